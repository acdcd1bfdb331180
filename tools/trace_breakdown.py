#!/usr/bin/env python3
"""Per-kernel totals and the stage timeline of the last frame from a rocprofv3 kernel-trace CSV directory."""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
per = defaultdict(list); rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        per[name].append(dur); rows.append((int(row["Start_Timestamp"]), name, dur))
tot = sum(sum(v) for v in per.values())
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("%-60s calls %5d total %9.3f ms mean %9.2f us %5.1f%%" % (k[:60], len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, 100 * sum(v) / tot))
rows.sort()
wf = [(n.split("::")[-1][:14], round(dd / 1e3, 1)) for _, n, dd in rows if "k_wf_" in n or "megakernel" in n]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 29
print("last frame:", wf[-n:])
print("last frame total us:", sum(x[1] for x in wf[-n:]))
