#!/usr/bin/env python3
"""Mean PMC counter values per kernel from a rocprofv3 --pmc CSV directory."""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in acc.items():
    if not any(k in name for k in ("k_wf_", "megakernel")):
        continue
    tot = {c: sum(v) for c, v in cs.items()}
    n = len(next(iter(cs.values())))
    line = "%-28s launches %4d " % (name[-28:], n)
    line += " ".join("%s=%.4g" % (c, t) for c, t in sorted(tot.items()))
    print(line)
    g = tot.get
    if g("SQ_WAVE_CYCLES"):
        wc = g("SQ_WAVE_CYCLES")
        print("    wait_any %.2f  wait_inst_any %.2f  active_inst_any %.2f   valu_lane_util %.2f   valu_insts/vmem_rd %.1f" % (
            g("SQ_WAIT_ANY", 0) / wc, g("SQ_WAIT_INST_ANY", 0) / wc, g("SQ_ACTIVE_INST_ANY", 0) / wc,
            g("SQ_THREAD_CYCLES_VALU", 0) / max(g("SQ_ACTIVE_INST_VALU", 1) * 64, 1), g("SQ_INSTS_VALU", 0) / max(g("SQ_INSTS_VMEM_RD", 1), 1)))
