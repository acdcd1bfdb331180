#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + PMC passes) into small text/JSON files for profiles/."""
import csv, glob, json, os, sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]


def find(pattern):
    return sorted(glob.glob(os.path.join(out_dir, pattern), recursive=True))


summary = {"tag": tag}
lines = []
# ---- kernel trace: per-kernel count / total / mean duration
kt = find("trace/**/*kernel_trace.csv")
per = defaultdict(list)
for f in kt:
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name") or row.get("kernel_name") or ""
        try:
            dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        except Exception:
            continue
        per[name].append((dur, row.get("VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"), row.get("Scratch_Size"), row.get("Grid_Size"), row.get("Workgroup_Size")))
tot = sum(sum(d[0] for d in v) for v in per.values()) or 1
lines.append("kernel-trace summary (%s): name, calls, total_ms, mean_us, pct, vgpr, sgpr, lds, scratch, grid, wg" % tag)
kernels = {}
for name, v in sorted(per.items(), key=lambda kv: -sum(d[0] for d in kv[1])):
    t = sum(d[0] for d in v)
    short = name[:90]
    lines.append("%-90s %6d %10.3f %10.2f %6.2f%%  %s %s %s %s %s %s" % (short, len(v), t / 1e6, t / len(v) / 1e3, 100.0 * t / tot, v[0][1], v[0][2], v[0][3], v[0][4], v[0][5], v[0][6]))
    kernels[name] = {"calls": len(v), "total_ms": t / 1e6, "mean_us": t / len(v) / 1e3}
summary["kernels"] = {k[:120]: v for k, v in kernels.items()}
# ---- PMC passes
def pmc(sub, counters):
    res = defaultdict(lambda: defaultdict(list))
    for f in find(sub + "/**/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name") or ""
            c = row.get("Counter_Name")
            if c in counters:
                res[name][c].append(float(row["Counter_Value"]))
    return res
mk = None
for name in per:
    if "pt_megakernel" in name:
        mk = name
for sub, cs in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE"]), ("pmc_l2", ["TCC_HIT_sum", "TCC_MISS_sum"])):
    res = pmc(sub, cs)
    for name, d in res.items():
        if "pt_megakernel" not in name:
            continue
        for c, vals in d.items():
            # skip warm-up launches: use the last `steps` launches
            lines.append("PMC %-14s %-40s launches %d mean %.4g (last 10 mean %.4g)" % (c, name[:40], len(vals), sum(vals) / len(vals), sum(vals[-10:]) / len(vals[-10:])))
            summary.setdefault("pmc", {})[c] = sum(vals[-10:]) / len(vals[-10:])
p = summary.get("pmc", {})
if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
    raw = (p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024.0
    corrected = (2.0 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024.0
    summary["hbm_bytes_per_launch_raw"] = raw
    # gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section); this kernel's loads are
    # scattered 16-B (dwordx4) gathers, an uncalibrated pattern: both figures are kept, the x2-corrected one is the upper bound.
    summary["hbm_bytes_per_launch"] = corrected
    lines.append("HBM bytes / pt_megakernel launch: raw (FETCH+WRITE)*1024 = %.4g, with gfx950 FETCH_SIZE x2 correction = %.4g" % (raw, corrected))
if "TCC_HIT_sum" in p and "TCC_MISS_sum" in p:
    hr = p["TCC_HIT_sum"] / max(p["TCC_HIT_sum"] + p["TCC_MISS_sum"], 1)
    summary["l2_hit_rate"] = hr
    lines.append("L2 hit rate pt_megakernel: %.4f" % hr)
for f in ("bench_trace.log",):
    fp = os.path.join(out_dir, f)
    if os.path.exists(fp):
        last = [l for l in open(fp).read().splitlines() if l.startswith("{")]
        if last:
            lines.append("bench line under the profiler: " + last[-1])
open(os.path.join(out_dir, "summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(summary, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
print("\n".join(lines))
