#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + PMC passes of the SAME bench command) into small text / JSON
files for profiles/.  One "launch" of the hot path = one pt_trace call = all path-tracing kernels of one frame
(wavefront: k_wf_generate + (k_wf_trace, k_wf_shade, k_wf_shadow) x (max_bounces + 1) + k_wf_resolve;
megakernel: one pt_megakernel), so per-launch figures are per-frame sums.

usage: summarize_profile.py <dir> <tag> <frames in the PMC runs> [<frames in the trace run>]"""
import csv, glob, json, os, sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]
pmc_frames = int(sys.argv[3]) if len(sys.argv) > 3 else 12
trace_frames = int(sys.argv[4]) if len(sys.argv) > 4 else pmc_frames


def find(pattern):
    return sorted(glob.glob(os.path.join(out_dir, pattern), recursive=True))


def is_pt(name):
    return "k_wf_" in name or "pt_megakernel" in name


summary = {"tag": tag}
lines = []
_fp = os.path.join(out_dir, "bench_trace.log")
if os.path.exists(_fp):
    _last = [l for l in open(_fp).read().splitlines() if l.startswith("{")]
    if _last:
        try:
            summary["bench"] = json.loads(_last[-1])
        except Exception:
            pass
per = defaultdict(list)
for f in find("trace/**/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        try:
            dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        except Exception:
            continue
        per[row.get("Kernel_Name", "")].append((dur, row.get("VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"), row.get("Scratch_Size")))
tot = sum(sum(d[0] for d in v) for v in per.values()) or 1
lines.append("kernel-trace summary (%s): name | calls | total ms | mean us | %% of GPU time | vgpr sgpr lds scratch" % tag)
kernels = {}
pt_total = 0
for name, v in sorted(per.items(), key=lambda kv: -sum(d[0] for d in kv[1])):
    t = sum(d[0] for d in v)
    short = name.split("(")[0].replace("void ", "")[:60]
    lines.append("%-60s %6d %10.3f %10.2f %6.2f%%  %s %s %s %s" % (short, len(v), t / 1e6, t / len(v) / 1e3, 100.0 * t / tot, v[0][1], v[0][2], v[0][3], v[0][4]))
    kernels[short] = {"calls": len(v), "total_ms": t / 1e6, "mean_us": t / len(v) / 1e3}
    if is_pt(name):
        pt_total += t
summary["kernels"] = kernels
summary["pt_kernel_ms_per_frame"] = pt_total / 1e6 / max(trace_frames, 1)
lines.append("path-tracing kernels: %.3f ms per frame (sum over %d frames / %d)" % (summary["pt_kernel_ms_per_frame"], trace_frames, trace_frames))


def pmc_total(sub, counters):
    res = defaultdict(float)
    for f in find(sub + "/**/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if is_pt(row.get("Kernel_Name", "")) and row.get("Counter_Name") in counters:
                res[row["Counter_Name"]] += float(row["Counter_Value"])
    return res


p = {}
for sub, cs in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE"]), ("pmc_l2", ["TCC_HIT_sum", "TCC_MISS_sum"])):
    for c, v in pmc_total(sub, cs).items():
        p[c] = v / max(pmc_frames, 1)
        lines.append("PMC %-14s per frame (sum over path-tracing kernels, %d frames): %.5g" % (c, pmc_frames, p[c]))
summary["pmc_per_frame"] = p
if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
    raw = (p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024.0
    corrected = (2.0 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024.0
    summary["hbm_bytes_per_launch_raw"] = raw
    # gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section).  These kernels
    # mostly issue scattered 16-B (dwordx4) gathers, an uncalibrated pattern: both figures are kept; the x2-corrected one is
    # the upper bound and is what bench.py reports as roofline.traffic.
    summary["hbm_bytes_per_launch"] = corrected
    lines.append("HBM bytes per frame: raw (FETCH+WRITE)*1024 = %.4g ; with the gfx950 FETCH_SIZE x2 correction = %.4g" % (raw, corrected))
if "TCC_HIT_sum" in p and "TCC_MISS_sum" in p:
    hr = p["TCC_HIT_sum"] / max(p["TCC_HIT_sum"] + p["TCC_MISS_sum"], 1)
    summary["l2_hit_rate"] = hr
    lines.append("L2 hit rate (path-tracing kernels): %.4f" % hr)

# ---- per-kernel PMC split (FETCH / WRITE / TCC hit and miss per stage, so every fraction of bench.py's roofline
# object can be recomputed from profiles/ alone).  Stage = the kernel family; per launch = per pt_trace = per frame of the run.
def stage_of(name):
    for key, st in (("k_wf_traverse", "traversal"), ("k_wf_trace", "traversal"), ("k_wf_shadow", "traversal"), ("k_wf_shade", "shade"), ("k_wf_generate", "generate+resolve"), ("k_wf_resolve", "generate+resolve"),
                    ("k_wf_tail", "tail"), ("pt_megakernel", "megakernel")):
        if key in name:
            return st
    return None


per_stage = defaultdict(lambda: defaultdict(float))
for sub in ("pmc_fetch", "pmc_write", "pmc_l2"):
    for f in find(sub + "/**/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            st = stage_of(row.get("Kernel_Name", ""))
            if st:
                per_stage[st][row["Counter_Name"]] += float(row["Counter_Value"]) / max(pmc_frames, 1)
stage_ms = defaultdict(float)
for name, v in per.items():
    st = stage_of(name)
    if st:
        stage_ms[st] += sum(d[0] for d in v) / 1e6 / max(trace_frames, 1)
stages = {}
lines.append("per-stage PMC, per launch (= per pt_trace): stage | ms | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes raw | HBM bytes (FETCH x2) | HBM GB/s (x2) | TCC hit rate")
tot_c = defaultdict(float)
for st, cs in sorted(per_stage.items()):
    fe, wr, hit, miss = cs.get("FETCH_SIZE", 0.0), cs.get("WRITE_SIZE", 0.0), cs.get("TCC_HIT_sum", 0.0), cs.get("TCC_MISS_sum", 0.0)
    for k, v in cs.items():
        tot_c[k] += v
    raw, corr = (fe + wr) * 1024.0, (2.0 * fe + wr) * 1024.0
    ms = stage_ms.get(st, 0.0)
    stages[st] = {"kernel_ms_per_launch": ms, "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "TCC_HIT_sum": hit, "TCC_MISS_sum": miss,
                  "hbm_bytes_per_launch_raw": raw, "hbm_bytes_per_launch": corr, "hbm_GBps": corr / max(ms, 1e-9) / 1e6,
                  "l2_hit_rate": hit / max(hit + miss, 1.0), "l2_requests_per_launch": hit + miss}
    lines.append("%-18s %9.3f %14.5g %14.5g %14.5g %14.5g %10.1f %8.4f" % (st, ms, fe, wr, raw, corr, corr / max(ms, 1e-9) / 1e6, hit / max(hit + miss, 1.0)))
if stages:
    fe, wr, hit, miss = tot_c.get("FETCH_SIZE", 0.0), tot_c.get("WRITE_SIZE", 0.0), tot_c.get("TCC_HIT_sum", 0.0), tot_c.get("TCC_MISS_sum", 0.0)
    ms = sum(stage_ms.values())
    stages["pipeline"] = {"kernel_ms_per_launch": ms, "hbm_bytes_per_launch_raw": (fe + wr) * 1024.0, "hbm_bytes_per_launch": (2 * fe + wr) * 1024.0,
                          "hbm_GBps": (2 * fe + wr) * 1024.0 / max(ms, 1e-9) / 1e6, "l2_hit_rate": hit / max(hit + miss, 1.0)}
    spp = (summary.get("bench") or {}).get("config", {}).get("samples_per_step")
    # the same per individual kernel (bench.py's roofline names the single dominant kernel, k_wf_traverse, beside its family)
    per_kernel = defaultdict(lambda: defaultdict(float))
    kname = lambda nm: next((k for k in ("k_wf_traverse", "k_wf_trace", "k_wf_shadow", "k_wf_shade", "k_wf_generate", "k_wf_resolve", "pt_megakernel") if k in nm), None)
    for sub in ("pmc_fetch", "pmc_write", "pmc_l2"):
        for f in find(sub + "/**/*counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                k = kname(row.get("Kernel_Name", ""))
                if k:
                    per_kernel[k][row["Counter_Name"]] += float(row["Counter_Value"]) / max(pmc_frames, 1)
    kern = {}
    for name, v in per.items():
        k = kname(name)
        if k:
            kern.setdefault(k, {"kernel_ms_per_launch": 0.0, "launches_per_pt_trace": 0.0})
            kern[k]["kernel_ms_per_launch"] += sum(d[0] for d in v) / 1e6 / max(trace_frames, 1)
            kern[k]["launches_per_pt_trace"] += len(v) / max(trace_frames, 1)
    lines.append("per-kernel PMC, per launch (= per pt_trace): kernel | launches | ms | HBM bytes (FETCH x2 + WRITE) | TCC hit rate")
    for k, cs in sorted(per_kernel.items()):
        fe, wr, hit, miss = cs.get("FETCH_SIZE", 0.0), cs.get("WRITE_SIZE", 0.0), cs.get("TCC_HIT_sum", 0.0), cs.get("TCC_MISS_sum", 0.0)
        e = kern.setdefault(k, {"kernel_ms_per_launch": 0.0, "launches_per_pt_trace": 0.0})
        e.update({"FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "TCC_HIT_sum": hit, "TCC_MISS_sum": miss, "hbm_bytes_per_launch_raw": (fe + wr) * 1024.0,
                  "hbm_bytes_per_launch": (2.0 * fe + wr) * 1024.0, "l2_hit_rate": hit / max(hit + miss, 1.0)})
        lines.append("%-16s %6.1f %9.3f %14.5g %8.4f" % (k, e["launches_per_pt_trace"], e["kernel_ms_per_launch"], e["hbm_bytes_per_launch"], e["l2_hit_rate"]))
    head = os.environ.get("MIPT_GIT_HEAD") or (open(".build_head").read().strip() if os.path.exists(".build_head") else None)
    pk = {"source": "tools/profile_gpu.sh %s: rocprofv3 --kernel-trace pass + separate --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum passes of the same bench.py command; values per launch (= per pt_trace), summed over the launches of each kernel family" % tag,
          "correction": "hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: gfx950 reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md); our scattered 16-B gathers are an uncalibrated pattern, so the raw figure is kept beside it",
          "samples_per_launch": spp, "git_head": head, "stages": stages, "kernels": kern}
    summary["pmc_per_kernel"] = pk

fp = os.path.join(out_dir, "bench_trace.log")
if os.path.exists(fp):
    last = [l for l in open(fp).read().splitlines() if l.startswith("{")]
    if last:
        lines.append("bench line under the profiler: " + last[-1])
        try:
            summary["bench"] = json.loads(last[-1])
        except Exception:
            pass
if "pmc_per_kernel" in summary:
    if summary["pmc_per_kernel"].get("samples_per_launch") is None:
        summary["pmc_per_kernel"]["samples_per_launch"] = (summary.get("bench") or {}).get("config", {}).get("samples_per_step")
    json.dump(summary["pmc_per_kernel"], open(os.path.join(out_dir, "pmc_per_kernel.json"), "w"), indent=1)
open(os.path.join(out_dir, "summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(summary, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
print("\n".join(lines))
