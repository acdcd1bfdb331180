#!/bin/bash
# Run ON THE GPU BOX: kernel trace of single-sample launches (bench.py --spp 1): per-launch kernel durations in launch order and the
# gaps between them, for the reference-semantics figure (one 1-spp pt_trace).   usage: tools/trace_1spp.sh [tag] [spp]
set -o pipefail
SPP=${2:-1}; TAG=${1:-1spp}; OUT=gpurun_out/trace_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --spp $SPP --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pt::", "")))
rows.sort()
# the last complete frame: from the last k_wf_generate to the k_wf_resolve after it
gens = [i for i, r in enumerate(rows) if r[2].startswith("k_wf_generate")]
i0 = gens[-2]; i1 = next(i for i in range(i0, len(rows)) if rows[i][2].startswith("k_wf_resolve"))
fr = rows[i0:i1 + 1]
t0 = fr[0][0]; busy = 0; prev_end = None
print("%-28s %9s %9s %9s" % ("kernel", "start us", "dur us", "gap us"))
for s, e, n in fr:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%-28s %9.1f %9.1f %9.1f" % (n[:28], (s - t0) / 1e3, (e - s) / 1e3, gap)); busy += e - s; prev_end = e
print("frame: %.1f us wall, %.1f us in kernels, %d launches" % ((fr[-1][1] - t0) / 1e3, busy / 1e3, len(fr)))
PY
