#!/bin/bash
# Run ON THE GPU BOX: dynamic instruction mix of the stage kernels (wave-instructions per launch by class), three --pmc passes of a short
# bench run.   usage: tools/pmc_instmix_probe.sh [tag]
set -o pipefail
TAG=${1:-mix}; OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
P() { timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/p$N -- python3 bench.py $ARGS > $OUT/p$N.log 2>&1 || { tail -5 $OUT/p$N.log; exit 1; }; N=$((N+1)); }
N=0
P SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_WAVES
P SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH
P SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pt::", "")
        if "k_wf_" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
launches = 4.0     # 3 timed + 1 warm-up 8-spp launches
for k, c in sorted(acc.items()):
    v = c.get("SQ_INSTS_VALU", 1.0)
    print("%s: per 8-spp launch" % k)
    for name in sorted(c):
        print("    %-26s %12.4g   (%5.1f %% of VALU)" % (name, c[name] / launches, 100.0 * c[name] / v))
PY
