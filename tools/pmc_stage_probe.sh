#!/bin/bash
# Run ON THE GPU BOX: SQ-side counters of the stage kernels (what the waves wait on), two --pmc passes of a short bench run.
set -o pipefail
OUT=gpurun_out/pmc_stage; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/a -- python3 bench.py $ARGS > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -- python3 bench.py $ARGS > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/c -- python3 bench.py $ARGS > $OUT/c.log 2>&1 || { tail -5 $OUT/c.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/d -- python3 bench.py $ARGS > $OUT/d.log 2>&1 || { tail -5 $OUT/d.log; exit 1; }
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pt::", "")
        if "k_wf_" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    print(k)
    for name, v in sorted(c.items()):
        print("    %-28s %.4g   (/wave_cycles %.3f)" % (name, v, v / wc))
PY
