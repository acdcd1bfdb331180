#!/bin/bash
# Run ON THE GPU BOX: memory-side counters of the stage kernels (L1 / TLB / texture addresser / L2 / fabric), one --pmc pass each of
# a short bench run.  What it is for: which unit is busy or stalled while the waves wait.
OUT=gpurun_out/pmc_mem; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
pass() {
  name=$1; shift
  timeout -k 10 100 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $ARGS > $OUT/$name.log 2>&1
  rc=$?
  if grep -q "exceeds the capabilities" $OUT/$name.log; then echo "pass $name: too many counters of one block for one pass"; return; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name timed out: stopping"; exit 1; fi
  if [ $rc -ne 0 ]; then echo "pass $name failed (rc $rc):"; tail -3 $OUT/$name.log; fi
}
pass a TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass a2 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass b TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_SERIALIZATION_STALL_sum
pass b2 TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum
pass c TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
pass c2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_TCP_LATENCY_sum
pass d TCC_BUSY_sum TCC_CYCLE_sum TCC_REQ_sum TCC_TAG_STALL_sum
pass d2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum
pass f SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); launches = defaultdict(lambda: defaultdict(int))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pt::", "")
        if "k_wf_" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[k][r["Counter_Name"]] += 1
with open("$OUT/summary.txt", "w") as out:
    for k, c in sorted(acc.items()):
        print(k, file=out)
        for name, v in sorted(c.items()):
            print("    %-44s %.5g   (%d launches)" % (name, v, launches[k][name]), file=out)
print(open("$OUT/summary.txt").read())
PY
