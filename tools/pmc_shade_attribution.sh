#!/bin/bash
# Run ON THE GPU BOX: where do the shade stage's fabric bytes go?  FETCH_SIZE / WRITE_SIZE of k_wf_shade (separate --pmc passes) for the
# product library and for PROBE builds that take one class of its traffic away (tools/build_variant.sh; their images are wrong, their
# timing and counters are the point):  notex = every texel footprint reads the texture's first texels; nopacket = every hit reads one of 64
# shading packets; nolp = the radiance / pending-term records are neither read nor written (same paths).  Queue entries and the
# beta / throughput records cannot be taken away without changing the paths: their share is what is left.
# usage: tools/pmc_shade_attribution.sh <variant | base> ...
export TMPDIR=/tmp
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --no-roofline"
mkdir -p gpurun_out/shade_attr
for n in "$@"; do
  if [ "$n" = base ]; then unset MIPT_LIBRARY; else export MIPT_LIBRARY=$PWD/variants/libmipt_$n.so; [ -f "$MIPT_LIBRARY" ] || { echo "no $MIPT_LIBRARY"; continue; }; fi
  OUT=gpurun_out/shade_attr/$n; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 bench.py $ARGS > $OUT/f.log 2>&1 || { echo "$n: FETCH pass failed"; tail -3 $OUT/f.log; continue; }
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 bench.py $ARGS > $OUT/w.log 2>&1 || { echo "$n: WRITE pass failed"; tail -3 $OUT/w.log; continue; }
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py $ARGS > $OUT/t.log 2>&1 || { echo "$n: trace pass failed"; tail -3 $OUT/t.log; continue; }
  python3 - "$OUT" "$n" 5 <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out, name, frames = sys.argv[1], sys.argv[2], float(sys.argv[3])
acc = defaultdict(lambda: defaultdict(float))
for sub in ("f", "w"):
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for k in ("k_wf_shade", "k_wf_traverse", "k_wf_trace", "k_wf_generate", "k_wf_resolve", "k_wf_shadow"):
                if k in r["Kernel_Name"]:
                    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]) / frames
ms = defaultdict(float)
for f in glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("k_wf_shade", "k_wf_traverse", "k_wf_trace", "k_wf_generate", "k_wf_resolve", "k_wf_shadow"):
            if k in r["Kernel_Name"]:
                ms[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 / frames
line = [l for l in open(out + "/t.log") if l.startswith("{")]
rays = json.loads(line[-1])["config"]["rays_per_frame"] if line else 0
for k in ("k_wf_shade", "k_wf_traverse", "k_wf_trace", "k_wf_generate", "k_wf_resolve"):
    fe, wr = acc[k].get("FETCH_SIZE", 0) * 1024, acc[k].get("WRITE_SIZE", 0) * 1024
    print("%-10s %-14s %7.3f ms/launch  FETCH x2 %7.2f GB  WRITE %7.2f GB  fabric %7.2f GB   (rays per 1-spp frame %.0f)" % (name, k, ms[k], 2 * fe / 1e9, wr / 1e9, (2 * fe + wr) / 1e9, rays))
PY
done
