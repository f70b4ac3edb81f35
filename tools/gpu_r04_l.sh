#!/bin/bash
# session l: pruning depth of the C = 320 levels inside the bench, with seeds (VTM_DEBUG_KP5: 2 = shipped, 3 = 60 % depth)
TAG=${1:-r04_l}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for rep in 1 2; do
  for kp in 2 3; do
    VTM_DEBUG_KP5=$kp timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench_kp${kp}_$rep.json 2>> $O/bench.err
  done
done
for d in n01 corr01 flat25; do
  for kp in 2 3; do
    VTM_DEBUG_KP5=$kp timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --data $d > $O/bench_${d}_kp$kp.json 2>> $O/bench.err
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["matching"]["matching_ms_per_step"], d["roofline"]["attention_ms_per_step"])
    except Exception as e: print(f, "failed", e)
PY
