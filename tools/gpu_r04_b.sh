#!/bin/bash
# round 4, session b: partial-sum pruning in the filter (parity + KP sweep), fp16 chain fixtures, cfg-4 vs oracle, bench
TAG=${1:-r04_b}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "${KSEL:-match or planted or default_fp16_path or chain_golden or e2e or end_to_end}" > $O/tests_new.log 2>&1; echo "tests(new) rc=$?"; tail -4 $O/tests_new.log
grep -E "filtered ms / exact ms|worst block-output|FAILED|Error" $O/tests_new.log | head -20
timeout 300 python -m pytest tests/test_gpu_chunk_parallel.py -m gpu -q -x -k "cfg4" > $O/tests_cfg4.log 2>&1; echo "tests(cfg4) rc=$?"; tail -3 $O/tests_cfg4.log
for kp in 0 2 3 4; do
  for shape in top_l1 top_l2 top_g; do
    echo "== VTM_DEBUG_KP=$kp $shape" >> $O/kp_sweep.txt
    VTM_DEBUG_KP=$kp timeout 300 python tools/kbench.py match --shape $shape --data all --iters 7 2>&1 | grep -v amdgpu.ids | grep -v "^match\|^data" >> $O/kp_sweep.txt
  done
done
for kp in 0 4 5 6 7; do
  for shape in mid_l1 mid_g; do
    echo "== VTM_DEBUG_KP=$kp $shape" >> $O/kp_sweep.txt
    VTM_DEBUG_KP=$kp timeout 300 python tools/kbench.py match --shape $shape --data all --iters 7 2>&1 | grep -v amdgpu.ids | grep -v "^match\|^data" >> $O/kp_sweep.txt
  done
done
cat $O/kp_sweep.txt
VTM_DEBUG_KP=0 timeout 600 python bench.py --no-cpu-baseline > $O/bench_kp0.json 2> $O/bench.err; echo "bench kp0 rc=$?"
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2>> $O/bench.err; echo "bench rc=$?"
for d in n01 corr01; do
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --data $d > $O/bench_$d.json 2>> $O/bench.err; echo "bench $d rc=$?"
done
python - <<PY
import json
for n in ("bench_kp0","bench","bench_n01","bench_corr01"):
    try:
        d=json.load(open("$O/%s.json"%n))
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["matching"]["matching_ms_per_step"], d["matching"]["counters"], d["roofline"]["attention_ms_per_step"], d["event_pass_ms"], d["unaccounted_ms_per_step"], d["box"]["sclk_mhz"]["mean"], d["box"]["power_w"]["mean"])
    except Exception as e:
        print(n, "failed", e)
PY
