#!/bin/bash
# session i: per-call effect of the guess seeds (kbench), and where the attention kernel's +0.5 ms in seeded runs comes from
# (VTM_DEBUG_SEED_DRY: every seeded launch / allocation happens, the filter does not see the seeds)
TAG=${1:-r04_i}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for shape in top_l1 top_l2 top_g mid_l1 mid_g; do
  timeout 200 python tools/kbench.py match --shape $shape --data corr05 --iters 9 --no-seed 2>&1 | grep -v amdgpu.ids >> $O/kbench_noseed.txt
  timeout 200 python tools/kbench.py match --shape $shape --data corr05 --iters 9 2>&1 | grep -v amdgpu.ids >> $O/kbench_seed.txt
done
for rep in 1 2; do
  VIDTOME_SEED=0 timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench_noseed$rep.json 2>> $O/bench.err
  VTM_DEBUG_SEED_DRY=1 timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench_dry$rep.json 2>> $O/bench.err
  timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench_seed$rep.json 2>> $O/bench.err
done
cat $O/kbench_noseed.txt $O/kbench_seed.txt
python - <<PY
import json
for n in ("noseed1","dry1","seed1","noseed2","dry2","seed2"):
    try:
        d=json.load(open("$O/bench_%s.json"%n))
        print(n, d["value"], d["ms_per_step"], d["matching"]["matching_ms_per_step"], d["roofline"]["attention_ms_per_step"], d["roofline"]["top_block"]["avg_ms"], d["box"]["sclk_mhz"]["mean"], d["box"]["power_w"]["mean"] if "power_w" in d["box"] else d["box"])
    except Exception as e: print(n, "failed", e)
PY
