#!/bin/bash
# session g: dead waves read hot fragments (filter), refine 64-channel variant A/B
TAG=${1:-r04_g}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "match or planted or fuzz" > $O/tests_new.log 2>&1; echo "tests(new) rc=$?"; tail -2 $O/tests_new.log
for shape in top_l1 top_l2 top_g mid_g; do
  timeout 300 python tools/kbench.py match --shape $shape --data all --iters 7 2>&1 | grep -v amdgpu.ids >> $O/match_regimes.txt
done
cat $O/match_regimes.txt
timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
VIDTOME_HIP_LIB=$R/vidtome_amd/lib/variants/refine64/libvidtome_hip.so timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench_refine64.json 2>> $O/bench.err; echo "bench refine64 rc=$?"
timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench2.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/tools/match_mix.py > $O/prof.log 2>&1
python $R/profiles/summarize_rocpd.py $O/prof/k_results.db 2>&1 | head -12 | cut -c1-150; rm -f $O/prof/k_results.db
VIDTOME_HIP_LIB=$R/vidtome_amd/lib/variants/refine64/libvidtome_hip.so rocprofv3 --kernel-trace --stats -d $O/prof2 -o k -- python $R/tools/match_mix.py > $O/prof2.log 2>&1
python $R/profiles/summarize_rocpd.py $O/prof2/k_results.db 2>&1 | head -12 | cut -c1-150; rm -f $O/prof2/k_results.db
python - <<PY
import json
for n in ("bench","bench_refine64","bench2"):
    d=json.load(open("$O/%s.json"%n))
    print(n, d["value"], d["ms_per_step"], d["matching"]["matching_ms_per_step"], d["roofline"]["attention_ms_per_step"], d["box"]["sclk_mhz"]["mean"])
PY
