#!/bin/bash
TAG=${1:-r04_f}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python tools/diag/modes_bf16.py 2>&1 | grep -v amdgpu.ids | tail -30
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "${KSEL:-attention or live or compact or cfg5 or cfg3 or full_size or end_to_end or match_filtered_hard}" > $O/tests_new.log 2>&1; echo "tests(new) rc=$?"; tail -3 $O/tests_new.log
grep -E "FAILED|Error" $O/tests_new.log | head -20
VIDTOME_ATT_SPLIT_ALL=0 timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench_nosplit.json 2> $O/bench.err; echo "bench nosplit rc=$?"
timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench.json 2>> $O/bench.err; echo "bench rc=$?"
VIDTOME_ATT_SPLIT_ALL=0 timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench_nosplit2.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --steps 30 > $O/bench2.json 2>> $O/bench.err
python - <<PY
import json
for n in ("bench_nosplit","bench","bench_nosplit2","bench2"):
    d=json.load(open("$O/%s.json"%n))
    print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["matching"]["matching_ms_per_step"], d["roofline"]["attention_ms_per_step"], d["roofline"]["top_block"], d["box"]["sclk_mhz"]["mean"])
PY
