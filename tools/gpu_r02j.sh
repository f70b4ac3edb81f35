#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02j
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "match or golden or fuzz or planted" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for sh in top_l1 top_l2 top_g mid_l1 mid_g; do python tools/kbench.py match --shape $sh --iters 6 2>&1 | grep match_filtered; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['top_block'],d['matching']['matching_ms_per_step'],d.get('projections'))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1; echo "prof rc=$?"
