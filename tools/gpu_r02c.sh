#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
for sh in top_l1 top_l2 top_g mid_l1 mid_g; do python tools/kbench.py match --shape $sh --iters 6 2>&1 | grep match_filtered; done
for proj in rows blas; do
  VIDTOME_PROJ=$proj python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$proj.json 2> $O/bench_$proj.err; echo "bench $proj rc=$?"
  python -c "
import json;d=json.load(open('$O/bench_$proj.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['top_block'],d['matching']['matching_ms_per_step'],d.get('projections'))"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1; echo "prof rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc -o ${ctr}_attn --output-format csv -- python $R/tools/kbench.py attn --Mq 34816 --M 52224 --d 40 --iters 3 > $O/pmc_${ctr}_attn.log 2>&1
done
ls $O/pmc | head
