#!/bin/bash
# one GPU session: full GPU test suite, kernel A/B runs, bench, rocprof kernel trace, PMC traffic of the HBM-bound kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
for v in default prio flat; do
  if [ $v = default ]; then unset VIDTOME_HIP_LIB; else export VIDTOME_HIP_LIB=$R/vidtome_amd/lib/variants/$v/libvidtome_hip.so; fi
  echo "== attn variant $v"; python tools/kbench.py attn --Mq 34816 --M 52224 --d 40 --iters 6 2>&1 | tail -1
done
unset VIDTOME_HIP_LIB
python tools/kbench.py attn --Mq 8704 --M 13056 --d 80 --iters 6 2>&1 | tail -1
for sh in top_l1 top_l2 top_g mid_l1 mid_g; do python tools/kbench.py match --shape $sh --iters 6 2>&1 | grep match_filtered; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['top_block'],d['matching'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1; echo "prof rc=$?"
for what in gather unmerge layernorm; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc -o ${ctr}_$what --output-format csv -- python $R/tools/kbench.py $what --B 4 --n 147456 --iters 3 > $O/pmc_${ctr}_$what.log 2>&1
  done
  tail -1 $O/pmc_WRITE_SIZE_$what.log
done
ls $O $O/pmc | head -40
