#!/bin/bash
# session k: the combine kernel of short split tails (two passes; accumulator groups shared out over blockIdx.y)
TAG=${1:-r04_k}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "attention" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for v in parts one; do
  if [ $v = one ]; then export VTM_DEBUG_COMBINE_PARTS=1; fi
  rocprofv3 --kernel-trace --stats -d $O/prof_$v -o k -- python $R/tools/kbench.py attn --Mq 8704 --M 13056 --d 80 --iters 20 > $O/prof_$v.log 2>&1
  python $R/profiles/summarize_rocpd.py $O/prof_$v/k_results.db 2>&1 | head -6 | cut -c1-150
  rm -rf $O/prof_$v
done
