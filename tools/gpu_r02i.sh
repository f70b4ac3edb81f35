#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
for ex in neighbour ring allgather; do
  VIDTOME_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --exchange $ex > $O/bench2_$ex.json 2> $O/bench2_$ex.err; echo "bench2 $ex rc=$?"
  tail -c 600 $O/bench2_$ex.json; echo; tail -3 $O/bench2_$ex.err
done
