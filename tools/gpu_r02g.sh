#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "linear_rows or projection_paths" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
python tools/kbench.py linear --B 2 --n 65536 --M 52224 --Mq 34816 --C 320 --iters 6
python tools/kbench.py linear --B 2 --n 16384 --M 13056 --Mq 8704 --C 640 --iters 6
python tools/kbench.py linear --B 32 --n 256 --M 256 --C 1280 --iters 6
for proj in rows blas; do
  VIDTOME_PROJ=$proj python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$proj.json 2> $O/bench_$proj.err; echo "bench $proj rc=$?"
  python -c "
import json;d=json.load(open('$O/bench_$proj.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['top_block'],d['matching']['matching_ms_per_step'],d.get('projections'))"
done
