#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for shape in top_l1 top_g mid_g; do
  python tools/kbench.py match --shape $shape --data corr01 --plan range --iters 9 2>&1 | grep -v amdgpu.ids | tail -1
  python tools/kbench.py match --shape $shape --data corr002 --plan range --iters 9 2>&1 | grep -v amdgpu.ids | tail -1
  python tools/kbench.py match --shape $shape --data n01 --iters 9 2>&1 | grep -v amdgpu.ids | tail -1
done
python tools/kbench.py match --shape top_g --data corr01 --plan range --ordered --shuffle --iters 9 2>&1 | grep -v amdgpu.ids | tail -2
python tools/kbench.py match --shape top_l1 --data all --iters 5 2>&1 | grep -v amdgpu.ids | tail -9
