#!/bin/bash
# device-planned (geometric tail) vs split-all query-bounded launches of the wide-tile d = 40 attention kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
K="python tools/kbench.py attn --d 40 --iters 7 --Mq 34816 --M 52224"
for frac in 0.83 0.5 0.97 0.25 0.05; do
  for env in "VTM_ATT16=0" "VTM_ATT_DEVPLAN=0" "VTM_ATT_DEVPLAN=1"; do
    env $env $K --bounded $frac --check 2>&1 | grep -v amdgpu.ids
  done
done
python tools/kbench.py attn --d 40 --iters 7 --Mq 8704 --M 13056 --bounded 0.8 --check 2>&1 | grep -v amdgpu.ids
VTM_ATT16=0 python tools/kbench.py attn --d 40 --iters 7 --Mq 8704 --M 13056 --bounded 0.8 --check 2>&1 | grep -v amdgpu.ids
