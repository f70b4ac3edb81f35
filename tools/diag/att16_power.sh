#!/bin/bash
# ms AND joules of the top-block attention launch: attention_kernel (32 queries per wave) vs the wide-tile kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
K="python tools/kbench.py attn --d 40 --iters 5 --Mq 34816 --M 52224 --power 3"
for rep in 1 2; do
for env in "VTM_ATT16=0" "VTM_ATT16_SKEW=0" "VTM_ATT16_SKEW=1"; do
  echo "-- $env"
  env $env $K 2>&1 | grep -v amdgpu.ids
  env $env $K --bounded 0.83 2>&1 | grep -v amdgpu.ids
done; done
echo "-- zero operands (no switching activity in the data)"
VTM_ATT16=0 $K --data zeros 2>&1 | grep -v amdgpu.ids
VTM_ATT16_SKEW=1 $K --data zeros 2>&1 | grep -v amdgpu.ids
