#!/bin/bash
# A/B of the wide-tile d = 40 attention kernels (csrc/attention16.hip) against attention_kernel on the cfg-2 / cfg-3 launches.
#   gpurun -- 'bash tools/diag/att16_ab.sh > gpurun_out/att16_ab.txt 2>&1'
cd ${GRAFT_REPO_ROOT:-/root/repo}
K="python tools/kbench.py attn --d 40 --iters 7"
VARIANTS=("VTM_ATT16=0" "VTM_ATT16=1 VTM_ATT16_SKEW=0" "VTM_ATT16=1 VTM_ATT16_SKEW=1" "VTM_ATT16=1 VTM_ATT16_SKEW=1 VTM_ATT16_WAVES=4")
echo "== correctness (fp32 reference rows) =="
for env in "${VARIANTS[@]}"; do
  echo "-- $env"
  env $env $K --Mq 34816 --M 52224 --check 2>&1 | grep -v amdgpu.ids
  env $env $K --Mq 34816 --M 52224 --bounded 0.83 --check 2>&1 | grep -v amdgpu.ids
  env $env $K --Mq 8600 --M 9001 --check 2>&1 | grep -v amdgpu.ids
  env $env $K --Mq 300 --M 130 --check 2>&1 | grep -v amdgpu.ids
  env $env $K --Mq 700 --M 64 --check 2>&1 | grep -v amdgpu.ids
done
echo "== timing: cfg-2 top block =="
for rep in 1 2; do
for env in "${VARIANTS[@]}"; do
  echo "-- $env"
  env $env $K --Mq 34816 --M 52224 2>&1 | grep -v amdgpu.ids
  env $env $K --Mq 34816 --M 52224 --bounded 0.83 2>&1 | grep -v amdgpu.ids
  env $env $K --Mq 32768 --M 52224 2>&1 | grep -v amdgpu.ids
done; done
