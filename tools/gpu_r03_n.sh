mkdir -p gpurun_out/r03_n; O=gpurun_out/r03_n
for what in layernorm gather unmerge; do
  python tools/kbench.py $what --B 4 --n 147456 --iters 9 2>/dev/null | grep -v amdgpu >> $O/hbm_default.txt
  VIDTOME_HIP_LIB=$PWD/vidtome_amd/lib/variants/nt/libvidtome_hip.so python tools/kbench.py $what --B 4 --n 147456 --iters 9 2>/dev/null | grep -v amdgpu >> $O/hbm_nt.txt
done
for what in layernorm gather unmerge; do
  python tools/kbench.py $what --B 2 --n 65536 --iters 9 2>/dev/null | grep -v amdgpu >> $O/hbm_default.txt
  VIDTOME_HIP_LIB=$PWD/vidtome_amd/lib/variants/nt/libvidtome_hip.so python tools/kbench.py $what --B 2 --n 65536 --iters 9 2>/dev/null | grep -v amdgpu >> $O/hbm_nt.txt
done
echo default; cat $O/hbm_default.txt; echo nt; cat $O/hbm_nt.txt
