mkdir -p gpurun_out/r03_v; O=gpurun_out/r03_v
for v in "" lnall; do
  if [ -n "$v" ]; then export VIDTOME_HIP_LIB=$PWD/vidtome_amd/lib/variants/$v/libvidtome_hip.so; fi
  echo "== variant ${v:-default}" >> $O/ln.txt
  python tools/kbench.py layernorm --B 2 --n 16384 --C 640 --iters 9 2>/dev/null | grep -v amdgpu >> $O/ln.txt
  python tools/kbench.py layernorm --B 4 --n 73728 --C 640 --iters 9 2>/dev/null | grep -v amdgpu >> $O/ln.txt
  python tools/kbench.py layernorm --B 2 --n 4096 --C 1280 --iters 9 2>/dev/null | grep -v amdgpu >> $O/ln.txt
  python tools/kbench.py layernorm --B 4 --n 36864 --C 1280 --iters 9 2>/dev/null | grep -v amdgpu >> $O/ln.txt
done
cat $O/ln.txt
