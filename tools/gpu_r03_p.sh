mkdir -p gpurun_out/r03_p; O=gpurun_out/r03_p
for v in "" lnr3 lnr4; do
  if [ -n "$v" ]; then export VIDTOME_HIP_LIB=$PWD/vidtome_amd/lib/variants/$v/libvidtome_hip.so; fi
  echo "== variant ${v:-default}" >> $O/ln.txt
  python tools/kbench.py layernorm --B 4 --n 147456 --iters 9 2>/dev/null | grep -v amdgpu >> $O/ln.txt
  python tools/kbench.py layernorm --B 2 --n 65536 --iters 9 2>/dev/null | grep -v amdgpu >> $O/ln.txt
done
unset VIDTOME_HIP_LIB
python tools/kbench.py gather --B 4 --n 147456 --iters 9 2>/dev/null | grep -v amdgpu >> $O/ln.txt
python tools/kbench.py unmerge --B 4 --n 147456 --iters 9 2>/dev/null | grep -v amdgpu >> $O/ln.txt
cat $O/ln.txt
