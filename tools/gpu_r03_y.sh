# LayerNorm PMC passes (after the one-round-per-wave change) + C = 640 / 1280 beyond the cache, then the final check
R=$PWD; O=$R/gpurun_out/r03_q2; mkdir -p $O/pmc
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc -o ${ctr}_layernorm --output-format csv -- python $R/tools/kbench.py layernorm --B 4 --n 147456 --iters 3 > $O/pmc_${ctr}_layernorm.log 2>&1
done
ls $O/pmc | head
cd $R
bash tools/gpu_r03_final.sh r03_z2
