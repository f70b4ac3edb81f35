mkdir -p gpurun_out/r03_g; O=gpurun_out/r03_g
for cfg in "131072 320" "32768 640" "8192 1280" "2048 1280"; do set -- $cfg; python tools/kbench.py ff --n $1 --C $2 --iters 5 >> $O/ff.txt 2>&1; done
cat $O/ff.txt | grep -v amdgpu.ids
python bench.py --no-cpu-baseline --full-block > $O/bench_full.json 2> $O/err.txt
VIDTOME_FF=blas python bench.py --no-cpu-baseline --full-block > $O/bench_full_blas.json 2>> $O/err.txt
tail -3 $O/err.txt
python -c "
import json
for n in ('bench_full.json','bench_full_blas.json'):
    d=json.load(open('$O/'+n)); print(n, d['value'], d['ms_per_step'], json.dumps(d['full_block'])[:900])
"
