mkdir -p gpurun_out/r03_h; O=$PWD/gpurun_out/r03_h; R=$PWD
python bench.py --no-cpu-baseline --full-block > $O/bench_full.json 2> $O/err.txt
VIDTOME_FF=blas python bench.py --no-cpu-baseline --full-block > $O/bench_full_blas.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --full-block > $O/prof.log 2>&1
cd $R
python profiles/summarize_rocpd.py $O/prof/k_results.db > $O/kernel_stats_full.txt 2>&1
rm -f $O/prof/k_results.db
head -45 $O/kernel_stats_full.txt | cut -c1-90,100-170
python -c "
import json
for n in ('bench_full.json','bench_full_blas.json'):
    d=json.load(open('$O/'+n)); print(n, d['value'], d['ms_per_step'], json.dumps(d['full_block'])[180:1100])
"
