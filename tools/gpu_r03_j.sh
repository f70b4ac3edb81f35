mkdir -p gpurun_out/r03_j; O=$PWD/gpurun_out/r03_j; R=$PWD
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/err.txt
VIDTOME_PROJ=blas python bench.py --no-cpu-baseline > $O/bench_proj_blas.json 2>> $O/err.txt
python bench.py --no-cpu-baseline --full-block > $O/bench_full.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --steps 7 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
cd $R
python profiles/summarize_rocpd.py $O/prof/k_results.db > $O/kernel_stats.txt 2>&1
rm -f $O/prof/k_results.db
head -40 $O/kernel_stats.txt | cut -c1-90,100-170
tail -3 $O/err.txt
python -c "
import json
for n in ('bench.json','bench_proj_blas.json','bench_full.json'):
    d=json.load(open('$O/'+n)); print(n, d['value'], d['ms_per_step'], d['roofline']['top_block'], d['roofline']['attention_ms_per_step'], d['matching']['matching_ms_per_step'], d['projections'])
"
