#!/bin/bash
# Round-4 check session: full GPU suite, smoke, bench (+ CPU baseline), full-block bench, data regimes, kernel trace of the bench,
# PMC counters of the dominant kernels.    gpurun --timeout 2400 -- 'bash tools/gpu_r04_final.sh r04_z'
TAG=${1:-r04_z}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
grep -E "FAILED|ERROR" $O/tests.log | head
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --no-cpu-baseline --full-block > $O/bench_full_block.json 2>> $O/bench.err; echo "bench full rc=$?"
for d in n01 corr01 flat25 dup; do
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --data $d > $O/bench_$d.json 2>> $O/bench.err; echo "bench $d rc=$?"
done
for ex in neighbour ring allgather; do
  timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --exchange $ex > $O/bench_n1_$ex.json 2>> $O/bench.err; echo "bench exchange $ex rc=$?"
done
for shape in top_l1 top_l2 top_g mid_l1 mid_g; do
  timeout 300 python tools/kbench.py match --shape $shape --data all --iters 7 2>&1 | grep -v amdgpu.ids >> $O/match_regimes.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --steps 7 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1; echo "prof rc=$?"
grep '"metric"' $O/prof.log > $O/bench_profiled.json
python $R/profiles/summarize_rocpd.py $O/prof/k_results.db > $O/kernel_stats.txt 2>&1; rm -f $O/prof/k_results.db
SQ="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
rocprofv3 --kernel-trace --pmc $SQ -d $O/pmc -o sq_attn --output-format csv -- python $R/tools/kbench.py attn --Mq 34816 --M 52224 --d 40 --iters 3 > $O/pmc_sq_attn.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ -d $O/pmc -o sq_match --output-format csv -- python $R/tools/kbench.py match --shape top_l1 --data corr05 --iters 3 > $O/pmc_sq_match.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ -d $O/pmc -o sq_ff --output-format csv -- python $R/tools/kbench.py ff --n 131072 --C 320 --iters 3 > $O/pmc_sq_ff.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc -o ${ctr}_ff --output-format csv -- python $R/tools/kbench.py ff --n 131072 --C 320 --iters 3 > $O/pmc_${ctr}_ff.log 2>&1
  rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc -o ${ctr}_attn --output-format csv -- python $R/tools/kbench.py attn --Mq 34816 --M 52224 --d 40 --iters 3 > $O/pmc_${ctr}_attn.log 2>&1
  rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc -o ${ctr}_match --output-format csv -- python $R/tools/kbench.py match --shape top_l1 --data corr05 --iters 3 > $O/pmc_${ctr}_match.log 2>&1
done
for what in gather unmerge layernorm; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc -o ${ctr}_$what --output-format csv -- python $R/tools/kbench.py $what --B 4 --n 147456 --iters 3 > $O/pmc_${ctr}_$what.log 2>&1
  done
done
ls $O/pmc | wc -l
python $R/profiles/summarize_pmc.py $O $TAG > $O/pmc_summary.txt 2>&1; tail -5 $O/pmc_summary.txt
rm -rf $O/pmc/*/ 2>/dev/null
python - <<PY
import json
d=json.load(open("$O/bench.json"));print(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["roofline"]["frac_at_sustained_clock"],d["roofline"]["top_block"],d["matching"]["matching_ms_per_step"],d["cpu_baseline"]["seconds_per_step"],d["box"])
f=json.load(open("$O/bench_full_block.json"));print("full block",f["value"],f["ms_per_step"])
for n in ("n01","corr01","flat25","dup","n1_neighbour","n1_ring","n1_allgather"):
    try:
        e=json.load(open("$O/bench_%s.json"%n));print(n,e["value"],e["ms_per_step"],e["matching"]["matching_ms_per_step"],e["matching"]["counters"],e["config"]["exchange"])
    except Exception as ex: print(n,"failed",ex)
PY
du -sh $O
