#!/bin/bash
# One GPU session producing everything profiles/ and DESIGN.md quote: full GPU test suite, smoke, bench (with the CPU
# baseline leg), rocprofv3 kernel trace of the bench, PMC counters of the dominant kernels.
#   gpurun --timeout 2400 -- 'bash tools/gpu_session.sh r06_final'
TAG=${1:-session}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for rep in 2 3; do python bench.py --no-cpu-baseline --regimes none --workloads none --no-inflight-line > $O/bench_run$rep.json 2>> $O/bench.err; done
python bench.py --no-cpu-baseline --regimes none --workloads none --no-inflight-line --full-block > $O/bench_full_block.json 2>> $O/bench.err; echo "bench full rc=$?"
for wlk in cfg3 cfg5; do python bench.py --no-cpu-baseline --regimes none --workloads none --no-inflight-line --workload $wlk > $O/bench_$wlk.json 2>> $O/bench.err; done
# the secondary options still run (N = 1 exchange modes in place, cfg-4's 8-frame chunk, local-only, rounds 1-2's regime)
for opt in "--exchange neighbour" "--exchange ring" "--exchange allgather" "--frames 8" "--local-only" "--same-chunk" "--data corr05"; do
  python bench.py --no-cpu-baseline --regimes none --workloads none --no-inflight-line --steps 8 $opt 2>> $O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('option $opt:', d['ms_per_step'], 'ms per step,', d['value'], 'steps/s')" ; done > $O/bench_options.txt 2>&1; cat $O/bench_options.txt
python - <<PY
import json
for n in ("bench","bench_run2","bench_run3","bench_full_block","bench_cfg3","bench_cfg5"):
    try:
        d=json.load(open("$O/%s.json"%n)); r=d["roofline"]
        print(n, d["value"], d["ms_per_step"], r["frac"], r["frac_at_sustained_clock"], r["top_block"], d["matching"]["matching_ms_per_step"], r["attention_ms_per_step"], d["side_launches_ms_per_step"], d["unaccounted_ms_per_step"], d["box"]["sclk_mhz"]["mean"], d.get("cpu_baseline",{}).get("seconds_per_step"))
        for k,v in d.get("regimes",{}).items(): print("   ",k,v["ms_per_step"],v["matching_ms"],v["attention_ms"],v["other_launches_ms"],v["pairs_per_row"],v["escaped"],v["pruned_block_fraction"],v["executed_mfma_fraction"],v["vs_corr05"])
    except Exception as e: print(n, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python $R/bench.py --steps 7 --warmup 1 --no-cpu-baseline --regimes none --workloads none --no-inflight-line > $O/prof.log 2>&1; echo "prof rc=$?"
grep '"metric"' $O/prof.log > $O/bench_profiled.json
python $R/profiles/summarize_rocpd.py $O/prof/k_results.db > $O/kernel_stats.txt 2>&1; rm -f $O/prof/k_results.db
rocprofv3 --kernel-trace --stats -d $O/prof3 -o k -- python $R/bench.py --workload cfg3 --steps 7 --warmup 1 --no-cpu-baseline --regimes none --workloads none --no-inflight-line > $O/prof3.log 2>&1; echo "prof cfg3 rc=$?"
python $R/profiles/summarize_rocpd.py $O/prof3/k_results.db > $O/kernel_stats_cfg3.txt 2>&1; rm -rf $O/prof3
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pmc -o sq_attn --output-format csv -- python $R/tools/kbench.py attn --Mq 34816 --M 52224 --d 40 --iters 3 > $O/pmc_sq_attn.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pmc -o sq_match --output-format csv -- python $R/tools/kbench.py match --shape top_l1 --data corr01 --iters 3 > $O/pmc_sq_match.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pmc -o sq_ff --output-format csv -- python $R/tools/kbench.py ff --n 131072 --C 320 --iters 3 > $O/pmc_sq_ff.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc -o ${ctr}_ff --output-format csv -- python $R/tools/kbench.py ff --n 131072 --C 320 --iters 3 > $O/pmc_${ctr}_ff.log 2>&1
  rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc -o ${ctr}_attn --output-format csv -- python $R/tools/kbench.py attn --Mq 34816 --M 52224 --d 40 --iters 3 > $O/pmc_${ctr}_attn.log 2>&1
  rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc -o ${ctr}_match --output-format csv -- python $R/tools/kbench.py match --shape top_l1 --data corr01 --iters 3 > $O/pmc_${ctr}_match.log 2>&1
done
for what in gather unmerge layernorm; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc -o ${ctr}_$what --output-format csv -- python $R/tools/kbench.py $what --B 4 --n 147456 --iters 3 > $O/pmc_${ctr}_$what.log 2>&1
  done
done
ls $O/pmc | wc -l
python $R/profiles/summarize_pmc.py $O $TAG > $O/pmc_summary.txt 2>&1; tail -5 $O/pmc_summary.txt
rm -rf $O/pmc/*/ 2>/dev/null; du -sh $O
