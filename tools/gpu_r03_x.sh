mkdir -p gpurun_out/r03_x; O=$PWD/gpurun_out/r03_x; R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in default pw2 pw1 pw8 pp6 pp14 pp20; do
  if [ "$v" != default ]; then export VIDTOME_HIP_LIB=$R/vidtome_amd/lib/variants/$v/libvidtome_hip.so; else unset VIDTOME_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats -d $O/p_$v -o k -- python $R/tools/match_mix.py > $O/log_$v.txt 2>&1
  python $R/profiles/summarize_rocpd.py $O/p_$v/k_results.db > $O/stats_$v.txt 2>&1; rm -rf $O/p_$v
  echo "$v: $(grep prep_operand $O/stats_$v.txt | cut -c100-170)  | total $(head -1 $O/stats_$v.txt | cut -c1-120)"
done
