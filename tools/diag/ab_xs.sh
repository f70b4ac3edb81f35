cd $GRAFT_REPO_ROOT
for lib in "" "vidtome_amd/lib/variants/xs256/libvidtome_hip.so"; do
  echo "== lib=$lib"
  for shape in top_l1 top_g mid_g; do
    VIDTOME_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python tools/kbench.py match --shape $shape --data flat25 --iters 5 2>&1 | grep -E "flat25"
    VIDTOME_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python tools/kbench.py match --shape $shape --data zero --iters 5 2>&1 | grep -E "zero"
  done
done
VIDTOME_HIP_LIB=$GRAFT_REPO_ROOT/vidtome_amd/lib/variants/xs256/libvidtome_hip.so python -m pytest tests/test_gpu_parity.py -m gpu -q -k "escape or hard_cases or worst" 2>&1 | tail -2
