#!/bin/bash
# A/B of environment switches on the headline bench region, same box, interleaved repetitions.
#   bash tools/diag/bench_ab.sh "VTM_ATT16=0" "VTM_ATT16=1" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for env in "$@"; do
    env $env python bench.py --no-cpu-baseline --regimes none --workloads none --no-inflight-line --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$env'.ljust(44), 'ms/step', d['ms_per_step'], 'attention', r['attention_ms_per_step'], 'top', r['top_block']['avg_ms'], 'matching', d['matching']['matching_ms_per_step'], 'side', d['side_launches_ms_per_step'], 'gaps', d['unaccounted_ms_per_step'], 'sclk', d['box']['sclk_mhz']['mean'], 'W', d['box']['power_w']['mean'], 'folded', r['folded_launches'])"
  done
done
