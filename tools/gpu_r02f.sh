#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in default lin_NOA lin_NOW lin_NOSTORE; do
  if [ $v = default ]; then unset VIDTOME_HIP_LIB; else export VIDTOME_HIP_LIB=$R/vidtome_amd/lib/variants/$v/libvidtome_hip.so; fi
  echo "== linear variant $v"
  python tools/kbench.py linear --B 2 --n 65536 --M 52224 --Mq 34816 --C 320 --iters 6 | grep "k   rows\|v^T rows"
  python tools/kbench.py linear --B 2 --n 16384 --M 13056 --Mq 8704 --C 640 --iters 6 | grep "k   rows"
  python tools/kbench.py linear --B 32 --n 256 --M 256 --C 1280 --iters 6 | grep "k   rows"
done
