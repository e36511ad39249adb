#!/bin/bash
# HBM traffic of the split-operand convolution kernels (rocprofv3 PMC, one counter per pass): FETCH_SIZE / WRITE_SIZE in KB per launch
# at B = $1 (default 32: a 40-channel activation tensor is 2 882 400 KB; 256: 23 059 200 KB).  usage (GPU box): bash tools/conv_traffic.sh [B]
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/tools/conv_split_pmc.py ${1:-32} 3 > /tmp/pmc.log 2>&1
  echo "== $c"
  python3 $R/tools/pmc_summary.py /tmp/pmc_$c | grep -A2 "^conv3x3_split"
done
