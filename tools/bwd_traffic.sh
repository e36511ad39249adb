#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (KB per launch) of the big ConvStack-backward kernels at B = 32.  usage (GPU box): bash tools/bwd_traffic.sh
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcb_$c -- python3 $R/tools/bwd_kernels_pmc.py 32 > /tmp/pmcb.log 2>&1
  echo "== $c"; grep "tensor KB" /tmp/pmcb.log
  python3 $R/tools/pmc_summary.py /tmp/pmcb_$c | grep -A1 "^conv3x3_wgrad\|^conv3x3_split\|^bn_bwd" | grep -v "^--"
done
