#!/bin/bash
# Round 6: occupancy caps re-measured with the mid-size step kernels in the bulk group (tools/step_time.py: means over 12 steps, 4 minibatches, fixed coins)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6a; mkdir -p $O
run() { echo "== $*"; env "$@" timeout 300 python tools/step_time.py 256 12 2>&1 | tail -1; }
{
run A2S_NOP=1
run A2S_ATTN_STRONG_CAP_SEGMENTS=3
run A2S_ATTN_STRONG_CAP_SEGMENTS=2
run A2S_ATTN_STRONG_CAP_SEGMENTS=1
run A2S_NOP=1
run A2S_ATTN_BULK_LDS_BWD=40960
run A2S_ATTN_BULK_LDS=98304
} > $O/cap_sweep.txt 2>&1
