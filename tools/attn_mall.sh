#!/bin/bash
# Does a clip block whose keys + encoder outputs fit the 256 MiB Infinity Cache decode faster per clip?  The same
# attention step launched 40 times on the SAME K / enc at B = 16 .. 256 (working set 59 .. 944 MB): microseconds per launch, GB/s, and the
# memory-side fetch counter per launch (rocprofv3 --pmc FETCH_SIZE, own pass; x2 on gfx950 for wide reads).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for B in 16 32 48 64 96 128 256; do
  python3 tools/attn_sweep.py $B 2>/dev/null | grep fwd
  rm -rf /tmp/pmc_attn; timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_attn -- python3 tools/attn_sweep.py $B > /tmp/pmc_attn.log 2>&1
  python3 tools/pmc_summary.py /tmp/pmc_attn | grep -A1 "^attn_fwd_split256$" | tail -1 | awk -v b=$B '{printf "    FETCH_SIZE x2 per launch: %.1f MB (working set %.1f MB)\n", $5 * 2 / 1024, b * 1201 * 768 * 4 / 1048576}'
done
