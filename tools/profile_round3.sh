#!/bin/bash
# The measurements kept under profiles/ for round 3 (run on the GPU box from the repository root; outputs under gpurun_out/r3/):
# rocprofv3 kernel-trace summary of the bench command, PMC counters of the row-streaming convolutions (own passes, no trace domains:
# matrix-pipe utilisation, instruction mix, LDS conflicts at B = 32; HBM traffic of the bench's own conv4 launch at B = 256), phase times.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-inference --no-straggler-sim > $O/bench_kt.log 2>&1
python tools/kernel_stats.py $O/kt 60 > $O/kernel_stats.txt 2>&1
python tools/trace_overlap.py $O/kt > $O/trace_overlap.txt 2>&1
rm -rf $O/kt
for c in "MfmaUtil SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc -- python3 tools/conv_rows_pmc.py 32 > $O/pmc.log 2>&1
done
python tools/pmc_summary.py $O/pmc > $O/conv_pmc_summary.txt
rm -rf $O/pmc
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 tools/conv_rows_pmc.py 256 > $O/pmc.log 2>&1
    echo "== $c (KB per launch, B = 256)"; python tools/pmc_summary.py $O/pmc_$c | grep -A1 "^conv3x3_rows\|^conv3x3_wgrad_rows"
    rm -rf $O/pmc_$c
done > $O/conv_traffic_b256.txt
timeout 300 python tools/phase_times.py --steps 6 --segments > $O/phase_times.txt 2>&1
timeout 300 python tools/wgrad_rows_check.py 256 --no-check > $O/wgrad_rows_b256.txt 2>&1
timeout 300 python tools/conv_rows_check.py 256 --no-check > $O/conv_rows_b256.txt 2>&1
timeout 100 python tools/staff_emb_time.py > $O/staff_emb.txt 2>&1
