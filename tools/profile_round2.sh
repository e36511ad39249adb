#!/bin/bash
# The measurements kept under profiles/ for round 2 (run on the GPU box from the repository root; outputs under gpurun_out/r2/):
# default bench line (spec'd workload, CPU baseline, loss parity), rocprofv3 kernel-trace summary of the same bench command, the PMC
# counters of the dominant convolution kernel (own passes, no trace domains), phase times, two-term fp16 convolution accuracy / timing.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_kt.log 2>&1
python tools/kernel_stats.py $O/kt 60 > $O/kernel_stats.txt 2>&1
python tools/trace_overlap.py $O/kt > $O/trace_overlap.txt 2>&1
rm -rf $O/kt
for c in "MfmaUtil SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE WRITE_SIZE"; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc -- python tools/conv_split_pmc.py 32 3 > $O/pmc.log 2>&1
done
python tools/pmc_summary.py $O/pmc > $O/conv_pmc_summary.txt
rm -rf $O/pmc
timeout 300 python tools/phase_times.py > $O/phase_times.txt 2>&1
timeout 300 python tools/phase_times.py --full-tail 0.0 > $O/phase_times_notail.txt 2>&1
timeout 600 python tools/conv_f16x2_check.py 64 2>&1 | grep -v amdgpu.ids > $O/conv_f16x2_check.txt
timeout 300 python tools/infer_bench.py 256 2>&1 | tail -1 > $O/infer_b256.json
timeout 300 python tools/infer_bench.py 8 2>&1 | tail -1 > $O/infer_b8.json
