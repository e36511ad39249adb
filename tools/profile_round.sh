#!/bin/bash
# The measurements kept under profiles/ for a round (run on the GPU box from the repository root; outputs under gpurun_out/):
# default bench line, B = 64 bench line, rocprofv3 kernel-trace summary of the bench, the conv kernels' launch table and the PMC
# counters of the split-operand convolution (counters in their own passes, no trace domains).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/conv_split_check.py 64 2>&1 | grep -v amdgpu.ids > gpurun_out/conv_split_check.txt
timeout 300 python tools/conv_bench.py 64 2>&1 | grep -v amdgpu.ids > gpurun_out/conv_bench_b64.txt
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout 600 python bench.py --batch 64 > gpurun_out/bench_b64.json 2> gpurun_out/bench_b64.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python bench.py --steps 3 --warmup 1 > gpurun_out/bench_kt.log 2>&1
python tools/kernel_stats.py gpurun_out/kt > gpurun_out/kernel_stats.txt 2>&1
rm -rf gpurun_out/kt
for c in "MfmaUtil SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_split -- python tools/conv_split_pmc.py 32 3 > gpurun_out/pmc_split.log 2>&1
done
python tools/pmc_summary.py gpurun_out/pmc_split > gpurun_out/pmc_split_summary.txt
rm -rf gpurun_out/pmc_split
timeout 300 python tools/phase_times.py > gpurun_out/phase_times.txt 2>&1
timeout 300 python tools/infer_bench.py 256 2>&1 | tail -1 > gpurun_out/infer_b256.json
