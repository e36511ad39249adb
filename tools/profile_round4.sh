#!/bin/bash
# The measurements kept under profiles/ for round 4 (run on the GPU box from the repository root; outputs under gpurun_out/r4p/):
# rocprofv3 kernel-trace summary of the bench command; PMC counters in passes of their own (no trace domains): HBM traffic of the split
# attention kernels at B = 256 (forward and backward, the round-4 kernels: DPP reductions, streaming loads), matrix-pipe utilisation of the
# row-streaming convolutions and of the Linear GEMMs AT B = 256; phase times; the microbenchmarks the DESIGN tables quote.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4p; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-inference --no-straggler-sim > $O/bench_kt.log 2>&1
python tools/kernel_stats.py $O/kt 70 > $O/kernel_stats.txt 2>&1
python tools/trace_overlap.py $O/kt > $O/trace_overlap.txt 2>&1
rm -rf $O/kt
# attention traffic at the bench's batch (FETCH_SIZE is in KB; x2 on gfx950 for the wide reads: MI355X_MICROARCH.md, HBM / rocprofv3 section)
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_a; timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_a -- python3 tools/attn_sweep.py 256 > $O/pmc.log 2>&1
    echo "== $c (KB per launch, B = 256, tools/attn_sweep.py: single-row forward / backward split kernels + combine)"; python tools/pmc_summary.py $O/pmc_a | grep -A1 "^attn_"
    rm -rf $O/pmc_a
done > $O/attn_traffic_b256.txt 2>&1
# matrix-pipe utilisation at B = 256 (round 3 measured it at B = 32 only)
for c in "MfmaUtil SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    timeout 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc -- python3 tools/conv_rows_pmc.py 256 > $O/pmc.log 2>&1
done
python tools/pmc_summary.py $O/pmc > $O/conv_pmc_summary_b256.txt 2>&1
rm -rf $O/pmc
for c in "MfmaUtil SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES"; do
    timeout 400 rocprofv3 --pmc $c --output-format csv -d $O/pmcl -- python3 tools/linear_bench.py 256 > $O/pmc.log 2>&1
done
python tools/pmc_summary.py $O/pmcl > $O/linear_pmc_summary_b256.txt 2>&1
rm -rf $O/pmcl
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf $O/pmcl; timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmcl -- python3 tools/linear_bench.py 256 > $O/pmc.log 2>&1
    echo "== $c (per launch, B = 256, tools/linear_bench.py: generic tile and csrc/a2s_linear.hip kernels)"; python tools/pmc_summary.py $O/pmcl | grep -A2 "^lin_\|^gemm_f32_kernel<256"
    rm -rf $O/pmcl
done > $O/linear_traffic_b256.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 tools/conv_rows_pmc.py 256 > $O/pmc.log 2>&1
    echo "== $c (KB per launch, B = 256)"; python tools/pmc_summary.py $O/pmc_$c | grep -A1 "^conv3x3_rows\|^conv3x3_wgrad_rows"
    rm -rf $O/pmc_$c
done > $O/conv_traffic_b256.txt 2>&1
timeout 300 python tools/phase_times.py --steps 8 --segments > $O/phase_times.txt 2>&1
timeout 200 python tools/linear_bench.py 256 > $O/linear_bench_b256.txt 2>&1
timeout 200 python tools/gru_step_bench.py 256 1201 > $O/gru_step_bench.txt 2>&1
timeout 200 python tools/attn_mq_bench.py 248 > $O/attn_mq_bench.txt 2>&1
for a in "8 1" "8 5"; do timeout 120 python tools/dec_persist_bench.py $a 398 2>&1 | grep persist=; done > $O/dec_persist_bench.txt 2>&1
timeout 200 python tools/infer_bench.py 8 > $O/infer_b8.txt 2>&1
timeout 200 python tools/infer_bench.py 256 > $O/infer_b256.txt 2>&1
timeout 300 python tools/conv_rows_check.py 256 --no-check > $O/conv_rows_b256.txt 2>&1
timeout 300 python tools/wgrad_rows_check.py 256 --no-check > $O/wgrad_rows_b256.txt 2>&1
