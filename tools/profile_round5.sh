#!/bin/bash
# Round-5 measurements kept under profiles/ (run on the GPU box from the repository root; outputs under gpurun_out/r5p/).
# PMC passes (own runs, no trace domains) over tools/step_once.py: the kernels the step REALLY launches at B = 256, among them the
# BatchNorm-backward-fused weight gradients conv3x3_wgrad_rows<..., true> (round 4 had measured the unfused instantiations).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5p; mkdir -p $O
KERN="^conv3x3_|^lin_|^bn_|^gemm_f32_kernel<256|^gemm_f32_kernel<128, 128, 2, 2, false, false, 2"
for c in "MfmaUtil SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    timeout 500 rocprofv3 --pmc $c --output-format csv -d $O/pmc -- python3 tools/step_once.py 256 2 > $O/pmc.log 2>&1
done
python tools/pmc_summary.py $O/pmc "$KERN" > $O/conv_pmc_summary_b256.txt 2>&1
rm -rf $O/pmc
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_t; timeout 500 rocprofv3 --pmc $c --output-format csv -d $O/pmc_t -- python3 tools/step_once.py 256 2 > $O/pmc.log 2>&1
    echo "== $c (KB per launch, B = 256; FETCH_SIZE x2 on gfx950 for the wide reads, MI355X_MICROARCH.md)"; python tools/pmc_summary.py $O/pmc_t "$KERN"
    rm -rf $O/pmc_t
done > $O/conv_traffic_b256.txt 2>&1
# kernel-trace summary of the bench command (5 steps under the profiler)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-inference --no-straggler-sim > $O/bench_kt.log 2>&1
python tools/kernel_stats.py $O/kt 70 > $O/kernel_stats.txt 2>&1
python tools/trace_overlap.py $O/kt > $O/trace_overlap.txt 2>&1
rm -rf $O/kt
timeout 300 python tools/phase_times.py --steps 8 --segments > $O/phase_times.txt 2>&1
timeout 200 python tools/linear_bench.py 256 > $O/linear_bench_b256.txt 2>&1
timeout 200 python tools/attn_mq_bench.py 248 > $O/attn_mq_bench.txt 2>&1
timeout 200 python tools/infer_bench.py 8 > $O/infer_b8.txt 2>&1
timeout 200 python tools/infer_bench.py 256 > $O/infer_b256.txt 2>&1
timeout 300 python tools/conv_rows_check.py 256 --no-check > $O/conv_rows_b256.txt 2>&1
timeout 300 python tools/wgrad_rows_check.py 256 --no-check > $O/wgrad_rows_b256.txt 2>&1
timeout 200 python tools/vqt_bench.py 64 5 > $O/vqt_bench.txt 2>&1
timeout 300 python tools/dp_settings.py 256 3 2>&1 | grep per_rank > $O/dp_settings.txt
