#!/bin/bash
# Round-6 measurements kept under profiles/ (run on the GPU box from the repository root; outputs under gpurun_out/r6p/).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6p; mkdir -p $O
rm -f gpurun_out/g4_parity_report.txt
# the -m gpu suite (writes gpurun_out/g4_parity_report.txt: the reference's optimizer step through two and THREE clip groups)
(timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v Warn | tail -6) > $O/gpu_tests.txt 2>&1
cp gpurun_out/g4_parity_report.txt $O/g4_parity_report.txt 2>/dev/null
# the bench at the driver's flags
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | grep '^{"metric' | tail -1 > $O/bench_driver_flags_b256.json
# kernel statistics + queue timelines of the bench command (4 steps under the profiler)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-inference --no-straggler-sim > $O/bench_kt.log 2>&1
python tools/kernel_stats.py $O/kt 70 > $O/kernel_stats.txt 2>&1
python tools/trace_overlap.py $O/kt > $O/trace_overlap.txt 2>&1
rm -rf $O/kt
timeout 300 python tools/phase_times.py --steps 8 --segments > $O/phase_times.txt 2>&1
# PMC passes (own runs, no trace domains) over the kernels the step really launches
KERN="^conv3x3_|^lin_|^dec_.*mid|^gemm_f32_kernel<256|^gemm_f32_kernel<128, 128, 2, 2, false, false, 2"
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
timeout 300 python tools/conv_rows_check.py 256 --no-check > $O/conv_rows_b256.txt 2>&1
A2S_AB_TAIL=0 timeout 400 python tools/ab_step.py --attr lib:dec_mid --pairs 6 2>&1 | tail -5 > $O/ab_dec_mid_notail.txt
timeout 400 python tools/ab_step.py --attr lib:dec_mid --pairs 8 2>&1 | tail -5 > $O/ab_dec_mid_tail.txt
timeout 300 python tools/alloc_by_stream.py 256 24 > $O/alloc_by_stream.txt 2>&1
