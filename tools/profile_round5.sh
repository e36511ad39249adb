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
