#!/bin/bash
# Round 6, first look at the mid-size step kernels in the step: A/B without the full-length tail, group timelines, kernel statistics of three steps.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6a; mkdir -p $O
A2S_AB_TAIL=0 timeout 400 python tools/ab_step.py --attr lib:dec_mid --pairs 6 2>&1 | tail -6 > $O/ab_mid_notail.txt
timeout 300 python tools/phase_times.py --steps 6 --segments > $O/phase_times.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/step_trace.py 256 3 0.01 > $O/kt.log 2>&1
python tools/kernel_stats.py $O/kt 60 > $O/kernel_stats.txt 2>&1
python tools/trace_overlap.py $O/kt > $O/trace_overlap.txt 2>&1
rm -rf $O/kt
