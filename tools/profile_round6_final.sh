#!/bin/bash
# End-of-round check at HEAD: the -m gpu suite, the bench at the driver's flags, kernel statistics of the bench command.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6f; mkdir -p $O
rm -f gpurun_out/g4_parity_report.txt
(timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -4) > $O/gpu_tests.txt 2>&1
cp gpurun_out/g4_parity_report.txt $O/g4_parity_report.txt 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | grep '^{"metric' | tail -1 > $O/bench_driver_flags_b256.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-inference --no-straggler-sim > $O/bench_kt.log 2>&1
python tools/kernel_stats.py $O/kt 70 > $O/kernel_stats.txt 2>&1
rm -rf $O/kt
python __graft_entry__.py smoke > $O/smoke.txt 2>&1
