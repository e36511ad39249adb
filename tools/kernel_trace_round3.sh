cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-inference --no-straggler-sim > $O/bench_kt.log 2>&1
python tools/kernel_stats.py $O/kt 70 > $O/kernel_stats.txt 2>&1
python tools/trace_overlap.py $O/kt > $O/trace_overlap.txt 2>&1
rm -rf $O/kt
timeout 300 python tools/phase_times.py --steps 6 --segments > $O/phase_times.txt 2>&1
