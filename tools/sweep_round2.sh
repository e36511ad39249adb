#!/bin/bash
# Round-2 measurements kept under profiles/: batch sweep at the spec'd workload, cost of the data-parallel collectives at world = 1
# (RCCL through one rank), synchronised BatchNorm on top.
for b in 192 256 288 320; do
  A2S_BENCH_BATCH=$b python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 2>/dev/null | grep '^{"metric' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b:', d['value'], 'clips/s', d['ms_per_step'], 'ms/step, peak', d['config']['peak_mem_GiB'], 'GiB')" || echo "batch $b: failed (out of memory?)"
done
for cfg in "A2S_NONE=1" "A2S_FORCE_DIST=1" "A2S_FORCE_DIST=1 A2S_SYNC_BN=1"; do
  env $cfg python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 2>/dev/null | grep '^{"metric' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg:', d['value'], 'clips/s', d['ms_per_step'], 'ms/step', d.get('data_parallel'))"
done
