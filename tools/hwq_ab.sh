#!/bin/bash
# A/B switches of the fused step on the tail workload; each configuration twice (run-to-run spread).
for cfg in "A2S_PIPELINE_GROUPS=1" "A2S_PIPELINE_GROUPS=0" "A2S_PIPELINE_GROUPS=1" "A2S_PIPELINE_GROUPS=0" "A2S_CLIP_GROUPS=0"; do
  env $cfg python bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 2 --full-tail ${TAIL:-0.01} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'])"
done
