#!/bin/bash
# A/B of clip groups / deferred streams on the fused step (tail workload); each configuration twice (run-to-run spread).
for cfg in "A2S_CLIP_GROUPS=1" "A2S_CLIP_GROUPS=0" "A2S_CLIP_GROUPS=1" "A2S_CLIP_GROUPS=0" "A2S_CLIP_GROUPS=1 A2S_DEFER_STREAM=1" "A2S_CLIP_GROUPS=0 A2S_DEFER_STREAM=1"; do
  env $cfg python bench.py --no-cpu-baseline --steps 4 --warmup 2 --full-tail ${TAIL:-0.01} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'])"
done
