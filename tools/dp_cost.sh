#!/bin/bash
for cfg in "A2S_FORCE_DIST=1" "A2S_FORCE_DIST=1 A2S_NO_EXCHANGE=1" "A2S_FORCE_DIST=1 A2S_CLIP_GROUPS=0" "A2S_FORCE_DIST=0"; do
  env $cfg python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 2>/dev/null | grep '^{"metric' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg:', d['value'], 'clips/s', d['ms_per_step'], 'ms/step', d.get('data_parallel'))"
done
