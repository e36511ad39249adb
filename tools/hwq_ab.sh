#!/bin/bash
for cfg in "A2S_ATTN_WGS=768" "A2S_ATTN_WGS=512" "A2S_ATTN_WGS=384" "A2S_ATTN_WGS=768" "A2S_ATTN_WGS=512" "A2S_ATTN_WGS=384"; do
  env $cfg python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 --full-tail ${TAIL:-0.01} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'])"
done
