#!/bin/bash
for cfg in "A2S_ATTN_MAX_SPLIT=16" "A2S_ATTN_MAX_SPLIT=32" "A2S_ATTN_MAX_SPLIT=64" "A2S_ATTN_MAX_SPLIT=16" "A2S_ATTN_MAX_SPLIT=64"; do
  env $cfg python tools/ab_step.py --attr env:A2S_NOOP --pairs 4 2>&1 | tail -1 | sed "s/^/$cfg  /"
done
