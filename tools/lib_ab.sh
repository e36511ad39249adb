#!/bin/bash
# old-vs-new library A/B on one box: alternate processes, same seeds
for i in 1 2; do
for lib in liba2s_hip_prev.so liba2s_hip.so; do
  A2S_LIB=$PWD/piano_a2s_amd/csrc/$lib python tools/ab_step.py --attr env:A2S_NOOP --pairs 4 2>&1 | tail -1 | sed "s/^/$lib  /"
done
done
