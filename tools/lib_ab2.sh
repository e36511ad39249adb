#!/bin/bash
# variant-vs-product library on one box: alternate processes, fixed minibatches and coins (tools/step_time.py).  usage: tools/lib_ab2.sh <variant .so> [rounds]
cd "$GRAFT_REPO_ROOT"
V=$PWD/$1; R=${2:-3}
for i in $(seq 1 $R); do
  echo "product: $(timeout 300 python tools/step_time.py 256 12 2>&1 | tail -1)"
  echo "variant: $(A2S_LIB=$V timeout 300 python tools/step_time.py 256 12 2>&1 | tail -1)"
done
