#!/bin/bash
# several variant libraries against the product one on one box: alternate processes, fixed minibatches and coins (tools/step_time.py).
# usage: tools/lib_ab3.sh <rounds> <variant .so> [<variant .so> ...]
cd "$GRAFT_REPO_ROOT"
R=$1; shift
for i in $(seq 1 $R); do
  echo "product: $(timeout 300 python tools/step_time.py 256 12 2>&1 | tail -1)"
  for V in "$@"; do
    echo "$(basename $V): $(A2S_LIB=$PWD/$V timeout 300 python tools/step_time.py 256 12 2>&1 | tail -1)"
  done
done
