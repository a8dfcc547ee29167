#!/bin/bash
# Same-box A/B of library builds on the x3 launches (random data, tools/bench_x3.py): two interleaved rounds.
#   bash tools/ab_x3.sh "LIB_A.so LIB_B.so ..." [bench_x3 flags]
LIBS=$1; shift
for r in 1 2; do
  for L in $LIBS; do
    echo "== $L"; UGN_LIB=$(pwd)/$L python tools/bench_x3.py --only-x3 "$@" 2>&1 | grep " x3 " 
  done
done
