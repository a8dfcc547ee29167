#!/bin/bash
# A/B of builds of the library on the f16-pipe kernels (one GPU box, cold cache): tools/ab_mm.sh "<bench_mm flags>" lib1.so lib2.so ...
FLAGS=$1; shift
for lib in "$@"; do
  echo "== $lib"
  UGN_LIB=$(pwd)/$lib timeout -k 10 200 python tools/bench_mm.py $FLAGS --only-mm --cold 2>&1 | grep -v "BENCH_MM\|amdgpu.ids"
done
