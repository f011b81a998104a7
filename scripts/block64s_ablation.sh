#!/bin/bash
# Times the streaming block kernel with parts of a step compiled out (libermi355_abl<mask>.so, built with -DNBS_ABLATE=<mask>:
# 1 matrix instructions, 2 epilogues, 4 row finishing, 8 x requests, 16 fragment reads), replayed from a graph.
# Build the variants first, on the CPU, from erfnet_pytorch_amd/csrc (after `make`):
#   for m in 0 1 2 4 8 16 3 19 31; do hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DNBS_ABLATE=$m \
#       -c nb1d_block64s.hip -o /tmp/nbs_$m.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libermi355_abl$m.so \
#       $(ls build/*.o | grep -v nb1d_block64s) /tmp/nbs_$m.o; done
# (the results differ from the product's: a removed part often takes dependent work with it -- without epilogues the compiler
# drops the matrix instructions too)
for m in 0 1 2 4 8 16 3 19 31; do
  cp erfnet_pytorch_amd/libermi355_abl$m.so erfnet_pytorch_amd/libermi355.so
  echo -n "mask $m: "; timeout -k 10 60 python scripts/block64s_stamps.py 2>&1 | grep "launch median"
done
