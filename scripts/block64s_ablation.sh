#!/bin/bash
# Times the streaming block kernel with parts of a step compiled out (libermi355_abl<mask>.so, built with -DNBS_ABLATE=<mask>:
# 1 matrix instructions, 2 epilogues, 4 row finishing, 8 x requests, 16 fragment reads), replayed from a graph.
for m in 0 1 2 4 8 16 3 19 31; do
  cp erfnet_pytorch_amd/libermi355_abl$m.so erfnet_pytorch_amd/libermi355.so
  echo -n "mask $m: "; timeout -k 10 60 python scripts/block64s_stamps.py 2>&1 | grep "launch median"
done
