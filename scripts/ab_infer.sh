#!/bin/bash
# A/B of two builds of the library on one box, inference only: erfnet_pytorch_amd/libermi355_prev.so against libermi355.so.
set -eo pipefail
L=erfnet_pytorch_amd
cp $L/libermi355.so /tmp/new.so
for k in 1 2 3; do
  for which in prev new; do
    if [ $which = prev ]; then cp $L/libermi355_prev.so $L/libermi355.so; else cp /tmp/new.so $L/libermi355.so; fi
    timeout -k 10 120 python bench.py --no-fp32 --no-train --no-cpu-baseline --no-roofline > gpurun_out/ab.json 2> gpurun_out/ab.err
    python3 -c "
import json;d=json.load(open('gpurun_out/ab.json'));print('$which', 'two chains', round(d['ms_per_step'],4), 'one chain', round(d['inference_one_chain']['ms_per_step'],4))"
  done
done
cp /tmp/new.so $L/libermi355.so
