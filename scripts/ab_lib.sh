#!/bin/bash
# A/B of two builds of the library on one box: erfnet_pytorch_amd/libermi355_prev.so against libermi355.so, bench lines to gpurun_out/ab_{prev,new}_K.json
set -eo pipefail
L=erfnet_pytorch_amd
cp $L/libermi355.so /tmp/new.so
for k in 1 2; do
  cp $L/libermi355_prev.so $L/libermi355.so
  timeout -k 10 250 python bench.py --no-cpu-baseline --no-fp32 --no-roofline > gpurun_out/ab_prev_$k.json 2> gpurun_out/ab_prev_$k.err
  cp /tmp/new.so $L/libermi355.so
  timeout -k 10 250 python bench.py --no-cpu-baseline --no-fp32 --no-roofline > gpurun_out/ab_new_$k.json 2> gpurun_out/ab_new_$k.err
done
python3 - <<'PY'
import json
for t in ("prev_1","new_1","prev_2","new_2"):
    d=json.load(open(f"gpurun_out/ab_{t}.json"))
    print(t, "two chains", round(d["ms_per_step"],4), "one chain", round(d["inference_one_chain"]["ms_per_step"],4), "train", round(d["training"]["ms_per_step"],4))
PY
