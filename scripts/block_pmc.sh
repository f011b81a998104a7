#!/bin/bash
# HBM traffic and time of the one-launch 64-channel block on the benchmark's map (16 x 128 x 256 x 64): two rocprofv3 counter
# passes (FETCH_SIZE, WRITE_SIZE) and a kernel trace over scripts/block_time.py; results under gpurun_out/blockpmc/.
set -eo pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/blockpmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/scripts/block_time.py" 64 > "$OUT/fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/scripts/block_time.py" 64 > "$OUT/write.log" 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/scripts/block_time.py" 64 > "$OUT/trace.log" 2>&1
cd "$ROOT"
python3 scripts/pmc_summary.py "$OUT/fetch" "$OUT/write" "$OUT/traffic"
python3 - <<PY
import json, csv, glob
d = json.load(open("$OUT/traffic.json"))
for k, v in d.items():
    if "nb1d" in k or "pair" in k:
        print(k[:80], round(v / 1e6, 1), "MB per launch")
for f in glob.glob("$OUT/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "nb1d" in r["Name"] or "pair" in r["Name"]:
            print(r["Name"][:80], r["Calls"], "calls, avg", round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
