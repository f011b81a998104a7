#!/bin/bash
# Where the waves of the streaming 64-channel block kernel spend their cycles: two rocprofv3 counter passes over scripts/block_time.py.
set -eo pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/blocksq
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/a" -- python3 "$ROOT/scripts/block_time.py" 64 > "$OUT/a.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/b" -- python3 "$ROOT/scripts/block_time.py" 64 > "$OUT/b.log" 2>&1
cd "$ROOT"
python3 - <<PY
import csv, glob, collections
for d in ("a", "b"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in agg.items():
        if "nb1d_block64s" in k or "pair_kernel" in k:
            print(k)
            for c, v in sorted(dd.items()):
                print(f"   {c:28s} {sum(v)/len(v):16.0f}")
PY
