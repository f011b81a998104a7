#!/bin/bash
# Regenerates profiles/$ROUND_* (ROUND defaults to r03) on the GPU box: the unprofiled bench line, the rocprofv3 kernel-trace summary of the same
# command, the HBM-traffic counters (separate FETCH_SIZE / WRITE_SIZE passes) and the matrix-core utilisation pass.
# Run from the repository root:  bash scripts/refresh_profiles.sh   (outputs land in gpurun_out/$ROUND/, summaries in profiles/).
# Through gpurun only gpurun_out/ comes back: afterwards  cp gpurun_out/$ROUND/${ROUND}_* profiles/
set -eo pipefail
ROOT=$(pwd)
ROUND=${ROUND:-r03}
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p "$OUT" "$ROOT/profiles"
timeout -k 10 300 python3 bench.py > "$OUT/bench_line.json"
cp "$OUT/bench_line.json" profiles/${ROUND}_bench_line.json
cd /tmp && export TMPDIR=/tmp
# Kernel trace and counters with ONE chain of launches (--infer-ways 1): a kernel's duration is then its own time and the launches
# have the shape the roofline probe times.  With the default two chains the launches of both overlap and a duration includes
# waiting for CUs; that trace is kept beside it as ${ROUND}_bench_kernel_stats_two_chains.csv.
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-fp32 --infer-ways 1 > "$OUT/trace.log" 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace2" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-fp32 > "$OUT/trace2.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-fp32 --infer-ways 1 --steps 3 --warmup 1 --train-steps 2 --train-warmup 3 > "$OUT/pmc_fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-fp32 --infer-ways 1 --steps 3 --warmup 1 --train-steps 2 --train-warmup 3 > "$OUT/pmc_write.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-fp32 --infer-ways 1 --steps 3 --warmup 1 --train-steps 2 --train-warmup 3 > "$OUT/pmc_mfma.log" 2>&1
cd "$ROOT"
cp "$(ls $OUT/trace/*/*kernel_stats.csv | tail -1)" profiles/${ROUND}_bench_kernel_stats.csv
cp "$(ls $OUT/trace2/*/*kernel_stats.csv | tail -1)" profiles/${ROUND}_bench_kernel_stats_two_chains.csv
python3 scripts/block_context.py "$OUT/trace" profiles/${ROUND}_block_kernel_by_context.csv > /dev/null
python3 scripts/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" profiles/${ROUND}_pmc_traffic
python3 scripts/pmc_mfma.py "$OUT/pmc_mfma" profiles/${ROUND}_pmc_mfma.csv > "$OUT/mfma.txt"
cp profiles/${ROUND}_* "$OUT/"
head -c 1500 profiles/${ROUND}_bench_line.json
