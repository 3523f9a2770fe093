#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + stats and the two HBM-traffic PMC passes of the default
# bench command, each in its own run as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# never combined with trace domains other than --kernel-trace).  Output: gpurun_out/$1/ ; summarise with
# tools/summarize_profiles.py and copy what is to be judged into profiles/.
set -o pipefail
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export OKGE_BENCH_OLP=0      # the S-OLP leg has its own line in the bench output; keep the per-kernel averages S-FB only
export OKGE_BENCH_CONFIGS=0  # ... and so do the other configurations (tools/collect_profiles_configs.sh profiles them)
CMD="python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $CMD > /dev/null 2> "$OUT/pmc_fetch.err" || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $CMD > /dev/null 2> "$OUT/pmc_write.err" || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS --output-format csv -d "$OUT/pmc_sq1" -- $CMD > /dev/null 2> "$OUT/pmc_sq1.err" || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- $CMD > /dev/null 2> "$OUT/pmc_sq2.err" || exit 1
cd "$R" && python3 tools/summarize_profiles.py "$OUT" > "$OUT/summary.txt"
# keep the summaries, drop the raw per-dispatch CSVs (gpurun copies back at most 64 MiB)
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq1" "$OUT/pmc_sq2"
tail -40 "$OUT/summary.txt"
