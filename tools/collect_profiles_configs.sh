#!/bin/bash
# Run on the GPU box (via gpurun): the same five rocprofv3 passes as tools/collect_profiles.sh (kernel trace + stats;
# FETCH_SIZE; WRITE_SIZE; two SQ counter sets -- each in its own run, never combined with other trace domains) for the
# OTHER BASELINE configurations, i.e. `python3 tools/bench_configs.py <workload>`:
#     S-DM (configs[2])   S-OLP-shard (configs[3], one GPU's shard)   S-OLP-tok (configs[4])
# Output: gpurun_out/$TAG/<workload>/ ; tools/summarize_profiles.py writes pmc_summary.txt / pmc_traffic.json there.
set -o pipefail
TAG=${1:-cfgprof}
shift
WORKLOADS=${@:-S-DM S-OLP-shard S-OLP-tok}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for W in $WORKLOADS; do
  OUT=$R/gpurun_out/$TAG/$W
  mkdir -p "$OUT"
  CMD="python3 $R/tools/bench_configs.py $W"
  echo "== $W" >&2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || exit 1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $CMD > /dev/null 2> "$OUT/pmc_fetch.err" || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $CMD > /dev/null 2> "$OUT/pmc_write.err" || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS --output-format csv -d "$OUT/pmc_sq1" -- $CMD > /dev/null 2> "$OUT/pmc_sq1.err" || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- $CMD > /dev/null 2> "$OUT/pmc_sq2.err" || exit 1
  (cd "$R" && OKGE_PROFILE_COMMAND="python3 tools/bench_configs.py $W" python3 tools/summarize_profiles.py "$OUT" > "$OUT/summary.txt")
  # keep the summaries, drop the raw per-dispatch CSVs (gpurun copies back at most 64 MiB)
  find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
  rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq1" "$OUT/pmc_sq2"
done
