#!/bin/bash
# Re-collects everything under profiles/ that bench.py's line refers to, on the GPU box, in the order that
# matters: kernel-trace statistics of the default bench and of the enumeration + pivot legs alone, the three
# PMC passes (separate runs, no trace domains), the JSON that ties the traffic figures to the hash of the
# kernel sources, and last an unprofiled bench run whose line then carries `traffic`.
#   gpurun -- 'bash scripts/refresh_profiles.sh gpurun_out/r04x'      then copy (see profiles/README.md):
#   stats/*/*kernel_stats.csv -> profiles/r04_bench_kernel_stats.csv, stats_pivot/… -> …_enum_pivot_only_…,
#   last line of bench_line.log -> profiles/r04_bench_line.json, r04_pmc_traffic.json -> profiles/;
#   pmc_batched.txt -> profiles/r04_pmc_batched.txt
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/${1:-gpurun_out/profiles_refresh}
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" > "$O/bench_profiled.log" 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_pivot" -- python3 "$R/bench.py" --no-batched --no-cpu-baseline --no-large-shape > "$O/bench_pivot_profiled.log" 2>&1
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 "$R/scripts/pmc_traffic.py" > "$O/pmc_fetch.log" 2>&1
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 "$R/scripts/pmc_traffic.py" > "$O/pmc_write.log" 2>&1
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d "$O/pmc_enum" -- python3 "$R/scripts/pmc_enum.py" > "$O/pmc_enum.log" 2>&1
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$O/pmc_fp64" -- python3 "$R/scripts/pmc_enum.py" > "$O/pmc_fp64.log" 2>&1
# the batched kernel under the counters that say what binds it: two passes
timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/pmc_batched_a" -- python3 "$R/scripts/pmc_batched.py" > "$O/pmc_batched_a.log" 2>&1
timeout -k 10 120 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d "$O/pmc_batched_b" -- python3 "$R/scripts/pmc_batched.py" > "$O/pmc_batched_b.log" 2>&1
cd "$R"
{ python3 scripts/pmc_summary.py "$O/pmc_batched_a" k_batched; python3 scripts/pmc_summary.py "$O/pmc_batched_b" k_batched; } > "$O/pmc_batched.txt" 2>&1 || true
python3 scripts/pmc_enum_to_json.py "$O/pmc_enum" "$O/pmc_fp64" > "$O/enum_valu.json"
python3 scripts/pmc_to_json.py "$O/pmc_fetch" "$O/pmc_write" "$O/enum_valu.json" "$O/stats_pivot" > "$O/pmc_to_json.log" 2>&1
cp profiles/r04_pmc_traffic.json "$O/"
timeout -k 10 400 python3 bench.py > "$O/bench_line.log" 2>&1
find "$O" -name "*kernel_trace.csv" -delete
tail -c 300 "$O/bench_line.log"
