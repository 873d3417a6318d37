#!/bin/bash
# VERDICT r02 item 5 / SURVEY row g: rocprofv3 evidence for the matrix-core Riccati sweeps (north_star: "MFMA ... evidenced with rocprof MFMA-utilisation counters").
# Three builds / settings of the SAME bench command (C2 workload, one instance per wavefront, 64 lanes per instance), each with one --kernel-trace --stats pass
# and one --pmc pass (no tracing options on the PMC pass; the program after `--` is python3 itself; the settings travel in the environment):
#   dpp     default row-parallel 64-bit-DPP sweep                                  (MPC_LANES_PER_STAGE=1 MPC_LANES_PER_INSTANCE=64)
#   mfma16  v_mfma_f64_16x16x4 chain, mpc_set_matrix_cores(1)                      (MPC_MATRIX_CORES=1)
#   mfma4   v_mfma_f64_4x4x4_4b blocks, the -DMPC_MFMA4 build (build/lib_mfma4.so)  (MPC_GPU_LIB=..., MPC_LANES_PER_STAGE=1 MPC_LANES_PER_INSTANCE=64)
# usage (GPU box, repo root): scripts/profile_mfma.sh    -> gpurun_out/r03_mfma_<variant>_{kernel_stats.csv,pmc_summary.json}
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extra --steps 2 --warmup 1"
PMC="SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_MFMA"
for V in dpp mfma16 mfma4; do
    unset MPC_MATRIX_CORES MPC_GPU_LIB; export MPC_LANES_PER_STAGE=1 MPC_LANES_PER_INSTANCE=64
    [ $V = mfma16 ] && export MPC_MATRIX_CORES=1
    [ $V = mfma4 ] && export MPC_GPU_LIB=$ROOT/build/lib_mfma4.so
    OUT=$ROOT/gpurun_out/prof_r03_mfma_$V; rm -rf "$OUT"; mkdir -p "$OUT"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
    rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc1" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc1.log" 2>&1 || echo "pmc pass failed ($OUT/pmc1.log)"
    python3 "$ROOT/scripts/summarize_profile.py" "$OUT" "r03_mfma_$V" 1024
    grep -h '"metric"' "$OUT/stats.log" | tail -1 > "$ROOT/gpurun_out/r03_mfma_${V}_bench.json"
done
