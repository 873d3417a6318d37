#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc_summary.json (run on the GPU box from the repo root):
#   scripts/profile_passes.sh <tag> [bench.py arguments]         (PROFILE_BATCH: the per-launch batch recorded in the summary; PROFILE_MFMA=1: one more pass with the
#                                                                  matrix-core busy counters -- the evidence runs of the opt-in MFMA sweep, MPC_MATRIX_CORES=1)
# One --kernel-trace --stats pass, then one PMC pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass; PMC
# passes carry no tracing options).  The program after `--` is python3 itself.
set -u
TAG=${1:-r02}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extra $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_THREAD_CYCLES_VALU" ${PROFILE_MFMA:+"SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"}; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i failed (see $OUT/pmc$i.log)"
done
python3 "$ROOT/scripts/summarize_profile.py" "$OUT" "$TAG" "${PROFILE_BATCH:-1024}"
