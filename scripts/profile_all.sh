#!/bin/bash
# Every profile a round commits under profiles/<tag>_*, in one go (GPU box, repo root; ~15 minutes):  scripts/profile_all.sh r06
#   <tag>_c2, _c3, _c5: the three BASELINE workloads on one stream;  _c4_share, _c5_share: one rank's share of the 8-GPU configurations;
#   _c5_share_mfma16: the opt-in v_mfma_f64_16x16x4 Riccati sweep (MPC_MATRIX_CORES=1) on C5's share -- row (g)'s negative result on C5's own workload.
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo "== $1 ($(date +%T))"; shift; "$@" 2>&1 | tail -4; }
run c2 env PROFILE_BATCH=1024 "$ROOT/scripts/profile_passes.sh" ${TAG}_c2
run c3 env PROFILE_BATCH=65536 "$ROOT/scripts/profile_passes.sh" ${TAG}_c3 --workload c3 --streams 1 --steps 2 --warmup 1
run c5 env PROFILE_BATCH=32768 "$ROOT/scripts/profile_passes.sh" ${TAG}_c5 --workload c5 --streams 1 --steps 2 --warmup 1
run c4_share env PROFILE_BATCH=32768 "$ROOT/scripts/profile_passes.sh" ${TAG}_c4_share --workload c4 --share 8 --streams 1 --steps 3 --warmup 1
run c5_share env PROFILE_BATCH=4096 "$ROOT/scripts/profile_passes.sh" ${TAG}_c5_share --workload c5 --share 8 --streams 1 --steps 3 --warmup 1
run c5_share_mfma16 env PROFILE_BATCH=4096 PROFILE_MFMA=1 MPC_MATRIX_CORES=1 "$ROOT/scripts/profile_passes.sh" ${TAG}_c5_share_mfma16 --workload c5 --share 8 --streams 1 --steps 3 --warmup 1
ls "$ROOT"/gpurun_out/${TAG}_*
