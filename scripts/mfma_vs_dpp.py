"""Row g of the scope table (north_star: "MFMA for the condensed-QP GEMM", BASELINE configs[4] "condensed-QP MFMA stress"): the three
implementations of the Riccati factorisation sweep that exist in this library, timed on the same problems with one instance per
wavefront (the only mapping the matrix-core variant has) -- v_mfma_f64_16x16x4 chain | row-parallel 64-bit-DPP FMAs | one-lane systolic --
for N = 20 / 3 obstacles and N = 50 / 10 obstacles (C5's problem), 1024 instances (one wavefront per SIMD), closed loop.
Writes gpurun_out/mfma_vs_dpp_<tag>.json.   usage (GPU box): python scripts/mfma_vs_dpp.py [tag]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np, torch
import mpc_gpu, bench

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
out = {}
for N, no in ((20, 3), (50, 10)):
    rng = np.random.default_rng(1234); B = 1024
    x0 = np.zeros((B, 5)); x0[:, :2] = rng.uniform(-6, 6, (B, 2)); x0[:, 2] = rng.uniform(-np.pi, np.pi, B)
    goal = rng.uniform(-6, 6, (B, 2))
    obst = np.zeros((B, no, 4)); obst[:, :, :2] = rng.uniform(-4.4, 6, (B, no, 2)); obst[:, :, 2:] = rng.uniform(-2, 2, (B, no, 2))
    row = {}
    for name, mfma, rowpar in (("matrix_cores_v_mfma_f64_16x16x4", 1, 0), ("row_parallel_dpp", 0, 1), ("one_lane_systolic", 0, 0)):
        loop = bench.Loop(mpc_gpu, torch, N, no, x0, goal, obst, dev)
        loop.m.set_lanes_per_stage(1); loop.m.set_lanes_per_instance(64); loop.m.set_row_parallel(bool(rowpar)); loop.m.set_matrix_cores(bool(mfma))
        r = bench.measure(torch, None, loop, 1, None, 2, 1, dev)
        row[name] = dict(kernel=loop.m.kernel_name(B), us_per_control_step=r["kern_ms"] / max(1, r["launches"]) * 1e3, mean_ipm_iters=r["mean_iters"],
                         us_per_ipm_iteration=r["kern_ms"] / max(1, r["launches"]) * 1e3 / r["mean_iters"])
        loop.m.close(); del loop
        print(N, no, name, row[name], flush=True)
    out[f"N{N}_obst{no}"] = row
# what the condensed form would cost per interior-point iteration: dense Hessian of size nu*N, n_c = (4 + 8 + 2*n_obst)*N rows
for N, no in ((20, 3), (50, 10)):
    nuN = 2 * N; n_c = (4 + 8 + 2 * no) * N
    out[f"N{N}_obst{no}"]["flops_per_ipm_iteration"] = dict(
        condensed_dense=dict(hessian_update_2_nc_nuN2=2 * n_c * nuN ** 2, cholesky_nuN3_over_3=nuN ** 3 / 3.0),
        riccati=N * (7.0 / 3 * 125 + 4 * 25 * 2 + 2 * 5 * 4 + 8 / 3.0) + N * (2 * 49 + 6 * (4 + 8 + 2 * no)))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"mfma_vs_dpp_{tag}.json"), "w"), indent=1)
