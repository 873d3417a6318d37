"""Large-sample GPU <-> oracle comparison (evidence behind DESIGN.md section 2; the pytest cases use small batches).
For each configuration: B random instances, first solve from set_initial_guess() and three further closed-loop steps in which
the oracle is fed the GPU's own shifted iterate (identical inputs per solve).  Writes gpurun_out/parity_sweep.json."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mpc_gpu
from oracle import oracle as orc
from helpers import oracle_P, qp_merit, random_batch

out = {}
# (lps, waves, lanes): lanes per horizon stage (1: rti_solve_kernel, 3 / 2: rti_split_kernel), wavefronts per SIMD of the split kernel,
# lanes per instance of the one-lane kernel (21: three instances per wavefront, compact LDS blocks; 0: automatic)
for N, no, B, lps, waves, lanes in [(20, 3, 20000, 1, 1, 32), (20, 3, 20000, 1, 1, 21), (20, 3, 20000, 3, 1, 0), (20, 3, 20000, 3, 2, 0),
                                    (20, 5, 20000, 1, 1, 32), (20, 5, 20000, 3, 1, 0), (10, 3, 20000, 1, 1, 16), (10, 3, 20000, 3, 1, 0),
                                    (30, 3, 8000, 2, 1, 0), (30, 3, 8000, 2, 2, 0), (20, 10, 8000, 3, 1, 0), (50, 10, 4000, 1, 1, 0), (5, 5, 20000, 1, 1, 0),
                                    (20, 4, 8000, 3, 1, 0), (40, 7, 3000, 1, 1, 0)]:      # obstacle counts between the instantiated row capacities
    x0, goal, obst = random_batch(B, no, seed=4242 + N + no)
    cfg = orc.config(N, no, 0.1 * N)
    res = []
    with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
        s.set_lanes_per_stage(lps); s.set_waves_per_simd(waves)
        if lanes:
            s.set_lanes_per_instance(lanes)
        kernel = s.kernel_name(B)
        s.reset_guess(x0)
        for step in range(4):
            Xin, Uin = s.get_traj(B)
            P = oracle_P(orc, cfg, obst)
            g = s.solve(x0, obst, goal); X, U = s.get_traj(B)
            t0 = time.perf_counter(); o = orc.rti_solve_batch(cfg, x0, P, goal, Xin, Uin); t_or = time.perf_counter() - t0
            ok = (o["status"] == 0) & (g["status"] == 0)
            d = np.abs(X - o["X"]).reshape(B, -1).max(1)[ok]
            du = (np.abs(g["u0"] - o["u0"]).max(1))[ok]
            # instances beyond 1e-6 are judged by the QP itself (helpers.qp_merit): feasible for the linearised dynamics and the boxes, objective
            # not above the oracle's (at most 40 per step are checked)
            idx = np.nonzero(ok)[0][d > 1e-6][:40]
            worse = 0
            for b in idx:
                fg, eqg, bg = qp_merit(orc, cfg, x0[b], P[b], goal[b], Xin[b], Uin[b], X[b], U[b])
                fo, _, _ = qp_merit(orc, cfg, x0[b], P[b], goal[b], Xin[b], Uin[b], o["X"][b], o["U"][b])
                worse += not (eqg <= 1e-7 and bg <= 1e-7 and fg <= fo + 1e-7 * max(1.0, abs(fo)))
            res.append(dict(step=step, kernel=kernel, outliers_checked_by_qp=int(len(idx)), outliers_worse_than_oracle=int(worse), status_equal=float((g["status"] == o["status"]).mean()), converged_both=float(ok.mean()),
                            status4_gpu=int((g["status"] == 4).sum()), status4_oracle=int((o["status"] == 4).sum()),
                            status2_gpu=int((g["status"] == 2).sum()), status2_oracle=int((o["status"] == 2).sum()),
                            iters_equal=float((g["iters"][ok] == o["iters"][ok]).mean()), mean_iters=float(g["iters"].mean()),
                            dX_median=float(np.median(d)), dX_q99=float(np.quantile(d, 0.99)), dX_q999=float(np.quantile(d, 0.999)), dX_max=float(d.max()),
                            du0_max=float(du.max()), frac_above_1e_6=float((d > 1e-6).mean()), oracle_seconds=t_or))
            print(N, no, B, kernel, res[-1], flush=True)
            # closed loop: plant + obstacles + shift, on the GPU's result
            x0 = s.plant_step(x0, g["u0"]); s.shift(B)
            obst = np.stack([np.array([orc.obstacle_step(cfg, ob, 0.1) for ob in obst[b]]) for b in range(B)]) if B <= 4000 else obst
    out[f"N{N}_obst{no}_B{B}_{kernel}"] = res
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity_sweep.json"), "w"), indent=1)
