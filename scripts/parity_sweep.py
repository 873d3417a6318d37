"""Large-sample GPU <-> oracle comparison (evidence behind DESIGN.md section 2; the pytest cases use small batches).
For each configuration: B random instances, first solve from set_initial_guess() and three further closed-loop steps in which
the oracle is fed the GPU's own shifted iterate (identical inputs per solve).  Writes gpurun_out/parity_sweep.json."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mpc_gpu
from oracle import oracle as orc
from helpers import oracle_P, random_batch

out = {}
# lps: lanes per horizon stage -- 1: rti_solve_kernel (what a batch of this size gets), 3 / 2: rti_split_kernel forced onto the same batch
for N, no, B, lps in [(20, 3, 20000, 1), (20, 3, 20000, 3), (20, 5, 20000, 1), (20, 5, 20000, 3), (10, 3, 20000, 1), (10, 3, 20000, 3),
                      (30, 3, 8000, 2), (20, 10, 8000, 3), (50, 10, 4000, 1), (5, 5, 20000, 1)]:
    x0, goal, obst = random_batch(B, no, seed=4242 + N + no)
    cfg = orc.config(N, no, 0.1 * N)
    res = []
    with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
        s.set_lanes_per_stage(lps)
        s.reset_guess(x0)
        for step in range(4):
            Xin, Uin = s.get_traj(B)
            P = oracle_P(orc, cfg, obst)
            g = s.solve(x0, obst, goal); X, U = s.get_traj(B)
            t0 = time.perf_counter(); o = orc.rti_solve_batch(cfg, x0, P, goal, Xin, Uin); t_or = time.perf_counter() - t0
            ok = (o["status"] == 0) & (g["status"] == 0)
            d = np.abs(X - o["X"]).reshape(B, -1).max(1)[ok]
            du = (np.abs(g["u0"] - o["u0"]).max(1))[ok]
            res.append(dict(step=step, status_equal=float((g["status"] == o["status"]).mean()), converged_both=float(ok.mean()),
                            status4_gpu=int((g["status"] == 4).sum()), status4_oracle=int((o["status"] == 4).sum()),
                            status2_gpu=int((g["status"] == 2).sum()), status2_oracle=int((o["status"] == 2).sum()),
                            iters_equal=float((g["iters"][ok] == o["iters"][ok]).mean()), mean_iters=float(g["iters"].mean()),
                            dX_median=float(np.median(d)), dX_q99=float(np.quantile(d, 0.99)), dX_q999=float(np.quantile(d, 0.999)), dX_max=float(d.max()),
                            du0_max=float(du.max()), frac_above_1e_6=float((d > 1e-6).mean()), oracle_seconds=t_or))
            print(N, no, B, "lanes per stage", lps, res[-1], flush=True)
            # closed loop: plant + obstacles + shift, on the GPU's result
            x0 = s.plant_step(x0, g["u0"]); s.shift(B)
            obst = np.stack([np.array([orc.obstacle_step(cfg, ob, 0.1) for ob in obst[b]]) for b in range(B)]) if B <= 4000 else obst
    out[f"N{N}_obst{no}_B{B}_lanes_per_stage{lps}"] = res
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity_sweep.json"), "w"), indent=1)
