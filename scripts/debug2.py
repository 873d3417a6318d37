import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mpc_gpu
from oracle import oracle as orc
from helpers import random_batch, oracle_P, oracle_guess
N, no, Tf, B = 20, 3, 2.0, 16
x0, goal, obst = random_batch(B, no, seed=7)
cfg = orc.config(N, no, Tf, qp_tol=1e-8)
P = oracle_P(orc, cfg, obst); X, U = oracle_guess(orc, cfg, x0)
rng = np.random.default_rng(0)
Xr = rng.normal(size=X.shape); Ur = rng.normal(size=U.shape)
with mpc_gpu.BatchedMpc(N, no, Tf, max_batch=B) as s:
    s.set_warmstart(Xr, Ur); s.shift(B); Xs, Us = s.get_traj(B)
    ref = [orc.shift(cfg, Xr[b], Ur[b]) for b in range(B)]
    print("shift dX", max(np.abs(Xs[b]-ref[b][0]).max() for b in range(B)), "dU", max(np.abs(Us[b]-ref[b][1]).max() for b in range(B)))
    # d0 != 0 on a cold iterate
    x0p = x0.copy(); x0p[:, :2] += 0.3; x0p[:, 3] = 0.5
    s.set_warmstart(X, U); g = s.solve(x0p, P, goal); Xg, Ug = s.get_traj(B)
    o = orc.rti_solve_batch(cfg, x0p, P, goal, X, U)
    print("d0 case: status", g["status"], o["status"], "iters", g["iters"], o["iters"])
    print("  dX per inst", np.abs(Xg-o["X"]).reshape(B,-1).max(1))
