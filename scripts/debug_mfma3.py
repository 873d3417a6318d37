import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mpc_gpu
from mpc_gpu import _lib
from oracle import oracle as orc
from helpers import random_batch, oracle_P, oracle_guess
N, no, B = 50, 10, 40
x0, goal, obst = random_batch(B, no, seed=51 + N)
cfg = orc.config(N, no, 5.0, qp_tol=1e-8)
P = oracle_P(orc, cfg, obst); X, U = oracle_guess(orc, cfg, x0)
o1 = orc.rti_solve_batch(cfg, x0, P, goal, X, U)
Xs = o1["X"].copy(); Us = o1["U"].copy()
for b in range(B): Xs[b], Us[b] = orc.shift(cfg, Xs[b], Us[b])
o2 = orc.rti_solve_batch(cfg, x0, P, goal, Xs, Us)
cfg12 = orc.config(N, no, 5.0, qp_tol=1e-12)
o2t = orc.rti_solve_batch(cfg12, x0, P, goal, Xs, Us)
for mf in (1, 0):
    with mpc_gpu.BatchedMpc(N, no, 5.0, max_batch=B) as s:
        _lib.check(_lib.lib().mpc_set_matrix_cores(s._h, mf)); _lib.check(_lib.lib().mpc_set_lanes_per_instance(s._h, 64))
        s.set_warmstart(Xs, Us); g = s.solve(x0, P, goal); Xg, Ug = s.get_traj(B)
    d = np.abs(Xg - o2["X"]).reshape(B, -1).max(1); dt = np.abs(Xg - o2t["X"]).reshape(B, -1).max(1)
    w = np.argsort(-d)[:3]
    print("mfma" if mf else "valu", "vs oracle(1e-8): worst", d[w], "inst", w, "| vs oracle(1e-12):", dt[w], "status", g["status"][w], o2["status"][w], "iters", g["iters"][w], o2["iters"][w])
print("oracle 1e-8 vs 1e-12 worst", np.abs(o2["X"] - o2t["X"]).reshape(B, -1).max(1).max())
