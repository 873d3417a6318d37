"""Debug helper (GPU box): find instances whose GPU status differs from the oracle's and trace where the iterate paths split."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mpc_gpu
from oracle import oracle as orc
from helpers import random_batch, oracle_P, oracle_guess

N, no, Tf, B = 20, 3, 2.0, 256
x0, goal, obst = random_batch(B, no, seed=100 + N)
cfg = orc.config(N, no, Tf, qp_tol=1e-8)
P = oracle_P(orc, cfg, obst); X, U = oracle_guess(orc, cfg, x0)
with mpc_gpu.BatchedMpc(N, no, Tf, max_batch=B) as s:
    s.set_warmstart(X, U); g = s.solve(x0, P, goal); Xg, Ug = s.get_traj(B)
o = orc.rti_solve_batch(cfg, x0, P, goal, X, U)
bad = np.nonzero(g["status"] != o["status"])[0]
print("mismatch", bad, g["status"][bad], o["status"][bad], g["iters"][bad], o["iters"][bad])
print("iters equal frac", (g["iters"] == o["iters"]).mean(), "max X diff (status equal)", np.abs(Xg - o["X"])[g["status"] == o["status"]].max())
for b in bad[:3]:
    print("instance", b, "x0", x0[b], "goal", goal[b])
    for cap in list(range(1, 26)) + [30, 40, 50]:
        c2 = orc.config(N, no, Tf, qp_tol=1e-8, qp_iter_max=cap)
        oo = orc.rti_solve(c2, x0[b], P[b], goal[b], X[b], U[b])
        with mpc_gpu.BatchedMpc(N, no, Tf, max_batch=1, qp_iter_max=cap) as s:
            s.set_warmstart(X[b:b+1], U[b:b+1]); gg = s.solve(x0[b:b+1], P[b:b+1], goal[b:b+1]); Xb, Ub = s.get_traj(1)
        print(f" cap {cap:2d} gpu st {gg['status'][0]} it {gg['iters'][0]:2d} | orc st {oo['status']} it {oo['iters']:2d} | dX {np.abs(Xb[0]-oo['X']).max():.3e} dU {np.abs(Ub[0]-oo['U']).max():.3e} kkt {oo['kkt']}")
