import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np, mpc_gpu
from oracle import oracle as orc
from helpers import oracle_P, oracle_guess, random_batch
N, no, B = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 3, 16
x0, goal, obst = random_batch(B, no, seed=100 + N)
cfg = orc.config(N, no, 0.1 * N, qp_tol=1e-8)
P = oracle_P(orc, cfg, obst); X0, U0 = oracle_guess(orc, cfg, x0)
o = orc.rti_solve_batch(cfg, x0, P, goal, X0, U0)
np.set_printoptions(linewidth=200, precision=2, suppress=False)
for on in (1, 0):
    with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
        s.set_block_riccati(bool(on))
        s.set_warmstart(X0, U0); g = s.solve(x0, P, goal); X, U = s.get_traj(B)
    dX = np.abs(X - o["X"]); dU = np.abs(U - o["U"])
    print("block2", on, s.__class__.__name__, "status", g["status"][:8], "iters", g["iters"][:8], o["iters"][:8])
    print(" max |dX| per component", dX.max(axis=(0, 1)), " per stage (max over comps, inst):")
    print(dX.max(axis=(0, 2)))
    print(" max |dU| per stage:"); print(dU.max(axis=(0, 2)))
