import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mpc_gpu
from mpc_gpu import _lib
from helpers import random_batch
for N, no, B in ((20, 3, 300), (50, 10, 40), (10, 5, 64)):
    x0, goal, obst = random_batch(B, no, seed=51 + N)
    out = {}
    for mf in (1, 0):
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
            _lib.check(_lib.lib().mpc_set_matrix_cores(s._h, mf)); _lib.check(_lib.lib().mpc_set_lanes_per_instance(s._h, 64))
            s.reset_guess(x0); g = s.solve(x0, obst, goal); X, U = s.get_traj(B)
            s.shift(B); g2 = s.solve(x0, obst, goal); X2, U2 = s.get_traj(B)
            out[mf] = (g, X, g2, X2)
    for step, (gi, xi) in enumerate(((0, 1), (2, 3))):
        d = np.abs(out[1][xi] - out[0][xi]).reshape(B, -1).max(1)
        st1, st0 = out[1][gi]["status"], out[0][gi]["status"]
        it1, it0 = out[1][gi]["iters"], out[0][gi]["iters"]
        w = np.argsort(-d)[:4]
        print(f"N={N} step {step}: status mismatch {(st1!=st0).sum()} iters mismatch {(it1!=it0).sum()} worst d {d[w]} st {st1[w]} {st0[w]} it {it1[w]} {it0[w]}")
