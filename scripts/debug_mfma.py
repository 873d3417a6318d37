import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mpc_gpu
from mpc_gpu import _lib
from helpers import random_batch
N, no, B = 20, 3, 300
x0, goal, obst = random_batch(B, no, seed=51 + N)
out = {}
for mf in (1, 0):
    with mpc_gpu.BatchedMpc(N, no, 2.0, max_batch=B) as s:
        _lib.check(_lib.lib().mpc_set_matrix_cores(s._h, mf)); _lib.check(_lib.lib().mpc_set_lanes_per_instance(s._h, 64))
        _lib.check(_lib.lib().mpc_debug_trace(s._h, 1, B, None))
        s.reset_guess(x0); g = s.solve(x0, obst, goal); X, U = s.get_traj(B)
        tr = np.zeros((B, 50, 4)); _lib.check(_lib.lib().mpc_debug_trace(s._h, 1, B, tr.ctypes.data))
        out[mf] = (g, X, tr)
d = np.abs(out[1][1] - out[0][1]).reshape(B, -1).max(1)
bad = np.argsort(-d)[:3]
print("worst", bad, d[bad], "iters", out[1][0]["iters"][bad], out[0][0]["iters"][bad], "status", out[1][0]["status"][bad], out[0][0]["status"][bad])
b = bad[0]
for it in range(int(max(out[1][0]["iters"][b], out[0][0]["iters"][b]))):
    print(it, "mfma mu %.9e sig %.6e al %.12f | valu mu %.9e sig %.6e al %.12f" % (out[1][2][b, it, 0], out[1][2][b, it, 1], out[1][2][b, it, 2], out[0][2][b, it, 0], out[0][2][b, it, 1], out[0][2][b, it, 2]))
