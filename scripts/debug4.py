import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mpc_gpu
from mpc_gpu import _lib
from oracle import oracle as orc
from helpers import random_batch, oracle_P, oracle_guess

N, no, Tf, B = 20, 3, 2.0, 128
x0, goal, obst = random_batch(B, no, seed=7)
cfg = orc.config(N, no, Tf, qp_tol=1e-8)
P = oracle_P(orc, cfg, obst); X, U = oracle_guess(orc, cfg, x0)
found = None
with mpc_gpu.BatchedMpc(N, no, Tf, max_batch=B) as s:
    for k in range(10):
        s.set_warmstart(X, U); g = s.solve(x0, P, goal); Xg, Ug = s.get_traj(B)
        o = orc.rti_solve_batch(cfg, x0, P, goal, X, U)
        d = np.abs(Xg - o["X"]).reshape(B, -1).max(1)
        b = int(np.argmax(d))
        if d[b] > 1e-6 and found is None:
            found = (k, b, X[b].copy(), U[b].copy(), d[b])
        X, U = o["X"].copy(), o["U"].copy()
        for i in range(B): X[i], U[i] = orc.shift(cfg, X[i], U[i])
k, b, Xb, Ub, db = found
print("step", k, "inst", b, "dX", db)
np.savez("gpurun_out/case.npz", x0=x0[b], P=P[b], goal=goal[b], X=Xb, U=Ub)
tr_o = np.zeros((50, 4)); orc.lib().orc_set_trace.argtypes = [np.ctypeslib.ndpointer(dtype=np.float64), C.c_int]
orc.lib().orc_set_trace(tr_o, 50)
oo = orc.rti_solve(cfg, x0[b], P[b], goal[b], Xb, Ub)
orc.lib().orc_set_trace.argtypes = [C.c_void_p, C.c_int]; orc.lib().orc_set_trace(None, 0)
with mpc_gpu.BatchedMpc(N, no, Tf, max_batch=1) as s:
    _lib.check(_lib.lib().mpc_debug_trace(s._h, 1, 1, None))
    s.set_warmstart(Xb[None], Ub[None]); gg = s.solve(x0[b:b+1], P[b:b+1], goal[b:b+1]); Xg, Ug = s.get_traj(1)
    tr_g = np.zeros((50, 4)); _lib.check(_lib.lib().mpc_debug_trace(s._h, 1, 1, tr_g.ctypes.data))
print("iters", gg["iters"], oo["iters"], "dX", np.abs(Xg[0] - oo["X"]).max())
for it in range(max(int(gg["iters"][0]), oo["iters"])):
    print(it, "gpu mu %.6e sig %.6e al %.12f cmax %.3e | orc mu %.6e sig %.6e al %.12f cmax %.3e" % (*tr_g[it], *tr_o[it]))
