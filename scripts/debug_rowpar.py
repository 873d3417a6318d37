"""Diagnostic: row-parallel vs systolic factorisation vs oracle, distribution of iterate differences."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mpc_gpu
from oracle import oracle as orc
from helpers import oracle_P, oracle_guess, random_batch
for N, no, B, lanes in [(20, 3, 300, 0), (20, 3, 65, 64), (10, 5, 130, 0), (50, 10, 40, 0), (5, 3, 77, 0), (20, 3, 2000, 0)]:
    x0, goal, obst = random_batch(B, no, seed=71 + N)
    out = {}
    for rp in (1, 0):
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
            s.set_row_parallel(rp)
            if lanes: s.set_lanes_per_instance(lanes)
            s.reset_guess(x0); g = s.solve(x0, obst, goal); X, U = s.get_traj(B)
            s.shift(B); g2 = s.solve(x0, obst, goal); X2, U2 = s.get_traj(B)
            out[rp] = (g, X, U, g2, X2, U2)
    cfg = orc.config(N, no, 0.1 * N)
    P = oracle_P(orc, cfg, obst); Xg, Ug = oracle_guess(orc, cfg, x0)
    o = orc.rti_solve_batch(cfg, x0, P, goal, Xg, Ug)
    Xs, Us = np.stack([orc.shift(cfg, o["X"][b], o["U"][b])[0] for b in range(B)]), np.stack([orc.shift(cfg, o["X"][b], o["U"][b])[1] for b in range(B)])
    for a, b in ((0, 1), (3, 4)):
        ok = (out[1][a]["status"] == 0) & (out[0][a]["status"] == 0)
        d = np.abs(out[1][b] - out[0][b]).reshape(B, -1).max(1)[ok]
        print(f"N={N} no={no} B={B} lanes={lanes} solve{a//3}: ok {ok.mean():.3f} iters-equal {(out[1][a]['iters'][ok] == out[0][a]['iters'][ok]).mean():.3f} "
              f"|rp-sys| med {np.median(d):.1e} q95 {np.quantile(d, .95):.1e} max {d.max():.1e}")
    ok = (o["status"] == 0) & (out[1][0]["status"] == 0) & (out[0][0]["status"] == 0)
    for rp in (1, 0):
        d = np.abs(out[rp][1] - o["X"]).reshape(B, -1).max(1)[ok]
        print(f"    first solve vs oracle, rp={rp}: med {np.median(d):.1e} q95 {np.quantile(d, .95):.1e} max {d.max():.1e} iters-equal {(o['iters'][ok] == out[rp][0]['iters'][ok]).mean():.3f}")
