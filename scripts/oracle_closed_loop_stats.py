#!/usr/bin/env python3
"""CPU ORACLE closed loops on random scenarios (the distributions of bench.py's C3 / C5): mean interior-point iterations and status counts per configuration
under a variant of the specification (--cfg overrides; --exp -> orc_set_investigation, the oracle's investigation switches).  Test infrastructure; no GPU.
    python scripts/oracle_closed_loop_stats.py --cfg polish_ratio=0.0,polish_tol=0.0"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default=""); ap.add_argument("--exp", type=int, default=0)
ap.add_argument("--B", type=int, default=512); ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--sizes", default="20:3,20:5,20:10,30:3,50:10,10:5")
ap.add_argument("--out", default=None)
a = ap.parse_args()
from oracle import oracle as orc
from helpers import random_batch
orc.build()
orc.set_investigation(a.exp)
over = {}
for kv in filter(None, a.cfg.split(",")):
    k, v = kv.split("="); over[k] = float(v) if ("." in v or "e" in v) else int(v)
res = {}
for sz in a.sizes.split(","):
    N, no = map(int, sz.split(":"))
    cfg = orc.config(N, no, 0.1 * N, **over)
    x0, goal, obst = random_batch(a.B, no, seed=1234)
    X = np.zeros((a.B, N + 1, 5)); U = np.zeros((a.B, N, 2))
    for b in range(a.B):
        X[b], U[b] = orc.initial_guess(cfg, x0[b])
    its = 0; st = {0: 0, 2: 0, 4: 0}; t0 = time.time(); first = 0
    for k in range(a.steps):
        P = np.stack([orc.predict_params(cfg, obst[b]) for b in range(a.B)])
        r = orc.rti_solve_batch(cfg, x0, P, goal, X, U, nthreads=8)
        its += int(r["iters"].sum())
        if k == 0: first = float(r["iters"].mean())
        for s in (0, 2, 4): st[s] += int((r["status"] == s).sum())
        X, U = r["X"], r["U"]
        for b in range(a.B):
            x0[b] = orc.dynamics(x0[b], r["u0"][b], 0.1)[0]
            for j in range(no): obst[b, j] = orc.obstacle_step(cfg, obst[b, j], 0.1)
            X[b], U[b] = orc.shift(cfg, X[b], U[b])
    res[sz] = dict(mean_iters=its / (a.B * a.steps), first_solve_iters=first, status=st, seconds=round(time.time() - t0, 1))
    print(sz, res[sz], flush=True)
if a.out:
    json.dump(dict(cfg=over, exp=a.exp, B=a.B, steps=a.steps, sizes=res), open(a.out, "w"), indent=1)
