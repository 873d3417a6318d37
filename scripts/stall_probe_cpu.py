"""CPU-only: would ending a solve at the FIRST iteration >= 20 with mu > mu0 (status 4) change any result?  Today that test is applied only AT the iteration cap
(oracle/mpc_oracle.c MU_CAP_SETTLED).  For closed loops of random scenarios the oracle's per-iteration mu trace is recorded; counted: solves that have mu > mu0 at
some iteration >= 20 ("stalled"), how they end today (status and iteration count), and whether any of them recovers (ends 0, or 2 with mu <= mu0)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from oracle import oracle as orc
from helpers import random_batch

N, no, B, steps, cap = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 50
cfg = orc.config(N, no, 0.1 * N, qp_iter_max=cap)
x0, goal, obst = random_batch(B, no, seed=2718)
X = np.zeros((B, N + 1, 5)); U = np.zeros((B, N, 2))
for b in range(B):
    X[b], U[b] = orc.initial_guess(cfg, x0[b])
stat = dict(solves=0, stalled=0, stalled_end_status={}, recovered=0, iters_saved=0, total_iters=0, long_ge_25=0, long_ge_25_stalled=0)
for k in range(steps):
    for b in range(B):
        P = orc.predict_params(cfg, obst[b])
        r = orc.rti_solve_trace(cfg, x0[b], P, goal[b], X[b], U[b])
        tr = r["trace"]; it = r["iters"]
        stat["solves"] += 1; stat["total_iters"] += it
        mu = tr[:, 0]
        first = next((i for i in range(20, len(mu)) if mu[i] > cfg.mu0), None)
        if it >= 25:
            stat["long_ge_25"] += 1; stat["long_ge_25_stalled"] += first is not None
        if first is not None:
            stat["stalled"] += 1
            stat["stalled_end_status"][r["status"]] = stat["stalled_end_status"].get(r["status"], 0) + 1
            if r["status"] == 0 or (r["status"] == 2):
                stat["recovered"] += 1
            stat["iters_saved"] += it - first
        X[b], U[b] = r["X"], r["U"]
        u = r["u0"].copy()
        if r["status"] == 4:
            X[b], U[b] = orc.initial_guess(cfg, x0[b])
        x0[b] = orc.dynamics(x0[b], u, 0.1)[0]
        for j in range(no):
            obst[b, j] = orc.obstacle_step(cfg, obst[b, j], 0.1)
        X[b], U[b] = orc.shift(cfg, X[b], U[b])
    print(k, stat, flush=True)
print(json.dumps(stat))
