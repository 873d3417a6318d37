"""CPU-only: distance of the oracle's interior point from the EXACT QP solution (helpers.exact_qp; cached in /tmp) on B random first solves.
Round 4 ran it on prototype builds of the oracle with the floor of t / lam (1e-13 ... 1e-9), the fraction to the boundary, a dual-residual gate and a
no-termination-after-a-jump gate switched through the environment (only the floor mattered; the shipped oracle has it at min(1e-11, qp_tol / 10)
and none of the switches).  usage: tail_scan_cpu.py N n_obst B"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from oracle import oracle as orc
from helpers import oracle_P, oracle_guess, random_batch, exact_qp, step_vector
N, no, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x0, goal, obst = random_batch(B, no, seed=99)
cfg = orc.config(N, no, 0.1 * N)
P = oracle_P(orc, cfg, obst); X0, U0 = oracle_guess(orc, cfg, x0)
cache = f"/tmp/exact_{N}_{no}_{B}.npz"
r = orc.rti_solve_batch(cfg, x0, P, goal, X0, U0)
if os.path.exists(cache):
    z = np.load(cache); VEX, OK = z["VEX"], z["OK"]
else:
    VEX = np.zeros((B, 7 * N)); OK = np.zeros(B, bool)
    for b in range(B):
        if r["status"][b] != 0: continue
        q = orc.export_qp(cfg, x0[b], P[b], goal[b], X0[b], U0[b])
        VEX[b], OK[b], _ = exact_qp(q, step_vector(N, X0[b], U0[b], r["X"][b], r["U"][b]))
    np.savez(cache, VEX=VEX, OK=OK)
V = np.stack([step_vector(N, X0[b], U0[b], r["X"][b], r["U"][b]) for b in range(B)])
ok = OK & (r["status"] == 0)
if ok.sum() == 0: print("nothing converged"); sys.exit()
d = np.abs(V - VEX).max(1)[ok]
print(f"verified {ok.sum()} of {B}; status!=0: {(r['status']!=0).sum()}; mean iters {r['iters'].mean():.3f}; dist to exact: median {np.median(d):.1e} q99 {np.quantile(d,.99):.1e} max {d.max():.1e} frac>1e-6 {(d>1e-6).mean():.4f} frac>1e-7 {(d>1e-7).mean():.4f}")
