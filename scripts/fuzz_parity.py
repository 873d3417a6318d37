"""Randomised configuration sweep: horizon, obstacle count, batch size and lane mapping drawn at random, GPU against the oracle on identical
inputs over two closed-loop steps (statuses equal; iterates to 1e-6 or judged by the QP, helpers.qp_merit).  A net for rarely taken dispatch
paths.  usage (GPU box): python scripts/fuzz_parity.py [seconds] [seed] [big]   -> gpurun_out/fuzz_parity.json   (big: batches of 1025 ... 20000 instances)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import mpc_gpu
from mpc_gpu import _lib
from oracle import oracle as orc
from helpers import adjudicate, oracle_P, oracle_guess, random_batch

# FUZZ_N="30,31,40,47,50,62": horizons to draw from; FUZZ_CFG="polish_step_frac=0.0,...": configuration overrides applied to BOTH sides (re-validation of a default)
N_CHOICES = [int(x) for x in os.environ["FUZZ_N"].split(",")] if os.environ.get("FUZZ_N") else [2, 3, 5, 9, 10, 14, 15, 17, 19, 20, 21, 25, 30, 31, 32, 40, 47, 50, 62]
CFG_OVER = {k: (float(v) if ("." in v or "e" in v) else int(v)) for k, v in (kv.split("=") for kv in filter(None, os.environ.get("FUZZ_CFG", "").split(",")))}
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
t0 = time.time(); log = []; fails = []; adjudicated = []
while time.time() - t0 < budget:
    N = int(rng.choice(N_CHOICES))
    no = int(rng.integers(1, 11))
    B = int(rng.choice([1, 2, 3, 7, 33, 64, 65, 100, 257]))
    if N > 31: B = min(B, 65)
    if BIG: B = int(rng.choice([1025, 3000, 4097, 8193, 12289, 20000])) if N <= 31 else int(os.environ.get("FUZZ_B_LONG", "1500"))      # batches deep enough for instance scheduling and every packing rule
    lps = int(rng.choice([0, 1, 2, 3])); lanes = int(rng.choice([0, 0, 16, 21, 32, 64])); waves = int(rng.choice([0, 1, 2]))
    soft = int(rng.random() > 0.15); bxt = int(rng.random() > 0.7)
    seed = int(rng.integers(1 << 30))
    x0, goal, obst = random_batch(B, no, seed=seed)
    cfg = orc.config(N, no, 0.1 * N, soft_h=soft, bx_terminal=bxt, **CFG_OVER)
    use_alpha = bool(soft and B <= 65 and rng.random() < 0.3)     # an explicit slack schedule (mpc_set_slack_schedule), some stages with zero weight
    alpha = rng.uniform(0.0, 3e4, (B, N + 1)) * (rng.random((B, N + 1)) > 0.15) if use_alpha else None
    rec = dict(N=N, n_obst=no, B=B, lps=lps, lanes=lanes, waves=waves, soft_h=soft, bx_terminal=bxt, seed=seed, explicit_slack_schedule=use_alpha)
    try:
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B, soft_h=soft, bx_terminal=bxt, **CFG_OVER) as s:
            L = _lib.lib()
            if lps: L.mpc_set_lanes_per_stage(s._h, lps)          # (an unsupported combination is rejected or ignored by the library: both are fine here)
            if lanes: L.mpc_set_lanes_per_instance(s._h, lanes)
            if waves: L.mpc_set_waves_per_simd(s._h, waves)
            rec["kernel"] = s.kernel_name(B)
            if use_alpha: s.set_slack_schedule(alpha)
            Xo, Uo = oracle_guess(orc, cfg, x0); s.reset_guess(x0)
            P = oracle_P(orc, cfg, obst)
            worst = 0.0
            for k in range(2):
                g = s.solve(x0, obst if k == 0 else P, goal); X, U = s.get_traj(B)
                if use_alpha:
                    rs = [orc.rti_solve(cfg, x0[b], P[b], goal[b], Xo[b], Uo[b], alpha=alpha[b]) for b in range(B)]
                    o = {kk: np.array([r[kk] for r in rs]) for kk in ("X", "U", "u0", "cost", "status", "iters")}
                else:
                    o = orc.rti_solve_batch(cfg, x0, P, goal, Xo, Uo)
                if not (g["status"] == o["status"]).all():
                    import ctypes as C
                    det = []
                    for b in np.nonzero(g["status"] != o["status"])[0]:          # how each side left the interior point
                        tr = np.zeros((64, 4)); orc.lib().orc_set_trace.argtypes = [C.c_void_p, C.c_int]; orc.lib().orc_set_trace(tr.ctypes.data, 64)
                        r1 = orc.rti_solve(cfg, x0[b], P[b], goal[b], Xo[b], Uo[b]); orc.lib().orc_set_trace(None, 0)
                        n = r1["iters"]
                        det.append(dict(inst=int(b), gpu_status=int(g["status"][b]), gpu_iters=int(g["iters"][b]), oracle_status=int(r1["status"]), oracle_iters=int(n),
                                        oracle_last_mu_sigma_alpha_cmax=tr[max(0, n - 3):n + 1].tolist()))
                    fails.append(dict(rec, step=k, why="status", mismatches=det)); break
                ok = o["status"] == 0
                d = np.abs(X - o["X"]).reshape(B, -1).max(1)
                for b in np.nonzero(ok & (d > 1e-6))[0]:
                    if use_alpha:          # (qp_merit assembles the QP with the built-in schedule: a plain bound instead)
                        if d[b] > 1e-4: fails.append(dict(rec, step=k, why="iterate (explicit slack schedule)", inst=int(b), d=float(d[b])))
                        continue
                    a = adjudicate(orc, cfg, x0[b], P[b], goal[b], Xo[b], Uo[b], X[b], U[b], o["X"][b], o["U"][b])      # against the exact QP solution
                    adjudicated.append(dict(N=N, n_obst=no, B=B, soft_h=soft, kernel=rec["kernel"], **a))
                    if not a["passed"]:
                        fails.append(dict(rec, step=k, why="adjudication", inst=int(b), d=float(d[b]), verdict=a))
                if ok.any(): worst = max(worst, float(d[ok].max()))
                Xo, Uo = o["X"].copy(), o["U"].copy()
                for b in range(B): Xo[b], Uo[b] = orc.shift(cfg, Xo[b], Uo[b])
                s.set_warmstart(Xo, Uo)
            rec["worst_dX"] = worst
    except Exception as e:                      # an API error is a finding too
        fails.append(dict(rec, why="exception", msg=str(e)[:300]))
    log.append(rec)
    if len(log) % (5 if BIG else 100) == 0: print(f"{len(log)} configurations, {len(fails)} findings, {time.time() - t0:.0f} s", flush=True)      # (a silent GPU job is taken for hung)
kernels = sorted({r.get("kernel", "?") for r in log})
ex = [a for a in adjudicated if a["kind"] == "exact"]
worst_cfg = max(log, key=lambda r: r.get("worst_dX", 0.0)) if log else None
out = dict(configurations=len(log), distinct_kernels=len(kernels), kernels=kernels, failures=fails, worst_dX=max((r.get("worst_dX", 0.0) for r in log), default=0.0), worst_dX_configuration=worst_cfg,
           adjudicated=len(adjudicated), adjudicated_exact=len(ex), adjudicated_by_merit=len(adjudicated) - len(ex), solves_compared=int(sum(2 * r["B"] for r in log)),
           worst_d_gpu_exact=max((a["d_gpu"] for a in ex), default=0.0), worst_d_oracle_exact=max((a["d_oracle"] for a in ex), default=0.0),
           gpu_farther_than_oracle=int(sum(a["d_gpu"] > a["d_oracle"] for a in ex)), adjudications=sorted(adjudicated, key=lambda a: -a["d_gpu_oracle"])[:60], seconds=time.time() - t0)
print(json.dumps({k: out[k] for k in ("configurations", "distinct_kernels", "failures", "worst_dX", "worst_dX_configuration", "adjudicated", "adjudicated_exact", "adjudicated_by_merit", "solves_compared",
                                      "worst_d_gpu_exact", "worst_d_oracle_exact", "gpu_farther_than_oracle")}, indent=1)[:6000])
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fuzz_parity.json"), "w"), indent=1)
sys.exit(1 if fails else 0)
