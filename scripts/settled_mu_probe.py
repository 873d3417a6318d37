#!/usr/bin/env python3
"""CPU probe (oracle only): is "from iteration 20 on, mu above mu0" decidable as a FAILURE the moment it first holds, instead of at the iteration cap only
?  Over three corpora -- the recorded-row replays (8 tables x 100 seeds, closed loops to 400 steps), the bench's C2 scenario (100 control
steps, plain loop) and random configurations as scripts/fuzz_parity.py draws them (hard and soft rows, N 2..62, 1..10 obstacles, two closed-loop steps) -- it
counts the solves at whose head mu stood above mu0 at some iteration >= 20 (oracle diagnostic orc_last_settled_it) by the status they ENDED with under the
at-the-cap rule.  The early rule is safe iff none of them ended 0 or 2.

    python scripts/settled_mu_probe.py [fuzz_configs] -> profiles/r06_settled_mu_probe.json
"""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]


def tally():
    return dict(solves=0, settled={0: 0, 2: 0, 4: 0}, status={0: 0, 2: 0, 4: 0}, first_it=[], examples=[])


def note(t, r, it, tag):
    t["solves"] += 1
    t["status"][r["status"]] += 1
    if it >= 0:
        t["settled"][r["status"]] += 1
        t["first_it"].append(it)
        if r["status"] != 4 and len(t["examples"]) < 20:
            t["examples"].append(dict(tag=tag, status=r["status"], iters=r["iters"], first_it=it))


def replay(job):
    stem, sp, seeds = job
    from oracle import oracle as orc
    from helpers import OracleLoop
    from mpc_gpu.world import reference_streams
    L_ = orc.lib()
    obst, noise = reference_streams(sp["scenario"], range(100), 5, 400)
    cfg = orc.config(sp["N_SOLV"], 5, float(sp["TF"]), qp_iter_max=sp["QP_ITER"])
    t = tally()
    for s in seeds:
        L = OracleLoop(orc, cfg, [-7.0, -7.0, np.pi / 4, 0, 0], [7.0, 7.0], obst[s], interp=bool(sp.get("interpolate_init")))
        for k in range(400):
            r = L.step(noise[k, s])
            if r is None:
                break
            note(t, r, L_.orc_last_settled_it(), f"{stem}:{s}:{k}")
    return t


def fuzz(job):
    seed, count = job
    from oracle import oracle as orc
    from helpers import random_batch
    L_ = orc.lib()
    rng = np.random.default_rng(seed)
    t = tally()
    for _ in range(count):
        N = int(rng.choice([2, 3, 5, 9, 10, 14, 15, 17, 19, 20, 21, 25, 30, 31, 32, 40, 47, 50, 62]))
        no = int(rng.integers(1, 11))
        B = 24 if N <= 31 else 8
        soft = int(rng.random() > 0.15); bxt = int(rng.random() > 0.7)
        sd = int(rng.integers(1 << 30))
        x0, goal, obst = random_batch(B, no, seed=sd)
        cfg = orc.config(N, no, 0.1 * N, soft_h=soft, bx_terminal=bxt)
        for b in range(B):
            X, U = orc.initial_guess(cfg, x0[b])
            P = orc.predict_params(cfg, obst[b])
            for k in range(2):
                r = orc.rti_solve(cfg, x0[b], P, goal[b], X, U)
                note(t, r, L_.orc_last_settled_it(), f"fuzz:{N}:{no}:{soft}:{bxt}:{sd}:{b}:{k}")
                X, U = orc.shift(cfg, r["X"], r["U"])
    return t


def c2(_):
    from oracle import oracle as orc
    from helpers import OracleLoop
    L_ = orc.lib()
    gold = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))      # bench.py::make_workload("c2")
    x0, goal, obst = [-6.0, -6.0, np.pi / 4, 0.0, 0.0], [6.0, 6.0], gold["gen_RANDOM_3"][0]
    t = tally()
    cfg = orc.config(20, 3, 2.0)
    L = OracleLoop(orc, cfg, x0, goal, obst, reset_on_fail=False, alias=False)
    L.flags = 0
    for k in range(100):
        L.flags &= ~1
        r = L.step(None)
        note(t, r, L_.orc_last_settled_it(), f"c2:{k}")
    return t


def merge(parts):
    out = tally()
    for p in parts:
        out["solves"] += p["solves"]
        for k in (0, 2, 4):
            out["settled"][k] += p["settled"][k]; out["status"][k] += p["status"][k]
        out["first_it"] += p["first_it"]; out["examples"] += p["examples"]
    fi = np.array(out.pop("first_it")) if out["first_it"] else np.zeros(0, int)
    out["first_it_hist"] = {int(k): int(v) for k, v in zip(*np.unique(fi, return_counts=True))}
    out["examples"] = out["examples"][:20]
    return out


def main():
    from oracle import oracle as orc
    orc.build()
    orc.lib().orc_last_settled_it
    nfuzz = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
    T = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")))["tables"]
    jobs = [(stem, t["spec"], list(range(lo, lo + 10))) for stem, t in T.items() for lo in range(0, 100, 10)]
    with mp.Pool(8) as pool:
        rep = merge(pool.map(replay, jobs, chunksize=1))
        print("replay", json.dumps(rep), flush=True)
        fz = merge(pool.map(fuzz, [(9000 + i, nfuzz // 64) for i in range(64)], chunksize=1))
        print("fuzz", json.dumps(fz), flush=True)
        cc = merge(pool.map(c2, [0]))
        print("c2", json.dumps(cc), flush=True)
    json.dump(dict(recorded_row_replays=rep, fuzz=fz, c2=cc), open(os.path.join(ROOT, "profiles", "r06_settled_mu_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
