#!/usr/bin/env python3
"""Hypothesis for the converged-but-unmatched seeds: acados' QP FAILED at some control step in a way that does not depend on QP_ITER (HPIPM's
minimum-step / NaN exits -> ACADOS_QP_FAILURE -> robot_ocp_problem.py:203-205 set_initial_guess()), which leaves the rows of the QP_ITER 100 and 50 tables
equal although not every solve converged.  Test: replay each seed on the oracle with ONE forced failure at control step k (the solve is skipped: iterate
untouched, stored u_0 applied, set_initial_guess with the aliasing defect), for every k, and report the k that land on the recorded row.
    python scripts/forced_failure_probe.py   -> profiles/r05_forced_failure_probe.json"""
import json, multiprocessing as mp, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
SEEDS = {"RANDOM": [28, 29, 37, 39, 62, 74, 94], "EDGE": [14, 23, 46, 52, 73]}
STEMS = {"RANDOM": "20221031_215846", "EDGE": "20221031_220136"}


def run(job):
    scen, seed, kfail = job
    from oracle import oracle as orc
    from helpers import OracleLoop
    from mpc_gpu.world import reference_streams
    obst, noise = reference_streams(scen, [seed], 5, 400)
    cfg = orc.config(20, 5, 2.0, qp_iter_max=100)
    lp = OracleLoop(orc, cfg, [-7.0, -7.0, np.pi / 4, 0, 0], [7.0, 7.0], obst[0], reset_on_fail=True, alias=True)
    real = orc.rti_solve
    for k in range(400):
        if lp.flags & 1:
            break
        if (k in kfail) if isinstance(kfail, tuple) else (k == kfail):      # the forced failure: status 4, iterate untouched, u = stored u_0
            orc.rti_solve = lambda cfg_, x, P, g, X, U, alpha=None: dict(X=X.copy(), U=U.copy(), u0=U[0].copy(), cost=0.0, status=4, iters=0, kkt=np.zeros(4))
        try:
            lp.step(noise[k, 0])
        finally:
            orc.rti_solve = real
    return scen, seed, kfail, lp.row()


def pairs():
    """second pass (argument `pairs`): the seeds ONE forced failure does not reproduce to 1e-6, with TWO forced failures at steps k1 < k2 (all pairs)"""
    T = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")))["tables"]
    first = json.load(open(os.path.join(ROOT, "profiles", "r05_forced_failure_probe.json")))
    todo = [k for k, v in first["seeds"].items() if not v["forced_failure_steps_that_reproduce_the_row_1e6"]]
    jobs = []
    for key in todo:
        sc, s = key.split("_"); s = int(s)
        n = int(T[STEMS[sc]]["rows"][s][4]) + 10
        jobs += [(sc, s, (a, b)) for a in range(n) for b in range(a + 1, n)]
    print(len(jobs), "episodes", flush=True)
    with mp.Pool(8) as pool:
        res = pool.map(run, jobs, chunksize=64)
    for key in todo:
        sc, s = key.split("_"); s = int(s)
        rec = T[STEMS[sc]]["rows"][s]
        hits, near = [], []
        for _, _, k, row in [r for r in res if r[0] == sc and r[1] == s]:
            fl = all(row[i] == rec[i] for i in (0, 1, 4, 5))
            d = max(abs(row[2] - rec[2]), abs(row[3] - rec[3]))
            if fl and d <= 1e-6: hits.append(list(k))
            elif fl and d <= 1e-3: near.append(list(k))
        first["seeds"][key]["two_forced_failures_that_reproduce_the_row_1e6"] = hits
        first["seeds"][key]["two_forced_failures_to_1e3"] = len(near)
        print(key, "recorded", rec, "-> two forced failures at", hits, f"({len(near)} pairs to 1e-3)", flush=True)
    json.dump(first, open(os.path.join(ROOT, "profiles", "r05_forced_failure_probe.json"), "w"), indent=1)


def main():
    T = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")))["tables"]
    jobs = [(sc, s, k) for sc, ss in SEEDS.items() for s in ss for k in range(-1, int(T[STEMS[sc]]["rows"][s][4]) + 30)]
    with mp.Pool(8) as pool:
        res = pool.map(run, jobs, chunksize=8)
    out = {"method": __doc__.split("python scripts")[0].strip(), "seeds": {}}
    for sc, ss in SEEDS.items():
        for s in ss:
            rec = T[STEMS[sc]]["rows"][s]
            hits, near = [], []
            for _, _, k, row in [r for r in res if r[0] == sc and r[1] == s]:
                fl = all(row[i] == rec[i] for i in (0, 1, 4, 5))
                d = max(abs(row[2] - rec[2]), abs(row[3] - rec[3]))
                if fl and d <= 1e-6: hits.append(k)
                elif fl and d <= 1e-3: near.append(k)
            out["seeds"][f"{sc}_{s}"] = dict(recorded=rec, forced_failure_steps_that_reproduce_the_row_1e6=hits, to_1e3=near)
            print(sc, s, "recorded", rec, "-> single forced failure at step", hits, "(1e-3:", near, ")", flush=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r05_forced_failure_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    pairs() if len(sys.argv) > 1 and sys.argv[1] == "pairs" else main()
