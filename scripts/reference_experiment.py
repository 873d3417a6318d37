"""Reproduce the protocol of the reference's src/simulation/experiments.py on the GPU episode harness:
2 scenarios x 100 seeds, start [-7,-7,pi/4,0,0], goal [7,7], N_OBST = 5, noisy obstacles, init_guess_when_error, max 400 steps.
Scenario draws are the reference generator's own (np.random.seed(i), reproduced bit for bit by the device generator); the obstacle noise stream is the GPU's
(torch), so per-seed rows are not comparable, only the statistics (the tables are chaotic anyway)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np
import mpc_gpu

gold = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")))["tables"]
out = {}
for scen, tf, n, qp in (("RANDOM", 2.0, 20, 100), ("EDGE", 2.0, 20, 100), ("RANDOM", 1.0, 10, 50), ("EDGE", 1.0, 10, 50)):
    obst = scen                                        # drawn on the device: np.random.seed(i) streams, i < 100 (== gold[f"gen_{scen}_5"])
    x0 = np.tile([-7.0, -7.0, np.pi / 4, 0, 0], (100, 1)); goal = np.tile([7.0, 7.0], (100, 1))
    r = mpc_gpu.run_episodes(x0, goal, obst, N=n, Tf=tf, max_iter=400, random_move=True, init_guess_when_error=True, seed=0, qp_iter_max=qp)
    tb = r["table"]
    mine = dict(hit=float(tb[:, 0].mean()), reached=float(tb[:, 1].mean()), mean_iters=float(tb[:, 4].mean()), oob=float(tb[:, 5].mean()))
    theirs = [v for v in ref.values() if v["spec"]["scenario"] == scen and v["spec"]["N_SOLV"] == n and v["spec"]["QP_ITER"] == qp]
    th = {k: theirs[0][k] for k in ("hit", "reached", "mean_iters", "oob")} if theirs else None
    out[f"{scen}_TF{tf:g}_QP{qp}"] = dict(gpu=mine, reference_recorded=th)
    print(scen, tf, qp, "gpu", mine, "| reference recorded", th)
    mpc_gpu.write_experiment(tb, {"slack": True, "random_move": True, "init_guess": True, "scenario": scen, "TF": tf, "N_SOLV": n, "N_OBST": 5, "QP_ITER": qp},
                             os.path.join(ROOT, "gpurun_out", "experiments"), stamp=f"r01_{scen}_TF{tf:g}")
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "reference_experiment_r01.json"), "w"), indent=1)
