"""Which combination of the unverifiable acados-semantics switches reproduces the reference's recorded
closed-loop statistics best?  Statistical evidence only (the recorded tables are chaotic per seed)."""
import itertools, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np
import mpc_gpu
gold = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))
ref = {"RANDOM": dict(hit=0.16, reached=0.99, iters=114.8), "EDGE": dict(hit=0.11, reached=0.88, iters=168.3)}
x0 = np.tile([-7.0, -7.0, np.pi / 4, 0, 0], (100, 1)); goal = np.tile([7.0, 7.0], (100, 1))
rows = []
for cs, ss, lms, alias in itertools.product((1, 0), (1, 0), (0, 1), (True, False)):
    rec = dict(cost_scale_dt=cs, slack_scale_dt=ss, lm_scaled=lms, bug_compat_alias=alias)
    for scen in ("RANDOM", "EDGE"):
        acc = []
        for seed in (0, 1, 2):
            r = mpc_gpu.run_episodes(x0, goal, gold[f"gen_{scen}_5"], N=20, Tf=2.0, max_iter=400, random_move=True, seed=seed, qp_iter_max=100,
                                     bug_compat_alias=alias, cost_scale_dt=cs, slack_scale_dt=ss, lm_scaled=lms)["table"]
            acc.append([r[:, 0].mean(), r[:, 1].mean(), r[:, 4].mean()])
        m = np.mean(acc, 0)
        rec[scen] = dict(hit=round(float(m[0]), 3), reached=round(float(m[1]), 3), iters=round(float(m[2]), 1))
    rows.append(rec)
    print(rec)
json.dump(dict(reference_recorded=ref, scan=rows), open(os.path.join(ROOT, "gpurun_out", "switch_scan_r01.json"), "w"), indent=1)
