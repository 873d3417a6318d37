"""Recorded rows reproduced as a function of this solver's own tolerance (GPU box): the two TF = 2 / QP_ITER = 100 tables at qp_tol 1e-6 ... 1e-12.
If acados' solutions were closer to the exact QP solutions than ours, a tighter tolerance would bring more rows back.  -> gpurun_out/r03_qp_tol_replay.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np
import mpc_gpu
from mpc_gpu.world import reference_streams
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")))["tables"]
x0 = np.tile([-7.0, -7.0, np.pi / 4, 0, 0], (100, 1)); goal = np.tile([7.0, 7.0], (100, 1))
out = {}
for stem in ("20221031_215846", "20221031_220136"):
    sp = ref[stem]["spec"]; rows = np.array(ref[stem]["rows"])
    obst, noise = reference_streams(sp["scenario"], range(100), 5, 400)
    for tol in (1e-6, 1e-8, 1e-10, 1e-12):
        r = mpc_gpu.run_episodes(x0, goal, obst, N=20, Tf=2.0, max_iter=400, random_move=True, init_guess_when_error=True, noise=noise, qp_iter_max=100, qp_tol=tol,
                                 status_log=True)
        tb = r["table"]
        fl = (tb[:, 0] == rows[:, 0]) & (tb[:, 1] == rows[:, 1]) & (tb[:, 5] == rows[:, 5]) & (tb[:, 4] == rows[:, 4])
        dm = np.abs(tb[:, 2] - rows[:, 2]); dd = np.abs(tb[:, 3] - rows[:, 3])
        rec = dict(matched_1e3=int((fl & (dm <= 1e-3) & (dd <= 1e-3)).sum()), matched_1e6=int((fl & (dm <= 1e-6) & (dd <= 1e-6)).sum()),
                   matched_1e8=int((fl & (dm <= 1e-8)).sum()), median_margin_deviation_of_matched=float(np.median(dm[fl & (dm <= 1e-3)])),
                   solves_at_cap=int(r["status2"].sum()), solves_failed=int(r["status4"].sum()))
        out[f"{stem} {sp['scenario']} qp_tol={tol:g}"] = rec
        print(stem, sp["scenario"], tol, rec, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_qp_tol_replay.json"), "w"), indent=1)
