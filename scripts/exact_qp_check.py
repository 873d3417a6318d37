"""Are this repository's QP solutions EXACT on the episodes that miss their recorded row by one control step?  (seeds whose recorded rows
agree between QP_ITER 100 and 50 but not 25.)  CPU only.  The episode is replayed on the oracle (experiments.py:20-36 with the reference's numpy streams); at
every control step the QP of the RTI step is exported (orc_export_qp), the ACTIVE SET is read off the interior point's answer, and the equality-constrained QP
on that active set is solved by one dense KKT factorisation (numpy): if its multipliers are non-negative and every inactive row is satisfied, that is the
unique solution of the strictly convex QP to rounding -- an answer that owes nothing to the interior point but the guess of the active set.
Output: per seed the largest |interior point - exact| over the episode, the step it occurs at and that step's iteration count.
usage: python scripts/exact_qp_check.py [SCENARIO:seed ...]      -> profiles/r03_exact_qp_check.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from oracle import oracle as orc
from helpers import OracleLoop, exact_from_active_set
from mpc_gpu.world import reference_streams


def main():
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")))["tables"]
    cases = sys.argv[1:] or ["RANDOM:62", "RANDOM:94", "EDGE:73", "EDGE:14", "RANDOM:0"]
    stems = {"RANDOM": "20221031_215846", "EDGE": "20221031_220136"}
    out = {"method": __doc__.split("Output")[0].strip(), "cases": []}
    for c in cases:
        scen, seed = c.split(":"); seed = int(seed)
        sp = ref[stems[scen]]["spec"]
        obst, noise = reference_streams(scen, [seed], 5, 400)
        cfg = orc.config(sp["N_SOLV"], 5, float(sp["TF"]), qp_iter_max=sp["QP_ITER"])
        lp = OracleLoop(orc, cfg, [-7.0, -7.0, np.pi / 4, 0, 0], [7.0, 7.0], obst[0], reset_on_fail=True, alias=True)
        worst = dict(dev=0.0, step=-1, iters=0); unverified = 0; max_it = 0; n = 0; worst_u0 = 0.0
        for k in range(400):
            if lp.flags & 1:
                break
            P = orc.predict_params(cfg, lp.obst)
            q = orc.export_qp(cfg, lp.x, P, lp.goal, lp.X, lp.U)
            X0, U0 = lp.X.copy(), lp.U.copy()
            r = lp.step(noise[k, 0])
            if r is None:
                break
            n += 1; max_it = max(max_it, int(r["iters"]))
            if r["status"] != 0:
                continue
            dX, dU = r["X"] - X0, r["U"] - U0
            v_ip = np.concatenate([np.concatenate([dU[i], dX[i + 1]]) for i in range(cfg.N)])
            v_ex, lam_min, feas, n_act, res = exact_from_active_set(q, v_ip)
            if lam_min < -1e-7 or feas < -1e-7 or res > 1e-9:
                unverified += 1        # the guessed active set is not the optimal one (a row within 1e-7 of its bound on the wrong side): no statement
                continue
            d = float(np.abs(v_ip - v_ex).max())
            du0 = float(np.abs(v_ip[:2] - v_ex[:2]).max())          # the control that is applied
            worst_u0 = max(worst_u0, du0)
            if d > worst["dev"]:
                worst = dict(dev=d, where="stage %d, variable %d of (ua, ual, x, y, psi, v, om)" % divmod(int(np.abs(v_ip - v_ex).argmax()), 7), step=k, iters=int(r["iters"]),
                             active_rows=n_act, dev_u0_at_that_step=du0)
        row = lp.row(); rec = ref[stems[scen]]["rows"][seed]
        rec_ = dict(case=c, control_steps=n, max_ipm_iters=max_it, steps_without_a_verified_active_set=unverified, worst_deviation_from_exact=worst, worst_deviation_of_the_applied_control=worst_u0,
                    oracle_row=row, recorded_row=rec)
        out["cases"].append(rec_); print(json.dumps(rec_), flush=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r03_exact_qp_check.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
