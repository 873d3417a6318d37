#!/usr/bin/env python3
"""The 12 seeds whose recorded rows prove that acados converged at every control step (rows equal between QP_ITER 100 and 50) and which the replay still
misses (profiles/r03_unmatched_rows.json): RANDOM 28, 29, 37, 39, 62, 74, 94 and EDGE 14, 23, 46, 52, 73 of
src/simulation/test_data/20221031_215846 / _220136 (QP_ITER 100) and _220735 / _220939 (QP_ITER 50).

Every remaining hypothesis about what acados / HPIPM did differently is replayed on the CPU oracle (scripts/oracle_variant_replay.py; test infrastructure, no GPU):
  tolerances      qp_tol 1e-6 / 1e-8 / 1e-12, polish off, stationarity gated as well (HPIPM's separate res_g test, at 1e-8)
  rows            obstacle rows present at stage 0; zero-penalty terminal rows kept as free-slack rows; both; the state box at stage N (bx_terminal)
and, as the yardstick, the SENSITIVITY of each of these closed loops: the same replay with the start position moved by 1e-9 and by 1e-12 -- a perturbation
far below every solver tolerance.  Output: per seed and table the recorded row, the replayed row per variant, whether any variant lands on the recorded row,
and |d min_margin|, d steps between the base replay and the perturbed ones.
    python scripts/converged_unmatched.py        -> profiles/r05_converged_unmatched.json"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEEDS = "RANDOM:28,29,37,39,62,74,94;EDGE:14,23,46,52,73"
TABLES = "20221031_215846,20221031_220136,20221031_220735,20221031_220939"
VARIANTS = {
    "base": [], "qp_tol_1e-6": ["--cfg", "qp_tol=1e-6"], "qp_tol_1e-8": ["--cfg", "qp_tol=1e-8"], "qp_tol_1e-12": ["--cfg", "qp_tol=1e-12"],
    "polish_off": ["--cfg", "polish_tol=0.0"], "stationarity_gated_1e-8": ["--exp", "4", "--cfg", "qp_tol=1e-8"],
    "stage0_rows": ["--exp", "1"], "terminal_free_slack_rows": ["--exp", "2"], "stage0_and_terminal_rows": ["--exp", "3"], "bx_terminal": ["--cfg", "bx_terminal=1"],
    "hpipm_like(all three, tol 1e-8)": ["--exp", "7", "--cfg", "qp_tol=1e-8"],
    "perturb_x0_1e-9": ["--perturb", "1e-9"], "perturb_x0_1e-12": ["--perturb", "1e-12"], "perturb_x0_1e-6": ["--perturb", "1e-6"],
}
res = {}
for name, args in VARIANTS.items():
    with tempfile.NamedTemporaryFile(suffix=".json") as f:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "oracle_variant_replay.py"), "--seeds", SEEDS, "--tables", TABLES, "--tag", name,
                               "--out", f.name] + args, stdout=subprocess.DEVNULL)
        res[name] = json.load(open(f.name))
out = {"method": __doc__.split("python scripts")[0].strip(), "variants": list(VARIANTS), "seeds": {}, "summary": {}}
recovered = {v: 0 for v in VARIANTS}
sens = []
for stem, t in res["base"]["tables"].items():
    for seed, p in t["per_seed"].items():
        key = f"{t['spec']['scenario']}_{seed}_QP{t['spec']['QP_ITER']}"
        rec = dict(table=stem, recorded=p["recorded"], clean_episode=p["status2"] + p["status4"] == 0, rows={}, matched_by=[])
        for v in VARIANTS:
            q = res[v]["tables"][stem]["per_seed"][seed]
            rec["rows"][v] = dict(row=q["row"], m3=q["m3"], nonconverged=q["status2"] + q["status4"])
            if q["m3"]:
                rec["matched_by"].append(v); recovered[v] += 1
        b = rec["rows"]["base"]["row"]
        for v in ("perturb_x0_1e-12", "perturb_x0_1e-9", "perturb_x0_1e-6"):
            r = rec["rows"][v]["row"]
            rec[f"d_{v}"] = dict(d_min_margin=abs(r[2] - b[2]), d_steps=r[4] - b[4], d_dist=abs(r[3] - b[3]))
        rec["d_recorded_vs_base"] = dict(d_min_margin=abs(p["recorded"][2] - b[2]), d_steps=p["recorded"][4] - b[4], flags_equal=[p["recorded"][k] == b[k] for k in (0, 1, 5)])
        out["seeds"][key] = rec
        sens.append((key, rec["d_recorded_vs_base"]["d_min_margin"], rec["d_recorded_vs_base"]["d_steps"], rec["d_perturb_x0_1e-9"]["d_min_margin"], rec["d_perturb_x0_1e-9"]["d_steps"],
                     rec["d_perturb_x0_1e-12"]["d_min_margin"], rec["d_perturb_x0_1e-12"]["d_steps"]))
out["summary"] = dict(rows=len(out["seeds"]), recovered_per_variant=recovered, clean_on_base=sum(r["clean_episode"] for r in out["seeds"].values()))
json.dump(out, open(os.path.join(ROOT, "profiles", "r05_converged_unmatched.json"), "w"), indent=1)
print(json.dumps(out["summary"]))
print("seed | recorded-vs-base d_margin d_steps | 1e-9 perturbation d_margin d_steps | 1e-12 perturbation d_margin d_steps | clean")
for s in sens:
    print(f"{s[0]:18s} {s[1]:.2e} {s[2]:+5.0f} | {s[3]:.2e} {s[4]:+5.0f} | {s[5]:.2e} {s[6]:+5.0f} | {out['seeds'][s[0]]['clean_episode']}")
