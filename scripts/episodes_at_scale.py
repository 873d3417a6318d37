"""The reference's experiment protocol (experiments.py:20-36: np.random.seed(i), scenario draw, 5 noisy obstacles, at most 400 control steps, stop at the goal) for MANY seeds at
once, everything random produced on the device (GPU box).  usage: python scripts/episodes_at_scale.py [count ...]  -> gpurun_out/<tag>_episodes_at_scale.json (MPC_PROFILE_TAG, default r05)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np, torch, mpc_gpu
out = {}
for B in [int(a) for a in sys.argv[1:]] or [100, 13000, 100000]:
    x0 = np.tile([-7.0, -7.0, np.pi / 4, 0, 0], (B, 1)); goal = np.tile([7.0, 7.0], (B, 1))
    for scen in ("RANDOM", "EDGE"):
        best = None
        for rep in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            r = mpc_gpu.run_episodes(x0, goal, scen, N=20, Tf=2.0, max_iter=400, random_move=True, init_guess_when_error=True, first_seed=0, qp_iter_max=100)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        tb = r["table"]
        out[f"{scen} x {B}"] = dict(seconds=best, control_steps_run=int(r["steps_run"]), solves=int(r["solves"]), solves_per_s=r["solves"] / best,
                                    reached=float(tb[:, 1].mean()), hit=float(tb[:, 0].mean()), mean_steps=float(tb[:, 4].mean()), out_of_bounds=float(tb[:, 5].mean()))
        print(scen, B, out[f"{scen} x {B}"], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", os.environ.get("MPC_PROFILE_TAG", "r05") + "_episodes_at_scale.json"), "w"), indent=1)
