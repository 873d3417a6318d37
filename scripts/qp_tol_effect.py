"""Diagnostic: what the interior-point tolerance costs.  The default qp_tol = 1e-8 is tighter than acados' 1e-6 [acados-knowledge: the
nlp tolerances, 1e-6, are handed to HPIPM when qp_solver_tol_* are not set, robot_ocp_problem.py:126-132 sets none]; this measures the C2 / C3
closed-loop rates and iteration counts at both, and how many recorded rows of the reference's RANDOM / TF = 2 table each reproduces.
usage (GPU box): python scripts/qp_tol_effect.py   -> gpurun_out/qp_tol_effect.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import mpc_gpu, bench
from mpc_gpu.sharding import shard_slice
from mpc_gpu.world import reference_streams
dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
TABLES = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")))["tables"]
STABLE = [0, 2, 3, 4, 5, 24, 25, 36, 41, 53, 63, 65, 66, 69, 76, 79, 80, 81, 82, 84, 95]
out = {}
for tol in (1e-8, 1e-6):
    rec = {}
    for wl in ("c2", "c3"):
        x0, goal, obst = bench.make_workload(wl, 1, 0, shard_slice)[:3]
        with mpc_gpu.BatchedMpc(20, 3, 2.0, max_batch=x0.shape[0], qp_tol=tol) as m:
            loop = bench.Loop.__new__(bench.Loop)          # bench.Loop around a solver with this tolerance
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            B = x0.shape[0]
            loop.torch, loop.m, loop.B, loop.N, loop.no = torch, m, B, 20, 3
            loop.x0, loop.goal, loop.obst = t(x0), t(goal), t(obst)
            loop.x0_init, loop.obst_init = loop.x0.clone(), loop.obst.clone()
            z = lambda *s, dt=torch.float64: torch.zeros(*s, dtype=dt, device=dev)
            loop.X, loop.U, loop.u0, loop.cost = z(B, 21, 5), z(B, 20, 2), z(B, 2), z(B)
            loop.status, loop.iters = z(B, dt=torch.int32), z(B, dt=torch.int32)
            loop.stream = torch.cuda.current_stream().cuda_stream
            it_acc = z(B, dt=torch.int32); st_acc = z(B, dt=torch.int32)
            best = 1e9
            for rep in range(3):
                loop.reset(); it_acc.zero_(); m.set_accumulators(it_acc, st_acc)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(100): loop.control_step()
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            m.set_accumulators(None, None)
            rec[wl] = {"solves_per_s": B * 100 / best, "mean_ipm_iters": float(it_acc.double().sum().item()) / (B * 100), "x_end": loop.x0.cpu().numpy()}
    t = TABLES["20221031_215846"]; rows = np.array(t["rows"])
    ob, noise = reference_streams("RANDOM", range(100), 5, 400)
    x0 = np.tile([-7.0, -7.0, np.pi / 4, 0, 0], (100, 1)); goal = np.tile([7.0, 7.0], (100, 1))
    tb = mpc_gpu.run_episodes(x0, goal, ob, N=20, Tf=2.0, max_iter=400, random_move=True, init_guess_when_error=True, noise=noise, qp_iter_max=100, qp_tol=tol)["table"]
    flags = (tb[:, 0] == rows[:, 0]) & (tb[:, 1] == rows[:, 1]) & (tb[:, 5] == rows[:, 5]) & (tb[:, 4] == rows[:, 4])
    dev_ = np.maximum(np.abs(tb[:, 2] - rows[:, 2]), np.abs(tb[:, 3] - rows[:, 3]))
    rec["recorded_rows_RANDOM_TF2"] = {"matched_1e-3": int((flags & (dev_ <= 1e-3)).sum()), "matched_1e-6": int((flags & (dev_ <= 1e-6)).sum()),
                                       "stable_seeds_matched_1e-4": int((flags & (dev_ <= 1e-4))[STABLE].sum()), "stable_seeds": len(STABLE),
                                       "stable_max_dev": float(dev_[STABLE][flags[STABLE]].max())}
    out[f"{tol:g}"] = rec
a, b = out["1e-08"], out["1e-06"]
for wl in ("c2", "c3"):
    d = np.abs(a[wl]["x_end"] - b[wl]["x_end"]).max(1)
    out[f"{wl}_end_state_deviation_between_tolerances"] = {"median": float(np.median(d)), "p99": float(np.quantile(d, 0.99)), "max": float(d.max())}
    for k in (a, b): del k[wl]["x_end"]
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "qp_tol_effect.json"), "w"), indent=1)
