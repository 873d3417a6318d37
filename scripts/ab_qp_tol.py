"""Control-step time and iteration count of the bench workloads as a function of qp_tol (GPU box).  usage: python scripts/ab_qp_tol.py [tol ...]"""
import functools, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, mpc_gpu, bench
from mpc_gpu.sharding import shard_slice
tols = [float(a) for a in sys.argv[1:]] or [1e-8, 1e-10]
dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
Base = mpc_gpu.BatchedMpc
out = {}
for wl in ("c2", "c3", "c5"):
    N, no = bench.WORKLOADS[wl][:2]
    x0, goal, obst, desc, _, G = bench.make_workload(wl, 1, 0, shard_slice)
    for tol in tols:
        mpc_gpu.BatchedMpc = functools.partial(Base, qp_tol=tol)
        loop = bench.Loop(mpc_gpu, torch, N, no, x0, goal, obst, dev)
        mpc_gpu.BatchedMpc = Base
        best = 1e9; its = []
        for rep in range(2):
            loop.reset(); torch.cuda.synchronize(); t = time.perf_counter()
            for k in range(100):
                loop.control_step()
                if rep == 1 and k % 10 == 0: its.append(float(loop.iters.double().mean()))
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 100)
        out[f"{wl} qp_tol={tol:g}"] = dict(ms_per_control_step=best * 1e3, solves_per_s=x0.shape[0] / best, mean_iters_sampled=float(np.mean(its)))
        print(wl, tol, out[f"{wl} qp_tol={tol:g}"], flush=True)
        loop.m.close()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_ab_qp_tol.json"), "w"), indent=1)
