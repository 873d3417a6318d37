"""C5's problem (N = 50, 10 obstacles) under different interior-point start thresholds thr0 (mpc_config.thr0: t = max(rho, thr0), s = max(0, -h) + thr0): solves/s and
mean iterations of the bench loop at 32768 and at the 4096-instance share.  usage (GPU box): python scripts/thr0_probe.py -> gpurun_out/thr0_probe.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import torch, bench, mpc_gpu
from mpc_gpu.sharding import shard_slice
dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
out = {}
for wl, share in (("c5", 1), ("c5", 8), ("c3", 1)):
    N, no = bench.WORKLOADS[wl][:2]
    x0, goal, obst, desc, _, _ = bench.make_workload(wl, share, 0, shard_slice)
    for thr0 in (0.1, 0.2, 0.3, 0.5):
        loop = bench.Loop(mpc_gpu, torch, N, no, x0, goal, obst, dev, streams=bench.pick_streams(len(x0)), thr0=thr0)
        r = bench.measure(torch, None, loop, 1, None, 2, 1, dev)
        rec = dict(workload=wl, batch=len(x0), thr0=thr0, solves_per_s=len(x0) * bench.EPISODE * 2 / r["elapsed"], mean_iters=r["mean_iters"], fail=r["fail"], cap=r["cap"])
        out[f"{wl}_{len(x0)}_{thr0}"] = rec; print(rec, flush=True)
        del loop
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "thr0_probe.json"), "w"), indent=1)
