"""Does pipelining independent sub-batches on several HIP streams hide the straggler tail of a launch?  (scripts/share_tail_probe.py: at 4096 x (N = 50, 10 obstacles)
a launch takes 1.34x its mean slot load because unpredicted 50-iteration solves start in the last round of wavefronts.)  The batch is cut into K contiguous
sub-batches, each with its own handle and stream; every control step launches K kernels that depend only on their own predecessor, so the tail of one overlaps
with the body of the others.  usage (GPU box): python scripts/streams_probe.py [workload] [share] -> gpurun_out/streams_probe_<workload>_<batch>.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np
import torch
import bench
import mpc_gpu
from mpc_gpu.sharding import shard_slice

wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
share = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
N, no = bench.WORKLOADS[wl][:2]
x0, goal, obst, desc, _, _ = bench.make_workload(wl, share, 0, shard_slice)
B = len(x0)
res = {}
for K in (1, 2, 4, 8):
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
    loops = []
    for k in range(K):
        lo, hi = shard_slice(B, k, K)
        with torch.cuda.stream(streams[k]):
            loops.append(bench.Loop(mpc_gpu, torch, N, no, x0[lo:hi], goal[lo:hi], obst[lo:hi], dev))

    def episode():
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                loops[k].reset()
        for _ in range(bench.EPISODE):
            for k in range(K):
                with torch.cuda.stream(streams[k]):
                    loops[k].control_step()
    episode(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        episode()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[K] = dict(streams=K, solves_per_s=B * bench.EPISODE * 3 / dt, ms_per_control_step=dt / (3 * bench.EPISODE) * 1e3, kernel=loops[0].m.kernel_name(loops[0].B))
    print(res[K], flush=True)
    del loops
out = dict(method=__doc__.split("usage")[0].strip(), workload=desc, batch=B, results=res)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"streams_probe_{wl}_{B}.json"), "w"), indent=1)
