"""Why is a GPU's SHARE of the 8-GPU long-horizon configuration (C5: 4096 x (N = 50, 10 obstacles) per GPU) slower per instance than the whole batch on one GPU?
One wavefront per SIMD = 1024 wavefront slots; 4096 instances are four instances per slot, and interior-point iteration counts are heavy-tailed (mean ~14, cap 50).
Per control step of the bench's C5-share loop this script records the iteration count of every instance and the order the launch dealt them in, and replays the
launch as list scheduling on 1024 slots (an instance costs its iterations + a constant for set-up and tail): makespan for (a) the order the library used (sorted
by the PREVIOUS step's counts), (b) the natural order, (c) the order sorted by the TRUE counts of this step (what a perfect predictor would give), against the
lower bound max(mean load, longest instance).  The measured kernel time per step sits beside it.
usage (GPU box): python scripts/share_tail_probe.py [batch] -> gpurun_out/share_tail_probe.json"""
import heapq, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np
import torch
import bench
import mpc_gpu
from mpc_gpu.sharding import shard_slice

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
SLOTS, OVERHEAD = 1024, 1.2      # wavefront slots of the chip (one 512-register wavefront per SIMD); set-up + tail of a solve in iteration equivalents (r02_c5_phase_timing.txt)


def makespan(durations):
    h = [0.0] * SLOTS
    heapq.heapify(h)
    for d in durations:
        heapq.heappush(h, heapq.heappop(h) + d)
    return max(h)


dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
N, no = bench.WORKLOADS["c5"][:2]
x0, goal, obst, desc, _, _ = bench.make_workload("c5", 32768 // B, 0, shard_slice)
assert len(x0) == B
loop = bench.Loop(mpc_gpu, torch, N, no, x0, goal, obst, dev)
rows = []
for ep in range(2):
    loop.reset()
    for k in range(bench.EPISODE):
        order = loop.m.instance_order(B)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); loop.control_step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if ep == 0 or k % 5:
            continue
        it = loop.iters.cpu().numpy().astype(float) + OVERHEAD
        order = np.arange(B) if order is None or len(order) == 0 else np.asarray(order)
        lb = max(it.sum() / SLOTS, it.max())
        rec = dict(step=k, ms=dt * 1e3, mean_iters=float(it.mean() - OVERHEAD), max_iters=float(it.max() - OVERHEAD), frac_ge_30=float((it - OVERHEAD >= 30).mean()),
                   lower_bound=lb, used_order=makespan(it[order]) / lb, natural_order=makespan(it) / lb, perfect_order=makespan(np.sort(it)[::-1]) / lb,
                   us_per_iteration_slot=dt * 1e6 / makespan(it[order]))
        rows.append(rec); print(rec, flush=True)
agg = {k: float(np.mean([r[k] for r in rows])) for k in rows[0] if k != "step"}
out = dict(method=__doc__.split("usage")[0].strip(), workload=desc, batch=B, slots=SLOTS, overhead_iterations=OVERHEAD, mean=agg, steps=rows)
print(json.dumps(agg, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"share_tail_probe_{B}.json"), "w"), indent=1)
