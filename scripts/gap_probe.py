"""Diagnostic: control-step time of the C2 loop with and without HIP events around every launch (what the events cost)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np, torch
import mpc_gpu, bench
from mpc_gpu.sharding import shard_slice
B = 1024
dev = torch.device("cuda:0"); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
x0, goal, obst = [a[:B] for a in bench.make_workload("c2", 1, 0, shard_slice)[:3]]
loop = bench.Loop(mpc_gpu, torch, 20, 3, x0, goal, obst, dev); loop.reset()
for _ in range(100): loop.control_step()
torch.cuda.synchronize()
for prof in (False, True, False, True):
    loop.m.profile_enable(prof)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): loop.control_step()
    torch.cuda.synchronize(); e = time.perf_counter() - t0
    extra = ""
    if prof:
        ms, n = loop.m.profile_read(); extra = f"  kernel {ms / n * 1e3:.1f} us"
    print(f"events around every launch: {prof}: {e / 500 * 1e6:.1f} us per control step{extra}")
