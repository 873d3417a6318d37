"""Diagnostic: per-launch time of the solve kernel at a given batch for each lanes-per-instance setting (C2 workload)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np, torch
import mpc_gpu, bench
from mpc_gpu.sharding import shard_slice
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0"); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
for lanes, lps in ((32, 1), (21, 1), (64, 1), (0, 2), (0, 3)):
    x0, goal, obst = [a[:B] for a in bench.make_workload("c2", 1, 0, shard_slice)[:3]]
    loop = bench.Loop(mpc_gpu, torch, 20, 3, x0, goal, obst, dev); loop.reset()
    loop.m.set_lanes_per_stage(lps)
    if lanes: loop.m.set_lanes_per_instance(lanes)
    for _ in range(100): loop.control_step()
    torch.cuda.synchronize()
    loop.m.profile_enable(True)
    for _ in range(300): loop.control_step()
    torch.cuda.synchronize()
    ms, n = loop.m.profile_read()
    print(f"B={B} lanes per instance={loop.m.lanes_per_instance(B)} per stage={loop.m.lanes_per_stage(B)} mean iters {loop.iters.double().mean().item():.2f}: {ms / n * 1e3:.1f} us per launch, {B / (ms / n * 1e-3):.3e} solves/s (kernel only)")
