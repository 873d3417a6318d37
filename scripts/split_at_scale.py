"""Diagnostic: the stage-split mapping (one instance per wavefront) forced onto a LARGE batch against the default mapping (one lane per
stage, two instances per wavefront): steady closed loop of the C3 workload."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np, torch
import mpc_gpu, bench
from mpc_gpu.sharding import shard_slice
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0"); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
for lps in (1, 3):
    x0, goal, obst = [a[:B] for a in bench.make_workload("c3", 1, 0, shard_slice)[:3]]
    loop = bench.Loop(mpc_gpu, torch, 20, 3, x0, goal, obst, dev); loop.reset()
    loop.m.set_lanes_per_stage(lps)
    for _ in range(10): loop.control_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): loop.control_step()
    torch.cuda.synchronize(); e = time.perf_counter() - t0
    print(f"B={B} lanes per stage {loop.m.lanes_per_stage(B)} (lanes per instance {loop.m.lanes_per_instance(B)}): {e / 30 * 1e3:.3f} ms per control step, {B * 30 / e:.3e} solves/s")
