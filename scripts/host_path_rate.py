"""PCIe-inclusive rate of the host-pointer API: mpc_solve_obst with host numpy arrays in and out, per call
H2D of (x0, obst, goal), one launch, D2H of (u0, cost, status, iters), one stream synchronize.  Never reported as `value`."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np
import bench, mpc_gpu
from mpc_gpu.sharding import shard_slice
for B in (1, 64, 1024, 65536):
    x0, goal, obst = bench.make_workload("c2" if B <= 1024 else "c3", 1, 0, shard_slice)[:3]
    x0, goal, obst = x0[:B], goal[:B], obst[:B]
    with mpc_gpu.BatchedMpc(20, 3, 2.0, max_batch=B) as s:
        s.reset_guess(x0)
        for _ in range(3): s.solve(x0, obst, goal); s.shift(B)
        n = 200 if B <= 1024 else 20
        t0 = time.perf_counter()
        for _ in range(n):
            r = s.solve(x0, obst, goal); s.shift(B)
        dt = (time.perf_counter() - t0) / n
        print(f"B={B:6d}: {1e3 * dt:8.3f} ms per solve+shift call pair  -> {B / dt:12.0f} solves/s PCIe-inclusive (mean iters {r['iters'].mean():.2f})")
