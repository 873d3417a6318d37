import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np, torch
import bench, mpc_gpu
dev = torch.device("cuda:0"); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
x0, goal, obst, desc = bench.make_workload("c2", 1024, 20, 3)
loop = bench.Loop(mpc_gpu, 20, 3, 1024, x0, goal, obst, dev)
for _ in range(5): loop.step()
torch.cuda.synchronize()
def timeit(name, fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:12s} host {1e6*(t1-t0)/n:9.1f} us/call   incl. drain {1e6*(t2-t0)/n:9.1f} us/call")
m, B, s = loop.m, loop.B, loop.stream
timeit("predict", lambda: m.predict_dev(B, loop.obst, loop.P, stream=s))
timeit("solve", lambda: m.solve_dev(B, loop.x0, loop.P, loop.goal, loop.X, loop.U, loop.u0, loop.cost, loop.status, loop.iters, stream=s))
timeit("plant", lambda: m.plant_step_dev(B, loop.x0, loop.u0, loop.x1, stream=s))
timeit("obst", lambda: m.obstacle_step_dev(B * 3, loop.obst, None, stream=s))
timeit("shift", lambda: m.shift_dev(B, loop.X, loop.U, stream=s))
timeit("step", loop.step)
m.profile_enable(True)
timeit("step+prof", loop.step)
print(m.profile_read())
