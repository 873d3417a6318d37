"""Diagnostic: build a -DMPC_PHASE_TIMING variant of the library and print the cycle share of each phase of the solve kernel."""
import sys, os, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")
sys.path[:0] = [ROOT, PKG]
import numpy as np, torch
so = os.path.join(ROOT, "gpurun_out", "libmpcgpu_timing.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                       "-DMPC_PHASE_TIMING"] + [a for a in sys.argv[2:] if a.startswith("-D")] + ["-o", so, os.path.join(PKG, "csrc", "mpc_api.hip")])
from mpc_gpu import _lib
_lib.LIB_PATH = so
import mpc_gpu, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x0, goal, obst, _ = bench.make_workload("c2" if B <= 4096 else "c3", B, 20, 3)
dev = torch.device("cuda:0"); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
loop = bench.Loop(mpc_gpu, 20, 3, B, x0, goal, obst, dev)
_lib.check(_lib.lib().mpc_debug_trace(loop.m._h, 1, B, None))
for _ in range(10): loop.step()
torch.cuda.synchronize()
tr = np.zeros((B, 50, 4)); _lib.check(_lib.lib().mpc_debug_trace(loop.m._h, 1, B, tr.ctypes.data))
t = tr.reshape(B, -1)[:, :16]
names = ["mu/conv check", "predictor assemble", "factor sweep", "rollout (affine)", "affine step + sigma", "corrector rhs", "corrector sweep", "rollout", "combined step + update", "(row-parallel: stage operands -> LDS)"]
tot = t[:, :10].sum(1).mean(); it = loop.iters.double().mean().item()
print(f"B={B} mean IPM iters {it:.2f}  total cycles/solve (IPM loop) {tot:.0f}  per iteration {tot/it:.0f}")
for k, n in zip(range(10, 15), ["loads + look-ahead", "iterate loads", "linearise + W staging", "row state init", "tail (step, plant, stores)"]):
    print(f"  outside the IPM loop: {n:28s} {t[:,k].mean():10.0f} cycles")
print(f"  (of the affine rollout: the sweep itself        {t[:,15].mean():10.0f} cycles, per iter {t[:,15].mean()/it:8.0f})")
for k, n in enumerate(names[:10]): print(f"  {n:24s} {t[:,k].mean():10.0f} cycles  {100*t[:,k].mean()/tot:5.1f}%   per iter {t[:,k].mean()/it:8.0f}")
