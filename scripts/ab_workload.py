"""Diagnostic: closed-loop control-step time of a RANDOMISED workload (C3's distributions) of any horizon / obstacle count / batch with several builds
of the library on ONE box.  usage (GPU box): python scripts/ab_workload.py N n_obst batch lib [lib ...]     ("default" = the in-tree library)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")

def child(N, no, B):
    sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, "tests")]
    import torch, mpc_gpu, bench
    from helpers import random_batch
    dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    x0, goal, obst = random_batch(B, no, seed=1234)
    loop = bench.Loop(mpc_gpu, torch, N, no, x0, goal, obst, dev)
    best = 1e9
    for rep in range(3):
        loop.reset(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(60): loop.control_step()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 60)
    print(json.dumps({"kernel": loop.m.kernel_name(B), "ms_per_control_step": best * 1e3, "solves_per_s": B / best, "iters_last": float(loop.iters.double().mean())}))

if __name__ == "__main__":
    if sys.argv[1] == "--child": child(*map(int, sys.argv[2:5])); sys.exit(0)
    N, no, B = sys.argv[1:4]
    for rnd in range(2):
        for lib in sys.argv[4:] or ["default"]:
            env = dict(os.environ)
            if lib != "default": env["MPC_GPU_LIB"] = os.path.join(ROOT, lib)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", N, no, B], env=env, capture_output=True, text=True)
            print(lib, r.stdout.strip().split("\n")[-1] if r.returncode == 0 else r.stderr[-400:], flush=True)
