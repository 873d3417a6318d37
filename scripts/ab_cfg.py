"""Diagnostic: closed-loop control-step time of a randomised workload (C3's distributions) under CONFIGURATION variants and / or library builds on ONE box, interleaved.
usage (GPU box): python scripts/ab_cfg.py N n_obst batch variant [variant ...]      variant = lib[:key=value[,key=value]]   ("default" = the in-tree library)
e.g.  python scripts/ab_cfg.py 50 10 32768 build/lib_r05.so default default:polish_res_g=0.0"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")

def child(N, no, B, cfg):
    sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, "tests")]
    import torch, mpc_gpu, bench
    from helpers import random_batch
    kw = {}
    for kv in filter(None, cfg.split(",")):
        k, v = kv.split("="); kw[k] = float(v) if ("." in v or "e" in v) else int(v)
    dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    x0, goal, obst = random_batch(B, no, seed=1234)
    loop = bench.Loop(mpc_gpu, torch, N, no, x0, goal, obst, dev, **kw)
    best = 1e9; its = 0.0
    for rep in range(3):
        loop.reset(); torch.cuda.synchronize(); t = time.perf_counter(); acc = 0.0
        for _ in range(60): loop.control_step()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 60)
    print(json.dumps({"kernel": loop.m.kernel_name(B), "ms_per_control_step": round(best * 1e3, 4), "solves_per_s": round(B / best), "iters_last": float(loop.iters.double().mean())}))

if __name__ == "__main__":
    if sys.argv[1] == "--child": child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] if len(sys.argv) > 5 else ""); sys.exit(0)
    N, no, B = sys.argv[1:4]
    for rnd in range(2):
        for var in sys.argv[4:] or ["default"]:
            lib, _, cfg = var.partition(":")
            env = dict(os.environ)
            if lib != "default": env["MPC_GPU_LIB"] = os.path.join(ROOT, lib)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", N, no, B, cfg], env=env, capture_output=True, text=True)
            print(var, r.stdout.strip().split("\n")[-1] if r.returncode == 0 else r.stderr[-400:], flush=True)
