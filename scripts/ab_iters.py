"""Diagnostic: per-control-step mean interior-point iterations and status counts of a closed loop with several builds (same inputs).
usage (GPU box): python scripts/ab_iters.py N n_obst batch steps lib [lib ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")

def child(N, no, B, steps):
    sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, "tests")]
    import torch, mpc_gpu, bench
    from helpers import random_batch
    dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    x0, goal, obst = random_batch(B, no, seed=1234)
    loop = bench.Loop(mpc_gpu, torch, N, no, x0, goal, obst, dev)
    loop.reset(); rows = []
    for k in range(steps):
        loop.control_step(); torch.cuda.synchronize()
        st = loop.status.cpu().numpy(); it = loop.iters.cpu().numpy()
        rows.append([k, float(it.mean()), int((st == 2).sum()), int((st == 4).sum()), float(loop.x0[:, :2].abs().double().mean())])
    print(json.dumps(rows))

if __name__ == "__main__":
    if sys.argv[1] == "--child": child(*map(int, sys.argv[2:6])); sys.exit(0)
    N, no, B, steps = sys.argv[1:5]
    res = {}
    for lib in sys.argv[5:]:
        env = dict(os.environ)
        if lib != "default": env["MPC_GPU_LIB"] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", N, no, B, steps], env=env, capture_output=True, text=True)
        if r.returncode: print(lib, r.stderr[-600:]); continue
        res[lib] = json.loads(r.stdout.strip().split("\n")[-1])
    for k in range(int(steps)):
        print(k, "  ".join(f"{lib.split('/')[-1]}: it {res[lib][k][1]:.3f} s2 {res[lib][k][2]} s4 {res[lib][k][3]} |x| {res[lib][k][4]:.5f}" for lib in res))
