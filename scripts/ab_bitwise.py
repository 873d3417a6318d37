"""Diagnostic: two builds of the library must produce bit-identical trajectories, iteration counts, statuses and costs (for changes that touch
scheduling or synchronisation but not arithmetic).  usage (GPU box): python scripts/ab_bitwise.py build/lib_head.so default"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")
CASES = [(20, 3, 0, 0, 1000), (20, 3, 0, 2, 5000), (20, 3, 21, 0, 3000), (20, 3, 32, 0, 1000), (10, 5, 16, 0, 1000), (31, 3, 0, 0, 500), (20, 10, 0, 0, 300),
         (50, 10, 64, 0, 200), (40, 3, 64, 0, 200), (20, 4, 0, 0, 300), (45, 7, 64, 0, 100), (20, 5, 21, 0, 600)]      # N, n_obst, lanes/instance (0: split), waves, batch

def child(out):
    sys.path[:0] = [ROOT, PKG]
    import numpy as np
    import mpc_gpu
    rng = np.random.default_rng(7)
    res = {}
    for N, no, G, W, B in CASES:
        x0 = np.zeros((B, 5)); x0[:, :2] = rng.uniform(-6, 6, (B, 2)); x0[:, 2] = rng.uniform(-np.pi, np.pi, B)
        goal = rng.uniform(-6, 6, (B, 2))
        obst = np.zeros((B, no, 4)); obst[:, :, :2] = rng.uniform(-4.4, 6, (B, no, 2)); obst[:, :, 2:] = rng.uniform(-2, 2, (B, no, 2))
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
            if G: s.set_lanes_per_stage(1); s.set_lanes_per_instance(G)
            if W: s.set_waves_per_simd(W)
            key = f"{s.kernel_name(B)}_{N}_{no}"
            s.reset_guess(x0)
            for k in range(4):
                g = s.solve(x0, obst, goal); X, U = s.get_traj(B); s.shift(B)
                res[f"{key}_{k}_X"] = X.copy(); res[f"{key}_{k}_U"] = U.copy(); res[f"{key}_{k}_c"] = np.asarray(g["cost"]).copy()
                res[f"{key}_{k}_it"] = np.asarray(g["iters"]).copy(); res[f"{key}_{k}_st"] = np.asarray(g["status"]).copy()
        print(key, flush=True)
    np.savez(out, **res)

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child": child(sys.argv[2]); sys.exit(0)
    import numpy as np
    libs = sys.argv[1:3]
    outs = []
    for i, lib in enumerate(libs):
        env = dict(os.environ)
        if lib != "default": env["MPC_GPU_LIB"] = os.path.join(ROOT, lib)
        out = os.path.join("/tmp", f"abbit_{i}.npz")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", out], env=env, stdout=subprocess.DEVNULL if i else None)
        outs.append(np.load(out))
    a, b = outs
    bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
    print(f"{len(a.files)} arrays compared, not bit-identical: {bad[:10]}")
    sys.exit(1 if bad else 0)
