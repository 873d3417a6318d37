"""Diagnostic: the factor sweep on 4x4x4 matrix-core blocks (-DMPC_MFMA4 build, csrc/rti_kernel.hpp::mfma4_factor) against the default
row-parallel DPP sweep: agreement of trajectories / iteration counts over closed-loop steps, then control-step time on the C2 workload and
on one instance per wavefront at N = 50.   usage (GPU box): python scripts/mfma4_check.py      (spawns itself once per library)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")
CASES = [(20, 3, 0, 1000), (20, 3, 64, 500), (31, 3, 0, 300), (20, 5, 0, 300), (50, 3, 64, 200), (9, 3, 0, 100)]      # N, n_obst, lanes/instance (0: split), batch

def child(out):
    sys.path[:0] = [ROOT, PKG]
    import numpy as np, torch
    import mpc_gpu, bench
    from mpc_gpu.sharding import shard_slice
    dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    rng = np.random.default_rng(99)
    res = {}
    for N, no, G, B in CASES:
        x0 = np.zeros((B, 5)); x0[:, :2] = rng.uniform(-6, 6, (B, 2)); x0[:, 2] = rng.uniform(-np.pi, np.pi, B)
        goal = rng.uniform(-6, 6, (B, 2))
        obst = np.zeros((B, no, 4)); obst[:, :, :2] = rng.uniform(-4.4, 6, (B, no, 2)); obst[:, :, 2:] = rng.uniform(-2, 2, (B, no, 2))
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
            if G: s.set_lanes_per_stage(1); s.set_lanes_per_instance(G)
            else: s.set_lanes_per_stage(3 if N <= 20 else 2); s.set_waves_per_simd(1)
            name = s.kernel_name(B)
            s.reset_guess(x0)
            for k in range(4):
                g = s.solve(x0, obst, goal); X, U = s.get_traj(B); s.shift(B)
                res[f"{N}_{no}_{G}_{k}_X"] = X.copy(); res[f"{N}_{no}_{G}_{k}_it"] = np.asarray(g["iters"]).copy(); res[f"{N}_{no}_{G}_{k}_st"] = np.asarray(g["status"]).copy()
        print(name, flush=True)
    times = {}
    x0, goal, obst, desc, _, _ = bench.make_workload("c2", 1, 0, shard_slice)
    loop = bench.Loop(mpc_gpu, torch, 20, 3, x0, goal, obst, dev)
    for rep in range(2):
        loop.reset(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(100): loop.control_step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 100
    times["c2"] = {"kernel": loop.m.kernel_name(1024), "us_per_control_step": dt * 1e6, "solves_per_s": 1024 / dt, "mean_iters": float(loop.iters.double().mean())}
    print(times["c2"], flush=True)
    np.savez(out, **res); json.dump(times, open(out + ".json", "w"))

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child": child(sys.argv[2]); sys.exit(0)
    import numpy as np
    od = os.path.join(ROOT, "gpurun_out"); os.makedirs(od, exist_ok=True)
    variant = os.path.join(ROOT, "build", "libmpcgpu_mfma4.so")          # the comparison build: compiled here when it is not there yet
    if not os.path.exists(variant) or "--rebuild" in sys.argv:
        os.makedirs(os.path.dirname(variant), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-DMPC_MFMA4",
                               "-o", variant, os.path.join(PKG, "csrc", "mpc_api.hip")])
    outs = {}
    for tag, lib in (("dpp", None), ("mfma4", os.path.join(ROOT, "build", "libmpcgpu_mfma4.so"))):
        env = dict(os.environ)
        if lib: env["MPC_GPU_LIB"] = lib
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", os.path.join(od, f"m4chk_{tag}")], env=env)
        outs[tag] = (dict(np.load(os.path.join(od, f"m4chk_{tag}.npz"))), json.load(open(os.path.join(od, f"m4chk_{tag}.json"))))
        os.remove(os.path.join(od, f"m4chk_{tag}.npz")); os.remove(os.path.join(od, f"m4chk_{tag}.json"))
    a, b = outs["dpp"][0], outs["mfma4"][0]
    rows = {}
    for N, no, G, B in CASES:
        for k in range(4):
            key = f"{N}_{no}_{G}_{k}"
            ok = (a[key + "_st"] == 0) & (b[key + "_st"] == 0)
            d = np.abs(a[key + "_X"] - b[key + "_X"]).reshape(B, -1).max(1)
            rows[key] = {"status_equal": float((a[key + "_st"] == b[key + "_st"]).mean()), "iters_equal": float((a[key + "_it"] == b[key + "_it"])[ok].mean()),
                         "dX_median": float(np.median(d[ok])), "dX_p99": float(np.quantile(d[ok], 0.99)), "dX_max": float(d[ok].max()), "nan": int(np.isnan(b[key + "_X"]).any(axis=(1, 2)).sum())}
            print(key, rows[key])
    summary = {"agreement": rows, "dpp": outs["dpp"][1], "mfma4": outs["mfma4"][1]}
    print(json.dumps({k: summary[k] for k in ("dpp", "mfma4")}, indent=1))
    json.dump(summary, open(os.path.join(od, "mfma4_check.json"), "w"), indent=1)
