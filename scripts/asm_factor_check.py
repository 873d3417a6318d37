"""Diagnostic: the one-block asm factor sweep on the compact LDS stage blocks (MPC_FACTOR_ASM_C) against the plain C++ sweep over the same
blocks (-DMPC_COMPACT_PLAIN build): bit-for-bit the same trajectories, iteration counts and costs, then the control-step time of both.
usage (GPU box): python scripts/asm_factor_check.py            (spawns itself once per library)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")
CASES = [(20, 3, 21, 6000, "c3"), (17, 3, 21, 3000, "c3"), (20, 5, 21, 3000, "c3"), (50, 10, 64, 1500, "c5"), (33, 3, 64, 1500, "c3"), (20, 10, 21, 1500, "c5")]

def child(out):
    sys.path[:0] = [ROOT, PKG]
    import numpy as np, torch
    import mpc_gpu, bench
    from mpc_gpu.sharding import shard_slice
    dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    rng = np.random.default_rng(99)
    res = {}
    for N, no, G, B, _ in CASES:
        x0 = np.zeros((B, 5)); x0[:, :2] = rng.uniform(-6, 6, (B, 2)); x0[:, 2] = rng.uniform(-np.pi, np.pi, B)
        goal = rng.uniform(-6, 6, (B, 2))
        obst = np.zeros((B, no, 4)); obst[:, :, :2] = rng.uniform(-4.4, 6, (B, no, 2)); obst[:, :, 2:] = rng.uniform(-2, 2, (B, no, 2))
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
            s.set_lanes_per_stage(1); s.set_lanes_per_instance(G)
            name = s.kernel_name(B)
            s.reset_guess(x0)
            for k in range(4):
                g = s.solve(x0, obst, goal); X, U = s.get_traj(B); s.shift(B)
                res[f"{N}_{no}_{G}_{k}_X"] = X.copy(); res[f"{N}_{no}_{G}_{k}_U"] = U.copy()
                res[f"{N}_{no}_{G}_{k}_it"] = np.asarray(g["iters"]).copy(); res[f"{N}_{no}_{G}_{k}_c"] = np.asarray(g["cost"]).copy()
        print(name, flush=True)
    # control-step time on the bench workloads
    times = {}
    for wl, B in (("c3", 65536), ("c5", 32768)):
        x0, goal, obst, desc, _, _ = bench.make_workload(wl, 1, 0, shard_slice)
        N, no = bench.WORKLOADS[wl][:2]
        loop = bench.Loop(mpc_gpu, torch, N, no, x0[:B], goal[:B], obst[:B], dev)
        loop.reset()
        for _ in range(20): loop.control_step()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(40): loop.control_step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 40
        times[wl] = {"kernel": loop.m.kernel_name(B), "ms_per_step": dt * 1e3, "solves_per_s": B / dt}
        print(wl, times[wl], flush=True)
    np.savez(out, **res); json.dump(times, open(out + ".json", "w"))

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child": child(sys.argv[2]); sys.exit(0)
    import numpy as np
    od = os.path.join(ROOT, "gpurun_out"); os.makedirs(od, exist_ok=True)
    variant = os.path.join(ROOT, "build", "libmpcgpu_plain.so")          # the comparison build: compiled here when it is not there yet
    if not os.path.exists(variant) or "--rebuild" in sys.argv:
        os.makedirs(os.path.dirname(variant), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-DMPC_COMPACT_PLAIN",
                               "-o", variant, os.path.join(PKG, "csrc", "mpc_api.hip")])
    outs = {}
    for tag, lib in (("asm", None), ("plain", os.path.join(ROOT, "build", "libmpcgpu_plain.so"))):
        env = dict(os.environ); 
        if lib: env["MPC_GPU_LIB"] = lib
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", os.path.join(od, f"asmchk_{tag}")], env=env)
        outs[tag] = (np.load(os.path.join(od, f"asmchk_{tag}.npz")), json.load(open(os.path.join(od, f"asmchk_{tag}.json"))))
    a, b = outs["asm"][0], outs["plain"][0]
    bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
    for tag in outs: os.remove(os.path.join(od, f"asmchk_{tag}.npz")); os.remove(os.path.join(od, f"asmchk_{tag}.json"))     # gpurun_out/ returns at most 64 MiB
    summary = {"arrays": len(a.files), "not_bitwise_equal": bad, "asm": outs["asm"][1], "plain": outs["plain"][1]}
    if bad:
        for k in bad[:8]: print(k, np.nanmax(np.abs(a[k] - b[k])))
    print(json.dumps(summary, indent=1))
    json.dump(summary, open(os.path.join(od, "asm_factor_check.json"), "w"), indent=1)
    sys.exit(1 if bad else 0)
