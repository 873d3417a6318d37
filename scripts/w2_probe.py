"""Stage-split kernel with two wavefronts per SIMD (compact LDS blocks, 256 registers) against the one-wavefront variant and the
one-lane-per-stage mapping: bitwise agreement of the two split variants, then control-step time over a range of batch sizes on the
randomized C3 workload.  usage (GPU box): python scripts/w2_probe.py [n_obst] [N]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np, torch
import mpc_gpu, bench
from mpc_gpu.sharding import shard_slice

no = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
rng = np.random.default_rng(1234)
G = 65536
x0 = np.zeros((G, 5)); x0[:, :2] = rng.uniform(-6, 6, (G, 2)); x0[:, 2] = rng.uniform(-np.pi, np.pi, G)
goal = rng.uniform(-6, 6, (G, 2))
obst = np.zeros((G, no, 4)); obst[:, :, :2] = rng.uniform(-4.4, 6, (G, no, 2)); obst[:, :, 2:] = rng.uniform(-2, 2, (G, no, 2))

# bitwise agreement of the two split variants over 3 control steps
B = 3000
res = {}
for w in (1, 2):
    with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
        s.set_lanes_per_stage(3 if N <= 20 else 2); s.set_waves_per_simd(w)
        assert s.waves_per_simd(B) == w
        s.reset_guess(x0[:B]); outs = []
        for k in range(3):
            g = s.solve(x0[:B], obst[:B], goal[:B]); X, U = s.get_traj(B); s.shift(B)
            outs.append((g, X, U))
        res[w] = outs
for k in range(3):
    a, b = res[1][k], res[2][k]
    same = np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[0]["iters"], b[0]["iters"]) and np.array_equal(a[0]["cost"], b[0]["cost"])
    print("step", k, "bitwise equal:", same, "max |dX|", np.abs(a[1] - b[1]).max(), flush=True)

# three instances per wavefront (G = 21) against two (G = 32): same arithmetic except the order of the wavefront reductions
if N <= 20:
    res = {}
    for G in (32 if N + 2 > 16 else 16, 21):
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
            s.set_lanes_per_stage(1); s.set_lanes_per_instance(G)
            s.reset_guess(x0[:B]); outs = []
            for k in range(3):
                g = s.solve(x0[:B], obst[:B], goal[:B]); X, U = s.get_traj(B); s.shift(B)
                outs.append((g, X, U))
            res[G] = outs
    Gs = sorted(res)
    for k in range(3):
        a, b = res[Gs[0]][k], res[Gs[1]][k]
        ok = (a[0]["status"] == 0) & (b[0]["status"] == 0)
        print("G=21 vs", Gs[1] if Gs[0] == 21 else Gs[0], "step", k, "status equal:", float((a[0]["status"] == b[0]["status"]).mean()), "iters equal:", float((a[0]["iters"] == b[0]["iters"]).mean()),
              "max |dX| (both converged):", float(np.abs(a[1] - b[1])[ok].max()), "median:", float(np.median(np.abs(a[1] - b[1])[ok].reshape(ok.sum(), -1).max(1))), flush=True)

out = []
for Bn in (1024, 2048, 4096, 8192, 16384, 65536):
    row = dict(batch=Bn)
    for name, lps, w in (("split_w1", 3 if N <= 20 else 2, 1), ("split_w2", 3 if N <= 20 else 2, 2), ("one_lane", 1, 1)) + ((("three_per_wave", 1, 21),) if N <= 20 else ()):
        mpc_gpu.BatchedMpc.default_lanes_per_stage = lps; mpc_gpu.BatchedMpc.default_waves_per_simd = w if w != 21 else 0
        loop = bench.Loop(mpc_gpu, torch, N, no, x0[:Bn], goal[:Bn], obst[:Bn], dev)
        if w == 21:
            loop.m.set_lanes_per_stage(1); loop.m.set_lanes_per_instance(21)
            assert loop.m.lanes_per_instance(Bn) == 21
        r = bench.measure(torch, None, loop, 1, None, 2, 1, dev)
        row[name] = dict(ms=r["elapsed"] / 200 * 1e3, Msolves=Bn * 200 / r["elapsed"] / 1e6, iters=r["mean_iters"])
        loop.m.close(); del loop
    print(json.dumps(row), flush=True); out.append(row)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"w2_probe_N{N}_no{no}.json"), "w"), indent=1)
