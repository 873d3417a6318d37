"""thr0 0.1 vs 0.3 on the other problem classes: C2 (value), N = 20 with 5 and with 10 obstacles at 65536 / 16384, and the replay of the ten recorded tables (rows reproduced)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts")]
import numpy as np, torch, bench, mpc_gpu
from mpc_gpu.sharding import shard_slice
from mpc_gpu.world import reference_streams
from unmatched_rows import matches
dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
out = {}
x0, goal, obst, _, _, _ = bench.make_workload("c2", 1, 0, shard_slice)
for thr0 in (0.1, 0.2, 0.3):
    loop = bench.Loop(mpc_gpu, torch, 20, 3, x0, goal, obst, dev, thr0=thr0)
    r = bench.measure(torch, None, loop, 1, None, 5, 1, dev)
    print("c2", thr0, 1024 * 100 * 5 / r["elapsed"], r["mean_iters"], r["fail"], r["cap"], flush=True); del loop
rng = np.random.default_rng(1234)
for no, B in ((5, 65536), (10, 16384)):
    xx = np.zeros((B, 5)); xx[:, :2] = rng.uniform(-6, 6, (B, 2)); xx[:, 2] = rng.uniform(-np.pi, np.pi, B); gg = rng.uniform(-6, 6, (B, 2))
    oo = np.zeros((B, no, 4)); oo[:, :, :2] = rng.uniform(-4.4, 6, (B, no, 2)); oo[:, :, 2:] = rng.uniform(-2, 2, (B, no, 2))
    for thr0 in (0.1, 0.2, 0.3):
        loop = bench.Loop(mpc_gpu, torch, 20, no, xx, gg, oo, dev, streams=2, thr0=thr0)
        r = bench.measure(torch, None, loop, 1, None, 2, 1, dev)
        print("N20", no, B, thr0, B * 100 * 2 / r["elapsed"], r["mean_iters"], r["fail"], r["cap"], flush=True); del loop
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")))["tables"]
x0 = np.tile([-7.0, -7.0, np.pi / 4, 0, 0], (100, 1)); goal = np.tile([7.0, 7.0], (100, 1))
streams = {s: reference_streams(s, range(100), 5, 400) for s in ("RANDOM", "EDGE")}
for thr0 in (0.1, 0.2, 0.3):
    tot3 = tot6 = 0
    for stem, t in ref.items():
        sp = t["spec"]; interp = bool(sp.get("interpolate_init")); ob, nz = streams[sp["scenario"]]; rows = np.array(t["rows"])
        best = (0, 0)
        for alias in ((True, False) if interp else (True,)):
            r = mpc_gpu.run_episodes(x0, goal, ob, N=sp["N_SOLV"], Tf=float(sp["TF"]), max_iter=400, random_move=True, init_guess_when_error=True, noise=nz,
                                     qp_iter_max=sp["QP_ITER"], bug_compat_alias=alias, interpolate_init=interp, thr0=thr0)
            m3, m6 = matches(r["table"], rows)
            if m3.sum() > best[0]: best = (int(m3.sum()), int(m6.sum()))
        tot3 += best[0]; tot6 += best[1]
    print("replay thr0", thr0, "rows reproduced", tot3, "/", tot6, flush=True)
