"""How much of a wavefront's time is lost to instances that share it but stop at different interior-point iterations, and how much an
ordering hint (instances sorted by the iteration count of their previous control step) would recover: C3 workload, closed loop.
usage (GPU box): python scripts/iters_order_probe.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np, torch
import mpc_gpu, bench
from mpc_gpu.sharding import shard_slice
dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
x0, goal, obst, desc, _, G = bench.make_workload("c3", 1, 0, shard_slice)
loop = bench.Loop(mpc_gpu, torch, 20, 3, x0, goal, obst, dev)
loop.reset()
hist = []
for k in range(100):
    loop.control_step(); torch.cuda.synchronize()
    hist.append(loop.iters.cpu().numpy().copy())
hist = np.array(hist)
# alternative predictors of an instance's next iteration count (three per wavefront)
for name, pred in (("previous", lambda k: hist[k - 1].astype(float)), ("max of last two", lambda k: np.maximum(hist[k - 1], hist[k - 2]).astype(float)),
                   ("sum of last two", lambda k: (hist[k - 1] + hist[k - 2]).astype(float)), ("0.7 prev + 0.3 prevprev", lambda k: 0.7 * hist[k - 1] + 0.3 * hist[k - 2]),
                   ("mean of last four", lambda k: hist[k - 4:k].mean(0))):
    vals = []
    for k in range(4, 100):
        it = hist[k].astype(float); n = (len(it) // 3) * 3
        vals.append(it[np.argsort(-pred(k), kind="stable")][:n].reshape(-1, 3).max(1).mean())
    print("predictor", name, float(np.mean(vals)))
out = []
for per in (2, 3):
    nat, srt, ideal = [], [], []
    for k in range(1, 100):
        it = hist[k].astype(float)
        n = (len(it) // per) * per
        nat.append(it[:n].reshape(-1, per).max(1).mean())
        order = np.argsort(hist[k - 1], kind="stable")
        srt.append(it[order][:n].reshape(-1, per).max(1).mean())
        ideal.append(np.sort(it)[:n].reshape(-1, per).max(1).mean())
    out.append(dict(instances_per_wavefront=per, mean_iters=float(hist[1:].mean()), mean_of_wave_max_natural_order=float(np.mean(nat)),
                    sorted_by_previous_step=float(np.mean(srt)), sorted_by_own_count_ideal=float(np.mean(ideal))))
    print(out[-1])
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "iters_order_probe.json"), "w"), indent=1)
