"""Summary of a scripts/parity_sweep.py record: per configuration and in total, instances beyond 1e-6 between GPU and oracle, the worst distance of either side from the exact QP solution,
GPU instances beyond 1e-6 FROM EXACT (each listed), status / iteration-count equality.   usage: python scripts/summarize_parity_sweep.py gpurun_out/parity_sweep.json [more.json ...]"""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    tot = dict(solves=0, outliers=0, gpu_beyond=0, oracle_beyond=0, worst_gpu=0.0, worst_oracle=0.0, gpu_farther=0)
    print(f, "seed_offset", d.get("seed_offset"))
    for k, steps in d["configurations"].items():
        B = int(k.split("_B")[1].split("_")[0])
        o = sum(s["outliers"] for s in steps); wg = max(s["worst_d_gpu_exact"] for s in steps); wo = max(s["worst_d_oracle_exact"] for s in steps)
        adj = [a for s in steps for a in s["adjudications"] if a["kind"] == "exact"]
        gb = [(a["inst"], a["d_gpu"], a["d_oracle"]) for a in adj if a["d_gpu"] > 1e-6]; ob = [a for a in adj if a["d_oracle"] > 1e-6]
        print(f"  {k}: beyond 1e-6 GPU-vs-oracle {o} of {B * len(steps)}; worst from exact GPU {wg:.2e} oracle {wo:.2e}; GPU beyond 1e-6 from exact {len(gb)} {[(i, float('%.2g' % g)) for i, g, _ in gb]}; "
              f"status equal {min(s['status_equal'] for s in steps):.4f} iters equal {min(s['iters_equal'] for s in steps):.4f} mean iters {sum(s['mean_iters'] for s in steps) / len(steps):.3f}")
        tot["solves"] += B * len(steps); tot["outliers"] += o; tot["gpu_beyond"] += len(gb); tot["oracle_beyond"] += len(ob)
        tot["worst_gpu"] = max(tot["worst_gpu"], wg); tot["worst_oracle"] = max(tot["worst_oracle"], wo); tot["gpu_farther"] += sum(s["gpu_farther_than_oracle"] for s in steps)
    print("  TOTAL", tot)
