"""Two wavefronts per SIMD, forced: what happens when the register allocator is told to fit the stage-split solve
kernel into 256 registers per lane (__launch_bounds__(64, 2), -DMPC_FORCE_WAVES2) on the large-batch workload C3 (65536 randomized
scenarios, steady closed loop), against the product build on the same mapping and against the product's own choice for that batch.
Writes gpurun_out/waves2_experiment_<tag>.json (resource usage from the compiler remarks + measured ms per control step).
usage (GPU box): python scripts/waves2_experiment.py [tag]"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
CSRC = os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd", "csrc")

RUN = r'''
import json, os, sys, time
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "dynamic-obstacle-avoidance-mpc_amd")]
import numpy as np, torch
import mpc_gpu, bench
from mpc_gpu.sharding import shard_slice
mpc_gpu.BatchedMpc.default_lanes_per_stage = %(lps)d
dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
x0, goal, obst, desc, _, G = bench.make_workload("c3", 1, 0, shard_slice)
loop = bench.Loop(mpc_gpu, torch, 20, 3, x0, goal, obst, dev)
r = bench.measure(torch, None, loop, 1, None, 2, 1, dev)
print("RESULT " + json.dumps(dict(ms_per_control_step=r["elapsed"] / 200 * 1e3, solves_per_s=G * 200 / r["elapsed"], mean_ipm_iters=r["mean_iters"],
                                  kernel_us=r["kern_ms"] / max(1, r["launches"]) * 1e3, lanes_per_stage=loop.m.lanes_per_stage(G), lanes_per_instance=loop.m.lanes_per_instance(G))))
'''


def resources(flags):
    so = os.path.join(OUT, "libmpcgpu_" + ("waves2" if flags else "plain") + ".so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", *flags,
                        "-Rpass-analysis=kernel-resource-usage", "-o", so, os.path.join(CSRC, "mpc_api.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    res, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1); res[cur] = {}
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch_bytes_per_lane", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("waves_per_simd", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds_static", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur:
                res[cur][key] = int(m.group(1))
    return so, {k: v for k, v in res.items() if "rti_split_kernelILi3ELi3" in k or "rti_solve_kernelILi3ELi32ELi2" in k}


def run(so, lps):
    env = dict(os.environ, MPC_GPU_LIB=so)
    r = subprocess.run([sys.executable, "-c", RUN % dict(root=ROOT, lps=lps)], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(line[-1][7:])


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    os.makedirs(OUT, exist_ok=True)
    plain, res_plain = resources([])
    forced, res_forced = resources(["-DMPC_FORCE_WAVES2"])
    sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")]
    out = dict(workload="C3: 65536 randomized scenarios, N=20, 3 obstacles, episodes of 100 control steps (1 untimed + 2 timed)",
               lds_bytes_per_wavefront_split_kernel="(RowLds::total(20,1) + 21*65 + 21*3*2) * 8 = 31.1 KB -> 5 wavefronts per CU by LDS (160 KB), whatever the register count",
               product_choice=dict(resources=res_plain, run=run(plain, 0)),
               split_kernel_512_registers=dict(resources=res_plain, run=run(plain, 3)),
               split_kernel_forced_256_registers=dict(resources=res_forced, run=run(forced, 3)))
    json.dump(out, open(os.path.join(OUT, f"waves2_experiment_{tag}.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
