#!/usr/bin/env python3
"""bench.py -- MPC solves/s of the RTI hot path on MI355X (contract: see the task statement / DESIGN.md section 6).

A "step" is one closed-loop control step for the whole batch, everything resident in HBM, ONE kernel launch:
    obstacle look-ahead (a9) -> RTI solve (a10/a11) -> plant step (a14) -> obstacle motion -> warm-start shift (a12)
Episodes of 100 control steps restart from the initial scenario, so the robots are always en route.
Workload (default, BASELINE.json configs[1] = "C2"): batch = 1024 identical scenarios, N = 20, Tf = 2 s, 3 moving obstacles
(positions and velocities of the reference generator's seed-0 RANDOM draw, tests/golden/), x0 = [-6,-6,pi/4,0,0], goal [6,6].
`--workload c3` runs 65536 randomized scenarios instead.

N > 1 GPUs (launched by torch.distributed.run, one rank per GPU): the batch is replicated per rank (weak scaling, no
data-path collective); the per-scenario costs are all-gathered over RCCL on a side stream, 50 control steps per message.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6     # MI355X FP64 vector peak


def algorithmic_bytes_per_solve(N, n_obst, fused=True):
    """HBM bytes one solve must move.  Reference-style explicit P: read x0 5 + goal 2 + P (N+1)*2*n_obst +
    X 5(N+1) + U 2N, write X, U, cost (f64) + status (4 B) = 3396 B at N=20 / 3 obstacles.  The fused closed-loop step the
    bench runs is the compact-obstacle variant: it reads the 4 n_obst obstacle states instead of P and additionally writes
    back x0, the obstacle states and u0 (+ status, iters): 2640 B at N=20 / 3 obstacles."""
    if not fused:
        rd = 5 + 2 + (N + 1) * 2 * n_obst + 5 * (N + 1) + 2 * N
        wr = 5 * (N + 1) + 2 * N + 1
        return 8 * (rd + wr) + 4
    rd = 5 + 2 + 4 * n_obst + 5 * (N + 1) + 2 * N
    wr = 5 * (N + 1) + 2 * N + 1 + 5 + 4 * n_obst + 2
    return 8 * (rd + wr) + 8


PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01_split_pmc_summary.json")   # rocprofv3 passes of this very command (scripts/profile_passes.sh)


def measured_valu_instructions(kernel_name, batch):
    """VALU wave-instructions per launch of the solve kernel from the committed PMC summary (SQ_INSTS_VALU); None if it is for
    another kernel variant or batch."""
    try:
        d = json.load(open(PMC_SUMMARY))
        if d["batch"] != batch or kernel_name.replace(" ", "") not in d["kernel"].replace(" ", ""):
            return None
        return d["counters"]["SQ_INSTS_VALU"]["mean_per_launch"]
    except Exception:
        return None


def measured_traffic(kernel_name, batch):
    """HBM bytes per launch of the solve kernel from the committed rocprofv3 PMC passes (PMC_SUMMARY, collected with the
    command recorded there); None when the summary is for another kernel variant or batch."""
    try:
        d = json.load(open(PMC_SUMMARY))
        if d["batch"] != batch or kernel_name.replace(" ", "") not in d["kernel"].replace(" ", ""):
            return None
        t = d["hbm_traffic_bytes_per_launch"]
        return t["fetch_raw_kb"] * 1024 + t["write_bytes"]
    except Exception:
        return None


def algorithmic_flops_per_solve(N, n_obst, k_iters):
    """F = N c_lin + K [N (7/3 nx^3 + 4 nx^2 nu + 2 nx nu^2 + nu^3/3) + N (2 (nx+nu)^2 + 6 n_ineq)]"""
    nx, nu = 5, 2
    n_ineq = 2 * nu + 2 * 4 + 2 * n_obst
    c_lin = 200.0
    per_iter = N * (7.0 / 3 * nx ** 3 + 4 * nx ** 2 * nu + 2 * nx * nu ** 2 + nu ** 3 / 3.0) + N * (2 * (nx + nu) ** 2 + 6 * n_ineq)
    return N * c_lin + k_iters * per_iter


def make_workload(name, batch, N, n_obst, rank=0):
    if name == "c2":
        gold = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))
        obst1 = gold[f"gen_RANDOM_{n_obst}"][0]               # seed-0 RANDOM draw of the reference generator
        x0 = np.tile(np.array([-6.0, -6.0, np.pi / 4, 0.0, 0.0]), (batch, 1))
        goal = np.tile(np.array([6.0, 6.0]), (batch, 1))
        obst = np.tile(obst1[None], (batch, 1, 1))
        desc = f"C2: batch={batch} identical scenarios, N={N}, Tf={0.1 * N:g}s, {n_obst} moving obstacles"
    else:
        rng = np.random.default_rng(1234 + rank)
        x0 = np.zeros((batch, 5)); x0[:, :2] = rng.uniform(-6, 6, (batch, 2)); x0[:, 2] = rng.uniform(-np.pi, np.pi, batch)
        goal = rng.uniform(-6, 6, (batch, 2))
        obst = np.zeros((batch, n_obst, 4)); obst[:, :, :2] = rng.uniform(-4.4, 6, (batch, n_obst, 2)); obst[:, :, 2:] = rng.uniform(-2, 2, (batch, n_obst, 2))
        desc = f"C3: batch={batch} randomized start/goal/obstacle velocities, N={N}, Tf={0.1 * N:g}s, {n_obst} obstacles"
    return x0, goal, obst, desc


class Loop:
    """closed-loop state on one GPU"""

    def __init__(self, mpc_gpu, N, n_obst, batch, x0, goal, obst, dev, episode_len=100, fused=True):
        self.m = mpc_gpu.BatchedMpc(N, n_obst, 0.1 * N, max_batch=batch, device=dev.index or 0)
        self.B, self.N, self.no = batch, N, n_obst
        self.k, self.episode_len, self.fused = 0, episode_len, fused
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        z = lambda *s, dt=torch.float64: torch.zeros(*s, dtype=dt, device=dev)
        self.x0, self.goal, self.obst = t(x0), t(goal), t(obst)
        self.x0_init, self.obst_init = self.x0.clone(), self.obst.clone()
        self.x1 = z(batch, 5)
        self.P = z(batch, N + 1, n_obst, 2)
        self.X, self.U = z(batch, N + 1, 5), z(batch, N, 2)
        self.u0, self.cost = z(batch, 2), z(batch)
        self.status, self.iters = z(batch, dt=torch.int32), z(batch, dt=torch.int32)
        self.stream = torch.cuda.current_stream().cuda_stream   # the caller runs us under `with torch.cuda.stream(...)`
        assert self.stream != 0, "run under an explicit torch stream so torch ops and library kernels share one queue"
        self.m.reset_guess_dev(batch, self.x0, self.X, self.U, stream=self.stream)

    def reset(self):
        """start a new episode: initial scenario, set_initial_guess() (device-to-device copies + one small kernel)"""
        self.x0.copy_(self.x0_init); self.obst.copy_(self.obst_init)
        self.m.reset_guess_dev(self.B, self.x0, self.X, self.U, stream=self.stream)

    def step(self, cost_out=None):
        """one control step of the whole batch = ONE kernel launch (look-ahead, solve, plant, obstacles, shift fused);
        cost_out: where the kernel writes the per-scenario costs of this step (default self.cost)"""
        if self.episode_len and self.k % self.episode_len == 0 and self.k > 0:
            self.reset()
        self.k += 1
        if self.fused:
            self.m.closed_loop_step_dev(self.B, self.x0, self.obst, self.goal, self.X, self.U, self.u0,
                                        self.cost if cost_out is None else cost_out, self.status, self.iters, None, stream=self.stream)
            return
        m, B, s = self.m, self.B, self.stream
        m.predict_dev(B, self.obst, self.P, stream=s)
        m.solve_dev(B, self.x0, self.P, self.goal, self.X, self.U, self.u0, self.cost, self.status, self.iters, stream=s)
        m.plant_step_dev(B, self.x0, self.u0, self.x1, stream=s)
        self.x0.copy_(self.x1)
        m.obstacle_step_dev(B * self.no, self.obst, None, stream=s)
        m.shift_dev(B, self.X, self.U, stream=s)


def cpu_baseline(N, n_obst, x0, goal, obst, warm_steps, target_s=12.0):
    """The oracle (CPU restatement, OpenMP over instances) on a bounded sample of the same workload, host cores of this box:
    a few untimed closed-loop steps, then timed steps until ~target_s of wall time is used (solve calls only); the thread count
    (all hardware threads or half of them) that gives the higher rate is the one reported in `cores`."""
    from oracle import oracle as orc
    cfg = orc.config(N, n_obst, 0.1 * N, qp_tol=1e-8)
    ncpu = os.cpu_count() or 1
    S = min(len(x0), max(64, 16 * ncpu))
    x0, goal, obst = x0[:S].copy(), goal[:S].copy(), obst[:S].copy()
    X = np.zeros((S, N + 1, 5)); U = np.zeros((S, N, 2))
    for b in range(S):
        X[b], U[b] = orc.initial_guess(cfg, x0[b])
    dt = 0.1
    candidates = sorted({ncpu, max(1, ncpu // 2)}, reverse=True)
    acc = {n: [0, 0.0] for n in candidates}       # threads -> [solves, seconds]
    warm = min(warm_steps, 10)
    steps = 0
    t_begin = time.perf_counter()
    while True:
        P = np.stack([orc.predict_params(cfg, obst[b]) for b in range(S)])
        nthreads = candidates[steps % len(candidates)]
        t0 = time.perf_counter()
        r = orc.rti_solve_batch(cfg, x0, P, goal, X, U, nthreads=nthreads)
        t1 = time.perf_counter()
        if steps >= warm:
            acc[nthreads][0] += S; acc[nthreads][1] += t1 - t0
        X, U = r["X"], r["U"]
        for b in range(S):
            x0[b] = orc.dynamics(x0[b], r["u0"][b], dt)[0]
            for j in range(n_obst):
                obst[b, j] = orc.obstacle_step(cfg, obst[b, j], dt)
            X[b], U[b] = orc.shift(cfg, X[b], U[b])
        steps += 1
        if (time.perf_counter() - t_begin > target_s and steps >= warm + 2 * len(candidates)) or steps >= 100:
            break
    best = max(candidates, key=lambda n: acc[n][0] / acc[n][1] if acc[n][1] > 0 else 0.0)
    return {"value": acc[best][0] / acc[best][1], "unit": "solves/s", "cores": best, "kind": "port",
            "sample": f"{S} instances x {acc[best][0] // S} closed-loop steps (after {warm} untimed) of the same workload, oracle/liborc.so "
                      f"(C, f64, OpenMP {best} threads; {', '.join(f'{n} threads: {acc[n][0] / acc[n][1]:.0f}/s' for n in candidates if acc[n][1] > 0)}), "
                      f"solve time only; acados itself cannot run here"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 1024 for c2, 65536 for c3)")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--n-obst", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the supplementary C3 measurement")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # MPC_BENCH_ONE_GPU=1: smoke-test the multi-process path on a single-GPU box (all ranks on cuda:0, gloo transport)
        one_gpu = os.environ.get("MPC_BENCH_ONE_GPU") == "1"
        dev_index = 0 if one_gpu else local_rank
        torch.cuda.set_device(dev_index)
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))   # one explicit queue for torch ops AND the library's kernels

    import __graft_entry__ as g
    g.build()
    import mpc_gpu

    N, no = args.horizon, args.n_obst
    batch = args.batch or (1024 if args.workload == "c2" else 65536)
    x0, goal, obst, desc = make_workload(args.workload, batch, N, no, rank)
    loop = Loop(mpc_gpu, N, no, batch, x0, goal, obst, dev)

    # Multi-GPU: the per-scenario costs of GATHER_EVERY consecutive control steps are all-gathered in one collective (RCCL over xGMI)
    # on a side stream.  Not every step: at batch 1024 the solve kernel fills every SIMD of the chip with exactly one 512-register
    # wavefront, so any kernel running beside it (the collective's) holds back the workgroups of the CUs it occupies -- one message
    # of GATHER_EVERY x 8 KB per rank costs that once instead of GATHER_EVERY times.
    GATHER_EVERY = 50
    # (the kernel writes each step's costs straight into its row of the history; two histories alternate, so that one can be in
    # flight while the next fills)
    cost_hist = torch.zeros(2, GATHER_EVERY, batch, dtype=torch.float64, device=dev) if world > 1 else None
    gathered = torch.zeros(world, GATHER_EVERY, batch, dtype=torch.float64, device=dev) if world > 1 else None
    side = torch.cuda.Stream(device=dev) if world > 1 else None
    handle = None
    nstep = 0

    def one_step():
        nonlocal handle, nstep
        if world == 1:
            loop.step()
            return
        buf = (nstep // GATHER_EVERY) % 2
        loop.step(cost_out=cost_hist[buf, nstep % GATHER_EVERY])
        nstep += 1
        if nstep % GATHER_EVERY == 0:
            if handle is not None:
                handle.wait()        # the previous message (it left the other history two fills ago)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                handle = dist.all_gather_into_tensor(gathered.view(-1), cost_hist[buf].view(-1), async_op=True)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # HIP events around every 7th launch of the timed region (pool created here, outside it): a pair of event records between two
    # back-to-back launches costs the stream ~7 us (scripts/gap_probe.py), 4 % of a control step -- sampled (0.7 % instead), not every launch
    EVENT_EVERY = 7          # coprime with the episode length: the samples visit every position of an episode
    loop.m.profile_enable(True, every=EVENT_EVERY)
    it_acc = torch.zeros(batch, dtype=torch.int32, device=dev)   # summed inside the solve kernel (mpc_set_accumulators)
    st_acc = torch.zeros(batch, dtype=torch.int32, device=dev)
    loop.m.set_accumulators(it_acc, st_acc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    if handle is not None:
        handle.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = loop.m.profile_read()
    loop.m.profile_enable(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    total_solves = world * batch * args.steps
    value = total_solves / elapsed
    mean_iters = float(it_acc.double().sum().item()) / (batch * args.steps)
    avg_kernel_s = kern_ms / max(1, launches) * 1e-3
    abytes = algorithmic_bytes_per_solve(N, no, fused=True) * batch
    lanes, lps = loop.m.lanes_per_instance(batch), loop.m.lanes_per_stage(batch)
    # <n_obst, lanes per instance, row-parallel sweeps> / small batches: <n_obst, lanes per stage> (one instance per wavefront)
    kname = f"rti_solve_kernel<{no}, {lanes}, 2>" if lps == 1 else f"rti_split_kernel<{no}, {lps}>"
    roof = {"bound": "hbm", "achieved": abytes / avg_kernel_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": abytes / avg_kernel_s / 1e9 / HBM_PEAK_GBS, "traffic": measured_traffic(kname, batch),
            "traffic_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/{os.path.basename(PMC_SUMMARY)} (FETCH_SIZE uncorrected: 8-byte-per-lane loads, see DESIGN.md section 5)",
            "kernel": kname, "avg_launch_us": avg_kernel_s * 1e6, "launches": launches,
            "launch_timing": f"HIP events on the launch stream around every {EVENT_EVERY}th launch of the timed region ({launches} of {args.steps} launches)",
            "algorithmic_bytes_per_launch": abytes,
            "fp64_valu": {"achieved": algorithmic_flops_per_solve(N, no, mean_iters) * batch / avg_kernel_s / 1e12,
                          "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                          "frac": algorithmic_flops_per_solve(N, no, mean_iters) * batch / avg_kernel_s / 1e12 / FP64_VALU_PEAK_TF,
                          "note": "the path is bound by the ISSUE of a serial FP64 instruction stream per wavefront (the stage recursions keep 8 or fewer lanes of 64 busy), not by HBM or by the FP64 flop peak (DESIGN.md section 5); SURVEY 8(d) flop model x measured mean IPM iterations"}}
    # the roof that actually binds: vector-instruction issue slots (one wave-instruction per 4 cycles per SIMD, 4 SIMDs x 256 CUs)
    n_valu = measured_valu_instructions(kname, batch)
    if n_valu is not None:
        peak = 1024 * 2.4e9 / 4
        roof["valu_issue"] = {"achieved": n_valu / avg_kernel_s / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s",
                              "frac": n_valu / avg_kernel_s / peak,
                              "note": f"SQ_INSTS_VALU per launch (profiles/{os.path.basename(PMC_SUMMARY)}) / measured launch time; one wavefront per "
                                      "SIMD (512-register kernel), so a wavefront's own dependent instruction stream sets the rate"}
    out = {"metric": "MPC solves/sec (N=20, 3 obstacles)", "value": value, "unit": "solves/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": desc, "per_gpu_batch": batch, "N": N, "n_obst": no, "qp_tol": 1e-8, "qp_iter_max": 50,
                      "step": "one fused launch: obstacle look-ahead + RTI solve + plant step + obstacle motion + warm-start shift, device resident; episodes of 100 control steps",
                      "parallelism": f"replicas x{world}, cost all-gather (RCCL, 50 control steps per message)" if world > 1 else "single GPU"},
           "mean_ipm_iters": mean_iters, "qp_failure_frac": float((st_acc % 65536).double().sum().item()) / (batch * args.steps),
           "qp_iter_cap_frac": float((st_acc // 65536).double().sum().item()) / (batch * args.steps),
           "lanes_per_instance": lanes, "lanes_per_stage": lps, "roofline": roof}

    if rank == 0 and world == 1 and not args.no_extra and args.workload == "c2":
        # supplementary: the large-batch configuration (configs[2]) on the same GPU
        xb, gb, ob, d3 = make_workload("c3", 65536, N, no)
        l3 = Loop(mpc_gpu, N, no, 65536, xb, gb, ob, dev)
        for _ in range(10):
            l3.step()
        it3 = torch.zeros(65536, dtype=torch.int32, device=dev)
        l3.m.set_accumulators(it3, None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(30):
            l3.step()
        torch.cuda.synchronize()
        e3 = time.perf_counter() - t1
        out["extra"] = {"workload": d3, "value": 65536 * 30 / e3, "unit": "solves/s", "ms_per_step": e3 / 30 * 1e3, "steps": 30, "warmup": 10,
                        "mean_ipm_iters": float(it3.double().sum().item()) / (65536 * 30)}
        del l3
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(N, no, x0, goal, obst, warm_steps=args.warmup)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
