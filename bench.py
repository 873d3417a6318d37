#!/usr/bin/env python3
"""bench.py -- MPC solves/s of the RTI hot path on MI355X (contract: the task statement / DESIGN.md section 6).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5]

One "step" = one EPISODE of the whole per-GPU batch: set_initial_guess() from the initial scenario, then 100 closed-loop control steps,
each control step ONE kernel launch with everything resident in HBM:
    obstacle look-ahead (a9) -> RTI solve (a10/a11) -> plant step (a14) -> obstacle motion -> warm-start shift (a12)
("100 steps" of the scenario.)  Every episode does exactly the same work, so `value` does not depend on --steps:
W untimed episodes, then exactly K timed ones between barrier + synchronize, value = global_batch * 100 * K / max-over-ranks time.

Workloads (BASELINE.json configs):
  c2 (default, configs[1], the one the metric is quoted on): 1024 identical scenarios per GPU, N = 20, Tf = 2 s, 3 moving obstacles
     (the reference generator's seed-0 RANDOM draw), x0 = [-6,-6,pi/4,0,0], goal [6,6]; N GPUs: 1024 per rank (weak scaling)
  c3 (configs[2]): 65536 randomized scenarios per GPU (weak)
  c4 (configs[3]): ONE global batch of 262144 randomized scenarios (seed 1234), rank r solves the contiguous slice
     mpc_gpu.sharding.shard_slice(262144, r, world) -- 32768 per GPU on 8 GPUs (strong scaling)
  c5 (configs[4]): N = 50, 10 obstacles, one global batch of 32768, sharded the same way (4096 per GPU on 8 GPUs)
Multi-GPU: one process per GPU, the SAME workload for every N unless --workload says otherwise (default c2: 1024 per GPU, weak -- one curve for the driver's 1 / 2 / 4 / 8 runs);
`--gpus N` without a torchrun environment starts the N ranks itself (torch.distributed.run, before
anything touches the GPU) and relays rank 0's JSON line.  No data-path collective; the per-scenario costs of 50 consecutive control
steps travel in one all-gather on a side stream -- through the library's own C-ABI collective (--exchange capi, the default: rank 0's
mpc_comm_unique_id is broadcast over the torch.distributed group that the launcher set up anyway, every rank calls mpc_comm_init, and each
message is one mpc_allgather_cost_dev = RCCL ncclAllGather over xGMI, issued by libmpcgpu itself) or through torch.distributed
(--exchange torch: mpc_gpu.sharding.gather_costs).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TF = 78.6     # MI355X FP64 vector peak.  MI355X_MICROARCH.md tabulates the FP32 vector peak only (157.3 TFLOP/s = 256 CUs x 4 SIMDs x 16 lanes
                             # x 2 flop x 2 (packed) x 2.4 GHz); the FP64 row it lacks is AMD's public specification for the part, 78.6 TFLOP/s vector AND matrix
                             # (= the same product without packing: one wave64 FP64 FMA occupies a SIMD for 4 cycles)
SIMDS = 1024                 # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4              # nominal engine clock the cycle figures are converted with
LONE_WAVE_VALU_CYCLES = 5.0  # what ONE wavefront per SIMD pays per independent FP64 VALU instruction (8.4 when it depends on the previous one) and per LDS instruction
LONE_WAVE_LDS_CYCLES = 14.0  # (issue cost, whatever the width): micro-benchmarks scripts/bin_src/dpp64_test.hip, lds_issue_test.hip (docs/HISTORY.md section 4.1c)
VALU_CYCLES_PER_INST = 4     # a wave64 VALU instruction holds its 16-lane SIMD for 4 cycles: the issue roof of one SIMD is 1 instruction per 4 cycles
EPISODE = 100                # control steps per episode = per bench step
GATHER_EVERY = 50            # control steps per cost all-gather message
PMC_STALE_TOL = 0.05         # a committed PMC profile speaks for this run only while the run's launch time is within 5 % of the profile's
EVENT_EVERY = 7              # HIP events around every 7th launch (coprime with EPISODE: the samples visit every position of an episode)

WORKLOADS = {   # name: (N, n_obst, per-GPU batch or None, global batch or None, scaling)
    "c2": (20, 3, 1024, None, "weak"),
    "c3": (20, 3, 65536, None, "weak"),
    "c4": (20, 3, None, 262144, "strong"),
    "c5": (50, 10, None, 32768, "strong"),
}


def algorithmic_bytes_per_solve(N, n_obst):
    """HBM bytes one fused control step must move per instance (the compact-obstacle variant of SURVEY.md 8(d)):
    read x0 5 + goal 2 + obstacle states 4 n_obst + X 5(N+1) + U 2N, write X, U, cost, x0 5, obstacle states 4 n_obst, u0 2 (f64)
    + status, iters (4 B each): 2640 B at N = 20 / 3 obstacles, 6448 B at N = 50 / 10 obstacles.  (Reference-style explicit P: 3396 / 13908 B.)"""
    rd = 5 + 2 + 4 * n_obst + 5 * (N + 1) + 2 * N
    wr = 5 * (N + 1) + 2 * N + 1 + 5 + 4 * n_obst + 2
    return 8 * (rd + wr) + 8


def algorithmic_flops_per_solve(N, n_obst, k_iters):
    """F = N c_lin + K [N (7/3 nx^3 + 4 nx^2 nu + 2 nx nu^2 + nu^3/3) + N (2 (nx+nu)^2 + 6 n_ineq)], K = measured mean
    interior-point iterations: 20*200 + K * 14807 at N = 20 / 3 obstacles."""
    nx, nu = 5, 2
    n_ineq = 2 * nu + 2 * 4 + 2 * n_obst
    per_iter = N * (7.0 / 3 * nx ** 3 + 4 * nx ** 2 * nu + 2 * nx * nu ** 2 + nu ** 3 / 3.0) + N * (2 * (nx + nu) ** 2 + 6 * n_ineq)
    return N * 200.0 + k_iters * per_iter


def lanes_useful(kernel_name, N, n_obst):
    """Lanes of a wavefront that carry data, averaged over the VALU instructions of one interior-point iteration (see roofline.lanes_note).
    Per iteration and wavefront: row phases R instructions on `stage_lanes` lanes, factor sweep F on 8 lanes per instance, vector sweeps V on 5 lanes per instance."""
    k = kernel_name.replace(" ", "")
    if k.startswith("rti_split_kernel<"):            # one instance per wavefront, LPS lanes per stage; per-stage counts from profiles/r01_split_phase_timing.txt
        lps = int(k.split("<")[1].split(",")[1])
        R, F, V, inst, stage_lanes = 830.0, 112.0 * N, 3 * 13.0 * N, 1, min(64, lps * (N + 1))
    elif k.startswith("rti_solve_kernel<"):          # one lane per stage; G lanes per instance; profiles/r03_solve_*_phase_instruction_counts.txt
        g = int(k.split("<")[1].split(",")[1])
        inst = 3 if g == 21 else 64 // g
        R = 2910.0 if n_obst <= 5 else 7000.0
        F, V, stage_lanes = 125.0 * N, 3 * 15.0 * N, inst * (N + 1)
    else:
        return None
    return (R * stage_lanes + F * 8 * inst + V * 5 * inst) / (R + F + V)


def measured_pmc(kernel_name, batch):
    """What the newest committed rocprofv3 PMC summary for this kernel and batch holds (profiles/*_pmc_summary.json, scripts/profile_passes.sh):
    HBM bytes per launch (a launch moves the same bytes whatever the iteration count), VALU instructions per launch with the kernel duration of
    that profile (-> the issue-slot fraction), and the mean number of active lanes per VALU instruction; else None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            d = json.load(open(f))
            if d["batch"] == batch and kernel_name.replace(" ", "") in d["kernel"].replace(" ", ""):
                t = d["hbm_traffic_bytes_per_launch"]
                c = d["counters"]
                best = dict(traffic=t["fetch_raw_kb"] * 1024 + t["write_bytes"], file=os.path.basename(f), kernel=d["kernel"],
                            valu_insts=c.get("SQ_INSTS_VALU", {}).get("mean_per_launch"), avg_ns=(d.get("kernel_stats") or {}).get("avg_ns"),
                            lds_insts=c.get("SQ_INSTS_LDS", {}).get("mean_per_launch"), wave_cycles=c.get("SQ_WAVE_CYCLES", {}).get("mean_per_launch"),
                            wait_any=c.get("SQ_WAIT_ANY", {}).get("mean_per_launch"),
                            lanes_active=(d.get("derived") or {}).get("valu_lanes_active"))
        except Exception:
            pass
    return best


def make_workload(name, world, rank, shard_slice):
    """This rank's slice of the workload's GLOBAL batch.  Returns x0, goal, obst, description, (lo, hi), global size."""
    N, n_obst, per_gpu, total, _ = WORKLOADS[name]
    G = total if total else per_gpu * world
    lo, hi = shard_slice(G, rank, world)
    if name == "c2":
        gold = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))
        obst1 = gold[f"gen_RANDOM_{n_obst}"][0]               # seed-0 RANDOM draw of the reference generator
        x0 = np.tile(np.array([-6.0, -6.0, np.pi / 4, 0.0, 0.0]), (hi - lo, 1))
        goal = np.tile(np.array([6.0, 6.0]), (hi - lo, 1))
        obst = np.tile(obst1[None], (hi - lo, 1, 1))
        desc = f"C2: {per_gpu} identical scenarios per GPU, N={N}, Tf={0.1 * N:g}s, {n_obst} moving obstacles"
    else:
        rng = np.random.default_rng(1234)                      # ONE global stream; every rank draws it and keeps its slice
        x0 = np.zeros((G, 5)); x0[:, :2] = rng.uniform(-6, 6, (G, 2)); x0[:, 2] = rng.uniform(-np.pi, np.pi, G)
        goal = rng.uniform(-6, 6, (G, 2))
        obst = np.zeros((G, n_obst, 4)); obst[:, :, :2] = rng.uniform(-4.4, 6, (G, n_obst, 2)); obst[:, :, 2:] = rng.uniform(-2, 2, (G, n_obst, 2))
        x0, goal, obst = x0[lo:hi].copy(), goal[lo:hi].copy(), obst[lo:hi].copy()
        desc = (f"{name.upper()}: global batch {G} randomized start/goal/obstacle velocities (seed 1234), "
                f"{'sharded' if total else 'per-GPU ' + str(per_gpu)}, N={N}, Tf={0.1 * N:g}s, {n_obst} obstacles")
    return x0, goal, obst, desc, (lo, hi), G


def pick_streams(batch):
    """sub-batches pipelined on separate streams (mpc_gpu.pipeline.PipelinedMpc): two once the batch is at least two rounds of wavefronts deep on the chip's 1024
    wavefront slots (measured: 4096 x (N = 50, 10 obstacles) +27 %, 32768 x (N = 20, 3 obstacles) +6 %, 65536 +2.4 %; four lose -- profiles/r04_streams_probe_*.json);
    one for the 1024 instances of C2, which are a single round"""
    return 2 if batch >= 2048 else 1


class Loop:
    """closed-loop state of one rank's slice on one GPU"""

    def __init__(self, mpc_gpu, torch, N, n_obst, x0, goal, obst, dev, streams=1, step_flags=None, **cfg):
        batch = x0.shape[0]
        self.torch = torch
        self.streams = streams
        from mpc_gpu import _lib as L
        self.flags = (L.STEP_SHIFT | L.STEP_PLANT | L.STEP_OBSTACLES) if step_flags is None else step_flags
        if streams > 1:
            from mpc_gpu.pipeline import PipelinedMpc
            self.m = PipelinedMpc(N, n_obst, 0.1 * N, max_batch=batch, device=dev.index or 0, streams=streams, **cfg)
        else:
            self.m = mpc_gpu.BatchedMpc(N, n_obst, 0.1 * N, max_batch=batch, device=dev.index or 0, **cfg)
        self.B, self.N, self.no = batch, N, n_obst
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        z = lambda *s, dt=torch.float64: torch.zeros(*s, dtype=dt, device=dev)
        self.x0, self.goal, self.obst = t(x0), t(goal), t(obst)
        self.x0_init, self.obst_init = self.x0.clone(), self.obst.clone()
        self.X, self.U = z(batch, N + 1, 5), z(batch, N, 2)
        self.u0, self.cost = z(batch, 2), z(batch)
        self.status, self.iters = z(batch, dt=torch.int32), z(batch, dt=torch.int32)
        self.stream = torch.cuda.current_stream().cuda_stream   # the caller runs us under an explicit torch stream
        assert self.stream != 0, "run under an explicit torch stream so torch ops and library kernels share one queue"

    def join(self):
        """the current stream waits for the sub-batch streams (a no-op with one stream)"""
        if self.streams > 1:
            self.m.join()

    def reset(self):
        """start an episode: initial scenario, set_initial_guess() (two device-to-device copies + one small kernel)"""
        if self.streams > 1:
            self.m.join()           # the copies below run on the current stream: behind the last control steps of every sub-batch ...
        self.x0.copy_(self.x0_init); self.obst.copy_(self.obst_init)
        if self.streams > 1:
            self.m.fork()           # ... and the sub-batch streams continue behind them
            self.m.reset_guess_dev(self.B, self.x0, self.X, self.U)
        else:
            self.m.reset_guess_dev(self.B, self.x0, self.X, self.U, stream=self.stream)

    def control_step(self, cost_out=None):
        """one control step of the whole batch = ONE kernel launch per sub-batch stream; cost_out: where the kernels write this step's per-scenario costs"""
        if self.streams > 1:
            self.m.closed_loop_step_dev(self.B, self.x0, self.obst, self.goal, self.X, self.U, self.u0,
                                        self.cost if cost_out is None else cost_out, self.status, self.iters, None, flags=self.flags)
        else:
            self.m.closed_loop_step_dev(self.B, self.x0, self.obst, self.goal, self.X, self.U, self.u0,
                                        self.cost if cost_out is None else cost_out, self.status, self.iters, None, flags=self.flags, stream=self.stream)


class CostExchange:
    """Per-scenario costs of GATHER_EVERY consecutive control steps, all-gathered in ONE message on a side stream (double buffered: one
    history is in flight while the next fills).  Not every control step: at batch 1024 the solve kernel fills every SIMD with exactly
    one 512-register wavefront, so any kernel beside it (the collective's) holds back the workgroups of the CUs it occupies."""

    def __init__(self, torch, world, batch, dev, gather_costs, join=None):
        self.torch, self.world, self.gather = torch, world, gather_costs
        self.join = join or (lambda: None)      # makes the current stream wait for the solver's sub-batch streams (Loop.join)
        self.hist = torch.zeros(2, GATHER_EVERY, batch, dtype=torch.float64, device=dev)
        self.out = torch.zeros(world * GATHER_EVERY * batch, dtype=torch.float64, device=dev)
        self.side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self.work, self.n = None, 0

    def row(self):
        return self.hist[(self.n // GATHER_EVERY) % 2, self.n % GATHER_EVERY]

    def stepped(self):
        buf = (self.n // GATHER_EVERY) % 2
        self.n += 1
        if self.n % GATHER_EVERY:
            return
        self.wait()                                              # the previous message left the other history two fills ago
        self.join()
        if self.side is not None:
            self.side.wait_stream(self.torch.cuda.current_stream())
            with self.torch.cuda.stream(self.side):
                _, self.work = self.gather(self.hist[buf].view(-1), self.world, async_op=True, out=self.out)
        else:
            _, self.work = self.gather(self.hist[buf].view(-1), self.world, async_op=True, out=self.out)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None

    def gathered(self):
        """[world][GATHER_EVERY][batch] view of the last completed message"""
        return self.out.view(self.world, GATHER_EVERY, -1)


class CApiCostExchange(CostExchange):
    """The same double-buffered exchange through the library's own collective (include/mpc_gpu.h: mpc_comm_init, mpc_allgather_cost_dev = RCCL's
    ncclAllGather issued by libmpcgpu on the side stream) -- the path a host without torch.distributed takes."""

    def __init__(self, torch, world, batch, dev, solver, join=None):
        super().__init__(torch, world, batch, dev, None, join=join)
        self.m = solver
        self.done = torch.cuda.Event()
        self.pending = False

    def stepped(self):
        buf = (self.n // GATHER_EVERY) % 2
        self.n += 1
        if self.n % GATHER_EVERY:
            return
        self.wait()
        self.join()
        self.side.wait_stream(self.torch.cuda.current_stream())
        self.m.allgather_cost_dev(GATHER_EVERY * self.hist.shape[2], self.hist[buf], self.out, stream=self.side.cuda_stream)
        self.done.record(self.side)
        self.pending = True

    def wait(self):
        if self.pending:
            self.done.synchronize()
            self.pending = False


def exchange_comm_id(dist, rank, make_id):
    """Rank 0 draws the communicator's unique id (mpc_comm_unique_id: 128 bytes) and every rank receives it over the process group the launcher
    has initialised anyway -- the only thing torch.distributed does for the C-ABI exchange."""
    box = [bytes(make_id()) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


# ------------------------------------------------------------------------------------------------------------------ CPU baseline

class _OracleAsAcados:
    """The oracle behind acados' method names, so that mpc_gpu.closed_loop.ShimLoop -- the reference's per-step call pattern, ~170
    Python <-> solver crossings per control step at N = 20 -- can drive it.  A mimic of the reference's Python-overhead
    regime, never presented as acados.  Lives here because only bench.py's cpu_baseline leg may use the oracle."""

    def __init__(self, orc, cfg, goal):
        self.orc, self.cfg = orc, cfg
        self.X = np.zeros((cfg.N + 1, 5)); self.U = np.zeros((cfg.N, 2)); self.P = np.zeros((cfg.N + 1, cfg.n_obst, 2))
        self.alpha = np.zeros(cfg.N + 1); self.goal = np.array(goal, float); self.x0 = np.zeros(5)

    def set(self, stage, fieldname, v):
        if fieldname == "x": self.X[stage] = v
        elif fieldname == "u": self.U[stage] = v
        elif fieldname == "p": self.P[stage] = np.asarray(v).reshape(-1, 2)
        else: self.x0 = np.array(v, float)

    def cost_set(self, stage, fieldname, v):
        self.alpha[stage] = v[0]

    def get(self, stage, fieldname):
        return (self.X if fieldname == "x" else self.U)[stage].copy()

    def reset(self):
        self.X[:] = 0; self.U[:] = 0

    def solve(self):
        r = self.orc.rti_solve(self.cfg, self.x0, self.P, self.goal, self.X, self.U, alpha=self.alpha)
        self.X, self.U = r["X"], r["U"]
        return r["status"]


class _OraclePlant:
    def __init__(self, orc, dt):
        self.orc, self.dt, self.x, self.u = orc, dt, np.zeros(5), np.zeros(2)

    def set(self, fieldname, v):
        if fieldname == "x": self.x = np.array(v, float)
        else: self.u = np.array(v, float)

    def solve(self):
        self.x = self.orc.dynamics(self.x, self.u, self.dt)[0]

    def get(self, fieldname):
        return self.x.copy()


def cpu_budget():
    """(hardware threads this process may run on, CPU quota of its cgroup in cores or None).  A GPU box of this pool shows 256 hardware threads but gives a
    one-GPU lease a quota of about 16 cores: OpenMP threads beyond the quota only time-slice (round 3's "15x on 128 cores" was this quota, not the allocator)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else int(t.split()[0]) / int(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))):
        try:
            quota = parse(open(path).read().strip())
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    return n, quota


def cpu_baseline(N, n_obst, x0, goal, obst, budget_s=18.0, steady_s=5.0):
    """The oracle (CPU restatement of the same RTI step; acados itself cannot run on this box) timed on this box's host cores on a
    bounded sample of the same workload: closed-loop control steps of the first S scenarios (a few untimed, then timed until the budget
    is used; solve calls only).  Three figures: all hardware threads or half of them, whichever is faster (`value`, `cores`); one thread;
    and one thread driven through the reference's per-step Python call pattern.  The timed copy is built -O3 -march=native
    (oracle/Makefile: liborc_bench.so); the checker the tests use stays -O2 without contraction."""
    from oracle import oracle as orc
    orc.use_bench_build()
    cfg = orc.config(N, n_obst, 0.1 * N)      # the library's defaults (qp_tol 1e-10)
    ncpu, quota = cpu_budget()
    dt = 0.1

    def closed_loop_rate(S, nthreads, budget, warm=5):
        """solve calls of closed-loop episodes of the first S scenarios (each episode from the initial scenario, as the GPU's) until `budget` seconds have passed;
        the first `warm` control steps of the first episode untimed"""
        solves, secs, steps, t_begin = 0, 0.0, 0, time.perf_counter()
        while True:
            xs, gs, os_ = x0[:S].copy(), goal[:S].copy(), obst[:S].copy()
            X = np.zeros((S, N + 1, 5)); U = np.zeros((S, N, 2))
            for b in range(S):
                X[b], U[b] = orc.initial_guess(cfg, xs[b])
            for _ in range(EPISODE):
                P = orc.predict_params_batch(cfg, os_)
                t0 = time.perf_counter()
                r = orc.rti_solve_batch(cfg, xs, P, gs, X, U, nthreads=nthreads)
                t1 = time.perf_counter()
                if steps >= warm:
                    solves += S; secs += t1 - t0
                X, U = r["X"], r["U"]
                orc.advance_batch(cfg, xs, r["u0"], os_, X, U)      # plant step, obstacle step, warm-start shift: in place, one call
                steps += 1
                if time.perf_counter() - t_begin > budget and steps >= warm + 3:
                    return solves / secs, steps - warm

    S_all = min(len(x0), max(64, 8 * ncpu))
    # doubling sweep over the OpenMP thread count, 8, 16, 32 ... up to the hardware threads this process may run on, ~2 s each; it stops at the plateau
    # (a doubling that gains < 5 %).  `value` is the plateau -- not a quota heuristic: on the driver's box (2 x EPYC 9575F, 256 threads) the rate
    # was still linear at 32 threads although the cgroup quota reads ~16 cores
    rates, n, prev = {}, min(8, ncpu), 0.0
    while True:
        rates[n] = closed_loop_rate(S_all, n, budget_s / 9)
        if n >= ncpu or rates[n][0] < 1.05 * prev:
            break
        prev, n = rates[n][0], min(ncpu, 2 * n)
    best = max(rates, key=lambda n: rates[n][0])
    # ... and the plateau thread count timed three times over >= 5 s of wall time each (the sweep's 2 s points differ by +-25 % between runs on the driver's box,
    # VERDICT r05 weak 7): `value` is the median, `spread` the three samples' minimum and maximum
    reps = sorted(closed_loop_rate(S_all, best, steady_s) for _ in range(3))
    steady, steady_steps = reps[1]
    one, one_steps = closed_loop_rate(min(len(x0), 16), 1, budget_s / 6)
    # reference call pattern: ONE scenario, the reference's per-step solver calls (ShimLoop) on oracle-backed objects, wall time of
    # everything in the loop (that is the point: the reference's step is Python overhead around the solve)
    from mpc_gpu.closed_loop import EpisodeState, ShimLoop
    from mpc_gpu.world import Obstacle
    ocp = _OracleAsAcados(orc, cfg, goal[0]); sim = _OraclePlant(orc, dt)
    st = EpisodeState(x=x0[0].copy(), goal=goal[0].copy(), obstacles=[Obstacle(*o, dt=dt) for o in obst[0]])
    loop = ShimLoop(ocp, sim, N)
    loop.cold_start(st)
    t0 = time.perf_counter(); n_py = 0
    while time.perf_counter() - t0 < budget_s / 8 and n_py < 4 * EPISODE:
        loop.control_step(st); loop.shift_warm_start(); n_py += 1
    py_rate = n_py / (time.perf_counter() - t0)
    return {"value": steady, "unit": "solves/s", "cores": best, "kind": "port", "host_cpu": orc._cpu_model(), "host_threads": ncpu,
            "spread": {"min": float(f"{reps[0][0]:.4g}"), "max": float(f"{reps[2][0]:.4g}"), "samples": 3, "seconds_each": steady_s},
            "cpu_quota_cores": quota,
            "one_thread": one, "per_core": steady / best, "scaling_efficiency": steady / (best * one),
            "python_call_pattern_one_thread": py_rate,
            "threads": {str(n): float(f"{r[0]:.4g}") for n, r in rates.items()},
            "sample": f"{S_all} scenarios x {steady_steps} closed-loop steps x 3 (median; spread = min / max), C oracle -O3 OpenMP at the plateau of a thread sweep, solve calls only",
            "sample_detail": f"first {S_all} scenarios x {steady_steps} closed-loop control steps (after 5 untimed), three times, of the same workload, oracle "
                             f"(C, f64, -O3 -march=native, OpenMP over instances), solve calls only; one_thread: 16 scenarios x {one_steps} steps; "
                             f"python_call_pattern: 1 scenario x {n_py} control steps through the reference's ~{8 * N + 12} solver calls per step "
                             "(whole loop timed); acados itself cannot run here, so this is a restatement, not the reference's solver",
            "note": "a reported baseline, not a target: the oracle is a dense, generic checker (no structure exploitation; per-thread workspace since round 4). "
                    "`cores` = the OpenMP thread count of `value`: the best of a doubling sweep (`threads`: the curve) that stops when a doubling gains < 5 %; "
                    "cpu_quota_cores is what the cgroup reports and is informational only; scaling_efficiency = value / (cores x one_thread)"}


# ------------------------------------------------------------------------------------------------------------------ measurement

def measure(torch, dist, loop, world, exch, steps, warmup, dev):
    """`warmup` untimed episodes, then exactly `steps` timed ones.  Returns elapsed (max over ranks), kernel ms sum, launches timed,
    mean interior-point iterations and failure / cap fractions of the timed region."""
    def episode():
        loop.reset()
        for _ in range(EPISODE):
            if exch is None:
                loop.control_step()
            else:
                loop.control_step(cost_out=exch.row()); exch.stepped()
    for _ in range(warmup):
        episode()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    loop.m.profile_enable(True, every=EVENT_EVERY)               # event pool created here, outside the timed region
    it_acc = torch.zeros(loop.B, dtype=torch.int32, device=dev)  # summed inside the solve kernel (mpc_set_accumulators)
    st_acc = torch.zeros(loop.B, dtype=torch.int32, device=dev)
    loop.m.set_accumulators(it_acc, st_acc)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        episode()
    if exch is not None:
        exch.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = loop.m.profile_read()
    loop.m.profile_enable(False)
    loop.m.set_accumulators(None, None)
    # what an event pair measures with NOTHING between its records, on the same stream (the figure a timed launch carries on top of its own duration)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    for a, b in evs:
        a.record(); b.record()
    torch.cuda.synchronize()
    pair_ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    n = loop.B * steps * EPISODE
    return dict(elapsed=elapsed, kern_ms=kern_ms, launches=launches, pair_ms=pair_ms, steps=steps, mean_iters=float(it_acc.double().sum().item()) / n,
                fail=float((st_acc % 65536).double().sum().item()) / n, cap=float((st_acc // 65536).double().sum().item()) / n)


def roofline(loop, N, no, r):
    """FP64 vector ALU is the roof that binds this path (~175 flop per algorithmic byte); HBM is the secondary figure.
    Everything from THIS run: kernel time from HIP events on the launch stream, flops from the measured mean iteration count."""
    batch = loop.B
    kname = loop.m.kernel_name(batch)
    raw_s = r["kern_ms"] / max(1, r["launches"]) * 1e-3
    avg_s = max(raw_s - r["pair_ms"] * 1e-3, 1e-9)          # event-bracketed duration minus what an empty bracket measures
    wall_per_launch = r["elapsed"] / (r["steps"] * EPISODE)
    if loop.streams > 1:
        # sub-batches pipelined on several streams: their launches overlap in time, so an event-bracketed launch duration is not the time the chip spent
        # on that launch's work.  The per-control-step wall time of the whole batch (all sub-launches) is -- `kernel_time_over_wall` is 1 by construction
        avg_s = wall_per_launch
    flops = algorithmic_flops_per_solve(N, no, r["mean_iters"]) * batch
    abytes = algorithmic_bytes_per_solve(N, no) * batch
    pm = measured_pmc(kname, batch)
    # The PMC-derived fields (issue, latency_frac, wait_frac, lanes_exec, traffic, sweep_frac) are NOT measured by this run: they are replayed from the newest
    # committed rocprofv3 profile of the same kernel and batch.  `pmc_source` names that file with the kernel duration it was taken at, and the fields are
    # withheld (null, with a note on stderr) when this run's own launch time differs from the profile's by more than PMC_STALE_TOL -- the kernel has changed
    # since the profile was taken.  Pipelined runs compare nothing (their launches overlap): they carry no PMC fields at all.
    pmc_source, pmc_stale = None, False
    if pm:
        prof_us = pm["avg_ns"] * 1e-3 if pm.get("avg_ns") else None
        pmc_source = {"file": "profiles/" + pm["file"], "kernel": pm["kernel"], "avg_launch_us": prof_us}
        if loop.streams > 1 or prof_us is None or abs(avg_s * 1e6 - prof_us) > PMC_STALE_TOL * prof_us:
            pmc_stale = True
            sys.stderr.write(f"bench.py: PMC-derived roofline fields withheld for {kname} at batch {batch}: this run {avg_s * 1e6:.1f} us per launch"
                             f"{' (pipelined streams)' if loop.streams > 1 else ''}, {pm['file']} was taken at {prof_us} us -- re-profile (scripts/profile_passes.sh)\n")
            pmc_source["stale"] = True
            pm = None
    issue = None
    if pm and pm["valu_insts"] and pm["avg_ns"]:
        # the roof that binds: VALU issue slots.  SQ_INSTS_VALU x 4 cycles / (SIMDs x kernel cycles), instructions and duration from the SAME profile
        issue = pm["valu_insts"] * VALU_CYCLES_PER_INST / (SIMDS * pm["avg_ns"] * CLOCK_GHZ)
    latency_frac = wait_frac = None
    if pm and pm.get("valu_insts") and pm.get("lds_insts") and pm.get("wave_cycles"):
        # The floor of what these wavefronts could take at all: their own instruction stream priced as a LONE wavefront issues it (one or two wavefronts per
        # SIMD: nothing else fills its bubbles) -- every VALU instruction independent (5.0 cycles; a dependent one costs 8.4, so the true floor is higher),
        # every LDS instruction 14 -- over the cycles the wavefronts were resident (SQ_WAVE_CYCLES counts quad-cycles).  1 - latency_frac is ALL the slack a
        # perfect schedule of the same instructions could recover; the rest of the distance to the FP64 peak is lanes without data and the lone stream.
        latency_frac = (LONE_WAVE_VALU_CYCLES * pm["valu_insts"] + LONE_WAVE_LDS_CYCLES * pm["lds_insts"]) / (4.0 * pm["wave_cycles"])
        wait_frac = pm["wait_any"] / pm["wave_cycles"] if pm.get("wait_any") else None
    sweep_frac = None
    try:        # per phase: the sweeps' loop bodies in the shipped listing at the same prices (dependent multiply-adds 8.4) over their measured cycles (scripts/critical_path_model.py);
        import glob      # a recorded figure like the PMC ones: the newest committed record, withheld together with them
        cps = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_critical_path_c2.json")))
        cp = json.load(open(cps[-1])) if cps and pm else {"kernels": []}
        for rec in cp["kernels"]:
            if rec["kernel"] == kname:
                sweep_frac = {"factor": rec["factor_sweep"]["model_over_measured"], "vector": rec["vector_sweep"]["model_over_measured"]}
                if "dep_floor" in rec["factor_sweep"]:      # (round 6) the factor sweep's own dependency chain per stage over the measured cycles per stage
                    sweep_frac["dep_floor"] = rec["factor_sweep"]["dep_floor"]
                pmc_source["sweep_frac_file"] = "profiles/" + os.path.basename(cps[-1])
    except (OSError, KeyError, ValueError):
        pass
    return {"bound": "fp64_valu", "achieved": flops / avg_s / 1e12, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
            "pmc_source": pmc_source, "pmc_stale": pmc_stale,
            "latency_frac": latency_frac, "wait_frac": wait_frac, "sweep_frac": sweep_frac,
            "latency_model": (f"(5.0 cycles x SQ_INSTS_VALU + 14 cycles x SQ_INSTS_LDS) / (4 x SQ_WAVE_CYCLES) of profiles/{pm['file']}: the kernel's own instruction stream at the "
                              "price a lone wavefront pays per instruction (micro-benchmarks, docs/HISTORY.md 4.1c; all VALU taken as independent: a lower bound of "
                              "the floor) over the cycles its wavefronts were resident") if latency_frac is not None else None,
            "frac": flops / avg_s / 1e12 / FP64_VALU_PEAK_TF,
            "peak_source": "AMD public specification of MI355X (FP64 vector = FP64 matrix = 78.6 TFLOP/s); MI355X_MICROARCH.md lists the FP32 vector peak (157.3) only",
            "issue": issue,
            "issue_note": ("fraction of the chip's VALU issue slots this kernel uses: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel cycles at 2.4 GHz), both from "
                           f"profiles/{pm['file']} -- the resource that binds (one or two wavefronts per SIMD, dependent FP64 chains), next to which `frac` is small "
                           "because few lanes of an instruction carry data") if issue is not None else None,
            "lanes_active": lanes_useful(kname, N, no),
            "lanes_exec": pm["lanes_active"] if pm else None,
            "lanes_note": ("lanes_active: lanes (of 64) that carry DATA per VALU instruction, a static model -- per interior-point iteration the row phases run on "
                           "every stage lane of the wavefront's instances, the Riccati factor sweep on 8 and the three vector sweeps on 5 lanes of a 16-lane DPP row "
                           "per instance, weighted with the phases' VALU instruction counts (profiles/r03_*_phase_instruction_counts.txt, r01_split_phase_timing.txt); "
                           "lanes_exec: lanes ENABLED per VALU instruction, SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU of the same profile as `issue` (the ratio reads "
                           "exactly K on a kernel with K lanes enabled: profiles/r04_lanes_counter_calibration.json) -- the sweeps keep EXEC full on purpose "
                           "(unconditional arithmetic, dead stores: no EXEC manipulation in the loops), so it bounds lanes_active from above"),
            "traffic": pm["traffic"] if pm else None,
            "traffic_source": (f"HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE passes of this command, profiles/{pm['file']} "
                               "(FETCH_SIZE uncorrected: 8-byte-per-lane loads)") if pm else None,
            "kernel": kname, "streams": loop.streams, "avg_launch_us": avg_s * 1e6, "avg_launch_us_raw": raw_s * 1e6, "event_pair_overhead_us": r["pair_ms"] * 1e3,
            "kernel_time_over_wall": avg_s / wall_per_launch, "launches_timed": r["launches"],
            "launch_timing": (f"HIP events on the launch stream around every {EVENT_EVERY}th launch of the timed region, minus the duration an empty event pair "
                              "measures on the same stream (median of 64); kernel_time_over_wall = that per-launch time / wall time per control step (one launch "
                              "each): <= 1, the rest is launch gaps and the episode resets"),
            "algorithmic_flops_per_launch": flops, "mean_ipm_iters": r["mean_iters"],
            "flop_model": "SURVEY.md 8(d): N*200 + K*[N*(7/3 nx^3 + 4 nx^2 nu + 2 nx nu^2 + nu^3/3) + N*(2 (nx+nu)^2 + 6 n_ineq)], K = mean_ipm_iters of this run",
            "hbm": {"achieved": abytes / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": abytes / avg_s / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": abytes},
            "note": "the stage recursions keep 8-24 of 64 lanes busy and a wavefront is limited by the issue rate of its own dependent "
                    "instruction stream (one or two wavefronts per SIMD), so the flop fraction is small by construction (DESIGN.md section 5)"}


def c1_latency(mpc_gpu, N, no):
    """BASELINE configs[0]: ONE scenario (3 static obstacles), host-pointer API as the reference's own loop would call it: per control
    step mpc_solve_obst + mpc_plant_step + mpc_shift with numpy arrays in and out (PCIe and one stream sync per call included)."""
    gold = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))
    obst = gold[f"gen_RANDOM_{no}"][0:1].copy(); obst[:, :, 2:] = 0.0
    x = np.array([[-6.0, -6.0, np.pi / 4, 0, 0]]); goal = np.array([[6.0, 6.0]])
    with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=1) as s:
        s.reset_guess(x)
        ts = []
        for k in range(150):
            t0 = time.perf_counter()
            out = s.solve(x, obst, goal)
            t1 = time.perf_counter()
            x = s.plant_step(x, out["u0"]); s.shift(1)
            if k >= 20:
                ts.append(t1 - t0)
    return {"workload": f"C1: single scenario, N={N}, {no} static obstacles, host-pointer API (numpy in / out, PCIe + stream sync included)",
            "ms_per_solve_median": float(np.median(ts)) * 1e3, "solves_per_s": 1.0 / float(np.median(ts)), "control_steps": len(ts)}


# ------------------------------------------------------------------------------------------------------------------ the output line

LINE_LIMIT = 4096            # the driver keeps only the tail of stdout: a line beyond a few KB is cut and cannot be parsed (BENCH_r04: 20.7 KB -> parsed null)
# the full record of a run: --record PATH, else $MPC_BENCH_RECORD, else gpurun_out/bench_last.json (scratch: a bench run must not dirty the tracked tree -- until
# round 5 it overwrote profiles/bench_last.json; records worth keeping are copied to profiles/rNN_* by hand)
FULL_RECORD = os.environ.get("MPC_BENCH_RECORD") or os.path.join(ROOT, "gpurun_out", "bench_last.json")


def _num(v, digits=5):
    """numbers of the line with a few significant digits (the full record keeps every bit)"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        return float(f"{v:.{digits}g}")
    return v


def compact_line(out):
    """The ONE JSON line of the contract, numbers only.  Every prose note, source, model description and nested detail of `out` stays in the full
    record (FULL_RECORD, also gpurun_out/ when that directory exists), never on stdout."""
    def pick(d, keys):
        return {k: _num(d[k]) for k in keys if d is not None and k in d and d[k] is not None}
    cfg = out["config"]
    line = pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    line["vs_baseline"] = None
    line.update(pick(out, ("dtype", "data")))
    line["config"] = pick(cfg, ("workload", "global_batch", "per_gpu_batch", "N", "n_obst", "qp_tol", "control_steps_per_step", "parallelism"))
    line.update(pick(out, ("mean_ipm_iters", "qp_failure_frac", "streams_per_gpu", "exchange", "rccl_ranks", "gather_check", "rccl_path", "torch_pg")))
    line["roofline"] = pick(out["roofline"], ("bound", "achieved", "peak", "unit", "frac", "issue", "latency_frac", "sweep_frac", "lanes_active", "lanes_exec", "traffic", "kernel", "avg_launch_us"))
    for k in ("issue", "latency_frac", "traffic"):          # the replayed fields are always present: a number, or null when their profile no longer describes the run
        line["roofline"].setdefault(k, None)
    src = out["roofline"].get("pmc_source")
    line["roofline"]["pmc_source"] = None if not src else f"{os.path.basename(src['file'])}@{_num(src.get('avg_launch_us'), 4)}us" + (" STALE" if src.get("stale") else "")
    for key, name in (("extra", "c3"), ("extra_c5", "c5"), ("extra_c4_share", "c4_share"), ("extra_c5_share", "c5_share")):
        e = out.get(key)
        if e:
            line[name] = {**pick(e, ("value", "value_one_stream", "mean_ipm_iters")), **pick(e["roofline"], ("frac", "issue", "latency_frac"))}
    for key in ("value_reset_on_fail", "value_qp_tol_1e-8"):
        if key in out:
            line[key] = _num(out[key]["value"])
    if "c1" in out:
        line["c1_ms_per_solve"] = _num(out["c1"]["ms_per_solve_median"])
    if "cpu_baseline" in out:
        line["cpu_baseline"] = pick(out["cpu_baseline"], ("value", "unit", "cores", "kind", "spread", "one_thread", "python_call_pattern_one_thread", "host_cpu",
                                                          "host_threads", "threads", "sample"))
    line["full_record"] = os.path.relpath(FULL_RECORD, ROOT) if os.path.abspath(FULL_RECORD).startswith(ROOT + os.sep) else FULL_RECORD
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, f"bench line is {len(text)} bytes: the driver cannot parse more than {LINE_LIMIT}"
    return text


def emit(out):
    """full record to the files, compact line to stdout (the LAST line of stdout)"""
    try:
        os.makedirs(os.path.dirname(os.path.abspath(FULL_RECORD)), exist_ok=True)
        with open(FULL_RECORD, "w") as f:
            json.dump(out, f, indent=1)
    except OSError as e:
        sys.stderr.write(f"bench.py: could not write {FULL_RECORD}: {e}\n")
    sys.stdout.flush()
    print(compact_line(out), flush=True)


# ------------------------------------------------------------------------------------------------------------------ entry points

def spawn_ranks(args, argv):
    """`python bench.py --gpus N` outside a torchrun environment: start the N ranks (one process per GPU) and relay rank 0's line.
    Nothing in this process has touched the GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    if r.returncode != 0 or not line:
        sys.stderr.write(r.stdout)
        sys.exit(r.returncode or 1)
    print(line[-1])
    sys.exit(0)


def dry_run(args, world, rank):
    """CPU rehearsal of the multi-rank plumbing (tests/test_host_logic.py, gloo): workload sharding, cost histories, the all-gather --
    no kernels.  Every 'control step' writes the instance's GLOBAL index + the step number as its cost; after the exchange every rank
    must hold exactly that for all ranks."""
    import torch
    import torch.distributed as dist
    from mpc_gpu.sharding import gather_costs, shard_slice
    N, no, _, _, scaling = WORKLOADS[args.workload]
    x0, goal, obst, desc, (lo, hi), G = make_workload(args.workload, world, rank, shard_slice)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    ok = True
    comm_id_sha1 = None
    if world > 1:
        # the id plumbing of --exchange capi: rank 0's unique id reaches every rank (without a GPU mpc_comm_unique_id refuses, as everything in the
        # library does; the rehearsal then ships 128 random bytes through the same function)
        import hashlib
        from mpc_gpu import BatchedMpc, MpcError

        def make_id():
            try:
                return BatchedMpc.comm_unique_id()
            except (MpcError, OSError):
                return os.urandom(128)
        uid = exchange_comm_id(dist, rank, make_id)
        digests = [None] * world
        dist.all_gather_object(digests, hashlib.sha1(uid).hexdigest())
        ok = ok and len(uid) == 128 and all(d == digests[0] for d in digests)
        comm_id_sha1 = digests[0]
        exch = CostExchange(torch, world, hi - lo, dev, gather_costs)
        idx = torch.arange(lo, hi, dtype=torch.float64)
        for k in range(2 * GATHER_EVERY):
            exch.row().copy_(idx + 1e6 * k); exch.stepped()
        exch.wait()
        got = exch.gathered()                                    # the second message: steps GATHER_EVERY .. 2 GATHER_EVERY - 1
        for r in range(world):
            rl, rh = shard_slice(G, r, world)
            want = torch.arange(rl, rh, dtype=torch.float64)[None] + 1e6 * torch.arange(GATHER_EVERY, 2 * GATHER_EVERY, dtype=torch.float64)[:, None]
            ok = ok and bool(torch.equal(got[r], want))
        flag = torch.tensor([1.0 if ok else 0.0]); dist.all_reduce(flag, op=dist.ReduceOp.MIN); ok = bool(flag.item() == 1.0)
    if rank == 0:
        print(json.dumps({"metric": "MPC solves/sec (N=20, 3 obstacles)", "value": 0.0, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "dry_run": True, "scaling": scaling, "gather_check": ok, "exchange": args.exchange, "comm_id_sha1": comm_id_sha1,
                          "config": {"workload": desc, "global_batch": G, "rank0_slice": [lo, hi], "x0_shape": list(x0.shape)}}))
    if world > 1:
        dist.destroy_process_group()


def rendezvous_guard(seconds, rank, _exit=os._exit):
    """A timer that ends THIS process with code 3 if it is not cancelled within `seconds`: mpc_comm_init (ncclCommInitRank) blocks in native code until every
    rank has joined, so a rank that never arrives would otherwise hang the job; torchrun then takes the other ranks down."""
    import threading

    def fire():
        sys.stderr.write(f"bench.py: rank {rank} still inside the mpc_comm_init rendezvous after {seconds:.0f} s -- giving up (MPC_BENCH_COMM_TIMEOUT)\n")
        sys.stderr.flush()
        _exit(3)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed episodes (100 control steps each)")
    ap.add_argument("--warmup", type=int, default=3, help="untimed episodes")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c2 for every N (BASELINE configs[1], the configuration the metric is quoted on: 1024 scenarios per GPU, weak scaling); c4 / c5: "
                         "the sharded global batches of configs[3] / [4] (262144 / 32768 scenarios over the ranks, strong scaling)")
    ap.add_argument("--exchange", default="capi", choices=["capi", "torch"],
                    help="multi-rank cost all-gather: capi = the library's own C-ABI collective (mpc_comm_init + mpc_allgather_cost_dev, RCCL called by "
                         "libmpcgpu; default), torch = torch.distributed.all_gather_into_tensor")
    ap.add_argument("--streams", type=int, default=0,
                    help="sub-batches pipelined on separate HIP streams per GPU (mpc_gpu.pipeline.PipelinedMpc); 0 = automatic: two from 2048 instances per GPU on, else one")
    ap.add_argument("--share", type=int, default=0,
                    help="single-GPU run of ONE rank's share of a K-way sharded workload (rank 0's slice of shard_slice(total, r, K)): the per-GPU work of "
                         "`--gpus K` without the other K - 1 GPUs (profiles/r04_c5_share_*: --workload c5 --share 8)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the cost exchange with a single rank too (RCCL accepts a communicator of one): rehearses on a one-GPU box exactly the code path "
                         "`--gpus N` takes -- id exchange aside -- including the join of pipelined sub-batch streams")
    ap.add_argument("--with-torch-pg", action="store_true",
                    help="one rank only: initialise a torch.distributed NCCL process group of world size 1 first (and run an all_reduce on it before the library's "
                         "communicator exists and another after its all-gathers) -- with --force-exchange the exact process state of a rank of `--gpus N`")
    ap.add_argument("--allow-exchange-fallback", action="store_true",
                    help="if the C-ABI exchange cannot be set up on every rank, run the torch.distributed exchange instead of exiting non-zero")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the supplementary C3, C5 and C1 measurements")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of the multi-rank plumbing (gloo), no kernels")
    ap.add_argument("--record", default=None, help="where the full record (every note and nested detail) goes; default $MPC_BENCH_RECORD or gpurun_out/bench_last.json")
    args = ap.parse_args()
    if args.record:
        global FULL_RECORD
        FULL_RECORD = os.path.abspath(args.record)
    if args.workload is None:
        # ONE workload for every N, so that the driver's 1 / 2 / 4 / 8-GPU values form one curve: the configuration the metric is quoted on (C2), 1024 scenarios PER
        # GPU -- weak scaling.  The sharded global batches of BASELINE configs[3] / [4] are `--workload c4 | c5` (strong scaling; one rank's share of them is
        # measured on one GPU by the default run: c4_share, c5_share)
        args.workload = "c2"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return dry_run(args, world, rank)

    import torch
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
        if args.with_torch_pg:
            # the process state of a rank of `--gpus N`, on one GPU: a torch.distributed NCCL (= RCCL) process group of one rank, its communicator created
            # (first collective) BEFORE the library binds librccl and makes its own -- two communicators, one process, one librccl mapping (rccl_path)
            import socket
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
            probe = torch.ones(4, dtype=torch.float64, device="cuda:0")
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            assert float(probe.sum().item()) == 4.0
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))   # one explicit queue for torch ops AND the library's kernels

    import __graft_entry__ as g
    if world > 1:           # one rank (re)builds if anything is stale; the others wait and then only load
        if rank == 0:
            g.build()
        dist.barrier()
    else:
        g.build()
    import mpc_gpu
    from mpc_gpu.sharding import gather_costs, shard_slice

    N, no, _, _, scaling = WORKLOADS[args.workload]
    x0, goal, obst, desc, (lo, hi), G = make_workload(args.workload, world, rank, shard_slice)
    if args.share > 1 and world == 1:
        x0, goal, obst, desc, (lo, hi), _ = make_workload(args.workload, args.share, 0, shard_slice)
        G = hi - lo
        desc += f" -- rank 0's slice of {args.share} ({G} instances) on this one GPU"
    loop = Loop(mpc_gpu, torch, N, no, x0, goal, obst, dev, streams=args.streams or pick_streams(hi - lo))
    exch, exchange, exchange_note, rccl_path = None, None, None, None
    if (world > 1 or args.force_exchange) and (hi - lo) * world == G:
        exchange = args.exchange if world > 1 else "capi"
        if exchange == "capi" and os.environ.get("MPC_BENCH_ONE_GPU") == "1":
            exchange = "torch"                    # RCCL refuses two ranks on one device: the one-GPU rehearsal keeps the gloo transport
        if exchange == "capi":
            # The library's own collective.  ncclCommInitRank is a rendezvous that blocks until ALL ranks have joined, so the ranks agree BEFORE it: every rank
            # probes the library locally (mpc_comm_unique_id: dlopen + dlsym + one RCCL call, no rendezvous), the flags are MIN-reduced over the launcher's
            # process group, and mpc_comm_init is entered only when every rank can.  A rank that cannot makes the whole job exit non-zero -- a SCALE record
            # must not silently measure another collective -- unless --allow-exchange-fallback asks for the torch.distributed exchange instead.
            def agree(ok, err):
                if dist is not None:
                    flag = torch.tensor([ok], dtype=torch.float64, device=dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    return float(flag.item()), err
                return ok, err
            ok, err = 1.0, ""
            try:
                mpc_gpu.BatchedMpc.comm_unique_id()
                rccl_path = mpc_gpu.BatchedMpc.comm_library_path()
            except Exception as e:      # noqa: BLE001 -- reported below
                ok, err = 0.0, f"{type(e).__name__}: {e}"[:200]
            ok, err = agree(ok, err)
            if ok == 1.0:
                # the rendezvous itself cannot be abandoned once entered (a native blocking call on every rank): a rank still inside it after
                # MPC_BENCH_COMM_TIMEOUT seconds (default 300) ends the job non-zero with a message instead of hanging the launcher
                guard = rendezvous_guard(float(os.environ.get("MPC_BENCH_COMM_TIMEOUT", "300")), rank)
                try:
                    uid = exchange_comm_id(dist, rank, mpc_gpu.BatchedMpc.comm_unique_id) if world > 1 else bytes(mpc_gpu.BatchedMpc.comm_unique_id())
                    loop.m.comm_init(rank, world, uid)
                except Exception as e:      # noqa: BLE001
                    ok, err = 0.0, f"{type(e).__name__}: {e}"[:200]
                ok, err = agree(ok, err)
                guard.cancel()
            if ok < 1.0:
                exchange_note = f"capi set-up failed on a rank ({err or 'another rank'})"
                try:
                    loop.m.comm_destroy()       # a rank whose mpc_comm_init succeeded keeps no live communicator
                except Exception:               # noqa: BLE001
                    pass
                if world == 1 or not args.allow_exchange_fallback:
                    sys.stderr.write(f"bench.py: {exchange_note}; --allow-exchange-fallback would run the torch.distributed exchange instead\n")
                    if dist is not None:
                        dist.destroy_process_group()
                    sys.exit(3)
                exchange, exchange_note = "torch", exchange_note + ": fell back to torch.distributed (--allow-exchange-fallback)"
        if exchange == "capi":
            exch = CApiCostExchange(torch, world, hi - lo, dev, loop.m, join=loop.join)
        else:
            exch = CostExchange(torch, world, hi - lo, dev, gather_costs, join=loop.join)
    r = measure(torch, dist, loop, world, exch, args.steps, args.warmup, dev)
    torch_pg = None
    if dist is not None:
        # the launcher's (or --with-torch-pg's) process group still works after the library's communicator has run beside it
        probe = torch.full((4,), float(rank + 1), dtype=torch.float64, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        torch_pg = {"backend": dist.get_backend(), "world": dist.get_world_size(), "all_reduce_after_exchange_ok": bool(float(probe[0].item()) == world * (world + 1) / 2)}
    rccl_maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l and "/" in l})
    gather_ok = None
    if exch is not None:         # the last completed message: on every rank, the rank's own block must be its own cost history, and every other block must have arrived
        got = exch.gathered()
        gather_ok = bool(torch.equal(got[rank], exch.hist[((exch.n - 1) // GATHER_EVERY) % 2]))
        gather_ok = gather_ok and all(bool(torch.isfinite(got[q]).all()) and float(got[q].abs().sum()) > 0.0 for q in range(world))      # every rank's block arrived
        if dist is not None:      # ... and the line reports the check of ALL ranks, not only rank 0's
            flag = torch.tensor([1.0 if gather_ok else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            gather_ok = bool(flag.item() == 1.0)

    value = G * EPISODE * args.steps / r["elapsed"]
    out = {"metric": "MPC solves/sec (N=20, 3 obstacles)" if N == 20 and no == 3 else f"MPC solves/sec (N={N}, {no} obstacles)",
           "value": value, "unit": "solves/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["elapsed"] / args.steps * 1e3, "higher_is_better": True,
           "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": desc, "global_batch": G, "per_gpu_batch": hi - lo, "N": N, "n_obst": no, "qp_tol": float(loop.m.cfg.qp_tol), "qp_iter_max": int(loop.m.cfg.qp_iter_max),
                      "step": f"one episode of the whole batch = set_initial_guess + {EPISODE} closed-loop control steps; a control step is ONE fused "
                              "launch (obstacle look-ahead + RTI solve + plant step + obstacle motion + warm-start shift), device resident",
                      "control_steps_per_step": EPISODE, "solves_per_step": G * EPISODE,
                      "comparability": ("whole episodes at qp_tol 1e-10 (round 3: the tolerance at which the reference's recorded closed loops come back best, DESIGN.md section 2; 11.2 interior-point "
                                        "iterations per solve); rounds 1-2 ran qp_tol 1e-8 (10.7 iterations: 7.4-7.5e6 solves/s with the same kernel)")
                                       if args.workload == "c2" else None,
                      "parallelism": (f"batch slices over {world} ranks (mpc_gpu.sharding.shard_slice), no data-path collective; per-scenario costs "
                                      f"all-gathered over RCCL, {GATHER_EVERY} control steps per message") if world > 1 else "single GPU"},
           "exchange": exchange, "exchange_note": exchange_note, "rccl_ranks": loop.m.comm_world() if exchange == "capi" else None, "gather_check": gather_ok,
           "rccl_path": rccl_path, "rccl_mapped": rccl_maps, "torch_pg": torch_pg,
           "streams_per_gpu": loop.streams,
           "ms_per_control_step": r["elapsed"] / (args.steps * EPISODE) * 1e3,
           "mean_ipm_iters": r["mean_iters"], "qp_failure_frac": r["fail"], "qp_iter_cap_frac": r["cap"],
           "lanes_per_instance": loop.m.lanes_per_instance(loop.B), "lanes_per_stage": loop.m.lanes_per_stage(loop.B),
           "waves_per_simd": loop.m.waves_per_simd(loop.B),
           "roofline": roofline(loop, N, no, r)}

    if rank == 0 and world == 1 and not args.no_extra and args.workload == "c2":
        # supplementary: the other BASELINE configurations on this one GPU.  Each is measured twice: on ONE stream (the kernel alone on the chip: `roofline`,
        # `value_one_stream`) and with its sub-batches pipelined on two streams (`value`: what `--workload` of that name reports; mpc_gpu.pipeline)
        def extra(wl, share, steps, what):
            Ne, noe = WORKLOADS[wl][:2]
            xe, ge, oe, de, _, _ = make_workload(wl, share, 0, shard_slice)
            Be = len(xe)
            l1 = Loop(mpc_gpu, torch, Ne, noe, xe, ge, oe, dev)
            r1 = measure(torch, None, l1, 1, None, steps, 1, dev)
            roof = roofline(l1, Ne, noe, r1)
            del l1
            K = pick_streams(Be)
            lK = Loop(mpc_gpu, torch, Ne, noe, xe, ge, oe, dev, streams=K)
            rK = measure(torch, None, lK, 1, None, steps, 1, dev)
            del lK
            e = {"workload": de + (f" -- rank 0's slice of {share} ({Be} instances) on this one GPU" if share > 1 else ""), "what": what,
                 "value": Be * EPISODE * steps / rK["elapsed"], "unit": "solves/s", "streams": K, "value_one_stream": Be * EPISODE * steps / r1["elapsed"],
                 "steps": steps, "warmup": 1, "ms_per_step": rK["elapsed"] / steps * 1e3, "ms_per_control_step": rK["elapsed"] / (steps * EPISODE) * 1e3,
                 "mean_ipm_iters": rK["mean_iters"], "qp_failure_frac": rK["fail"], "roofline": roof}
            if share > 1:
                e["predicted_value_on_%d_gpus" % share] = share * e["value"]
            return e
        out["extra"] = extra("c3", 1, 3, "BASELINE configs[2]: 65536 randomized scenarios on this GPU")
        out["value_throughput"] = out["extra"]["value"]
        out["value_throughput_config"] = "C3 (BASELINE configs[2]): 65536 randomized scenarios on this GPU -- `value` is the latency-shaped C2 (1024 scenarios = one wavefront per SIMD); `extra` has the details"
        out["extra_c5"] = extra("c5", 1, 2, "BASELINE configs[4] on ONE GPU: N = 50, 10 obstacles, 32768 scenarios")
        # the PER-GPU SHARES of the two 8-GPU configurations on this one GPU: what one rank of `--gpus 8 --workload c4 / c5` solves -- the strong-scaling
        # predictor (no data-path collective: 8 x the share rate is what 8 GPUs deliver if nothing else interferes)
        out["extra_c4_share"] = extra("c4", 8, 3, "BASELINE configs[3]: one rank's 32768 of 262144 scenarios")
        out["extra_c5_share"] = extra("c5", 8, 3, "BASELINE configs[4]: one rank's 4096 of 32768 scenarios (N = 50, 10 obstacles)")
        # ... and C2 at the tolerance rounds 1-2 ran (1e-8), for continuity with their lines: same kernel, 3-5 % fewer iterations
        l8 = Loop(mpc_gpu, torch, N, no, x0, goal, obst, dev, qp_tol=1e-8)
        r8 = measure(torch, None, l8, 1, None, 5, 1, dev)
        out["value_qp_tol_1e-8"] = {"value": G * EPISODE * 5 / r8["elapsed"], "unit": "solves/s", "mean_ipm_iters": r8["mean_iters"],
                                    "note": "the C2 workload with qp_tol = 1e-8, the default of rounds 1-2; `value` is at the round-3 default 1e-10 (DESIGN.md section 2)"}
        del l8
        # ... and C2 with the reference's reaction to a failed QP (status 4 -> set_initial_guess(), robot_ocp_problem.py:203-205, what experiments.py runs with
        # init_guess_when_error=True): `value` is the plain loop, in which an instance whose QP has become infeasible keeps failing (3 % of the solves)
        from mpc_gpu import _lib as L
        lr = Loop(mpc_gpu, torch, N, no, x0, goal, obst, dev, step_flags=L.STEP_SHIFT | L.STEP_PLANT | L.STEP_OBSTACLES | L.STEP_RESET_ON_FAIL | L.STEP_ALIAS_BUG)
        rr = measure(torch, None, lr, 1, None, 5, 1, dev)
        out["value_reset_on_fail"] = {"value": G * EPISODE * 5 / rr["elapsed"], "unit": "solves/s", "mean_ipm_iters": rr["mean_iters"], "qp_failure_frac": rr["fail"],
                                      "note": "the C2 workload with MPC_STEP_RESET_ON_FAIL | MPC_STEP_ALIAS_BUG (the reference's own protocol on a failed QP); every instance still solved 100 times per episode"}
        del lr
        out["c1"] = c1_latency(mpc_gpu, N, no)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(N, no, x0, goal, obst)
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
