"""The fused closed-loop control step (mpc_closed_loop_step_dev) against an ORACLE-side closed loop
(tests/helpers.py::OracleLoop = RobotOcpProblem.step's body, robot_ocp_problem.py:184-260, on the oracle's functions), plus the
sub-goal hook (set_subgoal, :279-284; x_N read-back, :232) and the explicit slack schedule (parameterize_slack, :145-152)."""
import numpy as np
import pytest

from helpers import OracleLoop, adjudicate, allowed_adjudications, oracle_P, oracle_guess, random_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["stage-split", "one-lane-per-stage"])
def env(built, request):
    import mpc_gpu
    from oracle import oracle as orc
    mpc_gpu.BatchedMpc.default_lanes_per_stage = 0 if request.param == "stage-split" else 1
    yield mpc_gpu, orc
    mpc_gpu.BatchedMpc.default_lanes_per_stage = 0


class GpuLoop:
    """device-resident closed loop on the handle-owned iterate, one launch per control step"""

    def __init__(self, mpc_gpu, N, no, Tf, x0, goal, obst, alias=True, **cfg):
        import torch
        from mpc_gpu import _lib
        self.torch, self.B = torch, x0.shape[0]
        self.m = mpc_gpu.BatchedMpc(N, no, Tf, max_batch=self.B, **cfg)
        dev = torch.device("cuda:0")
        self.stream = torch.cuda.Stream(device=dev)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        z = lambda *s, dt=torch.float64: torch.zeros(*s, dtype=dt, device=dev)
        with torch.cuda.stream(self.stream):
            self.x0, self.goal, self.obst = t(x0), t(goal), t(obst)
            if alias:
                self.x0[:, 3:] = 0.0
            self.u0, self.cost = z(self.B, 2), z(self.B)
            self.status, self.iters = z(self.B, dt=torch.int32), z(self.B, dt=torch.int32)
            self.margin = torch.full((self.B,), float("inf"), dtype=torch.float64, device=dev)
            self.flags, self.steps = z(self.B, dt=torch.int32), z(self.B, dt=torch.int32)
        self.stream.synchronize()
        self.dX, self.dU, _ = self.m.iterate_ptrs()              # the handle-owned iterate: terminal_state() / get_traj() see it
        self.m.reset_guess_dev(self.B, self.x0, self.dX, self.dU, stream=self.stream.cuda_stream)
        self.fl = (_lib.STEP_SHIFT | _lib.STEP_PLANT | _lib.STEP_OBSTACLES | _lib.STEP_METRICS | _lib.STEP_RESET_ON_FAIL
                   | (_lib.STEP_ALIAS_BUG if alias else 0))
        self.dev = dev

    def step(self, noise=None):
        nz = None if noise is None else self.torch.from_numpy(np.ascontiguousarray(noise)).to(self.dev)
        self.m.closed_loop_step_dev(self.B, self.x0, self.obst, self.goal, self.dX, self.dU, self.u0, self.cost, self.status, self.iters,
                                    nz, flags=self.fl, min_margin=self.margin, ep_flags=self.flags, ep_steps=self.steps,
                                    stream=self.stream.cuda_stream)
        self.stream.synchronize()

    def host(self):
        c = lambda a: a.cpu().numpy()
        X, U = self.m.get_traj(self.B)
        return dict(x0=c(self.x0), obst=c(self.obst), X=X, U=U, u0=c(self.u0), status=c(self.status), iters=c(self.iters),
                    margin=c(self.margin), flags=c(self.flags), steps=c(self.steps))

    def set_goal(self, goal):
        with self.torch.cuda.stream(self.stream):
            self.goal.copy_(self.torch.from_numpy(np.ascontiguousarray(goal)).to(self.dev))
        self.stream.synchronize()

    def close(self):
        self.m.close()


def test_fused_step_against_the_oracle_loop_with_resync(env):
    """24 instances x 70 control steps, moving noisy obstacles that interact with the robots (the reference's RANDOM draws), the
    oracle loop re-seeded with the GPU's state before every step: every control step is then ONE comparison of everything the
    fused launch does -- solve, u*, plant state, obstacle states (bit for bit), shifted warm start, margin, flags, step counter.
    At step 25 every instance gets a new sub-goal (set_subgoal); x_N is read back (terminal_state) after every step."""
    mpc_gpu, orc = env
    from mpc_gpu.world import reference_streams
    N, no, Tf, B, K = 20, 5, 2.0, 24, 70
    obst, noise = reference_streams("RANDOM", range(B), no, K)
    x0 = np.tile([-7.0, -7.0, np.pi / 4, 0, 0], (B, 1)); goal = np.tile([7.0, 7.0], (B, 1))
    x0[B // 2:, :2] = [[-6.0, 5.0]]; goal[B // 2:] = [5.0, -6.0]
    cfg = orc.config(N, no, Tf)
    g = GpuLoop(mpc_gpu, N, no, Tf, x0, goal, obst)
    loops = [OracleLoop(orc, cfg, x0[b], goal[b], obst[b]) for b in range(B)]
    worst = dict(X=0.0, x=0.0, u=0.0, margin=0.0)
    n_fail = n_cmp = n_out = 0
    for k in range(K):
        if k == 25:
            goal = goal[:, ::-1].copy() * 0.8                       # mid-episode sub-goal change on both sides
            g.set_goal(goal)
            for L in loops:
                L.goal = goal[loops.index(L)].copy()
        before = g.host()
        for b, L in enumerate(loops):                               # resync: the oracle continues from the GPU's state
            L.x, L.obst = before["x0"][b].copy(), before["obst"][b].copy()
            L.X, L.U = before["X"][b].copy(), before["U"][b].copy()
            L.min_margin, L.flags, L.steps = float(before["margin"][b]), int(before["flags"][b]), int(before["steps"][b])
        g.step(noise[k])
        after = g.host()
        xN = g.m.terminal_state(B)
        assert np.array_equal(xN, after["X"][:, -1])
        for b, L in enumerate(loops):
            r = L.step(noise[k, b])
            if r is None:                                           # finished episodes idle on both sides
                for key in ("x0", "obst", "X", "U"):
                    assert np.array_equal(after[key][b], before[key][b]), (k, b, key)
                assert after["steps"][b] == before["steps"][b]
                continue
            n_cmp += 1
            assert after["status"][b] == r["status"], (k, b, after["status"][b], r["status"])
            assert np.array_equal(after["obst"][b], L.obst), (k, b)                 # same noise, IEEE-exact obstacle motion
            assert after["flags"][b] == L.flags and after["steps"][b] == L.steps, (k, b)
            if r["status"] == 4:
                n_fail += 1
            if r["status"] != 0:
                continue                                            # capped / failed QPs: iterates need not agree (the statuses did)
            d = max(np.abs(after["X"][b] - L.X).max(), np.abs(after["U"][b] - L.U).max())
            if d > 1e-6:
                # an ill-conditioned QP at the float64 floor of the interior point: adjudicated against the EXACT solution of the
                # QP the oracle assembles from the same inputs (helpers.adjudicate) -- the GPU's step, un-shifted, is within a factor of the oracle's
                # distance from it and below the hard cap
                n_out += 1
                Xn = np.vstack([before["x0"][b][None], after["X"][b][:N]]); Un = np.vstack([after["u0"][b][None], after["U"][b][:N - 1]])
                P = orc.predict_params(cfg, before["obst"][b])
                a = adjudicate(orc, cfg, before["x0"][b], P, goal[b], before["X"][b], before["U"][b], Xn, Un, r["X"], r["U"])
                assert a["passed"], (k, b, d, a)
                continue
            worst["X"] = max(worst["X"], d)
            worst["x"] = max(worst["x"], np.abs(after["x0"][b] - L.x).max())
            worst["u"] = max(worst["u"], np.abs(after["u0"][b] - r["u0"]).max())
            worst["margin"] = max(worst["margin"], abs(after["margin"][b] - L.min_margin))
            assert abs(after["iters"][b] - r["iters"]) <= 1, (k, b)
    g.close()
    assert n_cmp > 0.7 * B * K and n_out <= max(2, 0.002 * n_cmp), (n_cmp, n_out)
    assert worst["X"] <= 1e-6 and worst["u"] <= 8e-6 and worst["x"] <= 1e-6 and worst["margin"] <= 1e-6, worst


def test_free_running_episodes_three_way(env):
    """No resync: the GPU episode harness, the oracle loop and the RECORDED reference rows for 12 seeds on which acados' QP always
    converged -- three independent computations of the same closed loop land on the same table rows."""
    import json
    import os
    mpc_gpu, orc = env
    from mpc_gpu.world import reference_streams
    rows = np.array(json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_tables.json")))["tables"]["20221031_215846"]["rows"])
    seeds = [0, 2, 3, 4, 5, 24, 25, 36, 41, 53, 63, 65]
    N, no, Tf = 20, 5, 2.0
    obst, noise = reference_streams("RANDOM", seeds, no, 400)
    B = len(seeds)
    x0 = np.tile([-7.0, -7.0, np.pi / 4, 0, 0], (B, 1)); goal = np.tile([7.0, 7.0], (B, 1))
    tb = mpc_gpu.run_episodes(x0, goal, obst, N=N, Tf=Tf, max_iter=400, noise=noise, qp_iter_max=100)["table"]
    cfg = orc.config(N, no, Tf, qp_iter_max=100)
    for b in range(B):
        L = OracleLoop(orc, cfg, x0[b], goal[b], obst[b])
        for k in range(400):
            if L.step(noise[k, b]) is None:
                break
        o = np.array(L.row())
        assert np.array_equal(o[[0, 1, 4, 5]], tb[b, [0, 1, 4, 5]]) and np.abs(o[2:4] - tb[b, 2:4]).max() <= 1e-5, (seeds[b], o, tb[b])
        assert np.array_equal(tb[b, [0, 1, 4, 5]], rows[seeds[b], [0, 1, 4, 5]]) and np.abs(tb[b, 2:4] - rows[seeds[b], 2:4]).max() <= 1e-5


def test_explicit_slack_schedule(env):
    """mpc_set_slack_schedule: (i) the reference's own schedule passed explicitly changes nothing (to rounding); (ii) a different
    schedule (constant weight, zero on the last three stages) agrees with the oracle given the same weights; (iii) NULL restores
    the built-in schedule; (iv) the shim forwards cost_set('zl'|'Zl') and refuses what the kernel cannot represent."""
    mpc_gpu, orc = env
    N, no, B = 20, 3, 48
    x0, goal, obst = random_batch(B, no, seed=77)
    cfg = orc.config(N, no, 2.0)
    P = oracle_P(orc, cfg, obst); Xg, Ug = oracle_guess(orc, cfg, x0)
    own = np.stack([orc.slack_alpha(cfg, x0[b], goal[b]) for b in range(B)])
    other = np.full((B, N + 1), 3.0e5); other[:, -3:] = 0.0
    with mpc_gpu.BatchedMpc(N, no, 2.0, max_batch=B) as s:
        s.set_warmstart(Xg, Ug); a = s.solve(x0, P, goal); Xa, Ua = s.get_traj(B)
        s.set_slack_schedule(own)
        s.set_warmstart(Xg, Ug); b = s.solve(x0, P, goal); Xb, Ub = s.get_traj(B)
        # (to rounding: the kernel evaluates the schedule with fused multiply-adds, the host formula without)
        assert np.abs(Xa - Xb).max() <= 1e-9 and np.abs(Ua - Ub).max() <= 1e-9 and np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["status"], b["status"])
        s.set_slack_schedule(other)
        s.set_warmstart(Xg, Ug); c = s.solve(x0, P, goal); Xc, Uc = s.get_traj(B)
        s.set_slack_schedule(None)
        s.set_warmstart(Xg, Ug); d = s.solve(x0, P, goal); Xd, _ = s.get_traj(B)
        assert np.array_equal(Xa, Xd)
        with pytest.raises(mpc_gpu.MpcError):
            s.set_slack_schedule(-other)
        # a schedule uploaded for k instances covers k instances: the rows behind them were never written, so a larger solve is refused
        # (it used to read them as slack weights) -- and works again once the schedule is dropped or uploaded for the whole batch
        s.set_slack_schedule(other[:16])
        s.set_warmstart(Xg[:16], Ug[:16]); e = s.solve(x0[:16], P[:16], goal[:16])
        assert np.array_equal(e["status"], c["status"][:16]) and np.array_equal(e["iters"], c["iters"][:16])
        with pytest.raises(mpc_gpu.MpcError, match="covers fewer instances"):
            s.solve(x0, P, goal)
        s.set_slack_schedule(other)
        s.set_warmstart(Xg, Ug); f = s.solve(x0, P, goal)
        assert np.array_equal(f["status"], c["status"]) and np.array_equal(f["u0"], c["u0"])
    moved = 0
    for i in range(B):
        r = orc.rti_solve(cfg, x0[i], P[i], goal[i], Xg[i], Ug[i], alpha=other[i])
        assert r["status"] == c["status"][i]
        if r["status"] == 0:
            assert np.abs(r["X"] - Xc[i]).max() <= 1e-6 and np.abs(r["U"] - Uc[i]).max() <= 8e-6
            assert abs(r["cost"] - c["cost"][i]) <= 1e-8 * max(1.0, abs(r["cost"]))
            moved += np.abs(Xc[i] - Xa[i]).max() > 1e-4
    assert moved >= 3                                               # the schedule matters wherever an obstacle row is active
    # the shim: what the caller sets is what the solve uses
    from mpc_gpu.acados_shim import AcadosOcpSolverShim
    with mpc_gpu.BatchedMpc(N, no, 2.0, max_batch=1) as m:
        sh = AcadosOcpSolverShim(N, no, 2.0, goal=goal[0], x0=x0[0], mpc=m)
        for i in range(N + 1):
            sh.set(i, "x", Xg[0, i]); sh.set(i, "p", P[0, i].reshape(-1))
            sh.cost_set(i, "zl", other[0, i] * np.ones(no)); sh.cost_set(i, "Zl", other[0, i] * np.ones(no))
            if i < N:
                sh.set(i, "u", Ug[0, i])
        assert sh.solve() == c["status"][0]
        assert np.array_equal(np.array([sh.get(i, "x") for i in range(N + 1)]), Xc[0])
        sh.cost_set(3, "Zl", 2 * other[0, 3] * np.ones(no))
        with pytest.raises(ValueError):
            sh.solve()                                              # zl != Zl is not the reference's schedule
        with pytest.raises(ValueError):
            sh.cost_set(2, "zl", np.array([1.0, 2.0, 3.0]))         # per-obstacle weights neither


def test_outliers_are_judged_by_the_qp_not_by_a_count(env):
    """N = 50 with 10 obstacles (C5's problem): a percent of the instances differ from the oracle by more than 1e-6 -- ill-conditioned
    QPs at the float64 floor of any interior point.  For EVERY converged instance the GPU's step must be a point of
    the QP the oracle assembles that is as good as the oracle's own: dynamics satisfied, inside the boxes, objective not worse."""
    mpc_gpu, orc = env
    N, no, B = 50, 10, 96
    x0, goal, obst = random_batch(B, no, seed=1234)
    cfg = orc.config(N, no, 5.0)
    P = oracle_P(orc, cfg, obst); Xg, Ug = oracle_guess(orc, cfg, x0)
    with mpc_gpu.BatchedMpc(N, no, 5.0, max_batch=B) as s:
        s.set_warmstart(Xg, Ug); g = s.solve(x0, P, goal); X1, U1 = s.get_traj(B)
        s.shift(B); Xs, Us = s.get_traj(B)
        g2 = s.solve(x0, P, goal); X2, U2 = s.get_traj(B)
    o1 = orc.rti_solve_batch(cfg, x0, P, goal, Xg, Ug)
    o2 = orc.rti_solve_batch(cfg, x0, P, goal, Xs, Us)
    n_out = 0
    for (gg, Xa, Ua, oo, Xin, Uin) in ((g, X1, U1, o1, Xg, Ug), (g2, X2, U2, o2, Xs, Us)):
        assert (gg["status"] == oo["status"]).mean() >= 0.97
        for b in np.nonzero((gg["status"] == 0) & (oo["status"] == 0))[0]:
            d = np.abs(Xa[b] - oo["X"][b]).max()
            if d <= 1e-6:
                continue
            n_out += 1
            a = adjudicate(orc, cfg, x0[b], P[b], goal[b], Xin[b], Uin[b], Xa[b], Ua[b], oo["X"][b], oo["U"][b])
            assert a["passed"], (b, d, a)
    assert n_out <= 2 * allowed_adjudications(cfg, B), n_out
    print("outliers adjudicated against the exact QP solution:", n_out)


def test_interpolate_init_guess(env):
    """The reference's other set_initial_guess() -- the commented straight-line block robot_ocp_problem.py:293-300 behind the two `interpolate_init`
    tables: (i) mpc_reset_guess_interp equals the oracle's restatement AND plain numpy of the reference's expressions bit for bit (its slips included:
    x does not move, psi = arctan2(dy, 0)); (ii) the fused control step with MPC_STEP_INTERP_GUESS writes that guess on a failed QP: episodes with the
    iteration cap at 3 (most solves end with status 2 or 4) against the oracle loop with the same guess, resynchronised per step."""
    import torch
    mpc_gpu, orc = env
    from mpc_gpu import _lib
    N, no, B = 20, 3, 48
    x0, goal, obst = random_batch(B, no, seed=123)
    x0[:, 3:] = np.random.default_rng(3).uniform(-1, 1, (B, 2)); goal[5, 1] = x0[5, 1]      # one instance with dy = 0
    cfg = orc.config(N, no, 2.0)
    with mpc_gpu.BatchedMpc(N, no, 2.0, max_batch=B) as s:
        s.reset_guess_interp(x0, goal)
        X, U = s.get_traj(B)
    i = np.arange(N + 1)
    for b in range(B):
        Xo, Uo = orc.initial_guess_interp(cfg, x0[b], goal[b])
        want = np.zeros((N + 1, 5))
        want[:, 0] = x0[b, 0] + i / N * (x0[b, 0] - x0[b, 0]); want[:, 1] = x0[b, 1] + i / N * (goal[b, 1] - x0[b, 1])
        want[:, 2] = np.arctan2(goal[b, 1] - x0[b, 1], goal[b, 0] - goal[b, 0])
        assert np.array_equal(X[b], want) and np.array_equal(Xo, want) and not U[b].any() and not Uo.any()
    assert X[5, 0, 2] == 0.0 and abs(abs(X[0, 0, 2]) - np.pi / 2) < 1e-15
    # fused reset: HARD obstacle rows and an obstacle parked on top of the robot make the QP of every second instance infeasible (status 4);
    # after every step both sides must hold the same iterate (the shifted solve, or the shifted straight line)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x0[:, 3:] = 0.0
    obst[::2, 0, :2] = x0[::2, :2] + 0.4; obst[::2, 0, 2:] = 0.0
    cfgh = orc.config(N, no, 2.0, soft_h=0)
    loops = [OracleLoop(orc, cfgh, x0[b], goal[b], obst[b], reset_on_fail=True, alias=False, interp=True) for b in range(B)]
    fl = _lib.STEP_SHIFT | _lib.STEP_PLANT | _lib.STEP_OBSTACLES | _lib.STEP_RESET_ON_FAIL | _lib.STEP_INTERP_GUESS
    resets = 0
    with mpc_gpu.BatchedMpc(N, no, 2.0, max_batch=B, soft_h=0) as s, torch.cuda.stream(torch.cuda.Stream(device=dev)):
        st = torch.cuda.current_stream().cuda_stream
        dx, dg, do = t(x0), t(goal), t(obst)
        X = torch.zeros(B, N + 1, 5, dtype=torch.float64, device=dev); U = torch.zeros(B, N, 2, dtype=torch.float64, device=dev)
        status = torch.zeros(B, dtype=torch.int32, device=dev)
        s.reset_guess_interp_dev(B, dx, dg, X, U, stream=st)
        for k in range(4):
            s.closed_loop_step_dev(B, dx, do, dg, X, U, None, None, status, None, None, flags=fl, stream=st)
            torch.cuda.synchronize()
            Xh, Uh, sh = X.cpu().numpy(), U.cpu().numpy(), status.cpu().numpy()
            for b in range(B):
                r = loops[b].step()
                assert r["status"] == sh[b], (k, b, r["status"], sh[b])
                resets += r["status"] == 4
                if r["status"] == 4:
                    assert np.array_equal(loops[b].X, Xh[b]) and not Uh[b].any()          # the straight line itself, shifted: exact
                elif r["status"] == 0:
                    assert np.abs(loops[b].X - Xh[b]).max() <= 1e-6 and np.abs(loops[b].U - Uh[b]).max() <= 8e-6, (k, b)
                loops[b].X, loops[b].U = Xh[b].copy(), Uh[b].copy()  # resynchronise: every comparison is one step on identical inputs
                loops[b].x = dx[b].cpu().numpy().copy(); loops[b].obst = do[b].cpu().numpy().copy()
    assert resets >= 5, resets


def test_c2_bench_workload_against_the_oracle_loop(built):
    """The HEADLINE workload itself: bench.py's C2 scenario -- make_workload("c2"): 1024 copies of the reference generator's seed-0
    RANDOM draw with its drawn velocities, x0 = [-6, -6, pi/4, 0, 0], goal [6, 6] -- driven exactly as bench.py's timed loop drives it (Loop.reset() +
    100 fused control steps with the plain flags: no reset after a failed QP, no stop at the goal; the dispatcher's own kernel choice for 1024 instances),
    and slot 0 and slot 1023 judged STEP BY STEP against tests/helpers.py::OracleLoop (robot_ocp_problem.py:184-260 on the oracle's functions, :195 the
    solve): status, u*, plant state, obstacle states (bit for bit) and the shifted iterate after every control step.  The oracle loop is re-seeded with
    the GPU's state before each step, so every step is one comparison on identical inputs; all 1024 slots must stay bitwise equal to slot 0."""
    import torch
    import bench
    import mpc_gpu
    from oracle import oracle as orc
    from mpc_gpu.sharding import shard_slice
    N, no = bench.WORKLOADS["c2"][:2]
    x0, goal, obst, desc, (lo, hi), G = bench.make_workload("c2", 1, 0, shard_slice)
    B = hi - lo
    assert B == 1024 and G == 1024 and desc.startswith("C2")
    dev = torch.device("cuda", 0)
    cfg = orc.config(N, no, 0.1 * N)
    slots = (0, B - 1)
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        loop = bench.Loop(mpc_gpu, torch, N, no, x0, goal, obst, dev, streams=bench.pick_streams(B))
        assert loop.streams == 1 and loop.m.kernel_name(B).startswith("rti_split_kernel<3, 3")      # the kernel the bench line names
        loop.reset()
        torch.cuda.current_stream().synchronize()
        ref = {b: OracleLoop(orc, cfg, x0[b], goal[b], obst[b], reset_on_fail=False, alias=False) for b in slots}
        host = lambda t: t.cpu().numpy()
        n_conv = n_fail = n_adj = 0
        worst = 0.0
        for k in range(bench.EPISODE):
            before = dict(x0=host(loop.x0), obst=host(loop.obst), X=host(loop.X), U=host(loop.U))
            loop.control_step()
            torch.cuda.current_stream().synchronize()
            after = dict(x0=host(loop.x0), obst=host(loop.obst), X=host(loop.X), U=host(loop.U), u0=host(loop.u0), status=host(loop.status), iters=host(loop.iters))
            for key in ("x0", "obst", "X", "U", "u0", "status", "iters"):      # identical scenarios: identical results in every slot
                assert (after[key] == after[key][0:1]).all(), (k, key)
            for b, L in ref.items():
                L.x, L.obst, L.X, L.U = before["x0"][b].copy(), before["obst"][b].copy(), before["X"][b].copy(), before["U"][b].copy()
                L.flags = 0                                                       # the plain loop never stops at the goal
                r = L.step()
                assert after["status"][b] == r["status"], (k, b, after["status"][b], r["status"], after["iters"][b], r["iters"])
                assert np.array_equal(after["obst"][b], L.obst), (k, b)
                if r["status"] == 4:
                    n_fail += 1
                    assert np.array_equal(after["u0"][b], before["U"][b][0])      # a failed QP applies the stored u_0 and leaves the iterate (shifted) as it was
                if r["status"] != 0:
                    continue
                n_conv += 1
                d = max(np.abs(after["X"][b] - L.X).max(), np.abs(after["U"][b] - L.U).max(), np.abs(after["x0"][b] - L.x).max())
                if d > 1e-6:
                    # the same adjudication as everywhere else (helpers.adjudicate: the GPU's step against the EXACT solution of the QP, cap EXACT_CAP), on the
                    # un-shifted step: the fused control step stores the iterate shifted, so row 0 of the step is x0 (the initial-state equality), rows 1 .. N are rows
                    # 0 .. N-1 of what was stored, the first input is the applied u0 and inputs 1 .. N-1 are stored inputs 0 .. N-2 (robot_ocp_problem.py:253-258)
                    P = orc.predict_params(cfg, before["obst"][b])
                    rr = orc.rti_solve(cfg, before["x0"][b], P, goal[b], before["X"][b], before["U"][b])
                    Xg = np.vstack([before["x0"][b][None], after["X"][b][:N]]); Ug = np.vstack([after["u0"][b][None], after["U"][b][:N - 1]])
                    a = adjudicate(orc, cfg, before["x0"][b], P, goal[b], before["X"][b], before["U"][b], Xg, Ug, rr["X"], rr["U"])
                    assert a["passed"], (k, b, d, a)
                    n_adj += 1
                else:
                    assert np.abs(after["u0"][b] - r["u0"]).max() <= 8e-6, (k, b)
                worst = max(worst, d)
        del loop
    assert n_conv >= 150 and n_adj <= 1, (n_conv, n_fail, n_adj, worst)
    print(f"C2 bench workload, slots {slots}: {n_conv} converged control steps compared, {n_fail} failed QPs (status equal), {n_adj} beyond 1e-6, worst |GPU - oracle| {worst:.2e}")
