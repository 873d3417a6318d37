"""GPU parity tests: HIP path through the C ABI vs the CPU oracle on identical inputs.

Tolerances (BASELINE.md section 3 / SURVEY.md 8(c)): linearisation 1e-12; trajectories |X - X_ref| <= 1e-6;
|u* - u*_ref| <= 1e-6 * C_MAX; cost relative 1e-8.  Both sides run the same interior-point specification with
qp_tol = 1e-10 (the library's and the oracle's default) and cap QP_ITER = 50, so the observed differences are far below these bounds.
"""
import numpy as np
import pytest

from helpers import adjudicate, adjudicate_batch, exact_qp, judge_against_oracle, oracle_P, oracle_guess, qp_merit, random_batch, step_vector

pytestmark = pytest.mark.gpu

TOL_X = 1e-6
TOL_U = 1e-6 * 8.0
TOL_LIN = 1e-12


@pytest.fixture(params=["stage-split", "one-lane-per-stage"])
def env(built, request):
    """Every parity test runs on both lane mappings of the solve kernel: the automatic choice (these batches are small, so the
    rows of a stage are split over 2-3 lanes wherever the horizon fits: rti_split_kernel) and one lane per stage
    (rti_solve_kernel, what large batches use)."""
    import mpc_gpu
    from oracle import oracle as orc
    mpc_gpu.BatchedMpc.default_lanes_per_stage = 0 if request.param == "stage-split" else 1
    yield mpc_gpu, orc
    mpc_gpu.BatchedMpc.default_lanes_per_stage = 0


def run_pair(mpc_gpu, orc, N, no, Tf, x0, goal, obst, steps=1, X=None, U=None, resync=True, **cfgkw):
    """steps RTI iterations with warm-start shift on both sides.  resync=True: before every solve the GPU is given the
    oracle's iterate, so each comparison is a single solve on IDENTICAL inputs (the parity contract); resync=False lets
    each side carry its own iterate (differences then compound through the closed loop)."""
    B = x0.shape[0]
    cfg = orc.config(N, no, Tf, **cfgkw)
    P = oracle_P(orc, cfg, obst)
    if X is None:
        X, U = oracle_guess(orc, cfg, x0)
    Xo, Uo = X.copy(), U.copy()
    outs = []
    with mpc_gpu.BatchedMpc(N, no, Tf, max_batch=B, **cfgkw) as s:
        s.set_warmstart(X, U)
        for k in range(steps):
            if resync and k > 0:
                s.set_warmstart(Xo, Uo)
            g = s.solve(x0, P, goal)
            Xg, Ug = s.get_traj(B)
            o = orc.rti_solve_batch(cfg, x0, P, goal, Xo, Uo)
            o["start"] = (Xo.copy(), Uo.copy()); o["cfg"], o["P"] = cfg, P
            Xo, Uo = o["X"].copy(), o["U"].copy()
            outs.append((g, Xg, Ug, o))
            if k + 1 < steps:
                s.shift(B)
                for b in range(B):
                    Xo[b], Uo[b] = orc.shift(cfg, Xo[b], Uo[b])
    return outs


def assert_close(g, Xg, Ug, o, allow_status_mismatch=0):
    ok = (g["status"] == o["status"])
    assert (~ok).sum() <= allow_status_mismatch, (g["status"][~ok], o["status"][~ok])
    # status 2 at the full cap means the interior point did not converge (infeasible QP): both sides then hold
    # unconverged iterates that need not agree; test_iteration_cap_status compares capped runs at a low cap instead
    sel = ok & (o["status"] == 0)
    assert sel.any()
    assert np.abs(Xg[sel] - o["X"][sel]).max() <= TOL_X
    assert np.abs(Ug[sel] - o["U"][sel]).max() <= TOL_U
    assert np.abs(g["u0"][sel] - o["u0"][sel]).max() <= TOL_U
    rel = np.abs(g["cost"][sel] - o["cost"][sel]) / np.maximum(1.0, np.abs(o["cost"][sel]))
    assert rel.max() <= 1e-8
    # failed QPs leave the iterate untouched on both sides
    bad = ok & (o["status"] == 4)
    if bad.any():
        assert np.abs(Xg[bad] - o["X"][bad]).max() == 0.0


def test_linearize_parity(env):
    mpc_gpu, orc = env
    import torch
    N, no, B = 20, 3, 64
    x0, goal, obst = random_batch(B, no, seed=3)
    cfg = orc.config(N, no, 2.0)
    P = oracle_P(orc, cfg, obst)
    rng = np.random.default_rng(5)
    X = rng.uniform(-7, 7, (B, N + 1, 5)); X[:, :, 2] = rng.uniform(-6, 6, (B, N + 1)); U = rng.uniform(-8, 8, (B, N, 2))
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    z = lambda *s: torch.zeros(*s, dtype=torch.float64, device=dev)
    with mpc_gpu.BatchedMpc(N, no, 2.0, max_batch=B) as s:
        A, Bm, b, q, hv, dh = z(B, N, 5, 5), z(B, N, 5, 2), z(B, N, 5), z(B, N + 1, 7), z(B, N + 1, no), z(B, N + 1, no, 2)
        args = [t(x0), t(P), t(goal), t(X), t(U)]
        s.linearize_dev(B, *args, A, Bm, b, q, hv, dh, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    for i in range(B):
        ref = orc.linearize(cfg, x0[i], P[i], goal[i], X[i], U[i])
        for name, got in (("A", A), ("B", Bm), ("b", b), ("q", q), ("h", hv), ("dh", dh)):
            assert np.abs(got[i].cpu().numpy() - ref[name]).max() <= TOL_LIN * max(1.0, np.abs(ref[name]).max()), name


@pytest.mark.parametrize("N,no,Tf", [(20, 3, 2.0), (10, 5, 1.0), (50, 10, 5.0), (5, 3, 0.5)])
def test_first_solve_parity(env, N, no, Tf):
    """cold RTI step from set_initial_guess() on randomized scenarios (config C3 distribution)"""
    mpc_gpu, orc = env
    B = 256 if N <= 20 else 64
    x0, goal, obst = random_batch(B, no, seed=100 + N)
    (g, Xg, Ug, o), = run_pair(mpc_gpu, orc, N, no, Tf, x0, goal, obst)
    n = judge_against_oracle(orc, o["cfg"], x0, o["P"], goal, o["start"][0], o["start"][1], g, Xg, Ug, o)
    assert n["converged"] >= 0.9 * B and n["status_borderline"] == 0, n


def test_closed_loop_sequence_parity(env):
    """10 consecutive RTI steps with warm-start shift; each side carries its own iterate"""
    mpc_gpu, orc = env
    N, no, B = 20, 3, 128
    x0, goal, obst = random_batch(B, no, seed=7)
    outs = run_pair(mpc_gpu, orc, N, no, 2.0, x0, goal, obst, steps=10)
    for g, Xg, Ug, o in outs:
        n = judge_against_oracle(orc, o["cfg"], x0, o["P"], goal, o["start"][0], o["start"][1], g, Xg, Ug, o)
        assert n["converged"] >= 0.9 * B, n


def test_c1_static_obstacles(env):
    """BASELINE config 1: x0 = [-6,-6,pi/4,0,0], goal [6,6], 3 static obstacles (first 3 of the seed-0 RANDOM draw)"""
    mpc_gpu, orc = env
    gold = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "reference_vectors.npz"))
    obst = gold["gen_RANDOM_3"][0:1].copy(); obst[:, :, 2:] = 0.0
    x0 = np.array([[-6.0, -6.0, np.pi / 4, 0, 0]]); goal = np.array([[6.0, 6.0]])
    outs = run_pair(mpc_gpu, orc, 20, 3, 2.0, x0, goal, obst, steps=5)
    for g, Xg, Ug, o in outs:
        assert_close(g, Xg, Ug, o)


def test_identical_scenarios_give_identical_outputs(env):
    """config C2 property: 1024 copies of one scenario -> bitwise identical results in every slot"""
    mpc_gpu, orc = env
    N, no, B = 20, 3, 1024
    x0, goal, obst = random_batch(1, no, seed=11)
    x0, goal, obst = np.repeat(x0, B, 0), np.repeat(goal, B, 0), np.repeat(obst, B, 0)
    with mpc_gpu.BatchedMpc(N, no, 2.0, max_batch=B) as s:
        s.reset_guess(x0)
        g = s.solve(x0, obst, goal)
        X, U = s.get_traj(B)
    assert (X == X[0]).all() and (U == U[0]).all() and (g["cost"] == g["cost"][0]).all()
    assert (g["status"] == 0).all()


def test_permutation_invariance_large_batch(env):
    """size-independent property at a large batch (config C3 scale-down: 16384): instances are independent"""
    mpc_gpu, orc = env
    N, no, B = 20, 3, 16384
    x0, goal, obst = random_batch(B, no, seed=21)
    perm = np.random.default_rng(0).permutation(B)
    with mpc_gpu.BatchedMpc(N, no, 2.0, max_batch=B) as s:
        lanes = s.lanes_per_instance(B)
        s.reset_guess(x0); g1 = s.solve(x0, obst, goal); X1, U1 = s.get_traj(B)
        s.reset_guess(x0[perm]); g2 = s.solve(x0[perm], obst[perm], goal[perm]); X2, U2 = s.get_traj(B)
    if lanes != 21:
        assert (X1[perm] == X2).all() and (U1[perm] == U2).all() and (g1["status"][perm] == g2["status"]).all()
    else:
        # three instances per wavefront: the wavefront sums of an instance are formed in an order that depends on which third of the
        # wavefront it occupies (the instances do not coincide with DPP rows), so a permuted batch agrees to rounding, not bit for bit
        same = g1["status"][perm] == g2["status"]
        assert same.mean() > 0.999
        d = np.abs(X1[perm] - X2).reshape(B, -1).max(1)[same & (g2["status"] == 0)]
        assert np.median(d) < 1e-12 and np.quantile(d, 0.999) < 1e-6 and d.max() < 1e-5
    # spot-check 64 of them against the oracle
    cfg = orc.config(N, no, 2.0)
    idx = perm[:64]
    P = oracle_P(orc, cfg, obst[idx]); Xg, Ug = oracle_guess(orc, cfg, x0[idx])
    o = orc.rti_solve_batch(cfg, x0[idx], P, goal[idx], Xg, Ug)
    sel = o["status"] != 4
    assert np.abs(o["X"][sel] - X1[idx][sel]).max() <= TOL_X


def test_obstacle_inside_safety_margin_and_bounds(env):
    """edge cases: robot starts deep inside an obstacle's margin (h << 0), on the state box, inputs saturating"""
    mpc_gpu, orc = env
    N, no = 20, 3
    x0 = np.array([[0.0, 0.0, 0.3, 0, 0], [6.9, 6.9, 0.7, 9.0, 0], [-6.99, 0.0, 3.0, 5.0, 2.0], [0, 0, 0, 0, 0]])
    goal = np.array([[3.0, 1.0], [-6.0, -6.0], [6.0, 0.0], [0.05, 0.0]])
    obst = np.zeros((4, no, 4))
    obst[0, :, :2] = [[0.3, 0.2], [1.5, 1.0], [-3, 3]]
    obst[1, :, :2] = [[5, 5], [0, 0], [-3, 3]]
    obst[2, :, :2] = [[-5, 0.5], [0, 0], [3, 3]]
    obst[3, :, :2] = [[5, 5], [-5, 5], [5, -5]]
    outs = run_pair(mpc_gpu, orc, N, no, 2.0, x0, goal, obst, steps=3)
    for g, Xg, Ug, o in outs:
        assert_close(g, Xg, Ug, o)


def test_iteration_cap_status(env):
    """qp_iter_max = 3: both sides stop at the cap with status 2 and still apply the (identical) step"""
    mpc_gpu, orc = env
    x0, goal, obst = random_batch(64, 3, seed=5)
    (g, Xg, Ug, o), = run_pair(mpc_gpu, orc, 20, 3, 2.0, x0, goal, obst, qp_iter_max=3)
    assert (g["status"] == 2).all() and (o["status"] == 2).all() and (g["iters"] == 3).all()
    assert np.abs(Xg - o["X"]).max() <= TOL_X


def test_switches(env):
    """acados-semantics switches are honoured identically on both sides"""
    mpc_gpu, orc = env
    x0, goal, obst = random_batch(32, 3, seed=9)
    for kw in (dict(cost_scale_dt=0), dict(slack_scale_dt=0), dict(lm_scaled=0), dict(bx_terminal=1), dict(soft_h=0),
               dict(bug_compat_predict=0)):
        (g, Xg, Ug, o), = run_pair(mpc_gpu, orc, 20, 3, 2.0, x0, goal, obst, **kw)
        n = judge_against_oracle(orc, o["cfg"], x0, o["P"], goal, o["start"][0], o["start"][1], g, Xg, Ug, o)
        assert n["converged"] >= 16, (kw, n)


def test_lanes_per_instance_packing(env):
    """2 or 4 instances sharing one wavefront (G = 32 / 16 lanes each) give bitwise the results of one instance per wavefront,
    also when the batch does not fill the last wavefront and when instances of one wavefront stop at different iterations"""
    mpc_gpu, orc = env
    from mpc_gpu import _lib
    for N, no, B in ((20, 3, 203), (10, 5, 77)):
        x0, goal, obst = random_batch(B, no, seed=31 + N)
        res = {}
        for lanes in (0, 64, 32) + ((16,) if N + 2 <= 16 else ()):
            with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
                _lib.check(_lib.lib().mpc_set_matrix_cores(s._h, 0))        # vector-ALU factorisation on every path: bitwise comparable
                s.set_lanes_per_stage(1)
                _lib.check(_lib.lib().mpc_set_lanes_per_instance(s._h, lanes))
                got = s.lanes_per_instance(B)
                s.reset_guess(x0); g = s.solve(x0, obst, goal); X, U = s.get_traj(B)
                res[lanes] = (got, g, X, U)
        assert res[0][0] == (32 if N == 20 else 16) and res[64][0] == 64
        assert len(set(res[0][1]["iters"].tolist())) > 3          # mixed iteration counts inside wavefronts
        for lanes in res:
            _, g, X, U = res[lanes]
            assert np.array_equal(X, res[64][2]) and np.array_equal(U, res[64][3])
            assert np.array_equal(g["iters"], res[64][1]["iters"]) and np.array_equal(g["status"], res[64][1]["status"])
            assert np.array_equal(g["cost"], res[64][1]["cost"])
    with mpc_gpu.BatchedMpc(20, 3, 2.0, max_batch=4) as s:
        assert _lib.lib().mpc_set_lanes_per_instance(s._h, 16) == _lib.MPC_ERR_ARG     # N + 1 = 21 does not fit 16 lanes


def test_matrix_core_factorisation_matches_vector_path(env):
    """v_mfma_f64_16x16x4 Riccati factorisation (one instance per wavefront) vs the vector-ALU systolic factorisation: same
    iteration counts and statuses, iterates equal to rounding; and both within tolerance of the oracle"""
    mpc_gpu, orc = env
    from mpc_gpu import _lib
    for N, no, B in ((20, 3, 300), (50, 10, 40), (10, 5, 64)):
        x0, goal, obst = random_batch(B, no, seed=51 + N)
        out = {}
        for mf in (1, 0):
            with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
                _lib.check(_lib.lib().mpc_set_matrix_cores(s._h, mf))
                s.set_row_parallel(False)      # the comparison is against the one-lane systolic sweep
                _lib.check(_lib.lib().mpc_set_lanes_per_instance(s._h, 64))
                s.reset_guess(x0); g = s.solve(x0, obst, goal); X, U = s.get_traj(B)
                s.shift(B); g2 = s.solve(x0, obst, goal); X2, U2 = s.get_traj(B)       # second step: d0 != 0, defects != 0
                out[mf] = (g, X, U, g2, X2, U2)
        ok = (out[1][0]["status"] == 0) & (out[0][0]["status"] == 0) & (out[1][3]["status"] == 0) & (out[0][3]["status"] == 0)
        assert ok.mean() > 0.9
        assert (np.abs(out[1][0]["iters"].astype(int) - out[0][0]["iters"]) <= 2).all()
        # the matrix-core path keeps the cost-to-go in a full (not triangular) tile and symmetrises every 4th stage: it is
        # ~100x less accurate than the vector paths on ill-conditioned stages (one reason it is not the default)
        d1 = np.abs(out[1][1] - out[0][1]).reshape(B, -1).max(1)[ok]; d2 = np.abs(out[1][4] - out[0][4]).reshape(B, -1).max(1)[ok]
        assert np.median(d1) < 1e-9 and np.median(d2) < 1e-9 and d1.max() < 1e-3 and d2.max() < 1e-3
        # ... and BOTH paths against the oracle on the first solve, instance by instance (the worst instance included: helpers.judge_against_oracle
        # judges an ill-conditioned QP by the QP itself)
        cfg = orc.config(N, no, 0.1 * N)
        P = oracle_P(orc, cfg, obst); X0, U0 = oracle_guess(orc, cfg, x0)
        o = orc.rti_solve_batch(cfg, x0, P, goal, X0, U0)
        for mf in (1, 0):
            n = judge_against_oracle(orc, cfg, x0, P, goal, X0, U0, out[mf][0], out[mf][1], out[mf][2], o, tol_x=1e-6 if mf == 0 else 5e-6)
            assert n["status_borderline"] == 0 and n["converged"] >= 0.9 * B, (mf, n)


@pytest.mark.gpu
@pytest.mark.parametrize("N,no,B,lanes", [(20, 3, 300, 0), (20, 3, 65, 64), (10, 5, 130, 0), (50, 10, 40, 0), (5, 3, 77, 0)])
def test_row_parallel_factorisation_matches_systolic_and_oracle(env, N, no, B, lanes):
    """row-parallel (v_fmac_f64_dpp row_newbcast) Riccati factorisation vs the one-lane systolic sweep and vs the oracle:
    same statuses and iteration counts, iterates equal to rounding; every lanes-per-instance packing"""
    mpc_gpu, orc = env
    x0, goal, obst = random_batch(B, no, seed=71 + N)
    out = {}
    for rp in (1, 0):
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
            s.set_row_parallel(rp)
            if lanes:
                s.set_lanes_per_instance(lanes)
            s.reset_guess(x0); g = s.solve(x0, obst, goal); X, U = s.get_traj(B)
            s.shift(B); g2 = s.solve(x0, obst, goal); X2, U2 = s.get_traj(B)       # second step: d0 != 0, defects != 0
            out[rp] = (g, X, U, g2, X2, U2)
    for a, b in ((0, 1), (3, 4)):
        assert (out[1][a]["status"] == out[0][a]["status"]).mean() > 0.98
        ok = (out[1][a]["status"] == 0) & (out[0][a]["status"] == 0)
        assert ok.mean() > 0.9
        assert (out[1][a]["iters"][ok] == out[0][a]["iters"][ok]).mean() > 0.95
        d = np.abs(out[1][b] - out[0][b]).reshape(B, -1).max(1)[ok]
        # both paths are equally close to the oracle (debug_rowpar.py); ill-conditioned long horizons move by ~1e-5 under rounding
        # (the second solves start from iterates that already differ by that much; one sensitive instance is allowed at N = 50)
        assert np.median(d) < 1e-10 and np.quantile(d, 0.95) < (1e-7 if N <= 20 else 2e-6)
        assert d.max() < 1e-6 if N <= 20 else (np.sort(d)[-2] < 1e-5 and d.max() < 1e-5)
    # against the oracle on the first solve
    cfg = orc.config(N, no, 0.1 * N)
    P = oracle_P(orc, cfg, obst); Xg, Ug = oracle_guess(orc, cfg, x0)
    o = orc.rti_solve_batch(cfg, x0, P, goal, Xg, Ug)
    n = judge_against_oracle(orc, cfg, x0, P, goal, Xg, Ug, out[1][0], out[1][1], out[1][2], o)
    assert n["converged"] >= 0.9 * B and n["status_borderline"] == 0, n


@pytest.mark.gpu
@pytest.mark.parametrize("N", [2, 3, 4, 14, 15, 30, 31, 62])
def test_horizon_extremes_and_packing_boundaries(env, N):
    """shortest / longest horizons and the horizons either side of a lanes-per-instance switch (N + 2 <= G), default
    (row-parallel) sweeps: the operand prefetch runs past both ends of the per-instance LDS region at small N, a tail
    wavefront has surplus instance slots, and the batch is not a multiple of the instances per wavefront"""
    mpc_gpu, orc = env
    no, B = 3, 37
    x0, goal, obst = random_batch(B, no, seed=300 + N)
    cfg = orc.config(N, no, 0.1 * N)
    P = oracle_P(orc, cfg, obst); Xg, Ug = oracle_guess(orc, cfg, x0)
    o = orc.rti_solve_batch(cfg, x0, P, goal, Xg, Ug)
    with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
        if s.lanes_per_stage(B) == 1:
            assert s.lanes_per_instance(B) == (16 if N + 2 <= 16 else 32 if N + 2 <= 32 else 64)
        else:
            assert s.lanes_per_stage(B) == (3 if N <= 20 else 2) and N <= 31 and s.lanes_per_instance(B) == 64
        s.reset_guess(x0); g = s.solve(x0, obst, goal); X, U = s.get_traj(B)
        s.shift(B); g2 = s.solve(x0, obst, goal)
    n = judge_against_oracle(orc, cfg, x0, P, goal, Xg, Ug, g, X, U, o)
    assert n["converged"] >= 0.9 * B and n["status_borderline"] == 0, n
    tol = 1e-6 if N <= 31 else 5e-5
    # second step from the shifted iterate, against the oracle fed with the GPU's own iterate
    Xs = np.stack([orc.shift(cfg, X[b], U[b])[0] for b in range(B)]); Us = np.stack([orc.shift(cfg, X[b], U[b])[1] for b in range(B)])
    o2 = orc.rti_solve_batch(cfg, x0, P, goal, Xs, Us)
    ok2 = (o2["status"] == 0) & (g2["status"] == 0)
    assert (g2["status"] == o2["status"]).mean() > 0.97 and ok2.mean() > 0.85
    assert np.abs(g2["u0"] - o2["u0"])[ok2].max() < tol * 8


@pytest.mark.gpu
@pytest.mark.parametrize("N", [62, 49])
def test_longest_horizon_with_ten_obstacles_takes_the_compact_stage_blocks(env, N):
    """N = 62 with 10 obstacles: the dense stage blocks plus the look-ahead staging would need ~66 KB of dynamic LDS per workgroup; the
    dispatcher takes the compact blocks instead (41 KB, look-ahead staged inside them, rti_solve_kernel<10, 64, 3>).  The result must match
    the oracle.  N = 49: an odd horizon on the compact blocks (the single-stage tail of the two-stages-per-pass asm sweep)"""
    mpc_gpu, orc = env
    no, B = 10, 6
    x0, goal, obst = random_batch(B, no, seed=977)
    cfg = orc.config(N, no, 0.1 * N)
    P = oracle_P(orc, cfg, obst); Xg, Ug = oracle_guess(orc, cfg, x0)
    o = orc.rti_solve_batch(cfg, x0, P, goal, Xg, Ug)
    with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
        assert s.kernel_name(B) == "rti_solve_kernel<10, 64, 3, false>"
        s.reset_guess(x0); g = s.solve(x0, obst, goal); X, U = s.get_traj(B)     # obstacle states in: look-ahead staged in LDS
    n = judge_against_oracle(orc, cfg, x0, P, goal, Xg, Ug, g, X, U, o)
    assert n["status_borderline"] == 0 and n["converged"] >= 1, n           # (an instance that does not converge from the cold start at this horizon does not on either side)


@pytest.mark.gpu
@pytest.mark.parametrize("N,no,B", [(20, 3, 300), (20, 5, 100), (20, 10, 60), (10, 5, 130), (31, 3, 50), (25, 10, 40), (2, 3, 33)])
def test_stage_split_matches_one_lane_per_stage_and_oracle(env, N, no, B):
    """rows of a stage dealt out to 3 / 2 lanes (rti_split_kernel) vs one lane per stage (rti_solve_kernel) vs the oracle:
    same statuses and iteration counts, iterates equal to rounding; explicit P and on-device look-ahead; two control steps"""
    mpc_gpu, orc = env
    x0, goal, obst = random_batch(B, no, seed=500 + N + no)
    cfg = orc.config(N, no, 0.1 * N)
    P = oracle_P(orc, cfg, obst); Xg, Ug = oracle_guess(orc, cfg, x0)
    o = orc.rti_solve_batch(cfg, x0, P, goal, Xg, Ug)
    out = {}
    for lps in (1, 2, 3) if N <= 20 else (1, 2):
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
            s.set_lanes_per_stage(lps)
            assert s.lanes_per_stage(B) == lps
            s.reset_guess(x0); g = s.solve(x0, obst, goal); X, U = s.get_traj(B)          # look-ahead in the kernel
            s.shift(B); g2 = s.solve(x0, P, goal); X2, U2 = s.get_traj(B)                # explicit P, warm start with defects
            out[lps] = (g, X, U, g2, X2, U2)
    ok = o["status"] == 0
    assert ok.mean() > 0.85
    for lps in out:
        g, X, U, g2, X2, U2 = out[lps]
        nj = judge_against_oracle(orc, cfg, x0, P, goal, Xg, Ug, g, X, U, o)
        assert nj["status_borderline"] == 0, nj
        tol = 1e-6 if N <= 20 else 5e-5           # longer horizons: an ill-conditioned instance or two sit at 1e-6 on either mapping
        d = np.abs(X - o["X"]).reshape(B, -1).max(1)[ok]; dU = np.abs(U - o["U"]).reshape(B, -1).max(1)[ok]
        # an ill-conditioned QP per batch (10 obstacles, long horizons) may sit at the float64 floor of the interior point on EVERY mapping (DESIGN.md
        # section 2): judge_against_oracle above has adjudicated it against the exact solution of the QP and bounded how many there are
        assert d.max() < 1e-5 and np.quantile(d, 0.9) < 1e-8
        for b in np.nonzero(ok)[0][(d > tol) | (dU > 8 * tol)]:
            assert no == 10 or N > 20, (b, d.max())
        rel = np.abs(g["cost"] - o["cost"])[ok] / np.maximum(1.0, np.abs(o["cost"][ok]))
        assert np.sort(rel)[-3 if no == 10 else -1] < (1e-8 if N <= 20 else 1e-6)
        if lps > 1:
            ref = out[1]
            both = ok & (g2["status"] == 0) & (ref[3]["status"] == 0)
            assert (g["status"] == ref[0]["status"]).all() and (g2["status"] == ref[3]["status"]).mean() > 0.97
            d1 = np.abs(X - ref[1]).reshape(B, -1).max(1)[ok]; d2 = np.abs(X2 - ref[4]).reshape(B, -1).max(1)[both]
            assert np.median(d1) < 1e-11 and np.sort(d1)[-3] < 10 * tol and np.median(d2) < 1e-10 and np.quantile(d2, 0.95) < 1e-6


@pytest.mark.gpu
def test_automatic_lane_mapping(env):
    """the dispatcher's choices (mpc_api.hip::pick_split / pick_waves, measured crossovers): the stage-split mapping wherever the horizon
    fits it, two wavefronts per SIMD for deep 3-obstacle batches, four instances per wavefront for short horizons at large batches"""
    mpc_gpu, orc = env
    keep = (mpc_gpu.BatchedMpc.default_lanes_per_stage, mpc_gpu.BatchedMpc.default_waves_per_simd)
    mpc_gpu.BatchedMpc.default_lanes_per_stage = mpc_gpu.BatchedMpc.default_waves_per_simd = 0
    try:
        from mpc_gpu import _lib
        with mpc_gpu.BatchedMpc(20, 3, 2.0, max_batch=70000) as s:
            assert s.lanes_per_stage(1) == 3 and s.lanes_per_stage(1024) == 3 and s.lanes_per_instance(1024) == 64
            assert s.lanes_per_stage(8192) == 3 and s.waves_per_simd(4096) == 1 and s.waves_per_simd(4097) == 2 and s.waves_per_simd(8192) == 2
            assert s.lanes_per_stage(8193) == 1 and s.lanes_per_instance(8193) == 21 and s.lanes_per_instance(65536) == 21     # three per wavefront
            assert s.kernel_name(65536) == "rti_solve_kernel<3, 21, 3, false>" and s.kernel_name(1024) == "rti_split_kernel<3, 3, false, false, false>"
            s.set_waves_per_simd(1)
            assert s.waves_per_simd(8192) == 1
            assert _lib.lib().mpc_set_waves_per_simd(s._h, 3) == _lib.MPC_ERR_ARG
            s.set_lanes_per_instance(64)
            assert s.lanes_per_stage(8) == 1 and s.waves_per_simd(65536) == 1 and s.lanes_per_instance(65536) == 64
        with mpc_gpu.BatchedMpc(20, 5, 2.0, max_batch=70000) as s:
            assert s.lanes_per_stage(7168) == 3 and s.waves_per_simd(7168) == 1 and s.lanes_per_stage(7169) == 1 and s.lanes_per_instance(7169) == 21
        with mpc_gpu.BatchedMpc(20, 10, 2.0, max_batch=70000) as s:
            assert s.lanes_per_stage(65536) == 3 and s.waves_per_simd(65536) == 1           # 10 obstacles: never the 256-register build
        with mpc_gpu.BatchedMpc(10, 3, 1.0, max_batch=70000) as s:
            assert s.lanes_per_stage(12288) == 3 and s.lanes_per_stage(12289) == 1 and s.lanes_per_instance(65536) == 16
        with mpc_gpu.BatchedMpc(50, 10, 5.0, max_batch=8) as s:
            assert s.kernel_name(8) == "rti_solve_kernel<10, 64, 3, false>"                        # long horizon: compact LDS stage blocks
        with mpc_gpu.BatchedMpc(31, 3, 3.1, max_batch=8) as s:
            assert s.lanes_per_stage(8) == 2
        with mpc_gpu.BatchedMpc(32, 3, 3.2, max_batch=8) as s:
            assert s.lanes_per_stage(8) == 1
            assert _lib.lib().mpc_set_lanes_per_stage(s._h, 2) == _lib.MPC_ERR_ARG
    finally:
        mpc_gpu.BatchedMpc.default_lanes_per_stage, mpc_gpu.BatchedMpc.default_waves_per_simd = keep


@pytest.mark.gpu
@pytest.mark.parametrize("N,no,B", [(20, 3, 700), (20, 5, 300), (31, 3, 200), (10, 10, 150), (2, 3, 33)])
def test_two_wavefronts_per_simd_variant_is_bitwise_the_one_wavefront_variant(env, N, no, B):
    """rti_split_kernel<.., W2 = true> (256 registers, compact LDS blocks, results overlaying the operand blocks, look-ahead staged in the
    operand region) computes exactly what the 512-register / dense-LDS variant computes: same instructions on the same operands in the
    same order, so every output agrees bit for bit -- over three closed-loop steps, explicit P and on-device look-ahead"""
    mpc_gpu, orc = env
    x0, goal, obst = random_batch(B, no, seed=900 + N + no)
    cfg = orc.config(N, no, 0.1 * N)
    P = oracle_P(orc, cfg, obst)
    res = {}
    for w in (1, 2):
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
            s.set_lanes_per_stage(3 if N <= 20 else 2); s.set_waves_per_simd(w)
            assert s.waves_per_simd(B) == w
            s.reset_guess(x0); outs = []
            for k in range(3):
                g = s.solve(x0, obst if k != 1 else P, goal); X, U = s.get_traj(B); s.shift(B)
                outs.append((g, X, U))
            res[w] = outs
    for (ga, Xa, Ua), (gb, Xb, Ub) in zip(res[1], res[2]):
        assert np.array_equal(Xa, Xb) and np.array_equal(Ua, Ub)
        for key in ("u0", "cost", "status", "iters"):
            assert np.array_equal(ga[key], gb[key]), key


@pytest.mark.gpu
@pytest.mark.parametrize("N,no,B", [(20, 3, 500), (20, 5, 100), (10, 3, 200), (17, 10, 40), (9, 10, 30), (8, 10, 30), (2, 3, 31)])
def test_three_instances_per_wavefront(env, N, no, B):
    """G = 21 (lanes [0,21), [21,42), [42,63) of a wavefront hold three instances; compact LDS stage blocks; the sweeps of the three
    instances run in DPP rows 0..2) against two / four instances per wavefront and against the oracle: statuses equal, iteration counts
    equal but for a rounding case, iterates equal to rounding (the wavefront reductions add in a different order); batch not a multiple of 3;
    look-ahead in the kernel and explicit P; three closed-loop steps"""
    mpc_gpu, orc = env
    x0, goal, obst = random_batch(B, no, seed=1300 + N + no)
    cfg = orc.config(N, no, 0.1 * N)
    P = oracle_P(orc, cfg, obst); Xg, Ug = oracle_guess(orc, cfg, x0)
    res = {}
    for G in (21, 32 if N + 2 > 16 else 16):
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
            s.set_lanes_per_stage(1); s.set_lanes_per_instance(G)
            assert s.lanes_per_instance(B) == G
            s.reset_guess(x0); outs = []
            for k in range(3):
                g = s.solve(x0, obst if k != 1 else P, goal); X, U = s.get_traj(B); s.shift(B)
                outs.append((g, X, U))
            res[G] = outs
    other = [k for k in res if k != 21][0]
    for (ga, Xa, Ua), (gb, Xb, Ub) in zip(res[21], res[other]):
        assert (ga["status"] == gb["status"]).mean() >= 0.99
        ok = (ga["status"] == 0) & (gb["status"] == 0)
        assert (ga["iters"][ok] == gb["iters"][ok]).mean() >= 0.95
        d = np.abs(Xa - Xb).reshape(B, -1).max(1)[ok]
        assert np.median(d) < 1e-10 and np.quantile(d, 0.95) < 1e-6 and d.max() < 1e-5
    o = orc.rti_solve_batch(cfg, x0, P, goal, Xg, Ug)
    g, X, U = res[21][0]
    assert (g["status"] == o["status"]).all()
    ok = o["status"] == 0
    d = np.abs(X - o["X"]).reshape(B, -1).max(1)[ok]
    assert np.quantile(d, 0.9) < 1e-8 and d.max() < 1e-5
    adjudicate_batch(orc, cfg, x0, P, goal, Xg, Ug, X, U, o, np.nonzero(ok)[0][d > 1e-6], what="three instances per wavefront")       # against the exact QP solution, as everywhere
    with mpc_gpu.BatchedMpc(21, 3, 2.1, max_batch=4) as s:
        from mpc_gpu import _lib
        assert _lib.lib().mpc_set_lanes_per_instance(s._h, 21) == _lib.MPC_ERR_ARG      # 22 stages do not fit 21 lanes


@pytest.mark.gpu
@pytest.mark.parametrize("B", [65536, 32768])
def test_full_size_batches_c3_and_c4_share(env, B):
    """BASELINE configs[2] (65536 on one GPU) and one GPU's share of configs[3] (32768 of 262144) at FULL size, whatever kernel the
    dispatcher gives them: two closed-loop steps; instances are independent (a permuted batch gives the permuted result -- bit for bit, or
    to rounding where three instances share a wavefront); 256 instances sampled from the whole range agree with the oracle"""
    mpc_gpu, orc = env
    N, no = 20, 3
    x0, goal, obst = random_batch(B, no, seed=1234)
    perm = np.random.default_rng(1).permutation(B)
    with mpc_gpu.BatchedMpc(N, no, 2.0, max_batch=B) as s:
        kernel = s.kernel_name(B, lookahead=False)
        s.reset_guess(x0); g1 = s.solve(x0, obst, goal); s.shift(B); Xs, Us = s.get_traj(B); h1 = s.solve(x0, obst, goal); X1, U1 = s.get_traj(B)
        s.reset_guess(x0[perm]); s.solve(x0[perm], obst[perm], goal[perm]); s.shift(B); h2 = s.solve(x0[perm], obst[perm], goal[perm]); X2, U2 = s.get_traj(B)
    same = h1["status"][perm] == h2["status"]
    if "21" in kernel.split(",")[1]:
        assert same.mean() > 0.999
        d = np.abs(X1[perm] - X2).reshape(B, -1).max(1)[same & (h2["status"] == 0)]
        assert np.median(d) < 1e-11 and np.quantile(d, 0.999) < 1e-6      # (round 5: 1e-5; the stated tolerance since the third polish indicator)
    else:
        assert same.all() and np.array_equal(X1[perm], X2) and np.array_equal(U1[perm], U2)
    assert (g1["status"] != 4).mean() > 0.99 and 5.0 < g1["iters"].mean() < 12.0
    idx = np.linspace(0, B - 1, 256).astype(int)
    cfg = orc.config(N, no, 2.0)
    P = oracle_P(orc, cfg, obst[idx])
    o = orc.rti_solve_batch(cfg, x0[idx], P, goal[idx], Xs[idx], Us[idx])          # second step, from the GPU's own shifted iterate
    assert (o["status"] == h1["status"][idx]).all()
    ok = o["status"] == 0
    d = np.abs(o["X"] - X1[idx]).reshape(256, -1).max(1)[ok]
    assert np.quantile(d, 0.98) < 1e-8 and d.max() < 1e-5
    adjudicate_batch(orc, cfg, x0[idx], P, goal[idx], Xs[idx], Us[idx], X1[idx], U1[idx], o, np.nonzero(ok)[0][d > TOL_X], what="permuted batch")


@pytest.mark.gpu
@pytest.mark.parametrize("G", [32, 21, 16, 0])
def test_instance_scheduling(env, G):
    """instances are dealt to wavefronts in the order of their previous iteration counts (schedule_kernel: stable counting sort on the device):
    the order is a permutation, sorted by the last counts (descending), and the results are those of the natural order -- bit for bit with one,
    two or four instances per wavefront, to rounding with three -- over four closed-loop steps; a batch of at most one wavefront per SIMD and a
    changed batch size fall back to the natural order"""
    mpc_gpu, orc = env
    N, no, B = (20, 3, 5003) if G != 16 else (10, 3, 6001)
    x0, goal, obst = random_batch(B, no, seed=2024 + G)
    res = {}
    for on in (1, 0):
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
            if G:
                s.set_lanes_per_stage(1); s.set_lanes_per_instance(G)
            else:
                s.set_lanes_per_stage(3)                                # G = 0: the stage-split mapping (one instance per wavefront: longest first)
            s.set_instance_scheduling(bool(on))
            assert s.instance_order(B) is None
            s.reset_guess(x0); outs = []
            xk = x0.copy()
            for k in range(4):
                g = s.solve(xk, obst, goal); X, U = s.get_traj(B)
                order = s.instance_order(B)
                if on:
                    assert order is not None and np.array_equal(np.sort(order), np.arange(B))          # a permutation
                    assert (np.diff(np.minimum(g["iters"][order], 31)) <= 0).all()                        # by the last counts, descending
                    assert len(set(g["iters"].tolist())) > 4
                else:
                    assert order is None
                outs.append((g, X, U))
                xk = s.plant_step(xk, g["u0"]); s.shift(B)
            if on:
                assert s.instance_order(B - 1) is None                  # another batch size: no order yet
                g = s.solve(xk[:900], obst[:900], goal[:900])            # at most one wavefront per SIMD: never scheduled
                assert s.instance_order(900) is None
            res[on] = outs
    for (ga, Xa, Ua), (gb, Xb, Ub) in zip(res[1], res[0]):
        if G != 21:
            assert np.array_equal(Xa, Xb) and np.array_equal(Ua, Ub) and np.array_equal(ga["iters"], gb["iters"]) and np.array_equal(ga["cost"], gb["cost"])
        else:
            same = ga["status"] == gb["status"]
            assert same.mean() > 0.999
            d = np.abs(Xa - Xb).reshape(B, -1).max(1)[same & (gb["status"] == 0)]
            assert np.median(d) < 1e-11 and np.quantile(d, 0.999) < 1e-6      # (round 5: 1e-5; the stated tolerance since the third polish indicator)


@pytest.mark.gpu
@pytest.mark.parametrize("N,no,B", [(20, 3, 200), (20, 5, 90), (10, 3, 64), (30, 3, 40), (20, 10, 48), (4, 3, 33)])
def test_block_riccati_matches_the_oracle_and_the_stagewise_recursion(built, N, no, B):
    """mpc_set_block_riccati(1): Riccati factorisation and vector recursions over PAIRS of stages (x_2m+1 eliminated, 4 x 4 input block, Blk2Lds /
    rowpar_factor2) in the stage-split kernel -- opt-in, because it measured slower.  Two closed-loop steps (cold start; warm start with
    defects and an explicit P) judged against the oracle instance by instance, and against the one-stage-per-step recursion: same statuses, iteration counts
    within the end-game, iterates equal to rounding.  Odd horizons and partial obstacle counts fall back to the stagewise kernel."""
    import mpc_gpu
    from oracle import oracle as orc
    x0, goal, obst = random_batch(B, no, seed=900 + N + no)
    cfg = orc.config(N, no, 0.1 * N)
    P = oracle_P(orc, cfg, obst); X0, U0 = oracle_guess(orc, cfg, x0)
    out = {}
    for on in (1, 0):
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
            s.set_lanes_per_stage(3 if N <= 20 else 2); s.set_block_riccati(bool(on))
            assert s.kernel_name(B).endswith("true>" if on else "false>"), s.kernel_name(B)
            s.reset_guess(x0); g = s.solve(x0, obst, goal); X, U = s.get_traj(B)
            s.shift(B); Xs, Us = s.get_traj(B); g2 = s.solve(x0, P, goal); X2, U2 = s.get_traj(B)
            out[on] = (g, X, U, g2, X2, U2, Xs, Us)
    g, X, U, g2, X2, U2, Xs, Us = out[1]
    o = orc.rti_solve_batch(cfg, x0, P, goal, X0, U0)
    n = judge_against_oracle(orc, cfg, x0, P, goal, X0, U0, g, X, U, o)
    assert n["converged"] >= 0.85 * B and n["status_borderline"] == 0, n
    o2 = orc.rti_solve_batch(cfg, x0, P, goal, Xs, Us)
    n2 = judge_against_oracle(orc, cfg, x0, P, goal, Xs, Us, g2, X2, U2, o2)
    assert n2["converged"] >= 0.8 * B, n2
    ref = out[0]
    assert np.array_equal(g["status"], ref[0]["status"]) and (np.abs(g["iters"].astype(int) - ref[0]["iters"]) <= 2).all()
    ok = g["status"] == 0
    d = np.abs(X - ref[1]).reshape(B, -1).max(1)[ok]
    assert np.median(d) < 1e-10 and d.max() < (1e-6 if no < 10 else 1e-5)
    with mpc_gpu.BatchedMpc(N + 1, no, 0.1 * (N + 1), max_batch=4) as s:      # odd horizon: no pairs
        s.set_lanes_per_stage(3 if N + 1 <= 20 else 2); s.set_block_riccati(True)
        assert s.kernel_name(4).endswith("false>")


@pytest.mark.gpu
def test_c5_kernel_at_a_scheduled_batch(built):
    """BASELINE configs[4]'s own instantiation, rti_solve_kernel<10, 64, 3, false>, at a batch the instance scheduling reorders (4096 = the per-GPU share of
    C5 on 8 GPUs; N = 50, 10 obstacles, fused closed-loop steps): after three control steps (the third runs in the order built from the second's iteration
    counts) 128 instances spread over the batch are judged against the oracle fed with the GPU's own iterate, and the batch solved again in a random
    permutation gives every instance the same result (one instance per wavefront: bit for bit)."""
    import torch
    import mpc_gpu
    from oracle import oracle as orc
    N, no, B = 50, 10, 4096
    x0, goal, obst = random_batch(B, no, seed=5050)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cfg = orc.config(N, no, 5.0)
    res = []
    perm = np.random.default_rng(1).permutation(B)
    for order in (np.arange(B), perm):
        with mpc_gpu.BatchedMpc(N, no, 5.0, max_batch=B) as s, torch.cuda.stream(torch.cuda.Stream(device=dev)):
            assert s.kernel_name(B) == "rti_solve_kernel<10, 64, 3, false>"
            st = torch.cuda.current_stream().cuda_stream
            dx, dg, do = t(x0[order]), t(goal[order]), t(obst[order])
            X = torch.zeros(B, N + 1, 5, dtype=torch.float64, device=dev); U = torch.zeros(B, N, 2, dtype=torch.float64, device=dev)
            u0 = torch.zeros(B, 2, dtype=torch.float64, device=dev); cost = torch.zeros(B, dtype=torch.float64, device=dev)
            status = torch.zeros(B, dtype=torch.int32, device=dev); iters = torch.zeros(B, dtype=torch.int32, device=dev)
            s.reset_guess_dev(B, dx, X, U, stream=st)
            for k in range(2):
                s.closed_loop_step_dev(B, dx, do, dg, X, U, u0, cost, status, iters, None, stream=st)
            torch.cuda.synchronize()
            assert s.instance_order(B) is not None
            before = (dx.cpu().numpy(), do.cpu().numpy(), X.cpu().numpy(), U.cpu().numpy())
            s.closed_loop_step_dev(B, dx, do, dg, X, U, u0, cost, status, iters, None, flags=0, stream=st)      # the solve alone: X, U = the new iterate, unshifted
            torch.cuda.synchronize()
            res.append((before, X.cpu().numpy(), U.cpu().numpy(), u0.cpu().numpy(), cost.cpu().numpy(), status.cpu().numpy(), iters.cpu().numpy()))
    (xb, ob, Xb, Ub), Xn, Un, u0n, cn, sn, itn = res[0]
    pick = np.arange(0, B, B // 128)
    P = np.stack([orc.predict_params(cfg, ob[b]) for b in pick])
    o = orc.rti_solve_batch(cfg, xb[pick], P, goal[pick], Xb[pick], Ub[pick])
    g = dict(status=sn[pick], iters=itn[pick], cost=cn[pick], u0=u0n[pick])
    n = judge_against_oracle(orc, cfg, xb[pick], P, goal[pick], Xb[pick], Ub[pick], g, Xn[pick], Un[pick], o)
    assert n["converged"] >= 100 and n["status_borderline"] <= 1, n
    _, Xp, Up, u0p, cp, sp, itp = res[1]
    assert np.array_equal(Xp, Xn[perm]) and np.array_equal(Up, Un[perm]) and np.array_equal(sp, sn[perm]) and np.array_equal(itp, itn[perm]) and np.array_equal(cp, cn[perm])


@pytest.mark.gpu
def test_instance_scheduling_across_streams(built):
    """The instance order belongs to the handle, the launches to the caller's stream: consecutive *_dev solves of one batch size on two DIFFERENT
    streams (no host synchronisation between them) must see a complete order -- the second launch waits for the event recorded behind the
    first one's sort.  Results equal those of the same sequence on one stream, bit for bit (two instances per wavefront)."""
    import torch
    import mpc_gpu
    N, no, B = 20, 3, 6000
    x0, goal, obst = random_batch(B, no, seed=99)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    res = []
    for two_streams in (False, True):
        sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        with mpc_gpu.BatchedMpc(N, no, 2.0, max_batch=B) as s:
            s.set_lanes_per_stage(1); s.set_lanes_per_instance(32)
            x0d, gd, od = t(x0), t(goal), t(obst)
            P = torch.zeros(B, N + 1, no, 2, dtype=torch.float64, device=dev)
            X = torch.zeros(B, N + 1, 5, dtype=torch.float64, device=dev); U = torch.zeros(B, N, 2, dtype=torch.float64, device=dev)
            it = torch.zeros(B, dtype=torch.int32, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            s.predict_dev(B, od, P, stream=sa.cuda_stream); s.reset_guess_dev(B, x0d, X, U, stream=sa.cuda_stream)
            for k in range(6):
                cur = (sa, sb)[k % 2] if two_streams else sa
                if two_streams and k:
                    cur.wait_stream((sa, sb)[(k - 1) % 2])      # the DATA dependency (X, U) is the caller's to order; the handle's own order array is not
                s.solve_dev(B, x0d, P, gd, X, U, None, None, st, it, stream=cur.cuda_stream)
            torch.cuda.synchronize()
            order = s.instance_order(B)
            assert order is not None and np.array_equal(np.sort(order), np.arange(B))
            res.append((X.cpu().numpy(), U.cpu().numpy(), it.cpu().numpy(), st.cpu().numpy(), order))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [0, 21, 32, 64])
def test_non_finite_inputs_are_contained(built, lanes):
    """NaN / Inf in one instance's x0, goal or obstacle state (a diverged plant, a bad sensor frame): that instance fails at once (status 4,
    no iterations, iterate untouched -- fmax() in the residual norms would otherwise swallow the NaN and report convergence); every OTHER instance of the
    batch -- including those sharing its wavefront (two or three instances per wavefront) -- is bit for bit what a clean batch gives."""
    import mpc_gpu
    mpc_gpu.BatchedMpc.default_lanes_per_stage = 0
    N, no, B = 20, 3, 90
    x0, goal, obst = random_batch(B, no, seed=4242)
    bad = {7: ("x0", np.nan), 8: ("goal", np.inf), 30: ("obst", np.nan), 31: ("x0", -np.inf), 64: ("obst", np.inf)}
    x0b, goalb, obstb = x0.copy(), goal.copy(), obst.copy()
    for b, (what, v) in bad.items():
        if what == "x0": x0b[b, 1] = v
        elif what == "goal": goalb[b, 0] = v
        else: obstb[b, 1, 3] = v          # vy: the look-ahead uses vy on both axes (defect D1), vx alone would never reach the solve
    res = []
    for xx, gg, oo in ((x0, goal, obst), (x0b, goalb, obstb)):
        with mpc_gpu.BatchedMpc(N, no, 2.0, max_batch=B) as s:
            if lanes: s.set_lanes_per_stage(1); s.set_lanes_per_instance(lanes)
            s.reset_guess(x0); Xg, Ug = s.get_traj(B)
            g = s.solve(xx, oo, gg); X, U = s.get_traj(B)
            res.append((g, X, U))
    (gc, Xc, Uc), (gb, Xb, Ub) = res
    clean = np.array([b not in bad for b in range(B)])
    assert np.array_equal(Xc[clean], Xb[clean]) and np.array_equal(Uc[clean], Ub[clean])
    assert np.array_equal(gc["status"][clean], gb["status"][clean]) and np.array_equal(gc["iters"][clean], gb["iters"][clean])
    for b in bad:
        assert gb["status"][b] == 4 and gb["iters"][b] == 0, (b, gb["status"][b], gb["iters"][b])      # as the oracle (test_oracle_math.py)
        assert np.array_equal(Xb[b], Xg[b]) and np.array_equal(Ub[b], Ug[b])                           # iterate untouched


@pytest.mark.gpu
@pytest.mark.parametrize("N,no,B", [(20, 1, 60), (20, 2, 60), (20, 4, 80), (20, 7, 40), (31, 8, 30), (10, 6, 50), (40, 4, 24), (40, 9, 12), (55, 8, 8), (62, 1, 6)])
def test_any_obstacle_count(built, N, no, B):
    """N_OBST is a free constant of the reference (world_specification.py:25; its tables use 5, BASELINE's workloads 3 and 10): any count
    1..10 runs on the instantiation with the next row capacity (3, 5 or 10), the rows of the missing obstacles switched off:
    rti_split_kernel<cap, LPS, false, true> for N <= 31, beyond that rti_solve_kernel<cap, 64, FACT, true>.  Same contract as everywhere: statuses
    equal, iterates to 1e-6 or judged by the QP; look-ahead in the kernel and explicit P; three closed-loop steps; a fused closed-loop step
    moves exactly `no` obstacles."""
    import mpc_gpu
    from oracle import oracle as orc
    mpc_gpu.BatchedMpc.default_lanes_per_stage = 0
    x0, goal, obst = random_batch(B, no, seed=77 + 13 * N + no)
    cfg = orc.config(N, no, 0.1 * N)
    P = oracle_P(orc, cfg, obst); Xo, Uo = oracle_guess(orc, cfg, x0)
    cap = 3 if no <= 3 else (5 if no <= 5 else 10)
    with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
        name = s.kernel_name(B)
        assert name.startswith(f"rti_split_kernel<{cap}," if N <= 31 else f"rti_solve_kernel<{cap}, 64,") and (name.endswith("true>") or name.endswith("true, false>")), name
        s.reset_guess(x0)
        for k in range(3):
            g = s.solve(x0, obst if k != 1 else P, goal); X, U = s.get_traj(B)
            o = orc.rti_solve_batch(cfg, x0, P, goal, Xo, Uo)
            assert (g["status"] == o["status"]).all(), (k, g["status"], o["status"])
            ok = o["status"] == 0
            assert (g["iters"][ok] == o["iters"][ok]).mean() >= 0.9
            d = np.abs(X - o["X"]).reshape(B, -1).max(1)
            adjudicate_batch(orc, cfg, x0, P, goal, Xo, Uo, X, U, o, np.nonzero(ok & (d > 1e-6))[0], what=f"{no} obstacles, step {k}")
            assert np.median(d[ok]) < 1e-9
            rel = np.abs(g["cost"][ok] - o["cost"][ok]) / np.maximum(1.0, np.abs(o["cost"][ok]))
            assert np.median(rel) < 1e-10
            # both sides continue from the oracle's iterate (single-solve parity on identical inputs)
            Xo, Uo = o["X"].copy(), o["U"].copy()
            for b in range(B):
                Xo[b], Uo[b] = orc.shift(cfg, Xo[b], Uo[b])
            s.set_warmstart(Xo, Uo)
    # fused closed-loop step: plant + obstacle motion over exactly `no` obstacles, against the oracle's pieces
    rng = np.random.default_rng(5)
    noise = rng.standard_normal((3, B, no, 2))
    r = mpc_gpu.run_episodes(x0, goal, obst, N=N, Tf=0.1 * N, max_iter=3, random_move=True, noise=noise, record=True)
    live = r["table"][:, 1] == 0                      # (an instance that reached its goal stops moving its obstacles)
    ob = obst.copy()
    for k in range(3):
        for b in range(B):
            for j in range(no):
                ob[b, j] = orc.obstacle_step(cfg, ob[b, j], 0.1, noise=noise[k, b, j])
    assert r["obst_traj"].shape == (4, B, no, 4)
    assert np.abs(r["obst_traj"][-1][live] - ob[live]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("N,no,B", [(20, 4, 3001), (40, 7, 2500), (20, 8, 2200)])
def test_instance_scheduling_with_fewer_obstacles_than_rows(built, N, no, B):
    """the kernels with a run-time obstacle count under instance scheduling (more than one wavefront per SIMD): one instance per wavefront,
    so the scheduled launches must give the natural order's results bit for bit, over three closed-loop steps, and match the oracle"""
    import mpc_gpu
    from oracle import oracle as orc
    mpc_gpu.BatchedMpc.default_lanes_per_stage = 0
    x0, goal, obst = random_batch(B, no, seed=99 + N + no)
    res = {}
    for on in (1, 0):
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
            assert "true" in s.kernel_name(B).split(",")[3]      # the masked (run-time obstacle count) instantiation
            s.set_instance_scheduling(bool(on)); s.reset_guess(x0); outs = []; xk = x0.copy()
            for k in range(3):
                g = s.solve(xk, obst, goal); X, U = s.get_traj(B)
                assert (s.instance_order(B) is not None) == bool(on)
                outs.append((g, X, U)); xk = s.plant_step(xk, g["u0"]); s.shift(B)
            res[on] = outs
    for (ga, Xa, Ua), (gb, Xb, Ub) in zip(res[1], res[0]):
        assert np.array_equal(Xa, Xb) and np.array_equal(Ua, Ub) and np.array_equal(ga["iters"], gb["iters"]) and np.array_equal(ga["status"], gb["status"])
    cfg = orc.config(N, no, 0.1 * N)
    Xg, Ug = oracle_guess(orc, cfg, x0[:200]); P = oracle_P(orc, cfg, obst[:200])
    o = orc.rti_solve_batch(cfg, x0[:200], P, goal[:200], Xg, Ug)
    g, X, U = res[1][0]
    assert (g["status"][:200] == o["status"]).all()
    ok = o["status"] == 0
    assert np.median(np.abs(X[:200] - o["X"]).reshape(200, -1).max(1)[ok]) < 1e-9


@pytest.mark.parametrize("N,no,B", [(20, 3, 8), (20, 5, 8), (50, 10, 4)])
def test_hip_path_against_the_exact_active_set_solution(env, N, no, B):
    """The HIP path against a solution that owes NO interior point anything (helpers.exact_from_active_set on the exported QP; the export is the linearisation the kernels
    are checked against to 1e-12 in test_linearize_parity): the first RTI step of B random scenarios, applied control to 1e-6, every variable to 1e-5."""
    from helpers import exact_from_active_set
    mpc_gpu, orc = env
    x0, goal, obst = random_batch(B, no, seed=900 + N + no)
    cfg = orc.config(N, no, 0.1 * N)
    P = oracle_P(orc, cfg, obst); X, U = oracle_guess(orc, cfg, x0)
    with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
        s.set_warmstart(X, U)
        g = s.solve(x0, P, goal)
        Xg, Ug = s.get_traj(B)
    verified = 0
    for b in range(B):
        if g["status"][b] != 0:
            continue
        q = orc.export_qp(cfg, x0[b], P[b], goal[b], X[b], U[b])
        dX, dU = Xg[b] - X[b], Ug[b] - U[b]
        v_gpu = np.concatenate([np.concatenate([dU[i], dX[i + 1]]) for i in range(N)])
        v_ex, lam_min, feas, _, res = exact_from_active_set(q, v_gpu)
        if lam_min < -1e-7 or feas < -1e-7 or res > 1e-9:
            continue
        verified += 1
        assert np.abs(v_gpu[:2] - v_ex[:2]).max() < 1e-6 and np.abs(v_gpu - v_ex).max() < 1e-5, (b, np.abs(v_gpu - v_ex).max())
    assert verified >= B // 2


@pytest.mark.gpu
def test_qp_fail_policy_truncate_against_the_oracle(env):
    """mpc_config.qp_fail_policy (docs/PROBLEM.md section 2): instances whose QP is infeasible -- the robot 0.1 from the state box with 6 .. 9.9 m/s towards it:
    x <= 7 at stage 1 cannot be met with |u| <= 8 -- next to feasible ones.  Policy 0: the divergence tests end them with status 4, iterate untouched.  Policy 1
    ("truncate", what acados did with a HPIPM MAX_ITER): at a cap of 10 iterations they run to the cap and the truncated step is applied, status 2, GPU and oracle
    agreeing on the iterate; at the default cap the interior point's step collapses first (status 4 on both sides, after a few iterations more than under
    policy 0) -- which is why the recorded tables cannot tell the policies apart (profiles/r04_fail_policy_replay.json).  Feasible instances do not notice."""
    mpc_gpu, orc = env
    B = 64
    x0, goal, obst = random_batch(B, 3, seed=77)
    bad = np.arange(0, B, 4); good = np.setdiff1d(np.arange(B), bad)
    x0[bad, 0] = 6.9; x0[bad, 2] = 0.0; x0[bad, 3] = np.linspace(6.0, 9.9, len(bad))
    res = {}
    for cap in (10, 50):
        for pol in (0, 1):
            cfg = orc.config(20, 3, 2.0, qp_fail_policy=pol, qp_iter_max=cap)
            P = oracle_P(orc, cfg, obst); X, U = oracle_guess(orc, cfg, x0)
            o = orc.rti_solve_batch(cfg, x0, P, goal, X, U)
            with mpc_gpu.BatchedMpc(20, 3, 2.0, max_batch=B, qp_fail_policy=pol, qp_iter_max=cap) as s:
                assert s.cfg.qp_fail_policy == pol
                s.set_warmstart(X, U); g = s.solve(x0, P, goal); Xg, Ug = s.get_traj(B)
            res[cap, pol] = (g, Xg, Ug, o)
            assert (g["status"] == o["status"]).all(), (cap, pol, g["status"], o["status"])
            # feasible instances: the same iteration count; a diverging one is recognised within two iterations of the oracle (round 6: the oracle mirrors the kernels'
            # two guards -- a NaN of the centring target, a non-finite step -- which brought the distance down from three; what remains is WHERE the overflow first
            # shows: the kernels form the affine complementarity from four running sums per lane (inf - inf one or two iterations before the oracle's sum of products)
            assert (np.abs(g["iters"].astype(int) - o["iters"]) <= 2).all() and (g["iters"][good] == o["iters"][good]).all()
            assert np.abs(Xg[good] - o["X"][good]).max() < 1e-6
            for b in bad[g["status"][bad] == 4]:
                assert np.array_equal(Xg[b], X[b]) and np.array_equal(Ug[b], U[b])          # a failed QP leaves the iterate untouched
    g0, _, _, _ = res[10, 0]; g1, X1, U1, o1 = res[10, 1]
    assert (g0["status"][bad] == 4).sum() >= len(bad) - 2              # (one of them reaches the cap of 10 before the divergence test fires)
    assert (g1["status"][bad] == 2).all() and (g1["iters"][bad] == 10).all()
    assert np.abs(X1[bad] - o1["X"][bad]).max() < 1e-6 and np.abs(X1[bad] - oracle_guess(orc, orc.config(20, 3, 2.0), x0[bad])[0]).max() > 1e-3      # the truncated step IS applied
    assert (res[50, 0][0]["status"][bad] == 4).all() and (res[50, 1][0]["status"][bad] == 4).all()
    assert (res[50, 1][0]["iters"][bad] >= res[50, 0][0]["iters"][bad]).all()      # without the divergence test the failure is noticed later (step collapse)
    for cap in (10, 50):           # feasible instances: bit for bit the same under both policies
        assert np.array_equal(res[cap, 0][1][good], res[cap, 1][1][good]) and np.array_equal(res[cap, 0][0]["status"][good], res[cap, 1][0]["status"][good])


def test_an_unsolved_end_game_is_reported_not_passed_off_as_converged(env):
    """Found by scripts/fuzz_parity.py (round 5; 1 of 2.6e6 solves): N = 47, one obstacle, a shifted warm start against an x0 that has NOT advanced.  Every
    termination test of the interior point reads zero (linear residuals, largest live complementarity product), the active set is the exact solution's, and the
    "solution" is 2e-2 from it (8e-3 under round 4's rules; stationarity residual 0.4): the end-game's Newton steps have lost their accuracy to the barrier
    weights lam / t_floor.  The polish's step estimate sees it -- after the two polish iterations it still stands at 2e-2, four orders of magnitude above what a
    healthy solve ends with -- and the solve is reported as NOT converged: status 2 (step applied, as after an iteration cap) on BOTH sides.  The instance's
    neighbours in the batch are unaffected and judged as usual."""
    mpc_gpu, orc = env
    N, no, B, seed, bad = 47, 1, 1500, 863992655, 155
    x0, goal, obst = random_batch(B, no, seed=seed)
    keep = np.arange(bad - 20, bad + 21)                       # the instance and forty neighbours
    x0, goal, obst = x0[keep], goal[keep], obst[keep]
    bad = 20
    cfg = orc.config(N, no, 0.1 * N)
    P = oracle_P(orc, cfg, obst); X, U = oracle_guess(orc, cfg, x0)
    o0 = orc.rti_solve_batch(cfg, x0, P, goal, X, U)
    Xs, Us = zip(*[orc.shift(cfg, o0["X"][b], o0["U"][b]) for b in range(len(keep))])
    Xs, Us = np.stack(Xs), np.stack(Us)
    o = orc.rti_solve_batch(cfg, x0, P, goal, Xs, Us)            # the same x0 again: the stale warm start
    with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=len(keep)) as s:
        s.set_warmstart(Xs, Us); g = s.solve(x0, P, goal); Xg, Ug = s.get_traj(len(keep))
    assert o["status"][bad] == 2 and g["status"][bad] == 2, (o["status"][bad], g["status"][bad], o["iters"][bad], g["iters"][bad])
    assert o["iters"][bad] < cfg.qp_iter_max and abs(int(g["iters"][bad]) - int(o["iters"][bad])) <= 2      # not the cap: the "unsolved" rule
    a = adjudicate(orc, cfg, x0[bad], P[bad], goal[bad], Xs[bad], Us[bad], Xg[bad], Ug[bad], o["X"][bad], o["U"][bad])
    assert a["kind"] == "exact" and a["d_oracle"] > 1e-3        # it really is far from the QP's solution: the status must not be 0
    n = judge_against_oracle(orc, cfg, x0, P, goal, Xs, Us, g, Xg, Ug, o)
    assert n["converged"] >= len(keep) - 6, n


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [0, 1])
def test_stationarity_indicator_closes_a_tail_instance_on_the_gpu(env, lanes):
    """Polish indicator (c) (round 6: the stationarity residual of the QP's Lagrangian, formed by the kernels with three levels of suffix sums over the stage lanes)
    doing its work ALONE on the GPU path: instance 715 of the parity sweep's C5 batch (N = 50, 10 obstacles, a first solve) meets every termination test 1.6e-6 from
    the exact solution of its QP with all three indicators off (the oracle: 1.7e-5 -- round 5's one GPU instance beyond 1e-6 in that configuration); with (a) and (b)
    off and (c) at its default 1e-7 the GPU lands within 2e-7 of the exact solution (measured 6.8e-8) at one or two further iterations, within one iteration of the
    oracle under the same settings (the residual sits a rounding error from its threshold on one side or the other).  Both lane mappings that can run N = 50 (the dispatcher's
    one-instance-per-wavefront kernel; `lanes` = 1 forces one lane per stage explicitly), the instance among neighbours so that the batch is not a special case."""
    mpc_gpu, orc = env
    N, no = 50, 10
    x0, goal, obst = random_batch(4000, no, seed=4242 + N + no)
    keep = np.arange(715 - 4, 715 + 5); b = 4
    x0, goal, obst = x0[keep], goal[keep], obst[keep]
    res = {}
    for name, kw in (("off", dict(polish_ratio=0.0, polish_tol=0.0, polish_res_g=0.0)), ("stationarity", dict(polish_ratio=0.0, polish_tol=0.0))):
        cfg = orc.config(N, no, 0.1 * N, **kw)
        P = oracle_P(orc, cfg, obst); X, U = oracle_guess(orc, cfg, x0)
        o = orc.rti_solve_batch(cfg, x0, P, goal, X, U)
        with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=len(keep), **kw) as s:
            if lanes:
                s.set_lanes_per_stage(1)
            assert s.cfg.polish_res_g == kw.get("polish_res_g", 1e-7)
            s.set_warmstart(X, U); g = s.solve(x0, P, goal); Xg, Ug = s.get_traj(len(keep))
        assert g["status"][b] == 0 and o["status"][b] == 0
        vex, ok, _ = exact_qp(orc.export_qp(cfg, x0[b], P[b], goal[b], X[b], U[b]), step_vector(N, X[b], U[b], Xg[b], Ug[b]))
        assert ok
        res[name] = (float(np.abs(step_vector(N, X[b], U[b], Xg[b], Ug[b]) - vex).max()), int(g["iters"][b]), int(o["iters"][b]))
    assert res["off"][0] > 1e-6 and res["stationarity"][0] < 2e-7, res
    assert 1 <= res["stationarity"][1] - res["off"][1] <= 2 and abs(res["stationarity"][1] - res["stationarity"][2]) <= 1 and res["off"][1] == res["off"][2], res


@pytest.mark.gpu
@pytest.mark.parametrize("G,L,N", [(16, 1, 14), (16, 1, 5), (21, 1, 20), (21, 1, 17), (32, 1, 30), (32, 1, 20), (64, 1, 50), (64, 1, 62), (64, 1, 33),
                                   (64, 2, 31), (64, 2, 25), (64, 3, 20), (64, 3, 10)])
def test_stationarity_sweep_against_the_adjoint_recursion(built, G, L, N):
    """The sweep behind polish indicator (c) on its own, through the C ABI (mpc_debug_adjoint_dev -> rti_kernel.hpp::adjoint_inputs): for RANDOM per-stage gradients
    g and a random iterate, the kernels' three levels of suffix sums over the stage lanes against the open-loop adjoint recursion written out in numpy over the
    oracle's linearisation, pi_N = g_x,N, pi_i = g_x,i + A_i' pi_{i+1}, ru_i = g_u,i + B_i' pi_{i+1} -- in every lane layout a solve kernel uses (16, 21, 32, 64 lanes per
    instance with one lane per stage: whole DPP rows, and the three-instances-per-wavefront packing whose segments straddle rows; 2 and 3 lanes per stage with one
    instance per wavefront), horizons that fill the layout and horizons that do not, a batch that ends in a partly filled wavefront.  1e-12 relative."""
    import torch
    import mpc_gpu
    from oracle import oracle as orc
    no, B = 3, 11
    cfg = orc.config(N, no, 0.1 * N)
    rng = np.random.default_rng(1000 + 100 * G + 10 * L + N)
    X = rng.uniform(-7, 7, (B, N + 1, 5)); X[:, :, 2] = rng.uniform(-6, 6, (B, N + 1)); U = rng.uniform(-8, 8, (B, N, 2))
    g = rng.standard_normal((B, N + 1, 7)) * rng.choice([1.0, 1e3, 1e6], (B, N + 1, 1))      # multipliers reach 1e6: so do the gradients the residual is made of
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ru = torch.full((B, N), -1.0, dtype=torch.float64, device=dev)
    with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s:
        s.debug_adjoint_dev(B, G, L, t(X), t(U), t(g), ru, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for Gbad, Lbad in ((16 if N + 1 > 16 else 17, 1), (21 if N + 1 > 21 else 20, 1), (64, 3 if N > 20 else 5), (64, 2 if N > 31 else 4)):
            with pytest.raises(mpc_gpu.MpcError):          # a layout the horizon does not fit, or one that does not exist, is refused
                s.debug_adjoint_dev(B, Gbad, Lbad, t(X), t(U), t(g), ru)
    ru = ru.cpu().numpy()
    zero = np.zeros(5); P0 = np.zeros((N + 1, no, 2))
    for b in range(B):
        lin = orc.linearize(cfg, zero, P0, np.zeros(2), X[b], U[b])
        pi = g[b, N, 2:].copy()
        for i in range(N - 1, -1, -1):
            r = g[b, i, :2] + lin["B"][i].T @ pi
            scale = max(1.0, np.abs(g[b, i:]).max())
            assert abs(ru[b, i] - np.abs(r).max()) <= 1e-12 * scale * (N - i + 1), (b, i, ru[b, i], np.abs(r).max())
            pi = g[b, i, 2:] + lin["A"][i].T @ pi
