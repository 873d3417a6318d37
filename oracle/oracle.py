"""ctypes binding of the CPU ORACLE (oracle/liborc.so) -- test infrastructure, NOT product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pinned per seed by the closed-loop tables the reference recorded (see oracle/mpc_oracle.h, tests/test_oracle_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_d = C.c_double
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class OrcConfig(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("n_obst", C.c_int), ("Tf", _d),
        ("W", _d * 6), ("We", _d * 4), ("lm", _d),
        ("bx_lo", _d * 4), ("bx_hi", _d * 4), ("bu_lo", _d * 2), ("bu_hi", _d * 2),
        ("r_safe", _d), ("slack_a", _d), ("slack_b", _d),
        ("qp_iter_max", C.c_int), ("qp_tol", _d),
        ("cost_scale_dt", C.c_int), ("slack_scale_dt", C.c_int), ("lm_scaled", C.c_int),
        ("bx_terminal", C.c_int), ("soft_h", C.c_int),
        ("arena", _d * 4), ("bug_compat_predict", C.c_int),
        ("mu0", _d), ("thr0", _d),
        ("qp_fail_policy", C.c_int),
        ("polish_ratio", _d), ("polish_tol", _d), ("polish_step_frac", _d), ("polish_res_g", _d),
    ]


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    src = [os.path.join(_HERE, f) for f in ("mpc_oracle.c", "mpc_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return so


_BENCH = False


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def use_bench_build():
    """bench.py's cpu_baseline leg only: switch this process to liborc_bench.so, the same source built -O3 -march=native ON THE BOX THAT
    RUNS IT (rebuilt when the CPU model differs from the one it was built on).  The checker the tests use stays -O2 -ffp-contract=off."""
    global _LIB, _BENCH
    so, tag = os.path.join(_HERE, "liborc_bench.so"), os.path.join(_HERE, "liborc_bench.host")
    src = [os.path.join(_HERE, f) for f in ("mpc_oracle.c", "mpc_oracle.h")]
    stale = (not os.path.exists(so) or not os.path.exists(tag) or open(tag).read() != _cpu_model()
             or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src))
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc_bench.so"], stdout=subprocess.DEVNULL)
        open(tag, "w").write(_cpu_model())
    _BENCH = True
    _LIB = None
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(os.path.join(_HERE, "liborc_bench.so") if _BENCH else build())
        L = _LIB
        cp = C.POINTER(OrcConfig)
        L.orc_default_config.argtypes = [cp, C.c_int, C.c_int, _d]
        L.orc_dynamics.argtypes = [_dp, _dp, _d, _dp, _dp, _dp]
        L.orc_dynamics_collocation.argtypes = [_dp, _dp, _d, C.c_int, _dp, _dp, _dp]
        L.orc_obstacle_step.argtypes = [cp, _dp, _d, C.c_void_p, _d, _d]
        L.orc_predict_trajectory.argtypes = [cp, _dp, C.c_int, _d, _dp]
        L.orc_predict_params.argtypes = [cp, _dp, _dp]
        L.orc_slack_alpha.argtypes = [cp, _dp, _dp, _dp]
        L.orc_initial_guess.argtypes = [cp, _dp, _dp, _dp]
        L.orc_initial_guess_interp.argtypes = [cp, _dp, _dp, _dp, _dp]
        L.orc_shift.argtypes = [cp, _dp, _dp]
        L.orc_linearize.argtypes = [cp] + [_dp] * 11
        L.orc_cost.argtypes = [cp] + [_dp] * 5
        L.orc_cost.restype = _d
        L.orc_rti_solve.argtypes = [cp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _dp]
        L.orc_rti_solve.restype = C.c_int
        L.orc_rti_solve_alpha.argtypes = [cp, _dp, _dp, _dp, C.c_void_p, _dp, _dp, _dp, _dp, _ip, _dp]
        L.orc_rti_solve_alpha.restype = C.c_int
        L.orc_rti_solve_batch.argtypes = [cp, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int]
        L.orc_export_qp.argtypes = [cp] + [_dp] * 15
        L.orc_export_qp.restype = C.c_int
    return _LIB


def config(N=20, n_obst=3, Tf=2.0, **kw):
    c = OrcConfig()
    lib().orc_default_config(C.byref(c), N, n_obst, Tf)
    for k, v in kw.items():
        cur = getattr(c, k)
        if hasattr(cur, "__len__"):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(c, k, v)
    return c


def _a(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def dynamics(x, u, dt):
    xn, A, B = np.zeros(5), np.zeros((5, 5)), np.zeros((5, 2))
    lib().orc_dynamics(_a(x), _a(u), dt, xn, A, B)
    return xn, A, B


def dynamics_collocation(x, u, dt, newton_iter=3):
    xn, A, B = np.zeros(5), np.zeros((5, 5)), np.zeros((5, 2))
    lib().orc_dynamics_collocation(_a(x), _a(u), dt, newton_iter, xn, A, B)
    return xn, A, B


def obstacle_step(cfg, state, dt, noise=None, randomness=0.1, vmax=2.0):
    st = _a(state).copy()
    nz = None if noise is None else _a(noise)
    lib().orc_obstacle_step(C.byref(cfg), st, dt, None if nz is None else nz.ctypes.data, randomness, vmax)
    return st


def predict_trajectory(cfg, state, n, dt):
    traj = np.zeros((n + 1, 2))
    lib().orc_predict_trajectory(C.byref(cfg), _a(state), n, dt, traj)
    return traj


def predict_params(cfg, obst):
    P = np.zeros((cfg.N + 1, cfg.n_obst, 2))
    lib().orc_predict_params(C.byref(cfg), _a(obst), P)
    return P


def slack_alpha(cfg, x0, goal):
    a = np.zeros(cfg.N + 1)
    lib().orc_slack_alpha(C.byref(cfg), _a(x0), _a(goal), a)
    return a


def initial_guess(cfg, x0):
    X, U = np.zeros((cfg.N + 1, 5)), np.zeros((cfg.N, 2))
    lib().orc_initial_guess(C.byref(cfg), _a(x0), X, U)
    return X, U


def initial_guess_interp(cfg, x0, goal):
    """the commented straight-line guess of robot_ocp_problem.py:293-300 (interpolate_init tables)"""
    X, U = np.zeros((cfg.N + 1, 5)), np.zeros((cfg.N, 2))
    lib().orc_initial_guess_interp(C.byref(cfg), _a(x0), _a(goal), X, U)
    return X, U


def shift(cfg, X, U):
    X, U = _a(X).copy(), _a(U).copy()
    lib().orc_shift(C.byref(cfg), X, U)
    return X, U


def linearize(cfg, x0, P, goal, X, U):
    N, no = cfg.N, cfg.n_obst
    A, B, b = np.zeros((N, 5, 5)), np.zeros((N, 5, 2)), np.zeros((N, 5))
    q, h, dh = np.zeros((N + 1, 7)), np.zeros((N + 1, no)), np.zeros((N + 1, no, 2))
    lib().orc_linearize(C.byref(cfg), _a(x0), _a(P), _a(goal), _a(X), _a(U), A, B, b, q, h, dh)
    return dict(A=A, B=B, b=b, q=q, h=h, dh=dh)


def cost(cfg, x0, P, goal, X, U):
    return lib().orc_cost(C.byref(cfg), _a(x0), _a(P), _a(goal), _a(X), _a(U))


def rti_solve(cfg, x0, P, goal, X, U, alpha=None):
    """One RTI step.  alpha: explicit slack weights zl_i = Zl_i per stage (N+1,), default the reference's schedule.
    Returns dict(X, U, u0, cost, status, iters, kkt)."""
    X, U = _a(X).copy(), _a(U).copy()
    u0, cst, it, kkt = np.zeros(2), np.zeros(1), np.zeros(1, np.int32), np.zeros(4)
    al = None if alpha is None else _a(alpha)
    st = lib().orc_rti_solve_alpha(C.byref(cfg), _a(x0), _a(P), _a(goal), None if al is None else al.ctypes.data, X, U, u0, cst, it, kkt)
    return dict(X=X, U=U, u0=u0, cost=float(cst[0]), status=int(st), iters=int(it[0]), kkt=kkt)


def rti_solve_trace(cfg, x0, P, goal, X, U, alpha=None):
    """rti_solve plus the per-iteration record (mu, sigma, alpha, largest complementarity product) of the interior point: (iters, 4)"""
    buf = np.zeros((cfg.qp_iter_max + 2, 4))
    lib().orc_set_trace.argtypes = [C.c_void_p, C.c_int]
    lib().orc_set_trace(buf.ctypes.data, buf.shape[0])
    try:
        r = rti_solve(cfg, x0, P, goal, X, U, alpha=alpha)
    finally:
        lib().orc_set_trace(None, 0)
    r["trace"] = buf[:max(r["iters"], 1)]
    return r


def set_investigation(switches):
    """investigation switches of scripts/converged_unmatched.py (mpc_oracle.c orc_set_investigation); process-wide, never called by tests"""
    lib().orc_set_investigation.argtypes = [C.c_int]
    lib().orc_set_investigation(int(switches))


def rti_solve_batch(cfg, x0, P, goal, X, U, nthreads=0):
    B = x0.shape[0]
    X, U = _a(X).copy(), _a(U).copy()
    u0, cst = np.zeros((B, 2)), np.zeros(B)
    st, it = np.zeros(B, np.int32), np.zeros(B, np.int32)
    lib().orc_rti_solve_batch(C.byref(cfg), B, _a(x0), _a(P), _a(goal), X, U, u0, cst, st, it, nthreads)
    return dict(X=X, U=U, u0=u0, cost=cst, status=st, iters=it)


def predict_params_batch(cfg, obst):
    """bench.py's cpu_baseline: look-ahead of a whole batch in one call"""
    obst = _a(obst); B = obst.shape[0]
    P = np.zeros((B, cfg.N + 1, cfg.n_obst, 2))
    lib().orc_predict_params_batch.argtypes = [C.POINTER(OrcConfig), C.c_int, _dp, _dp]
    lib().orc_predict_params_batch(C.byref(cfg), B, obst, P)
    return P


def advance_batch(cfg, x, u0, obst, X, U):
    """bench.py's cpu_baseline: plant step, noise-free obstacle step and warm-start shift of a whole batch, IN PLACE (C-contiguous float64 arrays)"""
    lib().orc_advance_batch.argtypes = [C.POINTER(OrcConfig), C.c_int, _dp, _dp, _dp, _dp, _dp]
    lib().orc_advance_batch(C.byref(cfg), x.shape[0], x, _a(u0), obst, X, U)


def export_qp(cfg, x0, P, goal, X, U):
    N, no = cfg.N, cfg.n_obst
    nv, nsm = 7 * N, max(1, N * no)
    H, g = np.zeros((nv, nv)), np.zeros(nv)
    Aeq, beq = np.zeros((5 * N, nv)), np.zeros(5 * N)
    lb, ub = np.zeros(nv), np.zeros(nv)
    Cs, hs, zs, Zs = np.zeros((nsm, nv)), np.zeros(nsm), np.zeros(nsm), np.zeros(nsm)
    ns = lib().orc_export_qp(C.byref(cfg), _a(x0), _a(P), _a(goal), _a(X), _a(U), H, g, Aeq, beq, lb, ub, Cs, hs, zs, Zs)
    return dict(H=H, g=g, Aeq=Aeq, beq=beq, lb=lb, ub=ub, Cs=Cs[:ns], hs=hs[:ns], zs=zs[:ns], Zs=Zs[:ns])
