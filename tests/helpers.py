"""Shared synthetic-input generators for tests (distributions: SURVEY.md 8(d), from obstacle_generator.py:10-22)."""
import numpy as np


def random_batch(B, n_obst, seed=1234, moving=True):
    rng = np.random.default_rng(seed)
    x0 = np.zeros((B, 5))
    x0[:, :2] = rng.uniform(-6, 6, (B, 2))
    x0[:, 2] = rng.uniform(-np.pi, np.pi, B)
    goal = rng.uniform(-6, 6, (B, 2))
    obst = np.zeros((B, n_obst, 4))
    obst[:, :, :2] = rng.uniform(-4.4, 6, (B, n_obst, 2))
    if moving:
        obst[:, :, 2:] = rng.uniform(-2, 2, (B, n_obst, 2))
    return x0, goal, obst


def oracle_reference(orc, cfg, x0, P, goal, X, U):
    """Run the oracle on a batch (OpenMP) and return its outputs."""
    return orc.rti_solve_batch(cfg, x0, P, goal, X, U, nthreads=0)


def oracle_P(orc, cfg, obst):
    return np.stack([orc.predict_params(cfg, o) for o in obst])


def oracle_guess(orc, cfg, x0):
    Xs, Us = zip(*[orc.initial_guess(cfg, x) for x in x0])
    return np.stack(Xs), np.stack(Us)


def qp_merit(orc, cfg, x0, P, goal, X, U, Xn, Un):
    """Solver-independent judgement of one RTI step (X, U) -> (Xn, Un): the step as a point of the QP the oracle assembles from
    (x0, P, goal, X, U) (orc_export_qp; slacks eliminated in closed form, s = max(0, -(Cs v + hs))).
    Returns (objective, max equality residual, max bound violation)."""
    q = orc.export_qp(cfg, x0, P, goal, X, U)
    N = cfg.N
    v = np.zeros(7 * N)
    for i in range(N):
        v[7 * i: 7 * i + 2] = Un[i] - U[i]
    for i in range(1, N + 1):
        v[7 * (i - 1) + 2: 7 * (i - 1) + 7] = Xn[i] - X[i]
    s = np.maximum(0.0, -(q["Cs"] @ v + q["hs"])) if len(q["hs"]) else np.zeros(0)
    f = 0.5 * v @ q["H"] @ v + q["g"] @ v + float((q["zs"] * s + 0.5 * q["Zs"] * s * s).sum())
    eq = float(np.abs(q["Aeq"] @ v - q["beq"]).max())
    bnd = float(max((q["lb"] - v).max(), (v - q["ub"]).max(), 0.0))
    return float(f), eq, bnd


def judge_against_oracle(orc, cfg, x0, P, goal, X0, U0, g, Xg, Ug, o, tol_x=1e-6, tol_u=8e-6, alpha=None):
    """GPU result (g, Xg, Ug) against the oracle's (o) for one RTI step of a batch from the iterate (X0, U0), instance by instance -- no "at most k
    instances may differ" clauses.  Per instance:
      * the statuses are equal, or the difference is an at-the-cap borderline: both sides ran to the iteration cap or one short of it (one meets the
        tolerance at iteration cap where the other is a rounding error above it; or the at-the-cap rule separates 2 from 4);
      * status 4 leaves the iterate untouched;
      * a converged instance (status 0 on both sides) is within the tolerance of the oracle -- or, where the QP is ill-conditioned at the float64 floor of an
        interior point, it is judged by the QP itself (qp_merit): the GPU's step satisfies the linearised dynamics and the boxes to 1e-7 and its QP objective
        does not exceed the oracle's;
      * the iteration counts are equal, or they differ by at most 2 AND the oracle's own record shows the end-game: where the earlier side stopped, the
        oracle's largest complementarity product was already below 1e-4 (the last, superlinear iterations: from there ONE step takes it to ~1e-10, and a
        rounding difference decides whether that step lands under the tolerance or just above it).
    Returns the numbers of instances that took each escape, for the caller to bound or report."""
    B = x0.shape[0]
    cap, tol = cfg.qp_iter_max, cfg.qp_tol
    n = dict(status_borderline=0, judged_by_qp=0, iter_borderline=0, converged=0)
    for b in range(B):
        sg, so, ig, io = int(g["status"][b]), int(o["status"][b]), int(g["iters"][b]), int(o["iters"][b])
        if sg != so:
            assert min(ig, io) >= cap - 1, f"instance {b}: status {sg} (GPU, {ig} iterations) vs {so} (oracle, {io}) away from the iteration cap {cap}"
            n["status_borderline"] += 1
            continue
        if so == 4:
            assert np.array_equal(Xg[b], X0[b]) and np.array_equal(Ug[b], U0[b]), f"instance {b}: a failed QP must leave the iterate untouched"
        if so != 0:
            continue
        n["converged"] += 1
        dx, du = np.abs(Xg[b] - o["X"][b]).max(), np.abs(Ug[b] - o["U"][b]).max()
        if dx > tol_x or du > tol_u:
            al = None if alpha is None else alpha[b]
            fg, eqg, bg = qp_merit(orc, cfg, x0[b], P[b], goal[b], X0[b], U0[b], Xg[b], Ug[b]) if al is None else (None, None, None)
            assert al is None, f"instance {b}: |dX| {dx:.2e} beyond the tolerance with an explicit slack schedule (no QP export for it)"
            fo, _, _ = qp_merit(orc, cfg, x0[b], P[b], goal[b], X0[b], U0[b], o["X"][b], o["U"][b])
            assert eqg <= 1e-7 and bg <= 1e-7 and fg <= fo + 1e-7 * max(1.0, abs(fo)), f"instance {b}: |dX| {dx:.2e}, QP objective {fg} vs oracle {fo}, eq {eqg:.1e}, box {bg:.1e}"
            n["judged_by_qp"] += 1
        else:
            assert abs(g["cost"][b] - o["cost"][b]) <= 1e-8 * max(1.0, abs(o["cost"][b])), (b, g["cost"][b], o["cost"][b])
            assert np.abs(g["u0"][b] - o["u0"][b]).max() <= tol_u
        if ig != io:
            assert abs(ig - io) <= 2, f"instance {b}: {ig} (GPU) vs {io} (oracle) interior-point iterations"
            tr = orc.rti_solve_trace(cfg, x0[b], P[b], goal[b], X0[b], U0[b], alpha=None if alpha is None else alpha[b])["trace"]
            m = min(ig, io)
            assert m >= 1 and tr[min(m, len(tr) - 1), 3] <= max(1e4 * tol, 1e-4), f"instance {b}: iteration counts {ig} / {io} differ away from the end-game (oracle cmax {tr[min(m, len(tr) - 1), 3]:.2e} at iteration {m})"
            n["iter_borderline"] += 1
    return n


class OracleLoop:
    """Oracle-side closed loop of ONE instance: the body of RobotOcpProblem.step (robot_ocp_problem.py:184-260) spelled out on the
    oracle's functions -- look-ahead, RTI solve, status-4 reset (with the aliasing defect D2 when alias=True), plant step, noisy
    obstacle motion, margin / arena / goal bookkeeping, warm-start shift.  Test infrastructure (checker for the fused GPU step)."""

    def __init__(self, orc, cfg, x0, goal, obst, reset_on_fail=True, alias=True, randomness=0.1, vmax=2.0, interp=False):
        self.orc, self.cfg = orc, cfg
        self.x = np.array(x0, dtype=np.float64); self.goal = np.array(goal, dtype=np.float64)
        self.obst = np.array(obst, dtype=np.float64)
        self.reset_on_fail, self.alias, self.randomness, self.vmax = reset_on_fail, alias, randomness, vmax
        self.interp = interp                      # set_initial_guess() = the commented straight-line variant (:293-300)
        if alias:
            self.x[3:] = 0.0                      # set_initial_guess() at the start of step() zeroes v, omega through the alias (:301-302)
        self.X, self.U = self.guess()
        self.min_margin, self.flags, self.steps = np.inf, 0, 0
        self.last = None

    def guess(self):
        return self.orc.initial_guess_interp(self.cfg, self.x, self.goal) if self.interp else self.orc.initial_guess(self.cfg, self.x)

    def step(self, noise=None):
        """one control step; returns the oracle's solve result"""
        orc, cfg = self.orc, self.cfg
        dt = cfg.Tf / cfg.N
        if self.flags & 1:
            return None                           # goal reached: the episode is over, the instance idles
        P = orc.predict_params(cfg, self.obst)
        r = orc.rti_solve(cfg, self.x, P, self.goal, self.X, self.U)
        self.X, self.U = r["X"], r["U"]
        u = r["u0"].copy()
        if r["status"] == 4 and self.reset_on_fail:
            if self.alias:
                self.x[3:] = 0.0
            self.X, self.U = self.guess()
        self.x = orc.dynamics(self.x, u, dt)[0]
        for j in range(cfg.n_obst):
            self.obst[j] = orc.obstacle_step(cfg, self.obst[j], dt, None if noise is None else noise[j], self.randomness, self.vmax)
        a = cfg.arena
        if self.x[0] < a[0] or self.x[0] > a[1] or self.x[1] < a[2] or self.x[1] > a[3]:
            self.flags |= 2
        margin = min(np.sqrt((self.x[0] - o[0]) ** 2 + (self.x[1] - o[1]) ** 2) - 1.2 for o in self.obst)
        self.min_margin = min(self.min_margin, margin)
        if self.min_margin <= 0:
            self.flags |= 4
        if np.linalg.norm(self.x[:2] - self.goal) <= 0.15:
            self.flags |= 1
        else:
            self.steps += 1
        self.X, self.U = orc.shift(cfg, self.X, self.U)
        self.last = r
        return r

    def row(self):
        return [float(bool(self.flags & 4)), float(bool(self.flags & 1)), self.min_margin, float(np.linalg.norm(self.x[:2] - self.goal)),
                float(self.steps), float(bool(self.flags & 2))]


class OracleAsAcados:
    """The oracle behind AcadosOcpSolver's method names (set / get / cost_set / reset / solve), so that host logic written against the
    shims (mpc_gpu.closed_loop.ShimLoop) can be exercised without a GPU.  Test infrastructure."""

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
        if fieldname == "yref": self.goal = np.array(v[:2], float)        # the shim applies the position to stage and terminal reference
        else: self.alpha[stage] = v[0]

    def get(self, stage, fieldname):
        return (self.X if fieldname == "x" else self.U)[stage].copy()

    def reset(self):
        self.X[:] = 0; self.U[:] = 0

    def solve(self):
        r = self.orc.rti_solve(self.cfg, self.x0, self.P, self.goal, self.X, self.U, alpha=self.alpha)
        self.X, self.U = r["X"], r["U"]
        return r["status"]


class OraclePlant:
    """AcadosSimSolver's surface over the oracle's integrator step."""

    def __init__(self, orc, dt):
        self.orc, self.dt, self.x, self.u = orc, dt, np.zeros(5), np.zeros(2)

    def set(self, fieldname, v):
        if fieldname == "x": self.x = np.array(v, float)
        else: self.u = np.array(v, float)

    def solve(self):
        self.x = self.orc.dynamics(self.x, self.u, self.dt)[0]

    def get(self, fieldname):
        return self.x.copy()


def exact_from_active_set(q, v0, tol=1e-7):
    """The solution of an exported QP (oracle.export_qp) that owes the interior point nothing but the GUESS of the active set: the active rows are read off the
    candidate v0 (bounds, soft rows h + C v + s >= 0, s >= 0 within `tol`), the equality-constrained QP on them is solved by one dense KKT system (equilibrated,
    iterative refinement with the residual in extended precision: the slack penalties reach 4e6 next to O(0.1) curvature), and the caller verifies the KKT
    conditions of the full QP with what comes back: (v, smallest multiplier of an active row, smallest constraint value, number of active rows, stationarity
    residual).  Multipliers >= 0 and constraints >= 0 to rounding => v is the unique minimiser of the strictly convex QP.  Test infrastructure."""
    nv, ns = q["H"].shape[0], len(q["hs"])
    n = nv + ns
    H = np.zeros((n, n)); H[:nv, :nv] = q["H"]; H[nv:, nv:] = np.diag(q["Zs"])
    g = np.concatenate([q["g"], q["zs"]])
    # slack values implied by v0: s = max(0, -(hs + Cs v)) is the minimiser for a given v when the penalty is positive ... take the interior point's own active set instead
    rho = q["hs"] + q["Cs"] @ v0
    s0 = np.maximum(0.0, -rho)
    rows = []      # equality rows E x = e
    rhs = []
    for r in range(q["Aeq"].shape[0]):
        rows.append(np.concatenate([q["Aeq"][r], np.zeros(ns)])); rhs.append(q["beq"][r])
    ineq = []      # (row, rhs, sign) of the active inequality rows, written as  a x >= b
    for v in range(nv):
        if np.isfinite(q["lb"][v]) and v0[v] - q["lb"][v] < tol:
            a = np.zeros(n); a[v] = 1.0; ineq.append((a, q["lb"][v]))
        if np.isfinite(q["ub"][v]) and q["ub"][v] - v0[v] < tol:
            a = np.zeros(n); a[v] = -1.0; ineq.append((a, -q["ub"][v]))
    for j in range(ns):
        if rho[j] + s0[j] < tol:          # hs + Cs v + s >= 0 active
            a = np.zeros(n); a[:nv] = q["Cs"][j]; a[nv + j] = 1.0; ineq.append((a, -q["hs"][j]))
        if s0[j] < tol:                   # s >= 0 active
            a = np.zeros(n); a[nv + j] = 1.0; ineq.append((a, 0.0))
    E = np.array(rows + [a for a, _ in ineq]); e = np.array(rhs + [b for _, b in ineq])
    m = E.shape[0]
    K = np.block([[H, -E.T], [E, np.zeros((m, m))]])
    b = np.concatenate([-g, e])
    # the slack penalties (up to 4e6) next to O(0.1) curvature make K ill-conditioned: equilibrate, then iterative refinement with the residual in extended precision
    dsc = 1.0 / np.sqrt(np.maximum(np.abs(K).max(axis=1), 1e-300))
    Ks = K * dsc[:, None] * dsc[None, :]
    Kl, bl = K.astype(np.longdouble), b.astype(np.longdouble)
    sol = dsc * np.linalg.lstsq(Ks, dsc * b, rcond=1e-15)[0]
    for _ in range(6):
        r = (bl - Kl @ sol.astype(np.longdouble)).astype(np.float64)
        sol = sol + dsc * np.linalg.lstsq(Ks, dsc * r, rcond=1e-15)[0]
    x, lam = sol[:n], sol[n:]
    lam_in = lam[len(rows):]
    # verification of the KKT conditions on the full QP
    v, s = x[:nv], x[nv:]
    feas = min(np.min(v - q["lb"]), np.min(q["ub"] - v), np.min(q["hs"] + q["Cs"] @ v + s) if ns else 0.0, np.min(s) if ns else 0.0)
    res = float(np.abs((bl - Kl @ sol.astype(np.longdouble)).astype(np.float64)[:n]).max())      # stationarity residual of the refined solve
    return v, float(lam_in.min()) if len(lam_in) else 0.0, float(feas), len(ineq), res
