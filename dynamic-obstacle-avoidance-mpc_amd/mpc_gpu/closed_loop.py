"""Single-scenario closed loop driven through the acados-shaped shims: the host-side check that a caller written against
`AcadosOcpSolver` / `AcadosSimSolver` (the reference's src/simulation/robot_ocp_problem.py) gets the same closed loop from
libmpcgpu.  It issues exactly the solver calls the reference issues per control step (`set(i,'p')`,
`cost_set(i,'zl'|'Zl')`, `set(0,'lbx'|'ubx')`, `solve()`, `get(0,'u')`, integrator `set/solve/get`, per-stage `get/set` for the
warm-start shift, `reset()`), organised here as a small state machine -- `EpisodeState` + `control_step()` + `rollout()` -- rather
than as the reference's monolithic `step()`.  The batched, device-resident equivalent is `episodes.run_episodes`.

`RobotOcpProblem` keeps the reference's constructor / `step()` / `set_subgoal()` / `set_up_new_experiment()` surface on top of it
so that scripts written for the reference run; horizon, obstacle count and QP cap are arguments instead of module constants.
Reference defects behind switches: D1 look-ahead `vx = vy` (bug_compat_predict), D2 `x_guess` aliases the
plant state (bug_compat_alias).
"""
from dataclasses import dataclass, field

import numpy as np

from . import world as W
from .acados_shim import AcadosOcpSolverShim, AcadosSimSolverShim
from .solver import BatchedMpc

_BOXED = [0, 1, 3, 4]          # state entries that enter the slack scale (robot_ocp_problem.py:146)


@dataclass
class EpisodeState:
    x: np.ndarray                                   # plant state
    goal: np.ndarray                                # current sub-goal
    obstacles: list
    steps: int = 0                                  # control steps completed by the current / last rollout() (the reference's local `i`, :183)
    total_steps: int = 0                            # ... and over all rollouts of this episode
    reached: bool = False                           # per rollout, as the reference's locals `reached_subgoal`, `out_of_bounds`, `u_max` (:177-182)
    left_arena: bool = False
    min_margin: float = np.inf                      # kept across rollouts, as the reference's self.min_margin_traj
    u_peak: float = 0.0
    xs: list = field(default_factory=list)          # visited plant states, starting with the initial one
    us: list = field(default_factory=list)
    predictions: list = field(default_factory=list)

    def table_row(self):
        """[hit, reached, min_margin, dist_to_goal, iters, out_of_bounds], robot_ocp_problem.py:277 / experiments.py:36"""
        return [self.min_margin <= 0, self.reached, self.min_margin, float(np.linalg.norm(self.x[:2] - self.goal)), self.steps,
                self.left_arena]


class ShimLoop:
    """One MPC instance in closed loop over (ocp_solver, ocp_integrator) objects with acados' method surface."""

    def __init__(self, ocp_solver, ocp_integrator, N, soft=True, reset_on_failure=False, alias_guess=True, keep_predictions=False):
        self.ocp, self.sim, self.N = ocp_solver, ocp_integrator, N
        self.soft, self.reset_on_failure, self.alias_guess, self.keep_predictions = soft, reset_on_failure, alias_guess, keep_predictions

    # -- uploads in front of a solve ------------------------------------------------------------------------------
    def upload_forecast(self, st):                  # parameterize_model, :154-166
        P = np.stack([o.predict_trajectory(self.N) for o in st.obstacles], axis=1)            # (N+1, n_obst, 2)
        for i, row in enumerate(P.reshape(self.N + 1, -1)):
            self.ocp.set(i, "p", row)

    def upload_slack(self, st):                     # parameterize_slack, :145-152
        dev = st.x[_BOXED] - np.concatenate([st.goal, [0.0, 0.0]])
        alpha = 1e4 * (dev @ dev + 50.0) * (self.N - np.arange(self.N + 1)) / self.N
        ones = np.ones(len(st.obstacles))
        for i, a in enumerate(alpha):
            self.ocp.cost_set(i, "zl", a * ones)
            self.ocp.cost_set(i, "Zl", a * ones)

    def cold_start(self, st):                       # set_initial_guess, :286-306
        self.ocp.reset()
        guess = st.x if self.alias_guess else st.x.copy()
        guess[3:] = 0.0                              # through the alias this also stops the plant (defect D2)
        for i in range(self.N + 1):
            self.ocp.set(i, "x", guess)
            if i < self.N:
                self.ocp.set(i, "u", np.zeros(2))

    def shift_warm_start(self):                     # :253-258: stage j takes stage j+1, the last input is zero, x_N stays
        xs = [self.ocp.get(j, "x") for j in range(1, self.N + 1)]
        us = [self.ocp.get(j, "u") for j in range(1, self.N)] + [np.zeros(2)]
        for j, (x, u) in enumerate(zip(xs, us)):
            self.ocp.set(j, "x", x)
            self.ocp.set(j, "u", u)

    # -- one control step ------------------------------------------------------------------------------------------
    def control_step(self, st):
        """:186-250.  Returns the solver status; st.reached tells whether the episode is over."""
        self.upload_forecast(st)
        if self.soft:
            self.upload_slack(st)
        for bound in ("ubx", "lbx"):
            self.ocp.set(0, bound, st.x)
        status = self.ocp.solve()
        u = self.ocp.get(0, "u")
        st.u_peak = max(st.u_peak, float(np.abs(u).max()))
        if status == 4 and self.reset_on_failure:
            self.cold_start(st)
        self.sim.set("x", st.x); self.sim.set("u", u); self.sim.solve()
        st.x = self.sim.get("x")
        st.left_arena |= bool(not (W.X_MIN <= st.x[0] <= W.X_MAX and W.Y_MIN <= st.x[1] <= W.Y_MAX))
        for o in st.obstacles:
            o.step()
        centres = np.array([[o.x, o.y] for o in st.obstacles]); radii = np.array([o.r for o in st.obstacles])
        st.min_margin = min(st.min_margin, float((np.linalg.norm(centres - st.x[:2], axis=1) - (radii + W.R_ROBOT)).min()))
        st.xs.append(st.x.copy()); st.us.append(u.copy())
        if self.keep_predictions:
            st.predictions.append(np.array([self.ocp.get(j, "x")[:2] for j in range(self.N + 1)]))
        st.reached = bool(np.linalg.norm(st.x[:2] - st.goal) <= W.TOL)
        return status

    def rollout(self, st, max_steps):
        """:177-260: cold start, then control steps until the goal region is reached or max_steps are done.  Step count, goal / arena
        flags and peak control are those of THIS call (locals of the reference's step(), :177-183): a caller that alternates
        set_subgoal() and step(n) -- the sub-goal hook of SURVEY 8(f)-3 -- gets n fresh control steps every time."""
        st.steps, st.reached, st.left_arena, st.u_peak = 0, False, False, 0.0
        self.cold_start(st)
        while st.steps < max_steps:
            self.control_step(st)
            if st.reached:
                break                                # the reference leaves its loop before `i += 1`
            self.shift_warm_start()
            st.steps += 1
            st.total_steps += 1
        return st


class RobotOcpProblem:
    """The reference's class surface (robot_ocp_problem.py:13-311) over ShimLoop and the libmpcgpu shims."""

    def __init__(self, robot_init, robot_end, scenario="RANDOM", slack=True, init_guess_when_error=False,
                 random_move=False, show_pred=False, N=W.N_SOLV, Tf=W.TF, n_obst=W.N_OBST, qp_iter=W.QP_ITER,
                 bug_compat_predict=True, bug_compat_alias=True, device=0, verbose=False):
        self.robot_init = np.array(robot_init, dtype=np.float64)
        self.robot_end = np.array(robot_end, dtype=np.float64)
        self.slack, self.N, self.Tf, self.n_obst = slack, N, Tf, n_obst
        self.bug_compat_predict, self.bug_compat_alias, self.verbose = bug_compat_predict, bug_compat_alias, verbose
        self.mpc = BatchedMpc(N, n_obst, Tf, max_batch=1, device=device, qp_iter_max=qp_iter,
                              soft_h=1 if slack else 0, bug_compat_predict=1 if bug_compat_predict else 0)
        self.ocp_solver = AcadosOcpSolverShim(N, n_obst, Tf, goal=self.robot_end, x0=self.robot_init, mpc=self.mpc)   # :135
        self.ocp_integrator = AcadosSimSolverShim(self.mpc)                                                          # :136
        self.init_experiment(scenario, init_guess_when_error, random_move, show_pred)

    def init_experiment(self, scenario, init_guess_when_error, random_move=False, show_pred=False):   # :35-51
        obstacles = W.generate_random_moving_obstacles(scenario, random_move, n_obst=self.n_obst, dt=self.Tf / self.N)
        for o in obstacles:
            o.bug_compat_predict = self.bug_compat_predict
        start = self.robot_init if self.bug_compat_alias else self.robot_init.copy()
        self.state = EpisodeState(x=start, goal=self.robot_end.copy(), obstacles=obstacles, xs=[start.copy()])
        self.loop = ShimLoop(self.ocp_solver, self.ocp_integrator, self.N, soft=self.slack, reset_on_failure=init_guess_when_error,
                             alias_guess=self.bug_compat_alias, keep_predictions=show_pred)

    def set_up_new_experiment(self, scenario="RANDOM", init_guess_when_error=False, random_move=False, show_pred=False):   # :309-311
        self.ocp_solver.reset()
        self.init_experiment(scenario, init_guess_when_error, random_move, show_pred)

    def set_subgoal(self, x, y):                     # :279-284 (the shim applies the position to stage and terminal reference)
        self.state.goal = np.array([x, y], dtype=np.float64)
        self.ocp_solver.cost_set(self.N, "yref", np.array([x, y, 0, 0, 0]))

    def step(self, max_iter, visualize=False):       # :168-277
        st = self.loop.rollout(self.state, max_iter)
        if self.verbose:
            print(f"min margin {st.min_margin}, distance to sub-goal {np.linalg.norm(st.x[:2] - st.goal)}, peak control {st.u_peak}, "
                  f"left the arena: {st.left_arena}")
        hit, reached, margin, dist, iters, oob = st.table_row()
        return st.xs[-1], hit, reached, margin, dist, iters, oob

    # views the reference's callers read
    obstacles = property(lambda self: self.state.obstacles)
    x0 = property(lambda self: self.state.x)
    subgoal = property(lambda self: self.state.goal)
    simX = property(lambda self: np.array(self.state.xs))
    simU = property(lambda self: np.array(self.state.us).reshape(-1, 2))
    pred = property(lambda self: np.array(self.state.predictions))
    min_margin_traj = property(lambda self: self.state.min_margin)
