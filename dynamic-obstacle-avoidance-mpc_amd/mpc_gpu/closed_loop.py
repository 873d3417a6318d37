"""RobotOcpProblem: the reference's closed-loop RTI simulator (src/simulation/robot_ocp_problem.py:13-311) with
the acados objects replaced by the libmpcgpu shims.  The loop body follows the reference line by line in behaviour
(citations inline); horizon, obstacle count and QP cap are constructor arguments instead of module constants.

Reference defects are reproduced behind switches so recorded statistics can be compared:
  D1 predictor uses vy for vx          -> bug_compat_predict (default True)
  D2 x_guess aliases self.x0           -> bug_compat_alias   (default True)
"""
import numpy as np

from . import world as W
from .acados_shim import AcadosOcpSolverShim, AcadosSimSolverShim
from .solver import BatchedMpc


class RobotOcpProblem:
    def __init__(self, robot_init, robot_end, scenario="RANDOM", slack=True, init_guess_when_error=False,
                 random_move=False, show_pred=False, N=W.N_SOLV, Tf=W.TF, n_obst=W.N_OBST, qp_iter=W.QP_ITER,
                 bug_compat_predict=True, bug_compat_alias=True, device=0, verbose=False):
        self.robot_init = np.array(robot_init, dtype=np.float64)
        self.robot_end = np.array(robot_end, dtype=np.float64)
        self.slack = slack
        self.N, self.Tf, self.n_obst = N, Tf, n_obst
        self.nx, self.nu = 5, 2
        self.bug_compat_predict, self.bug_compat_alias = bug_compat_predict, bug_compat_alias
        self.verbose = verbose
        self.mpc = BatchedMpc(N, n_obst, Tf, max_batch=1, device=device, qp_iter_max=qp_iter,
                              soft_h=1 if slack else 0, bug_compat_predict=1 if bug_compat_predict else 0)
        self.init_experiment(scenario, init_guess_when_error, random_move, show_pred)
        self.ocp_solver = AcadosOcpSolverShim(N, n_obst, Tf, goal=self.robot_end, x0=self.robot_init, mpc=self.mpc)   # :135
        self.ocp_integrator = AcadosSimSolverShim(self.mpc)                                                          # :136
        if self.slack:
            self.parameterize_slack()

    def init_experiment(self, scenario, init_guess_when_error, random_move=False, show_pred=False):   # :35-51
        self.subgoal = self.robot_end
        self.init_guess_when_error = init_guess_when_error
        self.show_pred = show_pred
        self.obstacles = W.generate_random_moving_obstacles(scenario, random_move, n_obst=self.n_obst, dt=self.Tf / self.N)
        for o in self.obstacles:
            o.bug_compat_predict = self.bug_compat_predict
        self.simX = np.ndarray((0, self.nx)); self.simU = np.ndarray((0, self.nu))
        self.pred = np.ndarray((0, self.N + 1, 2))
        self.x0 = self.robot_init if self.bug_compat_alias else self.robot_init.copy()
        self.simX = np.append(self.simX, self.x0.reshape((1, self.nx)), axis=0)
        self.reached_goal = False
        self.min_margin_traj = np.inf

    def parameterize_slack(self):                                                                     # :145-152
        scale = 1e4 * (np.sum((np.take(self.x0, [0, 1, 3, 4]) - np.append(self.subgoal, np.zeros(2))) ** 2) + 50)
        for i in range(self.N + 1):
            alpha_i = scale * (self.N - i) / self.N
            self.ocp_solver.cost_set(i, "zl", alpha_i * np.ones(len(self.obstacles)))
            self.ocp_solver.cost_set(i, "Zl", alpha_i * np.ones(len(self.obstacles)))

    def parameterize_model(self):                                                                     # :154-166
        P = np.ndarray((self.N + 1, self.n_obst, 2))
        for i, o in enumerate(self.obstacles):
            P[:, i, :] = o.predict_trajectory(self.N)
        for i in range(self.N + 1):
            self.ocp_solver.set(i, "p", P[i].flatten())

    def step(self, max_iter, visualize=False):                                                        # :168-277
        reached_subgoal = False
        out_of_bounds = False
        self.set_initial_guess()
        distance_to_goal = np.linalg.norm(np.take(self.x0, [0, 1, 3, 4]) - np.append(self.subgoal, [0, 0]))
        u_max = 0
        i = 0
        N = self.N
        while i < max_iter:
            self.parameterize_model()
            if self.slack:
                self.parameterize_slack()
            self.ocp_solver.set(0, "ubx", self.x0)
            self.ocp_solver.set(0, "lbx", self.x0)
            stat_solv = self.ocp_solver.solve()                                                       # :195
            u = self.ocp_solver.get(0, "u")
            u_max = max(u_max, np.max(np.abs(u)))
            if stat_solv in [4] and self.init_guess_when_error:                                       # :203-205
                self.set_initial_guess()
            self.ocp_integrator.set("x", self.x0)
            self.ocp_integrator.set("u", u)
            self.ocp_integrator.solve()
            self.x0 = self.ocp_integrator.get("x")                                                    # :212
            if self.x0[0] < W.X_MIN or self.x0[0] > W.X_MAX or self.x0[1] < W.Y_MIN or self.x0[1] > W.Y_MAX:
                out_of_bounds = True
            for o in self.obstacles:
                o.step()
            min_margin = np.inf
            for o in self.obstacles:
                margin = np.sqrt((self.x0[0] - o.x) ** 2 + (self.x0[1] - o.y) ** 2) - (o.r + W.R_ROBOT)
                min_margin = min(min_margin, margin)
            self.min_margin_traj = min(self.min_margin_traj, min_margin)
            self.simX = np.append(self.simX, self.x0.reshape((1, self.nx)), axis=0)
            self.simU = np.append(self.simU, u.reshape((1, self.nu)), axis=0)
            if self.show_pred:
                self.pred = np.append(self.pred, self.ocp_solver.X[None, :, :2], axis=0)
            distance_to_goal = np.linalg.norm(self.x0[:2] - self.subgoal)                             # :247
            if distance_to_goal <= W.TOL:
                reached_subgoal = True
                break
            for j in range(N - 1):                                                                    # :253-258
                self.ocp_solver.set(j, "x", self.ocp_solver.get(j + 1, "x"))
                self.ocp_solver.set(j, "u", self.ocp_solver.get(j + 1, "u"))
            self.ocp_solver.set(N - 1, "x", self.ocp_solver.get(N, "x"))
            self.ocp_solver.set(N - 1, "u", np.array([0, 0]))
            i += 1
        if self.verbose:
            print(f"Min margin to obstacle {self.min_margin_traj}")
            print(f"Final difference to sub goal state: {np.linalg.norm((self.simX[-1][0:2] - self.subgoal))}")
            print(f"maximal control along trajectory: {u_max}")
            print(f"left bounds: {out_of_bounds}")
        return (self.simX[-1], (self.min_margin_traj <= 0), reached_subgoal, self.min_margin_traj, distance_to_goal, i,
                out_of_bounds)

    def set_subgoal(self, x, y):                                                                      # :279-284
        self.subgoal = np.array([x, y], dtype=np.float64)
        self.ocp_solver.cost_set(self.N, "yref", np.array([x, y, 0, 0, 0]))

    def set_initial_guess(self):                                                                      # :286-306
        self.ocp_solver.reset()
        x_guess = self.x0 if self.bug_compat_alias else self.x0.copy()
        x_guess[3:] = np.zeros(1)          # with aliasing this zeroes the plant's v, omega (reference defect D2)
        for i in range(self.N + 1):
            if i < self.N:
                self.ocp_solver.set(i, "u", np.zeros(2))
            self.ocp_solver.set(i, "x", x_guess)

    def set_up_new_experiment(self, scenario="RANDOM", init_guess_when_error=False, random_move=False, show_pred=False):
        self.ocp_solver.reset()                                                                       # :309-311
        self.init_experiment(scenario, init_guess_when_error, random_move, show_pred)
