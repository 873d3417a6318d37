"""AcadosOcpSolver / AcadosSimSolver shaped shims over libmpcgpu, so the reference's closed-loop logic
(src/simulation/robot_ocp_problem.py:168-277) runs unchanged on top of the HIP solve.

Only the methods and fields the reference actually calls exist:
  ocp:  set(stage, 'x'|'u'|'p'|'lbx'|'ubx', v), set_params_sparse(stage, idx, v), cost_set(stage, 'zl'|'Zl'|'yref', v),
        solve() -> int, get(stage, 'x'|'u'), reset()
  sim:  set('x'|'u', v), solve(), get('x')
Host arrays are copied in on set and fresh copies are returned by get, as acados does.
"""
import numpy as np

from .solver import BatchedMpc


class AcadosOcpSolverShim:
    def __init__(self, N=20, n_obst=5, Tf=2.0, goal=(0.0, 0.0), x0=None, device=0, mpc=None, **cfg):
        self.N, self.n_obst = N, n_obst
        self.mpc = mpc if mpc is not None else BatchedMpc(N, n_obst, Tf, max_batch=1, device=device, **cfg)
        self.X = np.zeros((N + 1, 5)); self.U = np.zeros((N, 2))
        self.P = np.zeros((N + 1, n_obst, 2))
        self.goal = np.array(goal, dtype=np.float64)
        self.x0 = np.zeros(5) if x0 is None else np.array(x0, dtype=np.float64)   # constraints.x0, :87
        self.zl = np.zeros((N + 1, n_obst)); self.Zl = np.zeros((N + 1, n_obst))
        self._slack_touched = False     # until cost_set('zl'|'Zl') is called the kernel's built-in schedule (:145-148) applies
        self.status = 0; self.iters = 0; self.cost = 0.0

    # -- setters ---------------------------------------------------------------------------------------------------
    def set(self, stage, field, value):
        v = np.asarray(value, dtype=np.float64)
        if field == "x":
            self.X[stage] = v
        elif field == "u":
            self.U[stage] = v
        elif field == "p":
            self.P[stage] = v.reshape(self.n_obst, 2)              # p = [o0x, o0y, o1x, ...], :166
        elif field in ("lbx", "ubx"):
            if stage != 0:
                raise ValueError("only the initial-state bounds (stage 0) are settable, robot_ocp_problem.py:191-192")
            self.x0 = v.copy()
        else:
            raise ValueError(f"unsupported field {field!r}")

    def set_params_sparse(self, stage, idx, values):
        flat = self.P[stage].reshape(-1)
        flat[np.asarray(idx, dtype=int)] = np.asarray(values, dtype=np.float64)   # :165

    def cost_set(self, stage, field, value):
        v = np.asarray(value, dtype=np.float64)
        if field in ("zl", "Zl"):       # parameterize_slack, :149-152: forwarded to the solve (mpc_set_slack_schedule) at solve()
            v = np.broadcast_to(v, (self.n_obst,))
            if not np.all(v == v[0]) or not (v[0] >= 0 and np.isfinite(v[0])):
                raise ValueError("libmpcgpu takes one finite slack weight >= 0 per stage (zl_i = Zl_i = alpha_i * ones, robot_ocp_problem.py:149-150)")
            (self.zl if field == "zl" else self.Zl)[stage] = v
            self._slack_touched = True
        elif field == "yref":
            self.goal = v[:2].copy()    # set_subgoal writes [x, y, 0, 0, 0] (:284): only the position is meaningful
        else:
            raise ValueError(f"unsupported cost field {field!r}")

    def slack_schedule(self):
        """alpha_i of robot_ocp_problem.py:145-152 for the current (x0, goal)."""
        d = np.take(self.x0, [0, 1, 3, 4]) - np.append(self.goal, np.zeros(2))
        scale = 1e4 * (np.sum(d ** 2) + 50)
        return scale * (self.N - np.arange(self.N + 1)) / self.N

    # -- solve / get -----------------------------------------------------------------------------------------------
    def solve(self):
        if self._slack_touched:         # the caller's schedule goes to the kernel as it is
            if not np.array_equal(self.zl, self.Zl):
                raise ValueError("zl and Zl differ: libmpcgpu implements the reference's zl_i = Zl_i (robot_ocp_problem.py:149-152)")
            self.mpc.set_slack_schedule(self.zl[None, :, 0])
        self.mpc.set_warmstart(self.X[None], self.U[None])
        out = self.mpc.solve(self.x0[None], self.P[None], self.goal[None])
        X, U = self.mpc.get_traj(1)
        self.X, self.U = X[0], U[0]
        self.status, self.iters, self.cost = int(out["status"][0]), int(out["iters"][0]), float(out["cost"][0])
        return self.status

    def get(self, stage, field):
        if field == "x":
            return self.X[stage].copy()
        if field == "u":
            return self.U[stage].copy()
        raise ValueError(f"unsupported field {field!r}")

    def get_cost(self):
        return self.cost

    def reset(self):
        self.X[:] = 0.0; self.U[:] = 0.0


class AcadosSimSolverShim:
    """Plant integrator (robot_ocp_problem.py:136,207-212) on the device: same IRK map as the OCP."""

    def __init__(self, mpc):
        self.mpc = mpc
        self.x = np.zeros(5); self.u = np.zeros(2)

    def set(self, field, value):
        if field == "x":
            self.x = np.array(value, dtype=np.float64)
        elif field == "u":
            self.u = np.array(value, dtype=np.float64)
        else:
            raise ValueError(f"unsupported field {field!r}")

    def solve(self):
        self.x = self.mpc.plant_step(self.x[None], self.u[None])[0]
        return 0

    def get(self, field):
        if field == "x":
            return self.x.copy()
        raise ValueError(f"unsupported field {field!r}")
