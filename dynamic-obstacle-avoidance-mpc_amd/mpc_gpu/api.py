"""`solve(x0, obstacles, ref) -> u*`: the Python call surface of the hot path (BASELINE.json north_star).

In the reference this is the body of RobotOcpProblem.step around `ocp_solver.solve()`
(src/simulation/robot_ocp_problem.py:186-198).  Scalar and batched; state is kept per (N, n_obst, Tf) solver so
consecutive calls warm-start each other like the reference's solver object does.
"""
import numpy as np

from .solver import BatchedMpc
from .world import obstacle_states

_SOLVERS = {}


def get_solver(N=20, n_obst=3, Tf=2.0, max_batch=1, device=0, **cfg):
    key = (N, n_obst, Tf, device, tuple(sorted(cfg.items())))
    s = _SOLVERS.get(key)
    if s is None or s.max_batch < max_batch:
        if s is not None:
            s.close()
        s = BatchedMpc(N, n_obst, Tf, max_batch=max_batch, device=device, **cfg)
        s._primed = 0
        _SOLVERS[key] = s
    return s


def solve(x0, obstacles, ref, N=20, Tf=2.0, solver=None, shift=False, reset=False, full_output=False):
    """One real-time iteration.

    x0        (5,) or (B,5)   current state [x, y, psi, v, omega]
    obstacles list of Obstacle-like objects (.x .y .vx .vy), (n_obst,4) / (B,n_obst,4) states, or a precomputed
              look-ahead P (N+1,n_obst,2) / (B,N+1,n_obst,2)
    ref       (2,) or (B,2)   goal position (sets stage and terminal references; see DESIGN.md on set_subgoal)
    returns   u* = (2,) or (B,2); with full_output a dict(u0, cost, status, iters)
    The first call (or reset=True) initialises the iterate like set_initial_guess(); shift=True applies the reference's
    warm-start shift after the solve.
    """
    x0 = np.asarray(x0, dtype=np.float64)
    scalar = x0.ndim == 1
    x0b = np.atleast_2d(x0)
    B = x0b.shape[0]
    if not isinstance(obstacles, np.ndarray):
        obstacles = obstacle_states(obstacles)
    obs = np.asarray(obstacles, dtype=np.float64)
    if scalar:
        obs = obs[None]
    n_obst = obs.shape[-2]
    refb = np.broadcast_to(np.atleast_2d(np.asarray(ref, dtype=np.float64)), (B, 2)).copy()
    s = solver if solver is not None else get_solver(N, n_obst, Tf, max_batch=B)
    if reset or getattr(s, "_primed", 0) != B:
        s.reset_guess(x0b)
        s._primed = B
    out = s.solve(x0b, obs, refb)
    if shift:
        s.shift(B)
    if full_output:
        return {k: (v[0] if scalar else v) for k, v in out.items()}
    return out["u0"][0] if scalar else out["u0"]
