"""World constants, obstacle model and scenario generator: host-side mirror of the reference's
src/models/world_specification.py, src/utils/visualization.py:10-85 (Obstacle) and
src/utils/obstacle_generator.py:8-28, re-implemented (no matplotlib, no import-time N_OBST).

TF / N_SOLV / N_OBST are runtime parameters here (the reference edits them textually in the source file,
src/simulation/run_multiple_experiments.py:8-29).
"""
import numpy as np

# arena, world_specification.py:7-10
Y_MIN = -8
Y_MAX = -Y_MIN
X_MIN = Y_MIN
X_MAX = Y_MAX
# robot, :13-19
R_ROBOT = 0.2
V_MAX_ROBOT = 10
Y_MIN_ROBOT = Y_MIN + 2
Y_MAX_ROBOT = -Y_MIN_ROBOT
X_MIN_ROBOT = Y_MIN_ROBOT
X_MAX_ROBOT = Y_MAX_ROBOT
# controls, :22
C_MAX = 8
# obstacles, :25-40
N_OBST = 5
R_OBST = 1
RANDOMNESS = 0.1
V_MAX_OBST = 2
MARGIN = 1.2
R_MAX_OBST = 1.0
Y_MIN_OBST = Y_MIN_ROBOT + R_MAX_OBST + 3 * R_ROBOT
Y_MAX_OBST = -Y_MIN_ROBOT
X_MIN_OBST = Y_MIN_OBST
X_MAX_OBST = Y_MAX_ROBOT
# horizon defaults used by the recorded experiments (test_data/*_spec.json): TF = 2, N_SOLV = 20
TF = 2.0
N_SOLV = 20
TOL = 0.15          # :45
QP_ITER = 50        # :48


class Obstacle:
    """Constant-velocity disc with wall reflection (visualization.py:10-79).  `dt` replaces TF / N_SOLV (:26)."""

    def __init__(self, x_pos, y_pos, vx, vy, random_move=False, dt=0.1, bug_compat_predict=True, rng=None):
        self.x, self.y, self.vx, self.vy = float(x_pos), float(y_pos), float(vx), float(vy)
        self.r = R_OBST
        self.random_move = random_move
        self.dt = dt
        self.bug_compat_predict = bug_compat_predict
        self.rng = rng          # None -> numpy's global legacy stream, as the reference (np.random.normal, :31)
        self.traj = [[self.x, self.y]]

    @property
    def state(self):
        return np.array([self.x, self.y, self.vx, self.vy])

    def step(self):
        self.x, self.vx, self.y, self.vy = self.predict_step(self.x, self.vx, self.y, self.vy, noise=True)
        self.traj.append([self.x, self.y])

    def predict_step(self, x, vx, y, vy, noise=False):
        dt = self.dt
        if self.random_move and noise:      # :28-33
            n = (self.rng.normal(size=2) if self.rng is not None else np.random.normal(size=2))
            vx = min(max((1 + RANDOMNESS * n[0]) * vx, -V_MAX_OBST), V_MAX_OBST)
            vy = min(max((1 + RANDOMNESS * n[1]) * vy, -V_MAX_OBST), V_MAX_OBST)
        x, vx = _advance(x, vx, X_MIN, X_MAX, dt)
        y, vy = _advance(y, vy, Y_MIN, Y_MAX, dt)
        return x, vx, y, vy

    def predict_trajectory(self, n):
        """(n+1, 2) noise-free look-ahead; reproduces `vx = self.vy` (visualization.py:69) when bug_compat_predict."""
        x, y, vy = self.x, self.y, self.vy
        vx = self.vy if self.bug_compat_predict else self.vx
        traj = np.zeros((n + 1, 2))
        traj[0] = [x, y]
        for i in range(n):
            x, vx, y, vy = self.predict_step(x, vx, y, vy, noise=False)
            traj[i + 1] = [x, y]
        return traj

    def get_trajectory(self):
        return np.array(self.traj)


def _advance(p, v, lo, hi, dt):
    """one axis of visualization.py:35-59"""
    if v < 0:
        t_hit = (p - lo) / abs(v)
    elif v > 0:
        t_hit = (hi - p) / abs(v)
    else:
        t_hit = np.inf
    if t_hit <= dt:
        p += (v * t_hit - v * (dt - t_hit))
        v = -v
    else:
        p += v * dt
    return p, v


def generate_random_moving_obstacles(scenario="RANDOM", random_move=False, n_obst=N_OBST, dt=0.1, rng=None):
    """obstacle_generator.py:8-28.  With rng=None the draws come from numpy's global legacy stream in the reference's
    order (x block, y block, vx block, vy block), so np.random.seed(i) reproduces the reference's scenarios."""
    uni = (rng.uniform if rng is not None else np.random.uniform)
    if scenario == "RANDOM":
        xs = uni(X_MIN_OBST, X_MAX_OBST, (n_obst, 1))
        ys = uni(Y_MIN_OBST, Y_MAX_OBST, (n_obst, 1))
    elif scenario == "CENTER":
        xs = np.zeros((n_obst, 1)); ys = np.zeros((n_obst, 1))
    elif scenario == "EDGE":
        xs = 7 * np.ones((n_obst, 1)); ys = 7 * np.ones((n_obst, 1))
    else:
        raise ValueError(f"unknown scenario {scenario!r}")
    vx = uni(-V_MAX_OBST, V_MAX_OBST, (n_obst, 1))
    vy = uni(-V_MAX_OBST, V_MAX_OBST, (n_obst, 1))
    spec = np.hstack((xs, ys, vx, vy))
    return [Obstacle(*s, random_move, dt=dt, rng=rng) for s in spec]


def obstacle_states(obstacles):
    """list[Obstacle-like with .x .y .vx .vy] | (n_obst,4) array -> (n_obst,4) float64"""
    if isinstance(obstacles, np.ndarray):
        return np.ascontiguousarray(obstacles, dtype=np.float64)
    obstacles = list(obstacles)
    if obstacles and hasattr(obstacles[0], "vx"):
        return np.array([[o.x, o.y, o.vx, o.vy] for o in obstacles], dtype=np.float64)
    return np.ascontiguousarray(obstacles, dtype=np.float64)


def reference_streams(scenario, seeds, n_obst=N_OBST, steps=400):
    """What the reference's experiment loop draws from numpy's global legacy stream for each seed i (experiments.py:33-36):
    after np.random.seed(i) the generator's uniform blocks (obstacle_generator.py:10-22: x, y for RANDOM only, then vx, vy), then
    one np.random.normal(size=2) per obstacle per control step in `for o in self.obstacles: o.step()` (robot_ocp_problem.py:217-218,
    visualization.py:31).  The legacy Gaussian generator hands out its pairs in order, so one normal(size=(steps, n_obst, 2)) call is
    that sequence.  Plain numpy, no reference code: reproducible on any box.
    Returns obst (S, n_obst, 4) and noise (steps, S, n_obst, 2) -- the layout mpc_closed_loop_step_dev takes per control step."""
    seeds = list(seeds)
    obst = np.zeros((len(seeds), n_obst, 4))
    noise = np.zeros((steps, len(seeds), n_obst, 2))
    for k, seed in enumerate(seeds):
        rs = np.random.RandomState(int(seed))
        obst[k] = np.array([o.state for o in generate_random_moving_obstacles(scenario, True, n_obst=n_obst, rng=rs)])
        noise[:, k] = rs.normal(size=(steps, n_obst, 2))
    return obst, noise
