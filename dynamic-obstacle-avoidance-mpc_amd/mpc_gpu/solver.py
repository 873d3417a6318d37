"""BatchedMpc: host-side owner of one libmpcgpu handle (one device, one stream, up to max_batch instances).

Mirrors what the reference keeps inside its `AcadosOcpSolver` object (iterate X, U; parameters; options) for the
solve path of src/simulation/robot_ocp_problem.py:186-198, batched over independent MPC instances.
"""
import ctypes as C
import os

import numpy as np

from . import _lib


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    return C.c_void_p(a.data_ptr())      # torch tensor (device or host)


class BatchedMpc:
    # lanes per horizon stage / wavefronts per SIMD / lanes per instance applied to every new handle (0 = automatic); test / tuning hooks,
    # also settable from the environment for profiling runs of unmodified programs (MPC_LANES_PER_STAGE, MPC_WAVES_PER_SIMD, MPC_LANES_PER_INSTANCE)
    default_lanes_per_stage = int(os.environ.get("MPC_LANES_PER_STAGE", "0"))
    default_waves_per_simd = int(os.environ.get("MPC_WAVES_PER_SIMD", "0"))
    default_lanes_per_instance = int(os.environ.get("MPC_LANES_PER_INSTANCE", "0"))
    default_block_riccati = int(os.environ.get("MPC_BLOCK_RICCATI", "0"))      # 1: stage recursions on pairs of stages (A/B runs of unmodified programs)
    default_matrix_cores = int(os.environ.get("MPC_MATRIX_CORES", "0"))      # 1: the v_mfma_f64_16x16x4 Riccati sweep (evidence path, one instance per wavefront)

    def __init__(self, N=20, n_obst=3, Tf=2.0, max_batch=1, device=0, **cfg_overrides):
        self.cfg = _lib.default_config(N, n_obst, Tf, **cfg_overrides)
        self.N, self.n_obst, self.Tf = int(N), int(n_obst), float(Tf)
        self.dt = self.Tf / self.N
        self.max_batch = int(max_batch)
        self.device = int(device)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().mpc_create(C.byref(self.cfg), self.device, self.max_batch, C.byref(self._h)))
        if BatchedMpc.default_lanes_per_stage:
            _lib.check(_lib.lib().mpc_set_lanes_per_stage(self._h, int(BatchedMpc.default_lanes_per_stage)))
        if BatchedMpc.default_waves_per_simd:
            _lib.check(_lib.lib().mpc_set_waves_per_simd(self._h, int(BatchedMpc.default_waves_per_simd)))
        if BatchedMpc.default_lanes_per_instance:
            _lib.check(_lib.lib().mpc_set_lanes_per_instance(self._h, int(BatchedMpc.default_lanes_per_instance)))
        if BatchedMpc.default_block_riccati:
            _lib.check(_lib.lib().mpc_set_block_riccati(self._h, 1))
        if BatchedMpc.default_matrix_cores:
            _lib.check(_lib.lib().mpc_set_lanes_per_stage(self._h, 1))
            _lib.check(_lib.lib().mpc_set_lanes_per_instance(self._h, 64))
            _lib.check(_lib.lib().mpc_set_matrix_cores(self._h, 1))

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().mpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ host-pointer API (numpy in / numpy out)
    def reset_guess(self, x0):
        """set_initial_guess(), robot_ocp_problem.py:286-306."""
        x0 = _f64(np.atleast_2d(x0))
        _lib.check(_lib.lib().mpc_reset_guess(self._h, x0.shape[0], _ptr(x0)))

    def reset_guess_interp(self, x0, goal):
        """set_initial_guess() of the commented block :293-300 (the `interpolate_init` tables): straight line in y towards the goal."""
        x0 = _f64(np.atleast_2d(x0)); goal = _f64(np.atleast_2d(goal), (x0.shape[0], 2))
        _lib.check(_lib.lib().mpc_reset_guess_interp(self._h, x0.shape[0], _ptr(x0), _ptr(goal)))

    def set_warmstart(self, X, U):
        X, U = _f64(X), _f64(U)
        B = X.shape[0]
        if X.shape != (B, self.N + 1, 5) or U.shape != (B, self.N, 2):
            raise ValueError("X must be (B,N+1,5) and U (B,N,2)")
        _lib.check(_lib.lib().mpc_set_warmstart(self._h, B, _ptr(X), _ptr(U)))

    def get_traj(self, batch):
        X = np.empty((batch, self.N + 1, 5)); U = np.empty((batch, self.N, 2))
        _lib.check(_lib.lib().mpc_get_traj(self._h, batch, _ptr(X), _ptr(U)))
        return X, U

    def shift(self, batch):
        """warm-start shift, robot_ocp_problem.py:253-258."""
        _lib.check(_lib.lib().mpc_shift(self._h, batch))

    def solve(self, x0, obstacles, goal):
        """One RTI step for a batch.  `obstacles` is either the explicit parameter tensor P (B,N+1,n_obst,2)
        (reference API, parameterize_model) or obstacle states (B,n_obst,4) = (x,y,vx,vy) whose look-ahead is
        computed on the device.  Returns dict(u0, cost, status, iters)."""
        x0 = _f64(np.atleast_2d(x0)); B = x0.shape[0]
        goal = _f64(np.atleast_2d(goal), (B, 2))
        obstacles = _f64(obstacles)
        u0 = np.empty((B, 2)); cost = np.empty(B)
        status = np.empty(B, np.int32); iters = np.empty(B, np.int32)
        if obstacles.shape == (B, self.N + 1, self.n_obst, 2):
            fn = _lib.lib().mpc_solve
        elif obstacles.shape == (B, self.n_obst, 4):
            fn = _lib.lib().mpc_solve_obst
        else:
            raise ValueError(f"obstacles must be (B,{self.N + 1},{self.n_obst},2) or (B,{self.n_obst},4), got {obstacles.shape}")
        _lib.check(fn(self._h, B, _ptr(x0), _ptr(obstacles), _ptr(goal), _ptr(u0), _ptr(cost), _ptr(status), _ptr(iters)))
        return dict(u0=u0, cost=cost, status=status, iters=iters)

    def set_slack_schedule(self, alpha):
        """parameterize_slack(), robot_ocp_problem.py:145-152: explicit zl_i = Zl_i = alpha[b, i] for the following solves
        (alpha (B, N+1), or a device tensor of shape (max_batch, N+1) used in place); None = the reference's schedule, in-kernel."""
        if alpha is None:
            _lib.check(_lib.lib().mpc_set_slack_schedule(self._h, 0, None))
        elif isinstance(alpha, np.ndarray) or isinstance(alpha, (list, tuple)):
            alpha = _f64(np.atleast_2d(alpha))
            if alpha.shape[1] != self.N + 1:
                raise ValueError(f"alpha must be (B, {self.N + 1})")
            _lib.check(_lib.lib().mpc_set_slack_schedule(self._h, alpha.shape[0], _ptr(alpha)))
        else:
            if tuple(alpha.shape) != (self.max_batch, self.N + 1):
                raise ValueError(f"a device schedule must be ({self.max_batch}, {self.N + 1})")
            _lib.check(_lib.lib().mpc_set_slack_schedule_dev(self._h, _ptr(alpha)))

    def plant_step(self, x, u):
        """ocp_integrator set/solve/get, robot_ocp_problem.py:207-212."""
        x = _f64(np.atleast_2d(x)); u = _f64(np.atleast_2d(u), (x.shape[0], 2))
        xn = np.empty_like(x)
        _lib.check(_lib.lib().mpc_plant_step(self._h, x.shape[0], _ptr(x), _ptr(u), _ptr(xn)))
        return xn

    def predict(self, obst):
        """Obstacle.predict_trajectory for every obstacle -> P (B,N+1,n_obst,2), visualization.py:62-79."""
        obst = _f64(obst); B = obst.shape[0]
        if obst.shape != (B, self.n_obst, 4):
            raise ValueError("obst must be (B,n_obst,4)")
        P = np.empty((B, self.N + 1, self.n_obst, 2))
        _lib.check(_lib.lib().mpc_predict(self._h, B, _ptr(obst), _ptr(P)))
        return P

    SCENARIOS = {"RANDOM": 0, "CENTER": 1, "EDGE": 2}

    @staticmethod
    def _scenario_box():
        from . import world as w
        return np.array([w.X_MIN_OBST, w.X_MAX_OBST, w.Y_MIN_OBST, w.Y_MAX_OBST, w.V_MAX_OBST, 7.0], dtype=np.float64)

    def generate_scenarios(self, scenario, count, seed0=0):
        """generate_random_moving_obstacles for np.random.seed(seed0 + s), s < count (obstacle_generator.py:8-28) -> (count, n_obst, 4)"""
        obst = np.empty((count, self.n_obst, 4))
        box = self._scenario_box()
        _lib.check(_lib.lib().mpc_generate_scenarios(self._h, count, self.SCENARIOS[scenario], seed0, _ptr(box), _ptr(obst)))
        return obst

    def generate_scenarios_dev(self, scenario, count, obst, seed0=0, stream=None):
        box = self._scenario_box()
        _lib.check(_lib.lib().mpc_generate_scenarios_dev(self._h, count, self.SCENARIOS[scenario], seed0, _ptr(box), _ptr(obst), _ptr(stream)))

    def noise_state(self, count, scenario, seed0=0, stream=None):
        """device generator states of `count` instances after np.random.seed(seed0 + s) and the scenario draw (a torch int32 tensor)"""
        import torch
        st = torch.zeros(count, _lib.lib().mpc_noise_state_words(), dtype=torch.int32, device=torch.device("cuda", self.device))
        _lib.check(_lib.lib().mpc_noise_init_dev(self._h, count, self.SCENARIOS[scenario], seed0, _ptr(st), _ptr(stream)))
        return st

    def noise_draw_dev(self, count, state, noise, ep_flags=None, stream=None):
        """one control step's normals of the reference's stream -> noise (count, n_obst, 2); advances `state`"""
        _lib.check(_lib.lib().mpc_noise_draw_dev(self._h, count, _ptr(state), _ptr(noise), _ptr(ep_flags), _ptr(stream)))

    # ------------------------------------------------------------------ multi-GPU: all-gather of the costs, RCCL called by the library itself
    @staticmethod
    def comm_unique_id():
        """128 bytes that rank 0 creates and hands to every rank (any transport: a file, a socket, torch.distributed's store)"""
        buf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
        _lib.check(_lib.lib().mpc_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, rank, world, unique_id):
        """collective: this handle becomes rank `rank` of `world` of the cost exchange"""
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {_lib.COMM_ID_BYTES} bytes")
        buf = (C.c_ubyte * _lib.COMM_ID_BYTES).from_buffer_copy(unique_id)
        _lib.check(_lib.lib().mpc_comm_init(self._h, int(rank), int(world), buf))

    def comm_world(self):
        return int(_lib.lib().mpc_comm_world(self._h))

    def comm_destroy(self):
        _lib.check(_lib.lib().mpc_comm_destroy(self._h))

    @staticmethod
    def comm_library_path():
        """file name of the RCCL library the exchange is bound to (mpc_comm_library_path)"""
        buf = C.create_string_buffer(1024)
        _lib.check(_lib.lib().mpc_comm_library_path(buf, len(buf)))
        return buf.value.decode()

    def allgather_cost_dev(self, count, cost, cost_all, stream=None):
        """collective: cost (count,) of every rank -> cost_all (world, count), rank-major; device arrays, enqueued on `stream`"""
        _lib.check(_lib.lib().mpc_allgather_cost_dev(self._h, int(count), _ptr(cost), _ptr(cost_all), _ptr(stream)))

    def allgather_cost(self, cost):
        """collective, host arrays: returns (world, count)"""
        cost = _f64(cost).ravel()
        out = np.empty((self.comm_world(), cost.size))
        _lib.check(_lib.lib().mpc_allgather_cost(self._h, int(cost.size), _ptr(cost), _ptr(out)))
        return out

    def terminal_state(self, batch):
        """x_N of the current iterate (the reference reads it at robot_ocp_problem.py:232; hook for a sub-goal policy)"""
        return self.get_traj(batch)[0][:, -1].copy()

    # ------------------------------------------------------------------ device-pointer API (torch tensors or raw ints)
    def iterate_ptrs(self):
        dX, dU, st = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(_lib.lib().mpc_iterate_ptrs(self._h, C.byref(dX), C.byref(dU), C.byref(st)))
        return dX.value, dU.value, st.value

    def solve_dev(self, batch, x0, P, goal, X, U, u0=None, cost=None, status=None, iters=None, stream=None):
        _lib.check(_lib.lib().mpc_solve_dev(self._h, batch, _ptr(x0), _ptr(P), _ptr(goal), _ptr(X), _ptr(U), _ptr(u0),
                                            _ptr(cost), _ptr(status), _ptr(iters), _ptr(stream)))

    def closed_loop_step_dev(self, batch, x0, obst, goal, X, U, u0=None, cost=None, status=None, iters=None, noise=None,
                             randomness=0.1, vmax=2.0, flags=_lib.STEP_SHIFT | _lib.STEP_PLANT | _lib.STEP_OBSTACLES,
                             min_margin=None, ep_flags=None, ep_steps=None, stream=None):
        """One whole control step (look-ahead, solve, plant, obstacle motion, bookkeeping, shift) in one launch."""
        _lib.check(_lib.lib().mpc_closed_loop_step_dev(self._h, batch, _ptr(x0), _ptr(obst), _ptr(goal), _ptr(X), _ptr(U), _ptr(u0),
                                                       _ptr(cost), _ptr(status), _ptr(iters), _ptr(noise), randomness, vmax, flags,
                                                       _ptr(min_margin), _ptr(ep_flags), _ptr(ep_steps), _ptr(stream)))

    def predict_dev(self, batch, obst, P, stream=None):
        _lib.check(_lib.lib().mpc_predict_dev(self._h, batch, _ptr(obst), _ptr(P), _ptr(stream)))

    def shift_dev(self, batch, X, U, stream=None):
        _lib.check(_lib.lib().mpc_shift_dev(self._h, batch, _ptr(X), _ptr(U), _ptr(stream)))

    def reset_guess_dev(self, batch, x0, X, U, stream=None):
        _lib.check(_lib.lib().mpc_reset_guess_dev(self._h, batch, _ptr(x0), _ptr(X), _ptr(U), _ptr(stream)))

    def reset_guess_interp_dev(self, batch, x0, goal, X, U, stream=None):
        _lib.check(_lib.lib().mpc_reset_guess_interp_dev(self._h, batch, _ptr(x0), _ptr(goal), _ptr(X), _ptr(U), _ptr(stream)))

    def plant_step_dev(self, batch, x, u, xn, stream=None):
        _lib.check(_lib.lib().mpc_plant_step_dev(self._h, batch, _ptr(x), _ptr(u), _ptr(xn), _ptr(stream)))

    def obstacle_step_dev(self, count, obst, noise=None, randomness=0.1, vmax=2.0, stream=None):
        _lib.check(_lib.lib().mpc_obstacle_step_dev(self._h, count, _ptr(obst), _ptr(noise), randomness, vmax, _ptr(stream)))

    def linearize_dev(self, batch, x0, P, goal, X, U, A, B, b, q, hval, dh, stream=None):
        _lib.check(_lib.lib().mpc_linearize_dev(self._h, batch, _ptr(x0), _ptr(P), _ptr(goal), _ptr(X), _ptr(U), _ptr(A), _ptr(B),
                                                _ptr(b), _ptr(q), _ptr(hval), _ptr(dh), _ptr(stream)))

    def debug_adjoint_dev(self, batch, lanes_per_instance, lanes_per_stage, X, U, g, ru, stream=None):
        """the polish's stationarity sweep on its own (mpc_debug_adjoint_dev): ru[B][N] from a given gradient g[B][N+1][7] over the linearisation of (X, U)"""
        _lib.check(_lib.lib().mpc_debug_adjoint_dev(self._h, batch, int(lanes_per_instance), int(lanes_per_stage), _ptr(X), _ptr(U), _ptr(g), _ptr(ru), _ptr(stream)))

    # ------------------------------------------------------------------ measurement
    def set_accumulators(self, iters_acc=None, status_acc=None):
        _lib.check(_lib.lib().mpc_set_accumulators(self._h, _ptr(iters_acc), _ptr(status_acc)))

    def profile_enable(self, on=True, every=1):
        """HIP events around every `every`-th solve launch (on=False: off)."""
        _lib.check(_lib.lib().mpc_profile_enable(self._h, int(every) if on else 0))

    def profile_read(self):
        ms, n = C.c_double(), C.c_int()
        _lib.check(_lib.lib().mpc_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def set_row_parallel(self, on=True):
        """Riccati factorisation sweep: row-parallel 64-bit-DPP variant (True) or one-lane systolic sweep (False)."""
        _lib.check(_lib.lib().mpc_set_row_parallel(self._h, 1 if on else 0))

    def set_block_riccati(self, on=True):
        """stage recursions over pairs of stages (opt-in) or one stage per step (default) (mpc_set_block_riccati)"""
        _lib.check(_lib.lib().mpc_set_block_riccati(self._h, 1 if on else 0))

    def set_matrix_cores(self, on=True):
        _lib.check(_lib.lib().mpc_set_matrix_cores(self._h, 1 if on else 0))

    def set_lanes_per_instance(self, lanes):
        _lib.check(_lib.lib().mpc_set_lanes_per_instance(self._h, int(lanes)))

    def lanes_per_instance(self, batch):
        return _lib.lib().mpc_get_lanes_per_instance(self._h, batch)

    def set_lanes_per_stage(self, lanes):
        """0 automatic (small batches: rows of a stage split over 2-3 lanes), 1 one lane per stage, 2 / 3 split mapping."""
        _lib.check(_lib.lib().mpc_set_lanes_per_stage(self._h, int(lanes)))

    def lanes_per_stage(self, batch):
        return _lib.lib().mpc_get_lanes_per_stage(self._h, batch)

    def kernel_name(self, batch, lookahead=True):
        """the solve kernel instantiation a batch of this size runs, as a profiler prints it (without the namespace)"""
        buf = C.create_string_buffer(96)
        _lib.check(_lib.lib().mpc_get_kernel_name(self._h, batch, 1 if lookahead else 0, buf, 96))
        return buf.value.decode()

    def set_instance_scheduling(self, on=True):
        """deal instances to wavefronts in the order of their previous iteration counts (mappings with several instances per wavefront)"""
        _lib.check(_lib.lib().mpc_set_instance_scheduling(self._h, 1 if on else 0))

    def instance_order(self, batch):
        """the permutation in effect for the next launch of `batch` instances, or None for the natural order"""
        order = np.empty(batch, np.int32)
        rc = _lib.lib().mpc_get_instance_order(self._h, batch, _ptr(order))
        if rc < 0:
            _lib.check(rc)
        return order if rc == 1 else None

    def set_waves_per_simd(self, waves):
        """stage-split mapping: 0 automatic, 1 one wavefront per SIMD (512 registers), 2 two (256 registers, compact LDS blocks)"""
        _lib.check(_lib.lib().mpc_set_waves_per_simd(self._h, int(waves)))

    def waves_per_simd(self, batch):
        return _lib.lib().mpc_get_waves_per_simd(self._h, batch)
