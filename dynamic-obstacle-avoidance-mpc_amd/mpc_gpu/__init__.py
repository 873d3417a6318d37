"""mpc_gpu: MI355X-native batched RTI-NMPC for the Dynamic-Obstacle-Avoidance-MPC hot path.

Host-side mirror of the reference's interface for that path; all arithmetic runs in hand-written HIP kernels behind the
C ABI of include/mpc_gpu.h (libmpcgpu.so).  There is no CPU fallback.
"""
from . import _lib
from ._lib import MpcConfig, MpcError, build, default_config
from .solver import BatchedMpc
from .api import solve, get_solver
from .world import Obstacle, generate_random_moving_obstacles, obstacle_states
from .acados_shim import AcadosOcpSolverShim, AcadosSimSolverShim
from .closed_loop import RobotOcpProblem
from .episodes import run_episodes, visualisation_inputs, write_experiment

__all__ = ["MpcConfig", "MpcError", "build", "default_config", "BatchedMpc", "solve", "get_solver", "Obstacle",
           "generate_random_moving_obstacles", "obstacle_states", "AcadosOcpSolverShim", "AcadosSimSolverShim",
           "RobotOcpProblem", "run_episodes", "visualisation_inputs", "write_experiment"]
