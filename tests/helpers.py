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
