"""Oracle and host logic against the golden vectors captured from the reference's importable modules
(tests/golden/make_golden.py): scenario generator, obstacle motion, look-ahead, constants.  CPU only."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "reference_vectors.npz"))
TABLES = json.load(open(os.path.join(HERE, "golden", "reference_tables.json")))


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def test_constants_snapshot():
    from mpc_gpu import world as W
    c = TABLES["constants"]
    for k in ("X_MIN", "X_MAX", "Y_MIN", "Y_MAX", "R_ROBOT", "V_MAX_ROBOT", "C_MAX", "R_OBST", "RANDOMNESS", "V_MAX_OBST",
              "MARGIN", "X_MIN_OBST", "X_MAX_OBST", "Y_MIN_OBST", "Y_MAX_OBST", "TOL", "QP_ITER", "N_OBST"):
        assert getattr(W, k) == pytest.approx(c[k], abs=0), k
    # seed-0 RANDOM, first obstacle
    assert GOLD["gen_RANDOM_5"][0, 0] == pytest.approx([1.30766044, 2.31729878, 1.16690015, -1.65148280], abs=1e-8)


@pytest.mark.parametrize("n_obst", [3, 5, 10])
@pytest.mark.parametrize("scenario", ["RANDOM", "EDGE", "CENTER"])
def test_generator_reproduces_reference_draws(scenario, n_obst):
    from mpc_gpu import world as W
    ref = GOLD[f"gen_{scenario}_{n_obst}"]
    for seed in (0, 1, 17, 99):
        np.random.seed(seed)
        obs = W.generate_random_moving_obstacles(scenario, True, n_obst=n_obst)
        got = W.obstacle_states(obs)
        assert np.array_equal(got, ref[seed])       # bit-exact: same legacy RNG stream, same draw order


@pytest.mark.parametrize("n", [5, 20, 50])
def test_oracle_predictor_bit_exact(orc, n):
    cfg = orc.config(n, 3, 0.1 * n)
    for s, ref in zip(GOLD["pred_states"], GOLD[f"pred_traj_{n}"]):
        assert np.array_equal(orc.predict_trajectory(cfg, s, n, 0.1), ref)


def test_host_predictor_bit_exact_and_bug_switch():
    from mpc_gpu import world as W
    for s, ref in zip(GOLD["pred_states"], GOLD["pred_traj_20"]):
        assert np.array_equal(W.Obstacle(*s, False, dt=0.1).predict_trajectory(20), ref)
    # with the defect fixed the x look-ahead uses vx: differs whenever vx != vy
    s = GOLD["pred_states"][0]
    fixed = W.Obstacle(*s, False, dt=0.1, bug_compat_predict=False).predict_trajectory(5)
    assert fixed[1, 0] == pytest.approx(s[0] + 0.1 * s[2]) and not np.allclose(fixed, GOLD["pred_traj_5"][0])


def test_oracle_deterministic_step_bit_exact(orc):
    cfg = orc.config(20, 3, 2.0)
    for s, ref in zip(GOLD["pred_states"], GOLD["step_det"]):
        assert np.array_equal(orc.obstacle_step(cfg, s, 0.1), ref)


def test_noisy_motion_matches_reference_sequences(orc):
    from mpc_gpu import world as W
    cfg = orc.config(20, 3, 2.0)
    for i in range(GOLD["noisy_seq"].shape[0]):
        st = GOLD["noisy_seq"][i, 0].copy()
        for k in range(30):     # oracle driven by the recorded normals
            st = orc.obstacle_step(cfg, st, 0.1, noise=GOLD["noisy_noise"][i, k], randomness=0.1, vmax=2.0)
            assert np.array_equal(st, GOLD["noisy_seq"][i, k + 1])
        np.random.seed(100 + i)  # host model driven by the same legacy stream
        o = W.Obstacle(*GOLD["noisy_seq"][i, 0], True, dt=0.1)
        for k in range(30):
            o.step()
            assert np.array_equal(o.state, GOLD["noisy_seq"][i, k + 1])


def test_recorded_tables_are_the_expected_statistics():
    """G5: the summary of the reference's recorded closed-loop tables is what the fixtures hold."""
    t = TABLES["tables"]["20221031_215846"]
    assert t["spec"]["N_SOLV"] == 20 and t["spec"]["N_OBST"] == 5 and t["spec"]["QP_ITER"] == 100
    assert t["hit"] == pytest.approx(0.16) and t["reached"] == pytest.approx(0.99) and t["mean_iters"] == pytest.approx(114.78)
    assert t["rows_0_4"][0] == pytest.approx([0, 1, 1.2514572535, 0.1493760006, 105, 0], abs=1e-9)
