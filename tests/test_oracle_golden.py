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


# ---- the oracle against the closed loops the reference RECORDED (per seed) --------------------------------------------------------
# Seeds of four recorded tables whose row the oracle reproduces to 1e-6 (profiles/r02_oracle_seed_replay.json lists all of them: 243 of
# the 800 recorded rows to 1e-6, 342 to 1e-3; the others contain an acados QP that hit its iteration cap or failed, where the recorded
# tables themselves disagree between caps).  This is what pins the oracle's SOLVE -- cost scaling, LM term, slack
# schedule, integrator, status-4 handling, the aliasing defect D2 -- to what acados computed in October 2022.
RECORDED_SEEDS = {
    "20221031_215846": [0, 2, 3, 4, 5, 19, 24, 25, 31, 34],        # RANDOM, TF 2, N 20, QP_ITER 100
    "20221031_220136": [13, 19, 22, 41, 44, 47, 48, 53],           # EDGE,   TF 2, N 20, QP_ITER 100
    "20221031_224515": [0, 1, 2, 3, 4, 5, 6, 9, 14, 19],           # RANDOM, TF 1, N 10, QP_ITER 50
    "20221031_221613": [13, 19, 41, 44, 47, 48],                   # EDGE,   TF 2, N 20, QP_ITER 25
}


@pytest.mark.parametrize("stem", sorted(RECORDED_SEEDS))
def test_oracle_closed_loop_reproduces_recorded_rows(orc, stem):
    from helpers import OracleLoop
    from mpc_gpu.world import reference_streams
    t = TABLES["tables"][stem]; sp = t["spec"]; rows = np.array(t["rows"])
    seeds = RECORDED_SEEDS[stem]
    obst, noise = reference_streams(sp["scenario"], seeds, sp["N_OBST"], 400)
    cfg = orc.config(sp["N_SOLV"], sp["N_OBST"], float(sp["TF"]), qp_iter_max=sp["QP_ITER"])
    for b, seed in enumerate(seeds):
        L = OracleLoop(orc, cfg, [-7.0, -7.0, np.pi / 4, 0, 0], [7.0, 7.0], obst[b])        # experiments.py:20
        for k in range(400):
            if L.step(noise[k, b]) is None:
                break
        got = np.array(L.row())
        assert np.array_equal(got[[0, 1, 4, 5]], rows[seed, [0, 1, 4, 5]]), (seed, got, rows[seed])          # hit, reached, steps, oob
        assert np.abs(got[2:4] - rows[seed, 2:4]).max() <= 5e-6, (seed, got, rows[seed])                      # min margin, distance


def test_unscaled_lm_term_does_not_reproduce_the_recorded_rows(orc):
    """the other reading of levenberg_marquardt (added unscaled) is ruled out by the same data"""
    from helpers import OracleLoop
    from mpc_gpu.world import reference_streams
    t = TABLES["tables"]["20221031_215846"]; rows = np.array(t["rows"])
    seeds = [0, 2, 3, 4]
    obst, noise = reference_streams("RANDOM", seeds, 5, 400)
    cfg = orc.config(20, 5, 2.0, qp_iter_max=100, lm_scaled=0)
    for b, seed in enumerate(seeds):
        L = OracleLoop(orc, cfg, [-7.0, -7.0, np.pi / 4, 0, 0], [7.0, 7.0], obst[b])
        for k in range(400):
            if L.step(noise[k, b]) is None:
                break
        assert L.row()[4] != rows[seed, 4]
