"""Generate golden vectors by IMPORTING the reference's importable modules (this container only).

Run:  MPLBACKEND=Agg python tests/golden/make_golden.py
Writes tests/golden/reference_vectors.npz and tests/golden/reference_tables.json.

Only input/output DATA is stored; no reference source text is copied.  /root/reference does not exist
on the GPU box, so nothing at test time reads it -- tests consume the committed files.

What can be pinned: the scenario generator (utils/obstacle_generator.py:8-28), the obstacle
motion model and look-ahead (utils/visualization.py:20-79), the constants (models/world_specification.py) and
summary statistics of the recorded closed-loop tables (src/simulation/test_data/*.csv).  The MPC solve itself
(acados/HPIPM) is not importable; it is pinned through the recorded tables, replayed per seed (tests/test_oracle_golden.py, test_gpu_replay.py).
"""
import glob
import importlib
import json
import os
import sys

import numpy as np

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_ref(n_obst):
    """(Re)import the reference modules with N_OBST patched (it is an import-time constant, robot_model.py:36)."""
    for m in [k for k in sys.modules if k.split(".")[0] in ("models", "utils")]:
        del sys.modules[m]
    if REF not in sys.path:
        sys.path.insert(0, REF)
    ws = importlib.import_module("models.world_specification")
    ws.N_OBST = n_obst
    vis = importlib.import_module("utils.visualization")
    gen = importlib.import_module("utils.obstacle_generator")
    return ws, vis, gen


def main():
    os.environ.setdefault("MPLBACKEND", "Agg")
    out = {}
    # G4 constants
    ws, vis, gen = load_ref(5)
    consts = {k: getattr(ws, k) for k in dir(ws) if k.isupper()}
    # G1 scenario generator
    import io
    import contextlib
    for n_obst in (3, 5, 10):
        ws, vis, gen = load_ref(n_obst)
        for scen in ("RANDOM", "EDGE", "CENTER"):
            arr = np.zeros((100, n_obst, 4))
            for seed in range(100):
                np.random.seed(seed)
                with contextlib.redirect_stdout(io.StringIO()):
                    obs = gen.generate_random_moving_obstacles(scen, True)
                arr[seed] = [[o.x, o.y, o.vx, o.vy] for o in obs]
            out[f"gen_{scen}_{n_obst}"] = arr
    # G2 predictor: random interior cases + forced wall-bounce cases, N in {5, 20, 50}; dt = TF/N_SOLV = 0.1 (module constants)
    ws, vis, gen = load_ref(5)
    rng = np.random.RandomState(7)
    states = np.concatenate([
        np.column_stack([rng.uniform(-8, 8, 40), rng.uniform(-8, 8, 40), rng.uniform(-2, 2, 40), rng.uniform(-2, 2, 40)]),
        np.array([[7.95, 0.0, 2.0, 2.0], [-7.95, 7.9, -2.0, 2.0], [0.0, -7.99, 0.5, -2.0], [7.0, 7.0, 1.0, 1.5],
                  [8.0, 8.0, 2.0, 2.0], [-8.0, -8.0, -2.0, -2.0], [1.0, 2.0, 0.0, 0.0], [3.0, -3.0, 1.0, 0.0],
                  [3.0, -3.0, 0.0, -1.0], [7.9, -7.9, 0.3, -0.3]]),
    ])
    out["pred_states"] = states
    for n in (5, 20, 50):
        tr = np.zeros((len(states), n + 1, 2))
        for i, s in enumerate(states):
            tr[i] = vis.Obstacle(*s, False).predict_trajectory(n)
        out[f"pred_traj_{n}"] = tr
    # deterministic single steps (noise=False) incl. velocity sign flips
    one = np.zeros((len(states), 4))
    for i, s in enumerate(states):
        o = vis.Obstacle(*s, False)
        x, vx, y, vy = o.predict_step(o.x, o.vx, o.y, o.vy, noise=False)
        one[i] = [x, y, vx, vy]
    out["step_det"] = one
    # G3 noisy ground-truth motion: 30 steps for 8 obstacles, recording the normals consumed so a restatement
    # can be driven with the same noise (np.random.normal(size=2) per step, visualization.py:31)
    nz_states = states[:8]
    seq = np.zeros((8, 31, 4)); noise = np.zeros((8, 30, 2))
    for i, s in enumerate(nz_states):
        np.random.seed(100 + i)
        st = np.random.get_state()
        o = vis.Obstacle(*s, True)
        seq[i, 0] = [o.x, o.y, o.vx, o.vy]
        for k in range(30):
            o.step()
            seq[i, k + 1] = [o.x, o.y, o.vx, o.vy]
        np.random.set_state(st)
        for k in range(30):
            noise[i, k] = np.random.normal(size=2)
    out["noisy_seq"] = seq
    out["noisy_noise"] = noise
    np.savez_compressed(os.path.join(OUT, "reference_vectors.npz"), **out)

    # G5 recorded closed-loop tables -> summary statistics + the seeds that are bit-stable across QP_ITER caps
    tables = {}
    for f in sorted(glob.glob(os.path.join(REF, "simulation/test_data/*_experiment_data.csv"))):
        stem = os.path.basename(f).replace("_experiment_data.csv", "")
        d = np.loadtxt(f, delimiter=";")
        spec = json.load(open(f.replace("_data.csv", "_spec.json")))
        # rows: the full 100 x 6 table (data the reference holds: [hit, reached, min_margin, dist_to_goal, iters, out_of_bounds] per seed,
        # robot_ocp_problem.py:277 sliced [1:] at experiments.py:36), so that closed loops can be replayed PER SEED
        tables[stem] = dict(spec=spec, hit=float(d[:, 0].mean()), reached=float(d[:, 1].mean()),
                            mean_iters=float(d[:, 4].mean()), oob=float(d[:, 5].mean()),
                            min_margin_mean=float(d[:, 2].mean()), rows_0_4=d[:5].tolist(), rows=d.tolist())
    json.dump(dict(constants=consts, tables=tables), open(os.path.join(OUT, "reference_tables.json"), "w"), indent=1)
    print("wrote", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
