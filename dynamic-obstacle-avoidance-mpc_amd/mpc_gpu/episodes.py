"""On-device closed-loop episode harness: `RobotOcpProblem.step` + `experiments.py` semantics for a
whole batch of scenarios, one kernel launch per control step, everything resident in HBM.

Output table columns are the reference's (src/simulation/robot_ocp_problem.py:277, sliced [1:] at experiments.py:36):
    [hit, reached_goal, min_margin, dist_to_goal, iters, out_of_bounds]
and `write_experiment` stores them as `;`-separated CSV + spec JSON like experiments.py:40-43, so tooling written for the
reference's `test_data/` (evaluate_experiments.py:8-18) reads them.
"""
import json
import os
import time

import numpy as np

from . import _lib
from .solver import BatchedMpc


def run_episodes(x0, goal, obst, N=20, Tf=2.0, max_iter=400, random_move=True, init_guess_when_error=True,
                 bug_compat_alias=True, seed=0, device=0, solver=None, n_obst=5, first_seed=0, record=False, noise=None,
                 interpolate_init=False, status_log=False, compact_from=4096, **cfg):
    """x0 (B,5), goal (B,2), obst (B,n_obst,4) -- or a scenario name ("RANDOM" | "CENTER" | "EDGE"): instance s then starts
    from the reference generator's draw for np.random.seed(first_seed + s), produced on the device (experiments.py:26-29).
    record=True also returns simX (steps+1,B,5), obst_traj (steps+1,B,n_obst,4) and pred (steps,B,N+1,5): what the reference keeps
    for its visualisation (robot_ocp_problem.py:232-240,270-276).
    noise: an array (steps, B, n_obst, 2) -> exactly these normals, control step k using noise[k] (world.reference_streams gives the sequences the
    reference's own runs consumed, per seed); "reference" -> the same sequences produced ON THE DEVICE per instance (numpy's legacy generator for seed
    first_seed + s continued behind the scenario draw: mpc_noise_init_dev / mpc_noise_draw_dev, no host upload) -- the default when `obst` is a scenario
    name, i.e. run_episodes(x0, goal, "RANDOM", first_seed=0) IS experiments.py:20-36 for seeds 0 .. B-1; "torch" (and None with explicit obstacle states)
    -> standard normals from torch's generator (`seed`).
    interpolate_init: set_initial_guess() is the straight-line variant the reference keeps commented out (robot_ocp_problem.py:293-300; spec key
    `interpolate_init` of two recorded tables) -- at the start and on every status-4 reset.
    status_log: also return, per episode, how many of its solves ended with status 2 / status 4 and the first control step with a status != 0
    (-1: none) -- `status2`, `status4`, `first_bad`.
    compact_from: batches of at least this many episodes are COMPACTED while they run -- an episode that has reached its goal idles in its wavefront slot, and
    with the reference's protocol the mean episode is 120-170 of 400 control steps long: every 25 control steps, once a quarter of the episodes in the batch
    have finished, their results are parked and the live ones move together (device-side gathers, one host read of the count).  Each episode's arithmetic is
    its own: results are those of the uncompacted run, bit for bit where an instance's result does not depend on its wavefront neighbours (one instance
    per wavefront) and to the rounding of the wavefront sums otherwise (three per wavefront).  Off with record / status_log; None: never.
    Returns dict(table (B,6), x_last (B,5), steps_run, solves)."""
    import torch
    x0 = np.ascontiguousarray(x0, dtype=np.float64); B = x0.shape[0]
    scenario_name = obst if isinstance(obst, str) else None
    if interpolate_init and bug_compat_alias:
        import warnings
        warnings.warn("interpolate_init with bug_compat_alias=True: the straight-line guess is built from a plant state whose v, omega the aliasing defect has "
                      "zeroed; the two recorded `interpolate_init` tables replay with bug_compat_alias=False (tests/test_gpu_replay.py)", stacklevel=2)
    if scenario_name is not None:
        if noise is None and random_move:
            noise = "reference"
            if seed != 0:
                raise ValueError("`seed` selects torch's generator, which a scenario name does not use: the obstacle noise is the reference's own numpy stream "
                                 "per instance (first_seed + s).  Vary first_seed, or pass noise='torch' to draw from torch's generator with this seed")
        with BatchedMpc(N, n_obst, Tf, max_batch=B, device=device) as g:
            obst = g.generate_scenarios(obst, B, seed0=first_seed)
    obst = np.ascontiguousarray(obst, dtype=np.float64); n_obst = obst.shape[1]
    goal = np.ascontiguousarray(np.broadcast_to(goal, (B, 2)), dtype=np.float64)
    dev = torch.device("cuda", device)
    m = solver or BatchedMpc(N, n_obst, Tf, max_batch=B, device=device, **cfg)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        t = lambda a: torch.from_numpy(a.copy()).to(dev)
        dx0, dgoal, dobst = t(x0), t(goal), t(obst)
        if bug_compat_alias:                      # set_initial_guess() aliases self.x0 and zeroes v, omega (defect D2)
            dx0[:, 3:] = 0.0
        X = torch.zeros(B, N + 1, 5, dtype=torch.float64, device=dev); U = torch.zeros(B, N, 2, dtype=torch.float64, device=dev)
        status = torch.zeros(B, dtype=torch.int32, device=dev); iters = torch.zeros(B, dtype=torch.int32, device=dev)
        margin = torch.full((B,), float("inf"), dtype=torch.float64, device=dev)
        flags = torch.zeros(B, dtype=torch.int32, device=dev); steps = torch.zeros(B, dtype=torch.int32, device=dev)
        s = stream.cuda_stream
        if interpolate_init:
            m.reset_guess_interp_dev(B, dx0, dgoal, X, U, stream=s)
        else:
            m.reset_guess_dev(B, dx0, X, U, stream=s)        # set_initial_guess() at the start of step(), :180
        fl = _lib.STEP_SHIFT | _lib.STEP_PLANT | _lib.STEP_OBSTACLES | _lib.STEP_METRICS
        if init_guess_when_error:
            fl |= _lib.STEP_RESET_ON_FAIL | (_lib.STEP_ALIAS_BUG if bug_compat_alias else 0) | (_lib.STEP_INTERP_GUESS if interpolate_init else 0)
        if status_log:
            n2 = torch.zeros(B, dtype=torch.int32, device=dev); n4 = torch.zeros(B, dtype=torch.int32, device=dev)
            first_bad = torch.full((B,), -1, dtype=torch.int32, device=dev)
        gen = torch.Generator(device=dev); gen.manual_seed(seed)
        dnoise = None
        gen_state, nbuf = None, None
        if isinstance(noise, str) and noise == "torch":
            noise = None
        if isinstance(noise, str):
            if noise != "reference" or scenario_name is None:
                raise ValueError("noise='reference' continues the generator of a scenario draw: pass the scenario name as `obst`")
            if random_move:
                gen_state = m.noise_state(B, scenario_name, seed0=first_seed, stream=s)
                nbuf = torch.zeros(B, n_obst, 2, dtype=torch.float64, device=dev)
        elif noise is not None and random_move:
            noise = np.ascontiguousarray(noise, dtype=np.float64)
            if noise.shape[1:] != (B, n_obst, 2) or noise.shape[0] < max_iter:
                raise ValueError(f"noise must be (>= {max_iter}, {B}, {n_obst}, 2), got {noise.shape}")
            dnoise = torch.from_numpy(noise).to(dev)
        k = 0
        # compaction of finished episodes (compact_from): results of parked episodes live in full-size arrays, `ids` maps the live batch to them
        compact = compact_from is not None and B >= compact_from and not record and not status_log
        B0 = B
        if compact:
            ids = torch.arange(B, device=dev)
            full = dict(margin=margin.clone(), flags=flags.clone(), steps=steps.clone(), x=dx0.clone())
        rec_x, rec_o, rec_p = [dx0.clone()], [dobst.clone()], []
        # "every instance has reached its goal" without stalling the queue: every 25 control steps the flag goes to pinned host memory behind an event,
        # and is looked at only once that event has passed (so the loop runs at most ~25 steps longer than it has to -- on idle instances, which cost nothing)
        done_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        done_event = None
        while k < max_iter:
            if not random_move:
                nz = None
            elif gen_state is not None:
                m.noise_draw_dev(B, gen_state, nbuf, ep_flags=flags, stream=s); nz = nbuf
            elif dnoise is not None:
                nz = dnoise[k] if B == B0 else dnoise[k][ids]
            else:
                nz = torch.randn(B, n_obst, 2, dtype=torch.float64, device=dev, generator=gen)
            m.closed_loop_step_dev(B, dx0, dobst, dgoal, X, U, None, None, status, iters, nz, flags=fl,
                                   min_margin=margin, ep_flags=flags, ep_steps=steps, stream=s)
            if status_log:      # (an episode that has reached its goal idles: its status word keeps the last solve's value and is not counted again)
                live = (steps + (flags & 1)) > k
                n2 += (live & (status == 2)).int(); n4 += (live & (status == 4)).int()
                first_bad = torch.where(live & (status != 0) & (first_bad < 0), torch.full_like(first_bad, k), first_bad)
            k += 1
            if record:      # X holds the shifted prediction: stage j of the solve is X[j - 1], stage N is kept (:253-258)
                rec_x.append(dx0.clone()); rec_o.append(dobst.clone()); rec_p.append(X.clone())
            if done_event is not None and done_event.query():
                if int(done_host[0]) == 1:
                    break
                done_event = None
            if compact and k % 25 == 0:
                live = (flags & 1) == 0
                n_live = int(live.sum().item())                 # (the one host read: at these batch sizes 25 control steps take tens of milliseconds)
                if n_live == 0:
                    break
                if n_live <= 0.75 * B:
                    park = ids[~live]
                    full["margin"][park] = margin[~live]; full["flags"][park] = flags[~live]; full["steps"][park] = steps[~live]; full["x"][park] = dx0[~live]
                    take = lambda a: a[live].contiguous()
                    ids, dx0, dgoal, dobst, X, U = take(ids), take(dx0), take(dgoal), take(dobst), take(X), take(U)
                    status, iters, margin, flags, steps = take(status), take(iters), take(margin), take(flags), take(steps)
                    if gen_state is not None:
                        gen_state, nbuf = take(gen_state), take(nbuf)
                    B = n_live
                continue
            if k % 25 == 0 and done_event is None:
                done_host.copy_((flags & 1).min().to(torch.int32).reshape(1), non_blocking=True)
                done_event = torch.cuda.Event(); done_event.record(stream)
        stream.synchronize()
        if compact:
            full["margin"][ids] = margin; full["flags"][ids] = flags; full["steps"][ids] = steps; full["x"][ids] = dx0
            margin, flags, steps, dx0 = full["margin"], full["flags"], full["steps"], full["x"]
        fl_h = flags.cpu().numpy(); xl = dx0.cpu().numpy()
        table = np.column_stack([(fl_h & 4) != 0, (fl_h & 1) != 0, margin.cpu().numpy(),
                                 np.linalg.norm(xl[:, :2] - goal, axis=1), steps.cpu().numpy(), (fl_h & 2) != 0]).astype(np.float64)
        extra = {}
        if record:
            extra = dict(simX=torch.stack(rec_x).cpu().numpy(), obst_traj=torch.stack(rec_o).cpu().numpy(), pred=torch.stack(rec_p).cpu().numpy())
        if status_log:
            extra.update(status2=n2.cpu().numpy(), status4=n4.cpu().numpy(), first_bad=first_bad.cpu().numpy())
    if solver is None:
        m.close()
    return dict(table=table, x_last=xl, steps_run=k, solves=int(steps.sum().item()) + int((fl_h & 1).sum()), **extra)


def visualisation_inputs(rec, instance, steps=None):
    """What the reference hands to its `VisDynamicRobotEnv` (robot_ocp_problem.py:270-276, visualization.py:135-151) for ONE instance of a
    recorded batch (`rec` = run_episodes(..., record=True)):
        trajectory   (2, T)          -> vis.set_trajectory(simX[:, :2].T)
        pred         (T, N + 1, 2)   -> vis.set_pred_trajectories(pred): row 0 zeros (init_experiment, :48-49), row k the horizon solved at step k
        obstacles    [n_obst x (2, T)] -> vis.set_obst_trajectory([o.get_trajectory().T ...])
    with T = steps + 1 (default: the instance's own episode length).  The recorded iterate is the SHIFTED one (stage j of the solve sits at
    X[j - 1], stage N is kept, :253-258) and stage 0 of a solve is the plant state it started from, so the solved horizon is re-assembled here."""
    simX, obst, pred = rec["simX"], rec["obst_traj"], rec["pred"]
    if steps is None:
        steps = int(rec["table"][instance, 4]) + int(rec["table"][instance, 1])      # control steps run: i, plus the one that reached the goal
    steps = min(steps, pred.shape[0])
    T = steps + 1
    N = pred.shape[2] - 1
    horizon = np.zeros((T, N + 1, 2))
    for k in range(steps):
        horizon[k + 1, 0] = simX[k, instance, :2]
        horizon[k + 1, 1:N] = pred[k, instance, 0:N - 1, :2]
        horizon[k + 1, N] = pred[k, instance, N, :2]
    return dict(trajectory=simX[:T, instance, :2].T.copy(), pred=horizon,
                obstacles=[obst[:T, instance, j, :2].T.copy() for j in range(obst.shape[2])])


def write_experiment(table, spec, out_dir, stamp=None):
    """experiments.py:28-43 file format: `<stamp>_experiment_data.csv` (';' separated) + `<stamp>_experiment_spec.json`."""
    os.makedirs(out_dir, exist_ok=True)
    stamp = stamp or time.strftime("%Y%m%d_%H%M%S")
    np.savetxt(os.path.join(out_dir, f"{stamp}_experiment_data.csv"), table, delimiter=";")
    with open(os.path.join(out_dir, f"{stamp}_experiment_spec.json"), "w") as f:
        json.dump(spec, f)
    return stamp
