"""Host-side mirror of the reference interface (no GPU needed): shims' set/get plumbing, argument handling, sharding."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_set_get_plumbing(built):
    from mpc_gpu.acados_shim import AcadosOcpSolverShim, AcadosSimSolverShim

    class Fake:            # stands in for the device handle: the plumbing under test never launches anything
        pass
    s = AcadosOcpSolverShim(N=6, n_obst=3, Tf=0.6, mpc=Fake())
    s.set(2, "x", [1, 2, 3, 4, 5]); s.set(1, "u", [0.5, -0.5])
    assert (s.get(2, "x") == [1, 2, 3, 4, 5]).all() and (s.get(1, "u") == [0.5, -0.5]).all()
    g = s.get(2, "x"); g[:] = 0
    assert s.get(2, "x")[0] == 1                                   # get returns copies, like acados
    s.set(3, "p", [1, 2, 3, 4, 5, 6])                              # p = [o0x, o0y, o1x, ...], :166
    assert (s.P[3] == [[1, 2], [3, 4], [5, 6]]).all()
    s.set_params_sparse(3, [0, 5], [9, 8])                         # :165
    assert s.P[3, 0, 0] == 9 and s.P[3, 2, 1] == 8
    s.set(0, "lbx", [1, 1, 0, 0, 0]); s.set(0, "ubx", [1, 1, 0, 0, 0])
    assert (s.x0 == [1, 1, 0, 0, 0]).all()
    with pytest.raises(ValueError):
        s.set(1, "lbx", np.zeros(5))
    s.cost_set(6, "yref", [3, 4, 0, 0, 0])                         # set_subgoal's 5-vector (:284): position only
    assert (s.goal == [3, 4]).all()
    a = s.slack_schedule()                                         # :145-152
    assert a[0] == pytest.approx(1e4 * ((1 - 3) ** 2 + (1 - 4) ** 2 + 50)) and a[-1] == 0
    s.reset()
    assert (s.X == 0).all() and (s.U == 0).all()
    sim = AcadosSimSolverShim(Fake())
    sim.set("x", [1, 2, 3, 4, 5]); sim.set("u", [1, 1])
    assert (sim.get("x") == [1, 2, 3, 4, 5]).all()


def test_shard_slices_cover_the_batch(built):
    from mpc_gpu.sharding import shard_slice
    for total, world in ((262144, 8), (32768, 8), (1000, 3), (7, 8)):
        sl = [shard_slice(total, r, world) for r in range(world)]
        assert sl[0][0] == 0 and sl[-1][1] == total and all(sl[i][1] == sl[i + 1][0] for i in range(world - 1))
    assert shard_slice(262144, 3, 8) == (3 * 32768, 4 * 32768)     # C4: contiguous 32768 per GPU


WORKER = r'''
import os, sys
sys.path[:0] = [%r, %r]
import torch, torch.distributed as dist
from mpc_gpu.sharding import shard_slice, gather_costs, gather_costs_ragged
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
total = 10
lo, hi = shard_slice(total, rank, world)
cost = torch.arange(lo, hi, dtype=torch.float64) * 1.5          # this rank's per-scenario costs
full, _ = gather_costs(cost)
assert torch.equal(full, torch.arange(total, dtype=torch.float64) * 1.5), full
out, work = gather_costs(cost, async_op=True); work.wait()
assert torch.equal(out, full)
lo, hi = shard_slice(7, rank, world)                             # ragged: 4 + 3
rag = gather_costs_ragged(torch.arange(lo, hi, dtype=torch.float64), [4, 3])
assert torch.equal(rag, torch.arange(7, dtype=torch.float64)), rag
dist.destroy_process_group()
print("ok", rank)
'''


def test_cost_allgather_world_size_2_gloo(built, tmp_path):
    """N>1 path on CPU: two processes, gloo backend, contiguous shards + all-gather of per-scenario costs"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, os.path.join(ROOT, "dynamic-obstacle-avoidance-mpc_amd")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", MPC_GPU_NO_TORCH="")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", str(script)], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


_DRY = {}


def _bench_dry_run(workload, fresh=False, world=2):
    import json
    key = workload if world == 2 else (workload, world)
    if key in _DRY and not fresh:
        return _DRY[key]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry-run", "--workload", workload, "--steps", "1",
                        "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    _DRY.setdefault(key, line)
    return line


@pytest.mark.parametrize("workload,total", [("c4", 262144), ("c5", 32768), ("c2", 2048)])
def test_bench_multi_rank_path_world_size_2_gloo(built, workload, total):
    """`python bench.py --gpus 2` starts its two ranks itself (torch.distributed.run), every rank takes its shard_slice of the
    workload's ONE global batch and the per-scenario costs of 50 control steps are exchanged with sharding.gather_costs -- rehearsed
    on CPU with gloo (--dry-run: the same plumbing, no kernels; each cost is the instance's global index, so a wrong slice or a
    wrong gather order is visible)."""
    line = _bench_dry_run(workload)
    assert line["n_gpus"] == 2 and line["gather_check"] is True and line["config"]["global_batch"] == total
    assert line["config"]["rank0_slice"] == [0, total // 2] and line["scaling"] == ("weak" if workload == "c2" else "strong")
    # the id plumbing of the C-ABI exchange (--exchange capi, the default): every rank ended with rank 0's 128 bytes (gather_check covers it)
    assert line["exchange"] == "capi" and isinstance(line["comm_id_sha1"], str) and len(line["comm_id_sha1"]) == 40


@pytest.mark.parametrize("workload,total,per_rank", [("c4", 262144, 32768), ("c5", 32768, 4096)])
def test_bench_eight_rank_form_of_the_sharded_workloads_gloo(built, workload, total, per_rank):
    """BASELINE configs[3] and [4] in their 8-RANK form (the driver's `--gpus 8`): eight processes, gloo, the same plumbing as the two-rank rehearsal --
    rank r holds shard_slice(total, r, 8) = 32768 (C4) / 4096 (C5) instances, the cost histories of all eight ranks arrive rank-major, every rank ends with
    rank 0's communicator id.  (One GPU exercises a single rank's share of these workloads; this is the slicing and gather order of all eight.)"""
    line = _bench_dry_run(workload, world=8)
    assert line["n_gpus"] == 8 and line["gather_check"] is True and line["scaling"] == "strong"
    assert line["config"]["global_batch"] == total and line["config"]["rank0_slice"] == [0, per_rank] and line["config"]["x0_shape"] == [per_rank, 5]
    assert line["exchange"] == "capi" and len(line["comm_id_sha1"]) == 40


def test_comm_unique_id_differs_between_runs(built):
    """... and a fresh id per run: two launches never share a communicator id"""
    assert _bench_dry_run("c2")["comm_id_sha1"] != _bench_dry_run("c2", fresh=True)["comm_id_sha1"]


def test_step_counts_per_call_for_the_subgoal_hook(built):
    """RobotOcpProblem.step(n) / set_subgoal() / step(n) (the RL hook, robot_ocp_problem.py:168-284): step count, goal / arena flags and peak
    control are locals of every step() call in the reference (:177-183), min_margin_traj persists.  ShimLoop.rollout on oracle-backed
    acados-shaped objects, twice on one EpisodeState with a sub-goal change in between, against OracleLoop driven the same way."""
    from oracle import oracle as orc
    from helpers import OracleAsAcados, OracleLoop, OraclePlant
    from mpc_gpu.closed_loop import EpisodeState, ShimLoop
    from mpc_gpu.world import Obstacle
    N, no, k = 10, 3, 7
    cfg = orc.config(N, no, 1.0)
    x0 = np.array([-5.0, -4.0, 0.3, 0.0, 0.0]); g1, g2 = np.array([4.0, 3.0]), np.array([-2.0, 5.0])
    obst = np.array([[0.0, -1.0, 0.4, 0.6], [2.0, 2.5, -0.5, 0.3], [-3.0, 1.0, 0.2, -0.7]])
    ocp = OracleAsAcados(orc, cfg, g1); sim = OraclePlant(orc, 0.1)
    st = EpisodeState(x=x0.copy(), goal=g1.copy(), obstacles=[Obstacle(*o, dt=0.1) for o in obst], xs=[x0.copy()])
    loop = ShimLoop(ocp, sim, N, soft=True, reset_on_failure=True)
    ref = OracleLoop(orc, cfg, x0, g1, obst, reset_on_fail=True, alias=True)
    loop.rollout(st, k)
    for _ in range(k):
        ref.step()
    assert st.steps == k == ref.steps and not st.reached
    assert np.abs(st.x - ref.x).max() < 1e-9 and abs(st.min_margin - ref.min_margin) < 1e-9
    margin_after_first = st.min_margin
    # sub-goal change, then a second step(k): k MORE control steps (the base class ran zero), counted from zero, cold-started like the first
    st.goal = g2.copy(); ocp.cost_set(N, "yref", np.array([g2[0], g2[1], 0, 0, 0]))
    ref.goal = g2.copy(); ref.x[3:] = 0.0; ref.X, ref.U = orc.initial_guess(cfg, ref.x)
    loop.rollout(st, k)
    for _ in range(k):
        ref.step()
    assert st.steps == k and st.total_steps == 2 * k == ref.steps
    assert len(st.us) == 2 * k
    assert np.abs(st.x - ref.x).max() < 1e-9
    assert st.min_margin <= margin_after_first and abs(st.min_margin - ref.min_margin) < 1e-9     # min_margin_traj persists (:228-229)
    row = st.table_row()
    assert row[4] == k and row[1] is False


def test_bench_line_is_compact_and_parseable():
    """The driver parses the LAST stdout line of bench.py and keeps only a few KB of tail (BENCH_r04.json: a 20.7 KB line -> parsed null).  compact_line() of the
    largest full record any round produced (round 4's, kept under profiles/) must stay below the limit, parse, and carry the contract's keys; the prose
    stays in the full record."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    assert len(json.dumps(full)) > 15000
    text = bench.compact_line(full)
    assert len(text) < bench.LINE_LIMIT and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["vs_baseline"] is None and line["config"]["workload"].startswith("C2") and "model" not in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "issue", "lanes_active"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert abs(line["value"] / full["value"] - 1) < 1e-4 and abs(line["roofline"]["frac"] / full["roofline"]["frac"] - 1) < 1e-4
    for name in ("c3", "c5", "c4_share", "c5_share"):
        assert set(line[name]) <= {"value", "value_one_stream", "mean_ipm_iters", "frac", "issue", "latency_frac"}
    assert not any("note" in k for k in line) and not any("note" in k for k in line["roofline"])
    # a record that cannot be made compact is refused loudly, not printed
    bloated = dict(full, config=dict(full["config"], workload="x" * 5000))
    with pytest.raises(AssertionError):
        bench.compact_line(bloated)


def test_bench_roofline_names_its_profile_and_withholds_stale_fields(capsys):
    """roofline.issue / latency_frac / wait_frac / lanes_exec / traffic / sweep_frac are replayed from the newest committed rocprofv3 PMC profile of the same
    kernel and batch, not measured by the run: the record names that profile with the launch time it was taken at (`pmc_source`), and when the
    run's own launch time is more than 5 % away from it -- the kernel changed since -- the fields are null and stderr says so.  achieved / frac / avg_launch_us are
    always the run's own."""
    import glob
    import json
    sys.path.insert(0, ROOT)
    import bench
    prof = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")) if json.load(open(f))["batch"] == 1024 and "rti_split_kernel<3, 3" in json.load(open(f))["kernel"])[-1]
    d = json.load(open(prof))
    us = d["kernel_stats"]["avg_ns"] * 1e-3

    class M:
        def kernel_name(self, batch): return d["kernel"].split("(")[0]
    class L:
        B, streams, m = 1024, 1, M()
    def run(launch_us):
        return dict(kern_ms=launch_us * 1e-3 * 100, launches=100, pair_ms=0.0, elapsed=launch_us * 1e-6 * 100 * 1.02, steps=1, mean_iters=10.9)
    fresh = bench.roofline(L(), 20, 3, run(us * 1.02))
    assert fresh["pmc_source"]["file"] == "profiles/" + os.path.basename(prof) and abs(fresh["pmc_source"]["avg_launch_us"] - us) < 1e-9 and not fresh["pmc_stale"]
    assert fresh["issue"] > 0.3 and fresh["latency_frac"] > 0.5 and fresh["traffic"] > 1e6 and fresh["lanes_exec"] > 32
    stale = bench.roofline(L(), 20, 3, run(us * 1.10))
    assert stale["pmc_stale"] and stale["pmc_source"]["stale"] and stale["pmc_source"]["file"] == fresh["pmc_source"]["file"]
    for k in ("issue", "latency_frac", "wait_frac", "lanes_exec", "traffic", "sweep_frac"):
        assert stale[k] is None, k
    assert stale["achieved"] > 0 and abs(stale["avg_launch_us"] - us * 1.10) < 1e-6 and stale["frac"] == stale["achieved"] / stale["peak"]
    assert "withheld" in capsys.readouterr().err
    # the compact line carries the source (and null fields when stale)
    line = json.loads(bench.compact_line(dict(json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json"))), roofline=stale)))
    assert line["roofline"]["issue"] is None and line["roofline"]["traffic"] is None and line["roofline"]["pmc_source"].endswith("STALE")
    line = json.loads(bench.compact_line(dict(json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json"))), roofline=fresh)))
    assert line["roofline"]["pmc_source"].startswith(os.path.basename(prof) + "@") and line["roofline"]["issue"] > 0.3
    L.streams = 2                                    # pipelined sub-batches: launches overlap, nothing to compare -- no PMC fields
    assert bench.roofline(L(), 20, 3, run(us))["issue"] is None


def test_cpu_baseline_of_the_bench_runs_and_reports_its_spread():
    """bench.py::cpu_baseline on a small budget, in a child process (it switches the oracle module to its -O3 -march=native build, which must not become this
    process's checker): thread sweep to the plateau, the plateau timed three times -- `value` is the median and lies inside `spread` --, one thread, and the
    reference's call pattern; episodes roll over (more control steps than one episode holds) without the loop state running away."""
    import json
    import subprocess
    code = (
        "import sys, json; sys.path.insert(0, %r); import bench\n"
        "from mpc_gpu.sharding import shard_slice\n"
        "x0, goal, obst, *_ = bench.make_workload('c2', 1, 0, shard_slice)\n"
        "r = bench.cpu_baseline(20, 3, x0[:16], goal[:16], obst[:16], budget_s=2.0, steady_s=1.0)\n"
        "print(json.dumps({k: r[k] for k in ('value', 'cores', 'kind', 'spread', 'one_thread', 'python_call_pattern_one_thread', 'threads', 'sample')}))\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["kind"] == "port" and r["value"] > 100 and r["one_thread"] > 100 and r["python_call_pattern_one_thread"] > 10
    assert r["spread"]["samples"] == 3 and r["spread"]["min"] <= r["value"] * 1.0001 and r["value"] <= r["spread"]["max"] * 1.0001
    assert str(r["cores"]) in r["threads"] and "x 3 (median" in r["sample"]
    steps = int(r["sample"].split(" scenarios x ")[1].split(" ")[0])
    assert steps > 100, r["sample"]           # 16 scenarios at >= 2e3 solves/s for 1 s: more than one episode of 100 control steps


def test_bench_dry_run_last_stdout_line_parses(built):
    import json
    for wl in ("c2", "c4"):
        line = _bench_dry_run(wl)
        assert len(json.dumps(line)) < 4096 and line["metric"].startswith("MPC solves/sec")


def test_bench_default_workload_is_the_same_for_every_rank_count(built):
    """The driver runs `bench.py --gpus N` for N = 1, 2, 4, 8 without a --workload and computes scaling from the values: they must be ONE workload -- the
    configuration the metric is quoted on (C2), 1024 scenarios per GPU, weak scaling -- not C2 at N = 1 and the sharded C4 beyond (rounds 1-4)."""
    import json
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["workload"].startswith("C2") and line["config"]["global_batch"] == 2048
    assert line["metric"] == "MPC solves/sec (N=20, 3 obstacles)"


def test_bench_rendezvous_guard_fires_once_and_can_be_cancelled():
    """mpc_comm_init blocks in native code until every rank has joined: bench.py arms a timer around it that ends the process non-zero instead of hanging the
    launcher.  Here with a stand-in for os._exit: an armed guard fires with code 3, a cancelled one never does."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    fired = []
    g = bench.rendezvous_guard(0.05, 1, _exit=fired.append)
    time.sleep(0.5)
    assert fired == [3] and not g.is_alive()
    g = bench.rendezvous_guard(0.2, 0, _exit=fired.append)
    g.cancel()
    time.sleep(0.5)
    assert fired == [3]
