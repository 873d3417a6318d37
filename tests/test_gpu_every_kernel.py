"""EVERY kernel instantiation the dispatcher can reach x EVERY output pointer of the fused control step, inside guard bands.

Why this file exists: `rti_solve_kernel<3, 32, 2, false>` of round 3's first builds added the interior-point iteration count to the episode's
step counter.  Its source is right; in its listing the register allocator re-materialises the kernel-argument load of (ep_min_margin .. trace) into s[8:15] AFTER
`s_load_dwordx16 s[12:27]` has delivered (iters_acc, status_acc, ...), and then spills the clobbered s[12:13] as `iters_acc`: the accumulator pointer silently
became the step counter's.  No parity test saw it (accumulators and episode counters are optional outputs, and the workload kernels were not affected); the
closed-loop fuzz did.  So: for every instantiation, one fused step with ALL optional outputs wired to distinct guard-banded arrays, the result compared with the
default mapping's (statuses, iteration counts, flags, step counters exactly; controls, costs, margins to 1e-6), accumulators against what they accumulate, bands intact."""
import itertools

import numpy as np
import pytest

from helpers import random_batch

pytestmark = pytest.mark.gpu
GUARD = 64           # sentinel words on either side of every array


def _configs(mpc_gpu):
    """(N, n_obst, overrides) for every distinct kernel name the dispatcher reports"""
    seen, out = set(), []
    for N, no in itertools.product((10, 20, 31, 40), (3, 5, 10, 2, 4, 7)):
        for lanes, lps, waves, rowpar, mfma, blk2 in itertools.product((0, 16, 21, 32, 64), (0, 1, 2, 3), (0, 1, 2), (1, 0), (0, 1), (0, 1)):
            if mfma and (lanes != 64 or lps != 1 or not rowpar or blk2 or waves):
                continue
            if not rowpar and (lps != 1 or blk2 or waves):
                continue
            if blk2 and (lps == 1 or lanes or waves == 2):
                continue
            with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=4) as s:
                try:
                    s.set_lanes_per_stage(lps); s.set_waves_per_simd(waves)
                    if lanes:
                        s.set_lanes_per_instance(lanes)
                    s.set_row_parallel(bool(rowpar)); s.set_block_riccati(bool(blk2))
                    if mfma:
                        s.set_matrix_cores(True)
                    name = s.kernel_name(4)
                except mpc_gpu.MpcError:
                    continue
            if name not in seen:
                seen.add(name); out.append((N, no, dict(lanes=lanes, lps=lps, waves=waves, rowpar=rowpar, mfma=mfma, blk2=blk2), name))
    return out


class Banded:
    """device arrays carved out of one sentinel-filled buffer"""

    def __init__(self, torch, dev):
        self.torch, self.dev, self.items = torch, dev, []

    def f64(self, *shape, init=0.0):
        t = self.torch.full((int(np.prod(shape)) + 2 * GUARD,), -7.25e77, dtype=self.torch.float64, device=self.dev)
        v = t[GUARD:-GUARD].view(*shape); v.fill_(init); self.items.append((t, "f64")); return v

    def i32(self, *shape, init=0):
        t = self.torch.full((int(np.prod(shape)) + 2 * GUARD,), -1234567, dtype=self.torch.int32, device=self.dev)
        v = t[GUARD:-GUARD].view(*shape); v.fill_(init); self.items.append((t, "i32")); return v

    def intact(self):
        for t, kind in self.items:
            s = -7.25e77 if kind == "f64" else -1234567
            if not (bool((t[:GUARD] == s).all()) and bool((t[-GUARD:] == s).all())):
                return False
        return True


def _one_step(mpc_gpu, torch, N, no, ov, B, x0, goal, obst, noise):
    dev = torch.device("cuda:0")
    from mpc_gpu import _lib
    with mpc_gpu.BatchedMpc(N, no, 0.1 * N, max_batch=B) as s, torch.cuda.stream(torch.cuda.Stream(device=dev)):
        q = torch.cuda.current_stream().cuda_stream
        if ov is not None:
            s.set_lanes_per_stage(ov["lps"]); s.set_waves_per_simd(ov["waves"])
            if ov["lanes"]:
                s.set_lanes_per_instance(ov["lanes"])
            s.set_row_parallel(bool(ov["rowpar"])); s.set_block_riccati(bool(ov["blk2"]))
            if ov["mfma"]:
                s.set_matrix_cores(True)
        name = s.kernel_name(B)
        bd = Banded(torch, dev)
        t = lambda a, mk: mk(*a.shape).copy_(torch.from_numpy(np.ascontiguousarray(a)).to(dev))
        dx0, dg, do, dn = t(x0, bd.f64), t(goal, bd.f64), t(obst, bd.f64), t(noise, bd.f64)
        X, U = bd.f64(B, N + 1, 5), bd.f64(B, N, 2)
        u0, cost, margin = bd.f64(B, 2, init=-5.0), bd.f64(B, init=-5.0), bd.f64(B, init=float("inf"))
        status, iters, flags, steps = bd.i32(B, init=-9), bd.i32(B, init=-9), bd.i32(B), bd.i32(B, init=100)
        iacc, sacc = bd.i32(B, init=1000), bd.i32(B, init=0)
        s.set_accumulators(iacc, sacc)
        s.reset_guess_dev(B, dx0, X, U, stream=q)
        fl = _lib.STEP_SHIFT | _lib.STEP_PLANT | _lib.STEP_OBSTACLES | _lib.STEP_METRICS | _lib.STEP_RESET_ON_FAIL
        s.closed_loop_step_dev(B, dx0, do, dg, X, U, u0, cost, status, iters, dn, flags=fl, min_margin=margin, ep_flags=flags, ep_steps=steps, stream=q)
        torch.cuda.current_stream().synchronize()
        c = lambda a: a.cpu().numpy().copy()
        r = dict(name=name, x0=c(dx0), obst=c(do), X=c(X), U=c(U), u0=c(u0), cost=c(cost), margin=c(margin), status=c(status), iters=c(iters), flags=c(flags),
                 steps=c(steps), iacc=c(iacc), sacc=c(sacc), intact=bd.intact(), goal_unchanged=bool((dg.cpu().numpy() == goal).all()),
                 noise_unchanged=bool((dn.cpu().numpy() == noise).all()))
        s.set_accumulators(None, None)
    return r


def test_every_instantiation_writes_every_output_where_it_belongs(built):
    import torch
    import mpc_gpu
    cfgs = _configs(mpc_gpu)
    names = [c[3] for c in cfgs]
    assert len(cfgs) >= 40, names
    B = 37
    ref = {}
    bad, ran, refused = [], [], []
    for N, no, ov, name in cfgs:
        x0, goal, obst = random_batch(B, no, seed=100 + N + no)
        x0[0, :2] = goal[0] + 0.05                                  # instance 0 reaches its goal in this step: flag 1, step counter NOT advanced
        noise = np.random.default_rng(N * 31 + no).standard_normal((B, no, 2))
        if (N, no) not in ref:
            ref[(N, no)] = _one_step(mpc_gpu, torch, N, no, None, B, x0, goal, obst, noise)
        try:
            a, r = ref[(N, no)], _one_step(mpc_gpu, torch, N, no, ov, B, x0, goal, obst, noise)
        except mpc_gpu.MpcError as e:
            assert "no kernel variant" in str(e), (name, str(e))      # a combination of overrides the dispatcher refuses (it names a kernel that is not instantiated)
            refused.append(name)
            continue
        ran.append(r["name"])
        why = []
        if not (r["intact"] and r["goal_unchanged"] and r["noise_unchanged"]): why.append("a guard band or an input array was written")
        if not np.array_equal(r["status"], a["status"]): why.append("status")
        if not np.isin(r["status"], (0, 2, 4)).all(): why.append("status values")
        if np.abs(r["iters"] - a["iters"]).max() > 2: why.append(f"iters {r['iters'].tolist()} vs {a['iters'].tolist()}")
        if not np.array_equal(r["flags"], a["flags"]): why.append("episode flags")
        if not np.array_equal(r["steps"], 100 + ((r["flags"] & 1) == 0)): why.append(f"step counters {r['steps'].tolist()}")
        if not np.array_equal(r["iacc"], 1000 + r["iters"]): why.append(f"iteration accumulator {(r['iacc'] - 1000).tolist()} vs {r['iters'].tolist()}")
        if not np.array_equal(r["sacc"], (r["status"] == 4) + 65536 * (r["status"] == 2)): why.append("status accumulator")
        if not np.array_equal(r["obst"], a["obst"]): why.append("obstacle motion")
        ok = r["status"] == 0
        for key, tol in (("u0", 1e-6 * 8), ("x0", 1e-6), ("margin", 1e-6), ("X", 1e-5), ("U", 1e-5 * 8)):
            if np.abs(r[key][ok] - a[key][ok]).max() > tol: why.append(f"{key} differs by {np.abs(r[key][ok] - a[key][ok]).max():.2e}")
        if np.abs(r["cost"][ok] - a["cost"][ok]).max() > 1e-6 * max(1.0, np.abs(a["cost"][ok]).max()): why.append("cost")
        if (r["flags"][0] & 1) != 1: why.append("instance 0 did not reach its goal")
        if why:
            bad.append((name, why))
    assert not bad, bad
    assert len(set(ran)) >= 40 and any("rti_solve_kernel<3, 32, 2, false>" in n for n in ran), (sorted(set(ran)), refused)
