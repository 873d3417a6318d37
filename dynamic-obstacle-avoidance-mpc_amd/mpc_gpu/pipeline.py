"""PipelinedMpc: one batch of independent MPC instances cut into K contiguous sub-batches, each with its own libmpcgpu handle and HIP stream.

Why: one wavefront per SIMD gives the chip 1024 wavefront slots, and interior-point iteration counts are heavy-tailed (mean ~14 at N = 50 / 10 obstacles, cap 50).
A launch of a few thousand instances ends when its LAST wavefront ends, and a 50-iteration solve that the instance scheduling did not predict starts in the last
round of wavefronts: measured, a launch of 4096 such instances takes 1.34x its mean slot load (profiles/r04_share_tail_probe_4096.json).  Instances share nothing
(SURVEY.md 8(e)), so control step k + 1 of a sub-batch depends only on step k of the SAME sub-batch: with the sub-batches on separate streams the tail of one
launch overlaps with the body of the next launch of the other (the include/mpc_gpu.h threading contract: one handle per (device, stream)).  Measured with K = 2:
4096 x (N = 50, 10 obstacles) 1.47 -> 1.86e6 solves/s (+27 %), 32768 x (N = 20, 3 obstacles) 22.1 -> 23.4e6 (+6 %), 65536: +2.4 %; K = 4 loses (sub-batches of one
round of wavefronts are no longer reordered, and the dispatcher's crossovers assume the whole chip): profiles/r04_streams_probe_*.json.

Every instance's arithmetic is its own: results are those of one BatchedMpc on the whole batch, bit for bit with one instance per wavefront and to the rounding of
the wavefront sums with three per wavefront (the neighbours in a wavefront change, as under any permutation of the batch).

Device-pointer API only (torch tensors): the arrays are the caller's whole-batch arrays, every call slices them per sub-batch.  Calls return without joining the
streams; `join()` makes the caller's current stream wait for all sub-batches (before it reads whole-batch results), `fork()` makes the sub-batch streams wait for
the caller's current stream (after it has written whole-batch inputs)."""
import torch

from . import _lib
from .sharding import shard_slice
from .solver import BatchedMpc


class PipelinedMpc:
    def __init__(self, N=20, n_obst=3, Tf=2.0, max_batch=1, device=0, streams=2, **cfg_overrides):
        self.N, self.n_obst, self.Tf, self.max_batch, self.device = int(N), int(n_obst), float(Tf), int(max_batch), int(device)
        self.K = max(1, min(int(streams), self.max_batch))
        dev = torch.device("cuda", self.device)
        self.parts = []
        for k in range(self.K):
            lo, hi = shard_slice(self.max_batch, k, self.K)
            self.parts.append((lo, hi, BatchedMpc(N, n_obst, Tf, max_batch=hi - lo, device=device, **cfg_overrides), torch.cuda.Stream(device=dev)))
        self.cfg = self.parts[0][2].cfg

    # ------------------------------------------------------------------ lifetime
    def close(self):
        """Waits for every sub-batch stream before the handles go: mpc_destroy synchronises only the handle's OWN stream, and the launches were enqueued on
        these torch streams -- a close under in-flight kernels would free the handle's device buffers (instance order, iteration schedule) beneath them.
        The caller's whole-batch tensors must outlive join() (they are used on the sub-batch streams; no record_stream is taken on the slices)."""
        for _, _, m, s in self.parts:
            s.synchronize()
            m.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ stream plumbing
    def fork(self):
        """the sub-batch streams wait for everything enqueued so far on the caller's current stream"""
        cur = torch.cuda.current_stream()
        for _, _, _, s in self.parts:
            s.wait_stream(cur)

    def join(self):
        """the caller's current stream waits for everything enqueued so far on the sub-batch streams"""
        cur = torch.cuda.current_stream()
        for _, _, _, s in self.parts:
            cur.wait_stream(s)

    def _check(self, batch):
        if batch != self.max_batch:
            raise ValueError(f"PipelinedMpc is built for batches of exactly {self.max_batch} instances (got {batch}): the sub-batch slices are fixed")

    @staticmethod
    def _sl(t, lo, hi):
        return None if t is None else t[lo:hi]

    # ------------------------------------------------------------------ device-pointer API, per sub-batch on its own stream
    def reset_guess_dev(self, batch, x0, X, U):
        self._check(batch)
        for lo, hi, m, s in self.parts:
            m.reset_guess_dev(hi - lo, x0[lo:hi], X[lo:hi], U[lo:hi], stream=s.cuda_stream)

    def solve_dev(self, batch, x0, P, goal, X, U, u0=None, cost=None, status=None, iters=None):
        self._check(batch)
        for lo, hi, m, s in self.parts:
            m.solve_dev(hi - lo, x0[lo:hi], P[lo:hi], goal[lo:hi], X[lo:hi], U[lo:hi], self._sl(u0, lo, hi), self._sl(cost, lo, hi), self._sl(status, lo, hi),
                        self._sl(iters, lo, hi), stream=s.cuda_stream)

    def closed_loop_step_dev(self, batch, x0, obst, goal, X, U, u0=None, cost=None, status=None, iters=None, noise=None, randomness=0.1, vmax=2.0,
                             flags=_lib.STEP_SHIFT | _lib.STEP_PLANT | _lib.STEP_OBSTACLES, min_margin=None, ep_flags=None, ep_steps=None):
        """one control step of every sub-batch, each ONE launch on its own stream (mpc_closed_loop_step_dev)"""
        self._check(batch)
        for lo, hi, m, s in self.parts:
            m.closed_loop_step_dev(hi - lo, x0[lo:hi], obst[lo:hi], goal[lo:hi], X[lo:hi], U[lo:hi], self._sl(u0, lo, hi), self._sl(cost, lo, hi),
                                   self._sl(status, lo, hi), self._sl(iters, lo, hi), self._sl(noise, lo, hi), randomness, vmax, flags,
                                   self._sl(min_margin, lo, hi), self._sl(ep_flags, lo, hi), self._sl(ep_steps, lo, hi), stream=s.cuda_stream)

    # ------------------------------------------------------------------ the cost exchange lives on the first sub-batch's handle (include/mpc_gpu.h mpc_comm_*)
    def comm_init(self, rank, world, unique_id):
        self.parts[0][2].comm_init(rank, world, unique_id)

    def comm_world(self):
        return self.parts[0][2].comm_world()

    def comm_destroy(self):
        self.parts[0][2].comm_destroy()

    def comm_library_path(self):
        return self.parts[0][2].comm_library_path()

    def allgather_cost_dev(self, count, cost, cost_all, stream=None):
        """whole-batch costs (the caller joins the sub-batch streams first: join())"""
        self.parts[0][2].allgather_cost_dev(count, cost, cost_all, stream=stream)

    # ------------------------------------------------------------------ measurement / introspection (summed or taken from the first sub-batch)
    def set_accumulators(self, iters_acc=None, status_acc=None):
        for lo, hi, m, _ in self.parts:
            m.set_accumulators(self._sl(iters_acc, lo, hi), self._sl(status_acc, lo, hi))

    def profile_enable(self, on=True, every=1):
        for _, _, m, _ in self.parts:
            m.profile_enable(on, every=every)

    def profile_read(self):
        ms = n = 0
        for _, _, m, _ in self.parts:
            a, b = m.profile_read(); ms += a; n += b
        return ms, n

    def kernel_name(self, batch=None, lookahead=True):
        lo, hi, m, _ = self.parts[0]
        return m.kernel_name(hi - lo, lookahead)

    def lanes_per_instance(self, batch=None):
        lo, hi, m, _ = self.parts[0]
        return m.lanes_per_instance(hi - lo)

    def lanes_per_stage(self, batch=None):
        lo, hi, m, _ = self.parts[0]
        return m.lanes_per_stage(hi - lo)

    def waves_per_simd(self, batch=None):
        lo, hi, m, _ = self.parts[0]
        return m.waves_per_simd(hi - lo)
