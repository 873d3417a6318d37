// aux_kernels.hpp -- the small kernels either side of the solve: obstacle look-ahead (a9), warm-start shift (a12),
// initial guess (a13), plant step (a14), ground-truth obstacle motion and a dense dump of the linearisation for tests.
// File:line citations are relative to the reference repository root.
#pragma once
#include "rti_kernel.hpp"

namespace mpc {

// Obstacle.predict_trajectory (visualization.py:62-79) for every obstacle of every instance, written straight into the
// parameter tensor P[B][N+1][n_obst][2] (parameterize_model, robot_ocp_problem.py:154-166).  One thread per obstacle.
__global__ void predict_kernel(World w, int count, int n_obst, int N, double dt, const double *__restrict__ obst, double *__restrict__ P)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int inst = t / n_obst, j = t - inst * n_obst;
    const double *o = obst + (size_t)t * 4;
    double x = o[0], y = o[1], vy = o[3];
    double vx = w.bug_compat_predict ? o[3] : o[2];   // visualization.py:69 (reference defect D1)
    double *Pi = P + ((size_t)inst * (N + 1) * n_obst + j) * 2;
    Pi[0] = x; Pi[1] = y;
    for (int i = 1; i <= N; i++) {
        obstacle_advance(w, dt, x, vx, y, vy);
        Pi[(size_t)i * n_obst * 2] = x; Pi[(size_t)i * n_obst * 2 + 1] = y;
    }
}

// Obstacle.step(): ground-truth motion with optional multiplicative velocity noise, visualization.py:20-33.
__global__ void obstacle_step_kernel(World w, int count, double dt, double *__restrict__ obst, const double *__restrict__ noise,
                                     double randomness, double vmax)
{
#pragma clang fp contract(off)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    double *o = obst + (size_t)t * 4;
    double x = o[0], y = o[1], vx = o[2], vy = o[3];
    if (noise) obstacle_noise(randomness, vmax, noise[(size_t)t * 2], noise[(size_t)t * 2 + 1], vx, vy);
    obstacle_advance(w, dt, x, vx, y, vy);
    o[0] = x; o[1] = y; o[2] = vx; o[3] = vy;
}

// generate_random_moving_obstacles (obstacle_generator.py:8-28) for `count` seeds: thread s reproduces what the reference draws
// after np.random.seed(seed0 + s) -- numpy's legacy MT19937 (init_genrand seeding, standard tempering), random_sample() =
// ((a >> 5) * 2^26 + (b >> 6)) / 2^53 from two consecutive outputs, uniform(lo, hi) = lo + (hi - lo) * u (no contraction), drawn in
// the reference's order: x block, y block (RANDOM only), vx block, vy block.  CENTER / EDGE place every obstacle at 0 / `edge`.
// At most 8 * n_obst <= 80 outputs are needed, all from the first state regeneration, which touches state words < 80 + 398.
enum { kScenarioRandom = 0, kScenarioCenter = 1, kScenarioEdge = 2 };
__global__ void scenario_kernel(int count, int n_obst, int scenario, unsigned seed0, double x_lo, double x_hi, double y_lo, double y_hi,
                                double v_max, double edge, double *__restrict__ out)
{
#pragma clang fp contract(off)
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= count) return;
    constexpr int kWords = 80 + 398;
    unsigned mt[kWords];
    mt[0] = seed0 + (unsigned)s;
    for (int k = 1; k < kWords; k++) mt[k] = 1812433253u * (mt[k - 1] ^ (mt[k - 1] >> 30)) + (unsigned)k;
    int pos = 0;
    auto next32 = [&]() {
        const unsigned y0 = (mt[pos] & 0x80000000u) | (mt[pos + 1] & 0x7fffffffu);
        unsigned y = mt[pos + 397] ^ (y0 >> 1) ^ ((y0 & 1u) ? 0x9908b0dfu : 0u);
        pos++;
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    };
    auto uniform = [&](double lo, double hi) {
        const unsigned a = next32() >> 5, b = next32() >> 6;
        const double u = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
        return lo + (hi - lo) * u;
    };
    double *o = out + (size_t)s * n_obst * 4;
    for (int j = 0; j < n_obst; j++) o[j * 4 + 0] = scenario == kScenarioRandom ? uniform(x_lo, x_hi) : (scenario == kScenarioEdge ? edge : 0.0);
    for (int j = 0; j < n_obst; j++) o[j * 4 + 1] = scenario == kScenarioRandom ? uniform(y_lo, y_hi) : (scenario == kScenarioEdge ? edge : 0.0);
    for (int j = 0; j < n_obst; j++) o[j * 4 + 2] = uniform(-v_max, v_max);
    for (int j = 0; j < n_obst; j++) o[j * 4 + 3] = uniform(-v_max, v_max);
}

// ------------------------------------------------------------------------------------------------------------------
// THE REFERENCE'S NOISE STREAM ON THE DEVICE.  experiments.py:33-36 seeds numpy's global legacy generator per run (np.random.seed(i)), draws the
// scenario (obstacle_generator.py: uniform blocks) and then, in every control step, `for o in self.obstacles: o.step()` draws np.random.normal(size=2)
// per obstacle (visualization.py:31).  Instance s of a batch carries that generator for seed seed0 + s: MT19937 state (624 words), position, and the
// cached second value of the polar Gaussian method -- numpy's legacy_gauss:
//     if cached: return it;  else  do { x1 = 2 u - 1; x2 = 2 u - 1; r2 = x1 x1 + x2 x2 } while (r2 >= 1 || r2 == 0);  f = sqrt(-2 log(r2) / r2);
//     cache f x1, return f x2                       (u = random_sample() = ((a >> 5) 2^26 + (b >> 6)) / 2^53 from two outputs)
// Everything but the logarithm is IEEE arithmetic in numpy's order (no contraction) and therefore bit-exact.  The logarithm is evaluated to ~2^-80 in
// double-double arithmetic and rounded once, i.e. correctly rounded for all practical purposes; glibc's log (what numpy calls) is not quite -- it
// differs from the correctly rounded value on a few inputs in 10^5 -- so the stream equals the host's in all but that fraction of the draws, where
// one value differs in its last bit (measured: tests/test_gpu_aux.py).  The device library's own log differs from glibc's on 2.8 % of the inputs.
// State layout per instance (uint32 words): mt[624] | pos | has_gauss | gauss (2 words).
constexpr int kNoiseStateWords = 628;

__device__ __forceinline__ void dd_two_sum(double a, double b, double &s, double &e)
{
#pragma clang fp contract(off)
    s = a + b; const double bb = s - a; e = (a - (s - bb)) + (b - bb);
}
__device__ __forceinline__ void dd_fast_two_sum(double a, double b, double &s, double &e)       // |a| >= |b|
{
#pragma clang fp contract(off)
    s = a + b; e = b - (s - a);
}
// (ah, al) * (bh, bl) -> (ph, pl), error ~2^-104
__device__ __forceinline__ void dd_mul(double ah, double al, double bh, double bl, double &ph, double &pl)
{
    ph = ah * bh;
    pl = fma(ah, bh, -ph) + (ah * bl + al * bh);
    double s, e; dd_fast_two_sum(ph, pl, s, e); ph = s; pl = e;
}
__device__ __forceinline__ void dd_add(double ah, double al, double bh, double bl, double &sh, double &sl)
{
    double s, e; dd_two_sum(ah, bh, s, e);
    e += al + bl;
    dd_fast_two_sum(s, e, sh, sl);
}
// log(x) for a normal positive double, rounded once from a double-double value
__device__ inline double log_dd(double x)
{
    static constexpr double kLogTab[25][2] = {{-0x1.7fafa3bd8151cp-2, 0x1.219024acd3b77p-58}, {-0x1.522ae0738a3d8p-2, 0x1.8f7e9b38a6979p-57}, {-0x1.269621134db92p-2, -0x1.e0efadd9db02bp-56}, {-0x1.f991c6cb3b379p-3, -0x1.f665066f980a2p-57}, {-0x1.a93ed3c8ad9e3p-3, -0x1.bcafa9de97203p-57}, {-0x1.5bf406b543db2p-3, 0x1.1f5b44c0df7e7p-61}, {-0x1.1178e8227e47cp-3, 0x1.0e63a5f01c691p-58}, {-0x1.9335e5d594989p-4, 0x1.478a85704ccb7p-58}, {-0x1.08598b59e3a07p-4, 0x1.dd7009902bf32p-58}, {-0x1.0415d89e74444p-5, -0x1.c05cf1d753622p-59}, {0x0.0p+0, 0x0.0p+0}, {0x1.f829b0e783300p-6, 0x1.33e3f04f1ef23p-60}, {0x1.f0a30c01162a6p-5, 0x1.85f325c5bbacdp-59}, {0x1.6f0d28ae56b4cp-4, -0x1.906d99184b992p-58}, {0x1.e27076e2af2e6p-4, -0x1.61578001e0162p-60}, {0x1.29552f81ff523p-3, 0x1.301771c407dbfp-57}, {0x1.5ff3070a793d4p-3, -0x1.bc60efafc6f6ep-58}, {0x1.9525a9cf456b4p-3, 0x1.d904c1d4e2e26p-57}, {0x1.c8ff7c79a9a22p-3, -0x1.4f689f8434012p-57}, {0x1.fb9186d5e3e2bp-3, -0x1.caaae64f21acbp-57}, {0x1.1675cababa60ep-2, 0x1.ce63eab883717p-61}, {0x1.2e8e2bae11d31p-2, -0x1.8f4cdb95ebdf9p-56}, {0x1.4618bc21c5ec2p-2, 0x1.f42decdeccf1dp-56}, {0x1.5d1bdbf5809cap-2, 0x1.4236383dc7fe1p-56}, {0x1.739d7f6bbd007p-2, -0x1.8c76ceb014b04p-56}};
    int e;
    double m = frexp(x, &e);                        // m in [0.5, 1)
    if (m < 0.70710678118654752) { m *= 2.0; e -= 1; }      // m in [sqrt(1/2), sqrt(2)): log x = e ln 2 + log m without cancellation near x = 1
    const int i = (int)rint(m * 32.0);              // 22 .. 46; c = i / 32 exactly, |m - c| <= 1 / 64
    const double c = (double)i * 0.03125;
    // t = (m - c) / (m + c) in double-double: the numerator is exact (m and c within a factor of two: Sterbenz)
    const double num = m - c;
    double dh, dl; dd_two_sum(m, c, dh, dl);
    const double th = num / dh;
    const double tl = (fma(-th, dh, num) - th * dl) / dh;
    // log m = log c + 2 atanh t = log c + 2 (t + t^3 / 3 + t^5 / 5 + ...): t and t^3 / 3 in double-double, the rest (<= 3e-9 t) in double
    double t2h, t2l; dd_mul(th, tl, th, tl, t2h, t2l);
    double t3h, t3l; dd_mul(t2h, t2l, th, tl, t3h, t3l);
    double ch, cl; dd_mul(t3h, t3l, 0x1.5555555555555p-2, 0x1.5555555555555p-56, ch, cl);
    const double w = t2h;
    const double tail = th * (w * w) * (1.0 / 5.0 + w * (1.0 / 7.0 + w * (1.0 / 9.0 + w * (1.0 / 11.0 + w * (1.0 / 13.0)))));
    double ah, al; dd_add(th, tl, ch, cl, ah, al);
    al += tail;
    ah *= 2.0; al *= 2.0;
    double rh, rl; dd_add(kLogTab[i - 22][0], kLogTab[i - 22][1], ah, al, rh, rl);
    // e ln 2: the high part of ln 2 has 11 trailing zero bits, so e * ln2_hi is exact
    const double ed = (double)e;
    double eh = ed * 0x1.62e42fefa3800p-1, el = ed * 0x1.ef35793c76730p-45 + ed * 0x1.f97b57a079a19p-103;
    double sh, sl; dd_add(eh, el, rh, rl, sh, sl);
    return sh + sl;
}

struct NoiseGen {
    unsigned *st;      // this instance's state words
    __device__ __forceinline__ unsigned next32()
    {
        int pos = (int)st[624];
        if (pos >= 624) {       // regenerate (numpy's rk_random / init by genrand: the standard twist)
            for (int k = 0; k < 624; k++) {
                const unsigned y0 = (st[k] & 0x80000000u) | (st[(k + 1) % 624] & 0x7fffffffu);
                st[k] = st[(k + 397) % 624] ^ (y0 >> 1) ^ ((y0 & 1u) ? 0x9908b0dfu : 0u);
            }
            pos = 0;
        }
        unsigned y = st[pos];
        st[624] = (unsigned)(pos + 1);
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    __device__ __forceinline__ double next_double()
    {
#pragma clang fp contract(off)
        const unsigned a = next32() >> 5, b = next32() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    __device__ __forceinline__ double gauss()
    {
#pragma clang fp contract(off)
        if (st[625]) {
            st[625] = 0u;
            const double g = __hiloint2double((int)st[627], (int)st[626]);
            st[626] = 0u; st[627] = 0u;
            return g;
        }
        double x1, x2, r2;
        do {
            x1 = 2.0 * next_double() - 1.0;
            x2 = 2.0 * next_double() - 1.0;
            const double a = x1 * x1, b = x2 * x2;
            r2 = a + b;
        } while (r2 >= 1.0 || r2 == 0.0);
        const double lg = log_dd(r2);
        const double q = -2.0 * lg;
        const double f = sqrt(q / r2);
        const double g = f * x1;
        st[625] = 1u; st[626] = (unsigned)__double2loint(g); st[627] = (unsigned)__double2hiint(g);
        return f * x2;
    }
};

// np.random.seed(seed0 + s) followed by the scenario generator's uniform draws (their VALUES come from scenario_kernel; here they are only consumed):
// 4 n_obst doubles for RANDOM (x, y, vx, vy blocks), 2 n_obst otherwise
__global__ void noise_init_kernel(int count, int n_obst, int scenario, unsigned seed0, unsigned *__restrict__ state)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= count) return;
    unsigned *st = state + (size_t)s * kNoiseStateWords;
    st[0] = seed0 + (unsigned)s;
    for (int k = 1; k < 624; k++) st[k] = 1812433253u * (st[k - 1] ^ (st[k - 1] >> 30)) + (unsigned)k;
    st[624] = 624u; st[625] = 0u; st[626] = 0u; st[627] = 0u;
    NoiseGen g{st};
    const int draws = (scenario == kScenarioRandom ? 4 : 2) * n_obst;
    for (int k = 0; k < draws; k++) (void)g.next_double();
}

// one control step's np.random.normal(size=2) per obstacle, in the reference's order (obstacle 0 first): noise[s][j][0..1]
__global__ void noise_draw_kernel(int count, int n_obst, unsigned *__restrict__ state, double *__restrict__ noise, const int32_t *__restrict__ ep_flags)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= count) return;
    if (ep_flags && (ep_flags[s] & 1)) return;      // the episode is over: the reference's loop has left (robot_ocp_problem.py:247-250), nothing is drawn any more
    NoiseGen g{state + (size_t)s * kNoiseStateWords};
    double *o = noise + (size_t)s * n_obst * 2;
    for (int j = 0; j < n_obst; j++) { o[2 * j] = g.gauss(); o[2 * j + 1] = g.gauss(); }
}

// INSTANCE SCHEDULING.  Where several instances share a wavefront (one lane per stage: 2, 3 or 4 of them) the wavefront runs until its
// slowest instance has converged, and interior-point iteration counts are heavy-tailed: on the randomized C3 workload the mean is 6.6 but the
// mean of the per-wavefront maximum is 8.2 with two and 9.4 with three instances per wavefront.  Iteration counts of consecutive control
// steps of one instance are strongly correlated, so dealing the instances to wavefronts IN THE ORDER OF THEIR PREVIOUS COUNT brings the
// per-wavefront maximum down to 7.1 / 7.4 (scripts/iters_order_probe.py).  These kernels build that order: a stable counting sort of the
// instances by their last iteration count, descending (the longest-running wavefronts are dispatched first): deterministic, O(batch),
// ~10 us at 65536.
// order[] is a permutation of 0..batch-1; the solve kernel's slot s processes instance order[s].  Results are those of the natural
// order (bit for bit with 2 or 4 instances per wavefront; to the rounding of the wavefront sums with 3).
// Two launches: schedule_count_kernel (one histogram per block of kSchedChunk consecutive instances) and schedule_scatter_kernel (every block
// derives its global base per bin from all histograms, ranks its own instances stably and writes their slots).
constexpr int kSchedThreads = 256, kSchedBins = 32, kSchedPer = 4, kSchedChunk = kSchedThreads * kSchedPer;   // counts >= 31 share the first bin
__device__ __forceinline__ int sched_key(int it) { return (kSchedBins - 1) - (it < 0 ? 0 : (it > kSchedBins - 1 ? kSchedBins - 1 : it)); }   // descending counts

__global__ __launch_bounds__(kSchedThreads) void schedule_count_kernel(int batch, int nblk, const int32_t *__restrict__ iters, unsigned *__restrict__ ghist)
{
    __shared__ unsigned hist[kSchedBins];
    const int t = threadIdx.x, b = blockIdx.x;
    if (t < kSchedBins) hist[t] = 0u;
    __syncthreads();
    const int e0 = b * kSchedChunk + t * kSchedPer;
#pragma unroll
    for (int u = 0; u < kSchedPer; u++)
        if (e0 + u < batch) __hip_atomic_fetch_add(&hist[sched_key(iters[e0 + u])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if (t < kSchedBins) ghist[t * nblk + b] = hist[t];          // bin-major: the scatter pass reads a bin's blocks consecutively
}

__global__ __launch_bounds__(kSchedThreads) void schedule_scatter_kernel(int batch, int nblk, const int32_t *__restrict__ iters,
                                                                         const unsigned *__restrict__ ghist, int32_t *__restrict__ order)
{
    __shared__ unsigned cnt[kSchedBins * kSchedThreads];      // [bin][thread]: counts, then exclusive offsets in (bin, thread) order
    __shared__ unsigned part[kSchedThreads], tot[kSchedBins], pre[kSchedBins], base[kSchedBins], row0[kSchedBins];
    const int t = threadIdx.x, b = blockIdx.x;
    for (int f = t; f < kSchedBins * kSchedThreads; f += kSchedThreads) cnt[f] = 0u;
    if (t < kSchedBins) { tot[t] = 0u; pre[t] = 0u; }
    __syncthreads();
    // (a) where this block's instances of every bin start: everything in earlier bins, plus the same bin in earlier blocks
    for (int f = t; f < kSchedBins * nblk; f += kSchedThreads) {
        const int bin = f / nblk, blk = f - bin * nblk;
        const unsigned v = ghist[f];
        if (v) {
            __hip_atomic_fetch_add(&tot[bin], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (blk < b) __hip_atomic_fetch_add(&pre[bin], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    // (b) this block's own instances: thread t owns kSchedPer consecutive ones, counted in column t
    const int e0 = b * kSchedChunk + t * kSchedPer;
    int key[kSchedPer];
#pragma unroll
    for (int u = 0; u < kSchedPer; u++) {
        key[u] = (e0 + u < batch) ? sched_key(iters[e0 + u]) : -1;
        if (key[u] >= 0) cnt[key[u] * kSchedThreads + t] += 1u;
    }
    __syncthreads();
    if (t == 0) { unsigned run = 0u; for (int k = 0; k < kSchedBins; k++) { base[k] = run + pre[k]; run += tot[k]; } }
    // exclusive scan of cnt in (bin, thread) order: thread t owns the kSchedBins consecutive entries [t * kSchedBins, (t + 1) * kSchedBins)
    unsigned sum = 0u;
    for (int f = 0; f < kSchedBins; f++) sum += cnt[t * kSchedBins + f];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < kSchedThreads; d <<= 1) {
        const unsigned v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned run = part[t] - sum;
    for (int f = 0; f < kSchedBins; f++) { const unsigned c = cnt[t * kSchedBins + f]; cnt[t * kSchedBins + f] = run; run += c; }
    __syncthreads();
    if (t < kSchedBins) row0[t] = cnt[t * kSchedThreads];     // offset of the bin's first instance inside this block
    __syncthreads();
    // (c) slots: bins in descending-count order, inside a bin ascending instance index (blocks, threads, a thread's own instances in order)
#pragma unroll
    for (int u = 0; u < kSchedPer; u++) {
        if (key[u] >= 0) {
            const int k = key[u] * kSchedThreads + t;
            order[base[key[u]] + (cnt[k] - row0[key[u]])] = e0 + u;
            cnt[k] += 1u;
        }
    }
}

// Warm-start shift, robot_ocp_problem.py:253-258: X[j] <- X[j+1] (j < N), U[j] <- U[j+1] (j < N-1), U[N-1] <- 0.
// One wavefront per instance; every lane reads its successor stage before anyone writes.
__global__ __launch_bounds__(64) void shift_kernel(int batch, int N, double *__restrict__ X, double *__restrict__ U)
{
    const int inst = blockIdx.x;
    if (inst >= batch) return;
    const int i = threadIdx.x;
    double *Xg = X + (size_t)inst * (N + 1) * 5, *Ug = U + (size_t)inst * N * 2;
    double xv[5] = {0, 0, 0, 0, 0}, uv[2] = {0, 0};
    if (i < N) {
#pragma unroll
        for (int c = 0; c < 5; c++) xv[c] = Xg[(i + 1) * 5 + c];
    }
    if (i < N - 1) { uv[0] = Ug[(i + 1) * 2]; uv[1] = Ug[(i + 1) * 2 + 1]; }
    __syncthreads();
    if (i < N) {
#pragma unroll
        for (int c = 0; c < 5; c++) Xg[i * 5 + c] = xv[c];
        Ug[i * 2] = uv[0]; Ug[i * 2 + 1] = uv[1];
    }
}

// set_initial_guess, robot_ocp_problem.py:286-306: X[i] = [x0_x, x0_y, x0_psi, 0, 0], U = 0.
// goal != null: the straight-line variant of the commented block :293-300 (interp_guess, rti_kernel.hpp).
__global__ void reset_guess_kernel(int batch, int N, const double *__restrict__ x0, const double *__restrict__ goal, double *__restrict__ X, double *__restrict__ U)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * (N + 1)) return;
    const int inst = t / (N + 1), i = t - inst * (N + 1);
    const double *x = x0 + (size_t)inst * 5;
    double *Xi = X + (size_t)t * 5;
    Xi[0] = x[0]; Xi[1] = x[1]; Xi[2] = x[2]; Xi[3] = 0.0; Xi[4] = 0.0;
    if (goal) {
        const double xs[5] = {x[0], x[1], x[2], x[3], x[4]};
        double xg[5];
        interp_guess(xs, goal[(size_t)inst * 2 + 1], i, N, xg);
#pragma unroll
        for (int c = 0; c < 5; c++) Xi[c] = xg[c];
    }
    if (i < N) { double *Ui = U + ((size_t)inst * N + i) * 2; Ui[0] = 0.0; Ui[1] = 0.0; }
}

// Plant integrator, robot_ocp_problem.py:207-212.
__global__ void plant_step_kernel(int batch, double dt, const double *__restrict__ x, const double *__restrict__ u, double *__restrict__ xn)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch) return;
    double xi[5], ui[2], xo[5];
#pragma unroll
    for (int c = 0; c < 5; c++) xi[c] = x[(size_t)t * 5 + c];
    ui[0] = u[(size_t)t * 2]; ui[1] = u[(size_t)t * 2 + 1];
    dyn_step<false>(xi, ui, dt, xo, nullptr, nullptr);
#pragma unroll
    for (int c = 0; c < 5; c++) xn[(size_t)t * 5 + c] = xo[c];
}

// The stationarity sweep of the polish on its own (tests only): the open-loop adjoint of a GIVEN per-stage gradient g[B][N+1][7] over the linearisation of a given
// iterate, through rti_kernel.hpp::adjoint_inputs in the lane layouts of the solve kernels -- G lanes per instance with one lane per stage (G = 16, 21, 32, 64), or LPS
// lanes per stage with one instance per wavefront (rti_split_kernel: the stage's first lane holds g, the others enter the suffix sums with zero).
// ru[B][N] = max(|ru_t[0]|, |ru_t[1]|), ru_t = g_u,t + B_t' pi_{t+1}, pi_t = g_x,t + A_t' pi_{t+1}.
template <int G, int LPS>
__global__ __launch_bounds__(64) void adjoint_check_kernel(KParams p, const double *__restrict__ Xin, const double *__restrict__ Uin, const double *__restrict__ gin,
                                                            double *__restrict__ ru)
{
    constexpr int IPW = LPS > 1 ? 1 : (G == 21 ? 3 : 64 / G);
    const int lane = threadIdx.x, N = p.N;
    const int slot = LPS > 1 ? 0 : (G == 21 ? seg21_slot(lane) : lane / G);
    const int i = LPS > 1 ? lane / LPS : lane - slot * G;
    const bool own = LPS > 1 ? (lane % LPS == 0) : true;
    const int inst = blockIdx.x * IPW + slot;
    const bool act = inst < p.batch && i <= N, has_u = inst < p.batch && i < N;
    double xi[5] = {0, 0, 0, 0, 0}, ui[2] = {0, 0}, g[7] = {0, 0, 0, 0, 0, 0, 0};
    StageLin S;
    S.a02 = S.a03 = S.a04 = S.a12 = S.a13 = S.a14 = S.b00 = S.b01 = S.b10 = S.b11 = 0.0; S.dt = p.dt; S.h2 = p.h2;
    if (act) {
#pragma unroll
        for (int c = 0; c < 5; c++) xi[c] = Xin[((size_t)inst * (N + 1) + i) * 5 + c];
#pragma unroll
        for (int c = 0; c < 7; c++) g[c] = gin[((size_t)inst * (N + 1) + i) * 7 + c];
    }
    if (has_u) {
        ui[0] = Uin[((size_t)inst * N + i) * 2]; ui[1] = Uin[((size_t)inst * N + i) * 2 + 1];
        double xn[5], ae[6], be[4];
        dyn_step<true>(xi, ui, p.dt, xn, ae, be);
        S.a02 = ae[0]; S.a03 = ae[1]; S.a04 = ae[2]; S.a12 = ae[3]; S.a13 = ae[4]; S.a14 = ae[5];
        S.b00 = be[0]; S.b01 = be[1]; S.b10 = be[2]; S.b11 = be[3];
    }
    const double r = adjoint_inputs<(LPS > 1 ? 64 : G)>(own && act, own && has_u, S, g, lane);
    if (own && has_u) ru[(size_t)inst * N + i] = r;
}

// Dense dump of the linearisation of the iterate (tests only): same device functions the solve kernel uses.
__global__ void linearize_kernel(KParams p, int n_obst, const double *__restrict__ Xin, const double *__restrict__ Uin,
                                 double *__restrict__ A, double *__restrict__ B, double *__restrict__ b, double *__restrict__ q,
                                 double *__restrict__ hval, double *__restrict__ dh)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = p.N;
    if (t >= p.batch * (N + 1)) return;
    const int inst = t / (N + 1), i = t - inst * (N + 1);
    const double *Xg = Xin + (size_t)inst * (N + 1) * 5, *Ug = Uin + (size_t)inst * N * 2;
    double xi[5], ui[2] = {0, 0};
#pragma unroll
    for (int c = 0; c < 5; c++) xi[c] = Xg[i * 5 + c];
    const double gx = p.goal[(size_t)inst * 2], gy = p.goal[(size_t)inst * 2 + 1];
    double *qo = q + (size_t)t * 7;
    if (i < N) {
        ui[0] = Ug[i * 2]; ui[1] = Ug[i * 2 + 1];
        double xn[5], ae[6], be[4];
        dyn_step<true>(xi, ui, p.dt, xn, ae, be);
        double *Ao = A + ((size_t)inst * N + i) * 25, *Bo = B + ((size_t)inst * N + i) * 10, *bo = b + ((size_t)inst * N + i) * 5;
        for (int c = 0; c < 25; c++) Ao[c] = 0.0;
        for (int c = 0; c < 10; c++) Bo[c] = 0.0;
        for (int c = 0; c < 5; c++) { Ao[c * 5 + c] = 1.0; bo[c] = xn[c] - Xg[(i + 1) * 5 + c]; }
        Ao[0 * 5 + 2] = ae[0]; Ao[0 * 5 + 3] = ae[1]; Ao[0 * 5 + 4] = ae[2];
        Ao[1 * 5 + 2] = ae[3]; Ao[1 * 5 + 3] = ae[4]; Ao[1 * 5 + 4] = ae[5];
        Ao[2 * 5 + 4] = p.dt;
        Bo[0] = be[0]; Bo[1] = be[1]; Bo[2] = be[2]; Bo[3] = be[3];
        Bo[2 * 2 + 1] = p.h2; Bo[3 * 2 + 0] = p.dt; Bo[4 * 2 + 1] = p.dt;
        qo[0] = p.Wg[4] * ui[0]; qo[1] = p.Wg[5] * ui[1];
        qo[2] = p.Wg[0] * (xi[0] - gx); qo[3] = p.Wg[1] * (xi[1] - gy); qo[4] = 0.0; qo[5] = p.Wg[2] * xi[3]; qo[6] = p.Wg[3] * xi[4];
    } else {
        qo[0] = qo[1] = 0.0;
        qo[2] = p.Weg[0] * (xi[0] - gx); qo[3] = p.Weg[1] * (xi[1] - gy); qo[4] = 0.0; qo[5] = p.Weg[2] * xi[3]; qo[6] = p.Weg[3] * xi[4];
    }
    const double *Pg = p.P + (size_t)t * n_obst * 2;
    for (int j = 0; j < n_obst; j++) {
        const double ex = xi[0] - Pg[2 * j], ey = xi[1] - Pg[2 * j + 1];
        hval[(size_t)t * n_obst + j] = ex * ex + ey * ey - p.r2;
        dh[((size_t)t * n_obst + j) * 2] = 2 * ex; dh[((size_t)t * n_obst + j) * 2 + 1] = 2 * ey;
    }
}

}  // namespace mpc
