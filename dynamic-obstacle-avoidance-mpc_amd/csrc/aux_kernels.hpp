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
