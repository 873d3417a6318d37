// Bench-top measurement for row g (north_star: "MFMA ... for the small dense condensed-QP GEMM"; robot_ocp_problem.py:126 PARTIAL_CONDENSING_HPIPM):
// ONE interior-point iteration's Riccati FACTOR sweep in the block-of-5 formulation -- 5 stages condensed into one block whose homogeneous vector
// z^ = (u_a0, u_al0, ..., u_a4, u_al4 | x[5], 1) is exactly one 16 x 16 FP64 tile -- as the instruction skeleton a kernel would run, one instance per
// wavefront (the C2 regime: one 512-register wavefront per SIMD), timed with clock64():
//
//   assembly   H^_b = sum_{t in block} Gamma_t' H~_t Gamma_t        4 v_mfma_f64_16x16x4 per stage (X = H~ Gamma: 2, K = 8;  H^ += Gamma' X: 2), independent
//              across stages and blocks; H~_t (8 x 8, what the row phases produce per iteration) is fetched from LDS in A-operand layout, Gamma_t (8 x 16,
//              constant over the interior point) in B-operand layout -- and the D layout of this instruction IS its B layout per K-slice (reg s of D = slice s
//              of B: D[q + 4 r][j] sits in lane j + 16 q, B[4 s + k][j] in lane j + 16 k), so X feeds the second product without a re-layout, and Gamma's
//              B registers are Gamma''s A registers.
//   link       T = P~+ W^ (2 MFMAs, K = 8; P~+ symmetric: its D / LDS tile serves as the A operand),  M~ = H^_b + W^' T (2 MFMAs onto the assembled tile),
//              M~ -> LDS -> one column per lane of DPP row 0 (16 doubles per lane), elimination of the 10 inputs as pivots 0..9 (L D L' without square
//              roots: per pivot a broadcast of the diagonal, a reciprocal with two Newton steps, 15 - p fused multiply-adds with the pivot column read
//              through row_newbcast) -- what remains in rows / columns 10..15 is P~_b, the scaled pivot rows are the gains K^ -- and P~_b back to LDS in
//              A-operand layout for the next link.  NB dependent links per iteration (NB = 4 at N = 20, 10 at N = 50).
//
// The elimination is the real arithmetic (checked against a host L D L' below); the MFMA parts run the real instruction sequence on synthetic operands
// (their timing does not depend on the data).  Reference points measured with the kernels this repository ships (profiles/r01_split_phase_timing.txt,
// r02_c5_phase_timing.txt): the stage-by-stage DPP factor sweep costs 14.1k cycles per iteration at N = 20 and ~35k at N = 50 (55k with the vector sweeps).
// Output: cycles per iteration for NB = 4 and NB = 10, split into assembly-only, links-only and both interleaved by the compiler's scheduler.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double rcp_nr(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// v_fmac_f64_dpp acc += src(lane P of the DPP row) * b
template <int P>
__device__ __forceinline__ void fmac_bcast(double &acc, double src, double b)
{
    // (no wait states needed here: the DPP source m[i] was last written by the PREVIOUS pivot's multiply-add, at least five instructions earlier)
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(b), "n"(P));
}
template <int P>
__device__ __forceinline__ double mov_bcast(double src)
{
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(P));
    return r;
}

// elimination of pivots 0..9 of the symmetric 16 x 16 matrix whose column j lives in lane j (DPP row): m[i] = M[i][j].
// After it: m[i] for i, j >= 10 is the Schur complement; lanes keep w[p] = M[p][j] / d_p (row p of L' -- the gains for j >= 10).
template <int P>
__device__ __forceinline__ void pivot(double m[16], double &dinv)
{
    const double d = mov_bcast<P>(m[P]);          // d_p = M[p][p], from lane p
    dinv = rcp_nr(d);
    const double w = -m[P] * dinv;                // -M[p][j] / d_p in lane j
#pragma unroll
    for (int i = P + 1; i < 16; i++) fmac_bcast<P>(m[i], m[i], w);      // M[i][j] -= M[i][p] M[p][j] / d_p   (M[i][p]: lane p's m[i], read before it is written)
    m[P] = -w;                                     // row p of L'
}
// the same elimination with a caller's work issued between the pivots: stage(0..4) after pivots 1, 3, 5, 7, 9 (MODE 3: the four assembly MFMAs of one stage of the
// NEXT block go to the matrix pipe while the vector unit runs the next two pivots)
template <class F>
__device__ __forceinline__ void eliminate10_with(double m[16], F stage)
{
    double di;
    pivot<0>(m, di); pivot<1>(m, di); stage(0); pivot<2>(m, di); pivot<3>(m, di); stage(1); pivot<4>(m, di); pivot<5>(m, di); stage(2);
    pivot<6>(m, di); pivot<7>(m, di); stage(3); pivot<8>(m, di); pivot<9>(m, di); stage(4);
}
__device__ __forceinline__ void eliminate10(double m[16])
{
    double di;
    pivot<0>(m, di); pivot<1>(m, di); pivot<2>(m, di); pivot<3>(m, di); pivot<4>(m, di);
    pivot<5>(m, di); pivot<6>(m, di); pivot<7>(m, di); pivot<8>(m, di); pivot<9>(m, di);
}

// LDS map (doubles): HA[t][2][64] H~_t as A operand (slice, lane); GB[t][2][64] Gamma_t as B operand; WB[b][2][64] W^_b; PA[2][64] P~+ as A operand; MT[16][16] scratch tile
template <int NB, int MODE>      // MODE 0: assembly + links, 1: assembly only, 2: links only, 3: assembly of block b - 1 interleaved by hand into the elimination of block b
__global__ __launch_bounds__(64) void block5_iteration(const double *src, double *out, long long *cyc, int iters)
{
    extern __shared__ double lds[];
    constexpr int N = 5 * NB;
    double *HA = lds, *GB = HA + N * 128, *WB = GB + N * 128, *PA = WB + NB * 128, *MT = PA + 128;
    const int l = threadIdx.x, q = l >> 4, j = l & 15;
    for (int k = l; k < N * 128; k += 64) { HA[k] = src[k % 4096] * 1e-3; GB[k] = src[(k + 977) % 4096] * 1e-2; }
    for (int k = l; k < NB * 128; k += 64) WB[k] = src[(k + 311) % 4096] * 1e-2;
    for (int k = l; k < 128; k += 64) PA[k] = ((k & 15) == (k >> 4)) ? 1.0 : 0.0;
    __syncthreads();
    d4 Hb[NB];
    double acc = 0.0;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        int off = 0;
        asm volatile("" : "+v"(off));      // the operand tiles are rewritten by the row phases every iteration: their loads may not be hoisted out of the loop
        // ---- assembly: independent per stage ----
        auto assemble_stage = [&](d4 &H, int t) {
            const double a0 = HA[t * 128 + l + off], a1 = HA[t * 128 + 64 + l + off], g0 = GB[t * 128 + l + off], g1 = GB[t * 128 + 64 + l + off];
            d4 X = {0, 0, 0, 0};
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, g0, X, 0, 0, 0);
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, g1, X, 0, 0, 0);
            H = __builtin_amdgcn_mfma_f64_16x16x4f64(g0, X[0], H, 0, 0, 0);
            H = __builtin_amdgcn_mfma_f64_16x16x4f64(g1, X[1], H, 0, 0, 0);
        };
        if (MODE == 3) {      // only the LAST block is assembled ahead of the links
            d4 H = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 5; s++) assemble_stage(H, 5 * (NB - 1) + s);
            Hb[NB - 1] = H;
        } else
        if (MODE != 2) {
#pragma unroll
            for (int b = 0; b < NB; b++) {
                d4 H = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 5; s++) {
                    const int t = 5 * b + s;
                    const double a0 = HA[t * 128 + l + off], a1 = HA[t * 128 + 64 + l + off], g0 = GB[t * 128 + l + off], g1 = GB[t * 128 + 64 + l + off];
                    d4 X = {0, 0, 0, 0};
                    X = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, g0, X, 0, 0, 0);        // X = H~_t Gamma_t  (rows 0..7 of the tile)
                    X = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, g1, X, 0, 0, 0);
                    H = __builtin_amdgcn_mfma_f64_16x16x4f64(g0, X[0], H, 0, 0, 0);      // H^ += Gamma_t' X: Gamma's B registers are Gamma''s A registers, X's D registers its B slices
                    H = __builtin_amdgcn_mfma_f64_16x16x4f64(g1, X[1], H, 0, 0, 0);
                }
                Hb[b] = H;
            }
        } else {
#pragma unroll
            for (int b = 0; b < NB; b++) { Hb[b][0] = src[l] + b; Hb[b][1] = 0.0; Hb[b][2] = 0.0; Hb[b][3] = (q == 3 ? 4.0 : 0.0); }
        }
        // ---- links: dependent ----
        if (MODE != 1) {
#pragma unroll
            for (int b = NB - 1; b >= 0; b--) {
                const double p0 = PA[l], p1 = PA[64 + l], w0 = WB[b * 128 + l + off], w1 = WB[b * 128 + 64 + l + off];
                d4 T = {0, 0, 0, 0};
                T = __builtin_amdgcn_mfma_f64_16x16x4f64(p0, w0, T, 0, 0, 0);           // T = P~+ W^
                T = __builtin_amdgcn_mfma_f64_16x16x4f64(p1, w1, T, 0, 0, 0);
                d4 M = Hb[b];
                M = __builtin_amdgcn_mfma_f64_16x16x4f64(w0, T[0], M, 0, 0, 0);         // M~ = H^_b + W^' T
                M = __builtin_amdgcn_mfma_f64_16x16x4f64(w1, T[1], M, 0, 0, 0);
                // D layout -> one column per lane (row i = q + 4 r of column j)
#pragma unroll
                for (int r = 0; r < 4; r++) MT[(q + 4 * r) * 16 + j] = M[r];
                __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): a wavefront's LDS operations complete in order; the wait makes the stores visible to the loads
                double m[16];
#pragma unroll
                for (int i = 0; i < 16; i++) m[i] = MT[i * 16 + j] + (i == j ? 50.0 : 0.0);      // (+ a diagonal: keeps the synthetic tile positive definite)
                if (MODE == 3 && b > 0) {
                    d4 Hn = {0, 0, 0, 0};
                    eliminate10_with(m, [&](int s) { assemble_stage(Hn, 5 * (b - 1) + s); });
                    Hb[b - 1] = Hn;
                } else
                    eliminate10(m);
                // P~_b (rows / columns 10..15) -> A-operand tile for the next link: PA[slice][lane i + 16 k] = P[i][4 slice + k], zero outside 6 x 6
                if (j >= 10) {
#pragma unroll
                    for (int i = 10; i < 16; i++) { const int ii = i - 10, kk = j - 10; PA[(kk >> 2) * 64 + ii + 16 * (kk & 3)] = m[i] * 1e-3; }
                }
                acc += m[15];
                __builtin_amdgcn_s_waitcnt(0xc07f);
            }
        } else {
#pragma unroll
            for (int b = 0; b < NB; b++) acc += Hb[b][0] + Hb[b][1] + Hb[b][2] + Hb[b][3];
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + l] = acc;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

// numerical check of eliminate10: one 16 x 16 SPD matrix, Schur complement against the host
__global__ void check_elim(const double *Min, double *Sout)
{
    const int j = threadIdx.x & 15;
    double m[16];
    for (int i = 0; i < 16; i++) m[i] = Min[i * 16 + j];
    eliminate10(m);
    if (threadIdx.x < 16) for (int i = 0; i < 16; i++) Sout[i * 16 + j] = m[i];
}

template <int NB, int MODE>
static double run(const double *src, double *out, long long *cyc, int blocks, int iters)
{
    const size_t shm = (size_t)(2 * 5 * NB * 128 + NB * 128 + 128 + 256) * sizeof(double);
    (void)hipFuncSetAttribute((const void *)block5_iteration<NB, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    block5_iteration<NB, MODE><<<blocks, 64, shm>>>(src, out, cyc, iters);
    (void)hipDeviceSynchronize();
    block5_iteration<NB, MODE><<<blocks, 64, shm>>>(src, out, cyc, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return -1; }
    std::vector<long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0; for (long long v : h) s += (double)v;
    return s / blocks / iters;
}

int main()
{
    double *src, *out, *Min, *Sout; long long *cyc;
    const int blocks = 1024;
    (void)hipMalloc(&src, 4096 * 8); (void)hipMalloc(&out, blocks * 64 * 8); (void)hipMalloc(&cyc, blocks * 8); (void)hipMalloc(&Min, 256 * 8); (void)hipMalloc(&Sout, 256 * 8);
    std::vector<double> h(4096); for (int i = 0; i < 4096; i++) h[i] = 0.5 + 0.001 * ((i * 7919) % 997);
    (void)hipMemcpy(src, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    // --- elimination check ---
    double M[256], S[256], R[256];
    for (int i = 0; i < 16; i++) for (int jj = 0; jj < 16; jj++) M[i * 16 + jj] = (i == jj ? 20.0 + i : 0.0) + 0.3 * std::sin(1.0 + i * jj) + 0.3 * std::sin(1.0 + jj * i);
    for (int k = 0; k < 256; k++) R[k] = M[k];
    for (int p = 0; p < 10; p++) for (int i = p + 1; i < 16; i++) { const double lip = R[i * 16 + p] / R[p * 16 + p]; for (int jj = 0; jj < 16; jj++) if (jj != p) R[i * 16 + jj] -= lip * R[p * 16 + jj]; }
    (void)hipMemcpy(Min, M, sizeof(M), hipMemcpyHostToDevice);
    check_elim<<<1, 64>>>(Min, Sout); (void)hipMemcpy(S, Sout, sizeof(S), hipMemcpyDeviceToHost);
    double err = 0; for (int i = 10; i < 16; i++) for (int jj = 10; jj < 16; jj++) err = std::fmax(err, std::fabs(S[i * 16 + jj] - R[i * 16 + jj]));
    printf("elimination of 10 pivots: max |Schur complement - host| = %.3e\n", err);
    const int iters = 50;
    printf("cycles per interior-point iteration (factor sweep skeleton, one wavefront per SIMD, %d wavefronts):\n", blocks);
    const double a4 = run<4, 1>(src, out, cyc, blocks, iters), l4 = run<4, 2>(src, out, cyc, blocks, iters), b4 = run<4, 0>(src, out, cyc, blocks, iters);
    const double i4 = run<4, 3>(src, out, cyc, blocks, iters);
    printf("  N = 20 (4 blocks):  assembly only %.0f   links only %.0f (%.0f per link)   both %.0f   both, assembly interleaved by hand %.0f      [stage-by-stage DPP sweep today: 14100]\n", a4, l4, l4 / 4, b4, i4);
    const double a10 = run<10, 1>(src, out, cyc, blocks, iters), l10 = run<10, 2>(src, out, cyc, blocks, iters), b10 = run<10, 0>(src, out, cyc, blocks, iters);
    const double i10 = run<10, 3>(src, out, cyc, blocks, iters);
    printf("  N = 50 (10 blocks): assembly only %.0f   links only %.0f (%.0f per link)   both %.0f   both, assembly interleaved by hand %.0f      [today: ~35000]\n", a10, l10, l10 / 10, b10, i10);
    printf("  ratio to today's factor sweep: N = 20 %.2f (interleaved %.2f), N = 50 %.2f (interleaved %.2f)\n", b4 / 14100.0, i4 / 14100.0, b10 / 35000.0, i10 / 35000.0);
    return 0;
}
