// rti_kernel.hpp -- the RTI solve kernel: one SQP_RTI iteration per MPC instance, gfx950 (MI355X).
//
// What it replaces: everything `ocp_solver.solve()` does in the reference
// (src/simulation/robot_ocp_problem.py:195, options :126-132), plus the per-step parameter uploads around it
// (:145-152 slack schedule, :154-166 obstacle parameters, :191-192 initial-state bounds).  See DESIGN.md.
//
// Mapping (v1): ONE INSTANCE PER WAVEFRONT (64 lanes, one workgroup = one wave).
//   * lane i owns horizon stage i (N+1 <= 64): its linearisation, its inequality rows (multiplier lam, slack t for the
//     4 input-box, 8 state-box and 2*NOBST soft-obstacle rows live in that lane's REGISTERS for the whole solve);
//   * the per-stage blocks that the Riccati recursion consumes/produces are staged in LDS (92 N + 27 doubles
//     per instance: 14.9 KB at N=20, 37 KB at N=50), never in HBM;
//   * wavefront reductions (max step ratio, complementarity sum / max) are shuffle butterflies;
//   * the stage recursion (backward Riccati, forward rollout) is sequential in the stage index; inside a stage the
//     factorisation is spread column-per-lane over 7 lanes (M = H~ + W'PW, wave-uniform P re-broadcast by v_readlane),
//     hand-expanded for the sparsity of A_i = I + E_i (6 non-trivial entries) and B_i (4 non-trivial entries).
// HBM traffic is therefore the algorithmic minimum: read x0, goal, P, X, U once, write X, U, u0, cost, status once.
//
// Interior point method: Mehrotra predictor-corrector in residual ("delta") form.  The costates the stationarity
// residual needs come from the adjoint recursion pi_i = (H z + q - C'lam)_x + A_i' pi_{i+1}, fused into the backward
// Riccati sweep (it zeroes the state blocks of the residual exactly).  The corrector solves only the homogeneous
// system for the difference of right-hand sides.  Dynamics / initial-condition residuals decay by prod(1 - alpha_k).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Diagnostic build only (-DMPC_PHASE_TIMING, scripts/phase_timing.py): per-phase cycle counts into KParams::trace.
#ifdef MPC_PHASE_TIMING
#define MPC_T0() long long t_prev_ = clock64()
#define MPC_TICK(k) do { const long long t_now_ = clock64(); tacc_[k] += t_now_ - t_prev_; t_prev_ = t_now_; } while (0)
#else
#define MPC_T0() do {} while (0)
#define MPC_TICK(k) do {} while (0)
#endif

namespace mpc {

struct KParams {
    int N, batch;
    int soft_h, bx_terminal, iter_max;
    double dt, h2;            // dt, dt^2/2
    double Hd_stage[7];       // diag of the GN Hessian + LM, z order (ua, ual, x, y, psi, v, om), stages < N
    double Hd_term[5];        // terminal (x, y, psi, v, om)
    double Wg[6];             // cs * W  for y = [x, y, v, om, ua, ual]  (gradient and cost weights)
    double Weg[4];            // W_e
    double bx_lo[4], bx_hi[4], bu_lo[2], bu_hi[2];
    double r2;                // r_safe^2
    double slack_a, slack_b, ss;  // ss: penalty scale for stages < N (dt or 1)
    double tol, mu0, thr0;
    const double *x0, *P, *goal;
    double *X, *U, *u0, *cost;
    int32_t *status, *iters;
    double *trace;            // optional [batch][iter_max][4] = (mu, sigma, alpha, cmax) per IPM iteration (debug)
};

static constexpr double kTLMin = 1e-13;  // floor for lam and t (see oracle/mpc_oracle.c TL_MIN)

// Gauss-Legendre 4-point rule on [0,1]: the reference's IRK integrator (robot_ocp_problem.py:129) with acados defaults
// (GL, 4 stages, 1 step) collapses to closed-form psi,v,omega and this quadrature for x,y.
__device__ static constexpr double kGLC[4] = {0.069431844202973712388, 0.330009478207571867599,
                                              0.669990521792428132401, 0.930568155797026287612};
__device__ static constexpr double kGLB[4] = {0.173927422568726928687, 0.326072577431273071313,
                                              0.326072577431273071313, 0.173927422568726928687};

// One integrator step and the non-trivial entries of A = dF/dx, B = dF/du.
//   ae = {A02, A03, A04, A12, A13, A14}, be = {B00, B01, B10, B11};  A22.. = I, A24 = dt, B21 = dt^2/2, B30 = B41 = dt.
// src/models/robot_model.py:39-43
template <bool JAC>
__device__ __forceinline__ void dyn_step(const double x[5], const double u[2], double dt, double xn[5], double ae[6], double be[4])
{
    const double psi = x[2], v = x[3], om = x[4], a = u[0], al = u[1];
    double sx = 0, sy = 0, xpsi = 0, xv = 0, xom = 0, xa = 0, xal = 0, ypsi = 0, yv = 0, yom = 0, ya = 0, yal = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const double tau = kGLC[j] * dt, w = kGLB[j] * dt;
        const double vj = v + a * tau;
        const double pj = psi + om * tau + 0.5 * al * tau * tau;
        double sj, cj;
        sincos(pj, &sj, &cj);
        const double wvc = w * vj * cj, wvs = w * vj * sj;
        sx += wvc; sy += wvs;
        if (JAC) {
            const double ht2 = 0.5 * tau * tau;
            xpsi -= wvs;        ypsi += wvc;
            xv += w * cj;       yv += w * sj;
            xom -= wvs * tau;   yom += wvc * tau;
            xa += w * tau * cj; ya += w * tau * sj;
            xal -= wvs * ht2;   yal += wvc * ht2;
        }
    }
    xn[0] = x[0] + sx; xn[1] = x[1] + sy;
    xn[2] = psi + om * dt + 0.5 * al * dt * dt;
    xn[3] = v + a * dt;
    xn[4] = om + al * dt;
    if (JAC) {
        ae[0] = xpsi; ae[1] = xv; ae[2] = xom; ae[3] = ypsi; ae[4] = yv; ae[5] = yom;
        be[0] = xa; be[1] = xal; be[2] = ya; be[3] = yal;
    }
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// reciprocal: hardware seed + two Newton steps (1-2 ulp); used for 1/t of the inequality rows
__device__ __forceinline__ double rcp_nr(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// Sparse inner products with the columns of W = [B A] (5 x 7) of one stage.
struct StageLin {
    double a02, a03, a04, a12, a13, a14, b00, b01, b10, b11, dt, h2;
    __device__ __forceinline__ double dua(const double T[5]) const { return b00 * T[0] + b10 * T[1] + dt * T[3]; }
    __device__ __forceinline__ double dual(const double T[5]) const { return b01 * T[0] + b11 * T[1] + h2 * T[2] + dt * T[4]; }
    __device__ __forceinline__ double dpsi(const double T[5]) const { return a02 * T[0] + a12 * T[1] + T[2]; }
    __device__ __forceinline__ double dv(const double T[5]) const { return a03 * T[0] + a13 * T[1] + T[3]; }
    __device__ __forceinline__ double dom(const double T[5]) const { return a04 * T[0] + a14 * T[1] + dt * T[2] + T[4]; }
};

// LDS carve-up for one instance (doubles)
struct LdsMap {
    double *AE, *BE, *BB, *WC, *HQ, *GQ, *GX, *KK, *MI, *KV, *ZH;
    __device__ __forceinline__ LdsMap(double *base, int N)
    {
        AE = base;            // [N][6]
        BE = AE + 6 * N;      // [N][4]
        BB = BE + 4 * N;      // [N][5]   dynamics defects b_i of the SQP iterate
        WC = BB + 5 * N;      // [N][7][5] columns of W = [B A]: column c (ua, ual, x, y, psi, v, om) as a 5-vector
        HQ = WC + 35 * N;     // [N+1][8] barrier-modified Hessian: diagonal in z order (Ruu0, Ruu1, Qxx, Qyy, Qpsi, Qvv, Qww), then Qxy
        GQ = HQ + 8 * (N + 1);// [N+1][7] linear term: (l_u0, l_u1, cb_x[5]) for the predictor, the rhs difference for the corrector
        GX = GQ + 7 * (N + 1);// [N+1][5] local Lagrangian gradient (H z + q - C'lam)_x, input of the costate recursion
        KK = GX + 5 * (N + 1);// [N][10]  feedback gains K (2 x 5)
        MI = KK + 10 * N;     // [N][3]   LDL' factors of Muu: 1/d0, l, 1/d1
        KV = MI + 3 * N;      // [N][2]   feed-forward k
        ZH = KV + 2 * N;      // [N+1][7] Newton step dz = (du, dx)
    }
    static __host__ __device__ constexpr int doubles(int N) { return 92 * N + 27; }
};

// wave-uniform copy of lane `src`'s value (v_readlane_b32 x2 -> SGPR pair)
__device__ __forceinline__ double bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ StageLin load_stage_lin(const LdsMap L, int i, double dt, double h2)
{
    StageLin S;
    const double *ae = L.AE + 6 * i, *be = L.BE + 4 * i;
    S.a02 = ae[0]; S.a03 = ae[1]; S.a04 = ae[2]; S.a12 = ae[3]; S.a13 = ae[4]; S.a14 = ae[5];
    S.b00 = be[0]; S.b01 = be[1]; S.b10 = be[2]; S.b11 = be[3]; S.dt = dt; S.h2 = h2;
    return S;
}

// Backward Riccati factorisation + predictor right-hand side, COLUMN-PER-LANE: lane c < 7 owns column c of
// M = H~ + W'PW (z order ua, ual, x, y, psi, v, om); lanes >= 7 mirror lane 6.  The cost-to-go Hessian P (15 unique
// entries), its gradient pv and the costate pi are wave-uniform; after each stage the new P / pv are re-broadcast from
// the owning lanes with v_readlane.  Muu is solved by LDL' (see DESIGN.md section 2 on why not the closed-form inverse).
__device__ __forceinline__ void factor_sweep(const LdsMap L, int N, int lane, double dt, double h2, double rs)
{
    const int c = lane < 7 ? lane : 6;
    double P[5][5], pv[5], pi[5];
    {
        const double *hq = L.HQ + 8 * N, *gq = L.GQ + 7 * N, *gx = L.GX + 5 * N;
#pragma unroll
        for (int r = 0; r < 5; r++)
#pragma unroll
            for (int k = 0; k < 5; k++) P[r][k] = 0.0;
        P[0][0] = hq[2]; P[1][1] = hq[3]; P[0][1] = P[1][0] = hq[7]; P[2][2] = hq[4]; P[3][3] = hq[5]; P[4][4] = hq[6];
#pragma unroll
        for (int k = 0; k < 5; k++) { pv[k] = gq[2 + k]; pi[k] = gx[k]; }
    }
    for (int i = N - 1; i >= 0; i--) {
        const StageLin S = load_stage_lin(L, i, dt, h2);
        double w[5];
        {
            const double *wc = L.WC + 35 * i + 5 * c;
#pragma unroll
            for (int k = 0; k < 5; k++) w[k] = wc[k];
        }
        const double hd = L.HQ[8 * i + c], qxy = L.HQ[8 * i + 7];
        double g = L.GQ[7 * i + c];
        // stationarity residual of the input block with the costate of the NEXT stage; then this stage's costate
        {
            const double wpi = w[0] * pi[0] + w[1] * pi[1] + w[2] * pi[2] + w[3] * pi[3] + w[4] * pi[4];
            g += (c < 2) ? wpi : 0.0;
            const double *gx = L.GX + 5 * i;
            const double n0 = gx[0] + pi[0], n1 = gx[1] + pi[1], n2 = gx[2] + S.dpsi(pi), n3 = gx[3] + S.dv(pi), n4 = gx[4] + S.dom(pi);
            pi[0] = n0; pi[1] = n1; pi[2] = n2; pi[3] = n3; pi[4] = n4;
        }
        // column c of M
        double T[5];
#pragma unroll
        for (int k = 0; k < 5; k++) T[k] = P[k][0] * w[0] + P[k][1] * w[1] + P[k][2] * w[2] + P[k][3] * w[3] + P[k][4] * w[4];
        double Mc[7] = {S.dua(T), S.dual(T), T[0], T[1], S.dpsi(T), S.dv(T), S.dom(T)};
#pragma unroll
        for (int r = 0; r < 7; r++) Mc[r] += (r == c) ? hd : 0.0;
        Mc[3] += (c == 2) ? qxy : 0.0;
        Mc[2] += (c == 3) ? qxy : 0.0;
        // the two input rows of M, wave-uniform (lane 0 / lane 1 hold them as columns; M is symmetric)
        double Mu0[7], Mu1[7];
#pragma unroll
        for (int r = 0; r < 7; r++) { Mu0[r] = bcast(Mc[r], 0); Mu1[r] = bcast(Mc[r], 1); }
        const double i00 = 1.0 / Mu0[0];
        const double l = Mu0[1] * i00;
        const double i11 = 1.0 / (Mu1[1] - l * Mu0[1]);
        // gains of this lane's column: K[:, c] = -Muu^{-1} M[u, c]
        const double x1 = (Mc[1] - l * Mc[0]) * i11;
        const double K1c = -x1, K0c = -(Mc[0] * i00 - l * x1);
        // affine part (needs the OLD P): Pb = P r_b + p, r_b = rs * b_i
        double Pb[5];
        if (rs != 0.0) {
            const double *bb = L.BB + 5 * i;
            const double b0 = rs * bb[0], b1 = rs * bb[1], b2 = rs * bb[2], b3 = rs * bb[3], b4 = rs * bb[4];
#pragma unroll
            for (int k = 0; k < 5; k++) Pb[k] = pv[k] + P[k][0] * b0 + P[k][1] * b1 + P[k][2] * b2 + P[k][3] * b3 + P[k][4] * b4;
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) Pb[k] = pv[k];
        }
        const double m = g + (w[0] * Pb[0] + w[1] * Pb[1] + w[2] * Pb[2] + w[3] * Pb[3] + w[4] * Pb[4]);
        const double m0 = bcast(m, 0), m1 = bcast(m, 1);
        const double kx1 = (m1 - l * m0) * i11;
        const double k1 = -kx1, k0 = -(m0 * i00 - l * kx1);
        const double pvc = m + K0c * m0 + K1c * m1;           // lanes c >= 2: entry c-2 of the new cost-to-go gradient
        // column c-2 of the new cost-to-go Hessian (lanes c >= 2): P+[r][c-2] = M[2+r][c] + K0c M[0][2+r] + K1c M[1][2+r]
        double Pn[5];
#pragma unroll
        for (int r = 0; r < 5; r++) Pn[r] = Mc[2 + r] + K0c * Mu0[2 + r] + K1c * Mu1[2 + r];
        if (lane >= 2 && lane < 7) { L.KK[10 * i + (lane - 2)] = K0c; L.KK[10 * i + 5 + (lane - 2)] = K1c; }
        if (lane == 0) { L.MI[3 * i] = i00; L.MI[3 * i + 1] = l; L.MI[3 * i + 2] = i11; L.KV[2 * i] = k0; L.KV[2 * i + 1] = k1; }
#pragma unroll
        for (int r = 0; r < 5; r++)
#pragma unroll
            for (int cc = r; cc < 5; cc++) { const double v = bcast(Pn[r], 2 + cc); P[r][cc] = v; P[cc][r] = v; }
#pragma unroll
        for (int k = 0; k < 5; k++) pv[k] = bcast(pvc, 2 + k);
    }
}

// Corrector right-hand side: homogeneous system, reuses K and the LDL' factors; wave-uniform (every lane computes the same).
__device__ __forceinline__ void corrector_sweep(const LdsMap L, int N, int lane, double dt, double h2)
{
    double pv[5];
#pragma unroll
    for (int k = 0; k < 5; k++) pv[k] = L.GQ[7 * N + 2 + k];
    for (int i = N - 1; i >= 0; i--) {
        const StageLin S = load_stage_lin(L, i, dt, h2);
        const double *gq = L.GQ + 7 * i, *kk = L.KK + 10 * i, *mi = L.MI + 3 * i;
        const double i00 = mi[0], l = mi[1], i11 = mi[2];
        const double m0 = gq[0] + S.dua(pv), m1 = gq[1] + S.dual(pv);
        const double mx[5] = {gq[2] + pv[0], gq[3] + pv[1], gq[4] + S.dpsi(pv), gq[5] + S.dv(pv), gq[6] + S.dom(pv)};
        const double x1 = (m1 - l * m0) * i11;
        const double k0 = -(m0 * i00 - l * x1), k1 = -x1;
        if (lane == 0) { L.KV[2 * i] = k0; L.KV[2 * i + 1] = k1; }
#pragma unroll
        for (int k = 0; k < 5; k++) pv[k] = mx[k] + kk[k] * m0 + kk[5 + k] * m1;
    }
}

// Forward rollout of the Newton step through the stored gains; wave-uniform.  AFFINE: with the terms rs * b_i, rs * d0.
template <bool AFFINE>
__device__ __forceinline__ void forward_rollout(const LdsMap L, int N, int lane, double dt, double h2, const double d0[5], double rs)
{
    double x[5];
#pragma unroll
    for (int k = 0; k < 5; k++) x[k] = AFFINE ? rs * d0[k] : 0.0;
    for (int i = 0; i < N; i++) {
        const double *kk = L.KK + 10 * i, *ae = L.AE + 6 * i, *be = L.BE + 4 * i;
        double u0 = L.KV[2 * i], u1 = L.KV[2 * i + 1];
#pragma unroll
        for (int k = 0; k < 5; k++) { u0 += kk[k] * x[k]; u1 += kk[5 + k] * x[k]; }
        if (lane == 0) {
            double *zh = L.ZH + 7 * i;
            zh[0] = u0; zh[1] = u1;
#pragma unroll
            for (int k = 0; k < 5; k++) zh[2 + k] = x[k];
        }
        double xn0 = x[0] + ae[0] * x[2] + ae[1] * x[3] + ae[2] * x[4] + be[0] * u0 + be[1] * u1;
        double xn1 = x[1] + ae[3] * x[2] + ae[4] * x[3] + ae[5] * x[4] + be[2] * u0 + be[3] * u1;
        double xn2 = x[2] + dt * x[4] + h2 * u1;
        double xn3 = x[3] + dt * u0;
        double xn4 = x[4] + dt * u1;
        if (AFFINE) {
            const double *bb = L.BB + 5 * i;
            xn0 += rs * bb[0]; xn1 += rs * bb[1]; xn2 += rs * bb[2]; xn3 += rs * bb[3]; xn4 += rs * bb[4];
        }
        x[0] = xn0; x[1] = xn1; x[2] = xn2; x[3] = xn3; x[4] = xn4;
    }
    if (lane == 0) {
        double *zh = L.ZH + 7 * N;
        zh[0] = 0.0; zh[1] = 0.0;
#pragma unroll
        for (int k = 0; k < 5; k++) zh[2 + k] = x[k];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The solve kernel.  grid = batch workgroups of 64 threads; dynamic LDS = LdsMap::doubles(N) * 8 bytes.
// ------------------------------------------------------------------------------------------------------------------
template <int NOBST>
__global__ __launch_bounds__(64) void rti_solve_kernel(const KParams p)
{
    extern __shared__ double lds_raw[];
    const int inst = blockIdx.x;
    if (inst >= p.batch) return;
    const int lane = threadIdx.x;
    const int N = p.N;
    const int i = lane;                       // this lane's stage
    const bool act = (i <= N);
    const bool has_u = (i < N);
    const bool xb = (i >= 1) && (i < N || p.bx_terminal);
    const LdsMap L(lds_raw, N);
    const double dt = p.dt, h2 = p.h2;

    // ---- load (coalesced: consecutive lanes read consecutive stages of this instance's records) ----
    double x0v[5], gl[2];
#pragma unroll
    for (int c = 0; c < 5; c++) x0v[c] = p.x0[(size_t)inst * 5 + c];
    gl[0] = p.goal[(size_t)inst * 2]; gl[1] = p.goal[(size_t)inst * 2 + 1];
    double *Xg = p.X + (size_t)inst * (N + 1) * 5, *Ug = p.U + (size_t)inst * N * 2;
    const double *Pg = p.P + ((size_t)inst * (N + 1) + (act ? i : 0)) * NOBST * 2;
    double xi[5] = {0, 0, 0, 0, 0}, ui[2] = {0, 0}, xnext[5] = {0, 0, 0, 0, 0};
    if (act) {
#pragma unroll
        for (int c = 0; c < 5; c++) xi[c] = Xg[i * 5 + c];
    }
    if (has_u) {
        ui[0] = Ug[i * 2]; ui[1] = Ug[i * 2 + 1];
#pragma unroll
        for (int c = 0; c < 5; c++) xnext[c] = Xg[(i + 1) * 5 + c];
    }

    // ---- slack schedule, robot_ocp_problem.py:145-152 ----
    double zpen = 0.0;
    {
        const double ex = x0v[0] - gl[0], ey = x0v[1] - gl[1];
        const double scale = p.slack_a * (ex * ex + ey * ey + x0v[3] * x0v[3] + x0v[4] * x0v[4] + p.slack_b);
        const double alpha_i = scale * (double)(N - i) / (double)N;
        zpen = alpha_i * (has_u ? p.ss : 1.0);
    }
    const bool vs = act && (i >= 1) && (p.soft_h ? (zpen > 0.0) : true);   // obstacle rows present at this stage
    const bool soft = p.soft_h != 0;

    // ---- linearise ----
    double lin0 = 0.0;
    double d0[5] = {0, 0, 0, 0, 0};
    if (has_u) {
        double xn[5], ae[6], be[4];
        dyn_step<true>(xi, ui, dt, xn, ae, be);
#pragma unroll
        for (int c = 0; c < 6; c++) L.AE[6 * i + c] = ae[c];
#pragma unroll
        for (int c = 0; c < 4; c++) L.BE[4 * i + c] = be[c];
#pragma unroll
        for (int c = 0; c < 5; c++) { const double b = xn[c] - xnext[c]; L.BB[5 * i + c] = b; lin0 = fmax(lin0, fabs(b)); }
        // columns of W = [B A] as 5-vectors, for the column-per-lane factorisation
        const double wcol[7][5] = {{be[0], be[2], 0.0, dt, 0.0}, {be[1], be[3], h2, 0.0, dt}, {1.0, 0.0, 0.0, 0.0, 0.0}, {0.0, 1.0, 0.0, 0.0, 0.0},
                                   {ae[0], ae[3], 1.0, 0.0, 0.0}, {ae[1], ae[4], 0.0, 1.0, 0.0}, {ae[2], ae[5], dt, 0.0, 1.0}};
#pragma unroll
        for (int cc = 0; cc < 7; cc++)
#pragma unroll
            for (int k = 0; k < 5; k++) L.WC[35 * i + 5 * cc + k] = wcol[cc][k];
    }
    if (i == 0) {
#pragma unroll
        for (int c = 0; c < 5; c++) { d0[c] = x0v[c] - xi[c]; lin0 = fmax(lin0, fabs(d0[c])); }
    }
#pragma unroll
    for (int c = 0; c < 5; c++) d0[c] = bcast(d0[c], 0);      // wave-uniform: every lane runs the rollouts
    // Gauss-Newton gradient q and diagonal Hessian, z order (ua, ual, x, y, psi, v, om); robot_ocp_problem.py:59-83
    double q[7], Hd[7];
    if (has_u) {
        q[0] = p.Wg[4] * ui[0]; q[1] = p.Wg[5] * ui[1];
        q[2] = p.Wg[0] * (xi[0] - gl[0]); q[3] = p.Wg[1] * (xi[1] - gl[1]); q[4] = 0.0;
        q[5] = p.Wg[2] * xi[3]; q[6] = p.Wg[3] * xi[4];
#pragma unroll
        for (int c = 0; c < 7; c++) Hd[c] = p.Hd_stage[c];
    } else {
        q[0] = q[1] = 0.0;
        q[2] = p.Weg[0] * (xi[0] - gl[0]); q[3] = p.Weg[1] * (xi[1] - gl[1]); q[4] = 0.0;
        q[5] = p.Weg[2] * xi[3]; q[6] = p.Weg[3] * xi[4];
        Hd[0] = Hd[1] = 0.0;
#pragma unroll
        for (int c = 0; c < 5; c++) Hd[2 + c] = p.Hd_term[c];
    }

    // ---- inequality rows of this stage, in registers ----
    // box variables k: 0 ua, 1 ual, 2 x, 3 y, 4 v, 5 om  -> z index {0,1,2,3,5,6} (also the slot in Hq below)
    constexpr int NB = 6;
    const int zidx[NB] = {0, 1, 2, 3, 5, 6};
    double cl[NB], ch[NB], ll[NB], tl[NB], lh[NB], th[NB], rtl[NB], rth[NB], ppl[NB], pph[NB];
    bool vb[NB];
    {
        const double val[NB] = {ui[0], ui[1], xi[0], xi[1], xi[3], xi[4]};
        const double lo[NB] = {p.bu_lo[0], p.bu_lo[1], p.bx_lo[0], p.bx_lo[1], p.bx_lo[2], p.bx_lo[3]};
        const double hi[NB] = {p.bu_hi[0], p.bu_hi[1], p.bx_hi[0], p.bx_hi[1], p.bx_hi[2], p.bx_hi[3]};
#pragma unroll
        for (int k = 0; k < NB; k++) {
            vb[k] = (k < 2) ? has_u : xb;
            cl[k] = val[k] - lo[k]; ch[k] = hi[k] - val[k];
            tl[k] = fmax(cl[k], p.thr0); th[k] = fmax(ch[k], p.thr0);
            rtl[k] = rcp_nr(tl[k]); rth[k] = rcp_nr(th[k]);
            ll[k] = p.mu0 * rtl[k]; lh[k] = p.mu0 * rth[k];
            ppl[k] = pph[k] = 0.0;
            if (vb[k]) lin0 = fmax(lin0, fmax(tl[k] - cl[k], th[k] - ch[k]));
        }
    }
    // obstacle rows j: rho1 = h + a'dx + s >= 0 (lam1,t1), rho2 = s >= 0 (lam2,t2); robot_model.py:60-65
    double hh[NOBST], ax[NOBST], ay[NOBST], sv[NOBST], l1[NOBST], t1[NOBST], l2[NOBST], t2[NOBST], rt1[NOBST], rt2[NOBST], pp1[NOBST], pp2[NOBST];
#pragma unroll
    for (int j = 0; j < NOBST; j++) {
        double px = 0, py = 0;
        if (act) { px = Pg[2 * j]; py = Pg[2 * j + 1]; }
        const double ex = xi[0] - px, ey = xi[1] - py;
        hh[j] = ex * ex + ey * ey - p.r2; ax[j] = 2 * ex; ay[j] = 2 * ey;
        if (soft) {
            sv[j] = (hh[j] < 0 ? -hh[j] : 0.0) + p.thr0;
            t1[j] = fmax(hh[j] + sv[j], p.thr0);
            t2[j] = fmax(sv[j], p.thr0);
        } else {
            sv[j] = 0.0; t1[j] = fmax(hh[j], p.thr0); t2[j] = 1.0;
            if (vs) lin0 = fmax(lin0, t1[j] - hh[j]);
        }
        rt1[j] = rcp_nr(t1[j]); rt2[j] = rcp_nr(t2[j]);
        l1[j] = p.mu0 * rt1[j]; l2[j] = soft ? p.mu0 * rt2[j] : 0.0;
        pp1[j] = pp2[j] = 0.0;
    }
    int n_items_lane = 0;
#pragma unroll
    for (int k = 0; k < NB; k++) n_items_lane += vb[k] ? 2 : 0;
    n_items_lane += vs ? (soft ? 2 * NOBST : NOBST) : 0;
    const double n_items = wave_sum((double)n_items_lane);
    const double inv_items = n_items > 0 ? 1.0 / n_items : 0.0;
    lin0 = wave_max(lin0);

    double z[7] = {0, 0, 0, 0, 0, 0, 0};
    double rhoPi = 1.0;
    int status = 2, it = 0;

#ifdef MPC_PHASE_TIMING
    long long tacc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    MPC_T0();
    for (it = 0;; it++) {
        // ---- complementarity measures ----
        double msum = 0.0, cmax = 0.0;
#pragma unroll
        for (int k = 0; k < NB; k++) if (vb[k]) {
            const double a = ll[k] * tl[k], b = lh[k] * th[k];
            msum += a + b;
            if (!(tl[k] <= 2 * kTLMin || ll[k] <= 2 * kTLMin)) cmax = fmax(cmax, a);
            if (!(th[k] <= 2 * kTLMin || lh[k] <= 2 * kTLMin)) cmax = fmax(cmax, b);
        }
        if (vs) {
#pragma unroll
            for (int j = 0; j < NOBST; j++) {
                const double a = l1[j] * t1[j];
                msum += a;
                if (!(t1[j] <= 2 * kTLMin || l1[j] <= 2 * kTLMin)) cmax = fmax(cmax, a);
                if (soft) {
                    const double b = l2[j] * t2[j];
                    msum += b;
                    if (!(t2[j] <= 2 * kTLMin || l2[j] <= 2 * kTLMin)) cmax = fmax(cmax, b);
                }
            }
        }
        msum = wave_sum(msum);
        cmax = wave_max(cmax);
        const double mu = msum * inv_items;
        const double lin = rhoPi * lin0;
        if (!(mu == mu) || !(fabs(mu) <= 1e300)) { status = 4; break; }
        if (lin <= p.tol && cmax <= p.tol) { status = 0; break; }
        if (it >= p.iter_max) { status = 2; break; }
        MPC_TICK(0);

        // ---- predictor (sigma = 0): local gradient, barrier terms, reduced Hessian ----
        double Hq[8] = {Hd[0], Hd[1], Hd[2], Hd[3], Hd[4], Hd[5], Hd[6], 0.0};   // diagonal in z order, then Qxy
        double gloc[7], cb[7];                                             // (H z + q - C'lam), sum_c c beta_c
#pragma unroll
        for (int c = 0; c < 7; c++) { gloc[c] = Hd[c] * z[c] + q[c]; cb[c] = 0.0; }
        double rdl[NB], rdh[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            rdl[k] = rdh[k] = 0.0;
            if (vb[k]) {
                const double zk = z[zidx[k]];
                rdl[k] = (cl[k] + zk) - tl[k]; rdh[k] = (ch[k] - zk) - th[k];
                const double wl = ll[k] * rtl[k], wh = lh[k] * rth[k];
                const double bl = (ll[k] * tl[k] + ll[k] * rdl[k]) * rtl[k], bh = (lh[k] * th[k] + lh[k] * rdh[k]) * rth[k];
                Hq[zidx[k]] += wl + wh;
                gloc[zidx[k]] += lh[k] - ll[k];
                cb[zidx[k]] += bl - bh;
            }
        }
        double w1[NOBST], w2[NOBST], rD[NOBST], be1[NOBST], be2[NOBST], rs_[NOBST], rd1[NOBST], rd2[NOBST];
#pragma unroll
        for (int j = 0; j < NOBST; j++) {
            w1[j] = w2[j] = rD[j] = be1[j] = be2[j] = rs_[j] = rd1[j] = rd2[j] = 0.0;
            if (vs) {
                const double y = ax[j] * z[2] + ay[j] * z[3];
                w1[j] = l1[j] * rt1[j];
                double weff, geff;
                if (soft) {
                    rd1[j] = (hh[j] + y + sv[j]) - t1[j]; rd2[j] = sv[j] - t2[j];
                    be1[j] = (l1[j] * t1[j] + l1[j] * rd1[j]) * rt1[j];
                    w2[j] = l2[j] * rt2[j];
                    be2[j] = (l2[j] * t2[j] + l2[j] * rd2[j]) * rt2[j];
                    rs_[j] = zpen * sv[j] + zpen - l1[j] - l2[j];
                    const double D = zpen + w1[j] + w2[j];
                    rD[j] = 1.0 / D;
                    weff = w1[j] * (zpen + w2[j]) * rD[j];
                    geff = (be1[j] * (zpen + w2[j]) - w1[j] * (rs_[j] + be2[j])) * rD[j];
                } else {
                    rd1[j] = (hh[j] + y) - t1[j];
                    be1[j] = (l1[j] * t1[j] + l1[j] * rd1[j]) * rt1[j];
                    weff = w1[j]; geff = be1[j];
                }
                Hq[2] += weff * ax[j] * ax[j]; Hq[3] += weff * ay[j] * ay[j]; Hq[7] += weff * ax[j] * ay[j];
                gloc[2] -= l1[j] * ax[j]; gloc[3] -= l1[j] * ay[j];
                cb[2] += geff * ax[j]; cb[3] += geff * ay[j];
            }
        }
        if (act) {
#pragma unroll
            for (int c = 0; c < 8; c++) L.HQ[8 * i + c] = Hq[c];
            L.GQ[7 * i + 0] = gloc[0] + cb[0]; L.GQ[7 * i + 1] = gloc[1] + cb[1];
#pragma unroll
            for (int c = 0; c < 5; c++) { L.GQ[7 * i + 2 + c] = cb[2 + c]; L.GX[5 * i + c] = gloc[2 + c]; }
        }
        __syncthreads();
        MPC_TICK(1);
        factor_sweep(L, N, lane, dt, h2, rhoPi);
        __syncthreads();
        MPC_TICK(2);
        forward_rollout<true>(L, N, lane, dt, h2, d0, rhoPi);
        __syncthreads();
        MPC_TICK(3);
        double za[7] = {0, 0, 0, 0, 0, 0, 0};
        if (act) {
#pragma unroll
            for (int c = 0; c < 7; c++) za[c] = L.ZH[7 * i + c];
        }
        // affine step: dt, dlam per row, step ratios, products
        double rmax = 0.0, maff = 0.0;
        double dtl_[NB], dth_[NB], dll_[NB], dlh_[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            dtl_[k] = dth_[k] = dll_[k] = dlh_[k] = 0.0;
            if (vb[k]) {
                const double dzk = za[zidx[k]];
                dtl_[k] = dzk + rdl[k]; dth_[k] = -dzk + rdh[k];
                dll_[k] = -(ll[k] * tl[k] + ll[k] * dtl_[k]) * rtl[k]; dlh_[k] = -(lh[k] * th[k] + lh[k] * dth_[k]) * rth[k];
                ppl[k] = dll_[k] * dtl_[k]; pph[k] = dlh_[k] * dth_[k];
                rmax = fmax(rmax, fmax(-dtl_[k] * rtl[k], -dth_[k] * rth[k]));
                rmax = fmax(rmax, fmax(-dll_[k] * rcp_nr(ll[k]), -dlh_[k] * rcp_nr(lh[k])));
            }
        }
        double dt1_[NOBST], dl1_[NOBST], dt2_[NOBST], dl2_[NOBST], ds_[NOBST];
#pragma unroll
        for (int j = 0; j < NOBST; j++) {
            dt1_[j] = dl1_[j] = dt2_[j] = dl2_[j] = ds_[j] = 0.0;
            if (vs) {
                const double y = ax[j] * za[2] + ay[j] * za[3];
                if (soft) {
                    const double rsum = rs_[j] + be1[j] + be2[j];
                    ds_[j] = -(rsum + w1[j] * y) * rD[j];
                    dt1_[j] = rd1[j] + (y * (zpen + w2[j]) - rsum) * rD[j];     // y + ds without cancellation
                    dt2_[j] = rd2[j] + ds_[j];
                    dl2_[j] = -(l2[j] * t2[j] + l2[j] * dt2_[j]) * rt2[j];
                    pp2[j] = dl2_[j] * dt2_[j];
                    rmax = fmax(rmax, fmax(-dt2_[j] * rt2[j], -dl2_[j] * rcp_nr(l2[j])));
                } else dt1_[j] = rd1[j] + y;
                dl1_[j] = -(l1[j] * t1[j] + l1[j] * dt1_[j]) * rt1[j];
                pp1[j] = dl1_[j] * dt1_[j];
                rmax = fmax(rmax, fmax(-dt1_[j] * rt1[j], -dl1_[j] * rcp_nr(l1[j])));
            }
        }
        rmax = wave_max(rmax);
        const double a_aff = rmax > 1.0 ? 1.0 / rmax : 1.0;
#pragma unroll
        for (int k = 0; k < NB; k++) if (vb[k])
            maff += (ll[k] + a_aff * dll_[k]) * (tl[k] + a_aff * dtl_[k]) + (lh[k] + a_aff * dlh_[k]) * (th[k] + a_aff * dth_[k]);
        if (vs) {
#pragma unroll
            for (int j = 0; j < NOBST; j++) {
                maff += (l1[j] + a_aff * dl1_[j]) * (t1[j] + a_aff * dt1_[j]);
                if (soft) maff += (l2[j] + a_aff * dl2_[j]) * (t2[j] + a_aff * dt2_[j]);
            }
        }
        maff = wave_sum(maff) * inv_items;
        double sigma = mu > 0 ? maff / mu : 0.0;
        sigma = sigma * sigma * sigma;
        if (sigma > 1.0) sigma = 1.0;
        const double smu = sigma * mu;
        MPC_TICK(4);

        // ---- corrector: homogeneous system for the change of right-hand side, d beta_c = (dlam_aff dt_aff - sigma mu) / t ----
        double gc[7] = {0, 0, 0, 0, 0, 0, 0};
        double db1[NOBST], db2[NOBST];
#pragma unroll
        for (int k = 0; k < NB; k++) if (vb[k]) {
            const double dbl = (ppl[k] - smu) * rtl[k], dbh = (pph[k] - smu) * rth[k];
            gc[zidx[k]] += dbl - dbh;
        }
#pragma unroll
        for (int j = 0; j < NOBST; j++) {
            db1[j] = db2[j] = 0.0;
            if (vs) {
                db1[j] = (pp1[j] - smu) * rt1[j];
                double geff;
                if (soft) {
                    db2[j] = (pp2[j] - smu) * rt2[j];
                    geff = (db1[j] * (zpen + w2[j]) - w1[j] * db2[j]) * rD[j];
                } else geff = db1[j];
                gc[2] += geff * ax[j]; gc[3] += geff * ay[j];
            }
        }
        if (act) {
#pragma unroll
            for (int c = 0; c < 7; c++) L.GQ[7 * i + c] = gc[c];
        }
        __syncthreads();
        MPC_TICK(5);
        corrector_sweep(L, N, lane, dt, h2);
        __syncthreads();
        MPC_TICK(6);
        forward_rollout<false>(L, N, lane, dt, h2, d0, 0.0);
        __syncthreads();
        MPC_TICK(7);
        double dz[7] = {0, 0, 0, 0, 0, 0, 0};
        if (act) {
#pragma unroll
            for (int c = 0; c < 7; c++) dz[c] = za[c] + L.ZH[7 * i + c];
        }
        // ---- combined step ----
        rmax = 0.0;
#pragma unroll
        for (int k = 0; k < NB; k++) if (vb[k]) {
            const double dzk = dz[zidx[k]];
            dtl_[k] = dzk + rdl[k]; dth_[k] = -dzk + rdh[k];
            dll_[k] = -(ll[k] * tl[k] - smu + ppl[k] + ll[k] * dtl_[k]) * rtl[k];
            dlh_[k] = -(lh[k] * th[k] - smu + pph[k] + lh[k] * dth_[k]) * rth[k];
            rmax = fmax(rmax, fmax(-dtl_[k] * rtl[k], -dth_[k] * rth[k]));
            rmax = fmax(rmax, fmax(-dll_[k] * rcp_nr(ll[k]), -dlh_[k] * rcp_nr(lh[k])));
        }
#pragma unroll
        for (int j = 0; j < NOBST; j++) if (vs) {
            const double y = ax[j] * dz[2] + ay[j] * dz[3];
            if (soft) {
                const double rsum = rs_[j] + (be1[j] + db1[j]) + (be2[j] + db2[j]);
                ds_[j] = -(rsum + w1[j] * y) * rD[j];
                dt1_[j] = rd1[j] + (y * (zpen + w2[j]) - rsum) * rD[j];
                dt2_[j] = rd2[j] + ds_[j];
                dl2_[j] = -(l2[j] * t2[j] - smu + pp2[j] + l2[j] * dt2_[j]) * rt2[j];
                rmax = fmax(rmax, fmax(-dt2_[j] * rt2[j], -dl2_[j] * rcp_nr(l2[j])));
            } else dt1_[j] = rd1[j] + y;
            dl1_[j] = -(l1[j] * t1[j] - smu + pp1[j] + l1[j] * dt1_[j]) * rt1[j];
            rmax = fmax(rmax, fmax(-dt1_[j] * rt1[j], -dl1_[j] * rcp_nr(l1[j])));
        }
        rmax = wave_max(rmax);
        const double amax = rmax > 1.0 ? 1.0 / rmax : 1.0;
        const double alpha = (amax >= 1.0) ? 1.0 : 0.995 * amax;
#ifndef MPC_PHASE_TIMING
        if (p.trace && lane == 0) {
            double *tr = p.trace + ((size_t)inst * p.iter_max + it) * 4;
            tr[0] = mu; tr[1] = sigma; tr[2] = alpha; tr[3] = cmax;
        }
#endif
        if (!(alpha > 1e-14)) { status = 4; break; }
        // ---- update ----
#pragma unroll
        for (int c = 0; c < 7; c++) z[c] += alpha * dz[c];
#pragma unroll
        for (int k = 0; k < NB; k++) if (vb[k]) {
            tl[k] = fmax(tl[k] + alpha * dtl_[k], kTLMin); th[k] = fmax(th[k] + alpha * dth_[k], kTLMin);
            ll[k] = fmax(ll[k] + alpha * dll_[k], kTLMin); lh[k] = fmax(lh[k] + alpha * dlh_[k], kTLMin);
            rtl[k] = rcp_nr(tl[k]); rth[k] = rcp_nr(th[k]);
        }
        if (vs) {
#pragma unroll
            for (int j = 0; j < NOBST; j++) {
                t1[j] = fmax(t1[j] + alpha * dt1_[j], kTLMin); l1[j] = fmax(l1[j] + alpha * dl1_[j], kTLMin);
                rt1[j] = rcp_nr(t1[j]);
                if (soft) {
                    sv[j] += alpha * ds_[j];
                    t2[j] = fmax(t2[j] + alpha * dt2_[j], kTLMin); l2[j] = fmax(l2[j] + alpha * dl2_[j], kTLMin);
                    rt2[j] = rcp_nr(t2[j]);
                }
            }
        }
        rhoPi *= (1.0 - alpha);
        MPC_TICK(8);
    }
#ifdef MPC_PHASE_TIMING
    if (p.trace && lane == 0) { for (int k = 0; k < 10; k++) p.trace[((size_t)inst * p.iter_max) * 4 + k] = (double)tacc_[k]; }
#endif

    // ---- full step on the iterate; status 4 leaves it unchanged ----
    if (status != 4) {
        if (act) {
#pragma unroll
            for (int c = 0; c < 5; c++) { xi[c] += z[2 + c]; Xg[i * 5 + c] = xi[c]; }
        }
        if (has_u) { ui[0] += z[0]; ui[1] += z[1]; Ug[i * 2] = ui[0]; Ug[i * 2 + 1] = ui[1]; }
    }
    if (i == 0 && p.u0) { p.u0[(size_t)inst * 2] = ui[0]; p.u0[(size_t)inst * 2 + 1] = ui[1]; }
    // NLP objective at the returned iterate: LS cost + exact penalty of the obstacle violation
    if (p.cost) {
        double J = 0.0;
        if (act) {
            const double ex = xi[0] - gl[0], ey = xi[1] - gl[1];
            if (has_u) J = 0.5 * (p.Wg[0] * ex * ex + p.Wg[1] * ey * ey + p.Wg[2] * xi[3] * xi[3] + p.Wg[3] * xi[4] * xi[4]
                                  + p.Wg[4] * ui[0] * ui[0] + p.Wg[5] * ui[1] * ui[1]);
            else J = 0.5 * (p.Weg[0] * ex * ex + p.Weg[1] * ey * ey + p.Weg[2] * xi[3] * xi[3] + p.Weg[3] * xi[4] * xi[4]);
#pragma unroll
            for (int j = 0; j < NOBST; j++) {
                const double dx = xi[0] - Pg[2 * j], dy = xi[1] - Pg[2 * j + 1];
                const double hv = dx * dx + dy * dy - p.r2;
                const double v = hv < 0 ? -hv : 0.0;
                J += zpen * (v + 0.5 * v * v);
            }
        }
        J = wave_sum(J);
        if (lane == 0) p.cost[inst] = J;
    }
    if (lane == 0) {
        if (p.status) p.status[inst] = status;
        if (p.iters) p.iters[inst] = it;
    }
}

}  // namespace mpc
