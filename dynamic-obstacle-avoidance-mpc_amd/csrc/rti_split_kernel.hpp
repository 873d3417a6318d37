// rti_split_kernel.hpp -- the RTI solve kernel for SMALL batches (at most one instance per SIMD of the chip): one instance per
// wavefront, LPS LANES PER HORIZON STAGE (LPS = 3 for N <= 20, LPS = 2 for N <= 31).
//
// Same mathematics, same interior point method and the same row-parallel stage recursions as rti_solve_kernel<NOBST, G, 2>
// (rti_kernel.hpp; reference: src/simulation/robot_ocp_problem.py:126-132,145-166,186-198); what changes is who owns the
// inequality rows.  With one instance per wavefront a batch of 1024 fills the 1024 SIMDs of an MI355X with one wavefront
// each, and such a wavefront is bound by the length of its own instruction stream.  40 % of that stream
// are the lane-parallel "row phases" in which the lane that owns stage t walks through all 12 box rows and 2*NOBST obstacle
// rows of that stage while two thirds of the wavefront idle.  Here the rows of stage t are dealt out to LPS neighbouring
// lanes (lane = LPS * t + part):
//     box variables (ua, ual, x, y, v, om): 6 / LPS per lane (both sides of the box = 2 rows each),
//     obstacle rows: ceil(NOBST / LPS) pairs per lane (obstacle j = slot * LPS + part),
// so the row phases shrink by the factor LPS and the per-lane row state by as much (no AGPR round trips).  What the parts of
// a stage have to add up -- the barrier terms of the reduced Hessian and the gradients of the two Newton right-hand sides --
// travels to the stage's first lane (the "owner", which stages the stage's blocks in LDS) by one-lane DPP wave shifts
// (v_mov_b32_dpp wave_shl:1, once for the neighbour's value and twice for the next lane's) twice per interior-point iteration;
// the sums are formed in a fixed order, so the result is deterministic.  Everything a lane needs of the stage's Newton step it
// reads from the stage's LDS block (same address in the LPS lanes: a broadcast).
#pragma once
#include "rti_kernel.hpp"
#include <type_traits>

namespace mpc {

// W2 = false: dense stage blocks (RowLds) plus a result region of their own, 31 KB of LDS per wavefront at N = 20 -- the small-batch
//   variant, whose wavefront is alone on its SIMD anyway.
// W2 = true: TWO WAVEFRONTS PER SIMD for batches that are many rounds of wavefronts deep: compact stage blocks (RowLdsC), results
//   overlaying the consumed H~aug blocks (so every operand word is restaged per iteration) and the look-ahead staged in the same region:
//   14.7 KB per wavefront at N = 20, eight wavefronts per CU; the register allocator is held to 256 registers per lane.
template <int LPS, int NOBST, bool W2 = false, bool BLK2 = false>
struct SplitLds {
    using LT = typename std::conditional<W2, RowLdsC, RowLds>::type;
    static constexpr int NBL = 6 / LPS;                       // box variables per lane
    static constexpr int NSL = (NOBST + LPS - 1) / LPS;       // obstacle row pairs per lane
    static __host__ __device__ constexpr int results(int N) { return W2 ? 0 : (N + 1) * RowLds::HS; }     // result blocks of the sweeps, apart from the H~aug blocks
    static __host__ __device__ constexpr int total(int N, bool lookahead)
    {
        if (BLK2) return Blk2Lds::total(N) + (lookahead ? (N + 1) * NOBST * 2 : 0);
        return LT::total(N, 1) + results(N) + ((lookahead && !W2) ? (N + 1) * NOBST * 2 : 0);
    }
};

template <int K>
__device__ __forceinline__ double nth_of_six(double a0, double a1, double a2, double a3, double a4, double a5)
{
    if constexpr (K == 0) return a0; else if constexpr (K == 1) return a1; else if constexpr (K == 2) return a2;
    else if constexpr (K == 3) return a3; else if constexpr (K == 4) return a4; else return a5;
}

// Diagnostic build only (-DMPC_FORCE_WAVES2, scripts/waves2_experiment.py): the register allocator is told to fit TWO wavefronts per SIMD
// (256 registers per lane instead of 512).  What that costs is recorded in profiles/r02_waves2_experiment.json; the product is built without it.
#ifdef MPC_FORCE_WAVES2
#define MPC_SPLIT_BOUNDS(W2) __launch_bounds__(64, 2)
#else
#define MPC_SPLIT_BOUNDS(W2) __launch_bounds__(64, (W2) ? 2 : 1)
#endif
// MASKED: fewer obstacles than row pairs (p.n_obst < NOBST, see rti_solve_kernel): a template flag, because with the count known at compile time
// the per-slot row flags fold into the stage flags (measured: the run-time count costs 1 % at C2)
// BLK2: the stage recursions run on PAIRS of stages (Blk2Lds / rowpar_factor2, rti_kernel.hpp): even horizons, dense blocks, one wavefront per SIMD
template <int NOBST, int LPS, bool W2 = false, bool MASKED = false, bool BLK2 = false>
__global__ MPC_SPLIT_BOUNDS(W2) void rti_split_kernel(const KParams p)
{
    static_assert(LPS == 2 || LPS == 3, "two or three lanes per horizon stage");
    static_assert(!BLK2 || !W2, "the block-2 recursions exist on the dense layout only");
    using SL = SplitLds<LPS, NOBST, W2, BLK2>;
    using LT = typename SL::LT;
    constexpr int NBL = SL::NBL, NSL = SL::NSL;
    constexpr int kKK = 45;                   // free words 45, 46 of a stage block (RowVec uses 0..44, dead-store words start at RowLds::TAIL)
    const int lane = threadIdx.x;
    const int inst = p.order ? p.order[blockIdx.x] : (int)blockIdx.x;     // grid = batch: one instance per wavefront; instance scheduling: longest-running first
    const int N = p.N;
    const int nact = MASKED ? p.n_obst : NOBST;   // obstacles of the problem (<= NOBST, the row capacity this instantiation was built for)
    const int i = lane / LPS;                 // this lane's stage
    const int h = lane - i * LPS;             // ... and its part of the stage's rows
    const bool own = (h == 0);                // the part that stages the stage's blocks in LDS and stores the iterate
    const bool act = (i <= N);
    const bool has_u = (i < N);
    const bool xb = (i >= 1) && (i < N || (i == N && p.bx_terminal));
    const double dt = p.dt, h2 = p.h2;
    // part flags the optimiser cannot trace back to h (it would turn the select chains below into indexed loads from a stack array)
    int is_part1 = (h == 1), is_part2 = (h == 2);
    asm volatile("" : "+v"(is_part1), "+v"(is_part2));
    // value of the box variable k = part * NBL + s out of six values indexed by k (ua, ual, x, y, v, om); scalars, not an array:
    // the optimiser would turn selects between array elements into indexed loads from a stack copy
    auto part_of = [&](auto sc, double a0, double a1, double a2, double a3, double a4, double a5) {
        constexpr int s = decltype(sc)::value;
        double r = nth_of_six<s>(a0, a1, a2, a3, a4, a5);
        r = is_part1 ? nth_of_six<NBL + s>(a0, a1, a2, a3, a4, a5) : r;
        if (LPS == 3) r = is_part2 ? nth_of_six<(2 * NBL + s) % 6>(a0, a1, a2, a3, a4, a5) : r;
        return r;
    };
    // the same for vectors in z order (ua, ual, x, y, psi, v, om), whose box variables sit at 0, 1, 2, 3, 5, 6: compile-time slot, scalar
    // arguments (written with a run-time slot this became a table look-up in GLOBAL memory plus a seven-way select inside the iteration
    // loop; written on the array, a select between its elements becomes an indexed load from a stack copy)
    auto zpart_of = [&](auto sc, double v0, double v1, double v2, double v3, double v5, double v6) {
        return part_of(sc, v0, v1, v2, v3, v5, v6);
    };
    using slot0 = std::integral_constant<int, 0>; using slot1 = std::integral_constant<int, 1>; using slot2 = std::integral_constant<int, 2>;
    // the NBL slot values of a z-ordered vector for this lane
    auto slots_of = [&](const double (&v)[7], double (&out)[NBL]) {
        out[0] = zpart_of(slot0{}, v[0], v[1], v[2], v[3], v[5], v[6]);
        out[1] = zpart_of(slot1{}, v[0], v[1], v[2], v[3], v[5], v[6]);
        if constexpr (NBL > 2) out[2] = zpart_of(slot2{}, v[0], v[1], v[2], v[3], v[5], v[6]);
    };

#ifdef MPC_PHASE_TIMING
    long long tacc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    MPC_T0();
    // ---- load ----
    double x0v[5], gl[2];
#pragma unroll
    for (int c = 0; c < 5; c++) x0v[c] = p.x0[(size_t)inst * 5 + c];
    gl[0] = p.goal[(size_t)inst * 2]; gl[1] = p.goal[(size_t)inst * 2 + 1];
    // one instance per wavefront: its plant state and goal are the same in every lane.  In the 256-register build the 14 vector registers they hold to the end
    // of the kernel move to scalar ones (388 -> 308 B of scratch, +1.2 % at 8192); with 512 registers the scalar operands cost more than they free (-0.8 % at C2)
    if constexpr (W2) {
#pragma unroll
        for (int c = 0; c < 5; c++) x0v[c] = wave_uniform(x0v[c]);
        gl[0] = wave_uniform(gl[0]); gl[1] = wave_uniform(gl[1]);
    }
    double *Xg = p.X + (size_t)inst * (N + 1) * 5, *Ug = p.U + (size_t)inst * N * 2;
    const int ep_word = ((p.fused & kFuseMetrics) && p.ep_flags) ? p.ep_flags[inst] : 0;     // consumed after the look-ahead (no early wait)
    // the iterate: requested here, in front of the obstacle look-ahead, so that its global-memory latency passes behind that loop
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
    extern __shared__ double lds_raw[];
    const LT RL = [&]() {
        if constexpr (W2) return RowLdsC(lds_raw + RowLdsC::CT + RowLdsC::pad_front(N), N, lds_raw);
        else return RowLds(lds_raw + RowLds::pad_front(N), N, lds_raw + RowLds::total(N, 1));
    }();
    // block-2: blocks of stage pairs; the result blocks keep the dense layout (RV: what the vector sweeps walk, N / 2 + 1 blocks)
    const Blk2Lds BL(lds_raw, N);
    const RowLds RV(lds_raw, N / 2, BL.R);
    const int Mb = N >> 1;                    // blocks; stage i belongs to block i >> 1 (the terminal stage N = 2 Mb heads block Mb)
    const int mblk = i >> 1;
    const bool odd = (i & 1) != 0;
    // look-ahead staging: a region of its own, or (W2) the H~aug region, which is first written after the positions have been read
    double *lds_P = BLK2 ? lds_raw + Blk2Lds::total(N) : (W2 ? RL.H : lds_raw + LT::total(N, 1) + SL::results(N));
    // what part q of this lane's stage holds in the variable v: from_right shifts the whole wavefront by one lane, so the owner
    // (part 0) sees its neighbours' values; only the owner's result is meaningful
    auto of_part = [&](double v, int q) { return q == 0 ? v : (q == 1 ? from_right(v) : from_right(from_right(v))); };
    // obstacle positions of this lane's obstacle rows at its stage: explicit P (parameterize_model, robot_ocp_problem.py:154-166)
    // or the look-ahead computed here (Obstacle.predict_trajectory, src/utils/visualization.py:62-79)
    double pxy[NSL][2];
    if (p.obst) {
        if (lane < 2 * nact) {     // lane walks coordinate lane & 1 of obstacle lane >> 1 through the horizon
            const int j = lane >> 1, c = lane & 1;
            const double *o = p.obst + ((size_t)inst * nact + j) * 4;
            double q = o[c], v = (c == 0 && !p.world.bug_compat_predict) ? o[2] : o[3];      // defect D1: vx = self.vy (visualization.py:69)
            const double lo = c ? p.world.ymin : p.world.xmin, hi = c ? p.world.ymax : p.world.xmax;
            lds_P[lane] = q;
            for (int k = 1; k <= N; k++) {
                coord_advance(lo, hi, dt, q, v);
                lds_P[k * NOBST * 2 + lane] = q;
            }
        }
        wave_sync();
    }
    MPC_TICK(10);
    if (p.obst) {       // (two branches, not a select between an LDS and a global pointer: that would be a flat load)
#pragma unroll
        for (int s = 0; s < NSL; s++) {
            const int j = s * LPS + h, jj = j < nact ? j : nact - 1;
            const double *src = lds_P + ((act ? i : 0) * NOBST + jj) * 2;
            pxy[s][0] = src[0]; pxy[s][1] = src[1];
        }
    } else {
#pragma unroll
        for (int s = 0; s < NSL; s++) {
            const int j = s * LPS + h, jj = j < nact ? j : nact - 1;
            const double *src = p.P + (((size_t)inst * (N + 1) + (act ? i : 0)) * nact + jj) * 2;
            pxy[s][0] = src[0]; pxy[s][1] = src[1];
        }
    }
    const bool ep_done = (ep_word & 1) != 0;
    // Non-finite inputs must not pass as a converged solve (fmax() drops NaN, so the residual norm would not show them): one sum over
    // everything this lane read decides, the instance then fails at once (status 4) -- see rti_solve_kernel
    double fin = gl[0] + gl[1] + ui[0] + ui[1];
#pragma unroll
    for (int c = 0; c < 5; c++) fin += x0v[c] + xi[c] + xnext[c];
#pragma unroll
    for (int s = 0; s < NSL; s++) fin += pxy[s][0] + pxy[s][1];

    MPC_TICK(11);
    // ---- slack schedule, robot_ocp_problem.py:145-152 ----
    double zpen = 0.0;
    {
        const double ex = x0v[0] - gl[0], ey = x0v[1] - gl[1];
        const double scale = p.slack_a * (ex * ex + ey * ey + x0v[3] * x0v[3] + x0v[4] * x0v[4] + p.slack_b);
        const double alpha_i = p.alpha ? p.alpha[(size_t)inst * (N + 1) + (act ? i : N)] : scale * (double)(N - i) / (double)N;
        zpen = alpha_i * (has_u ? p.ss : 1.0);
    }
    const bool vs = act && (i >= 1) && (p.soft_h ? (zpen > 0.0) : true);
    const bool soft = p.soft_h != 0;

    // ---- linearise (every part of a stage computes the stage's linearisation: same instruction stream, no extra cost) ----
    double lin0 = 0.0;
    double d0[5] = {0, 0, 0, 0, 0};
    StageLin S;
    S.a02 = S.a03 = S.a04 = S.a12 = S.a13 = S.a14 = S.b00 = S.b01 = S.b10 = S.b11 = 0.0; S.dt = dt; S.h2 = h2;
    double bb[5] = {0, 0, 0, 0, 0};
    if (has_u) {
        double xn[5], ae[6], be[4];
        dyn_step<true>(xi, ui, dt, xn, ae, be);
        S.a02 = ae[0]; S.a03 = ae[1]; S.a04 = ae[2]; S.a12 = ae[3]; S.a13 = ae[4]; S.a14 = ae[5];
        S.b00 = be[0]; S.b01 = be[1]; S.b10 = be[2]; S.b11 = be[3];
#pragma unroll
        for (int c = 0; c < 5; c++) { bb[c] = xn[c] - xnext[c]; lin0 = fmax(lin0, fabs(bb[c])); }
    }
    if (i == 0) {
#pragma unroll
        for (int c = 0; c < 5; c++) { d0[c] = x0v[c] - xi[c]; lin0 = fmax(lin0, fabs(d0[c])); }
    }
    // ---- block-2: the pair's dynamics, once per solve ----
    // Sa, bba: linearisation and defect of the stage IN FRONT of an odd stage (its block partner a = i - 1), published by the owners and read by every lane
    // of the odd stage; even stages read their own (unused).  A^, B^, b^ of the pair are recomputed from (Sa, S) where needed (a few FMAs) rather than kept.
    StageLin Sa = S;
    double bba[5] = {0, 0, 0, 0, 0}, bh0[5] = {0, 0, 0, 0, 0};       // bh0: b^ = A_b b_a + b_b at rhoPi = 1 (odd owners)
    if constexpr (BLK2) {
        if (own && has_u) {     // [0] b2 [1] b3 [2] a02 [3] a03 [4] a04 [5] b0 [6] b00 [7] b01 | [8] b4 [9] - [10] a12 [11] a13 [12] a14 [13] b1 [14] b10 [15] b11
            double *w = BL.S + 16 * i;
            w[0] = bb[2]; w[1] = bb[3]; w[2] = S.a02; w[3] = S.a03; w[4] = S.a04; w[5] = bb[0]; w[6] = S.b00; w[7] = S.b01;
            w[8] = bb[4]; w[9] = 0.0; w[10] = S.a12; w[11] = S.a13; w[12] = S.a14; w[13] = bb[1]; w[14] = S.b10; w[15] = S.b11;
        }
        wave_sync();
        if (act && odd) {
            const double *w = BL.S + 16 * (i - 1);
            Sa.a02 = w[2]; Sa.a03 = w[3]; Sa.a04 = w[4]; Sa.b00 = w[6]; Sa.b01 = w[7];
            Sa.a12 = w[10]; Sa.a13 = w[11]; Sa.a14 = w[12]; Sa.b10 = w[14]; Sa.b11 = w[15];
            bba[0] = w[5]; bba[1] = w[13]; bba[2] = w[0]; bba[3] = w[1]; bba[4] = w[8];
        }
        if (own && odd && has_u) {      // W^ = [A^ b^ B^] rows 0..4, columns (x0..x4, 1, ua_a, ual_a, ua_b, ual_b); column 5 is rewritten every iteration
            // A^ = A_b A_a, B^ = [A_b B_a, B_b] with A = I + E (rows 0, 1 carry the six non-trivial entries, row 2 has dt at column 4), B = [b00 b01; b10 b11; 0 h2; dt 0; 0 dt]
            const double Wh[5][10] = {
                {1.0, 0.0, Sa.a02 + S.a02, Sa.a03 + S.a03, fma(S.a02, dt, Sa.a04) + S.a04, 0.0,
                 fma(S.a03, dt, Sa.b00), fma(S.a04, dt, fma(S.a02, h2, Sa.b01)), S.b00, S.b01},
                {0.0, 1.0, Sa.a12 + S.a12, Sa.a13 + S.a13, fma(S.a12, dt, Sa.a14) + S.a14, 0.0,
                 fma(S.a13, dt, Sa.b10), fma(S.a14, dt, fma(S.a12, h2, Sa.b11)), S.b10, S.b11},
                {0.0, 0.0, 1.0, 0.0, 2.0 * dt, 0.0, 0.0, fma(dt, dt, h2), 0.0, h2},
                {0.0, 0.0, 0.0, 1.0, 0.0, 0.0, dt, 0.0, dt, 0.0},
                {0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, dt, 0.0, dt}};
            double *w = BL.W + Blk2Lds::WS * mblk;
#pragma unroll
            for (int k = 0; k < 5; k++)
#pragma unroll
                for (int c = 0; c < 10; c++) w[k * 10 + c] = Wh[k][c];
            bh0[0] = fma(S.a04, bba[4], fma(S.a03, bba[3], fma(S.a02, bba[2], bba[0]))) + bb[0];
            bh0[1] = fma(S.a14, bba[4], fma(S.a13, bba[3], fma(S.a12, bba[2], bba[1]))) + bb[1];
            bh0[2] = fma(dt, bba[4], bba[2]) + bb[2]; bh0[3] = bba[3] + bb[3]; bh0[4] = bba[4] + bb[4];
        }
        if (own && act && (odd || i == N)) {      // H^ block of the pair (or of the terminal stage): the structural zeros once
            double *hc = BL.H + Blk2Lds::HS * mblk;
#pragma unroll
            for (int e = 0; e < 100; e++) hc[e] = 0.0;
        }
    } else
    if constexpr (W2) {
        // (the look-ahead positions staged in this region have been read above: LDS operations of a wavefront complete in order)
        if (own && act) { double *hc = RL.H + LT::HS * i; hc[49] = 0.0; hc[50] = 0.0; }      // rows 6, 7 of H~aug read these as their zeros; nothing else writes them
        if (lane < 8) {         // constant rows 2..4 of W~ for the columns other than the affine one: [3 j + m] = W~[2 + m][j]
            const double c2[8] = {0.0, 0.0, 1.0, 0.0, dt, 0.0, 0.0, h2}, c3[8] = {0.0, 0.0, 0.0, 1.0, 0.0, 0.0, dt, 0.0}, c4[8] = {0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, dt};
            double v2 = 0.0, v3 = 0.0, v4 = 0.0;
#pragma unroll
            for (int c = 0; c < 8; c++) if (lane == c) { v2 = c2[c]; v3 = c3[c]; v4 = c4[c]; }
            RL.C[3 * lane] = v2; RL.C[3 * lane + 1] = v3; RL.C[3 * lane + 2] = v4;
        }
        if (own && has_u) {     // rows 0, 1 of W~_t = [A b B]; b_t (words 5, 13, 16..18) is rewritten every iteration
            double *w = RL.W + LT::WS * i;
            const double Wrow[2][8] = {{1.0, 0.0, S.a02, S.a03, S.a04, 0.0, S.b00, S.b01}, {0.0, 1.0, S.a12, S.a13, S.a14, 0.0, S.b10, S.b11}};
#pragma unroll
            for (int k = 0; k < 2; k++)
#pragma unroll
                for (int c = 0; c < 8; c++) w[k * 8 + c] = Wrow[k][c];
        }
    } else {
    if (own && act) {           // H~aug_t: the structural zeros (and the constant psi diagonal) once; the 22 non-zeros are restaged every iteration
        double *hc = RL.H + LT::HS * i;
#pragma unroll
        for (int e = 0; e < 64; e++) hc[e] = 0.0;
    }
    if (own && has_u) {         // W~_t = [A b B] rows 0..4 (cols: x0..x4, b, ua, ual); column 5 is rewritten every iteration
        double *w = RL.W + LT::WS * i;
        const double Wrow[5][8] = {{1.0, 0.0, S.a02, S.a03, S.a04, 0.0, S.b00, S.b01}, {0.0, 1.0, S.a12, S.a13, S.a14, 0.0, S.b10, S.b11},
                                   {0.0, 0.0, 1.0, 0.0, dt, 0.0, 0.0, h2}, {0.0, 0.0, 0.0, 1.0, 0.0, 0.0, dt, 0.0}, {0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, dt}};
#pragma unroll
        for (int k = 0; k < 5; k++)
#pragma unroll
            for (int c = 0; c < 8; c++) w[k * 8 + c] = Wrow[k][c];
    }
    }

    MPC_TICK(12);
    // ---- this lane's box variables (slot s <-> variable k = part * NBL + s of ua, ual, x, y, v, om), in registers ----
    // cost: Gauss-Newton diagonal hd (+ LM) and gradient gc0 = W (y - yref) of the variable; robot_ocp_problem.py:59-83
    bool bp[NBL];
    double cl0[NBL], ch0[NBL], hq[NBL], hd[NBL], gc0[NBL];
    double ll[NBL], tl[NBL], lh[NBL], th[NBL], rtl[NBL], rth[NBL], zs[NBL];
    {
        auto slot_init = [&](auto sc) {     // slot index as a compile-time constant
            constexpr int s = decltype(sc)::value;
            const double val = part_of(sc, ui[0], ui[1], xi[0], xi[1], xi[3], xi[4]);
            const double lo = part_of(sc, p.bu_lo[0], p.bu_lo[1], p.bx_lo[0], p.bx_lo[1], p.bx_lo[2], p.bx_lo[3]);
            const double hi = part_of(sc, p.bu_hi[0], p.bu_hi[1], p.bx_hi[0], p.bx_hi[1], p.bx_hi[2], p.bx_hi[3]);
            const bool is_u = part_of(sc, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0) != 0.0;
            bp[s] = act && (is_u ? has_u : xb);
            hd[s] = part_of(sc, has_u ? p.Hd_stage[0] : 0.0, has_u ? p.Hd_stage[1] : 0.0, has_u ? p.Hd_stage[2] : p.Hd_term[0],
                            has_u ? p.Hd_stage[3] : p.Hd_term[1], has_u ? p.Hd_stage[5] : p.Hd_term[3], has_u ? p.Hd_stage[6] : p.Hd_term[4]);
            hq[s] = (is_u && !has_u) ? 1.0 : hd[s];       // the terminal stage has no inputs: unit block keeps Muu regular
            const double wg = part_of(sc, has_u ? p.Wg[4] : 0.0, has_u ? p.Wg[5] : 0.0, has_u ? p.Wg[0] : p.Weg[0], has_u ? p.Wg[1] : p.Weg[1],
                                      has_u ? p.Wg[2] : p.Weg[2], has_u ? p.Wg[3] : p.Weg[3]);
            gc0[s] = wg * (val - part_of(sc, 0.0, 0.0, gl[0], gl[1], 0.0, 0.0));
            cl0[s] = val - lo; ch0[s] = hi - val;
            tl[s] = fmax(cl0[s], p.thr0); th[s] = fmax(ch0[s], p.thr0);
            rtl[s] = rcp_nr(tl[s]); rth[s] = rcp_nr(th[s]);
            ll[s] = p.mu0 * rtl[s]; lh[s] = p.mu0 * rth[s];
            zs[s] = 0.0;
            if (bp[s]) lin0 = fmax(lin0, fmax(tl[s] - cl0[s], th[s] - ch0[s]));
        };
        slot_init(std::integral_constant<int, 0>{});
        slot_init(std::integral_constant<int, 1>{});
        if constexpr (NBL > 2) slot_init(std::integral_constant<int, 2>{});
    }
    const double hd_psi = has_u ? p.Hd_stage[4] : p.Hd_term[2];
    // ---- this lane's obstacle rows (slot s <-> obstacle j = s * LPS + part): rho1 = h + a'dx + s >= 0 (lam1, t1), rho2 = s >= 0
    //      (lam2, t2); robot_model.py:60-65 ----
    bool sp[NSL];
    double hh[NSL], ax[NSL], ay[NSL], sv[NSL], l1[NSL], t1[NSL], l2[NSL], t2[NSL], rt1[NSL], rt2[NSL];
#pragma unroll
    for (int s = 0; s < NSL; s++) {
        sp[s] = vs && (s * LPS + h < nact);
        const double ex = xi[0] - pxy[s][0], ey = xi[1] - pxy[s][1];
        hh[s] = ex * ex + ey * ey - p.r2; ax[s] = 2 * ex; ay[s] = 2 * ey;
        if (soft) {
            sv[s] = (hh[s] < 0 ? -hh[s] : 0.0) + p.thr0;
            t1[s] = fmax(hh[s] + sv[s], p.thr0);
            t2[s] = fmax(sv[s], p.thr0);
        } else {
            sv[s] = 0.0; t1[s] = fmax(hh[s], p.thr0); t2[s] = 1.0;
            if (sp[s]) lin0 = fmax(lin0, t1[s] - hh[s]);
        }
        rt1[s] = rcp_nr(t1[s]); rt2[s] = rcp_nr(t2[s]);
        l1[s] = p.mu0 * rt1[s]; l2[s] = soft ? p.mu0 * rt2[s] : 0.0;
    }
    int n_items_lane = 0;
#pragma unroll
    for (int s = 0; s < NBL; s++) n_items_lane += bp[s] ? 2 : 0;
#pragma unroll
    for (int s = 0; s < NSL; s++) n_items_lane += sp[s] ? (soft ? 2 : 1) : 0;
    const double n_items = seg_sum<64>((double)n_items_lane, lane);
    const double inv_items = wave_uniform(n_items > 0 ? 1.0 / n_items : 0.0);      // one instance per wavefront: per-instance scalars live in scalar registers
    if (!(fabs(fin) <= 1e300)) lin0 = INFINITY;
    lin0 = wave_uniform(seg_max<64>(lin0, lane));

    double z[7] = {0, 0, 0, 0, 0, 0, 0};
    double rhoPi = 1.0;
    int it = 0;
    IpmState ipm;             // rti_kernel.hpp: the interior point's scalar decisions are defined there, once
    ipm.running = !ep_done;
    int &status = ipm.status, &it_done = ipm.it_done;
    bool &running = ipm.running;
    float stepl = 0.0f;       // this lane's (stage's) last step norm, for the polish (ipm_polish_step)
    if (!(lin0 <= 1e300)) { status = 4; running = false; }

    // polish indicator (c) (rti_kernel.hpp: adjoint_inputs, ipm_head_g): the stationarity residual at the iterate.  Every lane forms its share of
    // g = H z + q - C' lam (its box variables, its obstacle rows), the shares travel to the stage's first lane as in the predictor (one-lane wave shifts, fixed
    // order), the open-loop adjoint sweep gives the input blocks, the slack equations are per row; max-norm over the wavefront.  About once per solve.
    auto stationarity = [&]() {
        double gsh[NBL], gk[6] = {0, 0, 0, 0, 0, 0}, sx = 0.0, sy = 0.0, rsm = 0.0;
#pragma unroll
        for (int s = 0; s < NBL; s++) { gsh[s] = gc0[s] + hd[s] * zs[s] + (bp[s] ? lh[s] - ll[s] : 0.0); gk[s] = gsh[s]; }
#pragma unroll
        for (int s = 0; s < NSL; s++) if (sp[s]) {
            sx -= l1[s] * ax[s]; sy -= l1[s] * ay[s];
            if (soft) rsm = fmax(rsm, fabs(zpen * sv[s] + zpen - l1[s] - l2[s]));
        }
        double shx = sx, shy = sy;
#pragma unroll
        for (int q = 1; q < LPS; q++) {
#pragma unroll
            for (int s = 0; s < NBL; s++) gsh[s] = from_right(gsh[s]);
            shx = from_right(shx); shy = from_right(shy);
#pragma unroll
            for (int s = 0; s < NBL; s++) gk[q * NBL + s] = gsh[s];
            sx += shx; sy += shy;
        }
        const double g[7] = {gk[0], gk[1], gk[2] + sx, gk[3] + sy, hd_psi * z[4], gk[4], gk[5]};      // (valid in the stage's first lane)
        const double ru = adjoint_inputs<64>(own && act, own && has_u, S, g, lane);
        return wave_uniform(seg_max<64>(fmax(ru, rsm), lane));
    };
    MPC_TICK(13);
    for (it = 0;; it++) {
        // ---- complementarity measures ----
        double msum = 0.0, cmax = 0.0;
#pragma unroll
        for (int s = 0; s < NBL; s++) if (bp[s]) {
            const double a = ll[s] * tl[s], b = lh[s] * th[s];
            msum += a + b;
            if (!(tl[s] <= 2 * p.tl_min || ll[s] <= 2 * p.tl_min)) cmax = fmax(cmax, a);
            if (!(th[s] <= 2 * p.tl_min || lh[s] <= 2 * p.tl_min)) cmax = fmax(cmax, b);
        }
#pragma unroll
        for (int s = 0; s < NSL; s++) if (sp[s]) {
            const double a = l1[s] * t1[s];
            msum += a;
            if (!(t1[s] <= 2 * p.tl_min || l1[s] <= 2 * p.tl_min)) cmax = fmax(cmax, a);
            if (soft) {
                const double b = l2[s] * t2[s];
                msum += b;
                if (!(t2[s] <= 2 * p.tl_min || l2[s] <= 2 * p.tl_min)) cmax = fmax(cmax, b);
            }
        }
        seg_reduce2<64, true>(msum, cmax, lane);
        const double mu = msum * inv_items;
        const double lin = rhoPi * lin0;
        ipm_head(p, ipm, it, mu, lin, cmax);
        if (__ballot(ipm.ask_g) != 0ull) ipm_head_g(p, ipm, it, stationarity());
        if (!running) break;      // wave-uniform: one instance per wavefront
        ipm.cprev = wave_uniform(cmax);
        MPC_TICK(0);

        // ---- predictor (sigma = 0): this lane's share of the local gradient, the barrier terms and the reduced Hessian ----
        double rdl[NBL], rdh[NBL];                  // residuals r_d = rho(z) - t of the box rows
        struct SoftT { double w1, w2, rD, be1, be2, rs, rd1, rd2; } so[NSL];
        double hdiag_[NBL], g_[NBL], ssum[5];       // this lane's share: per box variable (H diagonal, gradient); sxx, syy, sxy, sgx, sgy
        {
#pragma unroll
            for (int s = 0; s < NBL; s++) {
                rdl[s] = (cl0[s] + zs[s]) - tl[s];
                rdh[s] = (ch0[s] - zs[s]) - th[s];
                double hdiag = hq[s], g = gc0[s] + hd[s] * zs[s];           // (H z + q) of the variable
                if (bp[s]) {
                    const double wl = ll[s] * rtl[s], wh = lh[s] * rth[s];
                    const double bl = (ll[s] * tl[s] + ll[s] * rdl[s]) * rtl[s], bh = (lh[s] * th[s] + lh[s] * rdh[s]) * rth[s];
                    hdiag += wl + wh;
                    g += lh[s] - ll[s];                                     // - C'lam
                    g += bl - bh;                                           // sum_c c beta_c
                }
                hdiag_[s] = hdiag; g_[s] = g;
            }
            double sxx = 0.0, syy = 0.0, sxy = 0.0, glx = 0.0, gly = 0.0, cbx = 0.0, cby = 0.0;
#pragma unroll
            for (int s = 0; s < NSL; s++) {
                SoftT &o = so[s];
                const double y = ax[s] * z[2] + ay[s] * z[3];
                o.w1 = l1[s] * rt1[s];
                if (soft) {
                    o.rd1 = (hh[s] + y + sv[s]) - t1[s]; o.rd2 = sv[s] - t2[s];
                    o.be1 = (l1[s] * t1[s] + l1[s] * o.rd1) * rt1[s];
                    o.w2 = l2[s] * rt2[s];
                    o.be2 = (l2[s] * t2[s] + l2[s] * o.rd2) * rt2[s];
                    o.rs = zpen * sv[s] + zpen - l1[s] - l2[s];
                    o.rD = rcp_nr(zpen + o.w1 + o.w2);
                } else {
                    o.rd1 = (hh[s] + y) - t1[s]; o.rd2 = 0.0;
                    o.be1 = (l1[s] * t1[s] + l1[s] * o.rd1) * rt1[s];
                    o.w2 = 0.0; o.be2 = 0.0; o.rs = 0.0; o.rD = 0.0;
                }
                if (sp[s]) {
                    double weff, geff;
                    if (soft) {       // slack eliminated; cancellation-free forms
                        weff = o.w1 * (zpen + o.w2) * o.rD;
                        geff = (o.be1 * (zpen + o.w2) - o.w1 * (o.rs + o.be2)) * o.rD;
                    } else { weff = o.w1; geff = o.be1; }
                    sxx += weff * ax[s] * ax[s]; syy += weff * ay[s] * ay[s]; sxy += weff * ax[s] * ay[s];
                    glx -= l1[s] * ax[s]; gly -= l1[s] * ay[s];
                    cbx += geff * ax[s]; cby += geff * ay[s];
                }
            }
            ssum[0] = sxx; ssum[1] = syy; ssum[2] = sxy; ssum[3] = glx + cbx; ssum[4] = gly + cby;
        }
        MPC_TICK(1);
        double bbr[5], x_init[5];
#pragma unroll
        for (int c = 0; c < 5; c++) { bbr[c] = rhoPi * bb[c]; x_init[c] = rhoPi * d0[c]; }
        {   // the stage's values and sums in the owner lane, in a fixed order.  All first shifts, then all second shifts: a DPP read needs
            // two wait states after the VALU write of its source, which the other values' moves fill
            double Hk[6], gk[6], Ssum[5], sh_h[NBL], sh_g[NBL], sh_s[5];
#pragma unroll
            for (int s = 0; s < NBL; s++) { Hk[s] = hdiag_[s]; gk[s] = g_[s]; sh_h[s] = hdiag_[s]; sh_g[s] = g_[s]; }
#pragma unroll
            for (int e = 0; e < 5; e++) { Ssum[e] = ssum[e]; sh_s[e] = ssum[e]; }
#pragma unroll
            for (int q = 1; q < LPS; q++) {
#pragma unroll
                for (int s = 0; s < NBL; s++) { sh_h[s] = from_right(sh_h[s]); sh_g[s] = from_right(sh_g[s]); }
#pragma unroll
                for (int e = 0; e < 5; e++) sh_s[e] = from_right(sh_s[e]);
#pragma unroll
                for (int s = 0; s < NBL; s++) { Hk[q * NBL + s] = sh_h[s]; gk[q * NBL + s] = sh_g[s]; }
#pragma unroll
                for (int e = 0; e < 5; e++) Ssum[e] += sh_s[e];
            }
            const double Sxx = Ssum[0], Syy = Ssum[1], Sxy = Ssum[2], Sgx = Ssum[3], Sgy = Ssum[4];
            if constexpr (BLK2) {
                // the stage's cost in sparse form: Q = diag(q[0..4]) + qxy at (0, 1), gradient g[0..4], input block diag(r0, r1) with gradient (lu0, lu1)
                const double q[5] = {Hk[2] + Sxx, Hk[3] + Syy, hd_psi, Hk[4], Hk[5]};
                const double g[5] = {gk[2] + Sgx, gk[3] + Sgy, hd_psi * z[4], gk[4], gk[5]};
                const double r0 = Hk[0], r1 = Hk[1], lu0 = gk[0], lu1 = gk[1];
                // the even stage of a pair hands its 15 numbers to the odd stage's owner through the pair's (not yet used) result block
                if (own && act && !odd && i < N) {
                    double *x = BL.R + Blk2Lds::RS * mblk;
#pragma unroll
                    for (int c = 0; c < 5; c++) { x[c] = q[c]; x[5 + c] = g[c]; }
                    x[10] = Sxy; x[11] = r0; x[12] = r1; x[13] = lu0; x[14] = lu1;
                }
                if (own && i == N) {    // terminal stage: P~_N = H~aug_N[0..5][0..5]
                    double *hc = BL.H + Blk2Lds::HS * mblk;
#pragma unroll
                    for (int c = 0; c < 5; c++) { hc[11 * c] = q[c]; hc[10 * c + 5] = g[c]; hc[50 + c] = g[c]; }
                    hc[1] = Sxy; hc[10] = Sxy;
                }
                wave_sync();
                if (own && odd && act) {
                    const double *x = BL.R + Blk2Lds::RS * mblk;
                    double qa[5], ga[5];
#pragma unroll
                    for (int c = 0; c < 5; c++) { qa[c] = x[c]; ga[c] = x[5 + c]; }
                    const double qaxy = x[10], ra0 = x[11], ra1 = x[12], lua0 = x[13], lua1 = x[14];
                    // Y = C' Q C and y = C' (Q c + g) with C = [A_a B_a] (columns x0..x4, ua, ual), c = rhoPi b_a: the odd stage's cost seen from (x_a, u_a).
                    //   rows of C: r0 = (1, 0, a02, a03, a04, b00, b01), r1 = (0, 1, a12, a13, a14, b10, b11), r2 = (0, 0, 1, 0, dt, 0, h2), r3 = (0, 0, 0, 1, 0, dt, 0), r4 = (0, 0, 0, 0, 1, 0, dt)
                    //   Y_ij = r0_i s0_j + r1_i s1_j + q2 r2_i r2_j + q3 r3_i r3_j + q4 r4_i r4_j,   s0 = q0 r0 + qxy r1,  s1 = qxy r0 + q1 r1
                    const double C0[7] = {1.0, 0.0, Sa.a02, Sa.a03, Sa.a04, Sa.b00, Sa.b01}, C1[7] = {0.0, 1.0, Sa.a12, Sa.a13, Sa.a14, Sa.b10, Sa.b11};
                    double s0[7], s1[7];
                    s0[0] = q[0]; s0[1] = Sxy; s1[0] = Sxy; s1[1] = q[1];
#pragma unroll
                    for (int c = 2; c < 7; c++) { s0[c] = fma(q[0], C0[c], Sxy * C1[c]); s1[c] = fma(Sxy, C0[c], q[1] * C1[c]); }
                    double Y[7][7];
#pragma unroll
                    for (int c = 0; c < 7; c++) { Y[0][c] = s0[c]; Y[1][c] = s1[c]; }
#pragma unroll
                    for (int r = 2; r < 7; r++)
#pragma unroll
                        for (int c = r; c < 7; c++) Y[r][c] = fma(C0[r], s0[c], C1[r] * s1[c]);
                    const double q2dt = q[2] * dt, q2h2 = q[2] * h2, q3dt = q[3] * dt, q4dt = q[4] * dt;
                    Y[2][2] += q[2]; Y[2][4] += q2dt; Y[2][6] += q2h2;
                    Y[3][3] += q[3]; Y[3][5] += q3dt;
                    Y[4][4] += fma(q2dt, dt, q[4]); Y[4][6] += fma(q2dt, h2, q4dt);
                    Y[5][5] += q3dt * dt;
                    Y[6][6] += fma(q2h2, h2, q4dt * dt);
                    const double ca[5] = {rhoPi * bba[0], rhoPi * bba[1], rhoPi * bba[2], rhoPi * bba[3], rhoPi * bba[4]};
                    const double v[5] = {fma(q[0], ca[0], fma(Sxy, ca[1], g[0])), fma(Sxy, ca[0], fma(q[1], ca[1], g[1])), fma(q[2], ca[2], g[2]),
                                         fma(q[3], ca[3], g[3]), fma(q[4], ca[4], g[4])};
                    double y[7];
                    y[0] = v[0]; y[1] = v[1];
                    y[2] = fma(Sa.a02, v[0], fma(Sa.a12, v[1], v[2])); y[3] = fma(Sa.a03, v[0], fma(Sa.a13, v[1], v[3]));
                    y[4] = fma(Sa.a04, v[0], fma(Sa.a14, v[1], fma(dt, v[2], v[4])));
                    y[5] = fma(Sa.b00, v[0], fma(Sa.b10, v[1], dt * v[3])); y[6] = fma(Sa.b01, v[0], fma(Sa.b11, v[1], fma(h2, v[2], dt * v[4])));
                    // H^ in z^ order (x0..x4, 1, ua_a, ual_a, ua_b, ual_b): Y index 5, 6 (u_a) -> 6, 7; row-major 10 x 10, symmetric
                    double Hh[8][8];
                    const int zi[7] = {0, 1, 2, 3, 4, 6, 7};
#pragma unroll
                    for (int r = 0; r < 8; r++)
#pragma unroll
                        for (int c = 0; c < 8; c++) Hh[r][c] = 0.0;
#pragma unroll
                    for (int r = 0; r < 7; r++)
#pragma unroll
                        for (int c = r; c < 7; c++) { Hh[zi[r]][zi[c]] = Y[r][c]; Hh[zi[c]][zi[r]] = Y[r][c]; }
#pragma unroll
                    for (int c = 0; c < 5; c++) { Hh[c][c] += qa[c]; Hh[c][5] = ga[c] + y[c]; Hh[5][c] = Hh[c][5]; }
                    Hh[0][1] += qaxy; Hh[1][0] += qaxy;
                    Hh[6][6] += ra0; Hh[7][7] += ra1;
                    Hh[5][6] = lua0 + y[5]; Hh[6][5] = Hh[5][6]; Hh[5][7] = lua1 + y[6]; Hh[7][5] = Hh[5][7];
                    double *hc = BL.H + Blk2Lds::HS * mblk;
#pragma unroll
                    for (int r = 0; r < 8; r++)
#pragma unroll
                        for (int c = 0; c < 8; c++) hc[10 * r + c] = Hh[r][c];
                    hc[58] = lu0; hc[59] = lu1; hc[85] = lu0; hc[95] = lu1; hc[88] = r0; hc[99] = r1;
                    double *w = BL.W + Blk2Lds::WS * mblk;
#pragma unroll
                    for (int k = 0; k < 5; k++) w[10 * k + 5] = rhoPi * bh0[k];
                }
            } else
            if (own && act) {   // H~aug_t, dense 8 x 8, z~ order (x0..x4, 1, ua, ual); affine column of W~_t
                const double hxx = Hk[2] + Sxx, hyy = Hk[3] + Syy;
                const double gxs[5] = {gk[2] + Sgx, gk[3] + Sgy, hd_psi * z[4], gk[4], gk[5]};
                const double lu0 = gk[0], lu1 = gk[1];
                double *hc = RL.H + LT::HS * i;
                if constexpr (W2) {     // rows 0..5 of the block in full (the results of the last iteration overlay it), then H66, H77
                    const double Hrow[6][8] = {{hxx, Sxy, 0.0, 0.0, 0.0, gxs[0], 0.0, 0.0}, {Sxy, hyy, 0.0, 0.0, 0.0, gxs[1], 0.0, 0.0},
                                               {0.0, 0.0, hd_psi, 0.0, 0.0, gxs[2], 0.0, 0.0}, {0.0, 0.0, 0.0, Hk[4], 0.0, gxs[3], 0.0, 0.0},
                                               {0.0, 0.0, 0.0, 0.0, Hk[5], gxs[4], 0.0, 0.0}, {gxs[0], gxs[1], gxs[2], gxs[3], gxs[4], 0.0, lu0, lu1}};
#pragma unroll
                    for (int r = 0; r < 6; r++)
#pragma unroll
                        for (int c = 0; c < 8; c++) hc[r * 8 + c] = Hrow[r][c];
                    hc[48] = Hk[0]; hc[51] = Hk[1];
                    if (has_u) {
                        double *w = RL.W + LT::WS * i;
                        w[5] = bbr[0]; w[13] = bbr[1]; w[16] = bbr[2]; w[17] = bbr[3]; w[18] = bbr[4];
                    }
                } else {
                // row-major 8 x 8; everything else in the block is zero and stays zero
                hc[0] = hxx; hc[1] = Sxy; hc[8] = Sxy; hc[9] = hyy; hc[18] = hd_psi; hc[27] = Hk[4]; hc[36] = Hk[5]; hc[54] = Hk[0]; hc[63] = Hk[1];
#pragma unroll
                for (int c = 0; c < 5; c++) { hc[c * 8 + 5] = gxs[c]; hc[40 + c] = gxs[c]; }
                hc[46] = lu0; hc[47] = lu1; hc[53] = lu0; hc[61] = lu1;
                if (has_u) {
#pragma unroll
                    for (int k = 0; k < 5; k++) RL.W[LT::WS * i + k * 8 + 5] = bbr[k];
                }
                }
            }
        }
        wave_sync();
        MPC_TICK(9);
        // (the one-block asm variant rowpar_factor_fast saves 12 instructions per stage but claims 84 fixed registers: here, where the row
        // state lives in VGPRs next to the sweep, the extra AGPR round trips cost more than it gains -- measured 150.7 vs 148.0 us at C2)
        // K^ of the pair (rows u_a: 0, 1; u_b: 2, 3; column 5 = feed-forward) and, in the odd owner, the L D L' factors for the corrector
        double Kh[4][5], kh[4], fac[10];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            kh[u] = 0.0;
#pragma unroll
            for (int c = 0; c < 5; c++) Kh[u][c] = 0.0;
        }
#pragma unroll
        for (int e = 0; e < 10; e++) fac[e] = 0.0;
        StageFac F;
        F.i00 = 1.0; F.l = 0.0; F.i11 = 1.0; F.k0 = 0.0; F.k1 = 0.0;
#pragma unroll
        for (int c = 0; c < 5; c++) { F.K0[c] = 0.0; F.K1[c] = 0.0; }
        double za[7] = {0, 0, 0, 0, 0, 0, 0};
        // the affine part of x_b = A_a x_a + B_a u_a + c_a for the lanes of an odd stage (zero in the corrector's homogeneous solve)
        auto odd_state = [&](const double xa[5], double ua0, double ua1, const double ca[5], double xb_[5]) {
            xb_[0] = fma(Sa.b01, ua1, fma(Sa.b00, ua0, fma(Sa.a04, xa[4], fma(Sa.a03, xa[3], fma(Sa.a02, xa[2], xa[0]))))) + ca[0];
            xb_[1] = fma(Sa.b11, ua1, fma(Sa.b10, ua0, fma(Sa.a14, xa[4], fma(Sa.a13, xa[3], fma(Sa.a12, xa[2], xa[1]))))) + ca[1];
            xb_[2] = fma(h2, ua1, fma(dt, xa[4], xa[2])) + ca[2];
            xb_[3] = fma(dt, ua0, xa[3]) + ca[3];
            xb_[4] = fma(dt, ua1, xa[4]) + ca[4];
        };
        // the pair's step from the block's state: every lane of both stages evaluates it (same instruction stream)
        auto pair_step = [&](const double ca[5], double zz[7]) {
            const double *xx = BL.R + Blk2Lds::RS * mblk + RowVec::X;
            const double xa[5] = {xx[0], xx[1], xx[2], xx[3], xx[4]};
            double uu[4];
#pragma unroll
            for (int u = 0; u < 4; u++) uu[u] = fma(Kh[u][4], xa[4], fma(Kh[u][3], xa[3], fma(Kh[u][2], xa[2], fma(Kh[u][1], xa[1], fma(Kh[u][0], xa[0], kh[u])))));
            double xb_[5];
            odd_state(xa, uu[0], uu[1], ca, xb_);
#pragma unroll
            for (int c = 0; c < 5; c++) zz[2 + c] = odd ? xb_[c] : xa[c];
            zz[0] = odd ? uu[2] : uu[0]; zz[1] = odd ? uu[3] : uu[1];
            if (i == N) { zz[0] = 0.0; zz[1] = 0.0; }
            if (!act) {
#pragma unroll
                for (int c = 0; c < 7; c++) zz[c] = 0.0;
            }
        };
        double bbra[5];
#pragma unroll
        for (int c = 0; c < 5; c++) bbra[c] = rhoPi * bba[c];
        if constexpr (BLK2) {
            rowpar_factor2(lane, Mb, BL, lane < 16);
            wave_sync();
            if (has_u) {
                const double *ko = BL.R + Blk2Lds::RS * mblk;
#pragma unroll
                for (int u = 0; u < 4; u++) {
#pragma unroll
                    for (int c = 0; c < 5; c++) Kh[u][c] = ko[8 * u + c];
                    kh[u] = ko[8 * u + 5];
                }
                if (own && odd) {
#pragma unroll
                    for (int e = 0; e < 10; e++) fac[e] = ko[Blk2Lds::FAC + e];
                }
            }
            if (own && odd && has_u) {      // closed-loop pair: Acl^ = A^ + B^ K^ (row-major), c^ = rhoPi b^ + B^ k^; B^ rows 0, 1 recomputed, rows 2..4 are (0, h2 + dt^2, 0, h2), (dt, 0, dt, 0), (0, dt, 0, dt)
                const double B0[4] = {fma(S.a03, dt, Sa.b00), fma(S.a04, dt, fma(S.a02, h2, Sa.b01)), S.b00, S.b01};
                const double B1[4] = {fma(S.a13, dt, Sa.b10), fma(S.a14, dt, fma(S.a12, h2, Sa.b11)), S.b10, S.b11};
                const double A0[5] = {1.0, 0.0, Sa.a02 + S.a02, Sa.a03 + S.a03, fma(S.a02, dt, Sa.a04) + S.a04};
                const double A1[5] = {0.0, 1.0, Sa.a12 + S.a12, Sa.a13 + S.a13, fma(S.a12, dt, Sa.a14) + S.a14};
                const double h2d = fma(dt, dt, h2);
                double *acl = BL.R + Blk2Lds::RS * mblk + RowVec::ACL;
#pragma unroll
                for (int c = 0; c < 6; c++) {       // c = 5: the affine column (feed-forward)
                    const double k0 = c < 5 ? Kh[0][c] : kh[0], k1 = c < 5 ? Kh[1][c] : kh[1], k2 = c < 5 ? Kh[2][c] : kh[2], k3 = c < 5 ? Kh[3][c] : kh[3];
                    const double a0 = c < 5 ? A0[c] : rhoPi * bh0[0], a1 = c < 5 ? A1[c] : rhoPi * bh0[1];
                    const double a2 = c < 5 ? (c == 2 ? 1.0 : (c == 4 ? 2.0 * dt : 0.0)) : rhoPi * bh0[2];
                    const double a3 = c < 5 ? (c == 3 ? 1.0 : 0.0) : rhoPi * bh0[3], a4 = c < 5 ? (c == 4 ? 1.0 : 0.0) : rhoPi * bh0[4];
                    acl[0 * RowVec::RS + c] = fma(B0[3], k3, fma(B0[2], k2, fma(B0[1], k1, fma(B0[0], k0, a0))));
                    acl[1 * RowVec::RS + c] = fma(B1[3], k3, fma(B1[2], k2, fma(B1[1], k1, fma(B1[0], k0, a1))));
                    acl[2 * RowVec::RS + c] = fma(h2, k3, fma(h2d, k1, a2));
                    acl[3 * RowVec::RS + c] = fma(dt, k2, fma(dt, k0, a3));
                    acl[4 * RowVec::RS + c] = fma(dt, k3, fma(dt, k1, a4));
                }
            }
            MPC_TICK(2);
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < 5; c++) BL.R[RowVec::X + c] = x_init[c];
            }
            wave_sync();
            MPC_TICK(3);
            rowpar_vector_fast<true>(lane, Mb, RV, lane < 16);
            MPC_TICK(15);
            wave_sync();
            pair_step(bbra, za);
            MPC_TICK(3);
        } else {
#ifdef MPC_MFMA4
        if constexpr (!W2) mfma4_factor(lane, N, RL); else
#endif
        rowpar_factor(lane, N, RL, lane < 16);
        wave_sync();
        if (has_u) {
            const double *ko = RL.R + LT::HS * i;
#pragma unroll
            for (int c = 0; c < 5; c++) { F.K0[c] = ko[c]; F.K1[c] = ko[8 + c]; }
            F.k0 = ko[5]; F.k1 = ko[13]; F.i00 = ko[6]; F.l = ko[7]; F.i11 = ko[14];
        }
        // the parts of a stage must all have read the factors before the owner overwrites the block: LDS operations of one
        // wavefront complete in order, so no barrier is needed
        if (own && has_u) {     // closed-loop matrix Acl = A + B K, row-major, and c_t = r_b + B k, for the row-parallel vector recursions
            double *acl = RL.R + LT::HS * i + RowVec::ACL;
            const double Ar[2][5] = {{1.0, 0.0, S.a02, S.a03, S.a04}, {0.0, 1.0, S.a12, S.a13, S.a14}};
            const double Br[2][2] = {{S.b00, S.b01}, {S.b10, S.b11}};
#pragma unroll
            for (int c = 0; c < 5; c++) {
                acl[0 * RowVec::RS + c] = Ar[0][c] + Br[0][0] * F.K0[c] + Br[0][1] * F.K1[c];
                acl[1 * RowVec::RS + c] = Ar[1][c] + Br[1][0] * F.K0[c] + Br[1][1] * F.K1[c];
                acl[2 * RowVec::RS + c] = (c == 2 ? 1.0 : (c == 4 ? dt : 0.0)) + h2 * F.K1[c];
                acl[3 * RowVec::RS + c] = (c == 3 ? 1.0 : 0.0) + dt * F.K0[c];
                acl[4 * RowVec::RS + c] = (c == 4 ? 1.0 : 0.0) + dt * F.K1[c];
            }
            double *cc = acl + 5;
            cc[0 * RowVec::RS] = bbr[0] + S.b00 * F.k0 + S.b01 * F.k1; cc[1 * RowVec::RS] = bbr[1] + S.b10 * F.k0 + S.b11 * F.k1;
            cc[2 * RowVec::RS] = bbr[2] + h2 * F.k1; cc[3 * RowVec::RS] = bbr[3] + dt * F.k0; cc[4 * RowVec::RS] = bbr[4] + dt * F.k1;
        }
        MPC_TICK(2);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 5; c++) RL.R[RowVec::X + c] = x_init[c];
        }
        wave_sync();
        MPC_TICK(3);
        rowpar_vector_fast<true>(lane, N, RL, lane < 16);
        MPC_TICK(15);
        wave_sync();
        if (act) {
            const double *xx = RL.R + LT::HS * i + RowVec::X;
            double u0 = F.k0, u1 = F.k1;
#pragma unroll
            for (int c = 0; c < 5; c++) { za[2 + c] = xx[c]; u0 += F.K0[c] * xx[c]; u1 += F.K1[c] * xx[c]; }
            za[0] = u0; za[1] = u1;
        }
        MPC_TICK(3);

        }

        // ---- affine step: dt, dlam per row, step ratios, products dlam_aff * dt_aff ----
        double ppl[NBL], pph[NBL], pp1[NSL], pp2[NSL];
        double smu;
        {
            double rmax = 0.0, rmaxd = 0.0;      // largest -dt/t (primal) and -dlam/lam (dual) ratios
            double dtl_[NBL], dth_[NBL], dll_[NBL], dlh_[NBL], zas[NBL];
            slots_of(za, zas);
#pragma unroll
            for (int s = 0; s < NBL; s++) {
                const double dzk = zas[s];
                dtl_[s] = dzk + rdl[s]; dth_[s] = -dzk + rdh[s];
                dll_[s] = -(ll[s] * tl[s] + ll[s] * dtl_[s]) * rtl[s]; dlh_[s] = -(lh[s] * th[s] + lh[s] * dth_[s]) * rth[s];
                ppl[s] = dll_[s] * dtl_[s]; pph[s] = dlh_[s] * dth_[s];
                if (bp[s]) {
                    rmax = fmax(rmax, fmax(-dtl_[s] * rtl[s], -dth_[s] * rth[s]));
                    rmaxd = fmax(rmaxd, fmax(fma(dtl_[s], rtl[s], 1.0), fma(dth_[s], rth[s], 1.0)));      // -dlam/lam = 1 + dt/t when sigma = 0
                }
            }
            double dt1_[NSL], dl1_[NSL], dt2_[NSL], dl2_[NSL];
#pragma unroll
            for (int s = 0; s < NSL; s++) {
                const SoftT &o = so[s];
                const double y = ax[s] * za[2] + ay[s] * za[3];
                dt2_[s] = dl2_[s] = 0.0; pp2[s] = 0.0;
                if (soft) {
                    const double rsum = o.rs + o.be1 + o.be2;
                    const double ds = -(rsum + o.w1 * y) * o.rD;
                    dt1_[s] = o.rd1 + (y * (zpen + o.w2) - rsum) * o.rD;     // y + ds without cancellation
                    dt2_[s] = o.rd2 + ds;
                    dl2_[s] = -(l2[s] * t2[s] + l2[s] * dt2_[s]) * rt2[s];
                    pp2[s] = dl2_[s] * dt2_[s];
                    if (sp[s]) { rmax = fmax(rmax, -dt2_[s] * rt2[s]); rmaxd = fmax(rmaxd, fma(dt2_[s], rt2[s], 1.0)); }
                } else dt1_[s] = o.rd1 + y;
                dl1_[s] = -(l1[s] * t1[s] + l1[s] * dt1_[s]) * rt1[s];
                pp1[s] = dl1_[s] * dt1_[s];
                if (sp[s]) { rmax = fmax(rmax, -dt1_[s] * rt1[s]); rmaxd = fmax(rmaxd, fma(dt1_[s], rt1[s], 1.0)); }
            }
            seg_reduce2<64, false>(rmax, rmaxd, lane);
            double a_aff, a_affd;
            ipm_affine_steps(rmax, rmaxd, a_aff, a_affd);
            double maff = 0.0;
#pragma unroll
            for (int s = 0; s < NBL; s++) if (bp[s])
                maff += (ll[s] + a_affd * dll_[s]) * (tl[s] + a_aff * dtl_[s]) + (lh[s] + a_affd * dlh_[s]) * (th[s] + a_aff * dth_[s]);
#pragma unroll
            for (int s = 0; s < NSL; s++) if (sp[s]) {
                maff += (l1[s] + a_affd * dl1_[s]) * (t1[s] + a_aff * dt1_[s]);
                if (soft) maff += (l2[s] + a_affd * dl2_[s]) * (t2[s] + a_aff * dt2_[s]);
            }
            maff = seg_sum<64>(maff, lane) * inv_items;
            double sigma;
            smu = ipm_centring(maff, mu, cmax, sigma);
#ifndef MPC_PHASE_TIMING
            if (p.trace && lane == 0) {
                double *tr = p.trace + ((size_t)inst * p.iter_max + it) * 4;
                tr[0] = mu; tr[1] = sigma; tr[3] = cmax;
            }
#endif
        }
        MPC_TICK(4);

        // ---- corrector: homogeneous system for the change of right-hand side, d beta_c = (dlam_aff dt_aff - sigma mu) / t ----
        double gc[7];       // assembled in the owner lane only
        {
            double gcs[NBL];
#pragma unroll
            for (int s = 0; s < NBL; s++) {
                const double dbl = (ppl[s] - smu) * rtl[s], dbh = (pph[s] - smu) * rth[s];
                gcs[s] = bp[s] ? dbl - dbh : 0.0;
            }
            double sgx = 0.0, sgy = 0.0;
#pragma unroll
            for (int s = 0; s < NSL; s++) if (sp[s]) {
                const double db1 = (pp1[s] - smu) * rt1[s];
                double geff;
                if (soft) {
                    const SoftT &o = so[s];
                    const double db2 = (pp2[s] - smu) * rt2[s];
                    geff = (db1 * (zpen + o.w2) - o.w1 * db2) * o.rD;
                } else geff = db1;
                sgx += geff * ax[s]; sgy += geff * ay[s];
            }
            MPC_TICK(5);
            const int zidx[6] = {0, 1, 2, 3, 5, 6};
            gc[4] = 0.0;
            double Sgx = 0.0, Sgy = 0.0;
#pragma unroll
            for (int q = 0; q < LPS; q++) {
#pragma unroll
                for (int s = 0; s < NBL; s++) gc[zidx[q * NBL + s]] = of_part(gcs[s], q);
                const double vx = of_part(sgx, q), vy = of_part(sgy, q);
                Sgx = q == 0 ? vx : Sgx + vx; Sgy = q == 0 ? vy : Sgy + vy;
            }
            gc[2] += Sgx; gc[3] += Sgy;
            if constexpr (BLK2) {
                // the pair's right-hand side in (x_a, u_a, u_b): gc^_x = gc_a,x + A_a' gc_b,x,  gc^_ua = gc_a,u + B_a' gc_b,x,  gc^_ub = gc_b,u.
                // The even owner hands its seven numbers to the odd owner through free words of the pair's result block.
                constexpr int kGX = 50;
                if (own && act && !odd && i < N) {
                    double *x = BL.R + Blk2Lds::RS * mblk + kGX;
#pragma unroll
                    for (int c = 0; c < 7; c++) x[c] = gc[c];
                }
                if (own && i == N) {    // terminal stage: c~_N = gc_N,x
                    double *cc = BL.R + Blk2Lds::RS * mblk + RowVec::CT;
#pragma unroll
                    for (int c = 0; c < 5; c++) cc[c] = gc[2 + c];
                }
                wave_sync();
                double gu[4] = {0, 0, 0, 0};        // gc^_u in the order (ua_a, ual_a, ua_b, ual_b); odd owner
                if (own && odd && act) {
                    const double *x = BL.R + Blk2Lds::RS * mblk + kGX;
                    const double gb[5] = {gc[2], gc[3], gc[4], gc[5], gc[6]};
                    double gx[5];
                    gx[0] = x[2] + gb[0]; gx[1] = x[3] + gb[1];
                    gx[2] = x[4] + Sa.dpsi(gb); gx[3] = x[5] + Sa.dv(gb); gx[4] = x[6] + Sa.dom(gb);
                    gu[0] = x[0] + Sa.dua(gb); gu[1] = x[1] + Sa.dual(gb); gu[2] = gc[0]; gu[3] = gc[1];
                    double *cc = BL.R + Blk2Lds::RS * mblk + RowVec::CT;
#pragma unroll
                    for (int c = 0; c < 5; c++) cc[c] = fma(Kh[3][c], gu[3], fma(Kh[2][c], gu[2], fma(Kh[1][c], gu[1], fma(Kh[0][c], gu[0], gx[c]))));
                }
                wave_sync();
                rowpar_vector_fast<false>(lane, Mb, RV, lane < 16);
                wave_sync();
                if (own && odd && has_u) {      // feed-forward of the corrector: k^ = -Muu^-1 (gc^_u + B^' p_{a+2}),  B^' p = [B_a' (A_b' p); B_b' p]
                    const double *pp = BL.R + Blk2Lds::RS * (mblk + 1) + RowVec::P;
                    const double pv[5] = {pp[0], pp[1], pp[2], pp[3], pp[4]};
                    const double wv[5] = {pv[0], pv[1], S.dpsi(pv), S.dv(pv), S.dom(pv)};
                    const double ma0 = gu[0] + Sa.dua(wv), ma1 = gu[1] + Sa.dual(wv), mb0 = gu[2] + S.dua(pv), mb1 = gu[3] + S.dual(pv);
                    // substitution in the pivot order (u_b, u_a) with the stored factors: fac = 1/d0 1/d1 1/d2 1/d3 l10 l20 l30 l21 l31 l32
                    const double y0 = mb0, y1 = fma(-fac[4], y0, mb1), y2 = fma(-fac[7], y1, fma(-fac[5], y0, ma0)), y3 = fma(-fac[9], y2, fma(-fac[8], y1, fma(-fac[6], y0, ma1)));
                    const double k3 = -(y3 * fac[3]);
                    const double k2 = fma(-fac[9], k3, -(y2 * fac[2]));
                    const double k1 = fma(-fac[8], k3, fma(-fac[7], k2, -(y1 * fac[1])));
                    const double k0 = fma(-fac[6], k3, fma(-fac[5], k2, fma(-fac[4], k1, -(y0 * fac[0]))));
                    const double kc[4] = {k2, k3, k0, k1};      // (ua_a, ual_a, ua_b, ual_b)
                    // homogeneous pair dynamics: c^ = B^ k^; k^ itself for the lanes of both stages (free words 45..48 of the pair's block)
                    const double B0[4] = {fma(S.a03, dt, Sa.b00), fma(S.a04, dt, fma(S.a02, h2, Sa.b01)), S.b00, S.b01};
                    const double B1[4] = {fma(S.a13, dt, Sa.b10), fma(S.a14, dt, fma(S.a12, h2, Sa.b11)), S.b10, S.b11};
                    double *cc = BL.R + Blk2Lds::RS * mblk + RowVec::ACL + 5;
                    cc[0 * RowVec::RS] = fma(B0[3], kc[3], fma(B0[2], kc[2], fma(B0[1], kc[1], B0[0] * kc[0])));
                    cc[1 * RowVec::RS] = fma(B1[3], kc[3], fma(B1[2], kc[2], fma(B1[1], kc[1], B1[0] * kc[0])));
                    cc[2 * RowVec::RS] = fma(h2, kc[3], fma(dt, dt, h2) * kc[1]);
                    cc[3 * RowVec::RS] = fma(dt, kc[2], dt * kc[0]);
                    cc[4 * RowVec::RS] = fma(dt, kc[3], dt * kc[1]);
                    double *kk = BL.R + Blk2Lds::RS * mblk + kKK;       // words 45, 46, 47 and 49: word 48 (RowLds::TAIL) is where the idle lanes of the vector sweeps store
                    kk[0] = kc[0]; kk[1] = kc[1]; kk[2] = kc[2]; kk[4] = kc[3];
                }
            } else {
            if (own && act) {   // c~_t = gc_x + K' gc_u  (K = 0 in the terminal lane)
                double *cc = RL.R + LT::HS * i + RowVec::CT;
#pragma unroll
                for (int c = 0; c < 5; c++) cc[c] = gc[2 + c] + F.K0[c] * gc[0] + F.K1[c] * gc[1];
            }
            wave_sync();
            rowpar_vector_fast<false>(lane, N, RL, lane < 16);
            wave_sync();
            if (has_u) {        // feed-forward of the corrector right-hand side: k = -Muu^-1 (gc_u + B' p_{t+1}); owner lane (it has gc)
                const double *pp = RL.R + LT::HS * (i + 1) + RowVec::P;
                const double pv[5] = {pp[0], pp[1], pp[2], pp[3], pp[4]};
                const double m0 = gc[0] + S.dua(pv), m1 = gc[1] + S.dual(pv);
                F.k1 = fma(F.l, m0, -m1) * F.i11;
                F.k0 = fma(-F.l, F.k1, -(m0 * F.i00));
            }
            }
        }
        MPC_TICK(6);
        double dz[7] = {0, 0, 0, 0, 0, 0, 0};
        if constexpr (BLK2) {
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < 5; c++) BL.R[RowVec::X + c] = 0.0;
            }
            wave_sync();
            rowpar_vector_fast<true>(lane, Mb, RV, lane < 16);
            wave_sync();
            if (has_u) {
                const double *kk = BL.R + Blk2Lds::RS * mblk + kKK;
                kh[0] = kk[0]; kh[1] = kk[1]; kh[2] = kk[2]; kh[3] = kk[4];
            }
            const double zero5[5] = {0, 0, 0, 0, 0};
            pair_step(zero5, dz);
        } else {
        if (own && has_u) {     // homogeneous dynamics: c_t = B k; k itself for the other parts of the stage
            double *cc = RL.R + LT::HS * i + RowVec::ACL + 5;
            cc[0 * RowVec::RS] = S.b00 * F.k0 + S.b01 * F.k1; cc[1 * RowVec::RS] = S.b10 * F.k0 + S.b11 * F.k1;
            cc[2 * RowVec::RS] = h2 * F.k1; cc[3 * RowVec::RS] = dt * F.k0; cc[4 * RowVec::RS] = dt * F.k1;
            RL.R[LT::HS * i + kKK] = F.k0; RL.R[LT::HS * i + kKK + 1] = F.k1;
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 5; c++) RL.R[RowVec::X + c] = 0.0;
        }
        wave_sync();
        rowpar_vector_fast<true>(lane, N, RL, lane < 16);
        wave_sync();
        if (act) {
            const double *xx = RL.R + LT::HS * i + RowVec::X;
            double u0 = has_u ? RL.R[LT::HS * i + kKK] : 0.0, u1 = has_u ? RL.R[LT::HS * i + kKK + 1] : 0.0;
#pragma unroll
            for (int c = 0; c < 5; c++) { dz[2 + c] = xx[c]; u0 += F.K0[c] * xx[c]; u1 += F.K1[c] * xx[c]; }
            dz[0] = u0; dz[1] = u1;
        }
        }
#pragma unroll
        for (int c = 0; c < 7; c++) dz[c] += za[c];
        MPC_TICK(7);

        // ---- combined step: ratios, step length, update ----
        {
            double rmax = 0.0, rmaxd = 0.0;
            double dzs[NBL], dtl_[NBL], dth_[NBL], dll_[NBL], dlh_[NBL];
            slots_of(dz, dzs);
#pragma unroll
            for (int s = 0; s < NBL; s++) {
                dtl_[s] = dzs[s] + rdl[s]; dth_[s] = -dzs[s] + rdh[s];
                dll_[s] = -(ll[s] * tl[s] - smu + ppl[s] + ll[s] * dtl_[s]) * rtl[s];
                dlh_[s] = -(lh[s] * th[s] - smu + pph[s] + lh[s] * dth_[s]) * rth[s];
                if (bp[s]) {
                    rmax = fmax(rmax, fmax(-dtl_[s] * rtl[s], -dth_[s] * rth[s]));
                    rmaxd = fmax(rmaxd, fmax(-dll_[s] * rcp_nr(ll[s]), -dlh_[s] * rcp_nr(lh[s])));
                }
            }
            double dt1_[NSL], dl1_[NSL], dt2_[NSL], dl2_[NSL], ds_[NSL];
#pragma unroll
            for (int s = 0; s < NSL; s++) {
                const SoftT &o = so[s];
                const double y = ax[s] * dz[2] + ay[s] * dz[3];
                dt2_[s] = dl2_[s] = ds_[s] = 0.0;
                if (soft) {
                    const double db1 = (pp1[s] - smu) * rt1[s], db2 = (pp2[s] - smu) * rt2[s];
                    const double rsum = o.rs + (o.be1 + db1) + (o.be2 + db2);
                    ds_[s] = -(rsum + o.w1 * y) * o.rD;
                    dt1_[s] = o.rd1 + (y * (zpen + o.w2) - rsum) * o.rD;
                    dt2_[s] = o.rd2 + ds_[s];
                    dl2_[s] = -(l2[s] * t2[s] - smu + pp2[s] + l2[s] * dt2_[s]) * rt2[s];
                    if (sp[s]) { rmax = fmax(rmax, -dt2_[s] * rt2[s]); rmaxd = fmax(rmaxd, -dl2_[s] * rcp_nr(l2[s])); }
                } else dt1_[s] = o.rd1 + y;
                dl1_[s] = -(l1[s] * t1[s] - smu + pp1[s] + l1[s] * dt1_[s]) * rt1[s];
                if (sp[s]) { rmax = fmax(rmax, -dt1_[s] * rt1[s]); rmaxd = fmax(rmaxd, -dl1_[s] * rcp_nr(l1[s])); }
            }
            seg_reduce2<64, false>(rmax, rmaxd, lane);
            double alpha, alphad;
            ipm_step_lengths(rmax, rmaxd, alpha, alphad);
#ifndef MPC_PHASE_TIMING
            if (p.trace && lane == 0) p.trace[((size_t)inst * p.iter_max + it) * 4 + 2] = alpha;
#endif
            ipm_step_check(ipm, it, alpha, alphad, smu);
            ipm_polish_step<64>(p, ipm, lane, alpha, dz, stepl);
            if (running) {
#pragma unroll
                for (int c = 0; c < 7; c++) z[c] += alpha * dz[c];
#pragma unroll
                for (int s = 0; s < NBL; s++) {
                    zs[s] += alpha * dzs[s];
                    if (bp[s]) {
                        tl[s] = fmax(tl[s] + alpha * dtl_[s], p.tl_min); th[s] = fmax(th[s] + alpha * dth_[s], p.tl_min);
                        ll[s] = fmax(ll[s] + alphad * dll_[s], p.tl_min); lh[s] = fmax(lh[s] + alphad * dlh_[s], p.tl_min);
                        rtl[s] = rcp_nr(tl[s]); rth[s] = rcp_nr(th[s]);
                    }
                }
#pragma unroll
                for (int s = 0; s < NSL; s++) if (sp[s]) {
                    t1[s] = fmax(t1[s] + alpha * dt1_[s], p.tl_min); l1[s] = fmax(l1[s] + alphad * dl1_[s], p.tl_min);
                    rt1[s] = rcp_nr(t1[s]);
                    if (soft) {
                        sv[s] += alpha * ds_[s];
                        t2[s] = fmax(t2[s] + alpha * dt2_[s], p.tl_min); l2[s] = fmax(l2[s] + alphad * dl2_[s], p.tl_min);
                        rt2[s] = rcp_nr(t2[s]);
                    }
                }
                rhoPi = wave_uniform(rhoPi * (1.0 - alpha));
            }
        }
        MPC_TICK(8);
        if (!running) break;
    }

    if constexpr (!W2) {        // the plant state and the goal are read again here instead of being carried through the interior point (14 registers less across
        const double *xg = p.x0 + (size_t)inst * 5, *gg = p.goal + (size_t)inst * 2;      // the loop; the 256-register build keeps them in scalar registers)
        asm volatile("" : "+v"(xg), "+v"(gg));
#pragma unroll
        for (int c = 0; c < 5; c++) x0v[c] = xg[c];
        gl[0] = gg[0]; gl[1] = gg[1];
    }
    // ---- full step on the iterate; status 4 leaves it unchanged ----
    const bool store = !ep_done;
    status = ipm_finite_step<64>(status, z, lane);
    if (status != 4) {
#pragma unroll
        for (int c = 0; c < 5; c++) xi[c] += z[2 + c];
        ui[0] += z[0]; ui[1] += z[1];
    }
    const double u_apply[2] = {lane_value(ui[0], 0), lane_value(ui[1], 0)};   // u* = U[0]
    if ((p.fused & kFuseResetOnFail) && status == 4) {      // set_initial_guess(), robot_ocp_problem.py:203-205,286-306
        xi[0] = x0v[0]; xi[1] = x0v[1]; xi[2] = x0v[2]; xi[3] = 0.0; xi[4] = 0.0; ui[0] = ui[1] = 0.0;
        if (p.fused & kFuseInterpGuess) interp_guess(x0v, gl[1], i <= N ? i : N, N, xi);
    }
    if (store && own && (status != 4 || (p.fused & (kFuseResetOnFail | kFuseShift)))) {
        if (p.fused & kFuseShift) {                          // X[j] <- X[j+1], U[j] <- U[j+1], U[N-1] <- 0, X[N] kept (:253-258)
            if (act && i >= 1) {
#pragma unroll
                for (int c = 0; c < 5; c++) Xg[(i - 1) * 5 + c] = xi[c];
            }
            if (i == N) {
#pragma unroll
                for (int c = 0; c < 5; c++) Xg[N * 5 + c] = xi[c];
            }
            if (has_u && i >= 1) { Ug[(i - 1) * 2] = ui[0]; Ug[(i - 1) * 2 + 1] = ui[1]; }
            if (i == 0) { Ug[(N - 1) * 2] = 0.0; Ug[(N - 1) * 2 + 1] = 0.0; }
        } else {
            if (act) {
#pragma unroll
                for (int c = 0; c < 5; c++) Xg[i * 5 + c] = xi[c];
            }
            if (has_u) { Ug[i * 2] = ui[0]; Ug[i * 2 + 1] = ui[1]; }
        }
    }
    // ---- plant, obstacles, episode bookkeeping (fused closed-loop step) ----
    if (p.fused & (kFusePlant | kFuseObstacles | kFuseMetrics)) {
        double xp[5] = {x0v[0], x0v[1], x0v[2], x0v[3], x0v[4]};
        if ((p.fused & kFuseAliasBug) && (p.fused & kFuseResetOnFail) && status == 4) { xp[3] = 0.0; xp[4] = 0.0; }
        double xnew[5] = {xp[0], xp[1], xp[2], xp[3], xp[4]};
        if (p.fused & kFusePlant) dyn_step<false>(xp, u_apply, dt, xnew, nullptr, nullptr);     // every lane, same value
        if ((p.fused & kFusePlant) && lane == 0 && store && p.x0_rw) {
#pragma unroll
            for (int c = 0; c < 5; c++) p.x0_rw[(size_t)inst * 5 + c] = xnew[c];
        }
        double margin = INFINITY;
        if (p.obst && lane < nact) {                         // ground-truth motion of obstacle j = lane
            const double *o = p.obst + ((size_t)inst * nact + lane) * 4;
            double ox = o[0], oy = o[1], ovx = o[2], ovy = o[3];
            if (p.fused & kFuseObstacles) {
                if (p.noise) obstacle_noise(p.randomness, p.vmax, p.noise[((size_t)inst * nact + lane) * 2], p.noise[((size_t)inst * nact + lane) * 2 + 1], ovx, ovy);
                obstacle_advance(p.world, dt, ox, ovx, oy, ovy);
                if (store && p.obst_rw) { double *w = p.obst_rw + ((size_t)inst * nact + lane) * 4; w[0] = ox; w[1] = oy; w[2] = ovx; w[3] = ovy; }
            }
            const double ddx = xnew[0] - ox, ddy = xnew[1] - oy;
            margin = sqrt(ddx * ddx + ddy * ddy) - p.r_hit;  // :222-228
        }
        if (p.fused & kFuseMetrics) {
            margin = -seg_max<64>(-margin, lane);
            if (lane == 0 && store) {
                int fl = p.ep_flags[inst];
                if (xnew[0] < p.world.xmin || xnew[0] > p.world.xmax || xnew[1] < p.world.ymin || xnew[1] > p.world.ymax) fl |= 2;   // :213-214
                const double mm = fmin(p.ep_min_margin[inst], margin);
                p.ep_min_margin[inst] = mm;
                if (mm <= 0.0) fl |= 4;
                const double gx_ = xnew[0] - gl[0], gy_ = xnew[1] - gl[1];
                if (sqrt(gx_ * gx_ + gy_ * gy_) <= p.tol_goal) fl |= 1;      // :247-250: reached, the loop breaks before i += 1
                else p.ep_steps[inst] += 1;
                p.ep_flags[inst] = fl;
            }
        }
    }
    if (lane == 0 && p.u0 && store) { p.u0[(size_t)inst * 2] = u_apply[0]; p.u0[(size_t)inst * 2 + 1] = u_apply[1]; }
    // NLP objective at the returned iterate: LS cost (the stage's owner) + exact penalty of the obstacle violation (the rows' lanes)
    if (p.cost) {
        double J = 0.0;
        if (act) {
            if (own) {
                const double ex = xi[0] - gl[0], ey = xi[1] - gl[1];
                if (has_u) J = 0.5 * (p.Wg[0] * ex * ex + p.Wg[1] * ey * ey + p.Wg[2] * xi[3] * xi[3] + p.Wg[3] * xi[4] * xi[4]
                                      + p.Wg[4] * ui[0] * ui[0] + p.Wg[5] * ui[1] * ui[1]);
                else J = 0.5 * (p.Weg[0] * ex * ex + p.Weg[1] * ey * ey + p.Weg[2] * xi[3] * xi[3] + p.Weg[3] * xi[4] * xi[4]);
            }
#pragma unroll
            for (int s = 0; s < NSL; s++) if (s * LPS + h < nact) {
                const double dx = xi[0] - pxy[s][0], dy = xi[1] - pxy[s][1];
                const double hv = dx * dx + dy * dy - p.r2;
                const double v = hv < 0 ? -hv : 0.0;
                J += zpen * (v + 0.5 * v * v);
            }
        }
        J = seg_sum<64>(J, lane);
        if (lane == 0 && store) p.cost[inst] = J;
    }
    if (lane == 0 && store) {
        if (p.iters_acc) p.iters_acc[inst] += it_done;
        if (p.status_acc) p.status_acc[inst] += (status == 4 ? 1 : 0) + (status == 2 ? 65536 : 0);
        if (p.status) p.status[inst] = status;
        if (p.iters) p.iters[inst] = it_done;
    }
#ifdef MPC_PHASE_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    MPC_TICK(14);
    if (p.trace && lane == 0) { for (int k = 0; k < 16; k++) p.trace[((size_t)inst * p.iter_max) * 4 + k] = (double)tacc_[k]; }
#endif
}

}  // namespace mpc
