// rti_kernel.hpp -- the RTI solve kernel: one SQP_RTI iteration per MPC instance, gfx950 (MI355X).
//
// What it replaces: everything `ocp_solver.solve()` does in the reference
// (src/simulation/robot_ocp_problem.py:195, options :126-132), plus the per-step parameter uploads around it
// (:145-152 slack schedule, :154-166 obstacle parameters, :191-192 initial-state bounds).  See DESIGN.md.
//
// Mapping: G LANES PER INSTANCE (G = 64, 32 or 16 with N + 1 < G; one workgroup = one wavefront = 64/G instances that share
// one instruction stream).
//   * lane i owns horizon stage i (N+1 <= 64): its linearisation, its inequality rows (multiplier lam, slack t for the
//     4 input-box, 8 state-box and 2*NOBST soft-obstacle rows live in that lane's REGISTERS for the whole solve); everything
//     "per row" is lane-parallel over the stages;
//   * wavefront reductions (max step ratio, complementarity sum / max) are DPP butterflies;
//   * the stage recursions (backward Riccati, adjoint sweep, forward rollouts) are sequential in the stage index.  Two
//     implementations, selected by the template parameter FACT:
//       - ROW-PARALLEL (default): the algebra of ONE stage is spread over the 8 lanes of a 16-lane DPP row and runs as chains of
//         v_fmac_f64_dpp with row_newbcast operands; the stage operands travel through LDS (rowpar_factor, rowpar_vector);
//       - SYSTOLIC: the recursion state hops from lane to lane with one-lane DPP wave shifts and every stage is computed by the
//         lane that owns it, hand-expanded for the sparsity of A_i = I + E_i (6 non-trivial entries) and B_i (4); no LDS.
// HBM traffic is therefore the algorithmic minimum: read x0, goal, P, X, U once, write X, U, u0, cost, status once.
//
// Interior point method: Mehrotra predictor-corrector (separate primal / dual step lengths) in residual ("delta") form.  The costates the stationarity
// residual needs come from the adjoint recursion pi_i = (H z + q - C'lam)_x + A_i' pi_{i+1}, fused into the backward
// Riccati sweep (it zeroes the state blocks of the residual exactly).  The corrector solves only the homogeneous
// system for the difference of right-hand sides.  Dynamics / initial-condition residuals decay by prod(1 - alpha_k).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// Diagnostic builds (never the shipped library; each script compiles its own copy):
//   -DMPC_PHASE_TIMING   scripts/phase_timing.py      per-phase cycle counts into KParams::trace
//   -DMPC_COMPACT_PLAIN  scripts/asm_factor_check.py  compiler-scheduled factor sweep on the compact stage blocks instead of the one-block asm
//   -DMPC_MFMA4          scripts/mfma4_check.py       factor sweep on 4x4x4 matrix-core blocks (one instance per wavefront, dense blocks)
#ifdef MPC_PHASE_TIMING
#define MPC_T0() long long t_prev_ = clock64()
#define MPC_TICK(k) do { const long long t_now_ = clock64(); tacc_[k] += t_now_ - t_prev_; t_prev_ = t_now_; } while (0)
#else
#define MPC_T0() do {} while (0)
#define MPC_TICK(k) do {} while (0)
#endif

namespace mpc {

struct World {
    double xmin, xmax, ymin, ymax;
    int bug_compat_predict;
};

// fused closed-loop step: what the kernel does after the solve (bit flags of KParams::fused)
enum : int {
    kFuseShift = 1,        // store the iterate shifted by one stage (robot_ocp_problem.py:253-258)
    kFusePlant = 2,        // x0 <- F(x0, u*) in place (:207-212)
    kFuseObstacles = 4,    // obstacle states advance one step, optional velocity noise (visualization.py:20-33)
    kFuseResetOnFail = 8,  // status 4 -> set_initial_guess() before the plant step (:203-205)
    kFuseAliasBug = 16,    // ... which in the reference also zeroes the plant's v, omega (defect D2, :301-302)
    kFuseMetrics = 32,     // episode bookkeeping of RobotOcpProblem.step (:213-250)
    kFuseInterpGuess = 64  // ... and that set_initial_guess() is the straight-line variant the reference keeps commented out (:293-300, interp_guess below)
};

struct KParams {
    int N, batch;
    int n_obst;               // obstacles of the problem: 1 .. NOBST of the instantiation that runs it (rows beyond it do not exist)
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
    double polish_ratio;                 // polish of the interior point (kPolishMax), indicator (a); +inf = off (mpc_api.hip::make_params)
    float polish_tol;                    // ... indicator (b); +inf = off
    float polish_kappa;                  // floor of the step estimate as a fraction of the step (mpc_config.polish_step_frac)
    float polish_tol_unsolved;           // kPolishUnsolved x polish_tol: the estimate above which a solve that has used up its polish is reported as not converged (status 2)
    double polish_res_g;                 // ... indicator (c), round 6: the stationarity residual (adjoint_inputs) above which a polish iteration is taken; +inf = off
    double tl_min;                       // floor of t and lam: min(kTLMin, qp_tol / 10) (the floor must stay below the tolerance: an active row's rho - t is the floor)
    double mu_div, mu_cap, mu_settled;   // the divergence tests of the interior point as thresholds on mu (mpc_api.hip::make_params): kMuDiverged mu0, kMuCapFailed mu0, mu0 --
                                         // or, mpc_config.qp_fail_policy = 1 ("truncate"), 1e300 / inf / inf: a diverging solve runs to the iteration cap and ends as status 2
    const double *x0, *P, *goal;
    const double *alpha;      // optional [B][N+1]: explicit slack weights zl_i = Zl_i (mpc_set_slack_schedule); null = the schedule of robot_ocp_problem.py:145-148
    double *X, *U, *u0, *cost;
    int32_t *status, *iters;
    const int32_t *order;     // optional instance order (aux_kernels.hpp::schedule_kernel): wavefront slot s of the one-lane kernels processes instance order[s]
    int32_t *iters_acc, *status_acc;   // optional running sums over launches: IPM iterations; (status == 4) + 65536 * (status == 2)
    // ---- fused closed-loop step (all optional; see mpc_closed_loop_step_dev) ----
    const double *obst;       // [B][n_obst][4]: if set, the look-ahead P is computed in the kernel and p.P is ignored
    double *x0_rw, *obst_rw;  // in-place plant state / obstacle states
    const double *noise;      // [B][n_obst][2] standard normals or null
    double randomness, vmax, tol_goal, r_hit;
    World world;
    int fused;
    double *ep_min_margin;    // [B] running minimum of the margin to the obstacles
    int32_t *ep_flags;        // [B] bit0 reached goal (episode finished), bit1 left the arena, bit2 hit an obstacle
    int32_t *ep_steps;        // [B] completed control steps (the reference's `i`)
    double *trace;            // optional [batch][iter_max][4] = (mu, sigma, alpha, cmax) per IPM iteration (debug)
};

static constexpr double kTLMin = 1e-11;  // floor for lam and t (oracle/mpc_oracle.c TL_MIN: 1e-13 until round 3 -- the weights lam / t of collapsed pairs then cost the
                                         // end-game's Newton step its last digits: worst distance from the exact QP solution 7e-4 -> 8e-6 on first solves of C5's problem)
// Mehrotra constants shared with the oracle (oracle/mpc_oracle.c FRAC_TO_BOUNDARY, sigma): step = kFracToBoundary * (largest step that keeps t, lam > 0),
// centring sigma = (mu_aff / mu)^2.  Scanned on the oracle over three problem classes: 0.999995 / square needs 4 - 7 % fewer
// iterations than round 1's 0.9995 / cube at the same number of non-converged instances.
static constexpr double kFracToBoundary = 0.999995;
// mu of a healthy solve stays below ~1e2 mu0, that of an infeasible QP grows without bound: beyond kMuDiverged * mu0 the solve has failed (status 4)
static constexpr double kMuDiverged = 1e8;
// ... and a solve that reaches the iteration cap with mu above kMuCapFailed * mu0 was on its way there: status 4, not 2 (its step is not applied)
static constexpr double kMuCapFailed = 1e4;
static constexpr int kMuCapSettled = 20;       // ... and from this iteration on, above mu0 itself (oracle/mpc_oracle.c MU_CAP_SETTLED)

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

// One constant-velocity step with wall reflection, src/utils/visualization.py:35-59.  Plain IEEE operators in the
// reference's own operation order with FP contraction switched off for this block, so that the look-ahead is bit-exact
// against numpy (HIP's __dmul_rn/__dadd_rn are inlined plain operators that the backend would still fuse into FMAs).
// One coordinate (the reference treats x and y alike, :35-47 and :48-59):  t_hit = (distance to the wall ahead) / |v|;
// t_hit <= dt reflects.  The division (~40 instructions, and the look-ahead is a serial chain of N such steps) is only
// needed next to a wall: RN(n / |v|) <= dt is impossible when n > |v| dt (1 + 2^-50) -- the rounded product is at least
// |v| dt (1 + 2^-51), so the quotient exceeds dt by two ulps or more before rounding -- and then the branch outcome, hence
// every bit of the result, is that of the reference without computing t_hit.
__device__ __forceinline__ void coord_advance(double lo, double hi, double dt, double &x, double &v)
{
#pragma clang fp contract(off)
    const double av = fabs(v);
    const double n = v < 0 ? x - lo : hi - x;
    bool hit = false;
    double t_hit = 0.0;
    if (v != 0 && !(n > av * dt * (1.0 + 0x1p-50))) { t_hit = n / av; hit = t_hit <= dt; }
    if (hit) { const double a = v * t_hit, b = v * (dt - t_hit); x = x + (a - b); v = -v; }
    else { const double a = v * dt; x = x + a; }
}
__device__ __forceinline__ void obstacle_advance(const World w, double dt, double &x, double &vx, double &y, double &vy)
{
    coord_advance(w.xmin, w.xmax, dt, x, vx);
    coord_advance(w.ymin, w.ymax, dt, y, vy);
}
// The reference's OTHER initial guess (robot_ocp_problem.py:293-300, a commented block -- the code that recorded the two `interpolate_init` tables of
// src/simulation/test_data): stage i of N starts at
//     x = x0_x + i / N * (x0_x - x0_x) = x0_x      (sic: the x coordinate does not move),   y = x0_y + i / N * (goal_y - x0_y),
//     psi = arctan2(goal_y - x0_y, goal_x - goal_x) = arctan2(dy, 0) = +-pi/2 (0 when dy = 0),   v = omega = 0,   u = 0,
// defects included.  Plain IEEE operations in the reference's order (no contraction), so that the guess is bit for bit numpy's.
__device__ __forceinline__ void interp_guess(const double x0[5], double goal_y, int i, int N, double xg[5])
{
#pragma clang fp contract(off)
    const double dy = goal_y - x0[1];
    const double f = (double)i / (double)N;
    const double step = f * dy;
    xg[0] = x0[0] + f * (x0[0] - x0[0]);
    xg[1] = x0[1] + step;
    xg[2] = atan2(dy, 0.0);
    xg[3] = 0.0; xg[4] = 0.0;
}

// velocity noise of Obstacle.step(), visualization.py:28-33
__device__ __forceinline__ void obstacle_noise(double randomness, double vmax, double nx, double ny, double &vx, double &vy)
{
#pragma clang fp contract(off)
    const double rx = randomness * nx, ry = randomness * ny;
    const double fx = 1.0 + rx, fy = 1.0 + ry;
    vx = fmin(fmax(fx * vx, -vmax), vmax);
    vy = fmin(fmax(fy * vy, -vmax), vmax);
}

// Wavefront reductions with DPP (no LDS crossbar, no waits): four row-local butterfly steps, then the four row totals
// are combined through v_readlane.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// value held by the first lane (stage 0) of the calling lane's G-lane segment
__device__ __forceinline__ int seg21_slot(int lane);
template <int G>
__device__ __forceinline__ double lane_value_seg(double v, int lane)
{
    if (G == 21) { const double a = lane_value(v, 0), b = lane_value(v, 21), c = lane_value(v, 42); return lane < 21 ? a : (lane < 42 ? b : c); }
    if (G == 64) return lane_value(v, 0);
    if (G == 32) { const double a = lane_value(v, 0), b = lane_value(v, 32); return lane < 32 ? a : b; }
    const double a = lane_value(v, 0), b = lane_value(v, 16), c = lane_value(v, 32), d = lane_value(v, 48);
    return lane < 16 ? a : (lane < 32 ? b : (lane < 48 ? c : d));
}
template <bool SUM> __device__ __forceinline__ double seg21_reduce(double v, int lane);
// G = lanes per instance (64, 32, 21 or 16): the reduction is over the G-lane segment the calling lane belongs to.
template <int G>
__device__ __forceinline__ double seg_max(double v, int lane)
{
    if (G == 21) return seg21_reduce<false>(v, lane);
    v = fmax(v, dpp_f64<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmax(v, dpp_f64<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmax(v, dpp_f64<0x141>(v));   // row_half_mirror
    v = fmax(v, dpp_f64<0x140>(v));   // row_mirror: every lane of a 16-lane row now holds the row's result
    if (G == 16) return v;
    const double a = fmax(lane_value(v, 0), lane_value(v, 16)), b = fmax(lane_value(v, 32), lane_value(v, 48));
    if (G == 32) return lane < 32 ? a : b;
    return fmax(a, b);
}
template <int G>
__device__ __forceinline__ double seg_sum(double v, int lane)
{
    if (G == 21) return seg21_reduce<true>(v, lane);
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    if (G == 16) return v;
    const double a = lane_value(v, 0) + lane_value(v, 16), b = lane_value(v, 32) + lane_value(v, 48);
    if (G == 32) return lane < 32 ? a : b;
    return a + b;
}

// Two reductions at once (a: sum if SUM_A else max; b: max): the DPP steps of the two chains alternate, so that the wait states a DPP
// read needs after the VALU write of its source, and the latency of each dependent max / add, are filled by the other chain.
template <int G, bool SUM_A>
__device__ __forceinline__ void seg_reduce2(double &a, double &b, int lane)
{
    if (G == 21) { a = seg21_reduce<SUM_A>(a, lane); b = seg21_reduce<false>(b, lane); return; }
#define MPC_RED2_STEP(CTRL) { const double a2 = dpp_f64<CTRL>(a), b2 = dpp_f64<CTRL>(b); a = SUM_A ? a + a2 : fmax(a, a2); b = fmax(b, b2); }
    MPC_RED2_STEP(0xB1) MPC_RED2_STEP(0x4E) MPC_RED2_STEP(0x141) MPC_RED2_STEP(0x140)
#undef MPC_RED2_STEP
    if (G == 16) return;
    const double a0 = lane_value(a, 0), a1 = lane_value(a, 16), a2 = lane_value(a, 32), a3 = lane_value(a, 48);
    const double b0 = lane_value(b, 0), b1 = lane_value(b, 16), b2 = lane_value(b, 32), b3 = lane_value(b, 48);
    const double al = SUM_A ? a0 + a1 : fmax(a0, a1), ah = SUM_A ? a2 + a3 : fmax(a2, a3);
    const double bl = fmax(b0, b1), bh = fmax(b2, b3);
    if (G == 32) { a = lane < 32 ? al : ah; b = lane < 32 ? bl : bh; return; }
    a = SUM_A ? al + ah : fmax(al, ah); b = fmax(bl, bh);
}

// G = 21: THREE instances per wavefront in lanes [0, 21), [21, 42), [42, 63) (lane 63 idles).  The segments do not coincide with the
// 16-lane DPP rows -- rows 1 and 2 hold the tail of one instance and the head of the next -- so a reduction runs as two masked row
// reductions (pass L: in every row the lanes of the row's lower instance, pass U: the others; the neutral element elsewhere) whose six
// row results are combined per instance:  inst 0 = L0 + L1,  inst 1 = U1 + L2,  inst 2 = U2 + L3.  Fixed order: deterministic.
__device__ __forceinline__ int seg21_slot(int lane) { return lane >= 42 ? 2 : (lane >= 21 ? 1 : 0); }
__device__ __forceinline__ bool seg21_lower(int lane)
{   // lower instance of row r: rows 0, 1 -> 0; row 2 -> 1; row 3 -> 2
    const int row = lane >> 4, slot = seg21_slot(lane);
    return slot == (row <= 1 ? 0 : row - 1);
}
template <bool SUM>
__device__ __forceinline__ double row_reduce(double v)
{
#define MPC_ROW_STEP(CTRL) { const double w = dpp_f64<CTRL>(v); v = SUM ? v + w : fmax(v, w); }
    MPC_ROW_STEP(0xB1) MPC_ROW_STEP(0x4E) MPC_ROW_STEP(0x141) MPC_ROW_STEP(0x140)
#undef MPC_ROW_STEP
    return v;
}
template <bool SUM>
__device__ __forceinline__ double seg21_reduce(double v, int lane)
{
    const bool low = seg21_lower(lane);
    const double id = SUM ? 0.0 : -INFINITY;
    const double vl = row_reduce<SUM>(low ? v : id), vu = row_reduce<SUM>(low ? id : v);
    const double l0 = lane_value(vl, 0), l1 = lane_value(vl, 16), l2 = lane_value(vl, 32), l3 = lane_value(vl, 48);
    const double u1 = lane_value(vu, 16), u2 = lane_value(vu, 32);
    const double r0 = SUM ? l0 + l1 : fmax(l0, l1), r1 = SUM ? u1 + l2 : fmax(u1, l2), r2 = SUM ? u2 + l3 : fmax(u2, l3);
    const int slot = seg21_slot(lane);
    return slot == 0 ? r0 : (slot == 1 ? r1 : r2);
}

// ---- POLISH of the interior point (round 5; shared specification with oracle/mpc_oracle.c ipm_solve) ----
// Once the termination test holds, an instance takes up to kPolishMax further iterations while either indicator holds:
//  (a) slow end-game: its LAST iteration reduced the largest live complementarity product c_max by less than a factor 1 / polish_ratio
//      (c_max(k) > polish_ratio c_max(k - 1)) -- c_max is the termination test's own measure: no extra reduction;
//  (b) a multiplier collapsed to the floor on a weakly active row (complementary, primal feasible: invisible to the termination test) but the last primal
//      step is still long: for ANY stage, with s the max-norm of the stage's last step alpha * dz, s' of the one before and r = min(s / s', 1/2), the
//      estimate s r min(1, 10 r) = min(s / 2, s^2 / s', 10 s^3 / s'^2) of what remains exceeds polish_tol.  FLOAT arithmetic without a division, as the oracle's
//      polish_wanted(); a stage is a lane, so the indicator needs no cross-lane reduction: one ballot (seg_any).
// What solves left behind when their products slipped under qp_tol was the parity tail beyond 1e-6.
static constexpr int kPolishMax = 2;
static constexpr float kPolishUnsolved = 100.0f;      // oracle/mpc_oracle.c POLISH_UNSOLVED
static constexpr float kStationarityStep = 1e-6f;     // indicator (c), the stationarity residual (adjoint_inputs), is formed only behind a step that was still longer than this in
                                                      // some stage (oracle/mpc_oracle.c STATIONARITY_STEP): behind a shorter one the step length itself bounds what remains
// MPC_NAN_NOTE.  A NaN / overflow of the row state must end the solve (status 4) as it does in the oracle, where it surfaces in mu at the head of the next
// iteration.  In the kernels the floors of the update (t = fmax(t + a dt, floor): fmax drops a NaN) would wash it out of t and lam, and a solve that diverged
// under qp_fail_policy 1 would come back "converged" with every pair at the floor and the linear residual at 0 (found by the truncate-policy test once the
// centring target changed the path of an infeasible QP).  The affine sums carry the NaN -- sigma, hence smu = sigma min(mu, c_max), is NaN then -- and the
// step-length test of the same iteration looks at it.

// ---- THE INTERIOR POINT'S SCALAR DECISIONS, ONE DEFINITION (round 5) ----
// Everything of an iteration that is a decision on segment-uniform scalars -- the three status tests with the polish, the affine step lengths, sigma and the
// centring target, the step lengths of the combined step and its failure test -- is defined HERE, once, for the branch-free and the branched form of
// rti_solve_kernel and for rti_split_kernel (rti_split_kernel.hpp); the constants they use (kFracToBoundary, kMuCapSettled, kPolishMax, KParams::tol /
// mu_div / mu_cap / mu_settled / polish_ratio / iter_max) appear nowhere else in the loops.  The oracle's ipm_solve (oracle/mpc_oracle.c) states the same
// tests in its own words.  What stays per kernel is the ROW arithmetic, whose form IS the storage policy (rows in registers with 0 / 1 factors, lean rows
// recomputed per phase, rows under branches, rows dealt over the lanes of a stage).  __forceinline__: the same instructions as the inlined text they replace.
struct IpmState {            // per instance (segment-uniform)
    int status = 2, it_done = 0, npolish = 0;
    bool running = true;
    bool want_step = false;   // polish indicator (b) of the step just taken (ipm_polish_step) ...
    bool unsolved = false;    // ... and the same estimate against kPolishUnsolved x polish_tol: an end-game that is not a tail but a QP left unsolved
    bool long_step = false;   // some stage's last step was longer than kStationarityStep: only then is indicator (c) worth its sweep (carried here by
                              // ipm_polish_step<G, true>; the branch-free form of rti_solve_kernel keeps it at true and decides at its head instead)
    bool ask_g = false;       // the termination test holds and indicators (a), (b) are silent: the caller forms the stationarity residual and ipm_head_g decides (indicator (c))
    double cprev = INFINITY;  // c_max at the head of the previous iteration
};
// head of iteration `it`: failure by NaN / divergence, convergence (or a polish iteration), iteration cap
__device__ __forceinline__ void ipm_head(const KParams &p, IpmState &S, int it, double mu, double lin, double cmax)
{
    if (!S.running) return;
    if (!(mu == mu) || !(fabs(mu) <= p.mu_div)) { S.status = 4; S.running = false; S.it_done = it; }      // NaN, or diverged: an infeasible QP
    else if (lin <= p.tol && cmax <= p.tol) {
        // converged -- or one more iteration, the polish (kPolishMax; polish off: the ratio is +inf) -- or NOT SOLVED: the polish is used up and the step
        // estimate still stands two orders of magnitude above polish_tol (an end-game whose Newton steps have lost their accuracy to the barrier weights
        // lam / t_floor; oracle ipm_solve): the step is applied as after an iteration cap, status 2, instead of being reported as converged
        const bool want = cmax > p.polish_ratio * S.cprev || S.want_step;
        if (want && S.npolish >= kPolishMax && S.unsolved) { S.status = 2; S.running = false; S.it_done = it; }
        else if (want && S.npolish < kPolishMax && it < p.iter_max) S.npolish++;
        // indicator (c), round 6: undecided until the caller has formed the stationarity residual (ipm_head_g) -- asked only where it decides: a polish iteration is left
        else if (!want && S.long_step && S.npolish < kPolishMax && it > 0 && it < p.iter_max && p.polish_res_g < INFINITY) S.ask_g = true;
        else { S.status = 0; S.running = false; S.it_done = it; }
    }
    else if (it >= p.iter_max) {      // at the cap with mu above a healthy solve's: diverging or stalled, not slow
        S.status = (mu > p.mu_cap || (it >= kMuCapSettled && mu > p.mu_settled)) ? 4 : 2; S.running = false; S.it_done = it;
    }
}
// ... second half of the head for the instances that asked (IpmState::ask_g): converged, or one more iteration because the Lagrangian is not yet stationary
__device__ __forceinline__ void ipm_head_g(const KParams &p, IpmState &S, int it, double res_g)
{
    if (!S.ask_g) return;
    S.ask_g = false;
    if (res_g > p.polish_res_g) S.npolish++;
    else { S.status = 0; S.running = false; S.it_done = it; }
}
// largest affine steps that keep t, lam > 0 from the largest ratios -dt/t, -dlam/lam (true divisions, as the oracle: a 1-ulp reciprocal here moves a
// sensitive instance past the parity tolerance)
__device__ __forceinline__ void ipm_affine_steps(double rmax, double rmaxd, double &a_aff, double &a_affd)
{
    a_aff = rmax > 1.0 ? 1.0 / rmax : 1.0; a_affd = rmaxd > 1.0 ? 1.0 / rmaxd : 1.0;
}
// sigma = min(1, (mu_aff / mu)^2) and the centring target sigma * min(mu, c_max) (c_max: largest product of a pair off the floor; oracle ipm_solve mu_c)
__device__ __forceinline__ double ipm_centring(double maff, double mu, double cmax, double &sigma)
{
    sigma = mu > 0 ? maff / mu : 0.0;
    sigma = sigma * sigma;
    if (sigma > 1.0) sigma = 1.0;
    return sigma * fmin(mu, cmax);
}
// step lengths of the combined step: 1 if unblocked, else kFracToBoundary of the largest step that keeps t (alpha) and lam (alphad) positive
__device__ __forceinline__ void ipm_step_lengths(double rmax, double rmaxd, double &alpha, double &alphad)
{
    const double amax = rmax > 1.0 ? 1.0 / rmax : 1.0, amaxd = rmaxd > 1.0 ? 1.0 / rmaxd : 1.0;
    alpha = (amax >= 1.0) ? 1.0 : kFracToBoundary * amax;          // primal step: z, s, t
    alphad = (amaxd >= 1.0) ? 1.0 : kFracToBoundary * amaxd;       // dual step: lam
}
// polish indicator (b): est > tol with est = max(s r min(1, 10 r), kappa s), r = min(s / s', 1/2).  s r min(1, 10 r) is the SMALLEST of s / 2, s^2 / s' and 10 s^3 / s'^2 (r >= 1/2: the
// first; 0.1 < r < 1/2: the second; r <= 0.1: the third), so est > tol is the conjunction of three comparisons -- no division, no case selection, float
// arithmetic, every product left to right exactly as oracle/mpc_oracle.c::polish_wanted forms it.
__device__ __forceinline__ void polish_wanted(float s, float sp, float tol, float tol_unsolved, float kappa, bool &want, bool &unsolved)
{
    float pa = 0.5f * s, pb = s * s, qb = tol * sp, pc = 10.0f * s * s * s, qc = tol * sp * sp, rb = tol_unsolved * sp, rc = tol_unsolved * sp * sp, pd = kappa * s;
    asm volatile("" : "+v"(pa), "+v"(pb), "+v"(qb), "+v"(pc), "+v"(qc), "+v"(rb), "+v"(rc), "+v"(pd));      // (products pinned: what is left are compares and mask logic -- nothing to branch around)
    want = ((pa > tol) & (pb > qb) & (pc > qc)) | (pd > tol);      // kappa s: the floor of the estimate (KParams::polish_kappa, mpc_config.polish_step_frac)
    unsolved = ((pa > tol_unsolved) & (pb > rb) & (pc > rc)) | (pd > tol_unsolved);      // the same estimate against kPolishUnsolved x polish_tol (KParams::polish_tol_unsolved)
}
// does any lane of the calling lane's G-lane segment hold `w`?  One ballot; the segment mask is a per-lane constant.
template <int G>
__device__ __forceinline__ bool seg_any(bool w, int lane)
{
    const unsigned long long b = __ballot(w);
    if (G == 64) return b != 0ull;
    unsigned long long m;
    if (G == 32) m = lane < 32 ? 0xffffffffull : 0xffffffff00000000ull;
    else if (G == 16) m = 0xffffull << (lane & 48);
    else m = 0x1fffffull << (21 * seg21_slot(lane));      // G = 21: lanes [0, 21), [21, 42), [42, 63)
    return (b & m) != 0ull;
}
// after the step lengths are known: this lane's (stage's) step norm, its estimate against polish_tol, the segment's verdict for the next head
// CARRY: the trigger of indicator (c) -- some stage's last step longer than kStationarityStep -- is carried to the next head in S.long_step (one ballot per iteration);
// a kernel that decides it where it asks (the branch-free form of rti_solve_kernel) leaves S.long_step at true
template <int G, bool CARRY = true>
__device__ __forceinline__ void ipm_polish_step(const KParams &p, IpmState &S, int lane, double alpha, const double dz[7], float &stepl)
{
    // (float)max|dz| = max|(float)dz|: round-to-nearest is monotone, so the maximum is taken on the floats (v_max3_f32 with |.| modifiers: 3 instructions)
    const float f0 = (float)dz[0], f1 = (float)dz[1], f2 = (float)dz[2], f3 = (float)dz[3], f4 = (float)dz[4], f5 = (float)dz[5], f6 = (float)dz[6];
    const float dm = fmaxf(fmaxf(fmaxf(fabsf(f0), fabsf(f1)), fabsf(f2)), fmaxf(fmaxf(fmaxf(fabsf(f3), fabsf(f4)), fabsf(f5)), fabsf(f6)));
    const float sn = (float)alpha * dm;
    bool w, u;
    polish_wanted(sn, stepl, p.polish_tol, p.polish_tol_unsolved, p.polish_kappa, w, u);      // (polish_tol = +inf: indicator off)
    stepl = sn;
    S.want_step = seg_any<G>(w, lane); S.unsolved = seg_any<G>(u, lane);
    if constexpr (CARRY) S.long_step = seg_any<G>(sn > kStationarityStep, lane);
}
// A step that is not finite is not a step (MPC_NAN_NOTE): an overflow that reached z without passing through mu or sigma (the last iteration of a solve that
// diverged under qp_fail_policy 1 takes alpha = 1 on an infinite direction, and the floors then wash the row state clean) ends as status 4, iterate untouched,
// as in the oracle, whose measured residuals are NaN then.  Once per solve, after the loop.
template <int G>
__device__ __forceinline__ int ipm_finite_step(int status, const double z[7], int lane)
{
    const double fin = z[0] + z[1] + z[2] + z[3] + z[4] + z[5] + z[6];
    return seg_any<G>(!(fabs(fin) <= 1e300), lane) ? 4 : status;
}
// step collapse, or a NaN of the row state (MPC_NAN_NOTE): status 4
__device__ __forceinline__ void ipm_step_check(IpmState &S, int it, double alpha, double alphad, double smu)
{
    if (S.running && (!(alpha > 1e-14) || !(alphad > 1e-14) || !(smu == smu))) { S.status = 4; S.running = false; S.it_done = it; }
}

// reciprocal: hardware seed + two Newton steps (1-2 ulp); used for 1/t of the inequality rows
// a value every lane of the wavefront holds identically (one instance per wavefront), moved to a scalar register pair
__device__ __forceinline__ double wave_uniform(double x)
{
    const long long b = __double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

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

// wave-uniform copy of lane `src`'s value (v_readlane_b32 x2 -> SGPR pair)
__device__ __forceinline__ double bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// DPP whole-wave shifts: from_right(v)[j] = v[j+1] (wave_shl:1), from_left(v)[j] = v[j-1] (wave_shr:1); bound_ctrl makes
// the end lane read 0, so no "old" operand has to be materialised in front of every DPP move
__device__ __forceinline__ double from_right(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_left(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// ---- POLISH INDICATOR (c): THE STATIONARITY RESIDUAL (round 6; oracle/mpc_oracle.c ipm_solve, residuals()) ----
// With g_t = (H z + q - C' lam)_t per stage (z order: ua, ual, x, y, psi, v, om) and the costates of the OPEN-LOOP adjoint recursion
//     pi_N = g_x,N,    pi_t = g_x,t + A_t' pi_{t+1},
// the state blocks of the Lagrangian's gradient vanish and what is left are the input blocks  ru_t = g_u,t + B_t' pi_{t+1}  (HPIPM's res_g on this
// choice of costates; the slack equations are per row and stay with the callers).  The recursion is sequential in the stage index -- but A = I + E with E
// strictly upper triangular in the order (x, y | psi, v | om) (the plant is a chain of integrators), so it is THREE LEVELS OF SUFFIX SUMS over the stages:
//     pi_x = suf(g_x), pi_y = suf(g_y);     pi_psi = suf(g_psi + a02 pi_x+ + a12 pi_y+), pi_v = suf(g_v + a03 pi_x+ + a13 pi_y+);
//     pi_om = suf(g_om + a04 pi_x+ + a14 pi_y+ + dt pi_psi+)                    (+ : the value of stage t + 1)
// and a suffix sum over the lanes of an instance is four DPP shift-and-add steps per 16-lane row plus the totals of the rows behind it: ~30 instructions per
// level whatever the horizon, against 32 per STAGE for the systolic form of the same recursion (N = 50: 1600).  Formed about once per two solves.
// inclusive suffix sum over the calling lane's 16-lane DPP row: out[j] = v[j] + v[j + 1] + ... to the end of the row (row_shl:n reads lane j + n, zero beyond the row)
__device__ __forceinline__ double row_suffix(double v)
{
    v += dpp_f64<0x101>(v);
    v += dpp_f64<0x102>(v);
    v += dpp_f64<0x104>(v);
    v += dpp_f64<0x108>(v);
    return v;
}
// ... over the G-lane segment of the calling lane (G = 64, 32, 16: whole rows; G = 21: three instances in lanes [0, 21), [21, 42), [42, 63) -- two masked row
// passes as in seg21_reduce, L for the lanes of a row's lower instance and U for the others; an instance continues into the NEXT row's lower part).  Fixed order.
template <int G>
__device__ __forceinline__ double seg_suffix(double v, int lane)
{
    const int row = lane >> 4;
    if (G == 21) {
        const bool low = seg21_lower(lane);
        const double vl = row_suffix(low ? v : 0.0), vu = row_suffix(low ? 0.0 : v);
        const double l1 = lane_value(vl, 16), l2 = lane_value(vl, 32), l3 = lane_value(vl, 48);
        const double nxt = row == 0 ? l1 : (row == 1 ? l2 : (row == 2 ? l3 : 0.0));
        return low ? (row == 0 ? vl + nxt : vl) : vu + nxt;
    }
    v = row_suffix(v);
    if (G == 16) return v;
    const double t1 = lane_value(v, 16), t2 = lane_value(v, 32), t3 = lane_value(v, 48);
    if (G == 32) return (row & 1) ? v : v + (row == 0 ? t1 : t3);
    const double a2 = t2 + t3, a1 = t1 + a2;
    return v + (row == 0 ? a1 : (row == 1 ? a2 : (row == 2 ? t3 : 0.0)));
}
// max(|ru_t[0]|, |ru_t[1]|) of the calling lane's stage.  `src`: this lane holds a stage's g (one lane per stage: the stage exists; rti_split_kernel: the stage's
// first lane -- the other lanes of a stage enter the sums with zero and so hold the NEXT stage's suffix, which is what the first lane reads from its right
// neighbour); `nxt`: ... and the stage has a successor (t < N).
template <int G>
__device__ __forceinline__ double adjoint_inputs(bool src, bool nxt, const StageLin &S, const double g[7], int lane)
{
    const double s0 = from_right(seg_suffix<G>(src ? g[2] : 0.0, lane)), s1 = from_right(seg_suffix<G>(src ? g[3] : 0.0, lane));
    const double px = nxt ? s0 : 0.0, py = nxt ? s1 : 0.0;
    const double s2 = from_right(seg_suffix<G>(src ? g[4] + S.a02 * px + S.a12 * py : 0.0, lane));
    const double s3 = from_right(seg_suffix<G>(src ? g[5] + S.a03 * px + S.a13 * py : 0.0, lane));
    const double ppsi = nxt ? s2 : 0.0, pv = nxt ? s3 : 0.0;
    const double s4 = from_right(seg_suffix<G>(src ? g[6] + S.a04 * px + S.a14 * py + S.dt * ppsi : 0.0, lane));
    const double pom = nxt ? s4 : 0.0;
    const double ru0 = g[0] + S.b00 * px + S.b10 * py + S.dt * pv, ru1 = g[1] + S.b01 * px + S.b11 * py + S.h2 * ppsi + S.dt * pom;
    return nxt ? fmax(fabs(ru0), fabs(ru1)) : 0.0;
}

// Riccati factors of one stage, kept in the registers of the lane that owns the stage
struct StageFac {
    double K0[5], K1[5];   // feedback gains (2 x 5)
    double i00, l, i11;    // LDL' of Muu: 1/d0, l, 1/d1
    double k0, k1;         // feed-forward of the current right-hand side
};

// SYSTOLIC STAGE RECURSIONS.  The lane with stage index t (lane = slot * G + t when several instances share a wavefront) owns
// stage t and holds that stage's blocks in registers.  The recursion state (cost-to-go Hessian P and gradient q going
// backward; the state step dx going forward) travels from lane to lane by a one-lane DPP wave shift per stage: the value lane t
// computes in step t is the one lane t-1 (t+1) consumes in the next step.  Whatever the other lanes (idle lanes, the
// neighbouring instance) push into the chain arrives at a lane only AFTER that lane's own step, so it is never consumed.
// No LDS, no barriers, no global traffic inside the interior-point loop (this variant).
//
// Backward Riccati factorisation + predictor right-hand side.
//   q_t = gxs + A'(q+ + P+ r_b) + K'(lu + B'(q+ + P+ r_b)) is the cost-to-go gradient INCLUDING the costate of the
//   current multipliers: gxs = (H z + q - C'lam)_x + (sum_c c beta_c)_x, lu the same for the input block.  It is the
//   sum of the adjoint (costate) recursion and the Newton right-hand-side recursion, which share the propagator.
// Terminal lane N: W = 0 and incoming P = 0 give M = H~_N, K = 0, P_N = Q~_N, q_N = gxs, so one code serves all stages.
__device__ __forceinline__ void systolic_factor(int stage, int N, const StageLin &S, const double Hq[8], double lu0, double lu1,
                                                const double gxs[5], const double bbr[5], bool affine, StageFac &F)
{
    // Every lane executes the stage arithmetic in every step (a wave instruction costs the same for 1 or 64 active lanes);
    // only lane t holds a meaningful incoming state in step t, and only lane t keeps the factors it computed.  Computing
    // unconditionally lets the results flow through fresh registers straight into the DPP shift, without the masked copies
    // into loop-carried registers that an `if (stage == t)` around the arithmetic costs (~12 % of this loop).
    double P[5][5], qv[5];       // incoming state (valid in lane t at step t); symmetric entries share one value
#pragma unroll
    for (int r = 0; r < 5; r++) {
        qv[r] = 0.0;
#pragma unroll
        for (int c = 0; c < 5; c++) P[r][c] = 0.0;
    }
    const double dt = S.dt, h2 = S.h2;
    for (int t = N; t >= 0; t--) {
        double M[5][5], qo[5], mu0[5], mu1[5];
        // columns x, y of P W are columns 0, 1 of P
        mu0[0] = S.b00 * P[0][0] + S.b10 * P[1][0] + dt * P[3][0];
        mu0[1] = S.b00 * P[0][1] + S.b10 * P[1][1] + dt * P[3][1];
        mu1[0] = S.b01 * P[0][0] + S.b11 * P[1][0] + h2 * P[2][0] + dt * P[4][0];
        mu1[1] = S.b01 * P[0][1] + S.b11 * P[1][1] + h2 * P[2][1] + dt * P[4][1];
        M[0][0] = Hq[2] + P[0][0]; M[0][1] = Hq[7] + P[0][1]; M[1][1] = Hq[3] + P[1][1];
        double m00, m01, m11;
        {   // column ua
            double T[5];
#pragma unroll
            for (int k = 0; k < 5; k++) T[k] = P[k][0] * S.b00 + P[k][1] * S.b10 + P[k][3] * dt;
            m00 = Hq[0] + S.dua(T);
        }
        {   // column ual
            double T[5];
#pragma unroll
            for (int k = 0; k < 5; k++) T[k] = P[k][0] * S.b01 + P[k][1] * S.b11 + P[k][2] * h2 + P[k][4] * dt;
            m01 = S.dua(T); m11 = Hq[1] + S.dual(T);
        }
        {   // column psi
            double T[5];
#pragma unroll
            for (int k = 0; k < 5; k++) T[k] = P[k][0] * S.a02 + P[k][1] * S.a12 + P[k][2];
            mu0[2] = S.dua(T); mu1[2] = S.dual(T);
            M[0][2] = T[0]; M[1][2] = T[1]; M[2][2] = Hq[4] + S.dpsi(T);
        }
        {   // column v
            double T[5];
#pragma unroll
            for (int k = 0; k < 5; k++) T[k] = P[k][0] * S.a03 + P[k][1] * S.a13 + P[k][3];
            mu0[3] = S.dua(T); mu1[3] = S.dual(T);
            M[0][3] = T[0]; M[1][3] = T[1]; M[2][3] = S.dpsi(T); M[3][3] = Hq[5] + S.dv(T);
        }
        {   // column om
            double T[5];
#pragma unroll
            for (int k = 0; k < 5; k++) T[k] = P[k][0] * S.a04 + P[k][1] * S.a14 + P[k][2] * dt + P[k][4];
            mu0[4] = S.dua(T); mu1[4] = S.dual(T);
            M[0][4] = T[0]; M[1][4] = T[1]; M[2][4] = S.dpsi(T); M[3][4] = S.dv(T); M[4][4] = Hq[6] + S.dom(T);
        }
        // Muu = L D L' (backward stable; the closed-form inverse through det cancels catastrophically when a state
        // row's barrier weight makes B'PB nearly rank one)
        const double i00 = rcp_nr(m00);
        const double l = m01 * i00;
        const double i11 = rcp_nr(m11 - l * m01);
        double K0[5], K1[5];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            K1[c] = fma(l, mu0[c], -mu1[c]) * i11;            // -(mu1 - l mu0) / d1
            K0[c] = fma(-l, K1[c], -(mu0[c] * i00));          // -(mu0 / d0 + l K1)
        }
        // gradient part (needs the OLD P): qb = q+ + P+ r_b
        double qb[5];
        if (affine) {
#pragma unroll
            for (int k = 0; k < 5; k++) qb[k] = qv[k] + P[k][0] * bbr[0] + P[k][1] * bbr[1] + P[k][2] * bbr[2] + P[k][3] * bbr[3] + P[k][4] * bbr[4];
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) qb[k] = qv[k];
        }
        const double m0 = lu0 + S.dua(qb), m1 = lu1 + S.dual(qb);
        const double k1 = fma(l, m0, -m1) * i11;
        const double k0 = fma(-l, k1, -(m0 * i00));
        qo[0] = gxs[0] + qb[0] + K0[0] * m0 + K1[0] * m1;
        qo[1] = gxs[1] + qb[1] + K0[1] * m0 + K1[1] * m1;
        qo[2] = gxs[2] + S.dpsi(qb) + K0[2] * m0 + K1[2] * m1;
        qo[3] = gxs[3] + S.dv(qb) + K0[3] * m0 + K1[3] * m1;
        qo[4] = gxs[4] + S.dom(qb) + K0[4] * m0 + K1[4] * m1;
        if (stage == t) {      // the lane that owns stage t keeps its factors
            F.i00 = i00; F.l = l; F.i11 = i11; F.k0 = k0; F.k1 = k1;
#pragma unroll
            for (int c = 0; c < 5; c++) { F.K0[c] = K0[c]; F.K1[c] = K1[c]; }
        }
        // new cost-to-go Hessian P = Mxx + K'Mux (upper triangle), handed with q to the lane on the left (stage t-1)
#pragma unroll
        for (int r = 0; r < 5; r++) {
            qv[r] = from_right(qo[r]);
#pragma unroll
            for (int c = r; c < 5; c++) {
                const double v = from_right(fma(K1[r], mu1[c], fma(K0[r], mu0[c], M[r][c])));
                P[r][c] = v; P[c][r] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// ROW-PARALLEL RICCATI FACTORISATION (vector ALU + 64-bit DPP).  Same homogeneous formulation as the matrix-core variant
// below (z~ = (x[5], 1, ua, ual), W~ = [A b_r B; 0 1 0], P~ = [P q; q' .]), but the 8 columns of one stage's matrices are
// spread over lanes 0..7 of the first 16-lane row of the instance's lane group: lane j holds column j of P~, W~_t, H~aug_t,
// T = P~ W~ and M~ = H~aug + W~' T.  A product A B then is one  v_fmac_f64_dpp acc_r, A_r(row_newbcast:k), B_k  per (r, k)
// for ALL columns at once -- the DPP operand reads lane k's register, i.e. an element of the left factor, at full FP64 rate.
// Per stage: 30 + 40 + 12 DPP FMAs and ~35 ordinary instructions, against ~330 for the one-lane systolic sweep.
//   * only upper-triangle entries of P~ are ever read (element (r, k), r <= k, is lane k's register r), so the cost-to-go
//     stays symmetric by construction, exactly as in systolic_factor;
//   * the per-stage operands come from LDS, staged there by the lanes that own the stages (W~ once per solve, its affine
//     column and H~aug once per interior-point iteration) and fetched one stage ahead; K~, k and the LDL' factors return
//     through LDS (they overlay rows 0, 1 of the consumed H~aug_t);
//   * hazards: a DPP read of a VGPR needs 2 wait states after the VALU write (5 after an EXEC write); every block starts
//     with the s_nop that covers whatever the compiler scheduled in front of it, inside a block no DPP source is written.
// ------------------------------------------------------------------------------------------------------------------
struct RowLds {
    // per-stage strides are odd: the owning lanes (stage t in lane t) write their blocks in parallel, stride t * WS / t * HS
    // doubles between lanes, which an even stride would put on one LDS bank
    static constexpr int WS = 41, HS = 65;
    // The vector recursions request operands AHEAD stage blocks in advance without clamping the stage index: behind the last
    // instance's H that lands in a rear padding, in front of H in the instance's own W region (plus a front padding when
    // W is shorter than that).  Lanes that have nothing to store write into the unused tail [TAIL..HS) of the stage blocks.
    static constexpr int AHEAD = 3, TAIL = 48;
    static constexpr int DEADK = 48, DEADF = 50;      // dead-store words of idle lanes: K~ rows go to [x], [x + 8]; factors to [x], [x + 1], [x + 8]
    static constexpr bool COMPACT = false;
    static __host__ __device__ constexpr int per_instance(int N) { return WS * N + HS * (N + 1); }   // doubles
    static __host__ __device__ constexpr int pad_front(int N) { return AHEAD * HS > WS * (N - 1) ? AHEAD * HS - WS * (N - 1) : WS; }   // >= one W block
    static __host__ __device__ constexpr int pad_rear() { return AHEAD * HS; }
    static __host__ __device__ constexpr int total(int N, int instances) { return pad_front(N) + instances * per_instance(N) + pad_rear(); }
    double *W, *H;     // W[t][k][j] (5 x 8 per stage, t < N), H[t][i][j] (8 x 8 per stage, t <= N)
    double *R;         // results of the sweeps, one block of HS words per stage: overlays H (the blocks are dead once fetched) unless a
                       // region of its own is given -- then the structural zeros of the H~aug blocks survive an iteration and only their
                       // non-zeros have to be restaged (an LDS instruction costs a lone wavefront ~14 cycles)
    __device__ __forceinline__ RowLds(double *base, int N) : W(base), H(base + WS * N), R(base + WS * N) {}
    __device__ __forceinline__ RowLds(double *base, int N, double *results) : W(base), H(base + WS * N), R(results) {}
};

// COMPACT LAYOUT (three instances per wavefront, G = 21): the dense blocks above cost 17.5 KB of LDS per instance, which lets a CU hold 8
// instances (4 wavefronts x 2) -- 12 (4 x 3) need <= 13.6 KB each.  What is stored here is what the sweeps cannot synthesise:
//   W~_t: rows 0, 1 of [A b B] (16 words) and b_t[2..4] (3 words) -- rows 2..4 are constants (0, 1, dt, dt^2/2) except their affine entry,
//         so every lane but lane 5 reads them from a 24-word table shared by the wavefront (same instruction, per-lane base and stride);
//   H~aug_t: rows 0..5 (48 words) + [48] = H66, [49] = [50] = 0, [51] = H77: rows 6, 7 are zero except H[6][5] = H[5][6], H[7][5] = H[5][7]
//         (already in row 5, words 46, 47) and the two diagonal entries, so ONE ds_read2_b64 with a per-lane base
//         (lanes 0..4: 49, lane 5: 46, lane 6: 48, lane 7: 50) delivers (H[6][j], H[7][j]) to every lane.
// Results overlay the block exactly as in the dense layout (RowVec: words 0..44); dead-store words of idle lanes are 47, 48, 55.
struct RowLdsC {
    static constexpr int WS = 21, HS = 57;
    static constexpr int AHEAD = 3, TAIL = 48;
    static constexpr int DEADK = 47, DEADF = 47;      // K~ rows go to [x], [x + 8]; factors to [x], [x + 1], [x + 8]
    static constexpr int CT = 24;                     // constant table: [3 j + m] = W~[2 + m][j] for j != 5
    static constexpr bool COMPACT = true;
    static __host__ __device__ constexpr int per_instance(int N) { return WS * N + HS * (N + 1); }
    // front padding: what the operand requests ahead of stage 0 may touch -- and, with the lean row state, 5 words per instance of the wavefront for
    // the initial-condition residual (at least 16 words: N = 9 would leave 3, found by scripts/fuzz_parity.py)
    static __host__ __device__ constexpr int pad_front(int N)
    {
        return (AHEAD * HS > WS * (N - 1) ? AHEAD * HS - WS * (N - 1) : WS) < 16 ? 16 : (AHEAD * HS > WS * (N - 1) ? AHEAD * HS - WS * (N - 1) : WS);
    }
    static __host__ __device__ constexpr int pad_rear() { return AHEAD * HS; }
    static __host__ __device__ constexpr int total(int N, int instances) { return CT + pad_front(N) + instances * per_instance(N) + pad_rear(); }
    // with the obstacle look-ahead of the instances kept resident behind the blocks (10 obstacles, "lean" row state): the rear padding is only
    // ever READ (operand requests running ahead of the last stage), so the positions may occupy it
    static __host__ __device__ constexpr int positions_at(int N, int instances) { return CT + pad_front(N) + instances * per_instance(N); }
    static __host__ __device__ constexpr int total_with_positions(int N, int instances, int n_obst)
    {
        return positions_at(N, instances) + (instances * (N + 1) * n_obst * 2 > pad_rear() ? instances * (N + 1) * n_obst * 2 : pad_rear());
    }
    double *W, *H, *R, *C;
    __device__ __forceinline__ RowLdsC(double *base, int N, double *table) : W(base), H(base + WS * N), R(base + WS * N), C(table) {}
};

// Every solve kernel is ONE wavefront per workgroup, and the LDS executes a wavefront's requests in order: a write is visible to a later read of
// another lane without waiting for it.  A workgroup barrier (__syncthreads) would drain the LDS queue first (s_waitcnt lgkmcnt(0), ~100 cycles a
// lone wavefront cannot hide, eight to fourteen times per interior-point iteration); release / acquire at wavefront scope orders the accesses for
// the compiler and emits nothing.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t lds_address(const void *p) { return (uint32_t)(uintptr_t)p; }     // low half of a flat LDS address = LDS byte offset

template <class LT>
__device__ __forceinline__ void rowpar_factor(int lane, int N, const LT L, bool worker_row)
{
    constexpr int WS = LT::WS, HS = LT::HS;
    const int j = lane & 7, l15 = lane & 15;    // column; lanes 8..15 of a row mirror 0..7
    const bool store = worker_row && l15 < 6, store0 = worker_row && l15 == 0;
    const double d5 = (j == 5) ? 1.0 : 0.0;     // row 5 of W~ is e_5'
    // Results overlay rows 0, 1 of the consumed H~aug_t: K~[0][j] at [j], K~[1][j] at [8 + j], 1/d0, l, 1/d1 at [6], [7], [14].
    // Stores are unconditional: lanes with nothing to store write into dead parts of the same block ([16..63]).
    // (two per-lane pointers, the rest are immediate offsets: K~[1][j] is 8 words behind K~[0][j], 1/d1 8 words behind 1/d0, and the
    // dead words an idle lane hits instead -- 48, 56 resp. 50, 51, 58 -- lie in rows 6, 7 of the block, consumed one stage earlier)
    double *kp = L.R + (store ? j : LT::DEADK), *fp = L.R + (store0 ? 6 : LT::DEADF);
    const double *wp = L.W + j, *hp = L.H + j;
    // compact layout only: rows 2..4 of W~ (lane 5: the stage's b[2..4], stride WS; other lanes: the wavefront's constant table, stride 0)
    // and rows 6, 7 of H~aug through a per-lane base
    const double *bp = nullptr, *h6p = nullptr;
    int bstride = 0;
    if constexpr (LT::COMPACT) {
        bp = (j == 5) ? L.W + 16 : L.C + 3 * j;
        bstride = (j == 5) ? WS : 0;
        h6p = L.H + (j < 5 ? 49 : (j == 5 ? 46 : (j == 6 ? 48 : 50)));
    }
    auto fetch = [&](int t, double Wc[5], double Hc[8]) {
        if constexpr (LT::COMPACT) {
            Wc[0] = wp[WS * t]; Wc[1] = wp[WS * t + 8];
#pragma unroll
            for (int m = 0; m < 3; m++) Wc[2 + m] = bp[bstride * t + m];
#pragma unroll
            for (int i = 0; i < 6; i++) Hc[i] = hp[HS * t + i * 8];
            Hc[6] = h6p[HS * t]; Hc[7] = h6p[HS * t + 1];
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) Wc[k] = wp[WS * t + k * 8];
#pragma unroll
            for (int i = 0; i < 8; i++) Hc[i] = hp[HS * t + i * 8];
        }
    };
    // first half of a stage: T = P~ W~ from the previous stage's P~ (the upper-left 6 x 6 of its M registers)
    auto stepT = [&](const double Wc[5], const double Pc[8], double T[6]) {
#pragma unroll
        for (int r = 0; r < 6; r++) T[r] = Pc[r] * d5;                 // k = 5 term of T = P~ W~
        asm volatile(
            "s_nop 1\n"
            "v_fmac_f64_dpp %0, %6, %12 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %6, %12 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %6, %12 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %6, %12 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %6, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %6, %12 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %6, %13 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %7, %13 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %7, %13 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %7, %13 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %7, %13 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %7, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %6, %14 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %7, %14 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %8, %14 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %8, %14 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %8, %14 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %8, %14 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %6, %15 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %7, %15 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %8, %15 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %9, %15 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %9, %15 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %9, %15 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %6, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %7, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %8, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %9, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %10, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %10, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]), "+v"(T[4]), "+v"(T[5])
            : "v"(Pc[0]), "v"(Pc[1]), "v"(Pc[2]), "v"(Pc[3]), "v"(Pc[4]), "v"(Pc[5]), "v"(Wc[0]), "v"(Wc[1]), "v"(Wc[2]), "v"(Wc[3]), "v"(Wc[4]));
    };
    // second half: M~ (accumulated onto the H~aug column in M), L D L' of Muu, column j of K~, P~+ in M[0..5]
    auto stepM = [&](int t, const double Wc[5], const double Hc[8], const double T[6], double M[8]) {
        // M~ = H~aug + W~' T over the structural non-zeros of W~ (A = I + E, B: 24 of 40 products), rows 6, 7 (the input block) first;
        // then Muu = M~[6..7][6..7] -> L D L' (backward stable, see systolic_factor) and column j of K~ = -Muu^-1 M~[u, :],
        // computed in every lane and hand-interleaved with the remaining rows of M~ so that the serial reciprocal / Newton
        // chain (~15 dependent instructions) hides behind independent FMAs
        // the accumulators start as copies of the H~aug column made HERE, after the operands have arrived: accumulating in place
        // into registers that a ds_read2_b64 delivers makes the compiler copy them out right behind the request (a stall)
        asm volatile("v_mov_b64_e32 %0, %8\nv_mov_b64_e32 %1, %9\nv_mov_b64_e32 %2, %10\nv_mov_b64_e32 %3, %11\n"
                     "v_mov_b64_e32 %4, %12\nv_mov_b64_e32 %5, %13\nv_mov_b64_e32 %6, %14\nv_mov_b64_e32 %7, %15\n"
                     : "=&v"(M[0]), "=&v"(M[1]), "=&v"(M[2]), "=&v"(M[3]), "=&v"(M[4]), "=&v"(M[5]), "=&v"(M[6]), "=&v"(M[7])
                     : "v"(Hc[0]), "v"(Hc[1]), "v"(Hc[2]), "v"(Hc[3]), "v"(Hc[4]), "v"(Hc[5]), "v"(Hc[6]), "v"(Hc[7]));
        double K0, K1, i00, l, i11, m66, m67, m77, e_, r_;
        asm volatile(
            "s_nop 1\n"
            "v_fmac_f64_dpp %6, %18, %23 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %7, %18, %23 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %6, %19, %24 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %7, %19, %24 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %7, %20, %25 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %6, %21, %26 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %7, %22, %27 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %18, %23 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %18, %23 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %13, %6 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %14, %6 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %15, %7 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %18, %23 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_rcp_f64_e32 %17, %13\n"
            "v_fmac_f64_dpp %4, %18, %23 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fma_f64 %16, -%13, %17, 1.0\n"
            "v_fmac_f64_dpp %5, %18, %23 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fma_f64 %17, %16, %17, %17\n"
            "v_fmac_f64_dpp %1, %19, %24 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fma_f64 %16, -%13, %17, 1.0\n"
            "v_fmac_f64_dpp %2, %19, %24 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fma_f64 %10, %16, %17, %17\n"
            "v_fmac_f64_dpp %3, %19, %24 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_mul_f64 %11, %14, %10\n"
            "v_fmac_f64_dpp %4, %19, %24 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fma_f64 %15, -%11, %14, %15\n"
            "v_fmac_f64_dpp %5, %19, %24 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_rcp_f64_e32 %17, %15\n"
            "v_fmac_f64_dpp %2, %20, %25 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fma_f64 %16, -%15, %17, 1.0\n"
            "v_fmac_f64_dpp %4, %20, %25 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fma_f64 %17, %16, %17, %17\n"
            "v_fmac_f64_dpp %5, %20, %25 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fma_f64 %16, -%15, %17, 1.0\n"
            "v_fmac_f64_dpp %3, %21, %26 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fma_f64 %12, %16, %17, %17\n"
            "v_fmac_f64_dpp %5, %21, %26 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fma_f64 %16, %11, %6, -%7\n"
            "v_fmac_f64_dpp %4, %22, %27 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_mul_f64 %9, %16, %12\n"
            "v_mul_f64 %16, %6, %10\n"
            "v_fmac_f64_dpp %5, %22, %27 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fma_f64 %8, -%11, %9, -%16\n"
            : "+v"(M[0]), "+v"(M[1]), "+v"(M[2]), "+v"(M[3]), "+v"(M[4]), "+v"(M[5]), "+v"(M[6]), "+v"(M[7]),
              "=&v"(K0), "=&v"(K1), "=&v"(i00), "=&v"(l), "=&v"(i11), "=&v"(m66), "=&v"(m67), "=&v"(m77), "=&v"(e_), "=&v"(r_)
            : "v"(Wc[0]), "v"(Wc[1]), "v"(Wc[2]), "v"(Wc[3]), "v"(Wc[4]), "v"(T[0]), "v"(T[1]), "v"(T[2]), "v"(T[3]), "v"(T[4]));
        M[5] += T[5];                                                  // r = 5 term of W~' T
        kp[HS * t] = K0; kp[HS * t + 8] = K1;
        fp[HS * t] = i00; fp[HS * t + 1] = l; fp[HS * t + 8] = i11;
        // P~+ = M~[0..5][0..5] + M~[0..5][u] K~   (M~[i][6] = M~[6][i] is lane i's register 6)
        asm volatile(
            "s_nop 1\n"
            "v_fmac_f64_dpp %0, %6, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %6, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %6, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %6, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %6, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %7, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %7, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %7, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %7, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %7, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %7, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            : "+v"(M[0]), "+v"(M[1]), "+v"(M[2]), "+v"(M[3]), "+v"(M[4]), "+v"(M[5])
            : "v"(M[6]), "v"(M[7]), "v"(K0), "v"(K1));
    };
    // Two operand sets (W, H) and two result sets (P) alternate: while stage t runs on (Wa, Ha) with P~ taken from Pb and its
    // result going to Pa, the operands of stage t-1 are requested into (Wb, Hb) right after the T block -- a stage is ~600
    // cycles of arithmetic, far more than the LDS latency of a lone wavefront.  No register copies between stages.
    double Wa[5], Ha[8], Wb[5], Hb[8], Pa[8], Pb[8], T[6];
#pragma unroll
    for (int r = 0; r < 6; r++) Pb[r] = hp[HS * N + r * 8];           // P~_N = H~aug_N[0..5][0..5]
    fetch(N - 1, Wa, Ha);
    // DPP hazards: 5 wait states after an EXEC write (the branches in front of this sweep), covered once here; inside the loop nothing
    // writes EXEC (stores are unconditional), and the 2 wait states after a VALU write of a DPP source are the s_nop 1 at the head of
    // every block
    asm volatile("s_nop 4");
    int t = N - 1;
    for (; t >= 1; t -= 2) {
        stepT(Wa, Pb, T); fetch(t - 1, Wb, Hb); stepM(t, Wa, Ha, T, Pa);
        stepT(Wb, Pa, T); fetch(t - 2, Wa, Ha); stepM(t - 1, Wb, Hb, T, Pb);      // t - 2 = -1 reads the (dead) block in front
    }
    if (t == 0) { stepT(Wa, Pb, T); stepM(0, Wa, Ha, T, Pa); }
}

// ------------------------------------------------------------------------------------------------------------------
// BLOCK-2 (PARTIALLY CONDENSED) RICCATI FACTORISATION.  The stage recursions are sequential in the stage index and a lone wavefront is bound by the
// length of its instruction stream, so the horizon is solved in PAIRS of stages: block m = stages a = 2m and b = 2m + 1, state x_a, inputs (u_a, u_b);
// x_b = A_a x_a + B_a u_a + b_a is eliminated (HPIPM's own partial condensing is this idea):
//     dynamics   x_{a+2} = A^ x_a + B^ (u_a, u_b) + b^,    A^ = A_b A_a,  B^ = [A_b B_a, B_b],  b^ = A_b b_a + b_b
//     cost       H^ = H~_a (embedded) + V' H~_b V,         V = [A_a b_a B_a 0; 0 1 0 0; 0 0 0 I]   (assembled lane-parallel by the lanes that own the odd stages)
// in homogeneous coordinates z^ = (x[5], 1, u_a[2], u_b[2]): 10 columns in lanes 0..9 of a 16-lane DPP row.  Same chain per block as rowpar_factor per stage --
// T = P~ W^ (30 DPP FMAs), M~ = H^ + W^' T (32), Muu 4 x 4 -> L D L' in every lane, K^ = -Muu^-1 M~[u, :] by substitution, P~+ = M~xx + M~xu K^ (24) --
// ~170 vector instructions for two stages instead of 2 x 112, 15 LDS instructions instead of 26, and the three vector recursions run over N / 2 blocks.
// Pivot order u_b, u_a (what the stage-by-stage recursion does implicitly: the later input first).
// ------------------------------------------------------------------------------------------------------------------
struct Blk2Lds {
    static constexpr int WS = 51, HS = 101;          // W^ rows 0..4 x 10 columns; H^ 10 x 10 row-major (symmetric: lane j reads ROW j as its column), odd strides
    static constexpr int RS = RowLds::HS;            // result blocks: the layout and stride of the dense stage blocks, so that the vector sweeps are the same code
    static constexpr int KROW = 8;                   // K^[u][j] at [u * 8 + j] (u < 4, j < 6: gains and feed-forward), L D L' factors at [FAC .. FAC + 9]
    static constexpr int FAC = 32;                   // 1/d0 1/d1 1/d2 1/d3 l10 l20 l30 l21 l31 l32
    // dead stores of idle lanes, inside the lane's OWN result block and apart from everything the sweep's workers store (as RowLds::DEADK / DEADF):
    // the four K^ values go to column 6 of the K^ rows (words 6, 14, 22, 30: workers fill columns 0..5, nothing reads column 6 during the factor
    // sweep), the ten factors to words 42..51 (behind FAC; the corrector's hand-off words kKK / kGX of rti_split_kernel.hpp are written after the sweep)
    static constexpr int DEADK = 6, DEADF = 42;
    static_assert(DEADK >= 6 && DEADK < KROW && DEADK + 3 * KROW < FAC, "dead K^ stores stay in the unused columns of the K^ rows");
    static_assert(DEADF >= FAC + 10 && DEADF + 10 <= RowLds::HS, "dead factor stores stay behind the factors and inside the block");
    static __host__ __device__ constexpr int blocks(int N) { return N / 2; }
    static __host__ __device__ constexpr int pad_front() { return HS; }                                  // the factor sweep requests one block ahead of block 0
    static __host__ __device__ constexpr int pad_rear() { return RowLds::AHEAD * RS; }                   // the vector sweeps request three blocks ahead
    // W^ blocks | pad | H^ blocks (M + 1: the terminal cost-to-go starts from H~aug_N) | result blocks (M + 1) | pad | per-stage [A b B] rows 0, 1 (16 words, N stages)
    static __host__ __device__ constexpr int total(int N) { return WS * blocks(N) + pad_front() + HS * (blocks(N) + 1) + RS * (blocks(N) + 1) + pad_rear() + 16 * N; }
    double *W, *H, *R, *S;
    __device__ __forceinline__ Blk2Lds(double *base, int N)
        : W(base + pad_front()), H(base + pad_front() + WS * blocks(N)), R(H + HS * (blocks(N) + 1)), S(R + RS * (blocks(N) + 1) + pad_rear()) {}
};

__device__ __forceinline__ void rowpar_factor2(int lane, int M, const Blk2Lds L, bool worker_row)
{
    constexpr int WS = Blk2Lds::WS, HS = Blk2Lds::HS, RS = Blk2Lds::RS;
    const int l15 = lane & 15, j = l15 < 10 ? l15 : 9;      // column; lanes 10..15 of a row shadow column 9 and store nothing
    const bool storeK = worker_row && l15 < 6, store0 = worker_row && l15 == 0;
    const double d5 = (j == 5) ? 1.0 : 0.0;
    double *kp = L.R + (storeK ? j : Blk2Lds::DEADK), *fp = L.R + (store0 ? Blk2Lds::FAC : Blk2Lds::DEADF);
    const double *wp = L.W + j, *hp = L.H + 10 * j;
    auto fetch = [&](int m, double Wc[5], double Hc[10]) {
#pragma unroll
        for (int k = 0; k < 5; k++) Wc[k] = wp[WS * m + 10 * k];
#pragma unroll
        for (int i = 0; i < 10; i++) Hc[i] = hp[HS * m + i];
    };
    // T = P~ W^ : rows 0..5 (exactly stepT of rowpar_factor; the 10 columns are a matter of which lanes hold operands)
    auto stepT = [&](const double Wc[5], const double Pc[6], double T[6]) {
#pragma unroll
        for (int r = 0; r < 6; r++) T[r] = Pc[r] * d5;
        asm volatile(
            "s_nop 1\n"
            "v_fmac_f64_dpp %0, %6, %12 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %6, %12 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %6, %12 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %6, %12 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %6, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %6, %12 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %6, %13 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %7, %13 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %7, %13 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %7, %13 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %7, %13 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %7, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %6, %14 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %7, %14 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %8, %14 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %8, %14 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %8, %14 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %8, %14 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %6, %15 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %7, %15 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %8, %15 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %9, %15 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %9, %15 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %9, %15 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %6, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %7, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %8, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %9, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %10, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %10, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]), "+v"(T[4]), "+v"(T[5])
            : "v"(Pc[0]), "v"(Pc[1]), "v"(Pc[2]), "v"(Pc[3]), "v"(Pc[4]), "v"(Pc[5]), "v"(Wc[0]), "v"(Wc[1]), "v"(Wc[2]), "v"(Wc[3]), "v"(Wc[4]));
    };
    // M~ = H^ + W^' T, input rows 6..9 first; structural zeros of W^ = [A^ b^ B^] skipped:
    //   row 0 of W^: columns 0, 2..9      row 1: columns 1..9      row 2: columns 2, 4, 5, 7, 9      row 3: columns 3, 5, 6, 8      row 4: columns 4, 5, 7, 9
    auto stepM = [&](int m, const double Wc[5], const double Hc[10], const double T[6], double M[10]) {
        asm volatile("v_mov_b64_e32 %0, %10\nv_mov_b64_e32 %1, %11\nv_mov_b64_e32 %2, %12\nv_mov_b64_e32 %3, %13\nv_mov_b64_e32 %4, %14\n"
                     "v_mov_b64_e32 %5, %15\nv_mov_b64_e32 %6, %16\nv_mov_b64_e32 %7, %17\nv_mov_b64_e32 %8, %18\nv_mov_b64_e32 %9, %19\n"
                     : "=&v"(M[0]), "=&v"(M[1]), "=&v"(M[2]), "=&v"(M[3]), "=&v"(M[4]), "=&v"(M[5]), "=&v"(M[6]), "=&v"(M[7]), "=&v"(M[8]), "=&v"(M[9])
                     : "v"(Hc[0]), "v"(Hc[1]), "v"(Hc[2]), "v"(Hc[3]), "v"(Hc[4]), "v"(Hc[5]), "v"(Hc[6]), "v"(Hc[7]), "v"(Hc[8]), "v"(Hc[9]));
        // rows 6..9 (the input block and what the factorisation needs first)
        asm volatile(
            "s_nop 1\n"
            "v_fmac_f64_dpp %0, %4, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %4, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %4, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %4, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %5, %10 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %5, %10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %5, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %5, %10 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %6, %11 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %6, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %7, %12 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %7, %12 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %8, %13 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %8, %13 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
            : "+v"(M[6]), "+v"(M[7]), "+v"(M[8]), "+v"(M[9])
            : "v"(Wc[0]), "v"(Wc[1]), "v"(Wc[2]), "v"(Wc[3]), "v"(Wc[4]), "v"(T[0]), "v"(T[1]), "v"(T[2]), "v"(T[3]), "v"(T[4]));
        // Muu (rows / columns 6..9) to every lane: m[p][q] = M~[6 + p][6 + q], q >= p, is lane (6 + q)'s register (6 + p)
        double m66, m67, m68, m69, m77, m78, m79, m88, m89, m99;
        asm volatile(
            "s_nop 1\n"
            "v_mov_b64_dpp %0, %10 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %1, %10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %2, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %3, %10 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %4, %11 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %5, %11 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %6, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %7, %12 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %8, %12 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %9, %13 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
            : "=&v"(m66), "=&v"(m67), "=&v"(m68), "=&v"(m69), "=&v"(m77), "=&v"(m78), "=&v"(m79), "=&v"(m88), "=&v"(m89), "=&v"(m99)
            : "v"(M[6]), "v"(M[7]), "v"(M[8]), "v"(M[9]));
        // rows 0..5 of M~ (independent of the factorisation below: the scheduler interleaves them with its reciprocal chains)
        asm volatile(
            "s_nop 1\n"
            "v_fmac_f64_dpp %0, %6, %11 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %6, %11 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %6, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %6, %11 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %6, %11 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %7, %12 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %7, %12 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %7, %12 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %7, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %7, %12 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %8, %13 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %8, %13 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %8, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %9, %14 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %9, %14 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %10, %15 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %10, %15 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            : "+v"(M[0]), "+v"(M[1]), "+v"(M[2]), "+v"(M[3]), "+v"(M[4]), "+v"(M[5])
            : "v"(Wc[0]), "v"(Wc[1]), "v"(Wc[2]), "v"(Wc[3]), "v"(Wc[4]), "v"(T[0]), "v"(T[1]), "v"(T[2]), "v"(T[3]), "v"(T[4]));
        M[5] += T[5];                                                  // row 5 of W^~ is e_5'
        // L D L' of Muu in the pivot order u_b (rows 8, 9), u_a (rows 6, 7): p = (8, 9, 6, 7).  a[r][c] = Muu[p_r][p_c]
        const double a00 = m88, a10 = m89, a20 = m68, a30 = m78, a11 = m99, a21 = m69, a31 = m79, a22 = m66, a32 = m67, a33 = m77;
        const double i0 = rcp_nr(a00);
        const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
        const double d1 = fma(-l10, a10, a11), t21 = fma(-l20, a10, a21), t31 = fma(-l30, a10, a31);
        const double i1 = rcp_nr(d1);
        const double l21 = t21 * i1, l31 = t31 * i1;
        const double d2 = fma(-l21, t21, fma(-l20, a20, a22)), t32 = fma(-l31, t21, fma(-l30, a20, a32));
        const double i2 = rcp_nr(d2);
        const double l32 = t32 * i2;
        const double d3 = fma(-l32, t32, fma(-l31, t31, fma(-l30, a30, a33)));
        const double i3 = rcp_nr(d3);
        // column j of K^ = -Muu^-1 M~[u, j]: forward, scale, backward substitution in the pivot order; K^ rows are stored in the ORIGINAL order (u_a, u_b)
        const double y0 = M[8], y1 = fma(-l10, y0, M[9]), y2 = fma(-l21, y1, fma(-l20, y0, M[6])), y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, y0, M[7])));
        const double k3 = -(y3 * i3);
        const double k2 = fma(-l32, k3, -(y2 * i2));
        const double k1 = fma(-l31, k3, fma(-l21, k2, -(y1 * i1)));
        const double k0 = fma(-l30, k3, fma(-l20, k2, fma(-l10, k1, -(y0 * i0))));
        const double K6 = k2, K7 = k3, K8 = k0, K9 = k1;                // original rows 6, 7 (u_a), 8, 9 (u_b)
        double *kb = kp + RS * m, *fb = fp + RS * m;
        kb[0] = K6; kb[8] = K7; kb[16] = K8; kb[24] = K9;
        fb[0] = i0; fb[1] = i1; fb[2] = i2; fb[3] = i3; fb[4] = l10; fb[5] = l20; fb[6] = l30; fb[7] = l21; fb[8] = l31; fb[9] = l32;
        // P~+ = M~[0..5][0..5] + M~[0..5][u] K^   (M~[i][6 + u] = M~[6 + u][i] is lane i's register 6 + u)
        asm volatile(
            "s_nop 1\n"
            "v_fmac_f64_dpp %0, %6, %10 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %6, %10 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %6, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %6, %10 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %6, %10 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %6, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %7, %11 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %7, %11 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %7, %11 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %7, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %7, %11 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %7, %11 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %8, %12 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %8, %12 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %8, %12 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %8, %12 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %8, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %8, %12 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %9, %13 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %9, %13 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %9, %13 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %9, %13 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %9, %13 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %9, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            : "+v"(M[0]), "+v"(M[1]), "+v"(M[2]), "+v"(M[3]), "+v"(M[4]), "+v"(M[5])
            : "v"(M[6]), "v"(M[7]), "v"(M[8]), "v"(M[9]), "v"(K6), "v"(K7), "v"(K8), "v"(K9));
    };
    double Wa[5], Ha[10], Wb[5], Hb[10], Pa[10], Pb[10], T[6];
#pragma unroll
    for (int r = 0; r < 6; r++) Pb[r] = hp[HS * M + r];              // P~_N = H~aug_N[0..5][0..5] (terminal block: row j, entries 0..5)
    fetch(M - 1, Wa, Ha);
    asm volatile("s_nop 4");
    int m = M - 1;
    for (; m >= 1; m -= 2) {
        stepT(Wa, Pb, T); fetch(m - 1, Wb, Hb); stepM(m, Wa, Ha, T, Pa);
        stepT(Wb, Pa, T); fetch(m - 2, Wa, Ha); stepM(m - 1, Wb, Hb, T, Pb);      // m - 2 = -1 reads the (dead) padding in front
    }
    if (m == 0) { stepT(Wa, Pb, T); stepM(0, Wa, Ha, T, Pa); }
}

// THE SAME SWEEP AS ONE ASM BLOCK (rowpar_factor_fast; text generated from the three blocks above, identical arithmetic and results).
// What it saves per stage: the eight accumulator copies (the H~aug column is requested straight INTO the accumulator registers -- inside
// one block no compiler can touch a register with a load in flight), four address updates, and every wait is counted exactly
// (s_waitcnt lgkmcnt(7) before a T block: the 4 accumulator requests and 3 stores younger than its W~ operands; lgkmcnt(10) before an M
// block: 3 stores and the 7 requests of the next stage).  Two stages per loop pass (register sets a / b), the operands of a stage are
// requested right after the T block of the stage before; an odd horizon ends with a single stage.  Fixed registers v100..v183.
// per-layout pieces of the text below: MPC_FACTOR_ASM_T(FAD) = dense blocks (RowLds), MPC_FACTOR_ASM_T(FAC) = compact blocks (RowLdsC)
#define FAD_D5 "%5"
#define FAD_HN "%6"
#define FAD_ODD "%7"
#define FAD_W_U01 "%0 offset0:41 offset1:49"
#define FAD_W_U23 "%0 offset0:57 offset1:65"
#define FAD_W_U4 "%0 offset:584"
#define FAD_W_L23 "%0 offset0:16 offset1:24"
#define FAD_W_L4 "%0 offset:256"
#define FAD_H_U01 "%1 offset0:65 offset1:73"
#define FAD_H_U23 "%1 offset0:81 offset1:89"
#define FAD_H_U45 "%1 offset0:97 offset1:105"
#define FAD_H_U67 "%1 offset0:113 offset1:121"
#define FAD_H_L67 "%1 offset0:48 offset1:56"
#define FAD_K_U "offset0:65 offset1:73"
#define FAD_F_U "offset0:65 offset1:66"
#define FAD_F_U8 "offset:584"
#define FAD_W_DEC "0xfffffd70"
#define FAD_H_DEC "0xfffffbf0"
#define FAD_MORE_DEC ""
#define FAC_D5 "%8"
#define FAC_HN "%9"
#define FAC_ODD "%10"
#define FAC_W_U01 "%0 offset0:21 offset1:29"
#define FAC_W_U23 "%5 offset0:0 offset1:1"
#define FAC_W_U4 "%5 offset:16"
#define FAC_W_L23 "%6 offset0:0 offset1:1"
#define FAC_W_L4 "%6 offset:16"
#define FAC_H_U01 "%1 offset0:57 offset1:65"
#define FAC_H_U23 "%1 offset0:73 offset1:81"
#define FAC_H_U45 "%1 offset0:89 offset1:97"
#define FAC_H_U67 "%7 offset0:57 offset1:58"
#define FAC_H_L67 "%7 offset0:0 offset1:1"
#define FAC_K_U "offset0:57 offset1:65"
#define FAC_F_U "offset0:57 offset1:58"
#define FAC_F_U8 "offset:520"
#define FAC_W_DEC "0xfffffeb0"
#define FAC_H_DEC "0xfffffc70"
#define FAC_MORE_DEC "v_add_u32_e32 %5, %11, %5\nv_add_u32_e32 %6, %11, %6\nv_add_u32_e32 %7, 0xfffffc70, %7\n"
#define MPC_FACTOR_ASM_T(V) \
        "ds_read2_b64 v[116:119], " V##_HN " offset0:0 offset1:8\n" \
        "ds_read2_b64 v[120:123], " V##_HN " offset0:16 offset1:24\n" \
        "ds_read2_b64 v[124:127], " V##_HN " offset0:32 offset1:40\n" \
        "ds_read2_b64 v[144:147], " V##_W_U01 "\n" \
        "ds_read2_b64 v[148:151], " V##_W_U23 "\n" \
        "ds_read_b64 v[152:153], " V##_W_U4 "\n" \
        "ds_read2_b64 v[100:103], " V##_H_U01 "\n" \
        "ds_read2_b64 v[104:107], " V##_H_U23 "\n" \
        "ds_read2_b64 v[108:111], " V##_H_U45 "\n" \
        "ds_read2_b64 v[112:115], " V##_H_U67 "\n" \
        "s_waitcnt lgkmcnt(0)\n" \
        "s_nop 4\n" \
        "s_cmp_eq_u32 %4, 0\n" \
        "s_cbranch_scc1 2f\n" \
        "1:\n" \
        "s_waitcnt lgkmcnt(7)\n" \
        "v_mul_f64 v[132:133], v[116:117], " V##_D5 "\n" \
        "v_mul_f64 v[134:135], v[118:119], " V##_D5 "\n" \
        "v_mul_f64 v[136:137], v[120:121], " V##_D5 "\n" \
        "v_mul_f64 v[138:139], v[122:123], " V##_D5 "\n" \
        "v_mul_f64 v[140:141], v[124:125], " V##_D5 "\n" \
        "v_mul_f64 v[142:143], v[126:127], " V##_D5 "\n" \
        "s_nop 1\n" \
        "v_fmac_f64_dpp v[132:133], v[116:117], v[144:145] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[116:117], v[144:145] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[116:117], v[144:145] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[116:117], v[144:145] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[116:117], v[144:145] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[116:117], v[144:145] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[132:133], v[116:117], v[146:147] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[118:119], v[146:147] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[118:119], v[146:147] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[118:119], v[146:147] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[118:119], v[146:147] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[118:119], v[146:147] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[132:133], v[116:117], v[148:149] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[118:119], v[148:149] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[120:121], v[148:149] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[120:121], v[148:149] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[120:121], v[148:149] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[120:121], v[148:149] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[132:133], v[116:117], v[150:151] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[118:119], v[150:151] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[120:121], v[150:151] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[122:123], v[150:151] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[122:123], v[150:151] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[122:123], v[150:151] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[132:133], v[116:117], v[152:153] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[118:119], v[152:153] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[120:121], v[152:153] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[122:123], v[152:153] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[124:125], v[152:153] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[124:125], v[152:153] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "ds_read2_b64 v[154:157], %0 offset0:0 offset1:8\n" \
        "ds_read2_b64 v[158:161], " V##_W_L23 "\n" \
        "ds_read_b64 v[162:163], " V##_W_L4 "\n" \
        "ds_read2_b64 v[116:119], %1 offset0:0 offset1:8\n" \
        "ds_read2_b64 v[120:123], %1 offset0:16 offset1:24\n" \
        "ds_read2_b64 v[124:127], %1 offset0:32 offset1:40\n" \
        "ds_read2_b64 v[128:131], " V##_H_L67 "\n" \
        "s_waitcnt lgkmcnt(10)\n" \
        "s_nop 1\n" \
        "v_fmac_f64_dpp v[112:113], v[144:145], v[132:133] row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[114:115], v[144:145], v[132:133] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[112:113], v[146:147], v[134:135] row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[114:115], v[146:147], v[134:135] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[114:115], v[148:149], v[136:137] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[112:113], v[150:151], v[138:139] row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[114:115], v[152:153], v[140:141] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[100:101], v[144:145], v[132:133] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[104:105], v[144:145], v[132:133] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b64_dpp v[174:175], v[112:113] row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b64_dpp v[176:177], v[112:113] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b64_dpp v[178:179], v[114:115] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[106:107], v[144:145], v[132:133] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_rcp_f64_e32 v[182:183], v[174:175]\n" \
        "v_fmac_f64_dpp v[108:109], v[144:145], v[132:133] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], -v[174:175], v[182:183], 1.0\n" \
        "v_fmac_f64_dpp v[110:111], v[144:145], v[132:133] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[182:183], v[180:181], v[182:183], v[182:183]\n" \
        "v_fmac_f64_dpp v[102:103], v[146:147], v[134:135] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], -v[174:175], v[182:183], 1.0\n" \
        "v_fmac_f64_dpp v[104:105], v[146:147], v[134:135] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[168:169], v[180:181], v[182:183], v[182:183]\n" \
        "v_fmac_f64_dpp v[106:107], v[146:147], v[134:135] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_mul_f64 v[170:171], v[176:177], v[168:169]\n" \
        "v_fmac_f64_dpp v[108:109], v[146:147], v[134:135] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[178:179], -v[170:171], v[176:177], v[178:179]\n" \
        "v_fmac_f64_dpp v[110:111], v[146:147], v[134:135] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_rcp_f64_e32 v[182:183], v[178:179]\n" \
        "v_fmac_f64_dpp v[104:105], v[148:149], v[136:137] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], -v[178:179], v[182:183], 1.0\n" \
        "v_fmac_f64_dpp v[108:109], v[148:149], v[136:137] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[182:183], v[180:181], v[182:183], v[182:183]\n" \
        "v_fmac_f64_dpp v[110:111], v[148:149], v[136:137] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], -v[178:179], v[182:183], 1.0\n" \
        "v_fmac_f64_dpp v[106:107], v[150:151], v[138:139] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[172:173], v[180:181], v[182:183], v[182:183]\n" \
        "v_fmac_f64_dpp v[110:111], v[150:151], v[138:139] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], v[170:171], v[112:113], -v[114:115]\n" \
        "v_fmac_f64_dpp v[108:109], v[152:153], v[140:141] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_mul_f64 v[166:167], v[180:181], v[172:173]\n" \
        "v_mul_f64 v[180:181], v[112:113], v[168:169]\n" \
        "v_fmac_f64_dpp v[110:111], v[152:153], v[140:141] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[164:165], -v[170:171], v[166:167], -v[180:181]\n" \
        "v_add_f64 v[110:111], v[110:111], v[142:143]\n" \
        "ds_write2_b64 %2, v[164:165], v[166:167] " V##_K_U "\n" \
        "ds_write2_b64 %3, v[168:169], v[170:171] " V##_F_U "\n" \
        "ds_write_b64 %3, v[172:173] " V##_F_U8 "\n" \
        "s_nop 1\n" \
        "v_fmac_f64_dpp v[100:101], v[112:113], v[164:165] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[102:103], v[112:113], v[164:165] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[104:105], v[112:113], v[164:165] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[106:107], v[112:113], v[164:165] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[108:109], v[112:113], v[164:165] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[110:111], v[112:113], v[164:165] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[100:101], v[114:115], v[166:167] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[102:103], v[114:115], v[166:167] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[104:105], v[114:115], v[166:167] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[106:107], v[114:115], v[166:167] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[108:109], v[114:115], v[166:167] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[110:111], v[114:115], v[166:167] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "s_waitcnt lgkmcnt(7)\n" \
        "v_mul_f64 v[132:133], v[100:101], " V##_D5 "\n" \
        "v_mul_f64 v[134:135], v[102:103], " V##_D5 "\n" \
        "v_mul_f64 v[136:137], v[104:105], " V##_D5 "\n" \
        "v_mul_f64 v[138:139], v[106:107], " V##_D5 "\n" \
        "v_mul_f64 v[140:141], v[108:109], " V##_D5 "\n" \
        "v_mul_f64 v[142:143], v[110:111], " V##_D5 "\n" \
        "s_nop 1\n" \
        "v_fmac_f64_dpp v[132:133], v[100:101], v[154:155] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[100:101], v[154:155] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[100:101], v[154:155] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[100:101], v[154:155] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[100:101], v[154:155] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[100:101], v[154:155] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[132:133], v[100:101], v[156:157] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[102:103], v[156:157] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[102:103], v[156:157] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[102:103], v[156:157] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[102:103], v[156:157] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[102:103], v[156:157] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[132:133], v[100:101], v[158:159] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[102:103], v[158:159] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[104:105], v[158:159] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[104:105], v[158:159] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[104:105], v[158:159] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[104:105], v[158:159] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[132:133], v[100:101], v[160:161] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[102:103], v[160:161] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[104:105], v[160:161] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[106:107], v[160:161] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[106:107], v[160:161] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[106:107], v[160:161] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[132:133], v[100:101], v[162:163] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[102:103], v[162:163] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[104:105], v[162:163] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[106:107], v[162:163] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[108:109], v[162:163] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[108:109], v[162:163] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_add_u32_e32 %0, " V##_W_DEC ", %0\n" \
        "v_add_u32_e32 %1, " V##_H_DEC ", %1\n" V##_MORE_DEC \
        "ds_read2_b64 v[144:147], " V##_W_U01 "\n" \
        "ds_read2_b64 v[148:151], " V##_W_U23 "\n" \
        "ds_read_b64 v[152:153], " V##_W_U4 "\n" \
        "ds_read2_b64 v[100:103], " V##_H_U01 "\n" \
        "ds_read2_b64 v[104:107], " V##_H_U23 "\n" \
        "ds_read2_b64 v[108:111], " V##_H_U45 "\n" \
        "ds_read2_b64 v[112:115], " V##_H_U67 "\n" \
        "s_waitcnt lgkmcnt(10)\n" \
        "s_nop 1\n" \
        "v_fmac_f64_dpp v[128:129], v[154:155], v[132:133] row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[130:131], v[154:155], v[132:133] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[128:129], v[156:157], v[134:135] row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[130:131], v[156:157], v[134:135] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[130:131], v[158:159], v[136:137] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[128:129], v[160:161], v[138:139] row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[130:131], v[162:163], v[140:141] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[116:117], v[154:155], v[132:133] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[120:121], v[154:155], v[132:133] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b64_dpp v[174:175], v[128:129] row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b64_dpp v[176:177], v[128:129] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b64_dpp v[178:179], v[130:131] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[122:123], v[154:155], v[132:133] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_rcp_f64_e32 v[182:183], v[174:175]\n" \
        "v_fmac_f64_dpp v[124:125], v[154:155], v[132:133] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], -v[174:175], v[182:183], 1.0\n" \
        "v_fmac_f64_dpp v[126:127], v[154:155], v[132:133] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[182:183], v[180:181], v[182:183], v[182:183]\n" \
        "v_fmac_f64_dpp v[118:119], v[156:157], v[134:135] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], -v[174:175], v[182:183], 1.0\n" \
        "v_fmac_f64_dpp v[120:121], v[156:157], v[134:135] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[168:169], v[180:181], v[182:183], v[182:183]\n" \
        "v_fmac_f64_dpp v[122:123], v[156:157], v[134:135] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_mul_f64 v[170:171], v[176:177], v[168:169]\n" \
        "v_fmac_f64_dpp v[124:125], v[156:157], v[134:135] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[178:179], -v[170:171], v[176:177], v[178:179]\n" \
        "v_fmac_f64_dpp v[126:127], v[156:157], v[134:135] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_rcp_f64_e32 v[182:183], v[178:179]\n" \
        "v_fmac_f64_dpp v[120:121], v[158:159], v[136:137] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], -v[178:179], v[182:183], 1.0\n" \
        "v_fmac_f64_dpp v[124:125], v[158:159], v[136:137] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[182:183], v[180:181], v[182:183], v[182:183]\n" \
        "v_fmac_f64_dpp v[126:127], v[158:159], v[136:137] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], -v[178:179], v[182:183], 1.0\n" \
        "v_fmac_f64_dpp v[122:123], v[160:161], v[138:139] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[172:173], v[180:181], v[182:183], v[182:183]\n" \
        "v_fmac_f64_dpp v[126:127], v[160:161], v[138:139] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], v[170:171], v[128:129], -v[130:131]\n" \
        "v_fmac_f64_dpp v[124:125], v[162:163], v[140:141] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_mul_f64 v[166:167], v[180:181], v[172:173]\n" \
        "v_mul_f64 v[180:181], v[128:129], v[168:169]\n" \
        "v_fmac_f64_dpp v[126:127], v[162:163], v[140:141] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[164:165], -v[170:171], v[166:167], -v[180:181]\n" \
        "v_add_f64 v[126:127], v[126:127], v[142:143]\n" \
        "ds_write2_b64 %2, v[164:165], v[166:167] offset0:0 offset1:8\n" \
        "ds_write2_b64 %3, v[168:169], v[170:171] offset0:0 offset1:1\n" \
        "ds_write_b64 %3, v[172:173] offset:64\n" \
        "v_add_u32_e32 %2, " V##_H_DEC ", %2\n" \
        "v_add_u32_e32 %3, " V##_H_DEC ", %3\n" \
        "s_nop 1\n" \
        "v_fmac_f64_dpp v[116:117], v[128:129], v[164:165] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[118:119], v[128:129], v[164:165] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[120:121], v[128:129], v[164:165] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[122:123], v[128:129], v[164:165] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[124:125], v[128:129], v[164:165] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[126:127], v[128:129], v[164:165] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[116:117], v[130:131], v[166:167] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[118:119], v[130:131], v[166:167] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[120:121], v[130:131], v[166:167] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[122:123], v[130:131], v[166:167] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[124:125], v[130:131], v[166:167] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[126:127], v[130:131], v[166:167] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "s_sub_u32 %4, %4, 1\n" \
        "s_cmp_lg_u32 %4, 0\n" \
        "s_cbranch_scc1 1b\n" \
        "2:\n" \
        "s_cmp_eq_u32 " V##_ODD ", 0\n" \
        "s_cbranch_scc1 3f\n" \
        "s_waitcnt lgkmcnt(7)\n" \
        "v_mul_f64 v[132:133], v[116:117], " V##_D5 "\n" \
        "v_mul_f64 v[134:135], v[118:119], " V##_D5 "\n" \
        "v_mul_f64 v[136:137], v[120:121], " V##_D5 "\n" \
        "v_mul_f64 v[138:139], v[122:123], " V##_D5 "\n" \
        "v_mul_f64 v[140:141], v[124:125], " V##_D5 "\n" \
        "v_mul_f64 v[142:143], v[126:127], " V##_D5 "\n" \
        "s_nop 1\n" \
        "v_fmac_f64_dpp v[132:133], v[116:117], v[144:145] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[116:117], v[144:145] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[116:117], v[144:145] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[116:117], v[144:145] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[116:117], v[144:145] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[116:117], v[144:145] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[132:133], v[116:117], v[146:147] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[118:119], v[146:147] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[118:119], v[146:147] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[118:119], v[146:147] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[118:119], v[146:147] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[118:119], v[146:147] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[132:133], v[116:117], v[148:149] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[118:119], v[148:149] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[120:121], v[148:149] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[120:121], v[148:149] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[120:121], v[148:149] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[120:121], v[148:149] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[132:133], v[116:117], v[150:151] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[118:119], v[150:151] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[120:121], v[150:151] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[122:123], v[150:151] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[122:123], v[150:151] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[122:123], v[150:151] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[132:133], v[116:117], v[152:153] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[134:135], v[118:119], v[152:153] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[136:137], v[120:121], v[152:153] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[138:139], v[122:123], v[152:153] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[140:141], v[124:125], v[152:153] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[142:143], v[124:125], v[152:153] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "s_waitcnt lgkmcnt(3)\n" \
        "s_nop 1\n" \
        "v_fmac_f64_dpp v[112:113], v[144:145], v[132:133] row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[114:115], v[144:145], v[132:133] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[112:113], v[146:147], v[134:135] row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[114:115], v[146:147], v[134:135] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[114:115], v[148:149], v[136:137] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[112:113], v[150:151], v[138:139] row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[114:115], v[152:153], v[140:141] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[100:101], v[144:145], v[132:133] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[104:105], v[144:145], v[132:133] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b64_dpp v[174:175], v[112:113] row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b64_dpp v[176:177], v[112:113] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b64_dpp v[178:179], v[114:115] row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[106:107], v[144:145], v[132:133] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_rcp_f64_e32 v[182:183], v[174:175]\n" \
        "v_fmac_f64_dpp v[108:109], v[144:145], v[132:133] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], -v[174:175], v[182:183], 1.0\n" \
        "v_fmac_f64_dpp v[110:111], v[144:145], v[132:133] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[182:183], v[180:181], v[182:183], v[182:183]\n" \
        "v_fmac_f64_dpp v[102:103], v[146:147], v[134:135] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], -v[174:175], v[182:183], 1.0\n" \
        "v_fmac_f64_dpp v[104:105], v[146:147], v[134:135] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[168:169], v[180:181], v[182:183], v[182:183]\n" \
        "v_fmac_f64_dpp v[106:107], v[146:147], v[134:135] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_mul_f64 v[170:171], v[176:177], v[168:169]\n" \
        "v_fmac_f64_dpp v[108:109], v[146:147], v[134:135] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[178:179], -v[170:171], v[176:177], v[178:179]\n" \
        "v_fmac_f64_dpp v[110:111], v[146:147], v[134:135] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_rcp_f64_e32 v[182:183], v[178:179]\n" \
        "v_fmac_f64_dpp v[104:105], v[148:149], v[136:137] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], -v[178:179], v[182:183], 1.0\n" \
        "v_fmac_f64_dpp v[108:109], v[148:149], v[136:137] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[182:183], v[180:181], v[182:183], v[182:183]\n" \
        "v_fmac_f64_dpp v[110:111], v[148:149], v[136:137] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], -v[178:179], v[182:183], 1.0\n" \
        "v_fmac_f64_dpp v[106:107], v[150:151], v[138:139] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[172:173], v[180:181], v[182:183], v[182:183]\n" \
        "v_fmac_f64_dpp v[110:111], v[150:151], v[138:139] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[180:181], v[170:171], v[112:113], -v[114:115]\n" \
        "v_fmac_f64_dpp v[108:109], v[152:153], v[140:141] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_mul_f64 v[166:167], v[180:181], v[172:173]\n" \
        "v_mul_f64 v[180:181], v[112:113], v[168:169]\n" \
        "v_fmac_f64_dpp v[110:111], v[152:153], v[140:141] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fma_f64 v[164:165], -v[170:171], v[166:167], -v[180:181]\n" \
        "v_add_f64 v[110:111], v[110:111], v[142:143]\n" \
        "ds_write2_b64 %2, v[164:165], v[166:167] " V##_K_U "\n" \
        "ds_write2_b64 %3, v[168:169], v[170:171] " V##_F_U "\n" \
        "ds_write_b64 %3, v[172:173] " V##_F_U8 "\n" \
        "s_nop 1\n" \
        "v_fmac_f64_dpp v[100:101], v[112:113], v[164:165] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[102:103], v[112:113], v[164:165] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[104:105], v[112:113], v[164:165] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[106:107], v[112:113], v[164:165] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[108:109], v[112:113], v[164:165] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[110:111], v[112:113], v[164:165] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[100:101], v[114:115], v[166:167] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[102:103], v[114:115], v[166:167] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[104:105], v[114:115], v[166:167] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[106:107], v[114:115], v[166:167] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[108:109], v[114:115], v[166:167] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[110:111], v[114:115], v[166:167] row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "3:\n" \
        "s_waitcnt lgkmcnt(0)\n"

// THE SAME ONE-BLOCK SWEEP ON THE COMPACT STAGE BLOCKS (RowLdsC; the FAC_ pieces of the text above: same instructions, same registers, same
// wait counts -- the W~ and H~aug operands arrive through the same number of requests).  What differs: the offsets (21 + 57 words per stage), rows 2..4 of W~
// through two per-lane pointers (%5 upper / %6 lower stage of a pair; lane 5 walks the b_t words of the stage blocks with the per-lane stride %11, the other
// lanes stay on the wavefront's constant table), rows 6, 7 of H~aug through the per-lane pointer %7 (rti_kernel.hpp, RowLdsC).
#define MPC_FACTOR_ASM MPC_FACTOR_ASM_T(FAD)
#define MPC_FACTOR_ASM_C MPC_FACTOR_ASM_T(FAC)
#define MPC_FACTOR_ASM_CLOBBERS "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "scc", "vcc", "memory"

__device__ __forceinline__ void rowpar_factor_fast(int lane, int N, const RowLds L, bool worker_row)
{
    constexpr int WS = RowLds::WS, HS = RowLds::HS;
    const int j = lane & 7, l15 = lane & 15;
    const bool store = worker_row && l15 < 6, store0 = worker_row && l15 == 0;
    const double d5 = (j == 5) ? 1.0 : 0.0;
    // base pointers: the LOWER stage of the first pair (stage N - 2; the upper one sits one block above, as an immediate offset)
    uint32_t wp = lds_address(L.W + j + WS * (N - 2)), hp = lds_address(L.H + j + HS * (N - 2));
    uint32_t kp = lds_address(L.R + (store ? j : 48) + HS * (N - 2)), fp = lds_address(L.R + (store0 ? 6 : 50) + HS * (N - 2));
    const uint32_t hN = lds_address(L.H + j + HS * N);
    int passes = N >> 1;
    const int odd = N & 1;
    asm volatile(MPC_FACTOR_ASM : "+v"(wp), "+v"(hp), "+v"(kp), "+v"(fp), "+s"(passes) : "v"(d5), "v"(hN), "s"(odd) : MPC_FACTOR_ASM_CLOBBERS);
}

__device__ __forceinline__ void rowpar_factor_fast_c(int lane, int N, const RowLdsC L, bool worker_row)
{
    constexpr int WS = RowLdsC::WS, HS = RowLdsC::HS;
    const int j = lane & 7, l15 = lane & 15;
    const bool store = worker_row && l15 < 6, store0 = worker_row && l15 == 0;
    const double d5 = (j == 5) ? 1.0 : 0.0;
    // base pointers: the LOWER stage of the first pair (stage N - 2; the upper one sits one block above, as an immediate offset)
    uint32_t wp = lds_address(L.W + j + WS * (N - 2)), hp = lds_address(L.H + j + HS * (N - 2));
    uint32_t kp = lds_address(L.R + (store ? j : RowLdsC::DEADK) + HS * (N - 2)), fp = lds_address(L.R + (store0 ? 6 : RowLdsC::DEADF) + HS * (N - 2));
    const uint32_t hN = lds_address(L.H + j + HS * N);
    // rows 2..4 of W~: lane 5 reads b_t[2..4] of the stage block, the other lanes the constant table (stride 0)
    uint32_t bpu = lds_address(j == 5 ? L.W + 16 + WS * (N - 1) : L.C + 3 * j), bpl = lds_address(j == 5 ? L.W + 16 + WS * (N - 2) : L.C + 3 * j);
    const uint32_t bstride = j == 5 ? (uint32_t)(-2 * WS * 8) : 0u;
    // rows 6, 7 of H~aug: (zero, zero) | (H[5][6], H[5][7]) | (H66, zero) | (zero, H77)
    uint32_t h6p = lds_address(L.H + (j < 5 ? 49 : (j == 5 ? 46 : (j == 6 ? 48 : 50))) + HS * (N - 2));
    int passes = N >> 1;
    const int odd = N & 1;
    asm volatile(MPC_FACTOR_ASM_C : "+v"(wp), "+v"(hp), "+v"(kp), "+v"(fp), "+s"(passes), "+v"(bpu), "+v"(bpl), "+v"(h6p)
                 : "v"(d5), "v"(hN), "s"(odd), "v"(bstride) : MPC_FACTOR_ASM_CLOBBERS);
}

// ------------------------------------------------------------------------------------------------------------------
// THE FACTOR SWEEP ON THE MATRIX CORES, 4x4x4 BLOCKS (one instance per wavefront; dense stage blocks).  v_mfma_f64_4x4x4_4b multiplies four
// independent 4 x 4 blocks per instruction: the 8 x 8 matrices of a stage are 2 x 2 blocks, block (I, J) in block slot q = 2 I + J.  Lane layouts
// (scripts/bin_src/mfma4_test.hip, found by unit probes): A[q][i][k] in lane i + 4 q + 16 k, B[q][k][j] in lane j + 4 q + 16 k, C/D[q][i][j] in lane
// j + 4 q + 16 i.  So a result register serves as a B operand as it is and as an A operand TRANSPOSED block by block, and a block slot is a
// quad of a 16-lane DPP row: moving blocks between slots is a v_mov_b32_dpp row_shl / row_shr / row_ror by 4 or 8 lanes under a bank mask.
// Per stage (same homogeneous formulation as rowpar_factor):
//   T  = P~ W~            2 MFMAs   A: slot (I, J) <- P~(K, I) (as A: its transpose = P~(I, K), P~ symmetric);  B: W~(K, J) from LDS
//   M~ = H~aug + W~' T    2 MFMAs   A: W~(K, I) from LDS (as A: transposed);  B: slot (I, J) <- T(K, J);  C: H~aug from LDS
//   Muu -> L D L' on wave-uniform scalars (v_readlane), rows 6, 7 of M~ (DPP rows 2, 3 of block row 1) -> K~ after one v_permlane16_swap
//   P~+ = M~ + M~[:, u] K~  1 MFMA  A: slot (I, J) <- M~(1, I) (transposed: columns 6, 7 of M~);  B: rows 2, 3 = K~ (slots replicated), rows 0, 1 zero
// Rows / columns 6, 7 of P~ carry finite garbage: rows 6, 7 of the padded W~ are zero, so they never reach T's rows 0..5 or M~.
// ------------------------------------------------------------------------------------------------------------------
template <int CTRL, int BANKS>
__device__ __forceinline__ double dpp_blocks(double old, double src)       // quads named by BANKS take src moved by CTRL, the others keep old
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xf, BANKS, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xf, BANKS, false);
    return __hiloint2double(hi, lo);
}
// slot (I, J) <- block (0, I) of x: quads [0, 0, 1, 1];  block (1, I): quads [2, 2, 3, 3];  slot (I, J) <- block (K, J): [0, 1, 0, 1] / [2, 3, 2, 3]
__device__ __forceinline__ double blocks_0011(double x) { return dpp_blocks<0x118, 0x8>(dpp_blocks<0x114, 0x6>(x, x), x); }     // row_shr:4 -> quads 1, 2; row_shr:8 -> quad 3
__device__ __forceinline__ double blocks_2233(double x) { return dpp_blocks<0x108, 0x1>(dpp_blocks<0x104, 0x6>(x, x), x); }     // row_shl:4 -> quads 1, 2; row_shl:8 -> quad 0
__device__ __forceinline__ double blocks_0101(double x) { return dpp_blocks<0x128, 0xc>(x, x); }                                // row_ror:8 -> quads 2, 3
__device__ __forceinline__ double blocks_2323(double x) { return dpp_blocks<0x128, 0x3>(x, x); }                                // row_ror:8 -> quads 0, 1

template <class LT>
__device__ __forceinline__ void mfma4_factor(int lane, int N, const LT L)
{
    static_assert(!LT::COMPACT, "dense stage blocks");
    constexpr int WS = LT::WS, HS = LT::HS;
    const int dr = lane >> 4, I = (lane >> 3) & 1, J = (lane >> 2) & 1, jj = lane & 3;
    const double *wb = L.W + dr * 8 + 4 * J + jj;              // W~(K, J)[dr][jj]: + 32 K
    const double *wa = L.W + dr * 8 + 4 * I + jj;              // W~(K, I)[dr][jj]
    const double *hd = L.H + (4 * I + dr) * 8 + 4 * J + jj;    // H~aug(I, J)[dr][jj]
    const bool cst = dr >= 1;                                   // rows 5..7 of the padded W~: e_5', 0, 0
    const double cwb = (dr == 1 && 4 * J + jj == 5) ? 1.0 : 0.0, cwa = (dr == 1 && 4 * I + jj == 5) ? 1.0 : 0.0;
    const int c = 4 * J + jj;                                   // column of K~ this lane holds after the solve (DPP rows 2, 3, block row 1)
    double *kp = L.R + ((I == 1 && c < 6 && dr >= 2) ? (dr == 2 ? c : 8 + c) : LT::DEADK);
    double *fp = L.R + (lane == 0 ? 6 : (lane == 1 ? 7 : (lane == 2 ? 14 : LT::DEADF)));
    double P = hd[HS * N];
    double wb0 = wb[WS * (N - 1)], wb1 = wb[WS * (N - 1) + 32], wa0 = wa[WS * (N - 1)], wa1 = wa[WS * (N - 1) + 32], H = hd[HS * (N - 1)];
    for (int t = N - 1; t >= 0; t--) {
        const double b0 = wb0, b1 = cst ? cwb : wb1, a0 = wa0, a1 = cst ? cwa : wa1, Hc = H;
        // operands of the stage in front (t - 1 = -1 reads the dead block in front of the instance's blocks)
        wb0 = wb[WS * (t - 1)]; wb1 = wb[WS * (t - 1) + 32]; wa0 = wa[WS * (t - 1)]; wa1 = wa[WS * (t - 1) + 32]; H = hd[HS * (t - 1)];
        double T = __builtin_amdgcn_mfma_f64_4x4x4f64(blocks_0011(P), b0, 0.0, 0, 0, 0);
        T = __builtin_amdgcn_mfma_f64_4x4x4f64(blocks_2233(P), b1, T, 0, 0, 0);
        double M = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, blocks_0101(T), Hc, 0, 0, 0);
        M = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, blocks_2323(T), M, 0, 0, 0);
        // Muu = M~[6..7][6..7]: block (1, 1), rows / columns 2, 3 -> lanes 46, 47, 63
        const double m66 = lane_value(M, 46), m67 = lane_value(M, 47), m77 = lane_value(M, 63);
        double i00 = __builtin_amdgcn_rcp(m66);
        i00 = fma(fma(-m66, i00, 1.0), i00, i00); i00 = fma(fma(-m66, i00, 1.0), i00, i00);
        const double l = m67 * i00;
        const double d1 = fma(-l, m67, m77);
        double i11 = __builtin_amdgcn_rcp(d1);
        i11 = fma(fma(-d1, i11, 1.0), i11, i11); i11 = fma(fma(-d1, i11, 1.0), i11, i11);
        // rows 6 (DPP row 2) and 7 (DPP row 3) of M~ side by side: v_permlane16_swap exchanges row 3 of its first with row 2 of its second operand
        const auto slo = __builtin_amdgcn_permlane16_swap(__double2loint(M), __double2loint(M), false, false);
        const auto shi = __builtin_amdgcn_permlane16_swap(__double2hiint(M), __double2hiint(M), false, false);
        const double X = __hiloint2double(shi[0], slo[0]), Y = __hiloint2double(shi[1], slo[1]);      // X: row 3 <- row 2 (M~[6][.]);  Y: row 2 <- row 3 (M~[7][.])
        const double g0 = dr == 3 ? X : M, g1 = dr == 3 ? M : Y;
        const double K1 = fma(l, g0, -g1) * i11;
        const double K0 = fma(-l, K1, -(g0 * i00));
        const double Kr = dr == 2 ? K0 : K1;
        const double Kb = blocks_2323(dr >= 2 ? Kr : 0.0);
        P = __builtin_amdgcn_mfma_f64_4x4x4f64(blocks_2233(M), Kb, M, 0, 0, 0);
        kp[HS * t] = Kr;
        fp[HS * t] = lane == 0 ? i00 : (lane == 1 ? l : i11);
    }
}

// ROW-PARALLEL VECTOR RECURSIONS.  With the closed-loop matrix Acl_t = A_t + B_t K_t (5 x 5, computed by the lane that owns
// stage t, for all stages at once) in LDS, the forward rollout  dx_{t+1} = Acl_t dx_t + c_t  and the backward (adjoint)
// recursion  p_t = c~_t + Acl_t' p_{t+1}  are 5 x 5 matrix-vector products per stage: lane r of the instance's first DPP row
// holds row r (forward) resp. column r (backward) of Acl_t and element r of the travelling vector, and one product is five
// v_fmac_f64_dpp with the vector element broadcast from lane j (~20 wave instructions per stage instead of ~60 for the
// one-lane systolic sweeps).  Every stage's vector is left in LDS for the lane that owns the stage.
// LDS use (per stage, inside the H~aug_t block that is dead after the factorisation): [0..29] rows [Acl[r][0..4], c[r]],
// [30..34] c~, [35..39] dx_t, [40..44] p_t.
struct RowVec { static constexpr int ACL = 0, RS = 6, CT = 30, X = 35, P = 40; };

// FWD: dx_{t+1} = Acl_t dx_t + c_t for t = 0..N-1 (dx_0 from the stage-0 block).  !FWD: p_t = c~_t + Acl_t' p_{t+1} for
// t = N-1..1 (p_N = c~_N; p_0, the multiplier of the fixed initial state, is not needed).  A stage costs ~13 instructions, far
// less than the LDS latency of a lone wavefront, so the operands travel through a ring of four register sets and are requested
// three stages ahead (RowLds::AHEAD; running past the ends is harmless); stores are unconditional (idle lanes hit block tails).
template <bool FWD>
__device__ __forceinline__ void rowpar_vector(int lane, int N, const RowLds L, bool worker_row)
{
    constexpr int D = 4, HS = RowLds::HS;
    const int r = lane & 7, rc = r < 5 ? r : 0;
    const bool store = worker_row && (lane & 15) < 5;
    const int Q = FWD ? N : N - 1;                                  // number of stage steps
    double v = FWD ? L.R[RowVec::X + r] : L.R[HS * N + RowVec::CT + r];
    if (!FWD && store) L.R[HS * N + RowVec::P + r] = v;
    // step q works on stage t = q (FWD) / N-1-q (!FWD); src points at the lane's operands of step 0, dst at the place of
    // the vector produced by step 0; both move by one stage block per step (dst of an idle lane stays in the rear padding)
    const double *src = L.R + (FWD ? 0 : HS * (N - 1)) + RowVec::ACL + (FWD ? rc * RowVec::RS : rc);
    double *dst = L.R + (FWD ? HS : HS * (N - 1)) + (store ? (FWD ? RowVec::X : RowVec::P) + r : RowLds::TAIL);   // idle lanes: one dead word of the block
    constexpr int dstep = FWD ? HS : -HS;
    double A[D][5], c[D];
    const double *srcc = src + (FWD ? 5 : RowVec::CT);
    auto fetch = [&](const double *blk, const double *blkc, double a[5], double &cc) {
#pragma unroll
        for (int k = 0; k < 5; k++) a[k] = blk[FWD ? k : k * RowVec::RS];      // row r (FWD) / column r of Acl_t
        cc = *blkc;                                                            // c_t[r] / c~_t[r]
    };
    // The accumulator starts as a copy of c made INSIDE the block: an in-out operand that arrives as half of a ds_read2_b64
    // tuple would be copied out by the compiler behind an s_waitcnt right after the request was issued (a full LDS latency).
    auto stage = [&](const double a[5], double cc, double *out) {
        double acc;
        asm volatile(
                "s_nop 1\n"
                "v_mov_b64_e32 %0, %2\n"
                "v_fmac_f64_dpp %0, %1, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %0, %1, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %0, %1, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %0, %1, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %0, %1, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                : "=&v"(acc) : "v"(v), "v"(cc), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]));
        v = acc;
        *out = v;
    };
    constexpr int SS = FWD ? HS : -HS;                              // stage stride of src
#pragma unroll
    for (int u = 0; u < D - 1; u++) fetch(src + u * SS, srcc + u * SS, A[u], c[u]);
    int qb = 0;
    for (; qb + D <= Q; qb += D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            fetch(src + (u + D - 1) * SS, srcc + (u + D - 1) * SS, A[(u + D - 1) % D], c[(u + D - 1) % D]);
            stage(A[u], c[u], dst + u * SS);
        }
        src += D * SS; srcc += D * SS; dst += D * dstep;
    }
#pragma unroll
    for (int u = 0; u < D - 1; u++) {
        if (qb + u < Q) {
            fetch(src + (u + D - 1) * SS, srcc + (u + D - 1) * SS, A[(u + D - 1) % D], c[(u + D - 1) % D]);
            stage(A[u], c[u], dst + u * SS);
        }
    }
}

// HAND-SCHEDULED VARIANT of rowpar_vector (same arithmetic, same LDS layout, same results bit for bit).  For a lone wavefront an
// FP64 instruction issues every 5 cycles (a dependent one after 8.4) but an LDS instruction costs ~14 whatever its width
// (scripts/bin_src/lds_issue_test.hip), so the four LDS instructions of a stage weigh as much as its eight VALU instructions.
// Here ONE set of operand requests serves TWO stages: the travelling vector alternates between the halves of the DPP row --
//   even step: source = lanes 0..4 of B (row_newbcast:k),   accumulator A in lanes 8..12, whose own registers hold the even stage's rows,
//   odd step:  source = lanes 8..12 of A (row_newbcast:8+k), accumulator B in lanes 0..4,  whose registers hold the odd stage's rows --
// and one store writes both results (the halves merged by two masked v_mov_b32_dpp).  The whole sweep is one asm block: a ring of two
// operand sets in fixed registers (ds_read2_b64 needs register tuples whose halves are used separately, which asm operands cannot
// express), requested one pair ahead and placed into the bubbles of the dependent FMA chain; s_waitcnt lgkmcnt(1) at the head of a pair
// (only the previous pair's store is younger than its operands).  The compiler never sees a register with a load in flight: the block
// drains the LDS queue before it ends.  Four stages per loop pass; the leading N mod 4 stages run through the plain path; the adjoint
// sweep also computes p_0 (unused) so that both directions take N steps.
#define MPC_VEC_ASM_FWD_S(S) \
        "v_mov_b64_e32 v[230:231], %0\n" \
        "ds_read2_b64 v[180:183], %1 offset0:0 offset1:1\n" \
        "ds_read2_b64 v[184:187], %1 offset0:2 offset1:3\n" \
        "ds_read2_b64 v[188:191], %1 offset0:4 offset1:5\n" \
        "v_add_u32_e32 %1, " S ", %1\n" \
        "s_waitcnt lgkmcnt(0)\n" \
        "1:\n" \
        "s_waitcnt lgkmcnt(1)\n" \
        "v_mov_b64_e32 v[228:229], v[190:191]\n" \
        "ds_read2_b64 v[192:195], %1 offset0:0 offset1:1\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[180:181] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "ds_read2_b64 v[196:199], %1 offset0:2 offset1:3\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[182:183] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "ds_read2_b64 v[200:203], %1 offset0:4 offset1:5\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[184:185] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_add_u32_e32 %1, " S ", %1\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[186:187] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[188:189] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b64_e32 v[230:231], v[190:191]\n" \
        "s_nop 0\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[180:181] row_newbcast:8 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[182:183] row_newbcast:9 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[184:185] row_newbcast:10 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[186:187] row_newbcast:11 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[188:189] row_newbcast:12 row_mask:0xf bank_mask:0xf\n" \
        "s_nop 1\n" \
        "v_mov_b32_dpp v228, v230 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n" \
        "v_mov_b32_dpp v229, v231 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n" \
        "ds_write_b64 %2, v[228:229]\n" \
        "v_add_u32_e32 %2, " S ", %2\n" \
        "s_waitcnt lgkmcnt(1)\n" \
        "v_mov_b64_e32 v[228:229], v[202:203]\n" \
        "ds_read2_b64 v[180:183], %1 offset0:0 offset1:1\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[192:193] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "ds_read2_b64 v[184:187], %1 offset0:2 offset1:3\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[194:195] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "ds_read2_b64 v[188:191], %1 offset0:4 offset1:5\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[196:197] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_add_u32_e32 %1, " S ", %1\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[198:199] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[200:201] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b64_e32 v[230:231], v[202:203]\n" \
        "s_nop 0\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[192:193] row_newbcast:8 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[194:195] row_newbcast:9 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[196:197] row_newbcast:10 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[198:199] row_newbcast:11 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[200:201] row_newbcast:12 row_mask:0xf bank_mask:0xf\n" \
        "s_nop 1\n" \
        "v_mov_b32_dpp v228, v230 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n" \
        "v_mov_b32_dpp v229, v231 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n" \
        "ds_write_b64 %2, v[228:229]\n" \
        "v_add_u32_e32 %2, " S ", %2\n" \
        "s_sub_u32 %3, %3, 1\n" \
        "s_cmp_lg_u32 %3, 0\n" \
        "s_cbranch_scc1 1b\n" \
        "s_waitcnt lgkmcnt(0)\n"

#define MPC_VEC_ASM_BWD_S(S) \
        "v_mov_b64_e32 v[230:231], %0\n" \
        "ds_read2_b64 v[180:183], %1 offset0:0 offset1:6\n" \
        "ds_read2_b64 v[184:187], %1 offset0:12 offset1:18\n" \
        "ds_read2_b64 v[188:191], %1 offset0:24 offset1:30\n" \
        "v_add_u32_e32 %1, " S ", %1\n" \
        "s_waitcnt lgkmcnt(0)\n" \
        "1:\n" \
        "s_waitcnt lgkmcnt(1)\n" \
        "v_mov_b64_e32 v[228:229], v[190:191]\n" \
        "ds_read2_b64 v[192:195], %1 offset0:0 offset1:6\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[180:181] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "ds_read2_b64 v[196:199], %1 offset0:12 offset1:18\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[182:183] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "ds_read2_b64 v[200:203], %1 offset0:24 offset1:30\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[184:185] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_add_u32_e32 %1, " S ", %1\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[186:187] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[188:189] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b64_e32 v[230:231], v[190:191]\n" \
        "s_nop 0\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[180:181] row_newbcast:8 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[182:183] row_newbcast:9 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[184:185] row_newbcast:10 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[186:187] row_newbcast:11 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[188:189] row_newbcast:12 row_mask:0xf bank_mask:0xf\n" \
        "s_nop 1\n" \
        "v_mov_b32_dpp v228, v230 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n" \
        "v_mov_b32_dpp v229, v231 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n" \
        "ds_write_b64 %2, v[228:229]\n" \
        "v_add_u32_e32 %2, " S ", %2\n" \
        "s_waitcnt lgkmcnt(1)\n" \
        "v_mov_b64_e32 v[228:229], v[202:203]\n" \
        "ds_read2_b64 v[180:183], %1 offset0:0 offset1:6\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[192:193] row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "ds_read2_b64 v[184:187], %1 offset0:12 offset1:18\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[194:195] row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "ds_read2_b64 v[188:191], %1 offset0:24 offset1:30\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[196:197] row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_add_u32_e32 %1, " S ", %1\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[198:199] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[228:229], v[230:231], v[200:201] row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b64_e32 v[230:231], v[202:203]\n" \
        "s_nop 0\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[192:193] row_newbcast:8 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[194:195] row_newbcast:9 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[196:197] row_newbcast:10 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[198:199] row_newbcast:11 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f64_dpp v[230:231], v[228:229], v[200:201] row_newbcast:12 row_mask:0xf bank_mask:0xf\n" \
        "s_nop 1\n" \
        "v_mov_b32_dpp v228, v230 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n" \
        "v_mov_b32_dpp v229, v231 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x3\n" \
        "ds_write_b64 %2, v[228:229]\n" \
        "v_add_u32_e32 %2, " S ", %2\n" \
        "s_sub_u32 %3, %3, 1\n" \
        "s_cmp_lg_u32 %3, 0\n" \
        "s_cbranch_scc1 1b\n" \
        "s_waitcnt lgkmcnt(0)\n"

// stride of two stage blocks in bytes: 2 * HS * 8 (dense layout HS = 65: 1040; compact layout HS = 57: 912), negative for the adjoint sweep
#define MPC_VEC_ASM_FWD MPC_VEC_ASM_FWD_S("0x410")
#define MPC_VEC_ASM_BWD MPC_VEC_ASM_BWD_S("0xfffffbf0")
#define MPC_VEC_ASM_FWD_C MPC_VEC_ASM_FWD_S("0x390")
#define MPC_VEC_ASM_BWD_C MPC_VEC_ASM_BWD_S("0xfffffc70")

#define MPC_VEC_ASM_CLOBBERS "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v228", "v229", "v230", "v231", "scc", "memory"

template <bool FWD, class LT>
__device__ __forceinline__ void rowpar_vector_fast(int lane, int N, const LT L, bool worker_row)
{
    static_assert(LT::HS == 65 || LT::HS == 57, "the asm blocks carry the stage stride as a literal");
    constexpr int HS = LT::HS, SS = FWD ? HS : -HS;
    const int r = lane & 7, rc = r < 5 ? r : 0;
    const bool store = worker_row && r < 5;
    double v = FWD ? L.R[RowVec::X + r] : L.R[HS * N + RowVec::CT + r];
    if (!FWD && worker_row && (lane & 15) < 5) L.R[HS * N + RowVec::P + r] = v;
    // block of the first stage step: its operands (src) and the place of the vector it produces (dst)
    const double *src = L.R + (FWD ? 0 : HS * (N - 1)) + RowVec::ACL + (FWD ? rc * RowVec::RS : rc);
    double *dst = L.R + (FWD ? HS : HS * (N - 1));
    for (int q = 0; q < (N & 3); q++) {        // leading N mod 4 stages, every lane with r = lane & 7 (both halves of the row hold the vector)
        double a[5], cc, acc;
#pragma unroll
        for (int k = 0; k < 5; k++) a[k] = src[FWD ? k : k * RowVec::RS];
        cc = src[FWD ? 5 : RowVec::CT];
        asm volatile(
                "s_nop 1\n"
                "v_mov_b64_e32 %0, %2\n"
                "v_fmac_f64_dpp %0, %1, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %0, %1, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %0, %1, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %0, %1, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                "v_fmac_f64_dpp %0, %1, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                : "=&v"(acc) : "v"(v), "v"(cc), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]));
        v = acc;
        dst[(worker_row && (lane & 15) < 5) ? (FWD ? RowVec::X : RowVec::P) + r : LT::TAIL] = v;
        src += SS; dst += SS;
    }
    int passes = N >> 2;
    if (passes > 0) {
        // lanes 8..15 of a row take the first stage of a pair, lanes 0..7 the second one
        const bool first = (lane & 8) != 0;
        uint32_t ra = lds_address(src + (first ? 0 : SS));
        uint32_t rd = lds_address(dst + (first ? 0 : SS) + (store ? (FWD ? RowVec::X : RowVec::P) + r : LT::TAIL));
        if constexpr (LT::HS == 65) {
            if (FWD) asm volatile(MPC_VEC_ASM_FWD : "+v"(v), "+v"(ra), "+v"(rd), "+s"(passes) : : MPC_VEC_ASM_CLOBBERS);
            else     asm volatile(MPC_VEC_ASM_BWD : "+v"(v), "+v"(ra), "+v"(rd), "+s"(passes) : : MPC_VEC_ASM_CLOBBERS);
        } else {
            if (FWD) asm volatile(MPC_VEC_ASM_FWD_C : "+v"(v), "+v"(ra), "+v"(rd), "+s"(passes) : : MPC_VEC_ASM_CLOBBERS);
            else     asm volatile(MPC_VEC_ASM_BWD_C : "+v"(v), "+v"(ra), "+v"(rd), "+s"(passes) : : MPC_VEC_ASM_CLOBBERS);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// MATRIX-CORE RICCATI FACTORISATION (one instance per wavefront).  In homogeneous coordinates
//     x~ = (x[5], 1),   z~ = (x[5], 1, ua, ual),   W~ = [A b_r B; 0 1 0] (6 x 8),   P~ = [P q; q' .] (6 x 6)
// one stage is   M~ = H~aug + W~' P~ W~ (8 x 8),   K~ = -Muu^-1 M~[u,:],   P~+ = M~ + M~[:,u] K~,
// which carries the Hessian AND the gradient recursion.  Each product is a v_mfma_f64_16x16x4 (D = A B + C, the 8 x 8 /
// 6 x 8 blocks sit in the top-left corner of the 16 x 16 tiles).  Layouts (verified on gfx950): A[i][k] and B[k][j] in
// lane i + 16 k resp. j + 16 k; D[row][col] in lane col + 16 (row & 3), register row >> 2.  Because P~ and M~ are
// symmetric, an accumulator tile is directly the A or B operand of the next product, so the whole chain
//     T = P~ W~ (2 MFMA, K = 6)  ->  M~ = W~' T + H~aug (2)  ->  K~ (2, LDL' substitution)  ->  P~+ (1)
// runs without any cross-lane movement; only the 2 x 2 block Muu goes through v_readlane for its LDL' factors.
// The per-stage operands W~_t (constant during the solve) and H~aug_t (per interior-point iteration) are staged in LDS
// by the lanes that own the stages; K, k and the LDL' factors return to them through LDS.
// ------------------------------------------------------------------------------------------------------------------
typedef double mfma_acc_t __attribute__((ext_vector_type(4)));
#ifndef MPC_SYM_PERIOD
#define MPC_SYM_PERIOD 4
#endif
static constexpr int kSymPeriod = MPC_SYM_PERIOD;

struct MfmaLds {
    double *WB, *HC, *KO, *TT;
    __device__ __forceinline__ MfmaLds(double *base, int N)
    {
        WB = base;                    // [N][2][32]   W~_t operand registers: element (row k, col j) at [k >> 2][(k & 3) * 8 + j]
        HC = WB + 64 * N;             // [N+1][2][32] H~aug_t in accumulator layout, same indexing (row, col)
        KO = HC + 64 * (N + 1);       // [N][16]      K~ rows (2 x 6: K[a][0..4], k[a]) then 1/d0, l, 1/d1
        TT = KO + 16 * N;             // [2][32]      scratch tile for the transpose
    }
    static __host__ __device__ constexpr int doubles(int N) { return 64 * N + 64 * (N + 1) + 16 * N + 64; }
    static __device__ __forceinline__ int at(int row, int col) { return (row >> 2) * 32 + (row & 3) * 8 + col; }
};

__device__ __forceinline__ void mfma_factor(int lane, int N, const MfmaLds L, double rs)
{
    const int col = lane & 15, grp = lane >> 4;
    const bool in8 = col < 8;
    const int e = grp * 8 + (in8 ? col : 0);
    // terminal cost-to-go: P~_N = H~aug_N (rows / cols 0..5)
    mfma_acc_t Pt = {0.0, 0.0, 0.0, 0.0};
    Pt[0] = in8 ? L.HC[64 * N + e] : 0.0;
    Pt[1] = in8 ? L.HC[64 * N + 32 + e] : 0.0;
    // operands of stage t are loaded one stage ahead, so that the LDS latency hides behind the dependent MFMA chain
    double nB0 = in8 ? L.WB[64 * (N - 1) + e] : 0.0, nB1 = in8 ? L.WB[64 * (N - 1) + 32 + e] : 0.0;
    double nC0 = in8 ? L.HC[64 * (N - 1) + e] : 0.0, nC1 = in8 ? L.HC[64 * (N - 1) + 32 + e] : 0.0;
    for (int t = N - 1; t >= 0; t--) {
        double B0 = nB0, B1 = nB1;
        mfma_acc_t C = {nC0, nC1, 0.0, 0.0};
        if (t > 0) {
            nB0 = in8 ? L.WB[64 * (t - 1) + e] : 0.0; nB1 = in8 ? L.WB[64 * (t - 1) + 32 + e] : 0.0;
            nC0 = in8 ? L.HC[64 * (t - 1) + e] : 0.0; nC1 = in8 ? L.HC[64 * (t - 1) + 32 + e] : 0.0;
        }
        // column 5 of W~ is the affine term: rows 0..4 carry the dynamics residual r_b = rs * b_t, row 5 is the constant 1
        if (col == 5) { B0 *= rs; if (grp == 0) B1 *= rs; }
        mfma_acc_t T = {0.0, 0.0, 0.0, 0.0};
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(Pt[0], B0, T, 0, 0, 0);      // P~[:, 0..3] W~[0..3, :]
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(Pt[1], B1, T, 0, 0, 0);      // P~[:, 4..7] W~[4..7, :]  (rows 6, 7 of W~ are zero)
        mfma_acc_t M = C;
        M = __builtin_amdgcn_mfma_f64_16x16x4f64(B0, T[0], M, 0, 0, 0);       // W~'[:, 0..3] T[0..3, :]
        M = __builtin_amdgcn_mfma_f64_16x16x4f64(B1, T[1], M, 0, 0, 0);       // W~'[:, 4..7] T[4..7, :]
        // Muu = M~[6..7][6..7] -> LDL' (backward stable, see systolic_factor), explicit inverse from the factors
        const double m66 = lane_value(M[1], 6 + 32), m67 = lane_value(M[1], 7 + 32), m77 = lane_value(M[1], 7 + 48);
        const double i00 = rcp_nr(m66);
        const double l = m67 * i00;
        const double i11 = rcp_nr(m77 - l * m67);
        // K~ = -Muu^-1 M~[6..7, :] by the LDL' SUBSTITUTION, as two small products (never through an explicit inverse:
        // its entries cancel catastrophically on the gradient column when Muu is ill-conditioned):
        //   Y = D^-1 L^-1 M~[u,:]  (rows 2, 3),   K~ = -L^-T Y  (rows 2, 3, i.e. the B operand of the next product)
        double A1c = 0.0, A2c = 0.0;
        if (lane == 2 + 32) { A1c = i00; A2c = -1.0; }      // A[2][2]
        if (lane == 3 + 32) { A1c = -l * i11; }             // A[3][2]
        if (lane == 3 + 48) { A1c = i11; A2c = -1.0; }      // A[3][3]
        if (lane == 2 + 48) { A2c = l; }                    // A[2][3]
        mfma_acc_t Y = {0.0, 0.0, 0.0, 0.0};
        Y = __builtin_amdgcn_mfma_f64_16x16x4f64(A1c, M[1], Y, 0, 0, 0);
        mfma_acc_t Kt = {0.0, 0.0, 0.0, 0.0};
        Kt = __builtin_amdgcn_mfma_f64_16x16x4f64(A2c, Y[0], Kt, 0, 0, 0);
        // P~+ = M~ + M~[:, 6..7] K~   (A[i][k = 2, 3] = M~[6 + k - 2][i] by symmetry)
        const double Am = grp >= 2 ? M[1] : 0.0;
        Pt = __builtin_amdgcn_mfma_f64_16x16x4f64(Am, Kt[0], M, 0, 0, 0);
        // symmetrise P~+ (both triangles are computed independently and the next product reads the tile transposed; without
        // this the antisymmetric rounding part grows along the horizon, as in any Riccati recursion): transpose through LDS
        // (every kSymPeriod-th stage is enough to keep the antisymmetric part at rounding level; the transpose costs ~400 cycles)
        if ((t % kSymPeriod) == 0) {
            if (in8) { L.TT[MfmaLds::at(grp, col)] = Pt[0]; L.TT[MfmaLds::at(grp + 4, col)] = Pt[1]; }
            wave_sync();
            if (in8) {
                Pt[0] = 0.5 * (Pt[0] + L.TT[MfmaLds::at(col, grp)]);
                Pt[1] = 0.5 * (Pt[1] + L.TT[MfmaLds::at(col, grp + 4)]);
            }
            wave_sync();
        }
        if (grp >= 2 && col < 6) L.KO[16 * t + (grp - 2) * 6 + col] = Kt[0];
        if (lane == 0) { L.KO[16 * t + 12] = i00; L.KO[16 * t + 13] = l; L.KO[16 * t + 14] = i11; }
    }
}

// Corrector right-hand side: homogeneous dynamics, reuses K and the LDL' factors; linear term gc (7) per lane.
__device__ __forceinline__ void systolic_corrector(int stage, int N, const StageLin &S, const double gc[7], StageFac &F)
{
    double pv[5] = {0, 0, 0, 0, 0};
    for (int t = N; t >= 0; t--) {      // unconditional arithmetic, masked keep: see systolic_factor
        const double m0 = gc[0] + S.dua(pv), m1 = gc[1] + S.dual(pv);
        const double mx[5] = {gc[2] + pv[0], gc[3] + pv[1], gc[4] + S.dpsi(pv), gc[5] + S.dv(pv), gc[6] + S.dom(pv)};
        const double k1 = fma(F.l, m0, -m1) * F.i11;
        const double k0 = fma(-F.l, k1, -(m0 * F.i00));
        if (stage == t) { F.k0 = k0; F.k1 = k1; }
#pragma unroll
        for (int c = 0; c < 5; c++) pv[c] = from_right(mx[c] + F.K0[c] * m0 + F.K1[c] * m1);
    }
}

// Forward rollout of the Newton step: dx travels left to right, every lane keeps its own (du_t, dx_t) in dz.
// x_init is the initial-condition residual (meaningful in lane 0); bbr = rs * b_t for the affine (predictor) pass.
template <bool AFFINE>
__device__ __forceinline__ void systolic_rollout(int stage, int N, const StageLin &S, const StageFac &F, const double x_init[5],
                                                 const double bbr[5], double dz[7])
{
    double x[5];
#pragma unroll
    for (int c = 0; c < 5; c++) x[c] = AFFINE ? x_init[c] : 0.0;
    for (int t = 0; t <= N; t++) {      // unconditional arithmetic, masked keep: see systolic_factor
        double u0 = F.k0, u1 = F.k1, xo[5];
#pragma unroll
        for (int c = 0; c < 5; c++) { u0 += F.K0[c] * x[c]; u1 += F.K1[c] * x[c]; }
        if (stage == t) {
            dz[0] = u0; dz[1] = u1;
#pragma unroll
            for (int c = 0; c < 5; c++) dz[2 + c] = x[c];
        }
        xo[0] = x[0] + S.a02 * x[2] + S.a03 * x[3] + S.a04 * x[4] + S.b00 * u0 + S.b01 * u1;
        xo[1] = x[1] + S.a12 * x[2] + S.a13 * x[3] + S.a14 * x[4] + S.b10 * u0 + S.b11 * u1;
        xo[2] = x[2] + S.dt * x[4] + S.h2 * u1;
        xo[3] = x[3] + S.dt * u0;
        xo[4] = x[4] + S.dt * u1;
        if (AFFINE) {
#pragma unroll
            for (int c = 0; c < 5; c++) xo[c] += bbr[c];
        }
#pragma unroll
        for (int c = 0; c < 5; c++) x[c] = from_left(xo[c]);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The solve kernel.  grid = ceil(batch / (64 / G)) workgroups of one wavefront; dynamic LDS: row-parallel operands + look-ahead staging.
// ------------------------------------------------------------------------------------------------------------------
// FACT selects the Riccati factorisation sweep: 0 one-lane systolic, 1 matrix cores (G = 64 only), 2 row-parallel DPP on dense LDS stage
// blocks (RowLds), 3 row-parallel DPP on compact stage blocks (RowLdsC: what three instances per wavefront, G = 21, and long horizons need
// to keep four wavefronts on a CU).
// MASKED: the problem has p.n_obst < NOBST obstacles (any count the reference's N_OBST may take); the rows of obstacle j >= p.n_obst do not exist
// (skipped by wave-uniform branches), the input arrays are strided by p.n_obst, and the unused position slots replicate the last obstacle so
// that everything computed from them stays finite.  Instantiated for one instance per wavefront only; the stage-split kernel takes any count.
template <int NOBST, int G, int FACT, bool MASKED = false>
__global__ __launch_bounds__(64) void rti_solve_kernel(const KParams p)
{
    const int nact = MASKED ? p.n_obst : NOBST;
#define ROW_OFF(j) (MASKED && (j) >= nact)
#define OBST_IN(j) (MASKED ? ((j) < nact ? (j) : nact - 1) : (j))
    constexpr bool USE_MFMA = FACT == 1, ROWPAR = FACT >= 2, COMPACT = FACT == 3;
    // five and more obstacle pairs: recomputable row state is not carried (see obst_view below).  Measured per obstacle count (scripts/ab_workload.py,
    // 65536 random scenarios): 3 obstacles -4.6 % (their state fits the registers: recomputing only adds instructions), 5 obstacles +8 % at N = 20 and
    // +10 % at N = 10 (84 B of scratch and 58 accumulation registers less), 10 obstacles: the difference between 784 and 12 B of scratch
    constexpr bool LEAN = NOBST >= 5;
    constexpr bool PLDS = LEAN && COMPACT && NOBST >= 10;      // ... and, with ten, the obstacle positions of a stage stay in LDS behind the compact stage blocks (with five they
                                                               // would cost the CU its fourth wavefront of three instances: 44 KB each)
    constexpr bool SLDS = PLDS;                 // ... as do the A / B entries (re-read from the W~ block) and the initial residual (front padding); on the
                                                // 3-obstacle kernels, which do not spill, the same move costs 2 % (measured at C3) and is not made
    static_assert(!USE_MFMA || G == 64, "the matrix-core factorisation maps one instance per wavefront");
    static_assert(G != 21 || COMPACT, "three instances per wavefront exist for the row-parallel sweeps on compact stage blocks only");
    constexpr int IPW = 64 / G;               // instances per wavefront (G = 21: three, lanes [0,21), [21,42), [42,63); lane 63 idles)
    using LT = typename std::conditional<COMPACT, RowLdsC, RowLds>::type;     // LDS layout of the stage blocks
    const int lane = threadIdx.x;
    const int slot = (G == 21) ? seg21_slot(lane) : lane / G;
    const int inst_raw = blockIdx.x * IPW + slot;
    const bool valid = inst_raw < p.batch;    // tail wavefront: surplus slots replay the last instance and store nothing
    const int sidx = valid ? inst_raw : p.batch - 1;
    int inst_ = p.order ? p.order[sidx] : sidx;           // instance scheduling: which instance this slot works on
    // one instance per wavefront: the index is wave-uniform -> a scalar register.  Not a speed matter: a per-lane copy of it has to survive the
    // whole interior point in the register file, and a live-range split copy of exactly this value is what the toolchain once placed in front of
    // the exec restore of an if / else join (scripts/isa_audit.py rule P1) -- scalar spills (v_writelane) ignore EXEC
    if constexpr (G == 64) inst_ = __builtin_amdgcn_readfirstlane(inst_);
    const int inst = inst_;
    const int N = p.N;
    const int i = lane - slot * G;            // this lane's stage
    const bool act = (i <= N);
    const bool has_u = (i < N);
    const bool xb = (i >= 1) && (i < N || (i == N && p.bx_terminal));
    const double dt = p.dt, h2 = p.h2;

    // ---- load (coalesced: consecutive lanes read consecutive stages of this instance's records) ----
    double x0v[5], gl[2];
#pragma unroll
    for (int c = 0; c < 5; c++) x0v[c] = p.x0[(size_t)inst * 5 + c];
    gl[0] = p.goal[(size_t)inst * 2]; gl[1] = p.goal[(size_t)inst * 2 + 1];
    if constexpr (G == 64) { gl[0] = wave_uniform(gl[0]); gl[1] = wave_uniform(gl[1]); }      // one instance per wavefront: the goal is the same in every lane -> scalar registers
                                                                                              // (with ten obstacles these four registers are the difference between 28 and 0 B of scratch)
    double *Xg = p.X + (size_t)inst * (N + 1) * 5, *Ug = p.U + (size_t)inst * N * 2;
    // episode already finished (goal reached): the instance idles, nothing of it is touched
    const bool ep_done = (p.fused & kFuseMetrics) && p.ep_flags && (p.ep_flags[inst] & 1);
    // obstacle parameters of this stage: explicit P (reference API, parameterize_model) or the look-ahead computed here
    extern __shared__ double lds_raw[];
    const MfmaLds ML(lds_raw, N);             // used only when USE_MFMA (the launch sizes the allocation accordingly)
    // RL: the stage blocks of THIS lane's instance (row phases); RS: those of the instance whose sweeps this lane works on -- the same,
    // except with three instances per wavefront, where the sweep of instance w runs in DPP row w (lanes 16 w .. 16 w + 15)
    auto blocks_of = [&](int which) {
        if constexpr (COMPACT) return RowLdsC(lds_raw + RowLdsC::CT + RowLdsC::pad_front(N) + which * RowLdsC::per_instance(N), N, lds_raw);
        else return RowLds(lds_raw + (ROWPAR ? RowLds::pad_front(N) + which * RowLds::per_instance(N) : 0), N);
    };
    const LT RL = blocks_of(slot);                                  // used only when ROWPAR
    const LT RS = blocks_of(G == 21 ? (lane >> 4 < 3 ? lane >> 4 : 2) : slot);
    const bool sweep_worker = (G == 21) ? lane < 48 : i < 16;
    double *lds_P = lds_raw + (USE_MFMA ? MfmaLds::doubles(N) : (ROWPAR ? RowLds::total(N, IPW) : 0));
    double pxy[PLDS ? 1 : NOBST][2];
    // PLDS: this lane's stage positions, resident in LDS for the whole solve
    const double *myP = lds_raw;
    if (p.obst || PLDS) {
        // (compact blocks: the look-ahead is staged in the instance's own H~aug region, which is first written after the positions have been
        // read -- or, PLDS, behind the blocks, where it stays)
        double *Pl = PLDS ? lds_raw + RowLdsC::positions_at(N, IPW) + (size_t)slot * (N + 1) * NOBST * 2
                          : (COMPACT ? RL.H : lds_P + (size_t)slot * (N + 1) * NOBST * 2);
        if (!p.obst) {              // PLDS with explicit parameters (parameterize_model): every stage lane copies its row of P
            if (act) {
                const double *Pg = p.P + ((size_t)inst * (N + 1) + i) * nact * 2;
#pragma unroll
                for (int e = 0; e < 2 * NOBST; e++) Pl[i * NOBST * 2 + e] = Pg[2 * OBST_IN(e >> 1) + (e & 1)];
            }
        } else if (2 * NOBST <= G) {
            if (i < 2 * NOBST) {   // lane i walks coordinate i & 1 of obstacle i >> 1 through the horizon (Obstacle.predict_trajectory, visualization.py:62-79)
                const int j = OBST_IN(i >> 1), c = i & 1;
                const double *o = p.obst + ((size_t)inst * nact + j) * 4;
                double q = o[c], v = (c == 0 && !p.world.bug_compat_predict) ? o[2] : o[3];      // defect D1: vx = self.vy (:69)
                const double lo = c ? p.world.ymin : p.world.xmin, hi = c ? p.world.ymax : p.world.xmax;
                Pl[i] = q;
                for (int k = 1; k <= N; k++) {
                    coord_advance(lo, hi, dt, q, v);
                    Pl[k * NOBST * 2 + i] = q;
                }
            }
        } else if (i < NOBST) {       // lane j = i walks obstacle j through the horizon
            const double *o = p.obst + ((size_t)inst * nact + OBST_IN(i)) * 4;
            double ox = o[0], oy = o[1], ovy = o[3];
            double ovx = p.world.bug_compat_predict ? o[3] : o[2];
            Pl[i * 2] = ox; Pl[i * 2 + 1] = oy;
            for (int k = 1; k <= N; k++) {
                obstacle_advance(p.world, dt, ox, ovx, oy, ovy);
                Pl[(k * NOBST + i) * 2] = ox; Pl[(k * NOBST + i) * 2 + 1] = oy;
            }
        }
        wave_sync();
        if constexpr (PLDS) myP = Pl + (act ? i : 0) * NOBST * 2;
        else {
#pragma unroll
            for (int j = 0; j < NOBST; j++) { pxy[j][0] = act ? Pl[(i * NOBST + j) * 2] : 0.0; pxy[j][1] = act ? Pl[(i * NOBST + j) * 2 + 1] : 0.0; }
        }
    } else {
        const double *Pg = p.P + ((size_t)inst * (N + 1) + (act ? i : 0)) * nact * 2;
#pragma unroll
        for (int j = 0; j < NOBST; j++) { pxy[j][0] = act ? Pg[2 * OBST_IN(j)] : 0.0; pxy[j][1] = act ? Pg[2 * OBST_IN(j) + 1] : 0.0; }
    }
    // position of obstacle j at this lane's stage
    auto pos_x = [&](int j) { if constexpr (PLDS) return myP[2 * j]; else return pxy[j][0]; };
    auto pos_y = [&](int j) { if constexpr (PLDS) return myP[2 * j + 1]; else return pxy[j][1]; };
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
        const double alpha_i = p.alpha ? p.alpha[(size_t)inst * (N + 1) + (act ? i : N)] : scale * (double)(N - i) / (double)N;
        zpen = alpha_i * (has_u ? p.ss : 1.0);
    }
    const bool vs = act && (i >= 1) && (p.soft_h ? (zpen > 0.0) : true);   // obstacle rows present at this stage
    const bool soft = p.soft_h != 0;

    // ---- linearise ----
    double lin0 = 0.0;
    double d0[5] = {0, 0, 0, 0, 0};
    StageLin S;                    // this stage's non-trivial entries of A = dF/dx, B = dF/du (zero for the terminal lane)
    S.a02 = S.a03 = S.a04 = S.a12 = S.a13 = S.a14 = S.b00 = S.b01 = S.b10 = S.b11 = 0.0; S.dt = dt; S.h2 = h2;
    double bb[5] = {0, 0, 0, 0, 0};  // dynamics defect b_i = F(x_i, u_i) - x_{i+1} of the SQP iterate
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
        if constexpr (SLDS) {       // LEAN on compact blocks: the initial-condition residual is only ever used by the lane of stage 0 -- it waits in the
#pragma unroll                      // front padding of the stage blocks (5 words per instance) instead of in five registers of every lane
            for (int c = 0; c < 5; c++) lds_raw[RowLdsC::CT + slot * 5 + c] = d0[c];
        }
    }
    if (USE_MFMA) {
        // zero the operand tiles once, then every stage lane writes its W~_t = [A b B; 0 1 0] (cols: x0..x4, 1, ua, ual)
        for (int k = lane; k < 64 * N + 64 * (N + 1); k += 64) ML.WB[k] = 0.0;
        wave_sync();
        if (has_u) {
            double *w = ML.WB + 64 * i;
            const double Arow[5][5] = {{1.0, 0.0, S.a02, S.a03, S.a04}, {0.0, 1.0, S.a12, S.a13, S.a14}, {0.0, 0.0, 1.0, 0.0, dt},
                                       {0.0, 0.0, 0.0, 1.0, 0.0}, {0.0, 0.0, 0.0, 0.0, 1.0}};
            const double Brow[5][2] = {{S.b00, S.b01}, {S.b10, S.b11}, {0.0, h2}, {dt, 0.0}, {0.0, dt}};
#pragma unroll
            for (int k = 0; k < 5; k++) {
#pragma unroll
                for (int c = 0; c < 5; c++) w[MfmaLds::at(k, c)] = Arow[k][c];
                w[MfmaLds::at(k, 5)] = bb[k];
                w[MfmaLds::at(k, 6)] = Brow[k][0]; w[MfmaLds::at(k, 7)] = Brow[k][1];
            }
            w[MfmaLds::at(5, 5)] = 1.0;
        }
    }
    if constexpr (COMPACT) {    // compact blocks (RowLdsC): rows 0, 1 of W~_t, the shared table of the constant rows 2..4, the two zero words of H~aug_t
        if (has_u) {
            double *w = RL.W + LT::WS * i;
            const double Wrow[2][8] = {{1.0, 0.0, S.a02, S.a03, S.a04, 0.0, S.b00, S.b01}, {0.0, 1.0, S.a12, S.a13, S.a14, 0.0, S.b10, S.b11}};
#pragma unroll
            for (int k = 0; k < 2; k++)
#pragma unroll
                for (int c = 0; c < 8; c++) w[k * 8 + c] = Wrow[k][c];
        }
        if (act) { double *hc = RL.H + LT::HS * i; hc[49] = 0.0; hc[50] = 0.0; }      // (after the look-ahead positions staged here have been read)
        if (lane < 8) {
            const double c2[8] = {0.0, 0.0, 1.0, 0.0, dt, 0.0, 0.0, h2}, c3[8] = {0.0, 0.0, 0.0, 1.0, 0.0, 0.0, dt, 0.0}, c4[8] = {0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, dt};
            double v2 = 0.0, v3 = 0.0, v4 = 0.0;
#pragma unroll
            for (int c = 0; c < 8; c++) if (lane == c) { v2 = c2[c]; v3 = c3[c]; v4 = c4[c]; }
            RL.C[3 * lane] = v2; RL.C[3 * lane + 1] = v3; RL.C[3 * lane + 2] = v4;
        }
    } else
    if (ROWPAR && has_u) {      // W~_t = [A b B] rows 0..4 (cols: x0..x4, b, ua, ual); column 5 is rewritten every iteration
        double *w = RL.W + RowLds::WS * i;
        const double Wrow[5][8] = {{1.0, 0.0, S.a02, S.a03, S.a04, 0.0, S.b00, S.b01}, {0.0, 1.0, S.a12, S.a13, S.a14, 0.0, S.b10, S.b11},
                                   {0.0, 0.0, 1.0, 0.0, dt, 0.0, 0.0, h2}, {0.0, 0.0, 0.0, 1.0, 0.0, 0.0, dt, 0.0}, {0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, dt}};
#pragma unroll
        for (int k = 0; k < 5; k++)
#pragma unroll
            for (int c = 0; c < 8; c++) w[k * 8 + c] = Wrow[k][c];
    }
    // PLDS: the non-trivial entries of A, B are not carried through the interior point either -- rows 0, 1 of the stage's W~ block in LDS hold them
    // (words 2..4, 6, 7 and 10..12, 14, 15) for the whole solve
    auto stage_lin = [&]() {
        if constexpr (SLDS) {
            StageLin L;
            const double *w = RL.W + LT::WS * (has_u ? i : 0);
            L.a02 = has_u ? w[2] : 0.0; L.a03 = has_u ? w[3] : 0.0; L.a04 = has_u ? w[4] : 0.0; L.b00 = has_u ? w[6] : 0.0; L.b01 = has_u ? w[7] : 0.0;
            L.a12 = has_u ? w[10] : 0.0; L.a13 = has_u ? w[11] : 0.0; L.a14 = has_u ? w[12] : 0.0; L.b10 = has_u ? w[14] : 0.0; L.b11 = has_u ? w[15] : 0.0;
            L.dt = dt; L.h2 = h2;
            return L;
        } else return S;
    };
    // ---- inequality rows of this stage, in registers ----
    // box variables k: 0 ua, 1 ual, 2 x, 3 y, 4 v, 5 om  -> z index {0,1,2,3,5,6} (also the slot in Hq below)
    constexpr int NB = 6;
    const int zidx[NB] = {0, 1, 2, 3, 5, 6};
    double ll[NB], tl[NB], lh[NB], th[NB], rtl[NB], rth[NB];
    const bool vbu = has_u, vbx = xb;            // input-box rows (k < 2) / state-box rows (k >= 2) present at this stage
    double lo[NB] = {p.bu_lo[0], p.bu_lo[1], p.bx_lo[0], p.bx_lo[1], p.bx_lo[2], p.bx_lo[3]};
    double hi[NB] = {p.bu_hi[0], p.bu_hi[1], p.bx_hi[0], p.bx_hi[1], p.bx_hi[2], p.bx_hi[3]};
    // Inside the interior point the wave-uniform constants of the problem (12 bounds, 22 cost weights: 68 scalar registers) are RE-READ from the
    // kernel-argument segment at the head of every phase that uses them -- three scalar loads -- instead of staying live across the sweeps: held in
    // scalar registers they do not fit next to the lane masks and LDS addresses (102 registers), and every spilled one comes back through a
    // v_readlane, a VECTOR instruction of the wavefront's one issue stream (150 per iteration before).  The pointer passes through an opaque copy
    // per phase, so that the loads are neither hoisted out of the loop nor merged across phases.
    typedef const __attribute__((address_space(4))) KParams KArg;
    auto reload_bounds = [&]() {
#ifdef MPC_NO_RELOAD      // diagnostic build: the constants stay in (spilled) scalar registers, as in round 2
        return;
#endif
        KArg *pk = (KArg *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(pk));
        lo[0] = pk->bu_lo[0]; lo[1] = pk->bu_lo[1]; hi[0] = pk->bu_hi[0]; hi[1] = pk->bu_hi[1];
#pragma unroll
        for (int k = 0; k < 4; k++) { lo[2 + k] = pk->bx_lo[k]; hi[2 + k] = pk->bx_hi[k]; }
    };
    {
        const double val[NB] = {ui[0], ui[1], xi[0], xi[1], xi[3], xi[4]};
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const double cl = val[k] - lo[k], ch = hi[k] - val[k];
            tl[k] = fmax(cl, p.thr0); th[k] = fmax(ch, p.thr0);
            rtl[k] = rcp_nr(tl[k]); rth[k] = rcp_nr(th[k]);
            ll[k] = p.mu0 * rtl[k]; lh[k] = p.mu0 * rth[k];
            if ((k < 2) ? vbu : vbx) lin0 = fmax(lin0, fmax(tl[k] - cl, th[k] - ch));
        }
    }
    // obstacle rows j: rho1 = h + a'dx + s >= 0 (lam1,t1), rho2 = s >= 0 (lam2,t2); robot_model.py:60-65
    // LEAN (10 obstacles): the row state of ten obstacle pairs does not fit the register file next to the sweeps (784 B of scratch per lane
    // and 47 % of the wave cycles spent waiting for it, profiles/r02_c5_pmc_summary.json), so what is a pure function of the kept state --
    // h, dh/dx, dh/dy of a row and the reciprocals 1/t1, 1/t2 -- is recomputed where it is used instead of being carried: 50 doubles per
    // lane less.  `ObstView` is one row pair's view for one phase; per phase the inputs pass through an opaque zero so that the optimiser
    // cannot merge the recomputations back into long-lived registers.
    constexpr int NKEEP = LEAN ? 1 : NOBST;
    double hh[NKEEP], ax[NKEEP], ay[NKEEP], rt1[NKEEP], rt2[NKEEP];
    double sv[NOBST], l1[NOBST], t1[NOBST], l2[NOBST], t2[NOBST];
    struct ObstView { double hh, ax, ay, rt1, rt2; };
    auto obst_view = [&](int j, double zero) {
        ObstView v;
        if constexpr (LEAN) {
            const double ex = (xi[0] + zero) - pos_x(j), ey = (xi[1] + zero) - pos_y(j);
            v.hh = ex * ex + ey * ey - p.r2; v.ax = 2 * ex; v.ay = 2 * ey;
            v.rt1 = rcp_nr(t1[j] + zero); v.rt2 = rcp_nr(t2[j] + zero);
        } else { v.hh = hh[j]; v.ax = ax[j]; v.ay = ay[j]; v.rt1 = rt1[j]; v.rt2 = rt2[j]; }
        return v;
    };
#pragma unroll
    for (int j = 0; j < NOBST; j++) {
        const double ex = xi[0] - pos_x(j), ey = xi[1] - pos_y(j);
        const double h0 = ex * ex + ey * ey - p.r2;
        if (soft) {
            sv[j] = (h0 < 0 ? -h0 : 0.0) + p.thr0;
            t1[j] = fmax(h0 + sv[j], p.thr0);
            t2[j] = fmax(sv[j], p.thr0);
        } else {
            sv[j] = 0.0; t1[j] = fmax(h0, p.thr0); t2[j] = 1.0;
            if (vs && !ROW_OFF(j)) lin0 = fmax(lin0, t1[j] - h0);
        }
        const double r1 = rcp_nr(t1[j]), r2 = rcp_nr(t2[j]);
        l1[j] = p.mu0 * r1; l2[j] = soft ? p.mu0 * r2 : 0.0;
        if constexpr (!LEAN) { hh[j] = h0; ax[j] = 2 * ex; ay[j] = 2 * ey; rt1[j] = r1; rt2[j] = r2; }
    }
    int n_items_lane = 0;
#pragma unroll
    for (int k = 0; k < NB; k++) n_items_lane += ((k < 2) ? vbu : vbx) ? 2 : 0;
    n_items_lane += vs ? (soft ? 2 * nact : nact) : 0;
    const double n_items = seg_sum<G>((double)n_items_lane, lane);
    double inv_items = n_items > 0 ? 1.0 / n_items : 0.0;
    if constexpr (G == 64) inv_items = wave_uniform(inv_items);
    {   // Non-finite inputs (a diverged plant, a bad sensor frame, a poisoned warm start) must not pass as a converged solve: fmax() drops NaN, so
        // the residual norm above would not show them.  One sum over everything this lane read decides; the instance then fails at once (status 4).
        double fin = gl[0] + gl[1] + ui[0] + ui[1];
#pragma unroll
        for (int c = 0; c < 5; c++) fin += x0v[c] + xi[c] + xnext[c];
        if (act) {
#pragma unroll
            for (int j = 0; j < NOBST; j++) fin += pos_x(j) + pos_y(j);
        }
        if (!(fabs(fin) <= 1e300)) lin0 = INFINITY;
    }
    lin0 = seg_max<G>(lin0, lane);
    if constexpr (G == 64) lin0 = wave_uniform(lin0);

    double z[7] = {0, 0, 0, 0, 0, 0, 0};
    double rhoPi = 1.0;
    // BRANCH-FREE ROW PHASES.  Which rows exist differs per lane (no input rows in the terminal lane, no state box at stage 0, no obstacle rows at
    // stage 0 or where the penalty is zero), but a wave instruction costs the same for 1 or 64 active lanes and every `if (row exists)` around a
    // row's arithmetic is an EXEC save / restore plus a branch that also ends the scheduling region (a lone wavefront issues a dependent FP64
    // instruction every 8.4 cycles, an independent one every 5: the rows of a stage are independent of each other only inside ONE region).
    // So every lane computes all rows of its stage; rows that do not exist carry finite dummy state that nothing reads: their contributions to
    // whatever is shared (sums, maxima, Hessian / gradient entries) enter through the 0 / 1 factors below, as the multiplier of the fused
    // multiply-add that would have been an add.
    const double m_u = vbu ? 1.0 : 0.0, m_x = vbx ? 1.0 : 0.0, m_s = vs ? 1.0 : 0.0;
#define MPC_MK(k) ((k) < 2 ? m_u : m_x)
    int it = 0;
    IpmState ipm;             // per instance: instances sharing a wavefront stop at their own iteration and then idle
    ipm.running = !ep_done;
    int &status = ipm.status, &it_done = ipm.it_done;
    bool &running = ipm.running;
    float stepl = 0.0f;       // this lane's (stage's) last step norm, for the polish (ipm_polish_step)
    if (!(lin0 <= 1e300)) { status = 4; running = false; }

#ifdef MPC_PHASE_TIMING
    long long tacc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    MPC_T0();
    // Everything cheap is RECOMPUTED where it is needed instead of being kept in registers across the stage recursions: the
    // sweeps need ~100 doubles of their own and VALU operands can address only 256 VGPRs.  OPAQUE() hides a value's
    // history from the optimiser (no instruction is emitted), so that recomputations are neither hoisted nor merged back.
#define OPAQUE(x) asm volatile("" : "+v"(x))
    // ---- THE PARTS OF AN ITERATION THAT DO NOT DEPEND ON THE ROW-STORAGE POLICY, ONE DEFINITION (round 5) ----
    // The branch-free and the branched form below differ in how the ROWS of a stage are held and visited; what they shared as duplicated text -- the rows' residual
    // and weight formulas, the cost part of the predictor's right-hand side, the factor sweep with its staging, the three vector sweeps -- is defined here once
    // (lambdas, each called from exactly one place per instantiated kernel: inlined, the same instructions as the text they replace).
    // residual r_d = rho(z) - t of the box rows of variable k, from the iterate (never stored)
    auto box_rd = [&](int k, const double vals[NB], const double zz[7], double &rdl, double &rdh) {
        const double zk = zz[zidx[k]];
        rdl = ((vals[k] - lo[k]) + zk) - tl[k];
        rdh = ((hi[k] - vals[k]) - zk) - th[k];
    };
    // weights / residuals of obstacle row pair j at the iterate zz
    struct SoftT { double w1, w2, rD, be1, be2, rs, rd1, rd2; };
    auto soft_terms = [&](int j, const ObstView &v, const double zz[7]) {
        SoftT o;
        const double y = v.ax * zz[2] + v.ay * zz[3];
        o.w1 = l1[j] * v.rt1;
        if (soft) {
            o.rd1 = (v.hh + y + sv[j]) - t1[j]; o.rd2 = sv[j] - t2[j];
            o.be1 = (l1[j] * t1[j] + l1[j] * o.rd1) * v.rt1;
            o.w2 = l2[j] * v.rt2;
            o.be2 = (l2[j] * t2[j] + l2[j] * o.rd2) * v.rt2;
            o.rs = zpen * sv[j] + zpen - l1[j] - l2[j];
            o.rD = rcp_nr(zpen + o.w1 + o.w2);
        } else {
            o.rd1 = (v.hh + y) - t1[j]; o.rd2 = 0.0;
            o.be1 = (l1[j] * t1[j] + l1[j] * o.rd1) * v.rt1;
            o.w2 = 0.0; o.be2 = 0.0; o.rs = 0.0; o.rD = 0.0;
        }
        return o;
    };
    // an opaque 0.0 per phase (one v_mov; see obst_view)
    auto phase_zero = [&]() { double zz_ = 0.0; if constexpr (LEAN) asm volatile("" : "+v"(zz_)); return zz_; };
    // LEAN: the reciprocals of the box rows are phase-local as well (recomputed at the head of every phase that uses them)
    auto refresh_box_rcp = [&]() {
        if constexpr (LEAN) {
            const double pz = phase_zero();
#pragma unroll
            for (int k = 0; k < NB; k++) { rtl[k] = rcp_nr(tl[k] + pz); rth[k] = rcp_nr(th[k] + pz); }
        }
    };

    // predictor, cost part: bounds and iterate values of this phase, Gauss-Newton gradient (H z + q) and the diagonal of the reduced Hessian before the rows enter
    // (rows: the caller goes on to the box rows -- their reciprocals and bounds are made current here; the polish's stationarity residual needs neither)
    auto predictor_weights = [&](double (&vals)[NB], double (&Hq)[8], double (&gloc)[7], double (&cb)[7], auto rows) {
    if constexpr (decltype(rows)::value) {
    refresh_box_rcp();
    reload_bounds();
    }
    vals[0] = ui[0]; vals[1] = ui[1]; vals[2] = xi[0]; vals[3] = xi[1]; vals[4] = xi[3]; vals[5] = xi[4];
#pragma unroll
    for (int k = 0; k < NB; k++) OPAQUE(vals[k]);
    // Gauss-Newton gradient q and diagonal Hessian, z order (ua, ual, x, y, psi, v, om); robot_ocp_problem.py:59-83
    double Hd[7];
    {   // stage / terminal weights chosen per lane as a select of VALUES: the wave-uniform kernel arguments pass through an opaque
        // scalar copy first -- written as if / else on the argument arrays, the optimiser selects the ADDRESS and issues five
        // per-lane global loads from the kernel-argument segment inside the iteration loop
        double hs[7], ht[5], wg[6], we[4];
#ifdef MPC_NO_RELOAD
#pragma unroll
        for (int c = 0; c < 7; c++) { hs[c] = p.Hd_stage[c]; asm volatile("" : "+s"(hs[c])); }
#pragma unroll
        for (int c = 0; c < 5; c++) { ht[c] = p.Hd_term[c]; asm volatile("" : "+s"(ht[c])); }
#pragma unroll
        for (int c = 0; c < 6; c++) { wg[c] = p.Wg[c]; asm volatile("" : "+s"(wg[c])); }
#pragma unroll
        for (int c = 0; c < 4; c++) { we[c] = p.Weg[c]; asm volatile("" : "+s"(we[c])); }
#else
        KArg *pk = (KArg *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(pk));
#pragma unroll
        for (int c = 0; c < 7; c++) hs[c] = pk->Hd_stage[c];
#pragma unroll
        for (int c = 0; c < 5; c++) ht[c] = pk->Hd_term[c];
#pragma unroll
        for (int c = 0; c < 6; c++) wg[c] = pk->Wg[c];
#pragma unroll
        for (int c = 0; c < 4; c++) we[c] = pk->Weg[c];
#endif
        Hd[0] = has_u ? hs[0] : 0.0; Hd[1] = has_u ? hs[1] : 0.0;
#pragma unroll
        for (int c = 0; c < 5; c++) Hd[2 + c] = has_u ? hs[2 + c] : ht[c];
        gloc[0] = (has_u ? wg[4] : 0.0) * vals[0]; gloc[1] = (has_u ? wg[5] : 0.0) * vals[1];
        gloc[2] = (has_u ? wg[0] : we[0]) * (vals[2] - gl[0]); gloc[3] = (has_u ? wg[1] : we[1]) * (vals[3] - gl[1]); gloc[4] = 0.0;
        gloc[5] = (has_u ? wg[2] : we[2]) * vals[4]; gloc[6] = (has_u ? wg[3] : we[3]) * vals[5];
    }
    Hq[0] = has_u ? Hd[0] : 1.0; Hq[1] = has_u ? Hd[1] : 1.0; Hq[2] = Hd[2]; Hq[3] = Hd[3]; Hq[4] = Hd[4]; Hq[5] = Hd[5]; Hq[6] = Hd[6]; Hq[7] = 0.0;   // diagonal in z order, then Qxy
#pragma unroll
    for (int c = 0; c < 7; c++) { gloc[c] += Hd[c] * z[c]; cb[c] = 0.0; }                              // (H z + q - C'lam), sum_c c beta_c
    };
    // predictor, sweep part: H~aug_t and the affine column to LDS, Riccati factor sweep, this lane's gains back (F)
    auto factor_sweep = [&](const double (&Hq)[8], const double (&gloc)[7], const double (&cb)[7], const double (&bbr)[5], StageFac &F) {
    MPC_TICK(1);
    double gxs[5];
#pragma unroll
    for (int c = 0; c < 5; c++) gxs[c] = gloc[2 + c] + cb[2 + c];
    if (USE_MFMA) {
        if (act) {      // H~aug_t in accumulator layout: z~ order (x0..x4, 1, ua, ual)
            double *hc = ML.HC + 64 * i;
            const double lu0 = gloc[0] + cb[0], lu1 = gloc[1] + cb[1];
            hc[MfmaLds::at(0, 0)] = Hq[2]; hc[MfmaLds::at(1, 1)] = Hq[3]; hc[MfmaLds::at(2, 2)] = Hq[4];
            hc[MfmaLds::at(3, 3)] = Hq[5]; hc[MfmaLds::at(4, 4)] = Hq[6];
            hc[MfmaLds::at(0, 1)] = Hq[7]; hc[MfmaLds::at(1, 0)] = Hq[7];
            hc[MfmaLds::at(6, 6)] = Hq[0]; hc[MfmaLds::at(7, 7)] = Hq[1];
#pragma unroll
            for (int c = 0; c < 5; c++) { hc[MfmaLds::at(c, 5)] = gxs[c]; hc[MfmaLds::at(5, c)] = gxs[c]; }
            hc[MfmaLds::at(6, 5)] = lu0; hc[MfmaLds::at(5, 6)] = lu0;
            hc[MfmaLds::at(7, 5)] = lu1; hc[MfmaLds::at(5, 7)] = lu1;
        }
        wave_sync();
        mfma_factor(lane, N, ML, rhoPi);
        wave_sync();
        F.i00 = 1.0; F.l = 0.0; F.i11 = 1.0; F.k0 = 0.0; F.k1 = 0.0;
#pragma unroll
        for (int c = 0; c < 5; c++) { F.K0[c] = 0.0; F.K1[c] = 0.0; }
        if (has_u) {
            const double *ko = ML.KO + 16 * i;
#pragma unroll
            for (int c = 0; c < 5; c++) { F.K0[c] = ko[c]; F.K1[c] = ko[6 + c]; }
            F.k0 = ko[5]; F.k1 = ko[11]; F.i00 = ko[12]; F.l = ko[13]; F.i11 = ko[14];
        }
    } else if (ROWPAR) {
        if (act) {      // H~aug_t, dense 8 x 8, z~ order (x0..x4, 1, ua, ual); affine column of W~_t
            const double lu0 = gloc[0] + cb[0], lu1 = gloc[1] + cb[1];
            const double Hrow[8][8] = {{Hq[2], Hq[7], 0.0, 0.0, 0.0, gxs[0], 0.0, 0.0}, {Hq[7], Hq[3], 0.0, 0.0, 0.0, gxs[1], 0.0, 0.0},
                                       {0.0, 0.0, Hq[4], 0.0, 0.0, gxs[2], 0.0, 0.0}, {0.0, 0.0, 0.0, Hq[5], 0.0, gxs[3], 0.0, 0.0},
                                       {0.0, 0.0, 0.0, 0.0, Hq[6], gxs[4], 0.0, 0.0}, {gxs[0], gxs[1], gxs[2], gxs[3], gxs[4], 0.0, lu0, lu1},
                                       {0.0, 0.0, 0.0, 0.0, 0.0, lu0, Hq[0], 0.0}, {0.0, 0.0, 0.0, 0.0, 0.0, lu1, 0.0, Hq[1]}};
            double *hc = RL.H + LT::HS * i;
            if constexpr (COMPACT) {    // rows 0..5, then H66 and H77 (rows 6, 7 are synthesised by the sweep, RowLdsC)
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = 0; c < 8; c++) hc[r * 8 + c] = Hrow[r][c];
                hc[48] = Hq[0]; hc[51] = Hq[1];
                if (has_u) { double *w = RL.W + LT::WS * i; w[5] = bbr[0]; w[13] = bbr[1]; w[16] = bbr[2]; w[17] = bbr[3]; w[18] = bbr[4]; }
            } else {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int c = 0; c < 8; c++) hc[r * 8 + c] = Hrow[r][c];
            if (has_u) {
#pragma unroll
                for (int k = 0; k < 5; k++) RL.W[RowLds::WS * i + k * 8 + 5] = bbr[k];
            }
            }
        }
        wave_sync();
        MPC_TICK(9);
#ifdef MPC_FACTOR_PLAIN
        rowpar_factor(lane, N, RS, sweep_worker);
#else
        if constexpr (COMPACT) {
#ifdef MPC_COMPACT_PLAIN
            rowpar_factor(lane, N, RS, sweep_worker);
#else
            rowpar_factor_fast_c(lane, N, RS, sweep_worker);
#endif
        } else {
#ifdef MPC_MFMA4
            if constexpr (G == 64) mfma4_factor(lane, N, RS); else
#endif
            rowpar_factor_fast(lane, N, RS, sweep_worker);
        }
#endif
        wave_sync();
        F.i00 = 1.0; F.l = 0.0; F.i11 = 1.0; F.k0 = 0.0; F.k1 = 0.0;
#pragma unroll
        for (int c = 0; c < 5; c++) { F.K0[c] = 0.0; F.K1[c] = 0.0; }
        if (has_u) {
            const double *ko = RL.H + LT::HS * i;
#pragma unroll
            for (int c = 0; c < 5; c++) { F.K0[c] = ko[c]; F.K1[c] = ko[8 + c]; }
            F.k0 = ko[5]; F.k1 = ko[13]; F.i00 = ko[6]; F.l = ko[7]; F.i11 = ko[14];
        }
        // (no barrier: every lane overwrites only the block it has just read, and a wavefront's LDS operations complete in order)
        if (has_u) {    // closed-loop matrix Acl = A + B K of this stage, row-major, for the row-parallel vector recursions
            double *acl = RL.H + LT::HS * i + RowVec::ACL;
            const StageLin SL = stage_lin();
            const double Ar[2][5] = {{1.0, 0.0, SL.a02, SL.a03, SL.a04}, {0.0, 1.0, SL.a12, SL.a13, SL.a14}};
            const double Br[2][2] = {{SL.b00, SL.b01}, {SL.b10, SL.b11}};
#pragma unroll
            for (int c = 0; c < 5; c++) {
                acl[0 * RowVec::RS + c] = Ar[0][c] + Br[0][0] * F.K0[c] + Br[0][1] * F.K1[c];
                acl[1 * RowVec::RS + c] = Ar[1][c] + Br[1][0] * F.K0[c] + Br[1][1] * F.K1[c];
                acl[2 * RowVec::RS + c] = (c == 2 ? 1.0 : (c == 4 ? dt : 0.0)) + h2 * F.K1[c];
                acl[3 * RowVec::RS + c] = (c == 3 ? 1.0 : 0.0) + dt * F.K0[c];
                acl[4 * RowVec::RS + c] = (c == 4 ? 1.0 : 0.0) + dt * F.K1[c];
            }
        }
    } else
        systolic_factor(i, N, S, Hq, gloc[0] + cb[0], gloc[1] + cb[1], gxs, bbr, rhoPi != 0.0, F);
    };
    // forward vector sweep of the affine step: za
    auto affine_rollout = [&](const double (&bbr)[5], const double (&x_init)[5], StageFac &F, double (&za)[7]) {
    MPC_TICK(2);
    if (ROWPAR) {
        if (has_u) {        // c_t = r_b + B k
            double *cc = RL.H + LT::HS * i + RowVec::ACL + 5;      // c[r] closes row r
            const StageLin SL = stage_lin();
            cc[0 * RowVec::RS] = bbr[0] + SL.b00 * F.k0 + SL.b01 * F.k1; cc[1 * RowVec::RS] = bbr[1] + SL.b10 * F.k0 + SL.b11 * F.k1;
            cc[2 * RowVec::RS] = bbr[2] + h2 * F.k1; cc[3 * RowVec::RS] = bbr[3] + dt * F.k0; cc[4 * RowVec::RS] = bbr[4] + dt * F.k1;
        }
        if (i == 0) {
#pragma unroll
            for (int c = 0; c < 5; c++) RL.H[RowVec::X + c] = SLDS ? rhoPi * lds_raw[RowLdsC::CT + slot * 5 + c] : x_init[c];
        }
        wave_sync();
        rowpar_vector_fast<true>(lane, N, RS, sweep_worker);
        wave_sync();
        if (act) {
            const double *xx = RL.H + LT::HS * i + RowVec::X;
            double u0 = F.k0, u1 = F.k1;
#pragma unroll
            for (int c = 0; c < 5; c++) { za[2 + c] = xx[c]; u0 += F.K0[c] * xx[c]; u1 += F.K1[c] * xx[c]; }
            za[0] = u0; za[1] = u1;
        }
    } else
        systolic_rollout<true>(i, N, S, F, x_init, bbr, za);
    MPC_TICK(3);
    };
    // corrector: adjoint sweep of the right-hand side change gc, feed-forward, forward sweep, dz = corrector + affine step
    auto corrector_sweeps = [&](const double (&gc)[7], const double (&bbr)[5], const double (&x_init)[5], const double (&za)[7], StageFac &F, double (&dz)[7]) {
    MPC_TICK(5);
    if (ROWPAR) {
        if (act) {      // c~_t = gc_x + K' gc_u  (K = 0 in the terminal lane)
            double *cc = RL.H + LT::HS * i + RowVec::CT;
#pragma unroll
            for (int c = 0; c < 5; c++) cc[c] = gc[2 + c] + F.K0[c] * gc[0] + F.K1[c] * gc[1];
        }
        wave_sync();
        rowpar_vector_fast<false>(lane, N, RS, sweep_worker);
        wave_sync();
        if (has_u) {    // feed-forward of the corrector right-hand side: k = -Muu^-1 (gc_u + B' p_{t+1})
            const double *pp = RL.H + LT::HS * (i + 1) + RowVec::P;
            const double pv[5] = {pp[0], pp[1], pp[2], pp[3], pp[4]};
            const StageLin SL = stage_lin();
            const double m0 = gc[0] + SL.dua(pv), m1 = gc[1] + SL.dual(pv);
            F.k1 = fma(F.l, m0, -m1) * F.i11;
            F.k0 = fma(-F.l, F.k1, -(m0 * F.i00));
        }
    } else
        systolic_corrector(i, N, S, gc, F);
    MPC_TICK(6);
    if (ROWPAR) {
        if (has_u) {        // homogeneous dynamics: c_t = B k
            double *cc = RL.H + LT::HS * i + RowVec::ACL + 5;
            const StageLin SL = stage_lin();
            cc[0 * RowVec::RS] = SL.b00 * F.k0 + SL.b01 * F.k1; cc[1 * RowVec::RS] = SL.b10 * F.k0 + SL.b11 * F.k1;
            cc[2 * RowVec::RS] = h2 * F.k1; cc[3 * RowVec::RS] = dt * F.k0; cc[4 * RowVec::RS] = dt * F.k1;
        }
        if (i == 0) {
#pragma unroll
            for (int c = 0; c < 5; c++) RL.H[RowVec::X + c] = 0.0;
        }
        wave_sync();
        rowpar_vector_fast<true>(lane, N, RS, sweep_worker);
        wave_sync();
        if (act) {
            const double *xx = RL.H + LT::HS * i + RowVec::X;
            double u0 = F.k0, u1 = F.k1;
#pragma unroll
            for (int c = 0; c < 5; c++) { dz[2 + c] = xx[c]; u0 += F.K0[c] * xx[c]; u1 += F.K1[c] * xx[c]; }
            dz[0] = u0; dz[1] = u1;
        }
    } else
        systolic_rollout<false>(i, N, S, F, x_init, bbr, dz);
#pragma unroll
    for (int c = 0; c < 7; c++) dz[c] += za[c];
    MPC_TICK(7);
    };
    // polish indicator (c): the stationarity residual at the iterate -- g = H z + q - C' lam of this lane's stage (the cost part as the predictor forms it, the
    // rows' multipliers entered through the 0 / 1 factors in both forms of the row phases), the open-loop adjoint sweep for the input blocks, the slack equations
    // Z s + z - lam_1 - lam_2 of this lane's soft rows; max-norm over the instance.  Runs where ipm_head asks for it (IpmState::ask_g): about once per solve.
    auto stationarity = [&]() {
        double vals[NB], Hq[8], g[7], cb[7];
        predictor_weights(vals, Hq, g, cb, std::false_type{});
#pragma unroll
        for (int k = 0; k < NB; k++) g[zidx[k]] = fma(MPC_MK(k), lh[k] - ll[k], g[zidx[k]]);
        double gx = 0.0, gy = 0.0, rsm = 0.0;
        const double pz = phase_zero();
#pragma unroll
        for (int j = 0; j < NOBST; j++) {
            if (ROW_OFF(j)) continue;
            const ObstView v = obst_view(j, pz);
            gx += l1[j] * v.ax; gy += l1[j] * v.ay;
            if (soft) rsm = fmax(rsm, fabs(zpen * sv[j] + zpen - l1[j] - l2[j]));
        }
        g[2] = fma(-m_s, gx, g[2]); g[3] = fma(-m_s, gy, g[3]);
        const double ru = adjoint_inputs<G>(act, has_u, stage_lin(), g, lane);
        return seg_max<G>(fmax(ru, m_s * rsm), lane);
    };
    // Two forms of the iteration's row phases (same arithmetic, same interior point):
    //   BRANCHFREE (3 and 5 obstacle row pairs): every row of the stage computed by every lane, existence as 0 / 1 factors (see m_u, m_x, m_s);
    //   the branched form (10 row pairs): the rows under `if (row exists)`.  With ten obstacles the register file is full to the last
    //   register (256 + 240 and no scratch); the branch-free form needs ~30 registers more at its peak and pays for them in scratch memory
    //   (60..104 B per lane in every arrangement tried: measured 17.7 against 17.2 ms per C5 control step), so that kernel keeps the branches.
    constexpr bool BRANCHFREE = NOBST < 10;
    if constexpr (BRANCHFREE) {
    ipm.long_step = true;      // (not carried in this form: decided at the head, where the residual is asked for)
    for (it = 0;; it++) {
        // ---- complementarity measures ----
        double msum = 0.0, cmax = 0.0;
        {   // per row group (input box, state box, obstacle rows): sum and maximum over the group's rows, entered at once if the group exists
            double sg = 0.0, cg = 0.0;
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const double a = ll[k] * tl[k], b = lh[k] * th[k];
                sg += a + b;
                cg = fmax(cg, (tl[k] <= 2 * p.tl_min || ll[k] <= 2 * p.tl_min) ? 0.0 : a);
                cg = fmax(cg, (th[k] <= 2 * p.tl_min || lh[k] <= 2 * p.tl_min) ? 0.0 : b);
                if (k == 1 || k == NB - 1) { msum = fma(MPC_MK(k), sg, msum); cmax = fmax(cmax, MPC_MK(k) * cg); sg = 0.0; cg = 0.0; }
            }
#pragma unroll
            for (int j = 0; j < NOBST; j++) {
                if (ROW_OFF(j)) continue;
                const double a = l1[j] * t1[j];
                sg += a;
                cg = fmax(cg, (t1[j] <= 2 * p.tl_min || l1[j] <= 2 * p.tl_min) ? 0.0 : a);
                if (soft) {
                    const double b = l2[j] * t2[j];
                    sg += b;
                    cg = fmax(cg, (t2[j] <= 2 * p.tl_min || l2[j] <= 2 * p.tl_min) ? 0.0 : b);
                }
            }
            msum = fma(m_s, sg, msum); cmax = fmax(cmax, m_s * cg);
        }
        seg_reduce2<G, true>(msum, cmax, lane);
        const double mu = msum * inv_items;
        const double lin = rhoPi * lin0;
        ipm_head(p, ipm, it, mu, lin, cmax);
        // (this form decides the trigger of indicator (c) here, in cold code laid out behind the loop, instead of carrying it -- measured per kernel, one box, per control
        // step against round 5's library: rti_solve_kernel<3, 21, 3> -0.5 % this way, +0.6 % carried; the ten-obstacle form and rti_split_kernel lose 1.5 % this way
        // and carry it: profiles/r06_stationarity_placement.txt)
        if (__builtin_expect(__ballot(ipm.ask_g) != 0ull, 0)) {
            const bool asked = ipm.ask_g && seg_any<G>(stepl > kStationarityStep, lane);
            const double res_g = __ballot(asked) != 0ull ? stationarity() : 0.0;
            ipm_head_g(p, ipm, it, asked ? res_g : 0.0);
        }
        if (__ballot(running) == 0ull) break;
        ipm.cprev = cmax;
        MPC_TICK(0);

        // ---- predictor (sigma = 0): local gradient, barrier terms, reduced Hessian ----
        StageFac F;
        double za[7] = {0, 0, 0, 0, 0, 0, 0};
        double bbr[5], x_init[5];
#pragma unroll
        for (int c = 0; c < 5; c++) { bbr[c] = rhoPi * bb[c]; x_init[c] = SLDS ? 0.0 : rhoPi * d0[c]; }
        {
            double vals[NB], Hq[8], gloc[7], cb[7];
            predictor_weights(vals, Hq, gloc, cb, std::true_type{});
#pragma unroll
            for (int k = 0; k < NB; k++) {
                double rdl, rdh;
                box_rd(k, vals, z, rdl, rdh);
                const double wl = ll[k] * rtl[k], wh = lh[k] * rth[k];
                const double bl = (ll[k] * tl[k] + ll[k] * rdl) * rtl[k], bh = (lh[k] * th[k] + lh[k] * rdh) * rth[k];
                Hq[zidx[k]] = fma(MPC_MK(k), wl + wh, Hq[zidx[k]]);
                gloc[zidx[k]] = fma(MPC_MK(k), lh[k] - ll[k], gloc[zidx[k]]);
                cb[zidx[k]] = fma(MPC_MK(k), bl - bh, cb[zidx[k]]);
            }
            {
                const double pz = phase_zero();
                double hxx = 0.0, hyy = 0.0, hxy = 0.0, gx = 0.0, gy = 0.0, cx = 0.0, cy = 0.0;      // the obstacle rows' sums, entered below if the rows exist
#pragma unroll
                for (int j = 0; j < NOBST; j++) {
                    if (ROW_OFF(j)) continue;
                    const ObstView v = obst_view(j, pz);
                    const SoftT o = soft_terms(j, v, z);
                    double weff, geff;
                    if (soft) {
                        weff = o.w1 * (zpen + o.w2) * o.rD;
                        geff = (o.be1 * (zpen + o.w2) - o.w1 * (o.rs + o.be2)) * o.rD;
                    } else { weff = o.w1; geff = o.be1; }
                    hxx += weff * v.ax * v.ax; hyy += weff * v.ay * v.ay; hxy += weff * v.ax * v.ay;
                    gx += l1[j] * v.ax; gy += l1[j] * v.ay;
                    cx += geff * v.ax; cy += geff * v.ay;
                }
                Hq[2] = fma(m_s, hxx, Hq[2]); Hq[3] = fma(m_s, hyy, Hq[3]); Hq[7] = fma(m_s, hxy, Hq[7]);
                gloc[2] = fma(-m_s, gx, gloc[2]); gloc[3] = fma(-m_s, gy, gloc[3]);
                cb[2] = fma(m_s, cx, cb[2]); cb[3] = fma(m_s, cy, cb[3]);
            }
            factor_sweep(Hq, gloc, cb, bbr, F);
        }
        affine_rollout(bbr, x_init, F, za);

        // ---- affine step: dt, dlam per row, step ratios, products dlam_aff * dt_aff ----
        // The products dlam_aff * dt_aff of the rows are not carried from here to the combined step (18 doubles with 3 obstacles, 32 with 10: they would
        // cross two sweeps in accumulation registers -- six register moves per double -- and with ten obstacles they are what no longer fits): the combined
        // step recomputes them from the affine step za, which it keeps, and the corrector's right-hand side, linear in sigma * mu, is accumulated HERE
        // as  gc = G1 - sigma mu G0  (so the corrector has no pass over the rows at all).
        double G1[7] = {0, 0, 0, 0, 0, 0, 0}, G0[7] = {0, 0, 0, 0, 0, 0, 0};
        double smu;
        {
#pragma unroll
            for (int c = 0; c < 7; c++) OPAQUE(z[c]);
#pragma unroll
            for (int j = 0; j < NOBST; j++) { OPAQUE(l1[j]); OPAQUE(l2[j]); }
            refresh_box_rcp();
            reload_bounds();
            double vals[NB] = {ui[0], ui[1], xi[0], xi[1], xi[3], xi[4]};
#pragma unroll
            for (int k = 0; k < NB; k++) OPAQUE(vals[k]);
            double rmax = 0.0, rmaxd = 0.0;      // largest -dt/t (primal) and -dlam/lam (dual) ratios
            double rg = 0.0, rgd = 0.0;      // ... of one row group (see m_u, m_x, m_s), entered when the group is done
            // complementarity after the affine step, sum over the rows of (lam + ad dlam)(t + a dt) = S0 + a S1 + ad S2 + a ad S3 with
            // S0 = sum lam t, S1 = sum lam dt, S2 = sum dlam t, S3 = sum dlam dt per row group: four running sums instead of the rows' steps
            // kept alive across the ratio reduction (24 + 4 NOBST doubles: the register peak of the whole iteration)
            double S0 = 0.0, S1 = 0.0, S2 = 0.0, S3 = 0.0, T0 = 0.0, T1 = 0.0, T2 = 0.0, T3 = 0.0;      // S: the group in progress, T: this lane's totals
#pragma unroll
            for (int k = 0; k < NB; k++) {
                double rdl, rdh;
                box_rd(k, vals, z, rdl, rdh);
                const double dzk = za[zidx[k]];
                const double dtl = dzk + rdl, dth = -dzk + rdh;
                const double ltl = ll[k] * tl[k], lth = lh[k] * th[k];
                const double dll = -(ltl + ll[k] * dtl) * rtl[k], dlh = -(lth + lh[k] * dth) * rth[k];
                const double ppl = dll * dtl, pph = dlh * dth;
                G1[zidx[k]] = MPC_MK(k) * (ppl * rtl[k] - pph * rth[k]); G0[zidx[k]] = MPC_MK(k) * (rtl[k] - rth[k]);
                rg = fmax(rg, fmax(-dtl * rtl[k], -dth * rth[k]));
                rgd = fmax(rgd, fmax(fma(dtl, rtl[k], 1.0), fma(dth, rth[k], 1.0)));      // -dlam/lam = 1 + dt/t when sigma = 0
                S0 += ltl + lth; S1 = fma(ll[k], dtl, fma(lh[k], dth, S1)); S2 = fma(dll, tl[k], fma(dlh, th[k], S2)); S3 += ppl + pph;
                if (k == 1 || k == NB - 1) {
                    const double m = MPC_MK(k);
                    rmax = fmax(rmax, m * rg); rmaxd = fmax(rmaxd, m * rgd);
                    T0 = fma(m, S0, T0); T1 = fma(m, S1, T1); T2 = fma(m, S2, T2); T3 = fma(m, S3, T3);
                    rg = rgd = 0.0; S0 = S1 = S2 = S3 = 0.0;
                }
            }
            double cx1 = 0.0, cx0 = 0.0, cy1 = 0.0, cy0 = 0.0;      // the obstacle rows' part of G1, G0 (x and y entries)
            const double pz = phase_zero();
#pragma unroll
            for (int j = 0; j < NOBST; j++) {
                if (!ROW_OFF(j)) {
                    const ObstView v = obst_view(j, pz);
                    const SoftT o = soft_terms(j, v, z);
                    const double y = v.ax * za[2] + v.ay * za[3];
                    const double lt1 = l1[j] * t1[j];
                    double e1, e0, dt1, dl1;       // d beta_c eliminated to the x, y entries: geff = e1 - sigma mu e0
                    if (soft) {
                        const double rsum = o.rs + o.be1 + o.be2;
                        const double ds = -(rsum + o.w1 * y) * o.rD;
                        dt1 = o.rd1 + (y * (zpen + o.w2) - rsum) * o.rD;     // y + ds without cancellation
                        const double dt2 = o.rd2 + ds;
                        const double lt2 = l2[j] * t2[j];
                        const double dl2 = -(lt2 + l2[j] * dt2) * v.rt2;
                        dl1 = -(lt1 + l1[j] * dt1) * v.rt1;
                        const double pp2 = dl2 * dt2, pp1 = dl1 * dt1;
                        const double q = (zpen + o.w2) * v.rt1, r = o.w1 * v.rt2;
                        e1 = (pp1 * q - pp2 * r) * o.rD; e0 = (q - r) * o.rD;
                        rg = fmax(rg, -dt2 * v.rt2); rgd = fmax(rgd, fma(dt2, v.rt2, 1.0));
                        S0 += lt2; S1 = fma(l2[j], dt2, S1); S2 = fma(dl2, t2[j], S2); S3 += pp2 + pp1;
                    } else {
                        dt1 = o.rd1 + y;
                        dl1 = -(lt1 + l1[j] * dt1) * v.rt1;
                        e1 = dl1 * dt1 * v.rt1; e0 = v.rt1;
                        S3 = fma(dl1, dt1, S3);
                    }
                    S0 += lt1; S1 = fma(l1[j], dt1, S1); S2 = fma(dl1, t1[j], S2);
                    cx1 = fma(e1, v.ax, cx1); cx0 = fma(e0, v.ax, cx0); cy1 = fma(e1, v.ay, cy1); cy0 = fma(e0, v.ay, cy0);
                    rg = fmax(rg, -dt1 * v.rt1); rgd = fmax(rgd, fma(dt1, v.rt1, 1.0));
                }
            }
            G1[2] = fma(m_s, cx1, G1[2]); G0[2] = fma(m_s, cx0, G0[2]); G1[3] = fma(m_s, cy1, G1[3]); G0[3] = fma(m_s, cy0, G0[3]);
            rmax = fmax(rmax, m_s * rg); rmaxd = fmax(rmaxd, m_s * rgd);
            T0 = fma(m_s, S0, T0); T1 = fma(m_s, S1, T1); T2 = fma(m_s, S2, T2); T3 = fma(m_s, S3, T3);      // this lane's four sums over the rows that exist
            seg_reduce2<G, false>(rmax, rmaxd, lane);
            double a_aff, a_affd;
            ipm_affine_steps(rmax, rmaxd, a_aff, a_affd);
            double maff = fma(a_aff, fma(a_affd, T3, T1), fma(a_affd, T2, T0));
            maff = seg_sum<G>(maff, lane) * inv_items;
            double sigma;
            smu = ipm_centring(maff, mu, cmax, sigma);
#ifndef MPC_PHASE_TIMING
            if (p.trace && i == 0 && valid && running) {
                double *tr = p.trace + ((size_t)inst * p.iter_max + it) * 4;
                tr[0] = mu; tr[1] = sigma; tr[3] = cmax;
            }
#endif
        }
        MPC_TICK(4);

        // ---- corrector: homogeneous system for the change of right-hand side, d beta_c = (dlam_aff dt_aff - sigma mu) / t ----
        double dz[7] = {0, 0, 0, 0, 0, 0, 0};
        {
            double gc[7];
#pragma unroll
            for (int c = 0; c < 7; c++) gc[c] = fma(-smu, G0[c], G1[c]);
            corrector_sweeps(gc, bbr, x_init, za, F, dz);
        }

        // ---- combined step: ratios, step length, update (instances that have stopped keep their state) ----
        {
#pragma unroll
            for (int c = 0; c < 7; c++) { OPAQUE(z[c]); OPAQUE(za[c]); }      // (za: or the optimiser keeps the affine phase's products alive instead of recomputing them)
#pragma unroll
            for (int j = 0; j < NOBST; j++) { OPAQUE(l1[j]); OPAQUE(l2[j]); }
            refresh_box_rcp();
            reload_bounds();
            double vals[NB] = {ui[0], ui[1], xi[0], xi[1], xi[3], xi[4]};
#pragma unroll
            for (int k = 0; k < NB; k++) OPAQUE(vals[k]);
            double rmax = 0.0, rmaxd = 0.0;
            double rg = 0.0, rgd = 0.0;
            double dtl_[NB], dth_[NB], dll_[NB], dlh_[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                double rdl, rdh;
                box_rd(k, vals, z, rdl, rdh);
                const double dzk = dz[zidx[k]];
                dtl_[k] = dzk + rdl; dth_[k] = -dzk + rdh;
                // (dlam_aff dt_aff of the row, exactly as the affine phase formed it: same state, same za)
                const double dal = za[zidx[k]] + rdl, dah = -za[zidx[k]] + rdh;
                const double ppl = -(ll[k] * tl[k] + ll[k] * dal) * rtl[k] * dal, pph = -(lh[k] * th[k] + lh[k] * dah) * rth[k] * dah;
                dll_[k] = -(ll[k] * tl[k] - smu + ppl + ll[k] * dtl_[k]) * rtl[k];
                dlh_[k] = -(lh[k] * th[k] - smu + pph + lh[k] * dth_[k]) * rth[k];
                rg = fmax(rg, fmax(-dtl_[k] * rtl[k], -dth_[k] * rth[k]));
                rgd = fmax(rgd, fmax(-dll_[k] * rcp_nr(ll[k]), -dlh_[k] * rcp_nr(lh[k])));
                if (k == 1 || k == NB - 1) { rmax = fmax(rmax, MPC_MK(k) * rg); rmaxd = fmax(rmaxd, MPC_MK(k) * rgd); rg = rgd = 0.0; }
            }
            // the combined step of obstacle row pair j
            struct ObstStep { double dt1, dl1, dt2, dl2, ds; };
            auto obst_step = [&](int j, double pzero) {
                ObstStep q; q.dt2 = q.dl2 = q.ds = 0.0;
                const ObstView v = obst_view(j, pzero);
                const SoftT o = soft_terms(j, v, z);
                const double y = v.ax * dz[2] + v.ay * dz[3];
                const double ya = v.ax * za[2] + v.ay * za[3];
                double pp1;
                if (soft) {
                    const double rsa = o.rs + o.be1 + o.be2;
                    const double da1 = o.rd1 + (ya * (zpen + o.w2) - rsa) * o.rD, da2 = o.rd2 - (rsa + o.w1 * ya) * o.rD;
                    const double pp2 = -(l2[j] * t2[j] + l2[j] * da2) * v.rt2 * da2;
                    pp1 = -(l1[j] * t1[j] + l1[j] * da1) * v.rt1 * da1;
                    const double db1 = (pp1 - smu) * v.rt1, db2 = (pp2 - smu) * v.rt2;
                    const double rsum = o.rs + (o.be1 + db1) + (o.be2 + db2);
                    q.ds = -(rsum + o.w1 * y) * o.rD;
                    q.dt1 = o.rd1 + (y * (zpen + o.w2) - rsum) * o.rD;
                    q.dt2 = o.rd2 + q.ds;
                    q.dl2 = -(l2[j] * t2[j] - smu + pp2 + l2[j] * q.dt2) * v.rt2;
                    rg = fmax(rg, -q.dt2 * v.rt2); rgd = fmax(rgd, -q.dl2 * rcp_nr(l2[j]));
                } else {
                    const double da1 = o.rd1 + ya;
                    pp1 = -(l1[j] * t1[j] + l1[j] * da1) * v.rt1 * da1;
                    q.dt1 = o.rd1 + y;
                }
                q.dl1 = -(l1[j] * t1[j] - smu + pp1 + l1[j] * q.dt1) * v.rt1;
                rg = fmax(rg, -q.dt1 * v.rt1); rgd = fmax(rgd, -q.dl1 * rcp_nr(l1[j]));
                return q;
            };
            // TWOPASS (ten obstacles): the 50 step values of the obstacle rows are not kept across the ratio reduction -- they are the register peak of
            // the iteration, 100 registers that end up in scratch memory -- but computed again behind it (+6 % instructions, no scratch)
            constexpr bool TWOPASS = false;      // (tried with ten obstacles: +17 % instructions and the same scratch -- the peak is not here)
            double dt1_[TWOPASS ? 1 : NOBST], dl1_[TWOPASS ? 1 : NOBST], dt2_[TWOPASS ? 1 : NOBST], dl2_[TWOPASS ? 1 : NOBST], ds_[TWOPASS ? 1 : NOBST];
            {
                const double pz = phase_zero();
#pragma unroll
                for (int j = 0; j < NOBST; j++) {
                    if (ROW_OFF(j)) continue;
                    const ObstStep q = obst_step(j, pz);
                    if constexpr (!TWOPASS) { dt1_[j] = q.dt1; dl1_[j] = q.dl1; dt2_[j] = q.dt2; dl2_[j] = q.dl2; ds_[j] = q.ds; }
                }
            }
            rmax = fmax(rmax, m_s * rg); rmaxd = fmax(rmaxd, m_s * rgd);
            seg_reduce2<G, false>(rmax, rmaxd, lane);
            double alpha, alphad;
            ipm_step_lengths(rmax, rmaxd, alpha, alphad);
#ifndef MPC_PHASE_TIMING
            if (p.trace && i == 0 && valid && running) p.trace[((size_t)inst * p.iter_max + it) * 4 + 2] = alpha;
#endif
            ipm_step_check(ipm, it, alpha, alphad, smu);
            ipm_polish_step<G, false>(p, ipm, lane, alpha, dz, stepl);
            {   // An instance that has stopped keeps its step z (a select, not a step of length zero: what a converged instance computes while it idles
                // beside a neighbour that still iterates is a Newton step from a state with slacks at their floor -- it may be Inf or NaN, and 0 * NaN
                // is NaN).  Its row state is no longer read by anything and simply moves on, as do the rows that do not exist.
                const double ae = alpha;
                // rows that do not exist take no step (the 0 / 1 factors again): their dummy state stays at its benign initial value, so that everything
                // computed from it is bounded whatever the instance does -- outside the ratio test they would otherwise run into their floors and
                // overflow within a few iterations (lam grows by 1 / floor per step), and 0 * Inf would enter the shared sums as NaN
                const double aeg[3] = {alpha * m_u, alpha * m_x, alpha * m_s}, adg[3] = {alphad * m_u, alphad * m_x, alphad * m_s};
#pragma unroll
                for (int c = 0; c < 7; c++) z[c] = running ? fma(ae, dz[c], z[c]) : z[c];
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    const int g = k < 2 ? 0 : 1;
                    tl[k] = fmax(fma(aeg[g], dtl_[k], tl[k]), p.tl_min); th[k] = fmax(fma(aeg[g], dth_[k], th[k]), p.tl_min);
                    ll[k] = fmax(fma(adg[g], dll_[k], ll[k]), p.tl_min); lh[k] = fmax(fma(adg[g], dlh_[k], lh[k]), p.tl_min);
                    if constexpr (!LEAN) { rtl[k] = rcp_nr(tl[k]); rth[k] = rcp_nr(th[k]); }
                }
                const double pz2 = phase_zero();
#pragma unroll
                for (int j = 0; j < NOBST; j++) {
                    if (ROW_OFF(j)) continue;
                    ObstStep q;
                    if constexpr (TWOPASS) q = obst_step(j, pz2);      // (its ratio outputs are dead here)
                    else { q.dt1 = dt1_[j]; q.dl1 = dl1_[j]; q.dt2 = dt2_[j]; q.dl2 = dl2_[j]; q.ds = ds_[j]; }
                    t1[j] = fmax(fma(aeg[2], q.dt1, t1[j]), p.tl_min); l1[j] = fmax(fma(adg[2], q.dl1, l1[j]), p.tl_min);
                    if constexpr (!LEAN) rt1[j] = rcp_nr(t1[j]);
                    if (soft) {
                        sv[j] = fma(aeg[2], q.ds, sv[j]);
                        t2[j] = fmax(fma(aeg[2], q.dt2, t2[j]), p.tl_min); l2[j] = fmax(fma(adg[2], q.dl2, l2[j]), p.tl_min);
                        if constexpr (!LEAN) rt2[j] = rcp_nr(t2[j]);
                    }
                }
                rhoPi *= (1.0 - ae);
            }
        }
        MPC_TICK(8);
    }
    } else {
    for (it = 0;; it++) {
        // ---- complementarity measures ----
        double msum = 0.0, cmax = 0.0;
#pragma unroll
        for (int k = 0; k < NB; k++) if ((k < 2) ? vbu : vbx) {
            const double a = ll[k] * tl[k], b = lh[k] * th[k];
            msum += a + b;
            if (!(tl[k] <= 2 * p.tl_min || ll[k] <= 2 * p.tl_min)) cmax = fmax(cmax, a);
            if (!(th[k] <= 2 * p.tl_min || lh[k] <= 2 * p.tl_min)) cmax = fmax(cmax, b);
        }
        if (vs) {
#pragma unroll
            for (int j = 0; j < NOBST; j++) {
                if (ROW_OFF(j)) continue;
                const double a = l1[j] * t1[j];
                msum += a;
                if (!(t1[j] <= 2 * p.tl_min || l1[j] <= 2 * p.tl_min)) cmax = fmax(cmax, a);
                if (soft) {
                    const double b = l2[j] * t2[j];
                    msum += b;
                    if (!(t2[j] <= 2 * p.tl_min || l2[j] <= 2 * p.tl_min)) cmax = fmax(cmax, b);
                }
            }
        }
        seg_reduce2<G, true>(msum, cmax, lane);
        const double mu = msum * inv_items;
        const double lin = rhoPi * lin0;
        ipm_head(p, ipm, it, mu, lin, cmax);
        if (__ballot(ipm.ask_g) != 0ull) ipm_head_g(p, ipm, it, stationarity());
        if (__ballot(running) == 0ull) break;
        ipm.cprev = cmax;
        MPC_TICK(0);

        // ---- predictor (sigma = 0): local gradient, barrier terms, reduced Hessian ----
        StageFac F;
        double za[7] = {0, 0, 0, 0, 0, 0, 0};
        double bbr[5], x_init[5];
#pragma unroll
        for (int c = 0; c < 5; c++) { bbr[c] = rhoPi * bb[c]; x_init[c] = SLDS ? 0.0 : rhoPi * d0[c]; }
        {
            double vals[NB], Hq[8], gloc[7], cb[7];
            predictor_weights(vals, Hq, gloc, cb, std::true_type{});
#pragma unroll
            for (int k = 0; k < NB; k++) if ((k < 2) ? vbu : vbx) {
                double rdl, rdh;
                box_rd(k, vals, z, rdl, rdh);
                const double wl = ll[k] * rtl[k], wh = lh[k] * rth[k];
                const double bl = (ll[k] * tl[k] + ll[k] * rdl) * rtl[k], bh = (lh[k] * th[k] + lh[k] * rdh) * rth[k];
                Hq[zidx[k]] += wl + wh;
                gloc[zidx[k]] += lh[k] - ll[k];
                cb[zidx[k]] += bl - bh;
            }
            if (vs) {
                const double pz = phase_zero();
#pragma unroll
                for (int j = 0; j < NOBST; j++) {
                    if (ROW_OFF(j)) continue;
                    const ObstView v = obst_view(j, pz);
                    const SoftT o = soft_terms(j, v, z);
                    double weff, geff;
                    if (soft) {
                        weff = o.w1 * (zpen + o.w2) * o.rD;
                        geff = (o.be1 * (zpen + o.w2) - o.w1 * (o.rs + o.be2)) * o.rD;
                    } else { weff = o.w1; geff = o.be1; }
                    Hq[2] += weff * v.ax * v.ax; Hq[3] += weff * v.ay * v.ay; Hq[7] += weff * v.ax * v.ay;
                    gloc[2] -= l1[j] * v.ax; gloc[3] -= l1[j] * v.ay;
                    cb[2] += geff * v.ax; cb[3] += geff * v.ay;
                }
            }
            factor_sweep(Hq, gloc, cb, bbr, F);
        }
        affine_rollout(bbr, x_init, F, za);

        // ---- affine step: dt, dlam per row, step ratios, products dlam_aff * dt_aff ----
        double ppl[NB], pph[NB], pp1[NOBST], pp2[NOBST];
        double smu;
        {
#pragma unroll
            for (int c = 0; c < 7; c++) OPAQUE(z[c]);
#pragma unroll
            for (int j = 0; j < NOBST; j++) { OPAQUE(l1[j]); OPAQUE(l2[j]); }
            refresh_box_rcp();
            reload_bounds();
            double vals[NB] = {ui[0], ui[1], xi[0], xi[1], xi[3], xi[4]};
#pragma unroll
            for (int k = 0; k < NB; k++) OPAQUE(vals[k]);
            double rmax = 0.0, rmaxd = 0.0;      // largest -dt/t (primal) and -dlam/lam (dual) ratios
            double dtl_[NB], dth_[NB], dll_[NB], dlh_[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                dtl_[k] = dth_[k] = dll_[k] = dlh_[k] = 0.0; ppl[k] = pph[k] = 0.0;
                if ((k < 2) ? vbu : vbx) {
                    double rdl, rdh;
                    box_rd(k, vals, z, rdl, rdh);
                    const double dzk = za[zidx[k]];
                    dtl_[k] = dzk + rdl; dth_[k] = -dzk + rdh;
                    dll_[k] = -(ll[k] * tl[k] + ll[k] * dtl_[k]) * rtl[k]; dlh_[k] = -(lh[k] * th[k] + lh[k] * dth_[k]) * rth[k];
                    ppl[k] = dll_[k] * dtl_[k]; pph[k] = dlh_[k] * dth_[k];
                    rmax = fmax(rmax, fmax(-dtl_[k] * rtl[k], -dth_[k] * rth[k]));
                    rmaxd = fmax(rmaxd, fmax(fma(dtl_[k], rtl[k], 1.0), fma(dth_[k], rth[k], 1.0)));      // -dlam/lam = 1 + dt/t when sigma = 0
                }
            }
            double dt1_[NOBST], dl1_[NOBST], dt2_[NOBST], dl2_[NOBST];
            const double pz = phase_zero();
#pragma unroll
            for (int j = 0; j < NOBST; j++) {
                dt1_[j] = dl1_[j] = dt2_[j] = dl2_[j] = 0.0; pp1[j] = pp2[j] = 0.0;
                if (vs && !ROW_OFF(j)) {
                    const ObstView v = obst_view(j, pz);
                    const SoftT o = soft_terms(j, v, z);
                    const double y = v.ax * za[2] + v.ay * za[3];
                    if (soft) {
                        const double rsum = o.rs + o.be1 + o.be2;
                        const double ds = -(rsum + o.w1 * y) * o.rD;
                        dt1_[j] = o.rd1 + (y * (zpen + o.w2) - rsum) * o.rD;     // y + ds without cancellation
                        dt2_[j] = o.rd2 + ds;
                        dl2_[j] = -(l2[j] * t2[j] + l2[j] * dt2_[j]) * v.rt2;
                        pp2[j] = dl2_[j] * dt2_[j];
                        rmax = fmax(rmax, -dt2_[j] * v.rt2); rmaxd = fmax(rmaxd, fma(dt2_[j], v.rt2, 1.0));
                    } else dt1_[j] = o.rd1 + y;
                    dl1_[j] = -(l1[j] * t1[j] + l1[j] * dt1_[j]) * v.rt1;
                    pp1[j] = dl1_[j] * dt1_[j];
                    rmax = fmax(rmax, -dt1_[j] * v.rt1); rmaxd = fmax(rmaxd, fma(dt1_[j], v.rt1, 1.0));
                }
            }
            seg_reduce2<G, false>(rmax, rmaxd, lane);
            double a_aff, a_affd;
            ipm_affine_steps(rmax, rmaxd, a_aff, a_affd);
            double maff = 0.0;
#pragma unroll
            for (int k = 0; k < NB; k++) if ((k < 2) ? vbu : vbx)
                maff += (ll[k] + a_affd * dll_[k]) * (tl[k] + a_aff * dtl_[k]) + (lh[k] + a_affd * dlh_[k]) * (th[k] + a_aff * dth_[k]);
            if (vs) {
#pragma unroll
                for (int j = 0; j < NOBST; j++) {
                    if (ROW_OFF(j)) continue;
                    maff += (l1[j] + a_affd * dl1_[j]) * (t1[j] + a_aff * dt1_[j]);
                    if (soft) maff += (l2[j] + a_affd * dl2_[j]) * (t2[j] + a_aff * dt2_[j]);
                }
            }
            maff = seg_sum<G>(maff, lane) * inv_items;
            double sigma;
            smu = ipm_centring(maff, mu, cmax, sigma);
#ifndef MPC_PHASE_TIMING
            if (p.trace && i == 0 && valid && running) {
                double *tr = p.trace + ((size_t)inst * p.iter_max + it) * 4;
                tr[0] = mu; tr[1] = sigma; tr[3] = cmax;
            }
#endif
        }
        MPC_TICK(4);

        // ---- corrector: homogeneous system for the change of right-hand side, d beta_c = (dlam_aff dt_aff - sigma mu) / t ----
        double dz[7] = {0, 0, 0, 0, 0, 0, 0};
        {
            refresh_box_rcp();
            double gc[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < NB; k++) if ((k < 2) ? vbu : vbx) {
                const double dbl = (ppl[k] - smu) * rtl[k], dbh = (pph[k] - smu) * rth[k];
                gc[zidx[k]] += dbl - dbh;
            }
            if (vs) {
                const double pz = phase_zero();
#pragma unroll
                for (int j = 0; j < NOBST; j++) {
                    if (ROW_OFF(j)) continue;
                    const ObstView v = obst_view(j, pz);
                    const double db1 = (pp1[j] - smu) * v.rt1;
                    double geff;
                    if (soft) {
                        const double w1 = l1[j] * v.rt1, w2 = l2[j] * v.rt2;
                        const double db2 = (pp2[j] - smu) * v.rt2;
                        geff = (db1 * (zpen + w2) - w1 * db2) * rcp_nr(zpen + w1 + w2);
                    } else geff = db1;
                    gc[2] += geff * v.ax; gc[3] += geff * v.ay;
                }
            }
            corrector_sweeps(gc, bbr, x_init, za, F, dz);
        }

        // ---- combined step: ratios, step length, update (instances that have stopped keep their state) ----
        {
#pragma unroll
            for (int c = 0; c < 7; c++) OPAQUE(z[c]);
#pragma unroll
            for (int j = 0; j < NOBST; j++) { OPAQUE(l1[j]); OPAQUE(l2[j]); }
            refresh_box_rcp();
            reload_bounds();
            double vals[NB] = {ui[0], ui[1], xi[0], xi[1], xi[3], xi[4]};
#pragma unroll
            for (int k = 0; k < NB; k++) OPAQUE(vals[k]);
            double rmax = 0.0, rmaxd = 0.0;
            double dtl_[NB], dth_[NB], dll_[NB], dlh_[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                dtl_[k] = dth_[k] = dll_[k] = dlh_[k] = 0.0;
                if ((k < 2) ? vbu : vbx) {
                    double rdl, rdh;
                    box_rd(k, vals, z, rdl, rdh);
                    const double dzk = dz[zidx[k]];
                    dtl_[k] = dzk + rdl; dth_[k] = -dzk + rdh;
                    dll_[k] = -(ll[k] * tl[k] - smu + ppl[k] + ll[k] * dtl_[k]) * rtl[k];
                    dlh_[k] = -(lh[k] * th[k] - smu + pph[k] + lh[k] * dth_[k]) * rth[k];
                    rmax = fmax(rmax, fmax(-dtl_[k] * rtl[k], -dth_[k] * rth[k]));
                    rmaxd = fmax(rmaxd, fmax(-dll_[k] * rcp_nr(ll[k]), -dlh_[k] * rcp_nr(lh[k])));
                }
            }
            double dt1_[NOBST], dl1_[NOBST], dt2_[NOBST], dl2_[NOBST], ds_[NOBST];
            const double pz = phase_zero();
#pragma unroll
            for (int j = 0; j < NOBST; j++) {
                dt1_[j] = dl1_[j] = dt2_[j] = dl2_[j] = ds_[j] = 0.0;
                if (vs && !ROW_OFF(j)) {
                    const ObstView v = obst_view(j, pz);
                    const SoftT o = soft_terms(j, v, z);
                    const double y = v.ax * dz[2] + v.ay * dz[3];
                    if (soft) {
                        const double db1 = (pp1[j] - smu) * v.rt1, db2 = (pp2[j] - smu) * v.rt2;
                        const double rsum = o.rs + (o.be1 + db1) + (o.be2 + db2);
                        ds_[j] = -(rsum + o.w1 * y) * o.rD;
                        dt1_[j] = o.rd1 + (y * (zpen + o.w2) - rsum) * o.rD;
                        dt2_[j] = o.rd2 + ds_[j];
                        dl2_[j] = -(l2[j] * t2[j] - smu + pp2[j] + l2[j] * dt2_[j]) * v.rt2;
                        rmax = fmax(rmax, -dt2_[j] * v.rt2); rmaxd = fmax(rmaxd, -dl2_[j] * rcp_nr(l2[j]));
                    } else dt1_[j] = o.rd1 + y;
                    dl1_[j] = -(l1[j] * t1[j] - smu + pp1[j] + l1[j] * dt1_[j]) * v.rt1;
                    rmax = fmax(rmax, -dt1_[j] * v.rt1); rmaxd = fmax(rmaxd, -dl1_[j] * rcp_nr(l1[j]));
                }
            }
            seg_reduce2<G, false>(rmax, rmaxd, lane);
            double alpha, alphad;
            ipm_step_lengths(rmax, rmaxd, alpha, alphad);
#ifndef MPC_PHASE_TIMING
            if (p.trace && i == 0 && valid && running) p.trace[((size_t)inst * p.iter_max + it) * 4 + 2] = alpha;
#endif
            ipm_step_check(ipm, it, alpha, alphad, smu);
            ipm_polish_step<G>(p, ipm, lane, alpha, dz, stepl);
            if (running) {
#pragma unroll
                for (int c = 0; c < 7; c++) z[c] += alpha * dz[c];
#pragma unroll
                for (int k = 0; k < NB; k++) if ((k < 2) ? vbu : vbx) {
                    tl[k] = fmax(tl[k] + alpha * dtl_[k], p.tl_min); th[k] = fmax(th[k] + alpha * dth_[k], p.tl_min);
                    ll[k] = fmax(ll[k] + alphad * dll_[k], p.tl_min); lh[k] = fmax(lh[k] + alphad * dlh_[k], p.tl_min);
                    if constexpr (!LEAN) { rtl[k] = rcp_nr(tl[k]); rth[k] = rcp_nr(th[k]); }
                }
                if (vs) {
#pragma unroll
                    for (int j = 0; j < NOBST; j++) {
                        if (ROW_OFF(j)) continue;
                        t1[j] = fmax(t1[j] + alpha * dt1_[j], p.tl_min); l1[j] = fmax(l1[j] + alphad * dl1_[j], p.tl_min);
                        if constexpr (!LEAN) rt1[j] = rcp_nr(t1[j]);
                        if (soft) {
                            sv[j] += alpha * ds_[j];
                            t2[j] = fmax(t2[j] + alpha * dt2_[j], p.tl_min); l2[j] = fmax(l2[j] + alphad * dl2_[j], p.tl_min);
                            if constexpr (!LEAN) rt2[j] = rcp_nr(t2[j]);
                        }
                    }
                }
                rhoPi *= (1.0 - alpha);
            }
        }
        MPC_TICK(8);
    }
    }
#undef OPAQUE
#undef MPC_MK
#ifdef MPC_PHASE_TIMING
    if (p.trace && i == 0 && valid) { for (int k = 0; k < 10; k++) p.trace[((size_t)inst * p.iter_max) * 4 + k] = (double)tacc_[k]; }
#endif

    if constexpr (LEAN || G != 64) {       // the plant state and the goal are read again here instead of being carried through the interior point (7 doubles per
        const double *xg = p.x0 + (size_t)inst * 5;     // lane less; with one instance per wavefront the goal sits in scalar registers anyway)
        asm volatile("" : "+v"(xg));
#pragma unroll
        for (int c = 0; c < 5; c++) x0v[c] = xg[c];
        if constexpr (G != 64) {
            const double *gg = p.goal + (size_t)inst * 2;
            asm volatile("" : "+v"(gg));
            gl[0] = gg[0]; gl[1] = gg[1];
        }
    }
    typedef const __attribute__((address_space(4))) KParams KTail;
    // THE TAIL READS ITS KERNEL ARGUMENTS AGAIN.  Output pointers and fused-step parameters are used only from here on; carried from the prologue they live
    // in (spilled) scalar registers across the whole interior point -- and in one instantiation (rti_solve_kernel<3, 32, 2>) this toolchain's register allocator
    // re-materialised a kernel-argument load over the live half of another one in the prologue, so that `iters_acc` arrived here holding `ep_steps`
    // (tests/test_gpu_every_kernel.py).  Read here through an opaque copy of the segment pointer they are a handful of scalar
    // loads with live ranges of a few instructions, and the prologue no longer holds them (132 -> 75 spilled scalars at <3, 21, 3>, 152 -> 73 at <10, 64, 3>;
    // same speed).  The stage-split kernel keeps its arguments from the prologue: it spills 20 scalars, no instantiation of it shows the defect (audit rule P2
    // refuses a build that does), and there the six kernel-argument cache lines have left the scalar cache by the time the tail runs (-1 % at C2, measured).
    KTail *pt = (KTail *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(pt));
    World wld_t;
    wld_t.xmin = pt->world.xmin; wld_t.xmax = pt->world.xmax; wld_t.ymin = pt->world.ymin; wld_t.ymax = pt->world.ymax; wld_t.bug_compat_predict = pt->world.bug_compat_predict;
    // every field the tail uses, read in ONE block (the loads cluster and are waited for once; read where they are used they would be issued one by one inside the branches)
    int const t_fused = pt->fused;
    double *const t_x0_rw = pt->x0_rw;
    const double *const t_obst = pt->obst;
    double *const t_obst_rw = pt->obst_rw;
    const double *const t_noise = pt->noise;
    double const t_randomness = pt->randomness;
    double const t_vmax = pt->vmax;
    double const t_r_hit = pt->r_hit;
    double const t_tol_goal = pt->tol_goal;
    double const t_r2 = pt->r2;
    double *const t_ep_min_margin = pt->ep_min_margin;
    int32_t *const t_ep_flags = pt->ep_flags;
    int32_t *const t_ep_steps = pt->ep_steps;
    double *const t_u0 = pt->u0;
    double *const t_cost = pt->cost;
    int32_t *const t_status = pt->status;
    int32_t *const t_iters = pt->iters;
    int32_t *const t_iters_acc = pt->iters_acc;
    int32_t *const t_status_acc = pt->status_acc;
    const double t_Wg[6] = {pt->Wg[0], pt->Wg[1], pt->Wg[2], pt->Wg[3], pt->Wg[4], pt->Wg[5]}, t_Weg[4] = {pt->Weg[0], pt->Weg[1], pt->Weg[2], pt->Weg[3]};
    // ---- full step on the iterate; status 4 leaves it unchanged ----
    const bool store = valid && !ep_done;
    status = ipm_finite_step<G>(status, z, lane);
    if (status != 4) {
#pragma unroll
        for (int c = 0; c < 5; c++) xi[c] += z[2 + c];
        ui[0] += z[0]; ui[1] += z[1];
    }
    const double u_apply[2] = {lane_value_seg<G>(ui[0], lane), lane_value_seg<G>(ui[1], lane)};   // u* = U[0] of this instance
    if ((t_fused & kFuseResetOnFail) && status == 4) {      // set_initial_guess(), robot_ocp_problem.py:203-205,286-306
        xi[0] = x0v[0]; xi[1] = x0v[1]; xi[2] = x0v[2]; xi[3] = 0.0; xi[4] = 0.0; ui[0] = ui[1] = 0.0;
        if (t_fused & kFuseInterpGuess) interp_guess(x0v, gl[1], i <= N ? i : N, N, xi);
    }
    if (store && (status != 4 || (t_fused & (kFuseResetOnFail | kFuseShift)))) {
        if (t_fused & kFuseShift) {                          // X[j] <- X[j+1], U[j] <- U[j+1], U[N-1] <- 0, X[N] kept (:253-258)
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
    if (t_fused & (kFusePlant | kFuseObstacles | kFuseMetrics)) {
        double xp[5] = {x0v[0], x0v[1], x0v[2], x0v[3], x0v[4]};
        if ((t_fused & kFuseAliasBug) && (t_fused & kFuseResetOnFail) && status == 4) { xp[3] = 0.0; xp[4] = 0.0; }
        double xnew[5] = {xp[0], xp[1], xp[2], xp[3], xp[4]};
        if (t_fused & kFusePlant) dyn_step<false>(xp, u_apply, dt, xnew, nullptr, nullptr);     // every lane, same value
        if ((t_fused & kFusePlant) && i == 0 && store && t_x0_rw) {
#pragma unroll
            for (int c = 0; c < 5; c++) t_x0_rw[(size_t)inst * 5 + c] = xnew[c];
        }
        double margin = INFINITY;
        if (t_obst && i < nact) {                            // ground-truth motion of obstacle j = i
            const double *o = t_obst + ((size_t)inst * nact + i) * 4;
            double ox = o[0], oy = o[1], ovx = o[2], ovy = o[3];
            if (t_fused & kFuseObstacles) {
                if (t_noise) obstacle_noise(t_randomness, t_vmax, t_noise[((size_t)inst * nact + i) * 2], t_noise[((size_t)inst * nact + i) * 2 + 1], ovx, ovy);
                obstacle_advance(wld_t, dt, ox, ovx, oy, ovy);
                if (store && t_obst_rw) { double *w = t_obst_rw + ((size_t)inst * nact + i) * 4; w[0] = ox; w[1] = oy; w[2] = ovx; w[3] = ovy; }
            }
            const double ddx = xnew[0] - ox, ddy = xnew[1] - oy;
            margin = sqrt(ddx * ddx + ddy * ddy) - t_r_hit;  // :222-228
        }
        if (t_fused & kFuseMetrics) {
            margin = -seg_max<G>(-margin, lane);
            if (i == 0 && store) {
                int fl = t_ep_flags[inst];
                if (xnew[0] < wld_t.xmin || xnew[0] > wld_t.xmax || xnew[1] < wld_t.ymin || xnew[1] > wld_t.ymax) fl |= 2;   // :213-214
                const double mm = fmin(t_ep_min_margin[inst], margin);
                t_ep_min_margin[inst] = mm;
                if (mm <= 0.0) fl |= 4;
                const double gx_ = xnew[0] - gl[0], gy_ = xnew[1] - gl[1];
                if (sqrt(gx_ * gx_ + gy_ * gy_) <= t_tol_goal) fl |= 1;      // :247-250: reached, the loop breaks before i += 1
                else t_ep_steps[inst] += 1;
                t_ep_flags[inst] = fl;
            }
        }
    }
    if (i == 0 && t_u0 && store) { t_u0[(size_t)inst * 2] = u_apply[0]; t_u0[(size_t)inst * 2 + 1] = u_apply[1]; }
    // NLP objective at the returned iterate: LS cost + exact penalty of the obstacle violation
    if (t_cost) {
        double J = 0.0;
        if (act) {
            const double ex = xi[0] - gl[0], ey = xi[1] - gl[1];
            if (has_u) J = 0.5 * (t_Wg[0] * ex * ex + t_Wg[1] * ey * ey + t_Wg[2] * xi[3] * xi[3] + t_Wg[3] * xi[4] * xi[4]
                                  + t_Wg[4] * ui[0] * ui[0] + t_Wg[5] * ui[1] * ui[1]);
            else J = 0.5 * (t_Weg[0] * ex * ex + t_Weg[1] * ey * ey + t_Weg[2] * xi[3] * xi[3] + t_Weg[3] * xi[4] * xi[4]);
#pragma unroll
            for (int j = 0; j < NOBST; j++) {
                if (ROW_OFF(j)) continue;
                const double dx = xi[0] - pos_x(j), dy = xi[1] - pos_y(j);
                const double hv = dx * dx + dy * dy - t_r2;
                const double v = hv < 0 ? -hv : 0.0;
                J += zpen * (v + 0.5 * v * v);
            }
        }
        J = seg_sum<G>(J, lane);
        if (i == 0 && store) t_cost[inst] = J;
    }
    if (i == 0 && store) {
        if (t_iters_acc) t_iters_acc[inst] += it_done;
        if (t_status_acc) t_status_acc[inst] += (status == 4 ? 1 : 0) + (status == 2 ? 65536 : 0);
        if (t_status) t_status[inst] = status;
        if (t_iters) t_iters[inst] = it_done;
    }
}
#undef ROW_OFF
#undef OBST_IN

}  // namespace mpc
