/*
 * mpc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See mpc_oracle.h.
 *
 * Pinned per seed by the closed-loop tables the reference recorded (acados/HPIPM themselves are absent; see the header).
 *
 * Deliberately written as a dense, generic, residual-form ("delta form") primal-dual interior
 * point method with explicit costates and explicit KKT residuals, i.e. NOT the way the HIP
 * kernel is organised (which uses the structure-exploiting "absolute form" without costates).
 * Agreement of the two is therefore a real check.  The QP of one RTI step is strictly convex
 * (Gauss-Newton Hessian + LM*I, robot_ocp_problem.py:127-128) so its solution is unique.
 *
 * All paths cited are relative to /root/reference.
 */
#include "mpc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NX ORC_NX
#define NU ORC_NU
#define NZ ORC_NZ
/* floor of t and lam.  1e-13 until round 3; 1e-11 since round 4: a pair that collapses to the floor enters the reduced Hessian with the weight
 * lam / t, and at 1e-13 those weights (1e15 and more) cost the Newton step of the end-game its last digits -- measured against the EXACT solution of
 * the QP (tests/helpers.py::exact_qp) on first solves of C5's problem: worst 7.3e-4 -> 7.5e-6, 2.7 % -> 1.0 % of the instances beyond 1e-6, same
 * iteration counts (scripts/tail_scan_cpu.py).  Shared with the HIP kernels (rti_kernel.hpp kTLMin, KParams::tl_min).  It must stay
 * below qp_tol -- an active row's residual rho - t is the floor itself -- so the floor in effect is min(1e-11, qp_tol / 10). */
#define TL_MIN_MAX 1e-11
static _Thread_local double TL_MIN = TL_MIN_MAX;      /* the floor in effect: min(1e-11, qp_tol / 10), set per solve (ipm_solve) */

/* Per-thread workspace of the solve path (round 4): the ~35 arrays of one solve come from a thread-local arena that is reset, not freed, when the
 * outermost solve returns -- the batch driver's OpenMP threads used to malloc / free each of them per solve, and the all-core cpu_baseline of
 * bench.py measured glibc's allocator (15x on 128 cores).  Outside a solve (depth 0: orc_export_qp, orc_linearize, ...) WS_ALLOC is malloc.
 * The arithmetic is untouched: results are bit-identical to the malloc build (tests/test_oracle_golden.py). */
#define WS_CHUNKS 32
typedef struct { char *chunk[WS_CHUNKS]; size_t cap[WS_CHUNKS]; size_t used; int cur, n, depth; } ws_t;
static _Thread_local ws_t g_ws;
static void ws_enter(void) { g_ws.depth++; }
static void ws_leave(void) { if (--g_ws.depth == 0) { g_ws.cur = 0; g_ws.used = 0; } }
static void *ws_alloc(size_t n)
{
    if (g_ws.depth == 0) return malloc(n);
    n = (n + 63) & ~(size_t)63;
    while (g_ws.cur < g_ws.n && g_ws.used + n > g_ws.cap[g_ws.cur]) { g_ws.cur++; g_ws.used = 0; }
    if (g_ws.cur == g_ws.n) {
        if (g_ws.n == WS_CHUNKS) abort();
        size_t cap = n > ((size_t)1 << 20) ? n : ((size_t)1 << 20);
        g_ws.chunk[g_ws.n] = aligned_alloc(64, cap); g_ws.cap[g_ws.n] = cap; g_ws.n++; g_ws.used = 0;
        if (!g_ws.chunk[g_ws.cur]) abort();
    }
    void *r = g_ws.chunk[g_ws.cur] + g_ws.used; g_ws.used += n;
    return r;
}
static void ws_free(void *q) { if (g_ws.depth == 0) free(q); }
#define WS_ALLOC(n) ws_alloc(n)
#define WS_FREE(q) ws_free(q)

static const int IDXBX[4] = {0, 1, 3, 4}; /* robot_ocp_problem.py:93 */

/* Gauss-Legendre 4-point rule on [0,1] (acados IRK default: GL, 4 stages) */
static const double GLC[4] = {0.069431844202973712388, 0.330009478207571867599,
                              0.669990521792428132401, 0.930568155797026287612};
static const double GLB[4] = {0.173927422568726928687, 0.326072577431273071313,
                              0.326072577431273071313, 0.173927422568726928687};

void orc_default_config(orc_config *c, int N, int n_obst, double Tf)
{
    memset(c, 0, sizeof(*c));
    c->N = N; c->n_obst = n_obst; c->Tf = Tf;
    /* robot_ocp_problem.py:24-27: R = 0.15 I2, Q = 2 I4, Q_e = 5 I4 */
    for (int k = 0; k < 4; k++) { c->W[k] = 2.0; c->We[k] = 5.0; }
    c->W[4] = c->W[5] = 0.15;
    c->lm = 2.0;                                      /* :128 */
    c->bx_lo[0] = c->bx_lo[1] = -7.0; c->bx_hi[0] = c->bx_hi[1] = 7.0;    /* :91-92 */
    c->bx_lo[2] = c->bx_lo[3] = -10.0; c->bx_hi[2] = c->bx_hi[3] = 10.0;  /* V_MAX_ROBOT */
    c->bu_lo[0] = c->bu_lo[1] = -8.0; c->bu_hi[0] = c->bu_hi[1] = 8.0;    /* C_MAX :95-96 */
    c->r_safe = 1.0 + 0.2 + 1.2;                      /* robot_model.py:62 */
    c->slack_a = 1e4; c->slack_b = 50.0;              /* :146 */
    c->qp_iter_max = 50;                              /* world_specification.py:48 */
    c->qp_tol = 1e-10;
    c->cost_scale_dt = 1; c->slack_scale_dt = 1; c->lm_scaled = 1;   /* lm_scaled: see DESIGN.md section 2 (statistical pin) */ c->bx_terminal = 0; c->soft_h = 1;
    c->arena[0] = -8.0; c->arena[1] = 8.0; c->arena[2] = -8.0; c->arena[3] = 8.0;
    c->bug_compat_predict = 1;
    c->mu0 = 1e4;
    /* thr0: 0.1; from 8 obstacles on 0.3 (round 4, scripts/thr0_probe.py: -4 % iterations and +5 % solves/s on C5's problem, +2.6 % at N = 20 with 10 obstacles).  0.3 would
     * also gain 3 % at C2 and 1 % at C3 and reproduce 420 instead of 416 recorded rows, but one of the 41 seeds on which the recorded tables prove that acados converged
     * within 25 iterations then needs more than 25 here (profiles/r04_thr0_probe.txt): the reference's own problem size keeps the constant its pin was made with. */
    c->thr0 = n_obst >= 8 ? 0.3 : 0.1;
    c->qp_fail_policy = 0;
    c->polish_ratio = 1e-2; c->polish_tol = 1e-6;
    /* polish_step_frac (kappa): from N = 30 on the last Newton step of a solve is trusted to 1 % only.  Measured (scripts/fuzz_parity.py, fuzz_closed_loop.py, round 5): at
     * N = 30 .. 62 five solves of 1.2e7 ended 1e-5 .. 2e-5 from the exact solution on either side although the observed contraction (4e-3) promised 3e-8 -- the remainder of
     * their 2e-4 long last step was 1.5 % .. 10 % of it (barrier weights lam / t_floor through a long Riccati recursion); with kappa = 0.01 every one of them ends <= 3e-7.
     * Price: +0.4 .. 0.8 % iterations at N >= 30 with 5 .. 10 obstacles -- and +3.8 % at N = 20, where no solve of 1e7 needed it: hence the horizon in the default. */
    c->polish_step_frac = N >= 30 ? 0.01 : 0.0;
    c->polish_res_g = 1e-7;      /* indicator (c), round 6 (ipm_solve says what it is and what it was measured to buy) */
}

/* ------------------------------------------------------------------------------------------ */
/* Model: robot_model.py:39-43                                                                 */
/* ------------------------------------------------------------------------------------------ */
void orc_ode(const double *x, const double *u, double *xdot)
{
    xdot[0] = x[3] * cos(x[2]);
    xdot[1] = x[3] * sin(x[2]);
    xdot[2] = x[4];
    xdot[3] = u[0];
    xdot[4] = u[1];
}

static void ode_jac(const double *x, double *fx /*5x5*/, double *fu /*5x2*/)
{
    memset(fx, 0, 25 * sizeof(double));
    memset(fu, 0, 10 * sizeof(double));
    double c = cos(x[2]), s = sin(x[2]);
    fx[0 * 5 + 2] = -x[3] * s; fx[0 * 5 + 3] = c;
    fx[1 * 5 + 2] = x[3] * c;  fx[1 * 5 + 3] = s;
    fx[2 * 5 + 4] = 1.0;
    fu[3 * 2 + 0] = 1.0;
    fu[4 * 2 + 1] = 1.0;
}

/*
 * Closed form of one IRK-GL4 step for this ODE: psi, v, omega are polynomial
 * in t, and x,y are a 4-point Gauss-Legendre quadrature of v(t)cos(psi(t)), v(t)sin(psi(t)).
 * Plant integrator (robot_ocp_problem.py:136,207-212) and OCP integrator (:129) are the same map.
 */
void orc_dynamics(const double *x, const double *u, double dt, double *xn, double *A, double *B)
{
    double psi = x[2], v = x[3], om = x[4], a = u[0], al = u[1];
    double sx = 0, sy = 0;
    double dx_dpsi = 0, dx_dv = 0, dx_dom = 0, dx_da = 0, dx_dal = 0;
    double dy_dpsi = 0, dy_dv = 0, dy_dom = 0, dy_da = 0, dy_dal = 0;
    for (int j = 0; j < 4; j++) {
        double tau = GLC[j] * dt, w = GLB[j] * dt;
        double vj = v + a * tau;
        double pj = psi + om * tau + 0.5 * al * tau * tau;
        double cj = cos(pj), sj = sin(pj);
        sx += w * vj * cj;
        sy += w * vj * sj;
        dx_dpsi += -w * vj * sj;          dy_dpsi += w * vj * cj;
        dx_dv += w * cj;                  dy_dv += w * sj;
        dx_dom += -w * vj * sj * tau;     dy_dom += w * vj * cj * tau;
        dx_da += w * tau * cj;            dy_da += w * tau * sj;
        dx_dal += -w * vj * sj * 0.5 * tau * tau;
        dy_dal += w * vj * cj * 0.5 * tau * tau;
    }
    xn[0] = x[0] + sx;
    xn[1] = x[1] + sy;
    xn[2] = psi + om * dt + 0.5 * al * dt * dt;
    xn[3] = v + a * dt;
    xn[4] = om + al * dt;
    if (A) {
        memset(A, 0, 25 * sizeof(double));
        for (int k = 0; k < 5; k++) A[k * 5 + k] = 1.0;
        A[0 * 5 + 2] = dx_dpsi; A[0 * 5 + 3] = dx_dv; A[0 * 5 + 4] = dx_dom;
        A[1 * 5 + 2] = dy_dpsi; A[1 * 5 + 3] = dy_dv; A[1 * 5 + 4] = dy_dom;
        A[2 * 5 + 4] = dt;
    }
    if (B) {
        memset(B, 0, 10 * sizeof(double));
        B[0 * 2 + 0] = dx_da; B[0 * 2 + 1] = dx_dal;
        B[1 * 2 + 0] = dy_da; B[1 * 2 + 1] = dy_dal;
        B[2 * 2 + 1] = 0.5 * dt * dt;
        B[3 * 2 + 0] = dt;
        B[4 * 2 + 1] = dt;
    }
}

/* dense Gaussian elimination with partial pivoting: solves M X = R in place (M n x n, R n x m) */
static int gauss_solve(int n, int m, double *M, double *R)
{
    for (int k = 0; k < n; k++) {
        int p = k; double best = fabs(M[k * n + k]);
        for (int r = k + 1; r < n; r++) if (fabs(M[r * n + k]) > best) { best = fabs(M[r * n + k]); p = r; }
        if (best == 0.0) return -1;
        if (p != k) {
            for (int cidx = 0; cidx < n; cidx++) { double t = M[k * n + cidx]; M[k * n + cidx] = M[p * n + cidx]; M[p * n + cidx] = t; }
            for (int cidx = 0; cidx < m; cidx++) { double t = R[k * m + cidx]; R[k * m + cidx] = R[p * m + cidx]; R[p * m + cidx] = t; }
        }
        double inv = 1.0 / M[k * n + k];
        for (int r = k + 1; r < n; r++) {
            double f = M[r * n + k] * inv;
            if (f == 0.0) continue;
            for (int cidx = k; cidx < n; cidx++) M[r * n + cidx] -= f * M[k * n + cidx];
            for (int cidx = 0; cidx < m; cidx++) R[r * m + cidx] -= f * R[k * m + cidx];
        }
    }
    for (int k = n - 1; k >= 0; k--) {
        double inv = 1.0 / M[k * n + k];
        for (int cidx = 0; cidx < m; cidx++) {
            double s = R[k * m + cidx];
            for (int j = k + 1; j < n; j++) s -= M[k * n + j] * R[j * m + cidx];
            R[k * m + cidx] = s * inv;
        }
    }
    return 0;
}

/* Butcher matrix of the s=4 Gauss-Legendre collocation method: a_ij = int_0^{c_i} l_j(t) dt */
static void gl4_butcher(double a[4][4])
{
    for (int j = 0; j < 4; j++) {
        /* l_j(t) = prod_{m != j} (t - c_m)/(c_j - c_m): cubic with coefficients p[0..3] */
        double p[4] = {1, 0, 0, 0}; int deg = 0; double den = 1.0;
        for (int m = 0; m < 4; m++) {
            if (m == j) continue;
            double np_[4] = {0, 0, 0, 0};
            for (int d = 0; d <= deg; d++) { np_[d + 1] += p[d]; np_[d] += -GLC[m] * p[d]; }
            deg++;
            for (int d = 0; d < 4; d++) p[d] = np_[d];
            den *= (GLC[j] - GLC[m]);
        }
        for (int i = 0; i < 4; i++) {
            double t = GLC[i], acc = 0, tp = t;
            for (int d = 0; d < 4; d++) { acc += p[d] * tp / (d + 1); tp *= t; }
            a[i][j] = acc / den;
        }
    }
}

/*
 * General IRK Gauss-Legendre collocation (4 stages, 1 step) with `newton_iter` Newton iterations
 * from K = 0 and forward sensitivities by the implicit function theorem -- what acados'
 * integrator_type='IRK' (robot_ocp_problem.py:129) does with its defaults [acados-knowledge].
 * Used only to check the closed form above.
 */
void orc_dynamics_collocation(const double *x, const double *u, double dt, int newton_iter,
                              double *xn, double *A, double *B)
{
    double a[4][4]; gl4_butcher(a);
    double K[20]; memset(K, 0, sizeof(K));
    double J[400], R[20 * 8], fx[25], fu[10];
    for (int it = 0; it <= newton_iter; it++) {
        /* residual G = K - f(x + dt a K, u) and Jacobian dG/dK */
        memset(J, 0, sizeof(J));
        for (int s = 0; s < 4; s++) {
            double xs[5], f[5];
            for (int k = 0; k < 5; k++) { xs[k] = x[k]; for (int l = 0; l < 4; l++) xs[k] += dt * a[s][l] * K[l * 5 + k]; }
            orc_ode(xs, u, f); ode_jac(xs, fx, fu);
            for (int k = 0; k < 5; k++) {
                R[(s * 5 + k) * 8 + 0] = -(K[s * 5 + k] - f[k]);
                for (int m = 0; m < 5; m++) R[(s * 5 + k) * 8 + 1 + m] = fx[k * 5 + m];  /* rhs for dK/dx */
                for (int m = 0; m < 2; m++) R[(s * 5 + k) * 8 + 6 + m] = fu[k * 2 + m];  /* rhs for dK/du */
                J[(s * 5 + k) * 20 + (s * 5 + k)] += 1.0;
                for (int l = 0; l < 4; l++) for (int m = 0; m < 5; m++)
                    J[(s * 5 + k) * 20 + (l * 5 + m)] -= dt * a[s][l] * fx[k * 5 + m];
            }
        }
        gauss_solve(20, 8, J, R);
        if (it < newton_iter) { for (int i = 0; i < 20; i++) K[i] += R[i * 8 + 0]; }
        /* after the last Newton update, one more pass (it == newton_iter) evaluates sensitivities at the final K */
    }
    for (int k = 0; k < 5; k++) { xn[k] = x[k]; for (int s = 0; s < 4; s++) xn[k] += dt * GLB[s] * K[s * 5 + k]; }
    if (A) for (int k = 0; k < 5; k++) for (int m = 0; m < 5; m++) {
        double acc = (k == m) ? 1.0 : 0.0;
        for (int s = 0; s < 4; s++) acc += dt * GLB[s] * R[(s * 5 + k) * 8 + 1 + m];
        A[k * 5 + m] = acc;
    }
    if (B) for (int k = 0; k < 5; k++) for (int m = 0; m < 2; m++) {
        double acc = 0.0;
        for (int s = 0; s < 4; s++) acc += dt * GLB[s] * R[(s * 5 + k) * 8 + 6 + m];
        B[k * 2 + m] = acc;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Obstacles: utils/visualization.py:25-79                                                     */
/* ------------------------------------------------------------------------------------------ */
static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* visualization.py:25-60 (predict_step).  state = {x, y, vx, vy}. */
void orc_obstacle_step(const orc_config *c, double *st, double dt, const double *noise, double randomness, double vmax)
{
    double x = st[0], y = st[1], vx = st[2], vy = st[3];
    if (noise) { /* :28-33 */
        vx = clampd((1.0 + randomness * noise[0]) * vx, -vmax, vmax);
        vy = clampd((1.0 + randomness * noise[1]) * vy, -vmax, vmax);
    }
    double t_hit;
    if (vx < 0) t_hit = (x - c->arena[0]) / fabs(vx);        /* :35-40 */
    else if (vx > 0) t_hit = (c->arena[1] - x) / fabs(vx);
    else t_hit = INFINITY;
    if (t_hit <= dt) { x += (vx * t_hit - vx * (dt - t_hit)); vx = -vx; } else x += vx * dt;  /* :42-46 */
    if (vy < 0) t_hit = (y - c->arena[2]) / fabs(vy);        /* :48-53 */
    else if (vy > 0) t_hit = (c->arena[3] - y) / fabs(vy);
    else t_hit = INFINITY;
    if (t_hit <= dt) { y += (vy * t_hit - vy * (dt - t_hit)); vy = -vy; } else y += vy * dt;  /* :55-59 */
    st[0] = x; st[1] = y; st[2] = vx; st[3] = vy;
}

/* visualization.py:62-79: note :69 `vx = self.vy` (reference defect D1), reproduced when bug_compat_predict */
void orc_predict_trajectory(const orc_config *c, const double *state, int n, double dt, double *traj)
{
    double st[4] = {state[0], state[1], c->bug_compat_predict ? state[3] : state[2], state[3]};
    traj[0] = st[0]; traj[1] = st[1];
    for (int i = 0; i < n; i++) {
        orc_obstacle_step(c, st, dt, NULL, 0.0, 0.0);
        traj[2 * (i + 1)] = st[0]; traj[2 * (i + 1) + 1] = st[1];
    }
}

/* robot_ocp_problem.py:154-166: P[i] = [o0x,o0y,o1x,o1y,...] */
void orc_predict_params(const orc_config *c, const double *obst, double *P)
{
    int N = c->N, no = c->n_obst; double dt = c->Tf / N;
    double *traj = (double *)malloc(sizeof(double) * 2 * (N + 1));
    for (int j = 0; j < no; j++) {
        orc_predict_trajectory(c, obst + 4 * j, N, dt, traj);
        for (int i = 0; i <= N; i++) { P[(i * no + j) * 2] = traj[2 * i]; P[(i * no + j) * 2 + 1] = traj[2 * i + 1]; }
    }
    free(traj);
}

/* robot_ocp_problem.py:145-152 */
void orc_slack_alpha(const orc_config *c, const double *x0, const double *goal, double *alpha)
{
    double d0 = x0[0] - goal[0], d1 = x0[1] - goal[1];
    double scale = c->slack_a * (d0 * d0 + d1 * d1 + x0[3] * x0[3] + x0[4] * x0[4] + c->slack_b);
    for (int i = 0; i <= c->N; i++) alpha[i] = scale * (double)(c->N - i) / (double)c->N;
}

/* robot_ocp_problem.py:286-306: X[i] = [x0_x, x0_y, x0_psi, 0, 0], U = 0 (the aliasing of :301-302 is the caller's) */
void orc_initial_guess(const orc_config *c, const double *x0, double *X, double *U)
{
    for (int i = 0; i <= c->N; i++) { X[i * 5] = x0[0]; X[i * 5 + 1] = x0[1]; X[i * 5 + 2] = x0[2]; X[i * 5 + 3] = 0; X[i * 5 + 4] = 0; }
    for (int i = 0; i < c->N * 2; i++) U[i] = 0;
}

/* robot_ocp_problem.py:293-300, the commented "straight line guess" -- the code behind the two interpolate_init tables of
 * src/simulation/test_data: x_guess = x0_x + i/N (x0_x - x0_x), y_guess = x0_y + i/N (goal_y - x0_y), psi_guess = arctan2(goal_y - x0_y,
 * goal_x - goal_x), v = omega = 0, u = 0 (the two self-differences are the reference's) */
void orc_initial_guess_interp(const orc_config *c, const double *x0, const double *goal, double *X, double *U)
{
    const double dy = goal[1] - x0[1];
    const double psi = atan2(dy, goal[0] - goal[0]);
    for (int i = 0; i <= c->N; i++) {
        const double f = (double)i / (double)c->N;
        X[i * 5] = x0[0] + f * (x0[0] - x0[0]); X[i * 5 + 1] = x0[1] + f * dy; X[i * 5 + 2] = psi; X[i * 5 + 3] = 0; X[i * 5 + 4] = 0;
    }
    for (int i = 0; i < c->N * 2; i++) U[i] = 0;
}

/* robot_ocp_problem.py:253-258 */
void orc_shift(const orc_config *c, double *X, double *U)
{
    int N = c->N;
    for (int j = 0; j < N - 1; j++) {
        for (int k = 0; k < 5; k++) X[j * 5 + k] = X[(j + 1) * 5 + k];
        for (int k = 0; k < 2; k++) U[j * 2 + k] = U[(j + 1) * 2 + k];
    }
    for (int k = 0; k < 5; k++) X[(N - 1) * 5 + k] = X[N * 5 + k];
    U[(N - 1) * 2] = 0; U[(N - 1) * 2 + 1] = 0;
}

/* ------------------------------------------------------------------------------------------ */
/* OCP pieces                                                                                   */
/* ------------------------------------------------------------------------------------------ */
/* robot_model.py:60-65 */
static void obstacle_h(const orc_config *c, const double *x, const double *p, double *h, double *dh)
{
    for (int j = 0; j < c->n_obst; j++) {
        double ex = x[0] - p[2 * j], ey = x[1] - p[2 * j + 1];
        h[j] = ex * ex + ey * ey - c->r_safe * c->r_safe;
        if (dh) { dh[2 * j] = 2 * ex; dh[2 * j + 1] = 2 * ey; }
    }
}

static double stage_cs(const orc_config *c) { return c->cost_scale_dt ? c->Tf / c->N : 1.0; }
static double stage_ss(const orc_config *c) { return c->slack_scale_dt ? c->Tf / c->N : 1.0; }
static double stage_lm(const orc_config *c) { return c->lm_scaled ? c->lm * c->Tf / c->N : c->lm; }

/*
 * Gauss-Newton QP blocks: y = [x,y,v,w,ua,ual], W diag -> H diagonal.
 * z order is [u(2); x(5)].  Hd[7] diagonal of H, q[7] gradient.   robot_ocp_problem.py:59-83
 */
static void stage_cost_blocks(const orc_config *c, int i, const double *x, const double *u, const double *goal,
                              double *Hd, double *q)
{
    if (i < c->N) {
        double cs = stage_cs(c), lm = stage_lm(c);
        Hd[0] = cs * c->W[4] + lm; Hd[1] = cs * c->W[5] + lm;
        Hd[2] = cs * c->W[0] + lm; Hd[3] = cs * c->W[1] + lm; Hd[4] = lm; Hd[5] = cs * c->W[2] + lm; Hd[6] = cs * c->W[3] + lm;
        q[0] = cs * c->W[4] * u[0]; q[1] = cs * c->W[5] * u[1];
        q[2] = cs * c->W[0] * (x[0] - goal[0]); q[3] = cs * c->W[1] * (x[1] - goal[1]); q[4] = 0.0;
        q[5] = cs * c->W[2] * x[3]; q[6] = cs * c->W[3] * x[4];
    } else {
        double lm = c->lm; /* terminal: unscaled */
        Hd[0] = Hd[1] = 0; q[0] = q[1] = 0;
        Hd[2] = c->We[0] + lm; Hd[3] = c->We[1] + lm; Hd[4] = lm; Hd[5] = c->We[2] + lm; Hd[6] = c->We[3] + lm;
        q[2] = c->We[0] * (x[0] - goal[0]); q[3] = c->We[1] * (x[1] - goal[1]); q[4] = 0.0;
        q[5] = c->We[2] * x[3]; q[6] = c->We[3] * x[4];
    }
}

void orc_linearize(const orc_config *c, const double *x0, const double *P, const double *goal,
                   const double *X, const double *U,
                   double *A, double *B, double *b, double *q, double *h, double *dh)
{
    (void)x0;
    int N = c->N, no = c->n_obst; double dt = c->Tf / N; double Hd[7];
    for (int i = 0; i < N; i++) {
        double xn[5];
        orc_dynamics(X + 5 * i, U + 2 * i, dt, xn, A + 25 * i, B + 10 * i);
        for (int k = 0; k < 5; k++) b[5 * i + k] = xn[k] - X[5 * (i + 1) + k];
    }
    for (int i = 0; i <= N; i++) {
        stage_cost_blocks(c, i, X + 5 * i, i < N ? U + 2 * i : NULL, goal, Hd, q + 7 * i);
        obstacle_h(c, X + 5 * i, P + 2 * no * i, h + no * i, dh + 2 * no * i);
    }
}

/* alpha_in: explicit per-stage slack weights zl_i = Zl_i (what a caller of cost_set(i,'zl'/'Zl') supplies, robot_ocp_problem.py:149-152),
 * or NULL for the reference's own schedule of :145-148 */
static double cost_with_alpha(const orc_config *c, const double *x0, const double *P, const double *goal,
                              const double *X, const double *U, const double *alpha_in)
{
    int N = c->N, no = c->n_obst; double cs = stage_cs(c), ss = stage_ss(c);
    double *alpha = (double *)WS_ALLOC(sizeof(double) * (N + 1));
    double *h = (double *)WS_ALLOC(sizeof(double) * no);
    if (alpha_in) memcpy(alpha, alpha_in, sizeof(double) * (N + 1)); else orc_slack_alpha(c, x0, goal, alpha);
    double J = 0;
    for (int i = 0; i <= N; i++) {
        const double *x = X + 5 * i;
        double ex = x[0] - goal[0], ey = x[1] - goal[1];
        if (i < N) {
            const double *u = U + 2 * i;
            J += 0.5 * cs * (c->W[0] * ex * ex + c->W[1] * ey * ey + c->W[2] * x[3] * x[3] + c->W[3] * x[4] * x[4]
                             + c->W[4] * u[0] * u[0] + c->W[5] * u[1] * u[1]);
        } else {
            J += 0.5 * (c->We[0] * ex * ex + c->We[1] * ey * ey + c->We[2] * x[3] * x[3] + c->We[3] * x[4] * x[4]);
        }
        obstacle_h(c, x, P + 2 * no * i, h, NULL);
        double sc = (i < N) ? ss : 1.0;
        for (int j = 0; j < no; j++) {
            double v = h[j] < 0 ? -h[j] : 0.0;
            J += sc * alpha[i] * (v + 0.5 * v * v);
        }
    }
    WS_FREE(alpha); WS_FREE(h);
    return J;
}

double orc_cost(const orc_config *c, const double *x0, const double *P, const double *goal,
                const double *X, const double *U)
{
    return cost_with_alpha(c, x0, P, goal, X, U, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* Generic OCP-QP + interior point                                                              */
/* ------------------------------------------------------------------------------------------ */
/*
 * Inequality "items".  Every item is rho = c0 + cz' z_i (+ s_j) >= 0 with slack t and multiplier lam.
 * kind 0: plain row (box or hard obstacle row); kind 1: soft row (has +s_j); kind 2: s_j >= 0.
 */
typedef struct {
    int stage, kind, sidx;   /* sidx: index of the slack variable for kinds 1,2 */
    double c0, cz[NZ];
    double lam, t, dlam, dt_, dlam_aff, dt_aff;
} item_t;

typedef struct {
    int N, n_items, n_s;
    double (*Hd)[NZ];      /* diagonal Hessian per stage (z = [u;x]); stage N uses x part */
    double (*q)[NZ];
    double (*A)[25], (*B)[10], (*b)[NX];
    double d0[NX];         /* required dx_0 */
    item_t *it;
    double *zs, *Zs;       /* slack penalties per slack variable */
    int *s_stage;
} qp_t;

static void qp_free(qp_t *Q)
{
    WS_FREE(Q->Hd); WS_FREE(Q->q); WS_FREE(Q->A); WS_FREE(Q->B); WS_FREE(Q->b); WS_FREE(Q->it); WS_FREE(Q->zs); WS_FREE(Q->Zs); WS_FREE(Q->s_stage);
}

static void add_box(qp_t *Q, int stage, int zidx, double val, double lo, double hi)
{
    item_t *e = &Q->it[Q->n_items++];
    memset(e, 0, sizeof(*e)); e->stage = stage; e->kind = 0; e->cz[zidx] = 1.0; e->c0 = val - lo;
    e = &Q->it[Q->n_items++];
    memset(e, 0, sizeof(*e)); e->stage = stage; e->kind = 0; e->cz[zidx] = -1.0; e->c0 = hi - val;
}

/* INVESTIGATION switches (orc_set_investigation, default 0; scripts/converged_unmatched.py -- VERDICT r04 item 3): hypotheses about what acados / HPIPM
 * kept in the QP that change the interior point's PATH, not the QP's solution:  1 = the obstacle rows of stage 0 are present (x_0 fixed: they only set sl_0),
 * 2 = rows whose slack penalty is zero (the terminal stage of the reference's schedule) are kept as free-slack rows instead of being dropped,
 * 4 = the stationarity residual is gated by the termination test as well (HPIPM's res_g).  An explicit call of the investigation scripts, per process (round 6:
 * it was an environment variable, which a stray setting in a driver's environment would have turned into another checker); never set by tests or by the product. */
static int g_investigation = 0;
void orc_set_investigation(int switches) { g_investigation = switches; }
static int orc_investigation(void) { return g_investigation; }

/* Build the QP of one RTI step */
static void build_qp(const orc_config *c, const double *x0, const double *P, const double *goal,
                     const double *X, const double *U, qp_t *Q, const double *alpha_in)
{
    int N = c->N, no = c->n_obst; double dt = c->Tf / N;
    Q->N = N;
    Q->Hd = WS_ALLOC(sizeof(double[NZ]) * (N + 1)); Q->q = WS_ALLOC(sizeof(double[NZ]) * (N + 1));
    Q->A = WS_ALLOC(sizeof(double[25]) * N); Q->B = WS_ALLOC(sizeof(double[10]) * N); Q->b = WS_ALLOC(sizeof(double[NX]) * N);
    int max_items = (N + 1) * (4 + 8 + 2 * no);
    Q->it = WS_ALLOC(sizeof(item_t) * max_items); Q->n_items = 0;
    Q->zs = WS_ALLOC(sizeof(double) * (N + 1) * no); Q->Zs = WS_ALLOC(sizeof(double) * (N + 1) * no);
    Q->s_stage = WS_ALLOC(sizeof(int) * (N + 1) * no); Q->n_s = 0;
    double *alpha = WS_ALLOC(sizeof(double) * (N + 1));
    double *h = WS_ALLOC(sizeof(double) * no), *dh = WS_ALLOC(sizeof(double) * 2 * no);
    if (alpha_in) memcpy(alpha, alpha_in, sizeof(double) * (N + 1)); else orc_slack_alpha(c, x0, goal, alpha);
    for (int k = 0; k < 5; k++) Q->d0[k] = x0[k] - X[k];    /* lbx_0 = ubx_0 = x0, robot_ocp_problem.py:191-192 */
    for (int i = 0; i < N; i++) {
        double xn[5];
        orc_dynamics(X + 5 * i, U + 2 * i, dt, xn, Q->A[i], Q->B[i]);
        for (int k = 0; k < 5; k++) Q->b[i][k] = xn[k] - X[5 * (i + 1) + k];
    }
    for (int i = 0; i <= N; i++) {
        stage_cost_blocks(c, i, X + 5 * i, i < N ? U + 2 * i : NULL, goal, Q->Hd[i], Q->q[i]);
        if (i < N) for (int k = 0; k < 2; k++) add_box(Q, i, k, U[2 * i + k], c->bu_lo[k], c->bu_hi[k]);   /* :95-97 */
        if (i >= 1 && (i < N || c->bx_terminal))                                                         /* :91-93 */
            for (int k = 0; k < 4; k++) add_box(Q, i, 2 + IDXBX[k], X[5 * i + IDXBX[k]], c->bx_lo[k], c->bx_hi[k]);
        if (i >= 1 || (orc_investigation() & 1)) { /* stage 0: x_0 is fixed so its rows are decoupled; investigation switch 1 keeps them */
            obstacle_h(c, X + 5 * i, P + 2 * no * i, h, dh);
            double sc = (i < N) ? stage_ss(c) : 1.0;
            for (int j = 0; j < no; j++) {
                if (c->soft_h) {
                    double z = sc * alpha[i];
                    if (!(z > 0.0) && !(orc_investigation() & 2)) continue;   /* zero penalty: the slack is free, the row is vacuous (investigation switch 2 keeps it as a free-slack row) */
                    int si = Q->n_s++;
                    Q->zs[si] = z; Q->Zs[si] = z; Q->s_stage[si] = i;   /* zl = Zl = alpha_i, :149-152 */
                    item_t *e = &Q->it[Q->n_items++];
                    memset(e, 0, sizeof(*e)); e->stage = i; e->kind = 1; e->sidx = si; e->c0 = h[j]; e->cz[2] = dh[2 * j]; e->cz[3] = dh[2 * j + 1];
                    e = &Q->it[Q->n_items++];
                    memset(e, 0, sizeof(*e)); e->stage = i; e->kind = 2; e->sidx = si; e->c0 = 0.0;
                } else {
                    item_t *e = &Q->it[Q->n_items++];
                    memset(e, 0, sizeof(*e)); e->stage = i; e->kind = 0; e->c0 = h[j]; e->cz[2] = dh[2 * j]; e->cz[3] = dh[2 * j + 1];
                }
            }
        }
    }
    WS_FREE(alpha); WS_FREE(h); WS_FREE(dh);
}

/* dense helpers, row-major */
static void mat_mul(int m, int k, int n, const double *Aa, const double *Bb, double *C) /* C = A(mxk) B(kxn) */
{
    for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) {
        double s = 0; for (int l = 0; l < k; l++) s += Aa[i * k + l] * Bb[l * n + j];
        C[i * n + j] = s;
    }
}

typedef struct {
    double (*P)[25];   /* cost-to-go Hessians */
    double (*p)[NX];
    double (*K)[10];   /* 2x5 */
    double (*k)[NU];
    double (*L)[4];    /* LDL' of Muu: {d0, 0, l, d1} */
    double (*Mxu)[10]; /* 5x2 */
} ricc_t;

/*
 * Riccati solve of  min sum 0.5 z'Ht z + g'z  s.t. dx_{i+1} = A dx_i + B du_i + r_i,  dx_0 = e0.
 * Ht_i = diag(Hd) + Haug_i (7x7 dense), factorised when `factor`; returns z (7 per stage) and
 * costates pi_i (multiplier of the equation defining x_{i+1}; pi index i+1, pi[0] for the initial condition).
 */
static void riccati(const qp_t *Q, double (*Ht)[49], double (*g)[NZ], double (*r)[NX], const double *e0,
                    ricc_t *R, int factor, double (*z)[NZ], double (*pi)[NX])
{
    int N = Q->N;
    /* terminal */
    if (factor) for (int a = 0; a < 5; a++) for (int bq = 0; bq < 5; bq++) R->P[N][a * 5 + bq] = Ht[N][(2 + a) * 7 + (2 + bq)];
    for (int a = 0; a < 5; a++) R->p[N][a] = g[N][2 + a];
    for (int i = N - 1; i >= 0; i--) {
        double Wm[35]; /* 5x7: [B A] */
        for (int a = 0; a < 5; a++) { Wm[a * 7 + 0] = Q->B[i][a * 2]; Wm[a * 7 + 1] = Q->B[i][a * 2 + 1]; for (int bq = 0; bq < 5; bq++) Wm[a * 7 + 2 + bq] = Q->A[i][a * 5 + bq]; }
        double PW[35], M[49], m[7], Pr[5];
        mat_mul(5, 5, 7, R->P[i + 1], Wm, PW);
        if (factor) {
            for (int a = 0; a < 7; a++) for (int bq = 0; bq < 7; bq++) {
                double s = Ht[i][a * 7 + bq];
                for (int l = 0; l < 5; l++) s += Wm[l * 7 + a] * PW[l * 7 + bq];
                M[a * 7 + bq] = s;
            }
            /* Cholesky of Muu */
            /* Muu = L D L' without square roots: when a state row's barrier weight makes B'PB nearly rank one, the second
             * pivot M11 - l*M01 is a difference of ~1e13-sized numbers and can round to a tiny negative value; a Cholesky
             * square root would turn that into NaN, LDL' just carries the (rounding-sized) pivot. L[i] = {d0, 0, l, d1}. */
            double d0 = M[0], l10 = M[7] / d0, d1 = M[8] - l10 * M[7];
            R->L[i][0] = d0; R->L[i][1] = 0; R->L[i][2] = l10; R->L[i][3] = d1;
            for (int a = 0; a < 5; a++) { R->Mxu[i][a * 2] = M[(2 + a) * 7 + 0]; R->Mxu[i][a * 2 + 1] = M[(2 + a) * 7 + 1]; }
            /* K = -Muu^{-1} Mux */
            for (int a = 0; a < 5; a++) {
                double r0 = M[0 * 7 + 2 + a], r1 = M[1 * 7 + 2 + a];
                double x1 = (r1 - l10 * r0) / d1, x0_ = r0 / d0 - l10 * x1;
                R->K[i][0 * 5 + a] = -x0_; R->K[i][1 * 5 + a] = -x1;
            }
            /* P_i = Mxx + Mxu K */
            for (int a = 0; a < 5; a++) for (int bq = 0; bq < 5; bq++)
                R->P[i][a * 5 + bq] = M[(2 + a) * 7 + 2 + bq] + R->Mxu[i][a * 2] * R->K[i][0 * 5 + bq] + R->Mxu[i][a * 2 + 1] * R->K[i][1 * 5 + bq];
            /* symmetrise */
            for (int a = 0; a < 5; a++) for (int bq = a + 1; bq < 5; bq++) { double s = 0.5 * (R->P[i][a * 5 + bq] + R->P[i][bq * 5 + a]); R->P[i][a * 5 + bq] = R->P[i][bq * 5 + a] = s; }
        }
        /* m = g_i + W'(P r + p) */
        for (int a = 0; a < 5; a++) { double s = R->p[i + 1][a]; for (int l = 0; l < 5; l++) s += R->P[i + 1][a * 5 + l] * r[i][l]; Pr[a] = s; }
        for (int a = 0; a < 7; a++) { double s = g[i][a]; for (int l = 0; l < 5; l++) s += Wm[l * 7 + a] * Pr[l]; m[a] = s; }
        {
            double d0 = R->L[i][0], l10 = R->L[i][2], d1 = R->L[i][3];
            double x1 = (m[1] - l10 * m[0]) / d1, x0_ = m[0] / d0 - l10 * x1;
            R->k[i][0] = -x0_; R->k[i][1] = -x1;
        }
        for (int a = 0; a < 5; a++) R->p[i][a] = m[2 + a] + R->Mxu[i][a * 2] * R->k[i][0] + R->Mxu[i][a * 2 + 1] * R->k[i][1];
    }
    /* forward */
    double xc[5]; for (int a = 0; a < 5; a++) xc[a] = e0[a];
    for (int a = 0; a < 5; a++) { double s = R->p[0][a]; for (int l = 0; l < 5; l++) s += R->P[0][a * 5 + l] * xc[l]; pi[0][a] = s; }
    for (int i = 0; i < N; i++) {
        double u[2];
        for (int a = 0; a < 2; a++) { double s = R->k[i][a]; for (int l = 0; l < 5; l++) s += R->K[i][a * 5 + l] * xc[l]; u[a] = s; }
        z[i][0] = u[0]; z[i][1] = u[1]; for (int a = 0; a < 5; a++) z[i][2 + a] = xc[a];
        double xn[5];
        for (int a = 0; a < 5; a++) {
            double s = r[i][a];
            for (int l = 0; l < 5; l++) s += Q->A[i][a * 5 + l] * xc[l];
            s += Q->B[i][a * 2] * u[0] + Q->B[i][a * 2 + 1] * u[1];
            xn[a] = s;
        }
        for (int a = 0; a < 5; a++) xc[a] = xn[a];
        for (int a = 0; a < 5; a++) { double s = R->p[i + 1][a]; for (int l = 0; l < 5; l++) s += R->P[i + 1][a * 5 + l] * xc[l]; pi[i + 1][a] = s; }
    }
    z[N][0] = z[N][1] = 0; for (int a = 0; a < 5; a++) z[N][2 + a] = xc[a];
}

typedef struct {
    double (*z)[NZ];   /* primal per stage */
    double *s;         /* slacks */
    double (*pi)[NX];  /* pi[0]: initial condition; pi[i+1]: dynamics i */
} iter_t;

/* explicit KKT residuals; returns inf-norms in res[4] = {stationarity, equality, inequality, complementarity(max lam*t)} */
static void residuals(const qp_t *Q, iter_t *I, double (*rg)[NZ], double *rs, double (*rb)[NX], double *re0, double *res)
{
    int N = Q->N;
    /* Costates by the adjoint recursion pi_i = (H z + q - C'lam)_x + A_i' pi_{i+1}: any costate is a valid
     * linearisation point (the Newton target does not depend on it), this choice zeroes the x-blocks of the
     * stationarity residual exactly and avoids the ill-conditioned products P_i x_i late in the iteration. */
    for (int i = N; i >= 0; i--) {
        double acc[5];
        for (int a = 0; a < 5; a++) acc[a] = Q->Hd[i][2 + a] * I->z[i][2 + a] + Q->q[i][2 + a];
        for (int e = 0; e < Q->n_items; e++) { const item_t *it = &Q->it[e]; if (it->stage != i) continue; for (int a = 0; a < 5; a++) acc[a] -= it->cz[2 + a] * it->lam; }
        if (i < N) for (int a = 0; a < 5; a++) { double sacc = 0; for (int l = 0; l < 5; l++) sacc += Q->A[i][l * 5 + a] * I->pi[i + 1][l]; acc[a] += sacc; }
        for (int a = 0; a < 5; a++) I->pi[i][a] = acc[a];
    }
    for (int i = 0; i <= N; i++) {
        for (int a = 0; a < 7; a++) rg[i][a] = Q->Hd[i][a] * I->z[i][a] + Q->q[i][a];
        if (i == N) { rg[i][0] = rg[i][1] = 0; }
        /* + [B' ; A'] pi_{i+1} - [0; pi_i] */
        if (i < N) {
            for (int a = 0; a < 2; a++) { double sacc = 0; for (int l = 0; l < 5; l++) sacc += Q->B[i][l * 2 + a] * I->pi[i + 1][l]; rg[i][a] += sacc; }
            for (int a = 0; a < 5; a++) { double sacc = 0; for (int l = 0; l < 5; l++) sacc += Q->A[i][l * 5 + a] * I->pi[i + 1][l]; rg[i][2 + a] += sacc; }
        }
        for (int a = 0; a < 5; a++) rg[i][2 + a] -= I->pi[i][a];
    }
    for (int j = 0; j < Q->n_s; j++) rs[j] = Q->Zs[j] * I->s[j] + Q->zs[j];
    double rd = 0, rm = 0;
    for (int e = 0; e < Q->n_items; e++) {
        const item_t *it = &Q->it[e];
        double rho = it->c0;
        for (int a = 0; a < 7; a++) { rho += it->cz[a] * I->z[it->stage][a]; rg[it->stage][a] -= it->cz[a] * it->lam; }
        if (it->kind == 1 || it->kind == 2) { rho += I->s[it->sidx]; rs[it->sidx] -= it->lam; }
        double d = fabs(rho - it->t); if (d > rd) rd = d;
        /* a pair whose t (or lam) sits at the floor is numerically active (inactive): t below ~1e-13 is under the
         * rounding resolution of rho = c0 + c'z, so its product no longer measures distance from the solution */
        double mm = (it->t <= 2 * TL_MIN || it->lam <= 2 * TL_MIN) ? 0.0 : fabs(it->lam * it->t); if (mm > rm) rm = mm;
    }
    double ng = 0, nb = 0;
    for (int i = 0; i <= N; i++) for (int a = 0; a < 7; a++) { if (i == 0 && a >= 2) continue; /* x_0 is fixed: its multiplier pi_0 absorbs this block */ double v = fabs(rg[i][a]); if (v > ng) ng = v; }
    for (int j = 0; j < Q->n_s; j++) { double v = fabs(rs[j]); if (v > ng) ng = v; }
    for (int a = 0; a < 5; a++) { re0[a] = Q->d0[a] - I->z[0][2 + a]; double v = fabs(re0[a]); if (v > nb) nb = v; }
    for (int i = 0; i < N; i++) for (int a = 0; a < 5; a++) {
        double sacc = Q->b[i][a] - I->z[i + 1][2 + a];
        for (int l = 0; l < 5; l++) sacc += Q->A[i][a * 5 + l] * I->z[i][2 + l];
        sacc += Q->B[i][a * 2] * I->z[i][0] + Q->B[i][a * 2 + 1] * I->z[i][1];
        rb[i][a] = sacc; double v = fabs(sacc); if (v > nb) nb = v;
    }
    res[0] = ng; res[1] = nb; res[2] = rd; res[3] = rm;
}

/*
 * Mehrotra predictor-corrector primal-dual IPM, cold-started, separate step lengths for primal and dual,
 * Riccati factorisation of the reduced KKT system [acados-knowledge: this is the structure of HPIPM's
 * OCP-QP solver, which robot_ocp_problem.py:126 selects; tolerance/caps are ours].
 * Returns 0 converged, 2 max-iter, 4 failure (NaN / step collapse).
 */
/* optional per-iteration trace (mu, sigma, alpha, cmax) for debugging parity; not thread-safe */
#define MU_DIVERGED 1e8
#define MU_CAP_FAILED 1e4
#define MU_CAP_SETTLED 20
#define POLISH_MAX 2
#define POLISH_UNSOLVED 100.0f   /* a solve whose step estimate is still this many polish_tol after the polish is reported as not converged (status 2) */
#define STATIONARITY_STEP 1e-6f   /* indicator (c) of the polish is formed only if some stage's last step was longer than this (the stated parity tolerance): behind a shorter step
                                   * the iterate is within the tolerance by the step length itself, and the residual would be formed in every solve instead of every second */
#define FRAC_TO_BOUNDARY 0.999995   /* step = this fraction of the largest step that keeps t, lam > 0 */
static double *g_trace = NULL; static int g_trace_cap = 0;
void orc_set_trace(double *buf, int cap) { g_trace = buf; g_trace_cap = cap; }

/* the polish's step indicator: est > tol with est = max(s r min(1, 10 r), kappa s), r = min(s / s', 1/2).  s r min(1, 10 r) is the smallest of s / 2, s^2 / s' and
 * 10 s^3 / s'^2, so that part is the conjunction of three comparisons; kappa s is the floor of the estimate: what the last Newton step of a long horizon leaves
 * behind whatever contraction was observed (polish_step_frac).  Float arithmetic, no division, products left to right (rti_kernel.hpp::polish_wanted does exactly this) */
static _Thread_local int g_last_dead = 0, g_last_npolish = 0, g_last_settled_it = -1; static _Thread_local float g_last_step = 0;
static _Thread_local double g_last_weak = 0; double orc_last_weak(void) { return g_last_weak; }
int orc_last_dead_pairs(void) { return g_last_dead; }
int orc_last_settled_it(void) { return g_last_settled_it; }
int orc_last_npolish(void) { return g_last_npolish; }
double orc_last_step_norm(void) { return (double)g_last_step; }
static int polish_wanted(float s, float sp, float tol, float kappa)
{
    const float pa = 0.5f * s, pb = s * s, qb = tol * sp, pc = 10.0f * s * s * s, qc = tol * sp * sp, pd = kappa * s;
    return (pa > tol && pb > qb && pc > qc) || pd > tol;
}

static _Thread_local double g_last_res_g = 0;
double orc_last_res_g(void) { return g_last_res_g; }

static int ipm_solve(const orc_config *c, qp_t *Q, iter_t *I, int *iters_out, double *kkt)
{
    int N = Q->N, ni = Q->n_items, ns = Q->n_s;
    double (*rg)[NZ] = WS_ALLOC(sizeof(double[NZ]) * (N + 1));
    double (*rb)[NX] = WS_ALLOC(sizeof(double[NX]) * (N + 1));
    double *rs = WS_ALLOC(sizeof(double) * (ns + 1));
    double (*Ht)[49] = WS_ALLOC(sizeof(double[49]) * (N + 1));
    double (*gt)[NZ] = WS_ALLOC(sizeof(double[NZ]) * (N + 1));
    double (*dz)[NZ] = WS_ALLOC(sizeof(double[NZ]) * (N + 1));
    double (*dpi)[NX] = WS_ALLOC(sizeof(double[NX]) * (N + 1));
    double *ds = WS_ALLOC(sizeof(double) * (ns + 1)), *yds = WS_ALLOC(sizeof(double) * (ns + 1));
    double *w1 = WS_ALLOC(sizeof(double) * (ns + 1)), *w2 = WS_ALLOC(sizeof(double) * (ns + 1));
    double *be1 = WS_ALLOC(sizeof(double) * (ns + 1)), *be2 = WS_ALLOC(sizeof(double) * (ns + 1));
    int *soft_row = WS_ALLOC(sizeof(int) * (ns + 1)), *soft_pos = WS_ALLOC(sizeof(int) * (ns + 1));
    ricc_t R;
    R.P = WS_ALLOC(sizeof(double[25]) * (N + 1)); R.p = WS_ALLOC(sizeof(double[NX]) * (N + 1));
    R.K = WS_ALLOC(sizeof(double[10]) * N); R.k = WS_ALLOC(sizeof(double[NU]) * N); R.L = WS_ALLOC(sizeof(double[4]) * N); R.Mxu = WS_ALLOC(sizeof(double[10]) * N);
    double re0[5], res[4];
    int status = 2, it = 0, npolish = 0, long_step = 0; double cprev = INFINITY;      /* cprev: c_max at the head of the previous iteration */
    float *st_now = WS_ALLOC(sizeof(float) * 2 * (N + 1)), *st_prev = st_now + (N + 1);      /* per stage: max-norm of the last primal step and of the one before */
    for (int i = 0; i < 2 * (N + 1); i++) st_now[i] = 0.0f;
    TL_MIN = TL_MIN_MAX < 0.1 * c->qp_tol ? TL_MIN_MAX : 0.1 * c->qp_tol;
    g_last_settled_it = -1; g_last_res_g = 0;

    for (int e = 0; e < ni; e++) { if (Q->it[e].kind == 1) soft_row[Q->it[e].sidx] = e; if (Q->it[e].kind == 2) soft_pos[Q->it[e].sidx] = e; }

    /* cold start: z = 0, s = 0, pi = 0, t = max(rho, thr0), lam = mu0 / t */
    for (int i = 0; i <= N; i++) { memset(I->z[i], 0, sizeof(double[NZ])); memset(I->pi[i], 0, sizeof(double[NX])); }
    /* soft rows start strictly feasible: s = max(0, -h) + thr0, so rho1 = h + s >= thr0 and rho2 = s >= thr0 */
    for (int j = 0; j < ns; j++) { double h = Q->it[soft_row[j]].c0; I->s[j] = (h < 0 ? -h : 0.0) + c->thr0; }
    for (int e = 0; e < ni; e++) {
        item_t *q = &Q->it[e];
        double rho = q->c0 + (q->kind ? I->s[q->sidx] : 0.0);
        q->t = rho > c->thr0 ? rho : c->thr0; q->lam = c->mu0 / q->t;
    }

    for (it = 0; ; it++) {
        residuals(Q, I, rg, rs, rb, re0, res);
        double mu = 0; for (int e = 0; e < ni; e++) mu += Q->it[e].lam * Q->it[e].t; mu = ni ? mu / ni : 0.0;
        if (!(res[0] == res[0]) || !(res[1] == res[1]) || !(mu == mu)) { status = 4; break; }
        if (it >= MU_CAP_SETTLED && mu > c->mu0 && g_last_settled_it < 0) g_last_settled_it = it;      /* diagnostic (orc_last_settled_it) */
        /* divergence (an infeasible QP: the hard boxes cannot be met): the complementarity measure of a healthy solve never leaves [~0, 1e2 mu0]
         * (measured: <= 7e5 at mu0 = 1e4), that of an infeasible one grows without bound -- stop at 1e8 mu0 instead of iterating until the step
         * collapses or the cap; shared with the HIP kernels (status 4, iterate untouched) */
        if (c->qp_fail_policy == 0 && mu > MU_DIVERGED * c->mu0) { status = 4; break; }
        if (!(fabs(mu) <= 1e300)) { status = 4; break; }      /* (qp_fail_policy 1: only an overflow ends a diverging solve early) */
        /* Termination (shared spec with the HIP kernel): linear residuals (dynamics, initial condition, rho - t; they
         * all decay by the same factor prod(1 - alpha_k)) and the largest complementarity product below qp_tol.
         * The stationarity residual res[0] is REPORTED, not gated: late in the iteration its rounding floor is
         * ~ eps * lam^2 |z| / mu for active rows (multiplier accuracy), while the primal point is unaffected. */
        if (res[1] <= c->qp_tol && res[2] <= c->qp_tol && res[3] <= c->qp_tol && (!(orc_investigation() & 4) || res[0] <= c->qp_tol)) {
            /* POLISH (round 5; shared with the HIP kernels).  Meeting the tolerance does not bound the distance from the QP's solution; what solves left behind
             * when their products slipped under qp_tol was the parity tail beyond 1e-6 (up to 1.7e-5 from the exact solution on 0.7 % of
             * the first solves of C5's problem).  Two classes, two indicators; while either holds the solve continues (at most POLISH_MAX further iterations):
             *  (a) SLOW END-GAME -- pairs at the floor or weakly active rows make the last iterations contract by 0.3 .. 0.8 instead of superlinearly (one
             *      iteration of a healthy end-game takes the largest live product c_max down by four orders of magnitude and more):
             *          c_max(k) > polish_ratio * c_max(k - 1);
             *  (b) A MULTIPLIER COLLAPSED TO THE FLOOR on a weakly active row: complementary and primal feasible, hence invisible to the termination test (the
             *      stationarity residual is not gated: the kernels never form it), but the last primal step is still long.  Per stage i, with s_i the max-norm
             *      of the stage's last step alpha * dz_i, s_i' of the one before and r = min(s_i / s_i', 1/2) the observed contraction, the estimate
             *      s_i r min(1, 10 r) = min(s_i / 2, s_i^2 / s_i', 10 s_i^3 / s_i'^2) of what remains exceeds polish_tol for ANY stage.  Float arithmetic, as in
             *      polish_wanted(), so that both sides decide alike; per stage because a stage is a lane of the kernels: no cross-lane reduction, one ballot.
             *      From N = 30 on the estimate has the floor polish_step_frac * s_i (see orc_default_config).
             * Measured (scripts/polish_probe.py on 8000 first / second solves of C5's problem): beyond 1e-6 from the exact solution 55 -> 0, beyond 1e-7
             * 144 -> 25, at +0.5 % iterations. */
            int want = c->polish_ratio > 0 && res[3] > c->polish_ratio * cprev, unsolved = 0;
            /*  (c) (round 6) THE STATIONARITY RESIDUAL of the QP's Lagrangian (HPIPM's res_g, which acados' defaults gate: robot_ocp_problem.py:126-132 leaves HPIPM's
             *      tolerances alone) above polish_res_g.  With the adjoint costates the state blocks of the residual vanish, so res[0] is the input blocks
             *      (H z + q - C' lam)_u + B_i' pi_{i+1} and the slack equations Z s + z - lam_1 - lam_2: what the kernels form with one open-loop adjoint sweep
             *      (rti_kernel.hpp::adjoint_inputs) -- and only where it decides: the termination test holds, (a) and (b) are silent, a polish iteration is left and
             *      some stage's last step was longer than STATIONARITY_STEP (1e-6: behind a shorter step the step length itself bounds the remaining error;
             *      measured on 25000 converged solves of two problem sizes, 97 of the 98 with a residual above 1e-7 stand behind such a step, the other one's is 4e-10).
             *      It sees what (b) sees through the step length, directly: a multiplier that collapsed to the floor on a weakly active row leaves the gradient of
             *      the Lagrangian unbalanced.  Measured (scripts/polish_probe.py, 4000 first / second solves of C5's problem, polish_res_g 1e-7): beyond 1e-7 from
             *      the exact solution 5 -> 1, worst 1.3e-6 -> 2.4e-7, +0.11 % iterations. */
            if (c->polish_res_g > 0 && it > 0 && it < c->qp_iter_max && !want && npolish < POLISH_MAX && long_step) {
                g_last_res_g = res[0];
                if (res[0] > c->polish_res_g) want = 1;
            }
            if (c->polish_tol > 0) for (int i = 0; i <= N; i++) {
                want = want || polish_wanted(st_now[i], st_prev[i], (float)c->polish_tol, (float)c->polish_step_frac);
                unsolved = unsolved || polish_wanted(st_now[i], st_prev[i], POLISH_UNSOLVED * (float)c->polish_tol, (float)c->polish_step_frac);
            }
            /* NOT SOLVED: the polish is used up and the step estimate still stands two orders of magnitude above polish_tol.  That is not a tail but an end-game
             * whose Newton steps have lost their accuracy to the barrier weights lam / t_floor (found by scripts/fuzz_parity.py: a stale warm start, every
             * termination test at zero, the "solution" 2e-2 from the QP's -- and 8e-3 under round 4's rules): the step is applied as after an iteration cap,
             * status 2, instead of being reported as converged.  Healthy solves that use up the polish end with estimates <= 3e-6 (measured, 4 problem sizes). */
            if (want && npolish >= POLISH_MAX && unsolved) { status = 2; break; }
            if (!want || npolish >= POLISH_MAX || it >= c->qp_iter_max) { status = 0; break; }
            npolish++;
        }
        /* at the cap: a complementarity measure far above anything a healthy solve shows (<= ~1e2 mu0, early in the iteration) means the QP was on its way
         * to MU_DIVERGED (infeasible), not converging slowly -- its step is garbage and must not be applied (status 4, as every other failure);
         * otherwise: max-iter, step applied.  From iteration MU_CAP_SETTLED on the bar is mu0 itself: a healthy solve is three orders of
         * magnitude below its starting value by then (measured: <= 12 at iteration 20, but up to 1.01e4 at iteration 10), a stalled infeasible one orders above --
         * with the single bar at 1e4 mu0 the stalled ones straddled it (scripts/fuzz_parity.py, hard obstacle rows) */
        if (it >= c->qp_iter_max) { status = (c->qp_fail_policy == 0 && (mu > MU_CAP_FAILED * c->mu0 || (it >= MU_CAP_SETTLED && mu > c->mu0))) ? 4 : 2; break; }

        cprev = res[3];
        double sigma = 0.0; double alpha = 1.0, alphad = 1.0;
        /* centring target sigma * min(mu, cmax), cmax = res[3] the largest product of a pair OFF the floor (round 5): pairs whose t sits at the floor keep
         * products lam * t_floor of 1e-5 and more in the mean mu for ever, and a target sigma * mu dominated by them held the products of the live pairs at
         * ~1e-11 instead of letting them go to zero (end-game contraction 0.3 .. 0.8 per iteration instead of superlinear).  sigma itself is the ratio of
         * two means over ALL pairs, as before. */
        const double mu_c = mu < res[3] ? mu : res[3];
        for (int pass = 0; pass < 2; pass++) {
            /* reduced Hessian / gradient */
            for (int i = 0; i <= N; i++) {
                memset(Ht[i], 0, sizeof(double[49]));
                for (int a = 0; a < 7; a++) { Ht[i][a * 7 + a] = Q->Hd[i][a]; gt[i][a] = rg[i][a]; }
                if (i == N) { Ht[i][0] = Ht[i][8] = 1.0; gt[i][0] = gt[i][1] = 0; }
            }
            for (int e = 0; e < ni; e++) {
                item_t *q = &Q->it[e];
                double rd = q->c0; for (int a = 0; a < 7; a++) rd += q->cz[a] * I->z[q->stage][a];
                if (q->kind) rd += I->s[q->sidx];
                rd -= q->t;
                double rm = q->lam * q->t - sigma * mu_c;
                if (pass == 1) rm += q->dlam_aff * q->dt_aff;
                double w = q->lam / q->t, beta = (rm + q->lam * rd) / q->t;
                if (q->kind == 0) {
                    for (int a = 0; a < 7; a++) { if (q->cz[a] == 0.0) continue; gt[q->stage][a] += q->cz[a] * beta; for (int bq = 0; bq < 7; bq++) Ht[q->stage][a * 7 + bq] += w * q->cz[a] * q->cz[bq]; }
                } else if (q->kind == 1) { w1[q->sidx] = w; be1[q->sidx] = beta; }
                else { w2[q->sidx] = w; be2[q->sidx] = beta; }
            }
            for (int j = 0; j < ns; j++) {
                item_t *q = &Q->it[soft_row[j]];
                double D = Q->Zs[j] + w1[j] + w2[j];
                /* cancellation-free forms of w1 - w1^2/D and be1 - w1 (rs + be1 + be2)/D */
                double weff = w1[j] * (Q->Zs[j] + w2[j]) / D;
                double geff = (be1[j] * (Q->Zs[j] + w2[j]) - w1[j] * (rs[j] + be2[j])) / D;
                for (int a = 0; a < 7; a++) { if (q->cz[a] == 0.0) continue; gt[q->stage][a] += q->cz[a] * geff; for (int bq = 0; bq < 7; bq++) Ht[q->stage][a * 7 + bq] += weff * q->cz[a] * q->cz[bq]; }
            }
            riccati(Q, Ht, gt, rb, re0, &R, pass == 0, dz, dpi);
            /* recover ds, dt, dlam */
            for (int j = 0; j < ns; j++) {
                item_t *q = &Q->it[soft_row[j]];
                double y = 0; for (int a = 0; a < 7; a++) y += q->cz[a] * dz[q->stage][a];
                double D = Q->Zs[j] + w1[j] + w2[j];
                ds[j] = -(rs[j] + be1[j] + be2[j] + w1[j] * y) / D;
                /* y + ds without the cancellation it suffers when w1 >> Z + w2 (active row late in the iteration) */
                yds[j] = (y * (Q->Zs[j] + w2[j]) - (rs[j] + be1[j] + be2[j])) / D;
            }
            double amax = 1.0, amaxd = 1.0;   /* largest primal (t) and dual (lam) steps that keep positivity */
            for (int e = 0; e < ni; e++) {
                item_t *q = &Q->it[e];
                double rd = q->c0; for (int a = 0; a < 7; a++) rd += q->cz[a] * I->z[q->stage][a];
                if (q->kind) rd += I->s[q->sidx];
                rd -= q->t;
                double dt_ = rd;
                if (q->kind == 1) dt_ += yds[q->sidx];
                else { for (int a = 0; a < 7; a++) dt_ += q->cz[a] * dz[q->stage][a]; if (q->kind == 2) dt_ += ds[q->sidx]; }
                double rm = q->lam * q->t - sigma * mu_c; if (pass == 1) rm += q->dlam_aff * q->dt_aff;
                double dl = -(rm + q->lam * dt_) / q->t;
                q->dt_ = dt_; q->dlam = dl;
                if (dt_ < 0) { double a_ = -q->t / dt_; if (a_ < amax) amax = a_; }
                if (dl < 0) { double a_ = -q->lam / dl; if (a_ < amaxd) amaxd = a_; }
            }
            if (pass == 0) {
                double mu_aff = 0;
                for (int e = 0; e < ni; e++) { item_t *q = &Q->it[e]; mu_aff += (q->lam + amaxd * q->dlam) * (q->t + amax * q->dt_); q->dlam_aff = q->dlam; q->dt_aff = q->dt_; }
                mu_aff = ni ? mu_aff / ni : 0.0;
                double ratio = mu > 0 ? mu_aff / mu : 0.0;
                sigma = ratio * ratio;      /* centring: (mu_aff / mu)^2, shared with the HIP kernels (rti_kernel.hpp kFracToBoundary) */
                if (sigma > 1.0) sigma = 1.0;
                if (ni == 0) { alpha = 1.0; break; }
            } else {
                /* separate primal and dual step lengths (z, s, t move by alpha; lam by alphad) */
                alpha = FRAC_TO_BOUNDARY * amax; if (amax >= 1.0) alpha = 1.0; if (alpha > 1.0) alpha = 1.0;
                alphad = FRAC_TO_BOUNDARY * amaxd; if (amaxd >= 1.0) alphad = 1.0;
            }
        }
        if (g_trace && it < g_trace_cap) { g_trace[4 * it] = mu; g_trace[4 * it + 1] = sigma; g_trace[4 * it + 2] = alpha; g_trace[4 * it + 3] = res[3]; }
        /* step collapse -- or a NaN / overflow of the row state, which surfaces in the centring target sigma * mu_c first (the kernels' floors would wash it out of t and lam:
         * rti_kernel.hpp MPC_NAN_NOTE, ipm_step_check; mirrored here since round 6 so that both sides end such a solve at the same iteration) */
        if (!(alpha > 1e-14) || !(alphad > 1e-14) || !(sigma * mu_c == sigma * mu_c)) { status = 4; break; }
        for (int i = 0; i <= N; i++) {      /* per-stage step norms for the polish: (float)alpha * (float)max_a |dz_i[a]| */
            double m_ = 0; for (int a = 0; a < 7; a++) { double m = fabs(dz[i][a]); if (m > m_) m_ = m; }
            st_prev[i] = st_now[i]; st_now[i] = (float)alpha * (float)m_;
        }
        long_step = 0; for (int i = 0; i <= N; i++) if (st_now[i] > STATIONARITY_STEP) long_step = 1;      /* indicator (c) is asked only behind a step that was still long */
        for (int i = 0; i <= N; i++) { for (int a = 0; a < 7; a++) I->z[i][a] += alpha * dz[i][a]; }
        for (int j = 0; j < ns; j++) I->s[j] += alpha * ds[j];
        for (int e = 0; e < ni; e++) {
            item_t *q = &Q->it[e]; q->t += alpha * q->dt_; q->lam += alphad * q->dlam;
            if (q->t < TL_MIN) q->t = TL_MIN;      /* floors as in HPIPM's t_min / lam_min [acados-knowledge]: keep */
            if (q->lam < TL_MIN) q->lam = TL_MIN;  /* lam/t finite once a pair has collapsed below rounding        */
        }
    }
    if (status != 4) {   /* a step that is not finite is not a step (rti_kernel.hpp ipm_finite_step): an overflow that reached z without passing through mu or sigma */
        for (int i = 0; i <= N; i++) { double fin = 0; for (int a = 0; a < 7; a++) fin += I->z[i][a]; if (!(fabs(fin) <= 1e300)) status = 4; }
    }
    {   /* diagnostics of the last solve on this thread (orc_last_dead_pairs): pairs with BOTH t and lam at the floor -- complementary whatever the row does */
        int dead = 0; for (int e = 0; e < ni; e++) if (Q->it[e].t <= 2 * TL_MIN && Q->it[e].lam <= 2 * TL_MIN) dead++;
        g_last_dead = dead; g_last_npolish = npolish;
        { double w = INFINITY; for (int e = 0; e < ni; e++) { double m = Q->it[e].t > Q->it[e].lam ? Q->it[e].t : Q->it[e].lam; if (m < w) w = m; } g_last_weak = w; }
        g_last_step = 0; for (int i = 0; i <= N; i++) if (st_now[i] > g_last_step) g_last_step = st_now[i];
    }
    if (iters_out) *iters_out = it;
    if (kkt) for (int a = 0; a < 4; a++) kkt[a] = res[a];
    WS_FREE(st_now); WS_FREE(rg); WS_FREE(rb); WS_FREE(rs); WS_FREE(Ht); WS_FREE(gt); WS_FREE(dz); WS_FREE(dpi); WS_FREE(ds); WS_FREE(yds); WS_FREE(w1); WS_FREE(w2); WS_FREE(be1); WS_FREE(be2); WS_FREE(soft_row); WS_FREE(soft_pos);
    WS_FREE(R.P); WS_FREE(R.p); WS_FREE(R.K); WS_FREE(R.k); WS_FREE(R.L); WS_FREE(R.Mxu);
    return status;
}

/* robot_ocp_problem.py:186-198 around ocp_solver.solve(): one SQP_RTI iteration = QP + full step */
int orc_rti_solve_alpha(const orc_config *c, const double *x0, const double *P, const double *goal, const double *alpha,
                        double *X, double *U, double *u0, double *cost, int *iters, double *kkt)
{
    int N = c->N;
    ws_enter();
    {   /* non-finite inputs fail at once (status 4, iterate untouched): shared specification with the HIP kernels, which would otherwise
         * lose a NaN in the fmax() of their residual norms */
        double fin = goal[0] + goal[1];
        for (int k = 0; k < 5; k++) fin += x0[k];
        for (int k = 0; k < 5 * (N + 1); k++) fin += X[k];
        for (int k = 0; k < 2 * N; k++) fin += U[k];
        for (int k = 0; k < 2 * c->n_obst * (N + 1); k++) fin += P[k];
        if (!(fabs(fin) <= 1e300)) {
            if (u0) { u0[0] = U[0]; u0[1] = U[1]; }
            if (cost) *cost = NAN;
            if (iters) *iters = 0;
            if (kkt) for (int a = 0; a < 4; a++) kkt[a] = NAN;
            ws_leave();
            return 4;
        }
    }
    qp_t Q; build_qp(c, x0, P, goal, X, U, &Q, alpha);
    iter_t I;
    I.z = WS_ALLOC(sizeof(double[NZ]) * (N + 1)); I.pi = WS_ALLOC(sizeof(double[NX]) * (N + 1)); I.s = WS_ALLOC(sizeof(double) * (Q.n_s + 1));
    int status = ipm_solve(c, &Q, &I, iters, kkt);
    if (status != 4) { /* full step; a max-iter QP still has its step applied (3.2-6) */
        for (int i = 0; i <= N; i++) for (int k = 0; k < 5; k++) X[5 * i + k] += I.z[i][2 + k];
        for (int i = 0; i < N; i++) for (int k = 0; k < 2; k++) U[2 * i + k] += I.z[i][k];
    }
    if (u0) { u0[0] = U[0]; u0[1] = U[1]; }
    if (cost) *cost = cost_with_alpha(c, x0, P, goal, X, U, alpha);
    WS_FREE(I.z); WS_FREE(I.pi); WS_FREE(I.s); qp_free(&Q);
    ws_leave();
    return status;
}

int orc_rti_solve(const orc_config *c, const double *x0, const double *P, const double *goal,
                  double *X, double *U, double *u0, double *cost, int *iters, double *kkt)
{
    return orc_rti_solve_alpha(c, x0, P, goal, NULL, X, U, u0, cost, iters, kkt);
}

void orc_rti_solve_batch(const orc_config *c, int batch, const double *x0, const double *P, const double *goal,
                         double *X, double *U, double *u0, double *cost, int *status, int *iters, int nthreads)
{
    int N = c->N, no = c->n_obst;
    size_t sP = (size_t)(N + 1) * no * 2, sX = (size_t)(N + 1) * 5, sU = (size_t)N * 2;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int b = 0; b < batch; b++) {
        int it = 0;
        int st = orc_rti_solve(c, x0 + 5 * (size_t)b, P + sP * b, goal + 2 * (size_t)b, X + sX * b, U + sU * b,
                               u0 ? u0 + 2 * (size_t)b : NULL, cost ? cost + b : NULL, &it, NULL);
        if (status) status[b] = st;
        if (iters) iters[b] = it;
    }
    (void)nthreads;
}

/* bench.py's cpu_baseline only: what surrounds the solve in a closed loop, for a whole batch in one call (the ~6 Python -> C calls per scenario and control step
 * they replace took ten times as long as the timed solves themselves).  orc_predict_params_batch: P[b] from obst[b]; orc_advance_batch: x[b] <- F(x[b], u0[b])
 * (robot_ocp_problem.py:207-212), obstacles one noise-free step (visualization.py:20-23), warm-start shift (:253-258). */
void orc_predict_params_batch(const orc_config *c, int batch, const double *obst, double *P)
{
    size_t sP = (size_t)(c->N + 1) * c->n_obst * 2;
    for (int b = 0; b < batch; b++) orc_predict_params(c, obst + (size_t)b * c->n_obst * 4, P + sP * b);
}
void orc_advance_batch(const orc_config *c, int batch, double *x, const double *u0, double *obst, double *X, double *U)
{
    int N = c->N, no = c->n_obst; double dt = c->Tf / N;
    for (int b = 0; b < batch; b++) {
        double xn[5];
        orc_dynamics(x + 5 * (size_t)b, u0 + 2 * (size_t)b, dt, xn, NULL, NULL);
        memcpy(x + 5 * (size_t)b, xn, sizeof(xn));
        for (int j = 0; j < no; j++) orc_obstacle_step(c, obst + ((size_t)b * no + j) * 4, dt, NULL, 0.0, 0.0);
        orc_shift(c, X + (size_t)b * (N + 1) * 5, U + (size_t)b * N * 2);
    }
}

int orc_export_qp(const orc_config *c, const double *x0, const double *P, const double *goal,
                  const double *X, const double *U,
                  double *H, double *g, double *Aeq, double *beq, double *lb, double *ub,
                  double *Cs, double *hs, double *zs, double *Zs)
{
    int N = c->N; int nv = 7 * N;
    qp_t Q; build_qp(c, x0, P, goal, X, U, &Q, NULL);
    /* variable map: du_i -> 7i + {0,1} ; dx_i (i>=1) -> 7(i-1) + 2 + k */
    memset(H, 0, sizeof(double) * nv * nv); memset(g, 0, sizeof(double) * nv);
    memset(Aeq, 0, sizeof(double) * 5 * N * nv); memset(beq, 0, sizeof(double) * 5 * N);
    for (int v = 0; v < nv; v++) { lb[v] = -INFINITY; ub[v] = INFINITY; }
    for (int i = 0; i < N; i++) for (int k = 0; k < 2; k++) { int v = 7 * i + k; H[v * nv + v] = Q.Hd[i][k]; g[v] = Q.q[i][k]; }
    for (int i = 1; i <= N; i++) for (int k = 0; k < 5; k++) { int v = 7 * (i - 1) + 2 + k; H[v * nv + v] = Q.Hd[i][2 + k]; g[v] = Q.q[i][2 + k]; }
    /* dynamics rows: dx_{i+1} - A dx_i - B du_i = b_i  (dx_0 = d0 moved to rhs) */
    for (int i = 0; i < N; i++) for (int a = 0; a < 5; a++) {
        int r = 5 * i + a; double rhs = Q.b[i][a];
        Aeq[r * nv + 7 * i + 2 + a] = 1.0;
        for (int k = 0; k < 2; k++) Aeq[r * nv + 7 * i + k] = -Q.B[i][a * 2 + k];
        for (int l = 0; l < 5; l++) {
            if (i == 0) rhs += Q.A[i][a * 5 + l] * Q.d0[l];
            else Aeq[r * nv + 7 * (i - 1) + 2 + l] = -Q.A[i][a * 5 + l];
        }
        beq[r] = rhs;
    }
    int ns = 0;
    for (int e = 0; e < Q.n_items; e++) {
        item_t *it = &Q.it[e];
        if (it->kind == 0) {
            int nnz = 0, idx = -1; for (int a = 0; a < 7; a++) if (it->cz[a] != 0.0) { nnz++; idx = a; }
            int v = (idx < 2) ? 7 * it->stage + idx : 7 * (it->stage - 1) + idx;
            if (nnz == 1 && it->cz[idx] == 1.0) lb[v] = -it->c0;       /* z - (lo - val) >= 0 */
            else if (nnz == 1 && it->cz[idx] == -1.0) ub[v] = it->c0;
            else { /* hard obstacle row: export as a soft row with infinite penalty marker */
                memset(Cs + (size_t)ns * nv, 0, sizeof(double) * nv);
                for (int a = 2; a < 7; a++) Cs[(size_t)ns * nv + 7 * (it->stage - 1) + a] = it->cz[a];
                hs[ns] = it->c0; zs[ns] = INFINITY; Zs[ns] = INFINITY; ns++;
            }
        } else if (it->kind == 1) {
            memset(Cs + (size_t)ns * nv, 0, sizeof(double) * nv);
            for (int a = 2; a < 7; a++) Cs[(size_t)ns * nv + 7 * (it->stage - 1) + a] = it->cz[a];
            hs[ns] = it->c0; zs[ns] = Q.zs[it->sidx]; Zs[ns] = Q.Zs[it->sidx]; ns++;
        }
    }
    qp_free(&Q);
    return ns;
}
