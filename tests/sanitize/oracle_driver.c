/* AddressSanitizer / UBSan driver for the CPU oracle (tests/test_sanitizers.py builds it with -fsanitize=address,undefined together with
 * oracle/mpc_oracle.c and runs it on the CPU).  It walks every exported entry point on seeded random problems, including the shapes the
 * parity tests use (N = 2 .. 62, 3 / 5 / 10 obstacles, the batched OpenMP driver, the dense QP export).  Test infrastructure only. */
#include "mpc_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

static unsigned long long s_ = 88172645463325252ULL;
static double urand(double lo, double hi)
{
    s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17;
    return lo + (hi - lo) * (double)(s_ >> 11) / 9007199254740992.0;
}

static int one_config(int N, int no, int batch)
{
    orc_config c;
    orc_default_config(&c, N, no, 0.1 * N);
    size_t sP = (size_t)(N + 1) * no * 2, sX = (size_t)(N + 1) * 5, sU = (size_t)N * 2;
    double *x0 = malloc(sizeof(double) * 5 * batch), *goal = malloc(sizeof(double) * 2 * batch), *obst = malloc(sizeof(double) * 4 * no * batch);
    double *P = malloc(sizeof(double) * sP * batch), *X = malloc(sizeof(double) * sX * batch), *U = malloc(sizeof(double) * sU * batch);
    double *u0 = malloc(sizeof(double) * 2 * batch), *cost = malloc(sizeof(double) * batch), *alpha = malloc(sizeof(double) * (N + 1));
    int *status = malloc(sizeof(int) * batch), *iters = malloc(sizeof(int) * batch);
    for (int b = 0; b < batch; b++) {
        double *x = x0 + 5 * b;
        x[0] = urand(-6, 6); x[1] = urand(-6, 6); x[2] = urand(-3, 3); x[3] = 0; x[4] = 0;
        goal[2 * b] = urand(-6, 6); goal[2 * b + 1] = urand(-6, 6);
        for (int j = 0; j < no; j++) {
            double *o = obst + 4 * (no * b + j);
            o[0] = urand(-4.4, 6); o[1] = urand(-4.4, 6); o[2] = urand(-2, 2); o[3] = urand(-2, 2);
        }
        orc_predict_params(&c, obst + 4 * no * b, P + sP * b);
        orc_initial_guess(&c, x, X + sX * b, U + sU * b);
    }
    orc_rti_solve_batch(&c, batch, x0, P, goal, X, U, u0, cost, status, iters, 2);
    int bad = 0;
    for (int b = 0; b < batch; b++) {
        if (status[b] != 0 && status[b] != 2 && status[b] != 4) bad++;
        /* second step from the shifted iterate, with noise on the obstacles and an explicit slack schedule */
        orc_shift(&c, X + sX * b, U + sU * b);
        double nz[2] = {urand(-2, 2), urand(-2, 2)};
        for (int j = 0; j < no; j++) orc_obstacle_step(&c, obst + 4 * (no * b + j), 0.1, nz, 0.1, 2.0);
        orc_predict_params(&c, obst + 4 * no * b, P + sP * b);
        orc_slack_alpha(&c, x0 + 5 * b, goal + 2 * b, alpha);
        double kkt[4]; int it = 0;
        int st = orc_rti_solve_alpha(&c, x0 + 5 * b, P + sP * b, goal + 2 * b, alpha, X + sX * b, U + sU * b, u0 + 2 * b, cost + b, &it, kkt);
        if (st != 0 && st != 2 && st != 4) bad++;
    }
    /* the batch helpers of bench.py's cpu_baseline (look-ahead of a whole batch; plant step + obstacle step + warm-start shift in place), then one more batched
     * solve from what they left: the long-step trigger and the stationarity residual of the polish are on by default (orc_last_res_g reads what was formed) */
    orc_predict_params_batch(&c, batch, obst, P);
    orc_advance_batch(&c, batch, x0, u0, obst, X, U);
    orc_predict_params_batch(&c, batch, obst, P);
    orc_rti_solve_batch(&c, batch, x0, P, goal, X, U, u0, cost, status, iters, 2);
    for (int b = 0; b < batch; b++) if (status[b] != 0 && status[b] != 2 && status[b] != 4) bad++;
    if (!(orc_last_res_g() >= 0.0) || orc_last_settled_it() < -1) bad++;
    orc_set_investigation(0);
    /* linearisation products and the dense QP of instance 0 */
    {
        double *A = malloc(sizeof(double) * N * 25), *B = malloc(sizeof(double) * N * 10), *bb = malloc(sizeof(double) * N * 5);
        double *q = malloc(sizeof(double) * (N + 1) * 7), *h = malloc(sizeof(double) * (N + 1) * no), *dh = malloc(sizeof(double) * (N + 1) * no * 2);
        orc_linearize(&c, x0, P, goal, X, U, A, B, bb, q, h, dh);
        int nv = 7 * N, nsm = N * no;
        double *H = malloc(sizeof(double) * nv * nv), *g = malloc(sizeof(double) * nv), *Aeq = malloc(sizeof(double) * 5 * N * nv), *beq = malloc(sizeof(double) * 5 * N);
        double *lb = malloc(sizeof(double) * nv), *ub = malloc(sizeof(double) * nv), *Cs = malloc(sizeof(double) * nsm * nv), *hs = malloc(sizeof(double) * nsm);
        double *zs = malloc(sizeof(double) * nsm), *Zs = malloc(sizeof(double) * nsm);
        int ns = orc_export_qp(&c, x0, P, goal, X, U, H, g, Aeq, beq, lb, ub, Cs, hs, zs, Zs);
        if (ns < 0 || ns > nsm) bad++;
        free(A); free(B); free(bb); free(q); free(h); free(dh); free(H); free(g); free(Aeq); free(beq); free(lb); free(ub); free(Cs); free(hs); free(zs); free(Zs);
    }
    free(x0); free(goal); free(obst); free(P); free(X); free(U); free(u0); free(cost); free(alpha); free(status); free(iters);
    return bad;
}

int main(void)
{
    int bad = 0;
    const int cases[][3] = {{20, 3, 24}, {10, 5, 16}, {50, 10, 6}, {2, 3, 8}, {62, 10, 2}, {5, 3, 9}};
    for (unsigned k = 0; k < sizeof(cases) / sizeof(cases[0]); k++) bad += one_config(cases[k][0], cases[k][1], cases[k][2]);
    /* integrator identities and the look-ahead next to the walls */
    double x[5] = {0.3, -0.2, 0.7, 1.5, -0.4}, u[2] = {2.0, -1.0}, xn[5], xc[5], A[25], B[10], A2[25], B2[10];
    orc_dynamics(x, u, 0.1, xn, A, B);
    orc_dynamics_collocation(x, u, 0.1, 3, xc, A2, B2);
    for (int k = 0; k < 5; k++) if (fabs(xn[k] - xc[k]) > 1e-12) bad++;
    orc_config c; orc_default_config(&c, 50, 3, 5.0);
    double st[4] = {7.95, -7.99, 2.0, -2.0}, traj[51 * 2];
    orc_predict_trajectory(&c, st, 50, 0.1, traj);
    printf("oracle sanitizer driver: %d problems\n", bad);
    return bad ? 1 : 0;
}
