/*
 * mpc_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C float64 restatement of the reference's per-control-step MPC solve
 * ("linearise horizon -> build QP -> solve QP -> full step", i.e. one acados
 * SQP_RTI iteration) for abdelhakim96/Dynamic-Obstacle-Avoidance-MPC.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this library.  The product (libmpcgpu.so) never links or loads it.
 *
 * HOW THE ORACLE IS PINNED.  The reference's arithmetic lives in acados / HPIPM / BLASFEO / CasADi (un-vendored, un-pinned, absent
 * from /root/reference and from this image), and the reference holds no unit tests or golden vectors for (X+, U+, u*).  What it does
 * hold are the closed-loop tables it RECORDED (src/simulation/test_data/20221031_*_experiment_data.csv: 100 seeds x
 * [hit, reached, min_margin, dist_to_goal, iters, out_of_bounds], protocol src/simulation/experiments.py:20-36).  Driven with the
 * reference's own numpy random streams per seed, this oracle's closed loop (tests/helpers.py::OracleLoop) reproduces 342 of the 800
 * recorded rows with the control-step count exact and min_margin / dist_to_goal to 1e-3 (243 to 1e-6, median deviation per table 2e-9 .. 3e-7):
 * every row on which acados' QP converged throughout.  The remaining rows contain a QP that hit HPIPM's iteration cap or failed,
 * where the recorded tables themselves disagree between caps.  That pins cost scaling, the levenberg_marquardt term (scaled by the
 * stage interval), the slack schedule, the integrator, the status-4 reset and the aliasing defect D2 -- any other setting of the
 * unverifiable acados-semantics switches reproduces NO row (tests/test_oracle_golden.py, profiles/r02_oracle_seed_replay.json).
 * Single-solve outputs of a NON-converged QP (acados status 2 / 4) remain unpinned: nothing reference-held records them.
 * Further pins: obstacle predictor / scenario generator / constants bit for bit against vectors captured by importing the
 * reference's importable modules (tests/golden/), and solver-independent checks (GL4 collocation identity, finite differences,
 * scipy on the assembled QP, explicit KKT residuals).
 *
 * Reference lines each function follows are cited at its definition in
 * mpc_oracle.c (paths relative to /root/reference).
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NX 5
#define ORC_NU 2
#define ORC_NZ 7

typedef struct orc_config {
    int N;              /* horizon intervals: N_SOLV, world_specification.py:44            */
    int n_obst;         /* N_OBST, world_specification.py:25                               */
    double Tf;          /* TF, world_specification.py:43 (dt = Tf/N)                       */
    double W[6];        /* diag(W) for y=[x,y,v,w,ua,ualpha], robot_ocp_problem.py:24-26,78-80 */
    double We[4];       /* diag(W_e) for y_e=[x,y,v,w], robot_ocp_problem.py:27,83         */
    double lm;          /* levenberg_marquardt, robot_ocp_problem.py:128                   */
    double bx_lo[4];    /* lbx on idx [0,1,3,4], robot_ocp_problem.py:91-93                */
    double bx_hi[4];
    double bu_lo[2];    /* lbu/ubu, robot_ocp_problem.py:95-97                             */
    double bu_hi[2];
    double r_safe;      /* R_OBST+R_ROBOT+MARGIN, robot_model.py:62                        */
    double slack_a;     /* 1e4, robot_ocp_problem.py:146                                   */
    double slack_b;     /* 50,  robot_ocp_problem.py:146                                   */
    int qp_iter_max;    /* QP_ITER, robot_ocp_problem.py:131                               */
    double qp_tol;      /* IPM tolerance (residuals and complementarity)                   */
    /* acados-semantics switches, SURVEY.md 8(c) (defaults = 2022-era acados)            */
    int cost_scale_dt;  /* stage cost multiplied by dt                                     */
    int slack_scale_dt; /* slack penalties z,Z multiplied by dt for stages < N             */
    int lm_scaled;      /* LM term multiplied by dt for stages < N; default 1 */
    int bx_terminal;    /* path box also on stage N; default 0                             */
    int soft_h;         /* obstacle rows softened (slack=True), robot_ocp_problem.py:106   */
    /* obstacle world, world_specification.py:7-10 and visualization.py:62-79            */
    double arena[4];    /* X_MIN, X_MAX, Y_MIN, Y_MAX                                      */
    int bug_compat_predict; /* predict_trajectory uses vx = self.vy, visualization.py:69   */
    /* interior-point start (cold start every call, as HPIPM with warm_start=0)          */
    double mu0;
    double thr0;
    /* What a QP that does not converge does (round 4; shared with the HIP kernels, mpc_config.qp_fail_policy):
     *   0  the divergence tests are on: mu > 1e8 mu0 ends the solve at once (status 4, iterate untouched), and a solve that reaches qp_iter_max
     *      with mu > 1e4 mu0 -- or, from iteration 20 on, above mu0 -- is a failure (4), not a slow solve (2);
     *   1  "truncate": no divergence test -- the interior point runs to qp_iter_max and its step is applied (status 2), as acados' SQP_RTI did with a
     *      HPIPM solve that returned MAX_ITER (robot_ocp_problem.py:131, :203-205 only reacts to status 4); NaN / overflow / step collapse stay 4.
     * The default is the one the reference's recorded tables select (profiles/r04_fail_policy_replay.json). */
    int qp_fail_policy;
    /* polish (round 5; shared with the HIP kernels, mpc_config.polish_ratio): once the termination test holds, up to 2 further iterations while the last
     * iteration reduced the largest live complementarity product by less than 1 / polish_ratio (c_max(k) > polish_ratio c_max(k-1): not yet the superlinear
     * end-game), or (polish_tol) while for any stage the estimate s r min(1, 10 r) of the remaining primal error exceeds polish_tol (s: max-norm of the stage's
     * last step, r = min(s / previous s, 1/2)); 0 = that indicator off.  Defaults 1e-2 and 1e-6 (the stated parity tolerance). */
    double polish_ratio, polish_tol;
    double polish_step_frac;  /* floor of that estimate as a fraction of s: default 0.01 from N = 30 on, else 0 (orc_default_config says why) */
    /* polish indicator (c) (round 6): the STATIONARITY residual of the QP's Lagrangian (HPIPM's res_g; with the adjoint costates its state blocks vanish, so it is
     * the input blocks B_i' pi_{i+1} + (H z + q - C' lam)_u and the slack equations Z s + z - lam_1 - lam_2) above this value when the termination test holds
     * asks for a polish iteration; 0 = off. */
    double polish_res_g;
} orc_config;

void orc_default_config(orc_config *c, int N, int n_obst, double Tf);

/* a1/a14: unicycle step, closed form of IRK Gauss-Legendre(4 stages, 1 step). A 5x5, B 5x2 row-major, may be NULL */
void orc_dynamics(const double *x, const double *u, double dt, double *xn, double *A, double *B);
/* general GL4 collocation with Newton on the stage equations + IFT sensitivities (identity check only) */
void orc_dynamics_collocation(const double *x, const double *u, double dt, int newton_iter,
                              double *xn, double *A, double *B);
/* robot_model.py:39-43 */
void orc_ode(const double *x, const double *u, double *xdot);

/* a9: visualization.py:25-60 (noise = NULL -> deterministic; else 2 standard normals, randomness, vmax) */
void orc_obstacle_step(const orc_config *c, double *state /*x,y,vx,vy*/, double dt,
                       const double *noise, double randomness, double vmax);
/* a9: visualization.py:62-79 -> traj[(n+1)*2] */
void orc_predict_trajectory(const orc_config *c, const double *state, int n, double dt, double *traj);
/* a8: P[(N+1)*n_obst*2] from obst[n_obst*4] */
void orc_predict_params(const orc_config *c, const double *obst, double *P);

/* a7: robot_ocp_problem.py:145-152 */
void orc_slack_alpha(const orc_config *c, const double *x0, const double *goal, double *alpha);

/* a13 / a12 */
void orc_initial_guess(const orc_config *c, const double *x0, double *X, double *U);
void orc_initial_guess_interp(const orc_config *c, const double *x0, const double *goal, double *X, double *U);   /* :293-300 (commented variant) */
void orc_shift(const orc_config *c, double *X, double *U);

/* linearisation products, for parity tests of the linearise stage:
 * A[N*25], B[N*10], b[N*5], q[(N+1)*7] (order u,x; last stage x only in slots 2..6),
 * h[(N+1)*n_obst], dh[(N+1)*n_obst*2] */
void orc_linearize(const orc_config *c, const double *x0, const double *P, const double *goal,
                   const double *X, const double *U,
                   double *A, double *B, double *b, double *q, double *h, double *dh);

/* NLP objective at (X,U): LS cost + exact penalty of obstacle violation (north_star "per-scenario cost") */
double orc_cost(const orc_config *c, const double *x0, const double *P, const double *goal,
                const double *X, const double *U);

/* a10/a11: one RTI step.  X,U updated in place.  status: 0 ok, 2 max-iter (step applied), 4 QP failure (no step).
 * kkt[4] (optional): final inf-norm residuals {stationarity, equality, inequality, complementarity}. */
int orc_rti_solve(const orc_config *c, const double *x0, const double *P, const double *goal,
                  double *X, double *U, double *u0, double *cost, int *iters, double *kkt);

/* the same with explicit per-stage slack weights alpha[N+1] (zl_i = Zl_i = alpha_i, what the reference uploads with
 * cost_set(i,'zl'/'Zl'), robot_ocp_problem.py:149-152); alpha = NULL is orc_rti_solve */
int orc_rti_solve_alpha(const orc_config *c, const double *x0, const double *P, const double *goal, const double *alpha,
                        double *X, double *U, double *u0, double *cost, int *iters, double *kkt);

/* batched driver (OpenMP over instances); arrays are [batch][...] contiguous. */
void orc_rti_solve_batch(const orc_config *c, int batch, const double *x0, const double *P, const double *goal,
                         double *X, double *U, double *u0, double *cost, int *status, int *iters,
                         int nthreads);

/* bench.py's cpu_baseline: the untimed surroundings of the solve for a whole batch per call (look-ahead; plant step + obstacle step + warm-start shift, in place) */
void orc_predict_params_batch(const orc_config *c, int batch, const double *obst, double *P);
void orc_advance_batch(const orc_config *c, int batch, double *x, const double *u0, double *obst, double *X, double *U);

/* Assembled QP of one RTI step in dense form for cross-checking with scipy.
 * Variables v = [du_0, dx_1, du_1, ..., dx_N] (dx_0 eliminated), nv = 7N.
 *   min 0.5 v'Hv + g'v + sum_j (zs_j s_j + 0.5 Zs_j s_j^2)
 *   s.t. Aeq v = beq;  lb <= v <= ub;  Cs v + hs + s >= 0, s >= 0   (ns soft rows)
 * Returns ns. Arrays sized by caller: H[nv*nv], g[nv], Aeq[(5N)*nv], beq[5N], lb[nv], ub[nv],
 * Cs[ns_max*nv], hs[ns_max], zs[ns_max], Zs[ns_max] with ns_max = N*n_obst. */
int orc_export_qp(const orc_config *c, const double *x0, const double *P, const double *goal,
                  const double *X, const double *U,
                  double *H, double *g, double *Aeq, double *beq, double *lb, double *ub,
                  double *Cs, double *hs, double *zs, double *Zs);

/* debugging aid: number of pairs of the calling thread's last solve that ended with BOTH t and lam at the floor ("dead": complementary whatever the row does) */
int orc_last_dead_pairs(void);
/* ... polish iterations it took, and the largest per-stage norm of its last primal step (the polish's indicator (b) looks at these norms) */
int orc_last_npolish(void);
/* ... first iteration >= 20 at whose head mu stood above mu0 (-1: none): the settled-mu failure rule's trigger (ipm_solve) */
int orc_last_settled_it(void);
/* ... the reduced stationarity residual the polish's indicator (c) last formed on this thread (0: never formed) */
double orc_last_res_g(void);
double orc_last_step_norm(void);
/* investigation switches of scripts/converged_unmatched.py (mpc_oracle.c says which); process-wide, default 0, never set by tests */
void orc_set_investigation(int switches);
/* debugging aid: record (mu, sigma, alpha, cmax) of every IPM iteration of subsequent single solves into buf[4*cap] */
void orc_set_trace(double *buf, int cap);

#ifdef __cplusplus
}
#endif
#endif
