/*
 * mpc_gpu.h -- C ABI of libmpcgpu.so: batched real-time-iteration NMPC solve on AMD MI355X (gfx950).
 *
 * Drop-in boundary.  In the reference the hot path sits behind acados' ctypes objects
 * `AcadosOcpSolver` / `AcadosSimSolver`, used from src/simulation/robot_ocp_problem.py.  Each entry point
 * below names the reference call site(s) it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain C, no exceptions; every function returns 0 on success, a negative code on error
 *     (MPC_ERR_*), and mpc_last_error() returns a thread-local message for the last failure.
 *   - all arrays are float64, C-contiguous, batch-major:
 *       x0[B][5], goal[B][2], P[B][N+1][n_obst][2], obst[B][n_obst][4] = (x, y, vx, vy),
 *       X[B][N+1][5], U[B][N][2], u0[B][2], cost[B]; status/iters are int32[B].
 *     State is [x, y, psi, v, omega], control is [u_a, u_alpha] (src/models/robot_model.py:14-25).
 *   - per-instance solver status uses the acados codes the reference inspects
 *     (robot_ocp_problem.py:203): 0 ok, 2 QP not converged -- it hit qp_iter_max, or its polish ended with the step estimate still 100 polish_tol
 *     (mpc_config.polish_tol) -- and its step is still applied, 4 QP failure (no step).
 *   - functions without the _dev suffix take HOST pointers and copy; they synchronise before returning.
 *     _dev functions take DEVICE pointers, enqueue on `stream` (a hipStream_t passed as void*, NULL = the
 *     handle's own stream) and do not synchronise.
 *   - a handle is bound to one device and owns the warm-start iterate (X, U) for up to max_batch
 *     instances, as the acados solver object owns it in the reference.  Calls on one handle must be
 *     serialised by the caller; different handles may be used from different threads.
 *   - the library never falls back to a CPU path: without a usable HIP device mpc_create fails.
 */
#ifndef MPC_GPU_H
#define MPC_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_OK 0
#define MPC_ERR_ARG (-1)      /* bad argument / unsupported size */
#define MPC_ERR_HIP (-2)      /* HIP runtime error (message has the hipError string) */
#define MPC_ERR_NODEVICE (-3) /* no usable gfx950 device */

#define MPC_NX 5
#define MPC_NU 2

/* Version of this header's binary interface: bumped whenever `struct mpc_config` changes size or layout or an entry point changes its signature
 * (round 4: trailing field qp_fail_policy, +8 bytes; round 5, version 6: trailing fields polish_ratio, polish_tol, polish_step_frac, +24 bytes, mpc_abi_version itself;
 * round 6, version 7: trailing field polish_res_g, +8 bytes).  A host compares it with mpc_abi_version() of the library it loaded BEFORE it calls
 * mpc_default_config / mpc_create: a host built against an older struct would otherwise be written past its end.  (No reference counterpart: acados
 * regenerates and recompiles its C interface per problem.) */
#define MPC_ABI_VERSION 7

/* Problem definition.  Defaults (mpc_default_config) are the reference's constants. */
typedef struct mpc_config {
    int32_t N;              /* N_SOLV                     src/models/world_specification.py:44   */
    int32_t n_obst;         /* N_OBST, 1..10              world_specification.py:25              */
    double Tf;              /* TF                         world_specification.py:43              */
    double W[6];            /* diag W, y=[x,y,v,w,ua,ual] robot_ocp_problem.py:24-26,78-80       */
    double We[4];           /* diag W_e, y_e=[x,y,v,w]    robot_ocp_problem.py:27,83             */
    double lm;              /* levenberg_marquardt        robot_ocp_problem.py:128               */
    double bx_lo[4];        /* lbx on idx [0,1,3,4]       robot_ocp_problem.py:91-93             */
    double bx_hi[4];
    double bu_lo[2];        /* lbu / ubu                  robot_ocp_problem.py:95-97             */
    double bu_hi[2];
    double r_safe;          /* R_OBST+R_ROBOT+MARGIN      src/models/robot_model.py:62           */
    double slack_a;         /* 1e4                        robot_ocp_problem.py:146               */
    double slack_b;         /* 50                         robot_ocp_problem.py:146               */
    int32_t qp_iter_max;    /* QP_ITER                    robot_ocp_problem.py:131               */
    double qp_tol;          /* interior-point tolerance (linear residuals, complementarity)      */
    /* acados-semantics switches; defaults reproduce 2022-era acados           */
    int32_t cost_scale_dt;  /* stage cost x dt                                      default 1   */
    int32_t slack_scale_dt; /* slack penalties x dt for stages < N                  default 1   */
    int32_t lm_scaled;      /* LM term x dt for stages < N (pinned per seed) default 1 */
    int32_t bx_terminal;    /* path box also at stage N                              default 0   */
    int32_t soft_h;         /* obstacle rows softened (slack=True, :106)             default 1   */
    double arena[4];        /* X_MIN, X_MAX, Y_MIN, Y_MAX  world_specification.py:7-10           */
    int32_t bug_compat_predict; /* look-ahead uses vx = vy, src/utils/visualization.py:69  default 1 */
    double mu0;             /* interior-point cold start: lam = mu0 / t                          */
    double thr0;            /*                            t = max(rho, thr0)                     */
    int32_t qp_fail_policy; /* what a QP that does not converge does (robot_ocp_problem.py:131 qp_solver_iter_max, :203-205 status 4 -> set_initial_guess()):
                               0: the divergence tests are on -- mu > 1e8 mu0 ends the solve at once with status 4 (iterate untouched), and a solve that
                                  reaches qp_iter_max with mu > 1e4 mu0 (from iteration 20 on: above mu0) is a failure (4), not a slow solve (2);
                               1: "truncate" -- no divergence test: the interior point runs to qp_iter_max and its step is applied (status 2), as acados'
                                  SQP_RTI did with a HPIPM solve that returned MAX_ITER; NaN / overflow / step collapse stay status 4.
                               default: what the reference's recorded tables select                                    */
    double polish_ratio;    /* polish of the interior point (round 5): a solve that meets qp_tol takes up to 2 further iterations while (a) its last iteration reduced the largest
                               live complementarity product by less than 1 / polish_ratio (c_max(k) > polish_ratio c_max(k-1): not yet the superlinear end-game), or */
    double polish_tol;      /* (b) for any stage the estimate s r min(1, 10 r) of the remaining primal error exceeds polish_tol (s: max-norm of the stage's last step,
                               r = min(s / previous s, 1/2): a multiplier that collapsed to the floor on a weakly active row leaves the termination test blind but the
                               step long).  0 = that indicator off.  Defaults 1e-2 and 1e-6 (the stated parity tolerance): the solves that met the tolerance 1e-6 .. 2e-5
                               from the QP's exact solution (0.7 % of the first solves of BASELINE configs[4]'s problem) are gone at +0.5 % iterations (DESIGN.md
                               section 2).  (The qp_solver tolerances of robot_ocp_problem.py:126-132 are left at acados' defaults there.) */
    double polish_step_frac; /* floor of that estimate as a fraction of the step: est = max(s r min(1, 10 r), polish_step_frac s).  Default 0.01 from N = 30 on, else 0: at long
                               horizons the last Newton step leaves 1.5 .. 10 % of itself behind whatever contraction was observed (5 solves of 1.2e7 fuzz solves ended
                               1e-5 .. 2e-5 from the exact solution without it, none beyond 3e-7 with it; +0.4 .. 0.8 % iterations there, +3.8 % at N = 20 where nothing needs it) */
    double polish_res_g;    /* (c), round 6: the stationarity residual of the QP's Lagrangian -- HPIPM's res_g, which acados' default tolerances gate
                               (robot_ocp_problem.py:126-132 leaves them alone) -- above this value when the termination test holds asks for a polish iteration as well.
                               Formed once per solve by one open-loop adjoint sweep (input blocks B_i' pi_{i+1} + (H z + q - C' lam)_u; slack equations per row).
                               Default 1e-7: on BASELINE configs[4]'s problem the solves beyond 1e-7 from the QP's exact solution fall from 5 to 1 of 4000 and the worst
                               from 1.3e-6 to 2.4e-7 at +0.11 % iterations.  0 = off. */
} mpc_config;

typedef struct mpc_handle mpc_handle;

/* MPC_ABI_VERSION the library was built with */
int mpc_abi_version(void);

/* thread-local description of the last error returned on this thread */
const char *mpc_last_error(void);

/* number of visible HIP devices (0 when none); never initialises a context */
int mpc_device_count(void);

/* fill `cfg` with the reference's constants for a given horizon / obstacle count */
int mpc_default_config(mpc_config *cfg, int N, int n_obst, double Tf);

/* Replaces AcadosOcpSolver(...) / AcadosSimSolver(...) construction, robot_ocp_problem.py:135-136.
 * Allocates device buffers for up to max_batch instances on `device`; warm start is zero-initialised. */
int mpc_create(const mpc_config *cfg, int device, int max_batch, mpc_handle **out);
int mpc_destroy(mpc_handle *h);

/* Device pointers of the handle-owned iterate (X[max_batch][N+1][5], U[max_batch][N][2]) and the handle's stream */
int mpc_iterate_ptrs(mpc_handle *h, double **d_X, double **d_U, void **stream);

/* ocp_solver.set(i,'x',..)/set(i,'u',..) for all stages, robot_ocp_problem.py:254-258,305-306 */
int mpc_set_warmstart(mpc_handle *h, int batch, const double *X, const double *U);
/* ocp_solver.get(i,'x') / get(i,'u') for all stages, robot_ocp_problem.py:198,232,240 */
int mpc_get_traj(mpc_handle *h, int batch, double *X, double *U);
/* set_initial_guess(): reset(), X[i] = [x0_x, x0_y, x0_psi, 0, 0], U = 0, robot_ocp_problem.py:286-306 */
int mpc_reset_guess(mpc_handle *h, int batch, const double *x0);
/* set_initial_guess() as the commented block robot_ocp_problem.py:293-300 computes it -- the variant that recorded the two `interpolate_init`
 * tables of src/simulation/test_data (20221031_225145, _225445): stage i starts at (x0_x, x0_y + i/N (goal_y - x0_y), arctan2(goal_y - x0_y, 0), 0, 0),
 * U = 0; the reference's slips (`x0[0] - x0[0]`, `subgoal[0] - subgoal[0]`) included, bit for bit numpy's arithmetic. */
int mpc_reset_guess_interp(mpc_handle *h, int batch, const double *x0, const double *goal);
/* warm-start shift, robot_ocp_problem.py:253-258 */
int mpc_shift(mpc_handle *h, int batch);

/* The solve core, robot_ocp_problem.py:186-198: parameterize_model (P), parameterize_slack (from x0, goal),
 * lbx_0 = ubx_0 = x0, ocp_solver.solve(), u* = get(0,'u').  One SQP_RTI iteration per instance on the
 * handle-owned iterate.  u0/cost/status/iters may be NULL.  cost = NLP objective at the new iterate. */
int mpc_solve(mpc_handle *h, int batch, const double *x0, const double *P, const double *goal,
              double *u0, double *cost, int32_t *status, int32_t *iters);
/* Same with the obstacle look-ahead fused: obst[B][n_obst][4] -> P on device
 * (Obstacle.predict_trajectory, src/utils/visualization.py:62-79 + parameterize_model, :154-166) */
int mpc_solve_obst(mpc_handle *h, int batch, const double *x0, const double *obst, const double *goal,
                   double *u0, double *cost, int32_t *status, int32_t *iters);

/* parameterize_slack(), robot_ocp_problem.py:145-152: the reference uploads zl_i = Zl_i = alpha_i * ones(n_obst) for every stage i
 * before every solve (cost_set(i,'zl'|'Zl')).  By default the solve kernel evaluates that schedule itself from (x0, goal, slack_a,
 * slack_b); a caller who changes parameterize_slack passes the weights here instead: alpha[batch][N+1] (host; copied) applies to all
 * following solves of the first `batch` instances until it is replaced; alpha = NULL returns to the built-in schedule.  Weights must be
 * finite and >= 0 (0 = the row is vacuous, as at the reference's terminal stage).  slack_scale_dt still multiplies stages < N.
 * _dev: a device array [max_batch][N+1] used in place (not copied; NULL = built-in). */
int mpc_set_slack_schedule(mpc_handle *h, int batch, const double *alpha);
int mpc_set_slack_schedule_dev(mpc_handle *h, const double *d_alpha);

/* Plant integrator, ocp_integrator.set/solve/get, robot_ocp_problem.py:207-212 (same IRK as the OCP) */
int mpc_plant_step(mpc_handle *h, int batch, const double *x, const double *u, double *x_next);
/* Obstacle look-ahead only: obst[B][n_obst][4] -> P[B][N+1][n_obst][2] (visualization.py:62-79) */
int mpc_predict(mpc_handle *h, int batch, const double *obst, double *P);

/* ---- device-pointer (asynchronous) variants: inputs already resident in HBM ---- */
int mpc_solve_dev(mpc_handle *h, int batch, const double *d_x0, const double *d_P, const double *d_goal,
                  double *d_X, double *d_U, double *d_u0, double *d_cost, int32_t *d_status, int32_t *d_iters,
                  void *stream);
int mpc_predict_dev(mpc_handle *h, int batch, const double *d_obst, double *d_P, void *stream);

/* One whole control step of RobotOcpProblem.step (robot_ocp_problem.py:184-260) in ONE kernel launch, batched:
 *   look-ahead of the obstacles (visualization.py:62-79) -> RTI solve (:186-198) -> u* -> optional set_initial_guess on
 *   status 4 (:203-205) -> plant step, x0 updated in place (:207-212) -> obstacle motion with optional noise, obst updated
 *   in place (:217-218) -> margin / arena / goal bookkeeping (:213-250) -> warm-start shift (:253-258).
 * flags: OR of MPC_STEP_*.  d_noise: [B][n_obst][2] standard normals or NULL.  Metrics buffers (MPC_STEP_METRICS):
 * min_margin[B] (initialise to +inf), ep_flags int32[B] (bit0 goal reached -> the instance idles from then on, bit1 left the
 * arena, bit2 min_margin <= 0), ep_steps int32[B] (the reference's iteration counter i). */
#define MPC_STEP_SHIFT 1
#define MPC_STEP_PLANT 2
#define MPC_STEP_OBSTACLES 4
#define MPC_STEP_RESET_ON_FAIL 8
#define MPC_STEP_ALIAS_BUG 16
#define MPC_STEP_METRICS 32
#define MPC_STEP_INTERP_GUESS 64   /* with MPC_STEP_RESET_ON_FAIL: the reset writes the straight-line guess of mpc_reset_guess_interp */
int mpc_closed_loop_step_dev(mpc_handle *h, int batch, double *d_x0, double *d_obst, const double *d_goal, double *d_X, double *d_U,
                             double *d_u0, double *d_cost, int32_t *d_status, int32_t *d_iters, const double *d_noise,
                             double randomness, double vmax, int flags, double *d_min_margin, int32_t *d_ep_flags,
                             int32_t *d_ep_steps, void *stream);
int mpc_shift_dev(mpc_handle *h, int batch, double *d_X, double *d_U, void *stream);
int mpc_reset_guess_dev(mpc_handle *h, int batch, const double *d_x0, double *d_X, double *d_U, void *stream);
int mpc_reset_guess_interp_dev(mpc_handle *h, int batch, const double *d_x0, const double *d_goal, double *d_X, double *d_U, void *stream);
int mpc_plant_step_dev(mpc_handle *h, int batch, const double *d_x, const double *d_u, double *d_xnext, void *stream);
/* Obstacle.step() ground-truth motion (visualization.py:20-33); d_noise[B*n_obst][2] standard normals or NULL */
int mpc_obstacle_step_dev(mpc_handle *h, int count, double *d_obst, const double *d_noise,
                          double randomness, double vmax, void *stream);
/* The reference's NOISE stream on the device (experiments.py:33-36, visualization.py:28-33): instance s carries numpy's legacy generator after
 * np.random.seed(seed0 + s) and the scenario generator's uniform draws (mpc_generate_scenarios_dev produces those values; here they are consumed), and
 * mpc_noise_draw_dev writes one control step's np.random.normal(size=2) per obstacle -- d_noise[count][n_obst][2], the array mpc_closed_loop_step_dev
 * takes -- and advances the state.  d_state: count * mpc_noise_state_words() uint32.  d_ep_flags (optional): instances whose episode is over (bit 0)
 * draw nothing, like the reference's loop that has left.  Bit for bit numpy's stream except for the last bit of ~1 draw in 10^4 (glibc's log is not
 * correctly rounded there; the device evaluates it in double-double arithmetic).  No host upload: 13000 episodes x 400 steps of normals are 416 MB. */
int mpc_noise_state_words(void);
int mpc_noise_init_dev(mpc_handle *h, int count, int scenario, unsigned seed0, uint32_t *d_state, void *stream);
int mpc_noise_draw_dev(mpc_handle *h, int count, uint32_t *d_state, double *d_noise, const int32_t *d_ep_flags, void *stream);
/* MULTI-GPU: one process per GPU, every rank solves its own contiguous slice of the scenarios (the reference's 13 000 closed
 * loops, experiments.py:20-36, are independent), and the only exchange is an all-gather of the per-instance costs -- RCCL over xGMI, called directly
 * from this library (librccl.so.1 is loaded on first use; there is no link-time dependency and no other transport).  A C host does:
 *   rank 0: mpc_comm_unique_id(id), ships the 128 bytes to the other ranks by whatever means it has (a file, a socket, MPI);
 *   every rank: mpc_comm_init(h, rank, world, id)  [collective];  per round: mpc_allgather_cost_dev(h, count, d_cost, d_all, stream)  [collective,
 *   d_all[world][count], rank-major, equal counts on all ranks; enqueued on `stream` (NULL: the handle's), so it overlaps whatever runs on other streams];
 *   mpc_comm_destroy(h) (mpc_destroy does it too).  mpc_allgather_cost is the host-pointer form (stages through the handle's stream and waits). */
#define MPC_COMM_ID_BYTES 128
int mpc_comm_unique_id(unsigned char *id /* MPC_COMM_ID_BYTES */);
int mpc_comm_init(mpc_handle *h, int rank, int world, const unsigned char *id /* MPC_COMM_ID_BYTES */);
int mpc_comm_world(const mpc_handle *h);      /* ranks of the handle's communicator, 0 without one */
/* file name of the RCCL library the exchange is bound to (dladdr of its ncclAllGather): a process that has PyTorch's ROCm wheel loaded gets the wheel's
 * bundled librccl (same soname, already mapped), any other host /opt/rocm's -- measurement records name which one ran */
int mpc_comm_library_path(char *buf, int len);
int mpc_allgather_cost_dev(mpc_handle *h, int count, const double *d_cost, double *d_cost_all, void *stream);
int mpc_allgather_cost(mpc_handle *h, int count, const double *cost, double *cost_all);
int mpc_comm_destroy(mpc_handle *h);
/* generate_random_moving_obstacles (src/utils/obstacle_generator.py:8-28) for the seeds seed0 .. seed0+count-1: instance s gets
 * bit for bit what the reference draws after np.random.seed(seed0 + s) (numpy legacy MT19937 stream, reference draw order).
 * scenario: 0 RANDOM, 1 CENTER, 2 EDGE (:10-18).  box = {X_MIN_OBST, X_MAX_OBST, Y_MIN_OBST, Y_MAX_OBST, V_MAX_OBST, edge (7)}
 * (src/models/world_specification.py:25-40).  obst[count][n_obst][4] = (x, y, vx, vy). */
int mpc_generate_scenarios_dev(mpc_handle *h, int count, int scenario, unsigned seed0, const double *box, double *d_obst, void *stream);
int mpc_generate_scenarios(mpc_handle *h, int count, int scenario, unsigned seed0, const double *box, double *obst);
/* Linearisation products of the current iterate, for parity tests of the linearise stage:
 * A[B][N][5][5], Bm[B][N][5][2], b[B][N][5], q[B][N+1][7] (order u,x), hval[B][N+1][n_obst], dh[B][N+1][n_obst][2] */
int mpc_linearize_dev(mpc_handle *h, int batch, const double *d_x0, const double *d_P, const double *d_goal,
                      const double *d_X, const double *d_U,
                      double *d_A, double *d_B, double *d_b, double *d_q, double *d_hval, double *d_dh, void *stream);

/* The stationarity sweep of the polish (mpc_config.polish_res_g) on its own, for parity tests of that stage: the open-loop adjoint of a given per-stage gradient
 * g[B][N+1][7] (order u, x) over the linearisation of the iterate (X, U), in the lane layout of a solve kernel -- lanes_per_stage = 1 with lanes_per_instance 16, 21,
 * 32 or 64 (N + 1 lanes must fit), or lanes_per_stage 2 (N <= 31) / 3 (N <= 20) with one instance per wavefront.  ru[B][N] = max-norm of the input block
 * g_u,i + B_i' pi_{i+1} per stage, pi_i = g_x,i + A_i' pi_{i+1}.  (No reference counterpart: HPIPM forms res_g inside acados, robot_ocp_problem.py:126-132,195.) */
int mpc_debug_adjoint_dev(mpc_handle *h, int batch, int lanes_per_instance, int lanes_per_stage, const double *d_X, const double *d_U, const double *d_g,
                          double *d_ru, void *stream);

/* ---- measurement: HIP events around solve-kernel launches on the launch stream ----
 * on = 0: off; on = k > 0: events around every k-th launch (k = 1: every launch).  A pair of event records between two back-to-back
 * launches costs the stream ~7 us (measured, scripts/gap_probe.py), so a throughput run samples (bench.py: every 7th launch). */
int mpc_profile_enable(mpc_handle *h, int on);
/* synchronises; returns the summed duration and the number of solve-kernel launches since the last call */
int mpc_profile_read(mpc_handle *h, double *sum_ms, int *launches);

/* Optional device accumulators int32[max_batch] (NULL = off): every solve launch adds each instance's interior-point
 * iteration count to iters_acc and (status == 4) + 65536 * (status == 2) to status_acc, so a benchmark loop needs no
 * extra kernels to report mean iterations, QP failures and iteration-cap hits. */
int mpc_set_accumulators(mpc_handle *h, int32_t *d_iters_acc, int32_t *d_status_acc);

/* Debug aid for parity work: when enabled, every solve records (mu, sigma, alpha, cmax) of each interior-point
 * iteration into a device buffer [max_batch][qp_iter_max][4]; host_out (may be NULL) receives the first `batch` rows. */
int mpc_debug_trace(mpc_handle *h, int enable, int batch, double *host_out);

/* lanes per instance (64, 32, 16 or 8) the dispatcher picked for `batch`; 0 = automatic (default) */
int mpc_set_lanes_per_instance(mpc_handle *h, int lanes);
int mpc_get_lanes_per_instance(mpc_handle *h, int batch);
/* 1: for batches <= 1024 (one instance per wavefront) the Riccati factorisation runs on the matrix cores (v_mfma_f64_16x16x4,
 * homogeneous 8x8 stage blocks).  0 (default).  Measured on MI355X it is slower than both vector-ALU variants (FP64 MFMA rate =
 * FP64 vector rate, 116-cycle dependent MFMA links, 75 % tile padding) and less accurate on ill-conditioned stages; it is kept
 * as evidence, see DESIGN.md section 4.  No reference counterpart (tuning / test hook). */
int mpc_set_matrix_cores(mpc_handle *h, int on);
/* Riccati factorisation sweep of the interior point (same arithmetic specification, different lane mapping).
 * 1 (default): row-parallel -- the 8 columns of a stage's homogeneous blocks sit in 8 lanes of a 16-lane DPP row and the
 * products run as v_fmac_f64_dpp row_newbcast chains (~135 instead of ~330 wave instructions per stage); the forward and
 * adjoint vector recursions run the same way on the closed-loop matrix.
 * 0: one-lane systolic sweeps (no LDS), the independent implementation the default is tested against.
 * No reference counterpart (tuning / test hook). */
int mpc_set_row_parallel(mpc_handle *h, int on);
/* Block-2 (partially condensed) stage recursions, default OFF (opt-in: measured 5 % slower than one stage per step at C2): where the mapping has them (the stage-split kernel on dense blocks, even horizons, all of
 * the kernel's obstacle rows in use) the Riccati factorisation and the three vector recursions of an interior-point iteration run over PAIRS of stages
 * (state x_2m, inputs (u_2m, u_2m+1); x_2m+1 eliminated) -- half the sequential steps at ~0.75x the instructions.  What HPIPM's PARTIAL_CONDENSING
 * (robot_ocp_problem.py:126) does on the CPU.  0: one stage per step (the default, and the form the block form is tested against).  Same interior point, same QP solution. */
int mpc_set_block_riccati(mpc_handle *h, int on);
/* Lanes per horizon stage.  0 (default): automatic -- a batch of at most eight instances per SIMD of the device (8192 on
 * MI355X) runs one instance per wavefront with the inequality rows of every stage dealt out to 3 (N <= 20) or 2 (N <= 31)
 * neighbouring lanes, which shortens the instruction stream such a latency-bound wavefront is limited by (and lets every
 * instance stop at its own iteration); larger batches keep one lane per stage and pack 64/G instances into a wavefront
 * (measured crossover between 8192 and 16384 instances).  1: always one lane per stage.  2 / 3: the split mapping whenever
 * the horizon fits (any batch).  Setting lanes per INSTANCE, matrix cores or the systolic sweep implies 1.
 * Same arithmetic specification either way.  No reference counterpart (tuning / test hook). */
int mpc_set_lanes_per_stage(mpc_handle *h, int lanes);
int mpc_get_lanes_per_stage(mpc_handle *h, int batch);
/* Instance scheduling (default on).  Where several instances share a wavefront (one lane per stage: 2, 3 or 4 of them) the wavefront runs
 * until its slowest instance has converged; iteration counts are heavy-tailed (C3: mean 6.6, mean of the per-wavefront maximum 9.4 with
 * three per wavefront) but strongly correlated between consecutive control steps of an instance.  The library therefore deals the instances
 * to wavefronts in the order of their iteration counts in this handle's previous launch of the same batch size (a stable counting sort on
 * the device behind every launch; 7.4 instead of 9.4).  Which instances share a wavefront has no effect on their results beyond the
 * rounding of the wavefront sums with three instances per wavefront (bit-identical with two or four).  No reference counterpart.
 * mpc_get_instance_order: the permutation in effect for the next launch of `batch` instances (returns 1) or 0 when it is the natural order. */
int mpc_set_instance_scheduling(mpc_handle *h, int on);
int mpc_get_instance_order(mpc_handle *h, int batch, int32_t *order);
/* name of the solve kernel instantiation a batch of this size runs (as rocprofv3 prints it, without the namespace), for measurement
 * records: "rti_split_kernel<row capacity, lanes per stage, two wavefronts per SIMD, masked, block-2 recursions>" or "rti_solve_kernel<row capacity, lanes per
 * instance, sweeps, masked>" (row capacity: 3, 5 or 10 obstacle row pairs per stage, the smallest that holds n_obst; masked: n_obst is below it;
 * sweeps: 0 systolic, 1 matrix cores, 2 row-parallel on dense LDS blocks, 3 row-parallel on compact LDS blocks).  lookahead: whether the
 * obstacle look-ahead runs inside the kernel (mpc_closed_loop_step_dev) -- it enters the LDS budget that selects the block layout. */
int mpc_get_kernel_name(mpc_handle *h, int batch, int lookahead, char *buf, int len);
/* Wavefronts per SIMD of the stage-split mapping.  0 (default): automatic -- two (256 registers per lane, compact LDS blocks: 14.7 KB per
 * wavefront at N = 20, eight wavefronts per CU) only for 3-obstacle problems in batches of more than four instances per SIMD of the device,
 * where a second resident wavefront fills the LDS and dependent-issue stalls of the first; one (all 512 registers, dense LDS blocks)
 * otherwise -- with 5 or 10 obstacle row pairs the 256-register build spills and loses at every batch, and a problem that uses fewer rows than
 * its kernel's capacity always runs one.  1 / 2: forced (ignored for such partial-row problems).  The one-lane-per-stage mapping always runs one
 * wavefront per SIMD.  Same arithmetic specification either way.  No reference counterpart (tuning / test hook). */
int mpc_set_waves_per_simd(mpc_handle *h, int waves);
int mpc_get_waves_per_simd(mpc_handle *h, int batch);

#ifdef __cplusplus
}
#endif
#endif /* MPC_GPU_H */
