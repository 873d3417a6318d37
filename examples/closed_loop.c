/* Closed loop through the C ABI alone (include/mpc_gpu.h): what a host that is not Python links against.
 * The loop body is RobotOcpProblem.step's, src/simulation/robot_ocp_problem.py:184-260 of the reference, for a small batch of scenarios with
 * parked obstacles: look-ahead + RTI solve (mpc_solve_obst), plant step (mpc_plant_step), warm-start shift (mpc_shift).
 *   gcc -O2 -Iinclude examples/closed_loop.c -o closed_loop -L<dir of libmpcgpu.so> -lmpcgpu -Wl,-rpath,<dir> -lm
 *   ./closed_loop [steps]          prints one line per instance: final state, accumulated cost, failed solves */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mpc_gpu.h"

#define B 4
#define N_OBST 3

int main(int argc, char **argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 40;
    mpc_config cfg;
    mpc_handle *h = NULL;
    if (mpc_abi_version() != MPC_ABI_VERSION) { fprintf(stderr, "libmpcgpu has ABI version %d, this host was built against %d\n", mpc_abi_version(), MPC_ABI_VERSION); return 1; }
    if (mpc_default_config(&cfg, 20, N_OBST, 2.0) || mpc_create(&cfg, 0, B, &h)) { fprintf(stderr, "mpc_create: %s\n", mpc_last_error()); return 1; }
    double x[B][5], goal[B][2], obst[B][N_OBST][4], u0[B][2], cost[B], xn[B][5], total[B] = {0};
    int32_t status[B], iters[B], failed[B] = {0};
    for (int b = 0; b < B; b++) {
        x[b][0] = -6.0 + b; x[b][1] = -6.0; x[b][2] = 0.7853981633974483; x[b][3] = x[b][4] = 0.0;
        goal[b][0] = 5.0 - b; goal[b][1] = 5.0;
        for (int j = 0; j < N_OBST; j++) { obst[b][j][0] = -2.0 + 2.5 * j; obst[b][j][1] = -1.5 + 1.5 * j + 0.3 * b; obst[b][j][2] = obst[b][j][3] = 0.0; }
    }
    if (mpc_reset_guess(h, B, &x[0][0])) { fprintf(stderr, "%s\n", mpc_last_error()); return 1; }          /* set_initial_guess(), :286-306 */
    for (int k = 0; k < steps; k++) {
        if (mpc_solve_obst(h, B, &x[0][0], &obst[0][0][0], &goal[0][0], &u0[0][0], cost, status, iters) ||   /* :186-198 */
            mpc_plant_step(h, B, &x[0][0], &u0[0][0], &xn[0][0]) ||                                         /* :207-212 */
            mpc_shift(h, B)) { fprintf(stderr, "step %d: %s\n", k, mpc_last_error()); return 1; }            /* :253-258 */
        for (int b = 0; b < B; b++) {
            for (int c = 0; c < 5; c++) x[b][c] = xn[b][c];
            total[b] += cost[b]; failed[b] += status[b] == 4;
        }
    }
    for (int b = 0; b < B; b++)
        printf("%d %.17g %.17g %.17g %.17g %.17g %.17g %d\n", b, x[b][0], x[b][1], x[b][2], x[b][3], x[b][4], total[b], failed[b]);
    mpc_destroy(h);
    return 0;
}
