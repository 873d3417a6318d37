/* AddressSanitizer / UBSan driver for the HOST side of libmpcgpu.so (tests/test_sanitizers.py builds the library's host code with
 * -fsanitize=address,undefined -fno-gpu-sanitize and runs this on the CPU).  Without a GPU every entry point must fail cleanly:
 * argument validation, the no-device path of mpc_create (no allocation may survive it), NULL handles everywhere. */
#include "mpc_gpu.h"
#include <stdio.h>
#include <string.h>

int main(void)
{
    int bad = 0;
    mpc_config c;
    if (mpc_default_config(&c, 20, 3, 2.0) != MPC_OK) bad++;
    if (mpc_default_config(NULL, 20, 3, 2.0) != MPC_ERR_ARG) bad++;
    mpc_handle *h = NULL;
    const int ndev = mpc_device_count();
    if (ndev == 0) {
        if (mpc_create(&c, 0, 16, &h) != MPC_ERR_NODEVICE || h != NULL) bad++;
        if (!strstr(mpc_last_error(), "no CPU path")) bad++;
    }
    mpc_config c2 = c; c2.N = 63;
    if (mpc_create(&c2, 0, 1, &h) != MPC_ERR_ARG) bad++;
    c2 = c; c2.n_obst = 11;
    if (mpc_create(&c2, 0, 1, &h) != MPC_ERR_ARG) bad++;
    if (mpc_create(&c, 0, 0, &h) != MPC_ERR_ARG) bad++;
    if (mpc_create(NULL, 0, 1, &h) != MPC_ERR_ARG) bad++;
    double buf[8] = {0};
    int32_t ibuf[2] = {0};
    if (mpc_solve(NULL, 1, buf, buf, buf, buf, buf, ibuf, ibuf) != MPC_ERR_ARG) bad++;
    if (mpc_solve_obst(NULL, 1, buf, buf, buf, buf, buf, ibuf, ibuf) != MPC_ERR_ARG) bad++;
    if (mpc_set_slack_schedule(NULL, 1, buf) != MPC_ERR_ARG) bad++;
    if (mpc_set_slack_schedule_dev(NULL, buf) != MPC_ERR_ARG) bad++;
    if (mpc_shift(NULL, 1) != MPC_ERR_ARG) bad++;
    if (mpc_reset_guess(NULL, 1, buf) != MPC_ERR_ARG) bad++;
    if (mpc_plant_step(NULL, 1, buf, buf, buf) != MPC_ERR_ARG) bad++;
    if (mpc_predict(NULL, 1, buf, buf) != MPC_ERR_ARG) bad++;
    if (mpc_debug_adjoint_dev(NULL, 1, 64, 1, buf, buf, buf, buf, NULL) != MPC_ERR_ARG) bad++;
    if (mpc_linearize_dev(NULL, 1, buf, buf, buf, buf, buf, buf, buf, buf, buf, buf, buf, NULL) != MPC_ERR_ARG) bad++;
    if (mpc_profile_enable(NULL, 1) != MPC_ERR_ARG) bad++;
    if (mpc_set_lanes_per_stage(NULL, 0) != MPC_ERR_ARG) bad++;
    if (mpc_set_waves_per_simd(NULL, 2) != MPC_ERR_ARG) bad++;
    if (mpc_set_instance_scheduling(NULL, 1) != MPC_ERR_ARG) bad++;
    if (mpc_get_instance_order(NULL, 1, ibuf) != MPC_ERR_ARG) bad++;
    char name[8];
    if (mpc_get_kernel_name(NULL, 1, 1, name, 8) != MPC_ERR_ARG) bad++;
    if (mpc_destroy(NULL) != MPC_OK) bad++;
    printf("abi sanitizer driver: %d problems (devices: %d)\n", bad, ndev);
    return bad ? 1 : 0;
}
