// Micro-test: LDS bank conflicts of the sweeps' operand fetches with several instances per wavefront.  In the sweeps lane j (and its mirror j + 8)
// of DPP row r reads word j of instance r's stage block: 8 consecutive doubles per row, rows `stride` doubles apart (RowLdsC::per_instance(N) = 1617 at
// N = 20).  Which strides let the three (four) rows be served without conflicts?  Cycles per ds_read_b64 / ds_read2_b64 / ds_write_b64 for a range of strides.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int OP>
__global__ void k(double *o, long long *cyc, int iters, int stride, int rows)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 4 * 1700 + 128; i += 64) lds[i] = 1.0 + i;
    __syncthreads();
    const int lane = threadIdx.x, j = lane & 7, row = lane >> 4;
    const int r = row < rows ? row : 0;
    const unsigned a = (unsigned)(size_t)(lds + r * stride + j);
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, x = 1.0 + lane;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) asm volatile("ds_read_b64 %0, %4\nds_read_b64 %1, %4 offset:64\nds_read_b64 %2, %4 offset:128\nds_read_b64 %3, %4 offset:192\ns_waitcnt lgkmcnt(0)\n"
                                  : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(a));
        if (OP == 1) asm volatile("ds_read2_b64 %0, %2 offset0:0 offset1:8\nds_read2_b64 %1, %2 offset0:16 offset1:24\ns_waitcnt lgkmcnt(0)\n"
                                  : "=&v"(*(double2 *)&v[0]), "=&v"(*(double2 *)&v[2]) : "v"(a));
        if (OP == 2) asm volatile("ds_write_b64 %1, %0\nds_write_b64 %1, %0 offset:64\nds_write_b64 %1, %0 offset:128\nds_write_b64 %1, %0 offset:192\ns_waitcnt lgkmcnt(0)\n"
                                  :: "v"(x), "v"(a) : "memory");
    }
    long long t1 = clock64();
    double s = 0; for (int q = 0; q < 8; q++) s += v[q];
    o[blockIdx.x * 64 + threadIdx.x] = s + lds[lane];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main(int argc, char **argv)
{
    double *o; long long *cyc; hipMalloc(&o, 1024 * 64 * 8); hipMalloc(&cyc, 1024 * 8);
    const int iters = 2000, grid = 1024;
    const int lo = argc > 1 ? atoi(argv[1]) : 1600, hi = argc > 2 ? atoi(argv[2]) : 1664;
    const char *names[] = {"4 x ds_read_b64", "2 x ds_read2_b64", "4 x ds_write_b64"};
    const size_t shm = (4 * 1700 + 128) * 8;
    printf("cycles per pass (4 loads/stores of 8 consecutive doubles per 16-lane row, lanes 8..15 mirror 0..7), 4 wavefronts per CU; rows = instances per wavefront\n");
    printf("%8s", "stride");
    for (int rows : {1, 3, 4}) for (int op = 0; op < 3; op++) printf("  r%d:%-16s", rows, names[op]);
    printf("\n");
    for (int stride = lo; stride <= hi; stride++) {
        printf("%8d", stride);
        for (int rows : {1, 3, 4}) {
#define RUN(M) { hipLaunchKernelGGL(k<M>, dim3(grid), dim3(64), shm, 0, o, cyc, iters, stride, rows); hipDeviceSynchronize(); static long long h[1024]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost); \
                 double m = 0; for (int i = 0; i < grid; i++) m += h[i]; printf("  %-19.1f", m / grid / iters); }
            RUN(0) RUN(1) RUN(2)
        }
        printf("\n");
    }
    return 0;
}
