// Micro-test: do several lanes of a 16-lane group storing to the SAME LDS address serialise?  (The sweeps' idle lanes "store to a dead word":
// 10 of 16 lanes per K~ store, 15 of 16 per factor store, profiles/r02_c3_pmc_summary.json counts 46 % of the LDS cycles as bank conflicts.)
// Modes: every lane its own address (conflict-free) | lanes >= 6 of each 16-lane row one shared dead word | lanes >= 1 | those lanes switched off by EXEC.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double *o, long long *cyc, int iters)
{
    __shared__ double lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = 1.0 + i;
    __syncthreads();
    const int lane = threadIdx.x, l15 = lane & 15, row = lane >> 4;
    double c[6] = {1, 2, 3, 4, 5, 6}, x = 1.0 + lane * 1e-9;
    int word = row * 512 + l15;                              // conflict-free: 16 consecutive doubles per row
    if (MODE == 1 && l15 >= 6) word = row * 512 + 47;
    if (MODE == 2 && l15 >= 1) word = row * 512 + 47;
    if (MODE == 4 && l15 >= 6) word = row * 512 + 64 + 2 * l15;   // distinct dead words, distinct banks
    const unsigned a = (unsigned)(size_t)(lds + word);
    const unsigned long long mask = MODE == 3 ? 0x003f003f003f003full : ~0ull;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#define F(n) "v_fmac_f64_e32 %" #n ", %6, %6\n"
        if (MODE != 3) asm volatile(F(0) F(1) F(2) "ds_write2_b64 %7, %0, %1 offset0:0 offset1:8\n" F(3) F(4) F(5) "ds_write_b64 %7, %2 offset:1024\n" "s_waitcnt lgkmcnt(0)\n"
                                    : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]) : "v"(x), "v"(a));
        else asm volatile(F(0) F(1) F(2) "s_mov_b64 exec, %8\n" "ds_write2_b64 %7, %0, %1 offset0:0 offset1:8\n" "s_mov_b64 exec, -1\n" F(3) F(4) F(5)
                          "s_mov_b64 exec, %8\n" "ds_write_b64 %7, %2 offset:1024\n" "s_mov_b64 exec, -1\n" "s_waitcnt lgkmcnt(0)\n"
                          : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]) : "v"(x), "v"(a), "s"(mask));
    }
    long long t1 = clock64();
    double s = 0; for (int j = 0; j < 6; j++) s += c[j];
    o[blockIdx.x * 64 + threadIdx.x] = s + lds[lane];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double *o; long long *cyc; hipMalloc(&o, 4096 * 64 * 8); hipMalloc(&cyc, 4096 * 8);
    const int iters = 2000;
    const char *names[] = {"every lane its own word", "lanes 6..15 of a row share one dead word", "lanes 1..15 of a row share one dead word",
                           "lanes 6..15 switched off by EXEC (s_mov exec around the store)", "lanes 6..15 distinct dead words"};
    for (int grid : {256, 1024}) {
        printf("grid %d (wavefronts per CU: %s); 6 fmac + ds_write2_b64 + ds_write_b64 per pass, wait 0\n", grid, grid == 256 ? "1" : "4");
#define RUN(M) { hipLaunchKernelGGL(k<M>, dim3(grid), dim3(64), 0, 0, o, cyc, iters); hipDeviceSynchronize(); long long h[1024]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost); \
                 double m = 0; for (int i = 0; i < grid; i++) m += h[i]; printf("  %-64s %.1f cycles per pass\n", names[M], m / grid / iters); }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
    }
    return 0;
}
