// Micro-test: what an LDS instruction costs a LONE wavefront per SIMD (1 or 4 wavefronts per CU) when it is interleaved with FP64
// VALU work -- the regime of the stage recursions.  Prints cycles per loop pass for several instruction mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double *o, long long *cyc, int iters)
{
    __shared__ double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 1.0 + i;
    __syncthreads();
    double c[6] = {1, 2, 3, 4, 5, 6}, x = 1.0 + threadIdx.x * 1e-9;
    const unsigned a = (unsigned)(size_t)(lds + (threadIdx.x & 7) * 6);
    double r0, r1, r2, r3, r4, r5;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 q0, q1, q2;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#define F(n) "v_fmac_f64_e32 %" #n ", %6, %6\n"
        if (MODE == 0) asm volatile(F(0) F(1) F(2) F(3) F(4) F(5) : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]) : "v"(x));
        if (MODE == 1) asm volatile("ds_read_b64 %7, %13\n" F(0) "ds_read_b64 %8, %13 offset:8\n" F(1) "ds_read_b64 %9, %13 offset:16\n" F(2)
                                    "ds_read_b64 %10, %13 offset:24\n" F(3) "ds_read_b64 %11, %13 offset:32\n" F(4) "ds_read_b64 %12, %13 offset:40\n" F(5) "s_waitcnt lgkmcnt(0)\n"
                                    : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]) : "v"(x), "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(a));
        if (MODE == 2) asm volatile("ds_read2_b64 %7, %10 offset0:0 offset1:1\n" F(0) F(1) "ds_read2_b64 %8, %10 offset0:2 offset1:3\n" F(2) F(3) "ds_read2_b64 %9, %10 offset0:4 offset1:5\n" F(4) F(5) "s_waitcnt lgkmcnt(0)\n"
                                    : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]) : "v"(x), "v"(q0), "v"(q1), "v"(q2), "v"(a));
        if (MODE == 3) asm volatile(F(0) F(1) F(2) "ds_write_b64 %7, %0 offset:2048\n" F(3) F(4) F(5) "s_waitcnt lgkmcnt(0)\n"
                                    : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]) : "v"(x), "v"(a));
        if (MODE == 4) asm volatile("ds_read_b64 %7, %13\n" F(0) "ds_read_b64 %8, %13 offset:8\n" F(1) "ds_read_b64 %9, %13 offset:16\n" F(2)
                                    "ds_read_b64 %10, %13 offset:24\n" F(3) "ds_read_b64 %11, %13 offset:32\n" F(4) "ds_read_b64 %12, %13 offset:40\n" F(5) "s_waitcnt lgkmcnt(6)\n"
                                    : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]) : "v"(x), "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(a));
        if (MODE == 5) asm volatile("ds_read2_b64 %7, %10 offset0:0 offset1:1\n" F(0) F(1) "ds_read2_b64 %8, %10 offset0:2 offset1:3\n" F(2) F(3) "ds_read2_b64 %9, %10 offset0:4 offset1:5\n" F(4) F(5) "s_waitcnt lgkmcnt(3)\n"
                                    : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]) : "v"(x), "v"(q0), "v"(q1), "v"(q2), "v"(a));
        if (MODE == 6) asm volatile("s_mov_b64 s[20:21], exec\ns_mov_b64 exec, 0xffff\n" "ds_read2_b64 %7, %10 offset0:0 offset1:1\n" "ds_read2_b64 %8, %10 offset0:2 offset1:3\n" "ds_read2_b64 %9, %10 offset0:4 offset1:5\n" "s_mov_b64 exec, s[20:21]\n" F(0) F(1) F(2) F(3) F(4) F(5) "s_waitcnt lgkmcnt(3)\n"
                                    : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]) : "v"(x), "v"(q0), "v"(q1), "v"(q2), "v"(a) : "s20", "s21");
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    long long t1 = clock64();
    double s = 0; for (int j = 0; j < 6; j++) s += c[j];
    o[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double *o; long long *cyc; hipMalloc(&o, 4096 * 64 * 8); hipMalloc(&cyc, 4096 * 8);
    const int iters = 2000;
    const char *names[] = {"6 fmac", "6 fmac + 6 ds_read_b64, wait 0", "6 fmac + 3 ds_read2_b64, wait 0", "6 fmac + 1 ds_write_b64, wait 0",
                           "6 fmac + 6 ds_read_b64, one pass in flight", "6 fmac + 3 ds_read2_b64, one pass in flight", "6 fmac + 3 ds_read2_b64 (EXEC = 16 lanes), in flight"};
    for (int grid : {1, 256, 1024}) {
        printf("grid %d (wavefronts per CU: %s)\n", grid, grid == 1 ? "one on the chip" : grid == 256 ? "1" : "4");
#define RUN(M) { hipLaunchKernelGGL(k<M>, dim3(grid), dim3(64), 0, 0, o, cyc, iters); hipDeviceSynchronize(); long long h[1024]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost); \
                 double m = 0; for (int i = 0; i < grid; i++) m += h[i]; printf("  %-56s %.1f cycles per pass\n", names[M], m / grid / iters); }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    }
    return 0;
}
