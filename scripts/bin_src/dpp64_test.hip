// Micro-test: 64-bit DPP row_newbcast on gfx950 -- semantics and issue cost of v_fmac_f64_dpp / v_mov_b64_dpp.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void sem(double *o, const double *a, const double *b)
{
    double acc = 0.0, x = a[threadIdx.x], y = b[threadIdx.x];
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y));
    o[threadIdx.x] = acc;     // expect a[row*16+3] * b[lane]
}
template <int MODE>
__global__ void timing(double *o, const double *a, long long *cyc, int iters)
{
    double x = a[threadIdx.x];
    double c[8];
    for (int k = 0; k < 8; k++) c[k] = a[threadIdx.x + 64 * (k + 1)];
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(c[k & 7]) : "v"(x), "v"(x));
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(c[k & 7]) : "v"(x), "v"(x));
        } else if (MODE == 2) {      // dependent chain, dpp
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_fmac_f64_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(c[0]) : "v"(x));
        } else if (MODE == 3) {      // dependent chain, plain
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_fmac_f64_e32 %0, %1, %0" : "+v"(c[0]) : "v"(x));
        } else if (MODE == 4) {      // dependent through the DPP source: producer -> dpp consumer
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(c[0]) : "v"(x));
        } else if (MODE == 5) {      // v_mov_b64_dpp
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(c[k & 7]) : "v"(x));
        }
    }
    long long t1 = clock64();
    double s = 0; for (int k = 0; k < 8; k++) s += c[k];
    o[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double *a, *b, *o; long long *cyc;
    hipMalloc(&a, 64 * 16 * 8); hipMalloc(&b, 64 * 8); hipMalloc(&o, 64 * 8); hipMalloc(&cyc, 8 * 64);
    std::vector<double> ha(64 * 16), hb(64), ho(64);
    for (int i = 0; i < 64 * 16; i++) ha[i] = 1.0 + 1e-3 * (i % 64);
    for (int i = 0; i < 64; i++) hb[i] = 2.0 + i;
    hipMemcpy(a, ha.data(), ha.size() * 8, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 64 * 8, hipMemcpyHostToDevice);
    sem<<<1, 64>>>(o, a, b); hipMemcpy(ho.data(), o, 64 * 8, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; i++) if (ho[i] != ha[(i / 16) * 16 + 3] * hb[i]) bad++;
    printf("semantics: %s (lane 20: %g expect %g)\n", bad ? "MISMATCH" : "ok", ho[20], ha[19] * hb[20]);
    const int iters = 2000; long long h;
    const char *names[] = {"fmac_f64_dpp independent", "fmac_f64 independent", "fmac_f64_dpp dependent acc", "fmac_f64 dependent acc", "fmac_f64_dpp dependent dpp-src", "mov_b64_dpp"};
#define RUN(M) timing<M><<<1, 64>>>(o, a, cyc, iters); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-34s %.2f cycles/instr\n", names[M], (double)h / (iters * 64.0));
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    return 0;
}
