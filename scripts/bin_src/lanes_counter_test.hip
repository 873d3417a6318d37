// Calibration of the VALU lane-activity counters (bench.py roofline.lanes_active): kernels whose EXEC popcount is known --
// lanes_k<K>: every wavefront executes a long chain of FP64 FMAs with exactly K of 64 lanes enabled (the others branch around it).
// Run under rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES and compare the ratio
// SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (and / SQ_INSTS_VALU) per kernel with K: profiles/r04_lanes_counter_calibration.json.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K>
__global__ void lanes_k(double *o, const double *a, int iters)
{
    double x = a[threadIdx.x], c0 = x, c1 = x + 1, c2 = x + 2, c3 = x + 3;
    if ((int)threadIdx.x < K) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                asm volatile("v_fmac_f64_e32 %0, %1, %1" : "+v"(c0) : "v"(x));
                asm volatile("v_fmac_f64_e32 %0, %1, %1" : "+v"(c1) : "v"(x));
                asm volatile("v_fmac_f64_e32 %0, %1, %1" : "+v"(c2) : "v"(x));
                asm volatile("v_fmac_f64_e32 %0, %1, %1" : "+v"(c3) : "v"(x));
            }
        }
    }
    o[blockIdx.x * 64 + threadIdx.x] = c0 + c1 + c2 + c3;
}
int main()
{
    const int blocks = 1024, iters = 2000;
    double *a, *o;
    hipMalloc(&a, 64 * 8); hipMalloc(&o, blocks * 64 * 8);
    double ha[64]; for (int i = 0; i < 64; i++) ha[i] = 1e-9 * i;
    hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice);
#define RUN(K) for (int r = 0; r < 3; r++) lanes_k<K><<<blocks, 64>>>(o, a, iters); hipDeviceSynchronize(); printf("lanes_k<%d> done\n", K);
    RUN(64) RUN(32) RUN(16) RUN(8) RUN(1)
    return 0;
}
