// What does SQ_LDS_BANK_CONFLICT count?  Five access patterns whose conflicts are known by construction, one kernel each, to be run under
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE -- ./lds_conflict_counter_test
//   P0  ds_read_b32, 64 lanes x consecutive dwords        (256 B, every bank once: conflict-free in one pass)
//   P1  ds_read_b64, 64 lanes x consecutive doubles       (512 B: conflict-free, but twice the 64 x 4 B the banks deliver per cycle)
//   P2  ds_read_b64, all lanes the same double            (broadcast)
//   P3  ds_read_b64, the sweeps' operand fetch with one instance per wavefront: lane j and j + 8 of every row word j (8 distinct doubles)
//   P4  as P3 with three instances per wavefront, rows 1617 doubles apart (RowLdsC::per_instance(20)), fourth row mirrors the first
//   P5  ds_read_b64, 64 lanes, stride 32 doubles          (every lane the same two banks: a true 64-way conflict)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int P>
__global__ void pattern(double *o, int iters)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = 1.0 + i;
    __syncthreads();
    const int lane = threadIdx.x, j = lane & 7, row = lane >> 4;
    int word = 0;
    if (P == 0) word = lane;            // dword index (b32)
    if (P == 1) word = lane;
    if (P == 2) word = 5;
    if (P == 3) word = j;
    if (P == 4) word = (row < 3 ? row : 0) * 1617 + j;
    if (P == 5) word = lane * 32;
    const unsigned a = (unsigned)(size_t)lds + (P == 0 ? 4u : 8u) * (unsigned)word;
    double v = 0; float f = 0;
    for (int it = 0; it < iters; it++) {
        if (P == 0) asm volatile("ds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)\n" : "=&v"(f) : "v"(a));
        else asm volatile("ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)\n" : "=&v"(v) : "v"(a));
    }
    o[blockIdx.x * 64 + threadIdx.x] = v + f;
}
int main()
{
    double *o; hipMalloc(&o, 1024 * 64 * 8);
    const int iters = 1000, grid = 1024; const size_t shm = 8192 * 8;
    hipLaunchKernelGGL(pattern<0>, dim3(grid), dim3(64), shm, 0, o, iters);
    hipLaunchKernelGGL(pattern<1>, dim3(grid), dim3(64), shm, 0, o, iters);
    hipLaunchKernelGGL(pattern<2>, dim3(grid), dim3(64), shm, 0, o, iters);
    hipLaunchKernelGGL(pattern<3>, dim3(grid), dim3(64), shm, 0, o, iters);
    hipLaunchKernelGGL(pattern<4>, dim3(grid), dim3(64), shm, 0, o, iters);
    hipLaunchKernelGGL(pattern<5>, dim3(grid), dim3(64), shm, 0, o, iters);
    hipDeviceSynchronize();
    printf("done: 6 patterns x %d wavefronts x %d reads (+ 128 initialising ds_write_b64 per wavefront)\n", grid, iters);
    return 0;
}
