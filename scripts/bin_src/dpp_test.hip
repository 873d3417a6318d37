#include <hip/hip_runtime.h>
__device__ __forceinline__ double wave_shr1(double v) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_shl1(double v) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k(double* out, const double* in) {
    double v = in[threadIdx.x];
    out[threadIdx.x] = wave_shr1(v) + 2.0 * wave_shl1(v);
}
