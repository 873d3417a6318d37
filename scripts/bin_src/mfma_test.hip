#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void layout(const double* A, const double* B, double* D) {
    // A is 16x4 row-major, B is 4x16 row-major
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[l * 4 + r] = c[r];
}
__global__ void timing(double* out, int n) {
    int l = threadIdx.x;
    double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
    d4 c = {0, 0, 0, 0};
    long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        a = c[0] * 1e-30 + a;   // make next A depend on the result (dependent chain incl. one VALU op)
    }
    long long t1 = clock64();
    d4 e = {0,0,0,0};
    long long t2 = clock64();
    for (int i = 0; i < n; i++) { e = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e, 0, 0, 0); }
    long long t3 = clock64();
    out[l] = c[0] + c[1] + e[2];
    if (l == 0) { out[64] = (double)(t1 - t0) / n; out[65] = (double)(t3 - t2) / n; }
}
int main() {
    double hA[64], hB[64], hD[256];
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) hA[i * 4 + k] = 1 + i + 0.1 * k;
    for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) hB[k * 16 + j] = (k + 1) * 0.01 + j * 3;
    double *A, *B, *D, *T; hipMalloc(&A, 512); hipMalloc(&B, 512); hipMalloc(&D, 2048); hipMalloc(&T, 66 * 8);
    hipMemcpy(A, hA, 512, hipMemcpyHostToDevice); hipMemcpy(B, hB, 512, hipMemcpyHostToDevice);
    layout<<<1, 64>>>(A, B, D); hipMemcpy(hD, D, 2048, hipMemcpyDeviceToHost);
    // test hypothesis: D[row][col] with col = l & 15, row = (l >> 4) + 4 * r
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
        int col = l & 15, row = (l >> 4) + 4 * r; double ref = 0;
        for (int k = 0; k < 4; k++) ref += hA[row * 4 + k] * hB[k * 16 + col];
        if (fabs(ref - hD[l * 4 + r]) > 1e-9) bad++;
    }
    printf("layout hypothesis row=(l>>4)+4r, col=l&15: mismatches %d\n", bad);
    timing<<<1, 64>>>(T, 2000); double hT[66]; hipMemcpy(hT, T, 66 * 8, hipMemcpyDeviceToHost);
    printf("cycles per dependent (mfma + 1 fma): %.1f ; per back-to-back accumulate mfma: %.1f\n", hT[64], hT[65]);
    return 0;
}
