// v_mfma_f64_4x4x4f64 (4 blocks of 4x4x4 per wavefront): operand lane layouts found by unit probes, and the cost of dependent links.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void probe(unsigned long long *hit)   // block (la, lb): A = e_la, B = e_lb -> which D lanes see the product
{
    const int l = threadIdx.x, la = blockIdx.x >> 6, lb = blockIdx.x & 63;
    const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    const unsigned long long m = __ballot(d != 0.0);
    if (l == 0) hit[blockIdx.x] = m;
}
__global__ void timing(double *out, int n)
{
    const int l = threadIdx.x;
    double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l, c = 0.0;
    long long t0 = clock64();
    for (int i = 0; i < n; i++) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);              // accumulate chain (C <- D)
    long long t1 = clock64();
    double d = 0.0;
    for (int i = 0; i < n; i++) { d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0); a = d; }  // D feeds the next A directly
    long long t2 = clock64();
    double e = 0.0, a2 = a;
    for (int i = 0; i < n; i++) { e = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b, 0.0, 0, 0, 0); a2 = fma(e, 1e-30, a2); }   // D -> one VALU op -> A
    long long t3 = clock64();
    double f0 = 0, f1 = 0, f2 = 0, f3 = 0;
    for (int i = 0; i < n; i++) {     // four independent accumulators back to back (issue rate)
        f0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f0, 0, 0, 0); f1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f1, 0, 0, 0);
        f2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f2, 0, 0, 0); f3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f3, 0, 0, 0);
    }
    long long t4 = clock64();
    double g = 1.0 + l;
    for (int i = 0; i < n; i++) g = fma(g, 1.0000001, 1e-9);        // dependent FP64 FMA chain for scale
    long long t5 = clock64();
    out[l] = c + d + e + f0 + f1 + f2 + f3 + g + a2;
    if (l == 0) { out[64] = (double)(t1 - t0) / n; out[65] = (double)(t2 - t1) / n; out[66] = (double)(t3 - t2) / n; out[67] = (double)(t4 - t3) / (4.0 * n); out[68] = (double)(t5 - t4) / n; }
}
int main()
{
    unsigned long long *H, h[4096]; hipMalloc(&H, sizeof(h));
    probe<<<4096, 64>>>(H); hipMemcpy(h, H, sizeof(h), hipMemcpyDeviceToHost);
    // Found (printed below, 256 products): A lane = 16 k + 4 blk + i?  The table is fitted against
    //   A[blk][i][k] in lane 16 i' ... -- see the fit: every hit must satisfy lane(D) = f(i, j, blk).
    if (getenv("MFMA4_DUMP")) for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++) if (h[la * 64 + lb]) printf("%d %d %llx\n", la, lb, h[la * 64 + lb]);
    // fit: A lane = i + 4 k + 16 blk? B lane = j + 4 k + 16 blk? D lane = ?  try all assignments of the three 2-bit fields of a lane to (x, k, blk)
    const char *names[6] = {"x=l&3,k=(l>>2)&3,b=l>>4", "x=l&3,b=(l>>2)&3,k=l>>4", "k=l&3,x=(l>>2)&3,b=l>>4", "b=l&3,x=(l>>2)&3,k=l>>4", "k=l&3,b=(l>>2)&3,x=l>>4", "b=l&3,k=(l>>2)&3,x=l>>4"};
    auto field = [](int l, int perm, int which) {   // which: 0 = x (row/col index), 1 = k, 2 = blk
        const int f0 = l & 3, f1 = (l >> 2) & 3, f2 = l >> 4;
        const int tab[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};   // position of x, k, blk
        const int pos = tab[perm][which];
        return pos == 0 ? f0 : (pos == 1 ? f1 : f2);
    };
    for (int pa = 0; pa < 6; pa++) for (int pb = 0; pb < 6; pb++) for (int pd = 0; pd < 6; pd++) {
        // D lane fields: (i, j, blk) in the three positions given by pd (x -> i, k -> j, blk -> blk)
        int bad = 0;
        for (int la = 0; la < 64 && !bad; la++) for (int lb = 0; lb < 64; lb++) {
            const int i = field(la, pa, 0), ka = field(la, pa, 1), ba = field(la, pa, 2), j = field(lb, pb, 0), kb = field(lb, pb, 1), bb = field(lb, pb, 2);
            unsigned long long expect = 0;
            if (ka == kb && ba == bb) {
                const int tab[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
                int f[3]; f[tab[pd][0]] = i; f[tab[pd][1]] = j; f[tab[pd][2]] = ba;
                expect = 1ull << (f[0] + 4 * f[1] + 16 * f[2]);
            }
            if (h[la * 64 + lb] != expect) { bad = 1; break; }
        }
        if (!bad) printf("layout: A {%s}  B {%s}  D {i,j,blk as x,k,b in %s}\n", names[pa], names[pb], names[pd]);
    }
    double *T, hT[69]; hipMalloc(&T, sizeof(hT));
    timing<<<1, 64>>>(T, 4000); hipMemcpy(hT, T, sizeof(hT), hipMemcpyDeviceToHost);
    printf("cycles (clock64 ticks) per link: accumulate C<-D %.1f | D->A direct %.1f | D->fma->A %.1f | independent back-to-back %.1f | dependent v_fma_f64 %.1f\n", hT[64], hT[65], hT[66], hT[67], hT[68]);
    return 0;
}
