// Diagnostic (not shipped): prints the lane semantics the MLP kernel relies on --
// v_permlane16_swap / v_permlane32_swap and the v_mfma_f32_16x16x4_f32 operand/result maps.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o16, unsigned* o32, float* mf) {
    int l = threadIdx.x;
    unsigned a = 100 + l, b = 200 + l;
    u32x2 p = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o16[l] = p[0]; o16[64 + l] = p[1];
    u32x2 q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o32[l] = q[0]; o32[64 + l] = q[1];
    // A[i][k] = 10*i + k (i=l&15,k=l>>4) ; B[k][j] = (k==2) ? j+1 : 0  -> D[i][j] = (10 i + 2) (j+1)
    float av = 10.f * (l & 15) + (l >> 4);
    float bv = ((l >> 4) == 2) ? (float)((l & 15) + 1) : 0.f;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) mf[l * 4 + r] = acc[r];
}
int main() {
    unsigned *o16, *o32; float* mf;
    hipMalloc(&o16, 512); hipMalloc(&o32, 512); hipMalloc(&mf, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o16, o32, mf);
    unsigned h16[128], h32[128]; float hm[256];
    hipMemcpy(h16, o16, 512, hipMemcpyDeviceToHost); hipMemcpy(h32, o32, 512, hipMemcpyDeviceToHost);
    hipMemcpy(hm, mf, 1024, hipMemcpyDeviceToHost);
    int ok16 = 1, ok32 = 1, okm = 1;
    for (int l = 0; l < 64; ++l) {
        int row = l >> 4;
        unsigned e0 = (row & 1) ? 200 + (l - 16) : 100 + l;  // vdst: odd rows <- src0 even rows
        unsigned e1 = (row & 1) ? 200 + l : 100 + (l + 16);  // src0: even rows <- vdst odd rows
        if (h16[l] != e0 || h16[64 + l] != e1) ok16 = 0;
        unsigned f0 = l >= 32 ? 200 + (l - 32) : 100 + l;
        unsigned f1 = l >= 32 ? 200 + l : 100 + (l + 32);
        if (h32[l] != f0 || h32[64 + l] != f1) ok32 = 0;
        for (int r = 0; r < 4; ++r) {
            int i = 4 * (l >> 4) + r, j = l & 15;
            if (hm[l * 4 + r] != (10.f * i + 2) * (j + 1)) okm = 0;
        }
    }
    printf("permlane16_swap as assumed: %d\npermlane32_swap as assumed: %d\nmfma16x16x4 maps as assumed: %d\n", ok16, ok32, okm);
    if (!ok16) { printf("p16 vdst:"); for (int l = 0; l < 64; ++l) printf(" %u", h16[l]); printf("\np16 src0:"); for (int l = 0; l < 64; ++l) printf(" %u", h16[64 + l]); printf("\n"); }
    if (!ok32) { printf("p32 vdst:"); for (int l = 0; l < 64; ++l) printf(" %u", h32[l]); printf("\np32 src0:"); for (int l = 0; l < 64; ++l) printf(" %u", h32[64 + l]); printf("\n"); }
    if (!okm) { for (int l = 0; l < 64; ++l) printf("lane %d: %g %g %g %g\n", l, hm[l*4], hm[l*4+1], hm[l*4+2], hm[l*4+3]); }
    return (ok16 && ok32 && okm) ? 0 : 1;
}
