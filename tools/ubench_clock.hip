// ubench_clock.hip -- shader clock seen by a lone workgroup: s_memtime (shader clock) against s_memrealtime (100 MHz) around
// a dependent VALU chain, for (a) one 64-thread workgroup launched alone back to back (the by-word evaluation's pattern) and
// (b) the same chain on every CU.  hipcc --offload-arch=gfx950 -O2 tools/ubench_clock.hip -o tools/ubench_clock
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float *out, unsigned long long *stamps, int n) {
    float x = out[threadIdx.x];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamps[0] = c1 - c0;
        stamps[1] = r1 - r0;
    }
}
int main() {
    float *out;
    unsigned long long *st, h[2];
    hipMalloc(&out, 4096);
    hipMemset(out, 0, 4096);
    hipMalloc(&st, 16);
    for (int grid : {1, 256, 1, 2048, 1}) {
        for (int n : {2000, 20000}) {
            for (int rep = 0; rep < 300; ++rep) hipLaunchKernelGGL(chain, dim3(grid), dim3(64), 0, 0, out, st, n);
            hipDeviceSynchronize();
            hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
            printf("grid %4d  chain %6d: %8llu shader cycles, %7llu x 10 ns -> %.3f GHz, %.2f cycles per dependent fma\n", grid, n, h[0], h[1],
                   (double)h[0] / ((double)h[1] * 10.0), (double)h[0] / n);
        }
    }
    return 0;
}
