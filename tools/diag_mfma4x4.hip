// Diagnostic (not shipped): lane maps and issue cost of v_mfma_f32_4x4x1_16b_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void klay(float* out) {
    int l = threadIdx.x;
    // A value encodes (lane), B value encodes (lane): D[r] = a*b of the matching block; probe with one-hot
    for (int src = 0; src < 64; ++src) {
        float a = (l == src) ? 1.0f : 0.0f;
        float b = 1000.0f + l;  // B identifies its lane
        f32x4 c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[(src * 64 + l) * 4 + r] = c[r];
    }
}
__global__ __launch_bounds__(256) void kperf(unsigned long long* cyc, float* sink, int iters) {
    float x0 = threadIdx.x, x1 = x0 + 1;
    f32x4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, c3, 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ __launch_bounds__(256) void kperf_dep(unsigned long long* cyc, float* sink, int iters) {
    float x0 = threadIdx.x, x1 = x0 + 1;
    f32x4 c0 = {0,0,0,0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, c0, 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = c0[0];
}
int main() {
    float* out; hipMalloc(&out, 64 * 64 * 4 * 4);
    hipLaunchKernelGGL(klay, dim3(1), dim3(64), 0, 0, out);
    std::vector<float> h(64 * 64 * 4);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    // for A one-hot at lane src: which (lane l, reg r) outputs are nonzero, and which B lane they carry
    int ok = 1;
    for (int src = 0; src < 64; ++src)
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                float v = h[(src * 64 + l) * 4 + r];
                // hypothesis: A lane src = (block src/4, row src%4); D[l][r]: block l/4, row r, col l%4; B lane = (block, col) = l
                float expect = ((src / 4) == (l / 4) && (src % 4) == r) ? 1000.0f + l : 0.0f;
                if (v != expect) ok = 0;
            }
    printf("mfma 4x4x1 maps as assumed (A lane=(blk,row), B lane=(blk,col), D reg=row lane=(blk,col)): %d\n", ok);
    if (!ok) for (int src = 0; src < 8; ++src) { printf("src %d:", src); for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { float v = h[(src*64+l)*4+r]; if (v != 0) printf(" (l%d r%d %.0f)", l, r, v);} printf("\n"); }
    unsigned long long* cyc; float* sink;
    hipMalloc(&cyc, 256 * 4 * 4 * 8); hipMalloc(&sink, 256 * 4 * 256 * 4);
    for (int w : {1, 2, 4}) {
        hipLaunchKernelGGL(kperf, dim3(256 * w), dim3(256), 0, 0, cyc, sink, 2000);
        hipDeviceSynchronize();
        std::vector<unsigned long long> c(256 * w * 4); hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
        std::sort(c.begin(), c.end());
        printf("4x4x1 independent x4: waves/SIMD=%d cycles per MFMA per wave = %.2f\n", w, (double)c[c.size()/2] / (2000.0 * 32));
        hipLaunchKernelGGL(kperf_dep, dim3(256 * w), dim3(256), 0, 0, cyc, sink, 2000);
        hipDeviceSynchronize();
        hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
        std::sort(c.begin(), c.end());
        printf("4x4x1 dependent chain  : waves/SIMD=%d cycles per MFMA per wave = %.2f\n", w, (double)c[c.size()/2] / (2000.0 * 32));
    }
    return 0;
}
