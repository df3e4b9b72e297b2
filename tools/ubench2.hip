// Micro-benchmark 2 (not shipped): issue rate of the VALU forms a lane-per-symbol matmul would use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void kb(unsigned long long* cyc, float* sink, int iters, const float* __restrict__ w) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float h = 0.5f + threadIdx.x * 1e-3f;
    f32x2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, ph = {h, h};
    float s0 = w[0], s1 = w[1];
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {  // VOP2 fmac with SGPR multiplier: acc += s*h   (32 per iter)
            REP4(asm volatile("v_fmac_f32 %0, %8, %10\n v_fmac_f32 %1, %9, %10\n v_fmac_f32 %2, %8, %10\n v_fmac_f32 %3, %9, %10\n"
                              "v_fmac_f32 %4, %8, %10\n v_fmac_f32 %5, %9, %10\n v_fmac_f32 %6, %8, %10\n v_fmac_f32 %7, %9, %10\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(s0), "s"(s1), "v"(h));)
        } else if (KIND == 1) {  // pk_fma: 2 FMAs per lane per instr (16 per iter x4 = 32 instrs)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ph));)
        } else if (KIND == 2) {  // VOP2 fmac all-VGPR
            REP4(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(h), "v"(x0));)
        } else if (KIND == 3) {  // v_add + v_mul VOP2 mix
            REP4(asm volatile("v_add_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                              "v_add_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(h));)
        } else if (KIND == 4) {  // v_fmaak (VOP2 + literal)
            REP4(asm volatile("v_fmaak_f32 %0, %8, %0, 0x3f000000\n v_fmaak_f32 %1, %8, %1, 0x3f000000\n v_fmaak_f32 %2, %8, %2, 0x3f000000\n v_fmaak_f32 %3, %8, %3, 0x3f000000\n"
                              "v_fmaak_f32 %4, %8, %4, 0x3f000000\n v_fmaak_f32 %5, %8, %5, 0x3f000000\n v_fmaak_f32 %6, %8, %6, 0x3f000000\n v_fmaak_f32 %7, %8, %7, 0x3f000000\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(h));)
        } else if (KIND == 5) {  // min + cndmask + cmp (ACS-ish)
            REP4(asm volatile("v_min_f32 %0, %8, %0\n v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %1, %1, %2, vcc\n v_min_f32 %3, %8, %3\n"
                              "v_min_f32 %4, %8, %4\n v_cmp_lt_f32 vcc, %5, %6\n v_cndmask_b32 %5, %5, %6, vcc\n v_min_f32 %7, %8, %7\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(h) : "vcc");)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p1[1] + p2[0] + p3[1];
}

template <int KIND>
void run(const char* name, double ops_per_iter, int wps) {
    const int iters = 4000;
    int blocks = 256 * wps;
    unsigned long long* cyc; float *sink, *w;
    CHECK(hipMalloc(&cyc, blocks * 4 * 8)); CHECK(hipMalloc(&sink, blocks * 256 * 4)); CHECK(hipMalloc(&w, 64)); CHECK(hipMemset(w, 0, 64));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kb<KIND>, dim3(blocks), dim3(256), 0, 0, cyc, sink, 10, w);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kb<KIND>, dim3(blocks), dim3(256), 0, 0, cyc, sink, iters, w);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    double med = (double)h[h.size() / 2];
    double total_ops = (double)iters * ops_per_iter * blocks * 4;  // wave-instructions
    printf("%-26s waves/SIMD=%d  ticks/op/wave=%.2f  wall ns per op per SIMD=%.3f  (=%.2f cyc @2.4GHz)\n", name, wps,
           med / (iters * ops_per_iter), ms * 1e6 / (total_ops / 1024.0), ms * 1e6 / (total_ops / 1024.0) * 2.4);
    CHECK(hipFree(cyc)); CHECK(hipFree(sink)); CHECK(hipFree(w));
}
int main() {
    for (int w : {1, 2, 3, 4, 6, 8}) run<0>("v_fmac_f32 v,s,v (VOP2)", 32, w);
    for (int w : {1, 2, 4, 8}) run<1>("v_pk_fma_f32", 32, w);
    for (int w : {1, 2, 4, 8}) run<2>("v_fmac_f32 v,v,v (VOP2)", 32, w);
    for (int w : {1, 2, 4, 8}) run<3>("v_add/v_mul (VOP2)", 32, w);
    for (int w : {1, 2, 4, 8}) run<4>("v_fmaak_f32", 32, w);
    for (int w : {1, 2, 4, 8}) run<5>("min/cmp/cndmask", 32, w);
    return 0;
}
