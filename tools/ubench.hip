// Micro-benchmarks (not shipped) that size the fused ViterbiNet kernel: VALU / MFMA issue costs at 1,2,4
// waves per SIMD, cost of the sigmoid's special instructions, an exhaustive check of a cheap exact
// reciprocal on [1,inf], and the DPP lane maps the row-sweep relies on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

// ---- generic timing harness: kernel writes per-wave cycle counts
template <int KIND>
__global__ __launch_bounds__(256) void kbench(unsigned long long* cyc, float* sink, int iters, float seed) {
    float x0 = seed + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float a = 1.0001f, b = 0.5f;
    f32x4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    int qi = threadIdx.x;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {  // 32 independent-ish fma (8 chains x 4)
            REP4(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
        } else if (KIND == 1) {  // 4 independent MFMA 16x16x4 f32 x 8
            REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c1, 0, 0, 0);
                 c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c3, 0, 0, 0);)
        } else if (KIND >= 10 && KIND < 20) {  // per iteration: 8 x (4 MFMA + V fma), V = 8*(KIND-10)/... see below
            constexpr int V = (KIND - 10) * 8;  // VALU per 4 MFMAs
            REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c1, 0, 0, 0);
                 c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c3, 0, 0, 0);
                 for (int v = 0; v < V / 8; ++v) asm volatile("v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n"
                              "v_fma_f32 %4, %4, %6, %7\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n"
                              : "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
        } else if (KIND == 2) { REP32(asm volatile("v_rcp_f32 %0, %1\n v_rcp_f32 %2, %3\n" : "=v"(x0), "+v"(x1), "=v"(x2), "+v"(x3));)
        } else if (KIND == 3) { REP32(asm volatile("v_ldexp_f32 %0, %1, %4\n v_ldexp_f32 %2, %3, %4\n" : "=v"(x0), "+v"(x1), "=v"(x2), "+v"(x3) : "v"(qi));)
        } else if (KIND == 4) { REP32(asm volatile("v_div_fixup_f32 %0, %1, %4, %5\n v_div_fixup_f32 %2, %3, %4, %5\n" : "=v"(x0), "+v"(x1), "=v"(x2), "+v"(x3) : "v"(a), "v"(b));)
        } else if (KIND == 5) { REP32(asm volatile("v_rndne_f32 %0, %1\n v_cvt_i32_f32 %2, %3\n" : "=v"(x0), "+v"(x1), "=v"(x2), "+v"(x3));)
        } else if (KIND == 6) { REP32(asm volatile("v_div_scale_f32 %0, vcc, %1, %4, %5\n v_div_fmas_f32 %2, %3, %4, %5\n" : "=v"(x0), "+v"(x1), "=v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");)
        } else if (KIND == 7) { REP32(asm volatile("v_min_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_f32_dpp %2, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n" : "=v"(x0), "+v"(x1), "=v"(x2), "+v"(x3));)
        } else if (KIND == 8) { REP32(asm volatile("v_med3_f32 %0, %1, %4, %5\n v_mul_f32 %2, %3, %4\n" : "=v"(x0), "+v"(x1), "=v"(x2), "+v"(x3) : "v"(a), "v"(b));)
        } else if (KIND == 9) {  // dependent DPP-min chain (ACS recurrence latency): add + min_dpp, 32 steps
            REP32(asm volatile("v_add_f32 %0, %0, %1\n s_nop 1\n v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" : "+v"(x0) : "v"(a));)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + c0[0] + c1[1] + c2[2] + c3[3];
}

template <int KIND>
void run(const char* name, double ops_per_iter, int waves_per_simd) {
    const int iters = 2000;
    int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD of a CU
    unsigned long long* cyc; float* sink;
    CHECK(hipMalloc(&cyc, blocks * 4 * 8)); CHECK(hipMalloc(&sink, blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kbench<KIND>, dim3(blocks), dim3(256), 0, 0, cyc, sink, 10, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kbench<KIND>, dim3(blocks), dim3(256), 0, 0, cyc, sink, iters, 1.0f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    double med = (double)h[h.size() / 2];
    printf("%-28s waves/SIMD=%d  cycles/wave(med)=%.0f  cyc per op per wave=%.2f  SIMD cyc per op=%.2f  wall=%.3f ms (%.2f GHz-equiv)\n", name,
           waves_per_simd, med, med / (iters * ops_per_iter), med / (iters * ops_per_iter) / waves_per_simd, ms, med / (ms * 1e6));
    CHECK(hipFree(cyc)); CHECK(hipFree(sink));
}

// ---- exhaustive exact-reciprocal check on [1, inf]
__device__ __forceinline__ float fast_recip(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return __builtin_amdgcn_div_fixupf(r, x, 1.0f);
}
__device__ __forceinline__ float fast_recip1(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return __builtin_amdgcn_div_fixupf(r, x, 1.0f);
}
__global__ void krecip(unsigned long long* bad2, unsigned long long* bad1, unsigned* first_bad) {
    unsigned long long n2 = 0, n1 = 0;
    const unsigned lo = 0x3f800000u, hi = 0x7f800000u;  // [1.0, +inf]
    for (unsigned long long u = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u <= hi; u += (unsigned long long)gridDim.x * blockDim.x) {
        float x = __uint_as_float((unsigned)u);
        float ref = 1.0f / x;
        float f2 = fast_recip(x), f1 = fast_recip1(x);
        if (__float_as_uint(f2) != __float_as_uint(ref)) { ++n2; atomicMin(first_bad, (unsigned)u); }
        if (__float_as_uint(f1) != __float_as_uint(ref)) ++n1;
    }
    atomicAdd(bad2, n2); atomicAdd(bad1, n1);
}

// ---- DPP lane maps
__global__ void kdpp(int* out) {
    int l = threadIdx.x;
    out[0 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    out[1 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    out[2 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x141, 0xF, 0xF, false);  // row_half_mirror
    out[3 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x140, 0xF, 0xF, false);  // row_mirror
    out[4 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x128, 0xF, 0xF, false);  // row_ror:8
    out[5 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x104, 0xF, 0x5, false);  // row_shl:4, banks 0,2
    out[6 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x114, 0xF, 0xA, false);  // row_shr:4, banks 1,3
    out[7 * 64 + l] = __builtin_amdgcn_ds_bpermute(4 * ((2 * (l & 15)) % 16 + (l & 48)), l);
}

int main(int argc, char** argv) {
    int which = argc > 1 ? atoi(argv[1]) : 0;
    if (which == 0 || which == 1) {
        for (int w : {1, 2, 4}) run<0>("v_fma_f32 x32", 32, w);
        for (int w : {1, 2, 4}) run<1>("mfma16x16x4 x32 (4 acc)", 32, w);
        for (int w : {1, 2, 4}) run<10>("8x(4 mfma + 0 fma)", 8, w);
        for (int w : {1, 2, 4}) run<11>("8x(4 mfma + 8 fma)", 8, w);
        for (int w : {1, 2, 4}) run<12>("8x(4 mfma + 16 fma)", 8, w);
        for (int w : {1, 2, 4}) run<13>("8x(4 mfma + 24 fma)", 8, w);
        for (int w : {1, 2, 4}) run<14>("8x(4 mfma + 32 fma)", 8, w);
        for (int w : {1, 2, 4}) run<15>("8x(4 mfma + 40 fma)", 8, w);
        for (int w : {1, 4}) run<2>("v_rcp_f32 x64", 64, w);
        for (int w : {1, 4}) run<3>("v_ldexp_f32 x64", 64, w);
        for (int w : {1, 4}) run<4>("v_div_fixup_f32 x64", 64, w);
        for (int w : {1, 4}) run<5>("rndne+cvt_i32 x64", 64, w);
        for (int w : {1, 4}) run<6>("div_scale+div_fmas x64", 64, w);
        for (int w : {1, 4}) run<7>("v_min_f32_dpp x64", 64, w);
        for (int w : {1, 4}) run<8>("med3+mul x64", 64, w);
        for (int w : {1, 4}) run<9>("add+min_dpp dependent x32", 32, w);
    }
    if (which == 0 || which == 2) {
        unsigned long long *b2, *b1; unsigned* fb;
        CHECK(hipMalloc(&b2, 8)); CHECK(hipMalloc(&b1, 8)); CHECK(hipMalloc(&fb, 4));
        CHECK(hipMemset(b2, 0, 8)); CHECK(hipMemset(b1, 0, 8)); CHECK(hipMemset(fb, 0xff, 4));
        hipLaunchKernelGGL(krecip, dim3(4096), dim3(256), 0, 0, b2, b1, fb);
        unsigned long long h2, h1; unsigned hf;
        CHECK(hipMemcpy(&h2, b2, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&h1, b1, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&hf, fb, 4, hipMemcpyDeviceToHost));
        printf("exact reciprocal on [1,inf] (%u values): 2-step Newton+fixup mismatches=%llu (first 0x%08x), 1-step mismatches=%llu\n", 0x7f800000u - 0x3f800000u + 1, h2, hf, h1);
    }
    if (which == 0 || which == 3) {
        int* o; CHECK(hipMalloc(&o, 8 * 64 * 4));
        hipLaunchKernelGGL(kdpp, dim3(1), dim3(64), 0, 0, o);
        int h[8 * 64]; CHECK(hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost));
        const char* names[8] = {"quad_perm[1,0,3,2]", "quad_perm[2,3,0,1]", "row_half_mirror", "row_mirror", "row_ror:8", "row_shl:4 bank0101", "row_shr:4 bank1010", "bpermute 2s%16"};
        for (int k = 0; k < 8; ++k) { printf("%-20s:", names[k]); for (int l = 0; l < 32; ++l) printf(" %d", h[k * 64 + l]); printf("\n"); }
    }
    return 0;
}
