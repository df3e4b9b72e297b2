// Micro-benchmark 3 (not shipped): what a pure streaming read reaches on this part, as the practical ceiling the
// cost-streaming ACS sweep is compared with.  Reads N bytes once, sums them, writes 4 B per wave.
//   kind 0: float4 loads, grid-stride, default cache policy
//   kind 1: float4 loads, nontemporal
//   kind 2: LDS-DMA (global_load_lds_dwordx4), 1 KB per wave instruction, ring of 3, like sweep16_lds
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_hbm tools/ubench_hbm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void rd(const v4f* __restrict__ src, size_t n4, float* __restrict__ out) {
    float acc = 0.0f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        v4f a, b, c, d;
        if (KIND == 1) {
            a = __builtin_nontemporal_load(src + i);
            b = __builtin_nontemporal_load(src + i + stride);
            c = __builtin_nontemporal_load(src + i + 2 * stride);
            d = __builtin_nontemporal_load(src + i + 3 * stride);
        } else {
            a = src[i]; b = src[i + stride]; c = src[i + 2 * stride]; d = src[i + 3 * stride];
        }
        acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
    }
    for (; i < n4; i += stride) {
        v4f a = src[i];
        acc += a.x + a.y + a.z + a.w;
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;  // never true for the test data; keeps the loads alive
}

// each wave streams a contiguous segment in 1-KB pieces through a ring of 3 x 4 KB in LDS
__global__ __launch_bounds__(256) void rd_dma(const float* __restrict__ src, size_t bytes_per_wave, float* __restrict__ out) {
    __shared__ float lds[4][3][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t w = (size_t)blockIdx.x * 4 + wave;
    const char* base = (const char*)src + w * bytes_per_wave;
    const int nchunks = (int)(bytes_per_wave / 4096);
    auto issue = [&](int c, int slot) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const char* g = base + (size_t)c * 4096 + r * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)&lds[wave][slot][r * 256], 16, 0, 2);
        }
    };
    float acc = 0.0f;
    for (int k = 0; k < 3 && k < nchunks; ++k) issue(k, k);
    for (int c = 0; c < nchunks; ++c) {
        const int slot = c % 3;
        if (c + 3 <= nchunks - 1 + 1 && c + 2 < nchunks) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float v = lds[wave][slot][lane] + lds[wave][slot][512 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += v;
        if (c + 3 < nchunks) issue(c + 3, slot);
    }
    if (acc == 123.456f) out[w] = acc;
}


// like the sweeps: a wave owns NSTR row streams (rows row_bytes apart, contiguous over all waves) and per chunk moves
// PIECE 1-KB instructions from each; ring of 3 chunks.  LDS per wave = 3 * NSTR * PIECE KB.
template <int NSTR, int PIECE, int WR>  // WR 1: also write 64 B per stream and chunk (4 B per 64 B read, like the decisions)
__global__ __launch_bounds__(64) void rd_rows(const float* __restrict__ src, size_t row_bytes, float* __restrict__ out,
                                              float* __restrict__ wr) {
    __shared__ float lds[3][NSTR * PIECE * 256];
    const int lane = threadIdx.x;
    const char* base = (const char*)src + (size_t)blockIdx.x * NSTR * row_bytes;
    const int nchunks = (int)(row_bytes / (1024 * PIECE));
    auto issue = [&](int c, int slot) {
#pragma unroll
        for (int r = 0; r < NSTR; ++r)
#pragma unroll
            for (int p = 0; p < PIECE; ++p) {
                const char* g = base + (size_t)r * row_bytes + ((size_t)c * PIECE + p) * 1024 + lane * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)&lds[slot][(r * PIECE + p) * 256], 16, 0, 2);
            }
    };
    float acc = 0.0f;
    for (int k = 0; k < 3 && k < nchunks; ++k) issue(k, k);
    constexpr int N = NSTR * PIECE;
    for (int c = 0; c < nchunks; ++c) {
        const int slot = c % 3;
        // younger than G(c): G(c+1), G(c+2) and, with one store per chunk, S(c-3..c-1) (fewer in the first chunks and
        // for WR == 3: then the wait is merely a little early/late, this is a bandwidth test)
        constexpr int YOUNGER = 2 * N + (WR ? 3 : 0);
        if (c + 2 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER <= 63 ? YOUNGER : 63) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float v = lds[slot][lane] + lds[slot][(N - 1) * 256 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += v;
        if (c + 3 < nchunks) issue(c + 3, slot);
        if (WR) {  // row of stream r: 1000 floats, 16 floats per chunk of 1 KB
            constexpr int LPS = 64 / NSTR;  // lanes per stream
            float* wp = wr + ((size_t)blockIdx.x * NSTR + lane / LPS) * 1000 + (size_t)c * 16 * PIECE;
            if (WR == 1) {
                if (LPS == 4) *(v4f*)(wp + 4 * (lane % LPS)) = v4f{v, v, v, v};
                else if (LPS == 16) wp[lane % LPS] = v;
            } else if (WR == 2) {  // same volume, but one contiguous, 1-KB-aligned KB per wave and chunk (NSTR = 16)
                float* wq = wr + ((size_t)blockIdx.x * 64 + c) * 256;
                *(v4f*)(wq + 4 * lane) = v4f{v, v, v, v};
            } else if (WR == 3) {  // same volume, 4 KB contiguous per wave every 4th chunk
                float* wq = wr + ((size_t)blockIdx.x * 64 + (c & ~3)) * 256;
                if ((c & 3) == 3)
                    for (int j = 0; j < 4; ++j) *(v4f*)(wq + j * 256 + 4 * lane) = v4f{v, v, v, v};
            }
        }
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}
template <int NSTR, int PIECE, int WR = 0>
void run_rows(const float* src, size_t bytes, float* out, hipEvent_t e0, hipEvent_t e1, float* wr = nullptr) {
    const size_t row_bytes = 64000 / (1024 * PIECE) * (1024 * PIECE);  // ~ T = 1000 steps x 64 B
    const size_t stride = 64000;
    (void)stride;
    const int grid = (int)(bytes / (row_bytes * NSTR));
    auto launch = [&]() { hipLaunchKernelGGL((rd_rows<NSTR, PIECE, WR>), dim3(grid), dim3(64), 0, 0, src, row_bytes, out, wr); };
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < 10; ++i) {
        CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double moved = (double)grid * NSTR * row_bytes;
    if (WR) printf("  + 64-B writes per 1 KB read: ");
    printf("bytes %5zu MiB rows: %2d streams/wave x %d KB pieces (%3d KB LDS/wave, %d waves): median %.4f ms  %.0f GB/s  best %.0f GB/s\n",
           bytes >> 20, NSTR, PIECE, 3 * NSTR * PIECE, grid, ms[5], moved / ms[5] * 1e-6, moved / ms[0] * 1e-6);
}

// WR == 4 experiment: same traffic as rd_rows<16,1,2>, but the stores are issued by a second wave of the workgroup
// (paced through an LDS progress word), so they do not sit in the DMA wave's in-order vmcnt queue.
__global__ __launch_bounds__(128) void rd_rows_split(const float* __restrict__ src, size_t row_bytes, float* __restrict__ out,
                                                     float* __restrict__ wr) {
    constexpr int NSTR = 16, N = 16;
    __shared__ float lds[3][N * 256];
    __shared__ volatile int progress;
    const int lane = threadIdx.x & 63;
    const int nchunks = (int)(row_bytes / 1024);
    if (threadIdx.x == 0) progress = 0;
    __syncthreads();
    if (threadIdx.x >= 64) {  // writer wave
        for (int c = 0; c < nchunks; ++c) {
            while (progress <= c) __builtin_amdgcn_s_sleep(2);
            float* wq = wr + ((size_t)blockIdx.x * 64 + c) * 256;
            *(v4f*)(wq + 4 * lane) = v4f{1.0f, 2.0f, 3.0f, (float)c};
        }
        return;
    }
    const char* base = (const char*)src + (size_t)blockIdx.x * NSTR * row_bytes;
    auto issue = [&](int c, int slot) {
#pragma unroll
        for (int r = 0; r < NSTR; ++r) {
            const char* g = base + (size_t)r * row_bytes + (size_t)c * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)&lds[slot][r * 256], 16, 0, 2);
        }
    };
    float acc = 0.0f;
    for (int k = 0; k < 3 && k < nchunks; ++k) issue(k, k);
    for (int c = 0; c < nchunks; ++c) {
        const int slot = c % 3;
        if (c + 2 < nchunks) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float v = lds[slot][lane] + lds[slot][(N - 1) * 256 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += v;
        if (c + 3 < nchunks) issue(c + 3, slot);
        if (lane == 0) progress = c + 1;
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}
void run_split(const float* src, size_t bytes, float* out, hipEvent_t e0, hipEvent_t e1, float* wr) {
    const size_t row_bytes = 64000 / 1024 * 1024;
    const int grid = (int)(bytes / (row_bytes * 16));
    auto launch = [&]() { hipLaunchKernelGGL(rd_rows_split, dim3(grid), dim3(128), 0, 0, src, row_bytes, out, wr); };
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < 10; ++i) {
        CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double moved = (double)grid * 16 * row_bytes;
    printf("  + the same writes from a second wave: bytes %5zu MiB 16 streams/wave (%d workgroups): median %.4f ms  %.0f GB/s  best %.0f GB/s\n",
           bytes >> 20, grid, ms[5], moved / ms[5] * 1e-6, moved / ms[0] * 1e-6);
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    const size_t sizes[] = {(size_t)640 << 20, (size_t)2560 << 20};
    float* out; CHECK(hipMalloc(&out, 1 << 22));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (size_t bytes : sizes) {
        float* src; CHECK(hipMalloc(&src, bytes)); CHECK(hipMemset(src, 0, bytes));
        for (int kind = 0; kind < 3; ++kind) {
            for (int wgs_per_cu : {2, 4, 8, 16}) {
                const int grid = prop.multiProcessorCount * wgs_per_cu;
                size_t bpw = bytes / ((size_t)grid * 4) / 4096 * 4096;
                auto launch = [&]() {
                    if (kind == 0) hipLaunchKernelGGL(rd<0>, dim3(grid), dim3(256), 0, 0, (const v4f*)src, bytes / 16, out);
                    else if (kind == 1) hipLaunchKernelGGL(rd<1>, dim3(grid), dim3(256), 0, 0, (const v4f*)src, bytes / 16, out);
                    else hipLaunchKernelGGL(rd_dma, dim3(grid), dim3(256), 0, 0, src, bpw, out);
                };
                if (kind == 2 && wgs_per_cu > 3) continue;  // 48 KB of LDS per workgroup
                for (int i = 0; i < 3; ++i) launch();
                CHECK(hipDeviceSynchronize());
                std::vector<float> ms;
                for (int i = 0; i < 10; ++i) {
                    CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
                }
                std::sort(ms.begin(), ms.end());
                const double moved = kind == 2 ? (double)bpw * grid * 4 : (double)bytes;
                printf("bytes %5zu MiB kind %d wgs/cu %2d : median %.4f ms  %.0f GB/s   best %.0f GB/s\n", bytes >> 20, kind,
                       wgs_per_cu, ms[5], moved / ms[5] * 1e-6, moved / ms[0] * 1e-6);
            }
        }
        run_rows<4, 1>(src, bytes, out, e0, e1);
        {
            float* wr; CHECK(hipMalloc(&wr, bytes / 15 + (1 << 20)));  // 4000 B per 63 488 B read: < bytes / 15
            run_rows<4, 1, 1>(src, bytes, out, e0, e1, wr);
            run_rows<16, 1, 1>(src, bytes, out, e0, e1, wr);
            printf("  (next two: the same write volume as full, aligned 1-KB / 4-KB bursts)\n");
            run_rows<16, 1, 2>(src, bytes, out, e0, e1, wr);
            run_rows<16, 1, 3>(src, bytes, out, e0, e1, wr);
            run_split(src, bytes, out, e0, e1, wr);
            CHECK(hipFree(wr));
        }
        run_rows<16, 1>(src, bytes, out, e0, e1);
        run_rows<1, 4>(src, bytes, out, e0, e1);
        run_rows<2, 2>(src, bytes, out, e0, e1);
        run_rows<4, 2>(src, bytes, out, e0, e1);
        run_rows<4, 4>(src, bytes, out, e0, e1);
        run_rows<8, 2>(src, bytes, out, e0, e1);
        run_rows<1, 16>(src, bytes, out, e0, e1);
        CHECK(hipFree(src));
    }
    return 0;
}
