// Micro-benchmark 4 (not shipped): round-trip time of one vector-memory operation as seen through vmcnt, one wave on
// an otherwise idle GPU and the same wave while other workgroups stream HBM.  Motivation: on gfx9-family parts loads
// and stores share the in-order vmcnt counter, so a prefetched load cannot be waited for before every OLDER store has
// been acknowledged.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_lat tools/ubench_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

// KIND 0: load dwordx4 (fresh lines)   1: store dwordx4, 64 B per quad into 16 rows 4000 B apart (like the decisions)
//      2: same store, nontemporal       3: store dwordx4, fully contiguous 1 KB per instruction
//      4: store to the same 1 KB every time
template <int KIND>
__global__ __launch_bounds__(64) void lat(float* buf, size_t stride_f, unsigned long long* cyc, int iters, float* sink) {
    const int lane = threadIdx.x;
    if (blockIdx.x != 0) {  // background traffic: stream a private 8-MB region repeatedly
        // 8 MB per workgroup, all inside the 12-GB allocation: 256 MB + 1024 x 8 MB < 9 GB
        const v4f* p = (const v4f*)(buf + (size_t)(64 << 20)) + (size_t)(blockIdx.x & 1023) * (512 << 10);
        float acc = 0;
        for (int r = 0; r < iters / 8; ++r)
            for (int i = lane; i < (512 << 10); i += 64) {
                v4f a = __builtin_nontemporal_load(p + i);
                acc += a.x;
            }
        if (acc == 1.2345f) sink[blockIdx.x] = acc;
        return;
    }
    unsigned long long total = 0, worst = 0;
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
        float* p;
        if (KIND == 0 || KIND == 3) p = buf + (size_t)it * 256 + lane * 4;
        else if (KIND == 4) p = buf + lane * 4;
        else p = buf + (size_t)(lane >> 2) * stride_f + it * 16 + (lane & 3) * 4;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (KIND == 0) {
            v4f a = *(volatile v4f*)p;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += a.x;
        } else {
            v4f o = {(float)it, 1.0f, 2.0f, 3.0f};
            if (KIND == 2) __builtin_nontemporal_store(o, (v4f*)p);
            else *(v4f*)p = o;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        total += t1 - t0;
        worst = worst > t1 - t0 ? worst : t1 - t0;
    }
    if (lane == 0) { cyc[0] = total; cyc[1] = worst; }
    if (acc == 1.2345f) sink[0] = acc;
}

int main() {
    float* buf; CHECK(hipMalloc(&buf, (size_t)12 << 30)); CHECK(hipMemset(buf, 0, (size_t)1 << 30));
    unsigned long long* cyc; CHECK(hipMalloc(&cyc, 64)); float* sink; CHECK(hipMalloc(&sink, 1 << 20));
    const char* names[] = {"load dwordx4 (new lines)", "store dwordx4 16 rows x 64 B", "same, nontemporal", "store 1 KB contiguous", "store same 1 KB"};
    for (int bg : {0, 1024}) {
        for (int kind = 0; kind < 5; ++kind) {
            const int iters = 2000;
            auto launch = [&](int k) {
                dim3 g(1 + bg), b(64);
                if (k == 0) hipLaunchKernelGGL(lat<0>, g, b, 0, 0, buf, (size_t)1000, cyc, iters, sink);
                if (k == 1) hipLaunchKernelGGL(lat<1>, g, b, 0, 0, buf, (size_t)1000, cyc, iters, sink);
                if (k == 2) hipLaunchKernelGGL(lat<2>, g, b, 0, 0, buf, (size_t)1000, cyc, iters, sink);
                if (k == 3) hipLaunchKernelGGL(lat<3>, g, b, 0, 0, buf, (size_t)1000, cyc, iters, sink);
                if (k == 4) hipLaunchKernelGGL(lat<4>, g, b, 0, 0, buf, (size_t)1000, cyc, iters, sink);
            };
            launch(kind); CHECK(hipDeviceSynchronize());
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0)); launch(kind); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[2]; CHECK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
            // s_memtime counts at 100 MHz on this part family (constant-rate REFCLK)
            printf("background WGs %4d  %-32s avg %.0f ticks (%.2f us @100MHz)  worst %llu ticks; kernel %.3f ms for %llu timed ticks\n", bg, names[kind],
                   (double)h[0] / iters, (double)h[0] / iters / 100.0, h[1], ms, h[0]);
        }
    }
    return 0;
}
