// Assembly-level tuning harness (not shipped): loads a gfx950 code object holding vnet16_fusedn_kernel<false, 2>
// (hipModuleLoad), runs it on BASELINE configs[1]'s shape and prints the median launch time; with a second code
// object it also checks that both produce identical decisions (so an edited instruction stream can be validated
// against the compiler's).
//   fused_asm_bench edited.co [reference.co] [B]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
static const char* kSym = "_ZN12_GLOBAL__N_120vnet16_fusedn_kernelILb0ELi2EEEvPKflS2_S2_S2_S2_S2_S2_PflS3_S3_liS2_liPKhPyl";

struct Args {
    const float* y; long y_ld; const float *W1, *b1, *W2, *b2, *W3, *b3; float* dec; long dec_ld; float* logits; float* fm;
    long B; int T; const float* tx; long tx_ld; int K; const unsigned char* mask; unsigned long long* counters;
};

static double run(const char* path, Args a, int reps, std::vector<float>* out) {
    hipModule_t mod; hipFunction_t fn;
    CHECK(hipModuleLoad(&mod, path));
    CHECK(hipModuleGetFunction(&fn, mod, kSym));
    void* params[] = {&a.y, &a.y_ld, &a.W1, &a.b1, &a.W2, &a.b2, &a.W3, &a.b3, &a.dec, &a.dec_ld, &a.logits, &a.fm,
                      &a.B, &a.T, &a.tx, &a.tx_ld, &a.K, &a.mask, &a.counters};
    const unsigned grid = (unsigned)((a.B + 7) / 8);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<double> ts;
    for (int r = 0; r < reps + 3; ++r) {
        CHECK(hipEventRecord(e0));
        for (int k = 0; k < 5; ++k) CHECK(hipModuleLaunchKernel(fn, grid, 1, 1, 512, 1, 1, 0, 0, params, nullptr));
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) ts.push_back(ms / 5);
    }
    std::sort(ts.begin(), ts.end());
    if (out) { out->resize((size_t)a.B * a.T); CHECK(hipMemcpy(out->data(), a.dec, out->size() * 4, hipMemcpyDeviceToHost)); }
    CHECK(hipModuleUnload(mod));
    return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: %s edited.co [reference.co] [B]\n", argv[0]); return 2; }
    const long B = argc > 3 ? atol(argv[3]) : 10000; const int T = 1000;
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f); std::uniform_real_distribution<float> ud(-1.f, 1.f);
    std::vector<float> y((size_t)B * T), W1(100), b1(100), W2(5000), b2(50), W3(800), b3(16);
    for (auto& v : y) v = 1.3f * nd(rng);
    for (auto& v : W1) v = ud(rng); for (auto& v : b1) v = ud(rng);
    for (auto& v : W2) v = 0.1f * ud(rng); for (auto& v : b2) v = 0.1f * ud(rng);
    for (auto& v : W3) v = 0.14f * ud(rng); for (auto& v : b3) v = 0.14f * ud(rng);
    auto up = [](const std::vector<float>& h) { float* d; CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); return d; };
    Args a{}; a.y = up(y); a.y_ld = T; a.W1 = up(W1); a.b1 = up(b1); a.W2 = up(W2); a.b2 = up(b2); a.W3 = up(W3); a.b3 = up(b3);
    CHECK(hipMalloc(&a.dec, (size_t)B * T * 4)); CHECK(hipMemset(a.dec, 0, (size_t)B * T * 4)); a.dec_ld = T; a.B = B; a.T = T;
    std::vector<float> d0, d1;
    const double ms = run(argv[1], a, 9, &d0);
    printf("%-40s B=%ld  median %.4f ms  = %.1f cycles/symbol/SIMD @2.4GHz", argv[1], B, ms, ms * 1e-3 * 2.4e9 * 1024 / ((double)B * T));
    if (argc > 2 && strcmp(argv[2], "-") != 0) {
        CHECK(hipMemset(a.dec, 0, (size_t)B * T * 4));
        const double ms1 = run(argv[2], a, 5, &d1);
        size_t diff = 0; for (size_t i = 0; i < d0.size(); ++i) diff += d0[i] != d1[i];
        printf("   | reference %.4f ms, decisions differing: %zu of %zu", ms1, diff, d0.size());
    }
    printf("\n");
    return 0;
}
