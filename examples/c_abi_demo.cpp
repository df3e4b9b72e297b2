// c_abi_demo.cpp -- uses libmvn_hip.so through include/mvn.h ONLY (no Python, no torch): the boundary a non-Python
// host would bind.  Reads a little-endian binary problem file, runs the classical Viterbi detector, the ViterbiNet
// detector and the error counters on the GPU, writes the results.  tests/test_gpu_parity.py::test_c_abi_demo builds the
// file, runs this program and checks the output against the oracle.
//
//   build: hipcc --offload-arch=gfx950 -O2 -I include examples/c_abi_demo.cpp -o examples/c_abi_demo \
//                -L meta-viterbinet_amd -lmvn_hip -Wl,-rpath,'$ORIGIN/../meta-viterbinet_amd'
//   file : int64 B,T,S | y[B*T] | priors[S] | W1[100] b1[100] W2[5000] b2[50] W3[S*50] b3[S] | tx[B*T]   (all fp32)
//   out  : va_dec[B*T] | vnet_dec[B*T] | int64 va_counters[4] | int64 vnet_counters[4]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mvn.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP: %s (line %d)\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define MVN_OK_(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "mvn: %s (line %d)\n", mvn_strerror(rc_), __LINE__); return 3; } } while (0)

static float *to_device(const std::vector<float> &h) {
    float *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(float)) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s problem.bin result.bin\n", argv[0]); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    int64_t dims[3];
    if (fread(dims, 8, 3, f) != 3) return 1;
    const int64_t B = dims[0];
    const int T = (int)dims[1], S = (int)dims[2];
    auto rd = [&](size_t n) { std::vector<float> v(n); if (fread(v.data(), 4, n, f) != n) exit(1); return v; };
    std::vector<float> y = rd(B * T), pri = rd(S), W1 = rd(100), b1 = rd(100), W2 = rd(5000), b2 = rd(50), W3 = rd(S * 50),
                       b3 = rd(S), tx = rd(B * T);
    fclose(f);

    int n_cu = 0;
    char arch[64];
    MVN_OK_(mvn_device_info(&n_cu, nullptr, arch, sizeof arch));
    printf("mvn ABI %d on %s (%d CUs)\n", mvn_version(), arch, n_cu);

    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    float *dy = to_device(y), *dp = to_device(pri), *dtx = to_device(tx);
    float *dW1 = to_device(W1), *db1 = to_device(b1), *dW2 = to_device(W2), *db2 = to_device(b2), *dW3 = to_device(W3),
          *db3 = to_device(b3);
    float *dva = nullptr, *dvn = nullptr;
    int64_t *dc = nullptr;
    void *ws = nullptr;
    const size_t ws_bytes = mvn_vnet_workspace_bytes(B, T, S);
    HIP_OK(hipMalloc(&dva, B * T * 4));
    HIP_OK(hipMalloc(&dvn, B * T * 4));
    HIP_OK(hipMalloc(&dc, 8 * 8));
    HIP_OK(hipMalloc(&ws, ws_bytes));
    HIP_OK(hipMemsetAsync(dc, 0, 64, st));
    MVN_OK_(mvn_va_decode_f32(dy, T, dp, 1, dva, T, nullptr, B, T, S, st));
    MVN_OK_(mvn_vnet_decode_f32(dy, T, dW1, db1, dW2, db2, dW3, db3, dvn, T, nullptr, nullptr, ws, ws_bytes, B, T, S, st));
    MVN_OK_(mvn_count_errors(dva, T, dtx, T, nullptr, B, T, dc, st));
    MVN_OK_(mvn_count_errors(dvn, T, dtx, T, nullptr, B, T, dc + 4, st));
    HIP_OK(hipStreamSynchronize(st));

    std::vector<float> va(B * T), vn(B * T);
    int64_t c[8];
    HIP_OK(hipMemcpy(va.data(), dva, B * T * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(vn.data(), dvn, B * T * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(c, dc, 64, hipMemcpyDeviceToHost));
    FILE *o = fopen(argv[2], "wb");
    if (!o) { perror("open out"); return 1; }
    fwrite(va.data(), 4, va.size(), o);
    fwrite(vn.data(), 4, vn.size(), o);
    fwrite(c, 8, 8, o);
    fclose(o);
    printf("VA  : bit errors %lld / %lld, frame errors %lld / %lld\n", (long long)c[0], (long long)c[1], (long long)c[2], (long long)c[3]);
    printf("VNET: bit errors %lld / %lld, frame errors %lld / %lld\n", (long long)c[4], (long long)c[5], (long long)c[6], (long long)c[7]);
    return 0;
}
