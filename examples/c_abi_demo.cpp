// c_abi_demo.cpp -- uses libmvn_hip.so through include/mvn.h ONLY (no Python, no torch): the boundary a non-Python
// host would bind.  Reads a little-endian binary problem file, runs the classical Viterbi detector, the ViterbiNet
// detector and the error counters on the GPU, writes the results.  tests/test_gpu_parity.py::test_c_abi_demo builds the
// file, runs this program and checks the output against the oracle.
//
//   build: hipcc --offload-arch=gfx950 -O2 -I include examples/c_abi_demo.cpp -o examples/c_abi_demo \
//                -L meta-viterbinet_amd -lmvn_hip -Wl,-rpath,'$ORIGIN/../meta-viterbinet_amd'
//   file : int64 B,T,S | y[B*T] | priors[S] | W1[100] b1[100] W2[5000] b2[50] W3[S*50] b3[S] | tx[B*T]   (all fp32)
//   out  : va_dec[B*T] | vnet_dec[B*T] | int64 va_counters[4] | int64 vnet_counters[4]
//          and, at 16 states: int32 nerr[R] | enc[R*Tb] | theta[2*P]  -- one by-word block step for the first R = 4 words
//          (mvn_vnet_byword_step_f32: Tb = 8 floor(T/8) symbols, RS with 2 parity bytes) and two trials of 5 full-word
//          training iterations in one launch sequence (mvn_vnet_online_train_trials_f32, descriptors built here), and the ViterbiNet
//          decision with survivor-path traceback (mvn_vnet_decode_surv_f32 + mvn_traceback_f32)
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
    if (S != 16 || B < 4 || T < 32) return 0;

    // ---- one block step of the by-word evaluation for R words, then two independent trials of online training
    const int R = 4, Tb = (T / 8) * 8, nsym = 2, K = Tb - 8 * nsym, P = 250 + 50 * 100 + 51 * S;
    float *denc = nullptr, *dtheta = nullptr, *dm = nullptr, *dv = nullptr;
    int32_t *dlab = nullptr, *dnerr = nullptr, *dstatus = nullptr;
    HIP_OK(hipMalloc(&denc, (size_t)R * Tb * 4));
    HIP_OK(hipMalloc(&dlab, (size_t)R * Tb * 4));
    HIP_OK(hipMalloc(&dnerr, R * 4));
    HIP_OK(hipMalloc(&dstatus, 2 * 4));
    HIP_OK(hipMemsetAsync(dstatus, 0, 8, st));
    // (the message of a word = the first K transmitted bits; the words' own row stride T is passed as the leading dimension)
    MVN_OK_(mvn_vnet_byword_step_f32(dy, T, dtx, T, dW1, db1, dW2, db2, dW3, db3, nullptr, nullptr, Tb, nullptr, K, denc, Tb, nullptr,
                                     Tb, dlab, Tb, dnerr, R, Tb, nsym, 0, S, st));
    std::vector<float> theta0;  // flat parameters() order, one copy per trial
    for (int t = 0; t < 2; ++t)
        for (const std::vector<float> *a : {&W1, &b1, &W2, &b2, &W3, &b3}) theta0.insert(theta0.end(), a->begin(), a->end());
    dtheta = to_device(theta0);
    HIP_OK(hipMalloc(&dm, 2 * P * 4));
    HIP_OK(hipMalloc(&dv, 2 * P * 4));
    HIP_OK(hipMemsetAsync(dm, 0, 2 * P * 4, st));
    HIP_OK(hipMemsetAsync(dv, 0, 2 * P * 4, st));
    const int off[6] = {0, 100, 200, 5200, 5250, 5250 + 50 * S};
    mvn_train_trial_t trials[2] = {};
    for (int t = 0; t < 2; ++t) {
        trials[t].y = dy + (size_t)t * T;             // word t, its first Tb symbols
        trials[t].labels = dlab + (size_t)t * Tb;     // the trellis states the step kernel wrote for it
        for (int a = 0; a < 6; ++a) trials[t].w_in[a] = trials[t].w_out[a] = dtheta + (size_t)t * P + off[a];
        trials[t].adam_m = dm + (size_t)t * P;
        trials[t].adam_v = dv + (size_t)t * P;
        trials[t].status = dstatus + t;
        trials[t].b1pow = trials[t].b2pow = 1.0;      // no Adam step taken yet
        trials[t].n = 5;
    }
    mvn_train_trial_t *dtrials = nullptr;
    HIP_OK(hipMalloc(&dtrials, sizeof trials));
    HIP_OK(hipMemcpyAsync(dtrials, trials, sizeof trials, hipMemcpyHostToDevice, st));
    const size_t tws_bytes = mvn_vnet_train_trials_workspace_bytes(S, Tb, 1, 2);
    void *tws = nullptr;
    HIP_OK(hipMalloc(&tws, tws_bytes ? tws_bytes : 16));
    MVN_OK_(mvn_vnet_online_train_trials_f32(dtrials, 2, Tb, 0, 1e-3f, 0.9f, 0.999f, 1e-8f, S, tws, tws_bytes, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<int32_t> nerr(R), status(2);
    std::vector<float> enc((size_t)R * Tb), theta(2 * P);
    HIP_OK(hipMemcpy(nerr.data(), dnerr, R * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(status.data(), dstatus, 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(enc.data(), denc, enc.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(theta.data(), dtheta, theta.size() * 4, hipMemcpyDeviceToHost));
    if (status[0] || status[1]) { fprintf(stderr, "mvn: %s\n", mvn_strerror(MVN_E_BARRIER)); return 4; }
    o = fopen(argv[2], "ab");
    if (!o) { perror("open out"); return 1; }
    fwrite(nerr.data(), 4, nerr.size(), o);
    fwrite(enc.data(), 4, enc.size(), o);
    fwrite(theta.data(), 4, theta.size(), o);
    fclose(o);
    printf("by-word step: bit errors of the first %d words %d %d %d %d; 2 trials x 5 training iterations done\n", R, nerr[0], nerr[1],
           nerr[2], nerr[3]);

    // ---- ViterbiNet with survivor-path traceback (round 5): the sweep's survivors, then the maximum-likelihood path
    uint8_t *dsurv = nullptr;
    float *dfm = nullptr, *dbits = nullptr, *dlg = nullptr;
    const size_t lg_bytes = (size_t)B * T * S * 4;
    HIP_OK(hipMalloc(&dsurv, mvn_survivor_bytes(B, T, S)));
    HIP_OK(hipMalloc(&dfm, (size_t)B * S * 4));
    HIP_OK(hipMalloc(&dbits, (size_t)B * T * 4));
    HIP_OK(hipMalloc(&dlg, lg_bytes));
    MVN_OK_(mvn_vnet_decode_surv_f32(dy, T, dW1, db1, dW2, db2, dW3, db3, dvn, T, dfm, dsurv, dlg, lg_bytes, B, T, S, st));
    MVN_OK_(mvn_traceback_f32(dsurv, dfm, dbits, T, nullptr, B, T, S, st));
    HIP_OK(hipMemsetAsync(dc, 0, 64, st));
    MVN_OK_(mvn_count_errors(dbits, T, dtx, T, nullptr, B, T, dc, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<float> path((size_t)B * T);
    HIP_OK(hipMemcpy(path.data(), dbits, path.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(c, dc, 32, hipMemcpyDeviceToHost));
    o = fopen(argv[2], "ab");
    if (!o) { perror("open out"); return 1; }
    fwrite(path.data(), 4, path.size(), o);
    fclose(o);
    printf("VNET with traceback: bit errors %lld / %lld\n", (long long)c[0], (long long)c[1]);
    return 0;
}
