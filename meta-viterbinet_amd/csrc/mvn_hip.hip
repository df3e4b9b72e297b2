// mvn_hip.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI (include/mvn.h) for the
// Viterbi / ViterbiNet 'val' hot path of tomerraviv95/meta-viterbinet.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC  (see __graft_entry__.build)
// -ffp-contract=off is part of the arithmetic contract: every reference fp32 op rounds once.
//
// Kernels
//   sweep_kernel<S,MODE>   state-per-lane ACS recurrence (trellis_utils.py:16-30 inside the
//                          T loop of va_detector.py:89-97); MODE picks the branch-cost source:
//                          materialised costs, negated logits, or fused VA costs from y.
//   mlp_kernel<NT3>        ViterbiNet MLP (vnet_detector.py:27-33,49) on f32 MFMA 16x16x4:
//                          exact k-ordered fmaf chains == torch-CPU sgemm result order.
//   count_errors_kernel    metrics.py:7-17 as int64 counters.
// and, in the .inc files included below (each starts with its own description):
//   vnet16_fusedn.inc / vnet16_fused.inc   fused ViterbiNet detector at 16 states (MLP on MFMA + in-place DPP sweep)
//   sweep16_rows / _lds / _quad.inc        16-state sweeps over materialised costs (register prefetch, LDS-DMA)
//   sweep_inplace.inc                      the same sweep for any other S >= 4 (in-place recurrence + LDS-DMA)
//   va16_quad.inc, va_inplace.inc          fused classical Viterbi (16 states; any S >= 4)
//   rs_codec.inc                           Reed-Solomon encode/decode
//   online_train.inc, maml_train.inc       one-launch online training and MAML meta-learning steps
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <type_traits>

#include "../../include/mvn.h"

namespace {

#ifndef MVN_FUSEDN_DEFAULT
#define MVN_FUSEDN_DEFAULT 2
#endif
constexpr int kFusedNDefault = MVN_FUSEDN_DEFAULT;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kH1 = 100;  // vnet_detector.py:7
constexpr int kH2 = 50;   // vnet_detector.py:8
constexpr int kK2Steps = kH1 / 4;        // 25 MFMA k-steps for layer 2
constexpr int kK3Steps = (kH2 + 3) / 4;  // 13 MFMA k-steps for layer 3 (k 50,51 zero-padded)

// -------------------------------------------------------------------------------------------
// Deterministic sigmoid: 1/(1+expf_u10(d)), d = -z.  Same operation sequence as
// oracle/mvn_oracle.c:mvn_oracle_expf_u10 (SLEEF 1.0-ULP expf as ATen's vectorised sigmoid
// evaluates it).  v_ldexp_f32 replaces SLEEF's two-step scaling: the two differ only when the
// result is subnormal, and then 1+e == 1 either way.
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_from_neg(float d) {
    float dc = fminf(fmaxf(d, -128.0f), 128.0f);
    dc = d != d ? d : dc;  // NaN in -> NaN out, like the reference (the clamp alone would swallow it)
    float t = dc * 1.442695040888963407359924681001892137426645954152985934135449406931f;
    float qf = __builtin_rintf(t);
    int q = (int)qf;
    float s = __builtin_fmaf(qf, -0.693145751953125f, dc);
    s = __builtin_fmaf(qf, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = __builtin_fmaf(u, s, 0.00139304355252534151077271f);
    u = __builtin_fmaf(u, s, 0.00833336077630519866943359f);
    u = __builtin_fmaf(u, s, 0.0416664853692054748535156f);
    u = __builtin_fmaf(u, s, 0.166666671633720397949219f);
    u = __builtin_fmaf(u, s, 0.5f);
    u = 1.0f + __builtin_fmaf(s * s, u, s);
    float e = ldexpf(u, q);
    return 1.0f / (1.0f + e);  // IEEE division (hipcc default: correctly rounded)
}

// va_detector.py:64-68, four separately rounded ops (three instructions: see below).
__device__ __forceinline__ float va_cost(float y, float prior) {
    float d = y - prior;
    float sq = d * d;
    // sq * 0.5f is exact (or underflows far below half an ulp of the constant), so one fma rounds like the two ops
    return __builtin_fmaf(sq, 0.5f, -0.91893853320467274178f);
}

// -------------------------------------------------------------------------------------------
// Generic sweep: one lane per state (S<=64: 64/S blocks per wave; S=128/256: 2/4 states per
// lane, one block per wave).  Path metrics live in VGPRs; the predecessor shuffle
// out[s] = min(a[2s%S], a[(2s+1)%S]) goes through a per-wave LDS row (ds_write_b32 +
// ds_read_b64); the running argmin is a lexicographic (value,index) xor-butterfly.
// The ACS minimum here is torch.min's (NaN if either candidate is NaN, trellis_utils.py:30), not minNum: this kernel is
// also the GUARD of the specialised VA / two-kernel ViterbiNet routes, whose v_min_f32 stages drop a NaN that sits in one
// candidate only.  That can only matter when a state prior / a weight is non-finite or absurdly large, so the guard launch
// (GUARD = true, right after the fast kernel on the same stream) scans exactly those -- the priors of its own blocks, or
// the six weight arrays -- and returns at once unless it finds one, in which case it decodes its blocks again and
// overwrites the fast kernel's decisions and final metrics.
// -------------------------------------------------------------------------------------------
enum { MODE_COST = 0, MODE_NEGLOGIT = 1, MODE_VA = 2 };

constexpr float kStrictMinBoundG = 1e14f;  // = kStrictMinBound (vnet16_fused.inc, included below)
struct GuardWeights {  // the ViterbiNet weights a MODE_NEGLOGIT guard scans (lengths in floats); unused otherwise
    const float *w[6];
    int n[6];
};
// torch.min(dim) over (a, b) in index order: the first NaN wins, else the smaller, ties to a (oracle: min2_torch)
__device__ __forceinline__ float min2_torch_dev(float a, float b) { return a != a ? a : (!(b >= a) ? b : a); }

template <int S>
struct SweepCfg {
    static constexpr int LPB = S < 64 ? S : 64;  // lanes per block
    static constexpr int R = S / LPB;            // states per lane
    static constexpr int G = 64 / LPB;           // blocks per wave
    static constexpr int TC = S < 8 ? S : 8;     // steps per chunk (decisions stored TC at a time)
};

constexpr int kSweepWaves = 4;

template <int S, int MODE, bool GUARD = false>
__global__ __launch_bounds__(64 * kSweepWaves) void sweep_kernel(
    const float *__restrict__ src, int64_t src_ld, const float *__restrict__ priors, int64_t Bp,
    float *__restrict__ dec, int64_t dec_ld, float *__restrict__ final_metric, int64_t B, int T, const GuardWeights gw) {
    using C = SweepCfg<S>;
    constexpr int LPB = C::LPB, R = C::R, G = C::G, TC = C::TC;
    __shared__ float lds[kSweepWaves][64 * R];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / LPB;
    const int sl = lane % LPB;
    const int64_t b = ((int64_t)blockIdx.x * kSweepWaves + wave) * G + g;
    const bool active = b < B;
    const int64_t bc = active ? b : B - 1;
    float *row = &lds[wave][g * S];

    float m[R], pr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        m[r] = 0.0f;  // va_detector.py:84
        pr[r] = 0.0f;
        if (MODE == MODE_VA) pr[r] = priors[(bc % Bp) * S + sl + LPB * r];
    }
    if (GUARD) {  // wave-uniform: nothing to repair unless a prior / weight can produce partially-NaN branch costs
        bool odd = false;
        if (MODE == MODE_VA) {
#pragma unroll
            for (int r = 0; r < R; ++r) odd |= !(fabsf(pr[r]) < kStrictMinBoundG);
        } else {
#pragma unroll
            for (int a = 0; a < 6; ++a)
                for (int e = lane; e < gw.n[a]; e += 64) odd |= !(fabsf(gw.w[a][e]) < kStrictMinBoundG);
        }
        if (!__any(odd)) return;
    }

    const float *base = (MODE == MODE_VA) ? src + bc * src_ld : src + bc * (int64_t)T * S;
    auto load_chunk = [&](int t0, float (&c)[TC][R]) {
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            int t = t0 + i < T ? t0 + i : T - 1;
            if (MODE == MODE_VA) {
                float yv = base[t];
#pragma unroll
                for (int r = 0; r < R; ++r) c[i][r] = yv;
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) c[i][r] = base[(int64_t)t * S + sl + LPB * r];
            }
        }
    };

    float cur[TC][R], nxt[TC][R];
    load_chunk(0, nxt);
    float mydec = 0.0f;
    for (int t0 = 0; t0 < T; t0 += TC) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int r = 0; r < R; ++r) cur[i][r] = nxt[i][r];
        if (t0 + TC < T) load_chunk(t0 + TC, nxt);
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            if (t0 + i < T) {  // wave-uniform
                // ---- decision: argmin over states, first minimal index (torch.argmin) % 2
                float bv = m[0];
                int bi = sl;
#pragma unroll
                for (int r = 1; r < R; ++r)
                    if (m[r] < bv || (m[r] != m[r] && bv == bv)) {  // strict: lower index wins ties; first NaN wins
                        bv = m[r];
                        bi = sl + LPB * r;
                    }
#pragma unroll
                for (int off = 1; off < LPB; off <<= 1) {
                    float ov = __shfl_xor(bv, off);
                    int oi = __shfl_xor(bi, off);
                    // torch.argmin order: NaN sorts before everything, ties (and NaN vs NaN) go to the lower index
                    const bool onan = ov != ov, bnan = bv != bv;
                    bool take = onan ? (!bnan || oi < bi) : (!bnan && ((ov < bv) || (ov == bv && oi < bi)));
                    bv = take ? ov : bv;
                    bi = take ? oi : bi;
                }
                if (sl == i) mydec = (float)(bi & 1);
                // ---- ACS stage
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float c = cur[i][r];
                    if (MODE == MODE_VA) c = va_cost(c, pr[r]);
                    if (MODE == MODE_NEGLOGIT) c = -c;
                    row[sl + LPB * r] = m[r] + c;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    int p0 = (2 * (sl + LPB * r)) % S;
                    float2 v = *reinterpret_cast<const float2 *>(&row[p0]);
                    m[r] = min2_torch_dev(v.x, v.y);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        if (active && sl < TC && t0 + sl < T) dec[b * dec_ld + t0 + sl] = mydec;
    }
    if (active && final_metric) {
#pragma unroll
        for (int r = 0; r < R; ++r) final_metric[b * S + sl + LPB * r] = m[r];
    }
}

template <int MODE, bool GUARD = false>
int launch_sweep(const float *src, int64_t src_ld, const float *priors, int64_t Bp, float *dec,
                 int64_t dec_ld, float *final_metric, int64_t B, int T, int S, hipStream_t st,
                 const GuardWeights gw = GuardWeights{}) {
    if (B == 0 || T == 0) return MVN_OK;
#define MVN_SWEEP_CASE(SS)                                                                       \
    case SS: {                                                                                   \
        constexpr int per_wg = kSweepWaves * SweepCfg<SS>::G;                                    \
        int64_t grid = (B + per_wg - 1) / per_wg;                                                \
        hipLaunchKernelGGL((sweep_kernel<SS, MODE, GUARD>), dim3((unsigned)grid), dim3(64 * kSweepWaves), \
                           0, st, src, src_ld, priors, Bp, dec, dec_ld, final_metric, B, T, gw);  \
        break;                                                                                   \
    }
    switch (S) {
        MVN_SWEEP_CASE(2)
        MVN_SWEEP_CASE(4)
        MVN_SWEEP_CASE(8)
        MVN_SWEEP_CASE(16)
        MVN_SWEEP_CASE(32)
        MVN_SWEEP_CASE(64)
        MVN_SWEEP_CASE(128)
        MVN_SWEEP_CASE(256)
        default:
            return MVN_E_STATES;
    }
#undef MVN_SWEEP_CASE
    return (int)hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// ViterbiNet MLP on f32 MFMA (v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] B[4x16], exact
// k-ordered fmaf chain).  One wave owns a tile of 16 symbols:
//   layer 1+sigmoid : lane (j=l&15, q=l>>4) evaluates hidden-1 units k = 4i+q, i<25, for its
//                     symbol j straight into the B-operand layout (B[k=q][col=j]);
//   layer 2         : A = W2 rows (hidden-2 units, 50 padded to 64 = 4 row tiles), 100 VGPRs of
//                     weights held for the kernel's lifetime; 100 MFMAs per symbol tile;
//   bias + ReLU     : in the D layout (row = 4q+r);
//   (q,r) transpose : v_permlane32_swap + v_permlane16_swap turn D rows into the next
//                     B operand in natural k order (torch's accumulation order);
//   layer 3         : A = W3 rows (states, ceil(S/16) tiles) read from LDS; 13 MFMAs per tile.
// -------------------------------------------------------------------------------------------
constexpr int kMlpWaves = 4;

template <int NT3>
__global__ __launch_bounds__(64 * kMlpWaves, 2) void mlp_kernel(
    const float *__restrict__ y, int64_t y_ld, int T, int64_t N, const float *__restrict__ W1,
    const float *__restrict__ b1, const float *__restrict__ W2, const float *__restrict__ b2,
    const float *__restrict__ W3, const float *__restrict__ b3, float *__restrict__ out, int S) {
    __shared__ float ldsA3[NT3 * kK3Steps * 64];
    __shared__ float ldsB3[NT3 * 16];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 15;
    const int q = lane >> 4;

    // stage W3 in A-operand order: element (j3,i3,lane) = W3[16*j3 + (lane&15)][4*i3 + (lane>>4)]
    for (int e = threadIdx.x; e < NT3 * kK3Steps * 64; e += blockDim.x) {
        int l = e & 63, i3 = (e >> 6) % kK3Steps, j3 = (e >> 6) / kK3Steps;
        int st = 16 * j3 + (l & 15), k = 4 * i3 + (l >> 4);
        ldsA3[e] = (st < S && k < kH2) ? W3[st * kH2 + k] : 0.0f;
    }
    for (int e = threadIdx.x; e < NT3 * 16; e += blockDim.x) ldsB3[e] = e < S ? b3[e] : 0.0f;

    float nw1[kK2Steps], nb1[kK2Steps], a2[4][kK2Steps], b2v[4][4];
#pragma unroll
    for (int i = 0; i < kK2Steps; ++i) {
        nw1[i] = -W1[4 * i + q];
        nb1[i] = -b1[4 * i + q];
    }
#pragma unroll
    for (int tau = 0; tau < 4; ++tau) {
        int unit = 16 * tau + j;
#pragma unroll
        for (int i = 0; i < kK2Steps; ++i) a2[tau][i] = unit < kH2 ? W2[unit * kH1 + 4 * i + q] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int u = 16 * tau + 4 * q + r;
            b2v[tau][r] = u < kH2 ? b2[u] : 0.0f;
        }
    }
    __syncthreads();

    const int64_t ntiles = (N + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * kMlpWaves + wave; tile < ntiles;
         tile += (int64_t)gridDim.x * kMlpWaves) {
        const int64_t n = tile * 16 + j;
        const int64_t nc = n < N ? n : N - 1;
        const float yj = y[(nc / T) * y_ld + (nc % T)];

        f32x4 acc[4];
#pragma unroll
        for (int tau = 0; tau < 4; ++tau) acc[tau] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < kK2Steps; ++i) {
            // -(fma(y,w1,b1)) == fma(y,-w1,-b1): the argument of exp in torch's sigmoid
            float h = sigmoid_from_neg(__builtin_fmaf(yj, nw1[i], nb1[i]));
#pragma unroll
            for (int tau = 0; tau < 4; ++tau)
                acc[tau] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[tau][i], h, acc[tau], 0, 0, 0);
        }

        // bias + ReLU in D layout, then (q,r) transpose into B-operand order: bop[4*tau+r'] at
        // lane (j,q) = h2[unit 16*tau + 4*r' + q][symbol j]
        float bop[16];
#pragma unroll
        for (int tau = 0; tau < 4; ++tau) {
            unsigned u[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float z = acc[tau][r] + b2v[tau][r];
                u[r] = __float_as_uint(z < 0.0f ? 0.0f : z)  /* relu; NaN propagates like torch's */;
            }
            u32x2 p;
            p = __builtin_amdgcn_permlane32_swap(u[0], u[2], false, false);
            u[0] = p[0];
            u[2] = p[1];
            p = __builtin_amdgcn_permlane32_swap(u[1], u[3], false, false);
            u[1] = p[0];
            u[3] = p[1];
            p = __builtin_amdgcn_permlane16_swap(u[0], u[1], false, false);
            u[0] = p[0];
            u[1] = p[1];
            p = __builtin_amdgcn_permlane16_swap(u[2], u[3], false, false);
            u[2] = p[0];
            u[3] = p[1];
#pragma unroll
            for (int r = 0; r < 4; ++r) bop[4 * tau + r] = __uint_as_float(u[r]);
        }

        auto out_tile = [&](int j3) {
            f32x4 acc3 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i3 = 0; i3 < kK3Steps; ++i3)
                acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(ldsA3[(j3 * kK3Steps + i3) * 64 + lane], bop[i3],
                                                            acc3, 0, 0, 0);
            const int st0 = 16 * j3 + 4 * q;
            if (n < N) {
                float *o = out + n * S + st0;
                if (st0 + 3 < S) {
                    float4 v;
                    v.x = acc3[0] + ldsB3[st0 + 0];
                    v.y = acc3[1] + ldsB3[st0 + 1];
                    v.z = acc3[2] + ldsB3[st0 + 2];
                    v.w = acc3[3] + ldsB3[st0 + 3];
                    *reinterpret_cast<float4 *>(o) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (st0 + r < S) o[r] = acc3[r] + ldsB3[st0 + r];
                }
            }
        };
        if constexpr (NT3 <= 2) {
#pragma unroll
            for (int j3 = 0; j3 < NT3; ++j3) out_tile(j3);
        } else {
#pragma unroll 1
            for (int j3 = 0; j3 < NT3; ++j3) out_tile(j3);
        }
    }
}

int launch_mlp(const float *y, int64_t y_ld, int T, int64_t N, const float *W1, const float *b1,
               const float *W2, const float *b2, const float *W3, const float *b3, float *out, int S,
               hipStream_t st) {
    if (N == 0) return MVN_OK;
    const int64_t ntiles = (N + 15) / 16;
    int64_t want = (ntiles + kMlpWaves - 1) / kMlpWaves;
    const int64_t cap = 256 * 2 * 4;  // 2 workgroups per CU resident, a few rounds of slack
    unsigned grid = (unsigned)(want < cap ? want : cap);
    const int nt3 = (S + 15) / 16;
#define MVN_MLP_CASE(NT)                                                                          \
    case NT:                                                                                      \
        hipLaunchKernelGGL((mlp_kernel<NT>), dim3(grid), dim3(64 * kMlpWaves), 0, st, y, y_ld, T, N, \
                           W1, b1, W2, b2, W3, b3, out, S);                                       \
        break;
    switch (nt3) {
        MVN_MLP_CASE(1)
        MVN_MLP_CASE(2)
        MVN_MLP_CASE(4)
        MVN_MLP_CASE(8)
        MVN_MLP_CASE(16)
        default:
            return MVN_E_STATES;
    }
#undef MVN_MLP_CASE
    return (int)hipGetLastError();
}

int current_device_cus();  // CUs of the current device, cached per device (defined below the kernels)
int ensure_dynamic_lds(const void *fn, size_t bytes);  // raises a kernel's dynamic-LDS limit once per function and device (below)

#include "sweep_surv.inc"

#include "vnet16_fused.inc"
#include "vnet16_fusedn.inc"
#include "vnet16_dealt.inc"
#include "vnet16_coop.inc"
#include "va16_tile.inc"
#include "sweep16_rows.inc"
#include "sweep16_lds.inc"
#include "sweep16_quad.inc"
#include "va16_quad.inc"
#include "va_inplace.inc"
#include "va256_wave.inc"
#include "sweep_inplace.inc"
#include "vnet_fused_ip.inc"
#include "rs_codec.inc"
#include "byword_step.inc"
#include "online_train.inc"
#include "maml_train.inc"
#include "train_groups.inc"
#include "word_gen.inc"
#include "va_montecarlo.inc"

// -------------------------------------------------------------------------------------------
// metrics.py:7-17 as integer counters.  One wave per row, block-level reduction, one atomic
// per workgroup per counter.
// -------------------------------------------------------------------------------------------
constexpr int kCountWaves = 16;
__global__ __launch_bounds__(64 * kCountWaves) void count_errors_kernel(const float *__restrict__ dec, int64_t dec_ld,
                                                           const float *__restrict__ tx, int64_t tx_ld,
                                                           const int64_t *__restrict__ rows, int64_t n_rows,
                                                           int K, unsigned long long *counters) {
    __shared__ unsigned long long s_be, s_fe;
    if (threadIdx.x == 0) {
        s_be = 0;
        s_fe = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    unsigned long long be = 0, fe = 0;
    for (int64_t i = (int64_t)blockIdx.x * kCountWaves + wave; i < n_rows; i += (int64_t)gridDim.x * kCountWaves) {
        const int64_t r = rows ? rows[i] : i;
        int e = 0;
        const float *dr = dec + r * dec_ld, *tr = tx + r * tx_ld;
        if ((((uintptr_t)dr | (uintptr_t)tr) & 15) == 0) {  // 16-B aligned rows: 4 symbols per load
            const int K4 = K >> 2;
            // Rows of 1000 symbols start 0 / 32 / 64 / 96 bytes into a 128-B line.  A wave's load instruction then ends inside
            // a line and the next one starts in it, and the memory side saw that line requested twice (761 k read requests
            // per launch for 625 k lines, profiles/traffic.json of round 2).  When both rows sit at the same offset the loads
            // are issued from the line boundary below the row instead: every instruction covers whole lines, the lanes in
            // front of the row are masked.
            const int shift = (int)(((uintptr_t)dr & 127) >> 4);  // float4s between the line boundary and the row
            const bool same = (((uintptr_t)dr ^ (uintptr_t)tr) & 127) == 0;
            const int lo = same ? shift : 0, hi = lo + K4;
            const float4 *d4 = reinterpret_cast<const float4 *>(dr) - lo, *t4 = reinterpret_cast<const float4 *>(tr) - lo;
            for (int k0 = 0; k0 < hi; k0 += 256) {  // 8 loads in flight per lane (a 1000-symbol row is one trip)
                float4 a[4], c[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + 64 * j + lane;
                    const bool in = k >= lo && k < hi;
                    a[j] = in ? d4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                    c[j] = in ? t4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    e += ((long long)a[j].x != (long long)c[j].x) + ((long long)a[j].y != (long long)c[j].y) +
                         ((long long)a[j].z != (long long)c[j].z) + ((long long)a[j].w != (long long)c[j].w);  // .long() then eq
            }
            for (int k = 4 * K4 + lane; k < K; k += 64) e += ((long long)dr[k] != (long long)tr[k]) ? 1 : 0;
        } else {
            for (int k = lane; k < K; k += 64) e += ((long long)dr[k] != (long long)tr[k]) ? 1 : 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
        be += (unsigned long long)e;
        fe += e ? 1ull : 0ull;
    }
    if (lane == 0) {
        atomicAdd(&s_be, be);
        atomicAdd(&s_fe, fe);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_be) atomicAdd(&counters[0], s_be);
        if (s_fe) atomicAdd(&counters[2], s_fe);
        if (blockIdx.x == 0) {
            atomicAdd(&counters[1], (unsigned long long)n_rows * (unsigned long long)K);
            atomicAdd(&counters[3], (unsigned long long)n_rows);
        }
    }
}

// trellis_utils.py:16-30, one stage, one thread per (block,state).
__global__ __launch_bounds__(256) void acs_block_kernel(const float *__restrict__ in_prob,
                                                        const float *__restrict__ llrs, float *__restrict__ out,
                                                        long long *__restrict__ argmin_j, int64_t B, int S) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * S) return;
    const int64_t b = e / S;
    const int s = (int)(e % S);
    const int p0 = (2 * s) % S, p1 = (2 * s + 1) % S;
    const float v0 = in_prob[b * S + p0] + llrs[b * S + p0];
    const float v1 = in_prob[b * S + p1] + llrs[b * S + p1];
    const bool second = !(v1 >= v0) && v0 == v0;  // torch.min(dim): first minimal index wins, NaN propagates (first NaN wins)
    out[e] = second ? v1 : v0;
    if (argmin_j) argmin_j[e] = second ? 1 : 0;
}

// -------------------------------------------------------------------------------------------
// ISI-AWGN channel: zero padding by L (channel_dataset.py:71), BPSK 1-2c (modulator.py:12),
// anti-causal L-tap convolution + scaled white noise (channel.py:23-35), all in float64 like the reference's
// NumPy code, stored fp32 like channel_dataset.py:103.  One thread per received sample.
// -------------------------------------------------------------------------------------------
template <typename NoiseT>
__global__ __launch_bounds__(256) void isi_awgn_kernel(const float *__restrict__ bits, int64_t ld_bits, int K,
                                                       const NoiseT *__restrict__ noise, const double *__restrict__ h,
                                                       int64_t Bh, double sigma, float *__restrict__ y, int64_t y_ld,
                                                       int64_t B, int T, int L) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * T) return;
    const int64_t b = e / T;
    const int t = (int)(e % T);
    const double *hb = h + (b % Bh) * L;
    double acc = 0.0;
    for (int i = 0; i < L; ++i) {  // np.dot(h[:, ::-1], blockwise_s): row i of blockwise_s is s[t+i]
        const int tt = t + i;
        const double c = tt < K ? (double)bits[b * ld_bits + tt] : 0.0;
        acc += hb[L - 1 - i] * (1.0 - 2.0 * c);
    }
    if (noise) acc += sigma * (double)noise[b * T + t];
    y[b * y_ld + t] = (float)acc;
}

// The MVN_* switches (A/B runs, cross-checks in the tests).  Only the first character of a value matters.
// They are read from the environment ONCE per process -- the by-word evaluation calls into the library every few
// microseconds -- and again when the caller asks (mvn_reload_switches: the test-suite flips them between calls).
enum Switch { SW_UNFUSED, SW_COOP, SW_FUSEDN, SW_GENERIC_SWEEP, SW_VA256, SW_VA_INPLACE, SW_VA16, SW_SWEEP_INPLACE, SW_SWEEP16,
              SW_TRAIN_GROUPS, SW_TRAIN_PAIR, SW_FUSED_IP, SW_TRAIN_XCD, SW_DEALT, SW_COUNT };
const char *const kSwitchNames[SW_COUNT] = {"MVN_UNFUSED", "MVN_COOP", "MVN_FUSEDN", "MVN_GENERIC_SWEEP", "MVN_VA256",
                                            "MVN_VA_INPLACE", "MVN_VA16", "MVN_SWEEP_INPLACE", "MVN_SWEEP16", "MVN_TRAIN_GROUPS",
                                            "MVN_TRAIN_PAIR", "MVN_FUSED_IP", "MVN_TRAIN_XCD", "MVN_DEALT"};
// The library may be called from several host threads: the table is atomics (a reload while another thread launches gives that
// launch either the old or the new value of a switch, never a torn one), filled under a mutex.
std::atomic<char> g_switch[SW_COUNT];
std::atomic<bool> g_switches_loaded{false};
std::mutex g_switch_mutex;
void load_switches() {
    std::lock_guard<std::mutex> lock(g_switch_mutex);
    for (int i = 0; i < SW_COUNT; ++i) {
        const char *e = getenv(kSwitchNames[i]);
        g_switch[i].store(e ? e[0] : 0, std::memory_order_relaxed);
    }
    g_switches_loaded.store(true, std::memory_order_release);
}
inline char sw(Switch i) {
    if (!g_switches_loaded.load(std::memory_order_acquire)) load_switches();
    return g_switch[i].load(std::memory_order_relaxed);
}

// MVN_UNFUSED=1 forces the two-kernel ViterbiNet path (MLP -> logits -> sweep) for testing.
bool unfused_forced() { return sw(SW_UNFUSED) == '1'; }

// ViterbiNet at S != 16: vnet_fused_ip_kernel<LB> (MLP inside the in-place sweep, no logits in HBM, no scratch) is the default
// where it is also the faster route (round 5, tools/time_vnet_states.py -> profiles/r05_time_vnet_states.txt):
//   * 4, 8, 32, 64 states from ~1 500 blocks (three two-block waves per CU; 10 000 x 1000: 1.33-1.45 x the two-kernel route);
//   * 128 states from ~3 600 blocks (seven per CU; 4 000 x 1000: 1.21 x, 10 000: 1.26 x);
//   * smaller batches are a few lone waves, each running its two blocks' MLP serially, while mlp_kernel spreads the symbols of
//     ALL blocks over the chip: the two-kernel route wins (1 000 x 1000: 1.2-1.5 x);
//   * 256 states: opt-in (MVN_FUSED_IP=1): the W3 and chunk images leave four waves per CU, the two-kernel route is 1.35 x faster.
// MVN_FUSED_IP=1: whenever it can serve; MVN_FUSED_IP=0 / MVN_UNFUSED=1: never.  (Without a device the rule assumes 256 CUs.)
bool fused_ip_selected(int S, int64_t B) {
    if (unfused_forced() || S == 16 || S < 4 || S > 256 || (S & (S - 1))) return false;
    const char e = sw(SW_FUSED_IP);
    if (e == '0' || e == '1') return e == '1';
    if (S > 128) return false;
    const int cus = current_device_cus();
    return (B + 1) / 2 >= (int64_t)(S <= 64 ? 3 : 7) * (cus > 0 ? cus : 256);
}

// The cooperative kernel (vnet16_coop.inc: a 16-wave workgroup per block, one wave sweeping all T steps after the others' MLP
// tiles) against the dealt kernel run with one 8-wave ring per block (its eight waves sweep their units as they finish them):
// measured at 1 ... 768 blocks x 64 ... 1000 symbols (tools/time_coop_vs_dealt.py, profiles/r05_time_coop_vs_dealt.txt) the
// cooperative kernel wins where a block is SHORT and the batch fits one workgroup per CU -- T <= 384 up to CUs blocks (0.68-0.98 x
// the dealt kernel's time; T = 136, the by-word evaluation's call: 0.84), T <= 128 up to 2 CUs -- and loses everywhere else
// (1.05-1.7 x: a long block's sweep by one wave is a serial tail, and 16-wave workgroups run one per CU at a time).
// MVN_COOP=0|1 pins the choice (A/B runs, tests).  with_workspace = false: the caller gave no hand-off lines, the dealt kernel is
// not available and the cooperative one keeps its round-4 range (up to kCoopMaxBlocks blocks) against the one-wave-per-block kernel.
bool coop_selected(int64_t B, int T, bool with_workspace = true) {
    if (T > kCoopMaxT) return false;
    const char e = sw(SW_COOP);
    if (e == '0' || e == '1') return e == '1';
    if (!with_workspace || sw(SW_DEALT) == '0') return B <= kCoopMaxBlocks;
    const int n_cu = current_device_cus();
    const int64_t cus = n_cu > 0 ? n_cu : 256;
    return (B <= cus && T <= 384) || (B <= 2 * cus && T <= 128);
}

// MVN_FUSEDN=2|4 pins the tiles per super-tile of the fused kernel (vnet16_fusedn.inc); default 2 (6 waves/SIMD).
int fusedn_tiles() {
    const char e = sw(SW_FUSEDN);
    if (e == '2' || e == '4') return e - '0';
    return kFusedNDefault;
}

// The opt-in to more than 64 KB of dynamic LDS is per function (and device): raised once, remembered (under a mutex: the
// by-word evaluation of several host threads comes through here once per launch; uncontended it costs ~20 ns)
int ensure_dynamic_lds(const void *fn, size_t bytes) {
    struct Entry {
        const void *fn;
        int dev;
        size_t bytes;
    };
    static Entry seen[64];
    static int n_seen = 0;
    static std::mutex seen_mutex;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(seen_mutex);
    for (int i = 0; i < n_seen; ++i)
        if (seen[i].fn == fn && seen[i].dev == dev) {
            if (seen[i].bytes >= bytes) return 0;
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e != hipSuccess) return (int)e;
            seen[i].bytes = bytes;
            return 0;
        }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    if (n_seen < 64) seen[n_seen++] = Entry{fn, dev, bytes};
    return 0;
}

// counters[1] += K * (#counted rows), counters[3] += #counted rows  (rows with mask != 0, or all B when mask is NULL)
__global__ __launch_bounds__(1024) void count_totals_kernel(const unsigned char *__restrict__ row_mask, int64_t B, int K,
                                                            unsigned long long *counters) {
    __shared__ unsigned long long s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    unsigned long long n = 0;
    if (row_mask) {
        for (int64_t i = threadIdx.x; i < B; i += blockDim.x) n += row_mask[i] != 0;
        if (n) atomicAdd(&s_n, n);
    } else if (threadIdx.x == 0) {
        s_n = (unsigned long long)B;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&counters[1], s_n * (unsigned long long)K);
        atomicAdd(&counters[3], s_n);
    }
}

// The dealt form of the fused kernel (vnet16_dealt.inc): every batch the cooperative kernel does not take, when the caller gave
// the hand-off workspace (mvn_vnet_workspace_bytes) -- units of 32 symbols shared evenly by 3 workgroups per CU instead of whole
// blocks per wave.  MVN_DEALT=0: never (the one-wave-per-block kernel: A/B runs, cross-checks); MVN_FUSEDN=4 implies that too.
#ifdef MVN_TEST_HOOKS
std::atomic<int> g_dealt_skip_ring{-1};        // the tests' build: this group never publishes its cross-group hand-off ...
std::atomic<unsigned> g_dealt_spin_limit{kDealtSpinLimit};  // ... and the wait for it gives up after this many polls
inline int dealt_skip_ring() { return g_dealt_skip_ring.load(std::memory_order_relaxed); }
inline unsigned dealt_spin_limit() { return g_dealt_spin_limit.load(std::memory_order_relaxed); }
#else
inline int dealt_skip_ring() { return -1; }
inline unsigned dealt_spin_limit() { return kDealtSpinLimit; }
#endif
// A fresh non-zero 64-bit number per launch (splitmix64 of a process-wide counter that starts at a value drawn from the clock and
// the address space): the dealt kernel's hand-off flags are "set" when they hold it, so stale flags of earlier launches -- or
// whatever else the caller's workspace held -- never look set (2^-64 per flag and launch).
unsigned long long dealt_nonce() {
    static std::atomic<unsigned long long> counter{0};
    unsigned long long c = counter.load(std::memory_order_relaxed);
    if (c == 0) {
        timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        unsigned long long seed = ((unsigned long long)ts.tv_sec << 32) ^ (unsigned long long)ts.tv_nsec ^ reinterpret_cast<uintptr_t>(&counter) ^
                                  ((unsigned long long)getpid() << 48);
        if (seed == 0) seed = 1;
        counter.compare_exchange_strong(c, seed, std::memory_order_relaxed);  // (one of the racing first callers wins)
    }
    unsigned long long z = counter.fetch_add(0x9E3779B97F4A7C15ull, std::memory_order_relaxed) + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z ? z : 1;
}
// Workgroups ("groups" of 8 waves, 3 per CU) and ring size of a dealt launch.  A ring -- `ring` consecutive waves of a group --
// owns one contiguous range of the launch's units, which must hold at least one block: the more blocks, the smaller the rings
// (8 x groups blocks: every wave a ring of its own, the path metrics stay in its registers except at the two ends of its range;
// fewer: rings of 2, 4, 8 waves that hand the metrics round through LDS).  groups = 0: the shape is not dealt.
struct DealtPlan {
    int groups, ring;
    int rings() const { return groups * (kDealtWaves / ring); }
};
DealtPlan dealt_plan(int64_t B, int T, bool with_coop = true) {  // with_coop = false: the survivor form, which the cooperative kernel does not have
    DealtPlan p = {0, kDealtWaves};
    if (sw(SW_DEALT) == '0' || fusedn_tiles() != 2 || (with_coop && coop_selected(B, T))) return p;
    const int64_t units = ((int64_t)T + kDealtUnit - 1) / kDealtUnit;
    if (B < 1 || B * units >= ((int64_t)1 << 31)) return p;
    const int cus = current_device_cus();
    const int64_t slots = 3 * (int64_t)(cus > 0 ? cus : 256);  // 3 workgroups of 8 waves per CU (80 VGPRs, 36 KB of LDS each)
    p.groups = (int)(B < slots ? B : slots);
    const char e = sw(SW_DEALT);  // MVN_DEALT=8|4|2 pins the ring size (A/B runs, tests); a ring never gets less than a block
    int want = (e == '8' || e == '4' || e == '2') ? e - '0' : 1;
    while (want < kDealtWaves && (int64_t)p.groups * (kDealtWaves / want) > B) want *= 2;
    p.ring = want;
    return p;
}
size_t dealt_workspace_bytes_for(int64_t B, int T) {
    const DealtPlan p = dealt_plan(B, T);
    return p.groups ? dealt_workspace_bytes(p.rings()) : 0;
}

int launch_vnet16_fused(const float *y, int64_t y_ld, const float *W1, const float *b1, const float *W2, const float *b2,
                        const float *W3, const float *b3, float *dec, int64_t dec_ld, float *logits_out,
                        float *final_metric, int64_t B, int T, const float *tx, int64_t tx_ld, int K,
                        const unsigned char *row_mask, unsigned long long *counters, hipStream_t st, void *workspace = nullptr,
                        size_t workspace_bytes = 0) {
    const DealtPlan dplan = dealt_plan(B, T);
    const bool can_deal = dplan.groups && workspace && workspace_bytes >= dealt_workspace_bytes(dplan.rings()) &&
                          !(reinterpret_cast<uintptr_t>(workspace) & 127);
    if (coop_selected(B, T, can_deal)) {  // small batch: a 16-wave workgroup per block (MLP tiles in parallel, one sweeping wave)
        const size_t dyn = (size_t)((T + 15) / 16) * 1024;
        const void *fn = logits_out ? (const void *)vnet16_coop_kernel<true> : (const void *)vnet16_coop_kernel<false>;
        if (int e = ensure_dynamic_lds(fn, (size_t)(kCoopMaxT / 16 * 1024))) return e;
        if (logits_out)
            hipLaunchKernelGGL((vnet16_coop_kernel<true>), dim3((unsigned)B), dim3(64 * kCoopWaves), dyn, st, y, y_ld, W1, b1, W2, b2, W3,
                               b3, dec, dec_ld, logits_out, final_metric, B, T, tx, tx_ld, K, row_mask, counters);
        else
            hipLaunchKernelGGL((vnet16_coop_kernel<false>), dim3((unsigned)B), dim3(64 * kCoopWaves), dyn, st, y, y_ld, W1, b1, W2, b2, W3,
                               b3, dec, dec_ld, logits_out, final_metric, B, T, tx, tx_ld, K, row_mask, counters);
        if (tx) hipLaunchKernelGGL(count_totals_kernel, dim3(1), dim3(1024), 0, st, row_mask, B, K, counters);
        return (int)hipGetLastError();
    }
    if (can_deal) {
        const DealtPlan dp = dplan;
        const unsigned long long nonce = dealt_nonce();  // what a SET hand-off flag holds in this launch: nothing to clear
        if (logits_out)
            hipLaunchKernelGGL((vnet16_dealt_kernel<true>), dim3((unsigned)dp.groups), dim3(64 * kDealtWaves), 0, st, y, y_ld, W1, b1, W2, b2,
                               W3, b3, dec, dec_ld, logits_out, final_metric, (int)B, T, tx, tx_ld, K, row_mask, counters, (float *)workspace,
                               dp.groups, dp.ring, dealt_spin_limit(), dealt_skip_ring(), nonce, (unsigned char *)nullptr);
        else
            hipLaunchKernelGGL((vnet16_dealt_kernel<false>), dim3((unsigned)dp.groups), dim3(64 * kDealtWaves), 0, st, y, y_ld, W1, b1, W2, b2,
                               W3, b3, dec, dec_ld, logits_out, final_metric, (int)B, T, tx, tx_ld, K, row_mask, counters, (float *)workspace,
                               dp.groups, dp.ring, dealt_spin_limit(), dealt_skip_ring(), nonce, (unsigned char *)nullptr);
        if (tx) hipLaunchKernelGGL(count_totals_kernel, dim3(1), dim3(1024), 0, st, row_mask, B, K, counters);
        return (int)hipGetLastError();
    }
    const int nt = fusedn_tiles();  // NT = 2 runs 5 waves/SIMD (default); NT = 4 is the 64-symbol form, kept as a cross-check
    const unsigned gridn = (unsigned)((B + kFusedNWaves - 1) / kFusedNWaves);
#define MVN_FUSEDN_LAUNCH(WL, NT)                                                                                      \
    hipLaunchKernelGGL((vnet16_fusedn_kernel<WL, NT>), dim3(gridn), dim3(64 * kFusedNWaves), 0, st, y, y_ld, W1, b1, W2, b2, \
                       W3, b3, dec, dec_ld, logits_out, final_metric, B, T, tx, tx_ld, K, row_mask, counters)
    if (nt == 2) {
        if (logits_out) MVN_FUSEDN_LAUNCH(true, 2); else MVN_FUSEDN_LAUNCH(false, 2);
    } else {
        if (logits_out) MVN_FUSEDN_LAUNCH(true, 4); else MVN_FUSEDN_LAUNCH(false, 4);
    }
#undef MVN_FUSEDN_LAUNCH
    if (tx) hipLaunchKernelGGL(count_totals_kernel, dim3(1), dim3(1024), 0, st, row_mask, B, K, counters);
    return (int)hipGetLastError();
}

// MVN_GENERIC_SWEEP=1 forces the generic state-per-lane LDS sweep for every S (testing).
bool generic_sweep_forced() { return sw(SW_GENERIC_SWEEP) == '1'; }

// Which kernel serves a sweep: ONE decision function used by the dispatcher and by the mvn_*_kernel_name queries, so the
// name a caller is told is the kernel that runs (same environment switches, same alignment fall-backs).
enum SweepKind { SK_GENERIC, SK_VA256_WAVE, SK_VA_INPLACE, SK_SWEEP_INPLACE, SK_S16_QUAD, SK_S16_LDS, SK_S16_ROWS, SK_VA16_QUAD, SK_VA16_TILE,
                 SK_VA16_SPLIT };

template <int MODE>
SweepKind plan_sweep(const void *src, const void *dec, int64_t dec_ld, int64_t B, int S, int T = 1 << 30) {
    const bool generic = generic_sweep_forced();
    if constexpr (MODE == MODE_VA) {
        // classical VA: the lane-bits x register-bits in-place kernel serves every S >= 4 except S = 16, which has its
        // own 16-blocks-per-wave / row kernels (MVN_VA_INPLACE=1 sends S = 16 there too, for cross-checks)
        // (and S = 256, one block per wave with scalar decisions; MVN_VA256=inplace keeps the family kernel there)
        if (S == 256 && !generic && sw(SW_VA256) != 'i') return SK_VA256_WAVE;
        if (S >= 4 && !generic && (S != 16 || sw(SW_VA_INPLACE) == '1')) return SK_VA_INPLACE;
        if (S == 16 && !generic) {  // MVN_VA16 = "rows" | "quad" | "tile" | "split" pins a variant (A/B, tests); default by size:
            const char e = sw(SW_VA16);  // four waves per block for a few hundred short blocks, one wave per block below 6 000, 16 blocks per wave from there on
            if (e == 'r' || e == 'q' || e == 't') return e == 'q' ? SK_VA16_QUAD : e == 't' ? SK_VA16_TILE : SK_S16_ROWS;
            if (e == 's') return T <= kVaSplitMaxT ? SK_VA16_SPLIT : SK_VA16_TILE;
            return B >= kVaQuadMinBlocks ? SK_VA16_QUAD : va16_split_serves(B, T) ? SK_VA16_SPLIT : SK_VA16_TILE;
        }
        return SK_GENERIC;
    } else {
        // materialised costs: the LDS-DMA kernels move 16-byte pieces and store float4 decisions; buffers that are not
        // 16-byte aligned take the row kernel (S = 16) or the generic kernel
        const bool aligned = !((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dec) & 15) || (dec_ld & 3));
        if (S >= 4 && !generic && aligned && (S != 16 || sw(SW_SWEEP_INPLACE) == '1')) return SK_SWEEP_INPLACE;
        if (S == 16 && !generic) {
            if (reinterpret_cast<uintptr_t>(src) & 15) return SK_S16_ROWS;
            const char e = sw(SW_SWEEP16);  // "rows" | "lds" | "quad" pins a variant; default by size
            const char v = (e == 'q' || e == 'l' || e == 'r') ? e : (sweep16_quad_preferred(B) ? 'q' : 'l');
            return v == 'q' ? SK_S16_QUAD : v == 'l' ? SK_S16_LDS : SK_S16_ROWS;
        }
        return SK_GENERIC;
    }
}

int log2_states(int S) {
    int l = 0;
    while ((1 << l) < S) ++l;
    return l;
}

template <int MODE>
void sweep_kernel_name(SweepKind k, int S, const void *dec, int64_t dec_ld, char *name, size_t n) {
    const int lb = log2_states(S) - 2;
    switch (k) {
        case SK_VA256_WAVE: snprintf(name, n, "va256_wave_kernel"); break;
        case SK_VA_INPLACE: snprintf(name, n, "va_inplace_kernel<%d>", lb); break;
        case SK_SWEEP_INPLACE: snprintf(name, n, "sweep_inplace_kernel<%d, %d, %d>", lb, MODE, lb >= 2 ? 4 : lb == 1 ? 8 : 16); break;
        case SK_S16_QUAD:
            snprintf(name, n, "sweep16_quad_kernel<%d, %s>", MODE,
                     ((dec_ld % 4) == 0 && (reinterpret_cast<uintptr_t>(dec) % 16) == 0) ? "true" : "false");
            break;
        case SK_S16_LDS: snprintf(name, n, "sweep16_lds_kernel<%d>", MODE); break;
        case SK_S16_ROWS: snprintf(name, n, "sweep16_rows_kernel<%d>", MODE); break;
        case SK_VA16_QUAD: snprintf(name, n, "va16_quad_kernel"); break;
        case SK_VA16_TILE: snprintf(name, n, "va16_tile_kernel"); break;
        case SK_VA16_SPLIT: snprintf(name, n, "va16_split_kernel"); break;
        default: snprintf(name, n, "sweep_kernel<%d, %d>", S, MODE); break;
    }
}

// gw: the ViterbiNet weights when the costs are the logits of the two-kernel route (MODE_NEGLOGIT), else NULL
template <int MODE>
int dispatch_sweep(const float *src, int64_t src_ld, const float *priors, int64_t Bp, float *dec, int64_t dec_ld,
                   float *final_metric, int64_t B, int T, int S, hipStream_t st, const GuardWeights *gw = nullptr) {
    int rc = -1;
    const SweepKind kind = plan_sweep<MODE>(src, dec, dec_ld, B, S, T);
    switch (kind) {
        case SK_VA256_WAVE:
            if constexpr (MODE == MODE_VA) rc = launch_va256_wave(src, src_ld, priors, Bp, dec, dec_ld, final_metric, B, T, st);
            break;
        case SK_VA_INPLACE:
            if constexpr (MODE == MODE_VA) rc = launch_va_inplace(src, src_ld, priors, Bp, dec, dec_ld, final_metric, B, T, S, st);
            break;
        case SK_VA16_QUAD:
            if constexpr (MODE == MODE_VA) rc = launch_va16_quad(src, src_ld, priors, Bp, dec, dec_ld, final_metric, B, T, st);
            break;
        case SK_VA16_TILE:  // (strict by itself: va16_tile.inc)
            if constexpr (MODE == MODE_VA) return launch_va16_tile(src, src_ld, priors, Bp, dec, dec_ld, final_metric, B, T, st);
            break;
        case SK_VA16_SPLIT:  // (strict by itself too; a launch that cannot have its LDS falls back to one wave per block)
            if constexpr (MODE == MODE_VA) {
                const int rs = launch_va16_split(src, src_ld, priors, Bp, dec, dec_ld, final_metric, B, T, st);
                return rs < 0 ? launch_va16_tile(src, src_ld, priors, Bp, dec, dec_ld, final_metric, B, T, st) : rs;
            }
            break;
        case SK_SWEEP_INPLACE:
            if constexpr (MODE != MODE_VA) rc = launch_sweep_inplace<MODE>(src, dec, dec_ld, final_metric, B, T, S, st);
            break;
        case SK_S16_QUAD:
            if constexpr (MODE != MODE_VA) rc = launch_sweep16_quad<MODE>(src, dec, dec_ld, final_metric, B, T, st);
            break;
        case SK_S16_LDS:
            if constexpr (MODE != MODE_VA) rc = launch_sweep16_lds<MODE>(src, dec, dec_ld, final_metric, B, T, st);
            break;
        case SK_S16_ROWS: rc = launch_sweep16_rows<MODE>(src, src_ld, priors, Bp, dec, dec_ld, final_metric, B, T, st); break;
        default: break;
    }
    if (rc < 0)  // the generic kernel (its ACS minimum is torch.min's: no guard needed)
        return launch_sweep<MODE>(src, src_ld, priors, Bp, dec, dec_ld, final_metric, B, T, S, st);
    if (rc) return rc;
    // the specialised kernels' v_min_f32 stages drop a NaN that sits in one candidate only: the guard launch repairs the
    // blocks whose state priors (VA) / whose weights (ViterbiNet logits) can produce such costs, and is a no-op otherwise
    if constexpr (MODE == MODE_VA) return launch_sweep<MODE_VA, true>(src, src_ld, priors, Bp, dec, dec_ld, final_metric, B, T, S, st);
    if constexpr (MODE == MODE_NEGLOGIT)
        if (gw) return launch_sweep<MODE_NEGLOGIT, true>(src, src_ld, priors, Bp, dec, dec_ld, final_metric, B, T, S, st, *gw);
    return MVN_OK;
}

bool valid_states(int S) { return S >= 2 && S <= 256 && (S & (S - 1)) == 0; }

// CUs of the current device (cached per device): the one-workgroup-per-chunk training kernels need one CU per workgroup to be
// resident at once (each workgroup takes most of a CU's LDS)
int current_device_cus() {
    static std::atomic<int> cached[64];  // (zero-initialised; two threads that both miss store the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (!n) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}


// ---- launchers of the training kernels: one trial by value (`many` NULL) or R trials from a device array ----------------
// Workgroups per trial when a pass is spread over one workgroup per chunk (train_groups.inc); < 2 = the single-workgroup
// kernels serve the shape (one chunk, or more than kTrainMaxGroups)
int online_groups(int T) {
    const int g = (T + kTrainChunk - 1) / kTrainChunk;
    return g > kTrainMaxGroups ? 0 : g;
}
int maml_groups(int T, int W, int second_order) {
    const long long n_sup = (long long)W * T;
    const long long g = second_order ? (n_sup + kHvRows - 1) / kHvRows : (n_sup + kTrainChunk - 1) / kTrainChunk;
    return g > kTrainMaxGroups ? 0 : (int)g;
}
// floats between the workspace regions of consecutive trials (GroupSync, slots, private moment copies; 256-byte multiple)
size_t trial_workspace_floats(int S, int groups) {
    return ((maml_groups_workspace_bytes(S, groups) + 255) & ~(size_t)255) / sizeof(float);
}

// The chunked training launches place a trial's workgroups on ONE XCD (train_groups.inc, group_place): a 1-D grid in units of
// 8 x groups blocks, 8 trials per unit; the trial's gradient exchange then stays in that XCD's L2 (a whole-word iteration of one
// trial 9.4 -> 8.8 us, a second-order step 38 -> 36 us; profiles/r04_time_groups_xcd_ab.txt).  The units hold a few trials less
// per launch than CUs / groups (48 instead of 51 at 5 workgroups per trial): the (groups, trials) grid, whose workgroups of a
// trial land on different XCDs, stays where the XCD-aware one would need more launches.  MVN_TRAIN_XCD=0: never (A/B; same bits).
// The single-trial entry points (`many` false) may have launches in flight on several streams.  Each of them would put its G
// workgroups on the XCD the dispatcher starts a grid on, and two launches that each hold a part of that XCD's 32 CUs would wait
// for each other until the spin limit: their slot within the unit is therefore rotated from launch to launch (xcd_first), and
// above 8 workgroups per trial (T > 256) they keep the (groups, trials) grid, whose workgroups spread over all 256 CUs -- an XCD
// then holds at least 4 single-trial launches, 32 over the device, like the classic grid at that size.  (For either grid the
// chunked launches in flight together must fit the device's CUs; include/mvn.h.)
constexpr int kXcdSingleMaxGroups = 8;
std::atomic<unsigned> g_xcd_rotation{0};
bool xcd_grid(int groups, int cus, int R, bool many = true) {
    if (sw(SW_TRAIN_XCD) == '0' || groups < 2 || 8 * groups > cus) return false;
    if (!many && groups > kXcdSingleMaxGroups) return false;
    const int fit_x = 8 * (cus / (8 * groups)), fit_c = cus / groups;
    return (R + fit_x - 1) / fit_x <= (R + fit_c - 1) / fit_c;
}
int group_trials_fit(int groups, int cus, int R) { return xcd_grid(groups, cus, R) ? 8 * (cus / (8 * groups)) : cus / groups; }
int next_xcd_slot(bool xcd, bool many) { return xcd && !many ? (int)(g_xcd_rotation.fetch_add(1, std::memory_order_relaxed) & 7u) : 0; }
unsigned group_grid_blocks(int groups, int trials) { return (unsigned)(8 * groups * ((trials + 7) / 8)); }

// R trials, at most `fit` per launch (workgroups <= CUs): as few launches as possible, of equal size (29 trials with room for
// 28 run as 15 + 14, not 28 + 1)
int trials_per_launch(int R, int fit) {
    fit = std::max(1, fit);
    const int launches = (R + fit - 1) / fit;
    return (R + launches - 1) / launches;
}

// Many trials: one workgroup per CHUNK and trial finishes a trial soonest, one workgroup per TRIAL gets the most trials through
// a CU per second (no exchange, no workgroup waiting for a pass it has no chunk of): a full-word iteration costs 9.9 us on 5
// CUs or 31.0 us on one, a second-order meta-learning step 39 us on 5 or 133 us on one (first order 21 / 66;
// profiles/r03_time_online_training.txt).  With R trials the chunked form needs ceil(R / (CUs / groups)) launches one after
// the other, the one-workgroup form ceil(R / CUs) rounds of `slowdown` times the length: take whichever ends first.  (Both
// forms give the same bits.)
bool one_workgroup_per_trial_is_faster(int R, int groups, int cus, double slowdown) {
    if (groups < 2 || cus < groups) return true;
    const int fit = group_trials_fit(groups, cus, R);
    const double chunked = (double)((R + fit - 1) / fit), single = slowdown * (double)((R + cus - 1) / cus);
    return single < chunked;
}

// zero the GroupSync at the head of each of `trials` workspace regions
hipError_t clear_group_syncs(float *workspace, size_t stride_floats, int trials, hipStream_t st) {
    if (trials == 1) return hipMemsetAsync(workspace, 0, sizeof(GroupSync), st);
    return hipMemset2DAsync(workspace, stride_floats * sizeof(float), 0, sizeof(GroupSync), (size_t)trials, st);
}

template <class Launch>  // launch(SC tag): picks the instantiation for S
int dispatch_states(int S, bool many, Launch launch) {
    if (S == 16) return launch(std::integral_constant<int, 16>{});
    if (S == 32 && !many) return launch(std::integral_constant<int, 32>{});  // (R trials at S = 32 take the run-time form)
    if (S == 64) return launch(std::integral_constant<int, 64>{});   // } online training only (launch_online_train)
    if (S == 128) return launch(std::integral_constant<int, 128>{}); // }
    return launch(std::integral_constant<int, 0>{});
}

// Workgroups per trial a training call runs with (0 = one workgroup per trial): ONE decision function for the launchers and
// for mvn_vnet_train_kernel_name, so that the form a test or a profile is told is the form that runs.  `many`: the
// trial-batched entry points (R trials); has_workspace / workspace_bytes as the caller passed them.
int plan_online_groups(bool many, int R, int T, int M, int S, bool has_workspace, size_t workspace_bytes) {
    const int groups = M > 0 || S > 32 ? 0 : online_groups(T);  // minibatch iterations are one chunk: nothing to spread; 64 / 128 states: one workgroup
    const int cus = current_device_cus();
    if (groups < 2 || !has_workspace || sw(SW_TRAIN_GROUPS) == '0' || groups > cus) return 0;
    if (workspace_bytes < (many ? (size_t)R * trial_workspace_floats(S, groups) * sizeof(float) : train_groups_workspace_bytes(S, groups)))
        return 0;
    if (many && sw(SW_TRAIN_GROUPS) != '1' && one_workgroup_per_trial_is_faster(R, groups, cus, 3.1)) return 0;
    return groups;
}
int plan_maml_groups(bool many, int R, int T, int W, int second_order, int S, bool has_workspace, size_t workspace_bytes) {
    const int groups = maml_groups(T, W, second_order);  // one workgroup per chunk of the largest pass
    const int cus = current_device_cus();
    if (groups < 2 || !has_workspace || sw(SW_TRAIN_GROUPS) == '0' || groups > cus) return 0;
    if (workspace_bytes < (many ? (size_t)R * trial_workspace_floats(S, groups) * sizeof(float) : maml_groups_workspace_bytes(S, groups)))
        return 0;
    if (many && sw(SW_TRAIN_GROUPS) != '1' && one_workgroup_per_trial_is_faster(R, groups, cus, second_order ? 3.4 : 3.1)) return 0;
    return groups;
}

// More trials than CUs on one workgroup per trial: the 512-thread form of online_train_kernel puts TWO trials on a CU at once
// (online_train.inc); it needs parameters + gradient + a compact arena within half a CU's LDS (n_states <= 16).  Measured
// (tools/time_train_pair.py, profiles/r04_time_train_pair.txt): a 512-thread workgroup alone on a CU is 1.3-1.4 x slower than a
// 1024-thread one, two of them together 1.09-1.10 x faster on whole-word iterations (512 / 1024 trials: 11.5 -> 10.5 ms,
// 23.0 -> 20.8 ms per 200 iterations) and 0.92-0.94 x on 32-sample minibatch iterations, whose Adam update every 6.6 us waits for
// its moments from global memory.  So: whole-word iterations with more trials than CUs only.  MVN_TRAIN_PAIR=0|1 pins the choice
// (A/B, tests: the results are bit-identical either way).
bool online_pair_form(bool many, int R, int M, int S) {
    // 16 states only: the compact arena puts the column sums on the dlogits image, which is harmless only when EVERY column the
    // products read of that image is rewritten by every chunk (for n_states < 16 columns n_states..15 are padding that must stay 0;
    // 32 states do not fit half a CU's LDS)
    if (!many || S != 16 || online_train2_lds_bytes(S) > (size_t)80 * 1024) return false;
    const char e = sw(SW_TRAIN_PAIR);
    if (e == '0' || e == '1') return e == '1';
    return M == 0 && R > current_device_cus();
}

int launch_online_train(const mvn_train_trial_t &one, const mvn_train_trial_t *many, int R, int T, int M, float lr, float beta1,
                        float beta2, float eps, int S, void *workspace, size_t workspace_bytes, hipStream_t st) {
    const size_t lds = online_train_lds_bytes(S);
    const int lds_floats = (int)online_train_lds_floats(S);
    const int groups = plan_online_groups(many != nullptr, R, T, M, S, workspace != nullptr, workspace_bytes);
    const int cus = current_device_cus();
    const size_t stride = groups >= 2 ? trial_workspace_floats(S, groups) : 0;
    if (groups && (reinterpret_cast<uintptr_t>(workspace) & 15)) return MVN_E_WORKSPACE;
    if (!groups) {  // one workgroup per trial
        return dispatch_states(S, many != nullptr, [&](auto sc) -> int {
            constexpr int SC = decltype(sc)::value;
            if (many && online_pair_form(true, R, M, S)) {  // (16 states only)
                constexpr int SC2 = SC >= 32 ? 0 : SC;
                const size_t lds2 = online_train2_lds_bytes(S);
                if (int e = ensure_dynamic_lds((const void *)online_train_kernel<SC2, true, kTrainThreads2>, lds2)) return e;
                hipLaunchKernelGGL((online_train_kernel<SC2, true, kTrainThreads2>), dim3(1, (unsigned)R), dim3(kTrainThreads2),
                                   lds2, st, one, many, T, M, lr, beta1, beta2, eps, S, (int)online_train2_lds_floats(S));
            } else if (many) {
                if (int e = ensure_dynamic_lds((const void *)online_train_kernel<SC == 32 ? 0 : SC, true>, lds)) return e;
                hipLaunchKernelGGL((online_train_kernel<SC == 32 ? 0 : SC, true>), dim3(1, (unsigned)R), dim3(kTrainThreads), lds, st,
                                   one, many, T, M, lr, beta1, beta2, eps, S, lds_floats);
            } else {
                if (int e = ensure_dynamic_lds((const void *)online_train_kernel<SC, false>, lds)) return e;
                hipLaunchKernelGGL((online_train_kernel<SC, false>), dim3(1), dim3(kTrainThreads), lds, st, one, many, T, M, lr,
                                   beta1, beta2, eps, S, lds_floats);
            }
            return (int)hipGetLastError();
        });
    }
    // one workgroup per chunk and trial, never more workgroups in a launch than the device has CUs (all resident at once)
    const size_t ws_floats_one = train_groups_workspace_bytes(S, groups) / sizeof(float);
    const bool xcd = xcd_grid(groups, cus, R, many != nullptr);
    const int per_launch = many ? trials_per_launch(R, group_trials_fit(groups, cus, R)) : 1;
    for (int r0 = 0; r0 < R; r0 += per_launch) {
        const int nr = std::min(per_launch, R - r0);
        float *wsr = (float *)workspace + (size_t)r0 * stride;
        hipError_t e = clear_group_syncs(wsr, stride, nr, st);
        if (e != hipSuccess) return (int)e;
        const GroupLaunch gl = {wsr, (long long)stride, (unsigned)((many ? stride : ws_floats_one) * sizeof(float)), group_spin_limit(), group_phantoms(),
                                xcd ? groups : 0, nr, next_xcd_slot(xcd, many != nullptr)};
        const dim3 grid = xcd ? dim3(group_grid_blocks(groups, nr)) : dim3((unsigned)groups, (unsigned)nr);
        const int rc = dispatch_states(S, many != nullptr, [&](auto sc) -> int {
            constexpr int SC = decltype(sc)::value > 32 ? 0 : decltype(sc)::value;  // (never above 32 states: plan_online_groups)
            if (many) {
                if (int e2 = ensure_dynamic_lds((const void *)online_train_groups_kernel<SC == 32 ? 0 : SC, true>, lds)) return e2;
                hipLaunchKernelGGL((online_train_groups_kernel<SC == 32 ? 0 : SC, true>), grid,
                                   dim3(kTrainThreads), lds, st, one, many + r0, T, lr, beta1, beta2, eps, S, lds_floats, gl);
            } else {
                if (int e2 = ensure_dynamic_lds((const void *)online_train_groups_kernel<SC, false>, lds)) return e2;
                hipLaunchKernelGGL((online_train_groups_kernel<SC, false>), grid, dim3(kTrainThreads), lds, st, one,
                                   many, T, lr, beta1, beta2, eps, S, lds_floats, gl);
            }
            return (int)hipGetLastError();
        });
        if (rc) return rc;
    }
    return MVN_OK;
}

int launch_maml_train(const mvn_train_trial_t &one, const mvn_train_trial_t *many, int R, int T, int W, float meta_lr,
                      int second_order, float lr, float beta1, float beta2, float eps, int S, void *workspace,
                      size_t workspace_bytes, hipStream_t st) {
    const size_t lds = maml_train_lds_floats(S) * sizeof(float);
    const int lds_floats = (int)maml_train_lds_floats(S);
    const int groups = plan_maml_groups(many != nullptr, R, T, W, second_order, S, workspace != nullptr, workspace_bytes);
    const int cus = current_device_cus();
    const size_t stride = groups >= 2 ? trial_workspace_floats(S, groups) : 0;
    if (groups && (reinterpret_cast<uintptr_t>(workspace) & 15)) return MVN_E_WORKSPACE;
    if (!groups) {
        return dispatch_states(S, many != nullptr, [&](auto sc) -> int {
            constexpr int SC = decltype(sc)::value > 32 ? 0 : decltype(sc)::value;  // (the meta-learning kernels: up to 32 states)
            if (many) {
                if (int e = ensure_dynamic_lds((const void *)maml_train_kernel<SC == 32 ? 0 : SC, true>, lds)) return e;
                hipLaunchKernelGGL((maml_train_kernel<SC == 32 ? 0 : SC, true>), dim3(1, (unsigned)R), dim3(kTrainThreads), lds, st, one,
                                   many, T, W, meta_lr, second_order, lr, beta1, beta2, eps, S, lds_floats);
            } else {
                if (int e = ensure_dynamic_lds((const void *)maml_train_kernel<SC, false>, lds)) return e;
                hipLaunchKernelGGL((maml_train_kernel<SC, false>), dim3(1), dim3(kTrainThreads), lds, st, one, many, T, W, meta_lr,
                                   second_order, lr, beta1, beta2, eps, S, lds_floats);
            }
            return (int)hipGetLastError();
        });
    }
    const size_t ws_floats_one = maml_groups_workspace_bytes(S, groups) / sizeof(float);
    const bool xcd = xcd_grid(groups, cus, R, many != nullptr);
    const int per_launch = many ? trials_per_launch(R, group_trials_fit(groups, cus, R)) : 1;
    for (int r0 = 0; r0 < R; r0 += per_launch) {
        const int nr = std::min(per_launch, R - r0);
        float *wsr = (float *)workspace + (size_t)r0 * stride;
        hipError_t e = clear_group_syncs(wsr, stride, nr, st);
        if (e != hipSuccess) return (int)e;
        const GroupLaunch gl = {wsr, (long long)stride, (unsigned)((many ? stride : ws_floats_one) * sizeof(float)), group_spin_limit(), group_phantoms(),
                                xcd ? groups : 0, nr, next_xcd_slot(xcd, many != nullptr)};
        const dim3 grid = xcd ? dim3(group_grid_blocks(groups, nr)) : dim3((unsigned)groups, (unsigned)nr);
        const int rc = dispatch_states(S, many != nullptr, [&](auto sc) -> int {
            constexpr int SC = decltype(sc)::value > 32 ? 0 : decltype(sc)::value;  // (the meta-learning kernels: up to 32 states)
            if (many) {
                if (int e2 = ensure_dynamic_lds((const void *)maml_train_groups_kernel<SC == 32 ? 0 : SC, true>, lds)) return e2;
                hipLaunchKernelGGL((maml_train_groups_kernel<SC == 32 ? 0 : SC, true>), grid,
                                   dim3(kTrainThreads), lds, st, one, many + r0, T, W, meta_lr, second_order, lr, beta1, beta2, eps, S,
                                   lds_floats, gl);
            } else {
                if (int e2 = ensure_dynamic_lds((const void *)maml_train_groups_kernel<SC, false>, lds)) return e2;
                hipLaunchKernelGGL((maml_train_groups_kernel<SC, false>), grid, dim3(kTrainThreads), lds, st, one,
                                   many, T, W, meta_lr, second_order, lr, beta1, beta2, eps, S, lds_floats, gl);
            }
            return (int)hipGetLastError();
        });
        if (rc) return rc;
    }
    return MVN_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int mvn_version(void) { return MVN_ABI_VERSION; }

const char *mvn_strerror(int code) {
    switch (code) {
        case MVN_OK: return "ok";
        case MVN_E_DIMS: return "mvn: bad dimensions (negative size or T larger than a row stride)";
        case MVN_E_STATES: return "mvn: n_states must be a power of two in [2,256]";
        case MVN_E_PRIORS: return "mvn: state-prior table rows must be >=1 and divide the batch";
        case MVN_E_NULL: return "mvn: required pointer is NULL";
        case MVN_E_WORKSPACE: return "mvn: workspace smaller than one block of logits";
        case MVN_E_DEVICE: return "mvn: current HIP device is not gfx950";
        case MVN_E_BARRIER: return "mvn: a training launch abandoned its device-wide barrier (workgroups not co-resident); weights are NaN";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "mvn: unknown error";
}

int mvn_device_info(int *n_cu, int *lds_bytes_per_cu, char *arch_name, int arch_name_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return (int)e;
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    if (arch_name && arch_name_len > 0) {
        int i = 0;
        for (; i < arch_name_len - 1 && p.gcnArchName[i]; ++i) arch_name[i] = p.gcnArchName[i];
        arch_name[i] = 0;
    }
    const char *a = p.gcnArchName;
    return (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? MVN_OK
                                                                                                  : MVN_E_DEVICE;
}

int mvn_acs_block_f32(const float *in_prob, const float *llrs, float *out, int64_t *argmin_j, int64_t B,
                      int32_t S, mvn_stream_t stream) {
    if (B < 0) return MVN_E_DIMS;
    if (!valid_states(S)) return MVN_E_STATES;
    if (B == 0) return MVN_OK;
    if (!in_prob || !llrs || !out) return MVN_E_NULL;
    const int64_t n = B * S;
    hipLaunchKernelGGL(acs_block_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       in_prob, llrs, out, (long long *)argmin_j, B, S);
    return (int)hipGetLastError();
}

int mvn_acs_sweep_f32(const float *cost, float *dec, int64_t dec_ld, float *final_metric, int64_t B,
                      int32_t T, int32_t S, mvn_stream_t stream) {
    if (B < 0 || T < 0 || dec_ld < T) return MVN_E_DIMS;
    if (!valid_states(S)) return MVN_E_STATES;
    if (B == 0 || T == 0) return MVN_OK;
    if (!cost || !dec) return MVN_E_NULL;
    return dispatch_sweep<MODE_COST>(cost, 0, nullptr, 1, dec, dec_ld, final_metric, B, T, S,
                                   (hipStream_t)stream);
}

size_t mvn_survivor_bytes(int64_t B, int32_t T, int32_t S) {
    if (B <= 0 || T <= 0 || !valid_states(S)) return 0;
    return (size_t)B * (size_t)T * (size_t)(S >= 8 ? S / 8 : 1);
}

int mvn_acs_sweep_surv_f32(const float *cost, float *dec, int64_t dec_ld, float *final_metric, uint8_t *surv, int64_t B, int32_t T,
                           int32_t S, mvn_stream_t stream) {
    if (B < 0 || T < 0 || dec_ld < T) return MVN_E_DIMS;
    if (!valid_states(S)) return MVN_E_STATES;
    if (B == 0 || T == 0) return MVN_OK;
    if (!cost || !dec || !surv) return MVN_E_NULL;
    if (S == 16 && sw(SW_GENERIC_SWEEP) != '1' && sweep16_quad_surv_serves(cost, dec, dec_ld, surv, T))  // the HBM-bound form (sweep16_quad.inc)
        return launch_sweep16_quad_surv<MODE_COST>(cost, dec, dec_ld, final_metric, surv, B, T, (hipStream_t)stream);
    return launch_sweep_surv<MODE_COST>(cost, 0, nullptr, 1, dec, dec_ld, final_metric, surv, B, T, S, (hipStream_t)stream);
}

int mvn_va_decode_surv_f32(const float *y, int64_t y_ld, const float *state_priors, int64_t Bp, float *dec, int64_t dec_ld,
                           float *final_metric, uint8_t *surv, int64_t B, int32_t T, int32_t S, mvn_stream_t stream) {
    if (B < 0 || T < 0 || dec_ld < T || y_ld < T) return MVN_E_DIMS;
    if (!valid_states(S)) return MVN_E_STATES;
    if (Bp < 1 || (B % Bp) != 0) return MVN_E_PRIORS;
    if (B == 0 || T == 0) return MVN_OK;
    if (!y || !state_priors || !dec || !surv) return MVN_E_NULL;
    if (S == 16 && sw(SW_GENERIC_SWEEP) != '1' && va16_quad_surv_serves(dec, dec_ld, surv, T))  // 16 blocks per wave (va16_quad.inc)
        return launch_va16_quad_surv(y, y_ld, state_priors, Bp, dec, dec_ld, final_metric, surv, B, T, (hipStream_t)stream);
    return launch_sweep_surv<MODE_VA>(y, y_ld, state_priors, Bp, dec, dec_ld, final_metric, surv, B, T, S, (hipStream_t)stream);
}

// ViterbiNet with survivors on the dealt kernel (16 states): no logits in HBM.  -1: not served (the caller takes the two-kernel route)
int launch_vnet16_dealt_surv(const float *y, int64_t y_ld, const float *W1, const float *b1, const float *W2, const float *b2,
                             const float *W3, const float *b3, float *dec, int64_t dec_ld, float *final_metric, unsigned char *surv,
                             void *workspace, size_t workspace_bytes, int64_t B, int T, hipStream_t st) {
    if (unfused_forced() || (T & 3) || (reinterpret_cast<uintptr_t>(surv) & 7) || (reinterpret_cast<uintptr_t>(workspace) & 127)) return -1;
    const DealtPlan dp = dealt_plan(B, T, false);
    if (!dp.groups || workspace_bytes < dealt_workspace_bytes(dp.rings())) return -1;
    hipLaunchKernelGGL((vnet16_dealt_kernel<false, true>), dim3((unsigned)dp.groups), dim3(64 * kDealtWaves), 0, st, y, y_ld, W1, b1, W2, b2,
                       W3, b3, dec, dec_ld, (float *)nullptr, final_metric, (int)B, T, (const float *)nullptr, (int64_t)0, 0,
                       (const unsigned char *)nullptr, (unsigned long long *)nullptr, (float *)workspace, dp.groups, dp.ring, dealt_spin_limit(),
                       dealt_skip_ring(), dealt_nonce(), surv);
    return (int)hipGetLastError();
}

size_t mvn_vnet_surv_workspace_bytes(int64_t B, int32_t T, int32_t S) {
    if (B <= 0 || T <= 0 || !valid_states(S)) return 0;
    if (S == 16 && !unfused_forced() && !(T & 3)) {
        const DealtPlan dp = dealt_plan(B, T, false);
        if (dp.groups) return dealt_workspace_bytes(dp.rings());  // the fused detector: hand-off lines only
    }
    return (size_t)B * (size_t)T * (size_t)S * sizeof(float);  // the logits of the two-kernel route
}

int mvn_vnet_decode_surv_f32(const float *y, int64_t y_ld, const float *W1, const float *b1, const float *W2, const float *b2,
                             const float *W3, const float *b3, float *dec, int64_t dec_ld, float *final_metric, uint8_t *surv,
                             void *workspace, size_t workspace_bytes, int64_t B, int32_t T, int32_t S, mvn_stream_t stream) {
    if (B < 0 || T < 0 || dec_ld < T || y_ld < T) return MVN_E_DIMS;
    if (!valid_states(S)) return MVN_E_STATES;
    if (B == 0 || T == 0) return MVN_OK;
    if (!y || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !dec || !surv || !workspace) return MVN_E_NULL;
    // the logits of a slice of blocks in the workspace (mlp_kernel: the reference's logits bit for bit), then the sweep over them
    // with branch cost -logit (vnet_detector.py:57) and the survivors kept; torch.min's rule inside the sweeps (odd costs)
    hipStream_t st = (hipStream_t)stream;
    if (S == 16) {  // the fused detector itself (vnet16_dealt_kernel<false, true>: survivors out of its decision pass), when it serves the call
        const int rf = launch_vnet16_dealt_surv(y, y_ld, W1, b1, W2, b2, W3, b3, dec, dec_ld, final_metric, surv, workspace, workspace_bytes, B, T, st);
        if (rf >= 0) return rf;
    }
    const size_t per_block = (size_t)T * (size_t)S * sizeof(float);
    int64_t slice = (int64_t)(workspace_bytes / per_block);
    if (slice < 1 || (reinterpret_cast<uintptr_t>(workspace) & 15)) return MVN_E_WORKSPACE;
    if (slice > B) slice = B;
    const int SB = S >= 8 ? S / 8 : 1;
    float *lg = (float *)workspace;
    for (int64_t b0 = 0; b0 < B; b0 += slice) {
        const int64_t nb = (B - b0 < slice) ? B - b0 : slice;
        int rc = launch_mlp(y + b0 * y_ld, y_ld, T, nb * T, W1, b1, W2, b2, W3, b3, lg, S, st);
        if (rc) return rc;
        float *d = dec + b0 * dec_ld, *fm = final_metric ? final_metric + b0 * S : nullptr;
        unsigned char *sv = surv + (size_t)b0 * T * SB;
        if (S == 16 && sw(SW_GENERIC_SWEEP) != '1' && sweep16_quad_surv_serves(lg, d, dec_ld, sv, T))
            rc = launch_sweep16_quad_surv<MODE_NEGLOGIT>(lg, d, dec_ld, fm, sv, nb, T, st);
        else
            rc = launch_sweep_surv<MODE_NEGLOGIT>(lg, 0, nullptr, 1, d, dec_ld, fm, sv, nb, T, S, st);
        if (rc) return rc;
    }
    return MVN_OK;
}

int mvn_traceback_f32(const uint8_t *surv, const float *final_metric, float *bits, int64_t bits_ld, int32_t *states, int64_t B,
                      int32_t T, int32_t S, mvn_stream_t stream) {
    if (B < 0 || T < 0 || bits_ld < T) return MVN_E_DIMS;
    if (!valid_states(S)) return MVN_E_STATES;
    if (B == 0 || T == 0) return MVN_OK;
    if (!surv || !final_metric || !bits) return MVN_E_NULL;
    return launch_traceback(surv, final_metric, bits, bits_ld, states, B, T, S, (hipStream_t)stream);
}

int mvn_acs_sweep_kernel_name(const float *cost, const float *dec, int64_t dec_ld, int64_t B, int32_t T, int32_t S,
                              char *name, int32_t name_len) {
    (void)T;
    if (!valid_states(S)) return MVN_E_STATES;
    if (!name || name_len < 1) return MVN_E_NULL;
    sweep_kernel_name<MODE_COST>(plan_sweep<MODE_COST>(cost, dec, dec_ld, B, S), S, dec, dec_ld, name, (size_t)name_len);
    return MVN_OK;
}

int mvn_va_decode_kernel_name(int64_t B, int32_t T, int32_t S, char *name, int32_t name_len) {
    (void)T;
    if (!valid_states(S)) return MVN_E_STATES;
    if (!name || name_len < 1) return MVN_E_NULL;
    sweep_kernel_name<MODE_VA>(plan_sweep<MODE_VA>(nullptr, nullptr, 0, B, S, T), S, nullptr, 0, name, (size_t)name_len);
    return MVN_OK;
}

int mvn_vnet_decode_kernel_name(int64_t B, int32_t T, int32_t S, int32_t want_logits, char *name, int32_t name_len) {
    if (!valid_states(S)) return MVN_E_STATES;
    if (!name || name_len < 1) return MVN_E_NULL;
    if (S == 16 && !unfused_forced()) {
        if (coop_selected(B, T))
            snprintf(name, (size_t)name_len, "vnet16_coop_kernel<%s>", want_logits ? "true" : "false");
        else if (const DealtPlan dp = dealt_plan(B, T); dp.groups)  // (given the workspace mvn_vnet_workspace_bytes asks for; without it the kernel below)
            snprintf(name, (size_t)name_len, "vnet16_dealt_kernel<%s> rings of %d", want_logits ? "true" : "false", dp.ring);
        else
            snprintf(name, (size_t)name_len, "vnet16_fusedn_kernel<%s, %d>", want_logits ? "true" : "false", fusedn_tiles());
    } else if (!want_logits && fused_ip_selected(S, B)) {  // one kernel: the MLP fused into the in-place sweep
        snprintf(name, (size_t)name_len, "vnet_fused_ip_kernel<%d>", log2_states(S) - 2);
    } else {  // two launches: the MLP, then the sweep over its logits (scratch or logits_out: 16-byte aligned, row stride T)
        char sw[64];
        sweep_kernel_name<MODE_NEGLOGIT>(plan_sweep<MODE_NEGLOGIT>(nullptr, nullptr, 0, B, S), S, nullptr, 0, sw, sizeof sw);
        snprintf(name, (size_t)name_len, "mlp_kernel<%d> + %s", (S + 15) / 16, sw);
    }
    return MVN_OK;
}

int mvn_va_decode_f32(const float *y, int64_t y_ld, const float *state_priors, int64_t Bp, float *dec,
                      int64_t dec_ld, float *final_metric, int64_t B, int32_t T, int32_t S,
                      mvn_stream_t stream) {
    if (B < 0 || T < 0 || dec_ld < T || y_ld < T) return MVN_E_DIMS;
    if (!valid_states(S)) return MVN_E_STATES;
    if (Bp < 1 || (B % Bp) != 0) return MVN_E_PRIORS;
    if (B == 0 || T == 0) return MVN_OK;
    if (!y || !state_priors || !dec) return MVN_E_NULL;
    return dispatch_sweep<MODE_VA>(y, y_ld, state_priors, Bp, dec, dec_ld, final_metric, B, T, S,
                                 (hipStream_t)stream);
}

int mvn_vnet_logits_f32(const float *y, const float *W1, const float *b1, const float *W2, const float *b2,
                        const float *W3, const float *b3, float *logits, int64_t N, int32_t S,
                        mvn_stream_t stream) {
    if (N < 0) return MVN_E_DIMS;
    if (!valid_states(S)) return MVN_E_STATES;
    if (N == 0) return MVN_OK;
    if (!y || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !logits) return MVN_E_NULL;
    const int T = N < (int64_t)1 << 30 ? (int)N : 1 << 30;
    return launch_mlp(y, T, T, N, W1, b1, W2, b2, W3, b3, logits, S, (hipStream_t)stream);
}

size_t mvn_vnet_workspace_bytes(int64_t B, int32_t T, int32_t S) {
    if (B <= 0 || T <= 0 || S <= 0) return 0;
    if (S == 16 && !unfused_forced()) return dealt_workspace_bytes_for(B, T);  // hand-off lines of the dealt kernel (<= 100 KB), or 0
    if (fused_ip_selected(S, B)) return 0;  // the fused kernels keep the logits on chip
    return (size_t)B * (size_t)T * (size_t)S * sizeof(float);
}

int mvn_vnet_decode_f32(const float *y, int64_t y_ld, const float *W1, const float *b1, const float *W2,
                        const float *b2, const float *W3, const float *b3, float *dec, int64_t dec_ld,
                        float *logits_out, float *final_metric, void *workspace, size_t workspace_bytes,
                        int64_t B, int32_t T, int32_t S, mvn_stream_t stream) {
    if (B < 0 || T < 0 || dec_ld < T || y_ld < T) return MVN_E_DIMS;
    if (!valid_states(S)) return MVN_E_STATES;
    if (B == 0 || T == 0) return MVN_OK;
    if (!y || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !dec) return MVN_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (S == 16 && !unfused_forced())  // fused single-kernel path: no scratch, 8 B/symbol of HBM traffic
        return launch_vnet16_fused(y, y_ld, W1, b1, W2, b2, W3, b3, dec, dec_ld, logits_out, final_metric, B, T, nullptr, 0,
                                   0, nullptr, nullptr, st, workspace, workspace_bytes);
    if (!logits_out && fused_ip_selected(S, B))  // other state counts: the MLP fused into the in-place sweep
        return launch_vnet_fused_ip(y, y_ld, W1, b1, W2, b2, W3, b3, dec, dec_ld, final_metric, B, T, S, st);
    const size_t per_block = (size_t)T * (size_t)S * sizeof(float);
    int64_t slice = B;
    float *buf = logits_out;
    if (!buf) {
        if (!workspace) return MVN_E_NULL;
        slice = (int64_t)(workspace_bytes / per_block);
        if (slice < 1) return MVN_E_WORKSPACE;
        if (slice > B) slice = B;
        buf = (float *)workspace;
    }
    for (int64_t b0 = 0; b0 < B; b0 += slice) {
        const int64_t nb = (B - b0 < slice) ? B - b0 : slice;
        float *lg = logits_out ? logits_out + (size_t)b0 * T * S : buf;
        int rc = launch_mlp(y + b0 * y_ld, y_ld, T, nb * T, W1, b1, W2, b2, W3, b3, lg, S, st);
        if (rc) return rc;
        const GuardWeights gw = {{W1, b1, W2, b2, W3, b3}, {kH1, kH1, kH2 * kH1, kH2, S * kH2, S}};
        rc = dispatch_sweep<MODE_NEGLOGIT>(lg, 0, nullptr, 1, dec + b0 * dec_ld, dec_ld,
                                         final_metric ? final_metric + b0 * S : nullptr, nb, T, S, st, &gw);
        if (rc) return rc;
    }
    return MVN_OK;
}

int mvn_vnet_decode_count_f32(const float *y, int64_t y_ld, const float *W1, const float *b1, const float *W2,
                              const float *b2, const float *W3, const float *b3, const float *tx, int64_t tx_ld,
                              int32_t K, const uint8_t *row_mask, int64_t *counters, float *dec, int64_t dec_ld,
                              void *workspace, size_t workspace_bytes, int64_t B, int32_t T, int32_t S, mvn_stream_t stream) {
    if (B < 0 || T < 0 || K < 0 || K > T || y_ld < T || tx_ld < K || (dec && dec_ld < T)) return MVN_E_DIMS;
    if (S != 16) return MVN_E_STATES;  // fused epilogue exists for the 16-state kernel only
    if (!counters) return MVN_E_NULL;
    if (B == 0 || T == 0) return MVN_OK;
    if (!y || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !tx) return MVN_E_NULL;
    return launch_vnet16_fused(y, y_ld, W1, b1, W2, b2, W3, b3, dec, dec_ld, nullptr, nullptr, B, T, tx, tx_ld, K,
                               row_mask, (unsigned long long *)counters, (hipStream_t)stream, workspace, workspace_bytes);
}

int mvn_vnet_online_train_f32(const float *y, const int32_t *labels, int32_t T, const int32_t *batch_idx, int32_t M,
                              int32_t n_iter, float *W1, float *b1, float *W2, float *b2, float *W3, float *b3,
                              float *adam_m, float *adam_v, int64_t step0, float lr, float beta1, float beta2, float eps,
                              float *loss_out, int32_t S, mvn_stream_t stream) {
    return mvn_vnet_online_train_ws_f32(y, labels, T, batch_idx, M, n_iter, W1, b1, W2, b2, W3, b3, adam_m, adam_v, step0, lr,
                                        beta1, beta2, eps, loss_out, S, nullptr, 0, nullptr, stream);
}

size_t mvn_vnet_train_workspace_bytes(int32_t S) {
    if (!valid_states(S) || S > 32) return 0;
    return maml_groups_workspace_bytes(S, kTrainMaxGroups);  // the larger of the two kernels' needs
}

size_t mvn_vnet_train_trials_workspace_bytes(int32_t S, int32_t T, int32_t W, int32_t R) {
    if (!valid_states(S) || S > 32 || T < 1 || R < 1) return 0;
    const int g = std::max(online_groups(T), std::max(maml_groups(T, W, 1), maml_groups(T, W, 0)));
    return g < 2 ? 0 : (size_t)R * trial_workspace_floats(S, g) * sizeof(float);
}

int mvn_vnet_online_train_ws_f32(const float *y, const int32_t *labels, int32_t T, const int32_t *batch_idx, int32_t M,
                                 int32_t n_iter, float *W1, float *b1, float *W2, float *b2, float *W3, float *b3,
                                 float *adam_m, float *adam_v, int64_t step0, float lr, float beta1, float beta2, float eps,
                                 float *loss_out, int32_t S, void *workspace, size_t workspace_bytes, int32_t *status,
                                 mvn_stream_t stream) {
    if (T < 1 || n_iter < 0 || step0 < 0 || (batch_idx && M < 1)) return MVN_E_DIMS;
    if (!valid_states(S) || S > 128) return MVN_E_STATES;  // parameters + gradient + a chunk must fit the 160-KB LDS
    if (n_iter == 0) return MVN_OK;
    if (!y || !labels || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !adam_m || !adam_v) return MVN_E_NULL;
    mvn_train_trial_t d = {};
    d.y = y;
    d.labels = labels;
    d.idx = batch_idx;
    float *w[6] = {W1, b1, W2, b2, W3, b3};
    for (int a = 0; a < 6; ++a) {
        d.w_in[a] = w[a];
        d.w_out[a] = w[a];
    }
    d.adam_m = adam_m;
    d.adam_v = adam_v;
    d.loss_out = loss_out;
    d.status = status;
    d.b1pow = pow((double)beta1, (double)step0);
    d.b2pow = pow((double)beta2, (double)step0);
    d.n = n_iter;
    return launch_online_train(d, nullptr, 1, T, batch_idx ? M : 0, lr, beta1, beta2, eps, S, workspace, workspace_bytes,
                               (hipStream_t)stream);
}

int mvn_vnet_online_train_trials_f32(const mvn_train_trial_t *trials, int32_t R, int32_t T, int32_t M, float lr, float beta1,
                                     float beta2, float eps, int32_t S, void *workspace, size_t workspace_bytes,
                                     mvn_stream_t stream) {
    if (T < 1 || R < 0 || M < 0) return MVN_E_DIMS;
    if (!valid_states(S) || S > 128) return MVN_E_STATES;
    if (R == 0) return MVN_OK;
    if (!trials) return MVN_E_NULL;
    return launch_online_train(mvn_train_trial_t{}, trials, R, T, M, lr, beta1, beta2, eps, S, workspace, workspace_bytes,
                               (hipStream_t)stream);
}

int mvn_vnet_maml_train_f32(const float *rx_words, const int32_t *labels, int32_t T, const int32_t *support_idx, int32_t W,
                            const int32_t *query_idx, int32_t n_steps, float *W1, float *b1, float *W2, float *b2,
                            float *W3, float *b3, float *adam_m, float *adam_v, int64_t step0, float meta_lr,
                            int32_t second_order, float lr, float beta1, float beta2, float eps, float *loss_out, int32_t S,
                            mvn_stream_t stream) {
    return mvn_vnet_maml_train_ws_f32(rx_words, labels, T, support_idx, W, query_idx, n_steps, W1, b1, W2, b2, W3, b3, adam_m,
                                      adam_v, step0, meta_lr, second_order, lr, beta1, beta2, eps, loss_out, S, nullptr, 0,
                                      nullptr, stream);
}

int mvn_vnet_maml_train_ws_f32(const float *rx_words, const int32_t *labels, int32_t T, const int32_t *support_idx, int32_t W,
                               const int32_t *query_idx, int32_t n_steps, float *W1, float *b1, float *W2, float *b2,
                               float *W3, float *b3, float *adam_m, float *adam_v, int64_t step0, float meta_lr,
                               int32_t second_order, float lr, float beta1, float beta2, float eps, float *loss_out, int32_t S,
                               void *workspace, size_t workspace_bytes, int32_t *status, mvn_stream_t stream) {
    if (T < 1 || W < 1 || n_steps < 0 || step0 < 0) return MVN_E_DIMS;
    if (!valid_states(S) || S > 32) return MVN_E_STATES;  // four parameter-sized vectors + a chunk must fit the 160-KB LDS
    if (n_steps == 0) return MVN_OK;
    if (!rx_words || !labels || !support_idx || !query_idx || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !adam_m || !adam_v)
        return MVN_E_NULL;
    mvn_train_trial_t d = {};
    d.y = rx_words;
    d.labels = labels;
    d.idx = support_idx;
    d.query_idx = query_idx;
    float *w[6] = {W1, b1, W2, b2, W3, b3};
    for (int a = 0; a < 6; ++a) {
        d.w_in[a] = w[a];
        d.w_out[a] = w[a];
    }
    d.adam_m = adam_m;
    d.adam_v = adam_v;
    d.loss_out = loss_out;
    d.status = status;
    d.b1pow = pow((double)beta1, (double)step0);
    d.b2pow = pow((double)beta2, (double)step0);
    d.n = n_steps;
    return launch_maml_train(d, nullptr, 1, T, W, meta_lr, second_order, lr, beta1, beta2, eps, S, workspace, workspace_bytes,
                             (hipStream_t)stream);
}

int mvn_vnet_maml_train_trials_f32(const mvn_train_trial_t *trials, int32_t R, int32_t T, int32_t W, float meta_lr,
                                   int32_t second_order, float lr, float beta1, float beta2, float eps, int32_t S,
                                   void *workspace, size_t workspace_bytes, mvn_stream_t stream) {
    if (T < 1 || W < 1 || R < 0) return MVN_E_DIMS;
    if (!valid_states(S) || S > 32) return MVN_E_STATES;
    if (R == 0) return MVN_OK;
    if (!trials) return MVN_E_NULL;
    return launch_maml_train(mvn_train_trial_t{}, trials, R, T, W, meta_lr, second_order, lr, beta1, beta2, eps, S, workspace,
                             workspace_bytes, (hipStream_t)stream);
}

int mvn_vnet_train_kernel_name(int32_t kind, int32_t R, int32_t T, int32_t M_or_W, int32_t S, size_t workspace_bytes, char *name,
                               int32_t name_len) {
    if (kind < 0 || kind > 2 || R < 0 || T < 1 || M_or_W < 0 || (kind > 0 && M_or_W < 1)) return MVN_E_DIMS;
    if (!valid_states(S) || S > (kind == 0 ? 128 : 32)) return MVN_E_STATES;
    if (!name || name_len < 1) return MVN_E_NULL;
    const bool many = R > 0;
    const int n_trials = many ? R : 1;
    const int groups = kind == 0 ? plan_online_groups(many, n_trials, T, M_or_W, S, workspace_bytes > 0, workspace_bytes)
                                 : plan_maml_groups(many, n_trials, T, M_or_W, kind == 2, S, workspace_bytes > 0, workspace_bytes);
    const int sc = S == 16 ? 16 : (S == 32 && !many) ? 32 : S > 32 ? S : 0;  // dispatch_states
    const char *base = kind == 0 ? "online_train" : "maml_train";
    if (!groups && kind == 0 && online_pair_form(many, n_trials, M_or_W, S)) {
        snprintf(name, (size_t)name_len, "%s_kernel<%d, true, %d> 1x%d", base, sc, kTrainThreads2, n_trials);
    } else if (!groups) {
        snprintf(name, (size_t)name_len, "%s_kernel<%d, %s> 1x%d", base, sc, many ? "true" : "false", n_trials);
    } else {
        const int per_launch = many ? trials_per_launch(n_trials, group_trials_fit(groups, current_device_cus(), n_trials)) : 1;
        const int launches = (n_trials + per_launch - 1) / per_launch;
        const char *place = xcd_grid(groups, current_device_cus(), n_trials, many) ? " one XCD per trial" : "";
        if (launches > 1)
            snprintf(name, (size_t)name_len, "%s_groups_kernel<%d, %s> %dx%d in %d launches%s", base, sc, many ? "true" : "false", groups,
                     per_launch, launches, place);
        else
            snprintf(name, (size_t)name_len, "%s_groups_kernel<%d, %s> %dx%d%s", base, sc, many ? "true" : "false", groups, per_launch, place);
    }
    return MVN_OK;
}

int mvn_vnet_byword_step_f32(const float *rx, int64_t rx_ld, const float *tx, int64_t tx_ld, const float *W1, const float *b1,
                             const float *W2, const float *b2, const float *W3, const float *b3, const int64_t *w_stride,
                             float *dec, int64_t dec_ld, float *msg, int64_t msg_ld, float *enc, int64_t enc_ld,
                             float *label_word, int64_t lw_ld, int32_t *labels, int64_t lab_ld, int32_t *nerr, int64_t R,
                             int32_t T, int32_t nsym, int32_t pilot, int32_t S, mvn_stream_t stream) {
    if (S != 16) return MVN_E_STATES;  // the fused step exists for the 16-state detector
    if (R < 0 || T < 8 || (T & 7) || T > kCoopMaxT || nsym < 1 || nsym > 8 || T / 8 <= nsym) return MVN_E_DIMS;
    const int K = T - 8 * nsym;
    if (rx_ld < T || tx_ld < K || (dec && dec_ld < T) || (msg && msg_ld < K) || (enc && enc_ld < T) ||
        (label_word && lw_ld < T) || (labels && lab_ld < T))
        return MVN_E_DIMS;
    if (R == 0) return MVN_OK;
    if (!tx || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || (!pilot && !rx)) return MVN_E_NULL;
    WeightStrides ws;
    for (int a = 0; a < 6; ++a) ws.s[a] = w_stride ? (long long)w_stride[a] : 0;
    const size_t dyn = pilot ? 0 : (size_t)((T + 15) / 16) * 1024;
    hipStream_t st = (hipStream_t)stream;
#define MVN_STEP_LAUNCH(NS)                                                                                              \
    do {                                                                                                                \
        int e = ensure_dynamic_lds((const void *)byword_step_kernel<NS>, (size_t)(kCoopMaxT / 16 * 1024));               \
        if (e) return e;                                                                                                \
        hipLaunchKernelGGL((byword_step_kernel<NS>), dim3((unsigned)R), dim3(64 * kCoopWaves), dyn, st, rx, rx_ld, tx,    \
                           tx_ld, W1, b1, W2, b2, W3, b3, ws, dec, dec_ld, msg, msg_ld, enc, enc_ld, label_word, lw_ld,   \
                           labels, lab_ld, nerr, T, nsym, pilot);                                                       \
    } while (0)
    if (nsym <= 2) MVN_STEP_LAUNCH(2);
    else MVN_STEP_LAUNCH(8);
#undef MVN_STEP_LAUNCH
    return (int)hipGetLastError();
}

int mvn_va_byword_step_f32(const float *rx, int64_t rx_ld, const float *tx, int64_t tx_ld, const float *state_priors, int64_t Bp,
                           float *dec, int64_t dec_ld, float *msg, int64_t msg_ld, float *enc, int64_t enc_ld, float *label_word,
                           int64_t lw_ld, int32_t *labels, int64_t lab_ld, int32_t *nerr, int64_t R, int32_t T, int32_t nsym,
                           int32_t pilot, int32_t S, mvn_stream_t stream) {
    if (S != 16) return MVN_E_STATES;  // the fused step exists for the 16-state detectors
    if (R < 0 || T < 8 || (T & 7) || T > kCoopMaxT || nsym < 1 || nsym > 8 || T / 8 <= nsym) return MVN_E_DIMS;
    const int K = T - 8 * nsym;
    if (rx_ld < T || tx_ld < K || (dec && dec_ld < T) || (msg && msg_ld < K) || (enc && enc_ld < T) ||
        (label_word && lw_ld < T) || (labels && lab_ld < T))
        return MVN_E_DIMS;
    if (Bp < 1) return MVN_E_PRIORS;
    if (R == 0) return MVN_OK;
    if (!tx || (!pilot && (!rx || !state_priors))) return MVN_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (nsym <= 2)
        hipLaunchKernelGGL((byword_step_va_kernel<2>), dim3((unsigned)R), dim3(64), 0, st, rx, rx_ld, tx, tx_ld, state_priors, Bp, dec,
                           dec_ld, msg, msg_ld, enc, enc_ld, label_word, lw_ld, labels, lab_ld, nerr, T, nsym, pilot);
    else
        hipLaunchKernelGGL((byword_step_va_kernel<8>), dim3((unsigned)R), dim3(64), 0, st, rx, rx_ld, tx, tx_ld, state_priors, Bp, dec,
                           dec_ld, msg, msg_ld, enc, enc_ld, label_word, lw_ld, labels, lab_ld, nerr, T, nsym, pilot);
    return (int)hipGetLastError();
}

void mvn_reload_switches(void) { load_switches(); }

#ifdef MVN_TEST_HOOKS
/* Test hooks -- compiled into the tests' build of the library only (-DMVN_TEST_HOOKS: libmvn_hip_hooks.so, built by
 * __graft_entry__.build_hip_hooks and loaded through MVN_LIB_PATH / _lib.load_variant); the shipped libmvn_hip.so does not
 * export this.  Spin limit of the training kernels' device-wide barrier, and phantom workgroups every barrier additionally
 * waits for (> 0 forces the give-up path); negative arguments leave a setting unchanged. */
void mvn_test_hooks(int64_t group_spin_limit, int32_t group_phantoms) {
    if (group_spin_limit >= 0) g_group_spin_limit.store((unsigned)group_spin_limit, std::memory_order_relaxed);
    if (group_phantoms >= 0) g_group_phantoms.store(group_phantoms, std::memory_order_relaxed);
}
/* The one-XCD barrier of the chunked training launches (train_groups.inc, groups_barrier's `local` branch): workgroup `k` of every
 * trial never stores its tag, so its peers must abandon the wait (k < 0: off).  Shares the launch field of the phantom arrivals. */
void mvn_test_hooks_skip_tag(int32_t k) { g_group_phantoms.store(k < 0 ? 0 : -(k + 1), std::memory_order_relaxed); }
/* The dealt ViterbiNet kernel's hand-off between rings: ring `skip_ring` never publishes (-1: all do), and the waiting ring
 * gives up after `spin_limit` polls (< 0: unchanged). */
void mvn_test_hooks_dealt(int32_t skip_ring, int64_t spin_limit) {
    g_dealt_skip_ring.store(skip_ring, std::memory_order_relaxed);
    if (spin_limit >= 0) g_dealt_spin_limit.store((unsigned)spin_limit, std::memory_order_relaxed);
}
#endif

int mvn_isi_awgn_transmit(const float *bits, int64_t ld_bits, int32_t K, const void *noise, int32_t noise_is_f64,
                          const double *h, int64_t Bh, double sigma, float *y, int64_t y_ld, int64_t B, int32_t T,
                          int32_t L, mvn_stream_t stream) {
    if (B < 0 || T < 0 || K < 0 || L < 1 || L > 16 || ld_bits < K || y_ld < T || Bh < 1) return MVN_E_DIMS;
    if (B == 0 || T == 0) return MVN_OK;
    if (!bits || !h || !y) return MVN_E_NULL;
    const int64_t n = B * (int64_t)T;
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (noise_is_f64)
        hipLaunchKernelGGL((isi_awgn_kernel<double>), dim3(grid), dim3(256), 0, st, bits, ld_bits, K, (const double *)noise,
                           h, Bh, sigma, y, y_ld, B, T, L);
    else
        hipLaunchKernelGGL((isi_awgn_kernel<float>), dim3(grid), dim3(256), 0, st, bits, ld_bits, K, (const float *)noise, h,
                           Bh, sigma, y, y_ld, B, T, L);
    return (int)hipGetLastError();
}

int mvn_generate_words_f32(float *tx, int64_t tx_ld, float *y, int64_t y_ld, const double *h, int64_t Bh, double sigma,
                           uint64_t seed, int64_t B, int32_t T, int32_t L, mvn_stream_t stream) {
    if (B < 0 || T < 0 || L < 1 || L > 16 || y_ld < T || (tx && tx_ld < T) || Bh < 1) return MVN_E_DIMS;
    if (B == 0 || T == 0) return MVN_OK;
    if (!y || !h) return MVN_E_NULL;
    const int64_t n = B * (int64_t)((T + 3) / 4);
    hipLaunchKernelGGL(generate_words_kernel, dim3((unsigned)((n + kGenThreads - 1) / kGenThreads)), dim3(kGenThreads), 0,
                       (hipStream_t)stream, tx, tx_ld, y, y_ld, h, Bh, sigma, (uint32_t)seed, (uint32_t)(seed >> 32), B, T, L);
    return (int)hipGetLastError();
}

int mvn_va_montecarlo_f32(const double *h, int64_t Bh, double sigma, uint64_t seed, const float *state_priors, int64_t Bp,
                          int64_t *counters, int64_t B, int32_t T, int32_t L, int32_t S, mvn_stream_t stream) {
    if (B < 0 || T < 0 || L < 1 || L > 16 || Bh < 1) return MVN_E_DIMS;
    if (S != 16 && S != 256) return MVN_E_STATES;  // the two classical-Viterbi trellises of BASELINE (configs[0], configs[3])
    if (Bp < 1) return MVN_E_PRIORS;
    if (!counters) return MVN_E_NULL;
    if (B == 0 || T == 0) return MVN_OK;
    if (!h || !state_priors) return MVN_E_NULL;
    return launch_va_montecarlo(h, Bh, sigma, seed, state_priors, Bp, (unsigned long long *)counters, B, T, L, S, (hipStream_t)stream);
}

int mvn_rs_decode_bits_f32(const float *rx_bits, int64_t ld_in, float *msg_bits, int64_t ld_out, int32_t *status,
                           int64_t B, int32_t nbits, int32_t nsym, mvn_stream_t stream) {
    if (B < 0 || nbits < 0 || (nbits % 8) != 0 || nsym < 1 || nsym > 64) return MVN_E_DIMS;
    const int n = nbits / 8;
    if (n <= nsym || n > 255 || ld_in < nbits || ld_out < nbits - 8 * nsym) return MVN_E_DIMS;
    if (B == 0) return MVN_OK;
    if (!rx_bits || !msg_bits) return MVN_E_NULL;
    const unsigned grid = (unsigned)((B + kRsThreads - 1) / kRsThreads);
    hipStream_t st = (hipStream_t)stream;
#define MVN_RS_DEC(NS) \
    hipLaunchKernelGGL((rs_decode_kernel<NS>), dim3(grid), dim3(kRsThreads), 0, st, rx_bits, ld_in, msg_bits, ld_out, status, B, n, nsym)
    if (nsym <= 2) MVN_RS_DEC(2);
    else if (nsym <= 8) MVN_RS_DEC(8);
    else if (nsym <= 16) MVN_RS_DEC(16);
    else if (nsym <= 32) MVN_RS_DEC(32);
    else MVN_RS_DEC(64);
#undef MVN_RS_DEC
    return (int)hipGetLastError();
}

int mvn_rs_encode_bits_f32(const float *msg_bits, int64_t ld_in, float *cw_bits, int64_t ld_out, int64_t B,
                           int32_t kbits, int32_t nsym, mvn_stream_t stream) {
    if (B < 0 || kbits < 8 || (kbits % 8) != 0 || nsym < 1 || nsym > 64) return MVN_E_DIMS;
    const int k = kbits / 8;
    if (k + nsym > 255 || ld_in < kbits || ld_out < kbits + 8 * nsym) return MVN_E_DIMS;
    if (B == 0) return MVN_OK;
    if (!msg_bits || !cw_bits) return MVN_E_NULL;
    const unsigned grid = (unsigned)((B + kRsThreads - 1) / kRsThreads);
    hipStream_t st = (hipStream_t)stream;
#define MVN_RS_ENC(NS) \
    hipLaunchKernelGGL((rs_encode_kernel<NS>), dim3(grid), dim3(kRsThreads), 0, st, msg_bits, ld_in, cw_bits, ld_out, B, k, nsym)
    if (nsym <= 2) MVN_RS_ENC(2);
    else if (nsym <= 8) MVN_RS_ENC(8);
    else if (nsym <= 16) MVN_RS_ENC(16);
    else if (nsym <= 32) MVN_RS_ENC(32);
    else MVN_RS_ENC(64);
#undef MVN_RS_ENC
    return (int)hipGetLastError();
}

int mvn_count_errors(const float *dec, int64_t dec_ld, const float *tx, int64_t tx_ld, const int64_t *rows,
                     int64_t n_rows, int32_t K, int64_t *counters, mvn_stream_t stream) {
    if (n_rows < 0 || K < 0 || dec_ld < K || tx_ld < K) return MVN_E_DIMS;
    if (!counters) return MVN_E_NULL;
    if (n_rows == 0 || K == 0) return MVN_OK;
    if (!dec || !tx) return MVN_E_NULL;
    // few, large workgroups: the two global atomics per workgroup (all on the same two words) are the serial part
    int64_t want = (n_rows + kCountWaves - 1) / kCountWaves;
    unsigned grid = (unsigned)(want < 512 ? want : 512);
    hipLaunchKernelGGL(count_errors_kernel, dim3(grid), dim3(64 * kCountWaves), 0, (hipStream_t)stream, dec, dec_ld, tx,
                       tx_ld, rows, n_rows, K, (unsigned long long *)counters);
    return (int)hipGetLastError();
}

}  // extern "C"
