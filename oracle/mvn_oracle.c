/*
 * mvn_oracle.c -- CPU ORACLE (test infrastructure, NOT a product path).
 *
 * Plain-C restatement of the Viterbi / ViterbiNet detection hot path of
 * tomerraviv95/meta-viterbinet.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the shipped detectors
 * never call it (they fail loudly when the HIP library is missing).
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against
 * golden vectors captured by importing the reference in the build container
 * (tests/golden/make_golden.py -> tests/golden/ npz files, tests/test_oracle_golden.py).
 *
 * Arithmetic contract (all fp32, one rounding per written operation, compiled
 * with -ffp-contract=off; fmaf() is a single-rounding fused multiply-add):
 *   - ACS stage, reference python_code/utils/trellis_utils.py:16-30:
 *       out[s] = min(in[p0]+c[p0], in[p1]+c[p1]),  p_j = (2s+j) % S   (table :7-13)
 *   - VA branch cost, python_code/detectors/VA/va_detector.py:64-68:
 *       d = y - prior[s]; c = (d*d)*0.5f - (float)log(sqrt(2*pi))
 *   - ViterbiNet MLP, python_code/detectors/VNET/vnet_detector.py:27-33,49, as
 *     torch 2.10 CPU evaluates it (MKL sgemm + ATen vectorised sigmoid), measured
 *     in the build container to be exactly:
 *       z1[k] = fmaf(y, w1[k], b1[k])
 *       h1[k] = 1 / (1 + expf_u10(0 - z1[k]))      expf_u10 = SLEEF 1.0-ULP expf
 *       z2[m] = (k-ordered fmaf chain from 0 over k=0..99 of h1[k]*W2[m][k]) + b2[m]
 *       h2[m] = max(z2[m], 0)
 *       lg[s] = (k-ordered fmaf chain from 0 over k=0..49 of h2[k]*W3[s][k]) + b3[s]
 *     (torch's scalar tail of <32 activations per thread chunk uses libm expf and may
 *      differ from this by 1 ulp of the sigmoid; see DESIGN.md "Arithmetic contract".)
 *   - decision, va_detector.py:93 / vnet_detector.py:55: dec[b,t] = argmin_s(in)%2
 *     taken BEFORE stage t is absorbed, first minimal index wins (torch.argmin).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define MVN_H1 100
#define MVN_H2 50
#define MVN_MAX_S 256

int mvn_oracle_version(void) { return 1; }

int mvn_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* trellis_utils.py:7-13 -- row s = [(2s)%S, (2s+1)%S] */
void mvn_oracle_transition_table(int S, int32_t *table) {
    for (int s = 0; s < S; ++s) {
        table[2 * s + 0] = (2 * s) % S;
        table[2 * s + 1] = (2 * s + 1) % S;
    }
}

/* torch.min(dim) over two candidates in index order j=0,1 (first minimal index wins,
 * NaN propagates: ATen min_kernel_impl keeps the first NaN it meets). */
static inline float min2_torch(float a, float b, int64_t *j) {
    if (a != a) { *j = 0; return a; }
    if (!(b >= a)) { *j = 1; return b; }
    *j = 0;
    return a;
}

/* trellis_utils.py:16-30, one stage.  argmin_j may be NULL. */
void mvn_oracle_acs_block(const float *in_prob, const float *llrs, float *out, int64_t *argmin_j,
                          int64_t B, int S) {
    for (int64_t b = 0; b < B; ++b) {
        const float *ip = in_prob + b * S, *c = llrs + b * S;
        float tmp[MVN_MAX_S];
        for (int s = 0; s < S; ++s) {
            int p0 = (2 * s) % S, p1 = (2 * s + 1) % S;
            int64_t j;
            tmp[s] = min2_torch(ip[p0] + c[p0], ip[p1] + c[p1], &j);
            if (argmin_j) argmin_j[b * S + s] = j;
        }
        memcpy(out + b * S, tmp, sizeof(float) * (size_t)S);
    }
}

/* torch.argmin over a row: first minimal index, first NaN wins. */
static inline int argmin_torch(const float *v, int S) {
    int idx = 0;
    float m = v[0];
    if (m != m) return 0;
    for (int s = 1; s < S; ++s) {
        float x = v[s];
        if (!(x >= m)) {
            m = x;
            idx = s;
            if (x != x) break;
        }
    }
    return idx;
}

static inline void acs_stage_inplace(float *m, const float *c, int S) {
    float a[MVN_MAX_S];
    for (int p = 0; p < S; ++p) a[p] = m[p] + c[p];
    for (int s = 0; s < S; ++s) {
        int64_t j;
        m[s] = min2_torch(a[(2 * s) % S], a[(2 * s + 1) % S], &j);
    }
}

static int check_dims(int64_t B, int T, int S) {
    if (B < 0 || T < 0) return -1;
    if (S < 2 || S > MVN_MAX_S || (S & (S - 1))) return -2;
    return 0;
}

/* The T-step loop of va_detector.py:89-97 / vnet_detector.py:53-59 over materialised
 * branch costs cost[b][t][s] (row strides: cost_ld floats per block, dec_ld per dec row).
 * dec columns >= T are left untouched (the caller zero-fills, as the reference does :90). */
int mvn_oracle_acs_sweep(const float *cost, float *dec, int64_t dec_ld, float *final_metric,
                         int64_t B, int T, int S) {
    int rc = check_dims(B, T, S);
    if (rc) return rc;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        float m[MVN_MAX_S];
        for (int s = 0; s < S; ++s) m[s] = 0.0f;
        const float *cb = cost + (size_t)b * (size_t)T * (size_t)S;
        for (int t = 0; t < T; ++t) {
            dec[b * dec_ld + t] = (float)(argmin_torch(m, S) % 2);
            acs_stage_inplace(m, cb + (size_t)t * S, S);
        }
        if (final_metric) memcpy(final_metric + b * S, m, sizeof(float) * (size_t)S);
    }
    return 0;
}

/* The same stage keeping torch.min's indices (trellis_utils.py:30: `return torch.min(...)` = (values, argmin_j)): bit (s & 7) of
 * sv[s >> 3] = j, the surviving predecessor of state s being (2s + j) % S.  The reference's callers drop them
 * (va_detector.py:95, vnet_detector.py:57); the survivor entry points of the HIP library store them. */
static inline void acs_stage_inplace_surv(float *m, const float *c, int S, uint8_t *sv) {
    float a[MVN_MAX_S];
    const int SB = S >= 8 ? S / 8 : 1;
    for (int p = 0; p < S; ++p) a[p] = m[p] + c[p];
    memset(sv, 0, (size_t)SB);
    for (int s = 0; s < S; ++s) {
        int64_t j;
        m[s] = min2_torch(a[(2 * s) % S], a[(2 * s + 1) % S], &j);
        sv[s >> 3] |= (uint8_t)(j << (s & 7));
    }
}

/* mvn_oracle_acs_sweep + the survivors surv[b][t][max(1, S/8)] */
int mvn_oracle_acs_sweep_surv(const float *cost, float *dec, int64_t dec_ld, float *final_metric, uint8_t *surv,
                              int64_t B, int T, int S) {
    int rc = check_dims(B, T, S);
    if (rc) return rc;
    const int SB = S >= 8 ? S / 8 : 1;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        float m[MVN_MAX_S];
        for (int s = 0; s < S; ++s) m[s] = 0.0f;
        const float *cb = cost + (size_t)b * (size_t)T * (size_t)S;
        for (int t = 0; t < T; ++t) {
            dec[b * dec_ld + t] = (float)(argmin_torch(m, S) % 2);
            acs_stage_inplace_surv(m, cb + (size_t)t * S, S, surv + ((size_t)b * T + t) * SB);
        }
        if (final_metric) memcpy(final_metric + b * S, m, sizeof(float) * (size_t)S);
    }
    return 0;
}

/* Textbook traceback over those survivors: sigma_T = torch.argmin(final_metric[b]); sigma_t = (2 sigma_{t+1} + j) % S;
 * bits[b][t] = sigma_t & 1 (the state before stage t pays cost[b][t][sigma_t]; its LSB is symbol t's bit, trellis_utils.py:33-46). */
int mvn_oracle_traceback(const uint8_t *surv, const float *final_metric, float *bits, int64_t bits_ld, int32_t *states,
                         int64_t B, int T, int S) {
    int rc = check_dims(B, T, S);
    if (rc) return rc;
    const int SB = S >= 8 ? S / 8 : 1;
    for (int64_t b = 0; b < B; ++b) {
        int s = argmin_torch(final_metric + b * S, S);
        for (int t = T - 1; t >= 0; --t) {
            const int j = (surv[((size_t)b * T + t) * SB + (s >> 3)] >> (s & 7)) & 1;
            s = (2 * s + j) % S;
            bits[b * bits_ld + t] = (float)(s & 1);
            if (states) states[(size_t)b * T + t] = s;
        }
    }
    return 0;
}

/* va_detector.py:64-68: cost[b][t][s] for rows of y; priors is [Bp][S], Bp in {1,B}
 * (the reference repeats the [W,S] table B//W times, i.e. row b uses priors[b % Bp]). */
#define MVN_LOG_SQRT_2PI_F ((float)0.91893853320467274178)
static inline float va_cost(float y, float prior) {
    float d = y - prior;
    float sq = d * d;
    float half = sq * 0.5f; /* torch '/ 2' on fp32 == exact halving */
    return half - MVN_LOG_SQRT_2PI_F;
}

int mvn_oracle_va_costs(const float *y, int64_t y_ld, const float *priors, int64_t Bp, float *cost,
                        int64_t B, int T, int S) {
    int rc = check_dims(B, T, S);
    if (rc) return rc;
    if (Bp < 1) return -3;
    for (int64_t b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t)
            for (int s = 0; s < S; ++s)
                cost[((size_t)b * T + t) * S + s] = va_cost(y[b * y_ld + t], priors[(b % Bp) * S + s]);
    return 0;
}

/* VADetector.forward('val'), va_detector.py:73-98, given the [Bp,S] state priors. */
int mvn_oracle_va_decode(const float *y, int64_t y_ld, const float *priors, int64_t Bp, float *dec,
                         int64_t dec_ld, float *final_metric, int64_t B, int T, int S) {
    int rc = check_dims(B, T, S);
    if (rc) return rc;
    if (Bp < 1) return -3;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        float m[MVN_MAX_S], c[MVN_MAX_S];
        const float *pr = priors + (b % Bp) * S;
        for (int s = 0; s < S; ++s) m[s] = 0.0f;
        for (int t = 0; t < T; ++t) {
            dec[b * dec_ld + t] = (float)(argmin_torch(m, S) % 2);
            float yt = y[b * y_ld + t];
            for (int s = 0; s < S; ++s) c[s] = va_cost(yt, pr[s]);
            acs_stage_inplace(m, c, S);
        }
        if (final_metric) memcpy(final_metric + b * S, m, sizeof(float) * (size_t)S);
    }
    return 0;
}

/* ---- SLEEF expf, 1.0-ULP variant (Sleef_expf*_u10), as ATen's Vectorized<float>::exp()
 * evaluates it inside torch.sigmoid on CPU (vnet_detector.py:29 nn.Sigmoid).  SLEEF is a
 * third-party dependency of torch (not vendored in the reference); this restates its
 * published algorithm: Cody-Waite reduction by ln2 (L2U/L2L split), degree-5 polynomial,
 * two-step ldexp. ---- */
static inline float pow2if(int q) {
    int32_t bits = (int32_t)((uint32_t)(q + 0x7f) << 23);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
static inline float ldexp2kf(float d, int e) { return d * pow2if(e >> 1) * pow2if(e - (e >> 1)); }

float mvn_oracle_expf_u10(float d) {
    /* clamp the reduction input so the int conversion is defined; results outside
     * [-104, 100] are overwritten below exactly as SLEEF does. */
    float dc = d < -128.0f ? -128.0f : (d > 128.0f ? 128.0f : d);
    float t = dc * 1.442695040888963407359924681001892137426645954152985934135449406931f;
    float qf = rintf(t);
    int q = (int)qf;
    float s = fmaf(qf, -0.693145751953125f, dc);
    s = fmaf(qf, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    u = ldexp2kf(u, q);
    if (d < -104.0f) u = 0.0f;
    if (d > 100.0f) u = INFINITY;
    if (d != d) u = d;
    return u;
}

float mvn_oracle_sigmoid(float z) { return 1.0f / (1.0f + mvn_oracle_expf_u10(0.0f - z)); }

void mvn_oracle_sigmoid_array(const float *x, float *out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = mvn_oracle_sigmoid(x[i]);
}

/* One symbol through the MLP.  W2t is [100][50] (k-major), W3t is [50][S]. */
static inline void mlp_symbol(float y, const float *w1, const float *b1, const float *W2t,
                              const float *b2, const float *W3t, const float *b3, int S,
                              float *logits) {
    float h1[MVN_H1], acc2[MVN_H2], h2[MVN_H2], acc3[MVN_MAX_S];
    for (int k = 0; k < MVN_H1; ++k) h1[k] = mvn_oracle_sigmoid(fmaf(y, w1[k], b1[k]));
    for (int m = 0; m < MVN_H2; ++m) acc2[m] = 0.0f;
    for (int k = 0; k < MVN_H1; ++k) {
        const float hk = h1[k];
        const float *w = W2t + k * MVN_H2;
        for (int m = 0; m < MVN_H2; ++m) acc2[m] = fmaf(hk, w[m], acc2[m]);
    }
    for (int m = 0; m < MVN_H2; ++m) {
        float z = acc2[m] + b2[m];
        h2[m] = (z != z) ? z : (z > 0.0f ? z : 0.0f);
    }
    for (int s = 0; s < S; ++s) acc3[s] = 0.0f;
    for (int k = 0; k < MVN_H2; ++k) {
        const float hk = h2[k];
        const float *w = W3t + k * S;
        for (int s = 0; s < S; ++s) acc3[s] = fmaf(hk, w[s], acc3[s]);
    }
    for (int s = 0; s < S; ++s) logits[s] = acc3[s] + b3[s];
}

static void transpose_weights(const float *W2, const float *W3, int S, float *W2t, float *W3t) {
    for (int m = 0; m < MVN_H2; ++m)
        for (int k = 0; k < MVN_H1; ++k) W2t[k * MVN_H2 + m] = W2[m * MVN_H1 + k];
    for (int s = 0; s < S; ++s)
        for (int k = 0; k < MVN_H2; ++k) W3t[k * S + s] = W3[s * MVN_H2 + k];
}

/* net(y.reshape(-1,1)), vnet_detector.py:49 / meta_vnet_detector.py:26-32.
 * Weights in torch layout: W1[100,1] b1[100] W2[50,100] b2[50] W3[S,50] b3[S]. */
int mvn_oracle_vnet_logits(const float *y, const float *W1, const float *b1, const float *W2,
                           const float *b2, const float *W3, const float *b3, float *logits,
                           int64_t N, int S) {
    int rc = check_dims(N, 0, S);
    if (rc) return rc;
    float W2t[MVN_H1 * MVN_H2];
    float *W3t = (float *)malloc(sizeof(float) * MVN_H2 * (size_t)S);
    if (!W3t) return -9;
    transpose_weights(W2, W3, S, W2t, W3t);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) mlp_symbol(y[n], W1, b1, W2t, b2, W3t, b3, S, logits + n * S);
    free(W3t);
    return 0;
}

/* VNETDetector.forward('val'), vnet_detector.py:35-61 (== META_VNETDetector 'val',
 * meta_vnet_detector.py:24-45, with var = parameters()).  logits_out ([B,T,S]) and
 * final_metric ([B,S]) are optional. */
int mvn_oracle_vnet_decode(const float *y, int64_t y_ld, const float *W1, const float *b1,
                           const float *W2, const float *b2, const float *W3, const float *b3,
                           float *dec, int64_t dec_ld, float *logits_out, float *final_metric,
                           int64_t B, int T, int S) {
    int rc = check_dims(B, T, S);
    if (rc) return rc;
    float W2t[MVN_H1 * MVN_H2];
    float *W3t = (float *)malloc(sizeof(float) * MVN_H2 * (size_t)S);
    if (!W3t) return -9;
    transpose_weights(W2, W3, S, W2t, W3t);
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        float m[MVN_MAX_S], lg[MVN_MAX_S], c[MVN_MAX_S];
        for (int s = 0; s < S; ++s) m[s] = 0.0f;
        for (int t = 0; t < T; ++t) {
            dec[b * dec_ld + t] = (float)(argmin_torch(m, S) % 2);
            mlp_symbol(y[b * y_ld + t], W1, b1, W2t, b2, W3t, b3, S, lg);
            if (logits_out) memcpy(logits_out + ((size_t)b * T + t) * S, lg, sizeof(float) * (size_t)S);
            for (int s = 0; s < S; ++s) c[s] = -lg[s];
            acs_stage_inplace(m, c, S);
        }
        if (final_metric) memcpy(final_metric + b * S, m, sizeof(float) * (size_t)S);
    }
    free(W3t);
    return 0;
}

/* metrics.py:7-17 as integer counters over the selected rows:
 * counters = {bit_errors, bits, frame_errors, frames}.  rows may be NULL (= all rows). */
int mvn_oracle_count_errors(const float *dec, int64_t dec_ld, const float *tx, int64_t tx_ld,
                            const int64_t *rows, int64_t n_rows, int K, int64_t *counters) {
    int64_t be = 0, fe = 0;
    for (int64_t i = 0; i < n_rows; ++i) {
        int64_t r = rows ? rows[i] : i;
        int64_t e = 0;
        for (int k = 0; k < K; ++k) {
            int64_t p = (int64_t)dec[r * dec_ld + k], q = (int64_t)tx[r * tx_ld + k]; /* .long() */
            if (p != q) ++e;
        }
        be += e;
        if (e) ++fe;
    }
    counters[0] = be;
    counters[1] = n_rows * (int64_t)K;
    counters[2] = fe;
    counters[3] = n_rows;
    return 0;
}

/* trellis_utils.py:33-46: state[t] = sum_i 2^i * b[t+i], word zero-padded by L. */
int mvn_oracle_calculate_states(const float *words, int64_t B, int T, int L, int64_t *states) {
    if (L < 1 || L > 8) return -2;
    for (int64_t b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            float st = 0.0f; /* torch.sum over fp32 products, then .long() (:44-45) */
            for (int i = 0; i < L; ++i) {
                int tt = t + i;
                float bit = tt < T ? words[b * T + tt] : 0.0f;
                st += bit * (float)(1 << i);
            }
            states[b * T + t] = (int64_t)st;
        }
    return 0;
}

/* =====================================================================================================
 * Reed-Solomon outer code over GF(2^8) (SURVEY 8f "next #2"): restatement of python_code/ecc/
 * (rs_main.py:9-37, rs_encoder.py:7-37, rs_decoder.py:37-218, polynomials_manipulation.py:85-125), the
 * "Reed-Solomon codes for coders" codec: prim 0x11d, generator 2, fcr 0, systematic, nsym parity bytes,
 * Berlekamp-Massey + brute-force root search + Forney.  Polynomials follow the reference's list
 * conventions (highest degree first unless noted).  Bits are packed MSB-first (np.packbits).
 * Pinned by tests/golden/g8_rs.npz.
 * ===================================================================================================== */
#define RS_MAX_N 255
#define RS_MAX_NSYM 64
static uint8_t rs_exp[512];
static uint8_t rs_log[256];
static int rs_ready = 0;

static void rs_init_tables(void) { /* polynomials_manipulation.py:85-110 */
    if (rs_ready) return;
    int x = 1;
    for (int i = 0; i < 255; ++i) {
        rs_exp[i] = (uint8_t)x;
        rs_log[x] = (uint8_t)i;
        x <<= 1;
        if (x & 0x100) x ^= 0x11d;
    }
    for (int i = 255; i < 512; ++i) rs_exp[i] = rs_exp[i - 255];
    rs_ready = 1;
}
static inline int rs_mul(int x, int y) { return (x == 0 || y == 0) ? 0 : rs_exp[rs_log[x] + rs_log[y]]; }
static inline int rs_inv(int x) { return rs_exp[255 - rs_log[x]]; }
static inline int rs_pow2(int power) { /* gf_pow(2, power), python % semantics */
    int e = power % 255;
    if (e < 0) e += 255;
    return rs_exp[e];
}
static inline int rs_div(int x, int y) { return x == 0 ? 0 : rs_exp[(rs_log[x] + 255 - rs_log[y]) % 255]; }
/* gf_poly_eval: Horner, highest degree first */
static int rs_poly_eval(const int *p, int len, int x) {
    int y = p[0];
    for (int i = 1; i < len; ++i) y = rs_mul(y, x) ^ p[i];
    return y;
}
/* gf_poly_mul: r has len p + len q - 1 */
static void rs_poly_mul(const int *p, int lp, const int *q, int lq, int *r) {
    for (int i = 0; i < lp + lq - 1; ++i) r[i] = 0;
    for (int j = 0; j < lq; ++j)
        for (int i = 0; i < lp; ++i) r[i + j] ^= rs_mul(p[i], q[j]);
}

/* rs_encoder.py:7-37 on bytes: msg[k] -> out[k+nsym] */
int mvn_oracle_rs_encode_bytes(const uint8_t *msg, int k, int nsym, uint8_t *out) {
    rs_init_tables();
    if (k < 1 || nsym < 1 || nsym > RS_MAX_NSYM || k + nsym > RS_MAX_N) return -1;
    int gen[RS_MAX_NSYM + 1], tmp[RS_MAX_NSYM + 2], glen = 1;
    gen[0] = 1;
    for (int i = 0; i < nsym; ++i) { /* rs_generator_poly */
        int f[2] = {1, rs_pow2(i)};
        rs_poly_mul(gen, glen, f, 2, tmp);
        glen += 1;
        memcpy(gen, tmp, sizeof(int) * (size_t)glen);
    }
    int buf[RS_MAX_N];
    for (int i = 0; i < k; ++i) buf[i] = msg[i];
    for (int i = k; i < k + nsym; ++i) buf[i] = 0;
    for (int i = 0; i < k; ++i) {
        int coef = buf[i];
        if (coef != 0)
            for (int j = 1; j < glen; ++j) buf[i + j] ^= rs_mul(gen[j], coef);
    }
    for (int i = 0; i < k; ++i) out[i] = msg[i];
    for (int i = k; i < k + nsym; ++i) out[i] = (uint8_t)buf[i];
    return 0;
}

/* rs_main.py:21-37 on bytes: rx[n] -> out[n-nsym]; returns 0 ok, 1 = "too many errors" branch (:31-33),
 * 2 = the reference would raise ValueError("Could not find error magnitude") (rs_decoder.py:123). */
int mvn_oracle_rs_decode_bytes(const uint8_t *rx, int n, int nsym, uint8_t *out) {
    rs_init_tables();
    if (nsym < 1 || nsym > RS_MAX_NSYM || n <= nsym || n > RS_MAX_N) return -1;
    const int k = n - nsym;
    int msg[RS_MAX_N];
    for (int i = 0; i < n; ++i) msg[i] = rx[i];
    /* rs_calc_syndromes (:37-47): synd = [0] + [eval(msg, 2^i)] */
    int synd[RS_MAX_NSYM + 1];
    synd[0] = 0;
    for (int i = 0; i < nsym; ++i) synd[i + 1] = rs_poly_eval(msg, n, rs_pow2(i));
    /* rs_find_error_locator (:140-205), no erasures; synd_shift = 1 */
    int err_loc[RS_MAX_NSYM + 2], old_loc[RS_MAX_NSYM + 3], new_loc[RS_MAX_NSYM + 3];
    int el = 1, ol = 1;
    err_loc[0] = 1;
    old_loc[0] = 1;
    for (int i = 0; i < nsym; ++i) {
        const int K = i + 1;
        int delta = synd[K];
        for (int j = 1; j < el; ++j) delta ^= rs_mul(err_loc[el - (j + 1)], synd[K - j]);
        old_loc[ol++] = 0; /* old_loc + [0] */
        if (delta != 0) {
            if (ol > el) {
                for (int t = 0; t < ol; ++t) new_loc[t] = rs_mul(old_loc[t], delta);
                const int inv = rs_inv(delta);
                for (int t = 0; t < el; ++t) old_loc[t] = rs_mul(err_loc[t], inv);
                const int nl = ol;
                ol = el;
                el = nl;
                memcpy(err_loc, new_loc, sizeof(int) * (size_t)el);
            }
            /* err_loc = gf_poly_add(err_loc, gf_poly_scale(old_loc, delta)): right-aligned xor */
            int rl = el > ol ? el : ol, r[RS_MAX_NSYM + 3];
            for (int t = 0; t < rl; ++t) r[t] = 0;
            for (int t = 0; t < el; ++t) r[t + rl - el] = err_loc[t];
            for (int t = 0; t < ol; ++t) r[t + rl - ol] ^= rs_mul(old_loc[t], delta);
            el = rl;
            memcpy(err_loc, r, sizeof(int) * (size_t)el);
        }
    }
    int lead = 0;
    while (lead < el && err_loc[lead] == 0) ++lead; /* drop leading zeros */
    const int errs = el - lead - 1;
    if (errs * 2 > nsym) { /* err_loc is None -> uncorrected systematic part */
        for (int i = 0; i < k; ++i) out[i] = rx[i];
        return 1;
    }
    /* rs_find_errors(err_loc[::-1], n) (:207-218) */
    int rev[RS_MAX_NSYM + 2], rl = el - lead;
    for (int t = 0; t < rl; ++t) rev[t] = err_loc[el - 1 - t];
    int pos[RS_MAX_N], npos = 0;
    for (int i = 0; i < n; ++i)
        if (rs_poly_eval(rev, rl, rs_pow2(i)) == 0) pos[npos++] = n - 1 - i;
    /* rs_correct_errata (:83-137) */
    int coef_pos[RS_MAX_N], loc[RS_MAX_N + 1], ll = 1, tmp[RS_MAX_N + 2];
    loc[0] = 1;
    for (int i = 0; i < npos; ++i) {
        coef_pos[i] = n - 1 - pos[i];
        int f[2] = {rs_pow2(coef_pos[i]), 1}; /* gf_poly_add([1],[a,0]) = [a,1] */
        rs_poly_mul(loc, ll, f, 2, tmp);
        ll += 1;
        memcpy(loc, tmp, sizeof(int) * (size_t)ll);
    }
    /* err_eval = (synd[::-1] * loc) mod x^(ll): the last ll coefficients; then reversed */
    int srev[RS_MAX_NSYM + 1], prod[RS_MAX_NSYM + RS_MAX_N + 2];
    for (int t = 0; t <= nsym; ++t) srev[t] = synd[nsym - t];
    rs_poly_mul(srev, nsym + 1, loc, ll, prod);
    const int pl = nsym + 1 + ll - 1;
    /* gf_poly_div by [1,0,...,0] (length ll+1): remainder = last ll coefficients (needs pl >= ll) */
    int ev[RS_MAX_N + 1]; /* err_eval[::-1] of the reference == remainder as is (highest degree first) */
    for (int t = 0; t < ll; ++t) ev[t] = prod[pl - ll + t];
    int X[RS_MAX_N];
    for (int i = 0; i < npos; ++i) X[i] = rs_pow2(-(255 - coef_pos[i]));
    int E[RS_MAX_N];
    for (int i = 0; i < n; ++i) E[i] = 0;
    for (int i = 0; i < npos; ++i) {
        const int Xi_inv = rs_inv(X[i]);
        int prime = 1;
        for (int j = 0; j < npos; ++j)
            if (j != i) prime = rs_mul(prime, 1 ^ rs_mul(Xi_inv, X[j]));
        int yv = rs_poly_eval(ev, ll, Xi_inv); /* gf_poly_eval(err_eval[::-1], Xi_inv) with err_eval = remainder[::-1] */
        yv = rs_mul(X[i], yv);
        if (prime == 0) {
            for (int t = 0; t < k; ++t) out[t] = rx[t];
            return 2;
        }
        E[pos[i]] = rs_div(yv, prime);
    }
    for (int i = 0; i < k; ++i) out[i] = (uint8_t)(msg[i] ^ E[i]);
    return 0;
}

/* bit-level wrappers (rs_main.py:9-37): fp32 {0,1} words, MSB-first packing */
int mvn_oracle_rs_encode_bits(const float *bits, int64_t ld_in, float *out, int64_t ld_out, int64_t B,
                              int kbits, int nsym) {
    if (kbits % 8) return -1;
    const int k = kbits / 8;
    for (int64_t b = 0; b < B; ++b) {
        uint8_t msg[RS_MAX_N], cw[RS_MAX_N];
        for (int i = 0; i < k; ++i) {
            int v = 0;
            for (int j = 0; j < 8; ++j) v = (v << 1) | ((int)bits[b * ld_in + 8 * i + j] & 1);
            msg[i] = (uint8_t)v;
        }
        int rc = mvn_oracle_rs_encode_bytes(msg, k, nsym, cw);
        if (rc) return rc;
        for (int i = 0; i < k + nsym; ++i)
            for (int j = 0; j < 8; ++j) out[b * ld_out + 8 * i + j] = (float)((cw[i] >> (7 - j)) & 1);
    }
    return 0;
}

int mvn_oracle_rs_decode_bits(const float *bits, int64_t ld_in, float *out, int64_t ld_out, int64_t B,
                              int nbits, int nsym, int32_t *status) {
    if (nbits % 8) return -1;
    const int n = nbits / 8;
    rs_init_tables();
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        uint8_t rx[RS_MAX_N], msg[RS_MAX_N];
        for (int i = 0; i < n; ++i) {
            int v = 0;
            for (int j = 0; j < 8; ++j) v = (v << 1) | ((int)bits[b * ld_in + 8 * i + j] & 1);
            rx[i] = (uint8_t)v;
        }
        int rc = mvn_oracle_rs_decode_bytes(rx, n, nsym, msg);
        if (status) status[b] = rc;
        for (int i = 0; i < n - nsym; ++i)
            for (int j = 0; j < 8; ++j) out[b * ld_out + 8 * i + j] = (float)((msg[i] >> (7 - j)) & 1);
    }
    return 0;
}
