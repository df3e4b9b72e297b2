// Micro-benchmark 3 (not shipped): do f32 MFMA and the sigmoid's VALU work overlap on one SIMD?
//
// One 1024-thread workgroup per CU (dynamic LDS padding keeps a second one out), so every SIMD holds exactly four
// waves.  Each wave reads its SIMD id from HW_ID, takes a slot number on that SIMD from an LDS counter, and the slot
// picks its ROLE for the launch: idle (exits), M (a long loop of independent v_mfma_f32_16x16x4_f32), or one of the V
// roles (a long loop of one VALU instruction mix).  All waves of the workgroup start together (one barrier); every
// wave stamps s_memtime around its loop and the workgroup's span max(end) - min(start) is reported as the median over
// the 256 workgroups of the median launch out of REPEATS, next to s_memrealtime (100 MHz) for the clock.
//   overlap  :  span(M M V V) ~= max(span(M M - -), span(- - V V))
//   additive :  span(M M V V) ~= span(M M - -) + span(- - V V)
// Second table: issue cost of every instruction form the fused kernel's sigmoid uses, at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)

enum Role { IDLE = 0, R_MFMA, R_SIGMOID, R_FMAC, R_INT, R_RCP, R_MOV, R_DPP, R_PERMSWAP, R_FMA3, R_FMAAK, R_MULLIT, R_LSHLADD,
            R_FMACLIT, R_LDSREAD, R_SIGMOID2, R_MFMA4x4, R_K0, R_K1, R_K2, R_MFMA_DIFF, R_K3, R_K4, R_PKMUL, R_PKADD, R_VCMP, R_VCMPS, R_SALU, R_MIN3, R_CNDMASK, R_MFMA2, R_MFMA1, N_ROLES };
static const char* kRoleName[N_ROLES] = {"-", "M", "Vsig", "Vfmac", "Vint", "Vrcp", "Vmov", "Vdpp", "Vswap", "Vfma3", "Vfmaak",
                                         "Vmullit", "Vlshladd", "Vfmaclit", "Vlds", "Vsig2", "M4x4", "K0", "K1", "K2", "Mdiff", "K3bar", "K4bar2", "Vpkmul", "Vpkadd", "Vcmp", "VcmpS", "Salu", "Vmin3", "Vcndmask", "M2acc", "M1acc"};
// wave-instructions per loop iteration of each role (for cycles/instruction)
static const int kOpsPerIter[N_ROLES] = {0, 16, 80, 64, 64, 32, 64, 64, 32, 64, 64, 64, 64, 64, 32, 160, 32, 92, 92, 92, 16, 92, 184, 64, 64, 64, 64, 64, 64, 64, 16, 16};

#define SIG1(D, Y)                                                                                                     \
    "v_fma_f32 %[t0], " Y ", %[a], %[b]\n v_mul_f32 %[t1], 0x3fb8aa3b, %[t0]\n v_add_f32 %[t1], 0x4b400000, %[t1]\n"      \
    "v_add_f32 %[t2], 0xcb400000, %[t1]\n v_fmac_f32 %[t0], 0xbf317200, %[t2]\n v_fmac_f32 %[t0], 0xb5bfbe8e, %[t2]\n"    \
    "v_fmamk_f32 %[t2], %[t0], 0x39502bda, %[c]\n v_fmaak_f32 %[t2], %[t2], %[t0], 0x3c0888a6\n"                          \
    "v_fmaak_f32 %[t2], %[t2], %[t0], 0x3d2aaa7a\n v_fmaak_f32 %[t2], %[t2], %[t0], 0x3e2aaaab\n"                         \
    "v_mul_f32 %[t3], %[t0], %[t0]\n v_fma_f32 %[t2], %[t2], %[t0], 0.5\n v_fmac_f32 %[t0], %[t3], %[t2]\n"               \
    "v_add_f32 %[t0], 1.0, %[t0]\n v_lshl_add_u32 %[t0], %[t1], 23, %[t0]\n v_add_f32 %[t0], 1.0, %[t0]\n"               \
    "v_rcp_f32 " D ", %[t0]\n s_nop 0\n v_fma_f32 %[t0], -%[t0], " D ", 1.0\n v_fmac_f32 " D ", %[t0], " D "\n"

struct Stamp { unsigned long long t0, t1, r0, r1; unsigned hwid, role; };

__global__ __launch_bounds__(1024) void kroles(Stamp* out, float* sink, const int* roles /*[4] per slot*/, const int* iters /*[N_ROLES]*/) {
    extern __shared__ float dyn[];
    __shared__ int slot_cnt[4];
    __shared__ float ldsbuf[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 4) slot_cnt[threadIdx.x] = 0;
    ldsbuf[threadIdx.x] = threadIdx.x * 0.25f;
    __syncthreads();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const int simd = (hwid >> 4) & 3;
    int slot = 0;
    if (lane == 0) slot = atomicAdd(&slot_cnt[simd], 1);
    slot = __builtin_amdgcn_readfirstlane(slot);
    const int role = slot < 4 ? roles[slot] : IDLE;
    const int n = iters[role];
    float x0 = 1.0f + lane * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const float a = 1.0001f, b = 0.5f, c100 = 1.98527617612853646278381e-4f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == R_MFMA) {
        for (int it = 0; it < n; ++it) {
            REP4(c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c1, 0, 0, 0);
                 c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c3, 0, 0, 0);)
        }
    } else if (role == R_MFMA2) {  // two accumulators alternating (the training kernels' tile product): dependent distance 2
        for (int it = 0; it < n; ++it) {
            REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c1, 0, 0, 0);)
        }
    } else if (role == R_MFMA1) {  // one accumulator: every MFMA waits for its predecessor
        for (int it = 0; it < n; ++it) {
            REP16(c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c0, 0, 0, 0);)
        }
    } else if (role == R_MFMA4x4) {
        for (int it = 0; it < n; ++it) {
            REP8(c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, c1, 0, 0, 0);
                 c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, c3, 0, 0, 0);)
        }
    } else if (role == R_SIGMOID || role == R_SIGMOID2) {
        // the fused kernel's fast sigmoid, instruction for instruction (20 per activation; a dependent chain).
        // R_SIGMOID: 4 activations one after the other (like the kernel: sched_barrier keeps them apart) = 80 instr/iter
        // R_SIGMOID2: two chains interleaved instruction by instruction, 8 activations = 160 instr/iter
        float t0_, t1_, t2_, t3_, u0_, u1_, u2_, u3_;
        if (role == R_SIGMOID) {
            for (int it = 0; it < n; ++it) {
                asm volatile(SIG1("%[d0]", "%[y0]") SIG1("%[d1]", "%[y1]") SIG1("%[d2]", "%[y2]") SIG1("%[d3]", "%[y3]")
                             : [d0] "=&v"(x4), [d1] "=&v"(x5), [d2] "=&v"(x6), [d3] "=&v"(x7), [t0] "=&v"(t0_), [t1] "=&v"(t1_),
                               [t2] "=&v"(t2_), [t3] "=&v"(t3_)
                             : [y0] "v"(x0), [y1] "v"(x1), [y2] "v"(x2), [y3] "v"(x3), [a] "v"(a), [b] "v"(b), [c] "v"(c100));
            }
        } else {
#define SIG2(D, Y, E, Z)                                                                                                \
    "v_fma_f32 %[t0], " Y ", %[a], %[b]\n v_fma_f32 %[u0], " Z ", %[a], %[b]\n"                                           \
    "v_mul_f32 %[t1], 0x3fb8aa3b, %[t0]\n v_mul_f32 %[u1], 0x3fb8aa3b, %[u0]\n"                                           \
    "v_add_f32 %[t1], 0x4b400000, %[t1]\n v_add_f32 %[u1], 0x4b400000, %[u1]\n"                                           \
    "v_add_f32 %[t2], 0xcb400000, %[t1]\n v_add_f32 %[u2], 0xcb400000, %[u1]\n"                                           \
    "v_fmac_f32 %[t0], 0xbf317200, %[t2]\n v_fmac_f32 %[u0], 0xbf317200, %[u2]\n"                                         \
    "v_fmac_f32 %[t0], 0xb5bfbe8e, %[t2]\n v_fmac_f32 %[u0], 0xb5bfbe8e, %[u2]\n"                                         \
    "v_fmamk_f32 %[t2], %[t0], 0x39502bda, %[c]\n v_fmamk_f32 %[u2], %[u0], 0x39502bda, %[c]\n"                           \
    "v_fmaak_f32 %[t2], %[t2], %[t0], 0x3c0888a6\n v_fmaak_f32 %[u2], %[u2], %[u0], 0x3c0888a6\n"                         \
    "v_fmaak_f32 %[t2], %[t2], %[t0], 0x3d2aaa7a\n v_fmaak_f32 %[u2], %[u2], %[u0], 0x3d2aaa7a\n"                         \
    "v_fmaak_f32 %[t2], %[t2], %[t0], 0x3e2aaaab\n v_fmaak_f32 %[u2], %[u2], %[u0], 0x3e2aaaab\n"                         \
    "v_mul_f32 %[t3], %[t0], %[t0]\n v_mul_f32 %[u3], %[u0], %[u0]\n"                                                     \
    "v_fma_f32 %[t2], %[t2], %[t0], 0.5\n v_fma_f32 %[u2], %[u2], %[u0], 0.5\n"                                           \
    "v_fmac_f32 %[t0], %[t3], %[t2]\n v_fmac_f32 %[u0], %[u3], %[u2]\n"                                                   \
    "v_add_f32 %[t0], 1.0, %[t0]\n v_add_f32 %[u0], 1.0, %[u0]\n"                                                         \
    "v_lshl_add_u32 %[t0], %[t1], 23, %[t0]\n v_lshl_add_u32 %[u0], %[u1], 23, %[u0]\n"                                   \
    "v_add_f32 %[t0], 1.0, %[t0]\n v_add_f32 %[u0], 1.0, %[u0]\n"                                                         \
    "v_rcp_f32 " D ", %[t0]\n v_rcp_f32 " E ", %[u0]\n"                                                                   \
    "v_fma_f32 %[t0], -%[t0], " D ", 1.0\n v_fma_f32 %[u0], -%[u0], " E ", 1.0\n"                                         \
    "v_fmac_f32 " D ", %[t0], " D "\n v_fmac_f32 " E ", %[u0], " E "\n"
            for (int it = 0; it < n; ++it) {
                asm volatile(SIG2("%[d0]", "%[y0]", "%[d1]", "%[y1]") SIG2("%[d2]", "%[y2]", "%[d3]", "%[y3]")
                             SIG2("%[d0]", "%[y1]", "%[d1]", "%[y2]") SIG2("%[d2]", "%[y3]", "%[d3]", "%[y0]")
                             : [d0] "=&v"(x4), [d1] "=&v"(x5), [d2] "=&v"(x6), [d3] "=&v"(x7), [t0] "=&v"(t0_), [t1] "=&v"(t1_),
                               [t2] "=&v"(t2_), [t3] "=&v"(t3_), [u0] "=&v"(u0_), [u1] "=&v"(u1_), [u2] "=&v"(u2_), [u3] "=&v"(u3_)
                             : [y0] "v"(x0), [y1] "v"(x1), [y2] "v"(x2), [y3] "v"(x3), [a] "v"(a), [b] "v"(b), [c] "v"(c100));
            }
        }
    } else if (role == R_MFMA_DIFF) {  // like M, but every MFMA has its own A and B registers
        for (int it = 0; it < n; ++it) {
            REP2(c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x2, x3, c1, 0, 0, 0);
                 c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x4, x5, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x6, x7, c3, 0, 0, 0);
                 c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, x2, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x3, x4, c1, 0, 0, 0);
                 c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x5, x6, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x7, x0, c3, 0, 0, 0);)
        }
    } else if (role == R_K0 || role == R_K1 || role == R_K2 || role == R_K3 || role == R_K4) {
        // the fused kernel's k-step pattern: 4 x (sigmoid chain, 3 MFMAs).  K0: the MFMAs' B operand is the sigmoid just
        // computed (as in the kernel); K1: fixed operands (no VALU -> MFMA dependence); K2: 4 sigmoids, then 12 MFMAs.
        f32x4 d0 = c0, d1 = c0, d2 = c0, d3 = c0, e0 = c0, e1 = c0, e2 = c0, e3 = c0;
        float t0_, t1_, t2_, t3_;
        for (int it = 0; it < n; ++it) {
#define KSIG(D, Y) asm volatile(SIG1("%[d0]", "%[y0]") : [d0] "=&v"(D), [t0] "=&v"(t0_), [t1] "=&v"(t1_), [t2] "=&v"(t2_), [t3] "=&v"(t3_) \
                                : [y0] "v"(Y), [a] "v"(a), [b] "v"(b), [c] "v"(c100))
#define KMF(A0, A1, A2, Bv, C0, C1, C2) C0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0, Bv, C0, 0, 0, 0); \
            C1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1, Bv, C1, 0, 0, 0); C2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A2, Bv, C2, 0, 0, 0)
            if (role == R_K0) {
                KSIG(x4, x0); __builtin_amdgcn_sched_barrier(0); KMF(x1, x2, x3, x4, c0, c1, c2); __builtin_amdgcn_sched_barrier(0);
                KSIG(x5, x1); __builtin_amdgcn_sched_barrier(0); KMF(x1, x2, x3, x5, c3, d0, d1); __builtin_amdgcn_sched_barrier(0);
                KSIG(x6, x2); __builtin_amdgcn_sched_barrier(0); KMF(x1, x2, x3, x6, d2, d3, e0); __builtin_amdgcn_sched_barrier(0);
                KSIG(x7, x3); __builtin_amdgcn_sched_barrier(0); KMF(x1, x2, x3, x7, e1, e2, e3); __builtin_amdgcn_sched_barrier(0);
            } else if (role == R_K1) {
                KSIG(x4, x0); __builtin_amdgcn_sched_barrier(0); KMF(x1, x2, x3, x0, c0, c1, c2); __builtin_amdgcn_sched_barrier(0);
                KSIG(x5, x1); __builtin_amdgcn_sched_barrier(0); KMF(x1, x2, x3, x0, c3, d0, d1); __builtin_amdgcn_sched_barrier(0);
                KSIG(x6, x2); __builtin_amdgcn_sched_barrier(0); KMF(x1, x2, x3, x0, d2, d3, e0); __builtin_amdgcn_sched_barrier(0);
                KSIG(x7, x3); __builtin_amdgcn_sched_barrier(0); KMF(x1, x2, x3, x0, e1, e2, e3); __builtin_amdgcn_sched_barrier(0);
            } else if (role == R_K3) {  // phases separated by workgroup barriers (every wave of the CU runs this role)
                KSIG(x4, x0); KSIG(x5, x1); KSIG(x6, x2); KSIG(x7, x3); __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
                KMF(x1, x2, x3, x4, c0, c1, c2); KMF(x1, x2, x3, x5, c3, d0, d1); KMF(x1, x2, x3, x6, d2, d3, e0);
                KMF(x1, x2, x3, x7, e1, e2, e3); __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
            } else if (role == R_K4) {  // two k-steps per phase: 8 sigmoids | barrier | 24 MFMAs | barrier
                float z0, z1, z2, z3;
                KSIG(x4, x0); KSIG(x5, x1); KSIG(x6, x2); KSIG(x7, x3); KSIG(z0, x3); KSIG(z1, x2); KSIG(z2, x1); KSIG(z3, x0);
                __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
                KMF(x1, x2, x3, x4, c0, c1, c2); KMF(x1, x2, x3, x5, c3, d0, d1); KMF(x1, x2, x3, x6, d2, d3, e0);
                KMF(x1, x2, x3, x7, e1, e2, e3);
                KMF(x1, x2, x3, z0, c0, c1, c2); KMF(x1, x2, x3, z1, c3, d0, d1); KMF(x1, x2, x3, z2, d2, d3, e0);
                KMF(x1, x2, x3, z3, e1, e2, e3); __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
            } else {
                KSIG(x4, x0); KSIG(x5, x1); KSIG(x6, x2); KSIG(x7, x3); __builtin_amdgcn_sched_barrier(0);
                KMF(x1, x2, x3, x4, c0, c1, c2); KMF(x1, x2, x3, x5, c3, d0, d1); KMF(x1, x2, x3, x6, d2, d3, e0);
                KMF(x1, x2, x3, x7, e1, e2, e3); __builtin_amdgcn_sched_barrier(0);
            }
        }
        c0 += d0 + d1 + d2 + d3 + e0 + e1 + e2 + e3;
    } else if (role == R_FMAC) {  // VOP2, all VGPR, 8 independent chains
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
        }
    } else if (role == R_INT) {  // integer VOP2 (no FP32 multiplier involved)
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("v_add_u32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n"
                              "v_add_u32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));)
        }
    } else if (role == R_RCP) {
        for (int it = 0; it < n; ++it) {
            REP4(asm volatile("v_rcp_f32 %0, %8\n v_rcp_f32 %1, %8\n v_rcp_f32 %2, %8\n v_rcp_f32 %3, %8\n"
                              "v_rcp_f32 %4, %8\n v_rcp_f32 %5, %8\n v_rcp_f32 %6, %8\n v_rcp_f32 %7, %8\n"
                              : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(a));)
        }
    } else if (role == R_MOV) {
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                              "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                              : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(a));)
        }
    } else if (role == R_DPP) {
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("v_min_f32_dpp %0, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_min_f32_dpp %1, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              "v_min_f32_dpp %2, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                              "v_min_f32_dpp %3, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                              "v_min_f32_dpp %4, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_min_f32_dpp %5, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              "v_min_f32_dpp %6, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                              "v_min_f32_dpp %7, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                              : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7) : "v"(a));)
        }
    } else if (role == R_PERMSWAP) {
        for (int it = 0; it < n; ++it) {
            REP4(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                              "v_permlane32_swap_b32 %1, %2\n v_permlane16_swap_b32 %3, %4\n v_permlane32_swap_b32 %5, %6\n v_permlane16_swap_b32 %7, %0\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
        }
    } else if (role == R_FMA3) {  // VOP3 v_fma_f32 with three VGPR sources
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
        }
    } else if (role == R_FMAAK) {
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("v_fmaak_f32 %0, %8, %0, 0x3f000000\n v_fmaak_f32 %1, %8, %1, 0x3f000000\n v_fmaak_f32 %2, %8, %2, 0x3f000000\n"
                              "v_fmaak_f32 %3, %8, %3, 0x3f000000\n v_fmaak_f32 %4, %8, %4, 0x3f000000\n v_fmaak_f32 %5, %8, %5, 0x3f000000\n"
                              "v_fmaak_f32 %6, %8, %6, 0x3f000000\n v_fmaak_f32 %7, %8, %7, 0x3f000000\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));)
        }
    } else if (role == R_MULLIT) {  // VOP2 with a 32-bit literal source
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %0\n v_add_f32 %1, 0x4b400000, %1\n v_mul_f32 %2, 0x3fb8aa3b, %2\n v_add_f32 %3, 0x4b400000, %3\n"
                              "v_mul_f32 %4, 0x3fb8aa3b, %4\n v_add_f32 %5, 0x4b400000, %5\n v_mul_f32 %6, 0x3fb8aa3b, %6\n v_add_f32 %7, 0x4b400000, %7\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
        }
    } else if (role == R_LSHLADD) {
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("v_lshl_add_u32 %0, %8, 23, %0\n v_lshl_add_u32 %1, %8, 23, %1\n v_lshl_add_u32 %2, %8, 23, %2\n v_lshl_add_u32 %3, %8, 23, %3\n"
                              "v_lshl_add_u32 %4, %8, 23, %4\n v_lshl_add_u32 %5, %8, 23, %5\n v_lshl_add_u32 %6, %8, 23, %6\n v_lshl_add_u32 %7, %8, 23, %7\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));)
        }
    } else if (role == R_FMACLIT) {
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("v_fmac_f32 %0, 0xbf317200, %8\n v_fmac_f32 %1, 0xb5bfbe8e, %8\n v_fmac_f32 %2, 0xbf317200, %8\n v_fmac_f32 %3, 0xb5bfbe8e, %8\n"
                              "v_fmac_f32 %4, 0xbf317200, %8\n v_fmac_f32 %5, 0xb5bfbe8e, %8\n v_fmac_f32 %6, 0xbf317200, %8\n v_fmac_f32 %7, 0xb5bfbe8e, %8\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));)
        }
    } else if (role == R_PKMUL || role == R_PKADD) {  // packed f32: two lanes of arithmetic per instruction
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 p0 = {x0, x1}, p1 = {x1, x2}, p2 = {x2, x3}, p3 = {x3, x4}, p4 = {x4, x5}, p5 = {x5, x6}, p6 = {x6, x7}, p7 = {x7, x0};
        const f32x2 pa = {a, b};
        for (int it = 0; it < n; ++it) {
            if (role == R_PKMUL) {
                REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                                  "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                                  : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));)
            } else {
                REP8(asm volatile("v_pk_add_f32 %0, %0, %8 op_sel_hi:[1,0]\n v_pk_add_f32 %1, %1, %8 op_sel_hi:[1,0]\n v_pk_add_f32 %2, %2, %8 op_sel_hi:[1,0]\n v_pk_add_f32 %3, %3, %8 op_sel_hi:[1,0]\n"
                                  "v_pk_add_f32 %4, %4, %8 op_sel_hi:[1,0]\n v_pk_add_f32 %5, %5, %8 op_sel_hi:[1,0]\n v_pk_add_f32 %6, %6, %8 op_sel_hi:[1,0]\n v_pk_add_f32 %7, %7, %8 op_sel_hi:[1,0]\n"
                                  : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));)
            }
        }
        x0 = p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
    } else if (role == R_VCMP) {  // VOPC into vcc
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("v_cmp_eq_f32 vcc, %0, %1\n v_cmp_eq_f32 vcc, %1, %2\n v_cmp_eq_f32 vcc, %2, %3\n v_cmp_eq_f32 vcc, %3, %4\n"
                              "v_cmp_eq_f32 vcc, %4, %5\n v_cmp_eq_f32 vcc, %5, %6\n v_cmp_eq_f32 vcc, %6, %7\n v_cmp_eq_f32 vcc, %7, %0\n"
                              :: "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7) : "vcc");)
        }
    } else if (role == R_VCMPS) {  // VOP3 compare into an SGPR pair, each followed by the scalar use of the mask
        unsigned long long acc = 0;
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("v_cmp_eq_f32 s[40:41], %1, %2\n v_cmp_eq_f32 s[42:43], %2, %3\n v_cmp_eq_f32 s[44:45], %3, %4\n v_cmp_eq_f32 s[46:47], %4, %5\n"
                              "v_cmp_eq_f32 s[48:49], %5, %6\n v_cmp_eq_f32 s[50:51], %6, %7\n v_cmp_eq_f32 s[52:53], %7, %8\n v_cmp_eq_f32 s[54:55], %8, %1\n"
                              "s_or_b64 %0, %0, s[40:41]\n s_or_b64 %0, %0, s[54:55]\n"
                              : "+s"(acc) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7)
                              : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");)
        }
        x0 += (float)(acc & 1);
    } else if (role == R_SALU) {  // dependent and independent scalar work, no VALU at all
        unsigned long long s0 = hwid, s1 = hwid + 1, s2 = hwid + 2, s3 = hwid + 3;
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %2\n s_xor_b64 %2, %2, %3\n s_andn2_b64 %3, %3, %0\n"
                              "s_and_b64 %0, %0, %2\n s_or_b64 %1, %1, %3\n s_xor_b64 %2, %2, %0\n s_andn2_b64 %3, %3, %1\n"
                              : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));)
        }
        x0 += (float)((s0 ^ s1 ^ s2 ^ s3) & 1);
    } else if (role == R_MIN3) {
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                              "v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
        }
    } else if (role == R_CNDMASK) {
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");)
        }
    } else if (role == R_LDSREAD) {  // ds_read_b128 stream (the kernel's weight fetches)
        const unsigned addr = (unsigned)(size_t)(&ldsbuf[0]) + (lane & 15) * 16;
        f32x4 v0, v1, v2, v3;
        for (int it = 0; it < n; ++it) {
            REP8(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:256\n ds_read_b128 %2, %4 offset:512\n ds_read_b128 %3, %4 offset:768\n"
                              "s_waitcnt lgkmcnt(0)\n"
                              : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr) : "memory");)
            x0 += v0[0] + v1[1] + v2[2] + v3[3];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        Stamp s{t0, t1, r0, r1, hwid, (unsigned)role};
        out[blockIdx.x * 16 + wave] = s;
    }
    sink[blockIdx.x * 1024 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + c0[0] + c1[1] + c2[2] + c3[3] + dyn[lane];
}

struct Result { double span_cyc, span_us, wall_ms, ghz; double role_cyc[4]; bool placement_ok; };

static Result run(const int roles[4], const int* iters_host, int repeats = 21) {
    static Stamp* d_out = nullptr; static float* d_sink; static int *d_roles, *d_iters;
    const int nwg = 256;
    if (!d_out) {
        CHECK(hipMalloc(&d_out, nwg * 16 * sizeof(Stamp))); CHECK(hipMalloc(&d_sink, nwg * 1024 * 4));
        CHECK(hipMalloc(&d_roles, 16)); CHECK(hipMalloc(&d_iters, N_ROLES * 4));
        CHECK(hipFuncSetAttribute((const void*)kroles, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    }
    CHECK(hipMemcpy(d_roles, roles, 16, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_iters, iters_host, N_ROLES * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<Stamp> h(nwg * 16);
    std::vector<double> spans, walls, uss; std::vector<double> rc[4];
    bool ok = true;
    for (int rep = 0; rep < repeats + 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kroles, dim3(nwg), dim3(1024), 96 * 1024, 0, d_out, d_sink, d_roles, d_iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        if (rep < 2) continue;  // warm-up launches
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
        std::vector<double> wg_span, wg_us; std::vector<double> per_role[4];
        for (int g = 0; g < nwg; ++g) {
            unsigned long long lo = ~0ull, hi = 0, rlo = ~0ull, rhi = 0;
            int per_simd[4] = {0, 0, 0, 0};
            for (int w = 0; w < 16; ++w) {
                const Stamp& s = h[g * 16 + w];
                per_simd[(s.hwid >> 4) & 3]++;
                if (s.role == IDLE) continue;
                lo = std::min(lo, s.t0); hi = std::max(hi, s.t1); rlo = std::min(rlo, s.r0); rhi = std::max(rhi, s.r1);
                for (int k = 0; k < 4; ++k) if ((int)s.role == roles[k]) { per_role[k].push_back((double)(s.t1 - s.t0)); break; }
            }
            for (int k = 0; k < 4; ++k) ok = ok && per_simd[k] == 4;
            wg_span.push_back((double)(hi - lo)); wg_us.push_back((double)(rhi - rlo) * 0.01);
        }
        auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        spans.push_back(med(wg_span)); uss.push_back(med(wg_us)); walls.push_back(ms);
        for (int k = 0; k < 4; ++k) rc[k].push_back(med(per_role[k]));
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    Result r;
    r.span_cyc = med(spans); r.span_us = med(uss); r.wall_ms = med(walls); r.ghz = r.span_cyc / (r.span_us * 1e3);
    for (int k = 0; k < 4; ++k) r.role_cyc[k] = med(rc[k]);
    r.placement_ok = ok;
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return r;
}

static void show(const char* tag, const int roles[4], const int* iters) {
    Result r = run(roles, iters);
    printf("%-34s [%-8s %-8s %-8s %-8s] span=%9.0f cyc  %8.1f us  (%.2f GHz)  wall=%.3f ms%s\n", tag, kRoleName[roles[0]],
           kRoleName[roles[1]], kRoleName[roles[2]], kRoleName[roles[3]], r.span_cyc, r.span_us, r.ghz, r.wall_ms,
           r.placement_ok ? "" : "  (!! a SIMD did not hold 4 waves)");
    fflush(stdout);
}

int main(int argc, char** argv) {
    int iters[N_ROLES];
    // ~2M cycles of single-role work per wave (MFMA: 32 cyc each; VALU: ~5 cyc each alone)
    const int mf = 4000;  // 16 MFMA x 32 cyc = 512 cyc / iter -> 2.05 M cycles
    for (int r = 0; r < N_ROLES; ++r) iters[r] = 4000;
    iters[R_MFMA] = mf; iters[R_MFMA4x4] = 8000; iters[R_SIGMOID] = 5000; iters[R_SIGMOID2] = 2500; iters[R_RCP] = 8000;
    iters[R_PERMSWAP] = 8000; iters[R_LDSREAD] = 4000;

    if (argc > 1 && !strcmp(argv[1], "chain")) {
        printf("# part 5: f32 MFMA 16x16x4 chains: 4 / 2 / 1 accumulators per wave, 1, 2, 4 waves per SIMD (cycles per MFMA per SIMD)\n");
        iters[R_MFMA2] = iters[R_MFMA1] = 4000;
        for (int role : {R_MFMA, R_MFMA2, R_MFMA1}) {
            for (int nw : {1, 2, 4}) {
                int roles[4] = {IDLE, IDLE, IDLE, IDLE};
                for (int k = 0; k < nw; ++k) roles[k] = role;
                Result r = run(roles, iters, 5);
                const double ops = (double)iters[role] * kOpsPerIter[role] * nw;
                printf("%-9s waves/SIMD=%d  span=%9.0f cyc  cyc/MFMA/SIMD=%6.2f  (%.2f GHz)%s\n", kRoleName[role], nw, r.span_cyc,
                       r.span_cyc / ops, r.ghz, r.placement_ok ? "" : " (!! placement)");
                fflush(stdout);
            }
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "va")) {
        printf("# part 4: instruction forms of the 256-state Viterbi kernel; Salu beside VALU roles on the same SIMD\n");
        for (int role : {R_SALU, R_MIN3, R_CNDMASK, R_VCMP, R_FMAC, R_PKMUL, R_DPP}) {  // R_VCMPS did not finish in 300 s on the box: not run
            for (int nw : {2, 4}) {
                int roles[4] = {IDLE, IDLE, IDLE, IDLE};
                for (int k = 0; k < nw; ++k) roles[k] = role;
                Result r = run(roles, iters, 5);
                const double ops = (double)iters[role] * kOpsPerIter[role] * nw;
                printf("%-9s waves/SIMD=%d  span=%9.0f cyc  cyc/instr/SIMD=%6.2f  (%.2f GHz)%s\n", kRoleName[role], nw, r.span_cyc,
                       r.span_cyc / ops, r.ghz, r.placement_ok ? "" : " (!! placement)");
                fflush(stdout);
            }
        }
        for (int v : {R_FMAC, R_DPP}) {
            int v2[4] = {IDLE, IDLE, v, v}, sv[4] = {R_SALU, R_SALU, v, v}, s2[4] = {R_SALU, R_SALU, IDLE, IDLE};
            char tag[64];
            show("Salu alone (2 waves)", s2, iters);
            snprintf(tag, sizeof tag, "%s alone (2 waves)", kRoleName[v]); show(tag, v2, iters);
            snprintf(tag, sizeof tag, "2 x Salu beside 2 x %s", kRoleName[v]); show(tag, sv, iters);
        }
        return 0;
    }
    if (argc <= 1 || strcmp(argv[1], "k"))
    printf("# part 1: issue cost per instruction form (one role on 1, 2, 4 waves of every SIMD; cycles per wave-instruction per SIMD)\n");
    for (int role = R_MFMA; role < R_K0 && (argc <= 1 || strcmp(argv[1], "k")); ++role) {
        for (int nw : {1, 2, 4}) {
            int roles[4] = {IDLE, IDLE, IDLE, IDLE};
            for (int k = 0; k < nw; ++k) roles[k] = role;
            Result r = run(roles, iters, 11);
            const double ops = (double)iters[role] * kOpsPerIter[role] * nw;
            printf("%-9s waves/SIMD=%d  span=%9.0f cyc  cyc/instr/SIMD=%6.2f  (%.2f GHz)%s\n", kRoleName[role], nw, r.span_cyc,
                   r.span_cyc / ops, r.ghz, r.placement_ok ? "" : " (!! placement)");
            fflush(stdout);
        }
    }
    printf("# part 3: the kernel's k-step pattern (4 x [20-instruction sigmoid chain + 3 MFMA]); nominal = 80 x VALU cost + 12 x 32\n");
    iters[R_K0] = iters[R_K1] = iters[R_K2] = 3000; iters[R_MFMA_DIFF] = 4000;
    iters[R_K3] = 3000; iters[R_K4] = 1500;
    for (int role : {R_K0, R_K2, R_K3, R_K4}) {
        for (int nw : {4}) {
            int roles[4] = {IDLE, IDLE, IDLE, IDLE};
            for (int k = 0; k < nw; ++k) roles[k] = role;
            Result r = run(roles, iters, 11);
            printf("%-6s waves/SIMD=%d  span=%9.0f cyc  per k-step per wave-slot: %7.1f cyc  (%.2f GHz)\n", kRoleName[role], nw, r.span_cyc,
                   r.span_cyc / iters[role] / nw / (role == R_K4 ? 2 : 1), r.ghz);
            fflush(stdout);
        }
    }
    if (argc > 1 && !strcmp(argv[1], "k")) return 0;
    printf("# part 2: two MFMA waves beside two VALU waves on every SIMD.  additive = sum of the two single-role spans, overlap = max\n");
    const int vroles[] = {R_SIGMOID, R_SIGMOID2, R_FMAC, R_FMA3, R_INT, R_RCP, R_MOV, R_DPP, R_PERMSWAP, R_LDSREAD, R_MFMA4x4};
    {
        int m2[4] = {R_MFMA, R_MFMA, IDLE, IDLE};
        show("MFMA alone (2 waves)", m2, iters);
    }
    for (int v : vroles) {
        int v2[4] = {IDLE, IDLE, v, v}, mv[4] = {R_MFMA, R_MFMA, v, v}, m1v1[4] = {R_MFMA, v, IDLE, IDLE};
        char tag[64];
        snprintf(tag, sizeof tag, "%s alone (2 waves)", kRoleName[v]); show(tag, v2, iters);
        snprintf(tag, sizeof tag, "2 x M beside 2 x %s", kRoleName[v]); show(tag, mv, iters);
        snprintf(tag, sizeof tag, "1 x M beside 1 x %s", kRoleName[v]); show(tag, m1v1, iters);
    }
    return 0;
}
