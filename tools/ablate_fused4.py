#!/usr/bin/env python3
"""Diagnostic builds of vnet16_fused4_kernel with one phase removed each (results are WRONG by construction; only the
run time is read).  f32 MFMA and every other instruction of a SIMD are mutually exclusive on gfx950
(profiles/r02_ubench3_mfma_valu_roles.txt), so the time a phase costs is the time the kernel loses when the phase is
taken out.

    python tools/ablate_fused4.py build        # here (CPU): patched copies of csrc/ -> .scratch/abl/libmvn_<name>.so
    python tools/ablate_fused4.py run          # on the GPU box: time every variant (own process each, MVN_LIB_PATH)
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "meta-viterbinet_amd", "csrc")
OUT = os.path.join(ROOT, ".scratch", "abl")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]

SWEEP_CALL = """                if (nsteps == 16) sweep_tile(std::true_type{});
                else sweep_tile(std::false_type{});"""
DECIDE = """                const int d0 = decide_lsb<0>(mrec[0], ulog[0]), d1 = decide_lsb<1>(mrec[1], ulog[1]);
                const int d2 = decide_lsb<2>(mrec[2], ulog[2]), d3 = decide_lsb<3>(mrec[3], ulog[3]);"""
L3 = "                    acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(bop[i3], ldsB3w[i3 * 64 + lane], acc3, 0, 0, 0);"
U4849 = """                transpose_rows4(hu);
                accL = __builtin_amdgcn_mfma_f32_4x4x1f32(wl.x, __uint_as_float(hu[0]), accL, 0, 0, 0);
                accL = __builtin_amdgcn_mfma_f32_4x4x1f32(wl.y, __uint_as_float(hu[1]), accL, 0, 0, 0);
                accL = __builtin_amdgcn_mfma_f32_4x4x1f32(wl.z, __uint_as_float(hu[2]), accL, 0, 0, 0);
                accL = __builtin_amdgcn_mfma_f32_4x4x1f32(wl.w, __uint_as_float(hu[3]), accL, 0, 0, 0);"""
SIG = "                    const float h = sigmoid(__builtin_fmaf(yv[u], wb.x, wb.y));"
MF2 = """                    acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, h, acc[u][0], 0, 0, 0);
                    acc[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, h, acc[u][1], 0, 0, 0);
                    acc[u][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, h, acc[u][2], 0, 0, 0);"""
BIASRELU = """                    transpose_rows4(v);
#pragma unroll
                    for (int r = 0; r < 4; ++r) bop[4 * tau + r] = __uint_as_float(v[r]);"""
BPERM = "                    cost[r] = -__int_as_float(__builtin_amdgcn_ds_bpermute(row_addr + 4 * ulog[r], __float_as_int(logit)));"

LDSK = """                const float4 av = ldsA2[i * 64 + lane];
                const float4 wl = ldsWL[i * 4 + (lane & 3)];
                const float2 wb = ldsWB[i * 4 + q];"""
TILE = "            if (tu < T) {  // wave-uniform"

WG4 = "constexpr int kFused4Waves = 4;"
LB = "__launch_bounds__(64 * kFused4Waves, 4)"
KTOP = """#else
#pragma unroll MVN_F4_UNROLL
            for (int i = 0; i < kK2Steps; ++i) {"""

P2A = """#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, hh[u], acc[u][0], 0, 0, 0);"""
P2B = """                accL = __builtin_amdgcn_mfma_f32_4x4x1f32(wl.w, __uint_as_float(hu[3]), accL, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);"""

KENTRY = "    constexpr int S = 16;\n    __shared__ float ldsB3w[kK3Steps * 64];   // W3 as the layer-3 B operand"
KSTART = "    if (b >= B) return;  // whole wave; no barriers below"
KEND = "    if (final_metric && q == 0) final_metric[b * S + logical_state(j, T & 3)] = m;"

VARIANTS = {
    "base": [],
    "nosweep": [(SWEEP_CALL, "                m += cost[0] + cost[1] + cost[2] + cost[3]; mrec[0] = m; mrec[1] = cost[1]; mrec[2] = cost[2]; mrec[3] = cost[3];")],
    "nodecide": [(DECIDE, "                const int d0 = __float_as_int(mrec[0]) & 1, d1 = __float_as_int(mrec[1]) & 1, d2 = __float_as_int(mrec[2]) & 1, d3 = __float_as_int(mrec[3]) & 1;")],
    "nol3": [(L3, "                    acc3[i3 & 3] += bop[i3] * ldsB3w[i3 * 64 + lane];")],
    "no4849": [(U4849, "                accL[0] += __uint_as_float(hu[0]) * wl.x + __uint_as_float(hu[1]) + __uint_as_float(hu[2]) + __uint_as_float(hu[3]);")],
    "no4849mfma": [(U4849, "                transpose_rows4(hu);\n                accL[0] += __uint_as_float(hu[0]) * wl.x + __uint_as_float(hu[1]) + __uint_as_float(hu[2]) + __uint_as_float(hu[3]);")],
    "nosig": [(SIG, "                    const float h = __builtin_fmaf(yv[u], wb.x, wb.y);")],
    "nomfma2": [(MF2, "                    acc[u][0][0] += av.x * h; acc[u][1][1] += av.y * h; acc[u][2][2] += av.z * h;")],
    "notranspose": [(BIASRELU, "#pragma unroll\n                    for (int r = 0; r < 4; ++r) bop[4 * tau + r] = __uint_as_float(v[r]);")],
    "nobperm": [(BPERM, "                    cost[r] = -logit;")],
    "nolds": [(LDSK, "                const float4 av = make_float4(1.0f + i, 2.0f, 3.0f, 0.f);\n                const float4 wl = make_float4(0.5f, 0.25f, i, 0.f);\n                const float2 wb = make_float2(0.01f * (i + 1), 0.02f);")],
    "emptyk": [(SIG, "                    const float h = __builtin_fmaf(yv[u], wb.x, wb.y);"),
               (MF2, "                    acc[u][0][0] += av.x * h; acc[u][1][1] += av.y * h; acc[u][2][2] += av.z * h;"),
               (U4849, "                accL[0] += __uint_as_float(hu[0]) * wl.x + __uint_as_float(hu[1]) + __uint_as_float(hu[2]) + __uint_as_float(hu[3]);")],
    # hypothesis test: 16-wave workgroups (4 waves per SIMD in ONE workgroup) kept in lockstep by a raw s_barrier per k-step
    "wg16": [(WG4, "constexpr int kFused4Waves = 16;"), (LB, "__launch_bounds__(64 * kFused4Waves, 1)")],
    "wg16_bar": [(WG4, "constexpr int kFused4Waves = 16;"), (LB, "__launch_bounds__(64 * kFused4Waves, 1)"),
                 (KTOP, KTOP + "\n                __builtin_amdgcn_s_barrier();")],
    "wg16_bar_tile": [(WG4, "constexpr int kFused4Waves = 16;"), (LB, "__launch_bounds__(64 * kFused4Waves, 1)"),
                      (KTOP, KTOP + "\n                __builtin_amdgcn_s_barrier();"),
                      (TILE, "            __builtin_amdgcn_s_barrier();\n" + TILE)],
    # phase separation: 16-wave workgroups, grouped k-step (MVN_F4_PIPE=2), a raw barrier between the sigmoid phase and the
    # MFMA phase and another after the MFMA phase, so a SIMD runs VALU-only and MFMA-only stretches
    "wg16_p2": [(WG4, "constexpr int kFused4Waves = 16;"), (LB, "__launch_bounds__(64 * kFused4Waves, 1)")],
    "wg16_p2bar2": [(WG4, "constexpr int kFused4Waves = 16;"), (LB, "__launch_bounds__(64 * kFused4Waves, 1)"),
                    (P2A, "__builtin_amdgcn_s_barrier();\n" + P2A), (P2B, P2B + "\n                __builtin_amdgcn_s_barrier();")],
    "wg16_p2bar1": [(WG4, "constexpr int kFused4Waves = 16;"), (LB, "__launch_bounds__(64 * kFused4Waves, 1)"),
                    (P2A, "__builtin_amdgcn_s_barrier();\n" + P2A)],
    # in-kernel clock: every wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its whole life and
    # writes both differences over its final-metric row (a buffer no output of this diagnostic build is read from)
    "clock": [(KENTRY, KENTRY + "\n    const unsigned long long st_re = __builtin_amdgcn_s_memrealtime();"),
              (KSTART, KSTART + "\n    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();"),
              (KEND, "    if (final_metric && lane == 0) { unsigned long long *o = reinterpret_cast<unsigned long long *>(final_metric + b * S); o[0] = __builtin_amdgcn_s_memtime() - st_t0; o[1] = __builtin_amdgcn_s_memrealtime() - st_r0; o[2] = st_re; o[3] = st_r0; unsigned hw; asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\" : \"=s\"(hw)); unsigned xcc; asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\" : \"=s\"(xcc)); o[4] = ((unsigned long long)xcc << 32) | hw; }")],
    "notile": [(TILE, "            if (tu < T && yv[0] == 12345.0f) {  // wave-uniform")],
}


# variants that only differ by -D switches of the product source
DEFS = {
    "pipe0": ["-DMVN_F4_PIPE=0"],
    "pipe1": ["-DMVN_F4_PIPE=1"],
    "pipe1_u1": ["-DMVN_F4_PIPE=1", "-DMVN_F4_UNROLL=1"],
    "pipe1_u25": ["-DMVN_F4_PIPE=1", "-DMVN_F4_UNROLL=25"],
    "pipe2": ["-DMVN_F4_PIPE=2"],
    "pipe3": ["-DMVN_F4_PIPE=3"],
    "clock": ["-DMVN_F4_PIPE=3"],
    "wg16_p2": ["-DMVN_F4_PIPE=2"], "wg16_p2bar2": ["-DMVN_F4_PIPE=2"], "wg16_p2bar1": ["-DMVN_F4_PIPE=2"],
}
for _k in DEFS:
    VARIANTS.setdefault(_k, [])


def build():
    os.makedirs(OUT, exist_ok=True)
    for name, patches in VARIANTS.items():
        if sys.argv[2:] and name not in sys.argv[2:]:
            continue
        d = os.path.join(OUT, "src_" + name)
        shutil.rmtree(d, ignore_errors=True)
        shutil.copytree(CSRC, os.path.join(d, "meta-viterbinet_amd", "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
        p = os.path.join(d, "meta-viterbinet_amd", "csrc", "vnet16_fused4.inc")
        s = open(p).read()
        for old, new in patches:
            assert s.count(old) == 1, (name, old[:60], s.count(old))
            s = s.replace(old, new)
        open(p, "w").write(s)
        so = os.path.join(OUT, f"libmvn_{name}.so")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + DEFS.get(name, []) + [os.path.join(d, "meta-viterbinet_amd", "csrc", "mvn_hip.hip"), "-o", so],
                       check=True)
        shutil.rmtree(d)
        print("built", so, flush=True)


TIMER = r"""
import os, sys, torch
sys.path.insert(0, %r)
import meta_viterbinet_amd as mvn
import numpy as np
dev = torch.device("cuda:0")
B, T, S, L = int(os.environ.get("MVN_ABL_B", "10000")), 1000, 16, 4
g = np.load(os.path.join(%r, "tests", "golden", "g7_by_word.npz"))
w = [torch.tensor(g[f"w{i}"], device=dev) for i in range(6)]
tx, y = mvn.synthetic_words(B, T, L, 10.0, 0.2, dev, seed=3450002)
lib = mvn._lib.load(); st = mvn._lib.current_stream(dev)
dec = torch.zeros(B, T, device=dev)
fm = torch.zeros(B, S, device=dev) if sys.argv[1] == "clock" else None
wp = [mvn._lib.ptr(t) for t in w]
def run():
    rc = lib.mvn_vnet_decode_f32(mvn._lib.ptr(y), T, *wp, mvn._lib.ptr(dec), T, None, mvn._lib.ptr(fm), None, 0, B, T, S, st)
    assert rc == 0
for _ in range(5): run()
torch.cuda.synchronize()
ts = []
for rep in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
ts.sort()
if fm is not None:
    raw = fm.cpu().numpy().view(np.uint64).reshape(B, -1)
    v = raw[:, :2].astype(np.float64)
    ghz = v[:, 0] / (v[:, 1] * 10.0)
    t_entry = raw[:, 2].astype(np.float64); t_start = raw[:, 3].astype(np.float64); t_end = t_start + v[:, 1]
    k0 = t_entry.min()
    print(f"last launch: kernel span {(t_end.max()-k0)/100:.1f} us; prologue (entry -> first symbol) median {np.median(t_start-t_entry)/100:.2f} us, "
          f"p95 {np.percentile(t_start-t_entry,95)/100:.2f} us")
    order = np.argsort(t_start)
    for lo, hi in ((0, 4096), (4096, 8192), (8192, B)):
        idx = order[lo:hi]
        if len(idx) == 0:
            continue
        print(f"  waves #{lo}..{hi} by start time: start {np.min(t_start[idx]-k0)/100:7.1f} .. {np.max(t_start[idx]-k0)/100:7.1f} us, "
              f"end {np.min(t_end[idx]-k0)/100:7.1f} .. {np.max(t_end[idx]-k0)/100:7.1f} us, life median {np.median(v[idx,1])/100:6.1f} us")
    hw = raw[:, 4]; cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xf)
    tail = order[8192:] if B > 8192 else order
    cnt = np.bincount(np.unique(cu[tail], return_inverse=True)[1])
    print(f"  tail waves per CU: CUs used {len(cnt)}, histogram of waves/CU {np.bincount(cnt).tolist()}")
    print(f"in-kernel clock over a wave's life: median {np.median(ghz):.3f} GHz (p5 {np.percentile(ghz,5):.3f}, p95 {np.percentile(ghz,95):.3f}); "
          f"wave life median {np.median(v[:,0]):.0f} cycles = {np.median(v[:,1])/100:.1f} us", flush=True)
print(f"{sys.argv[1]:14s} B={B} median {ts[3]:.4f} ms  min {ts[0]:.4f}  max {ts[-1]:.4f}   cycles/symbol/SIMD @2.4GHz: {ts[3]*1e-3*2.4e9*1024/(B*T):.1f}", flush=True)
"""


def run():
    names = sys.argv[2:] or list(VARIANTS)
    for name in names:
        env = dict(os.environ)
        if name != "product":  # "product" = the in-tree libmvn_hip.so
            env["MVN_LIB_PATH"] = os.path.join(OUT, f"libmvn_{name}.so")
        subprocess.run([sys.executable, "-c", TIMER % (ROOT, ROOT), name], env=env, check=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
