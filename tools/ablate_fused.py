#!/usr/bin/env python3
"""Diagnostic builds of vnet16_fusedn_kernel with one phase removed each (results are WRONG by construction; only the
run time is read).  f32 MFMA and every other instruction of a SIMD are mutually exclusive on gfx950
(profiles/r02_ubench3_mfma_valu_roles.txt), so the time a phase costs is the time the kernel loses when the phase is
taken out.

    python tools/ablate_fused.py build        # here (CPU): patched copies of csrc/ -> .scratch/abl/libmvn_<name>.so
    python tools/ablate_fused.py run          # on the GPU box: time every variant (own process each, MVN_LIB_PATH)
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "meta-viterbinet_amd", "csrc")
OUT = os.path.join(ROOT, ".scratch", "abl")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]

SRC = "vnet16_fusedn.inc"
SWEEP_CALL = """                if (nsteps == 16) sweep_tile(std::true_type{});
                else sweep_tile(std::false_type{});"""
DECIDE = """                const int d0 = decide_lsb<0>(mrec[0], ulog[0]), d1 = decide_lsb<1>(mrec[1], ulog[1]);
                const int d2 = decide_lsb<2>(mrec[2], ulog[2]), d3 = decide_lsb<3>(mrec[3], ulog[3]);"""
L3 = "                    acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(bop[i3], ldsB3w[i3 * 64 + lane], acc3, 0, 0, 0);"
SIG = "                        const float h = sigmoid(__builtin_fmaf(yv[u], wb.x, wb.y));"
MF2 = """                        acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(axy.x, h, acc[u][0], 0, 0, 0);
                        acc[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(axy.y, h, acc[u][1], 0, 0, 0);
                        acc[u][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(az, h, acc[u][2], 0, 0, 0);"""
CHAIN = "                    if (i > 0) chain_step(tr, wa, wb4);"
TILE = "            if (tu < T) {  // wave-uniform"

# phase knock-outs: patched copies of the kernel source (results wrong by construction, only the time is read)
VARIANTS = {
    "base": [],
    "nosweep": [(SWEEP_CALL, "                m += cost[0] + cost[1] + cost[2] + cost[3]; mrec[0] = m; mrec[1] = cost[1]; mrec[2] = cost[2]; mrec[3] = cost[3];")],
    "nodecide": [(DECIDE, "                const int d0 = __float_as_int(mrec[0]) & 1, d1 = __float_as_int(mrec[1]) & 1, d2 = __float_as_int(mrec[2]) & 1, d3 = __float_as_int(mrec[3]) & 1;")],
    "nol3": [(L3, "                    acc3[i3 & 3] += bop[i3] * ldsB3w[i3 * 64 + lane];")],
    "nosig": [(SIG, "                        const float h = __builtin_fmaf(yv[u], wb.x, wb.y);")],
    "nomfma2": [(MF2, "                        acc[u][0][0] += axy.x * h; acc[u][1][1] += axy.y * h; acc[u][2][2] += az * h;")],
    "nochain": [(CHAIN, "                    if (i > 0) ca += tr.x * wa.x;")],
    "notile": [(TILE, "            if (tu < T && yv[0] == 12345.0f) {  // wave-uniform")],
}

# variants that only differ by -D switches of the product source
DEFS = {
    "fn_clock": ["-DMVN_DIAG_STAMPS"],  # wave timeline stamps (see vnet16_fusedn.inc)
    "fn_w4g5": ["-DMVN_FN_WAVES=4", "-DMVN_FN_WGS=5"],
    "fn_nt4": ["-DMVN_FUSEDN_DEFAULT=4"],
    "fn_noprio": ["-DMVN_FN_PRIO=0"],
    "fn_w8g3": ["-DMVN_FN_WAVES=8", "-DMVN_FN_WGS=3"], "fn_w8g2": ["-DMVN_FN_WAVES=8", "-DMVN_FN_WGS=2"],
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
        p = os.path.join(d, "meta-viterbinet_amd", "csrc", SRC)
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
fm = torch.zeros(B, S, device=dev) if "clock" in sys.argv[1] else None
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
    SL = int(os.environ.get('MVN_ABL_SLOTS', '4096'))
    for lo, hi in ((0, SL), (SL, 2 * SL), (2 * SL, B)):
        idx = order[lo:hi]
        if len(idx) == 0:
            continue
        print(f"  waves #{lo}..{hi} by start time: start {np.min(t_start[idx]-k0)/100:7.1f} .. {np.max(t_start[idx]-k0)/100:7.1f} us, "
              f"end {np.min(t_end[idx]-k0)/100:7.1f} .. {np.max(t_end[idx]-k0)/100:7.1f} us, life median {np.median(v[idx,1])/100:6.1f} us")
    hw = raw[:, 4]; cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xf)
    tail = order[2 * SL:] if B > 2 * SL else order[SL:] if B > SL else order
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
