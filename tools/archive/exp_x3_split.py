"""Experiment: where the f32x3 forward kernel's time goes.  Scratch variants of conv.hip (text edits of the X3 K loop):
  nodma   = filter DMA for the first chunk only          nostage = halo loads + split/stores for the first chunk only
  nomfma  = MFMA phase for the last chunk only           noboth  = nodma + nostage (barriers + MFMAs remain)
build (here):  python tools/exp_x3_split.py build      run (GPU box):  python tools/exp_x3_split.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "textboxgan_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "scratch")
DMA = ("        issue_filter_dma(kc, As);\n        if (kc + 1 < kend) {\n          loadx((kc + 1) * CK);",
       "        if (kc == kbeg) issue_filter_dma(kc, As);\n        if (kc + 1 < kend) {\n          loadx((kc + 1) * CK);")
STG = [("        if (kc + 1 < kend) {\n          loadx((kc + 1) * CK);", "        if (kc + 1 < kend) {\n          if (kc < kbeg) loadx((kc + 1) * CK);"),
       ("          if (j < p.NJ) {\n#pragma unroll\n            for (int bt = 0; bt < NBT; ++bt) store_halo(kc * CK + bt * CB, bt, j, xr[j][bt], Xs);",
        "          if (j < p.NJ && kc == kbeg) {\n#pragma unroll\n            for (int bt = 0; bt < NBT; ++bt) store_halo(kc * CK + bt * CB, bt, j, xr[j][bt], Xs);"),
       ("          if (p.NJ == 1) asm volatile(\"s_waitcnt vmcnt(%0) lgkmcnt(0)\" ::\"n\"(V1) : \"memory\");\n"
        "          else if (p.NJ == 2) asm volatile(\"s_waitcnt vmcnt(%0) lgkmcnt(0)\" ::\"n\"(V2) : \"memory\");\n"
        "          else asm volatile(\"s_waitcnt vmcnt(%0) lgkmcnt(0)\" ::\"n\"(V3) : \"memory\");",
        "          asm volatile(\"s_waitcnt vmcnt(0) lgkmcnt(0)\" ::: \"memory\");")]
MFMA = ("        __builtin_amdgcn_s_barrier();\n        mfma_taps(As, Xs);\n      }\n    }\n  }\n  if (done) {",
        "        __builtin_amdgcn_s_barrier();\n        if (kc == kend - 1) mfma_taps(As, Xs);\n      }\n    }\n  }\n  if (done) {")
SKEW = lambda n: [("      if (kbeg < kend) loadx(kbeg * CK);\n", "      if (kbeg < kend) loadx(kbeg * CK);\n      if (__builtin_amdgcn_s_getreg(6148) & 1) __builtin_amdgcn_s_sleep(%d);  // HW_ID.wave_id parity\n" % n)]
PRIO = [("        __builtin_amdgcn_s_barrier();\n        mfma_taps(As, Xs);\n      }\n    }\n  }\n  if (done) {",
         "        __builtin_amdgcn_s_barrier();\n        __builtin_amdgcn_s_setprio(2);\n        mfma_taps(As, Xs);\n        __builtin_amdgcn_s_setprio(0);\n      }\n    }\n  }\n  if (done) {")]
VARIANTS = {"skew32": SKEW(32), "skew64": SKEW(64), "skew100": SKEW(100), "prio": PRIO, "skew64prio": SKEW(64) + PRIO}
NOLDS = [("          const int cb = st & 1;\n          if (st + 1 < NS) ld(st + 1, cb ^ 1);\n", "          const int cb = 0;\n")]
DMAFIRST = [("""#pragma unroll
        for (int j = 0; j < NJX; ++j)
          if (j < p.NJ) {
#pragma unroll
            for (int bt = 0; bt < NBT; ++bt) store_halo(kc * CK + bt * CB, bt * (CB / KP), j, xr[j][bt], Xs);
          }
        issue_filter_dma(kc, As);
""", """        issue_filter_dma(kc, As);
#pragma unroll
        for (int j = 0; j < NJX; ++j)
          if (j < p.NJ) {
#pragma unroll
            for (int bt = 0; bt < NBT; ++bt) store_halo(kc * CK + bt * CB, bt * (CB / KP), j, xr[j][bt], Xs);
          }
""")]
if os.environ.get("X3_ABLATE") == "3":
    VARIANTS = {"dmafirst": DMAFIRST}
elif os.environ.get("X3_ABLATE") == "2":
    VARIANTS = {"noboth": [DMA] + STG, "noboth_nolds": [DMA] + STG + NOLDS, "nolds": NOLDS}
elif os.environ.get("X3_ABLATE"):
    VARIANTS = {"nodma": [DMA], "nostage": STG, "nomfma": [MFMA], "noboth": [DMA] + STG}


def build():
    os.makedirs(OUT, exist_ok=True)
    base = open(os.path.join(SRC, "conv.hip")).read()
    for name, edits in VARIANTS.items():
        s = base
        for a, b in edits:
            assert s.count(a) == 1, (name, a[:70], s.count(a))
            s = s.replace(a, b)
        s = s.replace('#include "common.h"', f'#include "{SRC}/common.h"')
        path = os.path.join(OUT, f"convx3_{name}.hip")
        open(path, "w").write(s)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
               "-Wno-unused-value", "-shared", "-o", os.path.join(OUT, f"libx3_{name}.so"), path]
        print(" ".join(cmd)); subprocess.check_call(cmd)


def timeit(call, n=20):
    import torch
    for _ in range(5): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run():
    sys.path.insert(0, ROOT)
    import torch
    from textboxgan_amd import native as N, ops
    dev = torch.device("cuda:0")
    P = N.lib()
    def load(name):
        l = C.CDLL(os.path.join(OUT, f"libx3_{name}.so"))
        l.tbg_conv2d_x3.argtypes = P.tbg_conv2d_x3.argtypes
        return l
    libs = {"product": P, **{k: load(k) for k in VARIANTS}}
    B = 16
    S2 = ((128, 128, 65, 257, 32, 128, 2, 0), (128, 256, 33, 129, 16, 64, 2, 0), (128, 128, 32, 128, 65, 257, 2, 1),
          (256, 256, 16, 64, 33, 129, 2, 1))
    S1 = ((128, 128, 64, 256, 64, 256, 1, 0), (128, 128, 32, 128, 32, 128, 1, 0), (256, 256, 16, 64, 16, 64, 1, 0),
          (64, 64, 64, 256, 64, 256, 1, 0))
    shapes = {"strided": S2, "all": S1 + S2}.get(os.environ.get("X3_SHAPES"), S1)
    for Cc, M, H, W, OH, OW, st, T in shapes:
        x = torch.randn(B, Cc, H, W, device=dev)
        w = torch.randn(3, 3, Cc, M, device=dev)
        pf = ops.pack_filter(w, False, False, bf16="f32x3")  # timing only: the filter's orientation does not matter
        y = torch.empty(B, M, OH, OW, device=dev)
        pad = 1 if st == 1 else 0
        d = N.ConvDesc(B, Cc, M, H, W, OH, OW, 3, 3, st, st, pad, pad, T, 0, M, 1)
        e = N.epilogue()
        flops = 2.0 * B * Cc * M * 9 * (H * W if T else OH * OW)
        line = f"x3 fprop {Cc}->{M} {H}x{W}->{OH}x{OW} s{st} T{T} [{N.conv_kernel_name(d, False, 2)}]:"
        for name, l in libs.items():
            t = timeit(lambda: l.tbg_conv2d_x3(C.byref(d), N.ptr(x), N.ptr(pf.data), N.ptr(y), None, C.byref(e), N.stream()))
            line += f"  {name} {t:7.1f} us ({flops / t / 1e6:5.1f} TF)"
        print(line, flush=True)


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else run()
