"""Experiment: where conv_wgrad_x3_kernel's time goes.  Scratch variants of conv.hip (text edits of its chunk loop):
  nostage = split + LDS stores for the first chunk only      noload = global loads for the first chunk only
  nomfma  = MFMA phase for the first chunk only               noboth = nostage + noload
build (here):  python tools/exp_wgx3_split.py build      run (GPU box):  python tools/exp_wgx3_split.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "textboxgan_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "scratch")
STORE = ("    wgrad_x3_store<VEC>(p, rg, Ss, Ls, tid);\n    __syncthreads();\n    if (chunk + p.ksplit < p.nchunks) {",
         "    if (chunk == (int)blockIdx.z) wgrad_x3_store<VEC>(p, rg, Ss, Ls, tid);\n    __syncthreads();\n    if (chunk + p.ksplit < p.nchunks) {")
LOAD = ("      chunk_pos(chunk + p.ksplit, bg, u0, v0);\n      wgrad_x3_load<VEC>(p, rg, bg, u0, v0, cs0, cl0, tid);",
        "      chunk_pos(chunk + p.ksplit, bg, u0, v0);\n      if (chunk < 0) wgrad_x3_load<VEC>(p, rg, bg, u0, v0, cs0, cl0, tid);")
MFMA = ("#pragma unroll 1\n    for (int gp = 0; gp < PIX / 16; ++gp) {\n      const int pp = 16 * gp + 8 * half;  // first of this half-wave's 8 pixels (one 32-pixel tile row)\n      const __bf16 *Lg = Lp + (VEC == 1",
        "#pragma unroll 1\n    for (int gp = 0; gp < (chunk == (int)blockIdx.z ? PIX / 16 : 0); ++gp) {\n      const int pp = 16 * gp + 8 * half;  // first of this half-wave's 8 pixels (one 32-pixel tile row)\n      const __bf16 *Lg = Lp + (VEC == 1")
NOLDS = [("      for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const bf16x8 *>(Sp + pl * SPL + 16 * gp);",
          "      for (int pl = 0; pl < 3; ++pl) { i32x4 t = {gp + pl, lane, chunk, pl}; a[pl] = __builtin_bit_cast(bf16x8, t); }"),
         ("          const i32x4 e = *reinterpret_cast<const i32x4 *>(row);\n          const int e4 = *reinterpret_cast<const int *>(row + 8);",
          "          const i32x4 e = {gp, kh + pl, lane, (int)(size_t)row};\n          const int e4 = chunk;")]
VARIANTS = {"nostage": [STORE], "noload": [LOAD], "nomfma": [MFMA], "noboth": [STORE, LOAD]}
if os.environ.get("WG_ABLATE") == "2":
    VARIANTS = {"noboth": [STORE, LOAD], "noboth_nolds": [STORE, LOAD] + NOLDS}


def build():
    os.makedirs(OUT, exist_ok=True)
    base = open(os.path.join(SRC, "conv.hip")).read()
    for name, edits in VARIANTS.items():
        s = base
        for a, b in edits:
            assert s.count(a) == 1, (name, a[:70], s.count(a))
            s = s.replace(a, b)
        s = s.replace('#include "common.h"', f'#include "{SRC}/common.h"')
        path = os.path.join(OUT, f"convwg_{name}.hip")
        open(path, "w").write(s)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
               "-Wno-unused-value", "-shared", "-o", os.path.join(OUT, f"libwg_{name}.so"), path]
        print(" ".join(cmd)); subprocess.check_call(cmd)


def timeit(call, n=20):
    import torch
    for _ in range(8): call()
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
        l = C.CDLL(os.path.join(OUT, f"libwg_{name}.so"))
        l.tbg_conv2d_wgrad_x3.argtypes = P.tbg_conv2d_wgrad_x3.argtypes
        return l
    libs = {"product": P, **{k: load(k) for k in VARIANTS}, "product2": P}
    B = 16
    for CS, CL, H, W, st in ((128, 128, 64, 256, 1), (128, 128, 32, 128, 1), (256, 256, 16, 64, 1), (128, 64, 32, 128, 2)):
        Hl, Wl = (H, W) if st == 1 else (2 * H + 1, 2 * W + 1)
        S = torch.randn(B, CS, H, W, device=dev); L = torch.randn(B, CL, Hl, Wl, device=dev)
        dw = torch.empty(3, 3, CL, CS, device=dev)
        pad = 1 if st == 1 else 0
        d = N.WgradDesc(B, CS, CL, H, W, Hl, Wl, 3, 3, st, st, pad, pad, CL * CS, CS, 1, 1.0)
        nb = P.tbg_conv2d_wgrad_workspace_bytes(C.byref(d))
        ws = torch.empty(nb // 4 + 4, device=dev)
        flops = 2.0 * B * CS * CL * 9 * H * W
        line = f"wgrad x3 {CS}x{CL} S={H}x{W} s{st} [{N.wgrad_kernel_name(d, 2)}]:"
        for name, l in libs.items():
            t = timeit(lambda: l.tbg_conv2d_wgrad_x3(C.byref(d), N.ptr(S), N.ptr(L), N.ptr(dw), None, None, None, None, 0.0,
                                                     N.ptr(ws), nb, N.stream()))
            line += f"  {name} {t:7.1f} us ({flops / t / 1e6:5.1f} TF)"
        print(line, flush=True)


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else run()
