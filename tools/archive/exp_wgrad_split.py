"""Experiment: where does conv_wgrad_kernel's time go?  Builds two scratch variants of the library from conv.hip --
(a) tiles staged for the first pixel chunk only (MFMA + LDS-read bound), (b) staging only (no MFMA loop) -- and times them
against the product kernel on the step's big filter-gradient shapes.
build (here):  python tools/exp_wgrad_split.py build      run (GPU box):  python tools/exp_wgrad_split.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "textboxgan_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "scratch")
VARIANTS = {  # (the float4-staged instance: the shapes below all take it)
    "nostage": [("      wgrad_stage_vec<false>(p, Ss, Ls, bg, u0, v0, cs0, cl0, tid);",
                 "      if (chunk == (int)blockIdx.z) wgrad_stage_vec<false>(p, Ss, Ls, bg, u0, v0, cs0, cl0, tid);")],
    "nomfma": [("      for (int gp = 0; gp < PIX / 8; ++gp) {\n        const int pp = 8 * gp + 4 * half;",
                "      for (int gp = 0; gp < 1; ++gp) {\n        const int pp = 8 * gp + 4 * half;")],
}


def build():
    os.makedirs(OUT, exist_ok=True)
    base = open(os.path.join(SRC, "conv.hip")).read()
    for name, edits in VARIANTS.items():
        s = base
        for a, b in edits:
            assert s.count(a) == 1, (name, a[:50], s.count(a))
            s = s.replace(a, b)
        s = s.replace('#include "common.h"', f'#include "{SRC}/common.h"')
        path = os.path.join(OUT, f"conv_{name}.hip")
        open(path, "w").write(s)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
               "-Wno-unused-value", "-shared", "-o", os.path.join(OUT, f"libexp_{name}.so"), path]
        print(" ".join(cmd)); subprocess.check_call(cmd)


def run():
    sys.path.insert(0, ROOT)
    import torch
    from textboxgan_amd import native as N
    dev = torch.device("cuda:0")
    libs = {"product": N.lib()}
    for name in VARIANTS:
        l = C.CDLL(os.path.join(OUT, f"libexp_{name}.so"))
        l.tbg_conv2d_wgrad_ex_f32.argtypes = N.lib().tbg_conv2d_wgrad_ex_f32.argtypes
        l.tbg_conv2d_wgrad_workspace_bytes.argtypes = N.lib().tbg_conv2d_wgrad_workspace_bytes.argtypes
        l.tbg_conv2d_wgrad_workspace_bytes.restype = C.c_longlong
        libs[name] = l
    shapes = [(16, 128, 128, 64, 256), (16, 128, 128, 32, 128), (16, 256, 256, 16, 64), (16, 64, 64, 64, 256)]
    for B, CS, CL, H, W in shapes:
        S = torch.randn(B, CS, H, W, device=dev); L = torch.randn(B, CL, H, W, device=dev)
        dW = torch.empty(9, CL, CS, device=dev)
        d = N.WgradDesc(B=B, CS=CS, CL=CL, Hs=H, Ws=W, Hl=H, Wl=W, KH=3, KW=3, sy=1, sx=1, py=1, px=1, st_t=CL * CS, st_l=CS, st_s=1, alpha=1.0)
        flops = 2.0 * B * H * W * CS * CL * 9
        line = f"wgrad B={B} {CS}x{CL} {H}x{W}: "
        for name, l in libs.items():
            wsb = l.tbg_conv2d_wgrad_workspace_bytes(C.byref(d))
            ws = torch.empty(max(wsb, 4) // 4, device=dev)
            def call():
                rc = l.tbg_conv2d_wgrad_ex_f32(C.byref(d), N.ptr(S), N.ptr(L), N.ptr(dW), None, None, None, None, 0.0, N.ptr(ws), wsb, N.stream())
                assert rc == 0, rc
            for _ in range(3): call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): call()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            line += f"{name} {us:7.1f} us ({flops / us / 1e6:6.1f} TF)  "
        print(line)


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else run()
