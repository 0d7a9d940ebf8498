"""Experiment: conv_fprop_kernel time split (plain K loop instances) and the wave-priority idea for conv_wgrad_kernel.
Scratch variants of conv.hip:  fnostage = halo + filter staged for the first chunk only;  prio = co-resident waves of a SIMD
get different issue priorities (so that one block's MFMA phase runs while the other one stages).
build (here):  python tools/exp_fprop_split.py build      run (GPU box):  python tools/exp_fprop_split.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "textboxgan_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "scratch")
VARIANTS = {
    "fnostage": [("      // halo batch 0 first: its round trip hides under the issue phase of the filter DMA below\n",
                  "      if (kc == kbeg) {\n"),
                 ("      asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");  // the filter DMA has landed (explicit, not left to the fence)\n      __syncthreads();\n      mfma_taps(As, Xs);",
                  "      }\n      asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n      __syncthreads();\n      mfma_taps(As, Xs);")],
    "prio": [("  const int g_HALFW = VEC == 2 ? WgVec2<false>::HALFW : p.HALFW;\n",
              "  const int g_HALFW = VEC == 2 ? WgVec2<false>::HALFW : p.HALFW;\n"
              "  if (__builtin_amdgcn_s_getreg(6148) & 1) __builtin_amdgcn_s_setprio(2);  // HW_ID.wave_id parity\n")],
}


def build():
    os.makedirs(OUT, exist_ok=True)
    base = open(os.path.join(SRC, "conv.hip")).read()
    for name, edits in VARIANTS.items():
        s = base
        for a, b in edits:
            assert s.count(a) == 1, (name, a[:60], s.count(a))
            s = s.replace(a, b)
        s = s.replace('#include "common.h"', f'#include "{SRC}/common.h"')
        path = os.path.join(OUT, f"conv_{name}.hip")
        open(path, "w").write(s)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
               "-Wno-unused-value", "-shared", "-o", os.path.join(OUT, f"libexp_{name}.so"), path]
        print(" ".join(cmd)); subprocess.check_call(cmd)


def timeit(call, n=20):
    import torch
    for _ in range(3): call()
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
        l = C.CDLL(os.path.join(OUT, f"libexp_{name}.so"))
        for fn in ("tbg_conv2d_f32", "tbg_conv2d_wgrad_ex_f32", "tbg_conv2d_wgrad_workspace_bytes"):
            getattr(l, fn).argtypes = getattr(P, fn).argtypes
            getattr(l, fn).restype = getattr(P, fn).restype
        return l
    print("--- fprop (3x3 SAME, B=16): product vs halo/filter staged once")
    libs = {"product": P, "fnostage": load("fnostage")}
    for Cc, M, H, W in ((128, 128, 32, 128), (256, 256, 16, 64), (128, 128, 64, 256), (64, 64, 64, 256), (256, 256, 8, 32)):
        x = torch.randn(16, Cc, H, W, device=dev); w = torch.randn(3, 3, Cc, M, device=dev)
        pf = ops.pack_filter(w.reshape(9, Cc, M), False, False)
        y = torch.empty(16, M, H, W, device=dev)
        d = N.ConvDesc(16, Cc, M, H, W, H, W, 3, 3, 1, 1, 1, 1, 0, 0, pf.M, 1)
        e = N.epilogue()
        flops = 2.0 * 16 * M * Cc * 9 * H * W
        line = f"fprop {Cc}->{M} {H}x{W} [{N.conv_kernel_name(d, False, False)}]: "
        for name, l in libs.items():
            def call():
                assert l.tbg_conv2d_f32(C.byref(d), N.ptr(x), N.ptr(pf.data), N.ptr(y), None, C.byref(e), N.stream()) == 0
            us = timeit(call)
            line += f"{name} {us:7.1f} us ({flops / us / 1e6:6.1f} TF)  "
        print(line)
    print("--- wgrad (3x3): product vs wave-priority variant")
    libs = {"product": P, "prio": load("prio")}
    for B, CS, CL, H, W, st in ((16, 128, 128, 64, 256, 1), (16, 128, 128, 32, 128, 1), (16, 256, 256, 16, 64, 1), (16, 128, 128, 65, 257, 2)):
        if st == 1:
            Hs, Ws, pad = H, W, 1
        else:
            Hs, Ws, pad = (H - 3) // 2 + 1, (W - 3) // 2 + 1, 0
        S = torch.randn(B, CS, Hs, Ws, device=dev); L = torch.randn(B, CL, H, W, device=dev)
        dW = torch.empty(9, CL, CS, device=dev)
        d = N.WgradDesc(B, CS, CL, Hs, Ws, H, W, 3, 3, st, st, pad, pad, CL * CS, CS, 1, 1.0)
        flops = 2.0 * B * Hs * Ws * CS * CL * 9
        line = f"wgrad B={B} {CS}x{CL} S={Hs}x{Ws} s{st} [{N.wgrad_kernel_name(d)}]: "
        for name, l in libs.items():
            wsb = l.tbg_conv2d_wgrad_workspace_bytes(C.byref(d))
            ws = torch.empty(max(wsb, 4) // 4, device=dev)
            def call():
                assert l.tbg_conv2d_wgrad_ex_f32(C.byref(d), N.ptr(S), N.ptr(L), N.ptr(dW), None, None, None, None, 0.0, N.ptr(ws), wsb, N.stream()) == 0
            us = timeit(call)
            line += f"{name} {us:7.1f} us ({flops / us / 1e6:6.1f} TF)  "
        print(line)


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else run()
