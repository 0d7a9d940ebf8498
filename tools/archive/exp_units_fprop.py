"""time conv_units_fprop_kernel ablations (tools/exp_units_fprop.sh) on the largest layers: python tools/exp_units_fprop.py"""
import subprocess, sys, os
if len(sys.argv) > 1:
    sys.path.insert(0, ".")
    import torch
    from textboxgan_amd import native as N
    N.LIB_PATH = os.path.abspath(sys.argv[1])
    from textboxgan_amd import ops
    dev = torch.device("cuda:0")
    out = []
    for (B, C, M, H, W) in [(16, 128, 128, 64, 256), (32, 128, 128, 32, 128)]:
        x = torch.randn(B, C, H, W, device=dev); w = torch.randn(3, 3, C, M, device=dev) / (9 * C) ** 0.5
        dd, bs, nz, st = torch.rand(B, M, device=dev) + 0.5, torch.randn(M, device=dev), torch.randn(B, 1, H, W, device=dev), torch.tensor(0.1, device=dev)
        with ops.compute_dtype("f32x3"):
            pf = ops.pack_filter(w, False, False); XU = ops.units_pack(x); y = torch.empty(B, M, H, W, device=dev)
            f = lambda: ops.conv2d_units_raw(XU, pf, M, epi=N.epilogue(out_scale=dd, bias=bs, noise=nz, strength=st, act=N.ACT_LRELU), out=y)
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 100
            out.append(f"{t:7.1f} us ({2.0 * B * C * M * H * W * 9 / t / 1e6:6.1f} TF)")
    print(sys.argv[1].split("/")[-1], "  ".join(out), flush=True)
else:
    names = {0: "product", 1: "no epilogue", 2: "no DMA in the K loop", 3: "no sched_barrier pins", 4: "no operand reads in the K loop"}
    for e in range(5):
        print(f"EXP {e} ({names[e]}):", end=" ", flush=True)
        subprocess.run([sys.executable, __file__, f"tools/scratch/libexp{e}.so"])
