"""A/B of two builds of the library on the SAME box: graph-replayed plain step (f32x3, B=16), alternating child processes.
usage: python tools/ab_lib.py <libA.so> <libB.so> [rounds]     (child: python tools/ab_lib.py --child <lib.so>)
environment: AB_DTYPE (f32x3), AB_BATCH (16)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    from textboxgan_amd import native
    native.LIB_PATH = os.path.abspath(sys.argv[2])
    if len(sys.argv) > 3 and "=" in sys.argv[3]:  # an ops attribute that has to change with the library (mirrored rules)
        from textboxgan_amd import ops as _ops
        setattr(_ops.TUNING, sys.argv[3].split("=")[0], eval(sys.argv[3].split("=")[1]))
    import torch, time
    from textboxgan_amd.config import Config
    from textboxgan_amd.training_step import build_trainer_state
    from bench import synthetic_batch, bench_init_
    dev = torch.device('cuda:0')
    cfg = Config(batch_size_per_gpu=int(os.environ.get("AB_BATCH", "16")))
    b = synthetic_batch(cfg, dev, 1234)
    st = build_trainer_state(cfg, dev, seed=0, use_graphs=True, compute_dtype=os.environ.get("AB_DTYPE", "f32x3")); bench_init_(st)
    ts = st["training_step"]
    args = (b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], False, False, 1e-4)
    for _ in range(4): ts.dist_train_step(*args)
    res = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(32): ts.dist_train_step(*args)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 32 * 1e3)
    print("MS", min(res))
else:
    libs = sys.argv[1:3]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    extra = {libs[0]: sys.argv[4:5], libs[1]: sys.argv[5:6]}  # optional "attr=value" (an ops.TUNING attribute) per library
    out = {l: [] for l in libs}
    for r in range(rounds):
        for l in libs:
            o = subprocess.run([sys.executable, __file__, "--child", l] + extra[l], capture_output=True, text=True).stdout
            out[l].append(float([x for x in o.splitlines() if x.startswith("MS")][0].split()[1]))
    for l, v in out.items():
        print(f"{l}: " + " ".join(f"{x:.3f}" for x in v) + f"  mean {sum(v) / len(v):.3f} ms/step (plain step, graph replay)")
