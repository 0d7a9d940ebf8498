#!/usr/bin/env python
"""bench.py -- text-boxes/s through the full TextBoxGAN training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one ``TrainingStep.dist_train_step`` (G fwd, mask, D(fake), D(real), frozen OCR,
three backward passes, three Adam updates; path-length every 8th and R1 every 16th step per
the reference's config.py:81-94 and the caller protocol of train.py:178-208) + the g_clone EMA,
on synthetic words / N(0,1) latents / U(-1,1) "real" images already resident in HBM.
Workload = BASELINE.json configs[1]: per-GPU batch 16, 64x256 boxes, fp32, weak scaling
(global batch = 16 * N, the reference's rule config.py:141).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     : the dominant kernel -- the MFMA implicit-GEMM instantiation (forward / data gradient
                 or filter gradient) with the largest summed time -- algorithmic FLOPs of its
                 launches / their HIP-event durations, measured in a separate instrumented pass
                 over the same step (events on the launch stream); every other instantiation
                 is listed under roofline.all_kernels.  The epilogue instantiations of
                 conv_units_fprop_kernel<NP, WTM, OPT> (same main loop and tile, OPT = which optional
                 epilogue operands are compiled in) count as ONE kernel "<NP, WTM, *>": its members and
                 their committed rocprofv3 rows are listed in roofline.instantiations / timed_region.
  cpu_baseline : the CPU restatement of the same step (oracle/, PyTorch-oneDNN, "port") timed on
                 this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense (~2.5 PF; never the 2:1-sparsity figure)
CONV_GFLOP_PER_IMAGE = 85.2   # dense-conv work of one non-regularised step


def synthetic_batch(cfg, device, seed):
    """words U{1..69} of length U{1..8}, ASTER labels, U(-1,1) images zero-padded right."""
    from textboxgan_amd.char_tokens import main_to_aster_labels
    g = np.random.default_rng(seed)
    B = cfg.batch_size_per_gpu
    L = g.integers(1, cfg.max_char_number + 1, size=B)
    words = np.zeros((B, cfg.max_char_number), dtype=np.int32)
    for b in range(B):
        words[b, : L[b]] = g.integers(1, cfg.main_vocab + 1, size=L[b])
    real = g.uniform(-1, 1, size=(B, 3, cfg.char_height, cfg.image_width)).astype(np.float32)
    for b in range(B):
        real[b, :, :, cfg.char_width * L[b]:] = 0.0
    return dict(real_images=torch.from_numpy(real).to(device), ocr_images=torch.zeros((), device=device),
                input_words=torch.from_numpy(words).to(device),
                ocr_labels=torch.from_numpy(main_to_aster_labels(words)).to(device))


def bench_init_(state):
    """non-trivial epilogues: noise_strength 0.1, biases ~ N(0, 0.1)."""
    g = torch.Generator().manual_seed(4321)
    with torch.no_grad():
        for mod in (state["generator"], state["discriminator"]):
            for n, p in mod.named_parameters():
                if n.endswith("noise_strength"):
                    p.fill_(0.1)
                elif n.endswith(".b") or n.endswith(".bias"):
                    p.copy_((torch.randn(p.shape, generator=g) * 0.1).to(p.device))
        state["g_clone"].load_state_dict(state["generator"].state_dict())


def run_steps(state, batch, n, ocr_w_late=True):
    ts = state["training_step"]
    for _ in range(n):
        step = ts.g_optimizer.iterations
        do_r1 = (step + 1) % ts.d_reg_interval == 0
        do_pl = (step + 1) % ts.g_reg_interval == 0
        ts.dist_train_step(batch["real_images"], batch["ocr_images"], batch["input_words"], batch["ocr_labels"],
                           do_r1, do_pl, 1e-4 if ocr_w_late else 1e-8)
        state["g_clone"].set_as_moving_average_of(state["generator"])


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(batch_size=4, warmup=1, steps=3, threads=None):
    """CPU restatement of the reference path (PyTorch/oneDNN, "port"), NOT the TF2 reference (TensorFlow is absent; the
    reference's CPU mode additionally inserts NCHW<->NHWC transposes and uses the non-fused modconv, utils.py:146-154).
    The default bench line uses cpu_baseline_bounded() (about 25 s of host work) so that `python bench.py` stays within minutes; the full
    SURVEY 8(d) protocol (batch 16, 3 warm-up + 10 timed steps, all cores and 8 threads) is `--cpu-baseline-full`."""
    from oracle import ref_model as M
    from textboxgan_amd.aster import AsterLikeOCR
    from textboxgan_amd.config import Config
    prev = torch.get_num_threads()
    if threads:
        torch.set_num_threads(threads)
    try:
        cfg = Config(batch_size_per_gpu=batch_size)
        st = M.make_state(cfg, 0, bench_init=True)
        batch, rand = M.make_batch(cfg), M.make_rand(cfg, with_pl=False)
        from oracle.ref_ocr import OcrOracle  # the oracle's own network code; the product module only supplies the frozen weights
        ocr = OcrOracle(AsterLikeOCR(max_steps=cfg.max_char_number).state_dict(), max_steps=cfg.max_char_number, dtype=torch.float32)
        args = (batch["real_images"], batch["ocr_images"], batch["input_words"], batch["ocr_labels"], False, False, 1e-4)
        for _ in range(warmup):  # oneDNN primitive creation
            M.training_step(st, cfg, *args, rand, ocr.serve)
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            M.training_step(st, cfg, *args, rand, ocr.serve)
            ts.append(time.perf_counter() - t0)
        med = float(np.median(ts))
        return dict(value=round(batch_size / med, 3), unit="text-boxes/s", cores=torch.get_num_threads(), kind="port",
                    cpu_model=_cpu_model(), host_logical_cpus=os.cpu_count(), s_per_step_median=round(med, 2),
                    s_per_step_min=round(min(ts), 2),
                    sample=f"{steps} non-regularised full-size steps (64x256, full channel widths) at batch {batch_size} "
                           f"after {warmup} warm-up step(s), median; oracle/ref_model.py training_step (torch-CPU fp32, "
                           f"oneDNN; the per-sample OCR loop of aster_inferer.py:28-37 included); the GPU line is batch 16"
                           + ("" if batch_size == 16 else f" -- the bounded default samples batch {batch_size} because one "
                                                          f"batch-16 CPU step takes ~4x as long"))
    finally:
        torch.set_num_threads(prev)


def cpu_baseline_bounded():
    """the default bench line's CPU leg: batch 4, 1 warm-up + 3 timed steps at 8 and at 32 threads (about 25 s of host work
    together), the faster one reported with its thread count in `cores`.  All 128 hardware threads are SLOWER on this
    workload (profiles/r02_f_bench_cpu_baseline_full.json: 0.37 text-boxes/s at batch 16 on 128 threads against 1.19 on 8
    threads at batch 4 -- oneDNN over-subscription on small per-sample work), so they are left to --cpu-baseline-full."""
    runs = [cpu_baseline(batch_size=4, warmup=1, steps=3, threads=t) for t in (8, 32)]
    best = max(runs, key=lambda r: r["value"])
    best = dict(best)
    best["thread_counts_tried"] = {str(r["cores"]): r["value"] for r in runs}
    return best


def cpu_baseline_full_cached():
    """SURVEY 8(d) protocol (batch 16, 3 warm-up + 10 timed steps, ALL cores; plus 8 threads at batch 4) as last measured with
    `bench.py --cpu-baseline-full` and committed under profiles/ -- it takes ~15 min of host time, so the default line carries
    the cached record with its provenance next to the live bounded sample."""
    for name in ("r04_bench_cpu_baseline_full.json", "r02_f_bench_cpu_baseline_full.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        return {"source": "profiles/" + name, "command": "python bench.py --cpu-baseline-full",
                "all_cores_batch16": d.get("cpu_baseline"), "threads8_batch4": d.get("cpu_baseline_8_threads")}
    return None


class _TinyOCR(torch.nn.Module):
    """Stand-in for the measurement that EXCLUDES the OCR network (the real ASTER net is absent, so
    the headline contains an unknowable share of made-up work): logits = a fixed linear map of column-pooled pixels.
    The generator's ocr-pass backward (20.3 GFLOP/img of the 85.2) still runs -- only the recogniser is removed."""

    def __init__(self, steps=8, classes=97):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.steps, self.classes = steps, classes
        self.register_buffer("proj", torch.randn(3 * 32, classes, generator=g) * 0.1)

    def forward_logits(self, img_nchw):
        B = img_nchw.shape[0]
        cols = torch.nn.functional.adaptive_avg_pool2d(img_nchw, (32, self.steps))  # [B,3,32,steps]
        feats = cols.permute(0, 3, 1, 2).reshape(B, self.steps, -1)
        logits = feats @ self.proj
        return logits, torch.full((B,), self.steps, device=logits.device, dtype=torch.int64)


HBM_PEAK_TBS = 8.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (~6.3 achievable)


def _roofline_family(name):
    import re
    m = re.match(r"(conv_units_fprop_kernel<\d+, \d+), \d+>$", name)
    return m.group(1) + ", *>" if m else name


def roofline_record(recs, dtype="f32"):
    """dominant MFMA kernel instantiation (largest summed time over the forward / data-gradient AND filter-gradient
    instantiations): algorithmic FLOPs / HIP-event time of its launches.  (A filter-gradient record brackets the C-ABI call,
    i.e. the kernel plus its partial-tile reduce launch; rocprofv3's average in profiles/ is the kernel alone.)"""
    convs = {k: v for k, v in recs.items() if k.startswith("conv_")}
    # the epilogue instantiations of conv_units_fprop_kernel<NP, WTM, OPT> (OPT = which optional epilogue operands are compiled in:
    # same main loop, same tile) are ONE roofline entry "<NP, WTM, *>"; its members and their rocprofv3 rows are listed beside it
    fams = {}
    for k, v in convs.items():
        f = fams.setdefault(_roofline_family(k), {"n": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "members": {}})
        for q in ("n", "ms", "flops", "bytes"):
            f[q] += v[q]
        f["members"][k] = v
    name, r = max(((k, v) for k, v in fams.items() if v["flops"] > 0), key=lambda kv: kv[1]["ms"])
    members = r["members"]
    achieved = r["flops"] / (r["ms"] * 1e-3) / 1e12
    # HBM traffic cannot be read from inside the process: it comes from the committed rocprofv3 --pmc passes over
    # `bench.py --roofline-only` (tools/pmc_report.sh, tools/make_traffic_json.py), matched by kernel instantiation; null if
    # none is committed
    traffic, traffic_src, mfma_busy, timed = None, None, None, None
    try:
        tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "latest_traffic.json")))
        rows = {k: tj["kernels"][k] for k in members if k in tj["kernels"]}
        if rows and len(rows) == len(members):  # launch-weighted over the members' committed rows
            nl = sum(x["launches"] for x in rows.values())
            traffic = round(sum(x["hbm_bytes_per_launch"] * x["launches"] for x in rows.values()) / nl)
            traffic_src = next(iter(rows.values())).get("source", tj["source"])
            busy = [x for x in rows.values() if x.get("mfma_busy") is not None]
            if busy:  # time-weighted
                tw = sum(x["launches"] * x["avg_us"] for x in busy)
                mfma_busy = round(sum(x["mfma_busy"] * x["launches"] * x["avg_us"] for x in busy) / tw, 3)
            if len(rows) > 1:
                timed = ("avg_launch_us is the launch-weighted average over the epilogue instantiations of one kernel; rocprofv3's rows in "
                         f"{traffic_src}: " + "; ".join(f"{k}: {x['launches']} x {x['avg_us']} us" for k, x in sorted(rows.items())))
            elif "wgrad" in name:  # what avg_launch_us brackets, and the committed kernel-only duration it has to be read against
                timed = (f"avg_launch_us brackets one C-ABI call = {name} + its partial-tile reduce launch (conv_wgrad_reduce_*); "
                         f"rocprofv3's row for the kernel alone in {traffic_src}: {rows[name].get('avg_us')} us")
    except (OSError, ValueError, KeyError):
        pass
    if dtype == "f32x3":
        # split-operand arithmetic on the bf16 pipe: 6 bf16 products per fp32 product -> the roof in fp32-equivalent FLOPs is
        # the dense bf16 MFMA peak / 6
        peak = BF16_MFMA_PEAK_TFLOPS / 6.0
        return {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "peak_note": "fp32-equivalent: 2500 TFLOP/s dense bf16 MFMA / 6 partial products (3xbf16 split operands)",
                "frac_of_exact_f32_peak": round(achieved / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "algorithmic_bytes": round(r["bytes"] / r["n"]),
                "traffic_unit": "bytes/launch (HBM, PMC); algorithmic_bytes = input + output + filter once each, per launch",
                "traffic_source": traffic_src, "mfma_busy_pmc": mfma_busy, "timed_region": timed, "instantiations": {k: {"n": v["n"], "avg_launch_us": round(1e3 * v["ms"] / v["n"], 2)} for k, v in members.items()},
                "launches": r["n"], "avg_launch_us": round(1e3 * r["ms"] / r["n"], 2),
                "gflop_per_launch": round(r["flops"] / r["n"] / 1e9, 3),
                "all_kernels": {k: {"n": v["n"], "ms": round(v["ms"], 3),
                                    "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in convs.items()},
                "all_conv_kernels": (lambda fl, ms: {"achieved": round(fl / (ms * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                                                     "ms": round(ms, 3), "launches": sum(v["n"] for v in convs.values())})(
                    sum(v["flops"] for v in convs.values()), sum(v["ms"] for v in convs.values())),
                "hbm_bound_kernels": {k: {"n": v["n"], "ms": round(v["ms"], 3),
                                          "algorithmic_mb_per_launch": round(v["bytes"] / v["n"] / 1e6, 3),
                                          "achieved_tb_s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e12, 3),
                                          "frac_of_8tb_s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e12 / HBM_PEAK_TBS, 3)}
                                      for k, v in recs.items() if not k.startswith("conv_") and v["bytes"] > 0 and v["ms"] > 0}}
    if dtype == "bf16":
        # bf16-in MFMA with fp32 tensors in HBM: the dominant 3x3 convolutions sit below the bf16 ridge (SURVEY 8(d):
        # ~288 FLOP/B of fp32 traffic vs a ridge of 2500/8 = 312), so the bounding roof is HBM; both fractions are reported.
        tbs = r["bytes"] / (r["ms"] * 1e-3) / 1e12
        return {"bound": "hbm", "kernel": name, "achieved": round(tbs * 1e3, 1), "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s",
                "frac": round(tbs / HBM_PEAK_TBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "mfma_busy_pmc": mfma_busy, "timed_region": timed, "instantiations": {k: {"n": v["n"], "avg_launch_us": round(1e3 * v["ms"] / v["n"], 2)} for k, v in members.items()}, "algorithmic_bytes": round(r["bytes"] / r["n"]),
                "launches": r["n"], "avg_launch_us": round(1e3 * r["ms"] / r["n"], 2),
                "gflop_per_launch": round(r["flops"] / r["n"] / 1e9, 3), "achieved_tflops": round(achieved, 2),
                "mfma_peak_tflops": BF16_MFMA_PEAK_TFLOPS, "frac_of_mfma_peak": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4),
                "all_kernels": {k: {"n": v["n"], "ms": round(v["ms"], 3),
                                    "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                    "tb_s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e12, 3)} for k, v in convs.items()}}
    hbm = {}
    for k, v in recs.items():
        if not k.startswith("conv_") and v["bytes"] > 0 and v["ms"] > 0:
            tbs = v["bytes"] / (v["ms"] * 1e-3) / 1e12
            hbm[k] = {"n": v["n"], "ms": round(v["ms"], 3), "algorithmic_mb_per_launch": round(v["bytes"] / v["n"] / 1e6, 3),
                      "achieved_tb_s": round(tbs, 3), "frac_of_8tb_s": round(tbs / HBM_PEAK_TBS, 3)}
    return {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": F32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
            "algorithmic_bytes": round(r["bytes"] / r["n"]),
            "traffic_unit": "bytes/launch (HBM, PMC); algorithmic_bytes = input + output + filter once each, per launch",
            "traffic_source": traffic_src, "mfma_busy_pmc": mfma_busy, "timed_region": timed, "instantiations": {k: {"n": v["n"], "avg_launch_us": round(1e3 * v["ms"] / v["n"], 2)} for k, v in members.items()},
            "launches": r["n"], "avg_launch_us": round(1e3 * r["ms"] / r["n"], 2),
            "gflop_per_launch": round(r["flops"] / r["n"] / 1e9, 3),
            "all_kernels": {k: {"n": v["n"], "ms": round(v["ms"], 3),
                                "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in convs.items()},
            # every MFMA launch of the pass together (forward, data- and filter-gradient instantiations): FLOP-weighted
            "all_conv_kernels": (lambda fl, ms: {"achieved": round(fl / (ms * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                                                 "frac": round(fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                                                 "ms": round(ms, 3), "launches": sum(v["n"] for v in convs.values())})(
                sum(v["flops"] for v in convs.values()), sum(v["ms"] for v in convs.values())),
            "hbm_bound_kernels": hbm}


HEADLINE_ARITH = "f32x3"  # arithmetic of the default (BASELINE configs[1]) line; --dtype f32 gives the exact-fp32 MFMA line


def graph_mode(ts):
    """TrainingStep.graph_mode: "single" | "split" | "two-phase" | "eager"."""
    return ts.graph_mode


def timed_loop(state, batch, steps, warmup, world):
    """W untimed + K timed steps in the caller protocol's cadence, bracketed by barrier + synchronize; max over ranks."""
    run_steps(state, batch, warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(state, batch, steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=batch["real_images"].device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def variant_step_ms(state, batch, reps=(4, 2, 2), blocks=3):
    """milliseconds of ONE step of each lazy-regularisation variant (SURVEY 8(d) Config 2: "non-reg step and the 16-step
    average"): plain, +PL (every 8th step), +PL+R1 (every 16th).  Real optimisation steps, flags forced.  Each variant is timed as
    `blocks` back-to-back blocks of n steps (one synchronize per block) and the MEDIAN block is reported: a single host hiccup
    (seen once: +21 ms inside a 6-step block, profiles/r05_o_bench.json configs2_bf16) moves one block, not the figure."""
    ts = state["training_step"]
    a = (batch["real_images"], batch["ocr_images"], batch["input_words"], batch["ocr_labels"])
    out = {}
    for name, (r1, pl), n in (("plain", (False, False), reps[0]), ("pl", (False, True), reps[1]), ("pl_r1", (True, True), reps[2])):
        ts.dist_train_step(*a, r1, pl, 1e-4)  # untimed
        torch.cuda.synchronize()
        per = []
        for _ in range(blocks):
            t0 = time.perf_counter()
            for _ in range(n):
                ts.dist_train_step(*a, r1, pl, 1e-4)
                state["g_clone"].set_as_moving_average_of(state["generator"])
            torch.cuda.synchronize()
            per.append(1e3 * (time.perf_counter() - t0) / n)
        out[name] = round(sorted(per)[len(per) // 2], 3)
    return out


def cycle_value(step_ms, batch):
    """text-boxes/s over one aligned 16-step cycle: 14 plain + 1 PL + 1 PL+R1 steps (config.py:86,93)."""
    return round(batch * 16 / ((14 * step_ms["plain"] + step_ms["pl"] + step_ms["pl_r1"]) * 1e-3), 2)


WORKLOAD_ARITH = {
    "f32": "fp32, exact v_mfma_f32_32x32x2_f32 contractions",
    "f32x3": "fp32 tensors; every MFMA contraction (forward, data gradient, filter gradient) as 3xbf16 split operands, 6 products, "
             "fp32 accumulate (tbg_conv2d_x3 / tbg_conv2d_units: fp32-grade error, passes every fp32 parity test at the fp32 "
             "tolerances); the large 3x3 stride-1 layers read their operands from unit tensors (tbg.h) written once per tensor",
    "bf16": "bf16 MFMA operands / fp32 accumulate / fp32 master weights + Adam",
}


def build_state(cfg, device, dtype, graphs, tiny_ocr=False):
    from textboxgan_amd.training_step import build_trainer_state
    ocr = None
    if tiny_ocr:
        from textboxgan_amd.aster import AsterInferer
        ocr = AsterInferer(model=_TinyOCR(cfg.max_char_number))
    state = build_trainer_state(cfg, device, aster_ocr=ocr, seed=0, use_graphs=graphs, compute_dtype=dtype)  # identical replicas
    bench_init_(state)
    return state


def roofline_pass(state, batch, dtype, steps=2):
    """same step, eager, every instrumented launch bracketed by HIP events on its stream"""
    from textboxgan_amd import ops
    ts = state["training_step"]
    was = ts.use_graphs
    ts.use_graphs = False
    ops.PROFILE.enable()
    try:
        for _ in range(steps):
            ts.dist_train_step(batch["real_images"], batch["ocr_images"], batch["input_words"], batch["ocr_labels"], False,
                               False, 1e-4)
        torch.cuda.synchronize()
        recs = ops.PROFILE.collect()
    finally:
        ops.PROFILE.disable()
        ts.use_graphs = was
    return roofline_record(recs, dtype) if recs else None


def sub_record(device, dtype, per_gpu_batch, steps, warmup, graphs, with_ocr_excluded=True, with_roofline=True):
    """a complete single-GPU measurement of another configuration inside the same JSON line (configs[2], exact fp32)"""
    from textboxgan_amd.config import Config
    cfg = Config(batch_size_per_gpu=per_gpu_batch, num_replicas=1)
    batch = synthetic_batch(cfg, device, 1234)
    state = build_state(cfg, device, dtype, graphs)
    a = (batch["real_images"], batch["ocr_images"], batch["input_words"], batch["ocr_labels"])
    if graphs:
        state["training_step"].prepare_graphs(*a)
    dt = timed_loop(state, batch, steps, warmup, 1)
    rec = {"dtype": dtype, "per_gpu_batch": per_gpu_batch, "steps": steps, "warmup": warmup,
           "value": round(per_gpu_batch * steps / dt, 2), "unit": "text-boxes/s", "ms_per_step": round(1e3 * dt / steps, 3),
           "arithmetic": WORKLOAD_ARITH[dtype], "graph_mode": graph_mode(state["training_step"])}
    rec["step_ms"] = variant_step_ms(state, batch)
    rec["value_16step"] = cycle_value(rec["step_ms"], per_gpu_batch)
    rec["value_window"], rec["ms_per_step_window"] = rec["value"], rec["ms_per_step"]
    rec["value"] = rec["value_16step"]  # same basis as the headline: one aligned 16-step cycle
    rec["ms_per_step"] = round((14 * rec["step_ms"]["plain"] + rec["step_ms"]["pl"] + rec["step_ms"]["pl_r1"]) / 16, 3)
    if with_roofline:
        rec["roofline"] = roofline_pass(state, batch, dtype)
    del state
    torch.cuda.empty_cache()
    if with_ocr_excluded:
        st2 = build_state(cfg, device, dtype, graphs, tiny_ocr=True)
        if graphs:
            st2["training_step"].prepare_graphs(*a)
        dt2 = timed_loop(st2, batch, steps, warmup, 1)
        rec["ocr_excluded_value"] = round(per_gpu_batch * steps / dt2, 2)
        rec["ocr_excluded_ms_per_step"] = round(1e3 * dt2 / steps, 3)
        del st2
        torch.cuda.empty_cache()
    return rec


def dist_record(state, batch, world, backend, steps=6):
    """N > 1: what the first RCCL run needs on record: collective bandwidth of the three
    gradient buffers alone, and the step time with / without the exchange -> how much of it the backward passes hide."""
    ts = state["training_step"]
    dev = batch["real_images"].device
    bufs = [ts.g_grad, ts.o_grad, ts.d_grad]
    nbytes = sum(b.numel() for b in bufs) * 4
    for _ in range(2):
        for b in bufs:
            dist.all_reduce(b, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        for b in bufs:
            dist.all_reduce(b, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    t_ar = (time.perf_counter() - t0) / 5
    tt = torch.tensor([t_ar], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t_ar = float(tt.item())
    a = (batch["real_images"], batch["ocr_images"], batch["input_words"], batch["ocr_labels"], False, False, 1e-4)

    def plain_ms(n):
        for _ in range(2):
            ts.dist_train_step(*a)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            ts.dist_train_step(*a)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / n], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return 1e3 * float(t.item())

    with_x = plain_ms(steps)
    ts.exchange.muted = True  # same launches, no gradient collectives (replicas drift apart: measurement only, last thing run)
    try:
        without_x = plain_ms(steps)
    finally:
        ts.exchange.muted = False
    exposed = max(with_x - without_x, 0.0)
    return {"world_size": dist.get_world_size(), "backend": backend, "graph_mode": graph_mode(ts),
            "gradient_bytes_per_step": nbytes,
            "allreduce_alone_ms": round(1e3 * t_ar, 3),
            "bus_bw_GBps": round(2 * (world - 1) / world * nbytes / t_ar / 1e9, 1),
            "bus_bw_convention": "2(N-1)/N * S / t over the three flat gradient buffers (39.96 + 34.75 + 62.38 MB)",
            "plain_step_ms_with_exchange": round(with_x, 3), "plain_step_ms_without_exchange": round(without_x, 3),
            "overlap_frac": round(1.0 - min(exposed / (1e3 * t_ar), 1.0), 3) if t_ar > 0 else None,
            "capture_error": ts.capture_error}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (BASELINE configs[1]: 16; configs[2]: 32)")
    ap.add_argument("--dtype", choices=("auto", "f32", "f32x3", "bf16"), default="auto",
                    help="arithmetic of the MFMA contractions.  auto = the BASELINE configs[1] headline (fp32 tensors, "
                         f"{HEADLINE_ARITH} contractions); f32 = exact fp32 MFMA; bf16 = configs[2] (default batch 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="SURVEY 8(d) protocol instead of the bounded sample: batch 16, 3 warm-up + 10 timed steps at all "
                         "cores, plus a batch-4 1+3-step run at 8 threads (takes ~15 min of host time)")
    ap.add_argument("--tiny-ocr", action="store_true",
                    help="profiling aid: run the MAIN loop with the OCR network replaced by the trivial stand-in (the "
                         "G + D part of the step alone); the printed line is then not the BASELINE metric")
    ap.add_argument("--no-ocr-excluded", action="store_true",
                    help="skip the second timed loop that replaces the (guessed) OCR network by a trivial stand-in")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sub-records", action="store_true",
                    help="skip the extra single-GPU measurements appended to the default line: configs2_bf16 (BASELINE "
                         "configs[2]: bf16, batch 32) and exact_f32 (the same configs[1] step on exact fp32 MFMA)")
    ap.add_argument("--no-graphs", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: accept a run whose split-graph capture failed (graph_mode != 'split'); without it the job "
                         "exits non-zero instead of reporting a silently slower eager / two-phase number")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the instrumented roofline pass (eager, non-regularised steps): the command the "
                         "profiles/*_roofline_kernel_stats.txt rocprofv3 summaries are taken with, so that rocprof's "
                         "per-kernel average covers the same launches as roofline.avg_launch_us")
    args = ap.parse_args()
    headline = args.dtype == "auto"
    if headline:
        args.dtype = HEADLINE_ARITH
    if args.batch is None:
        args.batch = 32 if args.dtype == "bf16" else 16

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("TBG_BENCH_SINGLE_DEVICE"):  # launcher smoke test on a 1-GPU box: all ranks share cuda:0
        local_rank = 0
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("TBG_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from textboxgan_amd import ops
    from textboxgan_amd.config import Config

    cfg = Config(batch_size_per_gpu=args.batch, num_replicas=world)
    if args.tiny_ocr:
        args.no_ocr_excluded = True
    graphs = not args.no_graphs
    state = build_state(cfg, device, args.dtype, graphs, tiny_ocr=args.tiny_ocr)
    batch = synthetic_batch(cfg, device, 1234 + rank)
    a4 = (batch["real_images"], batch["ocr_images"], batch["input_words"], batch["ocr_labels"])

    if args.roofline_only:
        ts = state["training_step"]; ts.use_graphs = False
        a = (*a4, False, False, 1e-4)
        for _ in range(args.warmup):
            ts.dist_train_step(*a)
        torch.cuda.synchronize()
        ops.PROFILE.enable()
        for _ in range(args.steps):
            ts.dist_train_step(*a)
        print(json.dumps({"roofline": roofline_record(ops.PROFILE.collect(), args.dtype), "steps": args.steps,
                          "warmup": args.warmup}), flush=True)
        return
    if graphs:  # untimed: warm up + capture the three step variants (6 real steps)
        state["training_step"].prepare_graphs(*a4)
    ts = state["training_step"]
    lost = world > 1 and graphs and graph_mode(ts) != "split"
    if world > 1 and graphs:  # every rank takes the same branch: one rank leaving alone would hang the others' collectives
        flag = torch.tensor([1.0 if lost else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        lost_any = bool(flag.item() > 0)
    else:
        lost_any = False
    if lost_any and not args.allow_fallback:
        # the overlapped exchange IS the N > 1 design: a run that lost it must not produce a number
        sys.stderr.write(f"[bench] rank {rank}: graph_mode={graph_mode(ts)!r} (wanted 'split'); capture_error="
                         f"{ts.capture_error!r}; pass --allow-fallback to measure the fallback anyway\n")
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(3)
    dt = timed_loop(state, batch, args.steps, args.warmup, world)

    out = None
    if rank == 0:
        value = args.batch * world * args.steps / dt
        cfgname = "BASELINE configs[2]" if args.dtype == "bf16" else "BASELINE configs[1]"
        out = {
            "metric": "text-boxes/sec (G+D+OCR training_step)" + (" [OCR NETWORK EXCLUDED: --tiny-ocr]" if args.tiny_ocr else ""),
            "value": round(value, 2), "unit": "text-boxes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": (f"training_step G+D+OCR, bs={args.batch}/GPU, 64x256 boxes, max_char_number=8, " +
                                    f"{WORKLOAD_ARITH[args.dtype]} ({cfgname})" +
                                    "; PL every 8th / R1 every 16th step (config.py:81-94); "
                                    "g_clone EMA included; OCR = ASTER-shaped frozen net (synthetic weights)"),
                       "arithmetic": args.dtype,
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       # which BASELINE.json configuration this line is: N = 1 -> configs[1] / configs[2]; N > 1 -> the SAME per-GPU
                       # workload on every rank (weak scaling, the reference's own rule batch_size = per_gpu x replicas,
                       # config/config.py:140-141), so that the driver's 1/2/4/8 efficiency compares like with like; BASELINE
                       # configs[3] (8 GPUs, global 256 = 8 x 32) is this command with --batch 32 (any --dtype)
                       "baseline_config": ("configs[3] (8 x 32 = global 256)" if (world == 8 and args.batch == 32) else
                                           cfgname.split()[-1] + (f" per GPU x {world} ranks (weak scaling; configs[3] = --gpus 8 --batch 32)"
                                                                  if world > 1 else "")),
                       "conv_gflop_per_image_nonreg_step": CONV_GFLOP_PER_IMAGE},
            "conv_tflops_vs_step_time": round(value * CONV_GFLOP_PER_IMAGE / 1e3 / world, 2),
            "graph_mode": graph_mode(ts), "capture_error": ts.capture_error,
            # peak device memory of the timed loop (all lazy-regularisation variants captured; torch's caching allocator): the unit
            # tensors kept beside the fp32 activations (ops.TUNING.save_units) are in here
            "peak_hbm_gb": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
        }
    if world == 1:
        # headline = one ALIGNED 16-step cycle of the caller protocol (14 plain + 1 PL + 1 PL+R1 steps, config.py:86,93), from
        # the per-variant step times: the K-step window of the command line holds a K-dependent share of regularised steps
        # (20 steps: 2 + 2 where a cycle holds 2 + 1 in 16) and stays on record as value_window / ms_per_step_window
        sm = variant_step_ms(state, batch)
        out["step_ms"] = sm
        out["value_window"], out["ms_per_step_window"] = out["value"], out["ms_per_step"]
        out["value_16step"] = cycle_value(sm, args.batch)
        out["value"] = out["value_16step"]
        out["ms_per_step"] = round((14 * sm["plain"] + sm["pl"] + sm["pl_r1"]) / 16, 3)
        out["value_basis"] = ("value and ms_per_step are the average of ONE ALIGNED 16-STEP CYCLE (14 plain + 1 PL + 1 PL+R1) built "
                              "from step_ms (each variant: the median of 3 separately timed blocks of 4 / 2 / 2 steps) -- NOT window / steps: steps * "
                              "ms_per_step is not a wall interval.  The wall-clock figure of the --steps window (barrier + "
                              "synchronize on both sides, images * steps / wall time) is value_window / ms_per_step_window, and "
                              "steps * ms_per_step_window is that interval")
        out["conv_tflops_vs_step_time"] = round(out["value"] * CONV_GFLOP_PER_IMAGE / 1e3, 2)
    else:
        dr = dist_record(state, batch, world, backend)
        if rank == 0:
            out["dist"] = dr

    # ---- roofline pass: same step, every conv launch bracketed by HIP events on its stream
    if rank == 0 and world == 1 and not args.no_roofline:
        rl = roofline_pass(state, batch, args.dtype)
        if rl:
            out["roofline"] = rl
    if world > 1:
        dist.barrier()
    # ---- same loop with the OCR NETWORK excluded (the ASTER-shaped stand-in is a guess)
    del state, ts
    torch.cuda.empty_cache()
    if world == 1 and not args.no_ocr_excluded:
        st2 = build_state(cfg, device, args.dtype, graphs, tiny_ocr=True)
        if graphs:
            st2["training_step"].prepare_graphs(*a4)
        dt2 = timed_loop(st2, batch, args.steps, args.warmup, 1)
        out["ocr_excluded_value"] = round(args.batch * args.steps / dt2, 2)
        out["ocr_excluded_ms_per_step"] = round(1e3 * dt2 / args.steps, 3)
        del st2
        torch.cuda.empty_cache()
    # ---- the other single-GPU configurations, inside the same driver-run line
    if world == 1 and headline and not args.no_sub_records and not args.tiny_ocr:
        out["configs2_bf16"] = sub_record(device, "bf16", 32, 8, 2, graphs)
        out["configs2_bf16"]["workload"] = ("BASELINE configs[2]: training_step + R1 + path-length regularisation, bs=32, bf16 "
                                            "MFMA operands / fp32 accumulate / fp32 master weights + Adam, 1 MI355X; reg "
                                            "cadence of config.py (PL/8, R1/16)")
        if HEADLINE_ARITH != "f32":
            out["exact_f32"] = sub_record(device, "f32", 16, 8, 2, graphs, with_ocr_excluded=False)
            out["exact_f32"]["workload"] = "BASELINE configs[1] on exact fp32 MFMA (v_mfma_f32_32x32x2_f32) for every contraction"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if args.cpu_baseline_full:
            out["cpu_baseline"] = cpu_baseline(batch_size=16, warmup=3, steps=10)
            out["cpu_baseline_8_threads"] = cpu_baseline(batch_size=4, warmup=1, steps=3, threads=8)
        else:
            out["cpu_baseline"] = cpu_baseline_bounded()
            out["cpu_baseline"]["full_protocol_cached"] = cpu_baseline_full_cached()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
