"""-m gpu: the data-parallel step with world_size 2 (two processes; gloo collectives on one shared GPU, or RCCL when the
box has two GPUs -- the same code path bench.py drives): replicas stay bit-identical, losses are SUM-reduced, and the
exchanged flat gradient buffers equal the SUM of the per-replica gradients of the CPU oracle (each replica: its own
shard, its own randomness, minibatch-std inside the replica, losses divided by the GLOBAL batch -- exactly what
MirroredStrategy does in the reference, training_step.py:91-136,233-235)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, use_graphs, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ref_model as M  # only for the synthetic batch helper
    from textboxgan_amd.config import small_config
    from textboxgan_amd.training_step import build_trainer_state
    dev = torch.device("cuda:0")
    cfg = small_config(4, num_replicas=world)
    st = build_trainer_state(cfg, dev, seed=0, use_graphs=use_graphs)  # identical replicas
    ts = st["training_step"]
    assert ts.distributed and ts.batch_size == 8
    b = {k: v.to(dev) for k, v in M.make_batch(cfg, seed=1234, rank=rank).items()}  # different shard per rank
    torch.manual_seed(100 + rank)
    losses = None
    for i in range(5 if use_graphs else 2):
        losses = ts.dist_train_step(b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], False, False, 1e-4)
        st["g_clone"].set_as_moving_average_of(st["generator"])
    torch.cuda.synchronize()
    gsum = float(st["generator"]._flat.flat.double().sum())
    dsum = float(st["discriminator"]._flat.flat.double().sum())
    gabs = float(st["generator"]._flat.flat.double().abs().sum())
    q.put((rank, gsum, dsum, gabs, [float(x) for x in losses[0]] + [float(x) for x in losses[1]] + [float(losses[2])],
           ts.g_optimizer.iterations, int(ts.g_optimizer.step.item())))
    dist.destroy_process_group()


@pytest.mark.parametrize("use_graphs", [False, True], ids=["eager", "hip-graph"])
def test_two_rank_step_keeps_replicas_identical(dev, use_graphs):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, use_graphs, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=600) for _ in range(2))
    [p.join(timeout=120) for p in procs]
    (r0, g0, d0, a0, l0, it0, s0), (r1, g1, d1, a1, l1, it1, s1) = res
    assert g0 == g1 and d0 == d1 and a0 == a1, "replicas diverged: gradient exchange is not a plain SUM over identical states"
    assert l0 == l1, "loss scalars must be reduced over ranks"
    n = 5 if use_graphs else 2
    assert it0 == it1 == n and s0 == s1 == n
    assert all(abs(v) < 1e6 for v in l0)


def _cfg(full_width, world):
    from textboxgan_amd.config import Config, small_config
    return Config(batch_size_per_gpu=4, num_replicas=world) if full_width else small_config(4, num_replicas=world)


def _grad_worker(rank, world, port, backend, q, full_width=False):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle import ref_model as M
    from textboxgan_amd.config import small_config
    from textboxgan_amd.training_step import build_trainer_state
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = _cfg(full_width, world)
    init = M.make_state(cfg, seed=0, bench_init=True)
    st = build_trainer_state(cfg, dev, seed=0)
    st["generator"].load_state_dict({k: v.clone() for k, v in init["G"].items()})
    st["discriminator"].load_state_dict({k: v.clone() for k, v in init["D"].items()})
    ts = st["training_step"]
    assert ts.distributed and ts.d_cuts, "the bucketed D exchange must be active in a data-parallel run"
    b = {k: v.to(dev) for k, v in M.make_batch(cfg, seed=1234, rank=rank).items()}
    rand = M.make_rand(cfg, seed=99 + rank, with_pl=False)
    rand = {k: ([t.to(dev) for t in v] if isinstance(v, list) else (v.to(dev) if torch.is_tensor(v) else v)) for k, v in rand.items()}
    losses = ts.dist_train_step(b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], False, False, 1e-4, rand=rand)
    torch.cuda.synchronize()
    cat = lambda views: torch.cat([v.reshape(-1) for v in views]).cpu().numpy()  # numpy: pickled by value (no shm fd)
    fracs = [float(b1 - b0) / ts.d_grad.numel() for _, _, (b0, b1) in ts.d_stages]  # deepest stage first
    # post-Adam replica state (exact checksums: the replicas must stay BIT-identical) and the loss normalisation in use
    post = (float(st["generator"]._flat.flat.double().sum()), float(st["generator"]._flat.flat.double().abs().sum()),
            float(st["discriminator"]._flat.flat.double().sum()), float(st["discriminator"]._flat.flat.double().abs().sum()))
    q.put((rank, cat(ts.g_views), cat(ts.o_views), cat(ts.d_views),
           [float(x) for x in losses[0]] + [float(x) for x in losses[1]] + [float(losses[2])], fracs, post, ts.batch_size))
    dist.barrier()
    dist.destroy_process_group()


def _exchange_vs_oracle(backend, full_width=False, world=2):
    import torch.multiprocessing as mp
    from oracle import ref_model as M
    from conftest import ocr_oracle
    from textboxgan_amd.config import small_config
    cfg = _cfg(full_width, world)
    assert cfg.batch_size == 4 * world  # config/config.py:140-141: the losses are divided by the GLOBAL batch
    # (eight replicas: the oracle's recogniser in fp32 -- 3 s instead of 9 s per replica on 8 cores, 1e-7 against bars of 2e-3 / 1e-2;
    # the two-replica tests keep the float64 recogniser)
    ocr = ocr_oracle(cfg.max_char_number, dtype=torch.float32 if world > 2 else None)
    sums, loss_sum = None, None
    for rank in range(world):  # the oracle, replica by replica, from the same initial weights
        st = M.make_state(cfg, seed=0, bench_init=True)
        batch, rand = M.make_batch(cfg, seed=1234, rank=rank), M.make_rand(cfg, seed=99 + rank, with_pl=False)
        losses, grads = M.training_step(st, cfg, batch["real_images"], batch["ocr_images"], batch["input_words"],
                                        batch["ocr_labels"], False, False, 1e-4, rand, ocr.serve, return_grads=True)
        flat = [float(x) for x in losses[0]] + [float(x) for x in losses[1]] + [float(losses[2])]
        loss_sum = flat if loss_sum is None else [a + b for a, b in zip(loss_sum, flat)]
        if sums is None:
            sums = {k: {n: g.clone() for n, g in grads[k].items()} for k in ("g", "ocr", "d")}
        else:
            for k in ("g", "ocr", "d"):
                for n, g in grads[k].items():
                    sums[k][n] += g
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, backend, q, full_width)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    [p.join(timeout=120) for p in procs]
    l2 = lambda a, r: float((a.double() - r.double()).norm() / (r.double().norm() + 1e-30))
    res_fracs = [t[5] for t in res]
    assert all(t[6] == res[0][6] for t in res), "replicas are not bit-identical after the Adam updates"
    assert all(t[7] == cfg.batch_size for t in res), "loss normalisation is not the global batch"
    assert all(t[5] == res_fracs[0] for t in res), "ranks disagree on the D bucket table"
    res = [(r, torch.from_numpy(g), torch.from_numpy(o), torch.from_numpy(d), l) for r, g, o, d, l, _, _, _ in res]
    (_, g0, o0, d0, l0) = res[0]
    for (_, g1, o1, d1, l1) in res[1:]:
        assert torch.equal(g0, g1) and torch.equal(o0, o1) and torch.equal(d0, d1), "ranks hold different exchanged gradients"
        assert l0 == l1, "loss scalars must be reduced over ranks"
    # the oracle's gradient dicts, concatenated in the order of the product's flat buffers
    from textboxgan_amd.training_step import build_trainer_state
    names = build_trainer_state(cfg, torch.device("cpu"), seed=0)
    gnames = [n for n in names["generator"]._flat.names if n.startswith(("latent_encoder.", "synthesis."))]
    onames = [n for n in names["generator"]._flat.names if n.startswith(("synthesis.", "word_encoder."))]
    dnames = names["discriminator"]._flat.names
    cat = lambda d, order: torch.cat([d[n].reshape(-1) for n in order])
    assert g0.numel() == cat(sums["g"], gnames).numel() and d0.numel() == cat(sums["d"], dnames).numel()
    assert l2(g0, cat(sums["g"], gnames)) < 2e-3, "all-reduced G gradient != sum of the per-replica oracle gradients"
    assert l2(o0, cat(sums["ocr"], onames)) < 1e-2
    assert l2(d0, cat(sums["d"], dnames)) < 2e-3
    if full_width:  # the buckets of the staged D exchange are where DESIGN section 5 says: 74 % / 20 % / 6 % of the bytes
        fr = res_fracs[0]
        assert len(fr) == 3 and abs(fr[0] - 0.74) < 0.03 and abs(fr[1] - 0.20) < 0.03 and abs(fr[2] - 0.06) < 0.03, fr
    for a, e in zip(l0, loss_sum):  # strategy.reduce(SUM) of the loss scalars
        assert abs(a - e) <= 2e-4 * max(1.0, abs(e)), (l0, loss_sum)


@pytest.mark.parametrize("full_width", [False, True], ids=["small", "full-width"])
def test_exchanged_gradients_equal_sum_of_per_replica_oracle_gradients_gloo(dev, full_width):
    """full-width: the real channel widths (B = 4 per rank), so the D exchange's bucket boundaries (d_cuts = (5, 3): 74 % /
    20 % / 6 % of D's 62 MB) are exercised on the real layer sizes."""
    _exchange_vs_oracle("gloo", full_width)


def test_eight_rank_exchange_matches_sum_of_eight_oracle_replicas_gloo(dev):
    """BASELINE configs[3] has EIGHT replicas (global batch 8 x per-GPU batch): eight processes on one GPU over gloo run one
    step each on its own shard with its own randomness.  Asserted: the exchanged flat buffers equal the SUM of the eight
    per-replica CPU-oracle gradient sets (G, OCR, D), the seven loss scalars SUM to the oracle's (every replica divides by
    Bg = 8 * B, config/config.py:140-141), all eight replicas are bit-identical after the three Adam updates, and every rank
    uses the same deepest-first D bucket table (training_step.py:91-136,233-235 of the reference)."""
    _exchange_vs_oracle("gloo", full_width=False, world=8)


def test_exchanged_gradients_equal_sum_of_per_replica_oracle_gradients_rccl(dev):
    """the same over RCCL (backend "nccl" IS RCCL on ROCm), one process per GPU -- needs two GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the driver's multi-GPU node); the gloo variant covers the logic on one GPU")
    _exchange_vs_oracle("nccl")


def test_bench_launcher_two_ranks_gloo_smoke(dev):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` as the driver launches it, with both ranks
    on one GPU and gloo collectives (TBG_BENCH_SINGLE_DEVICE / TBG_DIST_BACKEND): one JSON line, whole-job throughput."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TBG_DIST_BACKEND="gloo", TBG_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "4", "--no-roofline", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 8 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["per_gpu_batch"] == 4 and "x 2 ranks" in rec["config"]["baseline_config"]  # (names what the line measures)


def _forced_exchange_worker(port, q, real_ocr=False):
    """one process, backend nccl (= RCCL), world size 1, TBG_FORCE_EXCHANGE=1: the split-graph step with its collectives
    against the plain single-graph step from the same seed."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TBG_FORCE_EXCHANGE="1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from oracle import ref_model as M
    from textboxgan_amd.config import small_config
    from textboxgan_amd.training_step import build_trainer_state
    cfg = small_config(4, num_replicas=1)
    b = {k: v.to(dev) for k, v in M.make_batch(cfg, seed=1234, rank=0).items()}
    # caller protocol of train.py:178-208 over one lazy-regularisation cycle boundary: plain, PL, PL+R1 variants all replay
    sched = [(False, False)] * 3 + [(False, True)] * 3 + [(True, True)] * 3
    if real_ocr:  # two steps of each variant (the second one replays the captured graphs): a last-bit difference has few sign-like
        sched = [(False, False)] * 2 + [(False, True)] * 2 + [(True, True)] * 2  # Adam steps to grow in

    from bench import _TinyOCR
    from textboxgan_amd.aster import AsterInferer

    def run(forced):
        os.environ["TBG_FORCE_EXCHANGE"] = "1" if forced else "0"
        # (a recogniser made of order-independent operations: see the test's docstring)
        kw = {} if real_ocr else dict(aster_ocr=AsterInferer(model=_TinyOCR(cfg.max_char_number)))
        st = build_trainer_state(cfg, dev, seed=0, use_graphs=True, **kw)
        ts = st["training_step"]
        assert ts.distributed == forced and bool(ts.d_cuts) == forced
        torch.manual_seed(77)
        losses = []
        for do_r1, do_pl in sched:
            l = ts.dist_train_step(b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], do_r1, do_pl, 1e-4)
            losses.append([float(x) for x in l[0]] + [float(x) for x in l[1]] + [float(l[2])])
        torch.cuda.synchronize()
        return (ts.graph_mode, ts.capture_error, losses, st["generator"]._flat.flat.clone(), st["discriminator"]._flat.flat.clone(),
                float(st["pl_mean"]))

    mode_f, err_f, loss_f, g_f, d_f, pl_f = run(True)
    mode_p, err_p, loss_p, g_p, d_p, pl_p = run(False)
    q.put(dict(mode_forced=mode_f, err_forced=err_f, mode_plain=mode_p, err_plain=err_p, losses_equal=loss_f == loss_p,
               loss_forced=loss_f[-1], loss_plain=loss_p[-1], g_equal=bool(torch.equal(g_f, g_p)), d_equal=bool(torch.equal(d_f, d_p)),
               g_maxdiff=float((g_f - g_p).abs().max()), d_maxdiff=float((d_f - d_p).abs().max()), pl_equal=pl_f == pl_p,
               losses_forced=loss_f, losses_plain=loss_p, pl=(pl_f, pl_p),
               g_moved=float((g_f != g_p).float().mean()), d_moved=float((d_f != d_p).float().mean())))
    dist.destroy_process_group()


def test_forced_exchange_rccl_split_graphs_equal_the_plain_step(dev):
    """TBG_FORCE_EXCHANGE=1 with backend "nccl" at world size 1: GradExchange is active, so the step is captured as SPLIT HIP
    graphs sharing one pool -- [forward + g-pass] [ocr-pass] [D stage 0..2] [3 x Adam] -- and ncclAllReduce (RCCL) is issued
    between their replays for every gradient slice (the bucketed D exchange included), for the plain, PL and PL+R1 variants:
    the call pattern of the first multi-GPU run (reference training_step.py:91-136,233-235, config/config.py:140-141) on the
    one GPU this suite has.  A 1-rank SUM is the identity, so everything the step touches -- seven losses per step, pl_mean, and
    every generator / discriminator weight after nine steps -- must be BIT-identical to the non-distributed single-graph step
    from the same seed; the capture must not have been lost (graph_mode "split", no capture_error).
    The frozen recogniser of this test is bench.py's linear stand-in: the ASTER-shaped network's rectifier differentiates
    torch's grid_sample, whose image gradient is accumulated with atomicAdd (several samples per pixel) -- measured here
    (tools/scratch experiments of round 5, profiles/r05_ab_one_box.txt): every other stage of the step repeats bit for bit,
    that one differs in the last bit on some runs, and nine Adam steps (beta1 = 0: sign-like updates) amplify it.  The bit-
    identity asserted here is a property of the graph / exchange mechanics, so it is tested on order-independent arithmetic."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_exchange_worker, args=(_free_port(), q))
    p.start()
    r = q.get(timeout=900)
    p.join(timeout=120)
    assert r["mode_forced"] == "split" and r["err_forced"] is None, r
    assert r["mode_plain"] == "single" and r["err_plain"] is None, r
    assert r["losses_equal"] and r["pl_equal"], r
    assert r["g_equal"] and r["d_equal"], r


def test_forced_exchange_rccl_with_the_real_recogniser_graph(dev):
    """the same split-graph / RCCL mechanics with the ASTER-shaped recogniser in the step (its rectifier, trunk on the small-map
    kernels, fused BiLSTM and decoder steps all captured in the [ocr-pass] graph): two steps of each lazy-regularisation variant.
    Not bit-identity -- grid_sample's backward accumulates with atomicAdd (see the test above): the capture must be intact (split
    graphs, no capture error), the first step's seven losses agree to 1e-6 relative, and the six-step trajectories stay together at
    the level the sign-like Adam updates (beta1 = 0) allow."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_exchange_worker, args=(_free_port(), q, True))
    p.start()
    r = q.get(timeout=900)
    p.join(timeout=120)
    assert r["mode_forced"] == "split" and r["err_forced"] is None, r
    assert r["mode_plain"] == "single" and r["err_plain"] is None, r
    for step, (lf, lp) in enumerate(zip(r["losses_forced"], r["losses_plain"])):
        # the first step sees identical weights: 1e-6; afterwards the two runs' weights differ wherever a last-bit gradient difference
        # flipped a sign-like Adam update, and the losses follow at the 1e-4 level (measured 2.6e-4 on the path-length term)
        tol = 1e-6 if step == 0 else 5e-2  # (observed up to 6e-3 on the path-length term of this reduced-channel model)
        for a, b in zip(lf, lp):
            assert abs(a - b) <= tol * max(1.0, abs(b)), (step, lf, lp)
    assert abs(r["pl"][0] - r["pl"][1]) <= 5e-2 * max(1.0, abs(r["pl"][1])), r["pl"]
    assert r["g_maxdiff"] < 1e-2 and r["d_maxdiff"] < 1e-2, (r["g_maxdiff"], r["d_maxdiff"])  # (lr 2e-3: a handful of flipped updates)
