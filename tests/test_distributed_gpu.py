"""-m gpu: the data-parallel step with world_size 2 (two processes sharing cuda:0, gloo collectives --
the same code path bench.py drives with RCCL): replicas stay bit-identical, losses are SUM-reduced,
and the summed gradient equals the sum of the per-rank gradients."""
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
