"""One optimisation step of TextBoxGAN on MI355X -- drop-in for the reference's ``TrainingStep``
(training_step.py:14-402) with the same constructor / ``dist_train_step`` signature and return
structure, plus ``build_trainer_state`` (the wiring of train.py:25-108).

Per step (training_step.py:138-222): G forward, mask, D(fake), D(real), OCR; three losses;
three gradient sets taken at the PRE-update weights; three Adam updates in the reference's
order (g, ocr, d).  Lazy regularisers: path length every ``g_reg_interval`` steps on B//2
samples, R1 every ``d_reg_interval`` (both need second-order gradients -> composable mode).

Data parallel = one process per GPU (torch.distributed / RCCL): losses are pre-divided by the
GLOBAL batch, so the exchange is a plain SUM all-reduce of the three flat gradient buffers,
issued asynchronously as soon as each backward pass has produced its buffer so it overlaps
the following backward pass (the reference hides this inside MirroredStrategy's
apply_gradients, training_step.py:233-235).  pl_mean / w_avg stay rank-local
(ONLY_FIRST_REPLICA, train.py:40-46).
"""
from __future__ import annotations

import math
from typing import Optional

import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import ops
from .config import Config, cfg as default_cfg
from .dist_utils import GradExchange
from .models import Discriminator, Generator, mask_text_box
from .optim import AdamTF, GradViews, flatten_generator, flatten_module, write_grads


def generator_loss(fake_scores, batch_size):
    """models/losses/gan_losses.py:8-10."""
    return F.softplus(-fake_scores).sum() / batch_size


def discriminator_loss(fake_scores, real_scores, batch_size):
    """models/losses/gan_losses.py:13-16."""
    return (F.softplus(fake_scores) + F.softplus(-real_scores)).sum() / batch_size


def softmax_cross_entropy_loss(logits, labels, batch_size):
    """models/losses/ocr_losses.py:8-11."""
    ce = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1).long(), reduction="none")
    return ce.sum() / batch_size


def mean_squared_loss(y_a, y_b, batch_size):
    """models/losses/ocr_losses.py:14-20."""
    return (y_a - y_b).square().mean(dim=-1).sum() / batch_size


class TrainingStep:
    def __init__(self, generator: Generator, discriminator: Discriminator, aster_ocr, g_optimizer: AdamTF,
                 ocr_optimizer: AdamTF, d_optimizer: AdamTF, g_reg_interval: int, d_reg_interval: int,
                 pl_mean: torch.Tensor, cfg: Config = default_cfg, process_group=None, use_graphs: bool = False,
                 compute_dtype: Optional[str] = "f32"):
        self.generator, self.discriminator, self.aster_ocr = generator, discriminator, aster_ocr
        self.g_optimizer, self.ocr_optimizer, self.d_optimizer = g_optimizer, ocr_optimizer, d_optimizer
        self.g_reg_interval, self.d_reg_interval = g_reg_interval, d_reg_interval
        self.cfg = cfg
        self.batch_size = cfg.batch_size
        self.batch_size_per_gpu = cfg.batch_size_per_gpu
        self.pl_mean = pl_mean
        self.pl_minibatch_shrink = 2 if self.batch_size_per_gpu // 2 >= 1 else self.batch_size_per_gpu
        self.pl_decay = 0.01
        self.r1_gamma = 10.0
        self.pl_noise_scaler = 1.0 / math.sqrt(float(cfg.image_width) * float(cfg.char_height))
        self.pg = process_group
        self.use_graphs = use_graphs
        # "f32": exact fp32 MFMA contractions.  "f32x3": fp32 tensors, forward / data-gradient contractions as three bf16 terms
        # per operand on the bf16 pipe (fp32-grade, tbg.h "f32x3 forms").  "bf16": conv / filter-gradient operands rounded to
        # bf16 on their way into LDS, fp32 accumulate, fp32 master weights + Adam (BASELINE configs[2]).  None: the
        # ops.compute_dtype scope in force when the step object is built.
        if compute_dtype is None:
            compute_dtype = ops.compute_mode()
        assert compute_dtype in ops.COMPUTE_MODES
        self.compute_dtype = compute_dtype
        # (Round 2 ran the frozen-OCR branch on a second HIP stream; profiles/r02_stream_concurrency.txt measured that
        # the fork hides nothing on this stack -- 29.11 vs 29.20 ms/step -- so the branch is issued in line again.)
        self.capture_error = None  # text of a failed HIP-graph capture (the step then runs eagerly): surfaced by bench.py
        self._graphs = {}
        self._packs = ops.PackedStore()
        self._warmed = set()
        self._static = None
        self.exchange = GradExchange(process_group)
        self.distributed = self.exchange.active

        gf, df = generator._flat, discriminator._flat
        self.g_range = gf.range_of(("latent_encoder.", "synthesis."))
        self.o_range = gf.range_of(("synthesis.", "word_encoder."))
        self.g_params = gf.select(("latent_encoder.", "synthesis."))
        self.o_params = gf.select(("synthesis.", "word_encoder."))
        self.d_params = list(df.params)
        self.g_grad, self.g_views = gf.make_grad_buffer(*self.g_range)
        self.o_grad, self.o_views = gf.make_grad_buffer(*self.o_range)
        self.d_grad, self.d_views = df.make_grad_buffer(0, df.total)
        # D's gradient exchange in BUCKETS, deepest layers first (the reference hides one 62 MB all-reduce
        # inside apply_gradients, training_step.py:233-235).  D's backward reaches the deep layers first: at the default
        # cuts (5, 3) the blocks[5:] + head hold 74% of D's gradient bytes after ~8% of its backward FLOPs and blocks[3:5]
        # another 20% after ~20%, so their all-reduces run under the backward of the high-resolution blocks, and only the
        # last 6% of the bytes is exchanged after the pass.  Stage boundaries = activations entering blocks[cut].
        nb = len(discriminator.blocks)
        self.d_cuts = tuple(c for c in (5, 3) if 0 < c < nb) if self.distributed else ()
        self.d_stages = self._d_stage_table(df)

    def _d_stage_table(self, df):
        """[(parameter list, views, flat slice)] per backward stage, deepest first; the slices tile d_grad."""
        names = df.names
        def first_index(prefix):
            return next(i for i, n in enumerate(names) if n.startswith(prefix))
        bounds = sorted([first_index(f"blocks.{c}.") for c in self.d_cuts]) + [len(names)]
        stages, hi = [], len(names)
        for lo in reversed([0] + bounds[:-1]):
            b0 = df.offsets[lo]
            b1 = df.offsets[hi] if hi < len(names) else df.total
            stages.append((self.d_params[lo:hi], GradViews(self.d_views[lo:hi]), (b0, b1)))
            hi = lo
        # the staged backward relies on the flat layout: stage slices are contiguous, deepest first, and tile d_grad
        assert self.d_params == list(df.params) and len(self.d_views) == len(self.d_params)
        assert stages[0][2][1] == df.total and stages[-1][2][0] == 0, "stage slices must cover the flat gradient buffer"
        assert all(a[2][0] == b[2][1] for a, b in zip(stages, stages[1:])), "stage slices must be contiguous"
        assert sum(len(st[0]) for st in stages) == len(self.d_params)
        for c in self.d_cuts:  # everything from blocks[c] on (deeper blocks, last_block, last_dense, last_bias) is one suffix
            i0 = first_index(f"blocks.{c}.")
            assert not any(n.startswith(("initial_fromrgb.",) + tuple(f"blocks.{k}." for k in range(c))) for n in names[i0:]), \
                "discriminator parameters are not ordered shallow -> deep"
        return stages

    # ------------------------------------------------------------------------------------
    def dist_train_step(self, real_images, ocr_images, input_words, ocr_labels, do_r1_reg: bool, do_pl_reg: bool,
                        ocr_loss_weight: float, rand: Optional[dict] = None):
        """training_step.py:57-136.  Inputs are this rank's shard of the global batch: exactly ``cfg.batch_size_per_gpu``
        samples -- the reference's loaders use drop_remainder=True (training_data_loader.py:93-97), z / noise / loss
        normalisation are sized by the configured batch, and a HIP graph is bound to its geometry, so a short batch is an
        error here, not a silent broadcast."""
        nb = self.batch_size_per_gpu
        if real_images.shape[0] != nb or input_words.shape[0] != nb or ocr_labels.shape[0] != nb:
            raise ValueError(f"dist_train_step expects {nb} samples per replica (drop_remainder=True semantics), got "
                             f"real_images {tuple(real_images.shape)}, input_words {tuple(input_words.shape)}, "
                             f"ocr_labels {tuple(ocr_labels.shape)}")
        if self.use_graphs and rand is None:
            gen_losses, disc_losses, ocr_loss = self._graphed_step(real_images, ocr_images, input_words, ocr_labels,
                                                                   bool(do_r1_reg), bool(do_pl_reg), ocr_loss_weight)
        else:
            gen_losses, disc_losses, ocr_loss = self._train_step(real_images, ocr_images, input_words, ocr_labels,
                                                                 do_r1_reg, do_pl_reg, ocr_loss_weight, rand)
        self._count_step()
        if self.distributed:  # strategy.reduce(SUM) of the 7 scalars, one collective
            r = self.exchange.reduce_scalars([*gen_losses, *disc_losses, ocr_loss])
            gen_losses, disc_losses, ocr_loss = tuple(r[0:3]), tuple(r[3:6]), r[6]
        return gen_losses, disc_losses, ocr_loss

    @property
    def graph_mode(self) -> str:
        """how the step actually ran: "single" (one HIP graph per step variant), "split" (per-gradient-set graphs with the
        all-reduces between them), "two-phase" (gradient graph | blocking exchange | update graph) or "eager"."""
        if not self.use_graphs or not self._graphs:
            return "eager"
        n = max(len(g[0]) for g in self._graphs.values())
        return "single" if n == 1 else "two-phase" if n == 2 else "split"

    def _count_step(self):
        for o in (self.g_optimizer, self.ocr_optimizer, self.d_optimizer):
            o._iterations += 1

    # ------------------------------------------------------------------------------------
    # HIP-graph replay of the whole step (one graph per lazy-regularisation variant)
    def _graphed_step(self, real_images, ocr_images, input_words, ocr_labels, do_r1, do_pl, ocr_w):
        if self._static is None:
            self._static = dict(real=real_images.clone(), ocr_img=ocr_images.clone(), words=input_words.clone(),
                                labels=ocr_labels.clone(), w=torch.zeros((), device=real_images.device))
        st = self._static
        st["real"].copy_(real_images); st["ocr_img"].copy_(ocr_images); st["words"].copy_(input_words)
        st["labels"].copy_(ocr_labels); st["w"].fill_(ocr_w)
        key = (do_r1, do_pl)
        if key not in self._warmed:
            # first use of a variant: a real eager step on a side stream (doubles as the capture warm-up)
            self._warmed.add(key)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                out = self._run_phases(st, do_r1, do_pl)
            torch.cuda.current_stream().wait_stream(side)
            return out
        if key not in self._graphs:
            torch.cuda.synchronize()
            if self.distributed and os.environ.get("TBG_GRAPH_SPLIT", "1") != "0":
                try:
                    self._graphs[key] = self._capture_split(st, do_r1, do_pl)
                except Exception as e:  # a failed capture must not take a multi-GPU job down: finish it eagerly
                    import sys
                    self.capture_error = f"split capture {key}: {type(e).__name__}: {e}"
                    print(f"[tbg] HIP-graph capture failed ({type(e).__name__}: {e}); continuing WITHOUT graphs",
                          file=sys.stderr, flush=True)
                    torch.cuda.synchronize()
                    self.use_graphs = False
                    return self._train_step(st["real"], st["ocr_img"], st["words"], st["labels"], do_r1, do_pl, st["w"], None)
            elif self.distributed:  # conservative fallback: one gradient graph, three blocking all-reduces, one update graph
                ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                    outs = self._compute_grads(st["real"], st["ocr_img"], st["words"], st["labels"], do_r1, do_pl,
                                               st["w"], {})
                with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
                    self._apply_updates()
                self._graphs[key] = ([ga, gb], outs)
            else:
                g = torch.cuda.CUDAGraph()
                # a process group may exist although this step does not exchange (world size 1): RCCL's watchdog thread polls its
                # events while we capture -- under the default "global" mode that call fails, kills the watchdog (and with it the
                # process) and invalidates the capture (seen once in tests/test_distributed_gpu.py: timing dependent)
                import torch.distributed as _dist
                mode = "thread_local" if (_dist.is_available() and _dist.is_initialized()) else "global"
                try:
                    with torch.cuda.graph(g, capture_error_mode=mode):
                        outs = self._compute_grads(st["real"], st["ocr_img"], st["words"], st["labels"], do_r1, do_pl,
                                                   st["w"], {})
                        self._apply_updates()
                except Exception as e:
                    import sys
                    self.capture_error = f"capture {key}: {type(e).__name__}: {e}"
                    print(f"[tbg] HIP-graph capture failed ({type(e).__name__}: {e}); continuing WITHOUT graphs",
                          file=sys.stderr, flush=True)
                    torch.cuda.synchronize()
                    self.use_graphs = False
                    return self._train_step(st["real"], st["ocr_img"], st["words"], st["labels"], do_r1, do_pl, st["w"], None)
                self._graphs[key] = ([g], outs)
        graphs, outs = self._graphs[key]
        # the graph's output tensors are static (overwritten by the next replay): hand a COPY to the caller -- the 7 scalars as one
        # stacked tensor (one launch, not seven clones)
        def fresh(o):
            v = torch.stack([*o[0], *o[1], o[2]]).unbind(0)
            return tuple(v[0:3]), tuple(v[3:6]), v[6]
        if len(graphs) == 1:
            graphs[0].replay()
            return fresh(outs)
        if len(graphs) == 2:
            graphs[0].replay()
            self.exchange.reduce_now((self.g_grad, self.o_grad, self.d_grad))
            graphs[1].replay()
            return fresh(outs)
        # data-parallel: [fwd + g-pass] -> all-reduce(g) || [ocr-pass] -> all-reduce(ocr) || [d stage 0 (deepest)] ->
        # all-reduce(bucket 0) || [d stage 1] -> all-reduce(bucket 1) || ... -> [Adam x3]
        handles = []
        for g, buf in zip(graphs[:-1], self._exchange_buffers(do_r1)):
            g.replay()
            handles.append(self.exchange.start(buf))  # ordered after the graph on this stream, runs on the RCCL stream
        for h in handles:
            GradExchange.finish(h)
        graphs[-1].replay()
        return fresh(outs)

    def _exchange_buffers(self, do_r1):
        """the flat gradient slices exchanged after each gradient graph, in order."""
        if self.d_cuts and not do_r1:
            return [self.g_grad, self.o_grad] + [self.d_grad[b0:b1] for _, _, (b0, b1) in self.d_stages]
        return [self.g_grad, self.o_grad, self.d_grad]

    def _capture_split(self, st, do_r1, do_pl):
        """Capture the step as SEVERAL HIP graphs sharing one memory pool (the autograd state of the forward lives across
        them, as in torch's make_graphed_callables): [forward + g-pass] [ocr-pass] [d-pass stage 0 .. n-1] [3x Adam].  The
        gradient all-reduces (training_step.py:233-235) are issued between the graphs and overlap the next backward graph
        instead of sitting in front of the Adam updates; D's is bucketed by backward stage, deepest layers first."""
        import gc
        n_grad = len(self._exchange_buffers(do_r1))
        graphs = [torch.cuda.CUDAGraph() for _ in range(n_grad + 1)]
        pool = torch.cuda.graph_pool_handle()
        cap = torch.cuda.Stream()
        torch.cuda.synchronize()
        gc.collect()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            idx = [0]
            # thread_local: the RCCL watchdog thread polls events while we capture; its calls must not abort the capture
            graphs[0].capture_begin(pool=pool, capture_error_mode="thread_local")

            def boundary():
                graphs[idx[0]].capture_end()
                idx[0] += 1
                graphs[idx[0]].capture_begin(pool=pool, capture_error_mode="thread_local")

            try:
                outs = self._compute_grads(st["real"], st["ocr_img"], st["words"], st["labels"], do_r1, do_pl, st["w"], {},
                                           None, boundary)
                assert idx[0] == n_grad - 1, "one capture boundary per exchanged gradient slice"
                graphs[idx[0]].capture_end()
                idx[0] = n_grad
                graphs[n_grad].capture_begin(pool=pool, capture_error_mode="thread_local")
                self._apply_updates()
                graphs[n_grad].capture_end()
            except Exception:
                try:
                    graphs[idx[0]].capture_end()  # leave capture mode before propagating
                except Exception:
                    pass
                raise
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        return (graphs, outs)

    def prepare_graphs(self, real_images, ocr_images, input_words, ocr_labels, ocr_loss_weight: float = 1e-4):
        """Warm up and capture the three step variants (plain, +PL, +PL+R1) ahead of time.  Every call
        below is a REAL optimisation step (2 per variant: eager warm-up, then capture + first replay)."""
        assert self.use_graphs
        for do_r1, do_pl in ((False, False), (False, True), (True, True)):
            for _ in range(2):
                self.dist_train_step(real_images, ocr_images, input_words, ocr_labels, do_r1, do_pl, ocr_loss_weight)

    def _run_phases(self, st, do_r1, do_pl):
        outs = self._compute_grads(st["real"], st["ocr_img"], st["words"], st["labels"], do_r1, do_pl, st["w"], {})
        self.exchange.reduce_now((self.g_grad, self.o_grad, self.d_grad))
        self._apply_updates()
        return outs

    def _all_reduce_async(self, buf):
        return self.exchange.start(buf)

    def _train_step(self, real_images, ocr_images, input_words, ocr_labels, do_r1_reg, do_pl_reg, ocr_loss_weight,
                    rand):
        """eager path: the three all-reduces are issued asynchronously right after their backward pass."""
        handles = []
        outs = self._compute_grads(real_images, ocr_images, input_words, ocr_labels, do_r1_reg, do_pl_reg,
                                   ocr_loss_weight, rand or {}, handles)
        self._apply_updates(handles)
        return outs

    def _compute_grads(self, *args, **kw):
        # packed filters are shared by the forward and the three backward passes
        with ops.STATE_LOCK, ops.filter_cache(), ops.compute_dtype(self.compute_dtype), self._packs.scope():
            self._packs.refresh()  # ONE launch re-packs every filter from the weights the last update left
            return self._compute_grads_impl(*args, **kw)

    def _compute_grads_impl(self, real_images, ocr_images, input_words, ocr_labels, do_r1_reg, do_pl_reg,
                            ocr_loss_weight, rand, handles=None, boundary=None):
        """``handles``: eager mode -- the async all-reduce of each gradient set is appended right after its pass.
        ``boundary``: split-capture mode -- called after the g-pass and after the ocr-pass (the capture of one HIP graph
        ends and the next begins there, so that the replay can overlap each all-reduce with the following graph)."""
        cfg, G, D = self.cfg, self.generator, self.discriminator
        dev = real_images.device
        zero = torch.zeros((), device=dev)
        z = rand["z"] if "z" in rand else torch.randn(self.batch_size_per_gpu, cfg.z_dim, device=dev)

        # generator forward; mask_text_box (training_step.py:160-163, utils/utils.py:11-45) is the last toRGB's epilogue
        fake_images = G((input_words, z), training=True, rand=rand, mask_words=input_words)

        # frozen OCR branch and its own backward, issued in line (training_step.py:375-402): only d(ocr_loss)/d(fake_images)
        # is kept for the generator's ocr-pass below
        # The recogniser always runs at fp32 grade (north_star: OCR loss within 1e-4): measured on the full-width step, its
        # image gradient amplifies operand rounding by ~1e4 (exact-fp32 kernels: 1.2e-3 relative L2 on the OCR gradient
        # set), so bf16 operands (2^-9) leave nothing of it (relative L2 1.09 against the fp32 oracle).  In bf16 mode its
        # convolutions therefore take the f32x3 kernels (the frozen filters are packed per arithmetic by their owner).
        with ops.compute_dtype("f32x3" if self.compute_dtype == "bf16" else self.compute_dtype):
            ocr_loss = self._get_ocr_loss(fake_images, ocr_labels, ocr_images)
            ocr_loss_w = ocr_loss_weight * ocr_loss
            (dfake_ocr,) = torch.autograd.grad(ocr_loss_w, fake_images, retain_graph=False)

        staged = bool(self.d_cuts) and not do_r1_reg  # R1 steps (1 in 16) keep the single exchange: their D graph is second order
        cuts = sorted(self.d_cuts) if staged else None
        # Plain steps: ONE discriminator pass over [fake; real] (discriminator.py is per-sample except the minibatch
        # statistics, which stay inside each half: parts=2) -- the same arithmetic as the reference's two calls
        # (training_step.py:165-173) in half the launches and with twice the pixels per tile on the small maps.  The
        # G-loss pass differentiates the fake half only (ops.FLAGS.d_first_half).  R1 steps keep two calls: the real
        # half needs the second-order graph.
        joint = not do_r1_reg
        nb = fake_images.shape[0]
        if joint:
            both = torch.cat([fake_images, real_images], dim=0)
            if staged:
                scores, d_taps = D(both, cuts=cuts, parts=2)
            else:
                scores = D(both, parts=2)
            fake_scores, real_scores = scores[:nb], scores[nb:]
        else:
            fake_scores = D(fake_images)
        g_loss = generator_loss(fake_scores, self.batch_size)
        pl_penalty = self._path_length_reg(input_words, rand) if do_pl_reg else zero
        reg_g_loss = g_loss + pl_penalty

        if do_r1_reg:
            real_scores, r1_penalty = self._r1_reg(real_images)
        else:
            r1_penalty = zero
        d_loss = discriminator_loss(fake_scores, real_scores, self.batch_size)
        reg_d_loss = d_loss + r1_penalty

        # --- three backward passes at the pre-update weights (training_step.py:194-213)
        ops.FLAGS.skip_d_wgrad = True
        ops.FLAGS.d_first_half = nb if joint else 0
        try:
            grads = torch.autograd.grad(reg_g_loss, self.g_params, retain_graph=True, allow_unused=True)
        finally:
            ops.FLAGS.skip_d_wgrad = False
            ops.FLAGS.d_first_half = 0
        write_grads(self.g_views, grads)
        if handles is not None:
            handles.append(self._all_reduce_async(self.g_grad))
        if boundary is not None:
            boundary()

        grads = torch.autograd.grad(fake_images, self.o_params, grad_outputs=dfake_ocr, retain_graph=True,
                                    allow_unused=True)
        write_grads(self.o_views, grads)
        if handles is not None:
            handles.append(self._all_reduce_async(self.o_grad))
        if boundary is not None:
            boundary()

        ops.FLAGS.skip_image_grad = True
        try:
            if not staged:
                grads = torch.autograd.grad(reg_d_loss, self.d_params, allow_unused=True)
                write_grads(self.d_views, grads)
                if handles is not None:
                    handles.append(self._all_reduce_async(self.d_grad))
            else:
                # staged backward, deepest layers first: stage i yields its parameters' gradients plus the gradients of the
                # activations entering it (the "taps"), from which stage i+1 continues; each bucket's all-reduce is
                # issued as soon as the stage has written it (eager: async handles; HIP graphs: a capture boundary)
                outs, gouts = [reg_d_loss], None
                for si, (params, views, (b0, b1)) in enumerate(self.d_stages):
                    last = si == len(self.d_stages) - 1
                    k = len(cuts) - 1 - si  # taps feeding this stage sit at cuts[k] (none for the last stage)
                    taps = [] if last else [d_taps[k]]
                    grads = torch.autograd.grad(outs, list(params) + taps, grad_outputs=gouts, retain_graph=not last,
                                                allow_unused=True)
                    write_grads(views, grads[:len(params)])
                    if handles is not None:
                        handles.append(self._all_reduce_async(self.d_grad[b0:b1]))
                    if not last:
                        outs, gouts = taps, list(grads[len(params):])
                        if boundary is not None:
                            boundary()
        finally:
            ops.FLAGS.skip_image_grad = False

        return ((reg_g_loss.detach(), g_loss.detach(), pl_penalty.detach()),
                (reg_d_loss.detach(), d_loss.detach(), r1_penalty.detach()),
                (ocr_loss_w / ocr_loss_weight).detach())

    def _apply_updates(self, handles=None):
        """three Adam updates in the reference's order (g, ocr, d)."""
        hs = list(handles) if handles else [None, None, None]
        for i, (opt, buf) in enumerate(zip((self.g_optimizer, self.ocr_optimizer, self.d_optimizer),
                                           (self.g_grad, self.o_grad, self.d_grad))):
            for h in (hs[i:i + 1] if i < 2 else hs[2:]):  # D's exchange may be several bucket handles
                GradExchange.finish(h)
            opt.apply_gradients(buf)

    # ------------------------------------------------------------------------------------
    def _path_length_reg(self, input_words, rand):
        """training_step.py:300-347."""
        cfg = self.cfg
        pl_mb = max(1, self.batch_size_per_gpu // self.pl_minibatch_shrink)
        dev = input_words.device
        pl_z = rand["pl_z"] if "pl_z" in rand else torch.randn(pl_mb, cfg.z_dim, device=dev)
        mode = "fused2" if (dev.type == "cuda" and ops.TUNING.use_fused2) else "composable"  # (models.py: execution modes)
        img, style = self.generator((input_words[:pl_mb], pl_z), batch_size=pl_mb, ret_style=True, training=False,
                                    rand=rand, noises_key="pl_noises", mode=mode)
        noise = rand["pl_noise"] if "pl_noise" in rand else torch.randn_like(img)
        ops.FLAGS.no_filter_grads = True  # this inner gradient ends at the latents
        try:
            (g,) = torch.autograd.grad((img * (noise * self.pl_noise_scaler)).sum(), style, create_graph=True)
        finally:
            ops.FLAGS.no_filter_grads = False
        lengths = g.square().sum(dim=2).mean(dim=1).sqrt()
        with torch.no_grad():  # assigned BEFORE use, read back as a constant (:336-342)
            self.pl_mean.copy_(self.pl_mean + self.pl_decay * (lengths.mean() - self.pl_mean))
        pen = (lengths - self.pl_mean.detach()).square() * self.pl_minibatch_shrink * self.g_reg_interval
        return pen.sum() / self.batch_size

    def _r1_reg(self, real_images):
        """training_step.py:349-373."""
        real = real_images.detach().clone().requires_grad_(True)
        real_scores = self.discriminator(real, mode="composable")
        ops.FLAGS.no_filter_grads = True  # this inner gradient ends at the image
        try:
            (g,) = torch.autograd.grad(real_scores.sum(), real, create_graph=True)
        finally:
            ops.FLAGS.no_filter_grads = False
        pen = g.square().sum(dim=(1, 2, 3)) * (0.5 * self.r1_gamma) * self.d_reg_interval
        return real_scores, pen.sum() / self.batch_size

    def _get_ocr_loss(self, fake_images, ocr_labels, ocr_images):
        """training_step.py:375-402."""
        inp = self.aster_ocr.convert_inputs(fake_images, ocr_labels, blank_label=1)
        logits = self.aster_ocr(inp)
        if self.cfg.ocr_loss_type == "mse":
            return mean_squared_loss(self.aster_ocr(ocr_images), logits, self.batch_size)
        return softmax_cross_entropy_loss(logits, ocr_labels, self.batch_size)


def build_trainer_state(cfg: Config, device, aster_ocr=None, seed: int = 0, process_group=None,
                        use_graphs: bool = False, compute_dtype: Optional[str] = None):
    """The wiring of reference train.py:25-108 / model_loader.py:13-20: models (g_clone starts as a
    copy of G), lazy-reg-rescaled optimiser settings, three Adam states, pl_mean, TrainingStep."""
    from .aster import AsterInferer
    torch.manual_seed(seed)
    G, D, g_clone = Generator(cfg), Discriminator(cfg), Generator(cfg)
    g_clone.load_state_dict(G.state_dict())
    gf = flatten_generator(G, device)
    flatten_generator(g_clone, device)
    df = flatten_module(D, device)
    g_opt, d_opt = cfg.g_opt.lazy_reg_rescaled(), cfg.d_opt.lazy_reg_rescaled()
    gb, ge = gf.range_of(("latent_encoder.", "synthesis."))
    ob, oe = gf.range_of(("synthesis.", "word_encoder."))
    g_optimizer = AdamTF(gf.flat[gb:ge], g_opt)
    ocr_optimizer = AdamTF(gf.flat[ob:oe], g_opt)
    d_optimizer = AdamTF(df.flat, d_opt)
    pl_mean = torch.zeros((), device=device)
    if aster_ocr is None:  # frozen OCR with its convolutions on the HIP kernels
        from .aster import AsterLikeOCRHip
        aster_ocr = AsterInferer(model=AsterLikeOCRHip(max_steps=cfg.max_char_number), char_width=cfg.char_width,
                                 max_char_number=cfg.max_char_number, image_dims=cfg.aster_image_dims)
    aster_ocr = aster_ocr.to(device)
    step = TrainingStep(G, D, aster_ocr, g_optimizer, ocr_optimizer, d_optimizer, cfg.g_opt.reg_interval,
                        cfg.d_opt.reg_interval, pl_mean, cfg, process_group, use_graphs, compute_dtype)
    # Replicas must start from IDENTICAL weights (same seed above) but draw INDEPENDENT z / z2 / mixing cut-off / noise /
    # dropout / pl_z / pl_noise afterwards, as MirroredStrategy's replicas do (training_step.py:147, latent_encoder.py:47-60,
    # noise.py:16-19): reseed every device generator per rank once the models exist.  The captured HIP graphs read the
    # same default generator (philox seed + offset registered at capture), so the streams stay distinct in replay too.
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        torch.manual_seed(seed + 1 + dist.get_rank(process_group))
    return dict(generator=G, discriminator=D, g_clone=g_clone, g_optimizer=g_optimizer,
                ocr_optimizer=ocr_optimizer, d_optimizer=d_optimizer, pl_mean=pl_mean, aster_ocr=aster_ocr,
                training_step=step)
