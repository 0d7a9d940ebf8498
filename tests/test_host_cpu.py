"""CPU (-m "not gpu"): host logic, the C-ABI surface, and the 2-process gradient exchange (gloo)."""
import ctypes
import os
import re
import socket

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tokenizers_known_answers():
    from textboxgan_amd.char_tokens import (main_to_aster_labels, string_to_aster_int_sequence,
                                            string_to_main_int_sequence)
    assert string_to_main_int_sequence(["Hello"]).tolist() == [[44, 15, 22, 22, 25, 0, 0, 0]]
    assert string_to_aster_int_sequence(["Hello"]).tolist() == [[45, 16, 23, 23, 26, 1, 1, 1]]
    assert string_to_main_int_sequence(["0", "\""]).tolist()[0][0] == 1 and string_to_main_int_sequence(["\""]).tolist()[0][0] == 69
    assert string_to_main_int_sequence(["aéb"]).tolist()[0][:3] == [11, 0, 12]  # OOV -> 0
    assert string_to_main_int_sequence(["ABCDEFGHIJ"]).tolist() == [[39, 40, 41, 42, 43, 44, 45, 46]]  # truncating="pre"
    w = string_to_main_int_sequence(["Hello", "a-'b"])
    assert main_to_aster_labels(w).tolist() == string_to_aster_int_sequence(["Hello", "a-'b"]).tolist()


def test_config_matches_reference_values():
    from textboxgan_amd.config import Config, cfg
    assert cfg.image_width == 256 and cfg.generator_feat_maps == [128, 512, 256, 256, 128, 128]
    assert cfg.discrim_feat_maps == [64, 128, 128, 256, 256, 512, 512] and cfg.batch_size == 4
    g, d = cfg.g_opt.lazy_reg_rescaled(), cfg.d_opt.lazy_reg_rescaled()
    assert abs(g.learning_rate - 1.7778e-3) < 1e-7 and abs(g.beta2 - 0.99111) < 1e-5
    assert abs(d.learning_rate - 1.8824e-3) < 1e-7 and abs(d.beta2 - 0.99059) < 1e-5 and g.beta1 == 0.0
    assert Config(batch_size_per_gpu=32, num_replicas=8).batch_size == 256
    with pytest.raises(AssertionError):
        Config(char_height=32)


def test_header_symbols_are_exported_by_the_library():
    """every function include/tbg.h declares must be an exported symbol of libtbg_hip.so (no compute call)."""
    from textboxgan_amd import native
    from textboxgan_amd.build import build_native
    lib = ctypes.CDLL(build_native(verbose=False))
    header = open(os.path.join(ROOT, "include", "tbg.h")).read()
    declared = sorted(set(re.findall(r"\b(tbg_[a-z0-9_]+)\s*\(", header)) - {"tbg_epilogue"})
    assert len(declared) >= 13
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in tbg.h but not exported"
    assert set(native.EXPORTS) == set(declared)
    lib.tbg_strerror.restype = ctypes.c_char_p
    assert lib.tbg_version() >= 100 and b"ok" in lib.tbg_strerror(0) and b"HIP" in lib.tbg_strerror(-3)
    # the chunk size is the library's own constant: read it from the source the library was built from
    src = open(os.path.join(ROOT, "textboxgan_amd", "csrc", "elementwise.hip")).read()
    chunk = int(re.search(r"#define\s+BA_CHUNK\s+(\d+)", src).group(1))
    for hw in (1, 64, chunk, chunk + 1, 16384, 5 * chunk - 1):
        assert lib.tbg_bias_act_bwd_chunks(hw) == -(-hw // chunk)
    assert lib.tbg_bias_act_bwd_chunks(0) == 0


def test_small_map_and_recurrent_entries_answer_their_queries_and_refuse_bad_arguments():
    """round 6 entries, host side only (no launch): the dispatch queries of tbg_conv2d_units_small are pure functions of the
    descriptor -- block counts of the geometries a step launches, the 32 / 64-pixel tile rule, the fused dot only where a tile stays
    inside one sample -- and every entry answers bad arguments with an error code BEFORE touching a pointer (upfirdn_2d.cu:241-266
    convention: TBG_EINVAL -1 / TBG_EUNSUPPORTED -4)."""
    import ctypes as C
    from textboxgan_amd import native as N
    lib = N.lib()
    D = N.ConvDesc
    q = lambda d, planes=3: (lib.tbg_conv2d_units_small_blocks(C.byref(d), planes), lib.tbg_conv2d_units_small_tile_pixels(C.byref(d), planes),
                             lib.tbg_conv2d_units_small_dot_slots(C.byref(d), planes))
    # recogniser stage 4 (2x25 maps, B = 16): 800 pixels = 25 tiles of 32 x 8 channel tiles; a tile straddles samples -> no fused dot
    assert q(D(16, 256, 256, 2, 25, 2, 25, 3, 3, 1, 1, 1, 1, 0, 0, 256, 1)) == (200, 32, 0)
    # generator 4x16 512 -> 512: 512 blocks of 32 pixels would be two rounds -> 64-pixel tiles, one sample per tile, 2 dot slots
    assert q(D(16, 512, 512, 4, 16, 4, 16, 3, 3, 1, 1, 1, 1, 0, 0, 512, 1)) == (256, 64, 2)
    # 1x1 stride (2, 1) of a stage's first unit, and its transposed data gradient
    assert q(D(16, 128, 256, 4, 25, 2, 25, 1, 1, 2, 1, 0, 0, 0, 0, 256, 1))[:2] == (200, 32)
    assert q(D(16, 256, 128, 2, 25, 4, 25, 1, 1, 2, 1, 0, 0, 1, 0, 128, 1))[:2] == (200, 32)
    # stride-2 transposed 3x3 (4x16 -> 9x33): four output-parity classes in one launch, no dot
    b, px, slots = q(D(16, 512, 256, 4, 16, 9, 33, 3, 3, 2, 2, 0, 0, 1, 0, 256, 1))
    assert px == 64 and slots == 0 and b == 8 * sum(-(-16 * u * v // 64) for u, v in ((5, 17), (5, 16), (4, 17), (4, 16)))
    for bad in (D(2, 64, 64, 8, 32, 8, 32, 3, 3, 1, 1, 1, 1, 0, 0, 64, 2),      # split K belongs to the NCHW entry
                D(2, 64, 64, 8, 32, 4, 16, 3, 3, 2, 2, 1, 1, 0, 0, 64, 1),      # padded strided 3x3
                D(2, 64, 64, 8, 2, 8, 2, 3, 3, 1, 1, 1, 1, 0, 0, 64, 1),        # 2-wide rows
                D(2, 64, 64, 8, 32, 8, 32, 5, 5, 1, 1, 2, 2, 0, 0, 64, 1)):     # 5x5
        assert q(bad) == (-4, -4, -4)
    assert q(D(2, 24, 64, 8, 32, 8, 32, 3, 3, 1, 1, 1, 1, 0, 0, 64, 1), planes=1) == (-4, -4, -4)  # bf16: whole 16-channel chunks
    assert lib.tbg_conv2d_units_small_blocks(None, 3) == -1 and q(D(2, 64, 64, 8, 32, 8, 32, 3, 3, 1, 1, 1, 1, 0, 0, 64, 1), planes=2)[0] == -4
    d = D(2, 64, 64, 8, 32, 8, 32, 3, 3, 1, 1, 1, 1, 0, 0, 64, 1)
    assert lib.tbg_conv2d_units_small(C.byref(d), None, 3, None, None, None, None) == -1          # NULL operands
    assert lib.tbg_conv2d_units_small(C.byref(d), C.c_void_p(8), 3, C.c_void_p(16), C.c_void_p(16), None, None) == -1  # misaligned unit tensor
    one = C.c_void_p(64)  # (never dereferenced: the entries validate first)
    assert lib.tbg_lstm_fused_fwd_f32(one, one, one, C.c_void_p(128), one, one, None, 2, 25, 16, 8, 0, None) == -4   # H % 32 != 0
    assert lib.tbg_lstm_fused_fwd_f32(one, one, one, one, one, one, None, 2, 25, 16, 256, 1, None) == -1             # hT_in == hT_out
    assert lib.tbg_lstm_fused_fwd_f32(one, one, None, C.c_void_p(128), one, one, None, 2, 25, 16, 256, 25, None) == -1  # s out of range
    assert lib.tbg_lstm_fused_bwd_f32(None, one, None, C.c_void_p(128), one, one, one, None, 2, 25, 16, 256, 24, 1, None) == -1  # first step needs dseq
    assert lib.tbg_lstm_cell_fused_fwd_f32(one, one, one, C.c_void_p(128), one, one, 8, 16, 256, 770, 0, None) == -4  # K % 16 != 0
    assert lib.tbg_rows_gemv_t_f32(one, one, one, 1000, 768, 16, None) == -4 and lib.tbg_rows_gemv_t_f32(None, one, one, 1024, 768, 16, None) == -1
    assert lib.tbg_dec_sample_fwd_f32(*([None] * 14), 16, 25, 256, 512, 97, 97, None) == -1   # neither logits nor a next step
    assert lib.tbg_dec_sample_fwd_f32(one, one, one, one, one, one, one, one, one, one, one, one, one, one, 16, 65, 256, 512, 97, 97, None) == -4  # T > 64
    assert lib.tbg_dec_sample_bwd_f32(*([None] * 16), 16, 25, 256, 512, 0, None) == -1


def test_ctypes_struct_layout_matches_header(tmp_path):
    """sizes and EVERY field offset of the ctypes mirrors against what a C compiler makes of include/tbg.h (gcc, host only)"""
    import subprocess
    from textboxgan_amd import native
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {"tbg_epilogue": native.Epilogue, "tbg_conv_desc": native.ConvDesc, "tbg_wgrad_desc": native.WgradDesc,
               "tbg_pack_item": native.PackItem, "tbg_dense_item": native.DenseItem}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "tbg.h"', 'int main(void) {']
    for cname, st in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        c, f, v = ln.split()
        got[(c, f)] = int(v)
    for cname, st in structs.items():
        assert got[(cname, "size")] == ctypes.sizeof(st), cname
        for fname, _ in st._fields_:
            assert got[(cname, fname)] == getattr(st, fname).offset, (cname, fname)
    assert ctypes.sizeof(native.ConvDesc) == 17 * 4


def test_product_refuses_cpu_tensors():
    """no CPU fallback in the product path: kernels need device tensors, the library must exist."""
    from textboxgan_amd import native, ops
    with pytest.raises(native.TbgError):
        ops.upfirdn2d_raw(torch.zeros(1, 1, 4, 4), torch.ones(4, 4))


def test_state_dict_is_the_reference_checkpoint_layout():
    from oracle import ref_model as M
    from textboxgan_amd.config import cfg
    from textboxgan_amd.models import Discriminator, Generator
    G, D = Generator(cfg), Discriminator(cfg)
    Pg, Pd = M.init_generator(cfg), M.init_discriminator(cfg)
    assert {k: tuple(v.shape) for k, v in G.state_dict().items()} == {k: tuple(v.shape) for k, v in Pg.items()}
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == {k: tuple(v.shape) for k, v in Pd.items()}
    assert "synthesis.synth_blocks.3.conv_1.w" in Pg and "blocks.5.conv_skip.w" in Pd and "latent_encoder.w_avg" in Pg
    assert G.n_style == 15


def test_flat_params_ranges_and_grad_views():
    from textboxgan_amd.config import small_config
    from textboxgan_amd.models import Generator
    from textboxgan_amd.optim import flatten_generator, write_grads
    G = Generator(small_config(2))
    before = {k: v.clone() for k, v in G.state_dict().items()}
    fp = flatten_generator(G, torch.device("cpu"))
    for k, v in G.state_dict().items():
        assert torch.equal(v, before[k])
    gb, ge = fp.range_of(("latent_encoder.", "synthesis."))
    ob, oe = fp.range_of(("synthesis.", "word_encoder."))
    assert gb == 0 and oe == fp.total and ob < ge  # synthesis is shared by both optimisers
    assert all(o % 4 == 0 for o in fp.offsets)  # 16-byte aligned conv weights
    buf, views = fp.make_grad_buffer(ob, oe)
    params = fp.select(("synthesis.", "word_encoder."))
    assert [v.shape for v in views] == [p.shape for p in params]
    write_grads(views, [torch.full_like(p, 2.0) if i % 2 else None for i, p in enumerate(params)])
    assert float(buf.sum()) == 2.0 * sum(p.numel() for i, p in enumerate(params) if i % 2)
    p0 = fp.params[0]
    fp.flat[fp.offsets[0]] = 123.0  # parameters are views of the flat buffer
    assert float(p0.detach().reshape(-1)[0]) == 123.0


def test_aster_wrapper_matches_oracle_wrapper_on_cpu():
    from oracle import ref_model as M
    from textboxgan_amd.aster import AsterInferer
    from textboxgan_amd.config import cfg
    o = AsterInferer()
    fake = torch.randn(3, 3, 64, 256)
    labels = torch.tensor([[5, 6, 7, 1, 1, 1, 1, 1], [2, 3, 4, 5, 6, 7, 8, 9], [4, 1, 1, 1, 1, 1, 1, 1]])
    np.testing.assert_allclose(o.convert_inputs(fake, labels).numpy(), M.ocr_convert_inputs(fake, labels, cfg).numpy(), atol=1e-5)
    logits = torch.randn(2, 8, 97, generator=torch.Generator().manual_seed(0)) * 0.01
    logits[:, :, 1] = -5.0  # nobody predicts EOS ...
    logits[0, 2, 1] = 5.0   # ... except sample 0 at step 2: the network's dynamic decode emits 3 steps for it
    lengths = o.model.decode_lengths(logits)
    assert lengths.tolist() == [3, 8]
    # the wrapper is the literal :116-151 rule; the oracle applies it sample by sample to each sample's own [1,T_i,C]
    out = o._postprocess_simple(logits, lengths)
    exp = torch.cat([M.ocr_postprocess_simple(logits[i:i + 1, : int(lengths[i])], 8) for i in range(2)])
    assert torch.equal(out, exp)
    assert torch.equal(out[0, :3], logits[0, :3]) and float(out[0, 3, 1]) == 1000.0 and float(out[0, 7].sum()) == 1000.0
    assert torch.equal(o._postprocess_simple(logits), logits)  # without lengths: plain truncate/pad only
    short = o._postprocess_simple(logits[:, :5])
    assert torch.equal(short, M.ocr_postprocess_simple(logits[:, :5], 8))
    # whole wrapper (batched) == the reference's per-sample loop over the serving signature (aster_inferer.py:28-37)
    x = o.convert_inputs(fake[:2], labels[:2])
    with torch.no_grad():
        np.testing.assert_allclose(o(x).numpy(), M.ocr_call(x, o.model.serve, 8).numpy(), atol=2e-5)
    assert not any(p.requires_grad for p in o.parameters())


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _exchange_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from textboxgan_amd.dist_utils import GradExchange
    ex = GradExchange()
    bufs = [torch.full((n,), float(rank + 1) * (i + 1)) for i, n in enumerate((1000, 37, 5003))]
    handles = [ex.start(b) for b in bufs]  # issued back to back, waited for in order (as the step does)
    for h in handles:
        GradExchange.finish(h)
    scal = ex.reduce_scalars([torch.tensor(float(rank + 1)), torch.tensor(10.0 * (rank + 1))])
    more = [torch.ones(4) * (rank + 1)]
    ex.reduce_now(more)
    q.put((rank, [float(b[0]) for b in bufs], [float(b.sum()) for b in bufs], [float(s) for s in scal], float(more[0][0]), ex.world_size()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gradient_exchange_processes_gloo(world):
    """the N>1 path: SUM all-reduce of the three flat gradient buffers + the loss scalars; world_size 2 and 8 (BASELINE
    configs[3] is eight replicas: the collectives and the SUM convention at the real replica count)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=240) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    tri = float(world * (world + 1) // 2)  # sum of (rank + 1)
    for rank, firsts, sums, scal, more, ws in res:
        assert firsts == [tri, 2 * tri, 3 * tri] and sums == [tri * 1000, 2 * tri * 37, 3 * tri * 5003]
        assert scal == [tri, 10 * tri] and more == tri and ws == world


def test_grad_exchange_is_a_noop_single_process():
    from textboxgan_amd.dist_utils import GradExchange
    ex = GradExchange()
    b = torch.ones(5)
    assert ex.start(b) is None and ex.world_size() == 1
    ex.reduce_now([b])
    assert float(b.sum()) == 5.0 and [float(v) for v in ex.reduce_scalars([torch.tensor(2.0)])] == [2.0]


def test_aster_wrapper_combine_forward_and_backward_matches_oracle_on_cpu():
    """combine_forward_and_backward=True (aster_inferer.py:39-114): hand-made logits with a known answer, then the whole
    wrapper against the oracle's per-sample restatement."""
    from oracle import ref_model as M
    from textboxgan_amd.aster import AsterInferer, AsterLikeOCR
    C = 7
    def row(cls, conf):
        r = torch.zeros(C); r[cls] = conf; return r
    # forward predictor: "ab" then blank; backward predictor (reads right-to-left): "b"(conf 9), "a"(conf 1), blank
    fwd = torch.stack([row(2, 3.0), row(3, 4.0), row(1, 5.0)])[None]
    bwd = torch.stack([row(3, 9.0), row(2, 1.0), row(1, 5.0)])[None]
    out = AsterInferer._combine_logits(fwd, bwd)
    # step 0: forward 3.0 vs reversed-backward 1.0 -> forward; step 1: 4.0 vs 9.0 -> backward
    assert out.shape == (1, 2, C) and torch.equal(out[0, 0], row(2, 3.0)) and torch.equal(out[0, 1], row(3, 9.0))
    assert torch.equal(out, M.ocr_combine_logits(fwd, bwd))
    net = AsterLikeOCR(max_steps=8, backward_predictor=True)
    plain = AsterLikeOCR(max_steps=8)
    assert torch.equal(net.out.weight, plain.out.weight) and torch.equal(net.cell.weight_ih, plain.cell.weight_ih)
    o = AsterInferer(model=net, combine_forward_and_backward=True)
    x = torch.randn(2, 64, 256, 3, generator=torch.Generator().manual_seed(3)) * 0.5
    with torch.no_grad():
        got = o(x)
        exp = M.ocr_call(x, net.serve, 8, combine_forward_and_backward=True)
    assert got.shape == (2, 8, 97)
    np.testing.assert_allclose(got.numpy(), exp.numpy(), atol=1e-5)
    full = o._postprocess_combine({"forward_logits": fwd, "backward_logits": bwd})
    assert full.shape == (1, 8, C) and torch.equal(full[0, 2], row(1, 5.0)) and float(full[0, 3, 1]) == 1000.0
    with pytest.raises(ValueError):
        AsterInferer(model=plain, combine_forward_and_backward=True)


def test_ops_host_state_is_lock_guarded():
    """pruning flags, arithmetic mode and packed-filter scopes are process-wide (torch's autograd thread must see them) and
    guarded by a re-entrant lock that a step holds across its passes: a second thread that wants
    to run its own step waits for the first one's scope to end and then finds the state restored."""
    import threading
    import time

    from textboxgan_amd import ops
    seen = {}

    def other():
        with ops.STATE_LOCK:  # what TrainingStep._compute_grads does first
            seen["flags"] = (ops.FLAGS.skip_d_wgrad, ops.FLAGS.d_first_half, ops.FLAGS.no_filter_grads)
            seen["mode"] = ops.compute_mode()
            seen["scopes"] = (ops._TLS.pack_step, ops._TLS.pack_store)
            seen["t"] = time.monotonic()

    with ops.STATE_LOCK, ops.STATE_LOCK:  # re-entrant: nested scopes of one thread
        ops.FLAGS.skip_d_wgrad, ops.FLAGS.d_first_half, ops.FLAGS.no_filter_grads = True, 7, True
        try:
            with ops.compute_dtype("f32x3"), ops.filter_cache(), ops.PackedStore().scope():
                t = threading.Thread(target=other)
                t.start()
                time.sleep(0.2)
                assert "flags" not in seen, "the second thread must wait for the lock"
                assert ops.compute_mode() == "f32x3"
        finally:
            ops.FLAGS.skip_d_wgrad, ops.FLAGS.d_first_half, ops.FLAGS.no_filter_grads = False, 0, False
        released = time.monotonic()
    t.join()
    assert seen["t"] >= released
    assert (seen["flags"], seen["mode"], seen["scopes"]) == ((False, 0, False), "f32", (None, None))
    with __import__("pytest").raises(AttributeError):
        ops.FLAGS.no_such_flag = 1
