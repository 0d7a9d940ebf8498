"""TensorFlow checkpoint files without TensorFlow (textboxgan_amd/tf_checkpoint.py): known-answer, structural and
round-trip tests.  (Parity with files written by TF itself is unpinned: no TF and no sample checkpoint here.)"""
import os
import struct

import numpy as np
import pytest
import torch

from textboxgan_amd import tf_checkpoint as T
from textboxgan_amd.config import small_config


def test_crc32c_known_answers_and_mask():
    # RFC 3720 appendix B.4 test vectors for CRC-32C
    assert T.crc32c(b"123456789") == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    data = np.random.default_rng(0).integers(0, 256, 10000, dtype=np.uint8)
    whole = T.crc32c(data)
    assert T.crc32c(data[4000:], T.crc32c(data[:4000])) == whole  # incremental == one shot
    saved, T._NATIVE = T._NATIVE, None  # pure-python path agrees with the native helper
    try:
        assert T.crc32c(data[:3000]) == T.crc32c(bytes(data[:3000]))
        py = T.crc32c(data[:3000])
    finally:
        T._NATIVE = saved
    assert py == T.crc32c(data[:3000])
    # crc32c.h: Mask(crc) = ((crc >> 15) | (crc << 17)) + 0xa282ead8
    assert T.mask_crc(0) == 0xA282EAD8 and T.unmask_crc(T.mask_crc(0xDEADBEEF)) == 0xDEADBEEF


def test_reader_on_a_hand_assembled_table(tmp_path):
    """a table built byte by byte from the leveldb table_format (independent of write_table)."""
    def block(entries):  # [(shared, key_delta, value)] with one restart at 0
        b = b"".join(bytes([s, len(kd), len(v)]) + kd + v for s, kd, v in entries)
        return b + struct.pack("<I", 0) + struct.pack("<I", 1)

    def trailer(contents):
        return b"\x00" + struct.pack("<I", T.mask_crc(T.crc32c(b"\x00", T.crc32c(contents))))

    data = block([(0, b"apple", b"1"), (3, b"ly", b"22"), (0, b"banana", b"")])  # "apple", "apply" (shares "app"), "banana"
    meta = block([])
    out = data + trailer(data)
    meta_off = len(out)
    out += meta + trailer(meta)
    index = block([(0, b"banana", bytes([0, len(data)]))])  # handle = varint offset 0, varint size
    index_off = len(out)
    out += index + trailer(index)
    footer = bytes([meta_off, len(meta)]) + bytes([index_off, len(index)])
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    p = tmp_path / "hand.index"
    p.write_bytes(out)
    assert T.read_table(str(p)) == [(b"apple", b"1"), (b"apply", b"22"), (b"banana", b"")]
    bad = bytearray(out); bad[2] ^= 1
    p.write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        T.read_table(str(p))


def test_table_roundtrip_and_structure(tmp_path):
    items = [(b"", b"header")] + [(f"key/{i:05d}/.ATTRIBUTES/VARIABLE_VALUE".encode(), os.urandom(20 + i % 7)) for i in range(700)]
    p = str(tmp_path / "t.index")
    T.write_table(p, items)
    assert T.read_table(p) == items
    raw = open(p, "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == T.TABLE_MAGIC and len(raw) > 3 * 4096  # several data blocks
    # footer -> index block -> every data block carries its own checksum and a restart array
    footer = raw[-48:]
    _, pos = T._read_varint(footer, 0); _, pos = T._read_varint(footer, pos)
    ioff, pos = T._read_varint(footer, pos); isize, pos = T._read_varint(footer, pos)
    handles = list(T._iter_block(T._read_block(raw, ioff, isize, True)))
    assert len(handles) >= 4 and [k for k, _ in handles] == sorted(k for k, _ in handles)
    off0, p2 = T._read_varint(handles[0][1], 0)
    size0, _ = T._read_varint(handles[0][1], p2)
    blk = T._read_block(raw, off0, size0, True)
    n_restarts = struct.unpack("<I", blk[-4:])[0]
    assert n_restarts >= 2 and struct.unpack("<I", blk[-4 - 4 * n_restarts:-4 * n_restarts])[0] == 0
    assert size0 < len(b"".join(k + v for k, v in T._iter_block(blk))) + 3 * 200  # prefix compression is in effect
    with pytest.raises(ValueError):
        T.write_table(p, [(b"b", b""), (b"a", b"")])


def test_bundle_roundtrip_dtypes_strings_and_protos(tmp_path):
    g = np.random.default_rng(1)
    tensors = {
        "a/w/.ATTRIBUTES/VARIABLE_VALUE": g.standard_normal((3, 3, 5, 7)).astype(np.float32),
        "a/scalar/.ATTRIBUTES/VARIABLE_VALUE": np.array(2.5, dtype=np.float32),
        "opt/iter/.ATTRIBUTES/VARIABLE_VALUE": np.array(225000, dtype=np.int64),
        "b/flags": np.array([True, False, True]),
        "b/i32": g.integers(-5, 5, (4, 2)).astype(np.int32),
        "b/f64": g.standard_normal(6),
        "empty": np.zeros((0, 4), dtype=np.float32),
        "_CHECKPOINTABLE_OBJECT_GRAPH": b"\x0a\x00 not really a proto \xff\x00",
    }
    prefix = str(tmp_path / "ckpt-7")
    T.write_bundle(prefix, tensors)
    assert sorted(os.listdir(tmp_path)) == ["ckpt-7.data-00000-of-00001", "ckpt-7.index"]
    header, entries = T.read_bundle_index(prefix)
    assert header == dict(num_shards=1, endianness=0, producer=1)
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(e.size for e in entries.values())
    e = entries["a/w/.ATTRIBUTES/VARIABLE_VALUE"]
    assert (e.dtype, e.shape, e.size) == (1, (3, 3, 5, 7), 4 * 315) and entries["opt/iter/.ATTRIBUTES/VARIABLE_VALUE"].dtype == 9
    # the index's raw table: first key is the empty header key, keys strictly increasing
    keys = [k for k, _ in T.read_table(prefix + ".index")]
    assert keys[0] == b"" and keys == sorted(keys)
    back = T.read_bundle(prefix)
    for k, v in tensors.items():
        if isinstance(v, bytes):
            assert back[k] == v
        else:
            assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v)
    # a flipped data byte is caught by the per-tensor checksum
    with open(prefix + ".data-00000-of-00001", "r+b") as f:
        f.seek(entries["b/f64"].offset + 3); b = f.read(1); f.seek(-1, 1); f.write(bytes([b[0] ^ 0x40]))
    with pytest.raises(ValueError):
        T.read_bundle(prefix, ["b/f64"])
    assert np.array_equal(T.read_bundle(prefix, ["b/i32"])["b/i32"], tensors["b/i32"])


def test_object_graph_follows_attribute_paths_and_slots():
    keys = ["generator/synthesis/synth_blocks/3/conv_1/w" + T.VAR_SUFFIX,
            "generator/synthesis/synth_blocks/3/conv_1/mod_dense/w" + T.VAR_SUFFIX,
            "generator/latent_encoder/w_avg" + T.VAR_SUFFIX,
            "g_optimizer/iter" + T.VAR_SUFFIX,
            "generator/synthesis/synth_blocks/3/conv_1/w" + T.SLOT_TAG + "g_optimizer/m" + T.VAR_SUFFIX,
            "generator/synthesis/synth_blocks/3/conv_1/w" + T.SLOT_TAG + "g_optimizer/v" + T.VAR_SUFFIX,
            "pl_mean" + T.VAR_SUFFIX]
    nodes = T.parse_object_graph(T.build_object_graph(keys))
    assert [n for n, _ in nodes[0].children] == ["generator", "g_optimizer", "pl_mean"]

    def walk(path):
        nid = 0
        for comp in path.split("/"):
            nid = dict(nodes[nid].children)[comp]
        return nid
    w = walk("generator/synthesis/synth_blocks/3/conv_1/w")
    assert nodes[w].attributes == [("VARIABLE_VALUE", "generator.synthesis.synth_blocks.3.conv_1.w", keys[0])]
    opt = walk("g_optimizer")
    assert sorted(s for _, s, _ in nodes[opt].slots) == ["m", "v"] and all(o == w for o, _, _ in nodes[opt].slots)
    slot_node = nodes[[sid for _, s, sid in nodes[opt].slots if s == "m"][0]]
    assert slot_node.attributes[0][2] == keys[4]
    paths = T.graph_paths(nodes)
    assert paths["generator/latent_encoder/w_avg"] == keys[2] and paths["pl_mean"] == keys[6]


def _state(seed):
    from textboxgan_amd.training_step import build_trainer_state
    return build_trainer_state(small_config(2), torch.device("cpu"), seed=seed)


def test_trainer_checkpoint_roundtrip_layout_and_partial_restore(tmp_path):
    """the reference's checkpoint contents (train.py:94-108) written, listed and restored: keys, object graph, Adam
    slots, iterations (the global step, train.py:179), pl_mean; inference-style partial restore of g_clone."""
    a = _state(1)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for name in ("g_optimizer", "ocr_optimizer", "d_optimizer"):
            a[name].m.copy_(torch.randn(a[name].m.shape, generator=g)); a[name].v.copy_(torch.rand(a[name].v.shape, generator=g))
            a[name].step.fill_(1234); a[name]._iterations = 1234
        a["pl_mean"].fill_(0.731)
        a["generator"].latent_encoder.w_avg.copy_(torch.randn(a["generator"].latent_encoder.w_avg.shape, generator=g))
    ckpt_dir = str(tmp_path / "experiments" / "run" / "checkpoints")
    prefix = T.save_checkpoint(ckpt_dir, a)
    assert prefix.endswith("ckpt-1234") and T.latest_checkpoint(ckpt_dir) == prefix
    assert open(os.path.join(ckpt_dir, "checkpoint")).read().splitlines()[0] == 'model_checkpoint_path: "ckpt-1234"'
    _, entries = T.read_bundle_index(prefix)
    for k in ("generator/synthesis/synth_blocks/3/conv_1/w", "g_clone/word_encoder/fc/kernel", "discriminator/last_dense/w",
              "generator/latent_encoder/w_avg", "g_optimizer/iter", "d_optimizer/beta_2", "pl_mean", "save_counter"):
        assert k + T.VAR_SUFFIX in entries, k
    assert "generator/synthesis/synth_blocks/3/conv_1/w/.OPTIMIZER_SLOT/ocr_optimizer/v" + T.VAR_SUFFIX in entries
    assert "generator/word_encoder/fc/kernel/.OPTIMIZER_SLOT/ocr_optimizer/m" + T.VAR_SUFFIX in entries
    assert "generator/word_encoder/fc/kernel/.OPTIMIZER_SLOT/g_optimizer/m" + T.VAR_SUFFIX not in entries  # not g_optimizer's
    assert entries["generator/synthesis/synth_blocks/3/conv_1/w" + T.VAR_SUFFIX].shape == (3, 3, 8, 8)  # HWIO, as TF stores it
    graph = T.parse_object_graph(T.read_bundle(prefix, [T.OBJECT_GRAPH_KEY])[T.OBJECT_GRAPH_KEY])
    assert sorted(n for n, _ in graph[0].children) == sorted(
        ["d_optimizer", "g_optimizer", "ocr_optimizer", "discriminator", "generator", "g_clone", "pl_mean", "save_counter"])

    b = _state(2)
    assert not torch.equal(b["generator"]._flat.flat, a["generator"]._flat.flat)
    rep = T.load_checkpoint(prefix, b)
    assert not rep["missing"]
    for name in ("generator", "g_clone", "discriminator"):
        for (k, va), (_, vb) in zip(a[name].state_dict().items(), b[name].state_dict().items()):
            assert torch.equal(va, vb), (name, k)
    for name, owners in T._optimizer_ranges(a).items():  # every parameter's slot slice (alignment padding is not a variable)
        n_checked = 0
        for _, flat, begin in owners:
            for p, off in zip(flat.params, flat.offsets):
                lo = off - begin
                if 0 <= lo and lo + p.numel() <= a[name].m.numel():
                    assert torch.equal(a[name].m[lo:lo + p.numel()], b[name].m[lo:lo + p.numel()])
                    assert torch.equal(a[name].v[lo:lo + p.numel()], b[name].v[lo:lo + p.numel()])
                    n_checked += 1
        assert n_checked > 10
        assert b[name].iterations == 1234 and int(b[name].step) == 1234
    assert abs(float(b["pl_mean"]) - 0.731) < 1e-7
    # restored parameters still alias the flat buffers the optimisers update
    assert b["generator"].synthesis.synth_blocks[0].conv_0.w.data_ptr() >= b["generator"]._flat.flat.data_ptr()

    # inference: only g_clone, expect_partial (infer.py / model_loader.py:74-77)
    c = _state(3)
    rep = T.load_checkpoint(prefix, dict(g_clone=c["g_clone"]), expect_partial=True)
    assert len(rep["restored"]) == len(a["g_clone"].state_dict()) and rep["unused"]
    for (k, va), (_, vc) in zip(a["g_clone"].state_dict().items(), c["g_clone"].state_dict().items()):
        assert torch.equal(va, vc), k
    # a checkpoint lacking requested values raises unless expect_partial
    t = T.trainer_state_tensors(a)
    del t["discriminator/last_dense/w" + T.VAR_SUFFIX]
    t[T.OBJECT_GRAPH_KEY] = T.build_object_graph(sorted(k for k in t if k != T.OBJECT_GRAPH_KEY))
    T.write_bundle(str(tmp_path / "partial"), t)
    with pytest.raises(KeyError):
        T.load_checkpoint(str(tmp_path / "partial"), _state(4))
    assert "discriminator/last_dense/w" in T.load_checkpoint(str(tmp_path / "partial"), _state(4), expect_partial=True)["missing"]


def test_checkpoint_manager_keeps_five(tmp_path):
    a = _state(5)
    d = str(tmp_path / "ck")
    for step in (10, 20, 30, 40, 50, 60, 70):
        T.save_checkpoint(d, a, step=step)
    names = sorted(f for f in os.listdir(d) if f.endswith(".index"))
    assert names == [f"ckpt-{s}.index" for s in (30, 40, 50, 60, 70)]  # max_to_keep = 5 (model_loader.py:64-66)
    lines = open(os.path.join(d, "checkpoint")).read().splitlines()
    assert lines[0] == 'model_checkpoint_path: "ckpt-70"' and len(lines) == 6
    assert int(T.read_bundle(os.path.join(d, "ckpt-70"), ["save_counter" + T.VAR_SUFFIX])["save_counter" + T.VAR_SUFFIX]) == 7


def test_aster_weight_import_from_a_savedmodel_variables_bundle(tmp_path):
    """AsterLikeOCR.load_weights_tf reads ``<saved_model>/variables/variables.{index,data-*}`` without TensorFlow and
    converts the TF layouts (HWIO conv, [in,out] dense, fused LSTM kernel with gate order i,j,f,o)."""
    from textboxgan_amd.aster import AsterLikeOCR
    net = AsterLikeOCR(max_steps=8)
    g = np.random.default_rng(0)
    H = net.hidden
    conv = g.standard_normal((3, 3, 3, 32)).astype(np.float32)               # stem conv, HWIO
    dense = g.standard_normal((H, net.num_classes)).astype(np.float32)        # Predictor/dense/kernel [in, out]
    bias = g.standard_normal((net.num_classes,)).astype(np.float32)
    n_in = net.cell.weight_ih.shape[1]
    kernel = g.standard_normal((n_in + H, 4 * H)).astype(np.float32)          # Predictor/lstm_cell/kernel
    sm = tmp_path / "aster_weights"
    os.makedirs(sm / "variables")
    T.write_bundle(str(sm / "variables" / "variables"), {
        "Forward/Predictor/dense/kernel" + T.VAR_SUFFIX: dense, "Forward/Predictor/dense/bias" + T.VAR_SUFFIX: bias,
        "Forward/Predictor/lstm_cell/kernel" + T.VAR_SUFFIX: kernel, "FeatureExtractor/stem/kernel" + T.VAR_SUFFIX: conv})
    listed = AsterLikeOCR.list_tf_variables(str(sm))
    assert listed["Forward/Predictor/dense/kernel"] == (H, net.num_classes) and len(listed) == 4
    missing = net.load_weights_tf(str(sm), {
        "stem.conv.weight": "FeatureExtractor/stem/kernel", "out.weight": "Forward/Predictor/dense/kernel",
        "out.bias": "Forward/Predictor/dense/bias", "cell.weight_ih": ("Forward/Predictor/lstm_cell/kernel", "lstm_kernel")})
    assert missing == []
    assert torch.equal(net.stem.conv.weight, torch.from_numpy(conv).permute(3, 2, 0, 1))
    assert torch.equal(net.out.weight, torch.from_numpy(dense).t()) and torch.equal(net.out.bias, torch.from_numpy(bias))
    i, j, f, o = np.split(kernel, 4, axis=1)
    torch_order = np.concatenate([i, f, j, o], axis=1)                          # torch gates: i, f, g(=j), o
    assert torch.equal(net.cell.weight_ih, torch.from_numpy(torch_order[:n_in].T.copy()))
    assert torch.equal(net.cell.weight_hh, torch.from_numpy(torch_order[n_in:].T.copy()))
    with pytest.raises(KeyError):
        net.load_weights_tf(str(sm), {"out.bias": "no/such/variable"})
    # the cell's single TF bias (gate order i, j, f, o): lands in one torch bias in torch order with TF's run-time
    # forget_bias added to the f gate; the paired torch bias is zeroed (torch adds both) -- ADVICE round 2
    lb = g.standard_normal((4 * H,)).astype(np.float32)
    T.write_bundle(str(sm / "variables" / "variables"), {"Forward/Predictor/lstm_cell/bias" + T.VAR_SUFFIX: lb})
    with torch.no_grad():
        net.cell.bias_hh.fill_(0.5)
    net.load_weights_tf(str(sm), {"cell.bias_ih": ("Forward/Predictor/lstm_cell/bias", "lstm_bias")}, forget_bias=1.0)
    bi, bj, bf, bo = np.split(lb, 4)
    assert torch.equal(net.cell.bias_ih, torch.from_numpy(np.concatenate([bi, bf + 1.0, bj, bo])))
    assert float(net.cell.bias_hh.abs().max()) == 0.0
