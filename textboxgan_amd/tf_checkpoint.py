"""TensorFlow object-graph checkpoints (``tf.train.Checkpoint`` / TensorBundle V2 files) WITHOUT TensorFlow.

The reference checkpoints with ``tf.train.Checkpoint(d_optimizer, g_optimizer, ocr_optimizer, discriminator, generator,
g_clone, pl_mean)`` + ``CheckpointManager`` (train.py:94-108, models/model_loader.py:57-81): files ``ckpt-<step>.index``,
``ckpt-<step>.data-00000-of-00001`` and the ``checkpoint`` state file.  BASELINE.json's north_star asks the build to keep
that layout so the published 225 K-step weights can be loaded and TF's ``infer.py`` can consume weights trained here.
This module implements the three pieces of the on-disk format:

* ``.index``  -- an SSTable in the leveldb table format (prefix-compressed data blocks with restart arrays, an index
  block, a 48-byte footer with the magic 0xdb4775248b80fb57, every block followed by a type byte + masked CRC-32C)
  mapping  ""  -> ``BundleHeaderProto``  and  tensor key -> ``BundleEntryProto`` (tensor_bundle.proto).
* ``.data-XXXXX-of-NNNNN`` -- raw little-endian tensor bytes; DT_STRING tensors as
  [varint64 lengths][masked crc32c of the lengths][bytes] (tensor_bundle.cc WriteStringTensor).
* ``_CHECKPOINTABLE_OBJECT_GRAPH`` -- a scalar string tensor holding the ``TrackableObjectGraph`` proto
  (trackable_object_graph.proto): the object tree TF walks by attribute name on restore.  Variable keys are
  ``<attribute path>/.ATTRIBUTES/VARIABLE_VALUE``, Adam slots
  ``<variable path>/.OPTIMIZER_SLOT/<optimizer>/{m,v}/.ATTRIBUTES/VARIABLE_VALUE``.

PARITY UNPINNED: there is no TensorFlow and no sample checkpoint in the build container (``experiments/.keep`` only), so
the format is implemented from the published specifications of TF 2.8 (poetry.lock:751-753) and leveldb and is checked by
write->read round trips and structural tests only.  The protobuf messages are encoded/decoded by hand (wire format:
varints, length-delimited fields, fixed32) -- the needed .proto files are not in the reference tree either.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

# ---------------------------------------------------------------------------------------------------------------
# CRC-32C (Castagnoli), masked as in tensorflow/core/lib/hash/crc32c.h
# ---------------------------------------------------------------------------------------------------------------
_CRC_TABLE: Optional[np.ndarray] = None
_MASK_DELTA = 0xA282EAD8


def _crc_table() -> np.ndarray:
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.zeros(256, dtype=np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t[i] = c
        _CRC_TABLE = t
    return _CRC_TABLE


def _native_crc():
    """tbg_crc32c (host code: SSE4.2 crc32 instruction).  Looked for first in libtbg_host.so -- the same source built with
    g++ and NO HIP runtime dependency, so a CPU-only inference host loading a checkpoint gets the native checksum too
     -- then in libtbg_hip.so; None if neither loads."""
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ("libtbg_host.so", "libtbg_hip.so"):
        path = os.path.join(here, name)
        try:
            if os.path.exists(path):
                fn = C.CDLL(path).tbg_crc32c
                fn.restype = C.c_uint32
                fn.argtypes = [C.c_void_p, C.c_longlong, C.c_uint32]
                return fn
        except (OSError, AttributeError):  # not loadable here (e.g. no HIP runtime): try the next one
            continue
    return None


_NATIVE = "unset"


def crc32c(data, crc: int = 0) -> int:
    """CRC-32C of ``data`` (bytes / bytearray / memoryview / contiguous ndarray), continuing from ``crc``."""
    global _NATIVE
    if isinstance(data, np.ndarray):
        data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    else:
        data = np.frombuffer(data, dtype=np.uint8)
    if data.size == 0:
        return crc
    if _NATIVE == "unset":
        _NATIVE = _native_crc()
    if _NATIVE is not None and data.size >= 64:
        return int(_NATIVE(data.ctypes.data, data.size, crc))
    if data.size >= (1 << 20):
        import warnings
        warnings.warn("crc32c: native library not loadable, checksumming %d bytes in pure Python (slow): build "
                      "textboxgan_amd/libtbg_host.so with `python -m textboxgan_amd.build`" % data.size)
    t = _crc_table()
    c = crc ^ 0xFFFFFFFF
    for b in data.tolist():
        c = int(t[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(m: int) -> int:
    rot = (m - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------------------
# protobuf wire format (the handful of messages the format needs)
# ---------------------------------------------------------------------------------------------------------------
def _varint(n: int) -> bytes:
    if n < 0:
        n += 1 << 64
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos: int) -> Tuple[int, int]:
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _field_varint(num: int, v: int) -> bytes:
    return _varint(num << 3) + _varint(v)


def _field_bytes(num: int, v: bytes) -> bytes:
    return _varint((num << 3) | 2) + _varint(len(v)) + v


def _field_fixed32(num: int, v: int) -> bytes:
    return _varint((num << 3) | 5) + struct.pack("<I", v)


def _parse_fields(buf: bytes) -> List[Tuple[int, int, object]]:
    """[(field number, wire type, value)]: varint -> int, fixed32/64 -> int, length-delimited -> bytes."""
    out, pos = [], 0
    while pos < len(buf):
        tag, pos = _read_varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + n]); pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        out.append((num, wt, v))
    return out


# tensorflow/core/framework/types.proto
DT = {"float32": 1, "float64": 2, "int32": 3, "uint8": 4, "int16": 5, "int8": 6, "string": 7, "int64": 9, "bool": 10,
      "uint16": 17, "float16": 19, "uint32": 22, "uint64": 23}
DT_INV = {v: k for k, v in DT.items()}
DT_BFLOAT16 = 14


def _shape_proto(shape: Sequence[int]) -> bytes:
    return b"".join(_field_bytes(2, _field_varint(1, int(d))) for d in shape)  # TensorShapeProto.dim[].size


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for num, _, v in _parse_fields(buf):
        if num == 2:
            size = 0
            for n2, _, v2 in _parse_fields(v):
                if n2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
    return tuple(dims)


def _entry_proto(dtype: int, shape, shard: int, offset: int, size: int, crc_masked: int) -> bytes:
    """BundleEntryProto (tensor_bundle.proto): dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32)."""
    out = _field_varint(1, dtype) + _field_bytes(2, _shape_proto(shape))
    if shard:
        out += _field_varint(3, shard)
    if offset:
        out += _field_varint(4, offset)
    out += _field_varint(5, size) + _field_fixed32(6, crc_masked)
    return out


def _header_proto(num_shards: int) -> bytes:
    """BundleHeaderProto: num_shards=1, endianness=2 (LITTLE = 0, omitted), version=3 {producer=1}."""
    return _field_varint(1, num_shards) + _field_bytes(3, _field_varint(1, 1))


# ---------------------------------------------------------------------------------------------------------------
# leveldb-format table (tensorflow/core/lib/io/table*.cc): writer and reader
# ---------------------------------------------------------------------------------------------------------------
TABLE_MAGIC = 0xDB4775248B80FB57
_BLOCK_SIZE = 4096        # table::Options::block_size default (the data is flushed once a block grows past it)
_RESTART_INTERVAL = 16    # table::Options::block_restart_interval default


class _BlockBuilder:
    def __init__(self, restart_interval=_RESTART_INTERVAL):
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last_key = b""
        self.interval = restart_interval

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count < self.interval:
            m = min(len(key), len(self.last_key))
            while shared < m and key[shared] == self.last_key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last_key = key
        self.count += 1

    def size_estimate(self) -> int:
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self) -> bool:
        return not self.buf

    def finish(self) -> bytes:
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _block_handle(offset: int, size: int) -> bytes:
    return _varint(offset) + _varint(size)


def write_table(path: str, items: Iterable[Tuple[bytes, bytes]]) -> None:
    """items must be sorted by key (bytewise).  No compression (as BundleWriter configures its table)."""
    out = bytearray()

    def emit_block(contents: bytes) -> bytes:
        handle = _block_handle(len(out), len(contents))
        trailer_type = b"\x00"  # kNoCompression
        crc = mask_crc(crc32c(trailer_type, crc32c(contents)))
        out.extend(contents + trailer_type + struct.pack("<I", crc))
        return handle

    data, index = _BlockBuilder(), _BlockBuilder(restart_interval=1)
    pending: Optional[Tuple[bytes, bytes]] = None  # (last key of the finished block, its handle)
    prev = None
    for key, value in items:
        if prev is not None and key <= prev:
            raise ValueError("table keys must be strictly increasing")
        prev = key
        if pending is not None:
            index.add(pending[0], pending[1])  # the last key of a block is a valid separator for it
            pending = None
        data.add(key, value)
        if data.size_estimate() >= _BLOCK_SIZE:
            pending = (data.last_key, emit_block(data.finish()))
            data = _BlockBuilder()
    if not data.empty():
        pending = (data.last_key, emit_block(data.finish()))
    if pending is not None:
        index.add(pending[0], pending[1])
    meta_handle = emit_block(_BlockBuilder().finish())  # empty metaindex block
    index_handle = emit_block(index.finish())
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(out)


def _read_block(buf: bytes, offset: int, size: int, verify: bool) -> bytes:
    contents = buf[offset:offset + size]
    btype = buf[offset + size]
    if verify:
        stored = struct.unpack_from("<I", buf, offset + size + 1)[0]
        if unmask_crc(stored) != crc32c(buf[offset + size:offset + size + 1], crc32c(contents)):
            raise ValueError("table block checksum mismatch")
    if btype != 0:
        raise NotImplementedError("compressed table blocks (snappy) are not supported; TF writes bundle indexes uncompressed")
    return contents


def _iter_block(block: bytes):
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path: str, verify: bool = True) -> List[Tuple[bytes, bytes]]:
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError(f"{path}: not a leveldb-format table (bad magic)")
    footer = buf[len(buf) - 48:]
    pos = 0
    _, pos = _read_varint(footer, pos); _, pos = _read_varint(footer, pos)  # metaindex handle
    ioff, pos = _read_varint(footer, pos); isize, pos = _read_varint(footer, pos)
    out = []
    for _, handle in _iter_block(_read_block(buf, ioff, isize, verify)):
        off, p2 = _read_varint(handle, 0)
        size, _ = _read_varint(handle, p2)
        out.extend(_iter_block(_read_block(buf, off, size, verify)))
    return out


# ---------------------------------------------------------------------------------------------------------------
# TensorBundle
# ---------------------------------------------------------------------------------------------------------------
def _encode_string_tensor(strings: Sequence[bytes]) -> Tuple[bytes, int]:
    """-> (bytes on disk, un-masked crc as BundleWriter accumulates it)."""
    lengths = b"".join(_varint(len(s)) for s in strings)
    crc = 0
    for s in strings:
        crc = crc32c(struct.pack("<I", len(s)) if len(s) <= 0xFFFFFFFF else struct.pack("<Q", len(s)), crc)
    cks = struct.pack("<I", mask_crc(crc))
    crc = crc32c(cks, crc)
    for s in strings:
        crc = crc32c(s, crc)
    return lengths + cks + b"".join(strings), crc


def _decode_string_tensor(raw: bytes, n: int) -> List[bytes]:
    pos, lens = 0, []
    for _ in range(n):
        l, pos = _read_varint(raw, pos)
        lens.append(l)
    pos += 4
    out = []
    for l in lens:
        out.append(bytes(raw[pos:pos + l])); pos += l
    return out


def write_bundle(prefix: str, tensors: Dict[str, object]) -> None:
    """tensors: key -> numpy array, or ``bytes`` (a scalar DT_STRING tensor).  One data shard."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)) or ".", exist_ok=True)
    entries = []
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for key in sorted(tensors, key=lambda k: k.encode()):
            v = tensors[key]
            if isinstance(v, (bytes, bytearray)):
                raw, crc = _encode_string_tensor([bytes(v)])
                dtype, shape = DT["string"], ()
            else:
                a = np.asarray(v)
                shape0 = a.shape  # (ascontiguousarray would turn a 0-d scalar into shape (1,))
                a = np.ascontiguousarray(a).reshape(shape0)
                if a.dtype.byteorder == ">":
                    a = a.astype(a.dtype.newbyteorder("<"))
                if a.dtype.name not in DT:
                    raise TypeError(f"{key}: dtype {a.dtype} has no TensorFlow DataType here")
                raw, crc, dtype, shape = a.tobytes(), crc32c(a), DT[a.dtype.name], a.shape
            f.write(raw)
            entries.append((key.encode(), _entry_proto(dtype, shape, 0, offset, len(raw), mask_crc(crc))))
            offset += len(raw)
    write_table(prefix + ".index", [(b"", _header_proto(1))] + entries)


class BundleEntry:
    __slots__ = ("dtype", "shape", "shard", "offset", "size", "crc")

    def __init__(self, proto: bytes):
        self.dtype, self.shape, self.shard, self.offset, self.size, self.crc = 0, (), 0, 0, 0, None
        for num, _, v in _parse_fields(proto):
            if num == 1: self.dtype = v
            elif num == 2: self.shape = _parse_shape(v)
            elif num == 3: self.shard = v
            elif num == 4: self.offset = v
            elif num == 5: self.size = v
            elif num == 6: self.crc = v
            elif num == 7: raise NotImplementedError("sliced (partitioned) variables are not supported")


def read_bundle_index(prefix: str) -> Tuple[dict, Dict[str, BundleEntry]]:
    items = read_table(prefix + ".index")
    if not items or items[0][0] != b"":
        raise ValueError("bundle index has no header entry")
    header = dict(num_shards=1, endianness=0, producer=0)
    for num, _, v in _parse_fields(items[0][1]):
        if num == 1: header["num_shards"] = v
        elif num == 2: header["endianness"] = v
        elif num == 3:
            for n2, _, v2 in _parse_fields(v):
                if n2 == 1: header["producer"] = v2
    if header["endianness"] != 0:
        raise NotImplementedError("big-endian bundles are not supported")
    return header, {k.decode(): BundleEntry(v) for k, v in items[1:]}


def read_bundle(prefix: str, keys: Optional[Iterable[str]] = None, verify: bool = True) -> Dict[str, object]:
    """-> key -> numpy array (numeric tensors) / bytes or list of bytes (DT_STRING)."""
    header, entries = read_bundle_index(prefix)
    n = header["num_shards"]
    shards = {}
    out = {}
    for key in (entries if keys is None else keys):
        e = entries[key]
        if e.shard not in shards:
            shards[e.shard] = np.memmap(f"{prefix}.data-{e.shard:05d}-of-{n:05d}", dtype=np.uint8, mode="r")
        raw = shards[e.shard][e.offset:e.offset + e.size]
        count = int(np.prod(e.shape)) if e.shape else 1
        if e.dtype == DT["string"]:
            strs = _decode_string_tensor(bytes(raw), count)
            if verify and e.crc is not None and unmask_crc(e.crc) != _encode_string_tensor(strs)[1]:
                raise ValueError(f"{key}: checksum mismatch")
            out[key] = strs[0] if not e.shape else strs
            continue
        if verify and e.crc is not None and unmask_crc(e.crc) != crc32c(raw):
            raise ValueError(f"{key}: checksum mismatch")
        if e.dtype == DT_BFLOAT16:  # widen to float32
            a = (np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16).view(np.float32)
        elif e.dtype in DT_INV:
            a = np.frombuffer(raw, dtype=np.dtype(DT_INV[e.dtype]).newbyteorder("<"))
        else:
            raise NotImplementedError(f"{key}: TensorFlow DataType {e.dtype}")
        out[key] = np.array(a).reshape(e.shape)
    return out


# ---------------------------------------------------------------------------------------------------------------
# TrackableObjectGraph
# ---------------------------------------------------------------------------------------------------------------
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
VAR_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
SLOT_TAG = "/.OPTIMIZER_SLOT/"


class GraphNode:
    def __init__(self):
        self.children: List[Tuple[str, int]] = []          # (local_name, node_id)
        self.attributes: List[Tuple[str, str, str]] = []   # (name, full_name, checkpoint_key)
        self.slots: List[Tuple[int, str, int]] = []        # (original_variable_node_id, slot_name, slot_variable_node_id)


def build_object_graph(variable_keys: Sequence[str]) -> bytes:
    """TrackableObjectGraph for a set of checkpoint keys: one node per path component (root = node 0, children found by
    attribute name exactly as TF's restore walks them), a VARIABLE_VALUE attribute on every leaf, and SlotVariableReference
    entries on the optimizer nodes for ``<var>/.OPTIMIZER_SLOT/<optimizer>/<slot>`` keys."""
    nodes: List[GraphNode] = [GraphNode()]
    index: Dict[Tuple[int, str], int] = {}

    def child(parent: int, name: str) -> int:
        if (parent, name) not in index:
            nodes.append(GraphNode())
            index[(parent, name)] = len(nodes) - 1
            nodes[parent].children.append((name, len(nodes) - 1))
        return index[(parent, name)]

    def walk(path: str) -> int:
        n = 0
        for comp in path.split("/"):
            n = child(n, comp)
        return n

    plain = [k for k in variable_keys if SLOT_TAG not in k and k.endswith(VAR_SUFFIX)]
    for key in plain:
        path = key[: -len(VAR_SUFFIX)]
        nodes[walk(path)].attributes.append(("VARIABLE_VALUE", path.replace("/", "."), key))
    for key in variable_keys:
        if SLOT_TAG not in key:
            continue
        var_path, rest = key[: -len(VAR_SUFFIX)].split(SLOT_TAG)
        opt_path, slot = rest.rsplit("/", 1)
        nodes.append(GraphNode())  # slot variables hang off the optimizer's slot table, not off the attribute tree
        sid = len(nodes) - 1
        nodes[sid].attributes.append(("VARIABLE_VALUE", f"{var_path.replace('/', '.')}.{slot}", key))
        nodes[walk(opt_path)].slots.append((walk(var_path), slot, sid))
    out = b""
    for nd in nodes:
        body = b"".join(_field_bytes(1, _field_varint(1, nid) + _field_bytes(2, name.encode())) for name, nid in nd.children)
        body += b"".join(_field_bytes(2, _field_bytes(1, a.encode()) + _field_bytes(2, f.encode()) + _field_bytes(3, k.encode()))
                         for a, f, k in nd.attributes)
        body += b"".join(_field_bytes(3, _field_varint(1, o) + _field_bytes(2, s.encode()) + _field_varint(3, v))
                         for o, s, v in nd.slots)
        out += _field_bytes(1, body)
    return out


def parse_object_graph(buf: bytes) -> List[GraphNode]:
    nodes = []
    for num, _, v in _parse_fields(buf):
        if num != 1:
            continue
        nd = GraphNode()
        for n2, _, v2 in _parse_fields(v):
            f = {a: b for a, _, b in _parse_fields(v2)} if isinstance(v2, bytes) else {}
            if n2 == 1:
                nd.children.append((f.get(2, b"").decode(), f.get(1, 0)))
            elif n2 == 2:
                nd.attributes.append((f.get(1, b"").decode(), f.get(2, b"").decode(), f.get(3, b"").decode()))
            elif n2 == 3:
                nd.slots.append((f.get(1, 0), f.get(2, b"").decode(), f.get(3, 0)))
        nodes.append(nd)
    return nodes


def graph_paths(nodes: List[GraphNode]) -> Dict[str, str]:
    """attribute path (as TF's restore reaches it from the root) -> checkpoint key, for every variable in the graph.
    Works for graphs written by TF too, where keys need not equal the shortest attribute path."""
    out, seen, queue = {}, {0}, [(0, "")]
    while queue:
        nid, path = queue.pop(0)
        for name, _, key in nodes[nid].attributes:
            if name == "VARIABLE_VALUE":
                out.setdefault(path, key)
        for name, cid in nodes[nid].children:
            if cid not in seen and cid < len(nodes):
                seen.add(cid)
                queue.append((cid, f"{path}/{name}" if path else name))
    return out


# ---------------------------------------------------------------------------------------------------------------
# the reference's checkpoint (train.py:94-108) <-> the trainer state of build_trainer_state
# ---------------------------------------------------------------------------------------------------------------
def _np(t):
    return t.detach().cpu().numpy()


def _module_tensors(prefix: str, module) -> Dict[str, np.ndarray]:
    return {f"{prefix}/{n.replace('.', '/')}{VAR_SUFFIX}": _np(t) for n, t in module.state_dict().items()}


def _optimizer_tensors(name: str, opt, owners: Sequence[Tuple[str, object, int]]) -> Dict[str, np.ndarray]:
    """Keras Adam: hyper-parameter variables + iter, slots m / v per trained variable.  owners: (model prefix,
    FlatParams, offset of the optimiser's flat slice inside that FlatParams.flat)."""
    out = {f"{name}/iter{VAR_SUFFIX}": np.array(opt.iterations, dtype=np.int64),
           f"{name}/learning_rate{VAR_SUFFIX}": np.array(opt.lr, dtype=np.float32),
           f"{name}/beta_1{VAR_SUFFIX}": np.array(opt.beta1, dtype=np.float32),
           f"{name}/beta_2{VAR_SUFFIX}": np.array(opt.beta2, dtype=np.float32),
           f"{name}/decay{VAR_SUFFIX}": np.array(0.0, dtype=np.float32)}
    m, v = _np(opt.m), _np(opt.v)
    for prefix, flat, begin in owners:
        for pname, p, off in zip(flat.names, flat.params, flat.offsets):
            lo = off - begin
            if 0 <= lo and lo + p.numel() <= m.size:
                var = f"{prefix}/{pname.replace('.', '/')}"
                out[f"{var}{SLOT_TAG}{name}/m{VAR_SUFFIX}"] = m[lo:lo + p.numel()].reshape(tuple(p.shape))
                out[f"{var}{SLOT_TAG}{name}/v{VAR_SUFFIX}"] = v[lo:lo + p.numel()].reshape(tuple(p.shape))
    return out


def _optimizer_ranges(state):
    gf, df = state["generator"]._flat, state["discriminator"]._flat
    gb, _ = gf.range_of(("latent_encoder.", "synthesis."))
    ob, _ = gf.range_of(("synthesis.", "word_encoder."))
    return {"g_optimizer": [("generator", gf, gb)], "ocr_optimizer": [("generator", gf, ob)],
            "d_optimizer": [("discriminator", df, 0)]}


def trainer_state_tensors(state, save_counter: int = 1) -> Dict[str, object]:
    """Everything ``tf.train.Checkpoint(**ckpt_kwargs)`` of train.py:94-108 saves, keyed as TF keys it."""
    t: Dict[str, object] = {}
    for name in ("generator", "g_clone", "discriminator"):
        t.update(_module_tensors(name, state[name]))
    for name, owners in _optimizer_ranges(state).items():
        opt = state[name]
        # only parameters the optimiser actually owns (its flat slice) get slots
        owned = []
        for prefix, flat, begin in owners:
            owned.append((prefix, flat, begin))
        t.update({k: v for k, v in _optimizer_tensors(name, opt, owned).items()})
    t[f"pl_mean{VAR_SUFFIX}"] = np.array(float(state["pl_mean"]), dtype=np.float32)
    t[f"save_counter{VAR_SUFFIX}"] = np.array(save_counter, dtype=np.int64)
    t[OBJECT_GRAPH_KEY] = build_object_graph(sorted(t))
    return t


def save_checkpoint(ckpt_dir: str, state, step: Optional[int] = None, max_to_keep: int = 5) -> str:
    """``manager.save(checkpoint_number=step)`` (train.py:227-228, 259-261): writes ``ckpt-<step>.index/.data-*`` and the
    ``checkpoint`` state file; keeps the ``max_to_keep`` newest (model_loader.py:64-66).  Returns the prefix."""
    step = state["g_optimizer"].iterations if step is None else step
    os.makedirs(ckpt_dir, exist_ok=True)
    prefix = os.path.join(ckpt_dir, f"ckpt-{step}")
    existing = _list_checkpoints(ckpt_dir)
    counter = 0  # Checkpoint.save_counter: number of saves so far, carried from the newest checkpoint on disk
    if existing:
        key = f"save_counter{VAR_SUFFIX}"
        try:
            counter = int(read_bundle(os.path.join(ckpt_dir, existing[-1]), [key])[key])
        except (KeyError, ValueError, OSError):
            counter = len(existing)
    write_bundle(prefix, trainer_state_tensors(state, save_counter=counter + 1))
    names = [n for n in existing if n != f"ckpt-{step}"] + [f"ckpt-{step}"]
    for old in names[:-max_to_keep] if max_to_keep else []:
        for f in os.listdir(ckpt_dir):
            if f.startswith(old + ".index") or f.startswith(old + ".data-"):
                os.remove(os.path.join(ckpt_dir, f))
    names = names[-max_to_keep:] if max_to_keep else names
    with open(os.path.join(ckpt_dir, "checkpoint"), "w") as f:  # CheckpointState text proto
        f.write(f'model_checkpoint_path: "{names[-1]}"\n')
        for n in names:
            f.write(f'all_model_checkpoint_paths: "{n}"\n')
    return prefix


def _list_checkpoints(ckpt_dir: str) -> List[str]:
    if not os.path.isdir(ckpt_dir):
        return []
    names = {f[: -len(".index")] for f in os.listdir(ckpt_dir) if f.startswith("ckpt-") and f.endswith(".index")}
    return sorted(names, key=lambda n: int(n.split("-")[1]) if n.split("-")[1].isdigit() else -1)


def latest_checkpoint(ckpt_dir: str) -> Optional[str]:
    """``manager.latest_checkpoint``: the ``checkpoint`` state file if present, else the highest step on disk."""
    state_file = os.path.join(ckpt_dir, "checkpoint")
    if os.path.exists(state_file):
        for line in open(state_file):
            if line.startswith("model_checkpoint_path:"):
                return os.path.join(ckpt_dir, line.split('"')[1])
    names = _list_checkpoints(ckpt_dir)
    return os.path.join(ckpt_dir, names[-1]) if names else None


def load_checkpoint(prefix: str, state, expect_partial: bool = False, verify: bool = True) -> dict:
    """``ckpt.restore(prefix)`` (model_loader.py:57-81) into the objects of ``state`` (any subset of generator, g_clone,
    discriminator, g_optimizer, ocr_optimizer, d_optimizer, pl_mean -- inference passes g_clone only with
    ``expect_partial=True``, infer.py / validation).  Variables are matched through the checkpoint's OBJECT GRAPH
    (attribute paths from the root), so a checkpoint written by TensorFlow restores even where its keys differ from the
    plain attribute path.  Returns {"restored": [...], "missing": [...], "unused": [...]}."""
    import torch
    header, entries = read_bundle_index(prefix)
    graph = []
    if OBJECT_GRAPH_KEY in entries:
        graph = parse_object_graph(read_bundle(prefix, [OBJECT_GRAPH_KEY], verify)[OBJECT_GRAPH_KEY])
    by_path = graph_paths(graph) if graph else {}
    for k in entries:  # keys that follow the naming convention are reachable even without / beside the graph
        if k.endswith(VAR_SUFFIX) and SLOT_TAG not in k:
            by_path.setdefault(k[: -len(VAR_SUFFIX)], k)
    wanted: Dict[str, object] = {}
    for name in ("generator", "g_clone", "discriminator"):
        if name in state and state[name] is not None:
            for n, tns in state[name].state_dict().items():
                wanted[f"{name}/{n.replace('.', '/')}"] = tns
    restored, missing = [], []
    keys_needed = {p: by_path[p] for p in wanted if p in by_path}
    data = read_bundle(prefix, sorted(set(keys_needed.values())), verify)
    with torch.no_grad():
        for path, tns in wanted.items():
            if path not in keys_needed:
                missing.append(path)
                continue
            a = data[keys_needed[path]]
            if tuple(a.shape) != tuple(tns.shape):
                raise ValueError(f"{path}: checkpoint shape {tuple(a.shape)} != model shape {tuple(tns.shape)}")
            tns.copy_(torch.from_numpy(np.ascontiguousarray(a)).reshape(tuple(a.shape)).to(tns.dtype))
            restored.append(path)
        if "pl_mean" in state and state["pl_mean"] is not None and "pl_mean" in by_path:
            state["pl_mean"].copy_(torch.tensor(float(read_bundle(prefix, [by_path["pl_mean"]], verify)[by_path["pl_mean"]])))
            restored.append("pl_mean")
        if all(k in state for k in ("generator", "discriminator")):
            for name, owners in _optimizer_ranges(state).items():
                if name not in state or state[name] is None:
                    continue
                opt = state[name]
                it_key = f"{name}/iter{VAR_SUFFIX}"
                if it_key not in entries:
                    missing.append(f"{name}/iter")
                    continue
                it = int(read_bundle(prefix, [it_key], verify)[it_key])
                m, v = opt.m.detach().cpu().numpy().copy(), opt.v.detach().cpu().numpy().copy()
                slot_keys = []
                for mprefix, flat, begin in owners:
                    for pname, p, off in zip(flat.names, flat.params, flat.offsets):
                        lo = off - begin
                        if 0 <= lo and lo + p.numel() <= m.size:
                            var = f"{mprefix}/{pname.replace('.', '/')}"
                            slot_keys.append((f"{var}{SLOT_TAG}{name}/m{VAR_SUFFIX}", f"{var}{SLOT_TAG}{name}/v{VAR_SUFFIX}",
                                              lo, p.numel()))
                have = [sk for sk in slot_keys if sk[0] in entries and sk[1] in entries]
                missing += [sk[0] for sk in slot_keys if sk not in have]
                sd = read_bundle(prefix, [k for sk in have for k in sk[:2]], verify)
                for km, kv, lo, n in have:
                    m[lo:lo + n] = sd[km].reshape(-1)
                    v[lo:lo + n] = sd[kv].reshape(-1)
                opt.m.copy_(torch.from_numpy(m)); opt.v.copy_(torch.from_numpy(v))
                opt.step.fill_(it)
                opt._iterations = it
                restored.append(name)
    used = set(keys_needed.values())
    unused = [k for k in entries if k not in used and k != OBJECT_GRAPH_KEY]
    if missing and not expect_partial:
        raise KeyError(f"checkpoint {prefix} lacks {len(missing)} values, e.g. {missing[:3]} (pass expect_partial=True to "
                       f"restore a subset, as infer.py does)")
    return dict(restored=restored, missing=missing, unused=unused, header=header)
