"""Checkpoint files in the format the reference's saveModel / loadModel exchange (GANMF.py:309-339,
DisGANMF.py:264-266): a TensorFlow "V2 tensor bundle" written by tf.train.Saver(...).save(sess, prefix,
write_meta_graph=False, write_state=False).

Two files per prefix:
  <prefix>.data-00000-of-00001   raw little-endian tensors, concatenated in key-sorted order, no padding
  <prefix>.index                 a LevelDB-format table (uncompressed, restart interval 16) mapping
                                 ""    -> BundleHeaderProto {num_shards = 1, version {producer = 1}}
                                 name  -> BundleEntryProto  {dtype, shape, offset, size, masked crc32c}

TensorFlow itself is not a dependency of the reference's checked-in artefacts' *format*: the layout is restated
here from the surviving checkpoint (tests/golden/kat1_GANMF_item.index is that file) and the published
table/bundle formats; oracle/make_golden.py verifies in the build container that write_bundle() reproduces the
reference's .index and .data byte for byte.  Only DT_FLOAT tensors are needed (GANMF.py:108).
"""
import ctypes as C
import os
import struct

import numpy as np

_TABLE_MAGIC = 0xdb4775248b80fb57
_RESTART_INTERVAL = 16
_DT_FLOAT = 1
_MASK_DELTA = 0xa282ead8


def crc32c(data, crc=0):
    """CRC-32C of a bytes-like / ndarray (C implementation in libganmf_hip.so, include/ganmf_hip.h)."""
    from . import _lib as L
    lib = L.load_library()
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    if buf.size == 0:
        return crc
    return int(lib.ganmf_crc32c(C.c_uint32(crc), buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.size)))


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xffffffff


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


# ---- protobuf (only the two messages of tensor_bundle.proto) -----------------------------------
def _entry_proto(shape, offset, size, crc_masked):
    dims = b"".join(b"\x12" + _varint(len(d)) + d for d in (b"\x08" + _varint(int(n)) for n in shape))
    out = b"\x08" + _varint(_DT_FLOAT) + b"\x12" + _varint(len(dims)) + dims
    if offset:
        out += b"\x20" + _varint(offset)
    if size:
        out += b"\x28" + _varint(size)
    return out + b"\x35" + struct.pack("<I", crc_masked)


def _parse_fields(buf):
    """{field number: [values]} for varint, 32-bit and length-delimited fields"""
    pos, out = 0, {}
    while pos < len(buf):
        tag, pos = _read_varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wire == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        else:
            raise ValueError("tf_bundle: unsupported protobuf wire type %d" % wire)
        out.setdefault(field, []).append(v)
    return out


def _parse_entry(buf):
    f = _parse_fields(buf)
    shape = []
    for shp in f.get(2, []):
        for dim in _parse_fields(shp).get(2, []):
            shape.append(_parse_fields(dim).get(1, [0])[0])
    return {"dtype": f.get(1, [0])[0], "shape": tuple(shape), "shard_id": f.get(3, [0])[0],
            "offset": f.get(4, [0])[0], "size": f.get(5, [0])[0], "crc32c": f.get(6, [0])[0]}


# ---- LevelDB table ------------------------------------------------------------------------------
def _block(entries):
    """prefix-compressed key/value block with restart points every 16 entries"""
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        if i % _RESTART_INTERVAL == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            while shared < min(len(last), len(k)) and last[shared] == k[shared]:
                shared += 1
        out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _with_trailer(block):
    return block + b"\x00" + struct.pack("<I", _mask(crc32c(block + b"\x00")))


def _short_successor(key):
    for i, b in enumerate(key):
        if b != 0xff:
            return key[:i] + bytes([b + 1])
    return key


def _parse_block(buf):
    n_restarts = struct.unpack_from("<I", buf, len(buf) - 4)[0]
    end = len(buf) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _read_varint(buf, pos)
        non_shared, pos = _read_varint(buf, pos)
        vlen, pos = _read_varint(buf, pos)
        key = key[:shared] + bytes(buf[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(buf[pos:pos + vlen])))
        pos += vlen
    return out


def _read_block(buf, offset, size):
    block, trailer = buf[offset:offset + size], buf[offset + size:offset + size + 5]
    if len(trailer) != 5 or trailer[0] != 0:
        raise ValueError("tf_bundle: compressed or truncated table block")
    if struct.unpack("<I", trailer[1:])[0] != _mask(crc32c(bytes(block) + b"\x00")):
        raise ValueError("tf_bundle: table block checksum mismatch")
    return _parse_block(block)


def read_index(path):
    """-> (header fields, {tensor name: entry dict}) from a .index file"""
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != _TABLE_MAGIC:
        raise ValueError("tf_bundle: %s is not a tensor-bundle index" % path)
    footer = buf[-48:]
    pos = 0
    _, pos = _read_varint(footer, pos)       # metaindex handle (unused)
    _, pos = _read_varint(footer, pos)
    ioff, pos = _read_varint(footer, pos)
    isize, pos = _read_varint(footer, pos)
    header, entries = None, {}
    for _, handle in _read_block(buf, ioff, isize):
        off, p = _read_varint(handle, 0)
        size, p = _read_varint(handle, p)
        for k, v in _read_block(buf, off, size):
            if k == b"":
                header = _parse_fields(v)
            else:
                entries[k.decode()] = _parse_entry(v)
    if header is None:
        raise ValueError("tf_bundle: index has no header entry")
    if header.get(1, [0])[0] != 1 or header.get(2, [0])[0] != 0:
        raise ValueError("tf_bundle: only single-shard little-endian bundles are supported")
    return header, entries


def read_bundle(prefix, verify=True):
    """{name: float32 ndarray} of every tensor under <prefix>.index / <prefix>.data-00000-of-00001"""
    _, entries = read_index(prefix + ".index")
    data = np.fromfile(prefix + ".data-00000-of-00001", dtype=np.uint8)
    out = {}
    for name, e in entries.items():
        if e["dtype"] != _DT_FLOAT or e["shard_id"] != 0:
            raise ValueError("tf_bundle: tensor %s is not a float32 tensor in shard 0" % name)
        n = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if e["size"] != 4 * n or e["offset"] + e["size"] > data.size:
            raise ValueError("tf_bundle: tensor %s has inconsistent size/offset" % name)
        raw = data[e["offset"]:e["offset"] + e["size"]]
        if verify and _mask(crc32c(raw)) != e["crc32c"]:
            raise ValueError("tf_bundle: checksum mismatch in tensor %s" % name)
        out[name] = raw.view("<f4").reshape(e["shape"]).copy()
    return out


def write_bundle(prefix, tensors):
    """tf.train.Saver(var_list).save(sess, prefix, write_meta_graph=False, write_state=False) for float32
    variables: `tensors` maps variable names (without ':0') to arrays."""
    items = sorted((k.encode(), np.ascontiguousarray(v, dtype="<f4")) for k, v in tensors.items())
    entries = [(b"", b"\x08\x01\x1a\x02\x08\x01")]      # num_shards = 1, version.producer = 1
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for k, a in items:
            f.write(a.tobytes())
            entries.append((k, _entry_proto(a.shape, offset, a.nbytes, _mask(crc32c(a)))))
            offset += a.nbytes
    data_block = _block(entries)
    out = bytearray(_with_trailer(data_block))
    meta_off = len(out)
    meta = _block([])
    out += _with_trailer(meta)
    index_off = len(out)
    index = _block([(_short_successor(entries[-1][0]), _varint(0) + _varint(len(data_block)))])
    out += _with_trailer(index)
    footer = _varint(meta_off) + _varint(len(meta)) + _varint(index_off) + _varint(len(index))
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _TABLE_MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
