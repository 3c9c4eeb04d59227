"""Reader for TensorFlow "bundle" checkpoints (the format tf.train.Saver writes and the reference's
pre-trained model ships in: /root/reference/trained_models/english/acoustic/acousticmodel.ckpt.*,
reference save/restore at models/AcousticModel.py:483-527) -- SURVEY.md 8f-2.

A bundle is `<prefix>.index` + `<prefix>.data-0000N-of-0000M`.  The index is an uncompressed
leveldb-style table: data blocks of prefix-compressed (key, value) entries with a restart array,
an index block, and a 48-byte footer ending in the magic 0xdb4775248b80fb57.  Keys are variable
names (the empty key holds the BundleHeaderProto); values are BundleEntryProto messages
{1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset, 5: size, 6: crc32c}.  Only what the
acoustic model needs is decoded (float32 / int32 tensors, little-endian, row-major).
"""
import os
import struct

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 3: np.int32, 9: np.int64, 2: np.float64}


def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _block(data, offset, size):
    """Entries of one table block -> list of (key bytes, value bytes)."""
    blk = data[offset:offset + size]
    n_restarts = struct.unpack_from("<I", blk, len(blk) - 4)[0]
    limit = len(blk) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < limit:
        shared, pos = _varint(blk, pos)
        non_shared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + bytes(blk[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(blk[pos:pos + vlen])))
        pos += vlen
    return out


def _proto_fields(buf):
    """Minimal protobuf wire decoder -> list of (field number, wire type, value)."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((field, wt, v))
    return out


def _entry(value):
    e = {"dtype": 0, "shape": [], "shard": 0, "offset": 0, "size": 0, "crc32c": None}
    for field, _, v in _proto_fields(value):
        if field == 1:
            e["dtype"] = v
        elif field == 2:                       # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }
            for f2, _, dim in _proto_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, s in _proto_fields(dim):
                        if f3 == 1:
                            size = s
                    e["shape"].append(size)
        elif field == 3:
            e["shard"] = v
        elif field == 4:
            e["offset"] = v
        elif field == 5:
            e["size"] = v
        elif field == 6:
            e["crc32c"] = v
    return e


def read_index(index_path):
    """{variable name: {dtype, shape, shard, offset, size, crc32c}} plus '' -> header info."""
    data = open(index_path, "rb").read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != _MAGIC:
        raise ValueError("%s is not a TensorFlow bundle index (bad table magic)" % index_path)
    footer = data[-48:]
    pos = 0
    _, pos = _varint(footer, pos)              # metaindex handle
    _, pos = _varint(footer, pos)
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)
    entries = {}
    for _, handle in _block(data, idx_off, idx_size):
        off, p = _varint(handle, 0)
        size, p = _varint(handle, p)
        for key, value in _block(data, off, size):
            name = key.decode("utf-8")
            if name == "":
                hdr = {f: v for f, _, v in _proto_fields(value)}
                entries[""] = {"num_shards": hdr.get(1, 1)}
            else:
                entries[name] = _entry(value)
    return entries


def read_bundle(prefix):
    """Load every float32/int32 tensor of the bundle `<prefix>.index` / `<prefix>.data-*`."""
    entries = read_index(prefix + ".index")
    n_shards = entries.get("", {}).get("num_shards", 1)
    out = {}
    for name, e in entries.items():
        if name == "":
            continue
        if e["dtype"] not in _DTYPES:
            raise ValueError("tensor %s: unsupported dtype enum %d" % (name, e["dtype"]))
        shard = "%s.data-%05d-of-%05d" % (prefix, e["shard"], n_shards)
        if os.path.getsize(shard) < e["offset"] + e["size"]:
            raise IOError("%s is %d bytes but %s needs [%d, %d) -- a git-LFS pointer instead of the blob?"
                          % (shard, os.path.getsize(shard), name, e["offset"], e["offset"] + e["size"]))
        with open(shard, "rb") as fh:
            fh.seek(e["offset"])
            raw = fh.read(e["size"])
        # per-tensor masked CRC32C of the raw bytes (BundleEntryProto.crc32c): a truncated / bit-rotted shard must
        # not load silently
        if e["crc32c"] is not None and _mask(crc32c(raw)) != e["crc32c"]:
            raise IOError("%s: tensor %s fails its crc32c check (stored %08x, computed %08x)"
                          % (shard, name, e["crc32c"], _mask(crc32c(raw))))
        out[name] = np.frombuffer(raw, dtype=np.dtype(_DTYPES[e["dtype"]]).newbyteorder("<")).reshape(e["shape"]).copy()
    return out


# ------------------------------------------------------------------------------- writer
def crc32c(data, crc=0):
    """CRC32C through the native helper of libamdspeech (host code, no GPU involved)."""
    import ctypes
    from . import lib as _lib
    buf = bytes(data)
    return int(_lib.load().amdspeech_crc32c(ctypes.c_char_p(buf), len(buf), crc))


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(num, wt, payload):
    tag = _put_varint((num << 3) | wt)
    if wt == 0:
        return tag + _put_varint(payload)
    if wt == 2:
        return tag + _put_varint(len(payload)) + payload
    if wt == 5:
        return tag + struct.pack("<I", payload)
    raise ValueError(wt)


def _table_block(items, restart_interval=16):
    """(key, value) pairs (sorted) -> block bytes incl. restart array; keys are stored unshared."""
    body, restarts = bytearray(), []
    for n, (key, value) in enumerate(items):
        if n % restart_interval == 0:
            restarts.append(len(body))
        body += _put_varint(0) + _put_varint(len(key)) + _put_varint(len(value)) + key + value
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def _with_trailer(block):
    return block + b"\x00" + struct.pack("<I", _mask(crc32c(block + b"\x00")))


_DTYPE_ENUM = {np.dtype(np.float32): 1, np.dtype(np.int32): 3, np.dtype(np.int64): 9, np.dtype(np.float64): 2}


def write_bundle(prefix, tensors):
    """Write `<prefix>.index` + `<prefix>.data-00000-of-00001` with the layout tf.train.Saver produces
    (single shard, keys sorted, little-endian row-major data, masked CRC32C per tensor and per block)."""
    names = sorted(tensors)
    entries, offset = [], 0
    with open(prefix + ".data-00000-of-00001", "wb") as fh:
        for name in names:
            arr = np.asarray(tensors[name])            # (ascontiguousarray would turn scalars into [1])
            arr = arr if arr.flags.c_contiguous else arr.copy(order="C")
            if arr.dtype not in _DTYPE_ENUM:
                raise ValueError("tensor %s: unsupported dtype %s" % (name, arr.dtype))
            raw = arr.astype(arr.dtype.newbyteorder("<"), copy=False).tobytes()
            fh.write(raw)
            shape = b"".join(_field(2, 2, _field(1, 0, int(d))) for d in arr.shape)
            msg = _field(1, 0, _DTYPE_ENUM[arr.dtype]) + _field(2, 2, shape)
            if offset:
                msg += _field(4, 0, offset)
            msg += _field(5, 0, len(raw)) + _field(6, 5, _mask(crc32c(raw)))
            entries.append((name.encode("utf-8"), msg))
            offset += len(raw)
    header = _field(1, 0, 1) + _field(3, 2, _field(1, 0, 1))      # num_shards = 1, version.producer = 1
    data_block = _with_trailer(_table_block([(b"", header)] + entries))
    meta_block = _with_trailer(_table_block([]))
    data_size = len(data_block) - 5
    meta_off = len(data_block)
    index_off = meta_off + len(meta_block)
    last_key = entries[-1][0] if entries else b""
    index_block = _with_trailer(_table_block([(last_key, _put_varint(0) + _put_varint(data_size))]))
    footer = _put_varint(meta_off) + _put_varint(len(meta_block) - 5) + _put_varint(index_off) + _put_varint(len(index_block) - 5)
    footer = footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
    with open(prefix + ".index", "wb") as fh:
        fh.write(data_block + meta_block + index_block + footer)


def verify_index_checksums(index_path):
    """True if every table block of the index carries a valid masked CRC32C (what TensorFlow checks)."""
    data = open(index_path, "rb").read()
    footer = data[-48:]
    pos, handles = 0, []
    for _ in range(2):
        off, pos = _varint(footer, pos)
        size, pos = _varint(footer, pos)
        handles.append((off, size))
    idx_off, idx_size = handles[1]
    for _, handle in _block(data, idx_off, idx_size):
        off, p = _varint(handle, 0)
        size, p = _varint(handle, p)
        handles.append((off, size))
    for off, size in handles:
        stored = struct.unpack_from("<I", data, off + size + 1)[0]
        if _mask(crc32c(data[off:off + size + 1])) != stored:
            return False
    return True
