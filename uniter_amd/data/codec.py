"""Record encodings of the reference's databases (decode half).

The reference stores
  * one text example per key as  lz4.frame.compress(msgpack.dumps(example, use_bin_type=True))      (data/data.py:156-169),
  * one image per key either as the bytes of an `.npz` archive (`compress=True`, data/data.py:104-110) or as
    msgpack with the `msgpack_numpy` extension for arrays (`compress=False`, data/data.py:111-113).
`msgpack` is in this image; `lz4` and `msgpack_numpy` are not, so their two small published formats are implemented here:

  * LZ4 frame (lz4/lz4 doc/lz4_Frame_format.md v1.6.x, block format doc/lz4_Block_format.md): a full DEcoder (linked and
    independent blocks, stored blocks, optional block / content checksums and content size), and an ENcoder that writes
    stored (uncompressed) blocks — a valid frame every LZ4 implementation reads; record writing is prepro-time tooling,
    the training path only decodes.  `lz4.frame` is used instead whenever it can be imported.
  * msgpack_numpy (lebedov/msgpack-numpy 0.4.x): an ndarray travels as the map
    {b'nd': True, b'type': dtype.str, b'kind': b'', b'shape': shape, b'data': raw bytes}, a numpy scalar as
    {b'nd': False, b'type': dtype.str, b'data': raw bytes}.

Pinned by tests/test_data_records.py: frames against hand-assembled vectors from the format documents, the xxHash32 below
against the `xxhash` package, compressed blocks against an independent greedy compressor written from the block document.
"""
import io
import struct

import msgpack
import numpy as np

try:                                        # the real thing when present (not in this image)
    import lz4.frame as _lz4frame
except ImportError:                         # pragma: no cover - depends on the environment
    _lz4frame = None

_MAGIC = 0x184D2204
_P1, _P2, _P3, _P4, _P5 = 2654435761, 2246822519, 3266489917, 668265263, 374761393
_M32 = 0xFFFFFFFF


def _rotl(x, r):
    return ((x << r) | (x >> (32 - r))) & _M32


def xxh32(data, seed=0):
    """xxHash32 (Cyan4973/xxHash doc/xxhash_spec.md) — the LZ4 frame header / block / content checksums."""
    data = bytes(data)
    n = len(data)
    i = 0
    if n >= 16:
        v = [(seed + _P1 + _P2) & _M32, (seed + _P2) & _M32, seed & _M32, (seed - _P1) & _M32]
        limit = n - 16
        while i <= limit:
            lanes = struct.unpack_from('<4I', data, i)
            for k in range(4):
                v[k] = (_rotl((v[k] + lanes[k] * _P2) & _M32, 13) * _P1) & _M32
            i += 16
        h = (_rotl(v[0], 1) + _rotl(v[1], 7) + _rotl(v[2], 12) + _rotl(v[3], 18)) & _M32
    else:
        h = (seed + _P5) & _M32
    h = (h + n) & _M32
    while i + 4 <= n:
        h = (_rotl((h + struct.unpack_from('<I', data, i)[0] * _P3) & _M32, 17) * _P4) & _M32
        i += 4
    while i < n:
        h = (_rotl((h + data[i] * _P5) & _M32, 11) * _P1) & _M32
        i += 1
    h ^= h >> 15
    h = (h * _P2) & _M32
    h ^= h >> 13
    h = (h * _P3) & _M32
    h ^= h >> 16
    return h


def lz4_block_decode(src, out):
    """Appends the decoded bytes of one compressed block to the bytearray `out`; matches may reach back into what `out`
    already holds (linked blocks).  Raises ValueError on malformed input."""
    src = memoryview(src)
    n = len(src)
    i = 0
    while i < n:
        token = src[i]
        i += 1
        lit = token >> 4
        if lit == 15:
            while True:
                if i >= n:
                    raise ValueError("lz4: truncated literal length")
                b = src[i]
                i += 1
                lit += b
                if b != 255:
                    break
        if i + lit > n:
            raise ValueError("lz4: literals run past the block")
        out += src[i:i + lit]
        i += lit
        if i >= n:                           # the last sequence has no match part
            break
        if i + 2 > n:
            raise ValueError("lz4: truncated match offset")
        offset = src[i] | (src[i + 1] << 8)
        i += 2
        if offset == 0 or offset > len(out):
            raise ValueError("lz4: match offset outside the window")
        mlen = token & 15
        if mlen == 15:
            while True:
                if i >= n:
                    raise ValueError("lz4: truncated match length")
                b = src[i]
                i += 1
                mlen += b
                if b != 255:
                    break
        mlen += 4
        start = len(out) - offset
        if offset >= mlen:
            out += out[start:start + mlen]
        else:                                # overlapping copy: the pattern of `offset` bytes repeats
            pattern = bytes(out[start:])
            reps, rest = divmod(mlen, offset)
            out += pattern * reps + pattern[:rest]
    return out


def lz4_frame_decode(data, verify=True):
    """bytes of ONE LZ4 frame -> the content (what `lz4.frame.decompress` returns)."""
    if _lz4frame is not None:
        return _lz4frame.decompress(bytes(data))
    data = memoryview(data).cast('B') if not isinstance(data, (bytes, bytearray)) else memoryview(data)
    if len(data) < 7 or struct.unpack_from('<I', data, 0)[0] != _MAGIC:
        raise ValueError("lz4: not an LZ4 frame")
    flg, bd = data[4], data[5]
    if (flg >> 6) != 1:
        raise ValueError("lz4: unsupported frame version")
    if flg & 0x02 or bd & 0x8F:
        raise ValueError("lz4: reserved bits set")
    independent, block_sum, has_size, content_sum, has_dict = flg & 0x20, flg & 0x10, flg & 0x08, flg & 0x04, flg & 0x01
    pos = 6
    content_size = None
    if has_size:
        content_size = struct.unpack_from('<Q', data, pos)[0]
        pos += 8
    if has_dict:
        raise ValueError("lz4: dictionary frames are not supported")
    if verify and ((xxh32(data[4:pos]) >> 8) & 0xFF) != data[pos]:
        raise ValueError("lz4: header checksum mismatch")
    pos += 1
    out = bytearray()
    while True:
        if pos + 4 > len(data):
            raise ValueError("lz4: truncated frame")
        size = struct.unpack_from('<I', data, pos)[0]
        pos += 4
        if size == 0:                        # EndMark
            break
        stored = bool(size & 0x80000000)
        size &= 0x7FFFFFFF
        block = data[pos:pos + size]
        if len(block) != size:
            raise ValueError("lz4: truncated block")
        pos += size
        if block_sum:
            if verify and struct.unpack_from('<I', data, pos)[0] != xxh32(block):
                raise ValueError("lz4: block checksum mismatch")
            pos += 4
        if stored:
            out += block
        elif independent:
            piece = lz4_block_decode(block, bytearray())
            out += piece
        else:
            lz4_block_decode(block, out)
    if content_sum:
        if verify and struct.unpack_from('<I', data, pos)[0] != xxh32(out):
            raise ValueError("lz4: content checksum mismatch")
    if content_size is not None and content_size != len(out):
        raise ValueError("lz4: content size mismatch")
    return bytes(out)


def lz4_frame_encode(content):
    """A valid LZ4 frame holding `content` in stored blocks (content size recorded, as `lz4.frame.compress` does by
    default); with `lz4.frame` importable the real compressor runs instead."""
    if _lz4frame is not None:
        return _lz4frame.compress(bytes(content))
    content = bytes(content)
    desc = bytes([0x40 | 0x20 | 0x08, 0x70]) + struct.pack('<Q', len(content))     # v1, independent blocks, content size; 4 MiB blocks
    parts = [struct.pack('<I', _MAGIC), desc, bytes([(xxh32(desc) >> 8) & 0xFF])]
    step = 4 << 20
    for start in range(0, len(content), step):
        block = content[start:start + step]
        parts.append(struct.pack('<I', len(block) | 0x80000000))
        parts.append(block)
    parts.append(struct.pack('<I', 0))
    return b''.join(parts)


# ---- msgpack with numpy arrays ------------------------------------------------------------------------------------------
def _np_default(obj):
    if isinstance(obj, np.ndarray):
        if obj.dtype.kind in 'OV':
            raise TypeError("object / structured arrays are not stored in the feature records")
        return {b'nd': True, b'type': obj.dtype.str, b'kind': b'', b'shape': list(obj.shape),
                b'data': np.ascontiguousarray(obj).tobytes()}
    if isinstance(obj, (np.bool_, np.number)):
        return {b'nd': False, b'type': obj.dtype.str, b'data': obj.tobytes()}
    raise TypeError("cannot serialise %r" % type(obj))


def _np_hook(obj):
    nd = obj.get(b'nd', obj.get('nd')) if isinstance(obj, dict) else None
    if nd is None:
        return obj

    def field(name):
        return obj[name.encode()] if name.encode() in obj else obj[name]
    dtype = field('type')
    dtype = np.dtype(dtype.decode() if isinstance(dtype, bytes) else dtype)
    if nd:
        return np.frombuffer(field('data'), dtype=dtype).reshape(field('shape'))
    return np.frombuffer(field('data'), dtype=dtype)[0]


def packb(obj):
    """msgpack.dumps(obj, use_bin_type=True) with msgpack_numpy's array encoding."""
    return msgpack.packb(obj, default=_np_default, use_bin_type=True)


def unpackb(data):
    """msgpack.loads(data, raw=False) with msgpack_numpy's array decoding (arrays are read-only views of `data`)."""
    return msgpack.unpackb(bytes(data), object_hook=_np_hook, raw=False, strict_map_key=False)


# ---- the two record kinds ---------------------------------------------------------------------------------------------------
def encode_txt_record(example):
    return lz4_frame_encode(msgpack.packb(example, use_bin_type=True))


def decode_txt_record(blob):
    return msgpack.unpackb(lz4_frame_decode(blob), raw=False, strict_map_key=False)


def encode_img_record(arrays, compress=True):
    """arrays: dict name -> ndarray ('features', 'norm_bb', 'conf', 'soft_labels', ...)."""
    if not compress:
        return packb(dict(arrays))
    buf = io.BytesIO()
    np.savez_compressed(buf, **arrays)
    return buf.getvalue()


def decode_img_record(blob, compress=True, fields=None):
    """-> dict name -> ndarray; `fields` limits what is decoded from an npz archive (each member is inflated on access)."""
    if not compress:
        rec = unpackb(blob)
        return rec if fields is None else {k: rec[k] for k in fields}
    with io.BytesIO(bytes(blob)) as reader:
        # feature records are plain numeric arrays (features, norm_bb, conf, soft_labels): no pickle, so that a crafted feature
        # database cannot execute code in the loader workers (the reference's allow_pickle=True is not needed for its own files)
        archive = np.load(reader, allow_pickle=False)
        names = archive.files if fields is None else fields
        try:
            return {k: archive[k] for k in names}
        except ValueError as e:
            raise ValueError("decode_img_record: the record holds an object array (pickled data), which this loader refuses: %s" % e)
