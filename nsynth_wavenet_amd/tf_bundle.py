"""Reader / writer for TensorFlow V2 checkpoints ("tensor bundles") without TensorFlow.

SURVEY section 8 row f2: the reference restores its generation graphs from TF checkpoints
`model.ckpt-N.{index,data-00000-of-00001}` (wavenet/fastgen.py:80-84, wavenet/parallelgen.py:29-41,
tools/make_eval_model.py:13-34) whose keys are `<variable>/ExponentialMovingAverage`.  TensorFlow
is not installed here, so this module restates the on-disk format:

  <prefix>.index   an SSTable (TensorFlow's port of the LevelDB table format), uncompressed:
                   data blocks of prefix-compressed (key, value) entries + restart array,
                   each block followed by a 1-byte compression type and a masked CRC32C;
                   an index block; a 48-byte footer (metaindex handle, index handle, padding,
                   magic 0xdb4775248b80fb57).  Key "" -> BundleHeaderProto, every other key
                   (tensor name) -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}.
  <prefix>.data-SSSSS-of-NNNNN   raw little-endian tensor bytes at [offset, offset+size).

Validated only against itself (writer <-> reader round trip and a hand-assembled table in
tests/test_tf_bundle.py): no TensorFlow-written file is available offline.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_LEN = 48
BLOCK_TRAILER = 5
MASK_DELTA = 0xa282ead8

# tensorflow DataType enum values -> numpy
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 4: np.uint8, 6: np.int8, 5: np.int16}
DTYPE_IDS = {np.dtype(v): k for k, v in DTYPES.items()}

# ---------------------------------------------------------------- crc32c (Castagnoli) ----
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t[i] = c
        _CRC_TABLE = t
    return _CRC_TABLE


_native = None


def _native_crc():
    """Host-side helper in libwnhip.so (wn_host.cpp); pure python is the fallback."""
    global _native
    if _native is None:
        _native = False
        try:
            import ctypes
            from . import _lib
            lib = ctypes.CDLL(_lib.LIB_PATH)
            lib.wn_crc32c.restype = ctypes.c_uint32
            lib.wn_crc32c.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32]
            _native = lib.wn_crc32c
        except (OSError, AttributeError):
            pass
    return _native


def crc32c(data, crc=0):
    data = bytes(data)
    fn = _native_crc() if len(data) > 256 else None
    if fn:
        return int(fn(data, len(data), crc))
    t = _crc_table()
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = int(t[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + MASK_DELTA) & 0xFFFFFFFF


# ---------------------------------------------------------------- varints / protobuf ----
def _get_varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf):
    """Flat protobuf parse: [(field, wire_type, value)] (value = int or bytes)."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type {}'.format(wt))
        out.append((field, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_entry(buf):
    """BundleEntryProto -> dict(dtype, shape, shard_id, offset, size, crc32c, sliced)."""
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': 0, 'sliced': False}
    for field, wt, v in _parse_proto(buf):
        if field == 1:
            e['dtype'] = v
        elif field == 2:                               # TensorShapeProto
            for f2, _, v2 in _parse_proto(v):
                if f2 == 2:                            # Dim
                    size = 0
                    for f3, _, v3 in _parse_proto(v2):
                        if f3 == 1:
                            size = _signed64(v3)
                    e['shape'].append(size)
        elif field == 3:
            e['shard_id'] = v
        elif field == 4:
            e['offset'] = _signed64(v)
        elif field == 5:
            e['size'] = _signed64(v)
        elif field == 6:
            e['crc32c'] = v
        elif field == 7:
            e['sliced'] = True
    return e


def _encode_entry(dtype_id, shape, shard_id, offset, size, crc):
    dims = b''.join(b'\x12' + _put_varint(len(d)) + d for d in (b'\x08' + _put_varint(s) for s in shape))
    out = b'\x08' + _put_varint(dtype_id)
    out += b'\x12' + _put_varint(len(dims)) + dims
    if shard_id:
        out += b'\x18' + _put_varint(shard_id)
    if offset:
        out += b'\x20' + _put_varint(offset)
    out += b'\x28' + _put_varint(size)
    out += b'\x35' + struct.pack('<I', crc)
    return out


# ---------------------------------------------------------------- table blocks ----
def _read_block(f_bytes, offset, size, verify=True):
    contents = f_bytes[offset:offset + size]
    ctype = f_bytes[offset + size]
    if verify:
        stored = struct.unpack_from('<I', f_bytes, offset + size + 1)[0]
        if mask_crc(crc32c(f_bytes[offset:offset + size + 1])) != stored:
            raise ValueError('index block checksum mismatch at offset {}'.format(offset))
    if ctype != 0:
        raise ValueError('compressed index blocks (type {}) are not supported'.format(ctype))
    return contents


def _block_entries(block):
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b''
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        value = bytes(block[pos:pos + vlen])
        pos += vlen
        yield key, value


def _block_handle(buf, pos):
    off, pos = _get_varint(buf, pos)
    size, pos = _get_varint(buf, pos)
    return off, size, pos


class BundleReader(object):
    """tf.train.load_checkpoint look-alike: get_variable_to_shape_map(), get_tensor(name)."""

    def __init__(self, prefix, verify_index=True):
        self.prefix = prefix
        with open(prefix + '.index', 'rb') as f:
            data = f.read()
        if len(data) < FOOTER_LEN or struct.unpack_from('<Q', data, len(data) - 8)[0] != TABLE_MAGIC:
            raise ValueError('{}.index is not a TensorFlow tensor-bundle index (bad magic)'.format(prefix))
        footer = data[len(data) - FOOTER_LEN:]
        _, _, pos = _block_handle(footer, 0)                       # metaindex (unused)
        ioff, isize, _ = _block_handle(footer, pos)
        self.entries = {}
        self.num_shards = 1
        index_block = _read_block(data, ioff, isize, verify_index)
        for _, handle in _block_entries(index_block):
            boff, bsize, _ = _block_handle(handle, 0)
            for key, value in _block_entries(_read_block(data, boff, bsize, verify_index)):
                if key == b'':
                    for field, _, v in _parse_proto(value):
                        if field == 1:
                            self.num_shards = v
                        elif field == 2 and v != 0:
                            raise ValueError('big-endian bundles are not supported')
                else:
                    self.entries[key.decode()] = _parse_entry(value)
        self._shards = {}

    def get_variable_to_shape_map(self):
        return {k: list(e['shape']) for k, e in self.entries.items()}

    def has_tensor(self, name):
        return name in self.entries

    def _shard(self, sid):
        if sid not in self._shards:
            path = '{}.data-{:05d}-of-{:05d}'.format(self.prefix, sid, self.num_shards)
            self._shards[sid] = np.memmap(path, dtype=np.uint8, mode='r')
        return self._shards[sid]

    def get_tensor(self, name, verify=False):
        e = self.entries[name]
        if e['sliced']:
            raise ValueError('{}: partitioned (sliced) variables are not supported'.format(name))
        if e['dtype'] not in DTYPES:
            raise ValueError('{}: unsupported dtype enum {}'.format(name, e['dtype']))
        raw = self._shard(e['shard_id'])[e['offset']:e['offset'] + e['size']]
        if verify and mask_crc(crc32c(raw)) != e['crc32c']:
            raise ValueError('{}: tensor checksum mismatch'.format(name))
        arr = np.frombuffer(bytes(raw), dtype=np.dtype(DTYPES[e['dtype']]).newbyteorder('<'))
        return arr.reshape(e['shape']).copy()


# ---------------------------------------------------------------- writer ----
def _build_block(items, restart_interval=16):
    out, restarts, last = bytearray(), [], b''
    for i, (key, value) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(key), len(last)) and key[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        out += key[shared:] + value
        last = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_bundle(prefix, tensors, block_size=4096):
    """Write {name: ndarray} as a single-shard V2 checkpoint (`<prefix>.index` + data file)."""
    data_path = '{}.data-00000-of-00001'.format(prefix)
    entries = []
    offset = 0
    with open(data_path, 'wb') as f:
        for name in sorted(tensors):
            arr = np.asarray(tensors[name])
            dt = np.dtype(arr.dtype)
            if dt not in DTYPE_IDS:
                raise ValueError('unsupported dtype {}'.format(dt))
            raw = arr.astype(dt.newbyteorder('<')).tobytes()
            f.write(raw)
            entries.append((name.encode(), _encode_entry(DTYPE_IDS[dt], arr.shape, 0, offset, len(raw),
                                                         mask_crc(crc32c(raw)))))
            offset += len(raw)
    header = b'\x08\x01' + b'\x1a\x02\x08\x01'          # num_shards=1, version{producer=1}
    items = [(b'', header)] + entries                  # "" sorts first
    out = bytearray()
    index_items = []

    def emit(block_items):
        block = _build_block(block_items)
        off = len(out)
        out.extend(block)
        out.append(0)                                  # kNoCompression
        out.extend(struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
        index_items.append((block_items[-1][0], _put_varint(off) + _put_varint(len(block))))

    cur, cur_size = [], 0
    for kv in items:
        cur.append(kv)
        cur_size += len(kv[0]) + len(kv[1]) + 3
        if cur_size >= block_size:
            emit(cur)
            cur, cur_size = [], 0
    if cur:
        emit(cur)
    # metaindex (empty) and index blocks
    meta = _build_block([])
    meta_off = len(out)
    out.extend(meta + b'\x00' + struct.pack('<I', mask_crc(crc32c(meta + b'\x00'))))
    index = _build_block(index_items, restart_interval=1)
    index_off = len(out)
    out.extend(index + b'\x00' + struct.pack('<I', mask_crc(crc32c(index + b'\x00'))))
    footer = _put_varint(meta_off) + _put_varint(len(meta)) + _put_varint(index_off) + _put_varint(len(index))
    footer += b'\x00' * (FOOTER_LEN - 8 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    out.extend(footer)
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out))
    return prefix


def is_bundle(prefix):
    return os.path.exists(prefix + '.index')
