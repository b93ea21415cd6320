"""Writer-independent TensorFlow V2 checkpoint ("tensor bundle") assembler for the tests.

Every byte is produced here from the format description -- the LevelDB table format (prefix-compressed key blocks
with restart arrays, block trailers with masked CRC32C, metaindex + index blocks, 48-byte footer) and
tensor_bundle.proto (BundleHeaderProto, BundleEntryProto) -- and never by nsynth_wavenet_amd.tf_bundle.write_bundle,
so that the reader is held to the format, not to its own writer.  (No TensorFlow-written file can exist in this
image: TensorFlow is absent and there is no network.)"""
import struct

import numpy as np

from nsynth_wavenet_amd import tf_bundle as tb

DTYPE = {np.dtype('<f4'): 1, np.dtype('<i8'): 9}


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def entry_proto(dtype, shape, shard, offset, size, crc, sliced=False):
    dims = b''.join(b'\x12' + varint(len(d)) + d for d in (b'\x08' + varint(n) for n in shape))
    out = b'\x08' + varint(dtype) + b'\x12' + varint(len(dims)) + dims
    if shard:
        out += b'\x18' + varint(shard)
    if offset:
        out += b'\x20' + varint(offset)
    out += b'\x28' + varint(size) + b'\x35' + struct.pack('<I', crc)
    if sliced:                       # repeated TensorSliceProto slices = 7: one slice with one extent {start 0, length 2}
        ext = b'\x08\x00\x10\x02'
        sl = b'\x0a' + varint(len(ext)) + ext
        out += b'\x3a' + varint(len(sl)) + sl
    return out


def table_block(items, restart_every):
    """(key, value) list -> block bytes with PREFIX-COMPRESSED keys and a restart array."""
    out, restarts, last = bytearray(), [], b''
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_every == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        last = k
    for r in restarts:
        out += struct.pack('<I', r)
    return bytes(out) + struct.pack('<I', len(restarts))


def frame(block):
    return block + b'\x00' + struct.pack('<I', tb.mask_crc(tb.crc32c(block + b'\x00')))


def assemble(prefix, tensors, shard_of=None, n_shards=1, extra_items=(), blocks=3, restart_every=2):
    """Write `<prefix>.index` and `<prefix>.data-0000i-of-0000n` holding `tensors` (name -> little-endian array,
    stored in exactly the shape given).  shard_of: name -> shard index.  Returns the sorted item list."""
    order = sorted(tensors)
    shard_of = shard_of or {}
    blobs = {i: bytearray() for i in range(n_shards)}
    offs = {}
    for k in order:
        sid = shard_of.get(k, 0)
        offs[k] = len(blobs[sid])
        blobs[sid] += np.ascontiguousarray(tensors[k]).tobytes()
    for sid in range(n_shards):
        with open('{}.data-{:05d}-of-{:05d}'.format(prefix, sid, n_shards), 'wb') as f:
            f.write(bytes(blobs[sid]))
    items = [(b'', b'\x08' + varint(n_shards) + b'\x1a\x02\x08\x01')]     # header: num_shards, version.producer 1
    for k in order:
        raw = np.ascontiguousarray(tensors[k]).tobytes()
        items.append((k.encode(), entry_proto(DTYPE[tensors[k].dtype], tensors[k].shape, shard_of.get(k, 0), offs[k],
                                              len(raw), tb.mask_crc(tb.crc32c(raw)))))
    items.extend(extra_items)
    items.sort(key=lambda kv: kv[0])
    per = max(1, -(-len(items) // blocks))
    groups = [items[i:i + per] for i in range(0, len(items), per)]
    out, index_items = bytearray(), []
    for g in groups:
        blk = table_block(g, restart_every=restart_every)
        index_items.append((g[-1][0], varint(len(out)) + varint(len(blk))))
        out += frame(blk)
    meta = struct.pack('<II', 0, 1)
    meta_off = len(out)
    out += frame(meta)
    index = table_block(index_items, restart_every=1)
    index_off = len(out)
    out += frame(index)
    footer = varint(meta_off) + varint(len(meta)) + varint(index_off) + varint(len(index))
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out) + footer)
    return items
