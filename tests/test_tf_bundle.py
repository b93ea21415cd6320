"""CPU tests of the TensorFlow V2 checkpoint (tensor bundle) reader/writer."""
import os
import struct

import numpy as np
import pytest

from conftest import load_json
from nsynth_wavenet_amd import config as cfg
from nsynth_wavenet_amd import tf_bundle as tb
from nsynth_wavenet_amd import weights as wts


def test_crc32c_known_answers():
    # RFC 3720 / LevelDB test vectors
    assert tb.crc32c(b'\x00' * 32) == 0x8a9136aa
    assert tb.crc32c(b'\xff' * 32) == 0x62a8ab43
    assert tb.crc32c(bytes(range(32))) == 0x46dd794e
    assert tb.crc32c(b'123456789') == 0xe3069283
    assert tb.mask_crc(tb.crc32c(b'foo')) != tb.crc32c(b'foo')


def test_round_trip_many_tensors_multi_block(tmp_path):
    rs = np.random.RandomState(0)
    tensors = {'scope_%03d/sub/W/ExponentialMovingAverage' % i: rs.standard_normal([1, 3, i % 5 + 1, 7]).astype(np.float32)
               for i in range(300)}
    tensors['global_step'] = np.array(123456, np.int64)
    tensors['empty'] = np.zeros([0, 4], np.float32)
    prefix = tb.write_bundle(str(tmp_path / 'model.ckpt-9'), tensors, block_size=512)
    r = tb.BundleReader(prefix)
    assert sorted(r.entries) == sorted(tensors)
    assert r.get_variable_to_shape_map()['global_step'] == []
    for k, v in tensors.items():
        got = r.get_tensor(k, verify=True)
        assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v)
    raw = bytearray(open(prefix + '.index', 'rb').read())
    assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57 and len(raw) > 48 + 5
    raw[10] ^= 0xff                                        # corrupt a data block -> checksum must notice
    open(prefix + '.index', 'wb').write(bytes(raw))
    with pytest.raises(ValueError):
        tb.BundleReader(prefix)
    open(prefix + '.index', 'wb').write(b'not a table')
    with pytest.raises(ValueError):
        tb.BundleReader(prefix)


def test_reader_on_hand_assembled_table(tmp_path):
    """An index file assembled byte by byte from the format description (independent of
    write_bundle): one data block {"": header, "a/b": float32[2,3]}, restart interval 16."""
    prefix = str(tmp_path / 'hand')
    vals = np.arange(6, dtype='<f4').reshape(2, 3)
    open(prefix + '.data-00000-of-00001', 'wb').write(vals.tobytes())
    header = bytes([0x08, 0x01])                                           # num_shards = 1
    shape = bytes([0x12, 0x02, 0x08, 0x02, 0x12, 0x02, 0x08, 0x03])        # dims 2, 3
    entry = bytes([0x08, 0x01, 0x12, len(shape)]) + shape + bytes([0x28, 24, 0x35]) + \
        struct.pack('<I', tb.mask_crc(tb.crc32c(vals.tobytes())))
    block = bytes([0, 0, len(header)]) + header + bytes([0, 3, len(entry)]) + b'a/b' + entry
    block += struct.pack('<II', 0, 1)                                      # one restart at 0
    def framed(b):
        return b + b'\x00' + struct.pack('<I', tb.mask_crc(tb.crc32c(b + b'\x00')))
    out = framed(block)
    meta = struct.pack('<II', 0, 1)
    meta_off = len(out)
    out += framed(meta)
    handle = bytes([0, len(block)])                                        # offset 0, size < 128
    index = bytes([0, 3, len(handle)]) + b'a/b' + handle + struct.pack('<II', 0, 1)
    index_off = len(out)
    out += framed(index)
    footer = bytes([meta_off, len(meta), index_off, len(index)])
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
    open(prefix + '.index', 'wb').write(out + footer)
    r = tb.BundleReader(prefix)
    assert r.num_shards == 1 and list(r.entries) == ['a/b']
    assert np.array_equal(r.get_tensor('a/b', verify=True), vals)


def test_weights_load_from_tf_bundle_and_cli_resolution(tmp_path):
    from nsynth_wavenet_amd import cli
    from nsynth_wavenet_amd.tools import make_eval_model
    import json
    d = dict(load_json('parallel_wavenet.json'), num_iaf_layers=[2, 1], num_iters=1)
    hp = cfg.load_hparams(d)
    w = wts.synthetic_weights(hp, seed=4)
    run = tmp_path / 'run'
    run.mkdir()
    # what slim's saver leaves behind: EMA shadows AND raw variables AND optimizer slots
    blob = {k + wts.EMA: v for k, v in w.items()}
    blob.update({k: v + 1.0 for k, v in w.items()})
    blob['global_step'] = np.array(7, np.int64)
    tb.write_bundle(str(run / 'model.ckpt-7'), blob)
    tb.write_bundle(str(run / 'model.ckpt-3'), blob)
    (run / 'parallel_wavenet.json').write_text(json.dumps(d))
    assert wts.latest_checkpoint(str(run)).endswith('model.ckpt-7')
    (run / 'checkpoint').write_text('model_checkpoint_path: "model.ckpt-3"\nall_model_checkpoint_paths: "model.ckpt-3"\n')
    hp2, ck = cli.resolve_model(str(run))
    assert ck.endswith('model.ckpt-3')
    back = wts.load_checkpoint(ck, hp2)
    assert all(np.array_equal(back[k], w[k]) for k in w)                   # the EMA shadow wins
    make_eval_model.main(['--ckpt_dir', str(run), '--out_dir', str(tmp_path / 'eval_tf'), '--format', 'tf'])
    make_eval_model.main(['--ckpt_dir', str(run), '--out_dir', str(tmp_path / 'eval_npz'), '--format', 'npz'])
    for sub in ('eval_tf', 'eval_npz'):
        hp3, ck3 = cli.resolve_model(str(tmp_path / sub))
        got = wts.load_checkpoint(ck3, hp3)
        assert all(np.array_equal(got[k], w[k]) for k in w)
    r = tb.BundleReader(str(tmp_path / 'eval_tf' / 'model.ckpt-3'))
    assert all(k.endswith(wts.EMA) for k in r.entries) and len(r.entries) == len(w)


# ---- a second, writer-independent fixture: tests/bundle_assembler.py builds every byte from the format description
# ---- (LevelDB table format + tensor_bundle.proto), never through tf_bundle.write_bundle
import bundle_assembler as ba          # noqa: E402


def test_reader_on_hand_assembled_multi_block_two_shard_bundle(tmp_path):
    """Two data shards, three data blocks, keys sharing long prefixes (prefix compression with restart interval 2,
    so both restart and non-restart entries occur), an int64 scalar, an empty tensor, a tensor whose stored
    shape differs from the variable's (Saver(reshape=True), parallelgen.py:40) and a partitioned entry
    (slices present), which must be refused by name."""
    prefix = str(tmp_path / 'model.ckpt-42')
    rs = np.random.RandomState(1)
    t = {
        'iaf_1/dilated_conv_1/W/ExponentialMovingAverage': rs.standard_normal([1, 3, 4, 8]).astype('<f4'),
        'iaf_1/dilated_conv_1/biases/ExponentialMovingAverage': rs.standard_normal([8]).astype('<f4'),
        'iaf_1/dilated_conv_2/W/ExponentialMovingAverage': rs.standard_normal([1, 3, 4, 8]).astype('<f4'),
        'iaf_1/out2_scale/W/ExponentialMovingAverage': rs.standard_normal([4, 1]).astype('<f4'),     # stored [4,1], variable [1,1,4,1]
        'global_step': np.array(42, '<i8'),
        'zero_len': np.zeros([0, 3], '<f4'),
    }
    order = sorted(t)
    shard_of = {k: (1 if 'dilated_conv_2' in k or k == 'zero_len' else 0) for k in order}
    items = ba.assemble(prefix, t, shard_of=shard_of, n_shards=2, blocks=3, restart_every=2,
                        extra_items=[(b'part/W', ba.entry_proto(1, [4, 2], 0, 0, 0, 0, sliced=True))])
    # the prefix compression really is exercised: some entry shares >= 20 key bytes with its predecessor
    assert any(a[0][:20] == b[0][:20] and len(a[0]) > 20 for a, b in zip(items, items[1:]))

    r = tb.BundleReader(prefix)
    assert r.num_shards == 2 and sorted(r.entries) == sorted(order + ['part/W'])
    for k in order:
        got = r.get_tensor(k, verify=True)
        assert got.dtype == t[k].dtype.newbyteorder('=') and got.shape == t[k].shape and np.array_equal(got, t[k])
    assert r.get_variable_to_shape_map()['global_step'] == [] and int(r.get_tensor('global_step')) == 42
    with pytest.raises(ValueError, match='part/W'):
        r.get_tensor('part/W')
    # reshape=True semantics of the loader: [4,1] in the file feeds a [1,1,4,1] variable; a wrong element count does not
    from nsynth_wavenet_amd.weights import _BundleView
    view = _BundleView(prefix)
    arr = np.asarray(view['iaf_1/out2_scale/W/ExponentialMovingAverage'])
    assert arr.reshape([1, 1, 4, 1]).shape == (1, 1, 4, 1)
    # `checkpoint` state file (run_all_eval.py:44-49 writes exactly these two lines): relative and absolute paths,
    # and a stale entry falls back to the highest-numbered bundle
    d = str(tmp_path)
    open(os.path.join(d, 'checkpoint'), 'wt').write('model_checkpoint_path: "model.ckpt-42"\nall_model_checkpoint_paths: "model.ckpt-42"\n')
    assert wts.latest_checkpoint(d) == prefix
    open(os.path.join(d, 'checkpoint'), 'wt').write('model_checkpoint_path: "{}"\n'.format(prefix))
    assert wts.latest_checkpoint(d) == prefix
    open(os.path.join(d, 'checkpoint'), 'wt').write('model_checkpoint_path: "model.ckpt-99"\n')
    assert wts.latest_checkpoint(d) == prefix


def test_teacher_owned_resize_conv_upsampler_keeps_raw_names_in_both_writers(tmp_path):
    """use_teacher_deconv + use_resize_conv: the reference's restore map (parallelgen.py:31-39) looks the
    teacher-owned upsampler up by RAW variable name; both checkpoint writers (npz and TF bundle) must store it
    that way (one rule: weights.raw_name_variables) and both must load back."""
    import json
    from nsynth_wavenet_amd import cli
    from nsynth_wavenet_amd.tools import make_eval_model
    d = dict(load_json('parallel_wavenet.json'), num_iaf_layers=[1, 1], use_teacher_deconv=True, use_resize_conv=True,
             deconv_config=[[7, 2], [12, 4]], num_iters=1)
    d.pop('use_share_deconv', None)
    hp = cfg.load_hparams(d)
    w = wts.synthetic_weights(hp, seed=6)
    raw = wts.raw_name_variables(w, hp)
    assert raw and all(k.startswith('iaf_share/resize_conv') for k in raw)
    run = tmp_path / 'run'
    run.mkdir()
    wts.save_checkpoint(str(run / 'model.ckpt-5'), w, hp)
    (run / 'cfg.json').write_text(json.dumps(d))
    for fmt in ('tf', 'npz'):
        out = tmp_path / ('eval_' + fmt)
        make_eval_model.main(['--ckpt_dir', str(run), '--out_dir', str(out), '--format', fmt])
        hp2, ck = cli.resolve_model(str(out))
        if fmt == 'tf':
            keys = set(tb.BundleReader(ck).entries)
        else:
            keys = set(np.load(ck if ck.endswith('.npz') else ck + '.npz').files)
        assert keys == {k if k in raw else k + wts.EMA for k in w}
        back = wts.load_checkpoint(ck, hp2)
        assert all(np.array_equal(back[k], w[k]) for k in w)
