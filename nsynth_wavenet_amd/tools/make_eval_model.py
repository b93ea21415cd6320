"""Strip / convert a checkpoint to what generation needs (the reference's tools/make_eval_model.py
keeps only the '<var>/ExponentialMovingAverage' tensors plus the config json).

    python -m nsynth_wavenet_amd.tools.make_eval_model --ckpt_dir RUN_DIR --out_dir EVAL_DIR [--format npz|tf]

Reads the newest checkpoint of RUN_DIR (TensorFlow V2 bundle or .npz) and the single *.json,
keeps the variables the generation graph of that config uses (EMA shadows; raw names for
teacher-owned deconv variables) and writes them as `.npz` or as a TF V2 bundle.
"""
import argparse
import os
import shutil

from .. import cli
from .. import config as cfg
from .. import tf_bundle
from .. import weights as wts


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument('--ckpt_dir', required=True)
    ap.add_argument('--out_dir', required=True)
    ap.add_argument('--format', default='npz', choices=['npz', 'tf'])
    args = ap.parse_args(argv)
    hp, ckpt = cli.resolve_model(args.ckpt_dir)
    w = wts.load_checkpoint(ckpt, hp)
    os.makedirs(args.out_dir, exist_ok=True)
    base = os.path.basename(ckpt)
    base = base[:-4] if base.endswith('.npz') else base
    out = os.path.join(args.out_dir, base)
    if args.format == 'npz':
        path = wts.save_checkpoint(out, w, hp)
    else:
        keys = wts.checkpoint_keys(w, hp)                 # one rule for both writers (weights.raw_name_variables)
        path = tf_bundle.write_bundle(out, {keys[k]: v for k, v in w.items()})
        with open(os.path.join(args.out_dir, 'checkpoint'), 'wt') as f:
            f.write('model_checkpoint_path: "{}"\n'.format(base))
    import glob
    for j in glob.glob(os.path.join(args.ckpt_dir, '*.json')):
        shutil.copy(j, args.out_dir)
    print('wrote', path, 'with', len(w), 'variables')


if __name__ == '__main__':
    main()
