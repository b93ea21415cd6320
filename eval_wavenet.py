"""Autoregressive WaveNet (teacher, "fastgen") generation CLI on MI355X.

    python eval_wavenet.py --ckpt_dir DIR --source_path WAVS_OR_NPYS --save_path OUT

Same flags and output naming as the reference's eval_wavenet.py."""
from nsynth_wavenet_amd import cli
from nsynth_wavenet_amd.wavenet import fastgen


def _synthesis(hparams, mel, save_names, checkpoint_path):
    encoding = fastgen.encode_mel(hparams, mel, checkpoint_path)
    fastgen.synthesis(hparams, encoding, save_names, checkpoint_path)


def generate(args):
    cli.run(args, _synthesis)


if __name__ == '__main__':
    generate(cli.build_parser(__doc__).parse_args())
