"""Parallel-WaveNet (IAF student) generation CLI on MI355X.

    python eval_parallel_wavenet.py --ckpt_dir DIR --source_path WAVS_OR_NPYS --save_path OUT

Same flags and output naming as the reference's eval_parallel_wavenet.py."""
from nsynth_wavenet_amd import cli
from nsynth_wavenet_amd.wavenet import parallelgen


def generate(args):
    cli.run(args, parallelgen.synthesis)


if __name__ == '__main__':
    generate(cli.build_parser(__doc__).parse_args())
