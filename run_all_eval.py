"""Batch evaluation over several experiment directories (local mirror of the reference's
run_all_eval.py:95-140).

    python run_all_eval.py -c sweep.json -w WAVS_OR_NPYS -t OUT [-g 0]

`sweep.json` keeps the reference's keys: "exp_dirs" (training log directories) and
"eval_scripts" (eval_wavenet.py | eval_parallel_wavenet.py, one per directory); "hosts",
"users", "passwords" are accepted for compatibility but every host must be empty / null /
"localhost": copying runs from remote machines over ssh is outside this package -- mount or
copy the directory first.  For every experiment the newest checkpoint (model.ckpt-N.index or
model.ckpt-N.npz) and the single *.json are staged into <target>-<mm_dd_HH>/<exp>-model with a
TensorFlow-style `checkpoint` state file, the eval script is run on it, the audio goes to
<target>-<mm_dd_HH>/waves/<exp>-iter_N, and the staging directory is removed again.
"""
import argparse
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
EVAL_SCRIPTS = ('eval_wavenet.py', 'eval_parallel_wavenet.py')


def get_last_model_prefix(names):
    """(prefix, iteration) of the highest-numbered model.ckpt-N.{index,npz} in `names`."""
    best = -1
    for n in names:
        m = re.match(r'model\.ckpt-(\d+)\.(index|npz)$', n)
        if m:
            best = max(best, int(m.group(1)))
    if best < 0:
        raise FileNotFoundError('no model.ckpt-N.index / model.ckpt-N.npz checkpoint found')
    return 'model.ckpt-{}'.format(best), best


def write_checkpoint(model_prefix, save_path):
    """The two-line state file tf.train.latest_checkpoint reads."""
    with open(save_path, 'wt') as f:
        f.write('model_checkpoint_path: "{}"\n'.format(model_prefix))
        f.write('all_model_checkpoint_paths: "{}"\n'.format(model_prefix))


def stage_experiment(source_logdir, target_dir):
    """Copy the newest checkpoint + config of one experiment; -> (model_dir, wave_dir, iteration)."""
    source_logdir = os.path.abspath(os.path.expanduser(source_logdir))
    names = os.listdir(source_logdir)
    jsons = [n for n in names if n.endswith('.json')]
    if not jsons:
        raise FileNotFoundError('no *.json config in {}'.format(source_logdir))
    prefix, last_iter = get_last_model_prefix(names)
    exp_tag = os.path.basename(source_logdir.rstrip(os.sep))
    model_dir = os.path.join(target_dir, exp_tag + '-model')
    wave_dir = os.path.join(target_dir, 'waves', '{}-iter_{}'.format(exp_tag, last_iter))
    os.makedirs(model_dir, exist_ok=True)
    os.makedirs(wave_dir, exist_ok=True)
    write_checkpoint(prefix, os.path.join(model_dir, 'checkpoint'))
    for path in glob.glob(os.path.join(source_logdir, prefix + '.*')) + [os.path.join(source_logdir, jsons[0])]:
        shutil.copy2(path, model_dir)
    events_dir = os.path.join(target_dir, exp_tag)
    events = [n for n in names if n.startswith('events.')]
    if events:
        os.makedirs(events_dir, exist_ok=True)
        for n in events:
            shutil.copy2(os.path.join(source_logdir, n), events_dir)
    return model_dir, wave_dir, last_iter


def syn_wave(eval_script, ckpt_dir, source_path, save_path, gpu_id):
    script = os.path.basename(eval_script)
    if script not in EVAL_SCRIPTS:
        raise ValueError('eval script must be one of {}, got {}'.format(EVAL_SCRIPTS, eval_script))
    cmd = [sys.executable, os.path.join(HERE, script), '--ckpt_dir', ckpt_dir, '--source_path', source_path,
           '--save_path', save_path, '--gpu_id', str(gpu_id)]
    print('Running evaluation:', ' '.join(cmd), flush=True)
    return subprocess.call(cmd)


def run_all(json_path, source_waves, target_dir, gpu_id, stamp=None):
    with open(json_path, 'rt') as f:
        configs = json.load(f)
    exp_dirs, scripts = configs['exp_dirs'], configs['eval_scripts']
    if len(exp_dirs) != len(scripts):
        raise ValueError('"exp_dirs" and "eval_scripts" must have the same length')
    for host in configs.get('hosts', []):
        if host not in (None, '', 'localhost', '127.0.0.1'):
            raise ValueError('remote host {!r}: only local experiment directories are supported'.format(host))
    target_dir = os.path.abspath(os.path.expanduser(target_dir))
    source_waves = os.path.abspath(os.path.expanduser(source_waves))
    target_dir = '-'.join([target_dir, stamp or time.strftime('%m_%d_%H', time.localtime())])
    print('Save all data to {}'.format(target_dir))
    print('Use source waves in {}'.format(source_waves))
    failed = []
    for src_logdir, eval_script in zip(exp_dirs, scripts):
        print('Running eval for {}'.format(os.path.basename(src_logdir.rstrip(os.sep))))
        model_dir, wave_dir, _ = stage_experiment(src_logdir, target_dir)
        try:
            if syn_wave(eval_script, model_dir, source_waves, wave_dir, gpu_id) != 0:
                failed.append(src_logdir)
        finally:
            shutil.rmtree(model_dir)
    return target_dir, failed


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    parser.add_argument('--config', '-c', required=True, help='Config json file')
    parser.add_argument('--wave_dir', '-w', required=True, help='Source wave directory')
    parser.add_argument('--target_dir', '-t', required=True, help='Target directory')
    parser.add_argument('--gpu_id', '-g', default='0', help='Gpu id')
    args = parser.parse_args()
    _, bad = run_all(args.config, args.wave_dir, args.target_dir, args.gpu_id)
    sys.exit(1 if bad else 0)
