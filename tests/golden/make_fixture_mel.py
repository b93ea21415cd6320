"""The mel of the reference's own test utterance, as data, for BASELINE configs[0] ("fastgen on tests/test_data mels").

Run in the build container only (needs /root/reference):
    python tests/golden/make_fixture_mel.py
Reads /root/reference/tests/test_data/test.wav (int16 mono 16 kHz, 154 480 samples) with this repository's loader, computes
its log-mel with this repository's host featuriser (nsynth_wavenet_amd/auxilaries/mel_extractor.py, the restatement of
auxilaries/mel_extractor.py:14-35,65-90 -- librosa is not installed here, so this is the only featuriser there is) and writes
tests/golden/fixture_mel.npz: the [773, 80] float32 mel, the sample count, and the first 2 048 samples of the utterance (the
prefix the K1 check of the fastgen test teacher-forces).  DATA only: no source text of the reference is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
WAV = '/root/reference/tests/test_data/test.wav'
OUT = os.path.join(HERE, 'fixture_mel.npz')


def main():
    from nsynth_wavenet_amd.auxilaries import mel_extractor as M, utils
    wav = utils.load_audio(WAV, sample_length=-1)
    assert wav.dtype == np.float32 and wav.shape == (154480,), wav.shape
    mel = M.melspectrogram(wav)
    assert mel.shape == (773, 80) and mel.dtype == np.float32 and 0.0 <= mel.min() and mel.max() <= 1.0
    np.savez_compressed(OUT, mel=mel, n_samples=np.int64(wav.shape[0]), wav_head=wav[:2048].copy(),
                        source=np.array('tests/test_data/test.wav of the reference; featuriser: nsynth_wavenet_amd.auxilaries.mel_extractor.melspectrogram'))
    print('wrote', OUT, os.path.getsize(OUT), 'bytes; mel', mel.shape, 'range', float(mel.min()), float(mel.max()),
          'mean', float(mel.mean()))


if __name__ == '__main__':
    main()
