"""Extract the structural known-answers the reference's committed OUTPUT wavs pin
 into a small data fixture.

Run in the build container only (needs /root/reference):
    python tests/golden/make_ref_facts.py
Writes tests/golden/ref_fixture_facts.npz.  The fixture holds DATA only: sample
counts, min/max, grid residue and the set of distinct float32 sample values of
the mu-law outputs (<= 256 per file; these are values the reference's TF
inv_mu_law produced, i.e. a genuine known-answer for auxilaries/utils.py:108-122).
"""
import glob
import os
import numpy as np
from scipy.io import wavfile

REF = '/root/reference/tests'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_fixture_facts.npz')


def main():
    facts = {}
    sr, src = wavfile.read(os.path.join(REF, 'test_data', 'test.wav'))
    facts['test_wav/sr'] = np.int64(sr)
    facts['test_wav/n'] = np.int64(src.shape[0])
    facts['test_wav/dtype'] = np.array(str(src.dtype))
    names = []
    for path in sorted(glob.glob(os.path.join(REF, 'pred_data-*', '*.wav'))):
        key = os.path.relpath(path, REF)
        sr, a = wavfile.read(path)
        a = np.asarray(a)
        names.append(key)
        facts[key + '/sr'] = np.int64(sr)
        facts[key + '/n'] = np.int64(a.shape[0])
        facts[key + '/dtype'] = np.array(str(a.dtype))
        facts[key + '/min'] = np.float64(a.min())
        facts[key + '/max'] = np.float64(a.max())
        # residue on the 2^-15 grid (0.0 for every non-mu-law output)
        g = a.astype(np.float64) * 32768.0
        facts[key + '/grid_residue'] = np.float64(np.abs(g - np.round(g)).max())
        u = np.unique(a)
        facts[key + '/n_unique'] = np.int64(u.shape[0])
        if 'use_mu_law' in key:
            facts[key + '/unique'] = u.astype(np.float32)
    facts['names'] = np.array(names)
    np.savez_compressed(OUT, **facts)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes;', len(names), 'wavs')


if __name__ == '__main__':
    main()
