"""nsynth_wavenet_amd — MI355X (gfx950) native generation path of bfs18/nsynth_wavenet.

Only what the generation hot path needs lives here:
  csrc/      hand-written HIP kernels + the C ABI (include/wnhip.h) -> lib/libwnhip.so
  _lib.py    ctypes binding of that ABI (fails loudly when the library is missing)
  engine.py  device buffers / streams (PyTorch-ROCm is plumbing only)
  config.py, weights.py   config JSON surface and the TF-variable-named weight container
  wavenet/   host-side mirror of the reference's python interface for this path
             (parallelgen.synthesis, fastgen.encode/synthesis, ParallelWavenet, Fastgen ...)
"""
__version__ = '0.1.0'
