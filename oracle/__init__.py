"""CPU oracle for the nsynth_wavenet generation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`nsynth_wavenet_amd/`)
may import, call, link or execute anything under `oracle/`.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it, and only
as the checker / the timed CPU baseline.

PARITY UNPINNED: the reference (bfs18/nsynth_wavenet) is TensorFlow-1.x Python;
TensorFlow is not installed here or on the GPU box and the reference's own tests
print instead of asserting, so no TF-derived golden vector exists.  The
restatement is pinned only by (a) the structural facts extracted from the
reference's committed output wavs (tests/golden/ref_fixture_facts.npz), (b) the
reference's own internal invariants (incremental == full-sequence teacher;
x == eps*scale_tot+mean_tot), and (c) an independent torch-CPU implementation
(oracle/torch_ref.py).
"""
