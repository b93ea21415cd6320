"""CPU oracle for the nsynth_wavenet generation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`nsynth_wavenet_amd/`)
may import, call, link or execute anything under `oracle/`.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it, and only
as the checker / the timed CPU baseline.

PARITY UNPINNED: the reference (bfs18/nsynth_wavenet) is TensorFlow-1.x Python;
TensorFlow is not installed here or on the GPU box and the reference's own tests
print instead of asserting, so no TF-derived golden vector exists.  What IS held
by the reference and checked here, row by row of SURVEY.md section 8(a):

  reference-pinned
    a8   _clip_quant_scale / cast_quantize: every non-mu-law output wav the reference commits lies on the
         2^-15 grid inside [-1, 1-2^-15] and is a fixed point of the restated quantiser
         (tests/golden/ref_fixture_facts.npz, tests/test_oracle.py::test_reference_outputs_on_grid_match_clip_quant)
    a11  inv_mu_law: the <= 256 distinct sample values of the reference's mu-law output wavs equal the restated
         decode table to 1 float32 ulp, 0 at index 0 (::test_inv_mu_law_table_matches_reference_outputs)
    a7 / a12 lengths: 154 480 input samples -> F = 773 -> 154 112 (IAF, centre crop 244) and 154 600 (AR)
         (::test_reference_fixture_lengths; GPU: tests/test_gpu_configs.py)
    a6   the scale path of the flow head as the reference's tests/test_scale.py:64-107 defines it in numpy
         (softplus -> clip(e^-9, e^7), product over four flows): closed-form moments on the reference's
         76 800-draw experiment (::test_scale_path_statistics_of_reference_test_scale; GPU:
         tests/test_gpu_iaf.py::test_flow_head_scale_path_on_test_scale_draws).  The statistics the reference
         prints beside it (scale.m 0.38296, scale.std 0.61160) are a log line of a weight-normalised TF model
         (they are the moments of parameters ~ N(-0.01, 0.93^2), not of N(0,1) draws): regime check only.
  restatement-only (checked against the reference's own invariants K1/K2/K3, against an independent torch-CPU
  implementation that shares no code -- oracle/torch_ref.py --, against TF op definitions, never against TF output)
    a1-a5, a7 (arithmetic), a9, a10, a13, a14 and the float values of a6.
"""
