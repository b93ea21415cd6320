"""CPU oracle for the nsynth_wavenet generation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`nsynth_wavenet_amd/`)
may import, call, link or execute anything under `oracle/`.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it, and only
as the checker / the timed CPU baseline.

PARITY -- what pins this oracle (round 5).  The reference (bfs18/nsynth_wavenet) is TensorFlow-1.x Python and
TensorFlow is not installed here or on the GPU box, and the reference's own tests print instead of asserting, so there
is NO TENSORFLOW-PRODUCED VECTOR.  What exists instead, strongest first:

  1. REFERENCE CODE EXECUTED (tests/golden/make_ref_float.py -> tests/golden/ref_float.npz, tests/test_ref_float.py).
     The reference's wavenet/masked.py, wavenet.py, parallel_wavenet.py, loss_func.py, fastgen.py, parallelgen.py and
     auxilaries/utils.py are imported UNMODIFIED from /root/reference and driven the way eval_parallel_wavenet.py /
     eval_wavenet.py drive them (parallelgen.synthesis; fastgen.load_deconv_stack, load_fastgen, synthesis,
     calculate_cond_vars; Wavenet.encode_signal + feed_forward), with `import tensorflow` resolved to
     tests/golden/tf_standin.py: a numpy evaluator of the ~70 TensorFlow primitives those files call (graph, session,
     placeholders, variable scopes, Saver.restore, conv2d, conv2d_transpose, pad / slice / reshape / transpose,
     FIFOQueue, random_uniform, ...).  Eleven small cases (the seven of make_golden.py plus weight-norm + resize-conv student
     and teacher, and a use_teacher_deconv student), each in float64 and float32, plus BASELINE configs[1] at its full size
     (one 76 800-sample utterance through parallelgen.synthesis) and wavenet_mol.json as shipped for 400 incremental steps.  This oracle agrees with those runs to
     float64 rounding (<= 1e-12 of the range) on x / mean_tot / scale_tot / log_scale_tot, the upsampler output, the
     full-sequence teacher, cond_vars and every network output of the free-running incremental loop; the sampled index
     streams are IDENTICAL; the variables the reference graphs create and the checkpoint keys their Savers request are
     exactly weights.expected_variables / checkpoint_keys.  Rows a1-a7, a9, a10, a12-a14 of SURVEY.md section 8.
     PINNED by this: everything the reference's own code decides (names, scopes, EMA keys, time_to_batch dilation
     arithmetic, causal padding, tap order, centre crop, gate order, residual / skip wiring, flow head, mean_tot /
     scale_tot recursion and clips, queue discipline, sampler formulas, quantiser, the driver loops down to the wav files).
     NOT pinned: TensorFlow's kernels themselves -- tf_standin.py restates their documented semantics (SAME-padding
     arithmetic of conv2d_transpose, nearest-neighbour resize, l2_normalize epsilon, softplus) -- float32 rounding order
     inside a TF kernel, and TF's random generators (randoms are injected; the 'ce' head's categorical draw is DEFINED as
     inverse-CDF from one uniform).  That remainder is why this header does not say "TensorFlow-pinned".
  2. REFERENCE NUMPY EXECUTED: auxilaries/utils.py's pure-numpy codecs (rows a8 / a11; tests/golden/make_ref_codec.py).
  3. REFERENCE-HELD FACTS:
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
  4. the reference's own invariants K1 / K2 / K3 and an independent torch-CPU implementation (oracle/torch_ref.py).
  Still restatement-only: the mel featuriser (oracle/mel_np.py: librosa is absent).
"""
