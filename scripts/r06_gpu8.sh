#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
O=gpurun_out/r06f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_iaf.py::test_fp32_forms_hoisted_against_fused_and_the_frame_axis_upsampler_against_the_phase_major_one tests/test_ref_float.py::test_fp32_engine_equals_the_reference_code_at_full_size tests/test_gpu_configs.py::test_pipelined_cli_writes_what_the_serial_loop_writes -x -q -m gpu -s 2>&1 | tail -8
timeout 700 python tests/tools/fuzz_gpu.py 600 > $O/fuzz.txt 2>&1
tail -4 $O/fuzz.txt
