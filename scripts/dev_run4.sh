python scripts/dev_ramp.py 2>&1 | grep -v amdgpu.ids
timeout 900 python tests/tools/fuzz_gpu.py 720 2>&1 | tail -25
