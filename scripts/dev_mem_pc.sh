#!/bin/bash
# energy ubenches on the GPU box: producer -> consumer windows through the memory side (scripts/ubench/mem_power.hip, `pc` mode) and the
# non-matrix instruction classes (scripts/ubench/lds_valu_power.hip).  usage: dev_mem_pc.sh [pc|lv|mf|all]
set -u
mkdir -p gpurun_out
HW=$(python - <<'PY'
import glob, torch
pr = torch.cuda.get_device_properties(0)
bdf = '{:04x}:{:02x}:{:02x}.0'.format(pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
hw = glob.glob('/sys/bus/pci/devices/{}/hwmon/hwmon*'.format(bdf))
print(hw[0] if hw else '')
PY
)
echo "hwmon $HW"
WHAT=${1:-all}
if [ "$WHAT" = pc ] || [ "$WHAT" = all ]; then
  cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o /tmp/mem_power "$GRAFT_REPO_ROOT/scripts/ubench/mem_power.hip" && cd "$GRAFT_REPO_ROOT"
  timeout 300 /tmp/mem_power "$HW" 2.0 pc 2>&1 | tee gpurun_out/mem_pc.txt
fi
if [ "$WHAT" = lv ] || [ "$WHAT" = all ]; then
  cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o /tmp/lds_valu_power "$GRAFT_REPO_ROOT/scripts/ubench/lds_valu_power.hip" && cd "$GRAFT_REPO_ROOT"
  timeout 300 /tmp/lds_valu_power "$HW" 2.5 2>&1 | tee gpurun_out/lds_valu_power.txt
fi
if [ "$WHAT" = mf ] || [ "$WHAT" = all ]; then
  cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o /tmp/mfma_power "$GRAFT_REPO_ROOT/scripts/ubench/mfma_power.hip" && cd "$GRAFT_REPO_ROOT"
  timeout 300 /tmp/mfma_power "$HW" 2.5 widths 2>&1 | tee gpurun_out/mfma_power_widths.txt
fi
