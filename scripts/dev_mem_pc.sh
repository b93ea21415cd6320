#!/bin/bash
# producer -> consumer windows through the memory side (scripts/ubench/mem_power.hip, `pc` mode): run on the GPU box
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
cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o /tmp/mem_power "$GRAFT_REPO_ROOT/scripts/ubench/mem_power.hip" && cd "$GRAFT_REPO_ROOT"
timeout 300 /tmp/mem_power "$HW" 2.0 pc 2>&1 | tee gpurun_out/mem_pc.txt
