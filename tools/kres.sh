#!/bin/bash
# resource usage of the device kernels (cross-compile, no GPU): tools/kres.sh [name-fragment ...] [-- extra hipcc flags]
cd "$(dirname "$0")/../spacepharer_amd/csrc" || exit 1
frags=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do frags+=("$1"); shift; done; [ "$1" = "--" ] && shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --cuda-device-only -c sp_gpu.hip -o /tmp/kres_sp_gpu.o -Rpass-analysis=kernel-resource-usage "$@" 2> /tmp/kres.txt
python3 - "${frags[@]}" <<'PY'
import re, sys
name = None; info = {}
for line in open("/tmp/kres.txt"):
    m = re.search(r"Function Name: (\S+)", line)
    if m: name = m.group(1); info[name] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", line)
    if m and name: info[name][m.group(1).strip()] = int(m.group(2))
for k, v in info.items():
    if not sys.argv[1:] or any(f in k for f in sys.argv[1:]):
        print(k[:60], {a: v[a] for a in ("VGPRs", "AGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]") if a in v})
PY
