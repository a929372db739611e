#!/bin/bash
# Host-side AddressSanitizer + UBSan build of the library and the host driver into build/asan/ (device code is
# compiled as usual: GPU sanitizers are not available on the pool).  Build here, then on the GPU box e.g.
#   export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 RELATE_CLI=$GRAFT_REPO_ROOT/build/asan/relateHMM_hip
#   python -u tools/cli_scale.py 190 2000        # 17 955 pairs: two fit populations, decode sub-blocks, text writer
# (detect_leaks=0: the HIP runtime keeps its own allocations until exit.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/asan && cd $R/build/asan
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -Wno-unused-function \
  -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
  -o librelate_hip.so $R/relate_amd/csrc/relate_hip.hip
hipcc -O1 -g -std=c++17 -fsanitize=address -fsanitize=undefined -fno-gpu-sanitize -fno-omit-frame-pointer \
  -o relateHMM_hip $R/relate_amd/csrc/host/relate_cli.cpp -I $R/include -L . -lrelate_hip '-Wl,-rpath,$ORIGIN'
ls -la $R/build/asan
