#!/bin/bash
# builds ablated forms of the attention forward (-DCFFM_EXPERIMENTS -DFWD_ABLATE=n) into build/: run HERE (hipcc cross-compiles),
# then on the GPU: scripts/r05_fwd_ab.sh ablate build/libcffm_fabl*.so
cd "$(dirname "$0")/.." && mkdir -p build
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DCFFM_EXPERIMENTS -DFWD_ABLATE=$n vss_cffm_amd/csrc/cffm_hip.hip -o build/libcffm_fabl$n.so 2>/dev/null &
done; wait; ls -la build/libcffm_fabl*.so
