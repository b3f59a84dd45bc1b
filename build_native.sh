#!/bin/bash
# Build libcffm_hip.so (gfx950, product) and, with --emu, tests/libcffm_emu.so (host emulator of the
# same sources: TEST INFRASTRUCTURE, see vss_cffm_amd/csrc/hipemu.h).
set -e
cd "$(dirname "$0")"
SRC=vss_cffm_amd/csrc/cffm_hip.hip
if [ "$1" == "--emu" ]; then
  /opt/rocm/lib/llvm/bin/clang++ -x c++ -std=c++17 -O2 -DCFFM_EMU -fPIC -shared -pthread $SRC -o tests/libcffm_emu.so
else
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value $SRC -o vss_cffm_amd/libcffm_hip.so "$@"
fi
