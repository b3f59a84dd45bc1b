#!/bin/bash
# Register / LDS / spill figures of the kernels of a library build from its code-object metadata (no GPU needed).
# usage: scripts/kregs.sh [lib.so] [filter-regex]
LIB=${1:-vss_cffm_amd/libcffm_hip.so}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$LIB --output=$T/k.co --unbundle 2>/dev/null || \
  { /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $LIB && /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | python3 -c "
import sys, re
txt = sys.stdin.read()
pat = re.compile(sys.argv[1])
for blk in txt.split('- .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s*(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    if pat.search(name):
        print('%-70s vgpr %4s agpr %4s sgpr %4s lds %7s spill(v) %s scratch %s' % (name[:70], g('vgpr_count'), blk.split()[0], g('sgpr_count'), g('group_segment_fixed_size'), g('vgpr_spill_count'), g('private_segment_fixed_size')))
" "${2:-.}"
rm -rf $T
