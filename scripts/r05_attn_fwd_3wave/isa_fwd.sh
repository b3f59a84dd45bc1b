#!/bin/bash
# ISA of k_cfm_attn_fwd of the current sources (extra flags as arguments) -> /tmp/isa/fwd_cur.s + register / instruction summary
mkdir -p /tmp/isa && cd /tmp/isa
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -S --cuda-device-only "$@" /root/repo/vss_cffm_amd/csrc/cffm_hip.hip -o cffm.s 2>/dev/null
L=$(grep -n "^_Z14k_cfm_attn_fwd.*:" cffm.s | head -1 | cut -d: -f1)
awk -v L=$L 'NR>=L' cffm.s | awk '/^\.Lfunc_end/{exit} {print}' > fwd_cur.s
awk -v L=$L 'NR>=L' cffm.s | grep -m5 -E "; (NumVgprs|NumAgprs|ScratchSize|Occupancy|codeLenInByte)"
grep -E "^\s+(v_|s_|ds_|buffer_|global_)" fwd_cur.s | awk '{print $1}' | sort | uniq -c | awk '{c[$2]=$1} END{v=0;s=0;d=0;m=0;vm=0; for(k in c){ if(k ~ /^v_mfma/) m+=c[k]; else if(k ~ /^v_/) v+=c[k]; else if(k ~ /^s_/) s+=c[k]; else if (k ~ /^ds_/) d+=c[k]; else vm+=c[k];} print "all three waves: VALU",v,"MFMA",m,"SALU",s,"LDS",d,"VMEM",vm}'
grep -c "s_waitcnt lgkmcnt(0)" fwd_cur.s
