cd "$(dirname "$0")/.."
for l in vss_cffm_amd/libcffm_hip.so; do python scripts/r06_fwd_b2b.py $l; done
for s in 162 128 112 96 64; do CFFM_ATTN_FWD_SLOTS=$s python scripts/r06_fwd_b2b.py build/libcffm_exp.so; done
for s in 162 128 96 81 64; do CFFM_ATTN_FWD_SLOTS=$s python scripts/r06_fwd_b2b.py build/libcffm_exp_occ3.so; done
for s in 200 128 96; do CFFM_ATTN_FWD_SLOTS=$s python scripts/r06_fwd_b2b.py build/libcffm_exp_occ3.so 2 64; CFFM_ATTN_FWD_SLOTS=$s python scripts/r06_fwd_b2b.py build/libcffm_exp.so 2 64; done
