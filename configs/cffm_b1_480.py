_base_ = ['./_base_/cffm_head.py']
# CFFM-B1, 480x480, T=4 (BASELINE configs 2/3): MiT-B1 feature widths, two CFFM blocks
