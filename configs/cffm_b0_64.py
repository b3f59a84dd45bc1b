_base_ = ['./_base_/cffm_head.py']
# CFFM-B0 plumbing case (BASELINE config 1): MiT-B0 feature widths, one CFFM block
model = dict(decode_head=dict(in_channels=[32, 64, 160, 256], decoder_params=dict(embed_dim=256, depths=1)))
