# Decode-head part of the CFFM model dict (what SURVEY.md 8b lists as the ctor contract), written for this
# repository's bench / tests.  The reference's own local_configs/cffm/*.py load unchanged through
# vss_cffm_amd.config.Config (tests/test_boundary.py checks that where /root/reference is present).
norm_cfg = dict(type='SyncBN', requires_grad=True)
model = dict(
    type='EncoderDecoder_clips',
    decode_head=dict(
        type='CFFMHead_clips_resize1_8',
        in_channels=[64, 128, 320, 512],
        in_index=[0, 1, 2, 3],
        feature_strides=[4, 8, 16, 32],
        channels=128,
        dropout_ratio=0.1,
        num_classes=124,
        norm_cfg=norm_cfg,
        align_corners=False,
        decoder_params=dict(embed_dim=256, depths=2),
        loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
        num_clips=4))
optimizer = dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
data = dict(samples_per_gpu=2)
