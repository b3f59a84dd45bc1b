_base_ = ['./_base_/cffm_head.py']
# CFFM++-B1 (BASELINE config 5): fine-tune with per-video prototype tokens; lr / schedule as the reference's
# *_fine_w_proto.40k.py; the optimizer dict replaces the base one instead of merging into it.
model = dict(decode_head=dict(type='CFFMHead_clips_resize1_8_finetune_w_prototype3'))
optimizer = dict(_delete_=True, type='AdamW', lr=2e-4, weight_decay=0.01)
