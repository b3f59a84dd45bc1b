import sys, torch
sys.path.insert(0, '.')
from tests.test_headfuse import *
from tests.golden.make_golden_head import feature_maps, labels
device = torch.device('cuda')
chans, size = (64, 128, 320, 512), 256
head = build_head(RI.head_cfg(in_channels=chans, depths=2))
head.load_state_dict(R.synth_state(head, seed=71), strict=False)
head.dropout.p = 0.0
Hd.revert_sync_batchnorm(head)
head.to(device).train()
feats = [f.to(device).requires_grad_(True) for f in feature_maps(2, 4, size, chans=chans, seed=72)]
lab = labels(2, 4, size, seed=73).to(device)
def step():
    for p in head.parameters(): p.grad = None
    for f in feats: f.grad = None
    out = head(feats, 2, 4)
    loss = head.losses(out, lab)
    loss['loss_seg'].backward()
    return out, loss['loss_seg']
snap = lambda: {k: p.grad.clone() for k, p in head.named_parameters() if p.grad is not None}
step(); step(); ref = snap()
e2 = None
step(); e2 = snap()
print('eager vs eager differing:', [k for k in ref if not torch.equal(ref[k], e2[k])])
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): step()
for i in range(3):
    g.replay(); torch.cuda.synchronize()
    r = snap()
    print('replay', i, 'differing:', [(k, float((r[k]-ref[k]).abs().max())) for k in ref if not torch.equal(ref[k], r[k])][:8])
