"""CPU oracle for the CFFM block (CFFA + CFM).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates, with explicit index maps, what the reference computes in
``mmseg/models/decode_heads/cffm_module/cffm_transformer.py``:

* ``BasicLayer3d3.forward``            :917-927  -> ``layer_forward``
* ``CffmTransformerBlock3d3.forward``  :709-832  -> ``block_forward`` (CFFA part + residual/MLP)
* ``WindowAttention3d3.forward``       :364-606  -> ``block_forward`` (CFM part)
* ``get_relative_position_index``      :158-185  -> ``rel_index``
* ``window_partition`` / ``_reverse``  :29-71    -> ``window_pixels``

Everything is plain torch on whatever dtype the inputs carry (fp32 for parity
with the reference, fp64 for a noise-floor reference); autograd through these
functions is the backward oracle.

Parameter names are the reference's state_dict keys of one block
(SURVEY.md Appendix C), e.g. ``attn.qkv.weight``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

WS = 7          # window size            (cffm_head.py:77)
EXPAND = 3      # expand_size            (cffm_head.py:89)
NUM_HEADS = 8   # num_heads              (cffm_head.py:76)
FOCAL_WINDOW = 5                 # cffm_head.py:88 -> pooled-target neighbourhood 5x5
FOCAL_L_CLIPS = (1, 2, 3)        # cffm_head.py:94
FOCAL_KERNEL_CLIPS = (7, 5, 3)   # cffm_head.py:95
MASK_VALUE = -100.0              # cffm_transformer.py:445,490


# --------------------------------------------------------------------------- index maps
def padded_size(h0, w0):
    """cffm_transformer.py:720-724."""
    return (h0 + WS - 1) // WS * WS, (w0 + WS - 1) // WS * WS


def window_pixels(hp, wp):
    """[nW,49] flat padded-pixel index of token (i,j) of window (wy,wx); row-major
    windows and tokens, as window_partition produces (cffm_transformer.py:29-41)."""
    gy, gx = hp // WS, wp // WS
    out = np.zeros((gy * gx, WS * WS), dtype=np.int64)
    for wy in range(gy):
        for wx in range(gx):
            for i in range(WS):
                for j in range(WS):
                    out[wy * gx + wx, i * WS + j] = (WS * wy + i) * wp + (WS * wx + j)
    return out


def ring_pixels(hp, wp):
    """[nW,132] flat padded-pixel index of the 3-px ring keys.

    Four cyclic rolls (cffm_transformer.py:389-400), window partition of each,
    concatenation in the order tl,tr,bl,br (:410) and selection by
    ``valid_ind_rolled`` (:280-285,:415).  torch.roll(t, (-e,-e)) puts pixel
    (y+e, x+e) at (y, x); there is no border mask -- indices wrap.
    """
    gy, gx = hp // WS, wp // WS
    e = EXPAND
    rolls = ((+e, +e, lambda i, j: i >= WS - e or j >= WS - e),   # tl: mask_tl[:-e,:-e]=0
             (+e, -e, lambda i, j: i >= WS - e or j < e),         # tr: mask_tr[:-e,e:]=0
             (-e, +e, lambda i, j: i < e or j >= WS - e),         # bl: mask_bl[e:,:-e]=0
             (-e, -e, lambda i, j: i < e or j < e))               # br: mask_br[e:,e:]=0
    rows = []
    for wy in range(gy):
        for wx in range(gx):
            r = []
            for oy, ox, keep in rolls:
                for i in range(WS):
                    for j in range(WS):
                        if keep(i, j):
                            y = (WS * wy + i + oy) % hp
                            x = (WS * wx + j + ox) % wp
                            r.append(y * wp + x)
            rows.append(r)
    out = np.asarray(rows, dtype=np.int64)
    assert out.shape[1] == 4 * WS * WS - 4 * (WS - e) * (WS - e)  # 132, :273-274
    return out


def unfold_cells(gy, gx, stride, kk, pad):
    """[nW,kk*kk] flat cell index into a (gy*stride, gx*stride) pooled grid, -1 where
    nn.Unfold(kernel=kk, stride=stride, padding=pad) reads its zero padding
    (cffm_transformer.py:298-301,:339-343 and the mask unfolds :433-446,:481-492)."""
    ph, pw = gy * stride, gx * stride
    out = np.full((gy * gx, kk * kk), -1, dtype=np.int64)
    for wy in range(gy):
        for wx in range(gx):
            for a in range(kk):
                for b in range(kk):
                    u = stride * wy - pad + a
                    v = stride * wx - pad + b
                    if 0 <= u < ph and 0 <= v < pw:
                        out[wy * gx + wx, a * kk + b] = u * pw + v
    return out


def rel_index(kk):
    """get_relative_position_index(q=(7,7), k=(kk,kk)) (cffm_transformer.py:158-185)
    -> int64 [49, kk*kk] into a table of (6+kk)^2 entries."""
    out = np.zeros((WS * WS, kk * kk), dtype=np.int64)
    for qi in range(WS):
        for qj in range(WS):
            for a in range(kk):
                for b in range(kk):
                    out[qi * WS + qj, a * kk + b] = (qi - a + kk - 1) * (WS + kk - 1) + (qj - b + kk - 1)
    return out


def rel_index_own():
    """relative_position_index buffer (cffm_transformer.py:257-267): [49,49] into a 169-row table."""
    out = np.zeros((WS * WS, WS * WS), dtype=np.int64)
    for qi in range(WS):
        for qj in range(WS):
            for ki in range(WS):
                for kj in range(WS):
                    out[qi * WS + qj, ki * WS + kj] = (qi - ki + WS - 1) * (2 * WS - 1) + (qj - kj + WS - 1)
    return out


# --------------------------------------------------------------------------- CFFA pieces
def pool_windows(z, wsg, weight, bias):
    """Window pooling by a learned Linear(wsg*wsg -> 1) (cffm_transformer.py:768-773, :797-802).
    z [B,Hh,Ww,C] with Hh,Ww multiples of wsg -> [B,Hh/wsg,Ww/wsg,C]."""
    b, hh, ww, c = z.shape
    t = z.view(b, hh // wsg, wsg, ww // wsg, wsg, c).permute(0, 1, 3, 2, 4, 5)
    t = t.reshape(b, hh // wsg, ww // wsg, wsg * wsg, c)
    return torch.einsum('buvkc,k->buvc', t, weight.reshape(-1)) + bias.reshape(())


def assemble_bias(p):
    """Additive bias [nH,49,289] in the key order own|ring|P0|f0|f1|f2
    (cffm_transformer.py:536-587)."""
    dev = p['attn.relative_position_bias_table'].device
    own = p['attn.relative_position_bias_table'][torch.from_numpy(rel_index_own()).to(dev).view(-1)]
    own = own.view(49, 49, -1).permute(2, 0, 1)                                    # :536-538
    ring = p['attn.relative_position_bias_table_to_neighbors'][0]                   # :544
    parts = [own, ring]
    t0 = p['attn.relative_position_bias_table_to_windows.0']
    parts.append(t0[:, torch.from_numpy(rel_index(FOCAL_WINDOW)).to(dev).view(-1)].view(-1, 49, FOCAL_WINDOW ** 2))
    for f, kk in enumerate(FOCAL_KERNEL_CLIPS):
        tf = p['attn.relative_position_bias_table_to_windows_clips.%d' % f]
        parts.append(tf[:, torch.from_numpy(rel_index(kk)).to(dev).view(-1)].view(-1, 49, kk * kk))
    return torch.cat(parts, dim=2)


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


# --------------------------------------------------------------------------- the block
def block_forward(x, p, want=False):
    """One CffmTransformerBlock3d3 (cffm_transformer.py:709-832).

    x [B,T,H0,W0,C] (NHWC per frame; frame T-1 is the target).  Returns the
    block output (same shape; only frame T-1 differs) and, if ``want``, a dict
    of intermediates for kernel-level parity tests.
    """
    bsz, t, h0, w0, c = x.shape
    assert t == len(FOCAL_L_CLIPS) + 1, 'the block indexes 3 reference frames + target (:780-792)'
    nh, hd = NUM_HEADS, c // NUM_HEADS
    hp, wp = padded_size(h0, w0)
    gy, gx = hp // WS, wp // WS
    nw = gy * gx
    dev = x.device
    ix = lambda a: torch.from_numpy(a).to(dev)

    # A.1  LayerNorm on all frames, then zero padding (:716, :721-724)
    z = F.layer_norm(x, (c,), p['norm1.weight'], p['norm1.bias'], 1e-5)
    zp = F.pad(z, (0, 0, 0, wp - w0, 0, hp - h0))
    zt = zp[:, t - 1]

    wq, bq = p['attn.qkv.weight'], p['attn.qkv.bias']

    # A.2  target q,k,v (:374-375); heads are contiguous 32-channel slices (:380)
    qkv = F.linear(zt, wq, bq).view(bsz, hp * wp, 3, c)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]

    win = ix(window_pixels(hp, wp))          # [nW,49]
    ring = ix(ring_pixels(hp, wp))           # [nW,132]
    q_w = q[:, win]                          # [B,nW,49,C]
    k_parts = [k[:, win], k[:, ring]]
    v_parts = [v[:, win], v[:, ring]]
    mask_parts = [torch.zeros(nw, 49 + 132, dtype=x.dtype, device=dev)]

    # A.5  pooled target windows (:741-776) and their 5x5 neighbourhood (:426-468)
    pooled = [pool_windows(zt, WS, p['pool_layers.0.weight'], p['pool_layers.0.bias'])]
    geo = [(1, FOCAL_WINDOW, FOCAL_WINDOW // 2)]
    # A.6  reference frames (:780-805, :470-518)
    for f, (s, kk) in enumerate(zip(FOCAL_L_CLIPS, FOCAL_KERNEL_CLIPS)):
        wsg = WS // s
        hpool, wpool = gy * s * wsg, gx * s * wsg
        zf = zp[:, f]
        if hpool != hp or wpool != wp:                                              # :794-795
            zf = F.interpolate(zf.permute(0, 3, 1, 2), size=(hpool, wpool), mode='bilinear').permute(0, 2, 3, 1)
        pooled.append(pool_windows(zf.contiguous(), wsg, p['pool_layers_clips.%d.weight' % f],
                                   p['pool_layers_clips.%d.bias' % f]))
        geo.append((s, kk, kk // 2))
    for pg, (s, kk, pad) in zip(pooled, geo):
        kv = F.linear(pg, wq, bq).view(bsz, -1, 3, c)                              # :449, :495 (q third unused)
        cells = ix(unfold_cells(gy, gx, s, kk, pad))                                # [nW,kk*kk]
        valid = (cells >= 0)
        safe = cells.clamp(min=0)
        k_parts.append(kv[:, :, 1][:, safe] * valid[None, :, :, None])              # unfold zero padding
        v_parts.append(kv[:, :, 2][:, safe] * valid[None, :, :, None])
        mask_parts.append(torch.where(valid, 0.0, MASK_VALUE).to(x.dtype))          # :445, :490

    k_all = torch.cat(k_parts, dim=2)        # [B,nW,289,C]   (:521)
    v_all = torch.cat(v_parts, dim=2)
    mask = torch.cat(mask_parts, dim=1)      # [nW,289]
    n = k_all.shape[2]

    heads = lambda a: a.view(bsz, nw, a.shape[2], nh, hd).permute(0, 1, 3, 2, 4)   # [B,nW,nH,*,hd]
    bias = assemble_bias(p)                  # [nH,49,289]
    attn = (heads(q_w) * (hd ** -0.5)) @ heads(k_all).transpose(-1, -2)            # :528-530
    attn = attn + bias[None, None] + mask[None, :, None, None, :]
    attn = torch.softmax(attn, dim=-1)                                              # :597
    ao = (attn @ heads(v_all)).permute(0, 1, 3, 2, 4).reshape(bsz, nw, 49, c)      # :601
    y = F.linear(ao, p['attn.proj.weight'], p['attn.proj.bias'])                    # :602

    # window_reverse + crop (:812-821)
    y_img = y.new_zeros(bsz, hp * wp, c)
    y_img[:, win.view(-1)] = y.view(bsz, nw * 49, c)
    y_img = y_img.view(bsz, hp, wp, c)[:, :h0, :w0]

    x1 = x[:, t - 1] + y_img                                                        # :823
    z2 = F.layer_norm(x1, (c,), p['norm2.weight'], p['norm2.bias'], 1e-5)
    hpre = F.linear(z2, p['mlp.fc1.weight'], p['mlp.fc1.bias'])
    x2 = x1 + F.linear(gelu_erf(hpre), p['mlp.fc2.weight'], p['mlp.fc2.bias'])     # :824
    out = torch.cat([x[:, :t - 1], x2.unsqueeze(1)], dim=1)                         # :826
    if not want:
        return out
    inter = dict(zt=zt, pooled=pooled, qkv_t=qkv, k_all=k_all, v_all=v_all, mask=mask, bias=bias,
                 ao=ao, y_img=y_img, x1=x1, z2=z2, hpre=hpre, x2=x2, n_keys=n)
    return out, inter


def split_block_params(state, i, prefix='blocks.'):
    """Pick block ``i``'s parameters out of a BasicLayer3d3 state_dict."""
    pre = '%s%d.' % (prefix, i)
    return {k[len(pre):]: v for k, v in state.items() if k.startswith(pre)}


def layer_forward(x, state, depth):
    """BasicLayer3d3.forward (cffm_transformer.py:917-927): x [B,T,C,H,W] -> same."""
    y = x.permute(0, 1, 3, 4, 2)
    for i in range(depth):
        y = block_forward(y, split_block_params(state, i))
    return y.permute(0, 1, 4, 2, 3).contiguous()


# --------------------------------------------------------------------------- CFFM++ (GTC)
def gtc_block_forward(x, h, w, centers, p):
    """SwinTransformerBlock_cluster.forward with shift 0 (pvt/swin_transformer_2d.py:605-665)
    around WindowAttention_cluster.forward (:208-262).

    x [B,h*w,C], centers [B,K,C].  Padding, window partition and reverse are kept
    as the reference does them even though they are numerically no-ops here.
    """
    bsz, l, c = x.shape
    nh, hd = NUM_HEADS, c // NUM_HEADS
    z = F.layer_norm(x, (c,), p['norm1.weight'], p['norm1.bias'], 1e-5).view(bsz, h, w, c)
    cn = F.layer_norm(centers, (c,), p['norm1.weight'], p['norm1.bias'], 1e-5)     # :622 same norm1
    hp, wp = padded_size(h, w)
    zp = F.pad(z, (0, 0, 0, wp - w, 0, hp - h)).view(bsz, hp * wp, c)
    win = torch.from_numpy(window_pixels(hp, wp)).to(x.device)
    nw = win.shape[0]
    xw = zp[:, win]                                                                  # [B,nW,49,C]
    q = F.linear(xw, p['attn.qkv.weight'], p['attn.qkv.bias'])[..., :c]             # :219-220 (k,v thirds unused)
    kvc = F.linear(cn, p['attn.qkv_cluster.weight'], p['attn.qkv_cluster.bias'])    # :223
    kc, vc = kvc[..., :c], kvc[..., c:]
    kq = centers.shape[1]
    qh = q.view(bsz, nw, 49, nh, hd).permute(0, 1, 3, 2, 4) * (hd ** -0.5)          # :226
    kh = kc.view(bsz, kq, nh, hd).permute(0, 2, 1, 3)[:, None]                      # [B,1,nH,K,hd]
    vh = vc.view(bsz, kq, nh, hd).permute(0, 2, 1, 3)[:, None]
    a = torch.softmax(qh @ kh.transpose(-1, -2), dim=-1)                             # :232, :248
    o = (a @ vh).permute(0, 1, 3, 2, 4).reshape(bsz, nw, 49, c)                      # :257
    o = F.linear(o, p['attn.proj_cluster.weight'], p['attn.proj_cluster.bias'])     # :258
    img = o.new_zeros(bsz, hp * wp, c)
    img[:, win.view(-1)] = o.view(bsz, nw * 49, c)
    img = img.view(bsz, hp, wp, c)[:, :h, :w].reshape(bsz, h * w, c)
    x1 = x + img                                                                     # :662
    z2 = F.layer_norm(x1, (c,), p['norm2.weight'], p['norm2.bias'], 1e-5)
    return x1 + F.linear(gelu_erf(F.linear(z2, p['mlp.fc1.weight'], p['mlp.fc1.bias'])),
                         p['mlp.fc2.weight'], p['mlp.fc2.bias'])                     # :663


def gtc_layer_forward(x, h, w, centers, state, depth=1):
    """BasicLayer_cluster.forward (pvt/swin_transformer_2d.py:1103-1148), element 0 of its
    6-tuple; depth 1 => shift 0 and the mask it builds is unused (:632-637)."""
    for i in range(depth):
        x = gtc_block_forward(x, h, w, centers, split_block_params(state, i))
    return x


# --------------------------------------------------------------------------- SegFormer embedding
def bilinear_matrix(n_in, n_out, dtype=torch.float32):
    """[n_out, n_in] matrix of F.interpolate(mode='bilinear', align_corners=False) along one dimension, restated from
    ATen's rule (UpSample.h area_pixel_compute_source_index; the reference reaches it through mmseg.ops.resize,
    ops/wrappers.py:8-29): scale = in/out in fp32, src = scale*(dst+0.5)-0.5 clamped at 0, i0 = floor(src),
    i1 = i0 + (i0 < in-1), weights 1-l and l with l = src - i0."""
    u = torch.zeros(n_out, n_in, dtype=dtype)
    scale = np.float32(n_in) / np.float32(n_out)
    for dst in range(n_out):
        src = np.float32(scale * np.float32(dst + 0.5)) - np.float32(0.5)
        src = np.float32(max(src, np.float32(0.)))
        i0 = min(int(src), n_in - 1)
        i1 = i0 + (1 if i0 < n_in - 1 else 0)
        l1 = float(np.float32(src - np.float32(i0)))
        u[dst, i0] += 1.0 - l1
        u[dst, i1] += l1
    return u


def segformer_fuse(feats, lin_w, lin_b, fuse_w):
    """Pre-BatchNorm output of the embedding in front of the hot path, cffm_head.py:102-119, in the reference's own order
    of operations: per scale `MLP` (flatten -> Linear, :26-37), reshape to a map, bilinear resize of c4, c3, c2 to c1's
    size (:106-113), torch.cat([_c4, _c3, _c2, _c1], dim=1) (:119), 1x1 `linear_fuse.conv` without bias.
    feats = [c1, c2, c3, c4] NCHW; lin_w / lin_b in the same order; fuse_w [E, 4E, 1, 1]."""
    n, _, hh, ww = feats[0].shape
    dt = feats[0].dtype
    maps = []
    for i in (3, 2, 1, 0):
        c = feats[i]
        m = F.linear(c.flatten(2).transpose(1, 2), lin_w[i], lin_b[i])             # [N, h*w, E]
        m = m.permute(0, 2, 1).reshape(n, -1, c.shape[2], c.shape[3])
        if i:
            uy, ux = bilinear_matrix(c.shape[2], hh, dt), bilinear_matrix(c.shape[3], ww, dt)
            m = torch.einsum('yh,nchw,xw->ncyx', uy, m, ux)
        maps.append(m)
    cat = torch.cat(maps, dim=1)                                                     # the [N, 4E, H, W] concat
    return torch.einsum('oc,nchw->nohw', fuse_w.reshape(fuse_w.shape[0], -1), cat)


# --------------------------------------------------------------------------- training loss
def resize_cross_entropy(logits, labels, ignore_index=255):
    """(sum of per-pixel losses, number of pixels whose arg-max equals the label) for logits [M,K,h,w] resized to the
    labels' [M,H,W] size: decode_head.py:805-835 (resize, bilinear, align_corners=False) -> cross_entropy_loss.py:9-40
    (F.cross_entropy(reduction='none', ignore_index): 0 at ignored pixels) -> accuracy.py:4-44 (top-1 over ALL pixels).
    Restated with the explicit resize matrices and a log-softmax."""
    H, W = labels.shape[1:]
    up = torch.einsum('yh,nchw,xw->ncyx', bilinear_matrix(logits.shape[2], H, logits.dtype), logits,
                      bilinear_matrix(logits.shape[3], W, logits.dtype))
    logp = up - torch.logsumexp(up, dim=1, keepdim=True)
    keep = labels != ignore_index
    safe = torch.where(keep, labels, torch.zeros_like(labels))
    nll = -logp.gather(1, safe.unsqueeze(1)).squeeze(1)
    loss_sum = torch.where(keep, nll, torch.zeros_like(nll)).sum()
    hits = (up.argmax(dim=1) == labels).sum()
    return loss_sum, hits


def head_losses(seg_logit, seg_label, ignore_index=255, loss_weight=1.0):
    """BaseDecodeHead_clips_flow.losses for the k+1 layout of the CFFM head (decode_head.py:744-835): 0.5 * CE(per-frame
    logits, every frame's labels) + CE(clip-level logits, last frame's labels), each a mean over all pixels; acc over frames."""
    b, t = seg_label.shape[:2]
    assert seg_logit.shape[1] == t + 1
    fl, cl = seg_logit[:, :t].flatten(0, 1), seg_logit[:, t:].flatten(0, 1)
    flab, clab = seg_label.flatten(0, 1).squeeze(1), seg_label[:, -1].squeeze(1)
    fs, fh = resize_cross_entropy(fl, flab, ignore_index)
    cs, _ = resize_cross_entropy(cl, clab, ignore_index)
    return 0.5 * loss_weight * fs / flab.numel() + loss_weight * cs / clab.numel(), fh.to(seg_logit.dtype) * (100.0 / flab.numel())
