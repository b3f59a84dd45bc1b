"""The three CFFM decode heads, registered under the reference's names with the reference's constructor
contract and state_dict keys, around the MI355X hot path.

* ``CFFMHead_clips_resize1_8``                      <- cffm_head.py:41-157
* ``CFFMHead_clips_resize1_8_gene_prototype``       <- cffm_head.py:161-300  (k-means prototypes per video)
* ``CFFMHead_clips_resize1_8_finetune_w_prototype3``<- cffm_head.py:304-535  (CFFM++)
* ``BaseDecodeHead_clips_flow``                     <- decode_head.py:513-835 (ctor contract, losses)
* ``CrossEntropyLoss`` / ``accuracy``               <- losses/cross_entropy_loss.py:141, losses/accuracy.py:4

For GPU tensors the whole head runs in libcffm_hip.so on token rows (``rows_impl='hip'``): the SegFormer embedding
(``ops.segformer_fuse``), ``linear_fuse``'s BatchNorm + ReLU + the 1/4 -> 1/8 resize (``ops.bn_relu_pool``), ``decoder_focal`` /
``decoder_swin`` (the hot path), the 1x1 classifiers (``ops.conv1x1``), the 1/8 -> 1/4 resize of the clip-level logits
(``ops.rows_resize``) and resize + cross entropy + accuracy (``ops.head_cross_entropy``); ``rows_impl='torch'``, CPU tensors and
non-default loss settings take the reference's op sequence in stock PyTorch around the hot path.
mmcv is absent on both boxes, so ``ConvModule`` / ``resize`` are re-provided with the same parameter names.
"""
import glob
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .modules import BasicLayer3d3, BasicLayer_cluster
from . import _lib
from .ops import bn_relu_pool, cat_into, cat_room, conv1x1, frame_logits_cat, late_params, head_cross_entropy, resize_cross_entropy, rows_resize, segformer_fuse
from .registry import HEADS, LOSSES, build_loss


def resize(input, size=None, scale_factor=None, mode='nearest', align_corners=None):
    """mmseg.ops.resize: F.interpolate (mmseg/ops/wrappers.py:8-29, minus its warning)."""
    return F.interpolate(input, size, scale_factor, mode, align_corners)


class ConvModule(nn.Module):
    """conv -> norm -> ReLU with mmcv's sub-module names (``conv``, ``bn``, ``activate``)."""

    def __init__(self, in_channels, out_channels, kernel_size, norm_cfg=None, act_cfg=dict(type='ReLU'), **kw):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, bias=norm_cfg is None, **kw)
        self.bn = None
        if norm_cfg is not None:
            kinds = {'BN': nn.BatchNorm2d, 'BN2d': nn.BatchNorm2d, 'SyncBN': nn.SyncBatchNorm}
            if norm_cfg['type'] not in kinds:
                raise KeyError('unsupported norm type %s' % norm_cfg['type'])
            self.bn = kinds[norm_cfg['type']](out_channels)
            for p in self.bn.parameters():
                p.requires_grad = norm_cfg.get('requires_grad', True)
        self.activate = nn.ReLU(inplace=True) if act_cfg is not None else None
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
        if self.conv.bias is not None:
            nn.init.zeros_(self.conv.bias)

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return self.activate(x) if self.activate is not None else x


def revert_sync_batchnorm(module):
    """SyncBatchNorm -> BatchNorm2d in place (what the reference's CPU tests do, tests/test_models/test_forward.py:186)."""
    for name, child in module.named_children():
        if isinstance(child, nn.SyncBatchNorm):
            bn = nn.BatchNorm2d(child.num_features, child.eps, child.momentum, child.affine, child.track_running_stats)
            bn.load_state_dict(child.state_dict())
            setattr(module, name, bn)
        else:
            revert_sync_batchnorm(child)
    return module


# ------------------------------------------------------------------------------------------------ losses
def accuracy(pred, target, topk=1):
    """Top-1 pixel accuracy in percent over ALL pixels (ignored ones count as wrong, as in the reference)."""
    if pred.size(0) == 0:
        return pred.new_tensor(0.)
    hit = pred.argmax(dim=1).eq(target)
    return hit.float().sum().reshape(1) * (100.0 / target.numel())


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        if use_sigmoid or use_mask:
            raise NotImplementedError('the CFFM configs use the softmax cross entropy only')
        self.reduction, self.class_weight, self.loss_weight = reduction, class_weight, loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, ignore_index=-100):
        cw = cls_score.new_tensor(self.class_weight) if self.class_weight is not None else None
        loss = F.cross_entropy(cls_score, label, weight=cw, reduction='none', ignore_index=ignore_index)
        if weight is not None:
            loss = loss * weight.float()
        red = reduction_override or self.reduction
        if avg_factor is not None:
            assert red == 'mean'
            loss = loss.sum() / avg_factor
        elif red == 'mean':
            loss = loss.mean()          # over all pixels, ignored ones included (cross_entropy_loss.py:18-30)
        elif red == 'sum':
            loss = loss.sum()
        return self.loss_weight * loss


# ------------------------------------------------------------------------------------------------ base class
class BaseDecodeHead_clips_flow(nn.Module):
    def __init__(self, in_channels, channels, *, num_classes, dropout_ratio=0.1, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), in_index=-1, input_transform=None,
                 loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0), decoder_params=None,
                 ignore_index=255, sampler=None, align_corners=False, num_clips=5):
        super().__init__()
        if input_transform is not None:
            assert input_transform in ('resize_concat', 'multiple_select')
            assert isinstance(in_channels, (list, tuple)) and isinstance(in_index, (list, tuple))
            assert len(in_channels) == len(in_index)
        else:
            assert isinstance(in_channels, int) and isinstance(in_index, int)
        if sampler is not None:
            raise NotImplementedError('pixel samplers are not used by the CFFM configs')
        self.input_transform, self.in_index = input_transform, in_index
        self.in_channels = sum(in_channels) if input_transform == 'resize_concat' else in_channels
        self.channels, self.num_classes, self.dropout_ratio = channels, num_classes, dropout_ratio
        self.conv_cfg, self.norm_cfg, self.act_cfg = conv_cfg, norm_cfg, act_cfg
        self.loss_decode = build_loss(loss_decode)
        self.ignore_index, self.align_corners, self.num_clips = ignore_index, align_corners, num_clips
        self.sampler = None
        self.conv_seg = nn.Conv2d(channels, num_classes, kernel_size=1)      # never used by the CFFM forward
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None
        self.fp16_enabled = False

    def init_weights(self):
        nn.init.normal_(self.conv_seg.weight, mean=0, std=0.01)
        nn.init.constant_(self.conv_seg.bias, 0)

    def _transform_inputs(self, inputs):
        if self.input_transform == 'multiple_select':
            return [inputs[i] for i in self.in_index]
        if self.input_transform == 'resize_concat':
            picked = [inputs[i] for i in self.in_index]
            return torch.cat([resize(x, size=picked[0].shape[2:], mode='bilinear', align_corners=self.align_corners)
                              for x in picked], dim=1)
        return inputs[self.in_index]

    def forward_train(self, inputs, img_metas, gt_semantic_seg, train_cfg, batch_size, num_clips, img=None):
        return self.losses(self.forward(inputs, batch_size, num_clips, img), gt_semantic_seg)

    def forward_test(self, inputs, img_metas, test_cfg, batch_size=None, num_clips=None, img=None):
        return self.forward(inputs, batch_size, num_clips, img)

    def cls_seg(self, feat):
        return self.conv_seg(self.dropout(feat) if self.dropout is not None else feat)

    # 'hip': resize + cross entropy + accuracy fused (ops.resize_cross_entropy) for GPU tensors when the configured loss is the
    # plain softmax cross entropy of every CFFM config (mean reduction, no class weights, align_corners False); 'torch': the
    # reference's op sequence (what CPU tensors and other loss settings get).
    loss_impl = 'hip'

    def _fused_loss_ok(self, seg_logit, size=None):
        """The fused resize + cross-entropy kernels cover what every CFFM config trains with; anything else -- other loss
        settings, CPU tensors, and shapes outside the kernels' limits (cffm_hip.h: 1 <= K <= 256 classes, upsampling by a
        factor 1..8 per side) -- takes the reference's op sequence in torch instead of raising."""
        ld = self.loss_decode
        ok = (self.loss_impl == 'hip' and (seg_logit.is_cuda or _lib._override is not None)
              and type(ld) is CrossEntropyLoss and ld.class_weight is None and ld.reduction == 'mean'
              and not self.align_corners and seg_logit.dtype == torch.float32)
        if ok and size is not None:
            k, h, w = seg_logit.shape[-3:]
            ok = 1 <= k <= 256 and h >= 1 and w >= 1 and h <= size[0] <= 8 * h and w <= size[1] <= 8 * w
        return ok

    def losses(self, seg_logit, seg_label):
        """0.5 * CE(per-frame logits, all frames) + CE(clip-level logits, last frame)  (decode_head.py:744-835).
        seg_logit [B, T+e, K, h, w] with e extra clip-level maps, seg_label [B, T, 1, H, W]."""
        assert seg_logit.dim() == 5 and seg_label.dim() == 5
        b, t = seg_label.shape[:2]
        n = seg_logit.shape[1]
        if n in (t + 1, t + 3):                       # k+1 / k+3: e clip-level maps, all judged on the last frame
            e, frame_labels = n - t, seg_label
        elif n in (2 * t, 2 * t + 1):                 # 2k / 2k+1: frame logits twice (minus one), labels repeated
            e, frame_labels = n - (2 * t - 1), torch.cat([seg_label, seg_label], 1)[:, :-1]
        else:
            raise AssertionError('unsupported logit layout %d for %d frames' % (n, t))
        size = seg_label.shape[3:]
        if self._fused_loss_ok(seg_logit, size) and seg_logit.dtype == torch.float32 and seg_label.dtype == torch.int64:
            # resize + cross entropy + accuracy of ALL maps in one forward and one backward kernel of libcffm_hip.so, on the logits
            # as they lie (the rows path hands token rows viewed as [B,n,K,h,w]): nothing resized, converted or re-assembled
            w = self.loss_decode.loss_weight
            nf, pix = n - e, float(b * size[0] * size[1])
            lidx = [i % t for i in range(nf)] + [t - 1] * e
            loss, hits = head_cross_entropy(seg_logit, seg_label.squeeze(2), lidx, [0.5 * w / (nf * pix)] * nf + [w / (e * pix)] * e,
                                            [100.0 / (nf * pix)] * nf + [0.0] * e, self.ignore_index)
            return dict(loss_seg=loss, acc_seg=hits.reshape(1))
        frame_logits = seg_logit[:, :n - e].flatten(0, 1)
        clip_logits = seg_logit[:, n - e:].flatten(0, 1)
        frame_labels = frame_labels.flatten(0, 1).squeeze(1)
        clip_labels = seg_label[:, -1:].expand(-1, e, -1, -1, -1).flatten(0, 1).squeeze(1)
        if self._fused_loss_ok(seg_logit, size):
            w = self.loss_decode.loss_weight
            fsum, fhits = resize_cross_entropy(frame_logits, frame_labels, self.ignore_index)
            csum, _ = resize_cross_entropy(clip_logits, clip_labels, self.ignore_index)
            loss = 0.5 * w * fsum / frame_labels.numel() + w * csum / clip_labels.numel()
            return dict(loss_seg=loss, acc_seg=(fhits * (100.0 / frame_labels.numel())).reshape(1))
        frame_logits = resize(frame_logits, size=size, mode='bilinear', align_corners=self.align_corners)
        clip_logits = resize(clip_logits, size=size, mode='bilinear', align_corners=self.align_corners)
        loss = 0.5 * self.loss_decode(frame_logits, frame_labels, weight=None, ignore_index=self.ignore_index) \
            + self.loss_decode(clip_logits, clip_labels, weight=None, ignore_index=self.ignore_index)
        return dict(loss_seg=loss, acc_seg=accuracy(frame_logits, frame_labels))


class MLP(nn.Module):
    """SegFormer linear embedding: [N,C,H,W] -> [N,H*W,embed] (cffm_head.py:26-37)."""

    def __init__(self, input_dim=2048, embed_dim=768):
        super().__init__()
        self.proj = nn.Linear(input_dim, embed_dim)

    def forward(self, x):
        return self.proj(x.flatten(2).transpose(1, 2))


def _focal_layer(embed_dim, depths):
    return BasicLayer3d3(dim=embed_dim, depth=depths, num_heads=8, window_size=7, mlp_ratio=4., qkv_bias=True,
                         qk_scale=None, drop=0., attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, pool_method='fc',
                         downsample=None, focal_level=2, focal_window=5, expand_size=3, use_conv_embed=False,
                         use_shift=False, use_pre_norm=False, use_checkpoint=False, focal_l_clips=[1, 2, 3],
                         focal_kernel_clips=[7, 5, 3])


class _CffmHeadBase(BaseDecodeHead_clips_flow):
    """What the three heads share: SegFormer MLP decoder, fuse conv, per-frame classifier, hot path."""

    def __init__(self, feature_strides, **kwargs):
        super().__init__(input_transform='multiple_select', **kwargs)
        assert len(feature_strides) == len(self.in_channels)
        assert min(feature_strides) == feature_strides[0]
        self.feature_strides = feature_strides
        c1, c2, c3, c4 = self.in_channels
        dp = kwargs['decoder_params']
        e = dp['embed_dim']
        self.linear_c4, self.linear_c3 = MLP(c4, e), MLP(c3, e)
        self.linear_c2, self.linear_c1 = MLP(c2, e), MLP(c1, e)
        self.linear_fuse = ConvModule(e * 4, e, kernel_size=1, norm_cfg=dict(type='SyncBN', requires_grad=True))
        self.linear_pred = nn.Conv2d(e, self.num_classes, kernel_size=1)
        self.linear_pred2 = nn.Conv2d(e * 2, self.num_classes, kernel_size=1)
        self.decoder_focal = _focal_layer(e, dp['depths'])

    # 'hip': the embedding + resize + concat + 1x1 conv collapse into per-scale GEMMs with composed weights and one
    # full-resolution pass in libcffm_hip.so (ops.segformer_fuse: no 1024-channel concat) for GPU tensors -- it raises when
    # the library is missing; 'torch': the reference's own op sequence in stock PyTorch (what CPU tensors get, like every
    # other part of the head outside libcffm_hip.so; also the A/B partner in the tests).
    fuse_impl = 'hip'

    def _fuse(self, inputs):
        """cffm_head.py:102-119: 4 x (linear embed -> resize to 1/4) -> concat -> 1x1 conv + BN + ReLU."""
        c1, c2, c3, c4 = self._transform_inputs(inputs)
        if self.fuse_impl == 'hip' and (c1.is_cuda or _lib._override is not None):   # (CPU tensors: config 1's CPU plumbing case)
            lins = (self.linear_c1, self.linear_c2, self.linear_c3, self.linear_c4)
            x = segformer_fuse([c1, c2, c3, c4], [l.proj.weight for l in lins], [l.proj.bias for l in lins],
                               self.linear_fuse.conv.weight)
            x = self.linear_fuse.bn(x)
            return self.linear_fuse.activate(x)
        n, size = c4.shape[0], c1.shape[2:]
        maps = []
        for lin, c in ((self.linear_c4, c4), (self.linear_c3, c3), (self.linear_c2, c2), (self.linear_c1, c1)):
            m = lin(c).permute(0, 2, 1).reshape(n, -1, c.shape[2], c.shape[3])
            maps.append(m if c is c1 else resize(m, size=size, mode='bilinear', align_corners=False))
        return self.linear_fuse(torch.cat(maps, dim=1))

    def _classify(self, conv, feat, clips=0, extra=0):
        """a 1x1 classifier (`linear_pred*`): libcffm_hip.so's GEMM on token rows for GPU tensors, nn.Conv2d otherwise"""
        if (self.fuse_impl == 'hip' and (feat.is_cuda or _lib._override is not None) and feat.dtype == torch.float32
                and conv.in_channels % 4 == 0 and conv.out_channels % 4 == 0):      # 16-byte token rows (e.g. not 19 classes)
            return conv1x1(feat, conv.weight, conv.bias, clips, extra)
        y = conv(feat)
        return y.reshape(clips, y.shape[0] // clips, *y.shape[1:]) if clips else y

    def _frame_logits(self, fused, batch_size, num_clips, dropped=False, extra=0):
        # (the [B,T,K,h,w] view comes straight out of the classifier op: its gradient -- slices of the loss kernel's buffer, one
        #  block of rows per clip -- is then consumed in place instead of being gathered into one [B*T,...] tensor by autograd)
        return self._classify(self.linear_pred, self.dropout(fused) if (self.dropout is not None and not dropped) else fused, batch_size, extra)

    def _clip_features(self, fused, batch_size, num_clips):
        """1/4 -> 1/8 resize, then the hot path (cffm_head.py:131-145)."""
        h, w = fused.shape[2:]
        small = resize(fused, size=(int(h / 2), int(w / 2)), mode='bilinear', align_corners=False)
        stack = small.reshape(batch_size, num_clips, -1, int(h / 2), int(w / 2))
        mined = self.decoder_focal(stack)
        assert mined.shape == stack.shape
        return stack, mined

    def _clip_logits(self, stack, mined, size):
        feat = torch.cat([stack[:, -1], mined[:, -1]], 1)
        x2 = self._classify(self.linear_pred2, self.dropout(feat) if self.dropout is not None else feat)
        return resize(x2, size=size, mode='bilinear', align_corners=False).unsqueeze(1)


    # ------------------------------------------------------------------ the fused row path around the hot path
    rows_impl = 'hip'       # 'torch': keep BatchNorm / ReLU / resize / the NCHW layer call in stock PyTorch (A/B partner in the tests)

    def _rows_path_ok(self, inputs):
        """embedding -> [BN + ReLU + 2x2-average clip stack] -> the layer on token rows -> classifiers on rows; needs libcffm_hip.so,
        fp32, a 1/4 map with even sides (then the 1/2 bilinear resize IS the 2x2 average) and a plain / Sync BatchNorm.  Everything
        else takes the reference's op sequence (`_fuse` etc.)."""
        c1 = inputs[self.in_index[0]] if isinstance(self.in_index, (list, tuple)) else inputs[0]
        bn = getattr(self.linear_fuse, 'bn', None)
        return (self.fuse_impl == 'hip' and self.rows_impl == 'hip' and (c1.is_cuda or _lib._override is not None)
                and c1.dtype == torch.float32 and c1.shape[2] % 2 == 0 and c1.shape[3] % 2 == 0 and c1.shape[2] >= 2 and c1.shape[3] >= 2
                and isinstance(bn, torch.nn.modules.batchnorm._BatchNorm) and bn.affine and bn.track_running_stats
                and isinstance(self.linear_fuse.activate, nn.ReLU))

    def _rows_front(self, inputs, batch_size, num_clips):
        """-> (fused [N,256,h,w] channels-last, clip-stack rows [N,h/2,w/2,256] or None, frame logits [B,T,K,h,w])."""
        c1, c2, c3, c4 = self._transform_inputs(inputs)
        lins = (self.linear_c1, self.linear_c2, self.linear_c3, self.linear_c4)
        y = segformer_fuse([c1, c2, c3, c4], [l.proj.weight for l in lins], [l.proj.bias for l in lins], self.linear_fuse.conv.weight)
        need_clip = self.training or num_clips == self.num_clips
        # Dropout2d in front of `linear_pred` (cffm_head.py:120) rides in the BatchNorm + ReLU pass: a [N,256] table of 0 / 1/(1-p)
        # factors (whole channels of a frame, as nn.Dropout2d draws them) instead of two passes over the 118 MB map
        mask = None
        if self.dropout is not None and self.training and self.dropout.p > 0:
            keep = 1.0 - self.dropout.p
            mask = torch.bernoulli(torch.full((y.shape[0], y.shape[1]), keep, dtype=torch.float32, device=y.device)) / keep
        fused, stack = bn_relu_pool(y, self.linear_fuse.bn, want_stack=need_clip, drop_mask=mask)
        return fused, stack, mask is not None

    def _rows_logits(self, conv, feat_rows, drop, size):
        """A 1x1 classifier on [B,h2,w2,C] token rows + the 1/8 -> 1/4 resize (cffm_head.py:147-149) -> [B,1,K,h,w]."""
        feat = feat_rows.permute(0, 3, 1, 2)                                                      # channels-last [B,C,h2,w2] view
        x2 = self._classify(conv, drop(feat) if drop is not None else feat)
        if x2.permute(0, 2, 3, 1).is_contiguous() and x2.shape[1] % 4 == 0 and x2.dtype == torch.float32:
            # the classifier wrote token rows: the resize stays on rows (torch's bilinear kernels take 18 + 55 us forward + backward on
            # this 14 MB map; k_rows_resize_* ~5 + 8) and its gradient is read out of the loss kernel's buffer in place
            return rows_resize(x2.permute(0, 2, 3, 1), size).permute(0, 3, 1, 2).unsqueeze(1)
        return resize(x2, size=size, mode='bilinear', align_corners=False).unsqueeze(1)

    @staticmethod
    def _rows_cat(x, x2):
        if cat_room(x, x2.shape[1]) and x2.shape[0] == x.shape[0] and x2.shape[2:] == x.shape[2:]:
            return cat_into(x, x2)      # the frame logits were written with room behind them: only x2 moves
        if x.permute(0, 1, 3, 4, 2).is_contiguous() and x2.permute(0, 1, 3, 4, 2).is_contiguous():
            # the classifiers wrote token rows [.., h, w, K]: concatenate THERE (a plain copy; torch.cat of the [B,T,K,h,w] views
            # would transpose 71 MB into plain memory) and hand the result out as the [B,T+1,K,h,w] view the caller expects --
            # `losses` reads it as it lies, and cat's backward is two views of the loss kernel's gradient buffer
            rows = torch.cat([x.permute(0, 1, 3, 4, 2), x2.permute(0, 1, 3, 4, 2)], 1)
            return rows.permute(0, 1, 4, 2, 3)
        return torch.cat([x, x2], 1)


@HEADS.register_module()
class CFFMHead_clips_resize1_8(_CffmHeadBase):
    def _forward_rows(self, inputs, batch_size, num_clips):
        # (the frame classifier's parameters as they will be used at the END of the forward, through a node created FIRST: its backward runs
        #  late and joins the deferred branch before autograd accumulates their gradients -- ops._LateGradFn)
        lp_w, lp_b = late_params(self.linear_pred.weight, self.linear_pred.bias) if self.training else (self.linear_pred.weight, self.linear_pred.bias)
        fused, stack, dropped = self._rows_front(inputs, batch_size, num_clips)
        # Training with the clip-level path: the frame classifier `linear_pred` runs LAST, fused with the concatenation (ops.frame_logits_cat),
        # so that its backward is the first node of the backward pass and runs on the library's deferred branch beside the layer's backward.
        need_drop = self.dropout is not None and not dropped and self.dropout.p > 0          # (Dropout2d not yet folded into the BatchNorm pass)
        late = (self.training and stack is not None and not need_drop and fused.dtype == torch.float32
                and self.linear_pred.in_channels % 4 == 0 and self.linear_pred.out_channels % 4 == 0)
        x = None if late else self._frame_logits(fused, batch_size, num_clips, dropped=dropped, extra=1 if (self.training and stack is not None) else 0)
        if stack is None:
            return x[:, -1]                                   # short-circuit before CFFM (cffm_head.py:127-129)
        h, w = fused.shape[2:]
        h2, w2 = h // 2, w // 2
        x_rows = stack.view(batch_size, num_clips, h2 * w2, stack.shape[-1])
        mined = self.decoder_focal.forward_rows(x_rows, h2, w2)                                  # [B, h2*w2, 256]
        x2 = self._rows_logits(self.linear_pred2, torch.cat([x_rows[:, -1], mined], dim=-1).view(batch_size, h2, w2, -1), self.dropout, (h, w))
        if not self.training:
            return x2.squeeze(1)
        if late:
            return frame_logits_cat(fused, lp_w, lp_b, x2, batch_size)
        return self._rows_cat(x, x2)

    def forward(self, inputs, batch_size=None, num_clips=None, imgs=None):
        if self.training:
            assert self.num_clips == num_clips
        if self._rows_path_ok(inputs):
            return self._forward_rows(inputs, batch_size, num_clips)
        fused = self._fuse(inputs)
        x = self._frame_logits(fused, batch_size, num_clips)
        if not self.training and num_clips != self.num_clips:
            return x[:, -1]                                   # short-circuit before CFFM (cffm_head.py:127-129)
        stack, mined = self._clip_features(fused, batch_size, num_clips)
        x2 = self._clip_logits(stack, mined, fused.shape[2:])
        if not self.training:
            return x2.squeeze(1)
        return torch.cat([x, x2], 1)                          # [B, T+1, classes, h, w]


def _kmeans(x, k, iters=10):
    """Plain euclidean k-means with random initial centroids, like fast_pytorch_kmeans.KMeans(mode='euclidean',
    max_iter=10) which the reference calls (cffm_head.py:280-282; third-party, version unpinned, random init ->
    parity unpinned; outside the measured path)."""
    centers = x[torch.randperm(x.shape[0], device=x.device)[:k]].clone()
    for _ in range(iters):
        assign = torch.cdist(x, centers).argmin(dim=1)
        for j in range(k):
            sel = assign == j
            if sel.any():
                centers[j] = x[sel].mean(dim=0)
    return centers


@HEADS.register_module()
class CFFMHead_clips_resize1_8_gene_prototype(_CffmHeadBase):
    def __init__(self, feature_strides, **kwargs):
        super().__init__(feature_strides, **kwargs)
        self.n_clusters, self.save_path = 100, './cluster_centers/'

    def forward_test(self, inputs, img_metas, test_cfg, batch_size=None, num_clips=None, img=None):
        return self.forward(inputs, batch_size, num_clips, img, img_metas)

    def forward(self, inputs, batch_size=None, num_clips=None, imgs=None, img_metas=None):
        if self.training:
            assert self.num_clips == num_clips
        fused = self._fuse(inputs)
        x = self._frame_logits(fused, batch_size, num_clips)
        assert batch_size == 1
        h, w = fused.shape[2:]
        small = resize(fused, size=(int(h / 2), int(w / 2)), mode='bilinear', align_corners=False)
        feats = small.reshape(batch_size, num_clips, -1, int(h / 2), int(w / 2)).permute(0, 1, 3, 4, 2)
        feats = feats.reshape(batch_size, -1, feats.shape[-1])
        with torch.no_grad():
            centers = torch.stack([_kmeans(feats[i], self.n_clusters) for i in range(batch_size)], dim=0)
        video = img_metas[0]['filename'].split('/')[-3]
        os.makedirs(self.save_path + video, exist_ok=True)
        torch.save(centers, self.save_path + video + '/centers.pt')
        if not self.training:
            return x[:, -1]


@HEADS.register_module()
class CFFMHead_clips_resize1_8_finetune_w_prototype3(_CffmHeadBase):
    def __init__(self, feature_strides, **kwargs):
        super().__init__(feature_strides, **kwargs)
        e = kwargs['decoder_params']['embed_dim']
        focal = self._modules.pop('decoder_focal')           # keep the reference's registration (state_dict) order
        self.linear_pred3 = nn.Conv2d(e, self.num_classes, kernel_size=1)
        self.decoder_focal = focal
        self.n_clusters, self.save_path = 10, './cluster_centers/'
        self.dropout3 = nn.Dropout2d(self.dropout_ratio)
        self.decoder_swin = BasicLayer_cluster(dim=e, depth=1, num_heads=8, window_size=7, mlp_ratio=4., qkv_bias=True,
                                               qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                                               norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False)
        self.finetune = True

    def forward_train(self, inputs, img_metas, gt_semantic_seg, train_cfg, batch_size, num_clips, img=None):
        return self.losses(self.forward(inputs, batch_size, num_clips, img, img_metas), gt_semantic_seg)

    def forward_test(self, inputs, img_metas, test_cfg, batch_size=None, num_clips=None, img=None):
        return self.forward(inputs, batch_size, num_clips, img, img_metas)

    def _load_centers(self, img_metas, device, keep=0.8):
        """Per-video prototype tensors [1,K,C] (cffm_head.py:430-455); several files -> random 80 % subset."""
        out = []
        for meta in img_metas:
            video = meta['filename'].split('/')[-3]
            path = self.save_path + video + '/centers.pt'
            if os.path.isfile(path):
                out.append(torch.load(path, map_location='cpu'))
                continue
            parts = torch.cat([torch.load(p, map_location='cpu') for p in glob.glob(self.save_path + video + '/*.pt')], dim=1)
            assert parts.dim() == 3 and parts.shape[0] == 1, parts.shape
            pick = torch.topk(torch.rand(parts.shape[1]), int(parts.shape[1] * keep))[1].sort()[0]
            out.append(parts[:, pick])
        return torch.cat(out, dim=0).to(device)

    def _forward_rows(self, inputs, batch_size, num_clips, centers):
        """The same head on token rows (missing 3): the frozen embedding + BatchNorm + ReLU + clip stack in one pass, the CFFM
        layer and the prototype layer on rows (the target frame's stack rows ARE the tokens `decoder_swin` takes: no permute / reshape),
        all three classifiers and both resizes on rows."""
        with torch.no_grad():                                  # the fuse conv is frozen in eval (cffm_head.py:478-480)
            self.linear_fuse.eval()
            fused, stack, dropped = self._rows_front(inputs, batch_size, num_clips)
        x = self._frame_logits(fused, batch_size, num_clips, dropped=dropped, extra=1 if (self.training and stack is not None) else 0)
        if stack is None:
            return x[:, -1]
        h, w = fused.shape[2:]
        h2, w2 = h // 2, w // 2
        c = stack.shape[-1]
        x_rows = stack.view(batch_size, num_clips, h2 * w2, c)
        mined = self.decoder_focal.forward_rows(x_rows, h2, w2)
        x2 = self._rows_logits(self.linear_pred2, torch.cat([x_rows[:, -1], mined], dim=-1).view(batch_size, h2, w2, -1), self.dropout, (h, w))
        if self.finetune:                                      # the CFFM branch is detached (cffm_head.py:514-518)
            x_rows, x, x2 = x_rows.detach(), x.detach(), x2.detach()
        ctx = self.decoder_swin(x_rows[:, -1].contiguous(), h2, w2, centers)[0]                 # [B, h2*w2, C]
        x3 = self._rows_logits(self.linear_pred3, ctx.reshape(batch_size, h2, w2, c), self.dropout3, (h, w))
        if not self.training:
            return x2.squeeze(1) + 0.5 * x3.squeeze(1)
        return self._rows_cat(x, x3)

    def forward(self, inputs, batch_size=None, num_clips=None, imgs=None, img_metas=None):
        assert batch_size == len(img_metas)
        centers = self._load_centers(img_metas, inputs[0].device)
        if self.training:
            assert self.num_clips == num_clips
        if self._rows_path_ok(inputs):
            return self._forward_rows(inputs, batch_size, num_clips, centers)
        with torch.no_grad():                                  # the fuse conv is frozen in eval (cffm_head.py:478-480)
            self.linear_fuse.eval()
            fused = self._fuse(inputs)
        x = self._frame_logits(fused, batch_size, num_clips)
        if not self.training and num_clips != self.num_clips:
            return x[:, -1]
        stack, mined = self._clip_features(fused, batch_size, num_clips)
        x2 = self._clip_logits(stack, mined, fused.shape[2:])
        if self.finetune:                                      # the CFFM branch is detached (cffm_head.py:514-518)
            stack, x, x2 = stack.detach(), x.detach(), x2.detach()
        b, _, c, h2, w2 = stack.shape
        tokens = stack[:, -1].permute(0, 2, 3, 1).reshape(b, h2 * w2, c)
        ctx = self.decoder_swin(tokens, h2, w2, centers)[0]
        ctx = ctx.reshape(b, h2, w2, c).permute(0, 3, 1, 2)
        x3 = resize(self._classify(self.linear_pred3, self.dropout3(ctx)), size=fused.shape[2:], mode='bilinear', align_corners=False)
        x3 = x3.unsqueeze(1)
        if not self.training:
            return x2.squeeze(1) + 0.5 * x3.squeeze(1)
        return torch.cat([x, x3], 1)
