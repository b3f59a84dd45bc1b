"""Loading the reference's checkpoints into this package's heads.

The reference saves / loads through mmcv: a checkpoint file is ``{'meta': {...CLASSES, PALETTE, config, mmseg_version},
'state_dict': {...}, 'optimizer': {...}}`` whose state-dict keys carry the segmentor's attribute prefixes (``backbone.``,
``decode_head.``) and, when saved from a DistributedDataParallel wrapper, a leading ``module.`` (tools/test.py:133-135 loads
with ``load_checkpoint(model, path, map_location='cpu')``, non-strict; tools/convert_model.py:21-44 walks the same layout;
``--load-from`` feeds CFFM weights to the CFFM++ fine-tuning run, README.md:105).  The parameter / buffer names of this
package's heads equal the reference's (tests/test_boundary.py), so loading is a matter of the container format and prefixes.
"""
import torch


def extract_state_dict(checkpoint):
    """The tensor dict of an mmcv-format checkpoint (or of a bare state dict), without a leading ``module.``."""
    sd = checkpoint
    if isinstance(checkpoint, dict) and isinstance(checkpoint.get('state_dict'), dict):
        sd = checkpoint['state_dict']
    if not isinstance(sd, dict) or not all(isinstance(k, str) for k in sd):
        raise ValueError('not a checkpoint: expected a state dict or a dict with a "state_dict" entry')
    out = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items() if torch.is_tensor(v)}
    if not out:
        raise ValueError('not a checkpoint: no tensors in the state dict')
    return out


def load_reference_checkpoint(module, checkpoint, prefix='decode_head.', strict=False, map_location='cpu', trusted=False):
    """Load a reference (mmcv-format) checkpoint into `module` -- one of this package's decode heads by default.

    checkpoint: a path (``torch.load``-ed to `map_location`) or an already loaded dict.
    prefix: the sub-tree of the segmentor to take, with the prefix stripped (``'decode_head.'`` for a head, ``'backbone.'`` for
        a backbone, ``''`` for the whole segmentor).  Keys that already lack every segmentor prefix (a head-only state dict)
        are taken as they are.
    strict=False mirrors mmcv's ``load_checkpoint``: keys that are missing / unexpected / of a different shape are reported,
    not raised.  Returns ``(missing_keys, unexpected_keys, meta)``; `meta` is the checkpoint's ``meta`` dict (CLASSES, PALETTE,
    config text) or ``{}``.
    trusted: a checkpoint FILE is read with ``torch.load(weights_only=True)``.  mmcv metas may hold python objects that the safe
        unpickler refuses; only ``trusted=True`` (the caller vouches for the file: unpickling runs arbitrary code) retries with
        ``weights_only=False`` -- and only after an unpickling error, never to mask a missing / corrupt file."""
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, '__fspath__'):
        import pickle
        try:
            checkpoint = torch.load(checkpoint, map_location=map_location, weights_only=True)
        except pickle.UnpicklingError as e:
            if not trusted:
                raise pickle.UnpicklingError('%s\n(the checkpoint holds objects the safe loader refuses; pass trusted=True to '
                                             'load_reference_checkpoint if the file comes from a source you trust)' % e) from e
            checkpoint = torch.load(checkpoint, map_location=map_location, weights_only=False)
    sd = extract_state_dict(checkpoint)
    meta = checkpoint.get('meta', {}) if isinstance(checkpoint, dict) and isinstance(checkpoint.get('meta'), dict) else {}
    if prefix:
        sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        if not sub and not any(k.startswith(('backbone.', 'decode_head.', 'auxiliary_head.', 'neck.')) for k in sd):
            sub = sd                      # a bare head / layer state dict
        sd = sub
    own = module.state_dict()
    mismatched = [k for k, v in sd.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
    if strict:
        if mismatched:
            raise RuntimeError('shape mismatch for ' + ', '.join(mismatched))
        res = module.load_state_dict(sd, strict=True)
        return list(res.missing_keys), list(res.unexpected_keys), meta
    res = module.load_state_dict({k: v for k, v in sd.items() if k not in mismatched}, strict=False)
    return sorted(set(res.missing_keys) | set(mismatched)), list(res.unexpected_keys), meta
