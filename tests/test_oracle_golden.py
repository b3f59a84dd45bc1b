"""The oracle against the golden vectors the reference produced (tests/golden/make_golden.py).
Runs everywhere (CPU); this is what pins the oracle on the GPU box, where /root/reference is absent."""
import pytest
import torch

from oracle import cffm_oracle as O, recipe as R
from tests import helpers as H

TOL = 1e-5  # fp32 oracle vs fp32 reference: SURVEY.md 8(d) tolerance ladder


@pytest.mark.parametrize('case', H.LAYER_CASES)
def test_layer_forward_backward(case):
    g = H.load_golden(case)
    b, h, w, depth, st, x, gy = H.layer_case_inputs(g)
    if h * w > 1024:
        torch.set_num_threads(max(1, torch.get_num_threads()))
    st = {k: v.requires_grad_(True) for k, v in st.items()}
    x.requires_grad_(True)
    y = O.layer_forward(x, st, depth)
    assert torch.equal(y[:, :-1], x[:, :-1])          # reference frames pass through untouched
    H.check_layer_forward(g, y[:, -1].detach(), TOL)
    (y[:, -1] * gy).sum().backward()
    H.check_layer_backward(g, x.grad, {k: v.grad for k, v in st.items()}, 2e-5)


@pytest.mark.parametrize('case', H.GTC_CASES)
def test_gtc(case):
    g = H.load_golden(case)
    b, h, w, k = [int(v) for v in g['meta']]
    st = {kk: v.requires_grad_(True) for kk, v in R.gtc_layer_state(1, seed=3).items()}
    x = R.synth_input('gx', (b, h * w, 256), seed=4).requires_grad_(True)
    c = R.synth_input('gc', (b, k, 256), seed=5).requires_grad_(True)
    gg = R.synth_input('gg', (b, h * w, 256), seed=6, scale=1.0)
    y = O.gtc_layer_forward(x, h, w, c, st, 1)
    assert H.rel_err(y.detach(), g['y']) < TOL
    (y * gg).sum().backward()
    assert H.rel_err(x.grad, g['dx']) < 2e-5
    assert H.rel_err(c.grad, g['dc']) < 2e-5
    # parameters the reference leaves without gradient (table + proj unused)
    no_grad = set(str(s) for s in g['no_grad_keys'])
    for kk, v in st.items():
        if kk in no_grad:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0


def test_index_maps_shape():
    assert O.ring_pixels(14, 14).shape == (4, 132)
    assert O.window_pixels(63, 63).shape == (81, 49)
    cells = O.unfold_cells(9, 9, 3, 3, 1)
    assert cells.shape == (81, 9) and (cells[0] >= 0).sum() == 4      # stride-3/pad-1 unfold is off-centre
    # 12 ring positions are counted twice
    r = O.ring_pixels(63, 63)[40]
    assert len(set(r.tolist())) == 120
