"""Direct pin: the oracle against the reference's own modules imported from /root/reference.
Skipped where the reference tree is absent (the GPU box); the golden vectors cover that case."""
import pytest
import torch

from oracle import cffm_oracle as O, recipe as R, ref_import as RI
from tests import helpers as H

pytestmark = pytest.mark.skipif(not RI.available(), reason='/root/reference not present')


@pytest.mark.parametrize('b,h,w,depth', [(2, 8, 8, 1), (1, 14, 21, 2), (1, 13, 30, 2), (1, 9, 16, 1)])
def test_layer_fp64_bit_level(b, h, w, depth):
    m = RI.build_basic_layer(depth).double()
    st = R.layer_state(depth, seed=11, dtype=torch.float64)
    m.load_state_dict(st, strict=False)
    x = R.synth_input('x', (b, 4, 256, h, w), seed=12, dtype=torch.float64)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = m(xa), O.layer_forward(xb, st, depth)
    assert H.rel_err(yb, ya) < 1e-13
    ya[:, -1].square().sum().backward()
    yb[:, -1].square().sum().backward()
    assert H.rel_err(xb.grad, xa.grad) < 1e-12
    assert float(xa.grad[:, 0].abs().max()) > 0      # reference frames do receive gradient (A.10)


def test_negative_control_ring_bias_matters():
    m = RI.build_basic_layer(1).double()
    st = R.layer_state(1, seed=11, dtype=torch.float64)
    m.load_state_dict(st, strict=False)
    x = R.synth_input('x', (1, 4, 256, 8, 8), seed=12, dtype=torch.float64)
    st2 = dict(st)
    k = 'blocks.0.attn.relative_position_bias_table_to_neighbors'
    st2[k] = st[k].flip(-1)
    assert H.rel_err(O.layer_forward(x, st2, 1), m(x)) > 1e-4


@pytest.mark.parametrize('b,h,w,k', [(2, 8, 8, 8), (1, 13, 30, 100)])
def test_gtc_fp64(b, h, w, k):
    m = RI.build_cluster_layer(1).double()
    st = R.gtc_layer_state(1, seed=13, dtype=torch.float64)
    m.load_state_dict(st, strict=False)
    x = R.synth_input('gx', (b, h * w, 256), seed=14, dtype=torch.float64)
    c = R.synth_input('gc', (b, k, 256), seed=15, dtype=torch.float64)
    assert H.rel_err(O.gtc_layer_forward(x, h, w, c, st, 1), m(x, h, w, c)[0]) < 1e-13


def test_t2_raises_like_reference():
    m = RI.build_basic_layer(1)
    with pytest.raises(IndexError):
        m(torch.zeros(1, 2, 256, 8, 8))
    with pytest.raises(AssertionError):
        O.layer_forward(torch.zeros(1, 2, 256, 8, 8), R.layer_state(1), 1)
