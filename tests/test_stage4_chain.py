"""stage4.chain_forward on the CPU (stock-torch passes, oracle/torch_stock.py): the stacked rec || cv decoder pass and the way its
output is split (stage4._SplitRows) give the loss and gradients of the two separate passes of the reference
(train_gru_cyclevae_gauss_batch.py:1335-1336)."""
import numpy as np
import torch

import stage4
import synth
from train_util import cpu_step, make_masks


def test_stacked_decoder_pass_equals_the_two_separate_passes():
    P = synth.CycleVAEProblem(B=3, T=8, in_dim=10, out_dim=6, lat_dim=4, hidden=32, n_cyc=2, bias_scale=0.1, tag="splitchk")
    masks = make_masks(P, 4, 6)
    l0, g0 = cpu_step(P, masks, 2, False)
    l1, g1 = cpu_step(P, masks, 2, True)
    assert abs(l0 - l1) <= 1e-6 * abs(l0)
    for k in g0:
        for n in g0[k]:
            assert np.max(np.abs(g0[k][n] - g1[k][n])) <= 1e-5 * (np.max(np.abs(g0[k][n])) + 1e-30), (k, n)


def test_split_rows_backward_is_the_concatenation_and_tolerates_an_unused_half():
    out = torch.randn(6, 5, 4, requires_grad=True)
    a, b = stage4._SplitRows.apply(out, 2)
    assert a.shape == (2, 5, 4) and b.shape == (4, 5, 4)
    wa, wb = torch.randn_like(a), torch.randn_like(b)
    ((a * wa).sum() + (b * wb).sum()).backward()
    assert torch.equal(out.grad, torch.cat((wa, wb), 0))
    out.grad = None
    a, b = stage4._SplitRows.apply(out, 2)
    (b * wb).sum().backward()                          # rec unused: its half of the gradient is zero
    assert torch.equal(out.grad, torch.cat((torch.zeros_like(wa), wb), 0))
