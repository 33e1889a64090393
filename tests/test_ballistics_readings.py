"""Ballistics (core/envelope.py:63-101) has TWO readings and no external pin (torchcomp is absent):
"T" -- the recursion of torchcomp.compressor_core as the reference's code calls it (implemented: oracle.ballistics,
gfx_ballistics_f32) -- and "D" -- the reference docstring's formula (oracle.ballistics_docstring_reading).
Constant inputs above / below the initial state y[-1] = 1 have closed forms that tell them apart:

    x = 0 (below):  T: y[n] = (1 - sigmoid(z0))^(n+1)          D: y[n] = sigmoid(z1)^(n+1)
    x = 2 (above):  T: y[n] = 2 - (1 - sigmoid(z1))^(n+1)      D: y[n] = 2 - sigmoid(z0)^(n+1)

Every test here is `provisional` (named so): they document which reading is implemented, they pin nothing."""
import pytest
import torch

import oracle

Z = torch.tensor([[-1.0, 2.0], [0.5, -3.0], [3.0, 0.25]])
N = 64


def _closed_forms():
    n = torch.arange(1, N + 1, dtype=torch.float64)
    s0, s1 = torch.sigmoid(Z[:, 0].double())[:, None], torch.sigmoid(Z[:, 1].double())[:, None]
    return {"T_below": (1 - s0) ** n, "T_above": 2 - (1 - s1) ** n, "D_below": s1 ** n, "D_above": 2 - s0 ** n}


def test_ballistics_closed_forms_of_both_readings():
    cf = _closed_forms()
    below, above = torch.zeros(3, N, dtype=torch.float64), torch.full((3, N), 2.0, dtype=torch.float64)
    z = Z.double()
    assert torch.allclose(oracle.ballistics(below, z), cf["T_below"], rtol=1e-12)
    assert torch.allclose(oracle.ballistics(above, z), cf["T_above"], rtol=1e-12)
    assert torch.allclose(oracle.ballistics_docstring_reading(below, z), cf["D_below"], rtol=1e-12)
    assert torch.allclose(oracle.ballistics_docstring_reading(above, z), cf["D_above"], rtol=1e-12)
    # and they ARE different functions of z_alpha
    assert (cf["T_below"] - cf["D_below"]).abs().max() > 0.1 and (cf["T_above"] - cf["D_above"]).abs().max() > 0.1


@pytest.mark.gpu
def test_ballistics_kernel_implements_reading_T_not_D():
    from grafx_amd.processors import Ballistics

    cf = _closed_forms()
    for name, level in (("below", 0.0), ("above", 2.0)):
        x = torch.full((3, N), level)
        with torch.no_grad():
            y = Ballistics()(x.cuda(), Z.cuda()).cpu().double()
        assert (y - cf["T_" + name]).abs().max() <= 1e-6, name
        assert (y - cf["D_" + name]).abs().max() > 0.1, name
