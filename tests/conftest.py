import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is present and they were not deselected.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def __getitem__(self, k):
        return torch.from_numpy(self._z[k])

    def get(self, k, default=None):
        return torch.from_numpy(self._z[k]) if k in self._z.files else default

    def keys(self):
        return self._z.files

    def text(self, k):
        """A string entry (e.g. the JSON specs of a randomised fixture)."""
        return str(self._z[k])


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return load


def rel_err(y, ref):
    """Parity metric: max|y-ref|/max|ref| and relative L2."""
    y, ref = y.detach().double().cpu(), ref.detach().double().cpu()
    d = y - ref
    peak = float(d.abs().max() / ref.abs().max().clamp_min(1e-30))
    l2 = float(d.norm() / ref.norm().clamp_min(1e-30))
    return peak, l2


def assert_close(y, ref, tol=1e-5, what=""):
    assert y.shape == ref.shape, f"{what}: shape {tuple(y.shape)} vs {tuple(ref.shape)}"
    peak, l2 = rel_err(y, ref)
    assert peak <= tol and l2 <= tol, f"{what}: peak-rel {peak:.3e}, rel-L2 {l2:.3e} > {tol:g}"


def assert_parity(y, ref32, ref64, tol=1e-5, what=""):
    """Parity with the reference's fp32 output, with the float64 tie-breaker of SURVEY.md H6.

    Pass when y is within ``tol`` of the reference.  Where the reference's own fp32 result is
    farther than that from a float64 evaluation of the same formulas (its FFT/complex64 rounding
    amplified by a steep gain curve or a high-Q pole), y must instead be at least as close to
    the float64 result as the reference is (x1.5 slack) and within 10*tol of the reference -- or, when the
    reference's own noise is larger than that, within the triangle bound |y - f64| + |ref - f64|.
    """
    assert y.shape == ref32.shape, f"{what}: shape {tuple(y.shape)} vs {tuple(ref32.shape)}"
    peak, l2 = rel_err(y, ref32)
    if peak <= tol and l2 <= tol:
        return
    ours64, _ = rel_err(y, ref64)
    ref_noise, _ = rel_err(ref32, ref64)
    ok = ref_noise > 0.5 * tol and ours64 <= 1.5 * ref_noise and peak <= max(10 * tol, 1.05 * (ours64 + ref_noise))
    assert ok, (f"{what}: vs reference peak-rel {peak:.3e} / rel-L2 {l2:.3e} (tol {tol:g}); "
                f"vs float64: ours {ours64:.3e}, reference itself {ref_noise:.3e}")
