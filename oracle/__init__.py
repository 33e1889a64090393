"""CPU oracle for the grafx hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (CPU) restatement of the reference algorithms that sit on the
hot path.  It is used only by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg as the
checker / reported baseline.  ``grafx_amd`` never imports it.

Parity status: PINNED against outputs of the reference itself, generated in the
build container by ``tests/golden/make_golden.py`` (which imports
``/root/reference/src`` with four third-party shims) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every oracle
function against them.  One exception is stated where it applies:
``ballistics`` restates the published algorithm of the third-party
``torchcomp.compressor_core`` (Yu et al., DAFx 2024; unpinned upstream, wheel
absent here, no upstream golden vector) — "parity unpinned" for that function
only; see its docstring.
"""
from .lti import convolve, iir_fsm_fir, iir_fsm, one_pole_fir, truncated_one_pole  # noqa: F401
from .processors import (  # noqa: F401
    OracleBallistics,
    OracleBiquadFilter,
    OracleCompressor,
    OracleNoiseGate,
    OracleParametricEqualizer,
    OracleSTFTMaskedNoiseReverb,
    OracleStereoGain,
    ballistics,
    ballistics_coefficients,
    ballistics_docstring_reading,
    lr_to_ms,
    ms_to_lr,
    normalize_impulse,
    peq_biquad_coefficients,
    biquad_coefficients,
)
