"""The C-ABI library builds, loads and exports exactly what include/grafx_amd.h declares (CPU only:
no compute call is made here)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "grafx_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gfx_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_header_symbol():
    from grafx_amd import _lib
    from grafx_amd.build import LIB, build

    build()
    handle = _lib.lib()
    declared = header_symbols()
    assert declared, "no symbols parsed from the header"
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/grafx_amd.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes signature table and header disagree"
    exported = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True).stdout
    assert set(re.findall(r"\b(gfx_[a-z0-9_]+)\b", exported)) >= set(declared)


def test_size_queries_need_no_gpu():
    from grafx_amd import _lib

    lib = _lib.lib()
    assert lib.gfx_abi_version() == 1
    assert lib.gfx_fftconv_nparts(4001) == 1 and lib.gfx_fftconv_nparts(8193) == 1
    assert lib.gfx_fftconv_nparts(8194) == 2 and lib.gfx_fftconv_nparts(60001) == 8
    assert lib.gfx_fir_spectrum_bytes(3, 4001) == 3 * 17 * 256 * 16
    assert lib.gfx_fftconv_workspace_bytes(2, 2, 131072, 131072, 0, 4001) == 0
    assert lib.gfx_fftconv_workspace_bytes(2, 2, 131072, 131072, 0, 60001) == 2 * 2 * (16 + 7) * 17 * 256 * 16
    assert lib.gfx_iir_fsm_plan_bytes(4001) == (8192 + 4096 + 2 * 2052) * 8 and lib.gfx_iir_fsm_plan_bytes(5000) == 0
    assert lib.gfx_istft_basis_bytes(384) == (388 * 384 + 193 * 2 * 208 + 2 * 384) * 4   # full basis, half basis, FFT factors
    # the odd-length aliasing: one row per transform needs (3P - 1) / 2 points (25 tiles at P = 135 071), two rows per
    # transform 2P - 1 per pair (35 tiles); the pair form ends at 2^24 points, even lengths do not alias at all
    P = 131072 + 4000 - 1
    assert lib.gfx_odd_alias_workspace_bytes(2, P) == 2 * 25 * 8192 * 8
    # (+ 256: one word per row, rounded up -- the rows' max |z|, from which the second row of a pair is scaled to the first's binade)
    assert lib.gfx_odd_alias_pair_workspace_bytes(2, P) == 35 * 8192 * 8 + 256 == lib.gfx_odd_alias_pair_workspace_bytes(1, P)
    assert lib.gfx_odd_alias_pair_workspace_bytes(3, P) == 2 * 35 * 8192 * 8 + 256
    assert lib.gfx_odd_alias_pair_workspace_bytes(1024, P) == 512 * 35 * 8192 * 8 + 4096
    assert lib.gfx_odd_alias_pair_plan_bytes(P) == (P + (P - 1) + P + 2 * 35 * 8192) * 8
    assert lib.gfx_odd_alias_pair_precise_plan_bytes(P) == 2 * lib.gfx_odd_alias_pair_plan_bytes(P)
    assert lib.gfx_odd_alias_pair_workspace_bytes(2, 258047) == 63 * 8192 * 8 + 256       # the last length of one column pass
    assert lib.gfx_odd_alias_pair_workspace_bytes(2, 258049) == 4 * 16 * 8192 * 8 + 256   # then one outer radix-4 level
    assert lib.gfx_odd_alias_pair_workspace_bytes(2, 480000 + 4000 - 1) == 4 * 30 * 8192 * 8 + 256   # BASELINE configs[1], 4000 taps
    assert lib.gfx_odd_alias_pair_plan_bytes(8388607) > 0 and lib.gfx_odd_alias_pair_plan_bytes(8388609) == 0
    assert lib.gfx_odd_alias_pair_plan_bytes(4000) == 0 and lib.gfx_odd_alias_plan_bytes(8388609) > 0


def test_processors_refuse_cpu_tensors_and_missing_library(monkeypatch):
    import torch

    import grafx_amd.processors as P

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.StereoGain()(torch.zeros(1, 2, 8), torch.zeros(1, 2))
    from grafx_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB", "/nonexistent/libgrafx_amd.so")
    with pytest.raises(ImportError, match="not built"):
        _lib.lib()


def test_plain_c_program_links_and_runs_against_the_library(tmp_path):
    """The boundary is a C ABI: a C translation unit must be able to include the header and link the library."""
    import os
    import subprocess

    from grafx_amd import build

    lib = build.build()
    exe = tmp_path / "abi_smoke"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(lib)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "abi_smoke.c"),
           "-L", libdir, "-lgrafx_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True)
    assert "abi_smoke ok" in out.stdout


def test_small_composite_dfts_on_the_host(tmp_path):
    """csrc/small_dft.hpp (the in-register DFTs of 3, 5, 6, 7, 9, ... 30 points behind the chirp-z column passes) is
    __host__ __device__: tests/c/small_dft_host.hip runs every supported size, forward and inverse, float and double,
    against a direct sum in double -- on the CPU, no GPU needed."""
    import os
    import subprocess

    from grafx_amd import build

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "small_dft_host"
    cmd = [build.HIPCC, "--offload-arch=gfx950", "-std=c++17", "-O1", os.path.join(root, "tests", "c", "small_dft_host.hip"),
           "-o", str(exe)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "SMALL_DFT_OK" in out.stdout, out.stdout + out.stderr
