"""The C-ABI library loads and exports every symbol include/lic.h declares; argument validation
returns status codes without touching a GPU.  CPU only (no compute launches)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lic.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lic_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    from neural_image_compression_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib


def test_header_symbols_all_exported_and_bound(lib):
    syms = declared_symbols()
    assert len(syms) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (lic_[a-z0-9_]+)", out))
    missing = [s for s in syms if s not in exported]
    assert not missing, f"declared in lic.h but not exported: {missing}"
    unbound = [s for s in syms if s not in lib.SIGNATURES]
    assert not unbound, f"declared in lic.h but no ctypes signature: {unbound}"
    extra = [s for s in exported if s not in syms]
    assert not extra, f"exported but undeclared: {extra}"


def test_load_and_version(lib):
    L = lib.load()
    assert L.lic_version() == 2
    assert L.lic_arch() == b"gfx950"


def test_argument_validation_without_gpu(lib):
    L = lib.load()
    assert L.lic_igemm(None, None) == -1
    d = lib.IgemmDesc()
    assert L.lic_igemm(ctypes.byref(d), None) == -1            # null pointers
    assert L.lic_wgrad(None, None, 0, None) == -1
    assert L.lic_wgrad_workspace_bytes(None) == 0
    assert L.lic_permute3(None, None, 1, 1, 1, 1, 1, 1, 1, 1, 1, None) == -1
    assert L.lic_pack_weight(None, None, 1, 1, 1, 1, 1, 1, None) == -1
    assert L.lic_packed_weight_floats(25, 192, 192) == 25 * 12 * 192 * 16
    assert L.lic_packed_weight_floats(1, 80, 75) == 5 * 96 * 16
    assert L.lic_quantize(None, None, None, 4, 1, None) == -1
    assert L.lic_gmm_likelihood_fwd(None, None, None, None, 1, 1, 1, 1e-9, None) == -1
    assert L.lic_factorized_fwd(None, None, None, None, 1, 1, 1e-9, None) == -1
    assert L.lic_rd_loss_workspace_bytes(32) == 32 * 64 * 3 * 8
    assert L.lic_colsum_workspace_bytes(0, 4) == 0


def test_check_raises_with_status_name(lib):
    with pytest.raises(lib.LicError, match="LIC_ERR_INVALID"):
        lib.check(-1, "probe")
