"""CPU: the C-ABI library builds/loads without a GPU and exports exactly what include/golf_amd.h declares."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "golf_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(golf_[a-z0-9_]+)\s*\(", text)))


def test_header_matches_binding():
    from golf_amd import _lib

    assert header_symbols() == sorted(_lib.SIGNATURES)


def test_library_loads_and_exports_every_symbol():
    from golf_amd import _lib

    _lib.build()
    lib = _lib.load()  # getattr on every declared symbol
    assert lib.golf_abi_version() == _lib.ABI_VERSION == 6
    assert lib.golf_target_arch() == b"gfx950"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (golf_[a-z0-9_]+)", out))
    assert set(header_symbols()) <= exported
    # nothing but the C ABI leaks as a "golf_" symbol
    assert exported == set(header_symbols())


def test_argument_checks_without_gpu():
    """Bad arguments are rejected before any launch, so these calls are safe on a GPU-less host."""
    from golf_amd import _lib

    lib = _lib.load()
    assert lib.golf_ltv_allpole_workspace_bytes(32, 47761, 200, 22, 240) > 0
    assert lib.golf_ltv_allpole_workspace_bytes(0, 1, 1, 1, 1) == 0
    rc = lib.golf_ltv_allpole_fwd_f32(None, 0, None, None, None, 0, 2, 100, 3, 4, 10, None, 0, 0, None, None)
    assert rc == -1 and b"exceeds" in lib.golf_last_error()
    rc = lib.golf_ltv_allpole_fwd_f32(None, 0, None, None, None, 0, 2, 10, 3, 4, 8, None, 0, 0, None, None)
    assert rc == -1 and b"null" in lib.golf_last_error()
    rc = lib.golf_ltv_allpole_fwd_f32(None, 0, None, None, None, 0, 2, 10, 3, 99, 8, None, 0, 0, None, None)
    assert rc == -3
    rc = lib.golf_lti_frames_ola_fwd_f32(None, 0, None, None, None, None, 0, 1, 100, 5, 4, 8, 12, 0, None, 0, None)
    assert rc == -1
    rc = lib.golf_glottal_osc_fwd_f32(None, 0, 10, 1, None, 2, 8, None, 1, 16, 1, 0, None, 0, None, None, 0, 1, 10,
                                      None, 0, None, None, 0, 0, None)
    assert rc == -1
    # ABI 6: the source + transition-maps entry refuses what its two calls refuse (no launch), the fragment helpers are pure host logic
    rc = lib.golf_source_transitions_f32(None, 0, 10, 1, None, 2, 8, None, 1, 16, 4, 1, None, 129, None, 0, 1, 10, None, 0, None, 0, 0,
                                         None, None, 10, 2, 22, 240, None, 0, 2 | 128, None)
    assert rc == -1
    assert lib.golf_glottal_osc_tap_fragments_bytes(129, 4) == 2 * 4 * 16 * 64 * 4
    assert lib.golf_glottal_osc_tap_fragments_bytes(129, 2) == 0 and lib.golf_glottal_osc_tap_fragments_bytes(128, 4) == 0
    assert lib.golf_glottal_osc_tap_fragments_f32(None, 129, 4, None, 0, None) == -1
    assert lib.golf_glottal_osc_tap_fragments_f32(None, 129, 3, None, 0, None) == -3
    # peer-exchange entry points: null pointers / too many destinations are refused before anything touches a device
    assert lib.golf_peer_store_f32(None, 0, 1, 1, None, 0, 1, None) == -1 and b"null" in lib.golf_last_error()
    import ctypes
    many = (ctypes.c_void_p * 17)(*([1] * 17))
    assert lib.golf_peer_signal_u32(many, 17, 1, None) == -1 and b"at most" in lib.golf_last_error()
    assert lib.golf_peer_wait_u32(None, 1, 1, 1, 0, None, None) == -1
    assert lib.golf_peer_alloc(0, None) == -1


def test_ops_fail_loudly_on_cpu_tensors():
    import torch

    from golf_amd import functional as GF
    from golf_amd._lib import GolfError

    with pytest.raises(GolfError, match="no CPU path"):
        GF.ltv_allpole_ss(torch.zeros(1, 50), torch.ones(1, 3), torch.zeros(1, 3, 4), 24)
    with pytest.raises(GolfError):
        GF.glottal_osc(torch.zeros(1, 50), torch.zeros(1, 2), torch.zeros(4, 16), None, 1, 32)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no file under golf_amd/ may reference it."""
    pkg = os.path.join(ROOT, "golf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "libgolf_oracle" not in src, f
