"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/sincformer_hip.h declares, with the arity the ctypes binding assumes."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sincformer_hip.h")


def _header_decls():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:long long|int)\s+(sfm_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = n
    return decls


@pytest.fixture(scope="module")
def built():
    from sincformer_metacog_speech_enhancement_amd import build, lib
    build.build(verbose=False)
    return lib


def test_header_declares_the_bound_symbols(built):
    decls = _header_decls()
    assert len(decls) >= 18
    assert set(decls) == set(built.SIGNATURES), set(decls) ^ set(built.SIGNATURES)
    for name, n in decls.items():
        assert n == len(built.SIGNATURES[name]), (name, n, len(built.SIGNATURES[name]))


def test_library_exports_every_symbol(built):
    L = built.load()
    raw = ctypes.CDLL(built.LIB_PATH)
    for name in _header_decls():
        assert hasattr(raw, name), name
    assert L.sfm_abi_version() == 1


def test_missing_library_is_loud(monkeypatch, built):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "LIB_PATH", "/nonexistent/libsincformer_hip.so")
    with pytest.raises(built.HipExtensionMissing):
        built.load()


def test_null_pointers_are_rejected_without_a_gpu(built):
    """argument validation happens before any HIP call, so it can be exercised on CPU"""
    L = built.load()
    assert L.sfm_attention_fwd(None, None, 1, 1, 1, 64, 192, 64, 64, 128, 192, 64, 0.125, 0, None) == -1
    assert L.sfm_layernorm(None, None, None, None, None, 4, 256, 256, 256, 256, 1e-5, 0, 0, None) == -1
    one = ctypes.c_void_p(16)
    assert L.sfm_gemm16(one, one, None, one, None, None, 1, 8, 8, 12, 12, 1, 1, 0, 0, 32, 8, 64, 8, 0, 0, 0, 1.0, 0, 1,
                        0, 0, 0, None) == -2       # Cin not a multiple of 8
    assert L.sfm_bilstm_layer(one, one, one, 2, 5, 100, 0, None) == -2   # unsupported hidden size
    assert L.sfm_bilstm_layer_ex(one, one, one, 2, 5, 100, 1, None) == -2
