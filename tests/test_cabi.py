"""CPU: the C-ABI library builds/loads and exports every symbol include/texir_hip.h declares (no compute calls)."""
import os
import re

from conftest import ROOT


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "texir_hip.h")).read()
    return sorted(set(re.findall(r"TEXIR_API\s+[\w\s\*]+?\b(texir_\w+)\s*\(", txt)))


def test_header_symbols_exported():
    from texir_code_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 11
    for s in syms:
        assert hasattr(L, s), "libtexir_hip.so does not export %s" % s
    assert L.texir_version() >= 100


def test_missing_library_is_loud(monkeypatch):
    from texir_code_amd import _lib
    import pytest
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libtexir_hip.so")
    with pytest.raises(_lib.TexirError):
        _lib.lib()
