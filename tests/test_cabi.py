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


def test_job_structures_have_the_headers_layout(tmp_path):
    """the ctypes mirrors of the batched entry points' job structures (texir_code_amd/_lib.py) against the header, field by field: a C program that
    includes include/texir_hip.h prints sizeof / offsetof, compiled with the host compiler"""
    import ctypes
    import subprocess
    from texir_code_amd import _lib
    structs = {"texir_tex_fetch_job": _lib.TexFetchJob, "texir_tex_gather_job": _lib.TexGatherJob, "texir_adam_tex_job": _lib.AdamTexJob}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % os.path.join(ROOT, "include", "texir_hip.h"), "int main(void) {"]
    for cname, cls in structs.items():
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append('  printf("TEXIR_MAX_BATCH %d\\n", TEXIR_MAX_BATCH);')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-o", str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)
    assert int(got["TEXIR_MAX_BATCH"]) == _lib.MAX_BATCH
