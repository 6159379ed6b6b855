"""CPU checks of the drop-in boundary: libeeseg.so builds/loads and exports exactly
the symbols include/eeseg.h declares, with the argument counts the ctypes binding
uses.  No compute call is made (there is no GPU here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "eeseg.h")


def _prototypes():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"typedef struct \{.*?\} \w+;", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:const\s+char\s*\*|int64_t|int)\s+(eeseg_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        protos[m.group(1)] = n
    return protos


@pytest.fixture(scope="module")
def built_lib():
    from ee_semantic_segmentation_amd import build
    return build.build(verbose=False)


def test_header_and_binding_agree():
    from ee_semantic_segmentation_amd import _lib
    protos = _prototypes()
    assert len(protos) >= 30
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for name, n in protos.items():
        assert len(_lib.SIGNATURES[name][1]) == n, name


def test_library_exports_every_declared_symbol(built_lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", built_lib]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    missing = set(_prototypes()) - exported
    assert not missing, missing


def test_library_loads_and_reports_version(built_lib):
    from ee_semantic_segmentation_amd import _lib
    l = _lib.lib()
    assert l.eeseg_version() >= 100
    assert l.eeseg_conv_stats_tiles(16, 65, 65) == (16 * 65 * 65 + 127) // 128
    assert l.eeseg_colreduce_workspace(1, 64) > 0


def test_bad_arguments_fail_loudly_without_gpu(built_lib):
    """Argument validation happens before any launch, so it is checkable on CPU."""
    import ctypes as C
    from ee_semantic_segmentation_amd import _lib
    l = _lib.lib()
    a = _lib.ConvArgs()
    rc = l.eeseg_conv_igemm(C.byref(a), None)
    assert rc == -1 and b"null" in l.eeseg_last_error()
    with pytest.raises(_lib.EesegError):
        _lib.check(rc, "conv")


def test_no_cpu_fallback():
    import torch
    from ee_semantic_segmentation_amd import kernels, _lib
    with pytest.raises(_lib.EesegError):
        kernels.maxpool3x3s2.__wrapped__ if hasattr(kernels.maxpool3x3s2, "__wrapped__") else None
        kernels.conv_fwd(torch.zeros(1, 4, 4, 64), torch.zeros(64, 1, 1, 64))
