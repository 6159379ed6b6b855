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
    assert l.eeseg_version() == 106
    assert l.eeseg_conv_stats_tiles(16, 65, 65) == (16 * 65 * 65 + 63) // 64      # allocation bound: one row per 64 pixels
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


def test_option_table_round_trips(built_lib):
    """Every EESEG_OPT_* key of the header: the documented default comes back from eeseg_get_option, a legal value
    round-trips, an illegal one is refused with EESEG_ERR_ARG and leaves the setting alone.  Host-side state only."""
    from ee_semantic_segmentation_amd import _lib
    l = _lib.lib()
    txt = open(HEADER).read()
    keys = {m.group(1): int(m.group(2)) for m in re.finditer(r"(EESEG_OPT_\w+)\s*=\s*(\d+)", txt)}
    assert len(keys) >= 12 and len(set(keys.values())) == len(keys)
    # key: (default, another legal value, an illegal value)
    table = {"EESEG_OPT_CONV_PIPE": (3, 0, 9), "EESEG_OPT_CONV_TAP_INNER": (0, 1, 2), "EESEG_OPT_CONV_NARROW_MAX": (64, 128, 1),
             "EESEG_OPT_CONV_AUTO_NARROW": (0, 1, 2), "EESEG_OPT_CONV_TAIL_MIN": (224, 0, 999), "EESEG_OPT_CE_SPAN": (1, 0, 2),
             "EESEG_OPT_CONV_TAIL_MERGE": (1, 0, 2), "EESEG_OPT_CONV_CUS": (256, 240, 8), "EESEG_OPT_BN_REVERSE": (0, 3, 4),
             "EESEG_OPT_BN_ROWS": (2, 4, 3), "EESEG_OPT_COLREDUCE_BLOCKS": (512, 0, -1), "EESEG_OPT_CONV_SPLIT_MIN_K": (4, 8, 0),
             "EESEG_OPT_CONV_PW_MAX_K": (1280, 0, -1), "EESEG_OPT_CONV_PWS": (1, 2, 6), "EESEG_OPT_CONV_PW_ALL": (0, 1, 2), "EESEG_OPT_CONV_COUT_GROUP": (0, 2, 3), "EESEG_OPT_CONV_MFMA16": (1, 0, 2), "EESEG_OPT_BN_BWD_ROWS": (1, 2, 3), "EESEG_OPT_CONV_SWP": (1, 0, 2), "EESEG_OPT_BN_NT": (0, 3, 4),
             "EESEG_OPT_CONV_SMALL_M": (2, 0, 3), "EESEG_OPT_CONV_SMALL_M_MAX_K": (160, 64, -1), "EESEG_OPT_CONV_SMALL_M_DEEP": (1, 2, 3)}
    assert set(table) == set(keys), set(table) ^ set(keys)
    for name, (default, other, bad) in table.items():
        k = keys[name]
        assert l.eeseg_get_option(k) == default, name
        assert l.eeseg_set_option(k, other) == 0 and l.eeseg_get_option(k) == other, name
        assert l.eeseg_set_option(k, bad) != 0 and l.eeseg_get_option(k) == other, name
        assert l.eeseg_set_option(k, default) == 0
    assert l.eeseg_set_option(999, 0) != 0 and l.eeseg_get_option(999) < 0
