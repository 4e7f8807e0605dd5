"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads, and exports exactly the
symbols include/hpe.h declares (no compute calls -- there is no GPU in the build container)."""
import os
import re

import pytest

import hpe_amd
from hpe_amd import _lib, build as hbuild, resnet_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    hbuild.build()
    return _lib.load()


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "hpe.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hpe_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree(lib):
    hdr = _header_functions()
    assert hdr == _lib.declared_symbols(), (set(hdr) ^ set(_lib.declared_symbols()))
    for name in hdr:
        assert hasattr(lib, name), "library does not export %s" % name


def test_config_struct_matches_header(lib):
    """HpeConfig: the ctypes mirror has the header's fields in the header's order (all 4-byte scalars), and
    hpe_config_init fills the documented defaults with every plan option at -1 (= environment variable, else built-in)."""
    import ctypes as C

    txt = open(os.path.join(ROOT, "include", "hpe.h")).read()
    body = txt[txt.index("typedef struct HpeConfig {"):txt.index("} HpeConfig;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:int|float)\s+([a-z_0-9]+)\s*;", body)
    assert fields == [f[0] for f in _lib.HpeConfig._fields_]
    assert C.sizeof(_lib.HpeConfig) == 4 * len(fields)
    assert fields[0] == "struct_size" and tuple(fields[6:]) == _lib.PLAN_OPTIONS
    cfg = _lib.HpeConfig()
    lib.hpe_config_init(C.byref(cfg))
    assert cfg.struct_size == C.sizeof(_lib.HpeConfig)
    assert (cfg.device, cfg.max_batch, cfg.num_stage, cfg.encoder_dtype) == (0, 8, 3, 0) and abs(cfg.bn_eps - 1e-3) < 1e-9
    assert all(getattr(cfg, k) == -1 for k in _lib.PLAN_OPTIONS)


def test_create_refuses_foreign_config_struct(lib):
    """hpe_create checks HpeConfig.struct_size before it reads anything else: a zero-initialised struct (hpe_config_init not called: every
    plan option would read 0 = the slowest plan) and a struct of another header revision (shorter: the library would read plan options
    from past its end) are refused with HPE_ERR_INVALID and a message naming both sizes.  Runs without a GPU: the check comes first."""
    import ctypes as C

    h = C.c_void_p()
    zeroed = _lib.HpeConfig()  # ctypes zero-fills
    assert lib.hpe_create(C.byref(zeroed), C.byref(h)) == 1 and not h.value
    assert b"struct_size is 0" in lib.hpe_last_error()

    class OldConfig(C.Structure):  # the round-3 layout: no struct_size, 17 scalars
        _fields_ = [(n, t) for n, t in _lib.HpeConfig._fields_[1:-1]]

    old = OldConfig(device=0, max_batch=8, num_stage=3, bn_eps=1e-3, encoder_dtype=0, n_streams=-1)
    assert lib.hpe_create(C.cast(C.byref(old), C.POINTER(_lib.HpeConfig)), C.byref(h)) == 1 and not h.value
    short = _lib.HpeConfig()
    lib.hpe_config_init(C.byref(short))
    short.struct_size -= 4  # one field fewer: a stub built against the previous header
    assert lib.hpe_create(C.byref(short), C.byref(h)) == 1 and not h.value
    msg = lib.hpe_last_error().decode()
    assert str(C.sizeof(_lib.HpeConfig) - 4) in msg and str(C.sizeof(_lib.HpeConfig)) in msg, msg
    good = _lib.HpeConfig()
    lib.hpe_config_init(C.byref(good))
    rc = lib.hpe_create(C.byref(good), C.byref(h))
    assert rc in (0, 4), rc  # 4 = HPE_ERR_NO_DEVICE in the build container: the struct itself was accepted
    if rc == 0:
        lib.hpe_destroy(h)


def test_layer_table_matches_host_spec(lib):
    import ctypes as C

    geo = (C.c_int * 7)()
    for i, s in enumerate(resnet_spec.CONV_SPECS):
        assert lib.hpe_conv_layer_name(i).decode() == s.name
        assert lib.hpe_bn_layer_name(i).decode() == s.bn_name
        assert lib.hpe_conv_layer_geometry(i, geo) == 0
        assert tuple(geo) == (s.kh, s.kw, s.cin, s.cout, s.stride, s.hin, s.hout)
    assert lib.hpe_conv_layer_name(53) is None


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hpe_amd.HpeError):
        hpe_amd.HpeEngine()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "human-pose-estimation_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
