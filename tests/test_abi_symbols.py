"""The C-ABI library must load (no GPU needed) and export every function include/trase_rast.h
declares; the ctypes structure layouts must match the header's field order."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "trase_rast.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(trase_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from trase_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 12
    bound = {name for name, _, _ in _lib.SYMBOLS}
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in trase_rast.h but not exported"
        assert name in bound, f"{name} has no ctypes prototype in trase_amd/_lib.py"
    assert lib.trase_version().startswith(b"trase_amd")


def test_struct_field_order_matches_header():
    from trase_amd import _lib
    src = open(os.path.join(ROOT, "include", "trase_rast.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for cname, cls in (("TraseRastSettings", _lib.RastSettings), ("TraseRastInputs", _lib.RastInputs),
                       ("TraseRastOutputs", _lib.RastOutputs), ("TraseRastWorkspace", _lib.RastWorkspace),
                       ("TraseRastSizes", _lib.RastSizes), ("TraseRastGrads", _lib.RastGrads),
                       ("TraseRastRawInputs", _lib.RastRawInputs), ("TraseRastRawGrads", _lib.RastRawGrads),
                       ("TraseMlpWeights", _lib.MlpWeights)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                part = re.sub(r"\[[^\]]*\]", "", part).strip()          # drop array declarators
                names.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", part)[0])
        assert names == [f[0] for f in cls._fields_], cname


def test_sizes_call_works_without_gpu():
    import ctypes as C
    from trase_amd import _lib
    lib = _lib.load()
    sz = _lib.RastSizes()
    assert lib.trase_rast_sizes(1000, 640, 360, 32, 50000, C.byref(sz)) == 0
    assert sz.geom_bytes > 1000 * 48 and sz.bin_bytes >= 50000 * 8 and sz.bwd_tmp_bytes >= 50000 * 44 * 4
    assert lib.trase_rast_sizes(-1, 640, 360, 32, 1, C.byref(sz)) != 0
    assert b"bad arguments" in lib.trase_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    from trase_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libtrase_rast.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_cpu_tensors_are_rejected():
    import torch
    from tests.util import settings_for, small_case
    from diff_gaussian_rasterization import GaussianRasterizer
    act, cam = small_case(n=8, w=32, h=32)
    rast = GaussianRasterizer(raster_settings=settings_for(cam))
    with pytest.raises(RuntimeError, match="GPU only"):
        rast(means3D=act["means3D"], means2D=torch.zeros(8, 3), shs=act["shs"], sh_objs=act["sh_objs"],
             opacities=act["opacities"], scales=act["scales"], rotations=act["rotations"])
