"""CPU: the C-ABI library loads and exports every symbol include/misplat.h declares, the ctypes
mirror of misplat_params matches the C layout, and the product path fails loudly (no fallback,
no route through oracle/).  No compute calls: there is no GPU here."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "misplat.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(misplat_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from collab_splats_amd import _lib
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol(built_lib):
    from collab_splats_amd import _lib
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.misplat_version().decode().endswith("gfx950")
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (misplat_\w+)", out))
    assert exported == set(declared_symbols())          # nothing else leaks out of the C ABI


def test_library_contains_gfx950_code_object(built_lib):
    data = open(built_lib, "rb").read()
    assert b"gfx950" in data and b"blend_bwd_kernel" in data


def test_params_struct_layout_matches_c(built_lib):
    from collab_splats_amd._lib import Params
    fields = [f[0] for f in Params._fields_]
    src = "#include <stdio.h>\n#include <stddef.h>\n#include \"misplat.h\"\nint main(){printf(\"%zu\\n\", sizeof(misplat_params));\n"
    src += "".join(f'printf("%zu\\n", offsetof(misplat_params, {f}));\n' for f in fields) + "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        vals = [int(x) for x in subprocess.check_output([exe]).split()]
    assert vals[0] == C.sizeof(Params)
    assert vals[1:] == [getattr(Params, f).offset for f in fields]


@pytest.mark.parametrize("cname,pyname", [("misplat_raster_args", "RasterArgs"), ("misplat_raster_bwd_args", "RasterBwdArgs")])
def test_raster_args_struct_layout_matches_c(built_lib, cname, pyname):
    from collab_splats_amd import _lib
    RasterArgs = getattr(_lib, pyname)
    fields = [f[0] for f in RasterArgs._fields_]
    src = "#include <stdio.h>\n#include <stddef.h>\n#include \"misplat.h\"\nint main(){printf(\"%zu\\n\", sizeof(" + cname + "));\n"
    src += "".join(f'printf("%zu\\n", offsetof({cname}, {f}));\n' for f in fields) + "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        vals = [int(x) for x in subprocess.check_output([exe]).split()]
    assert vals[0] == C.sizeof(RasterArgs)
    assert vals[1:] == [getattr(RasterArgs, f).offset for f in fields]


def test_no_cpu_fallback():
    import collab_splats_amd as m
    z = torch.zeros
    with pytest.raises(m.MisplatError):
        m.rasterization(z(2, 3), z(2, 4), z(2, 3), z(2), z(2, 3), torch.eye(4)[None], torch.eye(3)[None], 8, 8)
    with pytest.raises(m.MisplatError):
        m.fully_fused_projection(z(2, 3), None, z(2, 4), z(2, 3), torch.eye(4)[None], torch.eye(3)[None], 8, 8)
    with pytest.raises(m.MisplatError):
        m.spherical_harmonics(0, z(2, 3), z(2, 1, 3))


def test_missing_library_fails_loudly(monkeypatch):
    from collab_splats_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmisplat.so")
    with pytest.raises(_lib.MisplatError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "collab_splats_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "craster" not in text and "torch_oracle" not in text, f
    code = "import sys; import collab_splats_amd; assert not any(m.split('.')[0]=='oracle' for m in sys.modules)"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_argument_validation_matches_reference_errors():
    import collab_splats_amd as m
    z = torch.zeros
    args = (z(2, 3), z(2, 4), z(2, 3), z(2), z(2, 3), torch.eye(4)[None], torch.eye(3)[None], 8, 8)
    with pytest.raises(ValueError, match="rasterize_mode"):      # rade_gs_model.py:150-151
        m.rasterization(*args, rasterize_mode="fancy")
    with pytest.raises(ValueError, match="render_mode"):
        m.rasterization(*args, render_mode="XYZ")
    with pytest.raises(NotImplementedError):
        m.rasterization(*args, packed=True)
    with pytest.raises(AssertionError):
        m.rasterization(z(2, 3), z(3, 4), z(2, 3), z(2), z(2, 3), torch.eye(4)[None], torch.eye(3)[None], 8, 8)


def test_gsplat_alias_resolves_reference_imports():
    code = ("import collab_splats_amd as m; m.install_gsplat_alias();"
            "from gsplat.rendering import rasterization;"
            "from gsplat.strategy import DefaultStrategy;"
            "from gsplat.cuda._wrapper import fully_fused_projection, spherical_harmonics;"
            "assert rasterization is m.rasterization and DefaultStrategy().absgrad is False")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_gsplat_alias_serves_nerfstudio_style_imports_and_names_what_is_missing():
    """nerfstudio's Splatfacto imports more of gsplat than the reference itself
    (``from gsplat.strategy import DefaultStrategy, MCMCStrategy``): those lines must resolve against the alias, and
    anything that is not built must say so instead of failing with a bare AttributeError."""
    import importlib
    import sys
    import collab_splats_amd
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "gsplat" or k.startswith("gsplat.")}
    try:
        collab_splats_amd.install_gsplat_alias()
        ns = {}
        exec("from gsplat.strategy import DefaultStrategy, MCMCStrategy\n"
             "from gsplat.rendering import rasterization\n"
             "from gsplat.cuda._wrapper import fully_fused_projection, spherical_harmonics\n"
             "from gsplat import rasterization as r2, DefaultStrategy as d2\n"
             "import gsplat", ns)
        assert ns["DefaultStrategy"] is collab_splats_amd.DefaultStrategy and ns["r2"] is collab_splats_amd.rasterization
        with pytest.raises(NotImplementedError):
            ns["MCMCStrategy"]()
        with pytest.raises(ImportError, match="not provided by collab_splats_amd"):
            exec("from gsplat import rasterization_2dgs", {})
        with pytest.raises(ImportError, match="not provided"):
            getattr(ns["gsplat"].cuda, "_backend")
    finally:
        for k in [k for k in sys.modules if k == "gsplat" or k.startswith("gsplat.")]:
            sys.modules.pop(k)
        sys.modules.update(saved)
