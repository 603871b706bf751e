"""CPU tests: libmmgen.so loads without a GPU and exports every symbol include/mmgen.h declares; wire struct sizes."""
import ctypes
import os
import re
import sys

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "mmgen.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmgen_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(mmgen_pkg):
    lib = ctypes.CDLL(mmgen_pkg.LIB_PATH)
    names = _declared()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"libmmgen.so lacks {n} declared in include/mmgen.h"


def test_wire_struct_sizes():
    """sizeof(CaveLayer)=12, FeaturePlacement=20 (feature@0,pos@4,canReplace@16), CaveFeaturePlacement=24."""
    class CaveLayer(ctypes.Structure):
        _fields_ = [("start", ctypes.c_int32), ("end", ctypes.c_int32), ("b", ctypes.c_uint8), ("t", ctypes.c_uint8), ("pad", ctypes.c_uint8 * 2)]

    class FP(ctypes.Structure):
        _fields_ = [("feature", ctypes.c_uint8), ("p0", ctypes.c_uint8 * 3), ("pos", ctypes.c_int32 * 3), ("r", ctypes.c_uint8), ("p1", ctypes.c_uint8 * 3)]

    class CFP(ctypes.Structure):
        _fields_ = [("feature", ctypes.c_uint8), ("p0", ctypes.c_uint8 * 3), ("pos", ctypes.c_int32 * 3), ("lh", ctypes.c_int32), ("r", ctypes.c_uint8),
                    ("p1", ctypes.c_uint8 * 3)]
    assert ctypes.sizeof(CaveLayer) == 12 and ctypes.sizeof(FP) == 20 and ctypes.sizeof(CFP) == 24
    assert FP.pos.offset == 4 and FP.r.offset == 16 and CFP.lh.offset == 16 and CFP.r.offset == 20
    text = open(os.path.join(ROOT, "include", "mmgen_types.h")).read()
    assert "MMB_SEA_LANTERN" in text and text.count("MMB_") > 140


def test_no_cpu_fallback(mmgen_pkg):
    """The product must fail loudly without a GPU instead of silently routing elsewhere."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        mmgen_pkg.MMGen(0)


def test_product_never_imports_oracle():
    """Nothing under mega-minecraft_amd/ or include/ includes, imports, links or loads anything from oracle/."""
    import re
    bad = re.compile(r'#include\s*[<"][^>"]*(oracle|mmo_)|\bimport\s+oracle|from\s+oracle|oracle_binding|libmmoracle|\bmmo_[a-z]|-lmmoracle|oracle/')
    for top in ("mega-minecraft_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp", ".hpp")) or f == "Makefile":
                    for n, line in enumerate(open(os.path.join(d, f), errors="ignore"), 1):
                        code = line.split("//")[0] if not f.endswith(".py") else line.split("#")[0]
                        if f.endswith(".py") and not re.search(r"^\s*(import|from)\b|CDLL|LoadLibrary|subprocess|open\(", line):
                            continue            # prose in docstrings may mention the oracle; code that could load it may not
                        assert not bad.search(code), f"{os.path.join(d, f)}:{n} references the oracle: {line.strip()}"


def test_the_library_is_loaded_behind_torch():
    """libmmgen.so needs libamdhip64.so.N and gets whichever copy the process holds already: loaded before torch it binds /opt/rocm's runtime,
    torch then brings its bundled one, and mmgen_init sees no device (build() followed by smoke() in one process, round 6).  load_library()
    therefore imports torch first: in a fresh interpreter that has not imported it, it is in sys.modules once the library is loaded."""
    import subprocess
    code = ("import importlib, sys; sys.path.insert(0, %r); assert 'torch' not in sys.modules; "
            "p = importlib.import_module('mega-minecraft_amd'); p.load_library(); assert 'torch' in sys.modules; print('ok')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
