"""The C-ABI library loads and exports every symbol include/flexs_amd.h declares
(no compute calls here -- there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

import pytest

from flexs_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "flexs_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    names = _declared_symbols()
    assert len(names) >= 30
    lib = ctypes.CDLL(_native.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in flexs_amd.h but not exported: {missing}"


def test_python_binding_covers_the_header():
    assert sorted(_native.SIGNATURES) == _declared_symbols()


def test_version_and_status_names():
    lib = _native.lib()
    assert lib.fx_version() == 100
    assert _native.status_name(0) == "FX_OK"
    assert _native.status_name(_native.FX_EBADCHAR) == "FX_EBADCHAR"
    assert _native.status_name(_native.FX_ESHAPE) == "FX_ESHAPE"
    assert _native.status_name(-99) == "FX_UNKNOWN"
    assert lib.fx_device_count() >= 0


def test_only_hip_runtime_dependency():
    """The boundary is a plain C-ABI .so: no torch / python symbols in its NEEDED list."""
    out = os.popen(f"readelf -d {_native.LIB_PATH}").read()
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert any(n.startswith("libamdhip64") for n in needed)
    assert not any("torch" in n or "python" in n or "c10" in n for n in needed)


def test_no_cpu_fallback_without_gpu():
    if _native.lib().fx_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _native.Engine(0)
    from flexs_amd.baselines.models import CNN

    cnn = CNN(8, 32, 100, "TGCA")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cnn.get_fitness(["ATGCATGC"])
    assert cnn.cost == 1          # landscape.py:44: cost is added before _fitness_function runs


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under flexs_amd/ may import or load it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "flexs_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "libfx_oracle" in txt or "fx_oracle" in txt:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
