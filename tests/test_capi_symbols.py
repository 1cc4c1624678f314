"""
`-m "not gpu"`: the real gfx950 library loads in a GPU-less container and exports every symbol include/phihip.h declares; no
compute call is made (there is no CPU fallback: creating a context must fail loudly).
"""
import os
import re

import pytest

from phiflow_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "phihip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(phihip_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_capi.EXPORTED_SYMBOLS)


@pytest.mark.skipif(not os.path.exists(_capi.DEFAULT_LIBRARY_PATH), reason="libphihip.so not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    import ctypes
    dll = ctypes.CDLL(_capi.DEFAULT_LIBRARY_PATH)
    for name in _declared_symbols():
        assert hasattr(dll, name), f"{name} declared in include/phihip.h but not exported"
    lib = _capi.Library(_capi.DEFAULT_LIBRARY_PATH)
    assert lib.version() == 102


@pytest.mark.skipif(not os.path.exists(_capi.DEFAULT_LIBRARY_PATH), reason="libphihip.so not built")
def test_library_was_built_from_the_sources_in_this_tree():
    """ the .so is not in git and travels to the GPU box as a file: its embedded source hash must match the sources next to it """
    import subprocess
    lib = _capi.Library(_capi.DEFAULT_LIBRARY_PATH)
    if not lib.built_from_tree() and os.path.exists("/opt/rocm/bin/hipcc"):      # sources edited since the last build: rebuild, like build()
        # (flock: a second make on the same build directory -- another test session, a developer's shell -- must not interleave object files)
        csrc = os.path.join(ROOT, "phiflow_amd", "csrc")
        subprocess.run(["flock", os.path.join(csrc, ".build.lock"), "make", "-C", csrc, f"-j{os.cpu_count() or 4}"], check=True, stdout=subprocess.DEVNULL)
        lib = _capi.Library(_capi.DEFAULT_LIBRARY_PATH)
    bid = lib.build_id()
    assert re.fullmatch(r"[0-9a-f]{12}(\+dirty)? src:[0-9a-f]{16}|nogit src:[0-9a-f]{16}", bid), bid
    assert lib.built_from_tree(), f"stale libphihip.so: built from {bid}, tree has src:{_capi.source_hash()} -- run __graft_entry__.build()"


@pytest.mark.skipif(not os.path.exists(_capi.DEFAULT_LIBRARY_PATH), reason="libphihip.so not built")
def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _capi.Library(_capi.DEFAULT_LIBRARY_PATH)
    with pytest.raises(_capi.PhiHipError) as e:
        _capi.Context(lib, 0)
    assert e.value.status == -4 and "no CPU fallback" in str(e.value)
    from phiflow_amd.backend import HipBackend
    with pytest.raises(RuntimeError):
        HipBackend()


def test_struct_layouts_match_the_header():
    import ctypes
    assert ctypes.sizeof(_capi.Grid) == 4 * 3 + 4 * 3 + 8 * 3 + 8 * 3 + 4 * 6 + 8 * 18
    assert ctypes.sizeof(_capi.Solve) == 32
    assert ctypes.sizeof(_capi.SolveInfo) == 32
    assert ctypes.sizeof(_capi.ObstacleStruct) == 4 * 4 + 8 * (3 + 3 + 3 + 3 + 9)      # kind, group, embed_mask, reserved + 21 doubles
