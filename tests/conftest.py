import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
EMU_DIR = os.path.join(ROOT, "tests", "hipemu")
EMU_LIB = os.environ.get("PHIHIP_EMU_LIB", os.path.join(EMU_DIR, "libphihip_emu.so"))   # tools/asan_emu.sh points it at the ASan build


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """ `-n auto` by default, but only where pytest-xdist exists (`addopts = -n auto` made every invocation fail with "unrecognized arguments: -n"
    on a machine without the plugin). Runs before xdist's own hook of this name, which turns "auto" into pytest_xdist_auto_num_workers below. """
    is_worker = hasattr(config, "workerinput") or "PYTEST_XDIST_WORKER" in os.environ        # (a worker must not start workers of its own)
    if (config.pluginmanager.hasplugin("xdist") and not is_worker and getattr(config.option, "numprocesses", None) is None
            and not getattr(config.option, "usepdb", False)):
        config.option.numprocesses = "auto"
    return None


@pytest.hookimpl(optionalhook=True)                       # (the hook exists only where pytest-xdist is installed)
def pytest_xdist_auto_num_workers(config):
    """ `-n auto` (pytest.ini): CPU suite on up to 6 cores; with a GPU present (the `-m gpu` tiers of the driver) everything stays in ONE process """
    try:
        import torch
        if torch.cuda.is_available():
            return 0
    except Exception:
        pass
    return max(1, min(6, (os.cpu_count() or 2) - 1))


def _emu_is_stale():
    if not os.path.exists(EMU_LIB):
        return True
    t = os.path.getmtime(EMU_LIB)
    srcs = [os.path.join(ROOT, "phiflow_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "phiflow_amd", "csrc"))
            if f.endswith((".hip", ".hpp"))]
    srcs += [os.path.join(ROOT, "include", "phihip.h"), os.path.join(EMU_DIR, "hipemu.cpp"),
             os.path.join(EMU_DIR, "include", "hip", "hip_runtime.h")]
    return any(os.path.getmtime(s) > t for s in srcs)


@pytest.fixture(scope="session")
def emu_library():
    """ TEST INFRASTRUCTURE: the kernel sources compiled with g++ against the fiber emulation (tests/hipemu). """
    from phiflow_amd import _capi
    # contexts on the emulation keep the analytic launch plan: timing candidates there is meaningless (and the bit-for-bit tests of
    # tests/test_parallel_gloo.py need the same launch geometry in every process)
    os.environ["PHIHIP_AUTOTUNE"] = "0"
    try:
        import fcntl                                                   # POSIX only -- like build_emu.sh itself (bash, g++); without it: no lock, run serially
    except ImportError:
        fcntl = None
    with open(os.path.join(EMU_DIR, ".build.lock"), "w") as lock:      # xdist workers: ONE of them rebuilds a stale library, the others wait
        if fcntl is not None:
            fcntl.flock(lock, fcntl.LOCK_EX)
        if _emu_is_stale():
            subprocess.run(["bash", os.path.join(EMU_DIR, "build_emu.sh")], check=True, stdout=subprocess.DEVNULL)
    return _capi.Library(EMU_LIB)


@pytest.fixture(scope="session")
def emu_ctx(emu_library):
    from phiflow_amd import _capi
    return _capi.Context(emu_library, 0)


@pytest.fixture(scope="session")
def emu_backend(emu_library):
    from phiflow_amd.backend import HipBackend
    return HipBackend(library=emu_library, device="cpu")


@pytest.fixture(scope="session")
def gpu_backend():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from phiflow_amd.backend import HipBackend
    return HipBackend()   # loads phiflow_amd/lib/libphihip.so; raises if it is missing
