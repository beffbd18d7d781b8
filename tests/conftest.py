import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running")
    # the built library is not in git: a fresh checkout builds it once (hipcc cross-compiles gfx950 without a GPU)
    lib = os.path.join(ROOT, "climt_amd", "_lib", "librrtmg_hip.so")
    if not os.path.exists(lib):
        import importlib.util   # by path: `import climt_amd` itself needs the library
        spec = importlib.util.spec_from_file_location("_rrtmg_build", os.path.join(ROOT, "climt_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)


@pytest.fixture(scope="session")
def gpu_ctx():
    """A librrtmg_hip context on device 0 with SW (and LW) tables initialised. Fails loudly, never falls back."""
    from climt_amd._lib import Context
    from oracle.ref_driver import CONSTANTS, CPDAIR
    ctx = Context(0)
    ctx.set_constants(**CONSTANTS)
    ctx.sw_init(CPDAIR)
    try:
        ctx.lw_init(CPDAIR)
    except Exception:
        pass
    yield ctx
    ctx.close()
