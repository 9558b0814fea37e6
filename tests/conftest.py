"""pytest config: registers the `gpu` marker and makes the repo root importable.

`-m "not gpu"` covers the oracle against the reference's known-answer tests, the
host logic and the C-ABI export check; `-m gpu` are the parity tests proper and
call the HIP kernels through the C-ABI (they fail loudly without the .so).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


def _loaded_lib():
    """The product library if a test already loaded it (never loads it: CPU-only runs stay GPU-free)."""
    mod = sys.modules.get("ferrum_infer_rs_amd")
    be = getattr(mod, "backend", None) if mod else None
    return getattr(be, "_lib", None) if be else None


@pytest.fixture(autouse=True)
def _knobs_follow_the_environment():
    """The library reads its FERRUM_HIP_* development knobs once (no getenv on a launch path).  Tests steer kernel forms
    with `knobs.set(...)`; whatever a test changed is re-read from the restored environment when it ends (this fixture is
    set up before `monkeypatch`, so it is torn down after monkeypatch has undone its setenv calls)."""
    yield
    lib = _loaded_lib()
    if lib is not None:
        lib.ferrum_hip_debug_reload_knobs()


class _Knobs:
    def __init__(self, monkeypatch):
        self.mp = monkeypatch

    def set(self, **kw):
        """knobs.set(ATTN_RS_MIN_WGS=1, NO_GRAPH=None): set / unset FERRUM_HIP_<NAME> and make the library re-read them."""
        for k, v in kw.items():
            name = "FERRUM_HIP_" + k
            if v is None:
                self.mp.delenv(name, raising=False)
            else:
                self.mp.setenv(name, str(v))
        lib = _loaded_lib()
        assert lib is not None, "load the library (pkg fixture) before steering its knobs"
        lib.ferrum_hip_debug_reload_knobs()


@pytest.fixture
def knobs(monkeypatch):
    return _Knobs(monkeypatch)


class _Forms:
    """Kernel-form counters of the launchers (ferrum_hip_debug_form_hits): `forms.reset()` … run … `forms.hits()` →
    {name: count}; `forms.require("attn_flash")` asserts the named form(s) ran since the reset."""

    def __init__(self, lib):
        import ctypes as C
        self.lib, self.C = lib, C
        lib.ferrum_hip_debug_form_name.restype = C.c_char_p
        self.names = [lib.ferrum_hip_debug_form_name(i).decode() for i in range(lib.ferrum_hip_debug_form_count())]

    def reset(self):
        self.lib.ferrum_hip_debug_form_reset()

    def hits(self):
        arr = (self.C.c_uint64 * len(self.names))()
        assert self.lib.ferrum_hip_debug_form_hits(arr, len(self.names)) == 0
        return {n: int(arr[i]) for i, n in enumerate(self.names) if arr[i]}

    def require(self, *names, absent=()):
        h = self.hits()
        for n in names:
            assert n in self.names, f"unknown kernel form {n}"
            assert h.get(n, 0) > 0, f"kernel form {n} did not run; forms that ran: {h}"
        for n in absent:
            assert h.get(n, 0) == 0, f"kernel form {n} ran but should not have: {h}"
        return h


@pytest.fixture
def forms():
    lib = _loaded_lib()
    assert lib is not None, "load the library (pkg fixture) first"
    f = _Forms(lib)
    f.reset()
    return f
