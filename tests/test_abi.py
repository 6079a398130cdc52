"""CPU: the C-ABI library loads and exports every symbol include/tgp.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "trieste_amd", "libtgp.so")


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "tgp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tgp_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__

        __graft_entry__.build()
    return ctypes.CDLL(LIB)


def test_every_declared_symbol_is_exported(lib):
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/tgp.h but not exported by libtgp.so"


def test_python_binding_covers_header():
    from trieste_amd import _lib

    assert sorted(_lib.SIGNATURES) == _header_symbols()


def test_version_and_create_fails_loudly_without_gpu(lib):
    lib.tgp_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.tgp_version()
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    rc = lib.tgp_create(0, 2, 3, ctypes.byref(h))
    assert rc != 0 and not h.value
    lib.tgp_last_error.restype = ctypes.c_char_p
    assert b"no CPU fallback" in lib.tgp_last_error(None)


def test_engine_raises_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from trieste_amd.engine import GPEngine

    with pytest.raises(Exception):
        GPEngine(2, "matern52")


def test_fake_engine_mirrors_the_engine_surface():
    """The CPU host-logic tests replace GPEngine by tests/fakes.py::FakeEngine at the engine boundary: the stand-in
    must offer every public method of the real class with the same parameter names, or those tests would exercise
    a different interface than the product's (and the trajectory stand-ins likewise)."""
    import inspect

    from tests.fakes import FakeEngine, FakeRffTrajectory, FakeTrajectory
    from trieste_amd.engine import GPEngine, Trajectory

    def public(cls):
        return {n: f for n, f in inspect.getmembers(cls, predicate=inspect.isfunction) if not n.startswith("_")}

    def params(f):
        return [p for p in inspect.signature(f).parameters if p != "self"]

    real, fake = public(GPEngine), public(FakeEngine)
    missing = sorted(set(real) - set(fake))
    assert not missing, f"FakeEngine lacks {missing}"
    for name, f in real.items():
        assert params(f) == params(fake[name]), (name, params(f), params(fake[name]))
    assert params(GPEngine.__init__) == params(FakeEngine.__init__)
    # one engine class serves both trajectory kinds (v: decoupled only, theta: RFF weight posterior only)
    both = set(public(FakeTrajectory)) | set(public(FakeRffTrajectory))
    lacking = sorted(set(public(Trajectory)) - both)
    assert not lacking, f"the trajectory stand-ins lack {lacking}"
    for stand_in, own in ((FakeTrajectory, "v"), (FakeRffTrajectory, "theta")):
        assert not sorted(set(public(Trajectory)) - set(public(stand_in)) - {"v", "theta"}) and own in public(stand_in)
