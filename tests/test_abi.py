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


def test_the_oracle_is_test_infrastructure_only():
    """The rule of the build: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch
    oracle/ (or tests/); the product package and the measurement tools never do, and nothing in the package falls
    back to a CPU evaluation when the HIP library is missing."""
    import ast
    import pathlib

    root = pathlib.Path(__file__).resolve().parents[1]

    def offending_imports(path, allowed_functions=()):
        tree = ast.parse(path.read_text())
        allowed_nodes = set()
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef) and node.name in allowed_functions:
                allowed_nodes.update(id(n) for n in ast.walk(node))
        bad = []
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom) and node.level == 0:
                names = [node.module or ""]
            for name in names:
                if name.split(".")[0] in ("oracle", "tests") and id(node) not in allowed_nodes:
                    bad.append((path.name, node.lineno, name))
        return bad

    bad = []
    for path in list((root / "trieste_amd").rglob("*.py")) + list((root / "tools").glob("*.py")):
        bad += offending_imports(path)
    bad += offending_imports(root / "bench.py", allowed_functions=("cpu_baseline",))
    bad += offending_imports(root / "__graft_entry__.py", allowed_functions=("smoke",))
    assert not bad, bad
    # a missing library is an error, never a fallback
    from trieste_amd import _lib

    src = (root / "trieste_amd" / "_lib.py").read_text()
    assert "no CPU fallback" in src or "no CPU path" in src
    assert callable(_lib.load)


def test_missing_library_is_an_import_error_not_a_fallback(monkeypatch, tmp_path):
    """No libtgp.so -> ImportError with build instructions from the loader, and therefore from every engine /
    model construction: nothing computes on the CPU instead."""
    import numpy as np

    import trieste_amd.models as M
    from trieste_amd import _lib
    from trieste_amd.data import Dataset
    from trieste_amd.engine import GPEngine
    from trieste_amd.space import Box

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libtgp.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
    with pytest.raises(ImportError):
        GPEngine(2, "matern52")
    x = np.random.default_rng(0).uniform(size=(5, 2))
    data = Dataset(x, x[:, :1])
    with pytest.raises(ImportError):
        M.GaussianProcessRegression(M.build_gpr(data, Box([0, 0], [1, 1])))
