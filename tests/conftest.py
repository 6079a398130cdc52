"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


try:  # property tests draw the same examples on every run: the suite is a gate, not a fuzzer
    from hypothesis import settings as _hyp_settings

    _hyp_settings.register_profile("deterministic", derandomize=True, deadline=None)
    _hyp_settings.load_profile("deterministic")
except ImportError:  # hypothesis is optional for everything but tests/test_host_properties.py
    pass


def pytest_addoption(parser):
    # like the reference (tests/conftest.py: --runslow): seed-tuned end-to-end runs are opt-in
    parser.addoption("--runslow", action="store", default="no", choices=("yes", "no", "only"),
                     help="whether to run tests marked slow")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: long, seed-tuned end-to-end runs (enable with --runslow yes)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    runslow = config.getoption("--runslow")
    for item in items:
        if runslow == "no" and "slow" in item.keywords:
            item.add_marker(pytest.mark.skip(reason="need --runslow yes to run"))
        if runslow == "only" and "slow" not in item.keywords:
            item.add_marker(pytest.mark.skip(reason="--runslow only"))
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _seeded_host_draws():
    """Every test starts from the same seed for the package's un-seeded host-side draws -- what the
    reference's tests do with their ``random_seed`` decorator (tests/util/misc.py)."""
    from trieste_amd import set_seed

    set_seed(20240916)
    yield
    set_seed(None)


def pytest_sessionfinish(session, exitstatus):
    """Write the parity margins of this session (tests/util.py MARGINS): per comparison the observed worst error as
    a fraction of its tolerance, worst first per test.  Goes to $TGP_MARGINS_FILE, else gpurun_out/parity_margins.txt
    when GPU tests ran."""
    try:
        from tests.util import MARGINS
    except Exception:
        return
    if not MARGINS:
        return
    path = os.environ.get("TGP_MARGINS_FILE")
    if not path:
        if not _has_gpu():
            return
        path = os.path.join(ROOT, "gpurun_out", "parity_margins.txt")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        agg = {}
        for test, what, worst, rtol, atol, n in MARGINS:
            key = (test, what, rtol, atol)
            w, cnt, els = agg.get(key, (0.0, 0, 0))
            agg[key] = (max(w, worst) if worst == worst else w, cnt + 1, els + n)
        rows = sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1][0]))
        summary = {}
        for (test, what, rtol, atol), (w, cnt, els) in agg.items():
            key = (test.split("[")[0], what)
            sw, sc, se = summary.get(key, (0.0, 0, 0))
            summary[key] = (max(sw, w), sc + cnt, se + els)
        with open(path, "w") as f:
            f.write("# observed worst |error| / tolerance per parity comparison (tolerance = rtol |ref| + atol); "
                    "1.0 = the tolerance is fully used\n")
            f.write(f"# {len(MARGINS)} comparisons\n# ---- summary: worst over the parametrisations of a test\n")
            f.write(f"{'worst/tol':>10} {'calls':>6} {'elements':>10}  test :: what\n")
            for (test, what), (w, cnt, els) in sorted(summary.items(), key=lambda kv: (kv[0][0], -kv[1][0])):
                f.write(f"{w:10.3g} {cnt:6d} {els:10d}  {test} :: {what}\n")
            f.write(f"# ---- every comparison ({len(agg)} rows)\n")
            f.write(f"{'worst/tol':>10} {'rtol':>8} {'atol':>10} {'calls':>6} {'elements':>10}  test :: what\n")
            for (test, what, rtol, atol), (w, cnt, els) in rows:
                f.write(f"{w:10.3g} {rtol:8.1e} {atol:10.2e} {cnt:6d} {els:10d}  {test} :: {what}\n")
    except OSError:
        pass
