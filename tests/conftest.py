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
