import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

PROFILES = ["n17_q32", "n167_q128", "n509_q2048", "n821_q4096", "n701_q8192"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def pure_golden():
    return load_golden("pure_functions.json")


@pytest.fixture(scope="session", params=PROFILES)
def scheme_golden(request):
    return load_golden("scheme_%s.json" % request.param)


def has_experiments(eng):
    """True when the loaded library was built with -DNTRU_EXPERIMENTS (`make -C ntru-circom_amd/csrc experiments`, then
    NTRU_ENGINE_LIB=ntru-circom_amd/lib/libntru_engine_experiments.so): kernel paths 6-12 exist only there."""
    try:
        eng.set_kernel_path(6)
    except Exception:
        return False
    eng.set_kernel_path(0)
    return True


def set_path_or_skip(eng, path):
    """ntru_engine_set_kernel_path(path); the measured-slower variants (paths 6-12) are skipped on the default library."""
    if path >= 6 and not has_experiments(eng):
        pytest.skip("kernel path %d needs the -DNTRU_EXPERIMENTS library (make experiments + NTRU_ENGINE_LIB)" % path)
    eng.set_kernel_path(path)


# Kernel paths 6-12 are compiled only into lib/libntru_engine_experiments.so: they join the parametrisations when that is the library
# under test (NTRU_ENGINE_LIB=.../libntru_engine_experiments.so), and are not collected otherwise.
EXPERIMENT_PATHS = [6, 7, 8, 9, 10, 11] if "experiments" in os.path.basename(os.environ.get("NTRU_ENGINE_LIB", "")) else []
