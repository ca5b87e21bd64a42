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
