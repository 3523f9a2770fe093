import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("OKGE_VALIDATE", "1")      # tests check id ranges before every kernel call (a bad id is an OOB access)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


@pytest.fixture(scope="session")
def okge_lib():
    """The product C-ABI library; GPU tests fail loudly (not skip) if it cannot be loaded."""
    from open_knowledge_graph_embeddings_amd import _native
    return _native.lib()
