import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# The GPU tests run the PRODUCTION configuration: OKGE_VALIDATE unset -- no host-side id range checks, no sync per call; the
# kernels' own id guard (checked_row -> okge_id_errors) is the only net, and `_id_guard_stays_silent` checks it after every
# GPU test.  `validate_config` opts a test into the integration-time checks.


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "both: host-side parity test (no GPU needed) that ALSO runs under `-m gpu`, so the "
                                       "GPU-box record covers the batch producer / file loader / checkpoint rows")


@pytest.hookimpl(tryfirst=True)
def pytest_collection_modifyitems(config, items):
    """`-m gpu` also selects the tests marked `both` (they run in the CPU suite as well)"""
    if (config.getoption("-m") or "").strip() == "gpu":
        for it in items:
            if it.get_closest_marker("both") is not None:
                it.add_marker(pytest.mark.gpu)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def fb15k237_dir(tmp_path):
    """the reference's FB15k-237 id files valid.txt / test.txt (data fixtures, tests/golden/fb15k237/*.gz) unpacked"""
    import gzip
    import shutil
    for f in ("valid.txt", "test.txt"):
        with gzip.open(os.path.join(GOLDEN, "fb15k237", f + ".gz"), "rb") as src, open(os.path.join(tmp_path, f), "wb") as dst:
            shutil.copyfileobj(src, dst)
    return str(tmp_path)


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


@pytest.fixture(scope="session")
def okge_lib():
    """The product C-ABI library; GPU tests fail loudly (not skip) if it cannot be loaded."""
    from open_knowledge_graph_embeddings_amd import _native
    return _native.lib()


@pytest.fixture(autouse=True)
def _id_guard_stays_silent(request):
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            from open_knowledge_graph_embeddings_amd import _native
            assert _native.id_errors() == 0, "a kernel met an out-of-range id (row 0 was substituted)"


@pytest.fixture
def validate_config(monkeypatch):
    """OKGE_VALIDATE=1: host-side id range checks with a sync per call (integration work)"""
    from open_knowledge_graph_embeddings_amd import hotpath
    monkeypatch.setattr(hotpath, "VALIDATE", True)


@pytest.fixture
def production_config(monkeypatch):
    """The configuration production runs: OKGE_VALIDATE unset -- no host-side id range checks, no sync per call; the
    kernels' own id guard (checked_row -> okge_id_errors) is the only net.  Used by the full-size / real-data parity tests;
    at teardown the device-side guard must not have seen a single bad id."""
    from open_knowledge_graph_embeddings_amd import _native, hotpath
    monkeypatch.setattr(hotpath, "VALIDATE", False)
    yield
    assert _native.id_errors() == 0
