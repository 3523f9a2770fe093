"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/okge.h declares.
No compute calls (there is no GPU here)."""
import os
import re

import pytest

from conftest import ROOT


def test_header_symbols_exported():
    from open_knowledge_graph_embeddings_amd import _native
    if _native.needs_build():
        _native.build_native()
    L = _native.lib()
    header = open(os.path.join(ROOT, "include", "okge.h")).read()
    declared = set(re.findall(r"\b(okge_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    for sym in declared:
        assert getattr(L, sym) is not None
    assert L.okge_abi_version() == 1


def test_workspace_query_and_argument_errors():
    from open_knowledge_graph_embeddings_amd import _native
    L = _native.lib()
    assert L.okge_train_workspace_bytes(512, 14541, 200) > 512 * 14541 * 4
    assert L.okge_train_workspace_bytes(0, 10, 16) == 0
    # NULL descriptors are rejected before anything touches a device
    rc = L.okge_score_prefixes(None, None, None, None, 0, None, 0, None)
    assert rc == -1 and b"null" in L.okge_last_error()
    rc = L.okge_adagrad_step(None, None, None, 0, 0.1, 0.0, 1e-8, 0, None)
    assert rc == -1


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the product package may reference it."""
    pkg = os.path.join(ROOT, "open_knowledge_graph_embeddings_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
                assert "libkge_oracle" not in text and "torch_twin" not in text, f


def test_hotpath_refuses_cpu():
    from open_knowledge_graph_embeddings_amd import OkgeError
    from open_knowledge_graph_embeddings_amd.hotpath import HotPath
    with pytest.raises(OkgeError):
        HotPath("cpu")


def test_integration_doc_maps_every_entry_point():
    """INTEGRATION.md is the binding guide: every function include/okge.h declares appears in its table"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "okge.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    declared = sorted(set(re.findall(r"\b(okge_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 30
    assert [f for f in declared if f not in doc] == []
