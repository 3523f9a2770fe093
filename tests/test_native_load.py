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


def test_streamk_run_and_slab_arithmetic():
    """The stream-K launch of fused_tile64k_kernel (okge_train64k.hip) and dc_reduce_streamk_kernel (okge_misc.hip) agree on
    who writes which partial-gradient slab by ARITHMETIC alone (no plan travels): this mirrors both sides' integer formulas and
    checks, for many (tiles, chunks per tile, workgroups), that the runs tile the units exactly once, that a workgroup writes
    at most two slabs (2p for its first segment, 2p + 1 for its last), and that the slabs the reduce kernel lists for a tile
    are exactly the ones written for it, in run order; tiles covered by one whole segment list nothing."""
    import random

    def kernel_segments(T, J, P):
        """per workgroup p: [(tile, j0, j1, slab or None)] as the tile kernel's segment loop derives them"""
        U = T * J
        out = []
        for p in range(P):
            u, u_end = U * p // P, U * (p + 1) // P
            u_first, segs = u, []
            while u < u_end:
                tile = u // J
                j0, j1 = u - tile * J, min(J, u_end - tile * J)
                whole = j0 == 0 and j1 == J
                segs.append((tile, j0, j1, None if whole else 2 * p + (0 if u == u_first else 1)))
                u = tile * J + j1
            out.append(segs)
        return out

    def reduce_slots(T, J, P, t):
        """the slab list dc_reduce_streamk_kernel builds for tile t ([] = the tile kernel stored the tile itself)"""
        U, lo, hi = T * J, t * J, t * J + J
        p = lo * P // U
        while p > 0 and U * p // P > lo:
            p -= 1
        while U * (p + 1) // P <= lo:
            p += 1
        slots = []
        while p < P and U * p // P < hi:
            ub, ue = U * p // P, U * (p + 1) // P
            j0, j1 = max(ub, lo) - lo, min(ue, hi) - lo
            if j1 > j0:
                if j0 == 0 and j1 == J:
                    return []
                slots.append(2 * p + (0 if ub >= lo else 1))
            p += 1
        return slots

    rng = random.Random(5)
    cases = [(157, 16, 256), (1, 1, 1), (1, 16, 256), (3, 5, 7), (300, 16, 256), (33, 8, 256), (2, 128, 256), (1000, 2, 256)]
    cases += [(rng.randint(1, 400), rng.randint(1, 40), rng.randint(1, 300)) for _ in range(200)]
    for T, J, P in cases:
        P = min(P, T * J)                                    # the launcher never starts more workgroups than units
        segs = kernel_segments(T, J, P)
        cover = [[0] * J for _ in range(T)]
        written = {t: [] for t in range(T)}
        for p, lst in enumerate(segs):
            assert sum(1 for s in lst if s[3] is not None) <= 2, (T, J, P, p)
            assert len({s[3] for s in lst if s[3] is not None}) == sum(1 for s in lst if s[3] is not None)
            for tile, j0, j1, slab in lst:
                for j in range(j0, j1):
                    cover[tile][j] += 1
                if slab is not None:
                    written[tile].append(slab)
        assert all(c == 1 for row in cover for c in row), (T, J, P)
        for t in range(T):
            assert reduce_slots(T, J, P, t) == written[t], (T, J, P, t)
            assert len(written[t]) != 1, (T, J, P, t)       # a partial tile has at least two partial segments


def test_deferred_decay_bookkeeping_on_the_host():
    """TokenPooledTrainStep's host side of the deferred decay (no GPU: a recording stand-in for the engine): an optimizer step
    leaves work pending under the hyper-parameters of ITS time, flush() settles it once with exactly those, a learning-rate
    change settles before the next update, a graph replay marks work pending again, state_tensors() flushes and carries the step
    counters, and small tables default to the eager sweep"""
    import torch
    from open_knowledge_graph_embeddings_amd.token_pooled import TokenPooledTrainStep, TokenSlot

    class Engine:
        def __init__(self):
            self.calls = []

        def adagrad_lazy(self, tensors, counters, window, flush, lr, wd, eps):
            self.calls.append(("flush" if flush else "step", len(tensors), window, lr, wd, eps))

        def adagrad_multi(self, tensors, lr, wd, eps):
            self.calls.append(("eager", len(tensors), lr))

    def make(**kw):
        tok = torch.zeros((5, 3), dtype=torch.int32)
        e = TokenSlot(torch.zeros(8, 4), tok, "sum", True)
        r = TokenSlot(torch.zeros(6, 4), tok, "sum", True)
        eng = Engine()
        return TokenPooledTrainStep(e, r, "complex", lr=0.1, engine=eng, **kw), eng
    st, eng = make()
    assert st.decay_window == 1                                   # 224 bytes of tables: the eager sweep
    st.optimizer_step()
    assert eng.calls == [("eager", 4, 0.1)] and st._pending is None
    st.flush()
    assert len(eng.calls) == 1                                    # nothing owed, nothing launched
    st, eng = make(decay_window=8)
    st.optimizer_step()
    assert eng.calls == [("step", 4, 8, 0.1, 1e-10, 1e-8)] and st._pending == (0.1, 1e-10, 1e-8)     # 2 tables + 2 batch-norm vectors
    st.lr = 0.05                                                   # the owed steps keep the rate of their time
    st.optimizer_step()
    assert eng.calls[1] == ("flush", 2, 8, 0.1, 1e-10, 1e-8) and eng.calls[2] == ("step", 4, 8, 0.05, 1e-10, 1e-8)
    st.flush()
    st.flush()
    assert [c[0] for c in eng.calls] == ["step", "flush", "step", "flush"] and st._pending is None
    st.mark_pending()                                              # GraphedTrainStep.replay: the graph ran the step's launches
    tensors = st.state_tensors()
    assert eng.calls[-1][0] == "flush" and len(eng.calls) == 5
    assert any(t is st._counters for t in tensors) and any(t is st.entity.row_steps for t in tensors)
