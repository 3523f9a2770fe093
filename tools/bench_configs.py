#!/usr/bin/env python3
"""One-GPU timings of the other BASELINE.json configurations (the bench.py line is configs[1] = S-FB):

  S-DM        configs[2]  DistMult d=512, B=512, batch-shared sampled list N=10 000 (unique ids)
  S-OLP-shard configs[3]  one GPU's share of the OLPBENCH-shaped run: 312 500 local candidates, B=4096, d=256
  S-OLP-tok   configs[4]  token-pooled ComplEx d=256, B=4096, batch-shared N=8192, 10 tokens per entity, batch-norm

Prints one JSON object per workload (ms/step, per-kernel averages from the library's HIP-event timers)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from open_knowledge_graph_embeddings_amd import hotpath as H  # noqa: E402
from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep  # noqa: E402


def positives(rng, B, N, per_row=2):
    rows = np.repeat(np.arange(B, dtype=np.int64), per_row)
    cols = rng.integers(0, N, rows.shape[0])
    key = np.unique(cols * B + rows)
    return (key % B).astype(np.int32), (key // B).astype(np.int32)


def run(name, step, batches, steps=30, warmup=5, ksteps=10, quiet=False):
    for i in range(warmup):
        step.step(batches[i % len(batches)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step.step(batches[i % len(batches)])
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    eng = step.engine
    eng.timing(True)
    for i in range(ksteps):
        step.step(batches[i % len(batches)])
    torch.cuda.synchronize()
    per = {k: round(v[0] / v[1] * 1e3, 2) for k, v in eng.timing_collect().items()}
    eng.timing(False)
    b = batches[0]
    out = {"workload": name, "ms_per_step": round(ms, 4), "prefixes_per_s": round(b.B / ms * 1e3), "kernels_us": per}
    if not quiet:
        print(json.dumps(out), flush=True)
    return out


def positives_batch(rng, t, n_ent, n_rel, B, N, per_row, cand_ids=None, cand_unique=False):
    pr, pc = positives(rng, B, N, per_row=per_row)
    kw = dict(cand_ids=cand_ids, cand_unique=cand_unique) if cand_ids is not None else dict(cand_first=2, n_cand=N)
    return H.PrefixBatch(po_rel=t(rng.integers(2, n_rel, B // 2).astype(np.int32)), po_obj=t(rng.integers(2, n_ent, B // 2).astype(np.int32)),
                         sp_subj=t(rng.integers(2, n_ent, B // 2).astype(np.int32)), sp_rel=t(rng.integers(2, n_rel, B // 2).astype(np.int32)),
                         pos_row=t(pr), pos_col=t(pc), **kw)


# ---- the workloads as (step object, resident batches, description, flops per step [6 B N d; KL: 8 B N d], flops of the tile launch) ----
def setup_s_dm(dev, rng, t):
    n_ent, n_rel, d, B, N = 14543, 239, 512, 512, 10000
    E, R = (rng.standard_normal((n, d), dtype=np.float32) * 0.1 for n in (n_ent, n_rel))
    step = FusedTrainStep(t(E), t(R), "distmult", lr=0.1, input_dropout=0.2, seed=1)
    batches = [positives_batch(rng, t, n_ent, n_rel, B, N, 2, cand_ids=t(rng.permutation(np.arange(2, n_ent))[:N].astype(np.int32)), cand_unique=True)
               for _ in range(4)]
    return step, batches, f"S-DM: DistMult d={d}, B={B}, batch-shared sampled N={N}, input_dropout 0.2, BCE, dense Adagrad", 6.0 * B * N * d, 4.0 * B * N * d


def setup_s_fb_kl(dev, rng, t):
    from open_knowledge_graph_embeddings_amd import synthetic
    w = synthetic.WORKLOADS["S-FB"]
    E, R = synthetic.make_tables(w, seed=1234)
    step = FusedTrainStep(t(E), t(R), w.scorer, loss="kl", lr=w.lr, input_dropout=w.input_dropout, seed=1)
    batches = []
    for i in range(4):
        hb = synthetic.make_batch(w, seed=1234 + i)
        batches.append(H.PrefixBatch(po_rel=t(hb["po_rel"]), po_obj=t(hb["po_obj"]), sp_subj=t(hb["sp_subj"]), sp_rel=t(hb["sp_rel"]),
                                     pos_row=t(hb["pos_row"]), pos_col=t(hb["pos_col"]), cand_first=2, n_cand=w.N))
    return step, batches, f"S-FB-kl: S-FB with the softmax / KL loss (one more score pass for the row log-sum-exp)", 8.0 * w.B * w.N * w.d, 4.0 * w.B * w.N * w.d


def setup_s_olp_tok(dev, rng, t):
    from open_knowledge_graph_embeddings_amd.synthetic import make_token_matrix
    from open_knowledge_graph_embeddings_amd.token_pooled import TokenPooledTrainStep, TokenSlot
    n_ent, n_rel, d, B, N, L, vt_e, vt_r = 2_500_000, 100_000, 256, 4096, 8192, 10, 200_000, 50_000
    ent = TokenSlot(torch.randn((vt_e, d), device=dev) * 0.1, t(make_token_matrix(rng, n_ent, vt_e, L)), "sum", True)
    rel = TokenSlot(torch.randn((vt_r, d), device=dev) * 0.1, t(make_token_matrix(rng, n_rel, vt_r, L)), "sum", True)
    step = TokenPooledTrainStep(ent, rel, "complex", lr=0.1, dropout=0.1, seed=1)
    # (eight resident batches: a rare token row named by one of them is named again eight steps later -- the deferred decay
    #  steps it catches up on then are as many as a training run's, where no batch repeats; with two the lag would be one)
    batches = [positives_batch(rng, t, n_ent, n_rel, B, N, 1, cand_ids=t(rng.choice(n_ent - 2, N, replace=False).astype(np.int32) + 2))
               for _ in range(int(os.environ.get("OKGE_TOK_BATCHES", "8")))]
    return step, batches, (f"S-OLP-tok: token-pooled ComplEx d={d}, B={B}, batch-shared N={N}, {L} tokens per entity from a {vt_e} / {vt_r} Zipf "
                           f"vocabulary, sum pooling + batch-norm, dropout 0.1, BCE, dense Adagrad over the token tables"), 6.0 * B * N * d, 4.0 * B * N * d


SETUPS = {"S-DM": setup_s_dm, "S-FB-kl": setup_s_fb_kl, "S-OLP-tok": setup_s_olp_tok}


def measure_config(name, dev, warmup=50, steps=200, peak_tflops=157.3):
    """one workload through the same code path as the command line, for bench.py's `configs` object"""
    rng = np.random.default_rng(1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)          # noqa: E731
    step, batches, what, flops_step, flops_tile = SETUPS[name](dev, rng, t)
    out = run(name, step, batches, steps=steps, warmup=warmup, ksteps=50, quiet=True)
    tile_us = out["kernels_us"].get("fused_tile_train")
    res = {"workload": what, "ms_per_step": out["ms_per_step"], "prefixes_per_s": out["prefixes_per_s"], "steps": steps, "warmup": warmup,
           "kernels_us": out["kernels_us"], "step_frac": flops_step / (out["ms_per_step"] * 1e-3) / 1e12 / peak_tflops}
    if tile_us:
        res["tile_us"] = tile_us
        res["tile_frac"] = flops_tile / (tile_us * 1e-6) / 1e12 / peak_tflops
    if name == "S-OLP-tok":
        # the same step on the Adagrad state of a run in progress (every accumulator has seen a real gradient: a decay-only update
        # then returns its input bits -- main() below, "accumulators of a run in progress"), and what the optimizer does per step
        res["decay_window"] = step.decay_window
        step.flush()
        for sl in (step.entity, step.relation):
            sl.sumW.fill_(1e-4)
        warm = run(name, step, batches, steps=steps, warmup=warmup, ksteps=10, quiet=True)
        res["ms_per_step_warm_accumulators"] = warm["ms_per_step"]
        res["adagrad_us_warm_accumulators"] = warm["kernels_us"].get("adagrad")
    del step, batches
    torch.cuda.empty_cache()
    return res


def rank_shape(world, dev, t):
    """what ONE of `world` ranks computes per step of `bench.py --gpus world` (weak scaling: global batch world x 512 rows
    against this rank's share of the FB15k-237-shaped candidates); the two all-reduces are not part of this timing"""
    from open_knowledge_graph_embeddings_amd import synthetic
    from open_knowledge_graph_embeddings_amd.sharded import shard_range
    import dataclasses
    w = synthetic.WORKLOADS["S-FB"]
    rank = world // 2
    wg = dataclasses.replace(w, n_po=w.n_po * world, n_sp=w.n_sp * world)
    E, R = synthetic.make_tables(w, seed=1234)
    lo, hi = shard_range(w.n_ent, world, rank)
    Et, Rt = t(E[lo:hi].copy()), t(R)
    eng = H.HotPath(dev)
    hb = synthetic.make_batch(wg, seed=5)
    batch = H.PrefixBatch(po_rel=t(hb["po_rel"]), po_obj=t(hb["po_obj"]), sp_subj=t(hb["sp_subj"]), sp_rel=t(hb["sp_rel"]),
                          pos_row=t(hb["pos_row"]), pos_col=t(hb["pos_col"]), cand_first=2, n_cand=w.N)
    shard = H.Shard(lo, hi, lo - 2)
    local = H.PrefixBatch(po_rel=batch.po_rel, po_obj=batch.po_obj, sp_subj=batch.sp_subj, sp_rel=batch.sp_rel,
                          pos_row=batch.pos_row, pos_col=batch.pos_col, cand_first=0, n_cand=hi - lo)
    dE, dR = torch.zeros_like(Et), torch.zeros_like(Rt)
    sE, sR = torch.zeros_like(Et), torch.zeros_like(Rt)
    loss = torch.zeros(1, dtype=torch.float64, device=dev)

    from open_knowledge_graph_embeddings_amd.sharded import make_row_segments
    segs = make_row_segments(hb["po_rel"], hb["po_obj"], hb["sp_subj"], hb["sp_rel"], dev) if os.environ.get("OKGE_ROW_SEGMENTS", "1") == "1" else None

    def one():
        qe = eng.encode_queries(Et, Rt, w.scorer, batch, shard)
        dq = torch.empty_like(qe[0])
        eng.train_tiles(Et, Rt, w.scorer, qe[0], local, shard, dE, dq, w.N, normalizer=float(wg.B) * w.N, loss_out=loss,
                        grads_zero=True)
        eng.prefix_backward(Et, Rt, w.scorer, batch, shard, dq, qe[1], dE, dR, rel_segments=segs)
        eng.adagrad2(Et, dE, sE, Rt, dR, sR, w.lr)
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        one()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 50
    eng.timing(True)
    for _ in range(10):
        one()
    torch.cuda.synchronize()
    per = {k: round(v[0] / v[1] * 1e3, 2) for k, v in eng.timing_collect().items()}
    eng.timing(False)
    print(json.dumps({"workload": f"S-FB-rank{world} (one rank's compute, no exchange)", "ms_per_step": round(ms, 4), "kernels_us": per}), flush=True)



def main():
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)          # noqa: E731
    which = sys.argv[1:] or ["S-DM", "S-OLP-shard", "S-OLP-tok"]
    if "S-FB-kl" in which:
        # configs[1] with the softmax / KL loss (trainer.py:99-101): one extra score pass for the row log-sum-exp
        step, batches = setup_s_fb_kl(dev, rng, t)[:2]
        run("S-FB-kl", step, batches, steps=100, warmup=10)
        del step, batches
    if "S-DM" in which:
        step, batches = setup_s_dm(dev, rng, t)[:2]
        run("S-DM", step, batches)
        del step, batches
    if "S-OLP-shard" in which:
        n_ent, n_rel, d, B = 312_500 + 2, 100_000, 256, 4096
        E = torch.randn((n_ent, d), device=dev) * 0.1
        R = torch.randn((n_rel, d), device=dev) * 0.1
        step = FusedTrainStep(E, R, "complex", lr=0.1, seed=1)
        batches = []
        for _ in range(2):
            pr, pc = positives(rng, B, n_ent - 2, per_row=1)
            batches.append(H.PrefixBatch(po_rel=t(rng.integers(2, n_rel, B // 2).astype(np.int32)), po_obj=t(rng.integers(2, n_ent, B // 2).astype(np.int32)),
                                         sp_subj=t(rng.integers(2, n_ent, B // 2).astype(np.int32)), sp_rel=t(rng.integers(2, n_rel, B // 2).astype(np.int32)),
                                         pos_row=t(pr), pos_col=t(pc), cand_first=2, n_cand=n_ent - 2))
        out = run("S-OLP-shard", step, batches, steps=5, warmup=2)
        flops = 6.0 * B * (n_ent - 2) * d
        print(json.dumps({"S-OLP-shard step TFLOP/s (6BNd)": round(flops / out["ms_per_step"] / 1e9, 1)}), flush=True)
        del step, batches, E, R
    for world in (2, 4, 8):
        if f"S-FB-rank{world}" in which:
            rank_shape(world, dev, t)
    if "S-OLP-tok" in which or "S-OLP-tok-short" in which:
        step, batches = setup_s_olp_tok(dev, rng, t)[:2]
        ent, rel = step.entity, step.relation
        run("S-OLP-tok", step, batches, steps=20, warmup=3)
        if "S-OLP-tok-short" in which:
            return
        # the same step on the Adagrad state of a run IN PROGRESS: once a row's accumulator has seen a real gradient, the
        # weight-decay-only update of a step that does not touch the row leaves its bits unchanged and the sweep skips the
        # two stores (okge_misc.hip adagrad_sweep).  Two resident batches touch ~15 % of the token rows; in the line above
        # the other rows' accumulators are still ~1e-22 and move every step, as in the first steps of a run.
        step.flush()
        for sl in (ent, rel):
            sl.sumW.fill_(1e-4)
        run("S-OLP-tok (accumulators of a run in progress)", step, batches, steps=20, warmup=3)
        # the same step replayed as a HIP graph (~50 launches collapse into one)
        from open_knowledge_graph_embeddings_amd.train_step import GraphedTrainStep
        g = GraphedTrainStep(step, batches[0], pos_capacity=max(b.nnz for b in batches))
        for i in range(3):
            g.step(batches[i % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            g.step(batches[i % 2])
        torch.cuda.synchronize()
        print(json.dumps({"workload": "S-OLP-tok (HIP graph replay)", "ms_per_step": round(1e3 * (time.perf_counter() - t0) / 20, 4)}), flush=True)


if __name__ == "__main__":
    main()
