#!/usr/bin/env python3
"""Benchmark of the open-KGE hot path on MI355X: training triples/sec, 1-vs-all ComplEx d=200
(BASELINE.json metric), synthetic FB15k-237-shaped data (workload "S-FB").

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: gather+dropout, score all candidates, BCE loss,
backward, dense Adagrad on both tables (what Trainer.compute_one_batch does, openkge/trainer.py:181-257).
All inputs (tables, batches) are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

N > 1: the entity table (and its Adagrad state) is row-sharded over the N GPUs and every rank scores the SAME
global batch of 512*N prefixes against its own candidates (open_knowledge_graph_embeddings_amd/sharded.py:
two small RCCL all-reduces per step).  Per-GPU work is constant in N ("weak" scaling); `value` counts each
global batch once.  The line also carries an `olp` object: the north-star's second shape (S-OLP, |E| = 2.5 M, d = 256,
B = 4096, Zipf(1.1) prefix entities) at the SAME global size for every N (strong scaling), a handful of steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
N_BATCHES = 8                   # distinct pre-generated batches cycled through


def to_dev_batch(hb, w, dev):
    from open_knowledge_graph_embeddings_amd.hotpath import PrefixBatch
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    return PrefixBatch(po_rel=t(hb["po_rel"]), po_obj=t(hb["po_obj"]), sp_subj=t(hb["sp_subj"]), sp_rel=t(hb["sp_rel"]),
                       pos_row=t(hb["pos_row"]), pos_col=t(hb["pos_col"]), cand_first=2, n_cand=w.N)


def host_cores():
    """Cores this process may actually use: affinity, then the cgroup CPU quota; the GPU boxes expose all 256
    host threads to os.cpu_count() but give a one-GPU job a 16-core share."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    if "OKGE_CPU_THREADS" in os.environ:
        n = int(os.environ["OKGE_CPU_THREADS"])
    elif n > 32:
        n = 16          # documented CPU share of a one-GPU box
    return n


def cpu_baseline(w, host_batches, budget_s=20.0):
    """The reference's ATen op sequence on the host cores (oracle/torch_twin.py, kind 'port'), bounded."""
    from oracle import torch_twin
    from open_knowledge_graph_embeddings_amd import synthetic
    # the reference uses every core it sees (trainer.py:136); here: every core this job may use
    torch.set_num_threads(host_cores())
    torch.manual_seed(1234)
    m = torch_twin.TwinModel(w.scorer, w.n_ent, w.n_rel, w.d, input_dropout=w.input_dropout, init_std=w.init_std)
    m.train()
    opt = torch_twin.make_adagrad(m, lr=w.lr)
    cand = torch.arange(w.n_ent)[2:].int().unsqueeze(1)
    prepared = []
    for hb in host_batches[:4]:
        y = torch.from_numpy(synthetic.dense_labels(hb, w.B, w.N))
        po = (torch.from_numpy(hb["po_rel"]).unsqueeze(1), torch.from_numpy(hb["po_obj"]).unsqueeze(1))
        sp = (torch.from_numpy(hb["sp_subj"]).unsqueeze(1), torch.from_numpy(hb["sp_rel"]).unsqueeze(1))
        prepared.append((po, sp, y, hb["n_pos"]))
    for i in range(3):
        po, sp, y, _ = prepared[i % len(prepared)]
        torch_twin.train_step(m, opt, po, sp, cand, y)
    t0 = time.perf_counter()
    steps, triples = 0, 0
    while True:
        po, sp, y, n_pos = prepared[steps % len(prepared)]
        torch_twin.train_step(m, opt, po, sp, cand, y)
        steps += 1
        triples += n_pos
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 200:
            break
    return {"value": triples / el, "unit": "triples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} steps of {w.name} (B={w.B}, N={w.N}, d={w.d}) in {el:.1f}s, torch-CPU twin of the "
                      f"reference op sequence, dense labels prebuilt, no dataloader",
            "ms_per_step": 1e3 * el / steps}


def run_dropin(w, host_batches, dev, steps=300, warmup=30):
    """ms/step of the drop-in path (tools/bench_dropin.py has the slower configurations beside it): AddLossModule with
    training_outputs=False, OkgeAdagrad, labels as column-sorted coordinates, the reference Trainer's statements verbatim"""
    from open_knowledge_graph_embeddings_amd.dataset import EntityRelationDatasetMeta
    from open_knowledge_graph_embeddings_amd.model import Models
    from open_knowledge_graph_embeddings_amd.optim import OkgeAdagrad
    from open_knowledge_graph_embeddings_amd.trainer import AddLossModule
    torch.manual_seed(0)
    m = Models.LookupComplexRelationModel(entity_slot_size=w.d, input_dropout=w.input_dropout, init_std=w.init_std, sparse=False,
                                          train_data=EntityRelationDatasetMeta(entities_size=w.n_ent, relations_size=w.n_rel)).to(dev)
    m.train()
    mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0, training_outputs=False)
    opt = OkgeAdagrad(m.parameters(), lr=w.lr, weight_decay=1e-10, eps=1e-8)
    cand = torch.arange(w.n_ent, device=dev)[2:].int().unsqueeze(1)
    t = lambda a: torch.from_numpy(a).to(dev).unsqueeze(1)  # noqa: E731
    batches = [([(t(hb["po_rel"]), t(hb["po_obj"])), (t(hb["sp_subj"]), t(hb["sp_rel"]))],
                (torch.from_numpy(hb["pos_row"]).to(dev), torch.from_numpy(hb["pos_col"]).to(dev))) for hb in host_batches]
    norm = float(w.B * w.N)

    def step(i):
        inputs, coords = batches[i % len(batches)]
        opt.zero_grad()
        loss, _, _ = mod(inputs=inputs, labels=coords, use_batch_shared_entities=False, batch_shared_entities=cand, epoch=1,
                         input_style_triple_or_prefix="right_and_left_prefix")
        (loss.sum() / norm).backward()
        opt.step()
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    windows = []
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        windows.append(1e3 * (time.perf_counter() - t0) / steps)
    return {"ms_per_step": float(np.median(windows)), "ms_per_step_min": min(windows), "ms_per_step_max": max(windows),
            "steps": steps, "path": "Models.LookupComplexRelationModel + AddLossModule(training_outputs=False) + OkgeAdagrad, "
                                    "coordinate labels, (loss.sum() / normalizer).backward() as trainer.py:217-234"}


def run_olp(world, rank, dev, dist, barrier, steps=6, warmup=2):
    """S-OLP at the full size on `world` GPUs: tables generated on the device from per-row-block seeds (identical for
    every N), Zipf(1.1) prefix entity ids, one positive per row, dropout 0 (the reference's OLPBENCH
    configs train without input dropout), BCE, dense Adagrad.  Returns the sub-object rank 0 prints."""
    from open_knowledge_graph_embeddings_amd import synthetic
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
    w = synthetic.WORKLOADS["S-OLP"]
    host_batches = [synthetic.make_batch(w, seed=4321 + i, zipf=True) for i in range(4)]
    batches = [to_dev_batch(hb, w, dev) for hb in host_batches]
    n_pos = [hb["n_pos"] for hb in host_batches]

    def table(rows_lo, rows_hi, d, seed):            # same values whatever the sharding: 65536-row blocks, one seed each
        out = torch.empty((rows_hi - rows_lo, d), dtype=torch.float32, device=dev)
        blk = 65536
        for b0 in range(rows_lo // blk * blk, rows_hi, blk):
            g = torch.Generator(device=dev).manual_seed(seed + b0 // blk)
            t = torch.randn((blk, d), generator=g, device=dev, dtype=torch.float32) * w.init_std
            lo, hi = max(b0, rows_lo), min(b0 + blk, rows_hi)
            out[lo - rows_lo:hi - rows_lo] = t[lo - b0:hi - b0]
        return out

    Rt = table(0, w.n_rel, w.d, 99)
    if world > 1:
        from open_knowledge_graph_embeddings_amd.sharded import ShardedTrainStep, make_exchange_plan, shard_range
        lo, hi = shard_range(w.n_ent, world, rank)
        step = ShardedTrainStep(table(lo, hi, w.d, 7), Rt, w.scorer, w.n_ent, lr=w.lr, loss=w.loss, seed=1234)
        plans = [make_exchange_plan(hb["po_obj"], hb["sp_subj"], w.n_ent, world, dev) for hb in host_batches]   # Zipf ids: may be None
        run = lambda i: step.step(batches[i % 4], plan=plans[i % 4])           # noqa: E731
    else:
        step = FusedTrainStep(table(0, w.n_ent, w.d, 7), Rt, w.scorer, loss=w.loss, lr=w.lr, seed=1234)
        run = lambda i: step.step(batches[i % 4])                               # noqa: E731
    for i in range(warmup):
        run(i)
    barrier()
    t0 = time.perf_counter()
    triples = 0
    for i in range(steps):
        run(i)
        triples += n_pos[i % 4]
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        el = float(tmax[0].item())
    flops = 6.0 * w.B * w.N * w.d
    return {"workload": f"S-OLP: OLPBENCH-shaped |E|={w.n_ent} |R|={w.n_rel} d={w.d} ComplEx 1-vs-all, B={w.B}, Zipf(1.1) prefix "
                        f"entities, BCE, dense Adagrad", "scaling": "strong", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * el / steps, "value": triples / el, "unit": "triples/s", "prefixes_per_s": w.B * steps / el,
            "step_frac": flops / (el / steps) / 1e12 / FP32_MFMA_PEAK_TFLOPS / world,
            "parallelism": f"entity table row-sharded x{world}" if world > 1 else "single"}


def main():
    # RCCL (and friends) print banners on stdout; the contract is ONE JSON line there.  Everything else -> stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="S-FB")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from open_knowledge_graph_embeddings_amd import synthetic
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # OKGE_BENCH_ONE_GPU=1: rehearsal of the multi-rank control flow on a one-GPU box -- every rank on cuda:0, gloo
    # instead of RCCL (which refuses two ranks per device).  Not a measurement.
    one_gpu = os.environ.get("OKGE_BENCH_ONE_GPU") == "1"
    dev = torch.device("cuda", 0 if one_gpu else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a dead or wedged rank ends the run with an error after five minutes (the process exits non-zero) instead of hanging
        import datetime
        if one_gpu:
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=5))
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=5))

    import dataclasses
    w = synthetic.WORKLOADS[args.workload]
    E, R = synthetic.make_tables(w, seed=1234)
    sharded = world > 1 or os.environ.get("OKGE_BENCH_FORCE_SHARDED") == "1"
    if sharded:
        if dist is None:                         # single-GPU rehearsal of the RCCL path
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            import datetime
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(minutes=5))
        from open_knowledge_graph_embeddings_amd.sharded import ShardedTrainStep, shard_range
        # global batch grows with the number of GPUs; identical on every rank (same seeds)
        wg = dataclasses.replace(w, n_po=w.n_po * world, n_sp=w.n_sp * world)
        host_batches = [synthetic.make_batch(wg, seed=1234 + i) for i in range(N_BATCHES)]
        lo, hi = shard_range(w.n_ent, world, rank)
        Et, Rt = torch.from_numpy(E[lo:hi].copy()).to(dev), torch.from_numpy(R).to(dev)
        step = ShardedTrainStep(Et, Rt, w.scorer, w.n_ent, lr=w.lr, loss=w.loss, input_dropout=w.input_dropout, seed=1234)
        batches = [to_dev_batch(hb, wg, dev) for hb in host_batches]
        # exchange 1 as an all-gather of the rows each rank owns: the plan is host work on ids the host already has
        from open_knowledge_graph_embeddings_amd.sharded import make_exchange_plan
        plans = [make_exchange_plan(hb["po_obj"], hb["sp_subj"], w.n_ent, world, dev) for hb in host_batches]
        # ... and so are the row segments of the prefix backward (relation / entity gradients without float atomics; None for
        # batches under 1024 rows, where the atomics are cheaper than the second launch)
        from open_knowledge_graph_embeddings_amd.sharded import make_row_segments
        segs = [make_row_segments(hb["po_rel"], hb["po_obj"], hb["sp_subj"], hb["sp_rel"], dev) for hb in host_batches]
        w_run = wg
    else:
        host_batches = [synthetic.make_batch(w, seed=1234 + i) for i in range(N_BATCHES)]
        Et, Rt = torch.from_numpy(E).to(dev), torch.from_numpy(R).to(dev)
        step = FusedTrainStep(Et, Rt, w.scorer, loss=w.loss, lr=w.lr, input_dropout=w.input_dropout, seed=1234)
        batches = [to_dev_batch(hb, w, dev) for hb in host_batches]
        w_run = w
    n_pos = [hb["n_pos"] for hb in host_batches]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # OKGE_BENCH_GRAPH=1 replays one captured HIP graph per resident batch instead of launching from Python (same
    # kernels; the dropout counter is a device scalar incremented inside the graph).  Measured on MI355X: 0.154 ms/step
    # replayed vs 0.144 launched -- graph nodes dispatch with wider gaps than back-to-back stream launches and the
    # host keeps ahead of a 0.15 ms step anyway -- so the default is plain launches.
    run = lambda i: step.step(batches[i % N_BATCHES])                       # noqa: E731
    if sharded:
        run = lambda i: step.step(batches[i % N_BATCHES], plan=plans[i % N_BATCHES], rel_segments=segs[i % N_BATCHES])   # noqa: E731
    # (sharded: the captured step carries its RCCL collectives -- exchange 1 as the all-reduce, since a plan's shapes vary per
    #  batch -- validated at world size 1 by tests/test_sharded.py::test_sharded_step_captured_in_a_hip_graph_rccl_one_rank; a
    #  rank's sharded step takes ~0.16 ms of host time to launch against ~0.15 ms on the device, so on a multi-GPU node the
    #  replay (0.05-0.06 ms of host time) is the lever if the ranks turn out host-bound.  Opt-in until it has run on N > 1.)
    if os.environ.get("OKGE_BENCH_GRAPH", "0") == "1":
        from open_knowledge_graph_embeddings_amd.train_step import GraphedTrainStep
        g0 = GraphedTrainStep(step, batches[0], pos_capacity=batches[0].nnz)
        graphs = [g0] + [GraphedTrainStep(step, b, pos_capacity=b.nnz, counter=g0.counter) for b in batches[1:]]
        run = lambda i: graphs[i % N_BATCHES].replay()                      # noqa: E731
    for i in range(args.warmup):
        run(i)

    def timed_window():
        """EXACTLY args.steps steps between two barrier + synchronize brackets; wall time = the slowest rank's"""
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            run(i)
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax[0].item())        # every rank processed the same global batches: count them once
        return el

    # A short window (the driver's --steps 20 is 2.8 ms of device time) is at the mercy of one clock ramp or one host
    # hiccup, so the measurement is repeated: up to 9 windows of `steps` steps each (fewer when one window is long: about
    # 20 s in total; the count follows from the rank-maximum time, so every rank runs the same number), and the line reports
    # the MEDIAN window as ms_per_step / value, with the fastest and slowest beside it.
    windows = [timed_window()]
    n_windows = max(1, min(9, int(20.0 / max(windows[0], 1e-6))))
    if n_windows > 1 and n_windows % 2 == 0:
        n_windows -= 1
    while len(windows) < n_windows:
        windows.append(timed_window())
    elapsed = float(np.median(windows))
    triples = sum(n_pos[i % N_BATCHES] for i in range(args.steps))
    loss_last = float((step.reduce_loss() if sharded else step.loss_out).item())

    # ---- roofline of the dominant kernel: HIP events on the launch stream, separate pass ------------------
    roof = None
    # (every rank runs these steps -- the sharded step contains collectives -- but only rank 0 times its kernels)
    eng = step.engine
    # (outside the timed windows: 200 launches for a stable per-kernel average, whatever --steps says -- with the driver's
    #  --steps 20 the whole run is a few tens of milliseconds and the first launches still see the clocks ramping)
    ksteps = 200
    for i in range(50):
        run(i)
    barrier()
    if rank == 0:
        eng.timing(True)
    for i in range(ksteps):
        run(i)
    barrier()
    if rank == 0:
        per_kernel = eng.timing_collect()
        eng.timing(False)
        tot_ms, cnt = per_kernel["fused_tile_train"]
        avg_s = tot_ms / cnt * 1e-3
        n_local = step.n_cand_local if sharded else w.N
        flops = 4.0 * w_run.B * n_local * w.d         # X = Q.C^T (2BNd) + dC = G^T.Q (2BNd) per launch, this rank
        achieved = flops / avg_s / 1e12
        # HBM bytes per launch of that kernel: from the committed PMC passes of this same command (tools/collect_profiles.sh:
        # separate FETCH_SIZE / WRITE_SIZE runs, FETCH doubled as the gfx950 guide says); counters cannot be read inside
        # a timed run, so the value is tagged with the file and the commit it was collected at
        traffic, traffic_src = None, None
        kname = "fused_tile64_kernel" if w.d <= 256 else "fused_tile32_kernel"
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and not sharded and args.workload == "S-FB":
            tj = json.load(open(tpath))
            hit = [v for k, v in tj.items() if k.startswith(kname)]
            if hit:
                traffic = hit[0]["hbm_bytes_per_launch"]
                traffic_src = {"file": "profiles/pmc_traffic.json", "collected_at_commit": tj.get("_commit"),
                               "command": tj.get("_command"), "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes"}
        step_s = elapsed / args.steps
        roof = {"bound": "mfma", "kernel": kname + "<train>", "achieved": achieved,
                "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": avg_s * 1e6,
                "kernels_us": {k: v[0] / v[1] * 1e3 for k, v in per_kernel.items()},
                "step_flops_6BNd": 6.0 * w_run.B * n_local * w.d, "step_bytes_20Nd": 20.0 * n_local * w.d,
                # the whole step against the same peak: 6 B N d flops (this rank's candidates) / measured step time
                "step_frac": 6.0 * w_run.B * n_local * w.d / step_s / 1e12 / FP32_MFMA_PEAK_TFLOPS}

    # ---- evaluation leg (single GPU): score every candidate + filtered ranks (dataset.py:423-446) --------------
    ev = None
    if rank == 0 and not sharded:
        hb = synthetic.make_eval_batch(w, seed=777)
        eb = to_dev_batch(hb, w, dev)
        t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
        csr = [t(hb[k]) for k in ("filt_ptr", "filt_col", "row_ptr", "grp_ptr", "ids")]
        from open_knowledge_graph_embeddings_amd.dataset import CollatedBatch
        from open_knowledge_graph_embeddings_amd.evaluate import FusedEvaluator, PipelinedEvaluator
        cb = CollatedBatch(eb, float(w.B * w.N), float(hb["n_pos"]), w.N, row_ptr=csr[2], grp_ptr=csr[3], ids=csr[4],
                           filt_ptr=csr[0], filt_col=csr[1])
        n_it = 70                        # one validation pass at the FB15k-237 shape: 2 x 17 535 prefixes / 512
        ev = {}
        # fused: point scores + tile sweep counting in registers + ranks/meters, no (B, N) score block (okge_evaluate_fused);
        # pipelined: the materialising path (scores, ranks, meters per batch; independent chains on three streams): any
        # slot size, dropout
        # (the materialising evaluator first: created after another evaluator has used its streams, its three chains shared
        #  hardware queues on the test boxes -- 0.052 instead of 0.039 ms per batch; each evaluator alone gets the lower figure)
        for name, cls in (("pipelined", PipelinedEvaluator), ("fused", FusedEvaluator)):
            ev_run = cls(Et, Rt, w.scorer, engine=step.engine)
            ev_run.run([cb] * 192)      # warm-up: fresh streams are slow until the runtime's per-queue pools have grown to the
                                        # depth a full run of batches keeps in flight (first 640-batch pass: 0.11 ms per batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res, n_groups = ev_run.run([cb] * n_it)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if name == "fused":
                ev.update({"ms_per_batch": 1e3 * el / n_it, "prefixes_per_s": w.B * n_it / el, "groups": n_groups // n_it,
                           "mrr": res["mrr"].avg, "path": "okge_evaluate_fused_batches", "batches": n_it})
                t0 = time.perf_counter()                 # steady state: start-up and the final host read amortised
                ev_run.run([cb] * 640)
                torch.cuda.synchronize()
                ev["steady_ms_per_batch"] = 1e3 * (time.perf_counter() - t0) / 640
            else:
                ev.update({"pipelined_ms_per_batch": 1e3 * el / n_it, "pipelined_mrr": res["mrr"].avg})
                t0 = time.perf_counter()
                ev_run.run([cb] * 640)
                torch.cuda.synchronize()
                ev["pipelined_steady_ms_per_batch"] = 1e3 * (time.perf_counter() - t0) / 640

    # ---- drop-in leg (single GPU, untimed region like `eval`): the reference Trainer's own step sequence (trainer.py:206-244:
    #      zero_grad, AddLossModule forward, (loss.sum() / normalizer).backward(), optimizer.step()) on this package's Models /
    #      AddLossModule / OkgeAdagrad -- what a user of INTEGRATION.md section 1 gets without touching the training loop
    dropin = None
    if rank == 0 and not sharded and args.workload == "S-FB" and os.environ.get("OKGE_BENCH_DROPIN", "1") == "1":
        dropin = run_dropin(w, host_batches, dev)

    # ---- the north-star's second shape: S-OLP (|E| = 2.5 M, |R| = 100 k, d = 256, B = 4096, Zipf(1.1) prefix entities),
    #      the SAME global problem at every N (strong scaling), entity table row-sharded over the ranks; a handful of steps
    olp = None
    if args.workload == "S-FB" and os.environ.get("OKGE_BENCH_OLP", "1") == "1":
        del step, batches, Et, Rt
        torch.cuda.empty_cache()
        olp = run_olp(world, rank, dev, dist, barrier)

    # ---- the other BASELINE configurations on the driver's clock (single GPU, untimed region like `eval` / `dropin`):
    #      configs[2] S-DM, configs[4] S-OLP-tok and configs[1] with the KL loss, through tools/bench_configs.py's code path,
    #      50 warm-up + 200 timed steps each, per-kernel HIP-event averages beside the step time
    configs = None
    if rank == 0 and world == 1 and not sharded and args.workload == "S-FB" and os.environ.get("OKGE_BENCH_CONFIGS", "1") == "1":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs
        configs = {name: bench_configs.measure_config(name, dev, peak_tflops=FP32_MFMA_PEAK_TFLOPS) for name in ("S-DM", "S-OLP-tok", "S-FB-kl")}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    cpu = None
    if not args.no_cpu_baseline and world == 1 and not sharded:
        cpu = cpu_baseline(w, host_batches)
    line = {
        "metric": "training triples/sec (1-vs-all ComplEx d=200)" if w.name.startswith("S-FB") and w.d == 200
                  else f"training triples/sec (1-vs-all {w.scorer} d={w.d})", "value": triples / elapsed, "unit": "triples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "ms_per_step_min": 1e3 * min(windows) / args.steps, "ms_per_step_max": 1e3 * max(windows) / args.steps,
        "timed_windows": len(windows),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{w.name}: {'FB15k-237' if w.name.startswith('S-FB') else 'OLPBENCH'}-shaped |E|={w.n_ent} |R|={w.n_rel} d={w.d} ComplEx 1-vs-all "
                               f"N={w.N}, B={w_run.B} ({w_run.n_po} po + {w_run.n_sp} sp), BCE, input_dropout "
                               f"{w.input_dropout}, dense Adagrad lr {w.lr}",
                   "global_batch": w_run.B,
                   "parallelism": f"entity table row-sharded x{world}, batch 512 x{world}" if sharded else "single"},
        "prefixes_per_s": w_run.B * args.steps / elapsed, "last_loss_sum": loss_last,
        "roofline": roof, "cpu_baseline": cpu, "eval": ev, "dropin": dropin, "olp": olp, "configs": configs,
    }
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
