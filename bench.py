#!/usr/bin/env python3
"""Headline benchmark: query placements/sec on the 10k-leaf tree, 150 bp reads.

    python bench.py --gpus N --steps K --warmup W [--config C3]

A "step" is one pass of the placement hot path (`cls_place_batch_device`, the
C-ABI entry a Rust caller would bind) over one batch of synthetic reads that is
already resident in HBM.  At N=1 the workload is BASELINE.json configs[2]
("C3": 10k-leaf tree, 1M x 150 bp reads, k=12).  At N>1 every rank holds the
same index and its own shard of ONE global read stream, and each step ends with
the single gather of the 24-byte placement records to rank 0 over RCCL:
  * default (C3): 1M reads per GPU -> "weak" scaling;
  * --config C4 (BASELINE configs[3]): ONE stream of 10M reads, ceil(10M/N) per
    rank -> "strong" scaling.
Other --config values are the shapes SURVEY.md 8 / VERDICT name: C2 (1k-leaf,
k=8), C5 (50k-leaf deep tree, 10 kb reads, k=15), C3s12 / C3s35 (the
support-collapsed 10k-leaf tree at k=12 and at the reference's default k=35), G35
(the reference's documented workload: ~1.9 kb gyrB queries, ~590-node
support-collapsed tree, k=35, m=4).

Rank 0 prints ONE JSON line; DESIGN.md "Measurement" says how `roofline`,
`host_window` and `cpu_baseline` are obtained.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def survey_model_bytes(read_lens, stats):
    """B(q) = L + 16*2(L-k+1) + 4*sum_{h in M(q)} |leaves(h)| + 24 -- a model that STREAMS posting lists;
    this index answers membership from 8-byte split halves instead, so the figure exceeds what any launch moves."""
    return int(read_lens.sum()) + 16 * int(stats["n_query_kmers"].astype(np.int64).sum()) + 4 * int(
        stats["leaf_postings"].astype(np.int64).sum()) + 24 * len(stats)


def needed_bytes(read_lens, stats):
    """Bytes THIS index must touch to place the batch, counted exactly by the statistics kernel
    (cls_query_stats.index_bytes: table entries, node records, split halves) + per read its bases, its two 8-byte
    offsets' share (8), its 4-byte entry of the read list and the 24-byte record.  None if the kernels that ran do
    not count index bytes."""
    ib = int(stats["index_bytes"].astype(np.int64).sum())
    if ib == 0:
        return None
    return int(read_lens.sum()) + ib + (8 + 4 + 24) * len(stats)


def needed_bytes_floor(read_lens, stats):
    """For the kernels that do not count their index bytes (the generic workgroup-per-read path): a LOWER bound from the
    oracle-checked counters -- one 16-byte hash-table slot per query k-mer (>= 1 probe each), one 16-byte set record per
    distinct matched k-mer, + bases, offsets, list entry, record.  Node records and split halves are not in it."""
    return (int(read_lens.sum()) + 16 * int(stats["n_query_kmers"].astype(np.int64).sum())
            + 16 * int(stats["n_matched"].astype(np.int64).sum()) + (8 + 4 + 24) * len(stats))


def source_sha16():
    """Fingerprint of the kernel sources a profile belongs to (the GPU box has no .git)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "classeq2_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".cpp", ".h")):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def traffic_lookup(config, reads, kernel):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE / WRITE_SIZE collected in separate --pmc runs of this same command, tools/profile.sh).  An entry
    only counts when it was taken from THESE sources and THIS kernel instance; otherwise (None, True)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            e = json.load(f).get(f"{config}:{reads}")
    except (OSError, ValueError):
        return None, False
    if not e:
        return None, False
    if e.get("source_sha16") != source_sha16() or e.get("kernel") != kernel:
        return None, True
    return e.get("hbm_bytes_per_launch"), False


def gather_reference():
    """Measured random-gather rate of this chip (tools/gather_probe.hip, committed summary), if present."""
    try:
        with open(os.path.join(ROOT, "profiles", "gather_probe.json")) as f:
            return json.load(f).get("summary")
    except (OSError, ValueError):
        return None


def cpu_baseline(synth, cfg, first, budget_s=15.0):
    """The C oracle (oracle/cls_oracle.c, kind "port") on this host's cores, on a bounded prefix of the same read
    stream.  Returns (json object, oracle records of that prefix)."""
    from oracle.oracle_port import OraclePort

    cores = min(os.cpu_count() or 1, 64)
    t0 = time.time()
    port = OraclePort(synth.flat)
    build_s = time.time() - t0
    probe = max(8, min(2000, cfg["n_reads"]) if cfg["read_len"] <= 1000 else 16)
    bases, offsets, _ = synth.reads(probe, cfg["read_len"], first=first)
    t0 = time.time()
    port.place_batch(bases, offsets, threads=cores)
    dt = max(time.time() - t0, 1e-4)
    n = int(min(cfg["n_reads"], max(probe, probe * budget_s / dt)))
    bases, offsets, _ = synth.reads(n, cfg["read_len"], first=first)
    t0 = time.time()
    recs = port.place_batch(bases, offsets, threads=cores)
    dt = time.time() - t0
    # what the unmodified Rust path would cost: the port + the reference's per-query deep clone of the index
    # (place_sequence.rs:77-80) and per-bucket key-set rebuild (kmers_map.rs:58-62); an estimate, on a small sample
    port.set_reference_cost(True)
    rc_threads = min(cores, 16)  # (3 M allocations per query: more threads only fight over the allocator)
    m = min(n, 2 * rc_threads)
    t0 = time.time()
    port.place_batch(bases[: m * cfg["read_len"]], offsets[: m + 1], threads=rc_threads)
    dt_ref = time.time() - t0
    port.close()
    return {
        "value": n / dt, "unit": "placements/s", "cores": cores, "kind": "port",
        "sample": f"first {n} reads of the same stream, {dt:.1f} s wall on {cores} threads (index build {build_s:.1f} s untimed)",
        "reference_cost_estimate": {"value": m / dt_ref, "unit": "placements/s", "cores": rc_threads,
                                    "sample": f"first {m} reads, {dt_ref:.1f} s wall",
                                    "what": "the port plus the reference's per-query index clone and bucket key-set rebuild "
                                            "(place_sequence.rs:77-80, kmers_map.rs:58-62); an estimate, not the Rust binary"},
    }, recs


def host_window(db, bases, offsets, n, reps=5):
    """SURVEY.md 8(d) window (i): `cls_place_batch` enter -> return, host buffers in, host records out (H2D of the
    reads, every kernel, D2H of the records; mod.rs:64-67 / 264-267 minus the file stages).  Median of `reps`
    after one warm-up, with pinned and with pageable host memory."""
    import torch

    from classeq2_amd import _abi

    res = {}
    for kind in ("pinned", "pageable"):
        if kind == "pinned":
            hb = torch.from_numpy(bases).pin_memory()
            ho = torch.from_numpy(offsets.view(np.int64)).pin_memory()
            hout = torch.zeros(n * 24, dtype=torch.uint8).pin_memory()
            b, o, out = hb.numpy(), ho.numpy().view(np.uint64), hout.numpy().view(_abi.PLACEMENT_DTYPE)
        else:
            b, o, out = bases, offsets, np.zeros(n, dtype=_abi.PLACEMENT_DTYPE)
        times = []
        for i in range(reps + 1):
            t0 = time.perf_counter()
            db.place_batch(b, o, out=out)
            times.append((time.perf_counter() - t0) * 1e3)
        ms = float(np.median(times[1:]))
        res[kind] = {"ms": ms, "placements_per_s": n / (ms * 1e-3)}
        last = out.copy()
    res["what"] = (f"cls_place_batch (host buffers -> host records) on the same {n} reads, median of {reps} after 1 warm-up; "
                   "`value` above is the device-resident window (cls_place_batch_device)")
    return res, last


def e2e_window(db, synth, cfg, bases, n):
    """SURVEY.md 8(d) window (ii) = the reference's UCPLACE0001 -> 0002 (mod.rs:64-67, 264-267): query FASTA file in,
    result file out (parse + placement + serialisation + write), database load excluded.  One warm-up, then one timed
    run per output format; `window_s` is what cls_place_sequences measures inside, `wall_s` the call."""
    import tempfile

    from classeq2_amd import _abi, engine

    kinds = ["ROOT", "NODE", "LEAF"]
    nodes = synth.flat.nodes
    sys.setrecursionlimit(1_000_000)

    def clade(r):
        d = {"id": int(nodes[r]["id"]), "parent": None if int(nodes[r]["parent"]) == _abi.NO_PARENT else int(nodes[r]["parent"]),
             "kind": kinds[int(nodes[r]["kind"])]}
        if nodes[r]["kind"] == 2:
            d["name"] = f"leaf_{int(nodes[r]['id'])}"
        else:
            d["support"] = 100.0
        d["length"] = 0.01
        if nodes[r]["has_children"]:
            d["children"] = [clade(int(nodes[r]["first_child"]) + i) for i in range(int(nodes[r]["n_children"]))]
        return d

    tmp = tempfile.mkdtemp(prefix="cls_e2e_")
    with open(os.path.join(tmp, "tree.json"), "w") as f:
        json.dump(clade(0), f)
    tree = engine.Tree(os.path.join(tmp, "tree.json"))
    rows = bases[: n * cfg["read_len"]].reshape(n, cfg["read_len"])
    query = os.path.join(tmp, "q.fasta")
    with open(query, "wb") as f:
        for lo in range(0, n, 100_000):
            f.write(b"".join(b">r%d\n" % (lo + i) + bytes(row) + b"\n" for i, row in enumerate(rows[lo:lo + 100_000])))
    out = {"what": "query FASTA file -> result file (cls_place_sequences: parse + placement + serialisation + write; database load "
                   "excluded), one warm-up then one timed run per format", "reads": n, "query_bytes": os.path.getsize(query)}
    for fmt, name in ((engine.FORMAT_JSONL, "jsonl"), (engine.FORMAT_YAML, "yaml")):
        engine.place_sequences(db, tree, query, os.path.join(tmp, "warm"), overwrite=True, fmt=fmt)
        t0 = time.perf_counter()
        got, sec = engine.place_sequences(db, tree, query, os.path.join(tmp, "res"), overwrite=True, fmt=fmt)
        wall = time.perf_counter() - t0
        out[name] = {"window_s": round(sec, 4), "wall_s": round(wall, 4), "placements_per_s": round(got / sec),
                     "result_bytes": os.path.getsize(os.path.join(tmp, "res." + name))}
        assert got == n
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "C4", "C5", "C3s12", "C3s35", "G35"])
    ap.add_argument("--e2e", action="store_true", help="add SURVEY.md 8(d) window (ii): query FASTA file -> result file (cls_place_sequences, "
                    "mod.rs:64-67 / 264-267), JSONL and YAML, on this config's reads")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: the config's; C4: total reads of the stream)")
    ap.add_argument("--scale", type=float, default=1.0, help="C5 only: fraction of the 50k leaves / 1M reads to run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-window", action="store_true")
    ap.add_argument("--dump-records", default="", help="rank 0 writes the records it holds after the last step (all ranks' "
                    "shards in rank order at N > 1) to this .npy file: lets a test check the gathered records against the oracle")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from classeq2_amd import _abi, engine
    from classeq2_amd.synth import CONFIGS, SynthDb

    engine.tuning_from_env()  # CLS_* experiment knobs for A/B runs (none changes a result; the library never reads them itself)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
        args.gpus = world
    # CLS_BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend (records staged through the host) -- lets the
    # N > 1 code path run on a one-GPU box; never a measurement
    rehearsal = os.environ.get("CLS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    cfg = dict(CONFIGS[args.config])
    strong = args.config == "C4"  # one stream of fixed size split over the ranks
    if args.config == "C5" and args.scale != 1.0:
        cfg["n_leaves"] = max(64, int(cfg["n_leaves"] * args.scale))
        cfg["n_reads"] = max(64, int(cfg["n_reads"] * args.scale))
    if strong:
        total_reads = args.reads or cfg["n_reads"]
        per_gpu = (total_reads + world - 1) // world          # contiguous ceil(N/G) blocks
        first = rank * per_gpu
        mine = max(0, min(per_gpu, total_reads - first))
    else:
        per_gpu = mine = args.reads or cfg["n_reads"]
        total_reads = per_gpu * world
        first = rank * per_gpu
    cfg["n_reads"] = per_gpu
    if args.config == "G35":
        engine.set_tuning("time_class", 2)  # 1.9 kb reads: the kernel of the gene-length classes (the LDS-tiled one) is the one to time and name
    t0 = time.time()
    synth = None
    if world == 1 or rank == 0:
        # N > 1: ONE rank generates the synthetic index and every rank's shard of the read stream with all the host's
        # threads and leaves them in shared memory; the others map them (8 generators side by side took minutes on C5)
        synth = SynthDb(cfg["n_leaves"], cfg["ref_len"], cfg["k_size"], cfg["m_size"], deep=cfg["deep"],
                        max_depth=cfg["max_depth"], collapse_prob=cfg.get("collapse_prob", 0.0), threads=os.cpu_count() or 8,
                        tips_only=cfg.get("tips_only", False))
        flat = synth.flat
    share = None
    if world > 1:
        share = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"cls_bench_{os.environ.get('MASTER_PORT', '0')}_{args.config}")
        fields = ("nodes", "bucket_key", "bucket_kmer_off", "kmer_hash", "kmer_node_off", "node_ids")
        if rank == 0:
            os.makedirs(share, exist_ok=True)
            for name in fields:
                np.save(os.path.join(share, name + ".npy"), getattr(flat, name))
            for rk in range(world):
                f_rk = rk * per_gpu
                m_rk = max(0, min(per_gpu, total_reads - f_rk)) if strong else per_gpu
                b_rk, o_rk, _ = synth.reads(max(m_rk, 1), cfg["read_len"], seed=3, first=f_rk)
                np.save(os.path.join(share, f"bases_{rk}.npy"), b_rk)
                np.save(os.path.join(share, f"offsets_{rk}.npy"), o_rk)
        dist.barrier()
        if rank != 0:
            from classeq2_amd.flatdb import FlatDb
            arrs = {name: np.load(os.path.join(share, name + ".npy"), mmap_mode="r") for name in fields}
            flat = FlatDb(k_size=cfg["k_size"], m_size=cfg["m_size"], leaves_only=bool(cfg.get("tips_only", False)), **arrs)
    gen_s = time.time() - t0
    t0 = time.time()
    db = engine.PlacementDb(flat, device=local_rank)
    create_s = time.time() - t0
    db.set_max_read_len(cfg["read_len"])  # the launch follows the longest read: classes beyond it are not launched
    # this rank's shard of the global read stream (seed 3); every rank allocates per_gpu records so that the
    # gather has one shape (the last shard of a strong-scaling split may be shorter: its tail stays zero)
    if world > 1:
        bases = np.load(os.path.join(share, f"bases_{rank}.npy"))
        offsets = np.load(os.path.join(share, f"offsets_{rank}.npy"))
        dist.barrier()  # everyone has read its files
        if rank == 0:
            import shutil
            shutil.rmtree(share, ignore_errors=True)
    else:
        bases, offsets, _ = synth.reads(max(mine, 1), cfg["read_len"], seed=3, first=first)
    setup = torch.tensor([gen_s, create_s], dtype=torch.float64)
    if world > 1:
        setup = setup.to("cpu" if rehearsal else dev)
        dist.all_reduce(setup, op=dist.ReduceOp.MAX)  # the slowest rank's setup
    gen_s, create_s = float(setup[0]), float(setup[1])
    d_bases = torch.from_numpy(bases).to(dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_outs = [torch.zeros(per_gpu * 24, dtype=torch.uint8, device=dev) for _ in range(2)]  # double-buffered records
    d_out = d_outs[0]
    d_stats = torch.zeros(per_gpu * 24, dtype=torch.uint8, device=dev)
    gathered = [[torch.empty_like(d_out) for _ in range(world)] for _ in range(2)] if (world > 1 and rank == 0) else [None, None]
    stream = torch.cuda.current_stream().cuda_stream
    pending = [None, None]  # the gather still reading each buffer
    host_gathered = [None]

    def step(stats_ptr=0, buf=0):
        if pending[buf] is not None:  # the records of two steps ago must have left before they are overwritten
            pending[buf].wait()
            pending[buf] = None
        if mine:
            db.place_batch_device(d_bases.data_ptr(), d_off.data_ptr(), mine, d_outs[buf].data_ptr(), None, stats_ptr, stream)

    def gather_records(buf=0):
        """The path's one collective: every rank's placement records -> rank 0 (RCCL over xGMI), asynchronous:
        the gather of step i runs while step i+1 places into the other buffer."""
        if rehearsal:
            h = d_outs[buf].cpu()
            lst = [torch.empty_like(h) for _ in range(world)] if rank == 0 else None
            dist.gather(h, lst, dst=0)
            host_gathered[0] = lst
        else:
            pending[buf] = dist.gather(d_outs[buf], gathered[buf], dst=0, async_op=True)

    def sync_all():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: per-query counters for the byte accounting
    step(d_stats.data_ptr())
    torch.cuda.synchronize()
    stats = d_stats.cpu().numpy().view(_abi.STATS_DTYPE)[:mine]
    lens = np.diff(offsets.astype(np.int64))[:mine]
    model_bytes = survey_model_bytes(lens, stats)
    need = needed_bytes(lens, stats)
    need_kind = "counted by the statistics kernel"
    if need is None and mine:
        need, need_kind = needed_bytes_floor(lens, stats), "lower bound: table slots + set records only (this kernel does not count its index bytes)"
    ref_out = d_out.cpu().numpy().view(_abi.PLACEMENT_DTYPE).copy()

    for i in range(args.warmup):
        step(buf=i & 1)
        if world > 1:
            gather_records(i & 1)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    sync_all()
    db.kernel_time(reset=True)  # HIP-event accumulators around the dominant kernel (cls_db_kernel_time)
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        step(buf=i & 1)
        ev[i][1].record()
        if world > 1:
            gather_records(i & 1)
    sync_all()
    elapsed = time.perf_counter() - t0
    call_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps          # whole C-ABI call (all its kernels)
    k_sum, k_cnt = db.kernel_time(reset=True)                             # the placement kernel alone, same launches
    kernel_ms = k_sum / max(1, k_cnt)
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # the timed steps must have produced the same records as the checked run
    now = d_outs[(args.steps - 1) & 1].cpu().numpy().view(_abi.PLACEMENT_DTYPE)
    if not os.environ.get("CLS_PROFILE_STOP"):
        for f in ("status", "one", "rest", "levels", "clade_id"):
            assert (now[f] == ref_out[f]).all(), "non-deterministic placement records"
    slots = db.refresh_info().scratch_slots

    if rank == 0:
        total = total_reads * args.steps
        kname = db.kernel_name()
        traffic, stale = traffic_lookup(args.config, mine, kname)
        counts = np.bincount(ref_out["status"][:mine], minlength=12)[:12]  # (a CLS_PROFILE_STOP run writes junk statuses)
        ksec = kernel_ms * 1e-3 if kernel_ms > 0 else float("nan")
        achieved = (need / ksec / 1e9) if need else None
        metric = {"C3": "query placements/sec, 10k-leaf tree, 150 bp reads", "C4": "query placements/sec, 10k-leaf tree, 150 bp reads",
                  "C2": "query placements/sec, 1k-leaf tree, 150 bp reads",
                  "C5": "query placements/sec, 50k-leaf deep tree, 10 kb reads",
                  "C3s12": "query placements/sec, support-collapsed 10k-leaf tree, 150 bp reads",
                  "C3s35": "query placements/sec, support-collapsed 10k-leaf tree, 150 bp reads, k=35",
                  "G35": "query placements/sec, 300-leaf support-collapsed tree, 1.9 kb reads, k=35 (the reference's documented workload)"}[args.config]
        line = {
            "metric": metric,
            "value": total / elapsed,
            "unit": "placements/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo; not a measurement)" if rehearsal else ""),
            "window": "device-resident: reads in HBM -> records in HBM (cls_place_batch_device, every kernel of the call"
                      + (" + the gather to rank 0" if world > 1 else "") + "); the host-buffer window is `host_window`",
            "config": {
                "workload": f"{args.config}: {cfg['n_leaves']}-leaf {'deep (depth <= %d) ' % cfg['max_depth'] if cfg['deep'] else ''}"
                            f"{'support-collapsed ' if cfg.get('collapse_prob') else ''}tree, "
                            + (f"{total_reads} x {cfg['read_len']} bp reads in ONE stream, ceil(N/G) = {per_gpu} per GPU, " if strong
                               else f"{per_gpu} x {cfg['read_len']} bp reads per GPU, ")
                            + f"k={cfg['k_size']}, m={cfg['m_size']} (seeds tree=1 refseq=2 reads=3)",
                "reads_per_gpu": per_gpu,
                "parallelism": f"reads sharded x{world}, index replicated, 1 gather/step" if world > 1 else "single GPU",
                "index": {"n_nodes": int(db.info.n_nodes), "n_kmers": int(db.info.n_kmers), "n_tip_sets": int(db.info.n_tip_sets), "max_depth": int(db.info.max_depth),
                          "hbm_bytes": int(db.info.hbm_bytes), "input": "tips only" if cfg.get("tips_only") else "explicit node sets"},
                "status_counts": {_abi.STATUS_NAMES[i]: int(c) for i, c in enumerate(counts) if c},
                "mean_levels": float(ref_out["levels"][:mine].mean()) if mine else 0.0,
                "setup_s": {"generate": round(gen_s, 1), "db_create": round(create_s, 1),
                            "what": "slowest rank; N > 1: rank 0 generates index + read shards once, the others map them from shared memory"},
                "scratch_slots": int(slots),
            },
            "roofline": {
                "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                # bytes this index must touch (counted by the statistics kernel) / time of the dominant kernel
                "achieved": achieved, "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "needed_bytes_per_launch": need, "needed_bytes_per_read": (need / mine) if need and mine else None, "needed_bytes_kind": need_kind,
                # bytes the kernel really moved (rocprofv3 FETCH_SIZE + WRITE_SIZE of a profile of THESE sources), else null
                "traffic": traffic, "traffic_stale": stale,
                "traffic_frac": (traffic / ksec / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "overfetch": (traffic / need) if traffic and need else None,
                "kernel": kname, "kernel_ms": kernel_ms, "kernel_launches_timed": int(k_cnt),
                "kernels_timed": "kernel_ms = the placement kernel named in `kernel` only (HIP events on the launch stream: the "
                                 "wave-per-read kernel of the <= 320-k-mer class, or all launches of the LDS-tiled kernel when the handle is "
                                 "provisioned for reads beyond 8192 k-mers or the config's reads are gene-length, G35); call_ms = the whole "
                                 "call: locality keys + sort + classify + every class",
                "call_ms": call_ms,
                # SURVEY.md 8(d)'s model prices streamed posting lists; > 1 by construction for an index that never streams them
                "survey_model_bytes": model_bytes, "survey_model_frac": model_bytes / ksec / 1e9 / HBM_PEAK_GBS,
            },
        }
        # the kernel reads at random, one 64-byte line per L2 miss: its second roofline is the rate at which the memory
        # side serves such lines, measured on this chip by tools/gather_probe.hip (profiles/gather_probe.json)
        gref = gather_reference()
        if gref:
            peak_lines = gref["glines_per_s"]["1024_MiB_table_hbm"]
            lines = (traffic / 64.0 / ksec / 1e9) if traffic else None
            line["roofline"]["random_line_rate"] = {"achieved_glines_per_s": lines, "peak_glines_per_s": peak_lines,
                                                    "frac": (lines / peak_lines) if lines else None,
                                                    "what": gref["what"]}
        if args.dump_records:
            last = (args.steps - 1) & 1
            if world == 1:
                allrec = now.copy()
            elif rehearsal:
                allrec = np.concatenate([x.numpy() for x in host_gathered[0]]).view(_abi.PLACEMENT_DTYPE)
            else:
                allrec = np.concatenate([x.cpu().numpy() for x in gathered[last]]).view(_abi.PLACEMENT_DTYPE)
            if strong:  # drop the zero tail of a short last shard
                allrec = allrec[:total_reads]
            np.save(args.dump_records, allrec)
            line["config"]["dumped_records"] = int(len(allrec))
        if not args.no_host_window and world == 1 and mine:
            hw, host_recs = host_window(db, bases, offsets, mine)
            for f in ("status", "one", "rest", "levels", "clade_id"):
                assert (host_recs[f] == ref_out[f][:mine]).all(), "host-buffer entry disagrees with the device-buffer entry"
            line["host_window"] = hw
        if args.e2e and world == 1 and mine:
            line["e2e"] = e2e_window(db, synth, cfg, bases, mine)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            cb, oracle_recs = cpu_baseline(synth, cfg, first)
            n_chk = min(len(oracle_recs), mine)
            for f in ("status", "one", "rest", "levels", "clade_id"):
                bad = np.nonzero(oracle_recs[f][:n_chk] != ref_out[f][:n_chk])[0]
                if len(bad):
                    raise SystemExit(f"PARITY FAILURE: read {bad[0]} field {f}: GPU {ref_out[f][bad[0]]} oracle {oracle_recs[f][bad[0]]} "
                                     f"({len(bad)} of {n_chk} reads differ)")
            line["cpu_baseline"] = cb
            line["parity_checked_reads"] = int(n_chk)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
