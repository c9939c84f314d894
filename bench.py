#!/usr/bin/env python3
"""Headline benchmark: query placements/sec on the 10k-leaf tree, 150 bp reads.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the placement hot path (`cls_place_batch_device`, the
C-ABI entry a Rust caller would bind) over one batch of synthetic reads that is
already resident in HBM.  At N=1 the workload is BASELINE.json configs[2]
("C3": 10k-leaf tree, 1M x 150 bp reads, k=12).  At N>1 every rank holds the
same index and its own shard of ONE global read stream (1M reads per GPU ->
"weak" scaling; configs[3] is this shape), and each step ends with the single
gather of the 24-byte placement records to rank 0 over RCCL.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for how `roofline` and
`cpu_baseline` are obtained.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(read_lens, stats):
    """B(q) = L + 16*2(L-k+1) + 4*sum_{h in M(q)} |leaves(h)| + 24."""
    return int(read_lens.sum()) + 16 * int(stats["n_query_kmers"].astype(np.int64).sum()) + 4 * int(
        stats["leaf_postings"].astype(np.int64).sum()) + 24 * len(stats)


def traffic_lookup(config, reads):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed under
    profiles/ (FETCH_SIZE / WRITE_SIZE collected in separate --pmc runs of this same command and
    corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes); null if this workload was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(f"{config}:{reads}", {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def db_kernel_name(db):
    return "place_fast_kernel<5,9,...>" if db.info.direct_table else "place_split_kernel<5,9,...>"


def cpu_baseline(synth, cfg, budget_s=15.0):
    """The C oracle (oracle/cls_oracle.c, kind "port") on this host's cores, on
    a bounded prefix of the same read stream."""
    from oracle.oracle_port import OraclePort

    cores = min(os.cpu_count() or 1, 64)
    t0 = time.time()
    port = OraclePort(synth.flat)
    build_s = time.time() - t0
    probe = 2000
    bases, offsets, _ = synth.reads(probe, cfg["read_len"])
    t0 = time.time()
    port.place_batch(bases, offsets, threads=cores)
    dt = max(time.time() - t0, 1e-4)
    n = int(min(cfg["n_reads"], max(probe, probe * budget_s / dt)))
    bases, offsets, _ = synth.reads(n, cfg["read_len"])
    t0 = time.time()
    port.place_batch(bases, offsets, threads=cores)
    dt = time.time() - t0
    # what the unmodified Rust path would cost: the port + the reference's per-query deep clone of the index
    # (place_sequence.rs:77-80) and per-bucket key-set rebuild (kmers_map.rs:58-62); an estimate, on a small sample
    port.set_reference_cost(True)
    rc_threads = min(cores, 16)  # (3 M allocations per query: more threads only fight over the allocator)
    m = 2 * rc_threads
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
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", choices=["C2", "C3"])
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from classeq2_amd import _abi, engine
    from classeq2_amd.synth import CONFIGS, SynthDb

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
    per_gpu = args.reads or cfg["n_reads"]
    cfg["n_reads"] = per_gpu
    threads = max(1, (os.cpu_count() or 8) // max(1, min(world, 8)))
    t0 = time.time()
    synth = SynthDb(cfg["n_leaves"], cfg["ref_len"], cfg["k_size"], cfg["m_size"], deep=cfg["deep"],
                    max_depth=cfg["max_depth"], threads=threads)
    gen_s = time.time() - t0
    t0 = time.time()
    db = engine.PlacementDb(synth.flat, device=local_rank)
    create_s = time.time() - t0
    # this rank's shard of the global read stream (seed 3)
    bases, offsets, _ = synth.reads(per_gpu, cfg["read_len"], seed=3, first=rank * per_gpu)
    d_bases = torch.from_numpy(bases).to(dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_outs = [torch.zeros(per_gpu * 24, dtype=torch.uint8, device=dev) for _ in range(2)]  # double-buffered records
    d_out = d_outs[0]
    d_stats = torch.zeros(per_gpu * 24, dtype=torch.uint8, device=dev)
    gathered = [[torch.empty_like(d_out) for _ in range(world)] for _ in range(2)] if (world > 1 and rank == 0) else [None, None]
    stream = torch.cuda.current_stream().cuda_stream
    pending = [None, None]  # the gather still reading each buffer

    def step(stats_ptr=0, buf=0):
        if pending[buf] is not None:  # the records of two steps ago must have left before they are overwritten
            pending[buf].wait()
            pending[buf] = None
        db.place_batch_device(d_bases.data_ptr(), d_off.data_ptr(), per_gpu, d_outs[buf].data_ptr(), None, stats_ptr, stream)

    def gather_records(buf=0):
        """The path's one collective: every rank's placement records -> rank 0 (RCCL over xGMI), asynchronous:
        the gather of step i runs while step i+1 places into the other buffer."""
        if rehearsal:
            h = d_outs[buf].cpu()
            dist.gather(h, [torch.empty_like(h) for _ in range(world)] if rank == 0 else None, dst=0)
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

    # untimed: per-query counters for the algorithmic-byte figure
    step(d_stats.data_ptr())
    torch.cuda.synchronize()
    stats = d_stats.cpu().numpy().view(_abi.STATS_DTYPE)
    alg_bytes = algorithmic_bytes(np.diff(offsets.astype(np.int64)), stats)
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

    if rank == 0:
        total = per_gpu * world * args.steps
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = traffic_lookup(args.config, per_gpu)
        counts = np.bincount(ref_out["status"], minlength=12)[:12]  # (a CLS_PROFILE_STOP run writes junk statuses)
        line = {
            "metric": "query placements/sec, 10k-leaf tree, 150 bp reads" if args.config == "C3"
            else "query placements/sec, 1k-leaf tree, 150 bp reads",
            "value": total / elapsed,
            "unit": "placements/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo; not a measurement)" if rehearsal else ""),
            "config": {
                "workload": f"{args.config}: {cfg['n_leaves']}-leaf Yule tree, {per_gpu} x {cfg['read_len']} bp reads per GPU, "
                            f"k={cfg['k_size']}, m={cfg['m_size']} (seeds tree=1 refseq=2 reads=3)",
                "reads_per_gpu": per_gpu,
                "parallelism": f"reads sharded x{world}, index replicated, 1 gather/step" if world > 1 else "single GPU",
                "index": {"n_nodes": int(db.info.n_nodes), "n_kmers": int(db.info.n_kmers), "n_tip_sets": int(db.info.n_tip_sets), "max_depth": int(db.info.max_depth),
                          "hbm_bytes": int(db.info.hbm_bytes)},
                "status_counts": {_abi.STATUS_NAMES[i]: int(c) for i, c in enumerate(counts) if c},
                "mean_levels": float(ref_out["levels"].mean()),
                "setup_s": {"generate": round(gen_s, 1), "db_create": round(create_s, 1)},
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                # the index answers set membership without streaming the posting lists the algorithmic figure
                # prices, so `frac` exceeds 1; the bytes the kernel really moved, against the same peak:
                "traffic_frac": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "kernel": db_kernel_name(db), "kernel_ms": kernel_ms, "kernel_launches_timed": int(k_cnt),
                "call_ms": call_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_bytes_per_read": alg_bytes / per_gpu,
            },
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(synth, cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
