"""bench.py's N > 1 code path on a one-GPU box: `CLS_BENCH_REHEARSAL=1` puts every rank on GPU 0 over gloo; run in a
FRESH child process (torch.distributed.run), rank 0's gathered records are dumped and checked against the oracle.
Covers the weak-scaling default (disjoint slices of one stream, 1 gather per step) and `--config C4` (ONE stream of
fixed size, ceil(N/G) reads per rank: BASELINE configs[3]'s shape, strong scaling) incl. a short last shard.
Never a measurement."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from classeq2_amd import _abi
from classeq2_amd.synth import CONFIGS, SynthDb
from oracle import oracle_port as op
from tests.helpers import records_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(tmp_path, world, extra):
    dump = str(tmp_path / "records.npy")
    env = dict(os.environ, CLS_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--dump-records", dump] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    return line, np.load(dump)


@pytest.mark.parametrize("config,reads,total", [("C3", 20000, 40000), ("C4", 30001, 30001)])
def test_two_ranks_gathered_records_match_the_oracle(tmp_path, config, reads, total):
    line, recs = _run_bench(tmp_path, 2, ["--config", config, "--reads", str(reads)])
    assert line["n_gpus"] == 2 and "REHEARSAL" in line["data"]
    assert line["scaling"] == ("strong" if config == "C4" else "weak")
    assert len(recs) == total == line["config"]["dumped_records"]
    cfg = CONFIGS[config]
    s = SynthDb(cfg["n_leaves"], cfg["ref_len"], cfg["k_size"], cfg["m_size"])
    bases, offsets, _ = s.reads(total, cfg["read_len"], seed=3, first=0)  # rank r holds reads [r * per, (r + 1) * per) of this stream
    want = op.OraclePort(s.flat).place_batch(bases, offsets, threads=16)
    bad = records_equal(recs.view(_abi.PLACEMENT_DTYPE), want)
    assert len(bad) == 0, f"{len(bad)} gathered records differ from the oracle, first {bad[0]}"
