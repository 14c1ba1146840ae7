"""CPU tests of the multi-rank Monte-Carlo harness logic (sharding, the one all-reduce, adaptive
sample counts) with world_size = 2 over gloo.  A deterministic stub backend stands in for the GPU
decode (which has no CPU form): its counters are a pure function of the global frame index, so the
totals must be identical for 1 and 2 ranks."""
import json
import os
import subprocess
import sys

import numpy as np

from channelcoding_amd import capi
from channelcoding_amd.montecarlo import awgn_simulation, samples, shannon_limit_ebno_db, shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StubCode:
    n, rate = 31, 16 / 31

    def to_string(self):
        return "(31, 16, 7)-STUB"


class StubBackend:
    """word error iff hash(global frame, ebno) falls under a threshold that shrinks with Eb/N0."""

    def run(self, ebno_db, seed, first_frame, frames):
        import torch
        idx = np.arange(first_frame, first_frame + frames, dtype=np.uint64)
        h = (idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)) >> np.uint64(40)
        thr = int((1 << 24) * 0.5 * 10 ** (-ebno_db / 4))
        err = h < np.uint64(thr)
        c = np.zeros(capi.MC_NCOUNTERS, np.int64)
        c[capi.MC_FRAMES] = frames
        c[capi.MC_WORD_ERRORS] = int(err.sum())
        c[capi.MC_BIT_ERRORS] = int((h[err] % np.uint64(7)).sum())
        return torch.from_numpy(c)


def test_helpers():
    assert samples(0.5) == 10000 and samples(1e-9) == 1000000 and samples(0.0) == 1000000  # simulation.c++:91-93
    assert samples(0.4806) == 10403  # second point of the (31,16,7) sweep
    cover = [shard(1000003, r, 8) for r in range(8)]
    assert cover[0][0] == 0 and sum(c for _, c in cover) == 1000003
    assert all(cover[i][0] + cover[i][1] == cover[i + 1][0] for i in range(7))
    assert abs(shannon_limit_ebno_db(0.5) - 0.187) < 0.01  # BPSK-constrained limit at R = 1/2
    assert abs(shannon_limit_ebno_db(231 / 255) - 3.3) < 0.1


def run_single():
    sim = awgn_simulation(StubCode(), backend=StubBackend(), max_samples=50000)
    return [(r["ebno"], r["frames"], r["word_errors"], r["bit_errors"]) for r in sim()]


def test_single_rank_ladder():
    res = run_single()
    assert res[0][0] == 1.0 and res[0][1] == 10000  # wer = 0.5 seeds 10000 samples
    for (e0, f0, w0, _), (e1, f1, _, _) in zip(res, res[1:]):
        assert abs(e1 - e0 - 0.5) < 1e-12
        assert f1 == min(50000, samples(w0 / f0))  # adaptive count uses the REDUCED wer of the previous point


WORKER = r"""
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch.distributed as dist
from channelcoding_amd.montecarlo import awgn_simulation
from test_montecarlo_dist import StubBackend, StubCode
dist.init_process_group("gloo")
sim = awgn_simulation(StubCode(), backend=StubBackend(), max_samples=50000)
res = [(r["ebno"], r["frames"], r["word_errors"], r["bit_errors"]) for r in sim()]
with open(os.path.join(%(out)r, "rank%%d.json" %% dist.get_rank()), "w") as f:  # one file per rank: shared stdout interleaves
    json.dump(res, f)
dist.barrier(); dist.destroy_process_group()
"""


def test_two_ranks_equal_one_rank(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = {}
    for rank in (0, 1):
        with open(tmp_path / ("rank%d.json" % rank)) as f:
            got[rank] = json.load(f)
    single = [list(x) for x in run_single()]
    assert got[0] == single and got[1] == single  # every rank holds the same reduced totals
