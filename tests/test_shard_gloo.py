"""CPU, world_size 2, gloo: the N > 1 harness bench.py uses (stream -> rank assignment, barrier,
max-over-ranks timing, whole-job aggregation).  The step itself runs the CPU oracle here -- the GPU
data path has no inter-rank traffic to test (streams are independent)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from sdrangel_amd import shard


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_streams, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    from tests import oracle_py as orc
    from tests import synth
    dist = shard.init_process_group("gloo", rank, world)
    mine = shard.streams_of_rank(n_streams, rank, world)
    decs = {s: orc.Decim(6, 2, 12) for s in mine}
    data = {s: synth.mix(16384, 40 + s, 2047, 500) for s in mine}
    outs = {s: [] for s in mine}

    def step():
        for s in mine:
            outs[s].append(decs[s].process(data[s]))

    el = shard.timed_region(step, steps=3, warmup=1, sync=lambda: None, dist=dist)
    # every rank must hold the SAME max-reduced time
    t = torch.tensor([el], dtype=torch.float64)
    lst = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(lst, t)
    same = all(abs(float(v) - el) < 1e-12 for v in lst)
    digest = {s: int(np.concatenate(outs[s]).astype(np.int64).sum()) for s in mine}
    q.put((rank, mine, el, same, digest))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_ranks_gloo_streams_shard_without_exchange():
    world, n_streams = 2, 5
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_streams, q)) for r in range(world)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in ps: p.join(30)
    assert all(p.exitcode == 0 for p in ps)
    owned = sorted(s for r in res for s in r[1])
    assert owned == list(range(n_streams))                       # disjoint and complete
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    assert all(r[3] for r in res) and abs(res[0][2] - res[1][2]) < 1e-12
    # per-stream results do not depend on which rank ran them
    from tests import oracle_py as orc
    from tests import synth
    for r in res:
        for s, dig in r[4].items():
            d = orc.Decim(6, 2, 12); x = synth.mix(16384, 40 + s, 2047, 500)
            want = int(np.concatenate([d.process(x) for _ in range(4)]).astype(np.int64).sum())
            assert dig == want
    rate = shard.aggregate_rate(16384 * 3, world, 3, res[0][2])
    assert rate > 0


@pytest.mark.timeout(180)
def test_bench_launcher_spawns_n_ranks_and_rank0_prints_one_line():
    """`python bench.py --gpus 2` with no torchrun around it: bench.py starts the ranks itself (before any HIP call),
    they rendezvous (gloo here), and exactly one JSON line with n_gpus == 2 comes out."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SDRX_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "launchcheck"],
                       env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    # a rank count that contradicts the environment is refused, not silently ignored
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "launchcheck"], env=env2, capture_output=True, text=True, timeout=60)
    assert r2.returncode != 0 and "WORLD_SIZE" in (r2.stderr + r2.stdout)


def test_stream_assignment_edges():
    assert shard.streams_of_rank(8, 3, 8) == [3]
    assert shard.streams_of_rank(3, 5, 8) == []                   # fewer streams than GPUs: idle rank ("replicas only" is bench's business)
    with pytest.raises(ValueError):
        shard.streams_of_rank(4, 2, 2)
