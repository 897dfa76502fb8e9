"""N > 1 path on the CPU: world_size-2 gloo processes, voices sharded across ranks, one sum-reduce of the bus.
Each rank renders its shard with the CPU harness (the engine's kernel code built for the host -- test
infrastructure; on the GPU box the same libzl_amd.sharding code drives the HIP engine over RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(world):
    from scenario import random_scene
    # 2 buses x 8 voice slots globally; every clip is started on an explicit (bus, slot) so the shard is unambiguous
    sc = random_scene(4242, num_buses=2, voices_per_bus=8, nclips=12, nframes=128, nblocks=10, events=False)
    ev = []
    for i, e in enumerate(sc.events[0]):
        bus, slot = i % 2, i // 2
        f = dict(e[1]); f["midiChannel"] = bus - 2
        ev.append(("start", bus, slot, f, 0))
    sc.events[0] = ev
    return sc


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_harness.sim import SimSynth
    from libzl_amd import sharding
    from scenario import run_backend
    sc = _scene(world)
    lo, hi = sharding.slots_for_rank(sc.voices_per_bus, world, rank)
    # this rank keeps only the voice slots [lo, hi) of every bus, renumbered from 0
    mine = [("start", b, s - lo, f, t) for (_, b, s, f, t) in sc.events[0] if lo <= s < hi]
    sc.events[0] = mine
    sc.voices_per_bus = hi - lo
    bus, rep, syn, _ = run_backend(sc, SimSynth, batch=5)
    t = torch.from_numpy(np.ascontiguousarray(bus))
    sharding.reduce_bus(t, dst=0)
    if rank == 0:
        q.put(t.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_sharded_voices_reduce_to_the_full_mix(built, world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenario import run_oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = _scene(world)
    seq, _, _ = run_oracle(sc)                                     # reference order: all 8 slots of a bus in sequence
    assert np.abs(seq - got).max() <= 1e-6 * max(1.0, float(np.abs(seq).max()))
    sc.mix_group = sc.voices_per_bus // world                      # the sharded order: per-rank partial sums, then a + b
    grouped, _, _ = run_oracle(sc)
    assert np.array_equal(grouped.view(np.int32), got.view(np.int32))


def test_partition_helpers():
    from libzl_amd import sharding
    assert [sharding.voice_range(1024, 8, r) for r in (0, 7)] == [(0, 128), (896, 1024)]
    assert sum(b - a for a, b in (sharding.voice_range(1000, 3, r) for r in range(3))) == 1000
    assert [sharding.bus_owner(b, 12, 4) for b in range(12)] == [0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3]
