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
    t_det, t_mesh = t.clone(), t.clone()
    sharding.reduce_bus(t, dst=0)
    sharding.reduce_bus_in_rank_order(t_det, dst=0)
    sharding.reduce_bus_mesh(t_mesh, dst=0)
    # the double-buffered, overlapped variant bench.py uses at N > 1: three more batches of the same voices
    from libzl_amd.engine import synthetic_clocks
    ov = sharding.OverlappedBusReduce(syn, lambda: torch.zeros((sc.num_buses, 2, 4 * sc.nframes), dtype=torch.float32), dst=0,
                                      algorithm="mesh")
    outs = []
    for i in range(3):
        b = ov.step(4, sc.nframes, synthetic_clocks(4, sc.nframes, sc.fs, start_block=sc.nblocks + 4 * i, bpm=sc.bpm))
        outs.append(b)
    ov.flush()
    tail = torch.cat([outs[0], outs[1], outs[2]], dim=2) if rank == 0 else None      # buffers 0 and 1 alternate: outs[2] is outs[0]
    if rank == 0:
        q.put((t.numpy().copy(), outs[1].numpy().copy(), outs[2].numpy().copy(), getattr(syn, "scanned_peaks", None), t_det.numpy().copy(), t_mesh.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_voices_reduce_to_the_full_mix(built, world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenario import run_oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, tail1, tail2, peaks, got_det, got_mesh = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = _scene(world)
    seq, _, _ = run_oracle(sc)                                     # reference order: all 8 slots of a bus in sequence
    assert np.abs(seq - got).max() <= 1e-6 * max(1.0, float(np.abs(seq).max()))
    sc.mix_group = sc.voices_per_bus // world                      # the sharded order: per-rank partial sums, then their sum
    grouped, _, _ = run_oracle(sc)
    # gather + sum in rank order: the oracle's grouped order bit for bit, for any number of ranks
    assert np.array_equal(grouped.view(np.int32), got_det.view(np.int32))
    assert np.array_equal(grouped.view(np.int32), got_mesh.view(np.int32))     # all-to-all + rank-order sum + gather
    if world == 2:
        assert np.array_equal(grouped.view(np.int32), got.view(np.int32))      # a + b has one order
    else:
        assert np.abs(grouped - got).max() <= 1e-6 * max(1.0, float(np.abs(grouped).max()))   # the backend's reduce order
    # overlapped path: blocks 10..21 of the same scene (batches 2 and 3 are what the two buffers hold at the end)
    sc.nblocks = 22
    longer, _, _ = run_oracle(sc)
    N = sc.nframes
    for want, have in ((longer[:, :, 14 * N:18 * N], tail1), (longer[:, :, 18 * N:22 * N], tail2)):
        assert np.array_equal(want.view(np.int32), have.view(np.int32))          # the mesh reduce sums in rank order
    exp = np.abs(np.float32(131072.0) * tail2.reshape(sc.num_buses, 2, 4, N)).astype(np.int64).max(axis=3).transpose(2, 0, 1)
    assert peaks is not None and np.array_equal(peaks, exp)          # levels were scanned on the reduced bus, on the root


def test_partition_helpers():
    from libzl_amd import sharding
    assert [sharding.voice_range(1024, 8, r) for r in (0, 7)] == [(0, 128), (896, 1024)]
    assert sum(b - a for a, b in (sharding.voice_range(1000, 3, r) for r in range(3))) == 1000
    assert [sharding.bus_owner(b, 12, 4) for b in range(12)] == [0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3]
