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
    t_det, t_mesh, t_fused = t.clone(), t.clone(), t.clone()       # (t shares the harness's buffer: clone before the in-place reduce)
    sharding.reduce_bus(t, dst=0)
    sharding.reduce_bus_in_rank_order(t_det, dst=0)
    sharding.reduce_bus_mesh(t_mesh, dst=0)
    # the exchange behind the C-ABI: all-to-all, zlhip_bus_reduce_sum_scan on every rank's pieces, gather of pieces + unit levels
    sharding.exchange_bus_mesh(syn, t_fused, sc.nblocks, sc.nframes, dst=0)
    fused_peaks, fused_sumsq = (getattr(syn, "scanned_peaks", None), getattr(syn, "scanned_sumsq", None)) if rank == 0 else (None, None)
    syn.scanned_peaks = None
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
        q.put((t.numpy().copy(), outs[1].numpy().copy(), outs[2].numpy().copy(), getattr(syn, "scanned_peaks", None), t_det.numpy().copy(), t_mesh.numpy().copy(),
               t_fused.numpy().copy(), fused_peaks, fused_sumsq))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_voices_reduce_to_the_full_mix(built, world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenario import run_oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, tail1, tail2, peaks, got_det, got_mesh, got_fused, fused_peaks, fused_sumsq = q.get(timeout=180)
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
    # the fused exchange: the same bits, and the levels of every (block, bus, channel) as the oracle defines them
    assert np.array_equal(grouped.view(np.int32), got_fused.view(np.int32))
    from oracle import zl_oracle as zo
    lib = zo.load()
    N0 = sc.nframes
    rows = grouped.reshape(sc.num_buses, 2, sc.nblocks, N0)
    assert np.array_equal(fused_peaks, np.abs(np.float32(131072.0) * rows).astype(np.int64).max(axis=3).transpose(2, 0, 1))
    for k in range(sc.nblocks):
        for b in range(sc.num_buses):
            for c in range(2):
                row = np.ascontiguousarray(rows[b, c, k])
                assert fused_sumsq[k, b, c] == lib.zlo_block_sumsq(row.ctypes.data, N0, 1), (k, b, c)
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


def _aligned_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every collective entry point is counted: the bus-aligned path must not call any
    calls = []
    for name in ("all_reduce", "reduce", "gather", "all_gather", "all_to_all_single", "all_to_all", "broadcast", "reduce_scatter", "scatter", "send", "recv"):
        if hasattr(dist, name):
            orig = getattr(dist, name)
            setattr(dist, name, (lambda *a, _n=name, _o=orig, **k: (calls.append(_n), _o(*a, **k))[1]))
    from cpu_harness.sim import SimSynth
    from libzl_amd import sharding
    from scenario import engine_cmd, run_backend
    sc = _aligned_scene()
    part = sharding.BusPartition(sc.num_buses, world, rank)
    # this rank's engine holds only its buses; commands addressed by global midi channel are routed by the partition
    mine = []
    for ev in sc.events[0]:
        local = part.local_command(engine_cmd(**ev[1]))
        if local is not None:
            f = dict(ev[1]); f["midiChannel"] = local.midi_channel
            mine.append(("cmd", f, ev[2]))
    sc.events[0] = mine
    sc.num_buses = part.num_local_buses
    bus, rep, syn, _ = run_backend(sc, SimSynth, batch=8)         # one batch: block_peaks() covers every block
    peaks = syn.block_peaks()
    q.put((rank, part.buses, bus.copy(), peaks.copy(), list(calls)))
    # (the test's own barrier, outside the data path)
    calls.clear()
    dist.barrier()
    dist.destroy_process_group()


def _aligned_scene():
    from scenario import random_scene
    return random_scene(777, num_buses=6, voices_per_bus=4, nclips=14, nframes=128, nblocks=8, events=False)


def test_bus_aligned_partition_needs_no_collective(built):
    """num_buses >= world_size: whole buses per rank.  Two ranks render their buses of one 6-bus scene; together they
    are the oracle's full mix bit for bit, levels included, and the data path made ZERO collective calls."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenario import run_oracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_aligned_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = _aligned_scene()
    full, _, _ = run_oracle(sc)
    seen = []
    for rank, buses, bus, peaks, calls in res:
        assert calls == [], f"rank {rank} called collectives on the data path: {calls}"
        seen += buses
        assert np.array_equal(bus.view(np.int32), full[buses].view(np.int32))
        exp = np.abs(np.float32(131072.0) * full[buses].reshape(len(buses), 2, sc.nblocks, sc.nframes)).astype(np.int64).max(axis=3).transpose(2, 0, 1)
        assert np.array_equal(peaks, exp)
    assert sorted(seen) == list(range(6))


def _fallback_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_harness.sim import SimSynth
    from libzl_amd import sharding
    from libzl_amd.engine import synthetic_clocks
    from scenario import run_backend
    sc = _scene(world)
    lo, hi = sharding.slots_for_rank(sc.voices_per_bus, world, rank)
    sc.events[0] = [("start", b, s - lo, f, t) for (_, b, s, f, t) in sc.events[0] if lo <= s < hi]
    sc.voices_per_bus = hi - lo
    _, _, syn, _ = run_backend(sc, SimSynth, batch=5)
    # the collective of the mesh exchange is refused from its 2nd call on, on every rank alike and before it moves anything --
    # what an RCCL build without that collective does
    orig, calls = dist.all_to_all_single, [0]

    def refusing(*a, **k):
        calls[0] += 1
        if calls[0] >= 2:
            raise RuntimeError("all_to_all_single: not supported by this build (injected)")
        return orig(*a, **k)
    dist.all_to_all_single = refusing
    renders = [0]
    render_batch = syn.render_batch
    syn.render_batch = lambda *a, **k: (renders.__setitem__(0, renders[0] + 1), render_batch(*a, **k))[1]
    ov = sharding.OverlappedBusReduce(syn, lambda: torch.zeros((sc.num_buses, 2, 4 * sc.nframes), dtype=torch.float32), dst=0, algorithm="mesh")
    logged, outs, algos, first = [], [], [], None
    for i in range(3):
        b, ov = ov.step_or_fall_back(4, sc.nframes, synthetic_clocks(4, sc.nframes, sc.fs, start_block=sc.nblocks + 4 * i, bpm=sc.bpm), log=logged.append)
        outs.append(b)
        algos.append(ov.algorithm)
        if i == 1:
            first = outs[0].numpy().copy()             # batch 0 (through the mesh exchange) before batch 2 reuses its buffer
    ov.flush()
    second, third = outs[1].numpy().copy(), outs[2].numpy().copy()
    peaks = getattr(syn, "scanned_peaks", None)
    # a reduce that is refused has nothing to fall back to: the error reaches the caller
    dist.reduce, orig_reduce = (lambda *a, **k: (_ for _ in ()).throw(RuntimeError("reduce refused (injected)"))), dist.reduce
    try:
        ov.step_or_fall_back(4, sc.nframes, synthetic_clocks(4, sc.nframes, sc.fs, start_block=sc.nblocks + 12, bpm=sc.bpm))
        raised = False
    except RuntimeError:
        raised = True
    dist.reduce = orig_reduce
    if rank == 0:
        q.put((first, second, third, peaks, algos, logged, renders[0], raised))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_exchange_falls_back_to_the_plain_reduce(built):
    """bench.py's warm-up safety net (OverlappedBusReduce.step_or_fall_back): the mesh exchange is refused on its second batch; the
    batch already rendered goes through a plain reduce over the same buffers, nothing is rendered twice, and all three batches are
    the oracle's grouped mix bit for bit (two ranks: a + b has one order), with the levels scanned on the root."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from scenario import run_oracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_fallback_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    b0, b1, b2, peaks, algos, logged, renders, raised = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert algos == ["mesh", "reduce", "reduce"]
    assert len(logged) == 1 and "falling back to dist.reduce" in logged[0] and "injected" in logged[0]
    assert renders == 4                                              # 3 batches + the one whose refused reduce was raised: none rendered twice
    assert raised
    sc = _scene(world)
    sc.mix_group = sc.voices_per_bus // world
    sc.nblocks = 22
    longer, _, _ = run_oracle(sc)
    N = sc.nframes
    for want, have in ((longer[:, :, 10 * N:14 * N], b0), (longer[:, :, 14 * N:18 * N], b1), (longer[:, :, 18 * N:22 * N], b2)):
        assert np.array_equal(want.view(np.int32), have.view(np.int32))
    exp = np.abs(np.float32(131072.0) * b2.reshape(sc.num_buses, 2, 4, N)).astype(np.int64).max(axis=3).transpose(2, 0, 1)
    assert peaks is not None and np.array_equal(peaks, exp)          # the plain reduce's levels: scanned on the reduced bus, on the root
