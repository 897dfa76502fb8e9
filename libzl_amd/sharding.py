"""Multi-GPU sharding of the sampler hot path: one process per GPU.

Voices are independent given the shared clock and clip parameters (SamplerSynthVoice::process has no
cross-voice state); the only coupling is the per-bus sum of SamplerChannel::process
(reference lib/SamplerSynth.cpp:134-140).  Two partitions follow from that:

* **bus-aligned** (`BusPartition`, used whenever the job has at least as many buses as GPUs): every GPU owns whole buses --
  their voices, their sources, their mix and their meters.  No bus spans GPUs, so NO data-path collective runs at all.
* **a bus spans GPUs** (`exchange_bus_mesh` / `OverlappedBusReduce`): every rank renders the voices it owns into a partial bus
  [num_buses, 2, frames]; the partial buses are exchanged piecewise over the point-to-point xGMI mesh (all-to-all), every rank
  sums the pieces it received in rank order AND scans them for AudioLevels with one HIP kernel behind the C-ABI
  (`zlhip_bus_reduce_sum_scan`), and the reduced pieces and their levels are gathered on the root.

`torch.distributed` with backend "nccl" is RCCL over xGMI on ROCm; the same code runs on "gloo" for
the CPU tests.  The reduce's operand order is the backend's (ring / tree), so for more than two ranks
the root's sum may differ from the reference's voice-order sum in the last bits (within 1e-6 of the
mix magnitude); with two ranks a + b is order-independent and the result is bit-exact.
"""
from __future__ import annotations

from typing import Tuple


def voice_range(num_voices: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Block partition: rank g owns voices [g*V/G, (g+1)*V/G) (sources never move between GPUs)."""
    return (num_voices * rank) // world_size, (num_voices * (rank + 1)) // world_size


def bus_owner(bus: int, num_buses: int, world_size: int) -> int:
    """Bus-aligned partition for the case num_buses >= world_size: whole buses per GPU, no collective."""
    return (bus * world_size) // num_buses


class BusPartition:
    """Bus-aligned partition: global bus g lives on rank bus_owner(g); a rank's engine holds its buses as local buses
    0..n-1 in global order.  The per-bus sum -- the only coupling of the path -- never crosses a GPU: rendering, mixing and
    metering of a rank's buses need no collective.  (Global midi channel of local bus b: first + b - 2, SamplerSynth.cpp:270.)"""

    def __init__(self, num_buses_global: int, world_size: int, rank: int):
        if num_buses_global < world_size:
            raise ValueError("bus-aligned partition needs at least one bus per rank; let the bus span ranks instead (exchange_bus_mesh)")
        self.num_buses_global, self.world_size, self.rank = num_buses_global, world_size, rank
        self.buses = [g for g in range(num_buses_global) if bus_owner(g, num_buses_global, world_size) == rank]
        self.first = self.buses[0]

    @property
    def num_local_buses(self) -> int:
        return len(self.buses)

    def local_bus(self, global_bus: int) -> int:
        """Local index of a global bus on this rank, or -1 if another rank owns it."""
        return global_bus - self.first if self.first <= global_bus < self.first + len(self.buses) else -1

    def local_command(self, cmd):
        """A ClipCommand addressed by GLOBAL midi channel (bus g has channel g - 2) -> the same command for this rank's engine,
        or None when the channel's bus lives on another rank (the command is simply not this rank's)."""
        b = self.local_bus(cmd.midi_channel + 2)
        if b < 0:
            return None
        import copy
        c = copy.copy(cmd)
        c.midi_channel = b - 2
        return c


def slots_for_rank(voices_per_bus_global: int, world_size: int, rank: int) -> Tuple[int, int]:
    """When a bus spans ranks: the slice of each bus's voice slots that lives on `rank`."""
    return voice_range(voices_per_bus_global, world_size, rank)


def reduce_bus(bus, dst: int = 0, group=None):
    """Sum the per-rank partial buses onto `dst` in place (one collective per batch).  `bus` is a torch tensor
    [num_buses, 2, frames] living where the backend expects it (HBM for nccl/RCCL, host for gloo)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(bus, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return bus


def reduce_bus_in_rank_order(bus, dst: int = 0, group=None):
    """Deterministic alternative: the partial buses are gathered onto `dst` and summed there in rank
    order, ((p0 + p1) + p2) + ..., so the result is the same bits on every run and for every world size -- it equals
    the oracle's two-level order with one mix group per rank.  Costs world_size x the bus in memory on `dst` and the
    root's inbound links carry every partial bus; the default `reduce_bus` lets RCCL choose the order."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bus
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = [torch.empty_like(bus) for _ in range(world)] if rank == dst else None
    dist.gather(bus, gather_list=parts, dst=dst, group=group)
    if rank == dst:
        acc = parts[0]
        for p in parts[1:]:
            acc = acc + p                                  # elementwise fp32 adds in rank order
        bus.copy_(acc)
    return bus


def reduce_bus_mesh(bus, dst: int = 0, group=None, scratch=None):
    """Sum of the partial buses on `dst` over the point-to-point mesh, in rank order.  xGMI links every pair of GPUs
    of a node directly, so instead of a ring (every link carries the whole bus) each rank sends 1/world of its bus to
    each peer (all-to-all: world - 1 links in parallel, 1/world of the bytes each), sums the world pieces it received
    in rank order -- deterministic, the oracle's grouped order bit for bit -- and the reduced pieces are gathered onto
    `dst` (again one piece per link).  Per link: 2/world of the bus instead of the whole of it.
    `scratch`: optional receive buffer of the bus's size (kept by the caller to avoid an allocation per batch)."""
    import os
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return bus
    if dist.get_world_size(group) == 1 and not os.environ.get("ZL_FORCE_COLLECTIVES"):   # (set to rehearse the calls with one rank)
        return bus
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    flat = bus.view(-1)
    if flat.numel() % world != 0:
        return reduce_bus_in_rank_order(bus, dst=dst, group=group)
    chunk = flat.numel() // world
    recv = scratch.view(-1) if scratch is not None else torch.empty_like(flat)
    dist.all_to_all_single(recv, flat, group=group)               # recv[r] = piece `rank` of rank r's partial bus
    parts = recv.view(world, chunk)
    acc = parts[0].clone()
    for r in range(1, world):
        acc += parts[r]                                            # rank order
    pieces = [flat[r * chunk:(r + 1) * chunk] for r in range(world)] if rank == dst else None
    dist.gather(acc, gather_list=pieces, dst=dst, group=group)    # piece r of the final bus comes from rank r
    return bus


def exchange_bus_mesh(synth, bus, nblocks: int, nframes: int, dst: int = 0, group=None, scratch=None, stream=None):
    """A bus that spans ranks, over the point-to-point mesh, with the sum and the meters in ONE kernel behind the C-ABI.
    `bus` [num_buses, 2, nblocks*nframes] is this rank's partial bus.  The flat bus is world equal pieces of whole units (unit =
    the nframes of one (bus, channel, block)):
      1. all-to-all: rank r receives piece r of every rank's partial bus (world - 1 links in parallel, 1/world of the bytes each);
      2. synth.bus_reduce_sum_scan: ((0 + p0) + p1) + ... per sample in rank order + the AudioLevels scan of every unit;
      3. gather of the reduced pieces (into `bus` on dst) and of their unit levels; dst imports the levels into its engine.
    Falls back to reduce_bus_mesh + a scan on the root when the units do not divide evenly among the ranks.
    scratch: optional dict the caller keeps between batches (receive / result buffers).  stream: the stream the kernels run on
    (an int handle) when it is not torch's current stream."""
    import os
    import torch
    import torch.distributed as dist
    single = not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1
    if single and not os.environ.get("ZL_FORCE_COLLECTIVES"):
        synth.levels_scan_device(bus.data_ptr(), nblocks, nframes, stream=stream)
        return bus
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = bus.shape[0]
    units_total = B * 2 * nblocks
    if units_total % world != 0:
        reduce_bus_mesh(bus, dst=dst, group=group)
        if rank == dst:
            synth.levels_scan_device(bus.data_ptr(), nblocks, nframes, stream=stream)
        return bus
    units = units_total // world
    chunk = units * nframes
    flat = bus.view(-1)
    sc = scratch if scratch is not None else {}
    key = (flat.numel(), flat.device, world)
    if sc.get("key") != key:
        sc.clear()
        sc["key"] = key
        sc["recv"] = torch.empty_like(flat)
        sc["acc"] = torch.empty(chunk, dtype=torch.float32, device=flat.device)
        sc["lv"] = torch.empty((units, 2), dtype=torch.int32, device=flat.device)          # zlhip_unit_levels {int32 peak; float sumsq}
        sc["lv_all"] = torch.empty((world, units, 2), dtype=torch.int32, device=flat.device) if rank == dst else None
    recv, acc, lv = sc["recv"], sc["acc"], sc["lv"]
    dist.all_to_all_single(recv, flat, group=group)               # recv[r] = piece `rank` of rank r's partial bus
    if flat.is_cuda and stream is None:
        stream = torch.cuda.current_stream(flat.device).cuda_stream   # the kernel follows the collective on its stream
    synth.bus_reduce_sum_scan(recv.data_ptr(), world, chunk, units, nframes, acc.data_ptr(), lv.data_ptr(), stream=stream)
    pieces = [flat[r * chunk:(r + 1) * chunk] for r in range(world)] if rank == dst else None
    dist.gather(acc, gather_list=pieces, dst=dst, group=group)    # piece r of the final bus comes from rank r
    lvs = [sc["lv_all"][r] for r in range(world)] if rank == dst else None
    dist.gather(lv, gather_list=lvs, dst=dst, group=group)
    if rank == dst:
        synth.levels_import_units(sc["lv_all"].data_ptr(), nblocks, nframes, stream=stream)
    return bus


def render_sharded(synth, nblocks: int, nframes: int, clocks, bus, dst: int = 0, stream=None, group=None):
    """One batch on this rank's voices into `bus` (device pointer of a torch tensor), then the bus reduce;
    on the root the reduced bus is scanned for AudioLevels.  `synth` is a libzl_amd.SamplerSynth."""
    import torch.distributed as dist
    synth.render_batch(nblocks, nframes, clocks, bus_out_dev=bus.data_ptr(), stream=stream)
    _order_collective_behind_render(synth, bus, stream)
    reduce_bus(bus, dst=dst, group=group)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_rank(group) == dst:
        synth.levels_scan_device(bus.data_ptr(), nblocks, nframes, stream=stream)
    return bus


def _order_collective_behind_render(synth, bus, stream):
    """The collective is ordered behind torch's CURRENT stream.  A render queued with stream=None runs on the engine's own
    non-blocking stream, which torch knows nothing about: wait for it on the host.  With an explicit stream the caller
    makes that stream current (bench.py) or the classes below order their communication stream behind it."""
    if stream is None and getattr(bus, "is_cuda", False):
        synth.synchronize()


class OverlappedBusReduce:
    """Double-buffered partial buses: the reduce of batch i runs while batch i+1 renders into the other buffer (a
    100+ MB bus takes about as long to cross xGMI as to render; hiding it keeps the weak-scaling curve flat).
    algorithm: "mesh" (reduce_bus_mesh: all-to-all + rank-order sum + gather, on a communication stream of its own),
    "reduce" (one RCCL reduce, async) or "rank-order" (reduce_bus_in_rank_order, on the communication stream)."""

    def __init__(self, synth, make_bus, dst: int = 0, group=None, algorithm: str = "mesh"):
        self.synth, self.dst, self.group, self.algorithm = synth, dst, group, algorithm
        self.bus = [make_bus(), make_bus()]
        self.cuda = self.bus[0].is_cuda
        self.work = [None, None]
        self.shape = [None, None]
        self.i = 0
        if algorithm != "reduce":
            import torch
            self.scratch = [{}, {}] if algorithm == "mesh" else [None, None]
            if self.cuda:
                self.comm = torch.cuda.Stream(device=self.bus[0].device)
                self.done = [torch.cuda.Event(), torch.cuda.Event()]

    def _finish(self, j, stream):
        import torch
        import torch.distributed as dist
        if self.work[j] is not None:
            if self.work[j] is True:
                if self.cuda:
                    dev = self.bus[j].device
                    (torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.current_stream(dev)).wait_event(self.done[j])
            elif self.cuda and stream:
                with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.bus[j].device)):
                    self.work[j].wait()               # the render stream waits for the collective
            else:
                self.work[j].wait()                   # the current stream waits for the collective
            self.work[j] = None
            if self.algorithm != "mesh" and dist.get_rank(self.group) == self.dst and self.shape[j] is not None:
                nb, nf = self.shape[j]
                self.synth.levels_scan_device(self.bus[j].data_ptr(), nb, nf, stream=stream)   # levels see the final mix
            # ("mesh": every rank scanned its reduced piece inside zlhip_bus_reduce_sum_scan; the root imported the levels)

    def _render(self, nblocks, nframes, clocks, stream):
        j = self.i & 1
        self._finish(j, stream)                       # buffer j was reduced two steps ago
        self.synth.render_batch(nblocks, nframes, clocks, bus_out_dev=self.bus[j].data_ptr(), stream=stream)
        _order_collective_behind_render(self.synth, self.bus[j], stream)
        return j

    def _exchange(self, j, nblocks, nframes, stream):
        import torch
        import torch.distributed as dist
        if self.algorithm == "reduce":
            if self.cuda and stream:                  # RCCL orders the collective behind torch's *current* stream
                with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.bus[j].device)):
                    self.work[j] = dist.reduce(self.bus[j], dst=self.dst, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                self.work[j] = dist.reduce(self.bus[j], dst=self.dst, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            fn = (lambda: exchange_bus_mesh(self.synth, self.bus[j], nblocks, nframes, dst=self.dst, group=self.group, scratch=self.scratch[j])) \
                if self.algorithm == "mesh" else (lambda: reduce_bus_in_rank_order(self.bus[j], dst=self.dst, group=self.group))
            if self.cuda:
                # the exchange, the ordered sum and the gather run on the communication stream, behind the render
                # (queued on `stream` when one is given, else on torch's current stream)
                dev = self.bus[j].device
                self.comm.wait_stream(torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.current_stream(dev))
                with torch.cuda.stream(self.comm):
                    fn()
                    self.done[j].record(self.comm)
            else:
                fn()
            self.work[j] = True
        self.shape[j] = (nblocks, nframes)
        self.i += 1
        return self.bus[j]

    def step(self, nblocks: int, nframes: int, clocks, stream=None):
        j = self._render(nblocks, nframes, clocks, stream)
        return self._exchange(j, nblocks, nframes, stream)

    def flush(self, stream=None):
        for j in ((self.i & 1), ((self.i + 1) & 1)):
            self._finish(j, stream)

    def step_or_fall_back(self, nblocks: int, nframes: int, clocks, stream=None, log=None):
        """step() with the safety net bench.py uses in its warm-up steps: a collective this RCCL build refuses raises the same
        synchronous RuntimeError on every rank, before it has moved anything.  The batch is rendered once; when its exchange is
        refused, the exchanges still in flight are finished, the exchange is rebuilt on the plain RCCL reduce -- the most basic
        of the three -- over the SAME two partial buses, and the rendered batch goes through that one (the voices are not rendered
        a second time).  Returns (bus, exchange): `exchange` is self, or the rebuilt one the caller goes on with."""
        j = self._render(nblocks, nframes, clocks, stream)
        try:
            return self._exchange(j, nblocks, nframes, stream), self
        except RuntimeError as err:
            if self.algorithm == "reduce":
                raise
            if log is not None:
                log(f"bus exchange '{self.algorithm}' failed ({str(err)[:200]}); falling back to dist.reduce")
            self.work[j] = None
            self._finish(j ^ 1, stream)               # the batch before this one went through the old exchange: complete it
            bufs = iter(self.bus)
            other = OverlappedBusReduce(self.synth, lambda: next(bufs), dst=self.dst, group=self.group, algorithm="reduce")
            other.i, other.shape = self.i, list(self.shape)
            return other._exchange(j, nblocks, nframes, stream), other
