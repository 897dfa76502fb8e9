"""ctypes wrapper of tests/cpu_harness/plan_host.cpp -- TEST HARNESS ONLY.

Runs the engine's __host__ __device__ planning / per-frame code on the CPU so the CPU-only test
tier can compare it with the oracle.  Not part of the product and never imported by libzl_amd.
"""
import ctypes as C

import numpy as np

from libzl_amd import build
from libzl_amd._abi import ClipCommand, ClipParams, Clock, VoiceReport

_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(build.build_cpu_harness())
        l.zlsim_create.restype = C.c_void_p
        l.zlsim_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_uint32, C.c_int]
        l.zlsim_destroy.argtypes = [C.c_void_p]
        l.zlsim_set_ctl_slots.argtypes = [C.c_void_p, C.c_int]
        l.zlsim_clip_set.argtypes = [C.c_void_p, C.c_int, C.POINTER(ClipParams)]
        l.zlsim_sound_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double]
        l.zlsim_handle_command.argtypes = [C.c_void_p, C.POINTER(ClipCommand), C.c_uint64]
        l.zlsim_start_voice.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(ClipCommand), C.c_uint64]
        l.zlsim_update_voice.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(ClipCommand)]
        l.zlsim_set_bus_enabled.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.zlsim_stop_voice.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        l.zlsim_render_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Clock), C.c_void_p, C.c_int]
        l.zlsim_reports.argtypes = [C.c_void_p, C.POINTER(VoiceReport)]
        l.zlsim_trace.argtypes = [C.c_void_p, C.c_void_p]
        l.zlsim_block_peaks.argtypes = [C.c_void_p, C.c_void_p]
        l.zlsim_slow_blocks.restype = C.c_ulonglong
        l.zlsim_slow_blocks.argtypes = [C.c_void_p]
        l.zlsim_source_bytes.restype = C.c_ulonglong
        l.zlsim_source_bytes.argtypes = [C.c_void_p]
        l.zlsim_nseg.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.zlsim_plan_flags.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.zlsim_check_linear_runs.restype = C.c_longlong
        l.zlsim_check_linear_runs.argtypes = [C.c_double, C.c_double, C.c_longlong, C.POINTER(C.c_longlong)]
        l.zlsim_steps_to_reach.argtypes = [C.c_double, C.c_double, C.c_double]
        _lib = l
    return _lib


class SimSynth:
    """Same surface as libzl_amd.SamplerSynth, executed by the CPU harness."""

    def __init__(self, num_buses=12, voices_per_bus=8, *, mode=0, playback_sample_rate=48000.0,
                 max_sounds=1024, voices_per_task=0, ctl_pool_slots=-1, **_):
        self.l = lib()
        self.num_buses, self.voices_per_bus, self.mode = num_buses, voices_per_bus, mode
        self.num_voices = num_buses * voices_per_bus
        self.s = C.c_void_p(self.l.zlsim_create(num_buses, voices_per_bus, max_sounds, playback_sample_rate, mode, voices_per_task))
        if ctl_pool_slots >= 0:                              # the window's control pool (-1: a slot for every (block, voice))
            self.l.zlsim_set_ctl_slots(self.s, ctl_pool_slots)
        self._last = (0, 0)
        self.force_slow = False
        self.no_periodic = False

    def close(self):
        if self.s:
            self.l.zlsim_destroy(self.s)
            self.s = None

    def register_clip(self, left, right, sample_rate):
        left = np.ascontiguousarray(left, dtype=np.float32)
        rp = None
        if right is not None:
            right = np.ascontiguousarray(right, dtype=np.float32)
            rp = right.ctypes.data
        i = self.l.zlsim_sound_upload(self.s, left.ctypes.data, rp, left.shape[0], float(sample_rate))
        assert i >= 0
        return i

    def set_clip_params(self, clip, params):
        self.l.zlsim_clip_set(self.s, clip, C.byref(params))

    def handle_clip_command(self, cmd, current_tick=0):
        return self.l.zlsim_handle_command(self.s, C.byref(cmd), current_tick)

    def start_voice(self, bus, slot, cmd, current_tick=0):
        return self.l.zlsim_start_voice(self.s, bus, slot, C.byref(cmd), current_tick)

    def set_bus_enabled(self, bus, enabled):
        self.l.zlsim_set_bus_enabled(self.s, bus, 1 if enabled else 0)

    def update_voice(self, bus, slot, cmd):
        return self.l.zlsim_update_voice(self.s, bus, slot, C.byref(cmd))

    def stop_voice(self, bus, slot, allow_tail_off=True):
        return self.l.zlsim_stop_voice(self.s, bus, slot, 1 if allow_tail_off else 0)

    def render_batch(self, nblocks, nframes, clocks, bus_out_dev=None, stream=None):
        self._bus = np.zeros((self.num_buses, 2, nblocks * nframes), dtype=np.float32)
        self.l.zlsim_render_batch(self.s, nblocks, nframes, clocks, self._bus.ctypes.data, (1 if self.force_slow else 0) | (2 if self.no_periodic else 0))
        self._last = (nblocks, nframes)
        if bus_out_dev:                                  # "device" buffer of the caller = host memory in this harness
            C.memmove(bus_out_dev, self._bus.ctypes.data, self._bus.nbytes)

    def levels_scan_device(self, bus_ptr, nblocks, nframes, stream=None):
        n = self.num_buses * 2 * nblocks * nframes
        buf = np.ctypeslib.as_array(C.cast(bus_ptr, C.POINTER(C.c_float)), (n,)).reshape(self.num_buses, 2, nblocks, nframes)
        v = np.abs(np.float32(131072.0) * buf)
        self.scanned_peaks = v.astype(np.int64).max(axis=3).transpose(2, 0, 1)      # [block][bus][channel]

    # host stand-ins of the two exchange kernels (zlhip_bus_reduce_sum_scan / zlhip_levels_import_units): the same defined
    # arithmetic -- ((0 + p0) + p1) + ... in fp32, integer peak, sums of squares through the oracle's order -- on host pointers
    def bus_reduce_sum_scan(self, pieces_ptr, npieces, piece_stride_floats, units, nframes, sum_out_ptr, levels_out_ptr, stream=None):
        from oracle import zl_oracle as zo
        lib = zo.load()
        src = np.ctypeslib.as_array(C.cast(pieces_ptr, C.POINTER(C.c_float)), (npieces * piece_stride_floats,)).reshape(npieces, piece_stride_floats)
        out = np.ctypeslib.as_array(C.cast(sum_out_ptr, C.POINTER(C.c_float)), (units * nframes,))
        lv = np.ctypeslib.as_array(C.cast(levels_out_ptr, C.POINTER(C.c_int32)), (units * 2,)).reshape(units, 2)
        acc = np.zeros(units * nframes, dtype=np.float32)
        for r in range(npieces):
            acc = acc + src[r, :units * nframes]
        out[:] = acc
        off = 0 if (self.mode & 2) else 1
        rows = np.ascontiguousarray(acc.reshape(units, nframes))
        lv[:, 0] = np.abs(np.float32(131072.0) * rows).astype(np.int64).max(axis=1)
        sq = np.array([lib.zlo_block_sumsq(rows[u].ctypes.data, nframes, off) for u in range(units)], dtype=np.float32)
        lv[:, 1] = sq.view(np.int32)

    def levels_import_units(self, units_ptr, nblocks, nframes, stream=None):
        u = np.ctypeslib.as_array(C.cast(units_ptr, C.POINTER(C.c_int32)), (self.num_buses * 2 * nblocks * 2,)).reshape(self.num_buses, 2, nblocks, 2)
        self.scanned_peaks = u[:, :, :, 0].transpose(2, 0, 1).astype(np.int64)                  # [block][bus][channel]
        self.scanned_sumsq = np.ascontiguousarray(u[:, :, :, 1]).view(np.float32).transpose(2, 0, 1)
        self._last = (nblocks, nframes)

    def read_bus(self):
        return self._bus

    def voice_reports(self):
        arr = (VoiceReport * self.num_voices)()
        self.l.zlsim_reports(self.s, arr)
        return arr

    def enable_trace(self, enable=True, force_slow=False, no_periodic=False):
        self.force_slow = force_slow
        self.no_periodic = no_periodic

    def read_trace(self):
        K, N = self._last
        out = np.empty((K, self.num_voices, N), dtype=np.int32)
        self.l.zlsim_trace(self.s, out.ctypes.data)
        return out

    def block_peaks(self):
        K, _ = self._last
        out = np.empty((K, self.num_buses, 2), dtype=np.int32)
        self.l.zlsim_block_peaks(self.s, out.ctypes.data)
        return out

    def slow_blocks(self):
        return self.l.zlsim_slow_blocks(self.s)

    def source_bytes(self):
        return self.l.zlsim_source_bytes(self.s)

    def nseg(self, k, v):
        return self.l.zlsim_nseg(self.s, k, v)
