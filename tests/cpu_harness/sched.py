"""ctypes wrapper of tests/cpu_harness/sched_host.cpp -- TEST HARNESS ONLY: the product's ClipCommand scheduler
(libzl_amd/csrc/zl_sched.h) compiled for the host, so the CPU tier can hold it against the oracle's restatement."""
import ctypes as C
import os

from libzl_amd import build
from libzl_amd._abi import ClipCommand, Clock


class Dispatch(C.Structure):
    _fields_ = [("cmd", ClipCommand), ("tick", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build.build_cpu_harness()
        l = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libzl_sched_host.so"))
        P = C.c_void_p
        l.zlsched_new.restype = P
        l.zlsched_free.argtypes = [P]
        l.zlsched_schedule.argtypes = [P, C.POINTER(ClipCommand), C.c_uint64]
        l.zlsched_set_latency.argtypes = [P, C.c_uint32, C.c_double]
        l.zlsched_set_bpm.argtypes = [P, C.c_uint64]
        l.zlsched_start.argtypes = [P, C.c_int]
        l.zlsched_stop.argtypes = [P]
        l.zlsched_timer_callback.argtypes = [P]
        l.zlsched_queue_start.argtypes = [P, C.c_int32, C.c_int]
        l.zlsched_queue_stop.argtypes = [P, C.c_int32, C.c_int]
        l.zlsched_process.restype = C.c_int32
        l.zlsched_process.argtypes = [P, C.c_uint32, C.c_uint64, C.c_uint64, C.c_float, C.POINTER(Dispatch), C.c_int32]
        l.zlsched_clock.argtypes = [P, C.POINTER(Clock)]
        l.zlsched_ext_schedule.argtypes = [P, C.POINTER(ClipCommand), C.c_uint64]
        l.zlsched_ext_process.restype = C.c_int32
        l.zlsched_ext_process.argtypes = [P, C.c_uint64, C.POINTER(Dispatch), C.c_int32]
        _lib = l
    return _lib


CMD_FIELDS = [f for f, _ in ClipCommand._fields_]


def cmd_tuple(c):
    return tuple(getattr(c, f) for f in CMD_FIELDS)


class ProductScheduler:
    def __init__(self):
        self.l = lib()
        self.s = self.l.zlsched_new()
        self._out = (Dispatch * 4096)()

    def close(self):
        if self.s:
            self.l.zlsched_free(self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def schedule(self, cmd, delay=0): self.l.zlsched_schedule(self.s, C.byref(cmd), delay)
    def set_latency(self, n, fs): self.l.zlsched_set_latency(self.s, n, fs)
    def set_bpm(self, bpm): self.l.zlsched_set_bpm(self.s, bpm)
    def start(self, bpm): self.l.zlsched_start(self.s, bpm)
    def stop(self): self.l.zlsched_stop(self.s)
    def timer_callback(self): self.l.zlsched_timer_callback(self.s)
    def queue_start(self, clip, ch): self.l.zlsched_queue_start(self.s, clip, ch)
    def queue_stop(self, clip, ch): self.l.zlsched_queue_stop(self.s, clip, ch)

    def _collect(self, n):
        assert n <= len(self._out)
        return [(cmd_tuple(self._out[i].cmd), int(self._out[i].tick)) for i in range(n)]

    def process(self, nframes, cu, nx):
        return self._collect(self.l.zlsched_process(self.s, nframes, cu, nx, C.c_float(float(nx - cu)), self._out, len(self._out)))

    def clock(self):
        c = Clock()
        self.l.zlsched_clock(self.s, C.byref(c))
        return (c.jack_playhead, c.jack_playhead_usecs, c.jack_subbeat_length_usecs)

    def ext_schedule(self, cmd, delay=0): self.l.zlsched_ext_schedule(self.s, C.byref(cmd), delay)

    def ext_process(self, playhead):
        return self._collect(self.l.zlsched_ext_process(self.s, playhead, self._out, len(self._out)))
