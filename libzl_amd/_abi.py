"""ctypes mirror of include/zlhip.h and the loader of libzl_amd/lib/libzlhip.so.

The loader fails loudly: there is no CPU render path and no fallback library.  If libzlhip.so is
missing or no HIP device is usable, the error says so.
"""
from __future__ import annotations

import ctypes as C
import os

ZLHIP_OK = 0
ZLHIP_ERR_INVALID = -1
ZLHIP_ERR_NO_DEVICE = -2
ZLHIP_ERR_HIP = -3
ZLHIP_ERR_CAPACITY = -4
ZLHIP_ERR_STATE = -5

MODE_FAITHFUL = 0
MODE_FIX_GAIN = 1
MODE_FIX_DELAY = 2
MODE_HERMITE = 4

MAX_SLICES = 128
BEAT_SUBDIVISIONS = 96


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("num_buses", C.c_int32),
        ("voices_per_bus", C.c_int32), ("max_frames", C.c_int32), ("max_batch_blocks", C.c_int32),
        ("max_sounds", C.c_int32), ("mode", C.c_uint32), ("playback_sample_rate", C.c_double),
        ("sound_arena_bytes", C.c_uint64), ("voices_per_task", C.c_int32), ("plan_window_blocks", C.c_int32),
        ("rt_idle_timeout_us", C.c_int32), ("sound_arena_max_bytes", C.c_uint64),
    ]


class Clock(C.Structure):
    _fields_ = [
        ("current_usecs", C.c_uint64), ("next_usecs", C.c_uint64), ("jack_playhead", C.c_uint64),
        ("jack_playhead_usecs", C.c_uint64), ("jack_subbeat_length_usecs", C.c_uint64),
    ]


class ClipParams(C.Structure):
    _fields_ = [
        ("start_position_seconds", C.c_float), ("length_seconds", C.c_float), ("length_in_beats", C.c_float),
        ("volume_absolute", C.c_float), ("pan", C.c_float), ("duration_seconds", C.c_float),
        ("adsr_attack", C.c_float), ("adsr_decay", C.c_float), ("adsr_sustain", C.c_float), ("adsr_release", C.c_float),
        ("root_note", C.c_int32), ("num_slice_positions", C.c_int32),
        ("slice_positions", C.c_double * MAX_SLICES),
    ]


class ClipCommand(C.Structure):
    _fields_ = [
        ("clip", C.c_int32), ("midi_note", C.c_int32), ("midi_channel", C.c_int32),
        ("start_playback", C.c_int32), ("stop_playback", C.c_int32),
        ("change_slice", C.c_int32), ("slice", C.c_int32),
        ("change_looping", C.c_int32), ("looping", C.c_int32),
        ("change_pitch", C.c_int32), ("pitch_change", C.c_float),
        ("change_speed", C.c_int32), ("speed_ratio", C.c_float),
        ("change_gain_db", C.c_int32), ("gain_db", C.c_float),
        ("change_volume", C.c_int32), ("volume", C.c_float),
    ]


class VoiceReport(C.Structure):
    _fields_ = [
        ("playing", C.c_int32), ("valid", C.c_int32), ("gain", C.c_float), ("progress", C.c_float),
        ("clip", C.c_int32), ("reserved", C.c_int32), ("source_sample_position", C.c_double),
    ]


class Levels(C.Structure):
    _fields_ = [
        ("peak_a", C.c_int32), ("peak_b", C.c_int32),
        ("peak_a_hold_signal", C.c_float), ("peak_b_hold_signal", C.c_float),
        ("peak_db_a", C.c_float), ("peak_db_b", C.c_float), ("combined_db", C.c_float),
        ("hold_db_a", C.c_float), ("hold_db_b", C.c_float), ("rms_a", C.c_float), ("rms_b", C.c_float),
    ]


class PassthroughParams(C.Structure):
    _fields_ = [
        ("dry_amount", C.c_float), ("wet_fx1_amount", C.c_float), ("wet_fx2_amount", C.c_float),
        ("pan_amount", C.c_float), ("muted", C.c_int32),
    ]


class Timings(C.Structure):
    _fields_ = [
        ("plan_ms", C.c_float), ("render_ms", C.c_float), ("finalize_ms", C.c_float), ("total_ms", C.c_float),
        ("source_bytes", C.c_uint64), ("slow_blocks", C.c_uint64), ("active_voice_frames", C.c_uint64),
        ("render_launches", C.c_int32), ("reserved", C.c_int32),
    ]


class RtCycleTrace(C.Structure):
    _fields_ = [
        ("cycle", C.c_uint64), ("resident", C.c_int32), ("reserved", C.c_int32),
        ("total_us", C.c_double), ("before_post_us", C.c_double), ("wait_us", C.c_double), ("after_us", C.c_double),
        ("max_poll_gap_us", C.c_double), ("device_us", C.c_double), ("involuntary_switches", C.c_int64),
    ]


# every symbol include/zlhip.h declares: name -> (restype, argtypes)
_F = C.POINTER(C.c_float)
_E = C.c_void_p
SIGNATURES = {
    "zlhip_abi_version": (C.c_int, []),
    "zlhip_config_default": (None, [C.POINTER(Config)]),
    "zlhip_engine_create": (C.c_int, [C.POINTER(Config), C.POINTER(_E)]),
    "zlhip_engine_destroy": (None, [_E]),
    "zlhip_last_error": (C.c_char_p, [_E]),
    "zlhip_strerror": (C.c_char_p, [C.c_int]),
    "zlhip_sound_upload": (C.c_int, [_E, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.POINTER(C.c_int32)]),
    "zlhip_sound_upload_device": (C.c_int, [_E, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.POINTER(C.c_int32)]),
    "zlhip_sound_upload_device_on": (C.c_int, [_E, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.POINTER(C.c_int32)]),
    "zlhip_sound_release": (C.c_int, [_E, C.c_int32]),
    "zlhip_clip_params_default": (None, [C.POINTER(ClipParams), C.c_float]),
    "zlhip_clip_set": (C.c_int, [_E, C.c_int32, C.POINTER(ClipParams)]),
    "zlhip_clip_command_clear": (None, [C.POINTER(ClipCommand)]),
    "zlhip_handle_command": (C.c_int, [_E, C.POINTER(ClipCommand), C.c_uint64]),
    "zlhip_handle_commands": (C.c_int, [_E, C.POINTER(ClipCommand), C.c_int32, C.c_uint64, C.POINTER(C.c_int32)]),
    "zlhip_handle_commands_voices": (C.c_int, [_E, C.POINTER(ClipCommand), C.c_int32, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "zlhip_bus_set_enabled": (C.c_int, [_E, C.c_int32, C.c_int]),
    "zlhip_start_voice": (C.c_int, [_E, C.c_int32, C.c_int32, C.POINTER(ClipCommand), C.c_uint64]),
    "zlhip_stop_voice": (C.c_int, [_E, C.c_int32, C.c_int32, C.c_int]),
    "zlhip_update_voice": (C.c_int, [_E, C.c_int32, C.c_int32, C.POINTER(ClipCommand)]),
    "zlhip_voice_is_playing": (C.c_int, [_E, C.c_int32, C.c_int32]),
    "zlhip_render": (C.c_int, [_E, C.c_int32, C.POINTER(Clock), C.c_void_p, C.c_void_p]),
    "zlhip_render_fanout": (C.c_int, [_E, C.c_int32, C.POINTER(Clock), C.c_void_p, C.c_void_p, C.POINTER(PassthroughParams), C.c_void_p]),
    "zlhip_render_batch": (C.c_int, [_E, C.c_int32, C.c_int32, C.POINTER(Clock), C.c_void_p, C.c_void_p]),
    "zlhip_render_batch_fanout": (C.c_int, [_E, C.c_int32, C.c_int32, C.POINTER(Clock), C.c_void_p, C.POINTER(PassthroughParams), C.c_void_p, C.c_void_p]),
    "zlhip_synchronize": (C.c_int, [_E]),
    "zlhip_read_bus": (C.c_int, [_E, C.c_void_p, C.c_size_t]),
    "zlhip_voice_reports": (C.c_int, [_E, C.POINTER(VoiceReport), C.c_int32]),
    "zlhip_debug_enable_trace": (C.c_int, [_E, C.c_int]),
    "zlhip_debug_read_trace": (C.c_int, [_E, C.c_void_p, C.c_size_t]),
    "zlhip_levels_tick": (C.c_int, [_E, C.c_int32, C.c_int32, C.POINTER(Levels)]),
    "zlhip_block_peaks": (C.c_int, [_E, C.c_void_p, C.c_size_t]),
    "zlhip_levels_scan_device": (C.c_int, [_E, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "zlhip_memory_bytes": (C.c_int, [_E, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "zlhip_bounce": (C.c_int, [_E, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "zlhip_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "zlhip_host_free": (None, [C.c_void_p]),
    "zlhip_bus_reduce_sum_scan": (C.c_int, [_E, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "zlhip_levels_import_units": (C.c_int, [_E, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "zlhip_passthrough_params_default": (None, [C.POINTER(PassthroughParams)]),
    "zlhip_passthrough_process": (C.c_int, [_E, C.POINTER(PassthroughParams), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "zlhip_set_profiling": (C.c_int, [_E, C.c_int]),
    "zlhip_last_timings": (C.c_int, [_E, C.POINTER(Timings)]),
    "zlhip_profile_totals": (C.c_int, [_E, C.POINTER(Timings), C.POINTER(C.c_int32), C.c_int]),
    "zlhip_bus_device_ptr": (C.c_void_p, [_E]),
    "zlhip_device_name": (C.c_int, [_E, C.c_char_p, C.c_size_t]),
    "zlhip_rt_stats": (C.c_int, [_E, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "zlhip_rt_residency": (C.c_int, [_E, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "zlhip_rt_last_cycle": (C.c_int, [_E, C.POINTER(RtCycleTrace)]),
}

# ZLHIP_LIBRARY selects another build of the same library (A/B measurements of kernel variants); never a fallback
LIB_PATH = os.environ.get("ZLHIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libzlhip.so")
_lib = None


def bind(lib, signatures=SIGNATURES):
    for name, (res, args) in signatures.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """Load libzlhip.so (built in-tree by libzl_amd/build.py).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m libzl_amd.build` "
                "(hipcc --offload-arch=gfx950).  libzl_amd has no CPU fallback.")
        _lib = bind(C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL))
    return _lib


class UnitLevels(C.Structure):
    _fields_ = [("peak", C.c_int32), ("sumsq", C.c_float)]


class ZlHipError(RuntimeError):
    pass


def check(lib, engine, rc, what):
    if rc < 0:
        detail = lib.zlhip_last_error(engine).decode() if engine else ""
        raise ZlHipError(f"{what}: {lib.zlhip_strerror(rc).decode()} ({rc}) {detail}")
    return rc
