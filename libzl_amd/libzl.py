"""ctypes binding of include/libzl_hotpath.h: the libzl.h-named functions a zynthbox-style host calls
(reference test/playtest.py:25-49 binds the same names on the reference's libzl.so)."""
from __future__ import annotations

import ctypes as C

from . import _abi
from ._abi import Clock, PassthroughParams

_P = C.c_void_p
CB = C.CFUNCTYPE(None, C.c_float)
CLOCK_MS = C.CFUNCTYPE(C.c_int64)
TIMER_CB = C.CFUNCTYPE(None, C.c_int)

SIGNATURES = {
    "ClipAudioSource_byID": (_P, [C.c_int]),
    "ClipAudioSource_new": (_P, [C.c_char_p, C.c_bool]),
    "ClipAudioSource_setProgressCallback": (None, [_P, CB]),
    "ClipAudioSource_play": (None, [_P, C.c_bool]),
    "ClipAudioSource_stop": (None, [_P]),
    "ClipAudioSource_playOnChannel": (None, [_P, C.c_bool, C.c_int]),
    "ClipAudioSource_stopOnChannel": (None, [_P, C.c_int]),
    "ClipAudioSource_getDuration": (C.c_float, [_P]),
    "ClipAudioSource_getFileName": (C.c_char_p, [_P]),
    "ClipAudioSource_setStartPosition": (None, [_P, C.c_float]),
    "ClipAudioSource_setLength": (None, [_P, C.c_float, C.c_int]),
    "ClipAudioSource_setPan": (None, [_P, C.c_float]),
    "ClipAudioSource_setSpeedRatio": (None, [_P, C.c_float]),
    "ClipAudioSource_setPitch": (None, [_P, C.c_float]),
    "ClipAudioSource_setGain": (None, [_P, C.c_float]),
    "ClipAudioSource_setVolume": (None, [_P, C.c_float]),
    "ClipAudioSource_setAudioLevelChangedCallback": (None, [_P, CB]),
    "ClipAudioSource_setSlices": (None, [_P, C.c_int]),
    "ClipAudioSource_keyZoneStart": (C.c_int, [_P]),
    "ClipAudioSource_setKeyZoneStart": (None, [_P, C.c_int]),
    "ClipAudioSource_keyZoneEnd": (C.c_int, [_P]),
    "ClipAudioSource_setKeyZoneEnd": (None, [_P, C.c_int]),
    "ClipAudioSource_rootNote": (C.c_int, [_P]),
    "ClipAudioSource_setRootNote": (None, [_P, C.c_int]),
    "ClipAudioSource_destroy": (None, [_P]),
    "ClipAudioSource_id": (C.c_int, [_P]),
    "ClipAudioSource_adsrAttack": (C.c_float, [_P]),
    "ClipAudioSource_setADSRAttack": (None, [_P, C.c_float]),
    "ClipAudioSource_adsrDecay": (C.c_float, [_P]),
    "ClipAudioSource_setADSRDecay": (None, [_P, C.c_float]),
    "ClipAudioSource_adsrSustain": (C.c_float, [_P]),
    "ClipAudioSource_setADSRSustain": (None, [_P, C.c_float]),
    "ClipAudioSource_adsrRelease": (C.c_float, [_P]),
    "ClipAudioSource_setADSRRelease": (None, [_P, C.c_float]),
    "SyncTimer_getMultiplier": (C.c_int, []),
    "SyncTimer_startTimer": (None, [C.c_int]),
    "SamplerSynth_setChannelEnabled": (None, [C.c_int, C.c_bool]),
    "SyncTimer_setBpm": (None, [C.c_uint]),
    "SyncTimer_stopTimer": (None, []),
    "SyncTimer_registerTimerCallback": (None, [TIMER_CB]),
    "SyncTimer_deregisterTimerCallback": (None, [TIMER_CB]),
    "SyncTimer_queueClipToStart": (None, [_P]),
    "SyncTimer_queueClipToStartOnChannel": (None, [_P, C.c_int]),
    "SyncTimer_queueClipToStop": (None, [_P]),
    "SyncTimer_queueClipToStopOnChannel": (None, [_P, C.c_int]),
    "libzl_hotpath_cycle": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint64, C.c_float, C.c_void_p, C.c_void_p]),
    "libzl_hotpath_cycle_fanout": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "libzl_hotpath_process_fanout": (C.c_int, [C.c_uint32, C.POINTER(Clock), C.c_void_p, C.c_void_p, C.c_void_p]),
    "libzl_hotpath_dropped_requests": (C.c_uint64, []),
    "libzl_hotpath_schedule_clip_command": (None, [C.POINTER(_abi.ClipCommand), C.c_uint64]),
    "libzl_hotpath_clip_params": (C.c_int, [C.c_void_p, C.POINTER(_abi.ClipParams)]),
    "libzl_hotpath_timer_tick": (None, []),
    "libzl_hotpath_bounce_to_wav": (C.c_int, [C.c_char_p, C.c_int64, C.c_uint32, C.c_uint64, C.c_int]),
    "libzl_hotpath_transport": (C.c_int, [C.POINTER(Clock)]),
    "initJuce": (None, []),
    "shutdownJuce": (None, []),
    "stopClips": (None, [C.c_int, C.POINTER(_P)]),
    "dBFromVolume": (C.c_float, [C.c_float]),
    "JackPassthrough_setPanAmount": (None, [C.c_int, C.c_float]),
    "JackPassthrough_getPanAmount": (C.c_float, [C.c_int]),
    "JackPassthrough_getWetFx1Amount": (C.c_float, [C.c_int]),
    "JackPassthrough_setWetFx1Amount": (None, [C.c_int, C.c_float]),
    "JackPassthrough_getWetFx2Amount": (C.c_float, [C.c_int]),
    "JackPassthrough_setWetFx2Amount": (None, [C.c_int, C.c_float]),
    "JackPassthrough_getDryAmount": (C.c_float, [C.c_int]),
    "JackPassthrough_setDryAmount": (None, [C.c_int, C.c_float]),
    "JackPassthrough_getMuted": (C.c_float, [C.c_int]),
    "JackPassthrough_setMuted": (None, [C.c_int, C.c_bool]),
    "JackPassthrough_getParams": (C.c_int, [C.c_int, C.POINTER(PassthroughParams)]),
    "libzl_hotpath_configure": (None, [C.POINTER(_abi.Config)]),
    "libzl_hotpath_status": (C.c_int, []),
    "libzl_hotpath_engine": (_P, []),
    "libzl_hotpath_set_clock_ms": (None, [CLOCK_MS]),
    "ClipAudioSource_newFromBuffer": (_P, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_char_p]),
    "libzl_hotpath_process": (C.c_int, [C.c_uint32, C.POINTER(Clock), C.c_void_p, C.c_void_p]),
    "ClipAudioSource_peakGain": (C.c_float, [_P]),
    "ClipAudioSource_firstProgress": (C.c_double, [_P]),
    "ClipAudioSource_volumeAbsolute": (C.c_float, [_P]),
    "ClipAudioSource_setVolumeAbsolute": (None, [_P, C.c_float]),
    "ClipAudioSource_engineClip": (C.c_int, [_P]),
    "libzl_wav_read": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "libzl_wav_free": (None, [C.POINTER(C.c_float)]),
    "libzl_wav_write": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int]),
    "libzl_wav_write_interleaved": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int]),
}


def load():
    """libzlhip.so with the libzl.h-named symbols bound (same library as the engine ABI)."""
    lib = _abi.load()
    _abi.bind(lib, SIGNATURES)
    return lib
