"""libzl_amd -- MI355X-native sample playback / mixing / metering engine behind libzl's hot path.

The product is libzl_amd/lib/libzlhip.so (HIP kernels for gfx950 + the C-ABI of include/zlhip.h and
the libzl.h-named symbols of include/libzl_hotpath.h).  This package is the thin ctypes binding a
Python host (zynthbox itself drives libzl through ctypes, reference test/playtest.py:25-49) uses.
"""
from .engine import (SamplerSynth, Clock, ClipCommand, ClipParams, Levels, PassthroughParams, VoiceReport,  # noqa: F401
                     MODE_FAITHFUL, MODE_FIX_GAIN, MODE_FIX_DELAY, MODE_HERMITE, ZlHipError, clip_command,
                     synthetic_clocks)

__version__ = "0.1.0"
