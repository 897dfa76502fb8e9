/*
 * libzl_hotpath.h -- the libzl.h-named entry points of the hot path, served by the MI355X engine.
 *
 * Same unmangled symbol names, argument orders and (absent) error behaviour as the reference's
 * extern "C" block (/root/reference/lib/libzl.h:18-179, implemented in lib/libzl.cpp:107-575), for
 * the functions that feed or read the sampler hot path.  A host that drives libzl through ctypes
 * (reference test/playtest.py:25-49) binds these exactly as before; handles stay opaque pointers.
 * Functions of libzl.h that belong to out-of-scope subsystems (SyncTimer scheduling, MIDI routing,
 * WAV recording, QML registration) are NOT declared here; see INTEGRATION.md for how they keep
 * linking against the unchanged reference objects.
 *
 * Differences that are visible at this boundary:
 *   - ClipAudioSource_new decodes RIFF/WAVE itself (PCM 8/16/24/32, float32/64; first two channels,
 *     as SamplerSynthSound.cpp:45) instead of going through JUCE / tracktion.
 *   - audio is pulled with libzl_hotpath_process() by whoever owns the JACK callback (the reference's
 *     SamplerChannel::process, SamplerSynth.cpp:116-148) instead of being pushed to JACK from inside.
 *   - ClipAudioSource_play/stop go through SyncTimer::scheduleClipCommand with delay 0 as in the reference
 *     (ClipAudioSource.cpp:428,437; SyncTimer.cpp:1011-1048): equivalent commands that meet in one step are merged, and
 *     a command reaches the sampler with the playhead of the step it is dispatched in (SyncTimer.cpp:553-558).  Two ways
 *     to run the cycle: libzl_hotpath_process (the host's own SyncTimer owns the transport and hands its getters over in
 *     zlhip_clock) and libzl_hotpath_cycle (this library runs the step ring itself, SyncTimer.cpp:452-702).
 *   - threading: setters, play / stop / queue calls and getters never take a lock the cycle holds (they publish a
 *     snapshot or post a request that the next cycle picks up); clip creation / destruction, initJuce / shutdownJuce do.
 *     The progress / level callbacks fire on the cycle's thread after its lock is released: they may call this API.
 */
#ifndef LIBZL_HOTPATH_H
#define LIBZL_HOTPATH_H

#include <stdbool.h>
#include <stdint.h>

#include "zlhip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ClipAudioSource ClipAudioSource;

/* ---- ClipAudioSource API bridge (libzl.h:23-61, libzl.cpp:107-302) ------------------------- */
ClipAudioSource *ClipAudioSource_byID(int id);                                          /* libzl.h:23 */
ClipAudioSource *ClipAudioSource_new(const char *filepath, bool muted);                 /* libzl.h:24 */
void ClipAudioSource_setProgressCallback(ClipAudioSource *c, void (*functionPtr)(float)); /* libzl.h:25 */
void ClipAudioSource_play(ClipAudioSource *c, bool loop);                               /* libzl.h:28 */
void ClipAudioSource_stop(ClipAudioSource *c);                                          /* libzl.h:29 */
void ClipAudioSource_playOnChannel(ClipAudioSource *c, bool loop, int midiChannel);     /* libzl.h:30 */
void ClipAudioSource_stopOnChannel(ClipAudioSource *c, int midiChannel);                /* libzl.h:31 */
float ClipAudioSource_getDuration(ClipAudioSource *c);                                  /* libzl.h:32 */
const char *ClipAudioSource_getFileName(ClipAudioSource *c);                            /* libzl.h:33 */
void ClipAudioSource_setStartPosition(ClipAudioSource *c, float startPositionInSeconds);/* libzl.h:34 */
void ClipAudioSource_setLength(ClipAudioSource *c, float beat, int bpm);                /* libzl.h:36 */
void ClipAudioSource_setPan(ClipAudioSource *c, float pan);                             /* libzl.h:37 */
void ClipAudioSource_setSpeedRatio(ClipAudioSource *c, float speedRatio);               /* libzl.h:38 (stored, unused by the voice: Q11) */
void ClipAudioSource_setPitch(ClipAudioSource *c, float pitchChange);                   /* libzl.h:39 (stored, unused by the voice: Q11) */
void ClipAudioSource_setGain(ClipAudioSource *c, float db);                             /* libzl.h:40 (stored, unused by the voice: Q11) */
void ClipAudioSource_setVolume(ClipAudioSource *c, float vol);                          /* libzl.h:41 */
void ClipAudioSource_setAudioLevelChangedCallback(ClipAudioSource *c, void (*functionPtr)(float)); /* libzl.h:42 */
void ClipAudioSource_setSlices(ClipAudioSource *c, int slices);                         /* libzl.h:44 */
int  ClipAudioSource_keyZoneStart(ClipAudioSource *c);                                  /* libzl.h:45 */
void ClipAudioSource_setKeyZoneStart(ClipAudioSource *c, int keyZoneStart);             /* libzl.h:46 */
int  ClipAudioSource_keyZoneEnd(ClipAudioSource *c);                                    /* libzl.h:47 */
void ClipAudioSource_setKeyZoneEnd(ClipAudioSource *c, int keyZoneEnd);                 /* libzl.h:48 */
int  ClipAudioSource_rootNote(ClipAudioSource *c);                                      /* libzl.h:49 */
void ClipAudioSource_setRootNote(ClipAudioSource *c, int rootNote);                     /* libzl.h:50 */
void ClipAudioSource_destroy(ClipAudioSource *c);                                       /* libzl.h:51 */
int  ClipAudioSource_id(ClipAudioSource *c);                                            /* libzl.h:52 */
float ClipAudioSource_adsrAttack(ClipAudioSource *c);                                   /* libzl.h:54 */
void ClipAudioSource_setADSRAttack(ClipAudioSource *c, float newValue);                 /* libzl.h:55 */
float ClipAudioSource_adsrDecay(ClipAudioSource *c);                                    /* libzl.h:56 */
void ClipAudioSource_setADSRDecay(ClipAudioSource *c, float newValue);                  /* libzl.h:57 */
float ClipAudioSource_adsrSustain(ClipAudioSource *c);                                  /* libzl.h:58 */
void ClipAudioSource_setADSRSustain(ClipAudioSource *c, float newValue);                /* libzl.h:59 */
float ClipAudioSource_adsrRelease(ClipAudioSource *c);                                  /* libzl.h:60 */
void ClipAudioSource_setADSRRelease(ClipAudioSource *c, float newValue);                /* libzl.h:61 */

/* ---- SyncTimer API bridge, the part that schedules ClipCommands (libzl.h:69-79, libzl.cpp:311-349) ----
 * Served by the step ring of libzl_hotpath_cycle; the registered timer callbacks fire on the cycle's thread after each cycle, once per
 * tick the timer went through.  SyncTimer_instance returns the reference's Qt object and is not declared here. */
void SyncTimer_startTimer(int interval);                                                /* libzl.h:70: SyncTimer::start(bpm), SyncTimer.cpp:870-879 */
/* SamplerSynth::setChannelEnabled(channel, enabled) (SamplerSynth.h:48-53, SamplerSynth.cpp:343-351; a method of the SamplerSynth singleton, not a
 * libzl.h symbol): channels -2 .. 9.  A disabled channel takes its commands, its voices stand still, its output is silent (zlhip_bus_set_enabled). */
void SamplerSynth_setChannelEnabled(int channel, bool enabled);
void SyncTimer_setBpm(unsigned int bpm);                                                /* libzl.h:71, SyncTimer.cpp:954-975 */
int  SyncTimer_getMultiplier(void);                                                     /* libzl.h:72, SyncTimer.cpp:946-948 */
void SyncTimer_stopTimer(void);                                                         /* libzl.h:73, SyncTimer.cpp:881-925 */
void SyncTimer_registerTimerCallback(void (*functionPtr)(int));                         /* libzl.h:74, SyncTimer.cpp:790-794: functionPtr(beat) per timer tick */
void SyncTimer_deregisterTimerCallback(void (*functionPtr)(int));                       /* libzl.h:75, SyncTimer.cpp:796-812 */
void SyncTimer_queueClipToStart(ClipAudioSource *clip);                                 /* libzl.h:76 */
void SyncTimer_queueClipToStartOnChannel(ClipAudioSource *clip, int midiChannel);       /* libzl.h:77, SyncTimer.cpp:815-832 */
void SyncTimer_queueClipToStop(ClipAudioSource *clip);                                  /* libzl.h:78 */
void SyncTimer_queueClipToStopOnChannel(ClipAudioSource *clip, int midiChannel);        /* libzl.h:79, SyncTimer.cpp:834-860 */

/* ---- misc (libzl.h:84-90) ----------------------------------------------------------------- */
void initJuce(void);                                                                    /* libzl.h:84: brings the engine up (12 channels x 8 voices) */
void shutdownJuce(void);                                                                /* libzl.h:85 */
void stopClips(int size, ClipAudioSource **clips);                                      /* libzl.h:89 */
float dBFromVolume(float vol);                                                          /* libzl.h:90 */

/* ---- JackPassthrough API bridge (libzl.h:117-175, libzl.cpp JackPassthrough_*) --------------- */
void  JackPassthrough_setPanAmount(int channel, float amount);
float JackPassthrough_getPanAmount(int channel);
float JackPassthrough_getWetFx1Amount(int channel);
void  JackPassthrough_setWetFx1Amount(int channel, float amount);
float JackPassthrough_getWetFx2Amount(int channel);
void  JackPassthrough_setWetFx2Amount(int channel, float amount);
float JackPassthrough_getDryAmount(int channel);
void  JackPassthrough_setDryAmount(int channel, float amount);
float JackPassthrough_getMuted(int channel);
void  JackPassthrough_setMuted(int channel, bool muted);

/* the parameter set of one passthrough client as zlhip_passthrough_process takes it (build-defined) */
int   JackPassthrough_getParams(int channel, zlhip_passthrough_params *out);

/* ---- build-defined additions (absent in libzl.h) ---------------------------------------------- */
/* engine configuration used by the next initJuce() (defaults: 12 x 8 voices, 48 kHz, faithful mode) */
void libzl_hotpath_configure(const zlhip_config *cfg);
/* 0 if the engine is up, else the zlhip status that initJuce() hit (e.g. ZLHIP_ERR_NO_DEVICE) */
int  libzl_hotpath_status(void);
zlhip_engine *libzl_hotpath_engine(void);
/* the millisecond clock behind the 30 ms / 100 ms rate limits and the positions model's time stamps (the reference reads
 * QDateTime::currentMSecsSinceEpoch(), ClipAudioSource.cpp:89,111,226,238): NULL = the wall clock.  A host that renders
 * faster or slower than real time (offline bounce, deterministic tests) installs its own. */
void libzl_hotpath_set_clock_ms(int64_t (*clock_ms)(void));
/* a clip from memory instead of a file (planar fp32, right == NULL for mono) */
ClipAudioSource *ClipAudioSource_newFromBuffer(const float *left, const float *right, int length, double sampleRate, const char *name);
/* the per-cycle seam: what SamplerChannel::process does for every channel (SamplerSynth.cpp:116-148).
 * out_left / out_right: [num_buses][nframes].  Afterwards the per-clip positions models are updated from
 * the voice reports and the progress / audio-level callbacks fire (ClipAudioSource.cpp:88-113,225-240). */
int  libzl_hotpath_process(uint32_t nframes, const zlhip_clock *clock, float *out_left, float *out_right);
/* The same cycle with the JackPassthrough clients behind the channels (JackPassthroughPrivate::process, JackPassthrough.cpp:45-115; clients
 * and their channels: MidiRouter.cpp:876-883 -- "GlobalPlayback" behind channel -1 = bus 1, "FXPassthrough-Channel<n>" behind channel n-1 = bus
 * n+1; a bus without a client in the reference gets one at its defaults): fan_out [num_buses][6][nframes] = dryL, dryR, fx1L, fx1R, fx2L, fx2R of
 * every bus, with the dry / wet / pan amounts and mute flags the JackPassthrough_set* calls stored -- read per cycle, no lock, no HIP call;
 * on the real-time cycle the resident kernel writes the rows from the registers that hold the mix.  fan_out == NULL: libzl_hotpath_process. */
int  libzl_hotpath_process_fanout(uint32_t nframes, const zlhip_clock *clock, float *out_left, float *out_right, float *fan_out);
/* The same cycle with this library's own transport: SyncTimerPrivate::process (SyncTimer.cpp:452-702) for the JACK cycle
 * [current_usecs, next_usecs) -- the steps of the 32768-step ring that fall due are played, their ClipCommands dispatched
 * with the playhead (:553-558), SetBpm commands applied, playhead and step clock rolled -- then every channel is rendered
 * with the clock SyncTimer's getters return (:990-1009), and, while the timer runs, the timer thread's tick
 * (hiResTimerCallback, :391-418) is taken once.  Arguments as jack_get_cycle_times returns them (SamplerSynth.cpp:128). */
int  libzl_hotpath_cycle(uint32_t nframes, uint64_t current_usecs, uint64_t next_usecs, float period_usecs, float *out_left, float *out_right);
int  libzl_hotpath_cycle_fanout(uint32_t nframes, uint64_t current_usecs, uint64_t next_usecs, float period_usecs, float *out_left, float *out_right,
                                float *fan_out /* as libzl_hotpath_process_fanout */);
/* play / stop / queue / timer calls are handed to the cycle through a bounded queue (4096 requests, FreshCommandStashSize of SyncTimer.cpp:252;
 * ClipAudioSource_stop(-3) alone is 12 of them) that only a cycle drains: how many requests were LOST because it was full -- no cycle ran
 * (before JACK starts, during libzl_hotpath_bounce_to_wav).  The callers return void as in libzl.h; a host that floods the queue reads this. */
uint64_t libzl_hotpath_dropped_requests(void);
/* Offline bounce of the running session to WAV files (BASELINE configs[4] from / to real files): what nblocks calls of
 * libzl_hotpath_cycle would render -- the library's own transport, JACK time start_usecs + k * round(1e6 * nframes / fs), commands
 * dispatched in the cycle their step falls due in -- rendered in batches through zlhip_bounce and written as one stereo WAV per sampler
 * channel, "<prefix>-channel_<bus>.wav": bits_per_sample 16 = the recorder's format (AudioLevels.cpp:53-58), 32 = float.  The
 * positions models and callbacks are not driven.  Returns 0 or a negative zlhip status. */
int  libzl_hotpath_bounce_to_wav(const char *prefix, int64_t nblocks, uint32_t nframes, uint64_t start_usecs, int bits_per_sample);
/* Extension: the parameters of a clip as the engine receives them at the next cycle -- among them the slice table that
 * ClipAudioSource_setSlices builds (ClipAudioSource.cpp:495-528; the reference exposes it as a Qt property only).  No device needed.
 * Returns 0, or -1 for a null argument. */
int  libzl_hotpath_clip_params(ClipAudioSource *c, zlhip_clip_params *out);
/* SyncTimer::scheduleClipCommand(command, delay) (SyncTimer.cpp:1011-1048): what ClipAudioSource_play / _stop call with delay 0;
 * command->clip is the zlhip clip id (ClipAudioSource_engineClip) */
void libzl_hotpath_schedule_clip_command(const zlhip_clip_command *command, uint64_t delay);
/* one more tick of the timer thread (hiResTimerCallback) before the next libzl_hotpath_cycle */
void libzl_hotpath_timer_tick(void);
/* SyncTimer::jackPlayhead / jackPlayheadUsecs / jackSubbeatLengthInMicroseconds (SyncTimer.cpp:990-1009) of the library's own
 * transport, in the playhead fields of *out */
int  libzl_hotpath_transport(zlhip_clock *out);
/* ClipAudioSourcePositionsModel read-outs of a clip (ClipAudioSourcePositionsModel.cpp:160-185) */
float  ClipAudioSource_peakGain(ClipAudioSource *c);
double ClipAudioSource_firstProgress(ClipAudioSource *c);
float  ClipAudioSource_volumeAbsolute(ClipAudioSource *c);                              /* ClipAudioSource.cpp:338-346 */
void   ClipAudioSource_setVolumeAbsolute(ClipAudioSource *c, float vol);                /* ClipAudioSource.cpp:328-336 */
int    ClipAudioSource_engineClip(ClipAudioSource *c);                                  /* zlhip clip id */
/* minimal RIFF/WAVE IO (decode side of SamplerSynthSound.cpp:28-59; record side of AudioLevels.cpp:35-119) */
int  libzl_wav_read(const char *path, float **left, float **right, int *length, double *sampleRate);  /* malloc'd planes; free with libzl_wav_free */
void libzl_wav_free(float *plane);
int  libzl_wav_write(const char *path, const float *left, const float *right, int length, double sampleRate, int bitsPerSample /* 16 or 32(float) */);
/* the same file from already interleaved frames: 16 bit = int16_t pairs as zlhip_bounce(ZLHIP_BOUNCE_PCM16_STEREO) delivers them per bus, 32 = float */
int  libzl_wav_write_interleaved(const char *path, const void *frames, int length, int channels, double sampleRate, int bitsPerSample);

#ifdef __cplusplus
}
#endif
#endif
