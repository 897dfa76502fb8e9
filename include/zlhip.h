/*
 * zlhip.h -- C-ABI of the MI355X sampler engine that replaces libzl's audio hot path.
 *
 * Drop-in seam (reference, paths under /root/reference/lib):
 *   SamplerChannel::process            SamplerSynth.cpp:116-148   -> zlhip_render / zlhip_render_batch
 *   SamplerChannel::handleCommand      SamplerSynth.cpp:187-230   -> zlhip_handle_command
 *   SamplerSynth::registerClip / SamplerSynthSound::loadSoundData
 *                                      SamplerSynth.cpp:285-295, SamplerSynthSound.cpp:28-59 -> zlhip_sound_upload
 *   ClipAudioSource getters read per block by the voice
 *                                      SamplerSynthVoice.cpp:189-196, ClipAudioSource.cpp:261-277,338-346,362,619,692
 *                                                                 -> zlhip_clip_set
 *   SamplerSynthVoice::process         SamplerSynthVoice.cpp:174-270 -> HIP kernels behind zlhip_render*
 *   positions-model report             SamplerSynthVoice.cpp:265-267 -> zlhip_voice_reports
 *   AudioLevels::timerCallback         AudioLevels.cpp:347-412    -> zlhip_levels_tick
 *   JackPassthroughPrivate::process    JackPassthrough.cpp:45-115 -> zlhip_passthrough_*
 *
 * Plain C: opaque handle, POD structs, raw pointers and sizes, int status codes.  No torch / Qt /
 * JUCE types.  All functions are thread-compatible (one caller at a time per engine), mirroring
 * the reference where each SamplerChannel is driven by one JACK thread.
 * The library has NO CPU render path: if no HIP device is usable, zlhip_engine_create fails with
 * ZLHIP_ERR_NO_DEVICE and nothing else can be called.
 */
#ifndef ZLHIP_H
#define ZLHIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: zlhip_config grew (rt_idle_timeout_us; struct_size tells the library which fields the caller knows); ZLHIP_MODE_HERMITE is
 *    the tap-weight form fixed in round 2 (it differs from the Horner form of ABI 1 in the last bits);
 *    new entry points: zlhip_bounce, zlhip_host_alloc/free, zlhip_bus_reduce_sum_scan, zlhip_levels_import_units,
 *    zlhip_sound_upload_device_on; zlhip_clip_set no longer waits for the device (the edit lands at the next render call).
 * 3: new entry points only (a caller built against 2 keeps working): zlhip_render_fanout (the JackPassthrough fan-out on the
 *    real-time cycle), zlhip_rt_residency, zlhip_rt_last_cycle; the resident real-time kernel takes any period (blocks longer than 256 frames too). */
#define ZLHIP_ABI_VERSION 3

/* status codes */
#define ZLHIP_OK                 0
#define ZLHIP_ERR_INVALID       -1   /* bad argument */
#define ZLHIP_ERR_NO_DEVICE     -2   /* no usable HIP device / kernels not loadable */
#define ZLHIP_ERR_HIP           -3   /* a HIP runtime call failed (see zlhip_last_error) */
#define ZLHIP_ERR_CAPACITY      -4   /* arena / table / batch capacity exceeded */
#define ZLHIP_ERR_STATE         -5   /* call not valid in the current state */

/* render modes: 0 reproduces the reference bit for bit (quirks Q1/Q2 of SURVEY.md section 0) */
#define ZLHIP_MODE_FAITHFUL      0u
#define ZLHIP_MODE_FIX_GAIN      1u  /* gain/envelope/volume scale the whole interpolated sample */
#define ZLHIP_MODE_FIX_DELAY     2u  /* frame f is written to out[f] instead of out[f+1] */
#define ZLHIP_MODE_HERMITE       4u  /* build-defined extension: 4-tap Catmull-Rom interpolation */

#define ZLHIP_MAX_SLICES         128
#define ZLHIP_BEAT_SUBDIVISIONS  96  /* SyncTimer.cpp:95 */

typedef struct zlhip_engine zlhip_engine;

typedef struct zlhip_config {
    uint32_t struct_size;            /* sizeof(zlhip_config), for ABI growth */
    int32_t  device;                 /* HIP device ordinal */
    int32_t  num_buses;              /* SamplerChannels (reference: 12, SamplerSynth.cpp:258) */
    int32_t  voices_per_bus;         /* voices per channel (reference: 8, SamplerSynth.cpp:23) */
    int32_t  max_frames;             /* largest nframes per block, 1 .. 4096 (any JACK period: 16, 32, 441, 480 ... -- a block runs on whole 64-lane waves) */
    int32_t  max_batch_blocks;       /* largest nblocks per zlhip_render_batch call */
    int32_t  max_sounds;             /* clip / sound table size */
    uint32_t mode;                   /* ZLHIP_MODE_* */
    double   playback_sample_rate;   /* jack_get_sample_rate, SamplerSynth.cpp:271-272 */
    uint64_t sound_arena_bytes;      /* HBM reserved for decoded sources */
    int32_t  voices_per_task;        /* voices summed sequentially by one wavefront (mix group); 0 = the whole bus,
                                        i.e. the reference's order.  Smaller groups = two-level order, more parallelism.
                                        (With 0, single real-time blocks of buses of >= 32 voices are rendered one voice
                                        per workgroup and added in voice order: the same order, bit for bit.) */
    int32_t  plan_window_blocks;     /* blocks planned per window (planning of window i+1 overlaps rendering of window i);
                                        0 = automatic: 512 Ki frames at 1024 voices (2048 blocks of 256), proportionally
                                        more frames for fewer voices (up to 16 Mi), never more than max_batch_blocks */
    int32_t  rt_idle_timeout_us;     /* (ABI 2) how long the resident real-time kernel behind zlhip_render stays on the device
                                        without a cycle before it leaves (it is started again by the next cycle); 0 = 200 000.
                                        A host that makes device-synchronising HIP calls of its own (hipFree, hipDeviceSynchronize)
                                        waits at most this long behind an idle engine; calls made through this library do not wait. */
    uint64_t sound_arena_max_bytes;  /* (ABI 2) sound_arena_bytes is what the engine reserves at creation; when a source no longer fits it
                                        allocates further segments of at least that size, up to this total (0 = no limit but the
                                        device's memory; = sound_arena_bytes: a fixed arena, uploads beyond it fail with
                                        ZLHIP_ERR_CAPACITY until clips are released) */
} zlhip_config;

/* clock inputs of one block: JACK cycle times + SyncTimer playhead getters
 * (SamplerSynth.cpp:128, SyncTimer.cpp:990-1009) */
typedef struct zlhip_clock {
    uint64_t current_usecs;
    uint64_t next_usecs;
    uint64_t jack_playhead;
    uint64_t jack_playhead_usecs;
    uint64_t jack_subbeat_length_usecs;
} zlhip_clock;

/* snapshot of the ClipAudioSource fields the voice reads (ClipAudioSource.cpp:63-82) */
typedef struct zlhip_clip_params {
    float   start_position_seconds;
    float   length_seconds;
    float   length_in_beats;
    float   volume_absolute;         /* tracktion fader position in [0,1], taken as an input */
    float   pan;
    float   duration_seconds;        /* getDuration() */
    float   adsr_attack, adsr_decay, adsr_sustain, adsr_release;
    int32_t root_note;
    int32_t num_slice_positions;
    double  slice_positions[ZLHIP_MAX_SLICES];
} zlhip_clip_params;

/* ClipCommand (ClipCommand.h:11-32); `clip` is the id returned by zlhip_sound_upload */
typedef struct zlhip_clip_command {
    int32_t clip;
    int32_t midi_note;
    int32_t midi_channel;
    int32_t start_playback, stop_playback;
    int32_t change_slice, slice;
    int32_t change_looping, looping;
    int32_t change_pitch;   float pitch_change;
    int32_t change_speed;   float speed_ratio;
    int32_t change_gain_db; float gain_db;
    int32_t change_volume;  float volume;
} zlhip_clip_command;

/* what SamplerSynthVoice.cpp:265-267 hands to ClipAudioSourcePositionsModel, per voice slot */
typedef struct zlhip_voice_report {
    int32_t playing;                 /* voice->isPlaying after the render */
    int32_t valid;                   /* 1 if (gain, progress) were reported for the last rendered block */
    float   gain;                    /* peakGain * 0.5f */
    float   progress;                /* sourceSamplePosition / sourceSampleLength */
    int32_t clip;                    /* clip id the voice plays, -1 if none */
    int32_t reserved;
    double  source_sample_position;  /* d->sourceSamplePosition after the render (parity checks) */
} zlhip_voice_report;

/* AudioLevels per-channel meter state + outputs (AudioLevels.cpp:359-398) */
typedef struct zlhip_levels {
    int32_t peak_a, peak_b;                          /* integer peaks after decay + scan */
    float   peak_a_hold_signal, peak_b_hold_signal;  /* 0.9x hold (playback channel) */
    float   peak_db_a, peak_db_b, combined_db, hold_db_a, hold_db_b;
    float   rms_a, rms_b;                            /* build-defined extension: RMS of the scanned block */
} zlhip_levels;

/* JackPassthrough parameters (JackPassthrough.cpp:27-31) */
typedef struct zlhip_passthrough_params {
    float   dry_amount, wet_fx1_amount, wet_fx2_amount, pan_amount;
    int32_t muted;
} zlhip_passthrough_params;

/* profiling counters of the last zlhip_render_batch (HIP events on the engine's stream) */
typedef struct zlhip_timings {
    float plan_ms;        /* planning (K0+K1+K1c) not hidden behind rendering: start of the call on its stream to first K2 */
    float render_ms;      /* gather-interp-mix kernel (K2), the dominant one: sum over the call's launches */
    float finalize_ms;    /* total - render - plan: K3 (bus reduce + levels), reports, gaps between launches */
    float total_ms;       /* first launch to last completion */
    uint64_t source_bytes;   /* algorithmic source bytes of the batch: sum (ceil(N*ratio)+taps-1)*ch*4 */
    uint64_t slow_blocks;    /* voice-blocks that needed the per-frame control path */
    uint64_t active_voice_frames; /* voice-samples rendered */
    int32_t  render_launches;  /* K2 launches of the call (one per plan window); render_ms is their sum */
    int32_t  reserved;
} zlhip_timings;

/* ---- lifecycle ---------------------------------------------------------------------------- */
int  zlhip_abi_version(void);
void zlhip_config_default(zlhip_config *cfg);
int  zlhip_engine_create(const zlhip_config *cfg, zlhip_engine **out);
void zlhip_engine_destroy(zlhip_engine *e);
const char *zlhip_last_error(const zlhip_engine *e);          /* borrowed, valid until the next call */
const char *zlhip_strerror(int status);

/* ---- sounds and clip parameters ----------------------------------------------------------- */
/* Upload a decoded source (planar fp32, right == NULL for mono) from host memory; returns its id
 * in *out_id.  The id doubles as the clip id (one SamplerSynthSound per ClipAudioSource). */
int zlhip_sound_upload(zlhip_engine *e, const float *left, const float *right, int32_t length,
                       double sample_rate, int32_t *out_id);
/* Same, but left/right are DEVICE pointers on the engine's device (no PCIe transfer).  The call waits for the device
 * (hipDeviceSynchronize) before it reads them: whatever stream produced the planes, they are complete. */
int zlhip_sound_upload_device(zlhip_engine *e, const float *left_dev, const float *right_dev, int32_t length,
                              double sample_rate, int32_t *out_id);
/* Same, with the stream the planes were produced on (hipStream_t; NULL = the null stream): the engine waits for that stream only
 * (an event), not for the device. */
int zlhip_sound_upload_device_on(zlhip_engine *e, const float *left_dev, const float *right_dev, int32_t length,
                                 double sample_rate, void *producer_stream, int32_t *out_id);
int zlhip_sound_release(zlhip_engine *e, int32_t id);          /* SamplerSynth::unregisterClip */
void zlhip_clip_params_default(zlhip_clip_params *p, float duration_seconds);   /* ClipAudioSource ctor defaults */
/* The parameters a voice reads per block (SamplerSynthVoice.cpp:189-196).  Host-only and wait-free: the edit is recorded and the
 * device applies it at the start of the next render call / real-time cycle, the block boundary at which the reference's voices
 * would read it; commands handled after the call see the new values (startNote, SamplerSynthVoice.cpp:115-121).  The resident
 * real-time kernel keeps running. */
int zlhip_clip_set(zlhip_engine *e, int32_t id, const zlhip_clip_params *p);

/* ---- commands ------------------------------------------------------------------------------ */
void zlhip_clip_command_clear(zlhip_clip_command *c);           /* ClipCommand.h:74-91 */
/* SamplerChannel::handleCommand for the bus whose midi channel matches (bus b has midi channel
 * b - 2, SamplerSynth.cpp:270).  Returns 1 if a voice took / merged the command, 0 if it was
 * dropped (no free voice, as in the reference), < 0 on error. */
int zlhip_handle_command(zlhip_engine *e, const zlhip_clip_command *cmd, uint64_t current_tick);
/* A block's worth of commands in one call (SyncTimer dispatches every command that is due in a cycle,
 * SyncTimer.cpp:553-558): handled in array order like the channel's command ring; they reach the device as ONE
 * voice-table update (K0) before the next rendered block.  taken[i] (optional) receives what zlhip_handle_command
 * would have returned for command i; the return value is their sum, < 0 on error. */
int zlhip_handle_commands(zlhip_engine *e, const zlhip_clip_command *cmds, int32_t count, uint64_t current_tick, int32_t *taken);
/* The same, and voices[i] (optional) receives the voice -- bus * voices_per_bus + slot, the index of zlhip_voice_reports -- that
 * command i STARTED (startNote, SamplerSynthVoice.cpp:110-144), -1 if it started none.  For a host that keeps per-voice state of its own
 * in the order the reference creates it (the libzl layer's playback-positions rows: created in command order at dispatch, :129). */
int zlhip_handle_commands_voices(zlhip_engine *e, const zlhip_clip_command *cmds, int32_t count, uint64_t current_tick, int32_t *taken, int32_t *voices);
/* SamplerSynth::setChannelEnabled(channel, enabled) for bus = channel + 2 (SamplerSynth.cpp:343-351; SamplerChannel::process :116-123): a
 * disabled bus still takes its commands, but its voices are not processed -- they keep position, envelope and clock state and go on
 * from there when the bus is enabled again -- and report no progress.  The bus renders silence meanwhile (the reference leaves its JACK
 * port buffers untouched).  Host-only, takes effect with the next rendered block. */
int zlhip_bus_set_enabled(zlhip_engine *e, int32_t bus, int enabled);
/* Same, addressed to an explicit voice slot of a bus (bypasses first-free allocation; used to
 * build large synthetic scenes deterministically). */
int zlhip_start_voice(zlhip_engine *e, int32_t bus, int32_t slot, const zlhip_clip_command *cmd, uint64_t current_tick);
/* The voice-level calls behind the JUCE SynthesiserVoice surface of SamplerSynthVoice (include/zlhip_voice_adapter.h):
 * stopNote(velocity, allowTailOff) (SamplerSynthVoice.cpp:146-169), setCurrentCommand on a playing voice (:58-93) and
 * the isPlaying flag (SamplerSynthVoice.h:31).  They apply before the next rendered block.  Return 1 / 0 (the voice
 * was playing / was not), < 0 on error. */
int zlhip_stop_voice(zlhip_engine *e, int32_t bus, int32_t slot, int allow_tail_off);
int zlhip_update_voice(zlhip_engine *e, int32_t bus, int32_t slot, const zlhip_clip_command *cmd);
int zlhip_voice_is_playing(zlhip_engine *e, int32_t bus, int32_t slot);

/* ---- render -------------------------------------------------------------------------------- */
/* One real-time block: renders nframes for every bus and delivers the mix to host memory.
 * out_left/out_right: [num_buses][nframes] each (host).  Synchronous.
 * Page-locked buffers (zlhip_host_alloc, or the caller's own hipHostMalloc / hipHostRegister -- e.g. of its JACK port area) are written by
 * the kernels DIRECTLY: no copy on the host behind the cycle (1.6 us for 12 buses x 256 frames; 5 us more with the fan-out's 72 KB).
 * Any other memory is served through the engine's staging rows and a copy.  (ZL_RT_DIRECT_OUT=0: always staged.) */
int zlhip_render(zlhip_engine *e, int32_t nframes, const zlhip_clock *clock, float *out_left, float *out_right);
/* The same cycle, and the JackPassthrough client behind every bus with it (JackPassthroughPrivate::process, JackPassthrough.cpp:45-115;
 * the client is the next node after a SamplerSynth channel in the reference's JACK graph): the three output pairs of every bus are
 * computed from the registers that hold the finished mix and written to host memory next to it -- by the resident real-time kernel
 * where zlhip_render uses it, else by the launched kernels.
 *   fan_params  host [num_buses]: this cycle's dry / wetFx1 / wetFx2 / pan amounts and mute flags.  Taken per cycle: a changed value
 *               costs no HIP call and does not disturb the resident kernel (it re-reads the table when it differs from the last cycle's)
 *   fan_out     host [num_buses][6][nframes] = dryL, dryR, fx1L, fx1R, fx2L, fx2R of every bus (the order of zlhip_passthrough_process)
 * Bit-identical to zlhip_render followed by the passthrough of its output.  fan_params == fan_out == NULL: zlhip_render. */
int zlhip_render_fanout(zlhip_engine *e, int32_t nframes, const zlhip_clock *clock, float *out_left, float *out_right,
                        const zlhip_passthrough_params *fan_params, float *fan_out);
/* Throughput mode: nblocks consecutive blocks in one pass.  clocks: host [nblocks].
 * bus_out_dev: DEVICE buffer laid out [num_buses][2][nblocks*nframes] fp32, or NULL to use the
 * engine's internal buffer (readable with zlhip_read_bus).  stream: hipStream_t or NULL for the
 * engine's own stream.  Asynchronous; zlhip_synchronize waits. */
int zlhip_render_batch(zlhip_engine *e, int32_t nblocks, int32_t nframes, const zlhip_clock *clocks,
                       float *bus_out_dev, void *stream);
int zlhip_synchronize(zlhip_engine *e);
/* copy the internal bus buffer of the last batch to host: out [num_buses][2][nblocks*nframes] */
int zlhip_read_bus(zlhip_engine *e, float *out, size_t out_floats);
/* per-voice reports of the last rendered block: out [num_buses*voices_per_bus] */
int zlhip_voice_reports(zlhip_engine *e, zlhip_voice_report *out, int32_t count);
/* debug: per-frame (int)sourceSamplePosition of every voice for the blocks of the NEXT batches,
 * kept on device and read back with zlhip_debug_read_trace: out [nblocks][voices][nframes] int32 */
int zlhip_debug_enable_trace(zlhip_engine *e, int enable);   /* bit 0: trace; bits 1, 2: planner test hooks */
int zlhip_debug_read_trace(zlhip_engine *e, int32_t *out, size_t out_ints);

/* ---- levels -------------------------------------------------------------------------------- */
/* One AudioLevels timer tick for every bus over block `block_index` of the last batch
 * (-1 = the last block; the reference only ever sees the most recent block, AudioLevels.cpp:361-384).
 * out: [num_buses].  with_hold_bus: index of the bus that keeps the 0.9x hold (reference:
 * channel index 1, AudioLevels.cpp:391-398), -1 for none. */
int zlhip_levels_tick(zlhip_engine *e, int32_t block_index, int32_t with_hold_bus, zlhip_levels *out);
/* raw per-block integer peaks of the last batch: out [nblocks][num_buses][2] */
int zlhip_block_peaks(zlhip_engine *e, int32_t *out, size_t out_ints);
/* levels of an arbitrary DEVICE bus buffer [num_buses][2][nblocks*nframes] (e.g. the result of a
 * multi-GPU reduce): recomputes the per-block peaks the next zlhip_levels_tick will use */
int zlhip_levels_scan_device(zlhip_engine *e, const float *bus_dev, int32_t nblocks, int32_t nframes, void *stream);

/* ---- multi-GPU exchange (a bus that spans GPUs) ----------------------------------------------------
 * The only coupling between voices is the per-bus sum of SamplerChannel::process (SamplerSynth.cpp:134-140).  When the
 * voices of a bus live on several GPUs, every rank renders a partial bus [num_buses][2][nblocks*nframes] and the ranks
 * exchange it piecewise (all-to-all over the xGMI mesh: rank r receives piece r of every rank's partial bus; the transport is
 * the host's, e.g. RCCL).  A piece is a run of whole UNITS, unit u = (bus * 2 + channel) * nblocks + block = nframes
 * consecutive floats of the bus buffer.
 *
 * zlhip_bus_reduce_sum_scan: one kernel, run by every rank on the pieces it received.  Sums them in piece (= rank) order,
 * ((0 + p0) + p1) + ... per sample -- deterministic, the same bits for any number of ranks, and equal to the single-GPU
 * result with voices_per_task = voices per rank -- writes the reduced piece and, in the same pass, the AudioLevels scan of
 * every unit (integer peak, AudioLevels.cpp:361-383; sum of squares of the RMS extension).
 *   pieces_dev      DEVICE [npieces] pieces of units * nframes floats, piece_stride_floats apart (the all-to-all receive buffer)
 *   sum_out_dev     DEVICE [units * nframes]
 *   levels_out_dev  DEVICE [units]
 * zlhip_levels_import_units: on the rank that meters (the root, after gathering the unit levels of all pieces in unit order
 * [num_buses * 2 * nblocks]): makes them the block levels the next zlhip_levels_tick / zlhip_block_peaks read. */
typedef struct zlhip_unit_levels { int32_t peak; float sumsq; } zlhip_unit_levels;
int zlhip_bus_reduce_sum_scan(zlhip_engine *e, const float *pieces_dev, int32_t npieces, int64_t piece_stride_floats, int64_t units,
                              int32_t nframes, float *sum_out_dev, zlhip_unit_levels *levels_out_dev, void *stream);
int zlhip_levels_import_units(zlhip_engine *e, const zlhip_unit_levels *units_dev, int32_t nblocks, int32_t nframes, void *stream);

/* ---- JackPassthrough fan-out ---------------------------------------------------------------- */
void zlhip_passthrough_params_default(zlhip_passthrough_params *p);
/* in_dev: [num_buses][2][frames]; out_dev: [num_buses][6][frames] = dryL,dryR,fx1L,fx1R,fx2L,fx2R.
 * params: host [num_buses]. */
int zlhip_passthrough_process(zlhip_engine *e, const zlhip_passthrough_params *params, const float *in_dev,
                              float *out_dev, int64_t frames, void *stream);

/* zlhip_render_batch with the fan-out (JackPassthroughPrivate::process, JackPassthrough.cpp:45-115) fused into the bus
 * write: the JackPassthrough client is the next node after a
 * SamplerSynth bus in the reference's graph, and its three output pairs are computed from the registers that hold the
 * finished mix (24 more bytes written per bus frame, no second pass over the bus).  fan_params: host [num_buses];
 * fan_out_dev: DEVICE [num_buses][6][nblocks*nframes] in the order of zlhip_passthrough_process.  Bit-identical to
 * zlhip_render_batch followed by zlhip_passthrough_process.  Both NULL = zlhip_render_batch. */
int zlhip_render_batch_fanout(zlhip_engine *e, int32_t nblocks, int32_t nframes, const zlhip_clock *clocks,
                              float *bus_out_dev, const zlhip_passthrough_params *fan_params, float *fan_out_dev,
                              void *stream);

/* ---- offline bounce (BASELINE configs[4]; the recorder side of the bus, AudioLevels.cpp:35-119) ----------------------------
 * Renders nblocks consecutive blocks exactly like consecutive zlhip_render_batch calls (voice state, levels and reports carry on;
 * afterwards zlhip_levels_tick / zlhip_block_peaks / zlhip_voice_reports see the last chunk) and delivers every bus to HOST memory.
 * The bounce is cut into chunks of sub_blocks blocks (0 = max_batch_blocks, the longest call the engine takes), each ONE render call
 * whose plan windows pipeline as in a device-resident batch; every window is handed to the copy engine as soon as its render kernel
 * has finished (16-bit: converted first), so PCIe runs next to the rendering of the following windows.  Synchronous.
 *   clocks    host [nblocks]
 *   host_out  ZLHIP_BOUNCE_F32_PLANAR:  float   [num_buses][2][nblocks*nframes]     (the layout of zlhip_render_batch)
 *             ZLHIP_BOUNCE_PCM16_STEREO: int16_t [num_buses][nblocks*nframes][2]     (the data chunk of one 16-bit stereo WAV per
 *             bus, converted on the GPU as the reference's recorder converts -- juce::WavAudioFormat 16 bit, AudioLevels.cpp:53-58;
 *             restated, JUCE version unpinned: clamp, x 0x7fffffff in double, round to nearest even, upper 16 bits; half the PCIe bytes)
 *             Page-locked memory (zlhip_host_alloc, or the caller's hipHostMalloc / hipHostRegister) gives the full PCIe rate;
 *             pageable memory works through the runtime's staging copies.
 * zlhip_host_alloc / zlhip_host_free: page-locked host memory for callers that do not link HIP. */
enum { ZLHIP_BOUNCE_F32_PLANAR = 0, ZLHIP_BOUNCE_PCM16_STEREO = 1 };
int  zlhip_bounce(zlhip_engine *e, int64_t nblocks, int32_t nframes, const zlhip_clock *clocks, void *host_out, int32_t format,
                  int32_t sub_blocks);
int  zlhip_host_alloc(size_t bytes, void **out);
void zlhip_host_free(void *p);

/* ---- introspection / measurement ------------------------------------------------------------ */
int zlhip_set_profiling(zlhip_engine *e, int enable);
int zlhip_last_timings(zlhip_engine *e, zlhip_timings *out);
/* Sums of the timings of every profiled zlhip_render_batch call since the last reset (and their number): lets a caller
 * queue calls back to back -- consecutive calls pipeline, the planning of call i+1 overlaps the rendering of call i --
 * and read the kernel times once at the end.  Waits for outstanding calls. */
int zlhip_profile_totals(zlhip_engine *e, zlhip_timings *totals, int32_t *calls, int reset);
float *zlhip_bus_device_ptr(zlhip_engine *e);                   /* internal [B][2][Kmax*Nmax] buffer */
/* the resident real-time kernel behind zlhip_render: how many times it was launched, how many cycles it rendered (a parameter edit,
 * a command or a quiet spell shorter than the idle timeout do not relaunch it; a batch, an upload or a block-size change do) */
int zlhip_rt_stats(zlhip_engine *e, uint64_t *kernel_starts, uint64_t *cycles_rendered);
/* Is this engine's resident kernel on the device right now (*resident), and which share of the device's resident-workgroup capacity
 * does it take (*share, 0..1)?  The resident kernels of ALL engines of a process together take at most three quarters of a device; an
 * engine that does not fit next to the others renders its cycles with launches (same results) until there is room. */
int zlhip_rt_residency(zlhip_engine *e, int32_t *resident, double *share);
/* HBM the engine allocated at creation: everything (source arena, voice / plan records, control pool, bus, levels) and the
 * arena's share of it */
/* Where the LAST zlhip_render / zlhip_render_fanout cycle spent its time, seen from the calling thread (engines created with ZL_RT_TRACE=1
 * in the environment; ZL_RT_TRACE_SLOW_US=<n> also reports every cycle longer than n microseconds on stderr).  total = before_post (command
 * upload, a restart of the resident kernel; for launches: the launch calls) + wait (for the device) + after (copies into the caller's
 * buffers).  max_poll_gap_us: the longest time between two polls of the waiting thread -- a gap of milliseconds means the THREAD was
 * off its core (scheduler, cgroup quota), not that the device was late; involuntary_switches: getrusage(RUSAGE_THREAD) over the cycle;
 * device_us: the resident kernel's own stage times for the cycle (with ZL_RT_STAMPS=1, else 0). */
typedef struct zlhip_rt_cycle_trace {
    uint64_t cycle;
    int32_t  resident, reserved;
    double   total_us, before_post_us, wait_us, after_us, max_poll_gap_us, device_us;
    int64_t  involuntary_switches;
} zlhip_rt_cycle_trace;
int zlhip_rt_last_cycle(zlhip_engine *e, zlhip_rt_cycle_trace *out);
int zlhip_memory_bytes(zlhip_engine *e, uint64_t *total_device_bytes, uint64_t *arena_bytes);
int zlhip_device_name(zlhip_engine *e, char *buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif
