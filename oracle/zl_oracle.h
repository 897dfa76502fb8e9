/*
 * zl_oracle.h -- CPU restatement of libzl's sampler hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the checker for the HIP engine in libzl_amd/: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call it.  The product path
 * (libzl_amd/, include/) never links or falls back to anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned".  The reference (zynthbox/libzl @ v1) cannot be compiled in
 * this image (needs JUCE, tracktion_engine [empty submodule], JACK, Qt >= 5.11, rtmidi) and its
 * test/ directory holds no golden vectors, known-answer tests or audio fixtures for this path.
 * The oracle is therefore a line-by-line behavioural restatement of the reference source text,
 * cross-checked against an independently written numpy restatement (oracle/np_restatement.py)
 * and hand-derived known answers (tests/test_oracle_kat.py).  juce::ADSR is third-party code
 * absent from /root/reference (un-vendored submodule tracktion_engine/modules/juce, version
 * unpinned); its published JUCE 6 algorithm is restated in zlo_adsr_*.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference/lib).
 * Compile with -ffp-contract=off: the oracle DEFINES the fp32 rounding sequence (one IEEE
 * operation per C operator, left to right as the reference's expressions associate).
 */
#ifndef ZL_ORACLE_H
#define ZL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- mode flags (0 = reference-faithful) -------------------------------------------------- */
#define ZLO_MODE_FAITHFUL        0u
#define ZLO_MODE_FIX_GAIN        1u /* undo quirk Q1: gain*env*volume scale the whole interpolated sample */
#define ZLO_MODE_FIX_DELAY       2u /* undo quirk Q2: frame f lands in out[f], last frame kept */
#define ZLO_MODE_HERMITE         4u /* build-defined extension: 4-tap Catmull-Rom (absent in reference) */

#define ZLO_MAX_SLICES      128
#define ZLO_POSITION_COUNT  32   /* ClipAudioSourcePositionsModel.cpp:5 */
#define ZLO_BEAT_SUBDIVISIONS 96 /* SyncTimer.cpp:95 */

/* ---- juce::ADSR (JUCE 6 juce_audio_basics/utilities/juce_ADSR.h; call sites
 *      SamplerSynthVoice.cpp:38,134-137,150,155,201,253,258) ------------------------------- */
enum { ZLO_ADSR_IDLE = 0, ZLO_ADSR_ATTACK = 1, ZLO_ADSR_DECAY = 2, ZLO_ADSR_SUSTAIN = 3, ZLO_ADSR_RELEASE = 4 };

typedef struct zlo_adsr_params { float attack, decay, sustain, release; } zlo_adsr_params;

typedef struct zlo_adsr {
    zlo_adsr_params p;
    double sampleRate;
    float  env;
    int32_t state;
    float  attackRate, decayRate, releaseRate;
} zlo_adsr;

void  zlo_adsr_init(zlo_adsr *a);                       /* defaults A=.1 D=.1 S=1 R=.1, sr 44100, idle */
void  zlo_adsr_default_params(zlo_adsr_params *p);
void  zlo_adsr_set_sample_rate(zlo_adsr *a, double sr);
void  zlo_adsr_set_parameters(zlo_adsr *a, const zlo_adsr_params *p);
void  zlo_adsr_reset(zlo_adsr *a);
void  zlo_adsr_note_on(zlo_adsr *a);
void  zlo_adsr_note_off(zlo_adsr *a);
float zlo_adsr_next(zlo_adsr *a);
int   zlo_adsr_is_active(const zlo_adsr *a);

/* ---- SamplerSynthSound (SamplerSynthSound.cpp:41-49,96-114) -------------------------------- */
typedef struct zlo_sound {
    const float *L;        /* channel 0, planar fp32 [length] */
    const float *R;        /* channel 1 or NULL for mono sources */
    int32_t length;
    int32_t valid;
    double  sampleRate;
} zlo_sound;

/* ---- ClipAudioSourcePositionsModel (ClipAudioSourcePositionsModel.cpp:7-28) ---------------- */
typedef struct zlo_position { int64_t id; float progress; float gain; int64_t lastUpdated; } zlo_position;
typedef struct zlo_positions {
    zlo_position pos[ZLO_POSITION_COUNT];
    int32_t updatePeakGain;
    float   peakGain;
} zlo_positions;

void    zlo_positions_init(zlo_positions *m);
int64_t zlo_positions_create(zlo_positions *m, float initialProgress, int64_t now_ms);
void    zlo_positions_set_gain_and_progress(zlo_positions *m, int64_t id, float gain, float progress, int64_t now_ms);
void    zlo_positions_remove(zlo_positions *m, int64_t id, int64_t now_ms);
float   zlo_positions_peak_gain(zlo_positions *m);
double  zlo_positions_first_progress(const zlo_positions *m);
int     zlo_positions_cleanup(zlo_positions *m, int64_t now_ms);

/* ---- ClipAudioSource: the parameters the voice reads (ClipAudioSource.cpp:63-82) ----------- */
typedef struct zlo_clip {
    float   startPositionInSeconds;   /* :63 */
    float   lengthInSeconds;          /* :64 */
    float   lengthInBeats;            /* :65 */
    float   volumeAbsolute;           /* :66 ; taken as an input in [0,1] (tracktion fader curve is out of scope) */
    float   pan;                      /* :69 */
    float   duration;                 /* getDuration() :367 = edit length in seconds (float) */
    int32_t rootNote;                 /* :81 */
    int32_t slices;                   /* :75 */
    int32_t nSlicePositions;          /* length of slicePositionsCache :77 */
    int32_t sliceBaseMidiNote;        /* :78 */
    int32_t keyZoneStart, keyZoneEnd; /* :79-80 */
    int32_t id;
    double  slicePositions[ZLO_MAX_SLICES];
    zlo_adsr adsr;                    /* :82 ; only its Parameters are consumed by voices */
    zlo_positions positions;          /* :73 */
} zlo_clip;

void  zlo_clip_init(zlo_clip *c, float durationSeconds, double sourceSampleRate);   /* ctor :135-205 */
float zlo_clip_get_start_position(const zlo_clip *c, int slice);                    /* :261-268 */
float zlo_clip_get_stop_position(const zlo_clip *c, int slice);                     /* :270-277 */
void  zlo_clip_set_start_position(zlo_clip *c, float startPositionInSeconds);       /* :255-259 */
void  zlo_clip_set_length(zlo_clip *c, float beat, int bpm);                        /* :352-360 */
void  zlo_clip_set_volume_absolute(zlo_clip *c, float vol);                         /* :328-336 */
void  zlo_clip_set_pan(zlo_clip *c, float pan);                                     /* :623-629 */
void  zlo_clip_set_slices(zlo_clip *c, int slices);                                 /* :495-528 */
int   zlo_clip_slice_for_midi_note(const zlo_clip *c, int midiNote);                /* :575-578 */
void  zlo_clip_set_adsr_attack(zlo_clip *c, float v);                               /* :636-643 (Q13) */
void  zlo_clip_set_adsr_decay(zlo_clip *c, float v);                                /* :650-657 */
void  zlo_clip_set_adsr_sustain(zlo_clip *c, float v);                              /* :664-671 */
void  zlo_clip_set_adsr_release(zlo_clip *c, float v);                              /* :678-685 */
float zlo_subbeat_count_to_seconds(uint64_t bpm, uint64_t subbeats);                /* SyncTimer.cpp:180-183,936-939 */

/* ---- per-clip level / progress chain (ClipAudioSource.cpp:68-69,84-113,225-240) ----
 * The meter state of ClipAudioSource::Private and the two rate-limited notifiers.  `now_ms` stands for
 * QDateTime::currentMSecsSinceEpoch().  Each returns 1 when the reference would have called the C callback and
 * stores the callback's float argument in *value.  juce::Decibels (third-party, restated from the public JUCE
 * source) is a template on its argument type: float for peakGain() (:92), double for :98 and :101. */
typedef struct zlo_clip_meter {
    double  currentLeveldB, prevLeveldB;      /* :68-69, both -400 */
    double  firstPositionProgress;            /* :85 */
    int64_t nextPositionUpdateTime;           /* :84 */
    int64_t nextGainUpdateTime;               /* :87 */
} zlo_clip_meter;
void zlo_clip_meter_init(zlo_clip_meter *m);
int  zlo_sync_audio_level(zlo_clip_meter *m, zlo_clip *clip, int64_t now_ms, float *value);                 /* :88-113 */
int  zlo_sync_progress(zlo_clip_meter *m, const zlo_clip *clip, int has_callback, int64_t now_ms, float *value);   /* :225-240 */

/* ---- ClipCommand (ClipCommand.h:11-39) ---------------------------------------------------- */
typedef struct zlo_clip_command {
    int32_t clip;            /* index of the clip (stands for the ClipAudioSource* identity) */
    int32_t midiNote;
    int32_t midiChannel;
    int32_t startPlayback, stopPlayback;
    int32_t changeSlice, slice;
    int32_t changeLooping, looping;
    int32_t changePitch;  float pitchChange;
    int32_t changeSpeed;  float speedRatio;
    int32_t changeGainDb; float gainDb;
    int32_t changeVolume; float volume;
} zlo_clip_command;

void zlo_clip_command_clear(zlo_clip_command *c);                                   /* ClipCommand.h:74-91 + ctor defaults */
int  zlo_clip_command_equivalent(const zlo_clip_command *a, const zlo_clip_command *b); /* :33-39 */

/* ---- SyncTimer: the step ring that carries ClipCommands to SamplerSynth -------
 * SyncTimer.cpp:43-79 (StepData), :364-378 (delayedStep), :391-418 (hiResTimerCallback), :452-702 (process),
 * :815-860 (queueClipToStart/StopOnChannel), :870-925 (start / stop), :954-975 (setBpm), :990-1009 (getters),
 * :1011-1048 (scheduleClipCommand).  MIDI buffers, the jack transport position, timer commands other than SetBpm and
 * the timer thread itself (its tick is zlo_sync_timer_callback) are outside the path. */
#define ZLO_STEP_RING_COUNT 32768        /* SyncTimer.cpp:253 */
typedef struct zlo_step {                /* StepData, :43-79 */
    zlo_clip_command *clipCommands; int32_t nClipCommands, capClipCommands;
    int32_t *bpmCommands;           int32_t nBpmCommands, capBpmCommands;    /* TimerCommand::SetBpmOperation parameters */
    int32_t played;                      /* starts true (:78) */
} zlo_step;
typedef struct zlo_dispatch { zlo_clip_command cmd; uint64_t tick; } zlo_dispatch;   /* samplerSynth->handleClipCommand(cmd, jackPlayhead), :553-558 */
typedef struct zlo_sync_timer {
    zlo_step *stepRing;                  /* [ZLO_STEP_RING_COUNT] */
    uint64_t stepReadHead;               /* index of *stepReadHead */
    uint64_t stepNextPlaybackPosition;
    uint64_t bpm;                        /* SyncTimerThread::bpm, :250 */
    int32_t  threadPaused;               /* SyncTimerThread::paused, :236 */
    int32_t  isPaused;                   /* SyncTimerPrivate::isPaused, :440,750-752 */
    uint64_t jackPlayhead;
    double   jackPlayheadBpm;
    uint64_t jackNextPlaybackPosition;
    uint64_t jackSubbeatLengthInMicroseconds;
    uint64_t jackLatency;
    uint64_t scheduleAheadAmount;
    uint64_t cumulativeBeat;
    int32_t  beat;
    uint64_t stepReadHeadOnStart;
} zlo_sync_timer;

zlo_sync_timer *zlo_sync_timer_new(void);
void     zlo_sync_timer_free(zlo_sync_timer *t);
/* the merge of scheduleClipCommand on one step's list (:1014-1047): returns 1 if `command` was appended, 0 if folded into
 * an equivalent command of the step.  list must have room for one more entry. */
int      zlo_step_schedule(zlo_clip_command *list, int32_t *n, const zlo_clip_command *command);
void     zlo_schedule_clip_command(zlo_sync_timer *t, const zlo_clip_command *command, uint64_t delay);   /* :1011-1048 */
void     zlo_sync_timer_set_latency(zlo_sync_timer *t, uint32_t bufferSize, double sampleRate);           /* :730-741,770-771 */
void     zlo_sync_timer_set_bpm(zlo_sync_timer *t, uint64_t bpm);                                         /* :954-975 */
void     zlo_sync_timer_start(zlo_sync_timer *t, int bpm);                                                /* :870-879 */
void     zlo_sync_timer_stop(zlo_sync_timer *t);                                                          /* :881-925 */
void     zlo_sync_timer_callback(zlo_sync_timer *t);                                                      /* hiResTimerCallback, :391-418 */
void     zlo_sync_timer_queue_clip_to_start_on_channel(zlo_sync_timer *t, int32_t clip, int midiChannel); /* :815-832 */
void     zlo_sync_timer_queue_clip_to_stop_on_channel(zlo_sync_timer *t, int32_t clip, int midiChannel);  /* :834-860 */
/* process (:452-702): plays the steps due in [current_usecs, next_usecs); their ClipCommands go to out[0 .. return value)
 * (at most max_out) with the playhead they are dispatched with */
int32_t  zlo_sync_timer_process(zlo_sync_timer *t, uint32_t nframes, uint64_t current_usecs, uint64_t next_usecs, float period_usecs,
                                zlo_dispatch *out, int32_t max_out);
uint64_t zlo_sync_timer_jack_playhead(const zlo_sync_timer *t);            /* :990-996 */
uint64_t zlo_sync_timer_jack_playhead_usecs(const zlo_sync_timer *t);      /* :998-1004 */
uint64_t zlo_sync_timer_jack_subbeat_length_usecs(const zlo_sync_timer *t);/* :1006-1009 */

/* ---- clock inputs: JACK cycle times + SyncTimer playhead getters -------------------------- */
typedef struct zlo_clock {
    uint64_t current_usecs, next_usecs;          /* jack_get_cycle_times, SamplerSynth.cpp:128 */
    uint64_t jackPlayhead;                       /* SyncTimer.cpp:990-996 */
    uint64_t jackPlayheadUsecs;                  /* SyncTimer.cpp:998-1004 */
    uint64_t jackSubbeatLengthInMicroseconds;    /* SyncTimer.cpp:1006-1009 */
} zlo_clock;

/* ---- SamplerSynthVoice (SamplerSynthVoice.cpp:20-39) --------------------------------------- */
typedef struct zlo_voice {
    /* SamplerSynthVoicePrivate */
    int32_t  hasCommand;          /* clipCommand != nullptr */
    zlo_clip_command cmd;         /* *clipCommand (the voice owns its copy) */
    int32_t  clip;                /* d->clip as index, -1 = nullptr */
    int64_t  clipPositionId;
    uint64_t startTick, nextLoopTick, nextLoopUsecs;
    double   pitchRatio, sourceSamplePosition, sourceSampleLength;
    float    lgain, rgain;
    zlo_adsr adsr;
    /* juce::SynthesiserVoice bookkeeping that the hot path observes */
    int32_t  sound;               /* getCurrentlyPlayingSound() as index, -1 = none */
    int32_t  isPlaying;           /* SamplerSynthVoice.h:31 */
} zlo_voice;

typedef struct zlo_report {       /* what :265-267 hands to the positions model this block */
    int32_t valid;                /* 0 when the voice stopped inside the block (d->clip == nullptr) */
    float   gain;                 /* peakGain * 0.5f */
    float   progress;             /* (float)(sourceSamplePosition / sourceSampleLength) */
} zlo_report;

void zlo_voice_init(zlo_voice *v);
/* setCurrentCommand :58-98 ; returns 1 if the incoming command was merged (and should be recycled) */
int  zlo_voice_set_current_command(zlo_voice *v, const zlo_clip_command *cmd,
                                   const zlo_clip *clips, const zlo_sound *sounds);
/* startNote :110-144 (after juce::Synthesiser::startVoice set currentlyPlayingSound) */
void zlo_voice_start_note(zlo_voice *v, int midiNote, float velocity, int soundIndex,
                          const zlo_sound *sounds, zlo_clip *clips, double playbackSampleRate, int64_t now_ms);
/* stopNote :146-169 */
void zlo_voice_stop_note(zlo_voice *v, int allowTailOff, zlo_clip *clips, int64_t now_ms);
/* process :174-270.  L/R have nframes floats and are accumulated into.  pos_trace (optional,
 * nframes int32) receives (int)sourceSamplePosition per rendered frame, -1 for unrendered ones. */
void zlo_voice_process(zlo_voice *v, float *L, float *R, uint32_t nframes, const zlo_clock *clk,
                       const zlo_sound *sounds, zlo_clip *clips, uint32_t mode, int64_t now_ms,
                       zlo_report *rep, int32_t *pos_trace);

/* ---- SamplerChannel (SamplerSynth.cpp:116-148,187-230) ------------------------------------- */
typedef struct zlo_channel {
    zlo_voice *voices;
    int32_t    nvoices;           /* reference: SAMPLER_CHANNEL_VOICE_COUNT = 8 (SamplerSynth.cpp:23) */
    int32_t    midiChannel;
    int32_t    enabled;
} zlo_channel;

/* handleCommand :187-230 ; clip->sound mapping is identity (sound index == clip index), as in
 * SamplerSynth::registerClip :285-295 (one SamplerSynthSound per ClipAudioSource).
 * Returns 1 if the command was consumed by a voice (start) or merged, 0 if dropped. */
int  zlo_channel_handle_command(zlo_channel *ch, const zlo_clip_command *cmd, uint64_t currentTick,
                                const zlo_sound *sounds, zlo_clip *clips, double playbackSampleRate, int64_t now_ms);
/* process :123-141 (after the command ring was drained): zero L/R then every playing voice in order.
 * reports: nvoices entries (may be NULL). */
void zlo_channel_process(zlo_channel *ch, float *L, float *R, uint32_t nframes, const zlo_clock *clk,
                         const zlo_sound *sounds, zlo_clip *clips, uint32_t mode, int64_t now_ms,
                         zlo_report *reports);

/* ---- AudioLevels metering (AudioLevels.cpp:234-236,330-412) -------------------------------- */
typedef struct zlo_levels_channel {
    int32_t peakA, peakB;
    float   peakAHoldSignal, peakBHoldSignal;
    /* outputs of the last tick */
    float   peakDbA, peakDbB, combinedDb, holdDbA, holdDbB;
} zlo_levels_channel;

float   zlo_convert_to_dbfs(float raw);                         /* :330-341 */
float   zlo_add_float_db(float db1, float db2);                 /* :234-236 */
int32_t zlo_sample_to_peak_int(float x);                        /* :367 / :378 */
/* one 50 ms tick for one channel over its latest block (:359-398).  L/R may be NULL with n = 0
 * (bufferReadSize == 0).  with_hold mirrors the channelIndex == 1 branch (:391-398). */
void    zlo_levels_tick(zlo_levels_channel *c, const float *L, const float *R, uint32_t n, int with_hold);
/* build-defined extension (absent in reference): block RMS = sqrtf(sumsq / n) in fp32; the sum of squares has a
 * fixed order -- tiles of 64 frames from frame `off` (1 with quirk Q2, 0 with ZLO_MODE_FIX_DELAY), a balanced pairwise
 * tree inside a tile, tiles added in order (zl_oracle.c) -- so that it is bit-exact across implementations. */
float   zlo_block_sumsq(const float *x, uint32_t n, uint32_t off);
float   zlo_block_rms(const float *x, uint32_t n, uint32_t off);

/* ---- recorder sample format (AudioLevels.cpp:53-58,72-76; juce::WavAudioFormat 16 bit, restated, version unpinned) ---- */
int16_t zlo_pcm16_sample(float x);
void    zlo_pcm16_stereo(const float *L, const float *R, uint32_t n, int16_t *out);     /* out [n][2] */

/* ---- JackPassthrough (JackPassthrough.cpp:45-115) ------------------------------------------ */
typedef struct zlo_passthrough {
    float dryAmount, wetFx1Amount, wetFx2Amount, panAmount;
    int32_t muted;
} zlo_passthrough;
void zlo_passthrough_init(zlo_passthrough *p);
/* out[0..5] = dryL, dryR, wetFx1L, wetFx1R, wetFx2L, wetFx2R */
void zlo_passthrough_process(const zlo_passthrough *p, const float *inL, const float *inR,
                             float *const out[6], uint32_t nframes);

/* ---- flat batch driver used by parity tests and the cpu_baseline leg ------------------------
 * Renders `nblocks` consecutive blocks of `nframes` for `nbuses` channels of `vpb` voices each.
 * busL/busR: [nbuses][nblocks*nframes].  clocks: [nblocks].  reports: last block, [nbuses*vpb] (may be NULL).
 * mix_group > 0 selects the engine's documented two-level summation order instead of the
 * reference's strictly sequential one (see DESIGN.md "bus summation order"); 0 = sequential.
 * threads > 1 partitions buses over that many pthreads (mirrors one RT thread per JACK client). */
void zlo_render_batch(zlo_channel *channels, int32_t nbuses, const zlo_sound *sounds, zlo_clip *clips,
                      const zlo_clock *clocks, uint32_t nblocks, uint32_t nframes, uint32_t mode,
                      int32_t mix_group, float *busL, float *busR, zlo_report *reports, int32_t threads);
void zlo_render_batch_at(zlo_channel *channels, int32_t nbuses, const zlo_sound *sounds, zlo_clip *clips,
                      const zlo_clock *clocks, uint32_t nblocks, uint32_t nframes, uint32_t mode,
                      int32_t mix_group, float *busL, float *busR, zlo_report *reports, int32_t threads, int64_t now_ms);

#ifdef __cplusplus
}
#endif
#endif
