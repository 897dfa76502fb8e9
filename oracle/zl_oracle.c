/*
 * zl_oracle.c -- CPU restatement of libzl's sampler hot path.  TEST INFRASTRUCTURE ONLY.
 * See zl_oracle.h for the parity status ("parity unpinned") and who may use this file.
 * References are file:line under /root/reference/lib unless stated.
 */
#include "zl_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* float -> quint64 as the reference's target (aarch64 fcvtzu) performs it: saturating, NaN -> 0.
 * On x86-64 a negative float -> unsigned conversion is undefined; the reference only reaches it
 * with lengthInBeats = -1 (quirk Q10, invalid input). */
static uint64_t f32_to_u64_sat(float f)
{
    if (!(f > 0.0f)) return 0;                       /* negatives, -0, NaN */
    if (f >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)f;
}
static uint64_t f64_to_u64_sat(double f)
{
    if (!(f > 0.0)) return 0;
    if (f >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)f;
}

/* =============================== juce::ADSR ================================================== */

void zlo_adsr_default_params(zlo_adsr_params *p)
{
    p->attack = 0.1f; p->decay = 0.1f; p->sustain = 1.0f; p->release = 0.1f;   /* juce::ADSR::Parameters defaults */
}

static void adsr_go_to_next_state(zlo_adsr *a)
{
    if (a->state == ZLO_ADSR_ATTACK) { a->state = (a->decayRate > 0.0f ? ZLO_ADSR_DECAY : ZLO_ADSR_SUSTAIN); return; }
    if (a->state == ZLO_ADSR_DECAY)  { a->state = ZLO_ADSR_SUSTAIN; return; }
    if (a->state == ZLO_ADSR_RELEASE) zlo_adsr_reset(a);
}

static float adsr_get_rate(float distance, float timeInSeconds, double sr)
{
    return timeInSeconds > 0.0f ? (float)(distance / (timeInSeconds * sr)) : -1.0f;
}

static void adsr_recalculate_rates(zlo_adsr *a)
{
    a->attackRate  = adsr_get_rate(1.0f, a->p.attack, a->sampleRate);
    a->decayRate   = adsr_get_rate(1.0f - a->p.sustain, a->p.decay, a->sampleRate);
    a->releaseRate = adsr_get_rate(a->p.sustain, a->p.release, a->sampleRate);
    if ((a->state == ZLO_ADSR_ATTACK && a->attackRate <= 0.0f)
        || (a->state == ZLO_ADSR_DECAY && (a->decayRate <= 0.0f || a->env <= a->p.sustain))
        || (a->state == ZLO_ADSR_RELEASE && a->releaseRate <= 0.0f))
        adsr_go_to_next_state(a);
}

void zlo_adsr_init(zlo_adsr *a)
{
    memset(a, 0, sizeof *a);
    zlo_adsr_default_params(&a->p);
    a->sampleRate = 44100.0;
    a->state = ZLO_ADSR_IDLE;
    a->env = 0.0f;
    adsr_recalculate_rates(a);
}

void zlo_adsr_set_sample_rate(zlo_adsr *a, double sr) { a->sampleRate = sr; }

void zlo_adsr_set_parameters(zlo_adsr *a, const zlo_adsr_params *p)
{
    a->p = *p;
    adsr_recalculate_rates(a);
}

void zlo_adsr_reset(zlo_adsr *a) { a->env = 0.0f; a->state = ZLO_ADSR_IDLE; }

void zlo_adsr_note_on(zlo_adsr *a)
{
    if (a->attackRate > 0.0f)      { a->state = ZLO_ADSR_ATTACK; }
    else if (a->decayRate > 0.0f)  { a->env = 1.0f; a->state = ZLO_ADSR_DECAY; }
    else                           { a->env = a->p.sustain; a->state = ZLO_ADSR_SUSTAIN; }
}

void zlo_adsr_note_off(zlo_adsr *a)
{
    if (a->state != ZLO_ADSR_IDLE) {
        if (a->p.release > 0.0f) {
            a->releaseRate = (float)(a->env / (a->p.release * a->sampleRate));
            a->state = ZLO_ADSR_RELEASE;
        } else {
            zlo_adsr_reset(a);
        }
    }
}

float zlo_adsr_next(zlo_adsr *a)
{
    switch (a->state) {
    case ZLO_ADSR_IDLE:
        return 0.0f;
    case ZLO_ADSR_ATTACK:
        a->env += a->attackRate;
        if (a->env >= 1.0f) { a->env = 1.0f; adsr_go_to_next_state(a); }
        break;
    case ZLO_ADSR_DECAY:
        a->env -= a->decayRate;
        if (a->env <= a->p.sustain) { a->env = a->p.sustain; adsr_go_to_next_state(a); }
        break;
    case ZLO_ADSR_SUSTAIN:
        a->env = a->p.sustain;
        break;
    case ZLO_ADSR_RELEASE:
        a->env -= a->releaseRate;
        if (a->env <= 0.0f) adsr_go_to_next_state(a);
        break;
    }
    return a->env;
}

int zlo_adsr_is_active(const zlo_adsr *a) { return a->state != ZLO_ADSR_IDLE; }

/* =============================== positions model ============================================= */

void zlo_positions_init(zlo_positions *m)
{
    for (int i = 0; i < ZLO_POSITION_COUNT; ++i) {       /* ClipAudioSourcePositionsModel.cpp:7-12,17-21 */
        m->pos[i].id = -1; m->pos[i].progress = 0.0f; m->pos[i].gain = 0.0f; m->pos[i].lastUpdated = 0;
    }
    m->updatePeakGain = 0;
    m->peakGain = 0.0f;
}

int zlo_positions_cleanup(zlo_positions *m, int64_t now_ms)      /* :191-209 */
{
    const int64_t allowedTime = now_ms - 1000;
    int removed = 0;
    for (int i = 0; i < ZLO_POSITION_COUNT; ++i) {
        zlo_position *p = &m->pos[i];
        if (p->id > -1 && p->lastUpdated < allowedTime) {
            p->id = -1; p->gain = 0.0f; p->progress = 0.0f; ++removed;
        }
    }
    return removed;
}

int64_t zlo_positions_create(zlo_positions *m, float initialProgress, int64_t now_ms)   /* :78-100 */
{
    zlo_position *position = NULL;
    int positionRow = -1;
    for (int i = 0; i < ZLO_POSITION_COUNT; ++i) {
        ++positionRow;
        if (m->pos[i].id == -1) { position = &m->pos[i]; break; }
    }
    if (position) {
        position->id = positionRow;
        position->progress = initialProgress;
        position->lastUpdated = now_ms;
        m->updatePeakGain = 1;
        zlo_positions_cleanup(m, now_ms);
    }
    /* Reference quirk: when all 32 rows are taken the loop ends with positionRow == 31 and no
     * position; 31 is returned although that row belongs to another voice (:82-99). */
    return positionRow;
}

void zlo_positions_set_gain_and_progress(zlo_positions *m, int64_t id, float gain, float progress, int64_t now_ms) /* :126-138 */
{
    if (id > -1 && id < ZLO_POSITION_COUNT) {
        zlo_position *p = &m->pos[id];
        p->gain = gain; p->progress = progress; p->lastUpdated = now_ms;
        m->updatePeakGain = 1;
    }
}

void zlo_positions_remove(zlo_positions *m, int64_t id, int64_t now_ms)   /* :140-153 */
{
    if (id > -1 && id < ZLO_POSITION_COUNT) {
        zlo_position *p = &m->pos[id];
        p->id = -1; p->gain = 0.0f; p->progress = 0.0f;
        m->updatePeakGain = 1;
    }
    zlo_positions_cleanup(m, now_ms);
}

float zlo_positions_peak_gain(zlo_positions *m)                     /* :160-173 */
{
    if (m->updatePeakGain) {
        float peak = 0.0f;
        for (int i = 0; i < ZLO_POSITION_COUNT; ++i) peak = peak > m->pos[i].gain ? peak : m->pos[i].gain;
        /* abs(float) > 0.01 : the int abs() overload is not viable for a float argument in C++,
         * so this is fabs in double against the double literal 0.01 */
        if (fabs((double)(m->peakGain - peak)) > 0.01) m->peakGain = peak;
        m->updatePeakGain = 0;
    }
    return m->peakGain;
}

double zlo_positions_first_progress(const zlo_positions *m)         /* :175-185 */
{
    double progress = -1.0;
    for (int i = 0; i < ZLO_POSITION_COUNT; ++i)
        if (m->pos[i].id > -1) { progress = m->pos[i].progress; break; }
    return progress;
}

/* =============================== ClipAudioSource parameters ================================== */

float zlo_subbeat_count_to_seconds(uint64_t bpm, uint64_t subbeats)   /* SyncTimer.cpp:180-183,936-939 */
{
    if (bpm < 50) bpm = 50;                 /* qBound(BPM_MINIMUM, bpm, BPM_MAXIMUM), SyncTimer.cpp:28-29 */
    if (bpm > 200) bpm = 200;
    const uint64_t ns = (subbeats * 60000000000ULL) / (bpm * (uint64_t)ZLO_BEAT_SUBDIVISIONS);
    return ns / (float)1000000000;          /* quint64 -> float, float division */
}

void zlo_clip_init(zlo_clip *c, float durationSeconds, double sourceSampleRate)
{
    memset(c, 0, sizeof *c);
    c->startPositionInSeconds = 0;          /* ClipAudioSource.cpp:63 */
    c->lengthInSeconds = durationSeconds;   /* :64 then :158 lengthInSeconds = edit->getLength() */
    c->lengthInBeats = -1;                  /* :65 */
    c->volumeAbsolute = 1.0f;               /* :66 cached fader position; build input, 1.0 = unity */
    c->pan = 0.0f;                          /* :69 */
    c->duration = durationSeconds;          /* :367 */
    c->rootNote = 60;                       /* :81 */
    c->sliceBaseMidiNote = 60;              /* :78 */
    c->keyZoneStart = 0; c->keyZoneEnd = 127;   /* :79-80 */
    c->slices = 0; c->nSlicePositions = 0;
    zlo_adsr_init(&c->adsr);
    {                                       /* :164-168 : default Parameters with attack 0, release .05 */
        zlo_adsr_params p; zlo_adsr_default_params(&p);
        p.attack = 0.0f; p.release = 0.05f;
        zlo_adsr_set_sample_rate(&c->adsr, sourceSampleRate);
        zlo_adsr_set_parameters(&c->adsr, &p);
    }
    zlo_positions_init(&c->positions);
    zlo_clip_set_slices(c, 16);             /* :204 */
}

float zlo_clip_get_start_position(const zlo_clip *c, int slice)      /* :261-268 */
{
    if (slice > -1 && slice < c->nSlicePositions)
        return (float)(c->startPositionInSeconds + (c->lengthInSeconds * c->slicePositions[slice]));
    return c->startPositionInSeconds;
}

float zlo_clip_get_stop_position(const zlo_clip *c, int slice)       /* :270-277 */
{
    if (slice > -1 && slice + 1 < c->nSlicePositions)
        return (float)(c->startPositionInSeconds + (c->lengthInSeconds * c->slicePositions[slice + 1]));
    return c->startPositionInSeconds + c->lengthInSeconds;
}

void zlo_clip_set_start_position(zlo_clip *c, float s)               /* :255-259 */
{
    c->startPositionInSeconds = s > 0.0f ? s : 0.0f;   /* jmax(0.0f, s) */
}

void zlo_clip_set_length(zlo_clip *c, float beat, int bpm)           /* :352-360 */
{
    c->lengthInSeconds = zlo_subbeat_count_to_seconds((uint64_t)bpm, f32_to_u64_sat(beat * (float)ZLO_BEAT_SUBDIVISIONS));
    c->lengthInBeats = beat;
}

void zlo_clip_set_volume_absolute(zlo_clip *c, float vol)            /* :328-336 (clamp; fader readback taken as identity) */
{
    const float lo = vol < 1.0f ? vol : 1.0f;      /* qMin(vol, 1.0f) */
    c->volumeAbsolute = 0.0f > lo ? 0.0f : lo;     /* qMax(0.0f, ...) */
}

void zlo_clip_set_pan(zlo_clip *c, float pan) { if (c->pan != pan) c->pan = pan; }   /* :623-629 */

void zlo_clip_set_slices(zlo_clip *c, int slices)                    /* :495-528 */
{
    if (slices > ZLO_MAX_SLICES) slices = ZLO_MAX_SLICES;            /* oracle storage bound only */
    if (c->slices != slices) {
        if (slices == 0) {
            c->nSlicePositions = 0;
        } else if (c->slices > slices) {
            while (c->nSlicePositions > slices) c->nSlicePositions--;
        } else if (c->slices < slices) {
            double lastSlicePosition = 0.0f;
            if (c->nSlicePositions > 0) lastSlicePosition = c->slicePositions[c->nSlicePositions - 1];
            double positionIncrement = (1.0f - lastSlicePosition) / (slices - c->slices);
            double newPosition = lastSlicePosition + positionIncrement;
            if (c->nSlicePositions == 0) c->slicePositions[c->nSlicePositions++] = 0.0f;
            while (c->nSlicePositions < slices) {
                c->slicePositions[c->nSlicePositions++] = newPosition;
                newPosition += positionIncrement;
            }
        }
        c->slices = slices;
    }
}

int zlo_clip_slice_for_midi_note(const zlo_clip *c, int midiNote)    /* :575-578 */
{
    return ((c->slices - (c->sliceBaseMidiNote % c->slices)) + midiNote) % c->slices;
}

/* Q13: each setter builds a fresh default Parameters and sets one field (:636-685) */
void zlo_clip_set_adsr_attack(zlo_clip *c, float v)
{
    if (c->adsr.p.attack != v) { zlo_adsr_params p; zlo_adsr_default_params(&p); p.attack = v; zlo_adsr_set_parameters(&c->adsr, &p); }
}
void zlo_clip_set_adsr_decay(zlo_clip *c, float v)
{
    if (c->adsr.p.decay != v) { zlo_adsr_params p; zlo_adsr_default_params(&p); p.decay = v; zlo_adsr_set_parameters(&c->adsr, &p); }
}
void zlo_clip_set_adsr_sustain(zlo_clip *c, float v)
{
    if (c->adsr.p.sustain != v) { zlo_adsr_params p; zlo_adsr_default_params(&p); p.sustain = v; zlo_adsr_set_parameters(&c->adsr, &p); }
}
void zlo_clip_set_adsr_release(zlo_clip *c, float v)
{
    if (c->adsr.p.release != v) { zlo_adsr_params p; zlo_adsr_default_params(&p); p.release = v; zlo_adsr_set_parameters(&c->adsr, &p); }
}

/* ====================== per-clip level / progress chain ======================= */

/* juce::Decibels::gainToDecibels<Type> / decibelsToGain<Type> (juce_Decibels.h, third-party, version unpinned):
 *   gain > 0 ? jmax(minusInfinityDb, static_cast<Type>(std::log10(gain)) * Type(20.0)) : minusInfinityDb
 *   decibels > minusInfinityDb ? std::pow(Type(10.0), decibels * Type(0.05)) : Type()
 * with minusInfinityDb = -100. */
static float  zlo_gain_to_db_f(float gain)   { if (!(gain > 0.0f)) return -100.0f; const float d = (float)log10f(gain) * 20.0f; return d > -100.0f ? d : -100.0f; }
static double zlo_gain_to_db_d(double gain)  { if (!(gain > 0.0)) return -100.0; const double d = log10(gain) * 20.0; return d > -100.0 ? d : -100.0; }
static double zlo_db_to_gain_d(double db)    { return db > -100.0 ? pow(10.0, db * 0.05) : 0.0; }

void zlo_clip_meter_init(zlo_clip_meter *m)
{
    m->currentLeveldB = -400.0; m->prevLeveldB = -400.0;            /* :68-69 */
    m->firstPositionProgress = 0.0;                                /* :85 */
    m->nextPositionUpdateTime = 0; m->nextGainUpdateTime = 0;      /* :84,87 */
}

int zlo_sync_audio_level(zlo_clip_meter *m, zlo_clip *clip, int64_t now_ms, float *value)   /* ClipAudioSource.cpp:88-113 */
{
    int fired = 0;
    if (m->nextGainUpdateTime < now_ms) {                          /* :89 */
        m->prevLeveldB = m->currentLeveldB;                        /* :90 */
        /* :92 -- qMax(gainToDecibels(float peakGain), levelClient dB): the tracktion level client never sees the
         * sampler's audio (SamplerSynth plays straight to JACK), it reads -100 dB = the floor of gainToDecibels */
        m->currentLeveldB = (double)zlo_gain_to_db_f(zlo_positions_peak_gain(&clip->positions));
        const double prevLevel = zlo_db_to_gain_d(m->prevLeveldB); /* :98 */
        if (m->prevLeveldB > m->currentLeveldB)                    /* :100-101 */
            m->currentLeveldB = zlo_gain_to_db_d(prevLevel * 0.94);
        if (fabs(m->currentLeveldB - m->prevLeveldB) > 0.1) {      /* :104 */
            *value = (float)m->currentLeveldB;                     /* :108: void (*)(float) called with a double */
            fired = 1;
        }
        m->nextGainUpdateTime = now_ms + 30;                       /* :111 */
    }
    return fired;
}

int zlo_sync_progress(zlo_clip_meter *m, const zlo_clip *clip, int has_callback, int64_t now_ms, float *value)   /* :225-240 */
{
    int fired = 0;
    if (m->nextPositionUpdateTime < now_ms) {                      /* :226 */
        double newPosition = clip->startPositionInSeconds / clip->duration;   /* :227, float / float -> double */
        if (has_callback && zlo_positions_first_progress(&clip->positions) > -1.0f)   /* :228 */
            newPosition = zlo_positions_first_progress(&clip->positions);
        if (fabs(m->firstPositionProgress - newPosition) > 0.001) {           /* :231 */
            m->firstPositionProgress = newPosition;
            /* :234 calls the callback unconditionally (a null pointer is the host's bug there); the restatement fires
             * only with a callback installed */
            if (has_callback) { *value = (float)(m->firstPositionProgress * clip->duration); fired = 1; }
            m->nextPositionUpdateTime = now_ms + 100;              /* :238 */
        }
    }
    return fired;
}

/* =============================== ClipCommand ================================================= */

void zlo_clip_command_clear(zlo_clip_command *c)    /* ClipCommand.h:13-32 defaults, :74-91 clear */
{
    memset(c, 0, sizeof *c);
    c->clip = -1; c->midiNote = -1; c->midiChannel = -1; c->slice = -1;
}

int zlo_clip_command_equivalent(const zlo_clip_command *a, const zlo_clip_command *b)   /* :33-39 */
{
    return a->clip == b->clip
        && ((a->changeSlice && b->changeSlice && a->slice == b->slice)
            || (!a->changeSlice && !b->changeSlice && a->midiNote == b->midiNote && a->midiChannel == b->midiChannel));
}

/* =============================== SamplerSynthVoice =========================================== */


/* ================================================================================================
 * SyncTimer: the step ring that carries ClipCommands (SyncTimer.cpp; header comment in zl_oracle.h)
 * ============================================================================================== */
#define ZLO_BEAT_SUBDIV_U64 ((uint64_t)ZLO_BEAT_SUBDIVISIONS)

static uint64_t st_subbeat_count_to_nanoseconds(uint64_t bpm, uint64_t subBeatCount)   /* :180-183 */
{
    return (subBeatCount * 60000000000ULL) / (bpm * ZLO_BEAT_SUBDIV_U64);
}
static float st_nanoseconds_to_subbeat_count(uint64_t bpm, uint64_t nanoseconds)        /* :184-187 */
{
    return (float)(nanoseconds / (60000000000ULL / (bpm * ZLO_BEAT_SUBDIV_U64)));
}

static void st_update_schedule_ahead(zlo_sync_timer *t)                                  /* :704-707 */
{
    /* nanosecondsToSubbeatCount(bpm, jackLatency * (float)1000000) + 1 : quint64 * float -> float -> quint64 argument */
    const float ns = (float)t->jackLatency * (float)1000000;
    t->scheduleAheadAmount = (uint64_t)(st_nanoseconds_to_subbeat_count(t->bpm, (uint64_t)ns) + 1);
}

static void st_step_ensure_fresh(zlo_step *s)                                            /* :50-62 */
{
    if (s->played) { s->played = 0; s->nClipCommands = 0; s->nBpmCommands = 0; }
}

static void st_step_reserve_clip(zlo_step *s, int32_t need)
{
    if (need > s->capClipCommands) {
        int32_t cap = s->capClipCommands ? s->capClipCommands * 2 : 4;
        while (cap < need) cap *= 2;
        s->clipCommands = (zlo_clip_command *)realloc(s->clipCommands, (size_t)cap * sizeof(zlo_clip_command));
        s->capClipCommands = cap;
    }
}
static void st_step_push_bpm(zlo_step *s, int32_t bpm)
{
    if (s->nBpmCommands + 1 > s->capBpmCommands) {
        const int32_t cap = s->capBpmCommands ? s->capBpmCommands * 2 : 4;
        s->bpmCommands = (int32_t *)realloc(s->bpmCommands, (size_t)cap * sizeof(int32_t));
        s->capBpmCommands = cap;
    }
    s->bpmCommands[s->nBpmCommands++] = bpm;
}

zlo_sync_timer *zlo_sync_timer_new(void)
{
    zlo_sync_timer *t = (zlo_sync_timer *)calloc(1, sizeof *t);
    t->stepRing = (zlo_step *)calloc(ZLO_STEP_RING_COUNT, sizeof(zlo_step));
    for (int i = 0; i < ZLO_STEP_RING_COUNT; ++i) t->stepRing[i].played = 1;            /* :78 */
    t->bpm = 120; t->threadPaused = 1; t->isPaused = 1;                                   /* :236,250,440 */
    t->jackPlayheadBpm = 120;                                                             /* :423 */
    t->jackSubbeatLengthInMicroseconds = st_subbeat_count_to_nanoseconds(t->bpm, 1) / 1000;   /* ctor, :749 */
    st_update_schedule_ahead(t);                                                          /* ctor, :771 */
    return t;
}

void zlo_sync_timer_free(zlo_sync_timer *t)
{
    if (!t) return;
    for (int i = 0; i < ZLO_STEP_RING_COUNT; ++i) { free(t->stepRing[i].clipCommands); free(t->stepRing[i].bpmCommands); }
    free(t->stepRing);
    free(t);
}

void zlo_sync_timer_set_latency(zlo_sync_timer *t, uint32_t bufferSize, double sampleRate)   /* :730-741 */
{
    const uint64_t newLatency = (uint64_t)((1000 * (double)bufferSize) / (double)sampleRate);
    if (newLatency != t->jackLatency) { t->jackLatency = newLatency; st_update_schedule_ahead(t); }
}

static zlo_step *st_delayed_step(zlo_sync_timer *t, uint64_t delay)                      /* :364-378 */
{
    uint64_t step;
    if (t->isPaused) {
        step = (t->stepReadHead + delay + 1) % ZLO_STEP_RING_COUNT;
    } else {
        const uint64_t a = t->cumulativeBeat + delay, b = t->jackPlayhead + 1;
        step = (t->stepReadHeadOnStart + (a > b ? a : b)) % ZLO_STEP_RING_COUNT;
    }
    zlo_step *s = &t->stepRing[step];
    st_step_ensure_fresh(s);
    return s;
}

int zlo_step_schedule(zlo_clip_command *list, int32_t *n, const zlo_clip_command *command)   /* :1014-1047 */
{
    int foundExisting = 0;
    for (int32_t i = 0; i < *n; ++i) {
        zlo_clip_command *existingCommand = &list[i];
        if (zlo_clip_command_equivalent(existingCommand, command)) {
            if (command->changeLooping) { existingCommand->looping = command->looping; existingCommand->changeLooping = 1; }
            if (command->changePitch)   { existingCommand->pitchChange = command->pitchChange; existingCommand->changePitch = 1; }
            if (command->changeSpeed)   { existingCommand->speedRatio = command->speedRatio; existingCommand->changeSpeed = 1; }
            if (command->changeGainDb)  { existingCommand->gainDb = command->gainDb; existingCommand->changeGainDb = 1; }
            if (command->changeVolume)  { existingCommand->volume = command->volume; existingCommand->changeVolume = 1; }
            if (command->startPlayback) { existingCommand->startPlayback = 1; }
            foundExisting = 1;
        }
    }
    if (foundExisting) return 0;                 /* deleteClipCommand(command) */
    list[(*n)++] = *command;                     /* stepData->clipCommands << command */
    return 1;
}

void zlo_schedule_clip_command(zlo_sync_timer *t, const zlo_clip_command *command, uint64_t delay)   /* :1011-1048 */
{
    zlo_step *s = st_delayed_step(t, delay);
    st_step_reserve_clip(s, s->nClipCommands + 1);
    zlo_step_schedule(s->clipCommands, &s->nClipCommands, command);
}

void zlo_sync_timer_set_bpm(zlo_sync_timer *t, uint64_t bpm)                             /* :954-975 */
{
    if (t->bpm != bpm) {
        t->bpm = bpm;                                                                     /* timerThread->setBPM */
        t->jackSubbeatLengthInMicroseconds = st_subbeat_count_to_nanoseconds(t->bpm, 1) / 1000;
        st_update_schedule_ahead(t);
        st_step_push_bpm(st_delayed_step(t, 0), (int32_t)bpm);                            /* scheduleTimerCommand(0, SetBpmOperation) */
    }
}

void zlo_sync_timer_start(zlo_sync_timer *t, int bpm)                                    /* :870-879 */
{
    zlo_sync_timer_set_bpm(t, (uint64_t)bpm);
    t->stepReadHeadOnStart = t->stepReadHead;
    t->threadPaused = 0; t->isPaused = 0;                                                 /* resume() -> pausedChanged (:750-752) */
}

void zlo_sync_timer_stop(zlo_sync_timer *t)                                              /* :881-925 */
{
    t->threadPaused = 1; t->isPaused = 1;                                                 /* pause() */
    t->beat = 0; t->cumulativeBeat = 0; t->jackPlayhead = 0;                              /* :888-890 */
    for (uint64_t step = 0; step < ZLO_STEP_RING_COUNT; ++step) {                         /* :893-916 */
        const uint64_t index = (step + t->stepReadHead) % ZLO_STEP_RING_COUNT;
        zlo_step *stepData = &t->stepRing[index];
        if (!stepData->played) {
            /* scheduleClipCommand(clipCommand, 0) addresses the step behind the read head.  When that is the step being
             * walked, every command is equivalent to itself: it folds into the list it is in, nothing is appended; the
             * step is then marked played and its commands never reach the sampler. */
            if (index != (t->stepReadHead + 1) % ZLO_STEP_RING_COUNT) {
                for (int32_t i = 0; i < stepData->nClipCommands; ++i) {
                    zlo_clip_command c = stepData->clipCommands[i];
                    c.changeVolume = 1; c.volume = 0;                                     /* :908-909 */
                    zlo_schedule_clip_command(t, &c, 0);
                }
            }
            stepData->played = 1;
        }
    }
}

void zlo_sync_timer_callback(zlo_sync_timer *t)                                          /* :391-418 */
{
    while (t->cumulativeBeat < (t->jackPlayhead + (t->scheduleAheadAmount * 2))) {
        t->beat = (t->beat + 1) % (ZLO_BEAT_SUBDIVISIONS * 4);
        ++t->cumulativeBeat;
    }
}

void zlo_sync_timer_queue_clip_to_start_on_channel(zlo_sync_timer *t, int32_t clip, int midiChannel)   /* :815-832 */
{
    zlo_clip_command command;
    zlo_clip_command_clear(&command);
    command.clip = clip; command.midiChannel = midiChannel; command.midiNote = 60;
    command.changeVolume = 1; command.volume = 1.0f; command.looping = 1;
    command.stopPlayback = 1; command.startPlayback = 1;
    const uint64_t bar = ZLO_BEAT_SUBDIV_U64 * 4;
    const uint64_t nextZeroBeat = t->threadPaused ? 0 : bar - (t->cumulativeBeat % bar);
    zlo_schedule_clip_command(t, &command, t->cumulativeBeat + nextZeroBeat < t->jackPlayhead ? nextZeroBeat + bar : nextZeroBeat);
}

void zlo_sync_timer_queue_clip_to_stop_on_channel(zlo_sync_timer *t, int32_t clip, int midiChannel)    /* :834-860 */
{
    for (uint64_t step = 0; step < ZLO_STEP_RING_COUNT; ++step) {                         /* :837-850: the first reference per unplayed step */
        zlo_step *stepData = &t->stepRing[step];
        if (!stepData->played) {
            for (int32_t i = 0; i < stepData->nClipCommands; ++i) {
                if (stepData->clipCommands[i].clip == clip) {
                    memmove(&stepData->clipCommands[i], &stepData->clipCommands[i + 1], (size_t)(stepData->nClipCommands - i - 1) * sizeof(zlo_clip_command));
                    stepData->nClipCommands -= 1;
                    break;
                }
            }
        }
    }
    zlo_clip_command command;
    zlo_clip_command_clear(&command);
    command.clip = clip; command.midiChannel = midiChannel; command.midiNote = 60; command.stopPlayback = 1;
    zlo_step *s = st_delayed_step(t, 0);                                                  /* appended without the merge, :858-859 */
    st_step_reserve_clip(s, s->nClipCommands + 1);
    s->clipCommands[s->nClipCommands++] = command;
}

int32_t zlo_sync_timer_process(zlo_sync_timer *t, uint32_t nframes, uint64_t current_usecs, uint64_t next_usecs, float period_usecs,
                               zlo_dispatch *out, int32_t max_out)                       /* :452-702 */
{
    (void)period_usecs;                                                                   /* (only the transport's bpm average uses it) */
    int32_t nout = 0;
    const uint64_t microsecondsPerFrame = (next_usecs - current_usecs) / nframes;         /* :482 */
    double thisStepBpm = t->jackPlayheadBpm;
    /* subbeatCountToNanoseconds takes const quint64 &bpm: the double playhead bpm is truncated on the way in (:484) */
    double thisStepSubbeatLengthInMicroseconds = (double)st_subbeat_count_to_nanoseconds((uint64_t)t->jackPlayheadBpm, 1) / 1000.0;
    if (!t->isPaused) {
        if (t->jackPlayhead == 0) t->jackNextPlaybackPosition = current_usecs;            /* :490-497 */
    }
    if (t->stepNextPlaybackPosition == 0) t->stepNextPlaybackPosition = current_usecs;    /* :500-502 */
    uint32_t firstAvailableFrame = 0, relativePosition = 0;
    while (t->stepNextPlaybackPosition < next_usecs && firstAvailableFrame < nframes) {   /* :512 */
        zlo_step *stepData = &t->stepRing[t->stepReadHead];
        t->stepReadHead = (t->stepReadHead + 1) % ZLO_STEP_RING_COUNT;                    /* stepReadHead->next */
        if (t->stepNextPlaybackPosition <= current_usecs) {                               /* :517-523 */
            relativePosition = firstAvailableFrame;
            ++firstAvailableFrame;
        } else {
            uint32_t p = microsecondsPerFrame ? (uint32_t)((t->stepNextPlaybackPosition - current_usecs) / microsecondsPerFrame) : 0;
            if (p < firstAvailableFrame) p = firstAvailableFrame;
            if (p > nframes - 1) p = nframes - 1;
            relativePosition = p;
            firstAvailableFrame = relativePosition;
        }
        (void)relativePosition;                /* where MIDI events of the step land; clip commands carry no frame offset */
        if (!stepData->played) {
            for (int32_t i = 0; i < stepData->nClipCommands; ++i) {                       /* :553-558 */
                if (nout < max_out) { out[nout].cmd = stepData->clipCommands[i]; out[nout].tick = t->jackPlayhead; }
                ++nout;
            }
            for (int32_t i = 0; i < stepData->nBpmCommands; ++i) {                        /* SetBpmOperation, :606-612 */
                uint64_t newBpm = (uint64_t)stepData->bpmCommands[i];
                if (newBpm < 50) newBpm = 50;
                if (newBpm > 200) newBpm = 200;
                zlo_sync_timer_set_bpm(t, newBpm);                                        /* q->setBpm(newBpm) */
                thisStepBpm = (double)newBpm;
            }
            stepData->played = 1;
        }
        if (t->jackPlayheadBpm != thisStepBpm) {                                          /* :634-639 */
            t->jackPlayheadBpm = thisStepBpm;
            thisStepSubbeatLengthInMicroseconds = (double)(st_subbeat_count_to_nanoseconds((uint64_t)t->jackPlayheadBpm, 1) / 1000);
        }
        if (!t->isPaused) {                                                               /* :660-667 */
            ++t->jackPlayhead;
            t->jackNextPlaybackPosition = (uint64_t)((double)t->jackNextPlaybackPosition + thisStepSubbeatLengthInMicroseconds);   /* quint64 += double */
        }
        t->stepNextPlaybackPosition = (uint64_t)((double)t->stepNextPlaybackPosition + thisStepSubbeatLengthInMicroseconds);       /* :671 */
    }
    return nout;
}

uint64_t zlo_sync_timer_jack_playhead(const zlo_sync_timer *t) { return t->threadPaused ? t->stepReadHead : t->jackPlayhead; }                           /* :990-996 */
uint64_t zlo_sync_timer_jack_playhead_usecs(const zlo_sync_timer *t) { return t->threadPaused ? t->stepNextPlaybackPosition : t->jackNextPlaybackPosition; }   /* :998-1004 */
uint64_t zlo_sync_timer_jack_subbeat_length_usecs(const zlo_sync_timer *t) { return t->jackSubbeatLengthInMicroseconds; }                               /* :1006-1009 */

static float velocity_to_gain(float velocity) { return velocity; }    /* SamplerSynthVoice.cpp:11-18 */

void zlo_voice_init(zlo_voice *v)
{
    memset(v, 0, sizeof *v);
    zlo_clip_command_clear(&v->cmd);
    v->clip = -1; v->clipPositionId = -1; v->sound = -1;
    zlo_adsr_init(&v->adsr);
}

int zlo_voice_set_current_command(zlo_voice *v, const zlo_clip_command *c,
                                  const zlo_clip *clips, const zlo_sound *sounds)   /* :58-98 */
{
    int merged = 0;
    if (v->hasCommand) {
        if (c->changeLooping) { v->cmd.looping = c->looping; v->cmd.changeLooping = 1; }
        if (c->changePitch)   { v->cmd.pitchChange = c->pitchChange; v->cmd.changePitch = 1; }
        if (c->changeSpeed)   { v->cmd.speedRatio = c->speedRatio; v->cmd.changeSpeed = 1; }
        if (c->changeGainDb)  { v->cmd.gainDb = c->gainDb; v->cmd.changeGainDb = 1; }
        if (c->changeVolume) {
            v->cmd.volume = c->volume; v->cmd.changeVolume = 1;
            v->lgain = velocity_to_gain(v->cmd.volume);
            v->rgain = velocity_to_gain(v->cmd.volume);
        }
        if (c->changeSlice)   { v->cmd.slice = c->slice; }
        if (c->startPlayback) {
            if (v->sound >= 0 && v->clip >= 0)   /* :89 ; d->clip is dereferenced unchecked at :90 */
                v->sourceSamplePosition = (int)(zlo_clip_get_start_position(&clips[v->clip], v->cmd.slice) * sounds[v->sound].sampleRate);
        }
        merged = 1;
    } else {
        v->cmd = *c;
        v->hasCommand = 1;
    }
    v->isPlaying = v->hasCommand;
    return merged;
}

void zlo_voice_start_note(zlo_voice *v, int midiNote, float velocity, int soundIndex,
                          const zlo_sound *sounds, zlo_clip *clips, double playbackSampleRate, int64_t now_ms)   /* :110-144 */
{
    const zlo_sound *sound = &sounds[soundIndex];
    v->sound = soundIndex;                                   /* juce::Synthesiser::startVoice: currentlyPlayingSound = sound */
    if (sound->valid) {                                      /* :114 (sound->clip() is the identity mapping here) */
        zlo_clip *clip = &clips[soundIndex];
        v->pitchRatio = pow(2.0, (midiNote - clip->rootNote) / 12.0) * sound->sampleRate / playbackSampleRate;   /* :115-116 */
        v->clip = soundIndex;                                                                                   /* :119 */
        v->sourceSampleLength = clip->duration * sound->sampleRate;                                             /* :120 */
        v->sourceSamplePosition = (int)(zlo_clip_get_start_position(clip, v->cmd.slice) * sound->sampleRate);   /* :121 */
        v->nextLoopTick = f32_to_u64_sat(v->startTick + clip->lengthInBeats * ZLO_BEAT_SUBDIVISIONS);           /* :123 (u64 + float -> float) */
        v->nextLoopUsecs = 0;                                                                                   /* :124 */
        if (v->clipPositionId > -1) zlo_positions_remove(&clip->positions, v->clipPositionId, now_ms);          /* :126-128 */
        v->clipPositionId = zlo_positions_create(&clip->positions, 0.0f, now_ms);                               /* :129 */
        v->lgain = velocity_to_gain(velocity);                                                                  /* :131-132 */
        v->rgain = velocity_to_gain(velocity);
        zlo_adsr_reset(&v->adsr);                                                                               /* :134-137 */
        zlo_adsr_set_sample_rate(&v->adsr, sound->sampleRate);
        zlo_adsr_set_parameters(&v->adsr, &clip->adsr.p);
        zlo_adsr_note_on(&v->adsr);
    }
}

void zlo_voice_stop_note(zlo_voice *v, int allowTailOff, zlo_clip *clips, int64_t now_ms)    /* :146-169 */
{
    if (allowTailOff) {
        zlo_adsr_note_off(&v->adsr);
    } else {
        v->sound = -1;                                         /* clearCurrentNote() */
        zlo_adsr_reset(&v->adsr);
        if (v->clip >= 0) {
            zlo_positions_remove(&clips[v->clip].positions, v->clipPositionId, now_ms);
            v->clip = -1;
            v->clipPositionId = -1;
        }
        if (v->hasCommand) {
            v->hasCommand = 0;
            zlo_clip_command_clear(&v->cmd);
            v->isPlaying = 0;
        }
        v->nextLoopTick = 0;
        v->nextLoopUsecs = 0;
    }
}

/* build-defined Hermite extension (absent in the reference): 4-point Catmull-Rom through x[pos-1 .. pos+2], evaluated in
 * TAP-WEIGHT form: the four cubic weights of the fractional position a are computed once per frame and shared by both
 * channels,
 *   w0 = ((-a/2 + 1) a - 1/2) a     w1 = (3/2 a - 5/2) a^2 + 1     w2 = ((-3/2 a + 2) a + 1/2) a     w3 = (a/2 - 1/2) a^2
 *   y  = w0 y0 + w1 y1 + w2 y2 + w3 y3
 * with the operation order and the fused multiply-adds fixed here (fmaf = one rounding, C99 7.12.13.1).  The same cubic as the
 * Horner form of JUCE's CatmullRomInterpolator, other roundings; 19 instead of 24 operations per stereo frame. */
typedef struct { float w0, w1, w2, w3; } hermite_w;
static hermite_w hermite_weights(float a)
{
    hermite_w w;
    const float t = a * a;
    w.w0 = fmaf(fmaf(-0.5f, a, 1.0f), a, -0.5f) * a;
    w.w1 = fmaf(fmaf(1.5f, a, -2.5f), t, 1.0f);
    w.w2 = fmaf(fmaf(-1.5f, a, 2.0f), a, 0.5f) * a;
    w.w3 = fmaf(0.5f, a, -0.5f) * t;
    return w;
}
static float hermite4(float y0, float y1, float y2, float y3, hermite_w w)
{
    return fmaf(w.w3, y3, fmaf(w.w2, y2, fmaf(w.w1, y1, w.w0 * y0)));
}

void zlo_voice_process(zlo_voice *v, float *leftBuffer, float *rightBuffer, uint32_t nframes, const zlo_clock *clk,
                       const zlo_sound *sounds, zlo_clip *clips, uint32_t mode, int64_t now_ms,
                       zlo_report *rep, int32_t *pos_trace)           /* :174-270 */
{
    if (rep) { rep->valid = 0; rep->gain = 0.0f; rep->progress = 0.0f; }
    if (pos_trace) for (uint32_t i = 0; i < nframes; ++i) pos_trace[i] = -1;
    if (v->sound < 0) return;                                                   /* :176 */
    const zlo_sound *playingSound = &sounds[v->sound];
    if (!(playingSound->valid && v->hasCommand)) return;                        /* :178 */
    zlo_clip *clip = &clips[v->clip];

    if (v->nextLoopUsecs == 0) {                                                /* :179-182 */
        const uint64_t differenceToPlayhead = v->nextLoopTick - clk->jackPlayhead;
        v->nextLoopUsecs = clk->jackPlayheadUsecs + (differenceToPlayhead * clk->jackSubbeatLengthInMicroseconds);
    }
    const double microsecondsPerFrame = (double)((clk->next_usecs - clk->current_usecs) / nframes);   /* :183 (integer division) */
    float peakGain = 0.0f;                                                      /* :184 */
    const float *const inL = playingSound->L;                                   /* :186 */
    const float *const inR = playingSound->R;                                   /* :187 */

    const float clipVolume = clip->volumeAbsolute;                              /* :189 */
    const int stopPosition = (int)(zlo_clip_get_stop_position(clip, v->cmd.slice) * playingSound->sampleRate);   /* :190, SamplerSynthSound.cpp:101-104 */
    const int sampleDuration = playingSound->length - 1;                        /* :191 */
    const float pan = (float)clip->pan;                                         /* :192 */
    const float lPan = (float)(0.5 * (1.0 + pan));                              /* :193 */
    const float rPan = (float)(0.5 * (1.0 - pan));                              /* :194 */
    const double sourceSampleRate = playingSound->sampleRate;                   /* :195 */
    const int isLooping = v->cmd.looping;                                       /* :196 */
    const int fixGain = (mode & ZLO_MODE_FIX_GAIN) != 0, fixDelay = (mode & ZLO_MODE_FIX_DELAY) != 0;
    const int hermite = (mode & ZLO_MODE_HERMITE) != 0;

    for (uint32_t frame = 0; frame < nframes; ++frame) {                        /* :197 */
        const int pos = (int)v->sourceSamplePosition;                           /* :198 */
        const float alpha = (float)(v->sourceSamplePosition - pos);             /* :199 */
        const float invAlpha = 1.0f - alpha;                                    /* :200 */
        const float envelopeValue = zlo_adsr_next(&v->adsr);                    /* :201 */
        if (pos_trace) pos_trace[frame] = pos;

        float l, r;
        if (hermite && sampleDuration > pos) {
            /* extension: whole-sample gain, the gain product formed first -- sample * ((gain * envelope) * volume) --; falls back
             * to 2-tap linear at the buffer edges */
            const int wide = (pos - 1 >= 0) && (pos + 2 <= sampleDuration);
            const hermite_w w = hermite_weights(alpha);
            const float sl = wide ? hermite4(inL[pos - 1], inL[pos], inL[pos + 1], inL[pos + 2], w)
                                  : (inL[pos] * invAlpha + inL[pos + 1] * alpha);
            l = sl * ((v->lgain * envelopeValue) * clipVolume);
            if (inR != NULL) {
                const float sr_ = wide ? hermite4(inR[pos - 1], inR[pos], inR[pos + 1], inR[pos + 2], w)
                                       : (inR[pos] * invAlpha + inR[pos + 1] * alpha);
                r = sr_ * ((v->rgain * envelopeValue) * clipVolume);
            } else {
                r = l;
            }
        } else if (fixGain) {
            l = sampleDuration > pos ? ((inL[pos] * invAlpha + inL[pos + 1] * alpha) * v->lgain * envelopeValue * clipVolume) : 0;
            r = (inR != NULL && sampleDuration > pos) ? ((inR[pos] * invAlpha + inR[pos + 1] * alpha) * v->rgain * envelopeValue * clipVolume) : l;
        } else {
            /* :204-205 -- Q1: the gain chain multiplies only the second tap */
            l = sampleDuration > pos ? (inL[pos] * invAlpha + inL[pos + 1] * alpha * v->lgain * envelopeValue * clipVolume) : 0;
            r = (inR != NULL && sampleDuration > pos) ? (inR[pos] * invAlpha + inR[pos + 1] * alpha * v->rgain * envelopeValue * clipVolume) : l;
        }

        const float mSignal = (float)(0.5 * (l + r));                           /* :208 */
        const float sSignal = l - r;                                            /* :209 */
        l = lPan * mSignal + sSignal;                                           /* :210 */
        r = rPan * mSignal - sSignal;                                           /* :211 */

        const float newGain = l + r;                                            /* :213-216 */
        if (newGain > peakGain) peakGain = newGain;

        /* :218-221 -- Q2: pointers are pre-incremented, so frame f lands in out[f+1]; the write for
         * f = nframes-1 is one past the JACK buffer in the reference and is dropped here. */
        if (fixDelay) {
            leftBuffer[frame] += l; rightBuffer[frame] += r;
        } else if (frame + 1 < nframes) {
            leftBuffer[frame + 1] += l; rightBuffer[frame + 1] += r;
        }

        v->sourceSamplePosition += v->pitchRatio;                               /* :223 */

        if (isLooping) {                                                        /* :225 */
            if (truncf(clip->lengthInBeats) == clip->lengthInBeats) {           /* :227 (trunc on float, compared as double) */
                if (clk->current_usecs + f64_to_u64_sat(frame * microsecondsPerFrame) >= v->nextLoopUsecs) {   /* :232 */
                    const uint64_t lengthInTicks = f32_to_u64_sat(clip->lengthInBeats * ZLO_BEAT_SUBDIVISIONS);   /* :234 */
                    v->nextLoopTick = v->nextLoopTick + lengthInTicks;          /* :235 */
                    const uint64_t differenceToPlayhead = v->nextLoopTick - clk->jackPlayhead;                 /* :236 */
                    v->nextLoopUsecs = clk->jackPlayheadUsecs + (differenceToPlayhead * clk->jackSubbeatLengthInMicroseconds);   /* :237 */
                    v->sourceSamplePosition = (int)(zlo_clip_get_start_position(clip, v->cmd.slice) * sourceSampleRate);     /* :241 */
                }
            } else if (v->sourceSamplePosition >= stopPosition) {               /* :243 */
                v->sourceSamplePosition = (int)(zlo_clip_get_start_position(clip, v->cmd.slice) * sourceSampleRate);         /* :246 */
            }
        } else {
            if (v->sourceSamplePosition >= stopPosition) {                      /* :249 */
                zlo_voice_stop_note(v, 0, clips, now_ms);                       /* :251 */
                break;
            } else if (v->sourceSamplePosition >= (stopPosition - (v->adsr.p.release * sourceSampleRate))) {   /* :253 */
                zlo_voice_stop_note(v, 1, clips, now_ms);                       /* :255 -- Q7: every frame */
            }
        }
        if (!zlo_adsr_is_active(&v->adsr)) {                                    /* :258-261 */
            zlo_voice_stop_note(v, 0, clips, now_ms);
            break;
        }
    }

    if (v->clip >= 0 && v->clipPositionId > -1) {                               /* :265-267 */
        const float gain = peakGain * 0.5f;
        const float progress = (float)(v->sourceSamplePosition / v->sourceSampleLength);
        zlo_positions_set_gain_and_progress(&clips[v->clip].positions, v->clipPositionId, gain, progress, now_ms);
        if (rep) { rep->valid = 1; rep->gain = gain; rep->progress = progress; }
    }
}

/* =============================== SamplerChannel ============================================== */

int zlo_channel_handle_command(zlo_channel *ch, const zlo_clip_command *c, uint64_t currentTick,
                               const zlo_sound *sounds, zlo_clip *clips, double playbackSampleRate, int64_t now_ms)   /* SamplerSynth.cpp:187-230 */
{
    const int sound = c->clip;                                  /* d->clipSounds[clipCommand->clip], :189 */
    int consumed = 0;
    if (c->stopPlayback || c->startPlayback) {
        if (c->stopPlayback) {                                  /* :191-203 */
            if (ch->midiChannel == c->midiChannel) {
                for (int i = 0; i < ch->nvoices; ++i) {
                    zlo_voice *voice = &ch->voices[i];
                    if (voice->sound >= 0 && voice->sound == sound && voice->hasCommand && zlo_clip_command_equivalent(&voice->cmd, c))
                        zlo_voice_stop_note(voice, 1, clips, now_ms);
                }
            }
        }
        if (c->startPlayback) {                                 /* :204-215 */
            if (ch->midiChannel == c->midiChannel) {
                for (int i = 0; i < ch->nvoices; ++i) {
                    zlo_voice *voice = &ch->voices[i];
                    if (!voice->isPlaying) {
                        zlo_voice_set_current_command(voice, c, clips, sounds);
                        voice->startTick = currentTick;
                        /* juce::Synthesiser::startVoice: a voice that still has a sound is hard-stopped first */
                        if (voice->sound >= 0) zlo_voice_stop_note(voice, 0, clips, now_ms);
                        zlo_voice_start_note(voice, c->midiNote, c->volume, sound, sounds, clips, playbackSampleRate, now_ms);
                        consumed = 1;
                        break;
                    }
                }
            }
        }
    } else {                                                    /* :216-229 */
        if (ch->midiChannel == c->midiChannel) {
            for (int i = 0; i < ch->nvoices; ++i) {
                zlo_voice *voice = &ch->voices[i];
                if (voice->sound >= 0 && voice->sound == sound && voice->hasCommand && zlo_clip_command_equivalent(&voice->cmd, c)) {
                    zlo_voice_set_current_command(voice, c, clips, sounds);
                    consumed = 1;
                }
            }
        }
    }
    return consumed;
}

void zlo_channel_process(zlo_channel *ch, float *L, float *R, uint32_t nframes, const zlo_clock *clk,
                         const zlo_sound *sounds, zlo_clip *clips, uint32_t mode, int64_t now_ms, zlo_report *reports)   /* :123-141 */
{
    if (reports) memset(reports, 0, sizeof(zlo_report) * (size_t)ch->nvoices);
    if (!ch->enabled) return;                                   /* :123 (buffers left untouched) */
    memset(L, 0, nframes * sizeof(float));                      /* :134-135 */
    memset(R, 0, nframes * sizeof(float));
    for (int i = 0; i < ch->nvoices; ++i) {                     /* :136-140 */
        zlo_voice *voice = &ch->voices[i];
        if (voice->isPlaying)
            zlo_voice_process(voice, L, R, nframes, clk, sounds, clips, mode, now_ms, reports ? &reports[i] : NULL, NULL);
    }
}

/* =============================== AudioLevels ================================================= */

float zlo_convert_to_dbfs(float raw)                            /* AudioLevels.cpp:330-341 */
{
    if (raw <= 0) return -200;
    const float fValue = 20 * log10f(raw);
    if (fValue < -200) return -200;
    return fValue;
}

float zlo_add_float_db(float db1, float db2)                    /* :234-236 : pow(int, float) -> double pow */
{
    return 10 * log10f((float)(pow(10, db1 / 10) + pow(10, db2 / 10)));
}

int32_t zlo_sample_to_peak_int(float x)                         /* :367 : abs(131072.f * x) -> int */
{
    const float v = fabsf(131072.0f * x);
    /* |v| >= 2^31 and NaN are undefined in the reference (float -> int); the build saturates / maps NaN to 0 */
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    return (int32_t)v;
}

void zlo_levels_tick(zlo_levels_channel *c, const float *L, const float *R, uint32_t n, int with_hold)   /* :347-398 */
{
    static const float intToFloatMultiplier = 0.00000152587;    /* :349 (Q12: 0.2/131072) */
    c->peakA = c->peakA - 10000 > 0 ? c->peakA - 10000 : 0;     /* :359-360 */
    c->peakB = c->peakB - 10000 > 0 ? c->peakB - 10000 : 0;
    if (n > 0) {                                                /* :361-384 */
        for (uint32_t i = 0; i < n; ++i) { const int32_t s = zlo_sample_to_peak_int(L[i]); if (s > c->peakA) c->peakA = s; }
        for (uint32_t i = 0; i < n; ++i) { const int32_t s = zlo_sample_to_peak_int(R[i]); if (s > c->peakB) c->peakB = s; }
    }
    const float peakA = c->peakA * intToFloatMultiplier, peakB = c->peakB * intToFloatMultiplier;   /* :385 */
    c->peakDbA = zlo_convert_to_dbfs(peakA);                    /* :386-387 */
    c->peakDbB = zlo_convert_to_dbfs(peakB);
    c->combinedDb = zlo_add_float_db(c->peakDbA, c->peakDbB);   /* :394 / :406 */
    if (with_hold) {                                            /* :395-398 */
        c->peakAHoldSignal = (peakA >= c->peakAHoldSignal) ? peakA : c->peakAHoldSignal * 0.9f;
        c->peakBHoldSignal = (peakB >= c->peakBHoldSignal) ? peakB : c->peakBHoldSignal * 0.9f;
        c->holdDbA = zlo_convert_to_dbfs(c->peakAHoldSignal);
        c->holdDbB = zlo_convert_to_dbfs(c->peakBHoldSignal);
    }
}

/* RMS extension (build-defined, absent in the reference; north_star "AudioLevels RMS/peak").  The order of the sum
 * of squares is part of the definition, so that every implementation gives the same bits:
 *   - the block is cut into tiles of 64 consecutive frames starting at frame `off` (off = 1 in the modes with quirk Q2,
 *     where frame 0 of a bus is the constant 0 and frame f of a voice lands in out[f + 1]; off = 0 with
 *     ZLO_MODE_FIX_DELAY); frames past the end count as 0;
 *   - a tile is summed by the balanced pairwise tree over its 64 squares (x0^2 + x1^2) + (x2^2 + x3^2) ... -- the
 *     tree a 64-lane wavefront reduction forms;
 *   - the tiles are added one after the other in tile order, starting from the sum of the squares of the frames before
 *     `off` (0 or x[0]^2). */
float zlo_block_sumsq(const float *x, uint32_t n, uint32_t off)
{
    float acc = 0.0f;
    for (uint32_t i = 0; i < off && i < n; ++i) acc += x[i] * x[i];
    for (uint32_t t0 = off; t0 < n; t0 += 64) {
        float a[64];
        for (uint32_t l = 0; l < 64; ++l) { const float v = (t0 + l < n) ? x[t0 + l] : 0.0f; a[l] = v * v; }
        for (uint32_t s_ = 1; s_ < 64; s_ *= 2)
            for (uint32_t i = 0; i < 64; i += 2 * s_) a[i] = a[i] + a[i + s_];
        acc += a[0];
    }
    return acc;
}

float zlo_block_rms(const float *x, uint32_t n, uint32_t off)
{
    return n ? sqrtf(zlo_block_sumsq(x, n, off) / (float)n) : 0.0f;
}

/* =============================== recorder sample format ====================================== */
/* 16-bit WAV samples as the reference's recorder writes them: AudioLevels.cpp:53-58 creates a juce::WavAudioFormat writer with
 * bitRate 16 and feeds it float blocks through AudioFormatWriter::ThreadedWriter (:72-76).  JUCE is not in the tree;
 * restated from its public source, version unpinned: AudioFormatWriter::write converts floats to 32-bit fixed point
 * (x <= -1 -> INT_MIN, x >= 1 -> INT_MAX, else roundToInt(INT_MAX * (double) x), round half to even) and the WAV writer stores
 * the upper 16 bits little-endian.  NaN (undefined in JUCE) is written as 0.  Stereo frames interleaved L, R. */
int16_t zlo_pcm16_sample(float x)
{
    const double d = (double)x;
    int32_t q;
    if (d <= -1.0) q = INT32_MIN;
    else if (d >= 1.0) q = INT32_MAX;
    else if (d != d) q = 0;
    else q = (int32_t)nearbyint(2147483647.0 * d);
    return (int16_t)(q >> 16);
}

void zlo_pcm16_stereo(const float *L, const float *R, uint32_t n, int16_t *out)
{
    for (uint32_t i = 0; i < n; ++i) { out[2 * i] = zlo_pcm16_sample(L[i]); out[2 * i + 1] = zlo_pcm16_sample(R[i]); }
}

/* =============================== JackPassthrough ============================================= */

void zlo_passthrough_init(zlo_passthrough *p)                   /* JackPassthrough.cpp:27-31 */
{
    p->dryAmount = 1.0f; p->wetFx1Amount = 1.0f; p->wetFx2Amount = 1.0f; p->panAmount = 0.0f; p->muted = 0;
}

void zlo_passthrough_process(const zlo_passthrough *p, const float *inL, const float *inR, float *const out[6], uint32_t nframes)   /* :45-115 */
{
    const size_t bytes = nframes * sizeof(float);
    if (p->muted) {                                             /* :55-61 */
        for (int k = 0; k < 6; ++k) memset(out[k], 0, bytes);
        return;
    }
    const float amounts[3] = { p->dryAmount, p->wetFx1Amount, p->wetFx2Amount };
    int output[3] = { 1, 1, 1 };
    for (int k = 0; k < 3; ++k) {                               /* :66-92 */
        if (p->panAmount == 0 && amounts[k] == 0) {
            output[k] = 0; memset(out[2 * k], 0, bytes); memset(out[2 * k + 1], 0, bytes);
        } else if (p->panAmount == 0 && amounts[k] == 1) {
            output[k] = 0; memcpy(out[2 * k], inL, bytes); memcpy(out[2 * k + 1], inR, bytes);
        }
    }
    if (p->panAmount != 0 || output[0] || output[1] || output[2]) {   /* :93-112 */
        const float la = 1 - p->panAmount, ra = 1 + p->panAmount;
        const float lm = (1.0f < la) ? 1.0f : la;               /* std::min(1 - panAmount, 1.0f) = (b < a) ? b : a */
        const float rm = (1.0f < ra) ? 1.0f : ra;
        for (uint32_t f = 0; f < nframes; ++f) {
            const float sl = inL[f], sr = inR[f];
            for (int k = 0; k < 3; ++k) {
                if (p->panAmount != 0 || output[k]) {
                    out[2 * k][f]     = amounts[k] * sl * lm;
                    out[2 * k + 1][f] = amounts[k] * sr * rm;
                }
            }
        }
    }
}

/* =============================== batch driver ================================================ */

typedef struct batch_job {
    zlo_channel *channels; int32_t bus_begin, bus_end;
    const zlo_sound *sounds; zlo_clip *clips; const zlo_clock *clocks;
    uint32_t nblocks, nframes, mode; int32_t mix_group;
    float *busL, *busR; zlo_report *reports;
    int64_t now_ms;                 /* the wall clock of the render (positions model: lastUpdated) */
} batch_job;

static void render_bus_block(zlo_channel *ch, float *L, float *R, uint32_t nframes, const zlo_clock *clk,
                             const zlo_sound *sounds, zlo_clip *clips, uint32_t mode, int32_t mix_group,
                             zlo_report *reports, float *tmpL, float *tmpR, int64_t now_ms)
{
    if (mix_group <= 0 || mix_group >= ch->nvoices) {
        zlo_channel_process(ch, L, R, nframes, clk, sounds, clips, mode, now_ms, reports);
        return;
    }
    /* engine summation order: consecutive groups of mix_group voices are summed sequentially into a
     * zeroed partial, then the partials of groups that rendered anything are added in group order */
    if (reports) memset(reports, 0, sizeof(zlo_report) * (size_t)ch->nvoices);
    if (!ch->enabled) return;
    memset(L, 0, nframes * sizeof(float));
    memset(R, 0, nframes * sizeof(float));
    for (int g0 = 0; g0 < ch->nvoices; g0 += mix_group) {
        const int g1 = g0 + mix_group < ch->nvoices ? g0 + mix_group : ch->nvoices;
        memset(tmpL, 0, nframes * sizeof(float));
        memset(tmpR, 0, nframes * sizeof(float));
        for (int i = g0; i < g1; ++i)
            if (ch->voices[i].isPlaying)
                zlo_voice_process(&ch->voices[i], tmpL, tmpR, nframes, clk, sounds, clips, mode, now_ms, reports ? &reports[i] : NULL, NULL);
        for (uint32_t f = 0; f < nframes; ++f) { L[f] += tmpL[f]; R[f] += tmpR[f]; }
    }
}

static void *batch_worker(void *arg)
{
    batch_job *j = (batch_job *)arg;
    const size_t total = (size_t)j->nblocks * j->nframes;
    float *tmpL = (float *)malloc(sizeof(float) * j->nframes * 2), *tmpR = tmpL + j->nframes;
    for (int32_t b = j->bus_begin; b < j->bus_end; ++b) {
        zlo_channel *ch = &j->channels[b];
        zlo_report *rp = NULL;
        for (uint32_t k = 0; k < j->nblocks; ++k) {
            if (j->reports && k + 1 == j->nblocks) {
                /* reports are laid out [bus][voice]; buses may have different voice counts only if the
                 * caller sized the array by the running sum -- the driver assumes a uniform nvoices */
                rp = j->reports + (size_t)b * ch->nvoices;
            }
            render_bus_block(ch, j->busL + b * total + (size_t)k * j->nframes, j->busR + b * total + (size_t)k * j->nframes,
                             j->nframes, &j->clocks[k], j->sounds, j->clips, j->mode, j->mix_group, rp, tmpL, tmpR, j->now_ms);
        }
    }
    free(tmpL);
    return NULL;
}

void zlo_render_batch(zlo_channel *channels, int32_t nbuses, const zlo_sound *sounds, zlo_clip *clips,
                      const zlo_clock *clocks, uint32_t nblocks, uint32_t nframes, uint32_t mode,
                      int32_t mix_group, float *busL, float *busR, zlo_report *reports, int32_t threads)
{
    zlo_render_batch_at(channels, nbuses, sounds, clips, clocks, nblocks, nframes, mode, mix_group, busL, busR, reports, threads, 0);
}

/* ... with the wall clock the positions models stamp their rows with (QDateTime::currentMSecsSinceEpoch() in
 * ClipAudioSourcePositionsModel.cpp:131,191-209: rows not updated for a second are taken for orphans at the next create / remove) */
void zlo_render_batch_at(zlo_channel *channels, int32_t nbuses, const zlo_sound *sounds, zlo_clip *clips,
                         const zlo_clock *clocks, uint32_t nblocks, uint32_t nframes, uint32_t mode,
                         int32_t mix_group, float *busL, float *busR, zlo_report *reports, int32_t threads, int64_t now_ms)
{
    if (threads < 1) threads = 1;
    if (threads > nbuses) threads = nbuses;
    batch_job jobs[256];
    pthread_t tids[256];
    if (threads > 256) threads = 256;
    for (int t = 0; t < threads; ++t) {
        batch_job *j = &jobs[t];
        j->channels = channels; j->bus_begin = (int32_t)((int64_t)nbuses * t / threads); j->bus_end = (int32_t)((int64_t)nbuses * (t + 1) / threads);
        j->sounds = sounds; j->clips = clips; j->clocks = clocks; j->nblocks = nblocks; j->nframes = nframes; j->mode = mode;
        j->mix_group = mix_group; j->busL = busL; j->busR = busR; j->reports = reports; j->now_ms = now_ms;
    }
    if (threads == 1) { batch_worker(&jobs[0]); return; }
    for (int t = 0; t < threads; ++t) pthread_create(&tids[t], NULL, batch_worker, &jobs[t]);
    for (int t = 0; t < threads; ++t) pthread_join(tids[t], NULL);
}
