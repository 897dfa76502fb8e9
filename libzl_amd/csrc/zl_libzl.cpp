// zl_libzl.cpp -- the libzl.h-named C-ABI of the hot path (include/libzl_hotpath.h) on top of the engine.
//
// Plain C++ restatement of the control-plane pieces of the reference that sit directly on the hot path:
//   ClipAudioSource parameter setters / getters   lib/ClipAudioSource.cpp:255-277,313-367,495-528,606-700
//   ClipAudioSource::play / stop                    lib/ClipAudioSource.cpp:415-455
//   ClipAudioSourcePositionsModel                   lib/ClipAudioSourcePositionsModel.cpp:78-209
//   level / progress callbacks                      lib/ClipAudioSource.cpp:88-113,225-240
//   JackPassthrough parameter bridge                lib/libzl.cpp:476-575
// No Qt, JUCE or tracktion: sources are decoded by the RIFF/WAVE reader below, audio is rendered by
// the HIP kernels behind zlhip_render, and this file only keeps parameters and forwards commands.
// -DZLHIP_NO_LIBZL_NAMES leaves this translation unit empty: libzlhip.so then exports only the zlhip_* engine ABI and a libzl
// build keeps its own ClipAudioSource_* / JackPassthrough_* bodies, calling the engine from them.
#ifndef ZLHIP_NO_LIBZL_NAMES
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/libzl_hotpath.h"
#include "zl_render.h"       // zl_pcm16: the recorder's 16-bit sample format
#include "zl_sched.h"        // the ClipCommand scheduling front-end (SyncTimer.cpp:452-702,1011-1048)
#include "zl_handoff.h"      // what crosses from the callers' threads to the cycle: the request queue, the parameter snapshots

namespace {

// QDateTime::currentMSecsSinceEpoch() of the reference (ClipAudioSource.cpp:89,111,226,238; ClipAudioSourcePositionsModel.cpp):
// the wall clock, or the host's clock when libzl_hotpath_set_clock_ms installed one (deterministic tests, offline bounces)
int64_t (*g_clock_ms)(void) = nullptr;
int64_t now_ms()
{
    if (g_clock_ms) return g_clock_ms();
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

// ---- tracktion_engine volume fader curve (third-party, absent from /root/reference; restated from the
//      public tracktion_engine source, version unpinned): used only by setVolume(dB) / dBFromVolume ----
float decibelsToVolumeFaderPosition(float db) { return (db > -100.0f) ? std::exp((db - 6.0f) * (1.0f / 20.0f)) : 0.0f; }
float volumeFaderPositionToDB(float pos) { return (pos > 0.0f) ? (20.0f * std::log(pos)) + 6.0f : -100.0f; }
// juce::Decibels (third-party template, restated from the public JUCE source): instantiated on the ARGUMENT type -- float for
// positionsModel->peakGain() (ClipAudioSource.cpp:92), double for prevLeveldB and prevLevel * 0.94 (:98,101)
float gainToDecibels(float gain) { return gain > 0.0f ? std::max(-100.0f, (float)std::log10(gain) * 20.0f) : -100.0f; }
double gainToDecibels(double gain) { return gain > 0.0 ? std::max(-100.0, std::log10(gain) * 20.0) : -100.0; }
double decibelsToGain(double db) { return db > -100.0 ? std::pow(10.0, db * 0.05) : 0.0; }

struct AdsrParams { float attack = 0.1f, decay = 0.1f, sustain = 1.0f, release = 0.1f; };   // juce::ADSR::Parameters defaults

// ClipAudioSourcePositionsModel (ClipAudioSourcePositionsModel.cpp)
struct PositionsModel {
    static const int COUNT = 32;                                   // :5
    struct Row { int64_t id = -1; float progress = 0.0f, gain = 0.0f; int64_t lastUpdated = 0; } rows[COUNT];
    bool updatePeakGain = false;
    float peak = 0.0f;
    void cleanUp(int64_t now)                                      // :191-209
    {
        for (Row &r : rows) if (r.id > -1 && r.lastUpdated < now - 1000) { r.id = -1; r.gain = 0.0f; r.progress = 0.0f; }
    }
    int64_t create(float initialProgress, int64_t now)             // :78-100
    {
        int row = -1; Row *pos = nullptr;
        for (Row &r : rows) { ++row; if (r.id == -1) { pos = &r; break; } }
        if (pos) { pos->id = row; pos->progress = initialProgress; pos->lastUpdated = now; updatePeakGain = true; cleanUp(now); }
        return row;
    }
    void set(int64_t id, float gain, float progress, int64_t now)  // :126-138
    {
        if (id > -1 && id < COUNT) { rows[id].gain = gain; rows[id].progress = progress; rows[id].lastUpdated = now; updatePeakGain = true; }
    }
    void remove(int64_t id, int64_t now)                           // :140-153
    {
        if (id > -1 && id < COUNT) { rows[id].id = -1; rows[id].gain = 0.0f; rows[id].progress = 0.0f; updatePeakGain = true; }
        cleanUp(now);
    }
    float peakGain()                                               // :160-173
    {
        if (updatePeakGain) {
            float p = 0.0f;
            for (Row &r : rows) p = std::max(p, r.gain);
            if (std::fabs((double)(peak - p)) > 0.01) peak = p;
            updatePeakGain = false;
        }
        return peak;
    }
    double firstProgress() const                                   // :175-185
    {
        for (const Row &r : rows) if (r.id > -1) return r.progress;
        return -1.0;
    }
};

// JackPassthrough.cpp:27-31.  Written by the host's setters on any thread, read by the cycle (libzl_hotpath_*_fanout) field by field,
// as the reference's process() reads the members its setters store: relaxed atomics, no lock on either side.
struct PassState { std::atomic<float> dry{1.0f}, fx1{1.0f}, fx2{1.0f}, pan{0.0f}; std::atomic<bool> muted{false}; };

}  // namespace

struct ClipAudioSource {
    int id = 0;
    int engineClip = -1;
    std::string fileName, filePath;
    double sourceSampleRate = 0.0;
    int lengthFrames = 0;
    std::mutex setMu;                                              // setters only
    // ClipAudioSource::Private (ClipAudioSource.cpp:63-82)
    float startPositionInSeconds = 0;
    float lengthInSeconds = -1;
    float lengthInBeats = -1;
    float volumeAbsolute = 1.0f;
    float pitchChange = 0, speedRatio = 1.0f, gainDb = 0;
    float pan = 0.0f;
    float duration = 0.0f;
    int slices = 0;
    std::vector<double> slicePositions;
    int sliceBaseMidiNote = 60, keyZoneStart = 0, keyZoneEnd = 127, rootNote = 60;
    AdsrParams adsr;
    ZlSnapshot<zlhip_clip_params> snap;
    // real-time side
    PositionsModel positions;
    std::atomic<float> startPositionRt{0.0f};                      // startPositionInSeconds as the cycle reads it (syncProgress, :227)
    std::atomic<void (*)(float)> progressCb{nullptr}, levelCb{nullptr};
    std::atomic<float> peakGainOut{0.0f};                          // positionsModel->peakGain() / firstProgress() after the last cycle
    std::atomic<double> firstProgressOut{-1.0};
    double currentLeveldB = -400.0, prevLeveldB = -400.0, firstPositionProgress = 0.0;
    int64_t nextGainUpdateTime = 0, nextPositionUpdateTime = 0;
};

namespace {

// ClipAudioSource::play / stop call SyncTimer::scheduleClipCommand on the caller's thread in the reference (the step lists are QLists
// touched without a lock); here the caller only posts the request (zl_handoff.h) and the cycle applies it -- in arrival order, before
// it looks at the steps that are due -- so the scheduler's state belongs to one thread and the caller never holds a lock the
// real-time thread wants.
struct Request {
    enum Kind : int32_t { Schedule, QueueStart, QueueStop, TimerStart, TimerStop, SetBpm, TimerTick, ChannelEnabled } kind;
    int32_t a, b;                                                  // QueueStart/Stop: clip, channel; TimerStart / SetBpm: bpm; ChannelEnabled: channel, flag
    uint64_t delay;
    zlhip_clip_command cmd;
};

struct Global {
    std::mutex mu;                             // the engine, the clip list, the scheduler: held by a cycle; by create / destroy / init / shutdown
    zlhip_config cfg;
    bool cfgSet = false;
    zlhip_engine *engine = nullptr;
    int status = ZLHIP_ERR_STATE;
    std::vector<ClipAudioSource *> clips;      // createdClips, libzl.cpp:126
    std::vector<ClipAudioSource *> byEngineClip; // engine clip id -> clip (live or parked)
    int nextClipId = 1;                        // libzl.cpp:122
    std::vector<int64_t> voicePositionId;      // per voice slot: row in its clip's positions model
    std::vector<int32_t> dueVoices;            // per command of a dispatched step: the voice it started (zlhip_handle_commands_voices)
    std::vector<ClipAudioSource *> voiceClip;  // per voice slot: clip being played (host view)
    std::vector<zlhip_voice_report> reports;
    ZlRequestQueue<Request, 4096> requests;          // FreshCommandStashSize, SyncTimer.cpp:252
    ZlStepSequencer seq;                       // SyncTimer's step ring (libzl_hotpath_cycle)
    ZlHostTransportSchedule ext;               // the host's SyncTimer owns the transport (libzl_hotpath_process)
    std::vector<ZlDispatch> due;               // commands of the steps due in this cycle
    std::vector<zlhip_clip_command> dueBatch;
    uint32_t lastNframes = 0;
    std::atomic<uint64_t> dropped{0};          // requests lost to a full queue
    std::mutex cbMu;                           // SyncTimer::addCallback / removeCallback (SyncTimer.cpp:790-812): ticks for the host's sequencer
    void (*timerCallbacks[16])(int) = {};      // CallbackSpaces, SyncTimer.cpp:249
    std::vector<int> beats;                    // beats the timer ticked through in this cycle (hiResTimerCallback, :397-399)
    PassState pass[11];                        // [0] GlobalPlayback (channel -1), [1..10] channels 0..9 (MidiRouter.cpp:876-883)
    std::vector<zlhip_passthrough_params> passNow;   // per bus: the passthrough parameters of the cycle being rendered
} G;

struct PendingCallback { void (*fn)(float); float value; };

// setters: the clip's fields -> its published snapshot (call with c->setMu held)
void fill_params(const ClipAudioSource *c, zlhip_clip_params &p)
{
    std::memset(&p, 0, sizeof p);
    p.start_position_seconds = c->startPositionInSeconds;
    p.length_seconds = c->lengthInSeconds;
    p.length_in_beats = c->lengthInBeats;
    p.volume_absolute = c->volumeAbsolute;
    p.pan = c->pan;
    p.duration_seconds = c->duration;
    p.adsr_attack = c->adsr.attack; p.adsr_decay = c->adsr.decay; p.adsr_sustain = c->adsr.sustain; p.adsr_release = c->adsr.release;
    p.root_note = c->rootNote;
    p.num_slice_positions = (int32_t)std::min<size_t>(c->slicePositions.size(), ZLHIP_MAX_SLICES);
    for (int i = 0; i < p.num_slice_positions; ++i) p.slice_positions[i] = c->slicePositions[(size_t)i];
}

void publish_params(ClipAudioSource *c)
{
    zlhip_clip_params p;
    fill_params(c, p);
    c->startPositionRt.store(c->startPositionInSeconds, std::memory_order_relaxed);
    c->snap.publish(p);
}

// the cycle: a dirty snapshot -> the engine.  Wait-free: a snapshot caught in the middle of a write stays dirty for the next cycle.
void apply_params(ClipAudioSource *c)
{
    if (!G.engine || c->engineClip < 0) return;
    zlhip_clip_params p;
    if (c->snap.take(p)) zlhip_clip_set(G.engine, c->engineClip, &p);
}

void set_slices(ClipAudioSource *c, int slices)                    // ClipAudioSource.cpp:495-528
{
    if (c->slices == slices) return;
    if (slices == 0) {
        c->slicePositions.clear();
    } else if (c->slices > slices) {
        while ((int)c->slicePositions.size() > slices) c->slicePositions.pop_back();
    } else {
        double lastSlicePosition = 0.0f;
        if (!c->slicePositions.empty()) lastSlicePosition = c->slicePositions.back();
        double positionIncrement = (1.0f - lastSlicePosition) / (slices - c->slices);
        double newPosition = lastSlicePosition + positionIncrement;
        if (c->slicePositions.empty()) c->slicePositions.push_back(0.0f);
        while ((int)c->slicePositions.size() < slices) { c->slicePositions.push_back(newPosition); newPosition += positionIncrement; }
    }
    c->slices = slices;
}

float subbeat_count_to_seconds(uint64_t bpm, uint64_t beats)       // SyncTimer.cpp:180-183,936-939
{
    bpm = std::min<uint64_t>(std::max<uint64_t>(bpm, 50), 200);     // qBound(BPM_MINIMUM, bpm, BPM_MAXIMUM)
    const uint64_t ns = (beats * 60000000000ULL) / (bpm * (uint64_t)ZLHIP_BEAT_SUBDIVISIONS);
    return ns / (float)1000000000;
}

uint64_t f32_to_u64_sat(float f)
{
    if (!(f > 0.0f)) return 0;
    if (f >= 18446744073709551616.0f) return ~0ull;
    return (uint64_t)f;
}

// call with G.mu held
ClipAudioSource *make_clip(const float *L, const float *R, int length, double sr, const char *path)
{
    ClipAudioSource *c = new ClipAudioSource();
    c->filePath = path ? path : "";
    const size_t slash = c->filePath.find_last_of('/');
    c->fileName = slash == std::string::npos ? c->filePath : c->filePath.substr(slash + 1);
    c->sourceSampleRate = sr;
    c->lengthFrames = length;
    c->duration = (float)(length / sr);                            // edit->getLength(), ClipAudioSource.cpp:158,367
    c->lengthInSeconds = c->duration;                              // :158
    c->adsr.attack = 0.0f; c->adsr.release = 0.05f;                // :164-168
    set_slices(c, 16);                                             // :204
    if (G.engine) {
        int32_t id = -1;
        const int rc = zlhip_sound_upload(G.engine, L, R, length, sr, &id);
        if (rc == ZLHIP_OK) c->engineClip = id;
        else {
            // the reference logs and carries on (libzl.cpp has no error returns); a clip without a source would be silent
            // for ever, so the bridge refuses it: ClipAudioSource_new* return NULL
            std::fprintf(stderr, "libzl hot path: cannot load %s into the engine: %s (%s)\n", c->filePath.c_str(), zlhip_strerror(rc), zlhip_last_error(G.engine));
            delete c;
            return nullptr;
        }
    }
    c->id = G.nextClipId++;                                        // libzl.cpp:122-124
    G.clips.push_back(c);
    if (c->engineClip >= 0) {
        if ((size_t)c->engineClip >= G.byEngineClip.size()) G.byEngineClip.resize((size_t)c->engineClip + 1, nullptr);
        G.byEngineClip[(size_t)c->engineClip] = c;
    }
    { std::lock_guard<std::mutex> sl(c->setMu); publish_params(c); }
    apply_params(c);
    return c;
}

void post(const Request &r)
{
    if (!G.requests.push(r)) {
        if (G.dropped.fetch_add(1, std::memory_order_relaxed) == 0) std::fprintf(stderr, "libzl hot path: request queue full (no cycle is draining it): requests are dropped\n");
    }
}

// SyncTimer::scheduleClipCommand(command, delay) from any thread
void schedule_command(const zlhip_clip_command &cmd, uint64_t delay)
{
    Request r; std::memset(&r, 0, sizeof r);
    r.kind = Request::Schedule; r.delay = delay; r.cmd = cmd;
    post(r);
}

zlhip_clip_command channel_command(ClipAudioSource *c, int midiChannel)   // ClipCommand::channelCommand, ClipCommand.h:66-72
{
    zlhip_clip_command cmd;
    zlhip_clip_command_clear(&cmd);
    cmd.clip = c->engineClip;
    cmd.midi_channel = midiChannel;
    return cmd;
}

void clip_play(ClipAudioSource *c, bool loop, int midiChannel)     // ClipAudioSource::play, ClipAudioSource.cpp:415-429
{
    zlhip_clip_command cmd = channel_command(c, midiChannel);
    cmd.midi_note = 60;
    cmd.change_volume = 1;
    cmd.volume = 1.0f;
    cmd.looping = loop ? 1 : 0;
    if (loop) cmd.stop_playback = 1;                               // stops any current loop plays, then starts a new one
    cmd.start_playback = 1;
    schedule_command(cmd, 0);                                      // d->syncTimer->scheduleClipCommand(command, 0), :428
}

void clip_stop(ClipAudioSource *c, int midiChannel)                // ClipAudioSource::stop, ClipAudioSource.cpp:431-455
{
    if (midiChannel > -3) {
        zlhip_clip_command cmd = channel_command(c, midiChannel);
        cmd.midi_note = 60; cmd.stop_playback = 1;
        schedule_command(cmd, 0);
    } else {
        zlhip_clip_command cmd = channel_command(c, -2);           // noEffectCommand: midi note 60 (ClipCommand.h:44-51)
        cmd.midi_note = 60; cmd.stop_playback = 1;
        schedule_command(cmd, 0);
        cmd = channel_command(c, -1);                              // effectedCommand
        cmd.midi_note = 60; cmd.stop_playback = 1;
        schedule_command(cmd, 0);
        for (int i = 0; i < 10; ++i) {
            cmd = channel_command(c, i);
            cmd.midi_note = 60; cmd.stop_playback = 1;
            schedule_command(cmd, 0);
        }
    }
}

void sync_audio_level(ClipAudioSource *c, int64_t now, std::vector<PendingCallback> &cbs)   // ClipAudioSource::Private::syncAudioLevel, :88-113
{
    if (c->nextGainUpdateTime < now) {
        c->prevLeveldB = c->currentLeveldB;
        c->currentLeveldB = gainToDecibels(c->positions.peakGain());   // the tracktion LevelMeasurer client stays silent here
        const double prevLevel = decibelsToGain(c->prevLeveldB);
        if (c->prevLeveldB > c->currentLeveldB) c->currentLeveldB = gainToDecibels(prevLevel * 0.94);   // double form, :100-101
        void (*cb)(float) = c->levelCb.load(std::memory_order_acquire);
        if (std::fabs(c->currentLeveldB - c->prevLeveldB) > 0.1 && cb) cbs.push_back(PendingCallback{ cb, (float)c->currentLeveldB });
        c->nextGainUpdateTime = now + 30;
    }
}

void sync_progress(ClipAudioSource *c, int64_t now, std::vector<PendingCallback> &cbs)      // ClipAudioSource::syncProgress, :225-240
{
    if (c->nextPositionUpdateTime < now) {
        void (*cb)(float) = c->progressCb.load(std::memory_order_acquire);
        double newPosition = c->startPositionRt.load(std::memory_order_relaxed) / c->duration;
        if (cb != nullptr && c->positions.firstProgress() > -1.0f) newPosition = c->positions.firstProgress();
        if (std::fabs(c->firstPositionProgress - newPosition) > 0.001) {
            c->firstPositionProgress = newPosition;
            if (cb) cbs.push_back(PendingCallback{ cb, (float)(c->firstPositionProgress * c->duration) });
            c->nextPositionUpdateTime = now + 100;
        }
    }
}

ClipAudioSource *clip_by_engine_id(int engineClip)                 // (a table: the cycle asks once per playing voice)
{
    return (engineClip >= 0 && (size_t)engineClip < G.byEngineClip.size()) ? G.byEngineClip[(size_t)engineClip] : nullptr;
}

PassState *pass_for(int channel)                                   // libzl.cpp:476-575 channel mapping
{
    if (channel == -1) return &G.pass[0];
    if (channel > -1 && channel < 10) return &G.pass[channel + 1];
    return nullptr;
}

bool bpm_usable(uint64_t bpm) { return bpm > 0 && bpm * (uint64_t)ZLHIP_BEAT_SUBDIVISIONS <= 60000000000ULL; }

// the requests posted since the last cycle, in arrival order, into the scheduler that serves this cycle (call with G.mu held)
void drain_requests(bool internalTransport)
{
    Request r;
    while (G.requests.pop(r)) {
        switch (r.kind) {
        case Request::Schedule:
            if (internalTransport) G.seq.scheduleClipCommand(r.cmd, r.delay); else G.ext.scheduleClipCommand(r.cmd, r.delay);
            break;
        case Request::QueueStart:
            if (internalTransport) G.seq.queueClipToStartOnChannel(r.a, r.b);
            else {                                                 // (a host-owned transport queues through its own SyncTimer; taken as "paused": now)
                zlhip_clip_command c; ZlStepSequencer::zlhip_clip_command_clear_inline(c);
                c.clip = r.a; c.midi_channel = r.b; c.midi_note = 60; c.change_volume = 1; c.volume = 1.0f; c.looping = 1; c.stop_playback = 1; c.start_playback = 1;
                G.ext.scheduleClipCommand(c, 0);
            }
            break;
        case Request::QueueStop:
            if (internalTransport) G.seq.queueClipToStopOnChannel(r.a, r.b);
            else {
                for (auto &st : G.ext.steps)
                    for (size_t i = 0; i < st.clipCommands.size(); ++i) if (st.clipCommands[i].clip == r.a) { st.clipCommands.erase(st.clipCommands.begin() + (long)i); break; }
                zlhip_clip_command c; ZlStepSequencer::zlhip_clip_command_clear_inline(c);
                c.clip = r.a; c.midi_channel = r.b; c.midi_note = 60; c.stop_playback = 1;
                size_t i = 0;
                while (i < G.ext.steps.size() && G.ext.steps[i].due < G.ext.lastPlayhead) ++i;
                if (i == G.ext.steps.size() || G.ext.steps[i].due != G.ext.lastPlayhead) G.ext.steps.insert(G.ext.steps.begin() + (long)i, ZlHostTransportSchedule::Step{ G.ext.lastPlayhead, {} });
                G.ext.steps[i].clipCommands.push_back(c);          // appended, not merged (SyncTimer.cpp:858-859)
            }
            break;
        // (a bpm of 0 -- or one so large that a subbeat has no nanoseconds, 60e9 / (bpm * 96) == 0 -- divides by zero in the step clock,
        // here on the audio thread where the reference would fault on the caller's: such a request is ignored.  Every other value is
        // taken as it is, as the reference does -- it clamps to 50..200 only when the SetBpm timer command plays, SyncTimer.cpp:606-612)
        case Request::TimerStart: if (internalTransport && bpm_usable((uint64_t)(uint32_t)r.a)) G.seq.start(r.a); break;
        case Request::TimerStop:  if (internalTransport) G.seq.stop(); break;
        case Request::SetBpm:     if (internalTransport && bpm_usable((uint64_t)(uint32_t)r.a)) G.seq.setBpm((uint64_t)(uint32_t)r.a); break;
        case Request::TimerTick:  if (internalTransport) { G.seq.beatSink = &G.beats; G.seq.hi_res_timer_callback(); } break;
        case Request::ChannelEnabled: if (G.engine) (void)zlhip_bus_set_enabled(G.engine, r.a + 2, r.b); break;   // (an unknown channel is ignored, SamplerSynth.cpp:345)
        }
    }
}

// the commands that fell due -> the engine: one zlhip_handle_commands call per run of equal ticks (a step), i.e. one voice-table
// update on the device however many commands the cycle carries (SamplerSynth::handleClipCommand per command, SyncTimer.cpp:553-558)
int dispatch_due(bool trackPositions = true)   // (false: the offline bounce, which does not drive the positions models)
{
    size_t i = 0;
    while (i < G.due.size()) {
        size_t j = i;
        G.dueBatch.clear();
        while (j < G.due.size() && G.due[j].tick == G.due[i].tick) G.dueBatch.push_back(G.due[j++].cmd);
        G.dueVoices.assign(G.dueBatch.size(), -1);
        const int rc = zlhip_handle_commands_voices(G.engine, G.dueBatch.data(), (int32_t)G.dueBatch.size(), G.due[i].tick, nullptr, G.dueVoices.data());
        if (rc < 0) return rc;
        // startNote creates the voice's row in its clip's positions model right here, command by command (SamplerSynthVoice.cpp:126-129)
        // -- before any voice of the cycle is processed; which row a voice gets, and so whose progress firstProgress() reports,
        // depends on that order
        const int64_t now = now_ms();
        for (size_t k = 0; trackPositions && k < G.dueBatch.size(); ++k) {
            const int v = G.dueVoices[k];
            if (v < 0 || v >= (int)G.voiceClip.size()) continue;
            ClipAudioSource *c = clip_by_engine_id(G.dueBatch[k].clip);
            if (G.voiceClip[(size_t)v]) G.voiceClip[(size_t)v]->positions.remove(G.voicePositionId[(size_t)v], now);   // (a slot whose end this layer has not seen yet)
            G.voiceClip[(size_t)v] = c;
            G.voicePositionId[(size_t)v] = c ? c->positions.create(0.0f, now) : -1;
        }
        i = j;
    }
    G.due.clear();
    return ZLHIP_OK;
}

// everything of a cycle behind the command dispatch: render, positions models, level / progress chains (call with G.mu held)
// The JackPassthrough client behind sampler channel `bus` (MidiRouter.cpp:876-883: "GlobalPlayback" behind channel -1 = bus 1,
// "FXPassthrough-Channel1..10" behind channels 0..9 = buses 2..11), as this cycle finds its members.  Bus 0 (channel -2, the
// un-effected global channel) and buses beyond 11 have no client in the reference: they get a client at its defaults (every pair = the bus).
zlhip_passthrough_params pass_of_bus(int bus)
{
    zlhip_passthrough_params p;
    zlhip_passthrough_params_default(&p);
    if (bus >= 1 && bus <= 11) {
        const PassState &q = G.pass[bus - 1];
        p.dry_amount = q.dry.load(std::memory_order_relaxed); p.wet_fx1_amount = q.fx1.load(std::memory_order_relaxed);
        p.wet_fx2_amount = q.fx2.load(std::memory_order_relaxed); p.pan_amount = q.pan.load(std::memory_order_relaxed);
        p.muted = q.muted.load(std::memory_order_relaxed) ? 1 : 0;
    }
    return p;
}

int render_and_report(uint32_t nframes, const zlhip_clock *clock, float *out_left, float *out_right, std::vector<PendingCallback> &cbs, float *fan_out = nullptr)
{
    int rc;
    if (fan_out) {
        // the passthrough clients' members as of this cycle (the setters are plain stores on the host's threads: no lock, no HIP call,
        // the resident kernel stays where it is)
        G.passNow.resize((size_t)(G.cfgSet ? G.cfg.num_buses : 12));       // (initJuce: 12 channels unless libzl_hotpath_configure said otherwise)
        for (size_t b = 0; b < G.passNow.size(); ++b) G.passNow[b] = pass_of_bus((int)b);
        rc = zlhip_render_fanout(G.engine, (int32_t)nframes, clock, out_left, out_right, G.passNow.data(), fan_out);
    } else {
        rc = zlhip_render(G.engine, (int32_t)nframes, clock, out_left, out_right);
    }
    if (rc != ZLHIP_OK) return rc;
    const int V = (int)G.reports.size();
    rc = zlhip_voice_reports(G.engine, G.reports.data(), V);
    if (rc != ZLHIP_OK) return rc;
    const int64_t now = now_ms();
    // route the per-voice reports into the per-clip positions models (SamplerSynthVoice.cpp:126-129,154-158,265-267)
    for (int v = 0; v < V; ++v) {
        const zlhip_voice_report &r = G.reports[(size_t)v];
        ClipAudioSource *cur = r.playing ? clip_by_engine_id(r.clip) : nullptr;
        if (G.voiceClip[(size_t)v] != cur) {
            if (G.voiceClip[(size_t)v]) G.voiceClip[(size_t)v]->positions.remove(G.voicePositionId[(size_t)v], now);
            G.voiceClip[(size_t)v] = cur;
            G.voicePositionId[(size_t)v] = cur ? cur->positions.create(0.0f, now) : -1;
        }
        if (cur && r.valid) cur->positions.set(G.voicePositionId[(size_t)v], r.gain, r.progress, now);
    }
    for (size_t i = 0; i < G.clips.size();) {
        ClipAudioSource *c = G.clips[i];
        if (c->id < 0) {            // destroyed by the host: release once no voice plays it any more
            bool used = false;
            for (int v = 0; v < V; ++v) used = used || (G.reports[(size_t)v].playing && G.reports[(size_t)v].clip == c->engineClip);
            if (!used) {
                if (c->engineClip >= 0) { zlhip_sound_release(G.engine, c->engineClip); G.byEngineClip[(size_t)c->engineClip] = nullptr; }
                delete c;
                G.clips.erase(G.clips.begin() + (long)i);
                continue;
            }
        } else {
            sync_audio_level(c, now, cbs);
            sync_progress(c, now, cbs);
            c->peakGainOut.store(c->positions.peakGain(), std::memory_order_relaxed);
            c->firstProgressOut.store(c->positions.firstProgress(), std::memory_order_relaxed);
        }
        ++i;
    }
    return ZLHIP_OK;
}

// ---- RIFF / WAVE --------------------------------------------------------------------------------
uint32_t rd32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

}  // namespace

extern "C" {

// ---- WAV IO -------------------------------------------------------------------------------------
int libzl_wav_read(const char *path, float **left, float **right, int *length, double *sampleRate)
{
    if (!path || !left || !right || !length || !sampleRate) return ZLHIP_ERR_INVALID;
    *left = *right = nullptr; *length = 0; *sampleRate = 0.0;
    FILE *f = std::fopen(path, "rb");
    if (!f) return ZLHIP_ERR_INVALID;
    std::vector<unsigned char> buf;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz < 44) { std::fclose(f); return ZLHIP_ERR_INVALID; }
    buf.resize((size_t)sz);
    const size_t got = std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (got != buf.size() || std::memcmp(buf.data(), "RIFF", 4) || std::memcmp(buf.data() + 8, "WAVE", 4)) return ZLHIP_ERR_INVALID;
    int fmt = 0, channels = 0, bits = 0; uint32_t rate = 0; const unsigned char *data = nullptr; uint32_t dataBytes = 0;
    for (size_t pos = 12; pos + 8 <= buf.size();) {
        const uint32_t csz = rd32(&buf[pos + 4]);
        const unsigned char *body = buf.data() + pos + 8;
        if (!std::memcmp(&buf[pos], "fmt ", 4) && csz >= 16 && pos + 8 + 16 <= buf.size()) {
            fmt = rd16(body); channels = rd16(body + 2); rate = rd32(body + 4); bits = rd16(body + 14);
            if (fmt == 0xFFFE && csz >= 26 && pos + 8 + 26 <= buf.size()) fmt = rd16(body + 24);   // WAVE_FORMAT_EXTENSIBLE sub-format
        } else if (!std::memcmp(&buf[pos], "data", 4)) {
            data = body; dataBytes = (uint32_t)std::min<size_t>(csz, buf.size() - (pos + 8));
        }
        pos += 8 + (size_t)csz + (csz & 1);
    }
    if (!data || channels < 1 || rate == 0 || !(fmt == 1 || fmt == 3)) return ZLHIP_ERR_INVALID;
    // sample formats of the decode side: integer PCM 8 / 16 / 24 / 32 bits, IEEE float 32 / 64 bits
    if (!((fmt == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32)) || (fmt == 3 && (bits == 32 || bits == 64)))) return ZLHIP_ERR_INVALID;
    const int bytesPer = bits / 8;
    const int frames = (int)(dataBytes / (uint32_t)(bytesPer * channels));
    const int outCh = std::min(2, channels);                                   // jmin(2, numChannels), SamplerSynthSound.cpp:45
    float *planes[2] = { (float *)std::malloc(sizeof(float) * (size_t)std::max(frames, 1)),
                         outCh > 1 ? (float *)std::malloc(sizeof(float) * (size_t)std::max(frames, 1)) : nullptr };
    if (!planes[0] || (outCh > 1 && !planes[1])) { std::free(planes[0]); std::free(planes[1]); return ZLHIP_ERR_CAPACITY; }
    for (int i = 0; i < frames; ++i) {
        for (int c = 0; c < outCh; ++c) {
            const unsigned char *p = data + ((size_t)i * channels + c) * bytesPer;
            float v;
            if (fmt == 3) {
                if (bits == 32) { std::memcpy(&v, p, 4); }
                else { double d; std::memcpy(&d, p, 8); v = (float)d; }
            } else {
                // JUCE convention: integer PCM is widened to left-justified int32 and scaled by 1 / 0x7fffffff
                int32_t s;
                if (bits == 8) s = ((int32_t)p[0] - 128) << 24;
                else if (bits == 16) s = (int32_t)((uint32_t)rd16(p) << 16);
                else if (bits == 24) s = (int32_t)(((uint32_t)p[0] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 24));
                else s = (int32_t)rd32(p);
                v = (float)s * (1.0f / 2147483648.0f);        // JUCE writes 1.0f / 0x7fffffff, which is this float
            }
            planes[c][i] = v;
        }
    }
    *left = planes[0]; *right = planes[1]; *length = frames; *sampleRate = (double)rate;
    return ZLHIP_OK;
}

void libzl_wav_free(float *plane) { std::free(plane); }

static void wav_header(unsigned char *h, uint32_t frames, int ch, double sampleRate, int bitsPerSample)
{
    const int bytesPer = bitsPerSample / 8;
    const uint32_t dataBytes = frames * (uint32_t)ch * (uint32_t)bytesPer, rate = (uint32_t)sampleRate;
    auto w32 = [&](int o, uint32_t v) { h[o] = v & 255; h[o + 1] = (v >> 8) & 255; h[o + 2] = (v >> 16) & 255; h[o + 3] = (v >> 24) & 255; };
    auto w16 = [&](int o, uint16_t v) { h[o] = v & 255; h[o + 1] = (v >> 8) & 255; };
    std::memcpy(h, "RIFF", 4); w32(4, 36 + dataBytes); std::memcpy(h + 8, "WAVEfmt ", 8); w32(16, 16);
    w16(20, bitsPerSample == 32 ? 3 : 1); w16(22, (uint16_t)ch); w32(24, rate); w32(28, rate * ch * bytesPer); w16(32, (uint16_t)(ch * bytesPer));
    w16(34, (uint16_t)bitsPerSample); std::memcpy(h + 36, "data", 4); w32(40, dataBytes);
}

int libzl_wav_write(const char *path, const float *left, const float *right, int length, double sampleRate, int bitsPerSample)
{
    if (!path || !left || length < 0 || !(bitsPerSample == 16 || bitsPerSample == 32)) return ZLHIP_ERR_INVALID;
    const int ch = right ? 2 : 1;
    // interleave (and convert) in memory, then one write: the format of the reference's recorder (AudioLevels.cpp:53-58), zl_pcm16
    std::vector<unsigned char> data((size_t)length * ch * (bitsPerSample / 8));
    for (int i = 0; i < length; ++i) {
        for (int c = 0; c < ch; ++c) {
            const float v = c ? right[i] : left[i];
            if (bitsPerSample == 32) std::memcpy(&data[((size_t)i * ch + c) * 4], &v, 4);
            else { const int16_t q = zl_pcm16(v); std::memcpy(&data[((size_t)i * ch + c) * 2], &q, 2); }
        }
    }
    return libzl_wav_write_interleaved(path, data.data(), length, ch, sampleRate, bitsPerSample);
}

int libzl_wav_write_interleaved(const char *path, const void *frames, int length, int channels, double sampleRate, int bitsPerSample)
{
    if (!path || (!frames && length > 0) || length < 0 || channels < 1 || channels > 2 || !(bitsPerSample == 16 || bitsPerSample == 32)) return ZLHIP_ERR_INVALID;
    if ((uint64_t)length * (uint64_t)channels * (uint64_t)(bitsPerSample / 8) > 0xffffffffull - 36) return ZLHIP_ERR_CAPACITY;   // RIFF sizes are 32 bit
    FILE *f = std::fopen(path, "wb");
    if (!f) return ZLHIP_ERR_INVALID;
    unsigned char h[44];
    wav_header(h, (uint32_t)length, channels, sampleRate, bitsPerSample);
    const size_t bytes = (size_t)length * channels * (bitsPerSample / 8);
    const bool ok = std::fwrite(h, 1, 44, f) == 44 && (bytes == 0 || std::fwrite(frames, 1, bytes, f) == bytes);
    return (std::fclose(f) == 0 && ok) ? ZLHIP_OK : ZLHIP_ERR_INVALID;
}

// ---- engine lifecycle -----------------------------------------------------------------------------
void libzl_hotpath_configure(const zlhip_config *cfg)
{
    std::lock_guard<std::mutex> lk(G.mu);
    if (cfg) { G.cfg = *cfg; G.cfgSet = true; } else G.cfgSet = false;
}

int libzl_hotpath_status(void) { return G.engine ? ZLHIP_OK : G.status; }
void libzl_hotpath_set_clock_ms(int64_t (*clock_ms)(void)) { std::lock_guard<std::mutex> lk(G.mu); g_clock_ms = clock_ms; }
zlhip_engine *libzl_hotpath_engine(void) { return G.engine; }

void initJuce(void)                                                 // libzl.cpp:358-410
{
    std::lock_guard<std::mutex> lk(G.mu);
    if (G.engine) return;
    zlhip_config cfg;
    if (G.cfgSet) cfg = G.cfg;
    else { zlhip_config_default(&cfg); cfg.max_batch_blocks = 16; }   // 12 channels x 8 voices (SamplerSynth.cpp:254-278)
    G.status = zlhip_engine_create(&cfg, &G.engine);
    if (G.status != ZLHIP_OK) {
        G.engine = nullptr;
        std::fprintf(stderr, "libzl hot path: engine not available: %s\n", zlhip_strerror(G.status));   // reference logs and carries on
        return;
    }
    const size_t V = (size_t)cfg.num_buses * cfg.voices_per_bus;
    G.voicePositionId.assign(V, -1);
    G.voiceClip.assign(V, nullptr);
    G.reports.assign(V, zlhip_voice_report{});
    G.requests.clear(); G.seq.reset(); G.ext.clear(); G.due.clear(); G.lastNframes = 0;
    G.byEngineClip.clear();
    for (ClipAudioSource *c : G.clips) c->engineClip = -1;
}

void shutdownJuce(void)                                             // libzl.cpp:412-419
{
    std::lock_guard<std::mutex> lk(G.mu);
    for (ClipAudioSource *c : G.clips) delete c;
    G.clips.clear(); G.byEngineClip.clear();
    if (G.engine) { zlhip_engine_destroy(G.engine); G.engine = nullptr; }
    G.requests.clear(); G.seq.reset(); G.ext.clear(); G.due.clear();
    G.status = ZLHIP_ERR_STATE;
    G.nextClipId = 1;
}

// ---- ClipAudioSource bridge -------------------------------------------------------------------------
// Which calls the real-time thread never waits for: every setter, play / stop / queue, the SyncTimer_* calls below (they post a
// request or publish a snapshot), and the getters (plain fields or atomics).  Creating and destroying clips, init and shutdown
// take the cycle's mutex -- the reference marshals them to its message thread and takes SamplerSynth's clip mutex
// (libzl.cpp:118-128,258-267; SamplerSynth.cpp:287,299).
ClipAudioSource *ClipAudioSource_byID(int id)                      // libzl.cpp:107-116
{
    std::lock_guard<std::mutex> lk(G.mu);
    for (ClipAudioSource *c : G.clips) if (c->id == id) return c;
    return nullptr;
}

ClipAudioSource *ClipAudioSource_newFromBuffer(const float *left, const float *right, int length, double sampleRate, const char *name)
{
    if (!left || length < 1 || !(sampleRate > 0.0)) return nullptr;
    std::lock_guard<std::mutex> lk(G.mu);
    return make_clip(left, right, length, sampleRate, name ? name : "");
}

ClipAudioSource *ClipAudioSource_new(const char *filepath, bool muted)   // libzl.cpp:118-128, ClipAudioSource.cpp:135-205
{
    float *L = nullptr, *R = nullptr; int n = 0; double sr = 0.0;
    if (libzl_wav_read(filepath, &L, &R, &n, &sr) != ZLHIP_OK || n < 1) {
        std::fprintf(stderr, "libzl hot path: cannot open %s\n", filepath ? filepath : "(null)");
        return nullptr;
    }
    ClipAudioSource *c;
    {
        std::lock_guard<std::mutex> lk(G.mu);
        c = make_clip(L, R, n, sr, filepath);
    }
    libzl_wav_free(L); libzl_wav_free(R);
    if (c && muted) ClipAudioSource_setVolume(c, -100.0f);         // ClipAudioSource.cpp:178-181
    return c;
}

void ClipAudioSource_destroy(ClipAudioSource *c)                   // libzl.cpp:258-267
{
    if (!c) return;
    std::lock_guard<std::mutex> lk(G.mu);
    clip_stop(c, -3);                                              // ~ClipAudioSource: stop(), ClipAudioSource.cpp:209
    G.clips.erase(std::remove(G.clips.begin(), G.clips.end(), c), G.clips.end());
    for (auto &vc : G.voiceClip) if (vc == c) vc = nullptr;
    // the engine keeps the sound until the voices that still play its release tail are done; the slot is
    // released lazily by the cycle once no voice reports it
    c->id = -c->id - 1;
    c->progressCb.store(nullptr); c->levelCb.store(nullptr);
    G.clips.push_back(c);   // parked (negative id) until its voices ended; see the reap in render_and_report
}

int ClipAudioSource_id(ClipAudioSource *c) { return c->id; }
int ClipAudioSource_engineClip(ClipAudioSource *c) { return c->engineClip; }
void ClipAudioSource_setProgressCallback(ClipAudioSource *c, void (*functionPtr)(float)) { c->progressCb.store(functionPtr, std::memory_order_release); }
void ClipAudioSource_setAudioLevelChangedCallback(ClipAudioSource *c, void (*functionPtr)(float)) { c->levelCb.store(functionPtr, std::memory_order_release); }

void ClipAudioSource_play(ClipAudioSource *c, bool loop) { clip_play(c, loop, -2); }                                              // play(loop) default channel -2
void ClipAudioSource_stop(ClipAudioSource *c) { clip_stop(c, -3); }                                                               // stop() default -3: everywhere
void ClipAudioSource_playOnChannel(ClipAudioSource *c, bool loop, int midiChannel) { clip_play(c, loop, midiChannel); }
void ClipAudioSource_stopOnChannel(ClipAudioSource *c, int midiChannel) { clip_stop(c, midiChannel); }
void stopClips(int size, ClipAudioSource **clips) { for (int i = 0; i < size; ++i) ClipAudioSource_stop(clips[i]); }               // libzl.cpp:87-94

float ClipAudioSource_getDuration(ClipAudioSource *c) { return c->duration; }
const char *ClipAudioSource_getFileName(ClipAudioSource *c) { return c->fileName.c_str(); }

void ClipAudioSource_setStartPosition(ClipAudioSource *c, float s)  // ClipAudioSource.cpp:255-259
{
    std::lock_guard<std::mutex> sl(c->setMu);
    c->startPositionInSeconds = std::max(0.0f, s);
    publish_params(c);
}

void ClipAudioSource_setLength(ClipAudioSource *c, float beat, int bpm)   // ClipAudioSource.cpp:352-360
{
    std::lock_guard<std::mutex> sl(c->setMu);
    c->lengthInSeconds = subbeat_count_to_seconds((uint64_t)bpm, f32_to_u64_sat(beat * ZLHIP_BEAT_SUBDIVISIONS));
    c->lengthInBeats = beat;
    publish_params(c);
}

void ClipAudioSource_setPan(ClipAudioSource *c, float pan) { std::lock_guard<std::mutex> sl(c->setMu); if (c->pan != pan) { c->pan = pan; publish_params(c); } }          // :623-629
void ClipAudioSource_setSpeedRatio(ClipAudioSource *c, float v) { c->speedRatio = v; }                                             // :292-303 (offline re-render, out of scope)
void ClipAudioSource_setPitch(ClipAudioSource *c, float v) { c->pitchChange = v; }                                                 // :279-290
void ClipAudioSource_setGain(ClipAudioSource *c, float db) { c->gainDb = db; }                                                     // :305-311

void ClipAudioSource_setVolume(ClipAudioSource *c, float vol)      // ClipAudioSource.cpp:313-326
{
    std::lock_guard<std::mutex> sl(c->setMu);
    c->volumeAbsolute = (vol <= -40.0f) ? 0.0f : decibelsToVolumeFaderPosition(vol);
    publish_params(c);
}

void ClipAudioSource_setVolumeAbsolute(ClipAudioSource *c, float vol)   // ClipAudioSource.cpp:328-336
{
    std::lock_guard<std::mutex> sl(c->setMu);
    c->volumeAbsolute = std::max(0.0f, std::min(vol, 1.0f));
    publish_params(c);
}

float ClipAudioSource_volumeAbsolute(ClipAudioSource *c) { return c->volumeAbsolute; }
float dBFromVolume(float vol) { return volumeFaderPositionToDB(vol); }                                                              // libzl.cpp:429

// extension: the parameters of a clip as the engine receives them (the slice table of setSlices has no getter in libzl.h: the
// reference exposes it as a QVariantList property, ClipAudioSource.h).  Works without a device.
int libzl_hotpath_clip_params(ClipAudioSource *c, zlhip_clip_params *out)
{
    if (!c || !out) return -1;
    std::lock_guard<std::mutex> sl(c->setMu);
    fill_params(c, *out);
    return 0;
}
void ClipAudioSource_setSlices(ClipAudioSource *c, int slices) { std::lock_guard<std::mutex> sl(c->setMu); set_slices(c, slices); publish_params(c); }
int  ClipAudioSource_keyZoneStart(ClipAudioSource *c) { return c->keyZoneStart; }
void ClipAudioSource_setKeyZoneStart(ClipAudioSource *c, int v) { c->keyZoneStart = v; }
int  ClipAudioSource_keyZoneEnd(ClipAudioSource *c) { return c->keyZoneEnd; }
void ClipAudioSource_setKeyZoneEnd(ClipAudioSource *c, int v) { c->keyZoneEnd = v; }
int  ClipAudioSource_rootNote(ClipAudioSource *c) { return c->rootNote; }
void ClipAudioSource_setRootNote(ClipAudioSource *c, int v) { std::lock_guard<std::mutex> sl(c->setMu); if (c->rootNote != v) { c->rootNote = v; publish_params(c); } }

// quirk Q13: every ADSR setter starts from a fresh default Parameters (ClipAudioSource.cpp:636-685)
float ClipAudioSource_adsrAttack(ClipAudioSource *c) { return c->adsr.attack; }
void  ClipAudioSource_setADSRAttack(ClipAudioSource *c, float v) { std::lock_guard<std::mutex> sl(c->setMu); if (c->adsr.attack != v) { AdsrParams p; p.attack = v; c->adsr = p; publish_params(c); } }
float ClipAudioSource_adsrDecay(ClipAudioSource *c) { return c->adsr.decay; }
void  ClipAudioSource_setADSRDecay(ClipAudioSource *c, float v) { std::lock_guard<std::mutex> sl(c->setMu); if (c->adsr.decay != v) { AdsrParams p; p.decay = v; c->adsr = p; publish_params(c); } }
float ClipAudioSource_adsrSustain(ClipAudioSource *c) { return c->adsr.sustain; }
void  ClipAudioSource_setADSRSustain(ClipAudioSource *c, float v) { std::lock_guard<std::mutex> sl(c->setMu); if (c->adsr.sustain != v) { AdsrParams p; p.sustain = v; c->adsr = p; publish_params(c); } }
float ClipAudioSource_adsrRelease(ClipAudioSource *c) { return c->adsr.release; }
void  ClipAudioSource_setADSRRelease(ClipAudioSource *c, float v) { std::lock_guard<std::mutex> sl(c->setMu); if (c->adsr.release != v) { AdsrParams p; p.release = v; c->adsr = p; publish_params(c); } }

// positionsModel->peakGain() / firstProgress() as the last cycle left them (the cycle calls peakGain() itself, ClipAudioSource.cpp:92,
// so a read between two cycles returns exactly this value)
float ClipAudioSource_peakGain(ClipAudioSource *c) { return c->peakGainOut.load(std::memory_order_relaxed); }
double ClipAudioSource_firstProgress(ClipAudioSource *c) { return c->firstProgressOut.load(std::memory_order_relaxed); }

// ---- SyncTimer bridge: the part that schedules ClipCommands (libzl.h:69-79, libzl.cpp:311-349) -------------------------------
int  SyncTimer_getMultiplier(void) { return ZLHIP_BEAT_SUBDIVISIONS; }   // SyncTimer.cpp:946-948
void SyncTimer_startTimer(int interval) { Request r; std::memset(&r, 0, sizeof r); r.kind = Request::TimerStart; r.a = interval; post(r); }   // syncTimer->start(interval): the argument is the bpm (SyncTimer.cpp:870-872)
void SyncTimer_stopTimer(void) { Request r; std::memset(&r, 0, sizeof r); r.kind = Request::TimerStop; post(r); }
// SamplerSynth::setChannelEnabled (SamplerSynth.cpp:343-351; not in libzl.h: the reference reaches it through the SamplerSynth singleton)
void SamplerSynth_setChannelEnabled(int channel, bool enabled) { Request r; std::memset(&r, 0, sizeof r); r.kind = Request::ChannelEnabled; r.a = channel; r.b = enabled ? 1 : 0; post(r); }
void SyncTimer_setBpm(unsigned int bpm) { Request r; std::memset(&r, 0, sizeof r); r.kind = Request::SetBpm; r.a = (int32_t)bpm; post(r); }
void SyncTimer_queueClipToStartOnChannel(ClipAudioSource *clip, int midiChannel) { Request r; std::memset(&r, 0, sizeof r); r.kind = Request::QueueStart; r.a = clip->engineClip; r.b = midiChannel; post(r); }
void SyncTimer_queueClipToStopOnChannel(ClipAudioSource *clip, int midiChannel) { Request r; std::memset(&r, 0, sizeof r); r.kind = Request::QueueStop; r.a = clip->engineClip; r.b = midiChannel; post(r); }
void SyncTimer_queueClipToStart(ClipAudioSource *clip) { SyncTimer_queueClipToStartOnChannel(clip, -1); }   // SyncTimer.cpp:862-864
void SyncTimer_queueClipToStop(ClipAudioSource *clip) { SyncTimer_queueClipToStopOnChannel(clip, -1); }     // :866-868
// SyncTimer::addCallback / removeCallback (SyncTimer.cpp:790-812).  The reference calls callbacks[i](beat) from its timer thread for every
// tick of cumulativeBeat; here they fire on the cycle's thread after the cycle (libzl_hotpath_cycle), once per tick the timer went
// through.  (The reference's removeCallback leaves a null entry and drops the callback behind it; here the entry is simply removed.)
void SyncTimer_registerTimerCallback(void (*functionPtr)(int))
{
    std::lock_guard<std::mutex> lk(G.cbMu);
    for (auto &cb : G.timerCallbacks) if (!cb) { cb = functionPtr; return; }
}
void SyncTimer_deregisterTimerCallback(void (*functionPtr)(int))
{
    std::lock_guard<std::mutex> lk(G.cbMu);
    for (auto &cb : G.timerCallbacks) if (cb == functionPtr) cb = nullptr;
}
void libzl_hotpath_schedule_clip_command(const zlhip_clip_command *command, uint64_t delay) { if (command) schedule_command(*command, delay); }
void libzl_hotpath_timer_tick(void) { Request r; std::memset(&r, 0, sizeof r); r.kind = Request::TimerTick; post(r); }

int libzl_hotpath_transport(zlhip_clock *out)
{
    if (!out) return ZLHIP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(G.mu);
    std::memset(out, 0, sizeof *out);
    out->jack_playhead = G.seq.jackPlayheadGetter();
    out->jack_playhead_usecs = G.seq.jackPlayheadUsecsGetter();
    out->jack_subbeat_length_usecs = G.seq.jackSubbeatLengthInMicroseconds;
    return ZLHIP_OK;
}

// ---- the per-cycle seam -------------------------------------------------------------------------------
static void fire(std::vector<PendingCallback> &cbs)
{
    // outside the cycle's mutex: a callback may call back into this API (peakGain, byID, a setter)
    for (const PendingCallback &cb : cbs) cb.fn(cb.value);
    cbs.clear();
}

// host-owned transport: the host's SyncTimer getters arrive in `clock`
int libzl_hotpath_process(uint32_t nframes, const zlhip_clock *clock, float *out_left, float *out_right)
{
    return libzl_hotpath_process_fanout(nframes, clock, out_left, out_right, nullptr);
}

int libzl_hotpath_process_fanout(uint32_t nframes, const zlhip_clock *clock, float *out_left, float *out_right, float *fan_out)
{
    static thread_local std::vector<PendingCallback> cbs;
    int rc;
    {
        std::lock_guard<std::mutex> lk(G.mu);
        if (!G.engine) return G.status;
        if (!clock) return ZLHIP_ERR_INVALID;
        G.ext.lastPlayhead = clock->jack_playhead;                  // a delay counts from the playhead of the cycle that takes the request
        drain_requests(false);
        for (ClipAudioSource *c : G.clips) if (c->id >= 0) apply_params(c);
        G.ext.process(clock->jack_playhead, G.due);                 // currentTick = the host's jackPlayhead (SyncTimer.cpp:553-558)
        rc = dispatch_due();
        if (rc == ZLHIP_OK) rc = render_and_report(nframes, clock, out_left, out_right, cbs, fan_out);
    }
    fire(cbs);
    return rc;
}

// the library's own transport: SyncTimerPrivate::process for this JACK cycle, then every SamplerChannel (the order in which the two
// JACK clients run inside a cycle is not defined in the reference; here the commands of a cycle reach the channels in the same cycle)
int libzl_hotpath_cycle(uint32_t nframes, uint64_t current_usecs, uint64_t next_usecs, float period_usecs, float *out_left, float *out_right)
{
    return libzl_hotpath_cycle_fanout(nframes, current_usecs, next_usecs, period_usecs, out_left, out_right, nullptr);
}

int libzl_hotpath_cycle_fanout(uint32_t nframes, uint64_t current_usecs, uint64_t next_usecs, float period_usecs, float *out_left, float *out_right, float *fan_out)
{
    static thread_local std::vector<PendingCallback> cbs;
    static thread_local std::vector<int> beats;
    int rc;
    {
        std::lock_guard<std::mutex> lk(G.mu);
        if (!G.engine) return G.status;
        if (nframes == 0 || next_usecs < current_usecs) return ZLHIP_ERR_INVALID;
        if (nframes != G.lastNframes) { G.seq.set_jack_latency(nframes, G.cfgSet ? G.cfg.playback_sample_rate : 48000.0); G.lastNframes = nframes; }
        drain_requests(true);
        for (ClipAudioSource *c : G.clips) if (c->id >= 0) apply_params(c);
        G.seq.process(nframes, current_usecs, next_usecs, period_usecs, G.due);
        rc = dispatch_due();
        if (rc == ZLHIP_OK) {
            zlhip_clock clk;
            clk.current_usecs = current_usecs; clk.next_usecs = next_usecs;
            clk.jack_playhead = G.seq.jackPlayheadGetter();                         // SamplerSynthVoice.cpp:179-182,232-237 read these getters
            clk.jack_playhead_usecs = G.seq.jackPlayheadUsecsGetter();
            clk.jack_subbeat_length_usecs = G.seq.jackSubbeatLengthInMicroseconds;
            rc = render_and_report(nframes, &clk, out_left, out_right, cbs, fan_out);
        }
        // the timer thread ticks once per subbeat while it runs (SyncTimer.cpp:117-163): modelled as one tick between two cycles
        G.seq.beatSink = &G.beats;
        if (!G.seq.threadPaused) G.seq.hi_res_timer_callback();
        beats.swap(G.beats);
    }
    fire(cbs);
    if (!beats.empty()) {
        void (*tcb[16])(int);
        { std::lock_guard<std::mutex> cl(G.cbMu); std::memcpy(tcb, G.timerCallbacks, sizeof tcb); }
        for (int b : beats) for (auto cb : tcb) if (cb) cb(b);
        beats.clear();
    }
    return rc;
}

// ---- offline bounce of the running session to WAV files (BASELINE configs[4] from / to real files) ------------------
// What nblocks calls of libzl_hotpath_cycle would render -- the library's own transport, JACK time advancing by the nominal period
// per cycle, commands dispatched in the cycle their step falls due in -- but rendered in batches (zlhip_bounce: the render kernel
// writes the recorder's 16-bit format, or floats, straight into page-locked host memory) and appended to one stereo WAV per sampler
// channel.  A batch ends where a cycle has commands to dispatch (they apply at that cycle's boundary) or at the engine's
// max_batch_blocks.  The positions models and callbacks are not driven (no real time passes).
int libzl_hotpath_bounce_to_wav(const char *prefix, int64_t nblocks, uint32_t nframes, uint64_t start_usecs, int bits_per_sample)
{
    if (!prefix || nblocks < 1 || nframes == 0 || !(bits_per_sample == 16 || bits_per_sample == 32)) return ZLHIP_ERR_INVALID;
    std::lock_guard<std::mutex> lk(G.mu);
    if (!G.engine) return G.status;
    zlhip_config cfg;
    if (G.cfgSet) cfg = G.cfg; else { zlhip_config_default(&cfg); cfg.max_batch_blocks = 16; }
    const int B = cfg.num_buses;
    const double fs = cfg.playback_sample_rate;
    const uint64_t period = (uint64_t)std::llround(1e6 * (double)nframes / fs);
    const uint64_t totalFrames = (uint64_t)nblocks * nframes;
    const int frameBytes = bits_per_sample == 16 ? 4 : 8;                       // stereo
    if (totalFrames * (uint64_t)frameBytes > 0xffffffffull - 36) return ZLHIP_ERR_CAPACITY;    // RIFF sizes are 32 bit
    const int64_t cap = std::max(1, cfg.max_batch_blocks);
    void *host = nullptr;
    // fp32 arrives planar per bus ([B][2][frames]) and is interleaved on the way to the file; 16 bit arrives as the file wants it
    if (zlhip_host_alloc((size_t)B * (size_t)cap * nframes * (size_t)frameBytes, &host) != ZLHIP_OK) return ZLHIP_ERR_CAPACITY;
    std::vector<FILE *> files((size_t)B, nullptr);
    int rc = ZLHIP_OK;
    for (int b = 0; b < B && rc == ZLHIP_OK; ++b) {
        const std::string path = std::string(prefix) + "-channel_" + std::to_string(b) + ".wav";
        files[(size_t)b] = std::fopen(path.c_str(), "wb");
        unsigned char h[44];
        wav_header(h, (uint32_t)totalFrames, 2, fs, bits_per_sample);
        if (!files[(size_t)b] || std::fwrite(h, 1, 44, files[(size_t)b]) != 44) rc = ZLHIP_ERR_INVALID;
    }
    std::vector<zlhip_clock> clocks;
    std::vector<float> inter;
    auto flush = [&]() -> int {
        const size_t n = clocks.size();
        if (n == 0) return ZLHIP_OK;
        const size_t frames = n * nframes;
        int r = zlhip_bounce(G.engine, (int64_t)n, (int32_t)nframes, clocks.data(), host, bits_per_sample == 16 ? ZLHIP_BOUNCE_PCM16_STEREO : ZLHIP_BOUNCE_F32_PLANAR, 0);
        if (r != ZLHIP_OK) return r;
        for (int b = 0; b < B; ++b) {
            if (bits_per_sample == 16) {
                if (std::fwrite((const char *)host + (size_t)b * frames * 4, 1, frames * 4, files[(size_t)b]) != frames * 4) return ZLHIP_ERR_INVALID;
            } else {
                const float *L = (const float *)host + (size_t)b * 2 * frames, *R = L + frames;
                inter.resize(frames * 2);
                for (size_t i = 0; i < frames; ++i) { inter[2 * i] = L[i]; inter[2 * i + 1] = R[i]; }
                if (std::fwrite(inter.data(), 1, frames * 8, files[(size_t)b]) != frames * 8) return ZLHIP_ERR_INVALID;
            }
        }
        clocks.clear();
        return ZLHIP_OK;
    };
    if (nframes != G.lastNframes) { G.seq.set_jack_latency(nframes, fs); G.lastNframes = nframes; }
    drain_requests(true);
    for (ClipAudioSource *c : G.clips) if (c->id >= 0) apply_params(c);
    for (int64_t k = 0; k < nblocks && rc == ZLHIP_OK; ++k) {
        const uint64_t cu = start_usecs + (uint64_t)k * period, nx = cu + period;
        G.seq.process(nframes, cu, nx, (float)period, G.due);
        if (!G.due.empty() || (int64_t)clocks.size() == cap) {
            rc = flush();                                                            // the blocks before this cycle's commands
            if (rc == ZLHIP_OK) rc = dispatch_due(false);
        }
        zlhip_clock clk;
        clk.current_usecs = cu; clk.next_usecs = nx;
        clk.jack_playhead = G.seq.jackPlayheadGetter(); clk.jack_playhead_usecs = G.seq.jackPlayheadUsecsGetter();
        clk.jack_subbeat_length_usecs = G.seq.jackSubbeatLengthInMicroseconds;
        clocks.push_back(clk);
        if (!G.seq.threadPaused) G.seq.hi_res_timer_callback();
    }
    if (rc == ZLHIP_OK) rc = flush();
    for (FILE *f : files) if (f && std::fclose(f) != 0 && rc == ZLHIP_OK) rc = ZLHIP_ERR_INVALID;
    zlhip_host_free(host);
    return rc;
}

// ---- JackPassthrough bridge (libzl.cpp:476-575) ------------------------------------------------------------
void  JackPassthrough_setPanAmount(int channel, float amount) { if (PassState *p = pass_for(channel)) p->pan.store(amount, std::memory_order_relaxed); }
float JackPassthrough_getPanAmount(int channel) { PassState *p = pass_for(channel); return p ? p->pan.load(std::memory_order_relaxed) : 0.0f; }
float JackPassthrough_getWetFx1Amount(int channel) { PassState *p = pass_for(channel); return p ? p->fx1.load(std::memory_order_relaxed) : 0.0f; }
void  JackPassthrough_setWetFx1Amount(int channel, float amount) { if (PassState *p = pass_for(channel)) p->fx1.store(amount, std::memory_order_relaxed); }
float JackPassthrough_getWetFx2Amount(int channel) { PassState *p = pass_for(channel); return p ? p->fx2.load(std::memory_order_relaxed) : 0.0f; }
void  JackPassthrough_setWetFx2Amount(int channel, float amount) { if (PassState *p = pass_for(channel)) p->fx2.store(amount, std::memory_order_relaxed); }
float JackPassthrough_getDryAmount(int channel) { PassState *p = pass_for(channel); return p ? p->dry.load(std::memory_order_relaxed) : 0.0f; }
void  JackPassthrough_setDryAmount(int channel, float amount) { if (PassState *p = pass_for(channel)) p->dry.store(amount, std::memory_order_relaxed); }
float JackPassthrough_getMuted(int channel) { PassState *p = pass_for(channel); return p ? (p->muted.load(std::memory_order_relaxed) ? 1.0f : 0.0f) : 0.0f; }
void  JackPassthrough_setMuted(int channel, bool muted) { if (PassState *p = pass_for(channel)) p->muted.store(muted, std::memory_order_relaxed); }
int   JackPassthrough_getParams(int channel, zlhip_passthrough_params *out)
{
    PassState *p = pass_for(channel);
    if (!p || !out) return ZLHIP_ERR_INVALID;
    out->dry_amount = p->dry.load(std::memory_order_relaxed); out->wet_fx1_amount = p->fx1.load(std::memory_order_relaxed);
    out->wet_fx2_amount = p->fx2.load(std::memory_order_relaxed); out->pan_amount = p->pan.load(std::memory_order_relaxed);
    out->muted = p->muted.load(std::memory_order_relaxed) ? 1 : 0;
    return ZLHIP_OK;
}

uint64_t libzl_hotpath_dropped_requests(void) { return G.dropped.load(std::memory_order_relaxed); }

}  // extern "C"
#endif  // ZLHIP_NO_LIBZL_NAMES
