// zl_host.h -- control-plane half of the reference's SamplerChannel / SamplerSynthVoice, HIP-free.
//
//   SamplerSynth::handleClipCommand   SamplerSynth.cpp:328-341  (bus routing by midi channel)
//   SamplerChannel::handleCommand     SamplerSynth.cpp:187-230  (stop-all-equivalent, first-free start, merge)
//   SamplerSynthVoice::setCurrentCommand  SamplerSynthVoice.cpp:58-98
//   SamplerSynthVoice::startNote      SamplerSynthVoice.cpp:110-144 (one-off fp64 math, ADSR noteOn)
//
// The host keeps only what it needs to allocate voices (command, sound, isPlaying); the evolving
// state (position, envelope, loop clock) lives in HBM.  Commands become ZlVoiceOp records that the
// device applies, in arrival order per voice, before the next block is planned.
// Used by zl_engine.cpp (product) and by tests/cpu_harness (CPU unit tests of the same logic).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/zlhip.h"
#include "zl_plan.h"
#include "zl_types.h"

// std::pow(2.0, x) exactly as the reference calls it (SamplerSynthVoice.cpp:115).  Called through a volatile
// pointer because clang rewrites pow(2.0, x) into exp2(x), which differs from glibc's pow in the last bit for some
// x -- enough to move sourceSamplePosition by an ulp after a few hundred frames.
static double (*volatile zl_libm_pow)(double, double) = static_cast<double (*)(double, double)>(std::pow);

struct ZlHostVoice {                     // control-plane view of one SamplerSynthVoice
    bool hasCommand = false;             // d->clipCommand != nullptr
    zlhip_clip_command cmd{};            // *d->clipCommand
    int  sound = -1;                     // getCurrentlyPlayingSound()
    bool isPlaying = false;              // SamplerSynthVoice.h:31
    uint64_t startTick = 0;
    bool cheapPlan = false;              // started at exactly the playback rate on a sample-space loop (or as a one-shot): K1 plans it in a handful of
                                         // exact runs whatever the window's length (a hint for the window size, zl_engine.cpp; never for results)
};

struct ZlHostControl {
    int num_buses = 0, voices_per_bus = 0;
    double playback_sample_rate = 48000.0;
    std::vector<ZlSound> sounds;
    std::vector<zlhip_clip_params> clipParams;
    std::vector<char> soundUsed;
    std::vector<ZlHostVoice> voices;
    std::vector<ZlVoiceOp> pendingOps;
    std::vector<uint32_t> opOrder;
    // clip-parameter edits since the last render call, one per clip (a later edit of the same clip replaces the earlier one):
    // the device applies them at the start of the next call / cycle, where the reference's voices read the parameters
    // (SamplerSynthVoice.cpp:189-196).  The host mirror clipParams is updated at once: commands handled after an edit see it,
    // as startNote does (SamplerSynthVoice.cpp:115-121).
    std::vector<ZlClipEdit> pendingClipEdits;

    void init(int B, int VPB, int max_sounds, double fs)
    {
        num_buses = B; voices_per_bus = VPB; playback_sample_rate = fs;
        sounds.assign((size_t)max_sounds, ZlSound{0, 0, 0, 0.0});
        clipParams.assign((size_t)max_sounds, zlhip_clip_params{});
        soundUsed.assign((size_t)max_sounds, 0);
        voices.assign((size_t)B * VPB, ZlHostVoice());
        pendingOps.clear();
        pendingClipEdits.clear();
    }

    void set_clip_params(int id, const zlhip_clip_params &p)
    {
        // the slice table travels only when it changed (a kilobyte; the knobs in front of it are 32 bytes)
        const zlhip_clip_params &old = clipParams[(size_t)id];
        const bool slices = old.num_slice_positions != p.num_slice_positions
            || std::memcmp(old.slice_positions, p.slice_positions, sizeof(double) * (size_t)std::max(p.num_slice_positions, 0)) != 0;
        clipParams[(size_t)id] = p;
        ZlClipEdit *slot = nullptr;
        for (ZlClipEdit &e : pendingClipEdits) if (e.clip == id) { slot = &e; break; }
        if (!slot) { pendingClipEdits.emplace_back(); slot = &pendingClipEdits.back(); slot->full = 0; }
        slot->clip = id;
        if (slices) slot->full = 1;
        fill_clip(slot->c, p);
    }
    // a sound slot that is (re)used: whatever the table held for it is unknown -- the next edit carries everything
    void forget_clip_params(int id)
    {
        clipParams[(size_t)id] = zlhip_clip_params{};
        clipParams[(size_t)id].num_slice_positions = -1;
    }

    static float adsr_rate(float distance, float timeInSeconds, double sr)      // juce::ADSR::recalculateRates
    {
        return timeInSeconds > 0.0f ? (float)(distance / (timeInSeconds * sr)) : -1.0f;
    }

    static float clip_start_position(const zlhip_clip_params &c, int slice)     // ClipAudioSource.cpp:261-268
    {
        if (slice > -1 && slice < c.num_slice_positions)
            return (float)(c.start_position_seconds + (c.length_seconds * c.slice_positions[slice]));
        return c.start_position_seconds;
    }

    static bool commands_equivalent(const zlhip_clip_command &a, const zlhip_clip_command &b)   // ClipCommand.h:33-39
    {
        return a.clip == b.clip
            && ((a.change_slice && b.change_slice && a.slice == b.slice)
                || (!a.change_slice && !b.change_slice && a.midi_note == b.midi_note && a.midi_channel == b.midi_channel));
    }

    // voices that ended on the device (stopNote(.., false) inside process) free their slot
    void absorb_reports(const ZlReport *reports)
    {
        for (size_t v = 0; v < voices.size(); ++v)
            if (voices[v].isPlaying && !reports[v].playing) voices[v] = ZlHostVoice();
    }

    int lastStartedVoice = -1;                 // the voice the last handled command started (-1: it started none)
    std::vector<uint8_t> busDisabled;          // SamplerChannel::enabled == false (SamplerSynth.cpp:60,123,343-351); empty = every bus enabled
    bool bus_enabled(int bus) const { return busDisabled.empty() || !busDisabled[(size_t)bus]; }
    // SamplerSynth::setChannelEnabled: the channel's voices are no longer processed (they keep their state and position and go on
    // when the channel is enabled again); its commands are still handled (:118-122 in front of the test at :123)
    int set_bus_enabled(int bus, bool enabled)
    {
        if (bus < 0 || bus >= num_buses) return 0;
        if (busDisabled.empty()) busDisabled.assign((size_t)num_buses, 0);
        if ((busDisabled[(size_t)bus] == 0) == enabled) return 1;
        busDisabled[(size_t)bus] = enabled ? 0 : 1;
        for (int i = 0; i < voices_per_bus; ++i) {
            const int v = bus * voices_per_bus + i;
            if (!voices[(size_t)v].isPlaying) continue;
            ZlVoiceOp op; std::memset(&op, 0, sizeof op);
            op.voice = v; op.kind = enabled ? ZL_OP_THAW : ZL_OP_FREEZE;
            pendingOps.push_back(op);
        }
        return 1;
    }

    void push_start(int v, const zlhip_clip_command &cmd, uint64_t tick)
    {
        // setCurrentCommand on an idle voice takes the command (SamplerSynthVoice.cpp:94-96), setStartTick,
        // juce::Synthesiser::startVoice -> startNote (SamplerSynthVoice.cpp:110-144)
        ZlHostVoice &hv = voices[(size_t)v];
        lastStartedVoice = v;
        hv.cmd = cmd; hv.hasCommand = true; hv.isPlaying = true; hv.startTick = tick; hv.sound = cmd.clip;
        const ZlSound &sd = sounds[(size_t)cmd.clip];
        const zlhip_clip_params &cp = clipParams[(size_t)cmd.clip];
        const double sr = sd.sample_rate;
        ZlVoiceOp op; std::memset(&op, 0, sizeof op);
        op.voice = v; op.kind = ZL_OP_START;
        ZlVoiceState &s = op.start;
        s.pitch_ratio = zl_libm_pow(2.0, (cmd.midi_note - cp.root_note) / 12.0) * sr / playback_sample_rate;   // :115-116
        hv.cheapPlan = s.pitch_ratio == 1.0 && !(cmd.looping && truncf(cp.length_in_beats) == cp.length_in_beats);
        s.src_len = cp.duration_seconds * sr;                                                               // :120
        s.P = (int)(clip_start_position(cp, cmd.slice) * sr);                                                // :121
        s.next_loop_tick = zl_f32_to_u64_sat(tick + cp.length_in_beats * ZLHIP_BEAT_SUBDIVISIONS);           // :123 (u64 + float -> float)
        s.next_loop_usecs = 0;                                                                               // :124
        s.lgain = cmd.volume; s.rgain = cmd.volume;     // :131-132; velocity = clipCommand->volume (SamplerSynth.cpp:210)
        // adsr.reset(); setSampleRate(sr); setParameters(clip adsr); noteOn()                              // :134-137
        s.adsr_sr = sr;
        s.sustain = cp.adsr_sustain; s.release = cp.adsr_release;
        s.attack_rate  = adsr_rate(1.0f, cp.adsr_attack, sr);
        s.decay_rate   = adsr_rate(1.0f - cp.adsr_sustain, cp.adsr_decay, sr);
        s.release_rate = adsr_rate(cp.adsr_sustain, cp.adsr_release, sr);
        s.env = 0.0f;
        if (s.attack_rate > 0.0f)     { s.adsr_state = ZL_ADSR_ATTACK; }
        else if (s.decay_rate > 0.0f) { s.env = 1.0f; s.adsr_state = ZL_ADSR_DECAY; }
        else                          { s.env = cp.adsr_sustain; s.adsr_state = ZL_ADSR_SUSTAIN; }
        s.clip = cmd.clip; s.slice = cmd.slice; s.looping = cmd.looping ? 1 : 0;
        s.playing = bus_enabled(v / voices_per_bus) ? 1 : 2;          // a voice started on a disabled channel waits for the channel
        pendingOps.push_back(op);
    }

    void push_merge(int v, const zlhip_clip_command &c)
    {
        // setCurrentCommand on a playing voice (SamplerSynthVoice.cpp:59-93)
        ZlHostVoice &hv = voices[(size_t)v];
        ZlVoiceOp op; std::memset(&op, 0, sizeof op);
        op.voice = v; op.kind = ZL_OP_PATCH;
        if (c.change_looping) { hv.cmd.looping = c.looping; hv.cmd.change_looping = 1; op.patch_mask |= ZL_PATCH_LOOPING; op.looping = c.looping ? 1 : 0; }
        if (c.change_pitch)   { hv.cmd.pitch_change = c.pitch_change; hv.cmd.change_pitch = 1; }
        if (c.change_speed)   { hv.cmd.speed_ratio = c.speed_ratio; hv.cmd.change_speed = 1; }
        if (c.change_gain_db) { hv.cmd.gain_db = c.gain_db; hv.cmd.change_gain_db = 1; }
        if (c.change_volume)  { hv.cmd.volume = c.volume; hv.cmd.change_volume = 1; op.patch_mask |= ZL_PATCH_GAIN; op.gain = c.volume; }
        if (c.change_slice)   { hv.cmd.slice = c.slice; op.patch_mask |= ZL_PATCH_SLICE; op.slice = c.slice; }
        if (c.start_playback && hv.sound >= 0) {
            op.patch_mask |= ZL_PATCH_POSITION;
            op.position = (int)(clip_start_position(clipParams[(size_t)hv.sound], hv.cmd.slice) * sounds[(size_t)hv.sound].sample_rate);   // :90
        }
        if (op.patch_mask) pendingOps.push_back(op);
    }

    // SamplerChannel::handleCommand for one bus.  forcedSlot >= 0 addresses a voice slot directly
    // (build extension for deterministic large scenes) and skips the midi-channel match.
    int handle_on_bus(int bus, const zlhip_clip_command &c, uint64_t tick, int forcedSlot)
    {
        const int VPB = voices_per_bus, base = bus * VPB;
        const int busMidi = bus - 2;                                   // SamplerSynth.cpp:270
        const int sound = c.clip;
        if (sound < 0 || sound >= (int)sounds.size() || !soundUsed[(size_t)sound]) return 0;   // :330 clipSounds.contains
        const bool mine = forcedSlot >= 0 || busMidi == c.midi_channel;
        int consumed = 0;
        if (c.stop_playback || c.start_playback) {
            if (c.stop_playback && mine) {                             // :191-203
                for (int i = 0; i < VPB; ++i) {
                    ZlHostVoice &hv = voices[(size_t)(base + i)];
                    if (hv.sound == sound && hv.hasCommand && commands_equivalent(hv.cmd, c)) {
                        ZlVoiceOp op; std::memset(&op, 0, sizeof op);
                        op.voice = base + i; op.kind = ZL_OP_NOTE_OFF;
                        pendingOps.push_back(op);
                    }
                }
            }
            if (c.start_playback && mine) {                            // :204-215
                for (int i = 0; i < VPB; ++i) {
                    if (forcedSlot >= 0 && i != forcedSlot) continue;
                    if (!voices[(size_t)(base + i)].isPlaying) { push_start(base + i, c, tick); consumed = 1; break; }
                }
            }
        } else if (mine) {                                             // :216-229
            for (int i = 0; i < VPB; ++i) {
                ZlHostVoice &hv = voices[(size_t)(base + i)];
                if (hv.sound == sound && hv.hasCommand && commands_equivalent(hv.cmd, c)) { push_merge(base + i, c); consumed = 1; }
            }
        }
        return consumed;
    }

    // SamplerSynthVoice::stopNote on one voice (SamplerSynthVoice.cpp:146-169): with tail-off the envelope is released
    // and the voice frees itself later; without, it stops before the next block
    int stop_voice(int bus, int slot, bool allowTailOff)
    {
        const int v = bus * voices_per_bus + slot;
        ZlHostVoice &hv = voices[(size_t)v];
        if (!hv.isPlaying) return 0;
        ZlVoiceOp op; std::memset(&op, 0, sizeof op);
        op.voice = v; op.kind = allowTailOff ? ZL_OP_NOTE_OFF : ZL_OP_HARD_STOP;
        pendingOps.push_back(op);
        if (!allowTailOff) hv = ZlHostVoice();
        return 1;
    }

    // setCurrentCommand on one playing voice (SamplerSynthVoice.cpp:58-93)
    int update_voice(int bus, int slot, const zlhip_clip_command &c)
    {
        const int v = bus * voices_per_bus + slot;
        if (!voices[(size_t)v].isPlaying) return 0;
        push_merge(v, c);
        return 1;
    }

    int handle_command(const zlhip_clip_command &c, uint64_t tick)
    {
        const int bus = c.midi_channel + 2;                            // SamplerSynth.cpp:330-331
        lastStartedVoice = -1;
        if (bus < 0 || bus >= num_buses) return 0;
        return handle_on_bus(bus, c, tick, -1);
    }

    // pending ops -> voice order (stable: arrival order per voice) written to out[0 .. pendingOps.size()) + per-voice
    // ranges.  Sorts indices, so every op is moved once -- straight into the caller's (pinned) buffer.
    void drain_ops_to(ZlVoiceOp *out, std::vector<ZlOpRange> &ranges)
    {
        const size_t n = pendingOps.size();
        opOrder.resize(n);
        for (size_t i = 0; i < n; ++i) opOrder[i] = (uint32_t)i;
        std::stable_sort(opOrder.begin(), opOrder.end(), [this](uint32_t a, uint32_t b) { return pendingOps[a].voice < pendingOps[b].voice; });
        ranges.clear();
        for (size_t i = 0; i < n; ++i) {
            out[i] = pendingOps[opOrder[i]];
            if (!ranges.empty() && ranges.back().voice == out[i].voice) ranges.back().count += 1;
            else ranges.push_back(ZlOpRange{ out[i].voice, (int32_t)i, 1, 0 });
        }
        pendingOps.clear();
    }
    void drain_ops(std::vector<ZlVoiceOp> &sorted, std::vector<ZlOpRange> &ranges)
    {
        sorted.resize(pendingOps.size());
        drain_ops_to(sorted.data(), ranges);
    }

    static void fill_clock(ZlClock &c, const zlhip_clock &k, int nframes)
    {
        c.current_usecs = k.current_usecs; c.next_usecs = k.next_usecs;
        c.playhead = k.jack_playhead; c.playhead_usecs = k.jack_playhead_usecs; c.subbeat_usecs = k.jack_subbeat_length_usecs;
        c.usecs_per_frame = (k.next_usecs - k.current_usecs) / (uint64_t)nframes;      // SamplerSynthVoice.cpp:183
    }

    static void fill_clip(ZlClip &c, const zlhip_clip_params &p)
    {
        std::memset(&c, 0, sizeof c);
        c.start_sec = p.start_position_seconds; c.length_sec = p.length_seconds; c.length_beats = p.length_in_beats;
        c.volume_abs = p.volume_absolute; c.pan = p.pan; c.duration = p.duration_seconds;
        c.n_slice_pos = p.num_slice_positions;
        for (int i = 0; i < p.num_slice_positions; ++i) c.slice_pos[i] = p.slice_positions[i];
    }

    static void default_clip_params(zlhip_clip_params *p, float duration_seconds)
    {
        std::memset(p, 0, sizeof *p);
        p->start_position_seconds = 0;            // ClipAudioSource.cpp:63
        p->length_seconds = duration_seconds;     // :158
        p->length_in_beats = -1;                  // :65
        p->volume_absolute = 1.0f;
        p->pan = 0.0f;                            // :69
        p->duration_seconds = duration_seconds;   // :367
        p->adsr_attack = 0.0f; p->adsr_decay = 0.1f; p->adsr_sustain = 1.0f; p->adsr_release = 0.05f;   // :164-168
        p->root_note = 60;                        // :81
        p->num_slice_positions = 16;              // setSlices(16), :204,:495-528
        double pos = 0.0;                         // :510-522 : positions accumulate 1/16 in double
        const double inc = (1.0f - 0.0) / 16;
        for (int i = 0; i < 16; ++i) { p->slice_positions[i] = pos; pos += inc; }
    }
};
