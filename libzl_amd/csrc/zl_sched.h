// zl_sched.h -- the ClipCommand scheduling front-end of the sampler, HIP-free host code.
//
// Restates the part of the reference's SyncTimer that carries ClipCommands to SamplerSynth (paths under /root/reference/lib):
//   SyncTimer::scheduleClipCommand        SyncTimer.cpp:1011-1048  same-step merge of equivalent commands
//   SyncTimerPrivate::delayedStep         SyncTimer.cpp:364-378    which step a delay addresses (paused / running)
//   StepData                              SyncTimer.cpp:43-79      the step ring (32768 steps), played / ensureFresh
//   SyncTimerPrivate::process             SyncTimer.cpp:452-702    steps due in a JACK cycle, dispatch with jackPlayhead (:553-558),
//                                                                  SetBpmOperation (:606-612), playhead / usecs roll (:634-672)
//   SyncTimerPrivate::hiResTimerCallback  SyncTimer.cpp:391-418    cumulativeBeat runs ahead of the playhead
//   SyncTimer::start / stop               SyncTimer.cpp:870-925    incl. the re-scheduling of unplayed commands at volume 0
//   SyncTimer::setBpm                     SyncTimer.cpp:954-975
//   SyncTimer::queueClipToStart/StopOnChannel   SyncTimer.cpp:815-860
//   SyncTimer::jackPlayhead / jackPlayheadUsecs / jackSubbeatLengthInMicroseconds   SyncTimer.cpp:990-1009
// Out of scope: MIDI buffers, the jack transport position, timer commands other than SetBpm, the timer THREAD itself
// (its tick is an explicit call here: hi_res_timer_callback), Qt signals.
//
// Everything is plain integer / double arithmetic in the reference's own order (u64 += double, integer nanoseconds).  Used by
// zl_libzl.cpp (product); the checker restates it independently (oracle/zl_oracle.c, zlo_sync_timer_*, and its numpy twin).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "../../include/zlhip.h"

// ClipCommand::equivalentTo (ClipCommand.h:33-39)
static inline bool zl_commands_equivalent(const zlhip_clip_command &a, const zlhip_clip_command &b)
{
    return a.clip == b.clip
        && ((a.change_slice && b.change_slice && a.slice == b.slice)
            || (!a.change_slice && !b.change_slice && a.midi_note == b.midi_note && a.midi_channel == b.midi_channel));
}

// The body of SyncTimer::scheduleClipCommand once the step is known (SyncTimer.cpp:1014-1047): a command equivalent to one
// already queued for the step is folded into it -- looping / pitch / speed / gain / volume when the new command changes them,
// startPlayback when it starts; stopPlayback, slice and the `looping` of a command without changeLooping are NOT copied -- and
// dropped; otherwise it is appended.  Returns true when it was appended.
static inline bool zl_schedule_into_step(std::vector<zlhip_clip_command> &step, const zlhip_clip_command &command)
{
    bool foundExisting = false;
    for (zlhip_clip_command &existing : step) {
        if (zl_commands_equivalent(existing, command)) {
            if (command.change_looping) { existing.looping = command.looping; existing.change_looping = 1; }
            if (command.change_pitch)   { existing.pitch_change = command.pitch_change; existing.change_pitch = 1; }
            if (command.change_speed)   { existing.speed_ratio = command.speed_ratio; existing.change_speed = 1; }
            if (command.change_gain_db) { existing.gain_db = command.gain_db; existing.change_gain_db = 1; }
            if (command.change_volume)  { existing.volume = command.volume; existing.change_volume = 1; }
            if (command.start_playback) existing.start_playback = 1;
            foundExisting = true;
        }
    }
    if (!foundExisting) step.push_back(command);
    return !foundExisting;
}

struct ZlDispatch { zlhip_clip_command cmd; uint64_t tick; };

struct ZlStepSequencer {
    static constexpr uint64_t StepRingCount = 32768;         // SyncTimer.cpp:253
    static constexpr uint64_t BeatSubdivisions = 96;         // :95
    struct Step {
        std::vector<zlhip_clip_command> clipCommands;
        std::vector<int> bpmCommands;                         // TimerCommand::SetBpmOperation parameters
        bool played = true;                                   // :78
        bool listed = false;                                  // (an entry of freshSteps points here)
    };
    std::vector<Step> ring;
    // the steps that have been made fresh (unplayed) since they were last played: the reference walks all 32768 steps in stop() and
    // queueClipToStopOnChannel() on the caller's thread; here those calls run on the cycle's thread, which only visits these
    std::vector<uint32_t> freshSteps;
    void ensureFresh(uint64_t index)                          // StepData::ensureFresh, :50-62
    {
        Step &sd = ring[index];
        if (sd.played) {
            sd.played = false; sd.clipCommands.clear(); sd.bpmCommands.clear();
            if (!sd.listed) { sd.listed = true; freshSteps.push_back((uint32_t)index); }
        }
    }
    void pruneFresh()
    {
        size_t n = 0;
        for (uint32_t i : freshSteps) { if (!ring[i].played) freshSteps[n++] = i; else ring[i].listed = false; }
        freshSteps.resize(n);
    }
    uint64_t stepReadHead = 0;                                // index of *stepReadHead
    uint64_t stepNextPlaybackPosition = 0;
    // SyncTimerThread
    uint64_t bpm = 120;                                       // :250
    bool threadPaused = true;                                 // :236
    // SyncTimerPrivate
    bool isPaused = true;                                     // :440 (follows the thread through pausedChanged, :750-752)
    uint64_t jackPlayhead = 0;
    double   jackPlayheadBpm = 120;
    uint64_t jackNextPlaybackPosition = 0;
    uint64_t jackSubbeatLengthInMicroseconds = 0;
    uint64_t jackLatency = 0;                                 // ms
    uint64_t scheduleAheadAmount = 0;
    uint64_t cumulativeBeat = 0;
    int      beat = 0;
    uint64_t stepReadHeadOnStart = 0;
    uint64_t jackPlayheadReturn = 0, jackSubbeatLengthInMicrosecondsReturn = 0;

    ZlStepSequencer() { reset(); }

    void reset()
    {
        ring.assign(StepRingCount, Step());
        freshSteps.clear();
        stepReadHead = 0; stepNextPlaybackPosition = 0;
        bpm = 120; threadPaused = true; isPaused = true;
        jackPlayhead = 0; jackPlayheadBpm = 120; jackNextPlaybackPosition = 0;
        jackSubbeatLengthInMicroseconds = subbeatCountToNanoseconds(bpm, 1) / 1000;     // SyncTimer ctor, :749
        jackLatency = 0; cumulativeBeat = 0; beat = 0; stepReadHeadOnStart = 0;
        jackPlayheadReturn = 0; jackSubbeatLengthInMicrosecondsReturn = 0;
        updateScheduleAheadAmount();                                                       // ctor, :771
    }

    static uint64_t subbeatCountToNanoseconds(uint64_t bpm, uint64_t subBeatCount)        // :180-183
    {
        return (subBeatCount * 60000000000ULL) / (bpm * BeatSubdivisions);
    }
    static float nanosecondsToSubbeatCount(uint64_t bpm, uint64_t nanoseconds)           // :184-187
    {
        return (float)(nanoseconds / (60000000000ULL / (bpm * BeatSubdivisions)));
    }
    static uint64_t add_u64_double(uint64_t a, double b) { return (uint64_t)((double)a + b); }   // quint64 += double

    // client_latency_callback / ctor (:730-741,770-771): latency in ms from the JACK buffer size and sample rate
    void set_jack_latency(uint32_t bufferSize, double sampleRate)
    {
        const uint64_t newLatency = (uint64_t)((1000 * (double)bufferSize) / sampleRate);
        if (newLatency != jackLatency) { jackLatency = newLatency; updateScheduleAheadAmount(); }
    }
    void updateScheduleAheadAmount()                                                       // :704-707
    {
        scheduleAheadAmount = (uint64_t)(nanosecondsToSubbeatCount(bpm, (uint64_t)(jackLatency * (float)1000000)) + 1);
    }

    Step &delayedStep(uint64_t delay, bool ensureFresh = true)                             // :364-378
    {
        uint64_t step;
        if (isPaused) step = (stepReadHead + delay + 1) % StepRingCount;
        else step = (stepReadHeadOnStart + std::max(cumulativeBeat + delay, jackPlayhead + 1)) % StepRingCount;
        if (ensureFresh) this->ensureFresh(step);
        return ring[step];
    }

    void scheduleClipCommand(const zlhip_clip_command &command, uint64_t delay)           // :1011-1048
    {
        zl_schedule_into_step(delayedStep(delay).clipCommands, command);
    }

    void setBpm(uint64_t newBpm)                                                          // :954-975
    {
        if (bpm != newBpm) {
            bpm = newBpm;
            jackSubbeatLengthInMicroseconds = subbeatCountToNanoseconds(bpm, 1) / 1000;
            updateScheduleAheadAmount();
            delayedStep(0).bpmCommands.push_back((int)newBpm);                              // scheduleTimerCommand(0, SetBpmOperation)
        }
    }

    void start(int newBpm)                                                                // :870-879
    {
        setBpm((uint64_t)newBpm);
        stepReadHeadOnStart = stepReadHead;
        threadPaused = false; isPaused = false;                                            // timerThread->resume() -> pausedChanged
    }

    // SyncTimer::stop (:881-925).  Every command of a step that has not been played is run through scheduleClipCommand at
    // delay 0 with its volume set to 0 ("so we don't make the users' ears bleed") and the step is marked played.  The target of
    // delay 0 is the step behind the read head -- itself one of the steps walked: commands of the read-head step and of the
    // target step fold into the target, which is then marked played with them (they never reach the sampler) unless a later
    // step holds commands too, whose re-scheduling finds the target played, clears it (ensureFresh) and fills it anew.
    void stop()
    {
        threadPaused = true; isPaused = true;
        beat = 0; cumulativeBeat = 0; jackPlayhead = 0;
        // the reference walks the whole ring from the read head; only unplayed steps matter, and only the target of delay 0 -- the step
        // behind the read head, offset 1 of the walk -- can become unplayed DURING the walk (when the read-head step re-schedules into
        // it).  So: the unplayed steps in walk order, with the target looked at in its turn whether it was unplayed before or not.
        pruneFresh();
        const uint64_t target = (stepReadHead + 1) % StepRingCount;
        std::vector<uint32_t> walk(freshSteps);
        std::sort(walk.begin(), walk.end(), [this](uint32_t a, uint32_t b) {
            return (a + StepRingCount - stepReadHead) % StepRingCount < (b + StepRingCount - stepReadHead) % StepRingCount; });
        bool targetDone = false;
        auto visit = [&](uint64_t idx) {
            Step &sd = ring[idx];
            if (sd.played) return;
            // (the target step itself: each of its commands is equivalent to itself, so it folds into the step it is already in
            // and nothing is appended -- the step is marked played with its commands, which are never dispatched)
            if (idx != target) {
                const std::vector<zlhip_clip_command> cmds(sd.clipCommands);
                for (zlhip_clip_command c : cmds) { c.change_volume = 1; c.volume = 0; scheduleClipCommand(c, 0); }
            }
            ring[idx].played = true;
        };
        for (uint32_t idx : walk) {
            const uint64_t off = (idx + StepRingCount - stepReadHead) % StepRingCount;
            if (off >= 1 && !targetDone) { visit(target); targetDone = true; if (idx == target) continue; }
            visit(idx);
        }
        if (!targetDone) visit(target);
        pruneFresh();
    }

    // the clock's tick: the timer thread calls this once per subbeat interval while it runs (:391-418)
    std::vector<int> *beatSink = nullptr;                     // where the registered timer callbacks' argument goes (callbacks[i](beat), :397-399)
    void hi_res_timer_callback()
    {
        while (cumulativeBeat < (jackPlayhead + (scheduleAheadAmount * 2))) {
            if (beatSink) beatSink->push_back(beat);
            beat = (beat + 1) % (int)(BeatSubdivisions * 4);
            ++cumulativeBeat;
        }
    }

    void queueClipToStartOnChannel(int32_t clip, int midiChannel)                          // :815-832
    {
        zlhip_clip_command command;
        zlhip_clip_command_clear_inline(command);
        command.clip = clip; command.midi_channel = midiChannel; command.midi_note = 60;
        command.change_volume = 1; command.volume = 1.0f; command.looping = 1;
        command.stop_playback = 1; command.start_playback = 1;
        const uint64_t nextZeroBeat = threadPaused ? 0 : (BeatSubdivisions * 4) - (cumulativeBeat % (BeatSubdivisions * 4));
        scheduleClipCommand(command, cumulativeBeat + nextZeroBeat < jackPlayhead ? nextZeroBeat + BeatSubdivisions * 4 : nextZeroBeat);
    }

    void queueClipToStopOnChannel(int32_t clip, int midiChannel)                           // :834-860
    {
        pruneFresh();
        for (uint32_t idx : freshSteps) {                                                  // every unplayed step (:837-850)
            Step &sd = ring[idx];
            for (size_t i = 0; i < sd.clipCommands.size(); ++i)
                if (sd.clipCommands[i].clip == clip) { sd.clipCommands.erase(sd.clipCommands.begin() + (long)i); break; }
        }
        zlhip_clip_command command;
        zlhip_clip_command_clear_inline(command);
        command.clip = clip; command.midi_channel = midiChannel; command.midi_note = 60; command.stop_playback = 1;
        delayedStep(0).clipCommands.push_back(command);                                    // appended, not merged (:858-859)
    }

    // SyncTimerPrivate::process (:452-702), the parts that concern clip commands and the playhead.  Commands of the steps that
    // fall due in [current_usecs, next_usecs) are appended to `out` with the playhead they are dispatched with (:553-558).
    void process(uint32_t nframes, uint64_t current_usecs, uint64_t next_usecs, float period_usecs, std::vector<ZlDispatch> &out)
    {
        (void)period_usecs;
        if (freshSteps.size() > 1024) pruneFresh();                                         // (entries of steps played long ago)
        double thisStepBpm = jackPlayheadBpm;
        double thisStepSubbeatLengthInMicroseconds = (double)subbeatCountToNanoseconds((uint64_t)jackPlayheadBpm, 1) / 1000.0;   // :484
        jackPlayheadReturn = jackPlayhead;
        jackSubbeatLengthInMicrosecondsReturn = (uint64_t)thisStepSubbeatLengthInMicroseconds;
        if (!isPaused) {
            if (jackPlayhead == 0) jackNextPlaybackPosition = current_usecs;                // :490-498
        }
        if (stepNextPlaybackPosition == 0) stepNextPlaybackPosition = current_usecs;        // :500-502
        uint32_t firstAvailableFrame = 0;
        while (stepNextPlaybackPosition < next_usecs && firstAvailableFrame < nframes) {    // :512
            Step &stepData = ring[stepReadHead];
            stepReadHead = (stepReadHead + 1) % StepRingCount;
            if (stepNextPlaybackPosition <= current_usecs) {                                // :517-523
                ++firstAvailableFrame;
            } else {
                const uint64_t microsecondsPerFrame = (next_usecs - current_usecs) / nframes;
                const uint64_t rel = microsecondsPerFrame ? (stepNextPlaybackPosition - current_usecs) / microsecondsPerFrame : 0;
                firstAvailableFrame = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(rel, firstAvailableFrame), nframes - 1);
            }
            if (!stepData.played) {
                for (const zlhip_clip_command &c : stepData.clipCommands) out.push_back(ZlDispatch{ c, jackPlayhead });   // :553-558
                // (indexed: setBpm below may append to the very step being played -- its target is the step behind the read head only
                // when the ring has wrapped, but the list must survive the append either way)
                for (size_t i = 0; i < stepData.bpmCommands.size(); ++i) {                  // SetBpmOperation, :606-612
                    const uint64_t newBpm = std::min<uint64_t>(std::max<uint64_t>((uint64_t)stepData.bpmCommands[i], 50), 200);
                    setBpm(newBpm);
                    thisStepBpm = (double)newBpm;
                }
                stepData.played = true;
            }
            if (jackPlayheadBpm != thisStepBpm) {                                           // :634-639
                jackPlayheadBpm = thisStepBpm;
                thisStepSubbeatLengthInMicroseconds = (double)(subbeatCountToNanoseconds((uint64_t)jackPlayheadBpm, 1) / 1000);
            }
            if (!isPaused) {                                                                // :660-667
                ++jackPlayhead;
                jackNextPlaybackPosition = add_u64_double(jackNextPlaybackPosition, thisStepSubbeatLengthInMicroseconds);
            }
            stepNextPlaybackPosition = add_u64_double(stepNextPlaybackPosition, thisStepSubbeatLengthInMicroseconds);   // :671
        }
    }

    // the getters SamplerSynthVoice::process reads (:990-1009)
    uint64_t jackPlayheadGetter() const { return threadPaused ? stepReadHead : jackPlayhead; }
    uint64_t jackPlayheadUsecsGetter() const { return threadPaused ? stepNextPlaybackPosition : jackNextPlaybackPosition; }

    static void zlhip_clip_command_clear_inline(zlhip_clip_command &c)                     // ClipCommand ctor defaults (ClipCommand.h:13-32)
    {
        c = zlhip_clip_command{};
        c.clip = -1; c.midi_note = -1; c.midi_channel = -1; c.slice = -1;
    }
};

// Host-owned transport (the reference's own SyncTimer stays in charge and the host hands its getters over in zlhip_clock): every
// command scheduled since the previous cycle with the same due tick forms one step -- with the merge above -- and is dispatched at
// the top of the first cycle whose playhead has reached it.  A command scheduled at playhead p with delay d is due at p + d.
struct ZlHostTransportSchedule {
    struct Step { uint64_t due; std::vector<zlhip_clip_command> clipCommands; };
    std::vector<Step> steps;                                     // ordered by due tick
    uint64_t lastPlayhead = 0;

    void scheduleClipCommand(const zlhip_clip_command &command, uint64_t delay)
    {
        const uint64_t due = lastPlayhead + delay;
        size_t i = 0;
        while (i < steps.size() && steps[i].due < due) ++i;
        if (i == steps.size() || steps[i].due != due) steps.insert(steps.begin() + (long)i, Step{ due, {} });
        zl_schedule_into_step(steps[i].clipCommands, command);
    }
    void process(uint64_t playhead, std::vector<ZlDispatch> &out)
    {
        lastPlayhead = playhead;
        size_t n = 0;
        while (n < steps.size() && steps[n].due <= playhead) {
            for (const zlhip_clip_command &c : steps[n].clipCommands) out.push_back(ZlDispatch{ c, playhead });
            ++n;
        }
        steps.erase(steps.begin(), steps.begin() + (long)n);
    }
    void clear() { steps.clear(); lastPlayhead = 0; }
};
