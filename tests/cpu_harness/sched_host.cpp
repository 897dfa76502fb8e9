// sched_host.cpp -- host build of libzl_amd/csrc/zl_sched.h for CPU-only tests: the product's ClipCommand scheduler
// (the step ring libzl_hotpath_cycle runs, and the host-transport schedule of libzl_hotpath_process) behind a flat C
// interface.  TEST INFRASTRUCTURE: the product reaches the same code through zl_libzl.cpp.
#include <cstring>
#include <vector>

#include "zl_sched.h"

extern "C" {

// (the engine library is not linked here)
void zlhip_clip_command_clear(zlhip_clip_command *c) { ZlStepSequencer::zlhip_clip_command_clear_inline(*c); }

struct ZlSchedHost { ZlStepSequencer seq; ZlHostTransportSchedule ext; std::vector<ZlDispatch> due; };

ZlSchedHost *zlsched_new(void) { return new ZlSchedHost(); }
void zlsched_free(ZlSchedHost *s) { delete s; }
void zlsched_schedule(ZlSchedHost *s, const zlhip_clip_command *c, uint64_t delay) { s->seq.scheduleClipCommand(*c, delay); }
void zlsched_set_latency(ZlSchedHost *s, uint32_t bufferSize, double sampleRate) { s->seq.set_jack_latency(bufferSize, sampleRate); }
void zlsched_set_bpm(ZlSchedHost *s, uint64_t bpm) { s->seq.setBpm(bpm); }
void zlsched_start(ZlSchedHost *s, int bpm) { s->seq.start(bpm); }
void zlsched_stop(ZlSchedHost *s) { s->seq.stop(); }
void zlsched_timer_callback(ZlSchedHost *s) { s->seq.hi_res_timer_callback(); }
void zlsched_queue_start(ZlSchedHost *s, int32_t clip, int channel) { s->seq.queueClipToStartOnChannel(clip, channel); }
void zlsched_queue_stop(ZlSchedHost *s, int32_t clip, int channel) { s->seq.queueClipToStopOnChannel(clip, channel); }
int32_t zlsched_process(ZlSchedHost *s, uint32_t nframes, uint64_t current_usecs, uint64_t next_usecs, float period_usecs, ZlDispatch *out, int32_t max_out)
{
    s->due.clear();
    s->seq.process(nframes, current_usecs, next_usecs, period_usecs, s->due);
    const int32_t n = (int32_t)s->due.size();
    for (int32_t i = 0; i < n && i < max_out; ++i) out[i] = s->due[(size_t)i];
    return n;
}
void zlsched_clock(const ZlSchedHost *s, zlhip_clock *out)
{
    out->jack_playhead = s->seq.jackPlayheadGetter();
    out->jack_playhead_usecs = s->seq.jackPlayheadUsecsGetter();
    out->jack_subbeat_length_usecs = s->seq.jackSubbeatLengthInMicroseconds;
}

// host-owned transport
void zlsched_ext_schedule(ZlSchedHost *s, const zlhip_clip_command *c, uint64_t delay) { s->ext.scheduleClipCommand(*c, delay); }
int32_t zlsched_ext_process(ZlSchedHost *s, uint64_t playhead, ZlDispatch *out, int32_t max_out)
{
    s->due.clear();
    s->ext.process(playhead, s->due);
    const int32_t n = (int32_t)s->due.size();
    for (int32_t i = 0; i < n && i < max_out; ++i) out[i] = s->due[(size_t)i];
    return n;
}

}  // extern "C"
