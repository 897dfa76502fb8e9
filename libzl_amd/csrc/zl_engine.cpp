// zl_engine.cpp -- host side of the C-ABI in include/zlhip.h.
//
// Owns the HBM layout (source arena, clip / sound / voice tables, per-batch plan records, bus and
// level buffers), mirrors the control-plane half of the reference's SamplerChannel (voice
// allocation and command merge, SamplerSynth.cpp:187-230; startNote's one-off math,
// SamplerSynthVoice.cpp:110-144) and launches the kernels of zl_kernels.hip.  There is no CPU
// render path in this library: every sample is produced by the HIP kernels.
#include <hip/hip_runtime.h>
#include <sys/resource.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/zlhip.h"
#include "zl_host.h"
#include "zl_kernels.h"
#include "zl_plan.h"
#include "zl_types.h"

// buses at least this wide are split into one voice per workgroup for single real-time blocks (pick_group)
#define ZL_RT_SPLIT_MIN_VOICES 32

namespace {

size_t g_alloc_bytes = 0;          // device bytes allocated by the engine being created (zlhip_engine_create is not re-entrant)
template <typename T> hipError_t dalloc(T **p, size_t n)
{
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    g_alloc_bytes += bytes;
    return hipMalloc((void **)p, bytes);
}

// host memory mapped into the device, grown by doubling (the buffer must be idle)
template <typename T> hipError_t grow_mapped(T **host, T **dev, size_t *cap, size_t need)
{
    if (need <= *cap) return hipSuccess;
    if (*host) { hipError_t r = hipHostFree(*host); *host = nullptr; *dev = nullptr; *cap = 0; if (r != hipSuccess) return r; }
    const size_t n = need * 2;
    hipError_t r = hipHostMalloc((void **)host, n * sizeof(T));
    if (r != hipSuccess) return r;
    r = hipHostGetDevicePointer((void **)dev, *host, 0);
    if (r == hipSuccess) *cap = n;
    return r;
}

}  // namespace

struct zlhip_engine {
    zlhip_config cfg{};
    int V = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    char devname[256] = {0};

    // HBM
    float *arena = nullptr; size_t arenaFloats = 0;
    // arena allocator: free extents (offset, floats), sorted by offset, neighbours coalesced on release -- clips are
    // loaded and destroyed freely (SamplerSynth::registerClip / unregisterClip, SamplerSynth.cpp:285-312)
    std::vector<std::pair<size_t, size_t>> arenaFree;
    // further arena segments, allocated when a source does not fit any more (clips are loaded freely; 288 GB of HBM): a source in
    // one of them is addressed like any other, by its float offset from `arena` -- taken modulo 2^64, so a segment below the first
    // one in the address space has a "negative" offset that the kernels' 64-bit address arithmetic wraps back
    struct ArenaSegment { float *p; size_t off, floats; };   // off: the segment's float offset from `arena`, modulo 2^64
    std::vector<ArenaSegment> arenaSegments;
    size_t arenaSegmentFloats = 0;       // floats in those segments (zlhip_memory_bytes)
    std::vector<size_t> soundFloats;     // per sound slot: floats it holds in the arena
    ZlSound *dSounds = nullptr; ZlClip *dClips = nullptr;
    ZlVoiceState *dVoices = nullptr;
    // K1 -> K2 records, double buffered so that planning window i+1 overlaps rendering window i
    struct PlanSet {
        ZlVoiceConst *vconst = nullptr; ZlRunList *runs = nullptr; ZlTSeg *tsegs = nullptr;
        ZlPlanHdr *hdr = nullptr; ZlPlanSeg0 *seg0 = nullptr; ZlPlanSeg1 *seg1 = nullptr;
        double *ctlP = nullptr; float *ctlEnv = nullptr;  // the window's pool of per-frame control slots (zl_plan.h, zl_ctl_alloc)
        unsigned long long *ctlNext = nullptr;            // its bump counter, never reset: a window's slots count from ctlBase
        unsigned long long ctlBase = 0;                   // host side: past every value the counter can have reached
        ZlSimConst *simConst = nullptr;
        float *partials = nullptr;
        hipEvent_t planned = nullptr, rendered = nullptr, k1done = nullptr;
        hipEvent_t renderedEv = nullptr; // the event that marks the end of the last rendering from this set (rendered, or a profiling event)
        bool used = false;               // `rendered` has been recorded at least once
    } ps[2];
    int windowBlocks = 0;                // plan_window_blocks when given (a fixed number of blocks per plan window)
    size_t windowFrames = 0;             // else a window is this many frames: 2048 blocks of 256 frames, more blocks when they are shorter
    int windowCap = 0;                   // blocks the K1 -> K2 record arrays hold
    size_t ctlPoolFrames = 0;            // frames of per-frame control a record set's pool holds (slots = this / nframes)
    int ctlSlotsOverride = -1;           // ZL_CTL_POOL_SLOTS (tests of the exhausted pool)
    hipStream_t planStream = nullptr;    // K0 + K1 (sequential per voice)
    hipStream_t lastRenderStream = nullptr;                          // the stream the previous call rendered on
    hipStream_t lastPlanStream = nullptr; hipEvent_t evPlanTail = nullptr;   // where the previous call planned (voice-state order)
    hipEvent_t lastPlanEvent = nullptr;  // marks the end of that planning: evPlanTail, or the call's `done` event when it planned on its render stream
    hipStream_t asmStream = nullptr;     // K1c of window w runs here, next to K1 of window w+1
    std::vector<std::pair<int, int>> wins;
    // Per-call resources, double buffered so that consecutive zlhip_render_batch calls pipeline: the host prepares
    // (and the planning stream plans) call i+1 while call i still renders; a slot is reused by call i+2.
    struct CallSlot {
        ZlClock *hClocks = nullptr, *hClocksDev = nullptr;   // the call's block clocks, in host memory mapped into the device: K1 reads them in place
        ZlPassParams *hPass = nullptr, *dPass = nullptr;   // fused fan-out parameters of the call
        // the call's voice operations, in host memory mapped into the device: K0 reads them in place (no copy command)
        ZlVoiceOp *hOps = nullptr, *hOpsDev = nullptr; ZlOpRange *hRanges = nullptr, *hRangesDev = nullptr;
        size_t opsCap = 0, rangesCap = 0;
        ZlClipEdit *hEdits = nullptr, *hEditsDev = nullptr; size_t editsCap = 0;   // the call's clip-parameter edits, same arrangement
        ZlReport *hReports = nullptr, *dReports = nullptr;
        float *hGain = nullptr;
        ZlBatchStats *hStats = nullptr, *dStats = nullptr;
        ZlReport *hReportsDev = nullptr; float *hGainDev = nullptr; ZlBatchStats *hStatsDev = nullptr;   // device views of the host buffers
        hipEvent_t evBegin = nullptr, evEnd = nullptr, done = nullptr;
        // `done` alternates between two events from one use of the slot to the next, so that the NEXT call (the other slot) can take this
        // call's completion as the begin of its own profile -- one event-record packet fewer between two calls' render kernels -- and
        // still find it intact when it is harvested, two calls later
        hipEvent_t doneEv[2] = {nullptr, nullptr}; unsigned doneGen = 0;
        hipEvent_t beginEv = nullptr;    // what this call's profile counts from: its own evBegin, or the previous call's `done`
        std::vector<hipEvent_t> evK2;    // [2 * max windows] start/end of every K2 launch (profiling)
        int windows = 0;
        bool inflight = false, profiled = false;
        bool fusedDone = false;          // the call's last K2 launch published the reports and carries `done` as its stop event (no report kernel)
    } slots[2];
    unsigned callIndex = 0, setPhase = 0;
    CallSlot *latest = nullptr;          // slot of the most recent call
    zlhip_timings totals{}; int totalCalls = 0;   // sums over harvested calls (zlhip_profile_totals)
    float *dGain = nullptr;
    ZlPassCache *dPassCache = nullptr;   // per voice: the recorded pass of its loop (zl_plan.h, replay_cached_pass)
    float *dBus = nullptr;
    ZlBlockLevels *dLevels = nullptr; ZlLevelsState *dLevelState = nullptr;
    int32_t *dTrace = nullptr; ZlPassParams *dPass = nullptr;
    size_t traceInts = 0;
    int maxGroups = 1;

    // pinned host staging
    float *hBus = nullptr, *hBusDev = nullptr;   // one real-time block, host memory mapped into the device
    // one real-time block's JackPassthrough fan-out [B][6][max_frames] (zlhip_render_fanout), written by the kernels like hBus; the
    // parameter table the resident kernel reads ([B], mapped host memory) and its version (never 0; moved when an entry changes)
    float *hFan = nullptr, *hFanDev = nullptr;
    ZlPassParams *hPassRt = nullptr, *hPassRtDev = nullptr;
    uint32_t passSeq = 1;
    // zero-copy delivery of a real-time cycle: when the caller's out_left / out_right (/ fan_out) are page-locked and mapped (zlhip_host_alloc,
    // hipHostMalloc, hipHostRegister) the kernels write them directly -- no copy on the host behind the cycle (24 KB + 72 KB for 12 buses:
    // 1.6 + 5 us of reading lines the device has just written).  The device views of the last buffers seen are kept.
    struct OutViews { const void *hL = nullptr, *hR = nullptr, *hF = nullptr; float *dL = nullptr, *dR = nullptr, *dF = nullptr; bool ok = false; } outViews;
    bool directOut = true;               // ZL_RT_DIRECT_OUT=0: always through the staging rows
    ZlLevelsState *hLevelState = nullptr;

    // host mirrors
    ZlHostControl hc;                    // voices / sounds / clip parameters / pending ops (zl_host.h)
    std::vector<ZlOpRange> ranges;

    // last batch
    int lastK = 0, lastN = 0, lastWindows = 0; float *lastBus = nullptr; bool outstanding = false; bool reportsFresh = false;
    bool trace = false; int traceK = 0, traceN = 0;
    int forceSlow = 0;
    size_t deviceBytes = 0;              // HBM the engine allocated at creation (arena included)
    int staged = 0;                      // K2 variant with LDS-staged source windows (zl_kernels.hip), chosen per mode at creation

    // resident real-time kernel (zl_k_rt_loop): the mailbox in mapped host memory, its stream, what it was launched for
    struct Rt {
        bool enabled = false, running = false; int wide = -1;    // wide: 1 always, 0 never, -1 only with one voice per workgroup
        ZlRtShared *h = nullptr, *d = nullptr;
        ZlRtDev *dev = nullptr;                            // the kernel's own hand-off words in HBM
        ZlOpRange *devRanges = nullptr;                    // wide buses: workgroup 0's copy of a block's operation ranges
        int capacity[2] = {-1, -1};                        // workgroups of the kernel the device holds at once (narrow, wide); -1 = not asked yet
        int vw = 0;                                        // wide buses: voices per resident workgroup (a divisor of voices_per_bus)
        double share = 0.0;                                // of the device's resident-workgroup capacity this engine's kernel takes (rt_eligible)
        std::atomic<bool> inCycle{false};                  // a cycle is being rendered through the resident kernel (its share stays taken even if the kernel has just left)
        int maxFrames = 4096;                              // longest period the resident kernel takes (ZL_RT_MAX_FRAMES)
        hipStream_t stream = nullptr;
        int nframes = 0;
        unsigned long long seq = 0;
        unsigned long long starts = 0, cycles = 0;        // launches of the resident kernel, cycles it rendered (zlhip_rt_stats)
        unsigned long long idleTicks = 20000000ull;       // 200 ms of the 100 MHz counter without a block: the kernel leaves
        bool stampsOn = false; double stampSum[6] = {0, 0, 0, 0, 0, 0}; unsigned long long stampN = 0;   // ZL_RT_STAMPS=1: stage times (us)
        // where the last real-time cycle spent its time, seen from the host (zlhip_rt_last_cycle): a worst case has to be attributable
        bool traceOn = false; double slowUs = 0.0;         // ZL_RT_TRACE=1; ZL_RT_TRACE_SLOW_US=<n>: cycles longer than that are reported on stderr
        zlhip_rt_cycle_trace last{};
    } rt;

    // offline bounce (zlhip_bounce): the device bus buffers of two chunks rendered into in turn, their 16-bit versions, the copy stream.
    // A chunk is ONE zlhip_render_batch call; every plan window of it is delivered as soon as its render kernel has finished
    // (the sink below, called from the window loop), while the following windows render.
    struct Bounce {
        static constexpr int NBUF = 2, NEV = 16;
        float *bus[NBUF] = {nullptr, nullptr}; int16_t *pcm[NBUF] = {nullptr, nullptr};
        size_t busFloats = 0, pcmFrames = 0;
        hipStream_t copyStream = nullptr;
        hipEvent_t copied[NBUF] = {nullptr, nullptr};      // the last delivery out of a chunk buffer
        hipEvent_t winEv[NEV] = {};                        // "window rendered (and converted)": waited for by the copy stream
        unsigned winNext = 0;
        bool active = false;                 // inside zlhip_bounce: every chunk plans on the planning stream
        struct Sink {
            bool on = false, pcm = false;
            char *hostBase = nullptr;        // the chunk's first frame in the caller's buffer
            void *hostDev = nullptr;         // direct delivery: the device view of hostBase (page-locked memory only), else nullptr
            size_t totalFrames = 0;          // frames per row of the caller's buffer (the whole bounce)
            int16_t *pcmDev = nullptr;       // the chunk's 16-bit staging buffer [B][chunk frames][2]
        } sink;
    } bnc;

    bool failed = false;                 // the resident kernel stopped answering in the middle of a cycle: the voice table is undefined (zlhip_render)

    // profiling
    bool profiling = false; hipEvent_t evJoin = nullptr;
    hipEvent_t joins[2] = {nullptr, nullptr};   // events on caller streams the host still has to wait for (engine_wait)
    zlhip_timings timings{};
};

#define ZL_HIP(e, call)                                                                        \
    do {                                                                                       \
        hipError_t st_ = (call);                                                               \
        if (st_ != hipSuccess) {                                                               \
            (e)->err = std::string(#call) + ": " + hipGetErrorString(st_);                     \
            return ZLHIP_ERR_HIP;                                                              \
        }                                                                                      \
    } while (0)

#define ZL_KERNEL(e, call)                                                                     \
    do {                                                                                       \
        int st_ = (call);                                                                      \
        if (st_ != 0) {                                                                        \
            (e)->err = std::string(#call) + ": " + hipGetErrorString((hipError_t)st_);         \
            return ZLHIP_ERR_HIP;                                                              \
        }                                                                                      \
    } while (0)

static int fail(zlhip_engine *e, int code, const char *msg)
{
    if (e) e->err = msg;
    return code;
}

// Host-side wait for everything the engine has queued, on its own stream and on caller-provided streams.  (Work on
// a caller's stream is joined through events waited for on the host, not by queueing a wait on the engine's stream:
// an idle HIP stream holds no hardware queue, and the process has few of them.)
static int engine_wait(zlhip_engine *e)
{
    for (hipEvent_t &ev : e->joins) {
        if (ev) { ZL_HIP(e, hipEventSynchronize(ev)); ev = nullptr; }
    }
    ZL_HIP(e, hipStreamSynchronize(e->stream));
    e->outstanding = false;
    return ZLHIP_OK;
}


// ---- resident kernels and device-synchronising HIP calls -----------------------------------------------
// hipFree, hipHostFree and hipDeviceSynchronize wait for every kernel on the device -- also for the resident real-time kernel of
// ANY engine of the process: up to its idle timeout, or for ever while its host keeps posting cycles.  Engines whose resident
// kernel may be on the device are registered here; a thread about to make such a call asks all of them (other than its own
// engine, which it stops itself) to leave through their mailbox (ZlRtShared::yield), waits until they have, and holds the
// request for the duration of the call.  The owners notice that their kernel has left at their next cycle (rt_render) and
// render that cycle with launches while a request is held; afterwards they start their kernel again.
namespace {
struct RtRegistry {
    std::mutex mu;
    std::vector<zlhip_engine *> engines;
    std::atomic<int> quiescing{0};
} g_rt;

void rt_unregister(zlhip_engine *e)
{
    std::lock_guard<std::mutex> lk(g_rt.mu);
    g_rt.engines.erase(std::remove(g_rt.engines.begin(), g_rt.engines.end(), e), g_rt.engines.end());
}

struct ZlQuiesce {
    explicit ZlQuiesce(const zlhip_engine *self)
    {
        g_rt.quiescing.fetch_add(1, std::memory_order_acq_rel);
        // The registry's lock is held in short bursts only -- to post the request and to look at the kernels' states -- never while
        // waiting: rt_start and rt_unregister take the same lock on a JACK thread (an audio thread whose kernel had just
        // left could sit behind this wait for up to a second).  The engine list is read afresh under the lock every round, so an
        // engine destroyed meanwhile is simply no longer there.
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            bool all = true;
            {
                std::lock_guard<std::mutex> lk(g_rt.mu);
                for (zlhip_engine *o : g_rt.engines) {
                    if (o == self || !o->rt.h) continue;
                    __atomic_store_n(&o->rt.h->yield, 1u, __ATOMIC_RELEASE);
                    // (a kernel that was just launched reads the request at its first idle poll)
                    if (__atomic_load_n(&o->rt.h->state, __ATOMIC_ACQUIRE) != 2u) all = false;
                }
            }
            // (one second: thousands of cycles)
            if (all || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(1)) break;
            for (int i = 0; i < 256; ++i) __builtin_ia32_pause();
        }
    }
    ~ZlQuiesce()
    {
        std::lock_guard<std::mutex> lk(g_rt.mu);
        if (g_rt.quiescing.fetch_sub(1, std::memory_order_acq_rel) == 1)
            for (zlhip_engine *o : g_rt.engines) if (o->rt.h) __atomic_store_n(&o->rt.h->yield, 0u, __ATOMIC_RELEASE);
    }
    ZlQuiesce(const ZlQuiesce &) = delete;
    ZlQuiesce &operator=(const ZlQuiesce &) = delete;
};
}  // namespace

// ---- resident real-time kernel ------------------------------------------------------------------
// Asks the resident kernel to leave and waits for it.  Called before anything else touches the voice table, the clip / sound
// tables, the arena or the plan records from outside (batches, uploads, parameter changes, destruction): while the kernel is
// resident its caches are not refreshed by other engines' writes.
static int rt_stop(zlhip_engine *e)
{
    if (!e->rt.running) return ZLHIP_OK;
    __atomic_store_n(&e->rt.h->stop, 1u, __ATOMIC_RELEASE);
    ZL_HIP(e, hipStreamSynchronize(e->rt.stream));
    __atomic_store_n(&e->rt.h->stop, 0u, __ATOMIC_RELEASE);
    e->rt.running = false;
    rt_unregister(e);
    return ZLHIP_OK;
}

extern "C" {

int zlhip_abi_version(void) { return ZLHIP_ABI_VERSION; }

const char *zlhip_strerror(int status)
{
    switch (status) {
    case ZLHIP_OK: return "ok";
    case ZLHIP_ERR_INVALID: return "invalid argument";
    case ZLHIP_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU render path)";
    case ZLHIP_ERR_HIP: return "HIP runtime error";
    case ZLHIP_ERR_CAPACITY: return "capacity exceeded";
    case ZLHIP_ERR_STATE: return "invalid state";
    default: return "unknown status";
    }
}

const char *zlhip_last_error(const zlhip_engine *e) { return e ? e->err.c_str() : "null engine"; }

void zlhip_config_default(zlhip_config *cfg)
{
    std::memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = sizeof *cfg;
    cfg->device = 0;
    cfg->num_buses = 12;             // SamplerSynth.cpp:258
    cfg->voices_per_bus = 8;         // SamplerSynth.cpp:23
    cfg->max_frames = 1024;
    cfg->max_batch_blocks = 64;
    cfg->max_sounds = 1024;
    cfg->mode = ZLHIP_MODE_FAITHFUL;
    cfg->playback_sample_rate = 48000.0;
    cfg->sound_arena_bytes = 256ull << 20;
    cfg->voices_per_task = 0;
    cfg->rt_idle_timeout_us = 0;     // 200 ms
    cfg->sound_arena_max_bytes = 0;  // the arena grows in segments as clips are loaded
}

void zlhip_engine_destroy(zlhip_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)rt_stop(e);
    rt_unregister(e);
    ZlQuiesce quiet(e);                 // the frees below wait for the device: no other engine's resident kernel may be on it
    if (e->rt.stampsOn && e->rt.stampN)
        std::fprintf(stderr, "zlhip resident kernel (workgroup 0), mean us per block over %llu blocks: K0 %.2f  K1 %.2f  K1c %.2f  K2 %.2f  reports + release %.2f\n", e->rt.stampN,
                     e->rt.stampSum[0] / e->rt.stampN, e->rt.stampSum[1] / e->rt.stampN, e->rt.stampSum[2] / e->rt.stampN, e->rt.stampSum[3] / e->rt.stampN,
                     e->rt.stampSum[4] / e->rt.stampN);
    if (e->rt.stream) (void)hipStreamDestroy(e->rt.stream);
    if (e->rt.h) (void)hipHostFree(e->rt.h);
    if (e->rt.dev) (void)hipFree(e->rt.dev);
    if (e->rt.devRanges) (void)hipFree(e->rt.devRanges);
    for (auto &c : e->slots) if (c.inflight && c.done) (void)hipEventSynchronize(c.done);   // calls queued on a caller's stream
    for (hipEvent_t ev : e->joins) if (ev) (void)hipEventSynchronize(ev);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->planStream) (void)hipStreamSynchronize(e->planStream);
    if (e->asmStream) (void)hipStreamSynchronize(e->asmStream);
    for (auto &seg : e->arenaSegments) if (seg.p) (void)hipFree(seg.p);
    void *dev[] = { e->arena, e->dSounds, e->dClips, e->dVoices, e->dGain, e->dBus, e->dLevels, e->dLevelState, e->dTrace, e->dPass, e->dPassCache };
    for (void *p : dev) if (p) (void)hipFree(p);
    for (auto &q : e->ps) {
        void *pd[] = { q.vconst, q.runs, q.tsegs, q.hdr, q.seg0, q.seg1, q.ctlP, q.ctlEnv, q.partials, q.ctlNext, q.simConst };
        for (void *p : pd) if (p) (void)hipFree(p);
        if (q.planned) (void)hipEventDestroy(q.planned);
        if (q.rendered) (void)hipEventDestroy(q.rendered);
        if (q.k1done) (void)hipEventDestroy(q.k1done);
    }
    for (auto &c : e->slots) {
        void *cd[] = { c.dReports, c.dStats, c.dPass };
        for (void *p : cd) if (p) (void)hipFree(p);
        void *ch[] = { c.hClocks, c.hReports, c.hGain, c.hStats, c.hPass, c.hOps, c.hRanges, c.hEdits };
        for (void *p : ch) if (p) (void)hipHostFree(p);
        for (auto &x : c.evK2) if (x) (void)hipEventDestroy(x);
        hipEvent_t evs[] = { c.evBegin, c.evEnd, c.doneEv[0], c.doneEv[1] };
        for (hipEvent_t x : evs) if (x) (void)hipEventDestroy(x);
    }
    if (e->bnc.copyStream) { (void)hipStreamSynchronize(e->bnc.copyStream); (void)hipStreamDestroy(e->bnc.copyStream); }
    for (int i = 0; i < zlhip_engine::Bounce::NBUF; ++i) {
        if (e->bnc.bus[i]) (void)hipFree(e->bnc.bus[i]);
        if (e->bnc.pcm[i]) (void)hipFree(e->bnc.pcm[i]);
        if (e->bnc.copied[i]) (void)hipEventDestroy(e->bnc.copied[i]);
    }
    for (hipEvent_t ev : e->bnc.winEv) if (ev) (void)hipEventDestroy(ev);
    if (e->planStream) (void)hipStreamDestroy(e->planStream);
    if (e->asmStream) (void)hipStreamDestroy(e->asmStream);
    void *host[] = { e->hBus, e->hLevelState, e->hFan, e->hPassRt };
    for (void *p : host) if (p) (void)hipHostFree(p);
    if (e->evJoin) (void)hipEventDestroy(e->evJoin);
    if (e->evPlanTail) (void)hipEventDestroy(e->evPlanTail);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int zlhip_engine_create(const zlhip_config *cfg_in, zlhip_engine **out)
{
    if (!cfg_in || !out) return ZLHIP_ERR_INVALID;
    *out = nullptr;
    // struct_size: a caller built against an earlier header passes a shorter struct; the fields it does not know keep their defaults
    zlhip_config cfg_full;
    zlhip_config_default(&cfg_full);
    {
        const size_t minSize = offsetof(zlhip_config, plan_window_blocks) + sizeof(int32_t);     // ABI version 1
        const size_t have = cfg_in->struct_size == 0 ? minSize : (size_t)cfg_in->struct_size;
        if (have < minSize) return ZLHIP_ERR_INVALID;
        std::memcpy(&cfg_full, cfg_in, std::min(have, sizeof cfg_full));
        cfg_full.struct_size = sizeof cfg_full;
    }
    const zlhip_config *cfg = &cfg_full;
    if (cfg->rt_idle_timeout_us < 0) return ZLHIP_ERR_INVALID;
    if (cfg->num_buses < 1 || cfg->voices_per_bus < 1 || cfg->max_frames < 1 || cfg->max_frames > 4096
        || cfg->max_batch_blocks < 1 || cfg->max_sounds < 1
        || !(cfg->playback_sample_rate > 0.0) || (cfg->mode & ~7u))
        return ZLHIP_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
        return ZLHIP_ERR_NO_DEVICE;
    if (hipSetDevice(cfg->device) != hipSuccess) return ZLHIP_ERR_NO_DEVICE;

    zlhip_engine *e = new zlhip_engine();
    g_alloc_bytes = 0;
    e->cfg = *cfg;
    e->device = cfg->device;
    e->V = cfg->num_buses * cfg->voices_per_bus;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) std::snprintf(e->devname, sizeof e->devname, "%s (%s)", prop.name, prop.gcnArchName);

    const size_t V = (size_t)e->V, K = (size_t)cfg->max_batch_blocks, N = (size_t)cfg->max_frames, B = (size_t)cfg->num_buses;
    e->maxGroups = cfg->voices_per_task > 0 ? (cfg->voices_per_bus + cfg->voices_per_task - 1) / cfg->voices_per_task : 1;
    e->arenaFloats = (size_t)(cfg->sound_arena_bytes / sizeof(float));

    int rc = ZLHIP_OK;
    auto chk = [&](hipError_t st, const char *what) {
        if (st != hipSuccess && rc == ZLHIP_OK) { e->err = std::string(what) + ": " + hipGetErrorString(st); rc = ZLHIP_ERR_HIP; }
    };
    chk(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking), "hipStreamCreate");
    chk(dalloc(&e->arena, e->arenaFloats + 1024), "arena");     // + 4 KB: wide loads at the end of the last source stay inside
    chk(dalloc(&e->dSounds, (size_t)cfg->max_sounds), "sounds");
    chk(dalloc(&e->dClips, (size_t)cfg->max_sounds), "clips");
    chk(dalloc(&e->dVoices, V), "voices");
    {
        // plan window: a call is cut into windows of ~512 Ki frames (2048 blocks of 256 frames; K2 launches of that
        // size reach the kernel's steady-state bandwidth) or of plan_window_blocks blocks when that is given; the
        // K1 -> K2 records are sized for one window and double buffered
        e->windowBlocks = cfg->plan_window_blocks > 0 ? cfg->plan_window_blocks : 0;
        // ... for 1024 voices; engines with fewer voices get proportionally longer windows (the same number of
        // voice-frames per K2 launch, the same record memory), up to 16 Mi frames
        e->windowFrames = std::min<size_t>((size_t)16 << 20, std::max<size_t>((size_t)2048 * 256, ((size_t)2048 * 256 * 1024) / V));
        int w = e->windowBlocks > 0 ? e->windowBlocks : (int)(e->windowFrames / 64);
        if (w > cfg->max_batch_blocks) w = cfg->max_batch_blocks;
        e->windowCap = w;
        size_t ctlFrames = e->windowBlocks > 0 ? (size_t)e->windowBlocks * N : std::max(e->windowFrames, N);
        ctlFrames = std::min(ctlFrames, K * N);
        const size_t W = (size_t)w;
        {
            // planning is a small latency-bound kernel that rendering waits for: give its stream the highest priority
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            chk(hipStreamCreateWithPriority(&e->planStream, hipStreamNonBlocking, hi), "plan stream");
            // a second planning stream pays only when it gets a hardware queue of its own (the runtime's default is 4
            // per process: two streams sharing one serialise behind each other's kernels)
            const char *hwq = std::getenv("GPU_MAX_HW_QUEUES");
            if (hwq && std::atoi(hwq) >= 6) chk(hipStreamCreateWithPriority(&e->asmStream, hipStreamNonBlocking, hi), "assemble stream");
        }
        const size_t minWindow = e->windowBlocks > 0 ? (size_t)e->windowBlocks : std::max<size_t>(1, e->windowFrames / N);
        // two record sets: windows of one call, and consecutive calls, are planned while the previous one renders
        // (engines for single real-time blocks keep one)
        const int nsets = (K > minWindow || K * N >= 65536) ? 2 : 1;
        for (int i = 0; i < 2; ++i) {
            zlhip_engine::PlanSet &q = e->ps[i];
            chk(hipEventCreateWithFlags(&q.planned, hipEventDisableTiming), "event");
            chk(hipEventCreateWithFlags(&q.rendered, hipEventDisableTiming), "event");
            chk(hipEventCreateWithFlags(&q.k1done, hipEventDisableTiming), "event");
            if (i >= nsets) continue;
            chk(dalloc(&q.vconst, V), "vconst");
            chk(dalloc(&q.runs, V), "run lists");
            chk(dalloc(&q.tsegs, V * (size_t)ZL_MAXTSEG), "segment streams");
            chk(dalloc(&q.hdr, W * V), "plan headers");
            chk(dalloc(&q.seg0, W * V), "plan segment 0");
            chk(dalloc(&q.seg1, W * V), "plan segment 1");
            // per-frame control of the window's slow (block, voice)s: a pool of block-sized slots.  One slot for every (block, voice)
            // of a window would be 12 bytes per voice-frame (6.4 GB per set for 1024 voices); slow blocks are sparse (release tails of
            // one-shots, the block behind a loop restart), so the pool is capped -- ZL_CTL_POOL_MB per set, default 256 -- and a
            // window that exhausts it has K2 recompute the control of the blocks that got no slot (zl_slow_control)
            {
                const char *mb = std::getenv("ZL_CTL_POOL_MB");
                const size_t capFrames = (size_t)(mb ? std::max(1, std::atoi(mb)) : 256) * ((size_t)1 << 20) / 12;
                e->ctlPoolFrames = std::max<size_t>(N, std::min(ctlFrames * V, capFrames));
                const char *so = std::getenv("ZL_CTL_POOL_SLOTS");
                if (so) e->ctlSlotsOverride = std::max(0, std::atoi(so));
            }
            chk(dalloc(&q.ctlP, e->ctlPoolFrames), "ctlP");
            chk(dalloc(&q.ctlEnv, e->ctlPoolFrames), "ctlEnv");
            chk(dalloc(&q.ctlNext, 1), "ctlNext");
            chk(dalloc(&q.simConst, V), "simConst");
            if (rc == ZLHIP_OK) chk(hipMemsetAsync(q.ctlNext, 0, sizeof(unsigned long long), e->stream), "memset ctlNext");
            {
                // mix-group partials of a window; or, for the per-voice split of single real-time blocks, of one block
                size_t pf = e->maxGroups > 1 ? ctlFrames * B * (size_t)e->maxGroups * 2 : 1;
                if (cfg->voices_per_task <= 0 && cfg->voices_per_bus >= ZL_RT_SPLIT_MIN_VOICES) pf = std::max(pf, N * V * 2);
                chk(dalloc(&q.partials, pf), "partials");
            }
        }
        const size_t nwin = (K + minWindow - 1) / minWindow + 16; // + the doubling windows at the start of a call
        e->wins.reserve(nwin);
        for (auto &c : e->slots) {
            c.evK2.assign(2 * nwin, nullptr);
            for (auto &x : c.evK2) chk(hipEventCreate(&x), "hipEventCreate");
            chk(hipEventCreate(&c.evBegin), "hipEventCreate");
            chk(hipEventCreate(&c.evEnd), "hipEventCreate");
            chk(hipEventCreate(&c.doneEv[0]), "hipEventCreate");   // (they can ride on a kernel dispatch as its stop event)
            chk(hipEventCreate(&c.doneEv[1]), "hipEventCreate");
            c.done = c.doneEv[0];

            chk(dalloc(&c.dReports, V), "reports");
            chk(dalloc(&c.dStats, 1), "stats");
            chk(hipHostMalloc((void **)&c.hClocks, K * sizeof(ZlClock)), "hClocks");
            if (rc == ZLHIP_OK) chk(hipHostGetDevicePointer((void **)&c.hClocksDev, c.hClocks, 0), "hClocks device view");
            chk(dalloc(&c.dPass, B), "fan-out params");
            chk(hipHostMalloc((void **)&c.hPass, std::max<size_t>(B, 1) * sizeof(ZlPassParams)), "hPass");
            chk(hipHostMalloc((void **)&c.hReports, V * sizeof(ZlReport)), "hReports");
            chk(hipHostMalloc((void **)&c.hGain, V * sizeof(float)), "hGain");
            chk(hipHostMalloc((void **)&c.hStats, sizeof(ZlBatchStats)), "hStats");
            // room for a cycle's clip-parameter edits without ever growing (growing frees pinned memory, which waits for the device)
            c.editsCap = (size_t)std::min(cfg->max_sounds, 64);
            chk(hipHostMalloc((void **)&c.hEdits, c.editsCap * sizeof(ZlClipEdit)), "hEdits");
            if (rc == ZLHIP_OK) chk(hipHostGetDevicePointer((void **)&c.hEditsDev, c.hEdits, 0), "map hEdits");
            // ... and for its voice operations: two per voice (a stop and a start of every voice in one cycle) before anything grows
            if (rc == ZLHIP_OK) chk(grow_mapped(&c.hOps, &c.hOpsDev, &c.opsCap, V + 32), "hOps");
            if (rc == ZLHIP_OK) chk(grow_mapped(&c.hRanges, &c.hRangesDev, &c.rangesCap, V + 32), "hRanges");
            if (rc == ZLHIP_OK) {
                chk(hipHostGetDevicePointer((void **)&c.hReportsDev, c.hReports, 0), "map hReports");
                chk(hipHostGetDevicePointer((void **)&c.hGainDev, c.hGain, 0), "map hGain");
                chk(hipHostGetDevicePointer((void **)&c.hStatsDev, c.hStats, 0), "map hStats");
            }
        }
    }
    chk(dalloc(&e->dGain, V), "gain");
    chk(dalloc(&e->dPassCache, V), "pass cache");
    chk(dalloc(&e->dBus, B * 2 * K * N), "bus");
    chk(dalloc(&e->dLevels, K * B), "levels");
    chk(dalloc(&e->dLevelState, B), "levelState");
    chk(dalloc(&e->dPass, B), "passthrough params");
    chk(hipHostMalloc((void **)&e->hBus, B * 2 * N * sizeof(float)), "hBus");
    if (rc == ZLHIP_OK) chk(hipHostGetDevicePointer((void **)&e->hBusDev, e->hBus, 0), "map hBus");
    chk(hipHostMalloc((void **)&e->hFan, B * 6 * N * sizeof(float)), "hFan");
    if (rc == ZLHIP_OK) chk(hipHostGetDevicePointer((void **)&e->hFanDev, e->hFan, 0), "map hFan");
    chk(hipHostMalloc((void **)&e->hPassRt, B * sizeof(ZlPassParams)), "hPassRt");
    if (rc == ZLHIP_OK) { chk(hipHostGetDevicePointer((void **)&e->hPassRtDev, e->hPassRt, 0), "map hPassRt"); std::memset(e->hPassRt, 0xff, B * sizeof(ZlPassParams)); }
    chk(hipHostMalloc((void **)&e->hLevelState, B * sizeof(ZlLevelsState)), "hLevelState");
    chk(hipEventCreateWithFlags(&e->evJoin, hipEventDisableTiming), "hipEventCreate");
    chk(hipEventCreateWithFlags(&e->evPlanTail, hipEventDisableTiming), "hipEventCreate");
    if (rc == ZLHIP_OK) {
        chk(hipMemsetAsync(e->dSounds, 0, (size_t)cfg->max_sounds * sizeof(ZlSound), e->stream), "memset sounds");
        chk(hipMemsetAsync(e->dClips, 0, (size_t)cfg->max_sounds * sizeof(ZlClip), e->stream), "memset clips");
        chk(hipMemsetAsync(e->dVoices, 0, V * sizeof(ZlVoiceState), e->stream), "memset voices");
        chk(hipMemsetAsync(e->dPassCache, 0, V * sizeof(ZlPassCache), e->stream), "memset pass cache");
        chk(hipMemsetAsync(e->dLevelState, 0, B * sizeof(ZlLevelsState), e->stream), "memset levels");
        for (auto &c : e->slots) chk(hipMemsetAsync(c.dReports, 0, V * sizeof(ZlReport), e->stream), "memset reports");
        for (auto &c : e->slots) chk(hipMemsetAsync(c.dStats, 0, sizeof(ZlBatchStats), e->stream), "memset stats");
        chk(hipMemsetAsync(e->dBus, 0, B * 2 * K * N * sizeof(float), e->stream), "memset bus");
        chk(hipStreamSynchronize(e->stream), "sync");
    }
    if (rc != ZLHIP_OK) {
        std::fprintf(stderr, "zlhip_engine_create: %s\n", e->err.c_str());
        zlhip_engine_destroy(e);
        return rc;
    }
    {
        // K2 variant with LDS-staged source windows: ZL_K2_STAGED = 0 never (default: the register gather is faster in every
        // measured configuration, profiles/round2_b_lds_staging_ab.txt), 1 in Hermite mode only, 2 always
        const char *st = std::getenv("ZL_K2_STAGED");
        const int sel = st ? std::atoi(st) : 0;
        e->staged = sel >= 2 ? 1 : (sel == 1 && (cfg->mode & ZLHIP_MODE_HERMITE)) ? 1 : 0;
    }
    {
        // zlhip_render goes through the resident kernel wherever one workgroup per bus can render a cycle (rt_eligible; measured
        // p50 25 us / p99 29 us against 41 / 65-76 us for three launches + a completion event);
        // ZL_RT_PERSISTENT=0 keeps the launched path
        const char *rp = std::getenv("ZL_RT_PERSISTENT");
        e->rt.enabled = !(rp && std::atoi(rp) == 0);
        // Wide buses (32 voices and more) through the resident kernel too, a workgroup per few voices.  With ONE voice per workgroup
        // (engines of up to ~384 voices) it equals the launched path's median and has a far shorter tail (256 voices: 46 / 50 us p50 / p99
        // against 45 / 80): the default there.  With 2-8 voices per workgroup (1024 voices = 256 workgroups of 4) every workgroup walks
        // the K2 chain of its voices one after the other: 73 / 77 us against 55 / 84 -- a better tail, a worse median: opt-in.
        // ZL_RT_WIDE=1: whenever the device holds the workgroups; =0: never; unset: one voice per workgroup only.
        // (profiles/round3_rt_inline_ab.txt; before the kernel lost its scratch memory the figures were 100 against 53 and 66 against 44.)
        const char *rw = std::getenv("ZL_RT_WIDE");
        e->rt.wide = rw ? (std::atoi(rw) == 1 ? 1 : 0) : -1;
        e->rt.stampsOn = std::getenv("ZL_RT_STAMPS") != nullptr;
        if (const char *mf = std::getenv("ZL_RT_MAX_FRAMES")) e->rt.maxFrames = std::max(1, std::atoi(mf));
        if (const char *su = std::getenv("ZL_RT_TRACE_SLOW_US")) e->rt.slowUs = std::atof(su);
        if (const char *dz = std::getenv("ZL_RT_DIRECT_OUT")) e->directOut = std::atoi(dz) != 0;
        e->rt.traceOn = std::getenv("ZL_RT_TRACE") != nullptr || e->rt.slowUs > 0.0;
        if (cfg->rt_idle_timeout_us > 0) e->rt.idleTicks = (unsigned long long)cfg->rt_idle_timeout_us * 100ull;   // 100 MHz counter
    }
    e->hc.init(cfg->num_buses, cfg->voices_per_bus, cfg->max_sounds, cfg->playback_sample_rate);
    e->soundFloats.assign((size_t)cfg->max_sounds, 0);
    e->arenaFree.assign(1, { (size_t)0, e->arenaFloats & ~(size_t)3 });
    for (auto &c : e->slots) { std::memset(c.hReports, 0, V * sizeof(ZlReport)); std::memset(c.hStats, 0, sizeof(ZlBatchStats)); }
    e->latest = &e->slots[0];
    e->deviceBytes = g_alloc_bytes;
    *out = e;
    return ZLHIP_OK;
}

int zlhip_device_name(zlhip_engine *e, char *buf, size_t len)
{
    if (!e || !buf || !len) return ZLHIP_ERR_INVALID;
    std::snprintf(buf, len, "%s", e->devname);
    return ZLHIP_OK;
}

// ---- sounds / clips ---------------------------------------------------------------------------
void zlhip_clip_params_default(zlhip_clip_params *p, float duration_seconds)
{
    ZlHostControl::default_clip_params(p, duration_seconds);
}

static int alloc_sound_slot(zlhip_engine *e, int32_t length, int channels, double sample_rate, int32_t *out_id, float **dst)
{
    if (length < 1 || !(sample_rate > 0.0) || !out_id) return fail(e, ZLHIP_ERR_INVALID, "bad sound arguments");
    int id = -1;
    for (int i = 0; i < e->cfg.max_sounds; ++i) if (!e->hc.soundUsed[i]) { id = i; break; }
    if (id < 0) return fail(e, ZLHIP_ERR_CAPACITY, "sound table full");
    const size_t pad = 8;
    size_t floats = ((size_t)length + pad) * (size_t)channels;
    floats = (floats + 3) & ~(size_t)3;                           // keep every source 16-byte aligned
    // first fit over the free extents (offsets and sizes are multiples of 4 floats, so every source stays aligned)
    size_t off = (size_t)-1;
    for (size_t i = 0; i < e->arenaFree.size(); ++i) {
        if (e->arenaFree[i].second >= floats) {
            off = e->arenaFree[i].first;
            e->arenaFree[i].first += floats; e->arenaFree[i].second -= floats;
            if (e->arenaFree[i].second == 0) e->arenaFree.erase(e->arenaFree.begin() + (long)i);
            break;
        }
    }
    if (off == (size_t)-1) {
        // no extent holds it: one more arena segment, at least as large as the first one (and as the source)
        const size_t segFloats = (std::max(floats, e->arenaFloats) + 3) & ~(size_t)3;
        if (e->cfg.sound_arena_max_bytes > 0 && (e->arenaFloats + e->arenaSegmentFloats + segFloats) * sizeof(float) > e->cfg.sound_arena_max_bytes)
            return fail(e, ZLHIP_ERR_CAPACITY, "sound arena full");
        float *seg = nullptr;
        if (hipMalloc((void **)&seg, (segFloats + 1024) * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return fail(e, ZLHIP_ERR_CAPACITY, "sound arena full (no memory for another segment)"); }
        e->arenaSegmentFloats += segFloats;
        e->deviceBytes += (segFloats + 1024) * sizeof(float);
        // float offset of the segment from `arena`, modulo 2^64 (exact: both are multiples of 4 bytes)
        const int64_t diffBytes = (int64_t)((uintptr_t)seg - (uintptr_t)e->arena);
        off = (size_t)(uint64_t)(diffBytes / 4);
        e->arenaSegments.push_back({ seg, off, segFloats });
        if (segFloats > floats) {
            const std::pair<size_t, size_t> rest(off + floats, segFloats - floats);
            e->arenaFree.insert(std::lower_bound(e->arenaFree.begin(), e->arenaFree.end(), rest), rest);
        }
    }
    ZlSound s; s.offset = off; s.length = length; s.channels = channels; s.sample_rate = sample_rate;
    // (integer arithmetic: a segment below the first arena in the address space has an offset that wraps -- no pointer ever leaves its allocation)
    *dst = reinterpret_cast<float *>((uintptr_t)e->arena + (uintptr_t)off * sizeof(float));
    e->hc.sounds[id] = s;
    e->hc.soundUsed[id] = 1;
    e->soundFloats[(size_t)id] = floats;
    *out_id = id;
    return ZLHIP_OK;
}

// the arena extent of a sound slot goes back to the free list (coalesced with its neighbours); the slot is free again
static void free_sound_slot(zlhip_engine *e, int id)
{
    const size_t off = (size_t)e->hc.sounds[id].offset, n = e->soundFloats[(size_t)id];
    e->hc.soundUsed[id] = 0;
    e->hc.sounds[id] = ZlSound{0, 0, 0, 0.0};
    e->soundFloats[(size_t)id] = 0;
    if (n == 0) return;
    auto it = std::lower_bound(e->arenaFree.begin(), e->arenaFree.end(), std::make_pair(off, (size_t)0));
    it = e->arenaFree.insert(it, {off, n});
    if (it + 1 != e->arenaFree.end() && it->first + it->second == (it + 1)->first) { it->second += (it + 1)->second; e->arenaFree.erase(it + 1); }
    if (it != e->arenaFree.begin() && (it - 1)->first + (it - 1)->second == it->first) { (it - 1)->second += it->second; e->arenaFree.erase(it); --it; }
    // a later arena segment whose every float is free again goes back to the device (the first arena stays for the engine's life).  The
    // caller has waited for the engine and stopped its resident kernel; the free waits for the device: other engines' kernels step aside.
    for (size_t si = 0; si < e->arenaSegments.size(); ++si) {
        const auto seg = e->arenaSegments[si];
        if (!(it->first <= seg.off && seg.off + seg.floats <= it->first + it->second)) continue;
        const std::pair<size_t, size_t> whole = *it;
        e->arenaFree.erase(it);
        if (whole.first < seg.off) { const std::pair<size_t, size_t> head(whole.first, seg.off - whole.first); e->arenaFree.insert(std::lower_bound(e->arenaFree.begin(), e->arenaFree.end(), head), head); }
        if (seg.off + seg.floats < whole.first + whole.second) {
            const std::pair<size_t, size_t> tail(seg.off + seg.floats, whole.first + whole.second - (seg.off + seg.floats));
            e->arenaFree.insert(std::lower_bound(e->arenaFree.begin(), e->arenaFree.end(), tail), tail);
        }
        { ZlQuiesce quiet(e); (void)hipFree(seg.p); }
        e->arenaSegmentFloats -= seg.floats;
        e->deviceBytes -= (seg.floats + 1024) * sizeof(float);
        e->arenaSegments.erase(e->arenaSegments.begin() + (long)si);
        break;                                                     // (one extent, at most one whole segment: segments are separate allocations)
    }
}

static int publish_sound(zlhip_engine *e, int id)
{
    ZL_HIP(e, hipMemcpyAsync(e->dSounds + id, &e->hc.sounds[id], sizeof(ZlSound), hipMemcpyHostToDevice, e->stream));
    zlhip_clip_params p;
    zlhip_clip_params_default(&p, (float)(e->hc.sounds[id].length / e->hc.sounds[id].sample_rate));
    { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    e->hc.forget_clip_params(id);                                  // (the first edit of a slot carries the whole record)
    return zlhip_clip_set(e, id, &p);
}

static int sound_upload_device(zlhip_engine *e, const float *left_dev, const float *right_dev, int32_t length, double sample_rate,
                               bool have_producer, hipStream_t producer, int32_t *out_id);

int zlhip_sound_upload_device(zlhip_engine *e, const float *left_dev, const float *right_dev, int32_t length,
                              double sample_rate, int32_t *out_id)
{
    return sound_upload_device(e, left_dev, right_dev, length, sample_rate, false, nullptr, out_id);
}

int zlhip_sound_upload_device_on(zlhip_engine *e, const float *left_dev, const float *right_dev, int32_t length,
                                 double sample_rate, void *producer_stream, int32_t *out_id)
{
    return sound_upload_device(e, left_dev, right_dev, length, sample_rate, true, (hipStream_t)producer_stream, out_id);
}

static int sound_upload_device(zlhip_engine *e, const float *left_dev, const float *right_dev, int32_t length, double sample_rate,
                               bool have_producer, hipStream_t producer, int32_t *out_id)
{
    if (!e || !left_dev) return ZLHIP_ERR_INVALID;
    ZL_HIP(e, hipSetDevice(e->device));
    { int r_ = rt_stop(e); if (r_ != ZLHIP_OK) return r_; }         // (the resident real-time kernel does not see other engines' writes)
    float *dst = nullptr;
    // queued batches may still gather from an extent that was freed and is handed out again here
    if (e->outstanding) { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    int rc = alloc_sound_slot(e, length, right_dev ? 2 : 1, sample_rate, out_id, &dst);
    if (rc != ZLHIP_OK) return rc;
    hipError_t st;
    if (have_producer) {
        // the planes are complete when `producer` has run dry: an event there, waited for by the engine's stream -- no device-wide wait
        st = hipEventRecord(e->evJoin, producer);
        if (st == hipSuccess) st = hipStreamWaitEvent(e->stream, e->evJoin, 0);
    } else {
        // left_dev / right_dev were produced on a stream this call knows nothing about (the engine's stream is non-blocking: not
        // even the null stream orders it): wait for the whole device -- with every other engine's resident kernel asked to leave
        // for the duration (they would make this wait last until their idle timeout, or for ever)
        ZlQuiesce quiet(e);
        st = hipDeviceSynchronize();
    }
    int krc = st == hipSuccess ? zl_launch_interleave(left_dev, right_dev, dst, length, 8, e->stream) : (int)st;
    if (krc != 0) {
        free_sound_slot(e, *out_id); *out_id = -1;
        e->err = std::string("sound_upload_device: ") + hipGetErrorString((hipError_t)krc);
        return ZLHIP_ERR_HIP;
    }
    rc = publish_sound(e, *out_id);
    if (rc != ZLHIP_OK) { free_sound_slot(e, *out_id); *out_id = -1; }
    return rc;
}

int zlhip_sound_upload(zlhip_engine *e, const float *left, const float *right, int32_t length, double sample_rate, int32_t *out_id)
{
    if (!e || !left) return ZLHIP_ERR_INVALID;
    ZL_HIP(e, hipSetDevice(e->device));
    { int r_ = rt_stop(e); if (r_ != ZLHIP_OK) return r_; }         // (the resident real-time kernel does not see other engines' writes)
    float *dst = nullptr;
    const int ch = right ? 2 : 1;
    if (e->outstanding) { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }     // see zlhip_sound_upload_device
    int rc = alloc_sound_slot(e, length, ch, sample_rate, out_id, &dst);
    if (rc != ZLHIP_OK) return rc;
    std::vector<float> tmp(((size_t)length + 8) * ch, 0.0f);
    if (right) for (int32_t i = 0; i < length; ++i) { tmp[2 * (size_t)i] = left[i]; tmp[2 * (size_t)i + 1] = right[i]; }
    else std::memcpy(tmp.data(), left, (size_t)length * sizeof(float));
    hipError_t st = hipMemcpyAsync(dst, tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice, e->stream);
    rc = st == hipSuccess ? engine_wait(e) : ZLHIP_ERR_HIP;
    if (st != hipSuccess) e->err = std::string("sound_upload: ") + hipGetErrorString(st);
    if (rc == ZLHIP_OK) rc = publish_sound(e, *out_id);
    if (rc != ZLHIP_OK) { free_sound_slot(e, *out_id); *out_id = -1; }   // a failed upload keeps neither the slot nor its extent
    return rc;
}

int zlhip_sound_release(zlhip_engine *e, int32_t id)
{
    if (!e || id < 0 || id >= e->cfg.max_sounds || !e->hc.soundUsed[id]) return ZLHIP_ERR_INVALID;
    ZL_HIP(e, hipSetDevice(e->device));
    if (e->outstanding) { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }     // queued batches still read the table
    { int r_ = rt_stop(e); if (r_ != ZLHIP_OK) return r_; }         // (the resident real-time kernel does not see other engines' writes)
    free_sound_slot(e, id);
    ZL_HIP(e, hipMemcpyAsync(e->dSounds + id, &e->hc.sounds[id], sizeof(ZlSound), hipMemcpyHostToDevice, e->stream));
    { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    return ZLHIP_OK;
}

int zlhip_clip_set(zlhip_engine *e, int32_t id, const zlhip_clip_params *p)
{
    if (!e || !p || id < 0 || id >= e->cfg.max_sounds || !e->hc.soundUsed[id]) return ZLHIP_ERR_INVALID;
    if (p->num_slice_positions < 0 || p->num_slice_positions > ZLHIP_MAX_SLICES) return fail(e, ZLHIP_ERR_INVALID, "too many slices");
    // No HIP call, no wait: the edit is recorded on the host and the DEVICE applies it at the start of the next render call or
    // real-time cycle (K0 / every workgroup of the resident kernel, in stream order behind the planning of every call already
    // queued) -- the block boundary at which the reference's voices read the parameters (SamplerSynthVoice.cpp:189-196).  The
    // resident kernel stays resident; a JACK thread is never made to wait behind a parameter change.
    e->hc.set_clip_params(id, *p);
    return ZLHIP_OK;
}

// ---- commands ----------------------------------------------------------------------------------
void zlhip_clip_command_clear(zlhip_clip_command *c)
{
    std::memset(c, 0, sizeof *c);
    c->clip = -1; c->midi_note = -1; c->midi_channel = -1; c->slice = -1;
}

static int refresh_host_voices(zlhip_engine *e)
{
    if (e->outstanding) {
        ZL_HIP(e, hipSetDevice(e->device));
        int rc = engine_wait(e);
        if (rc != ZLHIP_OK) return rc;
    }
    if (e->reportsFresh) {
        e->hc.absorb_reports(e->latest->hReports);
        e->reportsFresh = false;
    }
    return ZLHIP_OK;
}

int zlhip_handle_command(zlhip_engine *e, const zlhip_clip_command *cmd, uint64_t current_tick)
{
    if (!e || !cmd) return ZLHIP_ERR_INVALID;
    int rc = refresh_host_voices(e);
    if (rc != ZLHIP_OK) return rc;
    return e->hc.handle_command(*cmd, current_tick);
}

int zlhip_handle_commands(zlhip_engine *e, const zlhip_clip_command *cmds, int32_t count, uint64_t current_tick, int32_t *taken)
{
    return zlhip_handle_commands_voices(e, cmds, count, current_tick, taken, nullptr);
}

int zlhip_handle_commands_voices(zlhip_engine *e, const zlhip_clip_command *cmds, int32_t count, uint64_t current_tick, int32_t *taken, int32_t *voices)
{
    if (!e || (!cmds && count > 0) || count < 0) return ZLHIP_ERR_INVALID;
    int rc = refresh_host_voices(e);                               // once for the whole batch
    if (rc != ZLHIP_OK) return rc;
    int n = 0;
    for (int32_t i = 0; i < count; ++i) {
        const int t = e->hc.handle_command(cmds[i], current_tick); // arrival order, as the channel's command ring
        if (taken) taken[i] = t;
        if (voices) voices[i] = e->hc.lastStartedVoice;
        n += t;
    }
    return n;
}

int zlhip_bus_set_enabled(zlhip_engine *e, int32_t bus, int enabled)
{
    if (!e || bus < 0 || bus >= e->cfg.num_buses) return ZLHIP_ERR_INVALID;
    int rc = refresh_host_voices(e);
    if (rc != ZLHIP_OK) return rc;
    return e->hc.set_bus_enabled(bus, enabled != 0) ? ZLHIP_OK : ZLHIP_ERR_INVALID;
}

int zlhip_start_voice(zlhip_engine *e, int32_t bus, int32_t slot, const zlhip_clip_command *cmd, uint64_t current_tick)
{
    if (!e || !cmd || bus < 0 || bus >= e->cfg.num_buses || slot < 0 || slot >= e->cfg.voices_per_bus) return ZLHIP_ERR_INVALID;
    int rc = refresh_host_voices(e);
    if (rc != ZLHIP_OK) return rc;
    return e->hc.handle_on_bus(bus, *cmd, current_tick, slot);
}

int zlhip_stop_voice(zlhip_engine *e, int32_t bus, int32_t slot, int allow_tail_off)
{
    if (!e || bus < 0 || bus >= e->cfg.num_buses || slot < 0 || slot >= e->cfg.voices_per_bus) return ZLHIP_ERR_INVALID;
    int rc = refresh_host_voices(e);
    if (rc != ZLHIP_OK) return rc;
    return e->hc.stop_voice(bus, slot, allow_tail_off != 0);
}

int zlhip_update_voice(zlhip_engine *e, int32_t bus, int32_t slot, const zlhip_clip_command *cmd)
{
    if (!e || !cmd || bus < 0 || bus >= e->cfg.num_buses || slot < 0 || slot >= e->cfg.voices_per_bus) return ZLHIP_ERR_INVALID;
    int rc = refresh_host_voices(e);
    if (rc != ZLHIP_OK) return rc;
    return e->hc.update_voice(bus, slot, *cmd);
}

int zlhip_voice_is_playing(zlhip_engine *e, int32_t bus, int32_t slot)
{
    if (!e || bus < 0 || bus >= e->cfg.num_buses || slot < 0 || slot >= e->cfg.voices_per_bus) return ZLHIP_ERR_INVALID;
    int rc = refresh_host_voices(e);
    if (rc != ZLHIP_OK) return rc;
    return e->hc.voices[(size_t)(bus * e->cfg.voices_per_bus + slot)].isPlaying ? 1 : 0;
}

// ---- render ------------------------------------------------------------------------------------
static int pick_group(const zlhip_engine *e, int K, int N)
{
    // voices summed sequentially by one wavefront.  0 = the whole bus, i.e. the reference's summation order
    // (SamplerSynth.cpp:136-140) whatever the batch shape; smaller groups add parallelism for short batches
    // of wide buses at the price of a (documented, deterministic) two-level order.
    (void)N;
    const int VPB = e->cfg.voices_per_bus;
    if (e->cfg.voices_per_task > 0) return std::min(e->cfg.voices_per_task, VPB);
    // A single real-time block of a wide bus has no other parallelism than its voices: one voice per workgroup, and
    // K3 adds the voices in voice order -- the SAME order as the whole-bus walk (0 + v0 + v1 + ...), bit for bit, in
    // half the time (1024 voices on 8 buses: 106 -> 54 us)
    if (K == 1 && VPB >= ZL_RT_SPLIT_MIN_VOICES) return 1;
    return VPB;
}

// The pending voice operations of this call -> the slot's mapped host buffers (the slot is idle: its previous call was
// waited for), sorted by voice.  K0 reads them over the bus in place: a real-time block with a thousand commands costs
// no copy command and no staging.
static int upload_ops(zlhip_engine *e, zlhip_engine::CallSlot &c, ZlBatch &A)
{
    A.n_op_ranges = 0; A.ops = nullptr; A.op_ranges = nullptr; A.n_clip_edits = 0; A.clip_edits = nullptr;
    const size_t ne = e->hc.pendingClipEdits.size();
    if (ne > 0) {
        if (ne > c.editsCap) { ZlQuiesce quiet(e); ZL_HIP(e, grow_mapped(&c.hEdits, &c.hEditsDev, &c.editsCap, ne)); }
        std::memcpy(c.hEdits, e->hc.pendingClipEdits.data(), ne * sizeof(ZlClipEdit));
        e->hc.pendingClipEdits.clear();
        A.n_clip_edits = (int)ne; A.clip_edits = c.hEditsDev;
    }
    const size_t n = e->hc.pendingOps.size();
    if (n == 0) return ZLHIP_OK;
    if (n > c.opsCap || n > c.rangesCap) {
        ZlQuiesce quiet(e);                                        // growing frees pinned memory, which waits for the device
        ZL_HIP(e, grow_mapped(&c.hOps, &c.hOpsDev, &c.opsCap, n));
        ZL_HIP(e, grow_mapped(&c.hRanges, &c.hRangesDev, &c.rangesCap, n));
    }
    e->hc.drain_ops_to(c.hOps, e->ranges);
    ZL_HIP(e, grow_mapped(&c.hRanges, &c.hRangesDev, &c.rangesCap, e->ranges.size()));
    std::memcpy(c.hRanges, e->ranges.data(), e->ranges.size() * sizeof(ZlOpRange));
    A.n_op_ranges = (int)e->ranges.size(); A.ops = c.hOpsDev; A.op_ranges = c.hRangesDev;
    return ZLHIP_OK;
}

// HIP-event timings of a finished call -> e->timings (the last call) and e->totals (sums)
static int harvest_slot(zlhip_engine *e, zlhip_engine::CallSlot &c)
{
    if (!c.profiled) return ZLHIP_OK;
    c.profiled = false;
    zlhip_timings t; std::memset(&t, 0, sizeof t);
    hipEvent_t evEnd = c.fusedDone ? c.done : c.evEnd;
    ZL_HIP(e, hipEventSynchronize(evEnd));
    ZL_HIP(e, hipEventElapsedTime(&t.total_ms, c.beginEv, evEnd));
    float k2 = 0.0f;
    for (int w = 0; w < c.windows; ++w) {
        float x = 0.0f;
        ZL_HIP(e, hipEventElapsedTime(&x, c.evK2[2 * (size_t)w], (c.fusedDone && w == c.windows - 1) ? c.done : c.evK2[2 * (size_t)w + 1]));
        k2 += x;
    }
    t.render_ms = k2;                                              // sum over the K2 launches of the call
    float first = 0.0f;
    ZL_HIP(e, hipEventElapsedTime(&first, c.beginEv, c.evK2[0]));
    t.plan_ms = first;                                             // planning that is NOT hidden behind rendering
    t.finalize_ms = t.total_ms - k2 - first;                       // K3 + reports + gaps between launches
    t.render_launches = c.windows;
    t.source_bytes = c.hStats->source_bytes; t.slow_blocks = c.hStats->slow_blocks; t.active_voice_frames = c.hStats->active_frames;
    e->timings = t;
    e->totals.plan_ms += t.plan_ms; e->totals.render_ms += t.render_ms; e->totals.finalize_ms += t.finalize_ms; e->totals.total_ms += t.total_ms;
    e->totals.render_launches += t.render_launches; e->totals.source_bytes += t.source_bytes; e->totals.slow_blocks += t.slow_blocks;
    e->totals.active_voice_frames += t.active_voice_frames;
    e->totalCalls += 1;
    return ZLHIP_OK;
}

static ZlPassParams pass_params(const zlhip_passthrough_params &p)
{
    return ZlPassParams{ p.dry_amount, p.wet_fx1_amount, p.wet_fx2_amount, p.pan_amount, p.muted };
}

// Offline bounce, one plan window: blocks [k0, k0 + K) of the chunk being rendered into busDev ([B][2][Ktot * N]) are final on
// stream s -- converted there to 16 bit if asked (a small kernel between two render kernels; on the copy stream it would get no
// compute units while the next window's render kernel fills the chip) and handed to the copy engine as ONE strided copy.
static int bounce_deliver_window(zlhip_engine *e, const float *busDev, int b0, int b1, int k0, int K, int Ktot, int nframes, hipStream_t s)
{
    zlhip_engine::Bounce &q = e->bnc;
    const size_t rowFrames = (size_t)Ktot * (size_t)nframes, off = (size_t)k0 * (size_t)nframes, frames = (size_t)K * (size_t)nframes;
    const size_t nb = (size_t)(b1 - b0), total = q.sink.totalFrames;
    if (nb == 0) return ZLHIP_OK;
    if (q.sink.pcm) ZL_KERNEL(e, zl_launch_deliver(busDev + (size_t)b0 * 2 * rowFrames, q.sink.pcmDev + (size_t)b0 * rowFrames * 2, 1, (int)nb,
                                                   (long long)rowFrames, (long long)off, (long long)frames, (long long)rowFrames, s));
    hipEvent_t ev = q.winEv[q.winNext++ % zlhip_engine::Bounce::NEV];
    ZL_HIP(e, hipEventRecord(ev, s));
    ZL_HIP(e, hipStreamWaitEvent(q.copyStream, ev, 0));
    // rows of 4 bytes per frame in both formats: [B][total][2] 16-bit = one row per bus, [B][2][total] fp32 = one per bus channel
    const size_t rows = q.sink.pcm ? nb : nb * 2, row0 = q.sink.pcm ? (size_t)b0 : (size_t)b0 * 2;
    const char *src = q.sink.pcm ? (const char *)q.sink.pcmDev : (const char *)busDev;
    ZL_HIP(e, hipMemcpy2DAsync(q.sink.hostBase + (row0 * total + off) * 4, total * 4, src + (row0 * rowFrames + off) * 4, rowFrames * 4, frames * 4, rows,
                               hipMemcpyDeviceToHost, q.copyStream));
    return ZLHIP_OK;
}

int zlhip_render_batch(zlhip_engine *e, int32_t nblocks, int32_t nframes, const zlhip_clock *clocks, float *bus_out_dev, void *stream)
{
    return zlhip_render_batch_fanout(e, nblocks, nframes, clocks, bus_out_dev, nullptr, nullptr, stream);
}

static int render_batch_impl(zlhip_engine *e, int32_t nblocks, int32_t nframes, const zlhip_clock *clocks, float *bus_out_dev, long long bus_stride,
                             long long ch_stride, const zlhip_passthrough_params *fan_params, float *fan_out_dev, void *stream);

int zlhip_render_batch_fanout(zlhip_engine *e, int32_t nblocks, int32_t nframes, const zlhip_clock *clocks, float *bus_out_dev,
                              const zlhip_passthrough_params *fan_params, float *fan_out_dev, void *stream)
{
    return render_batch_impl(e, nblocks, nframes, clocks, bus_out_dev, 0, 0, fan_params, fan_out_dev, stream);
}

// bus_stride / ch_stride: 0 = the bus buffer's own layout [B][2][nblocks * nframes]; else the rows of a single real-time block at the
// caller's strides (zlhip_render_fanout delivering straight into page-locked out_left / out_right)
static int render_batch_impl(zlhip_engine *e, int32_t nblocks, int32_t nframes, const zlhip_clock *clocks, float *bus_out_dev, long long bus_stride,
                             long long ch_stride, const zlhip_passthrough_params *fan_params, float *fan_out_dev, void *stream)
{
    if (!e || !clocks || ((fan_params == nullptr) != (fan_out_dev == nullptr))) return ZLHIP_ERR_INVALID;
    if (nblocks < 1 || nblocks > e->cfg.max_batch_blocks) return fail(e, ZLHIP_ERR_CAPACITY, "nblocks exceeds max_batch_blocks");
    if (nframes < 1 || nframes > e->cfg.max_frames)
        return fail(e, ZLHIP_ERR_INVALID, "nframes must be in [1, max_frames]");
    if (e->failed) return fail(e, ZLHIP_ERR_STATE, "engine failed (the resident kernel stopped answering in the middle of a cycle): destroy it");
    ZL_HIP(e, hipSetDevice(e->device));
    { int r_ = rt_stop(e); if (r_ != ZLHIP_OK) return r_; }         // a batch shares the voice table and the plan records with the resident kernel
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;

    // this call's slot: wait for the call that used it two calls ago (the previous call may still be rendering)
    static const bool callStamps = std::getenv("ZL_CALL_STAMPS") != nullptr;   // diagnostics: host time of the call's phases (stderr)
    const auto tEnter = std::chrono::steady_clock::now();
    zlhip_engine::CallSlot &c = e->slots[e->callIndex & 1u];
    if (c.inflight) {
        ZL_HIP(e, hipEventSynchronize(c.done));
        c.inflight = false;
        int hrc = harvest_slot(e, c);
        if (hrc != ZLHIP_OK) return hrc;
    }
    c.doneGen ^= 1u; c.done = c.doneEv[c.doneGen];                  // (the event of this slot's previous use stays intact for the other slot's harvest)
    const auto tSlot = std::chrono::steady_clock::now();
    bool regular = true;                                           // monotone time, one period: lets K1 bisect for loop restarts
    for (int k = 0; k < nblocks; ++k) {
        ZlHostControl::fill_clock(c.hClocks[k], clocks[k], nframes);
        if (c.hClocks[k].usecs_per_frame >= (1ull << 21) || c.hClocks[k].usecs_per_frame != c.hClocks[0].usecs_per_frame
            || (k > 0 && c.hClocks[k].current_usecs < c.hClocks[k - 1].current_usecs)) regular = false;
    }

    ZlBatch A; std::memset(&A, 0, sizeof A);
    A.V = e->V; A.B = e->cfg.num_buses; A.VPB = e->cfg.voices_per_bus; A.N = nframes; A.Ktot = nblocks;
    A.G = pick_group(e, nblocks, nframes);
    A.groups = (A.VPB + A.G - 1) / A.G;
    // narrow buses in batches: several whole buses per K2 workgroup (voices of consecutive buses are contiguous); needs the
    // bus width to be a multiple of K2's chunk of 8 voices, no mix groups, one frame tile per block
    A.NB = 1;
    if (nblocks > 1 && A.groups == 1 && A.VPB <= 64 && (A.VPB % 8) == 0 && nframes <= 256)
        A.NB = std::max(1, std::min(128 / A.VPB, A.B));
    A.clocks_regular = regular ? 1 : 0;
    A.mode = e->cfg.mode;
    A.staged = (e->staged && !e->trace && nframes % 64 == 0) ? 1 : 0;   // (the position trace lives in the gather paths; a staged wave is a whole 64-frame tile)
    A.sounds = e->dSounds; A.clips = e->dClips; A.arena = e->arena;
    A.voices = e->dVoices; A.reports = c.dReports; A.pass_cache = e->dPassCache;
    A.bus = bus_out_dev ? bus_out_dev : e->dBus; A.stats = c.dStats;
    A.bus_stride = bus_stride; A.ch_stride = ch_stride;
    A.trace = 0; A.pos_trace = nullptr;
    int32_t *traceBase = nullptr;
    if (e->trace) {
        const size_t need = (size_t)nblocks * e->V * nframes;
        if (need > e->traceInts) {
            ZlQuiesce quiet(e);
            if (e->dTrace) ZL_HIP(e, hipFree(e->dTrace));
            ZL_HIP(e, dalloc(&e->dTrace, need));
            e->traceInts = need;
        }
        ZL_HIP(e, hipMemsetAsync(e->dTrace, 0xff, need * sizeof(int32_t), s));
        A.trace = 1; traceBase = e->dTrace; e->traceK = nblocks; e->traceN = nframes;
#ifdef ZL_STAMPS
        A.trace = 0;       // diagnostic build: the trace buffer receives per-workgroup timestamps instead
#endif
    }

    // ---- plan windows: K0/K1/K1c of window i+1 run on the planning stream while K2/K3 of window i render ----
    // Window layout: large windows keep K2 launches long (their ramp-up and drain are a fixed cost per launch), but
    // planning window i+1 must fit behind rendering window i, and the planning of the first window is hidden by
    // nothing but the previous call: windows start at 256 blocks and double up to the configured size.
    // (ZL_WINDOW_MUL=4: blocks longer than 64 frames may fill the record arrays -- up to four windows' worth of frames in one K2
    // launch.  Measured: K2 gains 4..9 % (fewer launch ramps; a bus's sources are re-read inside ONE launch, from the Infinity Cache),
    // the headline call 2..3.5 %, but a call is then a single window whose planning is hidden by nothing but the previous call's
    // launch: pitched voices -3 %, 128-frame blocks -5 %, 4096 voices -8 %, and one host-side stall of several milliseconds per
    // timed region.  The default stays the fixed number of frames per window.)
    static const int windowMulEnv = [] { const char *v = std::getenv("ZL_WINDOW_MUL"); const int m = v ? std::atoi(v) : 0; return m < 0 ? 0 : (m > 64 ? 64 : m); }();
    // Round 4: the choice is made per call from what planning will cost.  When every playing voice runs at exactly the playback rate on
    // a sample-space loop (libzl's common case: a clip at its own pitch and rate; the BASELINE workload) K1 plans a window of any length
    // in a handful of exact runs, so the window may be four times as long: one K2 launch per 8192-block call instead of four.  Pitched,
    // resampled or beat-locked voices keep the pipeline of shorter windows, whose planning hides behind rendering.  ZL_WINDOW_MUL=<n> forces n.
    // (measured on one box, alternating runs, profiles/round4_window_ab.txt: headline +2 %, 4096 voices at 96 kHz +5.5 %, Hermite at ratio 1 +5.5 %,
    // narrow buses +0.5..1 %; two 128-frame blocks per workgroup -3 % -- short blocks keep the fixed size)
    int windowMul = windowMulEnv;
    if (windowMul == 0) {
        bool cheap = nframes >= 256;
        for (const ZlHostVoice &hv : e->hc.voices) if (hv.isPlaying && !hv.cheapPlan) { cheap = false; break; }
        windowMul = cheap ? 4 : 1;
    }
    // (engines that split buses into mix groups keep the fixed size: their partial rows are sized for it)
    const size_t mul = e->maxGroups > 1 ? 1 : (size_t)windowMul;
    int W = e->windowBlocks > 0 ? e->windowBlocks : (int)std::max<size_t>(1, std::min<size_t>(mul * e->windowFrames / (size_t)nframes, (size_t)1 << 30));
    W = std::min(W, e->windowCap);
    W = std::min(W, (1 << 30) / nframes);                          // window time is a 32-bit frame index in K1 / K1c
    W = std::min(W, 60000);                                        // a K2 launch has one y slot per block (+ 1920 for a split tail): gridDim.y stays below 65536
    std::vector<std::pair<int, int>> &wins = e->wins;              // (first block, blocks); member: no allocation per call
    wins.clear();
    // when the previous call is still in flight its rendering hides the planning of this call's first window: no
    // need to start small (fewer, longer K2 launches)
    zlhip_engine::CallSlot &prev = e->slots[(e->callIndex + 1u) & 1u];
    const bool behindPrev = prev.inflight && hipEventQuery(prev.done) == hipErrorNotReady;
    if (nblocks <= W || e->ps[1].hdr == nullptr || behindPrev) {
        for (int k0 = 0; k0 < nblocks; k0 += W) wins.push_back({k0, std::min(W, nblocks - k0)});
    } else {
        // nothing hides the planning of this call's first window: a quarter-size window first (its planning is short,
        // and its rendering is long enough to hide the planning of a full window), then full windows.  (Doubling from
        // 64 Ki frames cost three small, inefficient K2 launches: +280 us per such call against +110 us.)
        const char *fw = std::getenv("ZL_FIRST_WINDOW_FRAMES");
        int size = std::min(W, std::max(1, (fw ? std::atoi(fw) : (int)std::min<size_t>(e->windowFrames / 4, (size_t)1 << 28)) / nframes));
        for (int k0 = 0; k0 < nblocks;) {
            const int n = std::min(size, nblocks - k0);
            wins.push_back({k0, n});
            k0 += n;
            size = W;
        }
    }
    const int nwin = (int)wins.size();
    // (a bounce queues its sub-batches back to back: planning on the planning stream from the first one on orders the second
    // sub-batch's planning behind the first one's PLANNING, not behind its rendering)
    const bool overlap = e->ps[1].hdr != nullptr && (nwin > 1 || behindPrev || e->bnc.active);
    hipStream_t ps = overlap ? e->planStream : s;
    // The call's inputs (clocks, voice operations, cleared statistics) go to the planning stream itself: it is in order
    // with the planning of the previous call, so window 0 of this call is planned while the previous call still renders.
    // the voice state is carried from call to call by K1: when this call plans on another stream than the previous one
    // did, order it behind that call's last planning kernel
    if (e->lastPlanStream && e->lastPlanStream != ps && e->lastPlanEvent) ZL_HIP(e, hipStreamWaitEvent(ps, e->lastPlanEvent, 0));
    // The block clocks stay where fill_clock wrote them, in host memory mapped into the device (like the voice operations): the planner
    // reads them at a voice's first block, in simulated blocks, and stage by stage for beat-locked loops.  (A copy command here blocked
    // the host for 6-7 ms every now and then -- hipMemcpyAsync on the planning stream, seen at the fifth call of a run.)
    if (nblocks == 1) { A.inline_clock = 1; A.clock0 = c.hClocks[0]; A.fuse_assemble = 1; }   // a real-time block: the clock travels with the kernel arguments
    if (fan_out_dev) {                                             // the rendering stream is ordered behind ps by the window events
        for (int b = 0; b < A.B; ++b) c.hPass[b] = pass_params(fan_params[b]);
        ZL_HIP(e, hipMemcpyAsync(c.dPass, c.hPass, (size_t)A.B * sizeof(ZlPassParams), hipMemcpyHostToDevice, ps));
        A.pass = c.dPass; A.fan = fan_out_dev;
    }
    int rc = upload_ops(e, c, A);
    if (rc != ZLHIP_OK) return rc;
    // (the slot's statistics were cleared by the report kernel of the call that used it before)
    if (e->profiling) {
        // the profile of a call queued behind a profiled call on the same stream counts from that call's completion event
        if (behindPrev && prev.profiled && e->lastRenderStream == s) c.beginEv = prev.done;
        else { ZL_HIP(e, hipEventRecord(c.evBegin, s)); c.beginEv = c.evBegin; }
    }
    e->lastRenderStream = s;
    // the record sets alternate across calls too, so that the first window of this call is not planned into the set
    // the previous call's last window still renders from
    const unsigned phase = e->ps[1].hdr != nullptr ? e->setPhase : 0u;
    // The call's reports come from K2 itself wherever one workgroup sums a whole bus of the last block (no mix groups, one frame tile):
    // the workgroups of that block publish gains, reports and statistics, and the launch carries the call's completion event -- between
    // the last render kernel of this call and the first of the next sits no report kernel any more (12 % of a 64-voice step were packets).
    // (Not while a bounce hands windows to the copy engine: its conversion kernels and copies follow the last render kernel.)
    const bool fusedReports = A.groups == 1 && nframes <= 256 && !(e->bnc.sink.on && !(e->bnc.sink.hostDev && A.groups == 1));
    c.fusedDone = fusedReports;
    for (int w = 0; w < nwin; ++w) {
        zlhip_engine::PlanSet &q = e->ps[(phase + (unsigned)w) & 1u];
        ZlBatch Aw = A;
        Aw.k0 = wins[(size_t)w].first;
        Aw.K = wins[(size_t)w].second;
        Aw.clocks = c.hClocksDev + Aw.k0;
        Aw.levels = e->dLevels + (size_t)Aw.k0 * A.B;
        Aw.pos_trace = traceBase ? traceBase + (size_t)Aw.k0 * e->V * nframes : nullptr;
        Aw.vconst = q.vconst; Aw.runs = q.runs; Aw.tsegs = q.tsegs; Aw.plan_hdr = q.hdr; Aw.plan_seg0 = q.seg0; Aw.plan_seg1 = q.seg1;
        Aw.ctl_P = q.ctlP; Aw.ctl_env = q.ctlEnv; Aw.partials = q.partials;
        Aw.ctl_next = q.ctlNext; Aw.sim_const = q.simConst;
        Aw.ctl_slots = e->ctlSlotsOverride >= 0 ? std::min<int>(e->ctlSlotsOverride, (int)(e->ctlPoolFrames / (size_t)nframes)) : (int)std::min<size_t>(e->ctlPoolFrames / (size_t)nframes, 0x7fffffff);
        // every (block, voice) of a window asks for at most one slot: the next window of this set counts from past that
        Aw.ctl_base = q.ctlBase;
        q.ctlBase += (unsigned long long)Aw.K * (unsigned long long)e->V + 1ull;
        if (w > 0) { Aw.n_op_ranges = 0; Aw.ops = nullptr; Aw.op_ranges = nullptr; Aw.n_clip_edits = 0; Aw.clip_edits = nullptr; }   // commands and parameter edits apply before the first block only
        // planning may not overwrite a record set while an earlier window (of this or the previous call) still renders from it
        if (ps != s && q.used) ZL_HIP(e, hipStreamWaitEvent(ps, q.renderedEv, 0));
        ZL_KERNEL(e, zl_launch_apply_ops(Aw, ps));
        ZL_KERNEL(e, zl_launch_plan(Aw, e->forceSlow, ps));
        if (ps != s && e->asmStream && !Aw.fuse_assemble) {
            // K1c (lane-parallel) on its own stream: it runs next to K1 of the following window, which writes the other set
            ZL_HIP(e, hipEventRecord(q.k1done, ps));
            ZL_HIP(e, hipStreamWaitEvent(e->asmStream, q.k1done, 0));
            ZL_KERNEL(e, zl_launch_assemble(Aw, e->asmStream));
            ZL_HIP(e, hipEventRecord(q.planned, e->asmStream));
            ZL_HIP(e, hipStreamWaitEvent(s, q.planned, 0));
        } else {
            if (!Aw.fuse_assemble) ZL_KERNEL(e, zl_launch_assemble(Aw, ps));
            if (ps != s) {
                ZL_HIP(e, hipEventRecord(q.planned, ps));
                ZL_HIP(e, hipStreamWaitEvent(s, q.planned, 0));
            }
        }
        // K2 scans the block for AudioLevels itself when one workgroup holds the whole block of the final mix
        const bool k3 = !(Aw.groups == 1 && nframes <= 256);
        // offline bounce, direct delivery: K2 itself also stores the finished bus into the caller's page-locked host buffer
        const bool direct = e->bnc.sink.on && e->bnc.sink.hostDev && Aw.groups == 1;
        if (direct) { Aw.host_out = e->bnc.sink.hostDev; Aw.host_fmt = e->bnc.sink.pcm ? 1 : 0; Aw.host_total = (long long)e->bnc.sink.totalFrames; Aw.host_k0 = 0; }
        // profiling: the K2 dispatch carries its own start / stop events (hipExtLaunchKernel); the call's last launch carries `done`
        const bool closes = fusedReports && w == nwin - 1;
        if (fusedReports) { Aw.fused_reports = 1; Aw.rep_gain = e->dGain; Aw.rep_host = c.hReportsDev; Aw.rep_host_gain = c.hGainDev; Aw.rep_host_stats = c.hStatsDev; }
        hipEvent_t k2stop = closes ? c.done : (e->profiling ? c.evK2[2 * (size_t)w + 1] : nullptr);
        ZL_KERNEL(e, zl_launch_render(Aw, s, e->profiling ? c.evK2[2 * (size_t)w] : nullptr, k2stop));
        if (k3) ZL_KERNEL(e, zl_launch_finalize(Aw, nullptr, s));
        // ... or through the copy engine, window by window: the window's columns of the bus are final now
        if (e->bnc.sink.on && !direct) { int d_ = bounce_deliver_window(e, A.bus, 0, Aw.B, Aw.k0, Aw.K, nblocks, nframes, s); if (d_ != ZLHIP_OK) return d_; }
        // (every event record is a packet the command processor handles between two K2 launches: when profiling, the
        // event that closes the K2 timing doubles as the set's "rendered" event)
        // (an engine with a single record set never plans on another stream: nobody waits for "rendered")
        if (e->ps[1].hdr != nullptr) {
            if (k2stop && !k3) q.renderedEv = k2stop;
            else { ZL_HIP(e, hipEventRecord(q.rendered, s)); q.renderedEv = q.rendered; }
            q.used = true;
        }
    }
    c.windows = nwin;
    // the end of this call's planning, for a next call that plans on another stream: an event on the planning stream, or
    // -- when the call planned on its render stream -- simply its `done` event (recorded below; one packet fewer)
    if (ps != s) { ZL_HIP(e, hipEventRecord(e->evPlanTail, ps)); e->lastPlanEvent = e->evPlanTail; }
    else e->lastPlanEvent = c.done;
    e->lastPlanStream = ps;
    if (e->ps[1].hdr != nullptr) e->setPhase = (phase + (unsigned)nwin) & 1u;
    // results go straight to mapped host memory (a copy command here would make the runtime wait for the stream)
    // (the report kernel is the call's last packet: its stop event is the call's completion event)
    if (fusedReports) {
        if (e->profiling) c.profiled = true;                      // (total = evBegin .. done, the last K2 launch's stop event)
    } else {
        ZL_KERNEL(e, zl_launch_reports(c.dReports, e->V, e->dGain, c.hReportsDev, c.hGainDev, c.dStats, c.hStatsDev, s, e->profiling ? nullptr : c.done));
        if (e->profiling) { ZL_HIP(e, hipEventRecord(c.evEnd, s)); c.profiled = true; ZL_HIP(e, hipEventRecord(c.done, s)); }
    }
    c.inflight = true;
    if (s != e->stream) e->joins[0] = c.done;                      // later engine work (levels, read-back) waits for it on the host
    e->latest = &c;
    e->callIndex += 1;
    e->lastK = nblocks; e->lastN = nframes; e->lastBus = bus_stride ? nullptr : A.bus; e->lastWindows = nwin;
    e->outstanding = true; e->reportsFresh = true;
    if (callStamps) {
        const auto tEnd = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        std::fprintf(stderr, "zlhip_render_batch #%u: slot wait %.0f us, commands %.0f us, %d window(s)\n", e->callIndex - 1, us(tEnter, tSlot), us(tSlot, tEnd), nwin);
    }
    return ZLHIP_OK;
}

// ---- offline bounce ---------------------------------------------------------------------------------
int zlhip_host_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0) return ZLHIP_ERR_INVALID;
    *out = nullptr;
    return hipHostMalloc(out, bytes) == hipSuccess ? ZLHIP_OK : ZLHIP_ERR_CAPACITY;
}

void zlhip_host_free(void *p) { if (p) { ZlQuiesce quiet(nullptr); (void)hipHostFree(p); } }

static int bounce_body(zlhip_engine *e, int64_t nblocks, int32_t nframes, const zlhip_clock *clocks, void *host_out, bool pcm, int64_t chunk)
{
    constexpr int NBUF = zlhip_engine::Bounce::NBUF;
    zlhip_engine::Bounce &q = e->bnc;
    const size_t total = (size_t)nblocks * (size_t)nframes;          // frames per bus channel in host_out
    static const bool stamps = std::getenv("ZL_BOUNCE_STAMPS") != nullptr;     // diagnostics: host time of every step (stderr)
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t00 = stamps ? now_us() : 0.0;
    // Direct delivery: the render kernel itself stores the finished bus into the caller's buffer over PCIe, in the requested format
    // (page-locked memory only: it needs a device view).  Measured on the per-GPU share of BASELINE configs[4] (profiles/round3_bounce.txt):
    // 16 bit 4.6 ms = 98 % of the device-resident rate (27 GB/s of stores next to the rendering: half the link) against 5.7 ms through the
    // copy engine; fp32 6.05 ms (the stores run at 40 GB/s) against 7.0 ms.  ZL_BOUNCE_DIRECT: 0 = always through the copy engine, window
    // by window; 1 = only the 16-bit format directly; 2 (default) = both.
    const int directMode = [] { const char *v = std::getenv("ZL_BOUNCE_DIRECT"); return v ? std::atoi(v) : 2; }();   // (read per bounce: tests switch it)
    void *hostDev = nullptr;
    if (directMode >= (pcm ? 1 : 2)) {
        if (hipHostGetDevicePointer(&hostDev, host_out, 0) != hipSuccess) { hostDev = nullptr; (void)hipGetLastError(); }   // pageable memory: copies
    }
    int j = 0;
    for (int64_t k0 = 0; k0 < nblocks; k0 += chunk, ++j) {
        const int nb = (int)std::min<int64_t>(chunk, nblocks - k0);
        const int i = j % NBUF;
        // the buffer is free again when the chunk rendered into it two chunks ago has left it
        if (j >= NBUF) ZL_HIP(e, hipStreamWaitEvent(e->stream, q.copied[i], 0));
        const double t0 = stamps ? now_us() : 0.0;
        q.sink.on = true; q.sink.pcm = pcm; q.sink.pcmDev = q.pcm[i]; q.sink.totalFrames = total;
        q.sink.hostBase = (char *)host_out + (size_t)k0 * (size_t)nframes * 4;      // 4 bytes per frame and row in both formats
        q.sink.hostDev = hostDev ? (void *)((char *)hostDev + (size_t)k0 * (size_t)nframes * 4) : nullptr;
        const int rc = zlhip_render_batch_fanout(e, nb, nframes, clocks + k0, q.bus[i], nullptr, nullptr, nullptr);
        q.sink.on = false;
        if (rc != ZLHIP_OK) return rc;
        ZL_HIP(e, hipEventRecord(q.copied[i], q.copyStream));
        if (stamps) std::fprintf(stderr, "zlhip_bounce chunk %d: at %.0f us, render + delivery commands %.0f us (%d windows)\n", j, t0 - t00, now_us() - t0, e->lastWindows);
    }
    const double t2 = stamps ? now_us() : 0.0;
    ZL_HIP(e, hipStreamSynchronize(q.copyStream));
    if (stamps) std::fprintf(stderr, "zlhip_bounce: commands done at %.0f us, delivered at %.0f us\n", t2 - t00, now_us() - t00);
    return ZLHIP_OK;
}

int zlhip_bounce(zlhip_engine *e, int64_t nblocks, int32_t nframes, const zlhip_clock *clocks, void *host_out, int32_t format, int32_t sub_blocks)
{
    if (!e || !clocks || !host_out || nblocks < 1 || !(format == ZLHIP_BOUNCE_F32_PLANAR || format == ZLHIP_BOUNCE_PCM16_STEREO)) return ZLHIP_ERR_INVALID;
    if (nframes < 1 || nframes > e->cfg.max_frames)
        return fail(e, ZLHIP_ERR_INVALID, "nframes must be in [1, max_frames]");
    ZL_HIP(e, hipSetDevice(e->device));
    constexpr int NBUF = zlhip_engine::Bounce::NBUF;
    const int B = e->cfg.num_buses;
    // chunks = the render calls of the bounce: as long as the engine takes (max_batch_blocks), so that its plan windows pipeline
    // as in a device-resident batch and a loop's sources are re-read from the Infinity Cache inside a window; the DELIVERY is per
    // plan window (bounce_deliver_window), so the first bytes cross PCIe after the first window, not after the first chunk
    int64_t chunk = sub_blocks > 0 ? sub_blocks : e->cfg.max_batch_blocks;
    chunk = std::min<int64_t>(std::min<int64_t>(chunk, e->cfg.max_batch_blocks), nblocks);
    const bool pcm = format == ZLHIP_BOUNCE_PCM16_STEREO;
    zlhip_engine::Bounce &q = e->bnc;
    if (!q.copyStream) {
        ZL_HIP(e, hipStreamCreateWithFlags(&q.copyStream, hipStreamNonBlocking));
        for (int i = 0; i < NBUF; ++i) ZL_HIP(e, hipEventCreateWithFlags(&q.copied[i], hipEventDisableTiming));
        for (hipEvent_t &ev : q.winEv) ZL_HIP(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const int nbuf = nblocks > chunk ? NBUF : 1;
    const size_t needFloats = (size_t)B * 2 * (size_t)chunk * (size_t)nframes, needFrames = (size_t)B * (size_t)chunk * (size_t)nframes;
    if (q.busFloats < needFloats || (pcm && q.pcmFrames < needFrames) || (nbuf > 1 && (!q.bus[1] || (pcm && !q.pcm[1])))) {
        { int r_ = rt_stop(e); if (r_ != ZLHIP_OK) return r_; }
        { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
        ZL_HIP(e, hipStreamSynchronize(q.copyStream));
        ZlQuiesce quiet(e);                                        // hipFree waits for the device
        const size_t nf = std::max(q.busFloats, needFloats), np = pcm ? std::max(q.pcmFrames, needFrames) : q.pcmFrames;
        for (int i = 0; i < nbuf; ++i) {
            if (!q.bus[i] || q.busFloats < nf) {
                if (q.bus[i]) { ZL_HIP(e, hipFree(q.bus[i])); q.bus[i] = nullptr; }
                ZL_HIP(e, hipMalloc((void **)&q.bus[i], nf * sizeof(float)));
            }
            if (pcm && (!q.pcm[i] || q.pcmFrames < np)) {
                if (q.pcm[i]) { ZL_HIP(e, hipFree(q.pcm[i])); q.pcm[i] = nullptr; }
                ZL_HIP(e, hipMalloc((void **)&q.pcm[i], np * 2 * sizeof(int16_t)));
            }
        }
        // (a second buffer allocated earlier at a smaller size is dropped: it is allocated again when a bounce needs it)
        for (int i = nbuf; i < NBUF; ++i) {
            if (q.bus[i] && q.busFloats < nf) { ZL_HIP(e, hipFree(q.bus[i])); q.bus[i] = nullptr; }
            if (q.pcm[i] && q.pcmFrames < np) { ZL_HIP(e, hipFree(q.pcm[i])); q.pcm[i] = nullptr; }
        }
        q.busFloats = nf; q.pcmFrames = np;
    }
    q.active = true;
    const int rc = bounce_body(e, nblocks, nframes, clocks, host_out, pcm, chunk);
    q.active = false; q.sink.on = false;
    // (also after an error: nothing of this call may still be writing the caller's buffer when it returns)
    const hipError_t cs = hipStreamSynchronize(q.copyStream);
    const int w = engine_wait(e);
    e->outstanding = false;
    if (rc != ZLHIP_OK) return rc;
    if (cs != hipSuccess) { e->err = std::string("bounce copy stream: ") + hipGetErrorString(cs); return ZLHIP_ERR_HIP; }
    return w;
}

int zlhip_synchronize(zlhip_engine *e)
{
    if (!e) return ZLHIP_ERR_INVALID;
    ZL_HIP(e, hipSetDevice(e->device));
    { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    e->outstanding = false;
    return ZLHIP_OK;
}

// wide buses: one resident workgroup per voice (the per-voice split of pick_group, inside the residency)
static bool rt_wide(const zlhip_engine *e) { return e->cfg.voices_per_bus >= ZL_RT_SPLIT_MIN_VOICES; }

static bool rt_eligible(zlhip_engine *e, int nframes)
{
    // the workgroups of one cycle are all resident: buses summed whole by one workgroup each (narrow) or one workgroup per voice and
    // the ordered sum by the bus's last arrival (wide); the reference's summation order either way (no mix groups), any period (a
    // workgroup walks the 256-frame tiles of a longer block one after the other), no debug trace
    if (!(e->rt.enabled && nframes <= e->rt.maxFrames && e->cfg.voices_per_task <= 0 && !e->trace)) return false;
    const bool wide = rt_wide(e);
    if (wide && e->rt.wide == 0) return false;
    static_assert(ZL_RT_DONE_SLOTS >= 64, "one completion word per resident workgroup of a narrow engine");
    if (e->cfg.num_buses > (wide ? ZL_RT_MAX_BUSES : 64)) return false;
    int &cap = e->rt.capacity[wide ? 1 : 0];
    if (cap < 0) cap = zl_rt_loop_capacity(e->cfg.mode, wide ? 1 : 0, 256, e->device);
    if (cap <= 0) return false;
    // (room is left for the level-tick and upload kernels of the same process: three quarters of the device at most -- for ALL the
    // resident kernels of the process together: rt_start adds up the shares of the engines that are resident, rt_fits below)
    if (!wide) {
        e->rt.share = (double)e->cfg.num_buses / (double)cap;
        return (long long)e->cfg.num_buses * 4 <= (long long)cap * 3;
    }
    // wide: as few voices per workgroup as the device holds (their K2 bodies run one after the other), never across a bus
    if (e->rt.vw == 0) {
        e->rt.vw = -1;
        for (int vw = 1; vw <= 8; vw *= 2)
            if (e->cfg.voices_per_bus % vw == 0 && (long long)(e->V / vw) * 4 <= (long long)cap * 3) { e->rt.vw = vw; break; }
    }
    if (e->rt.vw > 0) e->rt.share = (double)(e->V / e->rt.vw) / (double)cap;
    return e->rt.vw > 0 && (e->rt.wide == 1 || e->rt.vw == 1);     // (several voices per workgroup: opt-in, see zlhip_engine_create)
}

#define ZL_RT_BUSY 1     // rt_start / rt_render: a device-synchronising call is in progress somewhere in the process -- render this cycle with launches
#define ZL_RT_NOFIT 2    // rt_start: the resident kernels of the process's other engines leave no room on the device for this one -- render with launches

// Every workgroup of a resident kernel must be ON the device for a cycle to complete (the last arrival reports it): two engines
// that each fit alone may not fit together -- the second kernel's workgroups would wait for slots the first one's spinners hold, and
// its host would give up after two seconds.  So the shares of all engines whose kernel is (or is about to be) resident on this
// device are added up, and an engine that does not fit next to them renders with launches (same results).  Call with g_rt.mu held.
static bool rt_fits(const zlhip_engine *e)
{
    double used = 0.0;
    for (const zlhip_engine *o : g_rt.engines) {
        if (o == e || o->device != e->device || !o->rt.h) continue;
        if (__atomic_load_n(&o->rt.h->state, __ATOMIC_ACQUIRE) != 2u || o->rt.inCycle.load(std::memory_order_acquire)) used += o->rt.share;
    }
    return used + e->rt.share <= 0.75;
}

static int rt_start(zlhip_engine *e, int nframes)
{
    if (g_rt.quiescing.load(std::memory_order_acquire) > 0) return ZL_RT_BUSY;
    { std::lock_guard<std::mutex> lk(g_rt.mu); if (!rt_fits(e)) return ZL_RT_NOFIT; }   // (asked again below, where the kernel is launched)
    { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    for (auto &c : e->slots) if (c.inflight) { ZL_HIP(e, hipEventSynchronize(c.done)); c.inflight = false; int h_ = harvest_slot(e, c); if (h_ != ZLHIP_OK) return h_; }
    if (e->planStream) ZL_HIP(e, hipStreamSynchronize(e->planStream));
    if (!e->rt.h) {
        ZL_HIP(e, hipHostMalloc((void **)&e->rt.h, sizeof(ZlRtShared)));
        ZL_HIP(e, hipHostGetDevicePointer((void **)&e->rt.d, e->rt.h, 0));
        std::memset(e->rt.h, 0, sizeof(ZlRtShared));
        ZL_HIP(e, hipStreamCreateWithFlags(&e->rt.stream, hipStreamNonBlocking));
        ZL_HIP(e, hipMalloc((void **)&e->rt.dev, sizeof(ZlRtDev)));
        ZL_HIP(e, hipMalloc((void **)&e->rt.devRanges, (size_t)std::max(e->V, 1) * sizeof(ZlOpRange)));   // at most one range per voice
    }
    // the hand-off words start from the last block the previous residency finished
    {
        ZlRtDev z; std::memset(&z, 0, sizeof z); z.pub_seq = e->rt.seq;
        ZL_HIP(e, hipMemcpyAsync(e->rt.dev, &z, sizeof z, hipMemcpyHostToDevice, e->rt.stream));
        ZL_HIP(e, hipStreamSynchronize(e->rt.stream));
    }
    zlhip_engine::CallSlot &c = e->slots[0];
    zlhip_engine::PlanSet &q = e->ps[0];
    ZlBatch A; std::memset(&A, 0, sizeof A);
    A.V = e->V; A.B = e->cfg.num_buses; A.VPB = e->cfg.voices_per_bus; A.N = nframes; A.K = 1; A.Ktot = 1; A.k0 = 0;
    if (rt_wide(e)) { A.G = 1; A.groups = A.VPB; }                  // one workgroup per voice, the bus summed from partial rows (q.partials)
    else { A.G = A.VPB; A.groups = 1; }                             // one workgroup per bus
    A.NB = 1;
    A.clocks_regular = 1; A.inline_clock = 1; A.fuse_assemble = 1;
    A.rt_stamps = e->rt.stampsOn ? 1 : 0;
    A.mode = e->cfg.mode;
    A.sounds = e->dSounds; A.clips = e->dClips; A.arena = e->arena; A.voices = e->dVoices; A.reports = c.dReports; A.pass_cache = e->dPassCache;
    A.bus = e->hBusDev; A.stats = nullptr; A.levels = e->dLevels;
    A.fan = e->hFanDev; A.pass = e->hPassRtDev;                     // a cycle says whether it wants the fan-out (ZlRtShared::fan_seq)
    A.vconst = q.vconst; A.runs = q.runs; A.tsegs = q.tsegs; A.plan_hdr = q.hdr; A.plan_seg0 = q.seg0; A.plan_seg1 = q.seg1;
    A.ctl_P = q.ctlP; A.ctl_env = q.ctlEnv; A.partials = q.partials; A.ctl_next = q.ctlNext; A.sim_const = q.simConst;
    A.ctl_slots = e->ctlSlotsOverride >= 0 ? std::min<int>(e->ctlSlotsOverride, (int)(e->ctlPoolFrames / (size_t)nframes)) : (int)std::min<size_t>(e->ctlPoolFrames / (size_t)nframes, 0x7fffffff);
    // (every cycle is a plan window of its own: its pool base travels in the mailbox)
    // registered and launched under the registry's lock: a thread that starts a device-synchronising call either sees this engine
    // in the registry (and its kernel will read the request), or this thread sees its request and does not launch
    std::lock_guard<std::mutex> lk(g_rt.mu);
    if (g_rt.quiescing.load(std::memory_order_acquire) > 0) return ZL_RT_BUSY;
    if (!rt_fits(e)) return ZL_RT_NOFIT;
    __atomic_store_n(&e->rt.h->state, 0u, __ATOMIC_RELEASE);
    __atomic_store_n(&e->rt.h->yield, 0u, __ATOMIC_RELEASE);
    for (uint32_t &d : e->rt.h->wg_done) __atomic_store_n(&d, (uint32_t)e->rt.seq, __ATOMIC_RELAXED);   // (no workgroup has finished the cycle to come)
    ZL_KERNEL(e, zl_launch_rt_loop(A, e->rt.d, e->rt.dev, e->rt.seq, e->rt.idleTicks, e->dGain, c.hReportsDev, c.hGainDev, e->rt.devRanges, std::max(e->rt.vw, 1), std::min(256, (nframes + 63) & ~63), e->rt.stream));
    e->rt.running = true; e->rt.nframes = nframes; e->rt.starts += 1;
    if (std::find(g_rt.engines.begin(), g_rt.engines.end(), e) == g_rt.engines.end()) g_rt.engines.push_back(e);
    return ZLHIP_OK;
}

// ---- where a cycle's time went (ZL_RT_TRACE / zlhip_rt_last_cycle) ---------------------------------------------------------------
// A real-time engine is judged by its worst cycle, and a worst cycle has to be attributable: to the host before the cycle is posted
// (command upload, a kernel restart), to the wait for the device, or to the host afterwards -- and inside the wait, to the device or
// to the waiting THREAD having been taken off its core (the longest gap between two polls of the spin, or between the start and the
// end of the blocking wait; the thread's involuntary context switches).
namespace {
inline double us_between(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
inline long thread_nivcsw() { struct rusage ru; return getrusage(RUSAGE_THREAD, &ru) == 0 ? ru.ru_nivcsw : 0; }
void rt_trace_report(const zlhip_engine *e, const zlhip_rt_cycle_trace &t)
{
    if (e->rt.slowUs > 0.0 && t.total_us > e->rt.slowUs)
        std::fprintf(stderr, "zlhip slow real-time cycle %llu (%s): %.0f us = host before the post %.0f + wait for the device %.0f + host after %.0f; "
                             "longest gap between two polls of the waiting thread %.0f us, its involuntary context switches in the cycle: %ld; "
                             "device stages (ZL_RT_STAMPS) %.1f us\n",
                     (unsigned long long)t.cycle, t.resident ? "resident kernel" : "launches", t.total_us, t.before_post_us, t.wait_us, t.after_us,
                     t.max_poll_gap_us, (long)t.involuntary_switches, t.device_us);
}
}  // namespace

// The device views of the caller's output buffers, if all of them are page-locked and mapped (looked up once per set of pointers).
static bool rt_out_views(zlhip_engine *e, float *out_left, float *out_right, float *fan_out)
{
    zlhip_engine::OutViews &v = e->outViews;
    if (!e->directOut) return false;
    if (v.hL == out_left && v.hR == out_right && v.hF == fan_out) return v.ok;
    v.hL = out_left; v.hR = out_right; v.hF = fan_out; v.ok = false;
    void *dL = nullptr, *dR = nullptr, *dF = nullptr;
    if (hipHostGetDevicePointer(&dL, out_left, 0) != hipSuccess || hipHostGetDevicePointer(&dR, out_right, 0) != hipSuccess
        || (fan_out && hipHostGetDevicePointer(&dF, fan_out, 0) != hipSuccess)) { (void)hipGetLastError(); return false; }   // pageable memory: staged
    v.dL = (float *)dL; v.dR = (float *)dR; v.dF = (float *)dF;
    v.ok = ((uintptr_t)dL % 4 == 0) && ((uintptr_t)dR % 4 == 0);
    return v.ok;
}

// One real-time block through the resident kernel: post the block in the mailbox, spin until the kernel has published it.
static int rt_render(zlhip_engine *e, int32_t nframes, const zlhip_clock *clock, float *out_left, float *out_right,
                     const zlhip_passthrough_params *fan_params, float *fan_out)
{
    zlhip_engine::CallSlot &c = e->slots[0];
    zlhip_engine::PlanSet &q = e->ps[0];
    // from here to the end of the cycle this engine's share of the device counts as taken (rt_fits), whatever its kernel does meanwhile
    struct InCycle { std::atomic<bool> &f; explicit InCycle(std::atomic<bool> &x) : f(x) { f.store(true, std::memory_order_release); } ~InCycle() { f.store(false, std::memory_order_release); } } inCycle(e->rt.inCycle);
    const bool tr = e->rt.traceOn;
    const auto tEnter = tr ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    const long sw0 = tr ? thread_nivcsw() : 0;
    if (e->rt.running && (e->rt.nframes != nframes || __atomic_load_n(&e->rt.h->state, __ATOMIC_ACQUIRE) == 2u)) {
        int rc = rt_stop(e);                                       // another block size, or the kernel left after an idle spell
        if (rc != ZLHIP_OK) return rc;
    }
    ZlBatch A; std::memset(&A, 0, sizeof A);
    // the block's voice operations and clip edits: mapped host memory, read in place by the kernel.  (Growing the buffers frees
    // pinned memory, which waits for the device: stop the kernel first, and grow them before it is started again.)
    if (e->hc.pendingOps.size() > c.opsCap || e->hc.pendingOps.size() > c.rangesCap || e->hc.pendingClipEdits.size() > c.editsCap) {
        int rc = rt_stop(e);
        if (rc != ZLHIP_OK) return rc;
        ZlQuiesce quiet(e);
        ZL_HIP(e, grow_mapped(&c.hOps, &c.hOpsDev, &c.opsCap, e->hc.pendingOps.size()));
        ZL_HIP(e, grow_mapped(&c.hRanges, &c.hRangesDev, &c.rangesCap, e->hc.pendingOps.size()));
        ZL_HIP(e, grow_mapped(&c.hEdits, &c.hEditsDev, &c.editsCap, e->hc.pendingClipEdits.size()));
    }
    // (started before the operations are taken out of the host's pending list: when the start is refused, the cycle goes through
    // the launched path with everything still pending)
    if (!e->rt.running) { int rc = rt_start(e, nframes); if (rc != ZLHIP_OK) return rc; }
    ZlRtShared *sh = e->rt.h;
    unsigned long long *w = sh->cmd;                                // the cycle's command words (zl_types.h, ZlRtShared)
    {
        // knob edits (no slice table) ride in the mailbox itself, the first ZL_RT_INLINE_EDITS of them; the rest go through the edit buffer
        int ni = 0;
        auto &pe = e->hc.pendingClipEdits;
        for (size_t i = 0; i < pe.size() && ni < ZL_RT_INLINE_EDITS;) {
            if (pe[i].full) { ++i; continue; }
            unsigned long long *ew = w + ZL_RT_CMD_FIXED + ni * ZL_RT_EDIT_WORDS;
            ew[0] = (unsigned long long)(uint32_t)pe[i].clip;
            std::memcpy(ew + 1, &pe[i].c, ZL_CLIP_HEAD_BYTES);
            ++ni;
            pe.erase(pe.begin() + (long)i);
        }
        for (; ni < ZL_RT_INLINE_EDITS; ++ni) w[ZL_RT_CMD_FIXED + ni * ZL_RT_EDIT_WORDS] = (unsigned long long)(uint32_t)-1;
    }
    int rc = upload_ops(e, c, A);
    if (rc != ZLHIP_OK) return rc;
    ZlClock ck;
    ZlHostControl::fill_clock(ck, *clock, nframes);
    w[0] = (unsigned long long)(uint32_t)nframes | ((unsigned long long)(uint32_t)A.n_op_ranges << 32);
    w[1] = (unsigned long long)(uintptr_t)A.ops; w[2] = (unsigned long long)(uintptr_t)A.op_ranges;
    w[3] = q.ctlBase; q.ctlBase += (unsigned long long)e->V + 1ull;
    w[4] = ck.current_usecs; w[5] = ck.next_usecs; w[6] = ck.playhead; w[7] = ck.playhead_usecs; w[8] = ck.subbeat_usecs; w[9] = ck.usecs_per_frame;
    w[11] = (unsigned long long)(uintptr_t)A.clip_edits;
    const bool direct = rt_out_views(e, out_left, out_right, fan_out);
    if (direct) {
        w[12] = (unsigned long long)(uintptr_t)e->outViews.dL; w[13] = (unsigned long long)(uintptr_t)e->outViews.dF;
        w[14] = (unsigned long long)(long long)nframes; w[15] = (unsigned long long)(long long)(e->outViews.dR - e->outViews.dL);
    } else {
        w[12] = (unsigned long long)(uintptr_t)e->hBusDev; w[13] = (unsigned long long)(uintptr_t)e->hFanDev;
        w[14] = (unsigned long long)(2ll * nframes); w[15] = (unsigned long long)(long long)nframes;
    }
    uint32_t fanSeq = 0u;
    if (fan_out) {
        // the JackPassthrough parameters: a table in mapped host memory and its version.  A workgroup keeps its bus's entry across cycles
        // and reads the table again only when the version moved, so a quiet cycle makes no extra trip over PCIe
        bool moved = false;
        for (int b = 0; b < e->cfg.num_buses; ++b) {
            const ZlPassParams pp = pass_params(fan_params[b]);
            if (std::memcmp(&pp, &e->hPassRt[b], sizeof pp) != 0) { e->hPassRt[b] = pp; moved = true; }
        }
        if (moved && ++e->passSeq == 0u) e->passSeq = 1u;
        fanSeq = e->passSeq;
    }
    w[10] = (unsigned long long)(uint32_t)A.n_clip_edits | ((unsigned long long)fanSeq << 32);
    const unsigned long long seq = ++e->rt.seq;
    const auto tPost = tr ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    auto tPoll = tPost; double maxGap = 0.0;
    __atomic_store_n(&sh->cmd_seq, seq, __ATOMIC_RELEASE);
    // spin: a block takes some tens of microseconds.  The kernel may have left (idle timeout) just before the post: then start
    // it again -- it picks the posted block up at once (first_seq = the last block it saw finished).
    const auto spin0 = std::chrono::steady_clock::now();
    // narrow buses: every workgroup reports its own completion (wg_done); wide buses: the last arrival writes done_seq
    const int doneSlots = rt_wide(e) ? 0 : e->cfg.num_buses;
    auto cycle_done = [&]() -> bool {
        if (doneSlots == 0) return __atomic_load_n(&sh->done_seq, __ATOMIC_ACQUIRE) == seq;
        for (int z = 0; z < doneSlots; ++z) if (__atomic_load_n(&sh->wg_done[z], __ATOMIC_ACQUIRE) != (uint32_t)seq) return false;
        return true;
    };
    for (unsigned long long spins = 0;; ++spins) {
        if (cycle_done()) break;
        if (tr) { const auto n_ = std::chrono::steady_clock::now(); const double g = us_between(tPoll, n_); if (g > maxGap) maxGap = g; tPoll = n_; }
        if ((spins & 0xfffu) == 0xfffu) {
            if (__atomic_load_n(&sh->state, __ATOMIC_ACQUIRE) == 2u) {
                ZL_HIP(e, hipStreamSynchronize(e->rt.stream));
                e->rt.running = false;
                if (cycle_done()) break;
                rt_unregister(e);
                e->rt.seq = seq - 1;                               // the restarted kernel must see `seq` as new
                // (a device-synchronising call somewhere in the process made the kernel leave with this cycle posted but not taken:
                // wait for that call to end -- it is a one-off of some milliseconds -- the cycle's inputs are in the mailbox already)
                // (this engine's share of the device stayed taken through the cycle, so the kernel still fits)
                while (((rc = rt_start(e, nframes)) == ZL_RT_BUSY || rc == ZL_RT_NOFIT) && std::chrono::steady_clock::now() - spin0 < std::chrono::seconds(2)) { }
                e->rt.seq = seq;
                if (rc != ZLHIP_OK) return (rc == ZL_RT_BUSY || rc == ZL_RT_NOFIT) ? fail(e, ZLHIP_ERR_STATE, "resident real-time kernel kept out by a device-wide wait") : rc;
            }
            // (two seconds: hundreds of block periods)
            if (std::chrono::steady_clock::now() - spin0 > std::chrono::seconds(2)) {
                e->rt.enabled = false;
                (void)rt_stop(e);
                if (cycle_done()) break;                                   // it did finish the cycle after all: deliver it; launches from now on
                // The cycle was posted and its voice operations were taken: some buses may have applied and rendered it, others not.
                // The voice table no longer matches the host's control state -- not recoverable: every later render call fails.
                e->failed = true;
                return fail(e, ZLHIP_ERR_STATE, "resident real-time kernel does not answer; the engine is unusable (destroy it)");
            }
        }
    }
    const size_t B = (size_t)e->cfg.num_buses, N = (size_t)nframes;
    const auto tDone = tr ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    if (tr) { const double g = us_between(tPoll, tDone); if (g > maxGap) maxGap = g; }   // (the thread may be taken off between the post and its first poll)
    if (!direct) {
        for (size_t b = 0; b < B; ++b) {
            std::memcpy(out_left + b * N, e->hBus + (b * 2) * N, N * sizeof(float));
            std::memcpy(out_right + b * N, e->hBus + (b * 2 + 1) * N, N * sizeof(float));
        }
        if (fan_out) std::memcpy(fan_out, e->hFan, B * 6 * N * sizeof(float));
    }
    double devUs = 0.0;
    if (e->rt.stampsOn) {                                          // (workgroup 0's last stamp may still be in flight when another workgroup finishes the block)
        for (int i = 0; i < 5; ++i) { const long long d = (long long)(sh->stamps[i + 1] - sh->stamps[i]); if (d >= 0 && d < 100000000ll) { e->rt.stampSum[i] += (double)d * 0.01; devUs += (double)d * 0.01; } }
        e->rt.stampN += 1;
    }
    if (tr) {
        const auto tExit = std::chrono::steady_clock::now();
        zlhip_rt_cycle_trace &t = e->rt.last;
        t.cycle = e->rt.cycles; t.resident = 1; t.total_us = us_between(tEnter, tExit); t.before_post_us = us_between(tEnter, tPost);
        t.wait_us = us_between(tPost, tDone); t.after_us = us_between(tDone, tExit); t.max_poll_gap_us = maxGap;
        t.involuntary_switches = (int64_t)(thread_nivcsw() - sw0); t.device_us = devUs;
        rt_trace_report(e, t);
    }
    e->latest = &c;
    e->lastK = 1; e->lastN = nframes; e->lastBus = direct ? nullptr : e->hBusDev; e->lastWindows = 1;   // (delivered straight to the caller: nothing to read back)
    e->outstanding = false; e->reportsFresh = true;
    e->rt.cycles += 1;
    return ZLHIP_OK;
}

int zlhip_render(zlhip_engine *e, int32_t nframes, const zlhip_clock *clock, float *out_left, float *out_right)
{
    return zlhip_render_fanout(e, nframes, clock, out_left, out_right, nullptr, nullptr);
}

int zlhip_render_fanout(zlhip_engine *e, int32_t nframes, const zlhip_clock *clock, float *out_left, float *out_right,
                        const zlhip_passthrough_params *fan_params, float *fan_out)
{
    if (!e || !clock || !out_left || !out_right || ((fan_params == nullptr) != (fan_out == nullptr))) return ZLHIP_ERR_INVALID;
    if (e->failed) return fail(e, ZLHIP_ERR_STATE, "engine failed (the resident kernel stopped answering in the middle of a cycle): destroy it");
    if (nframes >= 1 && nframes <= e->cfg.max_frames && rt_eligible(e, nframes)) {
        ZL_HIP(e, hipSetDevice(e->device));
        const int rc = rt_render(e, nframes, clock, out_left, out_right, fan_params, fan_out);
        if (rc != ZL_RT_BUSY && rc != ZL_RT_NOFIT) return rc;
        // a device-synchronising call is in progress in the process, or other engines' resident kernels fill the device: this cycle is
        // rendered with launches (same results)
    }
    // the block's mix (and fan-out) is written by the kernels straight into mapped host memory (24 KB for 12 buses x 256 frames):
    // no copy command after the render, one wait for the call's completion event
    const bool tr = e->rt.traceOn;
    const auto tEnter = tr ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    const long sw0 = tr ? thread_nivcsw() : 0;
    const bool direct = rt_out_views(e, out_left, out_right, fan_out);
    int rc = direct ? render_batch_impl(e, 1, nframes, clock, e->outViews.dL, (long long)nframes, (long long)(e->outViews.dR - e->outViews.dL), fan_params,
                                        fan_out ? e->outViews.dF : nullptr, nullptr)
                    : render_batch_impl(e, 1, nframes, clock, e->hBusDev, 0, 0, fan_params, fan_out ? e->hFanDev : nullptr, nullptr);
    if (rc != ZLHIP_OK) return rc;
    const size_t B = (size_t)e->cfg.num_buses, N = (size_t)nframes;
    const auto tPost = tr ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    ZL_HIP(e, hipEventSynchronize(e->latest->done));
    const auto tDone = tr ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    e->outstanding = false;
    if (!direct) {
        for (size_t b = 0; b < B; ++b) {
            std::memcpy(out_left + b * N, e->hBus + (b * 2) * N, N * sizeof(float));
            std::memcpy(out_right + b * N, e->hBus + (b * 2 + 1) * N, N * sizeof(float));
        }
        if (fan_out) std::memcpy(fan_out, e->hFan, B * 6 * N * sizeof(float));
    }
    if (tr) {
        const auto tExit = std::chrono::steady_clock::now();
        zlhip_rt_cycle_trace &t = e->rt.last;
        t.cycle = e->callIndex; t.resident = 0; t.total_us = us_between(tEnter, tExit); t.before_post_us = us_between(tEnter, tPost);
        t.wait_us = us_between(tPost, tDone); t.after_us = us_between(tDone, tExit); t.max_poll_gap_us = t.wait_us;   // (a blocking wait: one "poll")
        t.involuntary_switches = (int64_t)(thread_nivcsw() - sw0); t.device_us = 0.0;
        rt_trace_report(e, t);
    }
    return ZLHIP_OK;
}

int zlhip_rt_last_cycle(zlhip_engine *e, zlhip_rt_cycle_trace *out)
{
    if (!e || !out) return ZLHIP_ERR_INVALID;
    if (!e->rt.traceOn) return fail(e, ZLHIP_ERR_STATE, "cycle tracing is off (ZL_RT_TRACE=1 at engine creation)");
    *out = e->rt.last;
    return ZLHIP_OK;
}

int zlhip_read_bus(zlhip_engine *e, float *out, size_t out_floats)
{
    if (!e || !out) return ZLHIP_ERR_INVALID;
    if (!e->lastBus) return fail(e, ZLHIP_ERR_STATE, "nothing to read back: no batch rendered yet, or the last real-time cycle went straight into the caller's page-locked buffers");
    const size_t need = (size_t)e->cfg.num_buses * 2 * (size_t)e->lastK * (size_t)e->lastN;
    if (out_floats < need) return fail(e, ZLHIP_ERR_INVALID, "output buffer too small");
    ZL_HIP(e, hipSetDevice(e->device));
    { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    e->outstanding = false;
    if (e->lastBus == e->hBusDev) { std::memcpy(out, e->hBus, need * sizeof(float)); return ZLHIP_OK; }   // a real-time block
    ZL_HIP(e, hipMemcpyAsync(out, e->lastBus, need * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    return ZLHIP_OK;
}

int zlhip_voice_reports(zlhip_engine *e, zlhip_voice_report *out, int32_t count)
{
    if (!e || !out || count < e->V) return ZLHIP_ERR_INVALID;
    ZL_HIP(e, hipSetDevice(e->device));
    if (e->outstanding) { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    for (int v = 0; v < e->V; ++v) {
        const ZlReport &r = e->latest->hReports[v];
        zlhip_voice_report &o = out[v];
        // (playing is 2 inside the engine for a voice on a disabled bus)
        o.playing = r.playing ? 1 : 0; o.valid = r.valid; o.gain = r.valid ? e->latest->hGain[v] : 0.0f; o.progress = r.progress;
        o.clip = r.clip; o.reserved = 0; o.source_sample_position = r.P;
    }
    return ZLHIP_OK;
}

int zlhip_debug_enable_trace(zlhip_engine *e, int enable)
{
    if (!e) return ZLHIP_ERR_INVALID;
    e->trace = (enable & 1) != 0;
    // test hooks: bit 1 routes every block through the per-frame control path, bit 2 plans periodic loops pass by pass
    e->forceSlow = ((enable & 2) ? 1 : 0) | ((enable & 4) ? 2 : 0);
    return ZLHIP_OK;
}

int zlhip_debug_read_trace(zlhip_engine *e, int32_t *out, size_t out_ints)
{
    if (!e || !out) return ZLHIP_ERR_INVALID;
    const size_t need = (size_t)e->traceK * e->V * e->traceN;
    if (!e->dTrace || need == 0) return fail(e, ZLHIP_ERR_STATE, "no trace recorded");
    if (out_ints < need) return fail(e, ZLHIP_ERR_INVALID, "trace buffer too small");
    ZL_HIP(e, hipSetDevice(e->device));
    { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    ZL_HIP(e, hipMemcpyAsync(out, e->dTrace, need * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    return ZLHIP_OK;
}

// ---- levels -------------------------------------------------------------------------------------
// libm calls exactly as the reference makes them (no compiler rewrite such as pow(10, x) -> exp10(x))
static double (*volatile zl_pow)(double, double) = static_cast<double (*)(double, double)>(std::pow);
static float (*volatile zl_log10f)(float) = static_cast<float (*)(float)>(log10f);

static float convert_to_dbfs(float raw)                            // AudioLevels.cpp:330-341
{
    if (raw <= 0) return -200;
    const float fValue = 20 * zl_log10f(raw);
    if (fValue < -200) return -200;
    return fValue;
}
static float add_float_db(float db1, float db2)                    // AudioLevels.cpp:234-236
{
    return 10 * zl_log10f((float)(zl_pow(10, db1 / 10) + zl_pow(10, db2 / 10)));
}

int zlhip_levels_tick(zlhip_engine *e, int32_t block_index, int32_t with_hold_bus, zlhip_levels *out)
{
    if (!e || !out) return ZLHIP_ERR_INVALID;
    ZL_HIP(e, hipSetDevice(e->device));
    const ZlBlockLevels *lv = nullptr;
    if (block_index != -2) {
        if (e->lastK <= 0) return fail(e, ZLHIP_ERR_STATE, "no batch rendered yet");
        const int k = block_index < 0 ? e->lastK - 1 : block_index;
        if (k >= e->lastK) return fail(e, ZLHIP_ERR_INVALID, "block index out of range");
        lv = e->dLevels + (size_t)k * e->cfg.num_buses;
    }
    const int B = e->cfg.num_buses;
    // the block levels may still be in flight on a caller's stream (zlhip_render_batch / zlhip_levels_scan_device with
    // stream != NULL): the engine's stream is non-blocking and nothing else orders it behind that work
    { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    ZL_KERNEL(e, zl_launch_levels_tick(e->dLevelState, lv, B, e->lastN, with_hold_bus, e->stream));
    ZL_HIP(e, hipMemcpyAsync(e->hLevelState, e->dLevelState, (size_t)B * sizeof(ZlLevelsState), hipMemcpyDeviceToHost, e->stream));
    { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    e->outstanding = false;
    static const float intToFloatMultiplier = 0.00000152587f;     // AudioLevels.cpp:349
    for (int b = 0; b < B; ++b) {
        const ZlLevelsState &s = e->hLevelState[b];
        zlhip_levels &o = out[b];
        o.peak_a = s.peak_a; o.peak_b = s.peak_b;
        o.peak_a_hold_signal = s.hold_a; o.peak_b_hold_signal = s.hold_b;
        const float peakA = s.peak_a * intToFloatMultiplier, peakB = s.peak_b * intToFloatMultiplier;   // :385
        o.peak_db_a = convert_to_dbfs(peakA); o.peak_db_b = convert_to_dbfs(peakB);                     // :386-387
        o.combined_db = add_float_db(o.peak_db_a, o.peak_db_b);                                         // :394 / :406
        o.hold_db_a = convert_to_dbfs(s.hold_a); o.hold_db_b = convert_to_dbfs(s.hold_b);               // :397-398
        o.rms_a = s.frames > 0 ? sqrtf(s.sumsq_a / (float)s.frames) : 0.0f;
        o.rms_b = s.frames > 0 ? sqrtf(s.sumsq_b / (float)s.frames) : 0.0f;
    }
    return ZLHIP_OK;
}

int zlhip_block_peaks(zlhip_engine *e, int32_t *out, size_t out_ints)
{
    if (!e || !out) return ZLHIP_ERR_INVALID;
    const size_t n = (size_t)e->lastK * e->cfg.num_buses;
    if (n == 0) return fail(e, ZLHIP_ERR_STATE, "no batch rendered yet");
    if (out_ints < n * 2) return fail(e, ZLHIP_ERR_INVALID, "output buffer too small");
    ZL_HIP(e, hipSetDevice(e->device));
    { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    e->outstanding = false;
    std::vector<ZlBlockLevels> tmp(n);
    ZL_HIP(e, hipMemcpyAsync(tmp.data(), e->dLevels, n * sizeof(ZlBlockLevels), hipMemcpyDeviceToHost, e->stream));
    { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    for (size_t i = 0; i < n; ++i) { out[2 * i] = tmp[i].peak_l; out[2 * i + 1] = tmp[i].peak_r; }
    return ZLHIP_OK;
}

int zlhip_levels_scan_device(zlhip_engine *e, const float *bus_dev, int32_t nblocks, int32_t nframes, void *stream)
{
    if (!e || !bus_dev || nblocks < 1 || nblocks > e->cfg.max_batch_blocks || nframes < 1 || nframes > e->cfg.max_frames) return ZLHIP_ERR_INVALID;
    ZL_HIP(e, hipSetDevice(e->device));
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    ZlBatch A; std::memset(&A, 0, sizeof A);
    A.V = e->V; A.B = e->cfg.num_buses; A.VPB = e->cfg.voices_per_bus; A.K = nblocks; A.Ktot = nblocks; A.k0 = 0; A.N = nframes; A.G = A.VPB; A.groups = 1; A.NB = 1;
    A.levels = e->dLevels; A.bus = nullptr;
    ZL_KERNEL(e, zl_launch_finalize(A, bus_dev, s));
    if (s != e->stream) { ZL_HIP(e, hipEventRecord(e->evJoin, s)); e->joins[1] = e->evJoin; }
    e->lastK = nblocks; e->lastN = nframes;
    e->outstanding = true;
    return ZLHIP_OK;
}

// ---- multi-GPU exchange ---------------------------------------------------------------------------
int zlhip_bus_reduce_sum_scan(zlhip_engine *e, const float *pieces_dev, int32_t npieces, int64_t piece_stride_floats, int64_t units,
                              int32_t nframes, float *sum_out_dev, zlhip_unit_levels *levels_out_dev, void *stream)
{
    static_assert(sizeof(zlhip_unit_levels) == sizeof(ZlUnitLevels), "ABI mirror");
    if (!e || !pieces_dev || !sum_out_dev || !levels_out_dev || npieces < 1 || units < 1 || nframes < 1
        || piece_stride_floats < units * (int64_t)nframes) return ZLHIP_ERR_INVALID;
    ZL_HIP(e, hipSetDevice(e->device));
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    const int off = (e->cfg.mode & ZLHIP_MODE_FIX_DELAY) ? 0 : 1;  // tile offset of the RMS order (zl_scan_rows)
    ZL_KERNEL(e, zl_launch_reduce_scan(pieces_dev, npieces, (long long)piece_stride_floats, (long long)units, nframes, off, sum_out_dev,
                                       reinterpret_cast<ZlUnitLevels *>(levels_out_dev), s));
    if (s != e->stream) { ZL_HIP(e, hipEventRecord(e->evJoin, s)); e->joins[1] = e->evJoin; }
    e->outstanding = true;
    return ZLHIP_OK;
}

int zlhip_levels_import_units(zlhip_engine *e, const zlhip_unit_levels *units_dev, int32_t nblocks, int32_t nframes, void *stream)
{
    if (!e || !units_dev || nblocks < 1 || nblocks > e->cfg.max_batch_blocks || nframes < 1 || nframes > e->cfg.max_frames) return ZLHIP_ERR_INVALID;
    ZL_HIP(e, hipSetDevice(e->device));
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    ZL_KERNEL(e, zl_launch_levels_import(reinterpret_cast<const ZlUnitLevels *>(units_dev), e->dLevels, e->cfg.num_buses, nblocks, s));
    if (s != e->stream) { ZL_HIP(e, hipEventRecord(e->evJoin, s)); e->joins[1] = e->evJoin; }
    e->lastK = nblocks; e->lastN = nframes;
    e->outstanding = true;
    return ZLHIP_OK;
}

// ---- JackPassthrough ------------------------------------------------------------------------------
void zlhip_passthrough_params_default(zlhip_passthrough_params *p)
{
    p->dry_amount = 1.0f; p->wet_fx1_amount = 1.0f; p->wet_fx2_amount = 1.0f; p->pan_amount = 0.0f; p->muted = 0;   // JackPassthrough.cpp:27-31
}

int zlhip_passthrough_process(zlhip_engine *e, const zlhip_passthrough_params *params, const float *in_dev, float *out_dev, int64_t frames, void *stream)
{
    if (!e || !params || !in_dev || !out_dev || frames < 1) return ZLHIP_ERR_INVALID;
    ZL_HIP(e, hipSetDevice(e->device));
    hipStream_t s = stream ? (hipStream_t)stream : e->stream;
    std::vector<ZlPassParams> pp((size_t)e->cfg.num_buses);
    for (int b = 0; b < e->cfg.num_buses; ++b) pp[(size_t)b] = pass_params(params[b]);
    ZL_HIP(e, hipMemcpyAsync(e->dPass, pp.data(), pp.size() * sizeof(ZlPassParams), hipMemcpyHostToDevice, s));
    ZL_KERNEL(e, zl_launch_passthrough(e->dPass, in_dev, out_dev, e->cfg.num_buses, (long long)frames, s));
    e->outstanding = true;
    return ZLHIP_OK;
}

// ---- measurement ----------------------------------------------------------------------------------
int zlhip_set_profiling(zlhip_engine *e, int enable)
{
    if (!e) return ZLHIP_ERR_INVALID;
    e->profiling = enable != 0;
    return ZLHIP_OK;
}

int zlhip_last_timings(zlhip_engine *e, zlhip_timings *out)
{
    if (!e || !out) return ZLHIP_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    ZL_HIP(e, hipSetDevice(e->device));
    { int w_ = engine_wait(e); if (w_ != ZLHIP_OK) return w_; }
    e->outstanding = false;
    // harvest in call order: the older slot first
    for (unsigned i = 0; i < 2; ++i) {
        zlhip_engine::CallSlot &c = e->slots[(e->callIndex + i) & 1u];
        int rc = harvest_slot(e, c);
        if (rc != ZLHIP_OK) return rc;
    }
    *out = e->timings;
    out->render_launches = e->lastWindows;
    out->source_bytes = e->latest->hStats->source_bytes;
    out->slow_blocks = e->latest->hStats->slow_blocks;
    out->active_voice_frames = e->latest->hStats->active_frames;
    return ZLHIP_OK;
}

int zlhip_profile_totals(zlhip_engine *e, zlhip_timings *totals, int32_t *calls, int reset)
{
    if (!e) return ZLHIP_ERR_INVALID;
    zlhip_timings last;
    int rc = zlhip_last_timings(e, &last);                         // waits for the outstanding calls and harvests them
    if (rc != ZLHIP_OK) return rc;
    if (totals) *totals = e->totals;
    if (calls) *calls = e->totalCalls;
    if (reset) { std::memset(&e->totals, 0, sizeof e->totals); e->totalCalls = 0; }
    return ZLHIP_OK;
}

float *zlhip_bus_device_ptr(zlhip_engine *e) { return e ? e->dBus : nullptr; }

int zlhip_rt_stats(zlhip_engine *e, uint64_t *kernel_starts, uint64_t *cycles_rendered)
{
    if (!e) return ZLHIP_ERR_INVALID;
    if (kernel_starts) *kernel_starts = e->rt.starts;
    if (cycles_rendered) *cycles_rendered = e->rt.cycles;
    return ZLHIP_OK;
}

int zlhip_rt_residency(zlhip_engine *e, int32_t *resident, double *share)
{
    if (!e) return ZLHIP_ERR_INVALID;
    if (resident) *resident = (e->rt.running && e->rt.h && __atomic_load_n(&e->rt.h->state, __ATOMIC_ACQUIRE) != 2u) ? 1 : 0;
    if (share) *share = e->rt.share;
    return ZLHIP_OK;
}

int zlhip_memory_bytes(zlhip_engine *e, uint64_t *total_device_bytes, uint64_t *arena_bytes)
{
    if (!e) return ZLHIP_ERR_INVALID;
    if (total_device_bytes) *total_device_bytes = (uint64_t)e->deviceBytes;
    if (arena_bytes) *arena_bytes = (uint64_t)(e->arenaFloats + e->arenaSegmentFloats) * sizeof(float);
    return ZLHIP_OK;
}

}  // extern "C"
