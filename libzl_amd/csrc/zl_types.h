// zl_types.h -- POD records shared by the host engine and the HIP kernels (HBM layout).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZL_HD __host__ __device__
#else
#define ZL_HD
#endif

#define ZL_MAX_SLICES    128
#define ZL_BEAT_SUBDIV   96

// juce::ADSR states (restated; see oracle/zl_oracle.h for the JUCE citation)
enum { ZL_ADSR_IDLE = 0, ZL_ADSR_ATTACK = 1, ZL_ADSR_DECAY = 2, ZL_ADSR_SUSTAIN = 3, ZL_ADSR_RELEASE = 4 };

// mode flags (mirror ZLHIP_MODE_*)
enum { ZL_MODE_FIX_GAIN = 1u, ZL_MODE_FIX_DELAY = 2u, ZL_MODE_HERMITE = 4u };

// One block's clock (zlhip_clock + the derived integer microseconds-per-frame of
// SamplerSynthVoice.cpp:183, which is block-constant and computed once on the host).
struct ZlClock {
    uint64_t current_usecs, next_usecs;
    uint64_t playhead, playhead_usecs, subbeat_usecs;
    uint64_t usecs_per_frame;     // (next_usecs - current_usecs) / nframes, integer division
};

// Decoded source in the HBM arena.  Stereo sources are stored INTERLEAVED [L0 R0 L1 R1 ...] so
// the two interpolation taps of both channels are one 16-byte load; mono sources are [x0 x1 ...].
struct ZlSound {
    uint64_t offset;              // float index into the arena (16-byte aligned)
    int32_t  length;              // frames
    int32_t  channels;            // 1 or 2; 0 = slot free / invalid
    double   sample_rate;
};

// ClipAudioSource fields the voice reads (ClipAudioSource.cpp:63-82), device copy.
struct ZlClip {
    float   start_sec, length_sec, length_beats, volume_abs, pan, duration;
    int32_t n_slice_pos;
    int32_t pad;
    double  slice_pos[ZL_MAX_SLICES];
};

// A parameter edit of one clip (zlhip_clip_set): the whole record, applied to the HBM clip table by the device at the
// start of the next render call / real-time cycle (K0, or every workgroup of the resident kernel) -- the boundary at which
// the reference's voices read the parameters (SamplerSynthVoice.cpp:189-196), and no HIP call on the host.
struct ZlClipEdit {
    int32_t clip, pad0, pad1;     // (the record below starts on a 16-byte boundary: it is copied with 16-byte accesses)
    int32_t full;                 // 1: the slice table changed too; 0: only the 32 bytes in front of it (a pan / volume / length knob: one
                                  // 16-byte load per lane for two lanes instead of a kilobyte over PCIe)
    ZlClip  c;
};
#define ZL_CLIP_HEAD_BYTES 32     // offsetof(ZlClip, slice_pos)

// SamplerSynthVoicePrivate + juce::ADSR state (SamplerSynthVoice.cpp:20-39), device resident.
struct ZlVoiceState {
    double   P;                   // sourceSamplePosition
    double   pitch_ratio;
    double   src_len;             // sourceSampleLength
    double   adsr_sr;             // ADSR sample rate (= source sample rate, quirk Q8)
    uint64_t next_loop_tick, next_loop_usecs;
    float    lgain, rgain;
    float    env;
    int32_t  adsr_state;
    float    attack_rate, decay_rate, release_rate;
    float    sustain, release;    // ADSR parameters still needed after noteOn
    int32_t  clip;                // clip == sound id; -1 none
    int32_t  slice;
    int32_t  looping;
    int32_t  playing;             // isPlaying && clipCommand != nullptr; 2 = ... on a DISABLED sampler channel: the voice keeps its state and is
                                  // not processed (SamplerChannel::process, SamplerSynth.cpp:123), commands still reach it
    int32_t  loop_phase1;         // 1 + frames since the last loop restart at the end of the last window; 0 = unknown (a hint for
                                  // the pass cache, verified against P before it is used)
};

// Host -> device voice operations, applied in order before the next block is planned.
enum { ZL_OP_START = 1, ZL_OP_NOTE_OFF = 2, ZL_OP_PATCH = 3, ZL_OP_HARD_STOP = 4, ZL_OP_FREEZE = 5, ZL_OP_THAW = 6 };   // 5 / 6: the voice's sampler channel was disabled / enabled
enum { ZL_PATCH_GAIN = 1, ZL_PATCH_LOOPING = 2, ZL_PATCH_SLICE = 4, ZL_PATCH_POSITION = 8 };
struct ZlVoiceOp {
    int32_t  voice;
    int32_t  kind;
    int32_t  patch_mask;
    int32_t  looping;
    int32_t  slice;
    float    gain;
    double   position;
    ZlVoiceState start;           // full post-startNote state for ZL_OP_START
};
struct ZlOpRange { int32_t voice, first, count, pad; };

// ---- K1 -> K2 records -------------------------------------------------------------------------
struct ZlVoiceConst {             // per voice, constant over a batch; 48 bytes (3 x 16 for LDS staging)
    uint64_t src_offset;
    int32_t  sample_duration;     // length - 1 (SamplerSynthVoice.cpp:191)
    int32_t  channels;
    float    lgain, rgain, clip_volume, lpan;
    float    rpan;
    float    env;                 // envelope of the implied (run) blocks: the sustain level
    int32_t  pad[2];
};

enum { ZL_PLAN_ACTIVE = 1, ZL_PLAN_SLOW = 2, ZL_PLAN_ENV = 4,      // ENV: the envelope ramps inside the block (ZlPlanSeg1 holds the slopes)
       // SLOW blocks keep their per-frame control in a slot of the window's control pool (slot index in ZlPlanSeg0::step).  When the
       // pool is exhausted the block carries what K2 needs to recompute its control instead:
       ZL_PLAN_NOSLOT_SIM = 8,      // a block K1 simulated: the voice state at the block's start (ZlPlanSeg0/1, zl_plan.h zl_snapshot_*)
       ZL_PLAN_NOSLOT_EXPAND = 16 };// a multi-segment block: its position in the segment stream (ZlPlanSeg1::n1 = idx0, P1 = base0)
// K1's output per voice and plan window is a stream of linear segments in window time: frames [t, next segment's t)
// are rendered at position P + (frame - t) * step with envelope E + (frame - t) * estep, both exactly (zl_plan.h).  A
// segment ends at a binade crossing of the position or of the envelope, an ADSR state change, a loop restart or the
// end of the voice.  K1c turns the stream into per-block plans, lane-parallel.
// ZL_TSEG_SLOW marks the start of blocks that K1 simulated per frame itself (envelope transients, release tails).
#define ZL_MAXTSEG   1024
#define ZL_TSEG_SLOW 1
struct ZlTSeg { double P, step; int32_t t, flags; float E, estep; };   // E: envelope of frame t, + estep per frame (exact fp32 ramp)

// Whole blocks inside one segment differ only in P0 and need no plan record at all: blocks [k0, k1) of a run
// are rendered whole, in sustain, with P0(k) = P + (k - k0) * N * step (exact).  The first ZL_MAXRUNS such
// stretches of a voice are kept inline (K2 resolves them without touching the plan arrays: all of the steady
// state of playback at the source rate); every other block gets an explicit plan from K1c.
#define ZL_MAXRUNS 6
struct ZlRun { double P, step; int32_t k0, k1; };
struct ZlRunList {
    int32_t n;                    // inline runs used
    int32_t dead_from;            // first block in which the voice no longer plays (K if it plays to the end; 0 = idle)
    int32_t nts;                  // segments in ZlBatch::tsegs
    int32_t t_end;                // frame (window time) at which the voice stopped; INT_MAX while it plays
    ZlRun   r[ZL_MAXRUNS];
    // A sample-space loop in sustain is exactly periodic (every restart sets the same integer position): K1 plans one
    // full pass -- segments [per_j0, per_j0 + per_n) of the stream, starting at frame per_t0 -- and the stream then
    // repeats every per_M frames up to the end of the window.  per_n == 0: no periodic part.
    int32_t per_t0, per_M, per_j0, per_n;
};

// The plan of one (block, voice).  In HBM it is split into three arrays so that K1 (one lane per voice)
// writes fully coalesced 16-byte lanes: ZlPlanHdr[K][V], ZlPlanSeg0[K][V] and, only for blocks with a
// second segment, ZlPlanSeg1[K][V].  K2 reassembles this 64-byte record in LDS.
struct ZlPlanHdr  { int32_t flags, n_active, nseg; float env; };
struct ZlPlanSeg0 { double P0, step; };
struct ZlPlanSeg1 { double P1, step1; int32_t n1; float estep0, E1, estep1; };   // second segment + the envelope slopes of both
struct ZlBlockPlan {              // per (block, voice); 64 bytes, the first two linear segments inline
    int32_t flags;
    int32_t n_active;             // frames rendered in this block (N unless the voice stopped inside it)
    int32_t nseg;                 // 1 or 2 for fast blocks (blocks with more are expanded to per-frame control by K1c)
    float   env;                  // envelope of frame 0 of a fast block (of every frame in sustain)
    double  P0;                   // position of frame 0
    double  step;                 // exact per-frame increment inside the first segment
    int32_t n1;                   // first frame of the second segment (INT_MAX when there is none)
    float   estep0;               // envelope slope of the first segment (0 in sustain); env is the envelope of frame 0
    double  P1;                   // position of frame n1
    double  step1;
    float   E1, estep1;           // envelope of frame n1 and the slope of the second segment
};
// What re-simulating a block needs of a voice besides the snapshot in its plan record: constant over a plan window (commands
// apply between calls only).  Written by K1 for every playing voice.
struct ZlSimConst {
    double   pitch_ratio, adsr_sr, tail_T;
    uint64_t length_ticks;
    float    attack_rate, decay_rate, sustain, release;
    int32_t  start_int, stop_pos, beat_locked, looping;
};

struct ZlReport {                 // device side of zlhip_voice_report
    int32_t playing, valid;
    uint32_t peak_bits;           // max over the last block of (l'+r'), as float bits (>= 0)
    float   progress;
    int32_t clip, pad;
    double  P;
};

static_assert(sizeof(ZlReport) == 32, "a report is two 16-byte words (zl_k2_body publishes it that way)");

struct ZlBlockLevels {            // per (block, bus)
    int32_t peak_l, peak_r;       // max over the block of (int)|131072*x|
    float   sumsq_l, sumsq_r;     // sum of squares (RMS extension)
};

struct ZlUnitLevels {             // levels of one (bus, channel, block) unit of a bus piece (multi-GPU exchange, zl_k_reduce_scan)
    int32_t peak;                 // max over the unit of (int)|131072*x|
    float   sumsq;                // sum of squares in the defined order (RMS extension)
};

struct ZlLevelsState {            // AudioLevelsChannel meter state per bus
    int32_t peak_a, peak_b;
    float   hold_a, hold_b;
    float   sumsq_a, sumsq_b;
    int32_t frames, pad;
};

// JackPassthrough parameters of one bus (JackPassthrough.cpp:27-31)
struct ZlPassParams { float dry, fx1, fx2, pan; int muted; };

// A sample-space loop in sustain repeats exactly: the segments of one pass (restart to restart, times relative to the
// restart), kept per voice across windows and calls and valid for the key (start, stop, ratio, sustain level).
#define ZL_PASS_MAXSEG 64
struct ZlPassCache {
    double  ratio;
    int32_t start_int, stop_pos;
    float   sustain;
    int32_t M, n, valid;
    ZlTSeg  seg[ZL_PASS_MAXSEG];
};

struct ZlBatchStats {
    unsigned long long source_bytes;
    unsigned long long slow_blocks;
    unsigned long long active_frames;
};

// Mailbox between zlhip_render and the resident real-time kernel (zl_k_rt_loop), in host memory mapped into the device.
// The host packs a cycle's inputs into `cmd` (8-byte words, the layout below), then writes cmd_seq; the kernel renders the cycle, writes
// the mix and the reports into host memory and reports completion.  state: 0 not started, 1 resident, 2 left (stop request or idle timeout).
//   Workgroup 0 alone polls cmd_seq; it republishes only the NUMBER in device memory, and every workgroup then reads the words itself,
//   with system-scope loads (one trip over PCIe, all words in flight together, past the caches -- no fence, no L2 invalidation).
//   Completion: one workgroup per bus (narrow buses) -- each writes its own word of wg_done, the host waits for all of them; no
//   arrival counter, no last-arrival store.  One workgroup per voice (wide buses): the arrival counter and done_seq, as before.
#define ZL_RT_INLINE_EDITS 2
#define ZL_RT_EDIT_WORDS (1 + ZL_CLIP_HEAD_BYTES / 8)      // clip id + the head, in 8-byte words
#define ZL_RT_CMD_FIXED 16                                 // the words in front of the inline knob edits
#define ZL_RT_CMD_WORDS (ZL_RT_CMD_FIXED + ZL_RT_INLINE_EDITS * ZL_RT_EDIT_WORDS)
#define ZL_RT_DONE_SLOTS 64                                // narrow buses: at most 64 resident workgroups (rt_eligible)
// cmd[0] nframes | n_op_ranges << 32      [1] ops  [2] op_ranges (device views of the cycle's voice operations, mapped host memory)
// cmd[3] ctl_base (base of this cycle's control-slot pool, zl_plan.h zl_ctl_alloc)        [4..9] ZlClock
// cmd[10] n_clip_edits | fan_seq << 32     [11] clip_edits (the cycle's clip-parameter edits, mapped host memory)
//         fan_seq: 0 = the cycle delivers the buses only; else it also delivers the JackPassthrough fan-out of every bus and this is the
//         version of the parameter table (ZlBatch::pass, mapped host memory): a workgroup re-reads its bus's entry only when it moved
// cmd[12] out_bus  [13] out_fan  [14] out_bus_stride  [15] out_ch_stride: where the cycle's rows go (device views) -- the engine's staging
//         rows in mapped host memory or, when the caller's buffers are page-locked, straight into them: row of (bus, channel) =
//         out_bus + bus * out_bus_stride + channel * out_ch_stride floats; the fan-out rows [B][6][nframes] at out_fan
// cmd[16..] up to ZL_RT_INLINE_EDITS knob edits (clip id, then the 32 bytes in front of the slice table): a pan or volume knob turned while
//         playing travels in the mailbox itself
struct ZlRtShared {
    unsigned long long cmd_seq, done_seq;
    uint32_t state, stop;
    uint32_t yield;               // another thread of the process is about to make a device-synchronising HIP call (hipFree, ...):
                                  // leave after the cycle in flight (zl_engine.cpp, ZlQuiesce)
    uint32_t pad;
    unsigned long long stamps[8]; // s_memrealtime (100 MHz) at the kernel's stages of the last block (diagnostic)
    unsigned long long cmd[ZL_RT_CMD_WORDS];
    uint32_t wg_done[ZL_RT_DONE_SLOTS];   // narrow buses: workgroup z's last finished cycle (low 32 bits of its sequence number)
};

// Device side of the resident kernel: workgroup 0 watches the mailbox and republishes every block in HBM for the other
// workgroups (agent-scope atomics: eight-byte words, visible across XCDs without cache fences); `arrive` counts the workgroups
// that have finished the block.
#define ZL_RT_MAX_BUSES 256
struct ZlRtDev {
    unsigned long long pub_seq;                  // the block being rendered (~0ull: leave)
    unsigned long long cmd[ZL_RT_CMD_WORDS];     // wide buses: workgroup 0's copy of the mailbox's words (a thousand readers would queue on PCIe)
    unsigned int arrive, pad;
    unsigned int bus_arrive[ZL_RT_MAX_BUSES];    // wide buses (one workgroup per voice): the voices of a bus that have written their partial mix
};

// Launch-wide arguments (passed by value to the kernels).
struct ZlBatch {
    int32_t V, B, VPB, K, N;      // voices, buses, voices per bus, blocks in this plan window, frames per block
    int32_t k0, Ktot;             // first block of the window inside the call, blocks of the whole call (bus stride)
    int32_t G;                    // voices per render task (mix group); groups per bus = ceil(VPB / G)
    int32_t groups;
    int32_t NB;                   // buses rendered by one K2 workgroup (narrow buses in batches), else 1
    int32_t tail_from, tail_split, tail_nb;   // narrow buses, one workgroup per block: launch slots from tail_from on render the window's LAST blocks with
                                  // tail_nb buses per workgroup, tail_split workgroups per block (zl_launch_render; 0 = no split tail)
    int32_t host_fmt;             // offline bounce, direct delivery: K2 also stores the finished bus into the caller's page-locked HOST buffer
                                  // (0 = fp32 planar [B][2][host_total], 1 = 16-bit stereo [B][host_total][2]) -- no conversion pass, no copy
    void   *host_out;             // device view of that buffer, or nullptr
    long long host_total, host_k0;// frames per row of the host buffer; first block of this call inside it
    uint32_t mode;
    int32_t n_op_ranges;
    int32_t trace;                // 1 = write pos trace
    int32_t clocks_regular;       // 1: current_usecs never decreases and usecs_per_frame is constant (< 2^21) over the call:
                                  //    the block in which a beat-locked loop restarts can be found by bisection
    int32_t inline_clock;         // 1: the (single) block's clock travels in clock0 with the kernel arguments
    int32_t fuse_assemble;        // 1: K1 assembles the plan records itself (single real-time block: one launch less)
    int32_t staged;               // 1: K2 stages source windows in LDS (LDS-DMA ring) instead of gathering into registers
    ZlClock clock0;
    const ZlClock      *clocks;   // [K]
    const ZlSound      *sounds;
    const ZlClip       *clips;
    const float        *arena;
    ZlVoiceState       *voices;   // [V]
    const ZlVoiceOp    *ops;
    const ZlOpRange    *op_ranges;
    const ZlClipEdit   *clip_edits; // [n_clip_edits] clip-parameter edits of the call (mapped host memory), applied by K0
    int32_t             n_clip_edits;
    int32_t             rt_stamps;  // resident kernel: 1 = workgroup 0 writes its stage times into the mailbox (ZL_RT_STAMPS; each is a store to host
                                    // memory that the stage's barrier then waits for -- 1 to 10 us per cycle depending on the box: off by default)
    ZlVoiceConst       *vconst;   // [V]
    ZlRunList          *runs;     // [V]
    ZlTSeg             *tsegs;    // [V][ZL_MAXTSEG] segment stream of the window
    ZlPlanHdr          *plan_hdr; // [K][V] explicit plans (blocks not covered by a run)
    ZlPlanSeg0         *plan_seg0;// [K][V]
    ZlPlanSeg1         *plan_seg1;// [K][V] valid where nseg >= 2
    double             *ctl_P;    // [ctl_slots][N]  per-frame control of slow blocks: a pool of block-sized slots per window
    float              *ctl_env;  // [ctl_slots][N]
    unsigned long long *ctl_next; // the pool's bump counter (monotone across windows; a window's slots count from ctl_base)
    unsigned long long  ctl_base;
    int32_t             ctl_slots;
    ZlSimConst         *sim_const;// [V]
    ZlReport           *reports;  // [V]
    float              *partials; // [K][B][groups][2][N]  (only when groups > 1)
    float              *bus;      // [B][2][Ktot*N] -- or, with the strides below, rows wherever the caller wants them
    long long           bus_stride, ch_stride;   // floats from one bus's rows to the next / from a bus's left row to its right row; 0 = the
                                                 // layout above (2 * Ktot * N, Ktot * N).  A real-time cycle delivered straight into the caller's
                                                 // page-locked out_left[B][N] / out_right[B][N] has bus_stride = N, ch_stride = out_right - out_left
    ZlPassCache        *pass_cache; // [V] or nullptr
    const ZlPassParams *pass;     // [B] JackPassthrough parameters of the fused fan-out (with fan)
    float              *fan;      // [B][6][Ktot*N] dry L,R / wetFx1 L,R / wetFx2 L,R of every bus, or nullptr
    ZlPassParams        pass0;    // pass_inline = 1 (resident kernel): the parameters of the workgroup's bus travel here instead of in pass[]
    int32_t             pass_inline;
    int32_t             pad_batch;
    int32_t             tile_accum; // 1 (resident kernel, blocks longer than 256 frames): ONE workgroup walks the frame tiles of its block in
                                    // order (bx = 0, 1, ...), and the fused level scan carries on from tile to tile (same defined order)
    // the call's per-voice reports, published by the K2 workgroups that render its LAST block (fused_reports = 1: every bus of that
    // block is summed whole by one workgroup, so each of them knows its voices' peaks without asking anybody): gain = peak * 0.5f
    // (SamplerSynthVoice.cpp:266) into rep_gain, the reports and gains into mapped host memory, the call's statistics too -- the report
    // kernel and its packet between two calls' render kernels are gone
    float              *rep_gain;       // [V] HBM
    ZlReport           *rep_host;       // [V] mapped host memory, or nullptr
    float              *rep_host_gain;  // [V]
    ZlBatchStats       *rep_host_stats; // mapped host memory, or nullptr
    int32_t             fused_reports;
    int32_t             pad_reports;
    ZlBlockLevels      *levels;   // [K][B]
    int32_t            *pos_trace;// [K][V][N] or null
    ZlBatchStats       *stats;
};
