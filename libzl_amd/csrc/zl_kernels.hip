// zl_kernels.hip -- HIP kernels of the sampler engine, written for gfx950 (CDNA4, wave64).
//
//   K0 zl_k0_apply_ops   device half of SamplerChannel::handleCommand (SamplerSynth.cpp:187-230)
//   K1 zl_k1_plan        per-voice control plan of SamplerSynthVoice::process (:174-270), zl_plan.h
//   K1c zl_k1c_assemble  segment streams -> per-block plan records, multi-segment blocks -> per-frame control
//                        (lane-parallel)
//   K2 zl_k2_render      gather + interpolate + gain/ADSR/pan + voice->bus sum (:198-221,
//                        SamplerSynth.cpp:134-140); HBM-bound, no MFMA (about 22 flop per 8 bytes)
//   K3 zl_k3_finalize    ordered sum of mix-group partials + AudioLevels block scan
//                        (AudioLevels.cpp:361-383) + report finalisation (:265-267)
//   zl_k_levels_tick     AudioLevels decay / hold tick (AudioLevels.cpp:359-360,395-396)
//   zl_k_passthrough     JackPassthrough fan-out (JackPassthrough.cpp:55-113)
//   zl_k_interleave      planar -> interleaved source upload (SamplerSynthSound.cpp:45-47 layout choice)
//
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off (the oracle defines an un-fused rounding
// sequence; see zl_render.h).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstddef>
#include "zl_types.h"
#include "zl_plan.h"
#include "zl_render.h"
#include "zl_kernels.h"

// ------------------------------------------------------------------------------------------------
// One clip-parameter edit (zlhip_clip_set) -> the HBM clip table, by the 64 lanes of a wave, word by word.  The record sits in
// mapped host memory; the table is read by the planner at the start of the plan window that follows (zl_plan.h, begin).
static __device__ __forceinline__ void zl_apply_clip_edit(const ZlBatch &A, const ZlClipEdit *ed, int lane)
{
    static_assert(offsetof(ZlClip, slice_pos) == ZL_CLIP_HEAD_BYTES && sizeof(ZlClip) % 16 == 0 && offsetof(ZlClipEdit, c) % 16 == 0, "clip record layout");
    const int clip = ed->clip;
    const int n16 = ed->full ? (int)(sizeof(ZlClip) / 16) : ZL_CLIP_HEAD_BYTES / 16;
    const uint4 *src = reinterpret_cast<const uint4 *>(&ed->c);
    uint4 *dst = reinterpret_cast<uint4 *>(const_cast<ZlClip *>(A.clips) + clip);
    for (int w = lane; w < n16; w += 64) dst[w] = src[w];
}

// workgroups [0, ceil(n_op_ranges / 64)): one lane per voice with operations; the following n_clip_edits workgroups: one edit each
__global__ void zl_k0_apply_ops(const ZlBatch A)
{
    const int opGroups = (A.n_op_ranges + 63) / 64;
    if ((int)blockIdx.x >= opGroups) {
        zl_apply_clip_edit(A, A.clip_edits + ((int)blockIdx.x - opGroups), (int)threadIdx.x);
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n_op_ranges) return;
    const ZlOpRange rg = A.op_ranges[i];
    ZlVoiceState st = A.voices[rg.voice];
    for (int j = 0; j < rg.count; ++j) zl_apply_op(st, A.ops[rg.first + j]);
    A.voices[rg.voice] = st;
}

// ------------------------------------------------------------------------------------------------
// K1c body for one wave: block k of the wave's 64 voices (one lane each).  Blocks with more than two position
// segments are expanded into per-frame control by the whole wave, lanes over frames.
static __device__ __forceinline__ void zl_k1c_block(const ZlBatch &A, ZlAssembler &as, int v, int lane, int k)
{
    int idx0 = 0, base0 = 0, n_active = 0;
    const int nseg = as.block(A, k, idx0, base0, n_active);
    unsigned long long m = __ballot(nseg > 2);
    while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        const int vv = __shfl(v, l, 64), ii = __shfl(idx0, l, 64), bb = __shfl(base0, l, 64), na = __shfl(n_active, l, 64);
        // a slot of the window's control pool for this (block, voice); none left: the block is marked and K2 recomputes it
        int slot = 0;
        if (lane == 0) slot = zl_expand_slot(A, (size_t)k * A.V + vv, ii, bb);
        slot = __shfl(slot, 0, 64);
        if (slot < 0) continue;
        ZlSegStream ss;
        ss.init(A, vv, A.runs[vv]);
        const size_t base = (size_t)slot * (size_t)A.N;
        for (int f = lane; f < A.N; f += 64) {
            float env;
            A.ctl_P[base + f] = zl_expand_frame(ss, A.N, k, ii, bb, f < na ? f : 0, env);
            A.ctl_env[base + f] = env;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1: one lane per voice; the block clocks are staged in LDS so the per-block step of the planner
// makes no dependent global load.  Cost is O(linear runs + events) per voice, not O(blocks) (zl_plan.h).
#define ZL_K1_CLOCKS 256     // (12 KB of LDS: a planner workgroup must fit next to the render kernel's workgroups on a CU -- with 1024 clocks it did not)
__global__ void __launch_bounds__(64) zl_k1_plan(const ZlBatch A, int force_slow)
{
    __shared__ ZlClock s_clk[ZL_K1_CLOCKS];
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = v < A.V;
    ZlPlanner pl;
    if (mine) pl.begin(A, v, force_slow);
    // Only beat-locked loops test the block clocks as they go (:227-241).  A wave without one plans its whole window in one sweep
    // with the clocks where they are, in global memory -- they are read at a voice's first block and in simulated blocks only --
    // instead of stage by stage: an iteration ends at the end of the staged clocks, and a window of 8192 64-frame blocks would cut
    // every run into 8 (planning 96 voices: 0.21 -> 0.0x ms per window).
    const bool needClocks = mine && pl.valid && pl.st.playing && pl.clockMode;
    if (!A.inline_clock && __ballot(needClocks) == 0ull) {
        if (mine) while (pl.t < A.K * A.N) pl.iterate(A, A.K, A.clocks, 0, force_slow);
    } else
    for (int kb = 0; kb < A.K; kb += ZL_K1_CLOCKS) {
        const int nk = (A.K - kb < ZL_K1_CLOCKS) ? A.K - kb : ZL_K1_CLOCKS;
        // every voice of this wave has reached the end of the window (idle, stopped, or a periodic loop whose remaining
        // passes are implied): the rest of the clocks is not needed.  (The workgroup is one wavefront; written as a
        // skipped body rather than a break, which crashes the register allocator of this compiler.)
        const bool lane_done = !mine || !(pl.valid && pl.st.playing) || pl.t >= A.K * A.N;
        if (__ballot(!lane_done) != 0ull) {
            __syncthreads();
            if (A.inline_clock) {
                if (threadIdx.x == 0) s_clk[0] = A.clock0;         // a single real-time block: no clock upload
            } else {
                const uint4 *g = reinterpret_cast<const uint4 *>(A.clocks + kb);
                uint4 *sh = reinterpret_cast<uint4 *>(s_clk);
                for (int i = threadIdx.x; i < nk * (int)(sizeof(ZlClock) / 16); i += blockDim.x) sh[i] = g[i];
            }
            __syncthreads();
            if (mine) {
                // idle voices leave at once (ZlRunList::dead_from); the others run the planner's state machine, whose
                // iterations are the same straight-line code for every lane (zl_plan.h)
                while (pl.t < (kb + nk) * A.N) pl.iterate(A, kb + nk, s_clk, kb, force_slow);
            }
        }
    }
    if (mine) {
        pl.end(A);
        if (A.stats) {
            if (pl.stats.source_bytes)  atomicAdd(&A.stats->source_bytes, pl.stats.source_bytes);
            if (pl.stats.slow_blocks)   atomicAdd(&A.stats->slow_blocks, pl.stats.slow_blocks);
            if (pl.stats.active_frames) atomicAdd(&A.stats->active_frames, pl.stats.active_frames);
        }
    }
    if (A.fuse_assemble) {
        // a single block: K1c's work for the wave's voices is done here (their records were written by this wave)
        __threadfence();
        ZlAssembler as;
        as.begin(A, mine ? v : 0, 0, mine ? A.K : 0);
        for (int k = 0; k < A.K; ++k) zl_k1c_block(A, as, v, threadIdx.x, k);
    }
}

// K1c: segment streams -> per-block plan records.  One lane per voice (coalesced 16-byte plan stores across the
// wave), ZL_K1C_BLOCKS consecutive blocks per wave (one binary search per lane, then a forward walk).  Blocks with
// more than two position segments (the block after a loop restart at a small position crosses ~log2(N) binades)
// are expanded into per-frame control by the whole wave, lanes over frames, so that K2 keeps one pipelined path.
#define ZL_K1C_BLOCKS 8
// Few voices = few waves, each walking its blocks one after the other: engines with fewer than 512 voices give a lane 2
// blocks instead of 8 (four times the waves, a quarter of the latency; the kernel is hidden behind K2 only if it is short)
static inline int zl_k1c_blocks_per_lane(const ZlBatch &A) { return A.V >= 512 ? ZL_K1C_BLOCKS : 2; }
__global__ void __launch_bounds__(64) zl_k1c_assemble(const ZlBatch A, int blocks_per_lane)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x;
    const bool mine = v < A.V;
    const int kbeg = blockIdx.y * blocks_per_lane;
    const int kend = kbeg + blocks_per_lane < A.K ? kbeg + blocks_per_lane : A.K;
    ZlAssembler as;
    as.begin(A, mine ? v : 0, kbeg, mine ? kend : kbeg);
    for (int k = kbeg; k < kend; ++k) zl_k1c_block(A, as, v, lane, k);
}

// ------------------------------------------------------------------------------------------------
// K2: one workgroup = one (bus, mix group, block); one lane = one output frame.  The voices of the
// group are walked sequentially in voice order, so the per-frame sum has the reference's order
// (SamplerSynth.cpp:136-140) and lives in two registers.  Every per-voice record is wave-uniform
// (scalar loads); the only vector memory traffic is the 16-byte two-tap stereo gather and the
// final coalesced store.
static __device__ __forceinline__ int zl_sample_to_peak_int(float x)
{
    // (int)fabsf(131072.f * x), AudioLevels.cpp:356,367, with the oracle's definition of the cases C leaves open: NaN -> 0, 2^31 and above
    // (and infinity) -> INT_MAX.  That is what v_cvt_i32_f32 does by itself (truncation, saturation, NaN -> 0): one instruction with the
    // |.| source modifier instead of two compares and three exec-masked branches per sample
    const float v = 131072.0f * x;
    int r;
    asm("v_cvt_i32_f32_e64 %0, |%1|" : "=v"(r) : "v"(v));
    return r;
}

// Wavefront reductions on the VALU (DPP row shifts / broadcasts, no LDS round trips): quad permutes, row_shr 4 and 8
// leave each 16-lane row's result in its last lane, row_bcast 15 / 31 carry it across the rows; the result of the
// whole wave is in lane 63.  `ident` fills lanes without a source (0 for max over non-negative ints and for sums).
template <int CTRL, int ROWMASK>
static __device__ __forceinline__ int zl_dpp_i(int ident, int v) { return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROWMASK, 0xf, false); }
static __device__ __forceinline__ int zl_wave_max_nonneg(int v)
{
    int t;
    t = zl_dpp_i<0xB1, 0xf>(0, v);  v = t > v ? t : v;             // quad_perm [1,0,3,2]
    t = zl_dpp_i<0x4E, 0xf>(0, v);  v = t > v ? t : v;             // quad_perm [2,3,0,1]
    t = zl_dpp_i<0x114, 0xf>(0, v); v = t > v ? t : v;             // row_shr 4
    t = zl_dpp_i<0x118, 0xf>(0, v); v = t > v ? t : v;             // row_shr 8
    t = zl_dpp_i<0x142, 0xa>(0, v); v = t > v ? t : v;             // row_bcast 15 -> rows 1, 3
    t = zl_dpp_i<0x143, 0xc>(0, v); v = t > v ? t : v;             // row_bcast 31 -> rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}
static __device__ __forceinline__ float zl_wave_sum(float x)
{
    int v = __float_as_int(x), t;
    t = zl_dpp_i<0xB1, 0xf>(0, v);  v = __float_as_int(__int_as_float(v) + __int_as_float(t));
    t = zl_dpp_i<0x4E, 0xf>(0, v);  v = __float_as_int(__int_as_float(v) + __int_as_float(t));
    t = zl_dpp_i<0x114, 0xf>(0, v); v = __float_as_int(__int_as_float(v) + __int_as_float(t));
    t = zl_dpp_i<0x118, 0xf>(0, v); v = __float_as_int(__int_as_float(v) + __int_as_float(t));
    t = zl_dpp_i<0x142, 0xa>(0, v); v = __float_as_int(__int_as_float(v) + __int_as_float(t));
    t = zl_dpp_i<0x143, 0xc>(0, v); v = __float_as_int(__int_as_float(v) + __int_as_float(t));
    return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}

// the four reductions of one bus's fused level scan (two integer peaks, two sums of squares) in ONE instruction sequence: the same
// DPP trees as zl_wave_max_nonneg / zl_wave_sum -- same operands, same order, same bits -- one instruction per step and value (the row
// broadcasts write their rows in place), the four chains interleaved so that no step waits out the DPP read-after-write hazard of its
// own predecessor (2 wait states: three other instructions stand between).  What the compiler makes of the four separate calls is
// ~50 VALU + ~20 s_nop per bus and wave; this is 24 + 4 v_readlane.  (s_nop 4 in front: 5 wait states cover every hazard a DPP
// read can have with the code before the block, which the compiler's hazard recogniser does not look into.)  All 64 lanes active.
#define ZL_DPP4(ctrl) "v_max_i32_dpp %0, %0, %0 " ctrl "\n\tv_max_i32_dpp %1, %1, %1 " ctrl "\n\tv_add_f32_dpp %2, %2, %2 " ctrl "\n\tv_add_f32_dpp %3, %3, %3 " ctrl "\n\t"
static __device__ __forceinline__ void zl_wave_levels4(int &pkL, int &pkR, float &sqL, float &sqR)
{
    int a = pkL, b = pkR, ra, rb;
    float c = sqL, d = sqR, rc, rd;
    asm("s_nop 4\n\t"
        ZL_DPP4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
        ZL_DPP4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1")
        ZL_DPP4("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        ZL_DPP4("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
        ZL_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
        ZL_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 0\n\t"
        "v_readlane_b32 %4, %0, 63\n\tv_readlane_b32 %5, %1, 63\n\tv_readlane_b32 %6, %2, 63\n\tv_readlane_b32 %7, %3, 63"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=s"(ra), "=s"(rb), "=s"(rc), "=s"(rd));
    pkL = ra; pkR = rb; sqL = rc; sqR = rd;
}
#undef ZL_DPP4

// maximum of a wave's non-negative floats (they order like their bit patterns), wave-uniform
static __device__ __forceinline__ float zl_wave_max(float x) { return __int_as_float(zl_wave_max_nonneg(__float_as_int(x))); }

// 16-byte gather of the two interpolation taps of both channels (interleaved stereo), 8-byte aligned
typedef float zl_f4a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef float zl_f2a4 __attribute__((ext_vector_type(2), aligned(4)));

#define ZL_K2_CHUNK 128      // voice records staged in LDS per pass
#define ZL_K2_MAXNB 16       // narrow buses per workgroup (128 voices / the minimum bus width of 8)
#ifndef ZL_K2_U
#define ZL_K2_U     8        // gathers in flight per wavefront
#endif
#ifndef ZL_K2_U_HERMITE
#define ZL_K2_U_HERMITE 4    // voices per chunk with 4-tap interpolation (two 16-byte gathers per voice)
#endif
#ifndef ZL_K2_WAVES_LINEAR
// linear mode, one block per workgroup (the headline shape): waves per SIMD asked of the register allocator.  The LDS cap of
// zl_launch_render holds this kernel at 5 workgroups per CU = 5 waves per SIMD whatever its registers: with 5 the allocator may use
// 96 registers instead of 80 and the kernel gains 0.5 % (headline) to 2 % (64-voice engines) -- profiles/round3_k2_roll_ab.txt
#define ZL_K2_WAVES_LINEAR 5
#endif
#ifndef ZL_K2_PREFETCH_BUS
// narrow buses: issue the next bus's gathers before the finished bus's epilogue (1).  Measured and NOT shipped: 1-2 % slower than 0 on 64- / 96-voice
// engines (profiles/round4_levels_epilogue_ab.txt) -- the epilogue's stores share vmcnt with the gathers, so the first mix waits for all of them
#define ZL_K2_PREFETCH_BUS 0
#endif
#ifndef ZL_K2_MINWAVES
#define ZL_K2_MINWAVES 1      // __launch_bounds__ minimum waves per SIMD (caps the VGPR budget)
#endif

// 16-byte gather of the interpolation taps: interleaved stereo [L0 R0 L1 R1] or mono [x0 x1 . .]
typedef float zl_f4a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float zl_f2a4b __attribute__((ext_vector_type(2), aligned(4)));

struct ZlK2Tap { ZlTaps t; float alpha, env; int flags; };   // flags: 1 act, 2 inb, 4 stereo, 16 wide

// One chunk of U consecutive voices for one lane (= one output frame).  Branch-free per voice so the
// U gathers (and, for CTL chunks, the U per-frame control loads before them) are issued back to back
// and stay in flight together; the voices are then mixed and accumulated in voice order.
// CTL = the chunk contains a block with per-frame control (envelope not in steady sustain, or a
// block expanded by K1c); regular voices of such a chunk read a dummy control word.
// SIMPLE chunks (the steady state): every voice of the chunk plays the whole block from one position segment, in
// sustain, from a stereo source.  No per-voice predicates are needed, which halves the VALU work.
// Packed (l, r) arithmetic of zl_mix_frame for the simple chunks: both channels go through the same expression, so
// the two lanes of a float2 are the reference's l and r sequences, evaluated by v_pk_mul_f32 / v_pk_add_f32 (the
// kernel's VALU budget is ~49 lane-operations per voice-sample at the HBM roofline; un-fused, -ffp-contract=off).
typedef float zl_f2 __attribute__((ext_vector_type(2)));

// unit-step blocks (playback at the source rate inside an exact run): integer part of P0 and the constant fraction
// + the gain products (gain * envelope) * volume of the Hermite mode's whole-sample gain (zl_render.h), one number per voice-block in sustain
struct ZlUnit { int ipos; float alpha; float gpl, gpr; };

// zl_hermite_weights / zl_hermite4 (zl_render.h) with both channels in one register pair: the weights are scalar work shared by
// the channels, the four multiply-adds are v_pk_fma_f32 (one rounding per fused operation, as fmaf)
static __device__ __forceinline__ zl_f2 zl_fma2(zl_f2 a, zl_f2 b, zl_f2 c) { return __builtin_elementwise_fma(a, b, c); }
static __device__ __forceinline__ zl_f2 zl_hermite4_pk(zl_f2 y0, zl_f2 y1, zl_f2 y2, zl_f2 y3, float a)
{
    const ZlHermiteW w = zl_hermite_weights(a);
    return zl_fma2((zl_f2){w.w3, w.w3}, y3, zl_fma2((zl_f2){w.w2, w.w2}, y2, zl_fma2((zl_f2){w.w1, w.w1}, y1, (zl_f2){w.w0, w.w0} * y0)));
}

// gprod (Hermite mode only): (gain * envelope) * volume of both channels, formed by the caller (per voice-block in sustain)
template <uint32_t MODE>
static __device__ __forceinline__ zl_f2 zl_mix_frame_pk(zl_f2 xm, zl_f2 x0, zl_f2 x1, zl_f2 x2, float alpha, bool inb, bool wide,
                                                         zl_f2 gain, float env, float vol, zl_f2 pan, zl_f2 gprod = (zl_f2){0.0f, 0.0f})
{
    const float invAlpha = 1.0f - alpha;                         // :200
    zl_f2 lr;
    if (MODE & ZL_MODE_HERMITE) {
        const zl_f2 h = zl_hermite4_pk(xm, x0, x1, x2, alpha);
        const zl_f2 lin = x0 * invAlpha + x1 * alpha;
        lr = (wide ? h : lin) * gprod;                              // whole-sample gain, the product formed first (zl_render.h)
    } else if (MODE & ZL_MODE_FIX_GAIN) {
        lr = (x0 * invAlpha + x1 * alpha) * gain * env * vol;
    } else {
        lr = x0 * invAlpha + x1 * alpha * gain * env * vol;      // :204-205, quirk Q1
    }
    if (!inb) lr = (zl_f2){0.0f, 0.0f};                          // stereo source: r = l = 0 out of range
    const float mSignal = 0.5f * (lr.x + lr.y);                  // :208
    const float sSignal = lr.x - lr.y;                           // :209
    return pan * mSignal + (zl_f2){sSignal, -sSignal};           // :210-211
}

// ZL_K2_PK_LINEAR / ZL_K2_PK_HERMITE: evaluate the simple chunks with the packed float2 arithmetic above (1) or with
// the scalar zl_mix_frame (0).  Same results bit for bit; the choice is a measured one.
#ifndef ZL_K2_PK_LINEAR
#define ZL_K2_PK_LINEAR 0
#endif
#ifndef ZL_K2_PK_HERMITE
#define ZL_K2_PK_HERMITE 1
#endif

// the gathers of one simple chunk, in flight: issued by zl_k2_simple_issue, consumed by zl_k2_simple_mix
template <bool HERM, int U> struct ZlSimpleTaps { zl_f4a4 d[U], e[HERM ? U : 1]; float alpha[U]; int widem; };

// first half of a simple chunk: the U positions and the U gathers, issued back to back.
// INT ("interior"): every frame of the block lies inside the source for every voice of the chunk (checked per voice at
// staging from the block's first and last position) -- no bounds guard, no tap selects, no read of the duration.
template <uint32_t MODE, bool SEG2, bool UNIT, bool INT, int U>
static __device__ __forceinline__ void zl_k2_simple_issue(const ZlBatch &A, const ZlBlockPlan *s_plan, const ZlVoiceConst *s_vc, const ZlUnit *s_unit,
                                                           int c0, int f, double fd, ZlSimpleTaps<(MODE & ZL_MODE_HERMITE) != 0, U> &T)
{
    constexpr bool HERM = (MODE & ZL_MODE_HERMITE) != 0;
    T.widem = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = c0 + u;
        double P;
        if (SEG2) {                                               // a binade crossing or loop restart inside the block
            const bool seg1 = f >= s_plan[i].n1;
            P = fma((double)(f - (seg1 ? s_plan[i].n1 : 0)), seg1 ? s_plan[i].step1 : s_plan[i].step, seg1 ? s_plan[i].P1 : s_plan[i].P0);
        } else if (!UNIT) {
            P = fma(fd, s_plan[i].step, s_plan[i].P0);            // exact, see zl_plan.h
        }
        // :198-199 -- P >= 0 here, so pos = floor(P) and alpha = (float)(P - pos) = (float)fract(P), both exact
        int pos;
        if (UNIT) {
            // step == 1 inside an exact run: every frame has the fractional part of P0 and the integer part moves by f
            pos = s_unit[i].ipos + f;
            T.alpha[u] = s_unit[i].alpha;
        } else {
            pos = (int)P;
            T.alpha[u] = (float)__builtin_amdgcn_fract(P);
        }
        const int dur = INT ? 0 : s_vc[i].sample_duration;
        const bool inb = INT || dur > pos;                        // :204 guard (Q5)
        // out of range: gather the zero padding behind the source (8 frames, written by zl_k_interleave), so that
        // l = r = 0 falls out of the arithmetic (0 * finite = +-0, +0 + -0 = +0) without a select per channel;
        // voices with a non-finite gain are not "simple"
        int p = inb ? pos : dur + 1;
        const uint64_t so = s_vc[i].src_offset;
        const float *src = A.arena + (((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(so >> 32)) << 32)
                                      | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)so));
        if (HERM) {
            // taps pos-1 .. pos+2 as two 16-byte loads; at the source edges (not wide) the pair pos, pos+1 comes first
            const bool wide = INT || (inb && pos >= 1 && pos + 2 <= dur);
            p -= wide ? 1 : 0;
            // (uniform 64-bit base + 32-bit byte offset: the scalar-base form of global_load, no 64-bit VALU address)
            const uint32_t ob = (uint32_t)p << 3;
            T.d[u] = *reinterpret_cast<const zl_f4a4 *>(reinterpret_cast<const char *>(src) + ob);
            T.e[u] = *reinterpret_cast<const zl_f4a4 *>(reinterpret_cast<const char *>(src) + ob + 16u);   // inside the arena padding at the end
            T.widem |= wide ? (1 << u) : 0;
        } else {
            T.d[u] = *reinterpret_cast<const zl_f4a4 *>(reinterpret_cast<const char *>(src) + ((uint32_t)p << 3));
        }
    }
}

// second half: the voices mixed and accumulated in voice order
template <uint32_t MODE, bool INT, int U>
static __device__ __forceinline__ void zl_k2_simple_mix(const ZlBatch &A, const ZlBlockPlan *s_plan, const ZlVoiceConst *s_vc, const ZlUnit *s_unit,
                                                         int c0, int vfirst, bool wantPeak, const ZlSimpleTaps<(MODE & ZL_MODE_HERMITE) != 0, U> &T,
                                                         float &accL, float &accR)
{
    constexpr bool HERM = (MODE & ZL_MODE_HERMITE) != 0;
    constexpr bool PK = HERM ? (ZL_K2_PK_HERMITE != 0) : (ZL_K2_PK_LINEAR != 0);
    zl_f2 acc = {accL, accR};
    // every lane of the wave has all four Hermite taps inside its source (true except in the blocks at a loop's ends):
    // the wave-uniform fast form needs no tap selects and no linear alternative
    const bool allWide = HERM && PK && (INT || __all(T.widem == (1 << U) - 1));
    if (allWide) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = c0 + u;
            const zl_f2 o = zl_mix_frame_pk<MODE>((zl_f2){T.d[u].x, T.d[u].y}, (zl_f2){T.d[u].z, T.d[u].w}, (zl_f2){T.e[HERM ? u : 0].x, T.e[HERM ? u : 0].y},
                                                  (zl_f2){T.e[HERM ? u : 0].z, T.e[HERM ? u : 0].w}, T.alpha[u], true, true,
                                                  (zl_f2){s_vc[i].lgain, s_vc[i].rgain}, s_plan[i].env, s_vc[i].clip_volume,
                                                  (zl_f2){s_vc[i].lpan, s_vc[i].rpan}, (zl_f2){s_unit[i].gpl, s_unit[i].gpr});
            acc += o;
            if (wantPeak) {
                const float ng = o.x + o.y;
                float pk = ng > 0.0f ? ng : 0.0f;
                pk = zl_wave_max(pk);
                if ((threadIdx.x & 63) == 0 && pk > 0.0f) atomicMax(&A.reports[vfirst + i].peak_bits, __float_as_uint(pk));
            }
        }
        accL = acc.x; accR = acc.y;
        return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = c0 + u;
        const bool wide = INT || ((T.widem >> u) & 1);
        float l, r;
        if (PK) {
            zl_f2 xm, x0, x1, x2;
            if (HERM) {
                xm = (zl_f2){T.d[u].x, T.d[u].y};
                x0 = wide ? (zl_f2){T.d[u].z, T.d[u].w} : (zl_f2){T.d[u].x, T.d[u].y};
                x1 = wide ? (zl_f2){T.e[u].x, T.e[u].y} : (zl_f2){T.d[u].z, T.d[u].w};
                x2 = (zl_f2){T.e[u].z, T.e[u].w};
            } else {
                x0 = (zl_f2){T.d[u].x, T.d[u].y}; x1 = (zl_f2){T.d[u].z, T.d[u].w};
                xm = x0; x2 = x1;
            }
            const zl_f2 o = zl_mix_frame_pk<MODE>(xm, x0, x1, x2, T.alpha[u], true, wide,
                                                  (zl_f2){s_vc[i].lgain, s_vc[i].rgain}, s_plan[i].env, s_vc[i].clip_volume,
                                                  (zl_f2){s_vc[i].lpan, s_vc[i].rpan}, (zl_f2){s_unit[i].gpl, s_unit[i].gpr});
            acc += o;                                             // :218-221 (index shift applied at the store)
            l = o.x; r = o.y;
        } else {
            ZlTaps t;
            if (HERM) {
                t.xml = T.d[u].x; t.xmr = T.d[u].y;
                t.x0l = wide ? T.d[u].z : T.d[u].x; t.x0r = wide ? T.d[u].w : T.d[u].y;
                t.x1l = wide ? T.e[u].x : T.d[u].z; t.x1r = wide ? T.e[u].y : T.d[u].w;
                t.x2l = T.e[u].z; t.x2r = T.e[u].w;
            } else {
                t.x0l = T.d[u].x; t.x0r = T.d[u].y; t.x1l = T.d[u].z; t.x1r = T.d[u].w;
                t.xml = t.xmr = t.x2l = t.x2r = 0.0f;
            }
            zl_mix_frame<MODE>(t, T.alpha[u], true, wide, true, s_vc[i].lgain, s_vc[i].rgain, s_plan[i].env,
                               s_vc[i].clip_volume, s_vc[i].lpan, s_vc[i].rpan, l, r);
            accL += l; accR += r;                                 // :218-221 (index shift applied at the store)
        }
        if (wantPeak) {                                           // :213-216, signed peak from 0 (Q6)
            const float ng = l + r;
            float pk = ng > 0.0f ? ng : 0.0f;
            pk = zl_wave_max(pk);
            if ((threadIdx.x & 63) == 0 && pk > 0.0f) atomicMax(&A.reports[vfirst + i].peak_bits, __float_as_uint(pk));
        }
    }
    if (PK) { accL = acc.x; accR = acc.y; }
}

template <uint32_t MODE, bool SEG2, bool UNIT, bool INT, int U>
static __device__ __forceinline__ void zl_k2_chunk_simple(const ZlBatch &A, const ZlBlockPlan *s_plan, const ZlVoiceConst *s_vc, const ZlUnit *s_unit,
                                                           int c0, int vfirst, int f, double fd, bool wantPeak, float &accL, float &accR)
{
    ZlSimpleTaps<(MODE & ZL_MODE_HERMITE) != 0, U> T;
    zl_k2_simple_issue<MODE, SEG2, UNIT, INT, U>(A, s_plan, s_vc, s_unit, c0, f, fd, T);
    zl_k2_simple_mix<MODE, INT, U>(A, s_plan, s_vc, s_unit, c0, vfirst, wantPeak, T, accL, accR);
}

// UNIT + INT chunks with SHARED taps (ZL_K2_UNIT_SHARE=1): at the playback rate inside an exact run, lane f + 1's first tap IS lane
// f's second (pos = ipos + f), so every lane gathers only its OWN frame -- 8 bytes instead of 16 (16 instead of 32 with four
// taps) -- and takes its neighbours' from their registers: v_mov_b32_dpp wave_shl:1 / wave_shr:1 (lane i reads lane i + 1 / i - 1;
// the lane at the end of the wave keeps the `old` operand).  The taps a wave does not hold (frame + 1 of lane 63; frame - 1 of
// lane 0 and frames + 1, + 2 of lane 63 with four taps) come from ONE extra load executed by those lanes only.  Half (a quarter)
// of the bytes through the texture addresser, the same operands in the same order: bit-identical results.
#ifndef ZL_K2_UNIT_SHARE
#define ZL_K2_UNIT_SHARE 0
#endif
#define ZL_DPP_WAVE_SHL1 0x130
#define ZL_DPP_WAVE_SHR1 0x138
static __device__ __forceinline__ float zl_dpp_next(float old, float v)    // lane i <- lane i + 1; lane 63 <- old
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), ZL_DPP_WAVE_SHL1, 0xf, 0xf, false));
}
static __device__ __forceinline__ float zl_dpp_prev(float old, float v)    // lane i <- lane i - 1; lane 0 <- old
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), ZL_DPP_WAVE_SHR1, 0xf, 0xf, false));
}

template <uint32_t MODE, int U>
static __device__ __forceinline__ void zl_k2_chunk_unit_shared(const ZlBatch &A, const ZlBlockPlan *s_plan, const ZlVoiceConst *s_vc, const ZlUnit *s_unit,
                                                                int c0, int vfirst, int f, bool wantPeak, float &accL, float &accR)
{
    constexpr bool HERM = (MODE & ZL_MODE_HERMITE) != 0;
    const int lane = (int)(threadIdx.x & 63u);
    zl_f2a4b b[U];                                                // the lane's own frame (L, R) of every voice
    zl_f2a4b ex2[HERM ? 1 : U];                                   // linear: frame + 1 of lane 63
    zl_f4a4  ex4[HERM ? U : 1];                                   // four taps: frames - 1, 0 of lane 0; frames + 1, + 2 of lane 63
    const char *src[U];
    uint32_t ob[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = c0 + u;
        const int pos = s_unit[i].ipos + f;
        const uint64_t so = s_vc[i].src_offset;
        src[u] = reinterpret_cast<const char *>(A.arena + (((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(so >> 32)) << 32)
                                                            | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)so)));
        ob[u] = (uint32_t)pos << 3;
        b[u] = *reinterpret_cast<const zl_f2a4b *>(src[u] + ob[u]);
    }
    if (HERM) {
        if (lane == 0 || lane == 63) {
#pragma unroll
            for (int u = 0; u < U; ++u) ex4[u] = *reinterpret_cast<const zl_f4a4 *>(src[u] + (lane == 0 ? ob[u] - 8u : ob[u] + 8u));
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) ex4[u] = (zl_f4a4){0.0f, 0.0f, 0.0f, 0.0f};
        }
    } else {
        if (lane == 63) {
#pragma unroll
            for (int u = 0; u < U; ++u) ex2[u] = *reinterpret_cast<const zl_f2a4b *>(src[u] + ob[u] + 8u);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) ex2[u] = (zl_f2a4b){0.0f, 0.0f};
        }
    }
    zl_f2 acc = {accL, accR};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = c0 + u;
        const float alpha = s_unit[i].alpha;
        float l, r;
        if (HERM) {
            const zl_f2 x0 = {b[u].x, b[u].y};
            const zl_f2 xm = {zl_dpp_prev(ex4[u].x, b[u].x), zl_dpp_prev(ex4[u].y, b[u].y)};
            const zl_f2 x1 = {zl_dpp_next(ex4[u].x, b[u].x), zl_dpp_next(ex4[u].y, b[u].y)};
            const zl_f2 x2 = {zl_dpp_next(ex4[u].z, x1.x), zl_dpp_next(ex4[u].w, x1.y)};
            const zl_f2 o = zl_mix_frame_pk<MODE>(xm, x0, x1, x2, alpha, true, true, (zl_f2){s_vc[i].lgain, s_vc[i].rgain}, s_plan[i].env, s_vc[i].clip_volume,
                                                  (zl_f2){s_vc[i].lpan, s_vc[i].rpan}, (zl_f2){s_unit[i].gpl, s_unit[i].gpr});
            acc += o;
            l = o.x; r = o.y;
        } else {
            ZlTaps t;
            t.x0l = b[u].x; t.x0r = b[u].y;
            t.x1l = zl_dpp_next(ex2[u].x, b[u].x); t.x1r = zl_dpp_next(ex2[u].y, b[u].y);
            t.xml = t.xmr = t.x2l = t.x2r = 0.0f;
            zl_mix_frame<MODE>(t, alpha, true, true, true, s_vc[i].lgain, s_vc[i].rgain, s_plan[i].env, s_vc[i].clip_volume, s_vc[i].lpan, s_vc[i].rpan, l, r);
            acc.x += l; acc.y += r;
        }
        if (wantPeak) {                                           // :213-216, signed peak from 0 (Q6)
            const float ng = l + r;
            float pk = ng > 0.0f ? ng : 0.0f;
            pk = zl_wave_max(pk);
            if ((threadIdx.x & 63) == 0 && pk > 0.0f) atomicMax(&A.reports[vfirst + i].peak_bits, __float_as_uint(pk));
        }
    }
    accL = acc.x; accR = acc.y;
}

// The same for chunks of mono sources: an 8-byte gather [x0 x1] (16 bytes [x-1 x0 x1 x2] for Hermite), r = l (:205, Q4).
template <uint32_t MODE, bool SEG2, bool UNIT, bool INT, int U>
static __device__ __forceinline__ void zl_k2_chunk_simple_mono(const ZlBatch &A, const ZlBlockPlan *s_plan, const ZlVoiceConst *s_vc, const ZlUnit *s_unit,
                                                                int c0, int vfirst, int f, double fd, bool wantPeak, float &accL, float &accR)
{
    constexpr bool HERM = (MODE & ZL_MODE_HERMITE) != 0;
    zl_f4a4 d4[HERM ? U : 1];
    zl_f2a4b d2[HERM ? 1 : U];
    float alpha[U];
    int   widem = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = c0 + u;
        double P;
        if (SEG2) {
            const bool seg1 = f >= s_plan[i].n1;
            P = fma((double)(f - (seg1 ? s_plan[i].n1 : 0)), seg1 ? s_plan[i].step1 : s_plan[i].step, seg1 ? s_plan[i].P1 : s_plan[i].P0);
        } else if (!UNIT) {
            P = fma(fd, s_plan[i].step, s_plan[i].P0);            // exact, see zl_plan.h
        }
        int pos;                                                  // :198-199 (P >= 0)
        if (UNIT) { pos = s_unit[i].ipos + f; alpha[u] = s_unit[i].alpha; }
        else      { pos = (int)P; alpha[u] = (float)__builtin_amdgcn_fract(P); }
        const int dur = INT ? 0 : s_vc[i].sample_duration;
        const bool inb = INT || dur > pos;                        // :204 guard (Q5)
        int p = inb ? pos : dur + 1;                              // out of range: the zero padding behind the source
        const uint64_t so = s_vc[i].src_offset;
        const float *src = A.arena + (((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(so >> 32)) << 32)
                                      | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)so));
        if (HERM) {
            const bool wide = INT || (inb && pos >= 1 && pos + 2 <= dur);
            p -= wide ? 1 : 0;
            d4[u] = *reinterpret_cast<const zl_f4a4 *>(reinterpret_cast<const char *>(src) + ((uint32_t)p << 2));
            widem |= wide ? (1 << u) : 0;
        } else {
            d2[u] = *reinterpret_cast<const zl_f2a4b *>(reinterpret_cast<const char *>(src) + ((uint32_t)p << 2));
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = c0 + u;
        const bool wide = INT || ((widem >> u) & 1);
        ZlTaps t;
        t.x0r = t.x1r = t.xmr = t.x2r = 0.0f;
        if (HERM) {
            t.xml = d4[u].x;
            t.x0l = wide ? d4[u].y : d4[u].x;
            t.x1l = wide ? d4[u].z : d4[u].y;
            t.x2l = d4[u].w;
        } else {
            t.x0l = d2[u].x; t.x1l = d2[u].y;
            t.xml = t.x2l = 0.0f;
        }
        float l, r;
        zl_mix_frame<MODE>(t, alpha[u], true, wide, false, s_vc[i].lgain, s_vc[i].rgain, s_plan[i].env,
                           s_vc[i].clip_volume, s_vc[i].lpan, s_vc[i].rpan, l, r);
        accL += l; accR += r;                                     // :218-221 (index shift applied at the store)
        if (wantPeak) {                                           // :213-216, signed peak from 0 (Q6)
            const float ng = l + r;
            float pk = ng > 0.0f ? ng : 0.0f;
            pk = zl_wave_max(pk);
            if ((threadIdx.x & 63) == 0 && pk > 0.0f) atomicMax(&A.reports[vfirst + i].peak_bits, __float_as_uint(pk));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-staged source windows (north_star: "one wavefront per voice-tile with source samples staged in LDS").
// For a (voice, block) whose frames advance along one exact line, the 64 frames of a wavefront read the contiguous source
// window [pos(first) - tb, pos(last) + ta] (tb / ta = taps before / after: 0 / 1 linear, 1 / 2 Hermite).  The wave fetches
// that window ONCE with a single LDS-DMA instruction (global_load_lds_dwordx4: lane i moves bytes [16 i, 16 i + 16) of the
// window straight into LDS, no register in between; lanes past the window's end are masked off) into a slot of its own
// ring, ZL_ST_D voices deep, and its lanes then read their taps from LDS.  Against the register gather this moves each
// source byte through the texture path once instead of 2x (linear) or 4x (Hermite: two overlapping 16-byte gathers per
// lane), and keeps ZL_ST_D - ZL_ST_U .. ZL_ST_D voices of every wave in flight without holding them in VGPRs.
// Windows are private to a wave: no barrier, the wave's own counted s_waitcnt vmcnt orders its reads behind its DMA.
#ifndef ZL_ST_SLOT
#define ZL_ST_SLOT 1024      // bytes per ring slot = one LDS-DMA piece: windows of up to 128 stereo frames (ratio <= 1.95)
#endif
#ifndef ZL_ST_D
#define ZL_ST_D 6            // ring depth in voices (slots per wave); measured: 6 > 10 > 16 (occupancy beats depth, profiles/round2_b_*)
#endif
#ifndef ZL_ST_U
#define ZL_ST_U 2            // voices per staged compute step
#endif
#ifndef ZL_ST_CHUNK
#define ZL_ST_CHUNK 64       // voice records per pass in the staged variant (64 or 128)
#endif
struct ZlWin { int a; int n16; };   // window of one (voice, wave): first source frame (even: 16-byte aligned), 16-byte pieces; n16 == 0: not staged
extern __shared__ __attribute__((aligned(16))) char zl_dyn_lds[];

static __device__ __forceinline__ uint32_t zl_lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}

// One LDS-DMA piece.  Lane address gaddr (the window's base + this lane's 16 bytes) -> LDS at dst (wave-uniform LDS byte
// address, in M0) + lane * 16, for the lanes with lane16 < nbytes.  The compiler neither sees the load nor counts it: the
// caller waits with zl_wait_vm<N>().  s_waitcnt lgkmcnt(0) first: every LDS read this wave has issued (the previous tenant of
// the slot among them) has returned before the DMA may overwrite it.  M0 is compiler-reserved: saved and restored.
static __device__ __forceinline__ void zl_glds_piece(const char *gaddr, uint32_t lane16, uint32_t nbytes, uint32_t dst)
{
#ifdef ZL_ST_DIAG_NODMA      // timing-only diagnostic build: no DMA (the taps read whatever the ring holds)
    return;
#endif
    unsigned keep; unsigned long long save;
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %5\n\t"
                 "v_cmp_lt_u32 vcc, %3, %4\n\t"
                 "s_and_saveexec_b64 %1, vcc\n\t"
                 "global_load_lds_dwordx4 %2, off\n\t"
                 "s_mov_b64 exec, %1\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(save)
                 : "v"(gaddr), "v"(lane16), "v"(nbytes), "s"(dst)
                 : "memory", "vcc", "scc");
}
template <int N> static __device__ __forceinline__ void zl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// What a lane needs of a staged voice besides its taps: wave-uniform values, read from the staged records ONE step ahead
// (they do not depend on the DMA), so that a step's only exposed LDS round trip is the tap read itself.
struct ZlStParams {
    double P0, step;
    int    a;                 // first source frame of this wave's window
    float  env, vol;
    zl_f2  gain, pan, gprod;  // (l, r) pairs
};
static __device__ __forceinline__ ZlStParams zl_st_params(const ZlBlockPlan *s_plan, const ZlVoiceConst *s_vc, const ZlWin *s_win, int i)
{
    ZlStParams q;
    q.P0 = s_plan[i].P0; q.step = s_plan[i].step; q.env = s_plan[i].env;
    q.a = s_win[i * 4].a;
    q.gain = (zl_f2){s_vc[i].lgain, s_vc[i].rgain}; q.vol = s_vc[i].clip_volume; q.pan = (zl_f2){s_vc[i].lpan, s_vc[i].rpan};
    q.gprod = (q.gain * q.env) * q.vol;                           // Hermite mode's whole-sample gain (zl_render.h)
    return q;
}
// The DMA of voice j of the pass, as per-lane values (no scalar round trip): lane address and bytes of the window.  Voices
// that are not staged fetch 16 bytes from the start of the arena (always there) by one lane, so that the wave's count of
// outstanding DMAs stays uniform.
struct ZlStDma { const char *g; uint32_t nbytes; };
static __device__ __forceinline__ ZlStDma zl_st_dma(const ZlBatch &A, const ZlVoiceConst *s_vc, const ZlWin *s_win, int j, uint32_t lane16)
{
    const ZlWin wn = s_win[j * 4];
    const uint64_t so = s_vc[j].src_offset;
    ZlStDma d;
    d.g = reinterpret_cast<const char *>(A.arena) + (wn.n16 ? (so << 2) + ((uint64_t)(uint32_t)wn.a << 3) + lane16 : 0ull);
    d.nbytes = wn.n16 ? (uint32_t)wn.n16 << 4 : 16u;
    return d;
}

// The taps of one staged voice for this lane, from ring slot `slot` of the wave.
struct ZlStTaps { zl_f2 xm, x0, x1, x2; float alpha; };
template <uint32_t MODE>
static __device__ __forceinline__ ZlStTaps zl_st_taps(const ZlStParams &q, const char *ring, int slot, double fd)
{
    constexpr bool HERM = (MODE & ZL_MODE_HERMITE) != 0;
    ZlStTaps t;
    const double P = fma(fd, q.step, q.P0);                       // exact, see zl_plan.h
    const int pos = (int)P;                                       // :198-199 (P >= 0)
    t.alpha = (float)__builtin_amdgcn_fract(P);
    const char *p = ring + slot * ZL_ST_SLOT + ((pos - (HERM ? 1 : 0) - q.a) << 3);
    if (HERM) {
        t.xm = *reinterpret_cast<const zl_f2 *>(p);      t.x0 = *reinterpret_cast<const zl_f2 *>(p + 8);
        t.x1 = *reinterpret_cast<const zl_f2 *>(p + 16); t.x2 = *reinterpret_cast<const zl_f2 *>(p + 24);
    } else {
        t.x0 = *reinterpret_cast<const zl_f2 *>(p); t.x1 = *reinterpret_cast<const zl_f2 *>(p + 8);
        t.xm = t.x0; t.x2 = t.x1;
    }
    return t;
}
// :200-221 of one staged voice ("interior": all taps inside the source; stereo, whole block, sustain) -- the same arithmetic
// as the interior variant of zl_k2_chunk_simple.
template <uint32_t MODE>
static __device__ __forceinline__ void zl_st_mix(const ZlBatch &A, const ZlStParams &q, const ZlStTaps &t, int voice, bool wantPeak, zl_f2 &acc)
{
    float l, r;
    if (MODE & ZL_MODE_HERMITE) {
        const zl_f2 o = zl_mix_frame_pk<MODE>(t.xm, t.x0, t.x1, t.x2, t.alpha, true, true, q.gain, q.env, q.vol, q.pan, q.gprod);
        acc += o;
        l = o.x; r = o.y;
    } else {
        ZlTaps tt;
        tt.x0l = t.x0.x; tt.x0r = t.x0.y; tt.x1l = t.x1.x; tt.x1r = t.x1.y;
        tt.xml = tt.xmr = tt.x2l = tt.x2r = 0.0f;
        zl_mix_frame<MODE>(tt, t.alpha, true, false, true, q.gain.x, q.gain.y, q.env, q.vol, q.pan.x, q.pan.y, l, r);
        acc.x += l; acc.y += r;                                   // :218-221 (index shift applied at the store)
    }
    if (wantPeak) {                                               // :213-216, signed peak from 0 (Q6)
        const float ng = l + r;
        float pk = ng > 0.0f ? ng : 0.0f;
        pk = zl_wave_max(pk);
        if ((threadIdx.x & 63) == 0 && pk > 0.0f) atomicMax(&A.reports[voice].peak_bits, __float_as_uint(pk));
    }
}

template <uint32_t MODE, bool CTL, int U>
static __device__ __forceinline__ void zl_k2_chunk(const ZlBatch &A, const ZlBlockPlan *s_plan, const ZlVoiceConst *s_vc,
                                                    const int *s_cls, int c0, size_t pbase, int vfirst, int f, bool wantPeak,
                                                    float &accL, float &accR)
{
    const int N = A.N;
    ZlK2Tap tap[U];
    double Pc[U];
    float  Ec[U];
    if (CTL) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = c0 + u;
            const bool ctl = s_cls[i] & 2;
            const bool act = (s_cls[i] & 1) && f < s_plan[i].n_active;
            if (__builtin_expect(ctl && (s_plan[i].flags & (ZL_PLAN_NOSLOT_SIM | ZL_PLAN_NOSLOT_EXPAND)), 0)) {
                // the window's control pool was exhausted when this block was planned: recompute its control (zl_plan.h)
                zl_slow_control(A, s_plan[i], vfirst + i, (int)(pbase / (size_t)A.V), act ? f : 0, Pc[u], Ec[u]);
            } else {
                const size_t off = ctl ? (size_t)(int)s_plan[i].step * (size_t)N + (act ? f : 0) : 0;   // the block's slot; regular voices: word 0 (always valid)
                Pc[u] = A.ctl_P[off];
                Ec[u] = A.ctl_env[off];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = c0 + u;
        const int cls = s_cls[i];
        const bool on = cls & 1;
        const bool act = on && f < s_plan[i].n_active;
        const int fe = act ? f : 0;                               // lanes past the voice's end gather a safe address
        const bool seg1 = fe >= s_plan[i].n1;                     // second linear segment of the block
        const double P0 = seg1 ? s_plan[i].P1 : s_plan[i].P0;
        const double st = seg1 ? s_plan[i].step1 : s_plan[i].step;
        double P = fma((double)(fe - (seg1 ? s_plan[i].n1 : 0)), st, P0);            // exact, see zl_plan.h
        tap[u].env = (float)fma((double)(fe - (seg1 ? s_plan[i].n1 : 0)), (double)(seg1 ? s_plan[i].estep1 : s_plan[i].estep0),
                                (double)(seg1 ? s_plan[i].E1 : s_plan[i].env));    // exact fp32 envelope ramp (0 slope in sustain)
        if (CTL) { if (cls & 2) P = Pc[u]; }
        int pos;
        zl_split_position(P, pos, tap[u].alpha);                  // :198-199
        const int dur = s_vc[i].sample_duration;
        const bool stereo = s_vc[i].channels > 1;
        const bool inb = on && (dur > pos) && (pos >= 0);         // :204 guard (Q5)
        const bool wide = (MODE & ZL_MODE_HERMITE) && inb && (pos - 1 >= 0) && (pos + 2 <= dur);
        const int p = inb ? pos : 0;
        const uint64_t so = on ? s_vc[i].src_offset : 0;
        const float *src = A.arena + (((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(so >> 32)) << 32)
                                      | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)so));
        const size_t e0 = stereo ? 2 * (size_t)p : (size_t)p;
        const zl_f4a4 d = *reinterpret_cast<const zl_f4a4 *>(src + e0);
        tap[u].t.x0l = d.x;
        tap[u].t.x0r = stereo ? d.y : 0.0f;
        tap[u].t.x1l = stereo ? d.z : d.y;
        tap[u].t.x1r = stereo ? d.w : 0.0f;
        if (MODE & ZL_MODE_HERMITE) {
            const size_t em = stereo ? 2 * (size_t)(wide ? p - 1 : p) : (size_t)(wide ? p - 1 : p);
            const size_t e2 = stereo ? 2 * (size_t)(wide ? p + 2 : p) : (size_t)(wide ? p + 2 : p);
            const zl_f2a4b m = *reinterpret_cast<const zl_f2a4b *>(src + em);
            const zl_f2a4b n = *reinterpret_cast<const zl_f2a4b *>(src + e2);
            tap[u].t.xml = m.x; tap[u].t.xmr = stereo ? m.y : 0.0f;
            tap[u].t.x2l = n.x; tap[u].t.x2r = stereo ? n.y : 0.0f;
        } else {
            tap[u].t.xml = tap[u].t.xmr = tap[u].t.x2l = tap[u].t.x2r = 0.0f;
        }
        tap[u].flags = (act ? 1 : 0) | (inb ? 2 : 0) | (stereo ? 4 : 0) | (wide ? 16 : 0);
        if (A.trace && on) A.pos_trace[(pbase + i) * (size_t)N + f] = act ? pos : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = c0 + u;
        const int tf = tap[u].flags;
        const bool act = tf & 1;
        float env = tap[u].env;
        if (CTL) { if (s_cls[i] & 2) env = Ec[u]; }
        float l, r;
        zl_mix_frame<MODE>(tap[u].t, tap[u].alpha, tf & 2, tf & 16, tf & 4, s_vc[i].lgain, s_vc[i].rgain, env,
                           s_vc[i].clip_volume, s_vc[i].lpan, s_vc[i].rpan, l, r);
        if (act) { accL += l; accR += r; }                        // :218-221 (index shift applied at the store)
        if (wantPeak) {                                           // :213-216, signed peak from 0 (Q6)
            const float ng = l + r;
            float pk = (act && ng > 0.0f) ? ng : 0.0f;
            pk = zl_wave_max(pk);
            if ((threadIdx.x & 63) == 0 && pk > 0.0f) atomicMax(&A.reports[vfirst + i].peak_bits, __float_as_uint(pk));
        }
    }
}

// BPW = blocks per workgroup.  Blocks shorter than 256 frames are rendered 256 / N at a time (BPW = 2 or 4: the 256
// threads are BPW groups of N frames), so the per-workgroup fixed costs -- launch, staging of the voice records -- are
// paid once per 256 frames whatever the block size is.  A wavefront never straddles two blocks (BPW > 1 only for N = 64 or 128).
// ST = the variant with LDS-staged source windows (zl_st_* above): its occupancy is set by the ring in LDS (2-3
// workgroups per CU), so it may take the registers of 3 waves per SIMD.
// (the body is a device function so that the persistent real-time kernel below can run it too: bx / by / bz / gdx / gdy stand for
// blockIdx.x / .y / .z and gridDim.x / .y of the batch launch)
// REPORTS: the workgroups of a call's last block also publish its per-voice reports (batch launches; the resident kernel has its own)
template <uint32_t MODE, int BPW, bool ST, bool REPORTS = true>
static __device__ __forceinline__ void zl_k2_body(const ZlBatch &A, const unsigned bx, const unsigned by, const unsigned bz, const unsigned gdx, const unsigned gdy)
{
    constexpr int U = (MODE & ZL_MODE_HERMITE) ? ZL_K2_U_HERMITE : ZL_K2_U;
    // voice records staged per pass (the staged variant trades them for ring space; four 64-frame blocks per workgroup stage
    // 4 x the block records: 64 voices per pass keep the workgroup at 25 KB of LDS -- 5 workgroups per CU instead of 3 at 50 KB)
    constexpr int CH = ST ? ZL_ST_CHUNK : (BPW == 4 ? 64 : ZL_K2_CHUNK);
    __shared__ ZlWin s_win_[ST ? CH * 4 : 1];             // [voice][wave of the workgroup]
    __shared__ unsigned long long s_stmask_[ST ? BPW * (CH / 64) : 1];   // per block: which voices of the pass are staged
    __shared__ ZlBlockPlan  s_plan_[BPW][CH];
    __shared__ ZlVoiceConst s_vc[CH];
    __shared__ int s_cls_[BPW][CH];                   // per voice: 1 = plays this block, 2 = per-frame control
    __shared__ int s_chunk_[BPW][CH / U];             // class of each chunk of U voices
    __shared__ ZlUnit s_unit_[BPW][CH];
    __shared__ int   s_pk[2][ZL_K2_MAXNB][4];         // fused level scan: per-wave partial results of each of the workgroup's buses
    __shared__ float s_sq[2][ZL_K2_MAXNB][4];

    const int N = A.N, V = A.V;
    const int blk = (BPW > 1) ? (int)threadIdx.x / N : 0;          // which of the workgroup's blocks this lane renders
    const int f = (BPW > 1) ? (int)threadIdx.x - blk * N : (int)(bx * blockDim.x + threadIdx.x);   // frame inside the block
    // XCD-aware block order: workgroups are dealt to the 8 XCDs round-robin in launch order, so launch slot y runs on XCD
    // y % 8; giving XCD x the contiguous blocks [x * KY / 8, (x + 1) * KY / 8) lets neighbouring blocks (which share the
    // cache line at their common edge of every source) meet in the same L2
    // Split tail (narrow buses, set by zl_launch_render): a launch ends when its last workgroups do, and those start one workgroup lifetime before
    // the end whatever the machine does meanwhile -- the launch's last blocks are therefore rendered by tail_split workgroups each, a few buses per
    // workgroup: a quarter of the lifetime, a quarter of the time the machine drains.  Slots [0, tail_from) are the blocks in front of them.
    const bool tailed = BPW == 1 && !ST && REPORTS && A.tail_from > 0;
    const bool inTail = tailed && (int)by >= A.tail_from;
    const int tj = inTail ? (int)by - A.tail_from : 0;
    const int ky = tailed ? A.tail_from : (int)gdy, xq = ky >> 3, xr = ky & 7, xx = (int)(by & 7u);
    const int yb = inTail ? A.tail_from + tj / A.tail_split
                          : xx * xq + (xx < xr ? xx : xr) + (int)(by >> 3);      // a bijection of [0, ky) for every ky
    const int k = yb * BPW + blk;
    const bool live = k < A.K;                                     // the last workgroup may hold fewer than BPW blocks
    const ZlBlockPlan *s_plan = s_plan_[blk];
    const int *s_cls = s_cls_[blk];
    const int *s_chunk = s_chunk_[blk];
    const ZlUnit *s_unit = s_unit_[blk];
    // Narrow buses (the reference's 8 voices per channel): one workgroup renders A.NB whole buses, one after the other,
    // from ONE staging pass over their NB * VPB <= 128 voices -- the gathers of consecutive buses keep flowing and the
    // fixed costs per workgroup are shared.  NB == 1: one (bus, mix group) per workgroup.
    const int NB = inTail ? A.tail_nb : A.NB;
    const int zz = inTail ? tj % A.tail_split : (int)bz;          // (a split tail exists only where one workgroup holds all the buses: bz == 0)
    const int bus0 = (NB > 1) ? zz * NB : zz / A.groups;
    const int g    = (NB > 1) ? 0 : zz - bus0 * A.groups;
    const int v0 = bus0 * A.VPB + g * A.G;
    const int vlim = (NB > 1) ? ((bus0 + NB) * A.VPB < V ? (bus0 + NB) * A.VPB : V) : (bus0 + 1) * A.VPB;
    const int v1 = (NB > 1) ? vlim : ((v0 + A.G < vlim) ? v0 + A.G : vlim);
    // the report covers the last block of the call (the same for every lane of a wave -- a wave is a 64-frame tile of one block --
    // and said so: the per-voice test is then a scalar branch instead of two vector instructions)
    const bool wantPeak = __builtin_amdgcn_readfirstlane((int)(live && (A.k0 + k == A.Ktot - 1))) != 0;
    // A block length that is no multiple of 64 (JACK periods of 16 or 32 frames, 441, 480 ...) leaves the last wave of a block with
    // lanes behind the block's end: they recompute the block's LAST frame -- every position, control word and tap they touch is
    // one a real frame touches -- and nothing of theirs is stored, scanned or reported (a duplicate of an existing value does not
    // move a maximum).  `f` below is the lane's own frame (stores, masks), `fc` the frame it computes.
    const int fc = f < N ? f : N - 1;
    const double fd = (double)fc;
#ifdef ZL_STAMPS
    unsigned long long zl_t0 = __builtin_amdgcn_s_memrealtime(), zl_t1 = 0, zl_paths = 0;
#endif

    float accL = 0.0f, accR = 0.0f;
    int curBus = bus0, busEnd = (bus0 + 1) * A.VPB;               // NB > 1: the bus being summed and its end in voice numbers

    // the finished mix of one bus (or mix group): bus row / partials row, and the fused AudioLevels block scan
    auto store_bus = [&](int bus) {
        float *outL, *outR;
        if (A.groups == 1) {
            const long long KN = (long long)A.Ktot * N;
            outL = A.bus + (long long)bus * (A.bus_stride ? A.bus_stride : 2 * KN) + (long long)(A.k0 + k) * N;
            outR = outL + (A.ch_stride ? A.ch_stride : KN);
        } else {
            outL = A.partials + ((((size_t)k * A.B + bus) * A.groups + g) * 2) * (size_t)N;
            outR = outL + N;
        }
        bool written = false;
        if (live) {
            if (MODE & ZL_MODE_FIX_DELAY) {
                written = f < N;
                if (written) { outL[f] = accL; outR[f] = accR; }
            } else {
                // quirk Q2: the reference pre-increments its output pointers, so frame f lands in out[f+1],
                // out[0] stays 0 and the sample of the last frame falls outside the buffer (dropped)
                written = f + 1 < N;
                if (written) { outL[f + 1] = accL; outR[f + 1] = accR; }
                if (f == 0)  { outL[0] = 0.0f;    outR[0] = 0.0f; }
            }
            if (A.host_out && A.groups == 1) {
                // offline bounce: the same samples once more, into the caller's page-locked host buffer (stores over PCIe; a wave's
                // 64 frames are 256 contiguous bytes per row), in the recorder's 16-bit format if asked (zl_pcm16)
                const size_t fo = (size_t)(A.host_k0 + A.k0 + k) * N;
                const int at = (MODE & ZL_MODE_FIX_DELAY) ? f : f + 1;
                if (A.host_fmt == 1) {
                    uint32_t *o = static_cast<uint32_t *>(A.host_out) + (size_t)bus * (size_t)A.host_total + fo;
                    if (at < N) __builtin_nontemporal_store((uint32_t)(uint16_t)zl_pcm16(accL) | ((uint32_t)(uint16_t)zl_pcm16(accR) << 16), o + at);
                    if (!(MODE & ZL_MODE_FIX_DELAY) && f == 0) __builtin_nontemporal_store(0u, o);
                } else {
                    float *oL = static_cast<float *>(A.host_out) + ((size_t)bus * 2) * (size_t)A.host_total + fo, *oR = oL + A.host_total;
                    if (at < N) { __builtin_nontemporal_store(accL, oL + at); __builtin_nontemporal_store(accR, oR + at); }
                    if (!(MODE & ZL_MODE_FIX_DELAY) && f == 0) { __builtin_nontemporal_store(0.0f, oL); __builtin_nontemporal_store(0.0f, oR); }
                }
            }
            if (A.fan && A.groups == 1) {
                // fused JackPassthrough fan-out of the finished bus (JackPassthrough.cpp:45-115): three more stereo pairs
                // written from the registers that hold the mix -- no second pass over the bus
                const size_t KN = (size_t)A.Ktot * N;
                const ZlPassParams pp = A.pass_inline ? A.pass0 : A.pass[bus];
                float *o = A.fan + ((size_t)bus * 6) * KN + (size_t)(A.k0 + k) * N;
                float lm, rm; zl_pass_pan(pp, lm, rm);
                const float amounts[3] = { pp.dry, pp.fx1, pp.fx2 };
                const int fo = (MODE & ZL_MODE_FIX_DELAY) ? f : f + 1;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float ol, orr;
                    zl_pass_pair(pp, amounts[c], lm, rm, accL, accR, ol, orr);
                    if (fo < N) { o[(size_t)(2 * c) * KN + fo] = ol; o[(size_t)(2 * c + 1) * KN + fo] = orr; }
                    if (!(MODE & ZL_MODE_FIX_DELAY) && f == 0) {   // the bus holds 0 in frame 0 (Q2)
                        zl_pass_pair(pp, amounts[c], lm, rm, 0.0f, 0.0f, ol, orr);
                        o[(size_t)(2 * c) * KN] = ol; o[(size_t)(2 * c + 1) * KN] = orr;
                    }
                }
            }
        }
        // ---- fused AudioLevels block scan (AudioLevels.cpp:361-383) when this workgroup holds the final mix of its
        //      whole block(s) (no mix groups, one frame tile): saves the K3 launch and its re-read of the bus
        if (A.groups == 1 && (gdx == 1 || A.tile_accum) && A.levels) {
            const float aL = written ? accL : 0.0f, aR = written ? accR : 0.0f;      // (a lane without a frame of its own scans a zero)
            int pkL = zl_sample_to_peak_int(aL), pkR = zl_sample_to_peak_int(aR);
            float sqL = aL * aL, sqR = aR * aR;
            zl_wave_levels4(pkL, pkR, sqL, sqR);
            // each wave leaves its partial result of this bus; they are combined once, after the last bus (no barrier here)
            const int w = threadIdx.x >> 6, bi = bus - bus0;
            if ((threadIdx.x & 63) == 0) { s_pk[0][bi][w] = pkL; s_pk[1][bi][w] = pkR; s_sq[0][bi][w] = sqL; s_sq[1][bi][w] = sqR; }
        }
    };
    // the block levels of the workgroup's buses from the per-wave partial results: one lane per (bus, block)
    auto combine_levels = [&](int nbus) {
        if (!(A.groups == 1 && (gdx == 1 || A.tile_accum) && A.levels)) return;
        __syncthreads();
        for (int idx = threadIdx.x; idx < nbus * BPW; idx += blockDim.x) {
            const int bi = idx / BPW, b = idx - bi * BPW;
            const int kk = yb * BPW + b;
            if (kk >= A.K) continue;
            // the waves of block b: all of the workgroup's (BPW == 1) or N / 64 of them
            const int w0 = (BPW > 1) ? b * (N >> 6) : 0;
            const int nw = (BPW > 1) ? (N >> 6) : (int)((blockDim.x + 63) >> 6);
            ZlBlockLevels lv; lv.peak_l = 0; lv.peak_r = 0; lv.sumsq_l = 0.0f; lv.sumsq_r = 0.0f;
            // a later frame tile of a block this workgroup walks tile by tile: carry on from what this very lane stored after the tile before
            // (read past the vector L1: agent-scope loads)
            if (bx > 0) {
                ZlBlockLevels *pl = &A.levels[(size_t)kk * A.B + bus0 + bi];
                lv.peak_l = __hip_atomic_load(&pl->peak_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lv.peak_r = __hip_atomic_load(&pl->peak_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lv.sumsq_l = __hip_atomic_load(&pl->sumsq_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lv.sumsq_r = __hip_atomic_load(&pl->sumsq_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (int i = w0; i < w0 + nw; ++i) {
                lv.peak_l = s_pk[0][bi][i] > lv.peak_l ? s_pk[0][bi][i] : lv.peak_l;
                lv.peak_r = s_pk[1][bi][i] > lv.peak_r ? s_pk[1][bi][i] : lv.peak_r;
                lv.sumsq_l += s_sq[0][bi][i]; lv.sumsq_r += s_sq[1][bi][i];
            }
            A.levels[(size_t)kk * A.B + bus0 + bi] = lv;
        }
    };

    int vbase = v0;                                               // first voice of the staging pass under way
    // gathers issued ahead of a bus's epilogue (linear modes; with four taps they would be 64 registers)
    constexpr bool PF = !ST && !(MODE & ZL_MODE_HERMITE) && ZL_K2_PREFETCH_BUS;
    // after a chunk of U voices: if it held the last voice of a narrow bus (VPB is a multiple of the chunk size there), write that
    // bus and start the next one
    auto chunk_end = [&](int c) {
        if (NB > 1 && vbase + c + U == busEnd) {
            store_bus(curBus);
            accL = 0.0f; accR = 0.0f;
            ++curBus; busEnd += A.VPB;
        }
    };
    for (int vb = v0; vb < v1; vb += CH) {
        const int nv = (v1 - vb < CH) ? v1 - vb : CH;
        vbase = vb;
        // ---- stage the per-voice records of this pass in LDS: one lane per (block, voice) issues every load it may
        //      need at once (voice constants, run list; plan header + first segment when no run covers the block), so
        //      the prologue is one memory round trip (two for blocks with a second segment) and one barrier
        // (the barrier keeps a pass from overwriting records that another wave still reads: nothing to wait for before the FIRST pass of a
        // batch workgroup -- its first wave issues the staging loads while the others are still being launched; the resident kernel runs
        // this body once per cycle and keeps the barrier)
        if (!REPORTS || vb != v0) __syncthreads();
#ifdef ZL_STAMPS
        const unsigned long long zl_ta = __builtin_amdgcn_s_memrealtime();
        unsigned long long zl_tb = zl_ta;
#endif
        for (int idx = threadIdx.x; idx < BPW * CH; idx += blockDim.x) {   // whole waves: blockDim.x is a multiple of 64
            const int b = idx / CH, i = idx - b * CH;
            const int kk = yb * BPW + b;
            ZlVoiceConst vc;
            ZlBlockPlan pl;
            zl_plan_clear(pl);
            vc.src_offset = 0; vc.sample_duration = 0; vc.channels = 2;
            vc.lgain = vc.rgain = vc.clip_volume = vc.lpan = vc.rpan = vc.env = 0.0f; vc.pad[0] = vc.pad[1] = 0;
            if (i < nv) {
                vc = A.vconst[vb + i];                             // K1 leaves a neutral record for voices that do not play
                if (kk < A.K) {
                    // zl_plan_lookup (implied by a run, explicit, or idle) with the run list AND the explicit record's header and first segment
                    // fetched together: ONE memory round trip per prologue where "look at the runs, then fetch the record" is two in a row (most
                    // blocks of a long window are explicit: a voice has ZL_MAXRUNS inline runs, a 2 s loop in a 43 s window restarts twenty times).
                    // The record of a block that a run covers is never written: its bits lose every select below.
                    const ZlRunList *rl = &A.runs[vb + i];
                    const size_t pidx = (size_t)kk * V + (size_t)(vb + i);
                    const ZlPlanHdr ph = A.plan_hdr[pidx];
                    const ZlPlanSeg0 ps = A.plan_seg0[pidx];
                    const int rn = rl->n, dead = rl->dead_from;
                    bool cov = false; double rP = 0.0, rstep = 0.0; int rk0 = 0;
#pragma unroll
                    for (int j = 0; j < ZL_MAXRUNS; ++j) {
                        const ZlRun r = rl->r[j];
                        const bool hit = !cov && j < rn && kk >= r.k0 && kk < r.k1;
                        rP = hit ? r.P : rP; rstep = hit ? r.step : rstep; rk0 = hit ? r.k0 : rk0;
                        cov = cov || hit;
                    }
                    if (kk < dead) {
                        pl.flags = cov ? (int32_t)ZL_PLAN_ACTIVE : ph.flags; pl.n_active = cov ? N : ph.n_active; pl.nseg = cov ? 1 : ph.nseg;
                        pl.env = cov ? vc.env : ph.env;
                        pl.P0 = cov ? fma((double)((kk - rk0) * N), rstep, rP) : ps.P0;   // exact: inside the linear run
                        pl.step = cov ? rstep : ps.step;
                        if (!cov && ((((ph.nseg >= 2 || (ph.flags & ZL_PLAN_ENV)) && !(ph.flags & ZL_PLAN_SLOW)) || (ph.flags & (ZL_PLAN_NOSLOT_SIM | ZL_PLAN_NOSLOT_EXPAND))))) {
                            const ZlPlanSeg1 s1 = A.plan_seg1[pidx];
                            pl.P1 = s1.P1; pl.step1 = s1.step1; pl.n1 = s1.n1; pl.estep0 = s1.estep0; pl.E1 = s1.E1; pl.estep1 = s1.estep1;
                        }
                    }
                }
            }
#ifdef ZL_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            zl_tb = __builtin_amdgcn_s_memrealtime();
#endif
            int cls = (pl.flags & ZL_PLAN_ACTIVE) ? (1 | ((pl.flags & ZL_PLAN_SLOW) ? 2 : 0)) : 0;
            // 4 = "simple": whole block, sustain (and no debug trace); 8 = it has a second position segment; 16 = mono source
            const float gprod = vc.lgain * vc.rgain * vc.clip_volume * pl.env;     // finite iff every factor is (or one is 0 * inf = NaN)
            // (sources of 4 GiB and more take the general path: the simple paths address a source with 32-bit byte offsets)
            if (cls == 1 && pl.nseg <= 2 && !(pl.flags & ZL_PLAN_ENV) && pl.n_active == N && (vc.channels == 1 || vc.channels == 2) && !A.trace
                && (gprod - gprod) == 0.0f && (uint32_t)vc.sample_duration < 0x1ffffff0u)
            {
                // 64 = "interior": first and last position of the block (P0 and P0 + (N-1) step, step > 0) leave room for
                // every tap: pos + 1 <= duration (pos >= 1 and pos + 2 <= duration with 4 taps)
                constexpr bool HM = (MODE & ZL_MODE_HERMITE) != 0;
                const double Pmax = fma((double)(N - 1), pl.step, pl.P0);
                const bool interior = pl.nseg == 1 && pl.P0 >= (HM ? 1.0 : 0.0) && Pmax < (double)(vc.sample_duration - (HM ? 1 : 0));
                cls |= 4 | (pl.nseg == 2 ? 8 : 0) | (vc.channels == 1 ? 16 : 0) | ((pl.nseg == 1 && pl.step == 1.0 && pl.P0 < 1073741824.0) ? 32 : 0)
                     | (interior ? 64 : 0);
            }
            if (ST) {
                // windows of the block's waves (BPW == 1: the workgroup's four frame tiles; else the N / 64 waves of block b).
                // Staged: an interior stereo voice on one line whose window fits a slot for every wave.
                constexpr int WPB = 4 / BPW;
                constexpr int TB = (MODE & ZL_MODE_HERMITE) ? 1 : 0, TA = (MODE & ZL_MODE_HERMITE) ? 2 : 1;
                const bool cand = (cls & (4 | 8 | 16 | 64)) == (4 | 64) && (uint32_t)vc.sample_duration < 0x0ffffff0u;
                int a[WPB], n16[WPB];
                bool fits = cand;
#pragma unroll
                for (int w = 0; w < WPB; ++w) {
                    const int f0 = (BPW > 1) ? 64 * w : (int)bx * 256 + 64 * w;
                    const int first = (int)fma((double)f0, pl.step, pl.P0) - TB, last = (int)fma((double)(f0 + 63), pl.step, pl.P0) + TA;
                    a[w] = first & ~1;                             // even frame = 16-byte aligned in the arena
                    n16[w] = (last - a[w] + 2) >> 1;               // 16-byte pieces covering frames [a, last]
                    fits = fits && n16[w] >= 1 && n16[w] <= ZL_ST_SLOT / 16;
                }
#pragma unroll
                for (int w = 0; w < WPB; ++w) { ZlWin wn; wn.a = fits ? a[w] : 0; wn.n16 = fits ? n16[w] : 0; s_win_[i * 4 + b * WPB + w] = wn; }
                const unsigned long long fm = __ballot(fits);      // the wave's 64 voices are one block's voices [64 m, 64 m + 64)
                if ((i & 63) == 0) s_stmask_[b * (CH / 64) + (i >> 6)] = fm;
            }
            { ZlUnit un; un.ipos = (int)pl.P0; un.alpha = (float)(pl.P0 - (double)un.ipos);
              un.gpl = (vc.lgain * pl.env) * vc.clip_volume; un.gpr = (vc.rgain * pl.env) * vc.clip_volume; s_unit_[b][i] = un; }
            if (b == 0) s_vc[i] = vc;
            s_plan_[b][i] = pl;                   // idle slots: a harmless record with no active frame
            s_cls_[b][i] = cls;
            // class of each chunk of U voices: OR of bits 1, 2, 8; 4 = every voice simple and of one source layout, 16 = all
            // mono (ballots over the wave's 64 voices)
            const unsigned long long m1 = __ballot(cls & 1), m2 = __ballot(cls & 2), m4 = __ballot(cls & 4), m8 = __ballot(cls & 8),
                                     m16 = __ballot(cls & 16), m32 = __ballot(cls & 32), m64 = __ballot(cls & 64);
            const int lane = i & 63;
            if (lane < 64 / U) {
                const unsigned long long full = (1ull << U) - 1ull;
                const int sh = lane * U;
                const unsigned long long mono = (m16 >> sh) & full;
                const int cc = (((m1 >> sh) & full) ? 1 : 0) | (((m2 >> sh) & full) ? 2 : 0) | (((m8 >> sh) & full) ? 8 : 0)
                             | (((((m4 >> sh) & full) == full) && (mono == 0 || mono == full)) ? 4 : 0) | (mono == full ? 16 : 0)
                             | ((((m32 >> sh) & full) == full) ? 32 : 0) | ((((m64 >> sh) & full) == full) ? 64 : 0);
                s_chunk_[b][(i >> 6) * (64 / U) + lane] = cc;
            }
        }
        __syncthreads();
#ifdef ZL_STAMPS
        zl_t1 = __builtin_amdgcn_s_memrealtime();
        zl_paths = ((zl_ta - zl_t0) & 0xffffull) | (((zl_tb - zl_t0) & 0xffffull) << 16);   // prologue break-down: first barrier, loads landed (10 ns ticks)
#endif
        const size_t pbase = (size_t)k * V + vb;
        if (ST) {
            // ---- LDS-staged pass: the wave's ring runs ZL_ST_D voices ahead of its compute.  Voice j of the pass owns slot j mod
            //      ZL_ST_D.  Before the step over voices [c0, c0 + US) everything up to voice c0 + US - 1 must have landed: at most
            //      the D - US youngest DMAs (voices c0 + US .. c0 + D - 1) may be outstanding.  Voices that are not staged (loop
            //      ends, events, mono, ...) take a one-lane dummy DMA so that the count stays uniform, and are rendered by the
            //      general single-voice path -- in voice order, as always.  A step's wave-uniform inputs (the next voices'
            //      parameters, the next DMAs' addresses) are read one step ahead, into registers.
            constexpr int D = ZL_ST_D, US = ZL_ST_U;
            static_assert(D % US == 0 && D > US && D <= 60 && 8 % US == 0, "ring shape");
            const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            const char *ring = zl_dyn_lds + wave * (D * ZL_ST_SLOT);
            const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)zl_lds_addr(ring));
            const ZlWin *s_win = s_win_ + wave;                    // entry of voice i: s_win[i * 4]
            const uint32_t lane16 = (threadIdx.x & 63u) << 4;
            unsigned long long stm[CH / 64];
#pragma unroll
            for (int m = 0; m < CH / 64; ++m) {
                const unsigned long long x = s_stmask_[blk * (CH / 64) + m];
                stm[m] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(x >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)x);
            }
            const int nvR = (nv + US - 1) / US * US;
            for (int j = 0; j < D && j < nv; ++j) {
                const ZlStDma d = zl_st_dma(A, s_vc, s_win, j, lane16);
                zl_glds_piece(d.g, lane16, d.nbytes, ring_lds + (uint32_t)j * ZL_ST_SLOT);
            }
            // parameters in two register sets that swap roles every step (no copies); the per-voice DMA descriptors likewise
            ZlStParams pa[US], pb[US];
#pragma unroll
            for (int u = 0; u < US; ++u) pa[u] = zl_st_params(s_plan, s_vc, s_win, u);           // (records of idle slots are harmless)
            int slot0 = 0;
            // one step over voices [c0, c0 + US): `cur` was read one step ago, `nxt` is read now for the following step
            auto step = [&](int c0, const ZlStParams (&cur)[US], ZlStParams (&nxt)[US]) {
                if (c0 + D < nv) zl_wait_vm<D - US>(); else zl_wait_vm<0>();     // (past that point nothing more is issued)
                const unsigned long long mw = (CH > 64 && c0 >= 64) ? stm[(CH / 64) - 1] : stm[0];
                const unsigned sm = (unsigned)(mw >> (c0 & 63)) & ((1u << US) - 1u);
                int slot[US];
#pragma unroll
                for (int u = 0; u < US; ++u) { slot[u] = slot0 + u; slot[u] -= slot[u] >= D ? D : 0; }
                ZlStDma dma[US];
                if (sm == (1u << US) - 1u) {
                    ZlStTaps t[US];
#pragma unroll
                    for (int u = 0; u < US; ++u) t[u] = zl_st_taps<MODE>(cur[u], ring, slot[u], fd);
                    // one step ahead, while the taps travel: the next voices' parameters and the next DMAs' addresses
#pragma unroll
                    for (int u = 0; u < US; ++u) {
                        nxt[u] = zl_st_params(s_plan, s_vc, s_win, c0 + US + u < CH ? c0 + US + u : CH - 1);
                        dma[u] = zl_st_dma(A, s_vc, s_win, c0 + D + u < CH ? c0 + D + u : CH - 1, lane16);
                    }
                    zl_f2 acc = {accL, accR};
#ifdef ZL_ST_DIAG_NOMIX     // timing-only diagnostic build: taps are read, nothing is computed
#pragma unroll
                    for (int u = 0; u < US; ++u) acc += t[u].xm + t[u].x2;
#else
#pragma unroll
                    for (int u = 0; u < US; ++u) zl_st_mix<MODE>(A, cur[u], t[u], vb + c0 + u, wantPeak, acc);
#endif
                    accL = acc.x; accR = acc.y;
                } else {
#pragma unroll
                    for (int u = 0; u < US; ++u) {
                        const int i = c0 + u;
                        if (i < nv) {
                            const int cl = __builtin_amdgcn_readfirstlane(s_cls[i]);
                            if ((sm >> u) & 1u) {
                                const ZlStTaps t = zl_st_taps<MODE>(cur[u], ring, slot[u], fd);
                                zl_f2 ac = {accL, accR};
                                zl_st_mix<MODE>(A, cur[u], t, vb + i, wantPeak, ac);
                                accL = ac.x; accR = ac.y;
                            } else if (cl == 0) { }                                                       // idle (SamplerSynth.cpp:137)
                            else if (cl & 2) zl_k2_chunk<MODE, true, 1>(A, s_plan, s_vc, s_cls, i, pbase, vb, fc, wantPeak, accL, accR);
                            else zl_k2_chunk<MODE, false, 1>(A, s_plan, s_vc, s_cls, i, pbase, vb, fc, wantPeak, accL, accR);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < US; ++u) {
                        nxt[u] = zl_st_params(s_plan, s_vc, s_win, c0 + US + u < CH ? c0 + US + u : CH - 1);
                        dma[u] = zl_st_dma(A, s_vc, s_win, c0 + D + u < CH ? c0 + D + u : CH - 1, lane16);
                    }
                }
                if (NB > 1 && vb + c0 + US == busEnd) {
                    // that step held the last voice of a bus (bus widths are multiples of 8 and US divides 8): write it, start the next
                    store_bus(curBus);
                    accL = 0.0f; accR = 0.0f;
                    ++curBus; busEnd += A.VPB;
                }
#pragma unroll
                for (int u = 0; u < US; ++u)
                    if (c0 + D + u < nv) zl_glds_piece(dma[u].g, lane16, dma[u].nbytes, ring_lds + (uint32_t)slot[u] * ZL_ST_SLOT);
                slot0 += US; slot0 -= slot0 >= D ? D : 0;
            };
            for (int c0 = 0; c0 < nvR; c0 += 2 * US) {
                step(c0, pa, pb);
                if (c0 + US < nvR) step(c0 + US, pb, pa);
            }
        } else
        for (int c0 = 0; c0 < nv; ) {
            const int cc = __builtin_amdgcn_readfirstlane(s_chunk[c0 / U]);
            if (cc == 0) {                                        // nobody in this chunk plays (SamplerSynth.cpp:137)
            } else if ((cc & 124) == 100) {
                // (the shared-tap form needs a wave's lanes to be consecutive frames of one block: true for every launch shape --
                // a wave is a 64-frame tile of its block)
                if (ZL_K2_UNIT_SHARE) zl_k2_chunk_unit_shared<MODE, U>(A, s_plan, s_vc, s_unit, c0, vb, fc, wantPeak, accL, accR);
                else zl_k2_chunk_simple<MODE, false, true, true, U>(A, s_plan, s_vc, s_unit, c0, vb, fc, fd, wantPeak, accL, accR);
            }
            else if ((cc & 92) == 68)   zl_k2_chunk_simple<MODE, false, false, true, U>(A, s_plan, s_vc, s_unit, c0, vb, fc, fd, wantPeak, accL, accR);
            else if ((cc & 28) == 4)    zl_k2_chunk_simple<MODE, false, false, false, U>(A, s_plan, s_vc, s_unit, c0, vb, fc, fd, wantPeak, accL, accR);
            else if ((cc & 20) == 4)    zl_k2_chunk_simple<MODE, true, false, false, U>(A, s_plan, s_vc, s_unit, c0, vb, fc, fd, wantPeak, accL, accR);
            else if ((cc & 124) == 116) zl_k2_chunk_simple_mono<MODE, false, true, true, U>(A, s_plan, s_vc, s_unit, c0, vb, fc, fd, wantPeak, accL, accR);
            else if ((cc & 92) == 84)   zl_k2_chunk_simple_mono<MODE, false, false, true, U>(A, s_plan, s_vc, s_unit, c0, vb, fc, fd, wantPeak, accL, accR);
            else if ((cc & 28) == 20)   zl_k2_chunk_simple_mono<MODE, false, false, false, U>(A, s_plan, s_vc, s_unit, c0, vb, fc, fd, wantPeak, accL, accR);
            else if ((cc & 20) == 20)   zl_k2_chunk_simple_mono<MODE, true, false, false, U>(A, s_plan, s_vc, s_unit, c0, vb, fc, fd, wantPeak, accL, accR);
            else {
                // general chunks (events, second segments, mixed layouts, per-frame control) are rare: run them as
                // two half-chunks so their extra per-voice registers do not set the kernel's register budget
                constexpr int H = U / 2;
                for (int h = 0; h < U; h += H) {
                    if (cc & 2) zl_k2_chunk<MODE, true, H>(A, s_plan, s_vc, s_cls, c0 + h, pbase, vb, fc, wantPeak, accL, accR);
                    else        zl_k2_chunk<MODE, false, H>(A, s_plan, s_vc, s_cls, c0 + h, pbase, vb, fc, wantPeak, accL, accR);
                }
            }
            // narrow buses: the chunk that ends a bus is followed by that bus's epilogue (store, level scan: ~100 instructions in a
            // dependent chain).  The NEXT bus's gathers go out first -- they travel while the epilogue runs -- when its first chunk is
            // of the steady-state kind (stereo, interior, one segment); with the reference's 8 voices per bus this inner loop walks
            // all the workgroup's buses.  (The taps in flight live only inside this loop: no register is held for them elsewhere.)
            while (PF && NB > 1 && vbase + c0 + U == busEnd && c0 + U < nv) {
                const int nc = __builtin_amdgcn_readfirstlane(s_chunk[c0 / U + 1]);
                const bool unit = (nc & 124) == 100;
                if (!unit && (nc & 92) != 68) break;
                ZlSimpleTaps<(MODE & ZL_MODE_HERMITE) != 0, U> pf;
                if (unit) zl_k2_simple_issue<MODE, false, true, true, U>(A, s_plan, s_vc, s_unit, c0 + U, fc, fd, pf);
                else      zl_k2_simple_issue<MODE, false, false, true, U>(A, s_plan, s_vc, s_unit, c0 + U, fc, fd, pf);
                chunk_end(c0);
                c0 += U;
                zl_k2_simple_mix<MODE, true, U>(A, s_plan, s_vc, s_unit, c0, vb, wantPeak, pf, accL, accR);
            }
            chunk_end(c0);
            c0 += U;
        }
    }

#ifdef ZL_STAMPS
    if (threadIdx.x == 0 && A.pos_trace) {
        unsigned long long *st = reinterpret_cast<unsigned long long *>(A.pos_trace) + 4 * ((size_t)bz * gdy + by);
        st[0] = zl_t0; st[1] = zl_t1; st[2] = __builtin_amdgcn_s_memrealtime();
        st[3] = zl_paths;
    }
#endif
    if (NB == 1) { store_bus(bus0); combine_levels(1); }
    else combine_levels(curBus - bus0);

    // ---- the call's reports (zl_k_reports' work), by the workgroups that hold the call's last block: this workgroup summed its buses
    //      whole, so its voices' peaks are complete once its own waves have issued their atomics (the barrier waits for them)
    if (REPORTS && A.fused_reports) {
        const int klast = A.Ktot - 1 - A.k0;                      // the call's last block in this window's numbering (uniform)
        if (klast >= 0 && klast < A.K && klast >= yb * BPW && klast < yb * BPW + BPW) {
            __syncthreads();
            for (int v = v0 + (int)threadIdx.x; v < v1; v += (int)blockDim.x) {
                // (a report is two 16-byte words: {playing, valid, peak bits, progress} {clip, pad, position}; the peak is read past the vector L1)
                const uint4 *src = reinterpret_cast<const uint4 *>(&A.reports[v]);
                uint4 a = src[0];
                const uint4 b = src[1];
                a.z = __hip_atomic_load(&A.reports[v].peak_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float gn = __uint_as_float(a.z) * 0.5f;
                A.rep_gain[v] = gn;
                if (A.rep_host) { uint4 *dst = reinterpret_cast<uint4 *>(&A.rep_host[v]); dst[0] = a; dst[1] = b; A.rep_host_gain[v] = gn; }
            }
            if (bus0 == 0 && g == 0 && threadIdx.x == 0 && A.rep_host_stats) {
                const unsigned long long sb = A.stats->source_bytes, sl = A.stats->slow_blocks, af = A.stats->active_frames;
                A.rep_host_stats->source_bytes = sb; A.rep_host_stats->slow_blocks = sl; A.rep_host_stats->active_frames = af;
                A.stats->source_bytes = 0; A.stats->slow_blocks = 0; A.stats->active_frames = 0;   // cleared for the call that reuses this slot
            }
        }
    }
}

// the kernel: one workgroup = one (bus or group of narrow buses, mix group, block or BPW short blocks, frame tile)
// (faithful linear mode, one block per workgroup -- the headline shape: ZL_K2_WAVES_LINEAR = 5 waves per SIMD, which is what the LDS
// cap of the launch allows anyway.  Two 128-frame blocks per workgroup: 5 waves -- what its 28.8 KB of
// LDS allow -- instead of the 4 its 104 registers give: +2..3 % with 10 spilled registers, profiles/round2_e_k2_experiments.txt;
// four 64-frame blocks: likewise 5, with 64 voices per staging pass)
template <uint32_t MODE, int BPW, bool ST>
__global__ void __launch_bounds__(256, ST ? 3 : (MODE & (ZL_MODE_HERMITE | ZL_MODE_FIX_DELAY)) == 0 ? (BPW == 1 ? ZL_K2_WAVES_LINEAR : 5) : ZL_K2_MINWAVES) zl_k2_render(const ZlBatch A)
{
    zl_k2_body<MODE, BPW, ST>(A, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

// ------------------------------------------------------------------------------------------------
// Resident real-time kernel: one workgroup PER BUS that stays on the GPU and renders a JACK cycle whenever the host
// posts one in the mailbox -- no kernel launch, no completion event, no command-processor packet per block.  A workgroup owns
// its bus: it applies the voice operations of the bus's voices (K0), plans them (K1 + K1c, one block), renders the bus (the K2
// body: the workgroup's lanes = the block's frames), and writes mix + reports straight into host memory.  Nothing is handed
// from one workgroup to another except the command: workgroup 0 watches the mailbox (host memory) and republishes each block in
// HBM with agent-scope atomics; the last workgroup to finish (an arrival counter) tells the host.  Only workgroup 0 decides to
// leave -- on a stop request or after idle_ticks (100 MHz) without a block -- and publishes that too, so every block is either
// rendered by all workgroups or by none, and every wave reaches the exit.
//
// WIDE (buses of 32 voices and more): one workgroup PER VOICE instead -- a single block has no other parallelism than its voices.
// Every workgroup renders its voice into the bus's partial rows (K2 body with one voice per mix group); the workgroup that is the
// last of its bus to arrive (a counter per bus, no waiting) sums the rows in voice order, 0 + v0 + v1 + ... -- the reference's
// order, bit for bit (K3 body) -- writes the bus into host memory and scans its levels.  The voice-operation ranges of the block
// are copied from host memory into HBM once, by workgroup 0 before it publishes the block (a thousand workgroups reading the
// same host memory would queue on PCIe).  Hand-offs between workgroups on different XCDs use agent-scope release / acquire fences.
static __device__ __forceinline__ void zl_k3_body(const ZlBatch &A, const float *bus_in, int k, int bus);
// (two waves per SIMD = two resident workgroups per CU: the budget the capacity rule of zl_engine.cpp rt_eligible counts on -- 256
// registers per lane, accumulation registers included)
template <uint32_t MODE, bool WIDE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) zl_k_rt_loop(const ZlBatch A0, ZlRtShared *sh, ZlRtDev *dev, unsigned long long first_seq, unsigned long long idle_ticks,
                                                    float *gain_out, ZlReport *host_reports, float *host_gain, ZlOpRange *dev_ranges, int vw)
{
    __shared__ unsigned long long s_cmd[ZL_RT_CMD_WORDS + 1];     // [0] = the block's sequence number (0 with s_go = 0: leave)
    __shared__ int s_go, s_last;
    __shared__ ZlClock s_clk0;
    __shared__ ZlPassParams s_pass;                               // JackPassthrough parameters of this workgroup's bus, as of version s_pass_seq
    __shared__ uint32_t s_pass_seq;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) s_pass_seq = 0u;
    const int z = blockIdx.x, W = gridDim.x;                       // one workgroup per bus (WIDE: per vw voices of one bus; vw divides VPB)
    const int vbeg = WIDE ? z * vw : z * A0.VPB, vend = WIDE ? vbeg + vw : vbeg + A0.VPB;
    unsigned long long last = first_seq;
    if (z == 0 && tid == 0) __hip_atomic_store(&sh->state, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (;;) {
        if (tid == 0) {
            int go = 0;
            if (z == 0) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    const unsigned long long q = __hip_atomic_load(&sh->cmd_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (q != last) {
                        // (the host wrote the words, then cmd_seq with a release store.)  Narrow buses: the NUMBER goes out to the other
                        // workgroups at once; each of them reads the words itself.  Wide buses: published below, with the words and the ranges
                        if (!WIDE) __hip_atomic_store(&dev->pub_seq, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_cmd[0] = q; go = 1;
                        break;
                    }
                    if (__hip_atomic_load(&sh->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) || __hip_atomic_load(&sh->yield, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                        || __builtin_amdgcn_s_memrealtime() - t0 > idle_ticks) {
                        __hip_atomic_store(&dev->pub_seq, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // everybody leaves
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            } else {
                for (;;) {
                    const unsigned long long q = __hip_atomic_load(&dev->pub_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (q == ~0ull) break;
                    if (q != last) { s_cmd[0] = q; go = 1; break; }
                    if (WIDE) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(1);   // (a thousand pollers: longer naps)
                }
            }
            s_go = go;
        }
        __syncthreads();
        if (!s_go) break;
        // ---- the cycle's command words, ONE WORD PER LANE (one trip, all words in flight together; a single lane's system-scope loads
        //      would go out one after the other: 34 trips over PCIe): out of the mailbox in host memory (system scope, past the caches)
        //      -- every workgroup of a narrow engine, workgroup 0 of a wide one, which also leaves them in HBM for the others
        if (tid < ZL_RT_CMD_WORDS) {
            unsigned long long wv;
            if (!WIDE || z == 0) wv = __hip_atomic_load(&sh->cmd[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else                 wv = __hip_atomic_load(&dev->cmd[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (WIDE && z == 0) __hip_atomic_store(&dev->cmd[tid], wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_cmd[1 + tid] = wv;
        }
        __syncthreads();
        last = s_cmd[0];
        if (A0.rt_stamps && z == 0 && tid == 0) sh->stamps[0] = __builtin_amdgcn_s_memrealtime();
        ZlBatch A = A0;
        A.N = (int)(uint32_t)s_cmd[1]; A.n_op_ranges = (int)(uint32_t)(s_cmd[1] >> 32);
        A.ops = reinterpret_cast<const ZlVoiceOp *>((uintptr_t)s_cmd[2]); A.op_ranges = reinterpret_cast<const ZlOpRange *>((uintptr_t)s_cmd[3]);
        A.ctl_base = s_cmd[4];
        A.clock0.current_usecs = s_cmd[5]; A.clock0.next_usecs = s_cmd[6]; A.clock0.playhead = s_cmd[7]; A.clock0.playhead_usecs = s_cmd[8];
        A.clock0.subbeat_usecs = s_cmd[9]; A.clock0.usecs_per_frame = s_cmd[10];
        A.n_clip_edits = (int)(uint32_t)s_cmd[11]; A.clip_edits = reinterpret_cast<const ZlClipEdit *>((uintptr_t)s_cmd[12]);
        A.inline_clock = 1; A.fuse_assemble = 1;
        // ---- JackPassthrough fan-out of this cycle (JackPassthrough.cpp:45-115): asked for per cycle; the parameters of the workgroup's
        //      bus are kept in LDS across cycles and read again from the host's table (mapped memory: a trip over PCIe, issued here and
        //      needed at the store, after planning) only when the host moved the table's version -- a knob turned while playing
        // where the cycle's rows go: the engine's staging rows, or straight into the caller's page-locked buffers
        A.bus = reinterpret_cast<float *>((uintptr_t)s_cmd[13]); A.fan = reinterpret_cast<float *>((uintptr_t)s_cmd[14]);
        A.bus_stride = (long long)s_cmd[15]; A.ch_stride = (long long)s_cmd[16];
        const uint32_t fan_seq = (uint32_t)(s_cmd[11] >> 32);
        if (fan_seq == 0u) A.fan = nullptr;
        else if (tid < 64) {
            // (wave 0 only: its lanes read the version before lane 0 writes it -- program order inside one wave; one word per lane, one trip)
            const bool refresh = fan_seq != s_pass_seq;
            const int pb = (WIDE ? vbeg / A0.VPB : z);
            const uint32_t *src = reinterpret_cast<const uint32_t *>(A0.pass + pb);
            uint32_t *dst = reinterpret_cast<uint32_t *>(&s_pass);
            if (refresh && tid < (int)(sizeof(ZlPassParams) / 4)) dst[tid] = __hip_atomic_load(src + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (refresh && tid == 0) s_pass_seq = fan_seq;
        }
        A.pass_inline = 1;
        A.tile_accum = 1;                                          // this workgroup walks every frame tile of its block (below)
        if (WIDE && z == 0) {
            // workgroup 0: the block's operation ranges host memory -> HBM (behind a system-scope acquire: the host reuses its buffers
            // every block and plain loads may hit stale cached lines), then the block is published for the other workgroups
            if (A.n_op_ranges > 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                for (int i = tid; i < A.n_op_ranges; i += (int)blockDim.x) dev_ranges[i] = A.op_ranges[i];
            }
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&dev->pub_seq, last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ---- K0: the operations of this workgroup's voices (the ranges are sorted by voice).  Narrow buses read ranges and
        //      operations in host memory in place (system-scope acquire, as above); wide buses read the ranges from workgroup 0's
        //      copy in HBM (agent-scope acquire: it may sit in another XCD's L2) and only their own operations from host memory
        if (A.n_op_ranges > 0 || A.n_clip_edits > 0) {
            if (WIDE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        }
        // ---- the cycle's clip-parameter edits: EVERY workgroup writes all of them into the HBM clip table (the same bytes; a few per
        //      cycle at most).  A workgroup then plans from what it wrote itself -- its own XCD's L2 -- so no hand-off between
        //      workgroups is needed, and the table is complete in HBM when the kernel leaves.
        for (int i = tid >> 6; i < A.n_clip_edits; i += (int)(blockDim.x >> 6)) zl_apply_clip_edit(A, A.clip_edits + i, lane);
        // (the knob edits that came with the command: 8 words each, from LDS)
        if (tid < ZL_RT_INLINE_EDITS * (ZL_CLIP_HEAD_BYTES / 4)) {
            const int ei = tid / (ZL_CLIP_HEAD_BYTES / 4), wi = tid % (ZL_CLIP_HEAD_BYTES / 4);
            const unsigned long long *ew = &s_cmd[1 + ZL_RT_CMD_FIXED + ei * ZL_RT_EDIT_WORDS];
            const int clip = (int)(uint32_t)ew[0];
            if (clip >= 0) reinterpret_cast<uint32_t *>(const_cast<ZlClip *>(A.clips) + clip)[wi] = reinterpret_cast<const uint32_t *>(ew + 1)[wi];
        }
        const ZlOpRange *ranges = WIDE ? dev_ranges : A.op_ranges;
        for (int i = tid; i < A.n_op_ranges; i += (int)blockDim.x) {
            const ZlOpRange rg = ranges[i];
            if (rg.voice < vbeg || rg.voice >= vend) continue;
            ZlVoiceState st = A.voices[rg.voice];
            for (int j = 0; j < rg.count; ++j) zl_apply_op(st, A.ops[rg.first + j]);
            A.voices[rg.voice] = st;
        }
        if (tid == 0) s_clk0 = A.clock0;
        __threadfence_block();
        __syncthreads();
        if (A0.rt_stamps && z == 0 && tid == 0) sh->stamps[1] = __builtin_amdgcn_s_memrealtime();
        // ---- K1: one lane per voice of the workgroup, the single block of this cycle
        for (int v = vbeg + tid; v < vend; v += (int)blockDim.x) {
            ZlPlanner pl;
            // (inlined here: as calls, the planner object and the cycle's ZlBatch live in scratch memory, 1 KB per lane)
            [[clang::always_inline]] pl.begin(A, v, 0);
            while (pl.t < A.N) { [[clang::always_inline]] pl.iterate(A, 1, &s_clk0, 0, 0); }
            [[clang::always_inline]] pl.end(A);
        }
        __threadfence_block();
        __syncthreads();
        if (A0.rt_stamps && z == 0 && tid == 0) sh->stamps[2] = __builtin_amdgcn_s_memrealtime();
        // ---- K1c: plan records of the block, multi-segment blocks expanded by whole waves
        for (int v0 = vbeg; v0 < vend; v0 += (int)blockDim.x) {
            const int v = v0 + tid;
            const bool mine = v < vend;
            ZlAssembler as;
            [[clang::always_inline]] as.begin(A, mine ? v : vbeg, 0, mine ? 1 : 0);
            zl_k1c_block(A, as, v, lane, 0);
        }
        __threadfence_block();
        __syncthreads();
        if (A0.rt_stamps && z == 0 && tid == 0) sh->stamps[3] = __builtin_amdgcn_s_memrealtime();
        // ---- K2: this bus (WIDE: each of this workgroup's voices into its own partial row: A.groups = voices per bus, one voice per group).
        //      A block longer than the workgroup (JACK periods of 512, 1024 ... frames) is walked tile by tile.
        if (A.fan) A.pass0 = s_pass;                               // (written before the barriers above)
        const unsigned tiles = ((unsigned)A.N + blockDim.x - 1u) / blockDim.x;
        for (unsigned bx = 0; bx < tiles; ++bx) {
            if (WIDE) { for (int v = vbeg; v < vend; ++v) zl_k2_body<MODE, 1, false, false>(A, bx, 0u, (unsigned)v, tiles, 1u); }
            else zl_k2_body<MODE, 1, false, false>(A, bx, 0u, (unsigned)z, tiles, 1u);
            __threadfence_block();
            __syncthreads();
        }
        if (WIDE) {
            // ---- the bus: the last of its workgroups to get here sums the partial rows in voice order, writes the mix, scans the levels
            const int bus = vbeg / A.VPB;
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // this voice's partial rows, for a workgroup that may run on another XCD
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned int old = __hip_atomic_fetch_add(&dev->bus_arrive[bus], (unsigned int)vw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int lastOne = old + (unsigned int)vw == (unsigned int)A.VPB;
                if (lastOne) __hip_atomic_store(&dev->bus_arrive[bus], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = lastOne;
            }
            __syncthreads();
            if (s_last) {                                              // (uniform over the workgroup)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                zl_k3_body(A, nullptr, 0, bus);
            }
            __threadfence_block();
            __syncthreads();
        }
        if (A0.rt_stamps && z == 0 && tid == 0) sh->stamps[4] = __builtin_amdgcn_s_memrealtime();
        // ---- reports (gain = peakGain * 0.5f, SamplerSynthVoice.cpp:266) straight into host memory
        for (int v = vbeg + tid; v < vend; v += (int)blockDim.x) {
            const ZlReport r = A.reports[v];
            const float g = __uint_as_float(r.peak_bits) * 0.5f;
            gain_out[v] = g; host_reports[v] = r; host_gain[v] = g;
        }
        __syncthreads();                                           // every wave's stores are issued and waited for (workgroup release)
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope: this bus's mix and reports (host memory), its levels (HBM)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (A0.rt_stamps && z == 0) sh->stamps[5] = __builtin_amdgcn_s_memrealtime();
            if (!WIDE) {
                // one workgroup per bus: each tells the host itself (the host waits for all of them) -- no counter round trip, no last-arrival store
                __hip_atomic_store(&sh->wg_done[z], (uint32_t)last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                const unsigned int old = __hip_atomic_fetch_add(&dev->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned int)W - 1u) {                 // the last workgroup: the block is complete
                    __hip_atomic_store(&dev->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&sh->done_seq, last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
    if (z == 0 && tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&sh->state, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The AudioLevels scan of one (block, bus) row pair by ONE wave: integer peaks (AudioLevels.cpp:361-383) and the sums of
// squares of the RMS extension in their defined order (oracle/zl_oracle.c zlo_block_sumsq): tiles of 64 frames from frame
// `off` (1 with quirk Q2: frame 0 of a bus is the constant 0 and lane f of K2 holds out[f + 1]; 0 with FIX_DELAY), the
// wave's DPP tree inside a tile, tiles added in order -- the same bits as K2's fused scan.  Any N: a short last tile is padded with zeros.
static __device__ __forceinline__ ZlBlockLevels zl_scan_rows(const float *inL, const float *inR, int N, int off, int lane)
{
    int pkL = 0, pkR = 0; float sqL = 0.0f, sqR = 0.0f;
    if (off) {                                                     // the frames in front of the first tile (uniform loads)
        const float l0 = inL[0], r0 = inR[0];
        pkL = zl_sample_to_peak_int(l0); pkR = zl_sample_to_peak_int(r0);
        sqL = l0 * l0; sqR = r0 * r0;
    }
    for (int t0 = off; t0 < N; t0 += 64) {
        const int i = t0 + lane;
        const float l = i < N ? inL[i] : 0.0f, r = i < N ? inR[i] : 0.0f;
        const int a = zl_sample_to_peak_int(l), c = zl_sample_to_peak_int(r);
        pkL = a > pkL ? a : pkL; pkR = c > pkR ? c : pkR;
        sqL += zl_wave_sum(l * l); sqR += zl_wave_sum(r * r);      // wave-uniform running sums, tile order
    }
    ZlBlockLevels out;
    out.peak_l = zl_wave_max_nonneg(pkL); out.peak_r = zl_wave_max_nonneg(pkR); out.sumsq_l = sqL; out.sumsq_r = sqR;
    return out;
}

// ------------------------------------------------------------------------------------------------
// K3: one workgroup per (block, bus).  Sums the mix-group partials in group order (when there are
// any), writes the bus, and scans it for the AudioLevels integer peak and the RMS extension.
static __device__ __forceinline__ void zl_k3_body(const ZlBatch &A, const float *bus_in, int k, int bus)
{
    const int N = A.N;
    const size_t KN = (size_t)A.Ktot * N;
    float *outL = A.bus ? A.bus + (long long)bus * (A.bus_stride ? A.bus_stride : 2 * (long long)KN) + (long long)(A.k0 + k) * N : nullptr;
    float *outR = outL ? outL + (A.ch_stride ? A.ch_stride : (long long)KN) : nullptr;
    const float *inL = bus_in ? bus_in + ((size_t)bus * 2) * KN + (size_t)(A.k0 + k) * N : outL;
    const float *inR = bus_in ? inL + KN : outR;

    for (int f = threadIdx.x; f < N && A.groups > 1 && !bus_in; f += blockDim.x) {
        float l, r;
        {
            l = 0.0f; r = 0.0f;
            const float *p = A.partials + (((size_t)k * A.B + bus) * A.groups) * 2 * (size_t)N;
            // the adds are sequential (group order = the summation order), the loads are not: eight groups' partials are
            // requested at once (voices_per_task = 1 makes every voice a group and this loop 128 long)
            int g = 0;
            for (; g + 8 <= A.groups; g += 8) {
                float pl[8], pr[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { pl[u] = p[(size_t)u * 2 * N + f]; pr[u] = p[(size_t)u * 2 * N + N + f]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { l += pl[u]; r += pr[u]; }
                p += 16 * (size_t)N;
            }
            for (; g < A.groups; ++g) { l += p[f]; r += p[N + f]; p += 2 * (size_t)N; }
            outL[f] = l; outR[f] = r;
            if (A.fan) {                                           // fused JackPassthrough fan-out, as in K2
                const ZlPassParams pp = A.pass_inline ? A.pass0 : A.pass[bus];
                float *o = A.fan + ((size_t)bus * 6) * KN + (size_t)(A.k0 + k) * N;
                float lm, rm; zl_pass_pan(pp, lm, rm);
                const float amounts[3] = { pp.dry, pp.fx1, pp.fx2 };
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float ol, orr;
                    zl_pass_pair(pp, amounts[c], lm, rm, l, r, ol, orr);
                    o[(size_t)(2 * c) * KN + f] = ol; o[(size_t)(2 * c + 1) * KN + f] = orr;
                }
            }
        }
    }
    // the rows of this (block, bus) are complete: wave 0 scans them in the defined order (zl_scan_rows) -- the workgroup
    // reads back its own stores (workgroup-scope fence + barrier; when nothing was summed the rows are the input's)
    if (A.groups > 1 && !bus_in) { __threadfence_block(); __syncthreads(); }
    if (threadIdx.x < 64) {
        const ZlBlockLevels lv = zl_scan_rows(inL, inR, N, (A.mode & ZL_MODE_FIX_DELAY) ? 0 : 1, (int)threadIdx.x);
        if (threadIdx.x == 0) A.levels[(size_t)k * A.B + bus] = lv;
    }
}
__global__ void __launch_bounds__(256) zl_k3_finalize(const ZlBatch A, const float *bus_in)
{
    zl_k3_body(A, bus_in, (int)blockIdx.x, (int)blockIdx.y);
}

// K3 without partials to sum (the scan of an existing bus: N > 256, or a bus reduced over several GPUs): one WAVE per
// (block, bus), reductions on the VALU -- no LDS, no barrier.
__global__ void __launch_bounds__(256) zl_k3_scan(const ZlBatch A, const float *bus)
{
    const int N = A.N;
    const long long pair = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);       // (block, bus) pairs, bus fastest
    if (pair >= (long long)A.K * A.B) return;                                     // whole waves leave
    const int k = (int)(pair / A.B), b = (int)(pair - (long long)k * A.B);
    const int lane = threadIdx.x & 63;
    const long long KN = (long long)A.Ktot * N;
    const bool own = bus == A.bus;                                 // the call's own bus may have the caller's strides
    const float *inL = bus + (long long)b * (own && A.bus_stride ? A.bus_stride : 2 * KN) + (long long)(A.k0 + k) * N;
    const ZlBlockLevels out = zl_scan_rows(inL, inL + (own && A.ch_stride ? A.ch_stride : KN), N, (A.mode & ZL_MODE_FIX_DELAY) ? 0 : 1, lane);
    if (lane == 0) A.levels[(size_t)k * A.B + b] = out;
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU exchange step: when a bus spans GPUs every rank receives, for its share of the bus, one piece
// per rank (all-to-all over the xGMI mesh).  This kernel sums the pieces IN RANK ORDER -- ((0 + p0) + p1) + ... per sample,
// the order of K3's mix-group sum and of the oracle's grouped mix, so the result is the same bits for any number of ranks --
// writes the reduced piece, and scans it for AudioLevels in the same pass: integer peak and sum of squares (defined order,
// zl_scan_rows) per unit.  A piece is a run of whole units; unit = the N frames of one (bus, channel, block).  One wave per
// unit; the loads of up to 8 pieces are in flight before the ordered adds.
__global__ void __launch_bounds__(256) zl_k_reduce_scan(const float *pieces, int npieces, long long stride, long long units, int N, int off,
                                                        float *out, ZlUnitLevels *lv)
{
    const long long u = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= units) return;                                        // whole waves leave
    const int lane = threadIdx.x & 63;
    const float *p0 = pieces + u * N;
    float *o = out + u * N;
    auto sum_at = [&](int idx) {
        float v = 0.0f;
        const float *p = p0 + idx;
        int r = 0;
        for (; r + 8 <= npieces; r += 8) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = p[(long long)(r + j) * stride];
#pragma unroll
            for (int j = 0; j < 8; ++j) v += x[j];
        }
        for (; r < npieces; ++r) v += p[(long long)r * stride];
        return v;
    };
    int pk = 0; float sq = 0.0f;
    if (off) {                                                     // the frame in front of the first tile (every lane: one address)
        const float v = sum_at(0);
        if (lane == 0) o[0] = v;
        pk = zl_sample_to_peak_int(v); sq = v * v;
    }
    for (int t0 = off; t0 < N; t0 += 64) {
        const int idx = t0 + lane;
        float v = 0.0f;
        if (idx < N) { v = sum_at(idx); o[idx] = v; }
        const int a = zl_sample_to_peak_int(v);
        pk = a > pk ? a : pk;
        sq += zl_wave_sum(v * v);                                  // wave-uniform running sum, tile order
    }
    pk = zl_wave_max_nonneg(pk);
    if (lane == 0) { ZlUnitLevels r; r.peak = pk; r.sumsq = sq; lv[u] = r; }
}

// unit levels of a whole bus ([bus][channel][block], as the exchange gathers them) -> the engine's per-block levels
__global__ void zl_k_levels_import(const ZlUnitLevels *units, ZlBlockLevels *levels, int B, int K)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;           // (block, bus), bus fastest
    if (i >= B * K) return;
    const int k = i / B, b = i - k * B;
    const ZlUnitLevels l = units[((size_t)b * 2) * K + k], r = units[((size_t)b * 2 + 1) * K + k];
    ZlBlockLevels o; o.peak_l = l.peak; o.peak_r = r.peak; o.sumsq_l = l.sumsq; o.sumsq_r = r.sumsq;
    levels[i] = o;
}

// AudioLevels::timerCallback state update for every bus (AudioLevels.cpp:359-360, 367-383 via the
// block scan of K3, 385, 395-396).  dBFS conversion (log10f) stays on the host, as in the reference.
__global__ void zl_k_levels_tick(ZlLevelsState *state, const ZlBlockLevels *levels, int B, int N, int with_hold_bus)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    ZlLevelsState s = state[b];
    s.peak_a = s.peak_a - 10000 > 0 ? s.peak_a - 10000 : 0;
    s.peak_b = s.peak_b - 10000 > 0 ? s.peak_b - 10000 : 0;
    if (levels) {
        const ZlBlockLevels lv = levels[b];
        s.peak_a = lv.peak_l > s.peak_a ? lv.peak_l : s.peak_a;
        s.peak_b = lv.peak_r > s.peak_b ? lv.peak_r : s.peak_b;
        s.sumsq_a = lv.sumsq_l; s.sumsq_b = lv.sumsq_r; s.frames = N;
    }
    if (b == with_hold_bus) {
        const float intToFloatMultiplier = 0.00000152587f;         // AudioLevels.cpp:349 (Q12)
        const float peakA = s.peak_a * intToFloatMultiplier, peakB = s.peak_b * intToFloatMultiplier;
        s.hold_a = (peakA >= s.hold_a) ? peakA : s.hold_a * 0.9f;
        s.hold_b = (peakB >= s.hold_b) ? peakB : s.hold_b * 0.9f;
    }
    state[b] = s;
}

// report finalisation: gain = peakGain * 0.5f (SamplerSynthVoice.cpp:266).  The call's results (reports, gains,
// statistics) are written straight into mapped host memory: no copy command sits between two calls on the stream.
__global__ void zl_k_reports(const ZlReport *reports, int V, float *gain_out, ZlReport *host_reports, float *host_gain,
                             const ZlBatchStats *stats, ZlBatchStats *host_stats)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v == 0 && host_stats) {
        *host_stats = *stats;
        ZlBatchStats z; z.source_bytes = 0; z.slow_blocks = 0; z.active_frames = 0;
        *const_cast<ZlBatchStats *>(stats) = z;                    // cleared for the call that reuses this slot
    }
    if (v >= V) return;
    const ZlReport r = reports[v];
    const float g = __uint_as_float(r.peak_bits) * 0.5f;
    gain_out[v] = g;
    if (host_reports) { host_reports[v] = r; host_gain[v] = g; }
}

// ------------------------------------------------------------------------------------------------
// JackPassthrough: in [B][2][frames] -> out [B][6][frames].  Pure streaming (8 B read, 24 B written per bus frame):
// VEC = 4 frames per lane with 16-byte accesses when frames % 4 == 0 and the buffers are 16-byte aligned.
template <int VEC>
__global__ void __launch_bounds__(256) zl_k_passthrough(const ZlPassParams *params, const float *in, float *out, long long frames)
{
    const int bus = blockIdx.y;
    const ZlPassParams p = params[bus];
    const float *inL = in + (size_t)bus * 2 * frames, *inR = inL + frames;
    float *o = out + (size_t)bus * 6 * frames;
    const float amounts[3] = { p.dry, p.fx1, p.fx2 };
    float lm, rm; zl_pass_pan(p, lm, rm);
    const long long nvec = frames / VEC;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        float sl[VEC], sr[VEC];
        if (VEC == 4) {
            const float4 a = reinterpret_cast<const float4 *>(inL)[i], b = reinterpret_cast<const float4 *>(inR)[i];
            sl[0] = a.x; sl[1 % VEC] = a.y; sl[2 % VEC] = a.z; sl[3 % VEC] = a.w;
            sr[0] = b.x; sr[1 % VEC] = b.y; sr[2 % VEC] = b.z; sr[3 % VEC] = b.w;
        } else { sl[0] = inL[i]; sr[0] = inR[i]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float ol[VEC], orr[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) zl_pass_pair(p, amounts[c], lm, rm, sl[j], sr[j], ol[j], orr[j]);
            float *oL = o + (size_t)(2 * c) * frames, *oR = oL + frames;
            if (VEC == 4) {
                reinterpret_cast<float4 *>(oL)[i] = make_float4(ol[0], ol[1 % VEC], ol[2 % VEC], ol[3 % VEC]);
                reinterpret_cast<float4 *>(oR)[i] = make_float4(orr[0], orr[1 % VEC], orr[2 % VEC], orr[3 % VEC]);
            } else { oL[i] = ol[0]; oR[i] = orr[0]; }
        }
    }
}

// Bounce delivery (zlhip_bounce): a rendered sub-batch [B][2][frames] fp32 leaves the device buffer for its place in the caller's
// buffer, whose rows are `total` frames long.  PCM = the recorder's 16-bit format (zl_render.h, zl_pcm16), [B][total][2]; otherwise
// the floats as they are, [B][2][total].  Four frames per lane: 16-byte loads, one (PCM) or two 16-byte stores.  The engine uses it
// for the 16-bit conversion into a device staging buffer (total = frames); the copy engine moves the rows to the host.
// VEC = 4: four frames per lane where rows, offset and length allow 16-byte accesses, and a per-frame tail for the last frames % 4 frames;
// VEC = 1: every frame on its own (rows or offsets that are no multiple of four frames: periods of 441 or 33 frames with odd windows).
template <bool PCM, int VEC>
__global__ void __launch_bounds__(256) zl_k_deliver(const float *bus, void *out, long long in_stride, long long off, long long frames, long long total)
{
    // frames [off, off + frames) of every bus row (rows are in_stride floats apart) -> the same frames of the output (rows of `total` frames)
    const int b = blockIdx.y;
    const float *Ls = bus + (size_t)b * 2 * in_stride + off, *Rs = bus + ((size_t)b * 2 + 1) * in_stride + off;
    auto pair = [](float a, float c) { return (uint32_t)(uint16_t)zl_pcm16(a) | ((uint32_t)(uint16_t)zl_pcm16(c) << 16); };
    const long long nvec = VEC == 4 ? frames / 4 : 0;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
    if (VEC == 4) {
        const float4 *L = reinterpret_cast<const float4 *>(Ls), *R = reinterpret_cast<const float4 *>(Rs);
        for (long long i = tid; i < nvec; i += nth) {
            const float4 l = L[i], r = R[i];
            if (PCM) {
                reinterpret_cast<uint4 *>(static_cast<int16_t *>(out) + ((size_t)b * total + off) * 2)[i] = make_uint4(pair(l.x, r.x), pair(l.y, r.y), pair(l.z, r.z), pair(l.w, r.w));
            } else {
                float *o = static_cast<float *>(out) + (size_t)b * 2 * total + off;
                reinterpret_cast<float4 *>(o)[i] = l;
                reinterpret_cast<float4 *>(o + total)[i] = r;
            }
        }
    }
    for (long long i = nvec * 4 + tid; i < frames; i += nth) {     // the tail (VEC = 4: at most three frames), or everything (VEC = 1)
        const float l = Ls[i], r = Rs[i];
        if (PCM) reinterpret_cast<uint32_t *>(static_cast<int16_t *>(out) + ((size_t)b * total + off) * 2)[i] = pair(l, r);
        else { float *o = static_cast<float *>(out) + (size_t)b * 2 * total + off; o[i] = l; o[total + i] = r; }
    }
}

// planar (L, R) -> interleaved arena layout; R == nullptr copies mono
__global__ void zl_k_interleave(const float *L, const float *R, float *dst, int length, int pad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= length + pad) return;
    const bool in = i < length;
    if (R) { dst[2 * i] = in ? L[i] : 0.0f; dst[2 * i + 1] = in ? R[i] : 0.0f; }
    else   { dst[i] = in ? L[i] : 0.0f; }
}

// ------------------------------------------------------------------------------------------------
// launchers (called from zl_engine.cpp)
#define ZL_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

int zl_launch_apply_ops(const ZlBatch &A, hipStream_t s)
{
    if (A.n_op_ranges <= 0 && A.n_clip_edits <= 0) return 0;
    hipLaunchKernelGGL(zl_k0_apply_ops, dim3((std::max(A.n_op_ranges, 0) + 63) / 64 + std::max(A.n_clip_edits, 0)), dim3(64), 0, s, A);
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_plan(const ZlBatch &A, int force_slow, hipStream_t s)
{
    hipLaunchKernelGGL(zl_k1_plan, dim3((A.V + 63) / 64), dim3(64), 0, s, A, force_slow);
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_assemble(const ZlBatch &A, hipStream_t s)
{
    const int bpl = zl_k1c_blocks_per_lane(A);
    hipLaunchKernelGGL(zl_k1c_assemble, dim3((A.V + 63) / 64, (A.K + bpl - 1) / bpl), dim3(64), 0, s, A, bpl);
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_render(const ZlBatch &A, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    // blocks of 64 / 128 frames: 4 / 2 blocks per workgroup (batches only; a single block keeps its small workgroup; other lengths
    // below 256 -- 16, 32, 48, 100 ... -- are real-time periods: one block per workgroup of whole waves)
    const int bpw = ((A.N == 64 || A.N == 128) && A.K > 1) ? 256 / A.N : 1;
    // (whole waves: a block of 100 frames runs on 128 lanes, one of 300 on two workgroups of 256)
    const int tpb = bpw > 1 ? 256 : (A.N < 256 ? ((A.N + 63) & ~63) : 256);
    dim3 grid(bpw > 1 ? 1 : (A.N + tpb - 1) / tpb, (A.K + bpw - 1) / bpw, A.NB > 1 ? (A.B + A.NB - 1) / A.NB : A.B * A.groups);
    const dim3 block(tpb);
    // split tail (see zl_k2_body): one workgroup per block holding ALL the buses, a window long enough to have a tail worth splitting
    static const int tail_env = [] { const char *e = getenv("ZL_K2_TAIL"); return e ? atoi(e) : 1; }();
    static const int tail_min = [] { const char *e = getenv("ZL_K2_TAIL_MIN_BLOCKS"); return e ? std::max(8, atoi(e)) : 2048; }();   // (the test tier lowers it)
    ZlBatch At = A;
    At.tail_from = 0; At.tail_split = 1; At.tail_nb = A.NB;
    if (tail_env && bpw == 1 && A.NB > 1 && A.NB == A.B && grid.z == 1 && grid.x == 1 && !(A.staged && A.K > 1 && tpb == 256) && A.K >= tail_min) {
        const int split = (A.NB % 4 == 0) ? 4 : (A.NB % 2 == 0) ? 2 : 1;
        if (split > 1) {
            const int T = std::min(A.K / 4, 640);                  // half a generation of workgroups (5 per CU x 256 CUs)
            At.tail_from = A.K - T; At.tail_split = split; At.tail_nb = A.NB / split;
            grid.y = (unsigned)(At.tail_from + T * split);
        }
    }
    // One-block-per-workgroup kernels fill every SIMD's register file (6 waves x 80 VGPRs; 5 x 96 with 4 taps) and
    // leave no room for a planning wave (88 VGPRs): a K1 launch that arrives after K2 has filled the machine then
    // crawls (measured 550 instead of 130 us).  Unused dynamic LDS caps K2 at 5 workgroups per CU (27 KB each of
    // 160 KB) -- one wave slot per SIMD stays free for the planner, and K2 itself is 0.5 % faster that way.
    static const int pad_env = [] { const char *e = getenv("ZL_K2_LDS_PAD"); return e ? atoi(e) : -1; }();
    static const int pad_env_h = [] { const char *e = getenv("ZL_K2_LDS_PAD_HERMITE"); return e ? atoi(e) : -1; }();
    // (ZL_K2_LDS_PAD=0 -- the sixth workgroup per CU -- was measured again after the planner became a single sweep: K2 itself gains
    // 1..3 %, but a planner launch that arrives just after K2 has filled the machine then waits for the whole K2 launch every now
    // and then (2.5 ms instead of 35 us), and across boxes the calls gain nothing: the cap stays.)
    const int pad = (A.mode & ZL_MODE_HERMITE) ? (pad_env_h >= 0 ? pad_env_h : 0) : (pad_env >= 0 ? pad_env : 10240);
    // ev_start / ev_stop (profiling): the kernel's own begin / end timestamps, taken by the dispatch packet itself -- no
    // event packets around the launch for the command processor to handle
    // LDS-staged source windows (A.staged): batches only, whole 256-thread workgroups; the ring is dynamic LDS
    const bool st = A.staged && A.K > 1 && tpb == 256;
    const int ring = 4 * ZL_ST_D * ZL_ST_SLOT;
    switch (A.mode & 7u) {
#define ZL_CASE(M) case M: \
        if (st && bpw == 4)      hipExtLaunchKernelGGL((zl_k2_render<M, 4, true>), grid, block, ring, s, ev_start, ev_stop, 0, At); \
        else if (st && bpw == 2) hipExtLaunchKernelGGL((zl_k2_render<M, 2, true>), grid, block, ring, s, ev_start, ev_stop, 0, At); \
        else if (st)             hipExtLaunchKernelGGL((zl_k2_render<M, 1, true>), grid, block, ring, s, ev_start, ev_stop, 0, At); \
        else if (bpw == 4)       hipExtLaunchKernelGGL((zl_k2_render<M, 4, false>), grid, block, 0, s, ev_start, ev_stop, 0, At); \
        else if (bpw == 2)       hipExtLaunchKernelGGL((zl_k2_render<M, 2, false>), grid, block, 0, s, ev_start, ev_stop, 0, At); \
        else                     hipExtLaunchKernelGGL((zl_k2_render<M, 1, false>), grid, block, pad, s, ev_start, ev_stop, 0, At); \
        break;
        ZL_CASE(0) ZL_CASE(1) ZL_CASE(2) ZL_CASE(3) ZL_CASE(4) ZL_CASE(5) ZL_CASE(6) ZL_CASE(7)
#undef ZL_CASE
    }
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_rt_loop(const ZlBatch &A, void *mailbox_dev, void *dev_state, unsigned long long first_seq, unsigned long long idle_ticks, float *gain_out,
                      ZlReport *host_reports, float *host_gain, ZlOpRange *dev_ranges, int vw, int threads, hipStream_t s)
{
    ZlRtShared *sh = reinterpret_cast<ZlRtShared *>(mailbox_dev);
    ZlRtDev *dv = reinterpret_cast<ZlRtDev *>(dev_state);
    const bool wide = A.groups > 1;                                // one workgroup per vw voices
    switch (A.mode & 7u) {
#define ZL_CASE(M) case M: \
        if (wide) hipLaunchKernelGGL((zl_k_rt_loop<M, true>), dim3(A.V / vw), dim3(threads), 0, s, A, sh, dv, first_seq, idle_ticks, gain_out, host_reports, host_gain, dev_ranges, vw); \
        else      hipLaunchKernelGGL((zl_k_rt_loop<M, false>), dim3(A.B), dim3(threads), 0, s, A, sh, dv, first_seq, idle_ticks, gain_out, host_reports, host_gain, dev_ranges, 0); \
        break;
        ZL_CASE(0) ZL_CASE(1) ZL_CASE(2) ZL_CASE(3) ZL_CASE(4) ZL_CASE(5) ZL_CASE(6) ZL_CASE(7)
#undef ZL_CASE
    }
    ZL_LAUNCH_CHECK();
    return 0;
}

// How many workgroups of the resident kernel the device holds at once (every one of them must be resident: a block is complete
// when all have arrived).  0 on error.
int zl_rt_loop_capacity(uint32_t mode, int wide, int threads, int device)
{
    int per_cu = 0, cus = 0;
    hipError_t st = hipErrorUnknown;
    switch (mode & 7u) {
#define ZL_CASE(M) case M: \
        st = wide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zl_k_rt_loop<M, true>, threads, 0) \
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, zl_k_rt_loop<M, false>, threads, 0); \
        break;
        ZL_CASE(0) ZL_CASE(1) ZL_CASE(2) ZL_CASE(3) ZL_CASE(4) ZL_CASE(5) ZL_CASE(6) ZL_CASE(7)
#undef ZL_CASE
    }
    if (st != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return per_cu * cus;
}

int zl_launch_finalize(const ZlBatch &A, const float *bus_in, hipStream_t s)
{
    // nothing to sum (K2 wrote the bus, or the caller hands one over) and 16-byte aligned rows: the wave-per-block scan
    const float *scan = bus_in ? bus_in : (A.groups == 1 ? A.bus : nullptr);
    if (scan) {
        const long long pairs = (long long)A.K * A.B;
        hipLaunchKernelGGL(zl_k3_scan, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, s, A, scan);
    } else {
        hipLaunchKernelGGL(zl_k3_finalize, dim3(A.K, A.B), dim3(A.N < 256 ? ((A.N + 63) & ~63) : 256), 0, s, A, bus_in);
    }
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_reports(const ZlReport *reports, int V, float *gain_out, ZlReport *host_reports, float *host_gain,
                      const ZlBatchStats *stats, ZlBatchStats *host_stats, hipStream_t s, hipEvent_t ev_done)
{
    // ev_done: the call's completion event rides on this dispatch (its stop event) instead of a packet of its own
    hipExtLaunchKernelGGL(zl_k_reports, dim3((V + 255) / 256), dim3(256), 0, s, nullptr, ev_done, 0, reports, V, gain_out, host_reports, host_gain, stats, host_stats);
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_levels_tick(ZlLevelsState *state, const ZlBlockLevels *levels, int B, int N, int with_hold_bus, hipStream_t s)
{
    hipLaunchKernelGGL(zl_k_levels_tick, dim3((B + 63) / 64), dim3(64), 0, s, state, levels, B, N, with_hold_bus);
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_reduce_scan(const float *pieces, int npieces, long long stride, long long units, int N, int off, float *out, ZlUnitLevels *lv, hipStream_t s)
{
    hipLaunchKernelGGL(zl_k_reduce_scan, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, pieces, npieces, stride, units, N, off, out, lv);
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_levels_import(const ZlUnitLevels *units, ZlBlockLevels *levels, int B, int K, hipStream_t s)
{
    hipLaunchKernelGGL(zl_k_levels_import, dim3((unsigned)((B * K + 255) / 256)), dim3(256), 0, s, units, levels, B, K);
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_passthrough(const void *params_dev, const float *in, float *out, int B, long long frames, hipStream_t s)
{
    const bool vec = (frames % 4) == 0 && (((uintptr_t)in | (uintptr_t)out) & 15u) == 0;
    long long nb = (frames / (vec ? 4 : 1) + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    if (vec) hipLaunchKernelGGL(zl_k_passthrough<4>, dim3((unsigned)nb, B), dim3(256), 0, s, (const ZlPassParams *)params_dev, in, out, frames);
    else     hipLaunchKernelGGL(zl_k_passthrough<1>, dim3((unsigned)nb, B), dim3(256), 0, s, (const ZlPassParams *)params_dev, in, out, frames);
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_deliver(const float *bus, void *out, int pcm16, int B, long long in_stride, long long off, long long frames, long long total, hipStream_t s)
{
    // 16-byte accesses need rows, offset and the output rows to start on multiples of four frames (the base pointers are allocations)
    const bool vec = ((in_stride | off | total) & 3) == 0 && (reinterpret_cast<uintptr_t>(bus) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    long long nb = ((vec ? frames / 4 : frames) + 255) / 256;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    if (pcm16) { if (vec) hipLaunchKernelGGL((zl_k_deliver<true, 4>), dim3((unsigned)nb, B), dim3(256), 0, s, bus, out, in_stride, off, frames, total);
                 else     hipLaunchKernelGGL((zl_k_deliver<true, 1>), dim3((unsigned)nb, B), dim3(256), 0, s, bus, out, in_stride, off, frames, total); }
    else       { if (vec) hipLaunchKernelGGL((zl_k_deliver<false, 4>), dim3((unsigned)nb, B), dim3(256), 0, s, bus, out, in_stride, off, frames, total);
                 else     hipLaunchKernelGGL((zl_k_deliver<false, 1>), dim3((unsigned)nb, B), dim3(256), 0, s, bus, out, in_stride, off, frames, total); }
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_interleave(const float *L, const float *R, float *dst, int length, int pad, hipStream_t s)
{
    hipLaunchKernelGGL(zl_k_interleave, dim3((length + pad + 255) / 256), dim3(256), 0, s, L, R, dst, length, pad);
    ZL_LAUNCH_CHECK();
    return 0;
}
