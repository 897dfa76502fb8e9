// zl_plan.h -- K1 "control plan": the per-voice control state of SamplerSynthVoice::process (reference
// lib/SamplerSynthVoice.cpp:174-270) advanced WITHOUT walking every frame, bit-exactly.
//
// The reference advances `sourceSamplePosition += pitchRatio` (fp64) once per frame (:223).  While
// P stays inside one binade and the exact sum P + r does not leave it, every such addition rounds
// to the same multiple of ulp(P), so the recurrence is an exact arithmetic progression
// P_n = P_0 + n*s with s = round_to_ulp(r) (ties handled by mantissa parity).  A voice is therefore
// described by a stream of "linear segments" {first frame t, P at that frame, s} in window time; lanes
// evaluate fma(frame - t, s, P), which is exact.  Segment boundaries are binade crossings (performed
// with one real fp64 addition) and the loop / stop events of :225-257.
//
//   K1  (one lane per voice, sequential):   ZlPlanner -- one iteration per linear run, no per-block work
//   K1c (one lane per voice x block chunk): ZlAssembler -- segment stream -> per-block plan records; blocks with
//                                           more than two segments -> per-frame control, the wave's lanes over frames
//
// Blocks whose envelope is not in steady sustain (attack, decay, release tail of quirk Q7, note-off)
// are simulated per frame by zl_sim_block(), which records (P, env) per frame for K2.
//
// Everything here is __host__ __device__ so the identical code is unit-tested on the CPU
// (tests/cpu_harness) against the oracle; the product only ever runs it inside HIP kernels.
#pragma once
#include "zl_types.h"
#include <math.h>
#include <limits.h>

#define ZL_INF_STEPS 0x7fffffff
#define ZL_RUN_CAP   0x3fffffff

ZL_HD inline uint64_t zl_bits(double d) { return __builtin_bit_cast(uint64_t, d); }
ZL_HD inline double   zl_from_bits(uint64_t u) { return __builtin_bit_cast(double, u); }

// float -> quint64 as on the reference's aarch64 target (fcvtzu): saturating, NaN -> 0
ZL_HD inline uint64_t zl_f32_to_u64_sat(float f)
{
    if (!(f > 0.0f)) return 0;
    if (f >= 18446744073709551616.0f) return ~0ull;
    return (uint64_t)f;
}
ZL_HD inline uint64_t zl_f64_to_u64_sat(double f)
{
    if (!(f > 0.0)) return 0;
    if (f >= 18446744073709551616.0) return ~0ull;
    return (uint64_t)f;
}

// ---- ClipAudioSource getters (ClipAudioSource.cpp:261-277) -----------------------------------
ZL_HD inline float zl_clip_start(const ZlClip &c, int slice)
{
    if (slice > -1 && slice < c.n_slice_pos)
        return (float)((double)c.start_sec + ((double)c.length_sec * c.slice_pos[slice]));
    return c.start_sec;
}
ZL_HD inline float zl_clip_stop(const ZlClip &c, int slice)
{
    if (slice > -1 && slice + 1 < c.n_slice_pos)
        return (float)((double)c.start_sec + ((double)c.length_sec * c.slice_pos[slice + 1]));
    return c.start_sec + c.length_sec;
}

// ---- juce::ADSR on the voice record ----------------------------------------------------------
ZL_HD inline void zl_adsr_reset(ZlVoiceState &s) { s.env = 0.0f; s.adsr_state = ZL_ADSR_IDLE; }

ZL_HD inline void zl_adsr_note_off(ZlVoiceState &s)
{
    if (s.adsr_state != ZL_ADSR_IDLE) {
        if (s.release > 0.0f) {
            s.release_rate = (float)((double)s.env / ((double)s.release * s.adsr_sr));
            s.adsr_state = ZL_ADSR_RELEASE;
        } else {
            zl_adsr_reset(s);
        }
    }
}

ZL_HD inline float zl_adsr_next(ZlVoiceState &s)
{
    switch (s.adsr_state) {
    case ZL_ADSR_IDLE:
        return 0.0f;
    case ZL_ADSR_ATTACK:
        s.env += s.attack_rate;
        if (s.env >= 1.0f) { s.env = 1.0f; s.adsr_state = (s.decay_rate > 0.0f ? ZL_ADSR_DECAY : ZL_ADSR_SUSTAIN); }
        break;
    case ZL_ADSR_DECAY:
        s.env -= s.decay_rate;
        if (s.env <= s.sustain) { s.env = s.sustain; s.adsr_state = ZL_ADSR_SUSTAIN; }
        break;
    case ZL_ADSR_SUSTAIN:
        s.env = s.sustain;
        break;
    case ZL_ADSR_RELEASE:
        s.env -= s.release_rate;
        if (s.env <= 0.0f) zl_adsr_reset(s);
        break;
    }
    return s.env;
}

// stopNote(velocity, false), SamplerSynthVoice.cpp:153-168
ZL_HD inline void zl_voice_hard_stop(ZlVoiceState &s)
{
    zl_adsr_reset(s);
    s.clip = -1;
    s.playing = 0;
    s.next_loop_tick = 0;
    s.next_loop_usecs = 0;
}

// ---- exact linear runs of P += r --------------------------------------------------------------
// Exponent of the lowest set bit of a finite positive double (the value is an odd multiple of 2^result).
ZL_HD inline int zl_lsb_exponent(double x)
{
    const uint64_t b = zl_bits(x);
    const int ex = (int)((b >> 52) & 0x7ff);
    const uint64_t m = (b & 0xfffffffffffffull) | (ex ? (1ull << 52) : 0ull);
    if (m == 0) return 4096;                                     // zero: a multiple of every power of two
    return (ex ? ex : 1) - 1075 + __builtin_ctzll(m);
}

// Returns the per-step increment s and the number L >= 0 of consecutive additions starting at P
// that are guaranteed to equal P + i*s exactly (i = 1..L).  L == 0 means "take one real addition".
// inv_r = 1 / r (computed once per voice and batch); it only shortens runs conservatively.
ZL_HD inline void zl_linear_run(double P, double r, double inv_r, double &s, int &L)
{
    // ---- exact-granule run: when P and r are both multiples of 2^g, every sum P + i*r is a multiple of 2^g and is
    //      exactly representable (no rounding at all, across binades) while it stays below 2^(53+g).  Covers playback
    //      at the source rate (r = 1), octaves and other dyadic ratios from integer loop starts.
    if (P >= 0.0 && r > 0.0 && P < 0x1p62 && r < 0x1p62) {
        const int gp = zl_lsb_exponent(P), gr = zl_lsb_exponent(r);
        const int g = gp < gr ? gp : gr;
        if (g > -1000 && g + 53 < 1000) {
            const double limit = zl_from_bits((uint64_t)(g + 53 + 1023) << 52);      // 2^(53+g)
            const double room = limit - P;                                           // exact (multiples of 2^g below 2^(53+g))
            if (room >= r) {
                double c0 = floor((room * inv_r) * (1.0 - 0x1p-30));
                if (c0 > (double)ZL_RUN_CAP) c0 = (double)ZL_RUN_CAP;
                for (int it = 0; it < 4 && c0 > 0.0 && fma(c0, r, -room) > 0.0; ++it) c0 -= 1.0;
                if (c0 >= 1024.0 && !(fma(c0, r, -room) > 0.0)) { s = r; L = (int)c0; return; }
            }
        }
    }
    s = 0.0; L = 0;
    const uint64_t pb = zl_bits(P);
    const int ex = (int)((pb >> 52) & 0x7ff);
    if ((pb >> 63) || ex <= 53 || ex >= 0x7fd) return;           // zero / tiny / huge / negative: step for real
    const double u   = zl_from_bits((uint64_t)(ex - 52) << 52);  // ulp(P) = 2^(e-52)
    const double top = zl_from_bits((uint64_t)(ex + 1) << 52);   // 2^(e+1)
    if (!(r > 0.0)) return;
    const double inv_u = zl_from_bits((uint64_t)(2046 - (ex - 52)) << 52);   // 1 / ulp(P), a power of two
    const double rq = r * inv_u;                                 // exact power-of-two scaling (or +inf)
    const double q  = floor(rq);
    if (!(q < 9007199254740992.0)) return;                       // r >= 2^53 ulps: leaves the binade at once
    const double rho  = r - q * u;                               // r mod u, exact
    const double half = 0.5 * u;
    double c;
    if (rho > half) c = 1.0;
    else if (rho < half) c = 0.0;
    else {                                                       // tie: round-half-even on the mantissa
        if (pb & 1ull) return;                                   // odd mantissa: one real step makes it even
        c = (((long long)q) & 1ll) ? 1.0 : 0.0;                  // even mantissa stays even from here on
    }
    const double sq = q + c;                                     // step in ulps
    s = sq * u;                                                  // exact
    if (sq <= 0.0) { L = ZL_RUN_CAP; return; }                   // r < ulp/2: P never moves
    // step i (0-based) starts at mantissa m_i = m0 + i*sq and is an in-binade step iff its result m_i + sq <= 2^53,
    // i.e. it lands inside the binade or exactly on 2^(e+1): the exact sum is then below 2^(e+1) + u/2 (c = 0) or
    // below 2^(e+1) (c = 1), where rounding to the spacing u (resp. to the top itself) gives m_i + sq.
    const double a = (top - P) - s;                              // exact
    if (a < 0.0) { s = 0.0; return; }
    // count = floor(a / s) + 1, estimated without a division (inv_r ~ 1 / s) and then corrected downwards
    // exactly: fma(c, s, -a) has the sign of the exact c*s - a
    double c0 = floor((a * inv_r) * (1.0 - 0x1p-30));
    if (!(c0 >= 0.0)) c0 = 0.0;
    for (int it = 0; c0 > 0.0 && fma(c0, s, -a) > 0.0; ++it) {
        if (it >= 4) {                                           // estimate far off (r tiny against P): divide once
            c0 = floor(a / s);
            while (c0 > 0.0 && fma(c0, s, -a) > 0.0) c0 -= 1.0;
            break;
        }
        c0 -= 1.0;
    }
    const double cnt = c0 + 1.0;
    L = cnt > (double)ZL_RUN_CAP ? ZL_RUN_CAP : (int)cnt;
}

// Envelope ramps (juce::ADSR attack / decay / release) are fp32 recurrences e <- fl32(e + d) with a constant d.  Inside
// one binade of e, and while the exact sum stays in it, every step adds the same multiple of ulp(e), so the values are
// an exact arithmetic progression e + i * es.  Returns es and the number Lc >= 0 of steps i = 1..Lc that are (a)
// exactly that progression and (b) do not reach `limit` (d > 0: e_i < limit, d < 0: e_i > limit), i.e. cause no ADSR
// state change.  Lc == 0: take one real step (zl_adsr_next).
ZL_HD inline void zl_env_linear_run(float e, float d, float limit, float &es, int &Lc)
{
    es = 0.0f; Lc = 0;
    const uint32_t eb = __builtin_bit_cast(uint32_t, e);
    const int ex = (int)((eb >> 23) & 0xffu);
    if ((eb >> 31) || ex == 0 || ex >= 0xfe || !(d == d) || d == 0.0f) return;   // e <= 0, subnormal, huge; no ramp
    const double u = zl_from_bits((uint64_t)(ex - 150 + 1023) << 52);            // ulp(e)
    const double inv_u = zl_from_bits((uint64_t)(150 - ex + 1023) << 52);
    const double m = (double)e * inv_u;                          // mantissa as an integer in [2^23, 2^24)
    const bool up = d > 0.0f;
    const double rq = fabs((double)d) * inv_u;                   // |d| in ulps, exact
    const double q = floor(rq);
    if (!(q < 16777216.0)) return;                               // leaves the binade at once
    const double fr = rq - q;
    double c;
    if (fr > 0.5) c = q + 1.0;
    else if (fr < 0.5) c = q;
    else {                                                       // tie: round-half-even on the mantissa
        if (eb & 1u) return;                                     // odd mantissa: one real step makes it even
        c = (((long long)q) & 1ll) ? q + 1.0 : q;                // even mantissa stays even from here on
    }
    if (c <= 0.0) {                                              // |d| <= ulp/2: the envelope does not move (es = 0) ...
        if (up || m - rq >= 8388608.0) Lc = ZL_RUN_CAP;           // ... unless it sits on the binade's bottom, going down
        return;
    }
    // steps that stay inside the binade: up, the result m_i + c may reach 2^24 (the top itself); down, the EXACT
    // difference m_i - rq must not fall below 2^23 (below it the spacing halves and the rounding changes)
    const double num = up ? 16777216.0 - m : m - 8388608.0 - rq + c;   // nb = floor(num / c), both cases
    double nb = 0.0;
    if (num >= c) {
        nb = floor(num / c);
        while (nb > 0.0 && fma(nb, c, -num) > 0.0) nb -= 1.0;
        while (fma(nb + 1.0, c, -num) <= 0.0) nb += 1.0;
    }
    // steps before the ADSR event: the values e + i * es are exact in double
    const double esd = up ? c * u : -(c * u);
    const double gap = up ? (double)limit - (double)e : (double)e - (double)limit;      // > 0 while the state lasts
    double nev = 0.0;
    if (gap > 0.0) {
        const double step = c * u;
        nev = ceil(gap / step) - 1.0;                            // largest i with i * step < gap, then corrected exactly
        if (nev < 0.0) nev = 0.0;
        while (nev > 0.0 && !(fma(nev, step, -gap) < 0.0)) nev -= 1.0;
        while (fma(nev + 1.0, step, -gap) < 0.0) nev += 1.0;
    }
    double n = nb < nev ? nb : nev;
    if (n > (double)ZL_RUN_CAP) n = (double)ZL_RUN_CAP;
    es = (float)esd;                                             // exact: c < 2^24 ulps
    Lc = (int)n;
}

// Smallest i in [1, L] with P + i*s >= X (values are exact), or ZL_INF_STEPS.
ZL_HD inline int zl_steps_to_reach(double P, double s, double inv_r, int L, double X)
{
    if (L < 1) return ZL_INF_STEPS;
    if (!(fma((double)L, s, P) >= X)) return ZL_INF_STEPS;
    if (P + s >= X) return 1;
    double g = ceil((X - P) * inv_r);                          // estimate; the exact search below corrects it
    if (!(g >= 1.0)) g = 1.0;
    if (g > (double)L) g = (double)L;
    int i = (int)g;
    while (i > 1 && fma((double)(i - 1), s, P) >= X) --i;
    while (i < L && fma((double)i, s, P) < X) ++i;
    return i;
}

// First frame f in [n, N) after whose rendering the beat-locked loop test of
// SamplerSynthVoice.cpp:232 fires: current_usecs + (u64)(f * usecsPerFrame) >= nextLoopUsecs.
// Returns N if it does not fire in this block.  Requires usecs_per_frame < 2^21 (caller checks).
ZL_HD inline int zl_clock_event_frame(const ZlClock &ck, uint64_t next_loop_usecs, int n, int N)
{
    const uint64_t U = ck.usecs_per_frame;
    if (ck.current_usecs + (uint64_t)n * U >= next_loop_usecs) return n;
    if (U == 0) return N;
    if (ck.current_usecs + (uint64_t)(N - 1) * U < next_loop_usecs) return N;
    const uint64_t D = next_loop_usecs - ck.current_usecs;
    const uint64_t f = (D + U - 1) / U;
    return f < (uint64_t)N ? (int)f : N;
}

struct ZlVoiceBatchConst {
    int      start_int;       // (int)(getStartPosition(slice) * sourceSampleRate), :241/:246
    int      stop_pos;        // SamplerSynthSound::stopPosition(slice), :190
    double   tail_T;          // stopPosition - release * sourceSampleRate, :253
    uint64_t length_ticks;    // (quint64)(lengthInBeats * multiplier), :234
    int      beat_locked;     // trunc(lengthInBeats) == lengthInBeats, :227
    int      clock_ok;        // usecs-per-frame small enough for the exact integer clock test
};

ZL_HD inline void zl_loop_restart(ZlVoiceState &st, const ZlVoiceBatchConst &c, const ZlClock &ck, bool clockMode)
{
    if (clockMode) {                                             // :234-237
        st.next_loop_tick = st.next_loop_tick + c.length_ticks;
        st.next_loop_usecs = ck.playhead_usecs + ((st.next_loop_tick - ck.playhead) * ck.subbeat_usecs);
    }
    st.P = (double)c.start_int;                                  // :241 / :246
}

// One frame of the reference loop minus the audio arithmetic (:223-261): advances the position and the loop / stop logic
// after frame `frame` was rendered.  Returns false when the voice stopped with that frame.
ZL_HD inline bool zl_sim_step(ZlVoiceState &st, const ZlVoiceBatchConst &c, const ZlClock &ck, double upf, int frame)
{
    st.P += st.pitch_ratio;                                      // :223
    if (st.looping) {
        if (c.beat_locked) {
            if (ck.current_usecs + zl_f64_to_u64_sat((double)(uint32_t)frame * upf) >= st.next_loop_usecs)   // :232
                zl_loop_restart(st, c, ck, true);
        } else if (st.P >= (double)c.stop_pos) {                 // :243
            zl_loop_restart(st, c, ck, false);
        }
    } else {
        if (st.P >= (double)c.stop_pos) {                        // :249-252
            zl_voice_hard_stop(st);
            return false;
        } else if (st.P >= c.tail_T) {                           // :253-256 (Q7: every frame)
            zl_adsr_note_off(st);
        }
    }
    if (st.adsr_state == ZL_ADSR_IDLE) {                         // :258-261
        zl_voice_hard_stop(st);
        return false;
    }
    return true;
}

// Per-frame simulation of one block.  Records (P, env) of every rendered frame (ctlP == nullptr: only advances the voice);
// returns the number of frames rendered.
ZL_HD inline int zl_sim_block(ZlVoiceState &st, const ZlVoiceBatchConst &c, const ZlClock &ck, int N,
                              double *ctlP, float *ctlEnv)
{
    const double upf = (double)ck.usecs_per_frame;               // :183
    for (int frame = 0; frame < N; ++frame) {
        const double P = st.P;
        const float env = zl_adsr_next(st);                      // :201
        if (ctlP) { ctlP[frame] = P; ctlEnv[frame] = env; }
        if (!zl_sim_step(st, c, ck, upf, frame)) return frame + 1;
    }
    return N;
}

// ---- the window's pool of per-frame control slots ------------------------------------------------------------------------
// One slot = the (P, env) of the N frames of one slow (block, voice).  K1 (simulated blocks) and K1c (multi-segment blocks)
// take slots with a bump counter that is never reset: a window's slots count from A.ctl_base, which the host raises past every
// value the counter can have reached (zl_engine.cpp).  -1: the pool is exhausted -- the block then carries a snapshot and K2
// recomputes its control (zl_slow_control).
ZL_HD inline int zl_ctl_alloc(const ZlBatch &A)
{
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax(A.ctl_next, A.ctl_base);
    const unsigned long long old = atomicAdd(A.ctl_next, 1ull);
#else
    if (*A.ctl_next < A.ctl_base) *A.ctl_next = A.ctl_base;
    const unsigned long long old = (*A.ctl_next)++;
#endif
    const unsigned long long sl = old - A.ctl_base;
    return sl < (unsigned long long)(A.ctl_slots > 0 ? A.ctl_slots : 0) ? (int)sl : -1;
}

// The voice state at the start of a simulated block, packed into the block's plan record (seg0 + seg1 = 48 bytes): everything
// of ZlVoiceState that moves inside a window.
ZL_HD inline void zl_snapshot_store(ZlBlockPlan &pl, const ZlVoiceState &st)
{
    pl.P0 = st.P;
    pl.step = zl_from_bits(st.next_loop_tick);
    pl.P1 = zl_from_bits(st.next_loop_usecs);
    pl.step1 = zl_from_bits((uint64_t)__builtin_bit_cast(uint32_t, st.env) | ((uint64_t)__builtin_bit_cast(uint32_t, st.release_rate) << 32));
    pl.n1 = st.adsr_state;
}
ZL_HD inline void zl_snapshot_load(const ZlBlockPlan &pl, const ZlSimConst &sc, ZlVoiceState &st, ZlVoiceBatchConst &c)
{
    st.P = pl.P0;
    st.next_loop_tick = zl_bits(pl.step);
    st.next_loop_usecs = zl_bits(pl.P1);
    const uint64_t er = zl_bits(pl.step1);
    st.env = __builtin_bit_cast(float, (uint32_t)er);
    st.release_rate = __builtin_bit_cast(float, (uint32_t)(er >> 32));
    st.adsr_state = pl.n1;
    st.pitch_ratio = sc.pitch_ratio; st.adsr_sr = sc.adsr_sr; st.src_len = 0.0;
    st.attack_rate = sc.attack_rate; st.decay_rate = sc.decay_rate; st.sustain = sc.sustain; st.release = sc.release;
    st.lgain = 0.0f; st.rgain = 0.0f; st.clip = 0; st.slice = 0; st.looping = sc.looping; st.playing = 1; st.loop_phase1 = 0;
    c.start_int = sc.start_int; c.stop_pos = sc.stop_pos; c.tail_T = sc.tail_T; c.length_ticks = sc.length_ticks;
    c.beat_locked = sc.beat_locked; c.clock_ok = 1;
}

struct ZlPlanStats { unsigned long long source_bytes, slow_blocks, active_frames; };

ZL_HD inline void zl_plan_clear(ZlBlockPlan &pl)
{
    pl.flags = 0; pl.n_active = 0; pl.nseg = 0; pl.env = 0.0f; pl.P0 = 0.0; pl.step = 0.0;
    pl.n1 = INT_MAX; pl.estep0 = 0.0f; pl.P1 = 0.0; pl.step1 = 0.0; pl.E1 = 0.0f; pl.estep1 = 0.0f;
}

ZL_HD inline void zl_plan_store(const ZlBatch &A, size_t pidx, const ZlBlockPlan &pl)
{
    ZlPlanHdr h; h.flags = pl.flags; h.n_active = pl.n_active; h.nseg = pl.nseg; h.env = pl.env;
    A.plan_hdr[pidx] = h;
    ZlPlanSeg0 s0; s0.P0 = pl.P0; s0.step = pl.step;
    A.plan_seg0[pidx] = s0;
    if (pl.nseg >= 2 || (pl.flags & (ZL_PLAN_ENV | ZL_PLAN_NOSLOT_SIM))) {
        ZlPlanSeg1 s1; s1.P1 = pl.P1; s1.step1 = pl.step1; s1.n1 = pl.n1; s1.estep0 = pl.estep0; s1.E1 = pl.E1; s1.estep1 = pl.estep1;
        A.plan_seg1[pidx] = s1;
    }
}

ZL_HD inline ZlBlockPlan zl_plan_load(const ZlBatch &A, size_t pidx)
{
    ZlBlockPlan pl;
    zl_plan_clear(pl);
    // header and first segment are independent loads (one round trip); the second segment only where there is one
    const ZlPlanHdr h = A.plan_hdr[pidx];
    const ZlPlanSeg0 s0 = A.plan_seg0[pidx];
    pl.flags = h.flags; pl.n_active = h.n_active; pl.nseg = h.nseg; pl.env = h.env;
    pl.P0 = s0.P0; pl.step = s0.step;
    if (((h.nseg >= 2 || (h.flags & ZL_PLAN_ENV)) && !(h.flags & ZL_PLAN_SLOW)) || (h.flags & (ZL_PLAN_NOSLOT_SIM | ZL_PLAN_NOSLOT_EXPAND))) {
        const ZlPlanSeg1 s1 = A.plan_seg1[pidx];
        pl.P1 = s1.P1; pl.step1 = s1.step1; pl.n1 = s1.n1; pl.estep0 = s1.estep0; pl.E1 = s1.E1; pl.estep1 = s1.estep1;
    }
    return pl;
}

// floor(t / N) for 0 <= t < 2^31 without an integer division (invN = 1.0 / N)
ZL_HD inline int zl_block_of(int t, int N, double invN)
{
    int k = (int)((double)t * invN);
    if ((k + 1) * N <= t) ++k;
    if (k * N > t) --k;
    return k;
}

// Plans a window for one voice and leaves the voice state as the reference would after rendering it.
// The planner is a flat state machine over window time t: one iterate() call = at most one new linear run
// (zl_linear_run + the search for the loop / stop event inside it), emitted as ONE segment whatever number of
// blocks it spans.  All lanes of a wavefront execute the same straight-line code per iteration whatever their
// positions in the window are; everything per block is left to K1c, which is lane-parallel.
struct ZlPlanner {
    ZlVoiceState st;
    ZlVoiceBatchConst c;
    ZlPlanStats stats;
    unsigned long long blockBytes;
    double X;                    // threshold of the position event that can occur in a fast (sustain) block
    double inv_r, invN;          // 1 / pitch_ratio, 1 / N
    double s;                    // current linear run: step, linear steps left, steps to the position event
    int L, ie;
    // current envelope run: frame t has envelope eNext, the following eLeft - 1 frames add es each (exactly, fp32)
    float eNext, es;
    int eLeft;
    bool haveERun, endAfter;     // endAfter: the envelope reached idle at frame t, the voice stops after rendering it
    int v;
    bool valid, posMode, clockMode, haveRun, slowNext, lastMarker;
    int t;                       // next frame to plan (window time); its position is st.P
    int nts;                     // segments emitted
    int t_end;                   // frame at which the voice stopped (INT_MAX while it plays)
    int dead_from;
    // state at the start of the block that contains frame t (restored when that block has to be simulated after all)
    double Pbs; uint64_t tick_bs, usecs_bs; int jbs; float env_bs; int adsr_bs;
    // inline runs: the open one lives in registers and is stored when the next one opens
    ZlRun cur;
    int  nruns;
    bool haveCur;
    // periodic sample-space loops: 0 = waiting for a restart, 1 = capturing the pass that started at tcap (segment
    // jcap), 2 = done / given up
    int perState, tcap, jcap;
    int per_t0, per_M, per_j0, per_n;
    // loop phase (frames since the last restart) for the pass cache: last restart seen in this window, or lost
    int lastRestartT; bool phaseLost, phaseSet;

    ZL_HD void begin(const ZlBatch &A, int voice, int force_slow = 0)
    {
        v = voice;
        lastRestartT = -1; phaseLost = false; phaseSet = false;
        perState = 0; tcap = 0; jcap = 0; per_t0 = 0; per_M = 0; per_j0 = 0; per_n = 0;
        st = A.voices[v];
        stats.source_bytes = 0; stats.slow_blocks = 0; stats.active_frames = 0;
        nruns = 0; haveCur = false; cur.P = 0.0; cur.step = 0.0; cur.k0 = 0; cur.k1 = 0;
        s = 0.0; L = 0; ie = ZL_INF_STEPS; haveRun = false; slowNext = false; lastMarker = false;
        eNext = 0.0f; es = 0.0f; eLeft = 0; haveERun = false; endAfter = false; env_bs = 0.0f; adsr_bs = 0;
        t = 0; nts = 0; t_end = INT_MAX; dead_from = 0;
        Pbs = 0.0; tick_bs = 0; usecs_bs = 0; jbs = 0;
        invN = 1.0 / (double)A.N;
        valid = st.playing == 1 && st.clip >= 0 && A.sounds[st.clip].channels > 0;   // (2: its channel is disabled -- idle for this window, state kept)
        posMode = false; clockMode = false; X = INFINITY; blockBytes = 0; inv_r = 0.0;
        if (!valid) {
            // a neutral record: K2 stages every voice slot of a bus and must find addressable constants in it
            ZlVoiceConst vc;
            vc.src_offset = 0; vc.sample_duration = 0; vc.channels = 2;
            vc.lgain = vc.rgain = vc.clip_volume = vc.lpan = vc.rpan = vc.env = 0.0f; vc.pad[0] = vc.pad[1] = 0;
            A.vconst[v] = vc;
            return;
        }
        dead_from = A.K;
        inv_r = 1.0 / st.pitch_ratio;
        const ZlClip &cl = A.clips[st.clip];
        const ZlSound sd = A.sounds[st.clip];
        const double sr = sd.sample_rate;
        c.start_int = (int)(zl_clip_start(cl, st.slice) * sr);
        c.stop_pos  = (int)(zl_clip_stop(cl, st.slice) * sr);
        c.tail_T    = (double)c.stop_pos - ((double)st.release * sr);
        c.length_ticks = zl_f32_to_u64_sat(cl.length_beats * (float)ZL_BEAT_SUBDIV);
        c.beat_locked = truncf(cl.length_beats) == cl.length_beats;
        c.clock_ok = 1;

        if (A.sim_const) {
            ZlSimConst sc;
            sc.pitch_ratio = st.pitch_ratio; sc.adsr_sr = st.adsr_sr; sc.tail_T = c.tail_T; sc.length_ticks = c.length_ticks;
            sc.attack_rate = st.attack_rate; sc.decay_rate = st.decay_rate; sc.sustain = st.sustain; sc.release = st.release;
            sc.start_int = c.start_int; sc.stop_pos = c.stop_pos; sc.beat_locked = c.beat_locked; sc.looping = st.looping;
            A.sim_const[v] = sc;
        }
        ZlVoiceConst vc;
        vc.src_offset = sd.offset;
        vc.sample_duration = sd.length - 1;                       // :191
        vc.channels = sd.channels;
        vc.lgain = st.lgain; vc.rgain = st.rgain;
        vc.clip_volume = cl.volume_abs;                           // :189
        vc.lpan = (float)(0.5 * (1.0 + (double)cl.pan));          // :193
        vc.rpan = (float)(0.5 * (1.0 - (double)cl.pan));          // :194
        vc.env = st.sustain;
        vc.pad[0] = 0; vc.pad[1] = 0;
        A.vconst[v] = vc;

        // algorithmic source bytes of one block of this voice
        const int taps = (A.mode & ZL_MODE_HERMITE) ? 4 : 2;
        blockBytes = (unsigned long long)((long long)ceil((double)A.N * st.pitch_ratio) + taps - 1) * (unsigned long long)sd.channels * 4ull;

        const bool posLoop = st.looping && !c.beat_locked;
        const bool oneShot = !st.looping;
        clockMode = st.looping && c.beat_locked;
        X = posLoop ? (double)c.stop_pos : (oneShot ? ((st.release > 0.0f) ? c.tail_T : (double)c.stop_pos) : INFINITY);
        posMode = posLoop || oneShot;
        if (posLoop && !force_slow && A.pass_cache) replay_cached_pass(A);
    }

    // The steady state of a sample-space loop costs no planning at all: when the voice sits ON the pass recorded for its
    // (start, stop, ratio, sustain) -- same position, bit for bit, at the recorded offset -- its future is that pass
    // from that offset, repeated.  The window's stream is the cached pass shifted back by the offset (per_t0 < 0); the
    // state at the window's end comes from (offset + window) mod M.  Anything else falls through to the planner.
    ZL_HD void replay_cached_pass(const ZlBatch &A)
    {
        const ZlPassCache &pc = A.pass_cache[v];
        const int r = st.loop_phase1 - 1;
        if (!(pc.valid && r >= 0 && r < pc.M && st.adsr_state == ZL_ADSR_SUSTAIN && st.env == st.sustain)) return;
        if (!(pc.ratio == st.pitch_ratio && pc.start_int == c.start_int && pc.stop_pos == c.stop_pos && pc.sustain == st.sustain)) return;
        const long long TW = (long long)A.K * A.N;
        // passes of one or two segments are cheaper as inline runs, unless the window holds more of them than the list
        if (!(pc.n >= 3 || TW / pc.M >= ZL_MAXRUNS)) return;
        int j = pc.n - 1;
        while (j > 0 && pc.seg[j].t > r) --j;
        if (!(fma((double)(r - pc.seg[j].t), pc.seg[j].step, pc.seg[j].P) == st.P)) return;   // not on the recorded pass
        ZlTSeg *ts = A.tsegs + (size_t)v * ZL_MAXTSEG;
        for (int i = 0; i < pc.n; ++i) { ZlTSeg e = pc.seg[i]; e.t -= r; ts[i] = e; }
        nts = pc.n; per_t0 = -r; per_M = pc.M; per_j0 = 0; per_n = pc.n; perState = 2;
        const ZlClock &ck = A.inline_clock ? A.clock0 : A.clocks[0];
        if (st.next_loop_usecs == 0)                                   // :179-182, at the start of the first block
            st.next_loop_usecs = ck.playhead_usecs + ((st.next_loop_tick - ck.playhead) * ck.subbeat_usecs);
        const int re = (int)(((long long)r + TW) % pc.M);
        int je = pc.n - 1;
        while (je > 0 && pc.seg[je].t > re) --je;
        st.P = fma((double)(re - pc.seg[je].t), pc.seg[je].step, pc.seg[je].P);   // exact: on the segment's line
        st.env = st.sustain;
        st.loop_phase1 = re + 1; phaseSet = true;
        stats.active_frames += (unsigned long long)TW;
        stats.source_bytes += blockBytes * (unsigned long long)A.K;
        t = (int)TW;
    }

    // The block that starts at tb (= the block containing t) has to be simulated per frame after all: forget what
    // was planned inside it.
    ZL_HD void rollback_block(int tb)
    {
        if (t > tb) {
            stats.active_frames -= (unsigned long long)(t - tb);
            stats.source_bytes -= blockBytes;
            nts = jbs;
            t = tb;
        }
        // the state at the start of the block (the envelope may have taken its step for frame t already)
        st.P = Pbs; st.next_loop_tick = tick_bs; st.next_loop_usecs = usecs_bs;
        st.env = env_bs; st.adsr_state = adsr_bs;
        if (perState == 1) perState = 2;
        haveRun = false; haveERun = false; endAfter = false;
        slowNext = true;
    }

    // One iteration.  clk0 is the clock of block kb; frames below kend * N may be planned.  Call while t < kend * N.
    // force_slow: bit 0 = simulate every block per frame, bit 1 = do not use the periodicity of loops (test hooks).
    ZL_HD void iterate(const ZlBatch &A, int kend, const ZlClock *clk0, int kb, int force_slow)
    {
        const int N = A.N;
        const int Tend = kend * N;
        if (!(valid && st.playing)) { t = Tend; return; }          // idle blocks are implied by ZlRunList::dead_from
        const int kcur = zl_block_of(t, N, invN);
        const int tb = kcur * N;
        const int n = t - tb;
        ZlTSeg *ts = A.tsegs + (size_t)v * ZL_MAXTSEG;
        if (n == 0) {
            // ---- start of block kcur ----
            const ZlClock &ck = clk0[kcur - kb];
            if (st.next_loop_usecs == 0)                            // :179-182
                st.next_loop_usecs = ck.playhead_usecs + ((st.next_loop_tick - ck.playhead) * ck.subbeat_usecs);
            // (attack / decay / release ramps are planned as exact fp32 runs; only the every-frame noteOff of a
            // one-shot's tail, quirk Q7, needs the per-frame path -- it arrives here through slowNext)
            const bool slow = (force_slow & 1) || slowNext || (clockMode && ck.usecs_per_frame >= (1ull << 21)) || nts >= ZL_MAXTSEG - 1
                              || (!st.looping && st.release > 0.0f && st.P >= c.tail_T);
            if (slow) {
                // envelope transient, release tail (Q7) or a pathological clock: per-frame simulation of this block
                const size_t pidx = (size_t)kcur * A.V + v;
                slowNext = false;
                if (perState == 1) perState = 2;                    // a simulated block inside the pass being captured
                if (!lastMarker && nts < ZL_MAXTSEG) {              // consecutive simulated blocks share one marker
                    ZlTSeg m; m.P = st.P; m.step = 0.0; m.t = t; m.flags = ZL_TSEG_SLOW;
                    ts[nts++] = m;
                    lastMarker = true;
                }
                ZlBlockPlan pl;
                zl_plan_clear(pl);
                pl.flags = ZL_PLAN_ACTIVE | ZL_PLAN_SLOW; pl.env = st.sustain; pl.P0 = st.P;
                const int slot = zl_ctl_alloc(A);
                if (slot >= 0) {
                    pl.step = (double)slot;                         // where K2 finds the block's control
                    pl.n_active = zl_sim_block(st, c, ck, N, A.ctl_P + (size_t)slot * (size_t)N, A.ctl_env + (size_t)slot * (size_t)N);
                } else {
                    pl.flags |= ZL_PLAN_NOSLOT_SIM;                 // pool exhausted: K2 re-simulates from this state
                    zl_snapshot_store(pl, st);
                    pl.n_active = zl_sim_block(st, c, ck, N, nullptr, nullptr);
                }
                haveRun = false; haveERun = false; endAfter = false;
                zl_plan_store(A, pidx, pl);
                stats.slow_blocks += 1;
                phaseLost = true;                                   // restarts inside a simulated block are not seen here
                stats.source_bytes += blockBytes;
                stats.active_frames += (unsigned long long)pl.n_active;
                if (!st.playing) { t_end = t + pl.n_active; dead_from = kcur + 1; }
                t += N;
                return;
            }
            Pbs = st.P; tick_bs = st.next_loop_tick; usecs_bs = st.next_loop_usecs; jbs = nts;
            env_bs = st.env; adsr_bs = st.adsr_state;
        }

        // ---- a new linear position run and / or a new envelope run when the previous one is used up: one segment ----
        if (!haveRun || !haveERun) {
            if (nts >= ZL_MAXTSEG - 1) { rollback_block(tb); return; }   // table full (one slot is kept for a marker): simulate this block
            if (!haveRun) {
                zl_linear_run(st.P, st.pitch_ratio, inv_r, s, L);
                haveRun = true;
                // the event inside the run's L linear steps, else at the real addition that closes it (step L + 1)
                ie = ZL_INF_STEPS;
                if (posMode) {
                    ie = zl_steps_to_reach(st.P, s, inv_r, L, X);
                    if (ie == ZL_INF_STEPS && (fma((double)L, s, st.P) + st.pitch_ratio) >= X) ie = L + 1;
                }
            }
            if (!haveERun) {
                // the envelope of frame t by one real juce::ADSR step (:201), then its exact linear continuation: no state
                // change and no change of the fp32 spacing inside it (sustain: constant for ever)
                eNext = zl_adsr_next(st);
                es = 0.0f; eLeft = 1; endAfter = false;
                if (st.adsr_state == ZL_ADSR_IDLE) endAfter = true;                    // :258-261, the voice stops after this frame
                else if (st.adsr_state == ZL_ADSR_SUSTAIN) eLeft = (eNext == st.sustain) ? ZL_RUN_CAP : 1;   // (the frame that enters sustain still has the ramp's last value)
                else {
                    int Lc = 0;
                    if (st.adsr_state == ZL_ADSR_ATTACK)       zl_env_linear_run(eNext, st.attack_rate, 1.0f, es, Lc);
                    else if (st.adsr_state == ZL_ADSR_DECAY)   zl_env_linear_run(eNext, -st.decay_rate, st.sustain, es, Lc);
                    else                                       zl_env_linear_run(eNext, -st.release_rate, 0.0f, es, Lc);
                    eLeft = Lc + 1;
                }
                haveERun = true;
                if (st.adsr_state != ZL_ADSR_SUSTAIN && perState == 1) perState = 2;   // a ramping pass is not a template
            }
            ZlTSeg sg; sg.P = st.P; sg.step = s; sg.t = t; sg.flags = 0; sg.E = eNext; sg.estep = es;
            ts[nts++] = sg;
            lastMarker = false;
        }

        // ---- how far this iteration goes: the run (L linear steps + 1 real addition), an event, the end of the chunk ----
        const int S = L + 1;
        int room = Tend - t;
        if (eLeft < room) room = eLeft;                              // ... or the end of the envelope run
        const int maxsteps = S < room ? S : room;
        int steps = maxsteps;
        bool event = false;
        int ke = kcur;                                              // block in which the event fires
        if (posMode) {
            if (ie <= maxsteps) {
                if (!st.looping && st.release > 0.0f) {
                    // the release tail starts after frame t + ie - 1 (Q7): that block is simulated from its start
                    const int be = zl_block_of(t + ie - 1, N, invN) * N;
                    if (be <= t) { rollback_block(be); return; }
                    steps = be - t;
                    slowNext = true;
                } else {
                    steps = ie;
                    event = true;
                }
            }
        } else if (clockMode) {
            // the frame after whose rendering the beat-locked test (:232) fires
            const int tlimit = t + maxsteps;
            if (A.clocks_regular) {
                // regular clocks: "the test can fire in block j" (current_usecs_j + (N-1) * usecs_per_frame >= next loop
                // time) is monotone in j, so the first such block is found by bisection instead of a walk
                const ZlClock &c0 = clk0[kcur - kb];
                int fa = zl_clock_event_frame(c0, st.next_loop_usecs, n, N);      // the rest of the current block
                int kf = kcur;
                if (fa >= N) {
                    int lo = kcur + 1, hi = zl_block_of(tlimit - 1, N, invN) + 1; // candidate blocks [lo, hi)
                    const int hi0 = hi;
                    const uint64_t span = (uint64_t)(N - 1) * c0.usecs_per_frame;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (clk0[mid - kb].current_usecs + span >= st.next_loop_usecs) hi = mid; else lo = mid + 1;
                    }
                    if (lo < hi0) { kf = lo; fa = zl_clock_event_frame(clk0[lo - kb], st.next_loop_usecs, 0, N); }
                }
                if (fa < N && kf * N + fa < tlimit) { steps = kf * N + fa - t + 1; event = true; ke = kf; }
            } else {
                int kk = kcur, nn = n;
                for (int bs = tb; bs < tlimit; bs += N, ++kk, nn = 0) {                // block by block
                    const ZlClock &cj = clk0[kk - kb];
                    if (nn == 0 && cj.usecs_per_frame >= (1ull << 21)) { steps = bs - t; break; }   // simulated block (bs > t)
                    const int fa = zl_clock_event_frame(cj, st.next_loop_usecs, nn, N);
                    if (fa < N) {
                        if (bs + fa < tlimit) { steps = bs + fa - t + 1; event = true; ke = kk; }
                        break;
                    }
                }
            }
        }

        // ---- frames t .. t+steps-1 are rendered from the line P + i * s (steps <= L + 1) ----
        const int t1 = t + steps;
        const int k1 = zl_block_of(t1, N, invN);                    // block of the next frame
        {
            const int k0 = (n == 0) ? kcur : kcur + 1;              // whole blocks inside [t, t1)
            stats.active_frames += (unsigned long long)steps;
            const int started = zl_block_of(t1 - 1, N, invN) - kcur + (n == 0 ? 1 : 0);
            stats.source_bytes += blockBytes * (unsigned long long)started;
            if (k1 > k0 && es == 0.0f && st.adsr_state == ZL_ADSR_SUSTAIN) {
                const double Pk0 = fma((double)(k0 * N - t), s, st.P);     // exact: on the line
                if (haveCur && cur.k1 == k0 && cur.step == s && fma((double)((k0 - cur.k0) * N), s, cur.P) == Pk0) {
                    cur.k1 = k1;                                    // the same run continued (chunk boundary)
                } else if (nruns + (haveCur ? 1 : 0) < ZL_MAXRUNS) {
                    if (haveCur) A.runs[v].r[nruns++] = cur;
                    cur.P = Pk0; cur.step = s; cur.k0 = k0; cur.k1 = k1;
                    haveCur = true;
                }
            }
        }
        const double Pold = st.P;
        const uint64_t tickOld = st.next_loop_tick, usecsOld = st.next_loop_usecs;
        bool jump = false;
        if (event) {
            // ---- event after rendering frame t1 - 1 ----
            haveRun = false;
            if (st.looping) {
                zl_loop_restart(st, c, clk0[ke - kb], clockMode);
                lastRestartT = t1;
                if (posMode && !force_slow) {                       // (force_slow bit 1: test hook, plan every pass)
                    // sample-space loop: the pass that begins now repeats exactly (same integer start, same ratio)
                    if (perState == 0 && st.adsr_state == ZL_ADSR_SUSTAIN) { perState = 1; tcap = t1; jcap = nts; }
                    else if (perState == 1) {
                        perState = 2;
                        if (A.pass_cache && nts - jcap >= 1 && nts - jcap <= ZL_PASS_MAXSEG) {
                            // the pass just completed (restart to restart, all in sustain) is the voice's steady state
                            ZlPassCache &pc = A.pass_cache[v];
                            pc.valid = 0;
                            pc.ratio = st.pitch_ratio; pc.start_int = c.start_int; pc.stop_pos = c.stop_pos; pc.sustain = st.sustain;
                            pc.M = t1 - tcap; pc.n = nts - jcap;
                            for (int i = 0; i < pc.n; ++i) { ZlTSeg e = ts[jcap + i]; e.t -= tcap; e.flags = 0; pc.seg[i] = e; }
                            pc.valid = 1;
                        }
                        // passes of one or two segments are cheaper as inline runs (K2 needs no record for their blocks) --
                        // unless more passes are left than the run list could hold: then the window is finished here too
                        const bool many = (A.K * N - t1) / (t1 - tcap) >= ZL_MAXRUNS;
                        if (nts - jcap <= 64 && (nts - jcap >= 3 || many)) {
                            per_t0 = tcap; per_M = t1 - tcap; per_j0 = jcap; per_n = nts - jcap;
                            jump = true;
                        }
                    }
                }
            } else {
                zl_voice_hard_stop(st);                             // :249-252, voice ends after frame t1 - 1
                t_end = t1;
                dead_from = zl_block_of(t1 - 1, N, invN) + 1;
            }
        } else if (steps == S) {
            st.P = fma((double)L, s, st.P) + st.pitch_ratio;        // the real addition that leaves the binade
            haveRun = false;
        } else {
            st.P = fma((double)steps, s, st.P);                     // stopped at the chunk end or before a simulated block
            L -= steps;
            if (ie != ZL_INF_STEPS) ie -= steps;
        }
        // ---- the envelope over the same frames: st.env is the value of the last rendered frame ----
        const float eFirst = eNext;
        if (st.playing) {
            st.env = (float)fma((double)(steps - 1), (double)es, (double)eFirst);      // exact: inside the envelope run
            eLeft -= steps;
            if (eLeft > 0) eNext = (float)((double)st.env + (double)es); else haveERun = false;
            if (endAfter) {
                // the envelope ran out at this frame (:258-261): the voice stops after rendering it
                zl_voice_hard_stop(st);
                t_end = t1;
                dead_from = zl_block_of(t1 - 1, N, invN) + 1;
                haveRun = false; haveERun = false; endAfter = false;
                jump = false;
            }
        } else {
            haveERun = false; endAfter = false;
        }
        // ---- state at the start of the block that now contains t1 ----
        if (k1 * N > t) {
            const int b = k1 * N;
            if (b == t1) { Pbs = st.P; tick_bs = st.next_loop_tick; usecs_bs = st.next_loop_usecs; env_bs = st.env; }
            else {                                                  // the event (if any) comes after b
                Pbs = fma((double)(b - t), s, Pold); tick_bs = tickOld; usecs_bs = usecsOld;
                env_bs = (float)fma((double)(b - t - 1), (double)es, (double)eFirst);
            }
            adsr_bs = st.adsr_state;
            jbs = nts;
        }
        t = t1;
        if (jump) {
            // ---- the rest of the window is whole repetitions of the captured pass: go straight to its end ----
            const int TW = A.K * N;
            const int Rm = TW - t1;                                 // frames left; frame t1 is offset 0 of a pass
            const int q = Rm / per_M, rem = Rm - q * per_M;
            int j = per_j0 + per_n - 1;                             // the pass segment that covers offset rem
            while (j > per_j0 && ts[j].t - per_t0 > rem) --j;
            st.P = fma((double)(rem - (ts[j].t - per_t0)), ts[j].step, ts[j].P);   // exact: on the segment's line
            st.env = st.sustain;
            st.loop_phase1 = rem + 1; phaseSet = true;
            haveRun = false; haveERun = false;
            stats.active_frames += (unsigned long long)Rm;
            stats.source_bytes += blockBytes * (unsigned long long)(A.K - (k1 + (t1 > k1 * N ? 1 : 0)));
            t = TW;
        }
    }

    ZL_HD void end(const ZlBatch &A)
    {
        ZlReport rep;
        rep.playing = st.playing; rep.valid = 0; rep.peak_bits = 0; rep.progress = 0.0f; rep.clip = st.clip; rep.pad = 0; rep.P = st.P;
        // :265-267 -- the report of the last block exists only if the voice still has its clip
        if (valid && st.playing && t >= A.K * A.N) {
            rep.valid = 1;
            rep.progress = (float)(st.P / st.src_len);
        }
        if (!phaseSet) {
            const long long TW = (long long)A.K * A.N;
            if (phaseLost || !(valid && st.playing && t >= TW)) st.loop_phase1 = 0;
            else if (lastRestartT >= 0) st.loop_phase1 = (int)(TW - lastRestartT) + 1;
            else if (st.loop_phase1 > 0) st.loop_phase1 = (st.loop_phase1 - 1 + TW < 0x7fffffffLL) ? (int)(st.loop_phase1 + TW) : 0;
        }
        A.reports[v] = rep;
        A.voices[v] = st;
        if (haveCur) A.runs[v].r[nruns++] = cur;
        A.runs[v].n = nruns;
        A.runs[v].dead_from = dead_from;                           // blocks >= dead_from are idle (voice ended or never played)
        A.runs[v].nts = nts;
        A.runs[v].t_end = t_end;
        A.runs[v].per_t0 = per_t0; A.runs[v].per_M = per_M; A.runs[v].per_j0 = per_j0; A.runs[v].per_n = per_n;
    }
};

// The plan of block k of voice v: implied by a run, explicit, or idle.  Used by K2's staging.
ZL_HD inline ZlBlockPlan zl_plan_lookup(const ZlBatch &A, int k, int v, float run_env)
{
    ZlBlockPlan pl;
    zl_plan_clear(pl);
    const ZlRunList rl_ = A.runs[v];                              // one struct copy: independent wide loads, one round trip
    const ZlRunList *rl = &rl_;
    if (k >= rl->dead_from) return pl;                            // idle
    const int n = rl->n;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < ZL_MAXRUNS; ++j) {
        const ZlRun r = rl->r[j];
        if (j < n && k >= r.k0 && k < r.k1) {
            pl.flags = ZL_PLAN_ACTIVE; pl.n_active = A.N; pl.nseg = 1; pl.env = run_env;
            pl.P0 = fma((double)((k - r.k0) * A.N), r.step, r.P);  // exact: inside the linear run
            pl.step = r.step;
            return pl;
        }
    }
    return zl_plan_load(A, (size_t)k * A.V + v);
}

// A voice's segment stream as K1c reads it: the stored entries, then -- for a periodic loop -- the captured pass
// repeated every per_M frames.  A position in the stream is (idx, base): entry ts[idx] shifted by base frames.
struct ZlSegStream {
    const ZlTSeg *ts;
    int nts, per_t0, per_M, per_j0, per_n;

    ZL_HD void init(const ZlBatch &A, int v, const ZlRunList &rl)
    {
        ts = A.tsegs + (size_t)v * ZL_MAXTSEG;
        nts = rl.nts; per_t0 = rl.per_t0; per_M = rl.per_M; per_j0 = rl.per_j0; per_n = rl.per_n;
    }
    // the following entry; false at the end of the stream
    ZL_HD bool next(int &idx, int &base) const
    {
        if (per_n > 0 && idx + 1 == per_j0 + per_n) { idx = per_j0; base += per_M; return true; }
        if (idx + 1 >= nts) return false;
        ++idx;
        return true;
    }
    // the last entry that starts at or before frame T (false: none)
    ZL_HD bool locate(int T, int &idx, int &base) const
    {
        int lo, hi, off = T;
        base = 0;
        if (per_n > 0 && T >= per_t0 + per_M) {                   // in a repetition of the pass
            const int q = (T - per_t0) / per_M;
            base = q * per_M;
            off = T - base;
            lo = per_j0; hi = per_j0 + per_n;
        } else {
            lo = 0; hi = nts;
        }
        const int lo0 = lo;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (ts[mid].t <= off) lo = mid + 1; else hi = mid; }
        idx = lo - 1;
        return idx >= lo0;
    }
};

// K1c body: the explicit plan record of one block of voice v from its segment stream.  Blocks inside an inline run,
// simulated blocks (K1 wrote their records) and idle blocks are left alone.  begin() positions the walker on block
// kbeg; block() must then be called for k = kbeg, kbeg + 1, ...
struct ZlAssembler {
    ZlRunList rl;
    ZlSegStream ss;
    int v, idx, base, kend;
    float env;

    ZL_HD void begin(const ZlBatch &A, int voice, int kbeg, int kend_)
    {
        v = voice;
        rl = A.runs[v];
        ss.init(A, v, rl);
        kend = kend_ < rl.dead_from ? kend_ : rl.dead_from;
        env = A.vconst[v].env;
        idx = -1; base = 0;
        if (kbeg >= kend || rl.nts <= 0) { kend = kbeg; return; }
        // the whole chunk inside one inline run (the steady state of unpitched playback): nothing to assemble
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int q = 0; q < ZL_MAXRUNS; ++q)
            if (q < rl.n && kbeg >= rl.r[q].k0 && kend <= rl.r[q].k1) { kend = kbeg; return; }
        if (!ss.locate(kbeg * A.N, idx, base)) kend = kbeg;       // cannot happen: the stream starts at t = 0
    }

    // Returns the number of position segments of block k (0: nothing to write).  A block with more than two is marked
    // ZL_PLAN_SLOW and the caller fills its per-frame control with zl_expand_frame(idx0, base0, ...) for frames < n_active.
    ZL_HD int block(const ZlBatch &A, int k, int &idx0, int &base0, int &n_active)
    {
        if (k >= kend) return 0;
        const int N = A.N;
        const int T = k * N;
        for (;;) {                                                // walk to the last entry that starts at or before T
            int i2 = idx, b2 = base;
            if (!ss.next(i2, b2) || ss.ts[i2].t + b2 > T) break;
            idx = i2; base = b2;
        }
        const ZlTSeg a = ss.ts[idx];
        if (a.flags & ZL_TSEG_SLOW) return 0;
        bool inrun = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int q = 0; q < ZL_MAXRUNS; ++q) inrun = inrun || (q < rl.n && k >= rl.r[q].k0 && k < rl.r[q].k1);
        if (inrun) return 0;
        n_active = (rl.t_end - T < N) ? rl.t_end - T : N;
        // the entries that start inside the block (up to two are looked at)
        int nseg = 1, i1 = idx, b1 = base;
        ZlTSeg b; b.P = 0.0; b.step = 0.0; b.t = 0; b.flags = 0; b.E = 0.0f; b.estep = 0.0f;
        int bt = 0;
        {
            int i2 = idx, b2 = base;
            while (nseg < 3 && ss.next(i2, b2) && ss.ts[i2].t + b2 < T + n_active) {
                if (nseg == 1) { i1 = i2; b1 = b2; }
                ++nseg;
            }
            if (nseg >= 2) { b = ss.ts[i1]; bt = b.t + b1; }
        }
        const size_t pidx = (size_t)k * A.V + v;
        const float E0 = (float)fma((double)(T - (a.t + base)), (double)a.estep, (double)a.E);    // exact: inside the envelope run
        const bool ramps = a.estep != 0.0f || (nseg == 2 && (b.estep != 0.0f || b.E != E0));
        ZlPlanHdr h; h.flags = ZL_PLAN_ACTIVE | (nseg > 2 ? ZL_PLAN_SLOW : 0) | (ramps && nseg <= 2 ? ZL_PLAN_ENV : 0);
        h.n_active = n_active; h.nseg = nseg; h.env = E0;
        ZlPlanSeg0 s0; s0.P0 = fma((double)(T - (a.t + base)), a.step, a.P); s0.step = a.step;   // exact: on the segment's line
        A.plan_hdr[pidx] = h;
        A.plan_seg0[pidx] = s0;
        if (nseg == 2 || (ramps && nseg == 1)) {
            ZlPlanSeg1 s1; s1.P1 = b.P; s1.step1 = b.step; s1.n1 = nseg == 2 ? bt - T : INT_MAX;
            s1.estep0 = a.estep; s1.E1 = b.E; s1.estep1 = b.estep;
            A.plan_seg1[pidx] = s1;
        }
        idx0 = idx; base0 = base;
        return nseg;
    }
};

// Per-frame control of a block with more than two segments: position and envelope of frame f (< n_active) of block k;
// (idx0, base0) = the stream position that covers the block's first frame.
ZL_HD inline double zl_expand_frame(const ZlSegStream &ss, int N, int k, int idx0, int base0, int f, float &env)
{
    const int T = k * N + f;
    int idx = idx0, base = base0;
    for (;;) {
        int i2 = idx, b2 = base;
        if (!ss.next(i2, b2) || ss.ts[i2].t + b2 > T) break;
        idx = i2; base = b2;
    }
    const ZlTSeg a = ss.ts[idx];
    env = (float)fma((double)(T - (a.t + base)), (double)a.estep, (double)a.E);                    // exact
    return fma((double)(T - (a.t + base)), a.step, a.P);          // exact
}

// Position and envelope of frame f (< n_active) of a SLOW block of voice v, block k of the window: from the block's slot of the
// control pool or, when the pool was exhausted, recomputed -- a multi-segment block straight from the segment stream, a
// simulated block by running the reference's per-frame recurrence again from the state at the block's start (O(f) steps per
// lane: the price of an exhausted pool, never of a normal run).
ZL_HD inline void zl_slow_control(const ZlBatch &A, const ZlBlockPlan &pl, int v, int k, int f, double &P, float &env)
{
    if (!(pl.flags & (ZL_PLAN_NOSLOT_SIM | ZL_PLAN_NOSLOT_EXPAND))) {
        const size_t o = (size_t)(int)pl.step * (size_t)A.N + (size_t)f;
        P = A.ctl_P[o]; env = A.ctl_env[o];
        return;
    }
    if (pl.flags & ZL_PLAN_NOSLOT_EXPAND) {
        ZlSegStream ss;
        ss.init(A, v, A.runs[v]);
        P = zl_expand_frame(ss, A.N, k, pl.n1, (int)(uint32_t)zl_bits(pl.P1), f, env);
        return;
    }
    ZlVoiceState st; ZlVoiceBatchConst c;
    zl_snapshot_load(pl, A.sim_const[v], st, c);
    const ZlClock ck = A.inline_clock ? A.clock0 : A.clocks[k];
    const double upf = (double)ck.usecs_per_frame;
    for (int frame = 0;; ++frame) {
        P = st.P; env = zl_adsr_next(st);
        if (frame >= f || !zl_sim_step(st, c, ck, upf, frame)) return;
    }
}

// A multi-segment block (K1c): takes a slot for its per-frame control or, when the pool is exhausted, marks the block for
// recomputation in K2.  Returns the slot (>= 0) the caller fills with zl_expand_frame, or -1.
ZL_HD inline int zl_expand_slot(const ZlBatch &A, size_t pidx, int idx0, int base0)
{
    const int slot = zl_ctl_alloc(A);
    if (slot >= 0) { A.plan_seg0[pidx].step = (double)slot; return slot; }
    A.plan_hdr[pidx].flags |= ZL_PLAN_NOSLOT_EXPAND;
    ZlPlanSeg1 s1; s1.P1 = zl_from_bits((uint64_t)(uint32_t)base0); s1.step1 = 0.0; s1.n1 = idx0; s1.estep0 = 0.0f; s1.E1 = 0.0f; s1.estep1 = 0.0f;
    A.plan_seg1[pidx] = s1;
    return -1;
}

// Whole window of one voice with the clocks read from A.clocks (host harness; the kernel stages them in LDS).
ZL_HD inline void zl_plan_voice(const ZlBatch &A, int v, int force_slow, ZlPlanStats &stats)
{
    ZlPlanner pl;
    pl.begin(A, v, force_slow);
    while (pl.t < A.K * A.N) pl.iterate(A, A.K, A.clocks, 0, force_slow);
    pl.end(A);
    stats = pl.stats;
}

// ---- voice operations (device half of SamplerChannel::handleCommand, SamplerSynth.cpp:187-230) --
ZL_HD inline void zl_apply_op(ZlVoiceState &st, const ZlVoiceOp &op)
{
    if (op.kind == ZL_OP_START) {
        st = op.start;                                            // startNote :110-144 computed by the host
    } else if (op.kind == ZL_OP_NOTE_OFF) {
        if (st.playing) zl_adsr_note_off(st);                     // stopNote(0, true) :148-151
    } else if (op.kind == ZL_OP_HARD_STOP) {
        if (st.playing) zl_voice_hard_stop(st);                   // stopNote(0, false) :153-168
    } else if (op.kind == ZL_OP_FREEZE) {                         // SamplerSynth::setChannelEnabled(channel, false), SamplerSynth.cpp:343-351
        if (st.playing) st.playing = 2;
    } else if (op.kind == ZL_OP_THAW) {
        if (st.playing) st.playing = 1;
    } else if (op.kind == ZL_OP_PATCH) {                          // setCurrentCommand merge :58-98
        if (!st.playing) return;
        if (op.patch_mask & ZL_PATCH_LOOPING)  st.looping = op.looping;
        if (op.patch_mask & ZL_PATCH_GAIN)     { st.lgain = op.gain; st.rgain = op.gain; }
        if (op.patch_mask & ZL_PATCH_SLICE)    st.slice = op.slice;
        if (op.patch_mask & ZL_PATCH_POSITION) st.P = op.position;
    }
}
