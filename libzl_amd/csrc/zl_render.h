// zl_render.h -- per-frame arithmetic of SamplerSynthVoice::process (reference
// lib/SamplerSynthVoice.cpp:198-216): fractional-index gather, linear (or Hermite) interpolation,
// gain / envelope / clip volume, M/S pan.  One call = one (voice, frame).  The operation order is
// the reference's expression order; the translation unit must be built with -ffp-contract=off so
// no multiply-add is fused (the oracle defines the same rounding sequence).
//
// __host__ __device__ so tests/cpu_harness can execute the identical code on the CPU.
#pragma once
#include "zl_types.h"
#include <math.h>

// Position and envelope of frame f of a planned FAST block (at most two inline segments; blocks with per-frame control: zl_plan.h,
// zl_slow_control).
ZL_HD inline void zl_eval_control(const ZlBlockPlan &pl, int f, double &P, float &env)
{
    const bool seg1 = f >= pl.n1;
    const int n0 = seg1 ? pl.n1 : 0;
    P = fma((double)(f - n0), seg1 ? pl.step1 : pl.step, seg1 ? pl.P1 : pl.P0);                     // exact (see zl_plan.h)
    env = (float)fma((double)(f - n0), (double)(seg1 ? pl.estep1 : pl.estep0), (double)(seg1 ? pl.E1 : pl.env));   // exact fp32 ramp
}

// 4-point Catmull-Rom (build-defined extension, absent in the reference) in TAP-WEIGHT form: the four cubic
// weights of the fractional position a,
//   w0 = ((-a/2 + 1) a - 1/2) a    w1 = (3/2 a - 5/2) a^2 + 1    w2 = ((-3/2 a + 2) a + 1/2) a    w3 = (a/2 - 1/2) a^2,
// are computed once per frame and shared by both channels; y = w0 y0 + w1 y1 + w2 y2 + w3 y3.  Fused multiply-adds in
// exactly this order (11 operations for the weights + 4 per channel; every fmaf is ONE rounding -- the oracle and the numpy
// restatement do the same).  The Horner form of round 1 cost 12 per channel: the kernel is bound by VALU issue in this mode.
struct ZlHermiteW { float w0, w1, w2, w3; };
ZL_HD inline ZlHermiteW zl_hermite_weights(float a)
{
    ZlHermiteW w;
    const float t = a * a;
    w.w0 = fmaf(fmaf(-0.5f, a, 1.0f), a, -0.5f) * a;
    w.w1 = fmaf(fmaf(1.5f, a, -2.5f), t, 1.0f);
    w.w2 = fmaf(fmaf(-1.5f, a, 2.0f), a, 0.5f) * a;
    w.w3 = fmaf(0.5f, a, -0.5f) * t;
    return w;
}
ZL_HD inline float zl_hermite4(float y0, float y1, float y2, float y3, const ZlHermiteW &w)
{
    return fmaf(w.w3, y3, fmaf(w.w2, y2, fmaf(w.w1, y1, w.w0 * y0)));
}

// The gathered samples of one frame: taps pos, pos+1 (and pos-1, pos+2 for Hermite) of both channels.
struct ZlTaps {
    float x0l, x0r, x1l, x1r;       // pos, pos+1
    float xml, xmr, x2l, x2r;       // pos-1, pos+2 (Hermite only)
};

// :198-199 -- integer position and fractional part
ZL_HD inline void zl_split_position(double P, int &pos, float &alpha)
{
    pos = (int)P;
    alpha = (float)(P - (double)pos);
}

// :204-211 on gathered taps.  inb = guard of :204 (Q5); wide = all four Hermite taps are inside the
// source; stereo = the source has a second channel.  Returns the panned (l', r') of :210-211.
template <uint32_t MODE>
ZL_HD inline void zl_mix_frame(const ZlTaps &t, float alpha, bool inb, bool wide, bool stereo,
                               float lgain, float rgain, float env, float vol, float lpan, float rpan,
                               float &lout, float &rout)
{
    const float invAlpha = 1.0f - alpha;                         // :200
    float l, r;
    if (MODE & ZL_MODE_HERMITE) {
        // build-defined extension: Catmull-Rom, linear at the source edges; whole-sample gain with the gain product formed first,
        // sample * ((gain * envelope) * volume) -- for a voice in sustain the product is one number per block
        const ZlHermiteW w = zl_hermite_weights(alpha);
        const float sl = wide ? zl_hermite4(t.xml, t.x0l, t.x1l, t.x2l, w) : (t.x0l * invAlpha + t.x1l * alpha);
        const float sr = wide ? zl_hermite4(t.xmr, t.x0r, t.x1r, t.x2r, w) : (t.x0r * invAlpha + t.x1r * alpha);
        l = sl * ((lgain * env) * vol);
        r = sr * ((rgain * env) * vol);
    } else if (MODE & ZL_MODE_FIX_GAIN) {
        l = (t.x0l * invAlpha + t.x1l * alpha) * lgain * env * vol;
        r = (t.x0r * invAlpha + t.x1r * alpha) * rgain * env * vol;
    } else {
        // :204-205 -- quirk Q1: the gain chain multiplies only the second tap
        l = t.x0l * invAlpha + t.x1l * alpha * lgain * env * vol;
        r = t.x0r * invAlpha + t.x1r * alpha * rgain * env * vol;
    }
    if (!inb) l = 0.0f;
    if (!(stereo && inb)) r = l;                                 // :205 mono source / out of range: r = l
    const float mSignal = 0.5f * (l + r);                        // :208 (0.5 * float in double, exact)
    const float sSignal = l - r;                                 // :209
    lout = lpan * mSignal + sSignal;                             // :210
    rout = rpan * mSignal - sSignal;                             // :211
}

// JackPassthrough fan-out of one bus frame (JackPassthrough.cpp:55-109): output pair c (0 dry, 1 wetFx1, 2 wetFx2) of
// input (sl, sr).  lm, rm = zl_pass_pan().  The memset / memcpy fast paths of the reference are value-identical to
// these selects (they differ from the multiply only in the sign of zero and for non-finite input, which is why they
// are kept).
ZL_HD inline void zl_pass_pan(const ZlPassParams &p, float &lm, float &rm)
{
    const float a = 1 - p.pan, b = 1 + p.pan;
    lm = (1.0f < a) ? 1.0f : a;                                  // std::min(1 - panAmount, 1.0f), :100
    rm = (1.0f < b) ? 1.0f : b;                                  // :101
}
ZL_HD inline void zl_pass_pair(const ZlPassParams &p, float amount, float lm, float rm, float sl, float sr, float &ol, float &orr)
{
    if (p.muted)                            { ol = 0.0f; orr = 0.0f; }                      // :55-61
    else if (p.pan == 0 && amount == 0)     { ol = 0.0f; orr = 0.0f; }                      // :66-69 memset
    else if (p.pan == 0 && amount == 1)     { ol = sl;   orr = sr; }                        // :70-73 memcpy
    else                                    { ol = amount * sl * lm; orr = amount * sr * rm; }   // :100-109
}

// Host-side convenience used by tests/cpu_harness: gather + mix for one (voice, frame).
// src points at the voice's source in the arena (interleaved stereo or mono).
template <uint32_t MODE>
ZL_HD inline void zl_render_frame(const ZlVoiceConst &vc, const float *src, double P, float env,
                                  float &lout, float &rout, int &pos_out)
{
    int pos; float alpha;
    zl_split_position(P, pos, alpha);
    const bool inb = vc.sample_duration > pos && pos >= 0;       // :204 guard; pos < 0 cannot occur for P >= 0
    const bool stereo = vc.channels > 1;
    const bool wide = (MODE & ZL_MODE_HERMITE) && inb && (pos - 1 >= 0) && (pos + 2 <= vc.sample_duration);
    const int p = inb ? pos : 0;
    ZlTaps t;
    t.xml = t.xmr = t.x2l = t.x2r = 0.0f;
    if (stereo) {
        t.x0l = src[2 * p]; t.x0r = src[2 * p + 1]; t.x1l = src[2 * p + 2]; t.x1r = src[2 * p + 3];
        if (wide) { t.xml = src[2 * p - 2]; t.xmr = src[2 * p - 1]; t.x2l = src[2 * p + 4]; t.x2r = src[2 * p + 5]; }
    } else {
        t.x0l = src[p]; t.x1l = src[p + 1]; t.x0r = 0.0f; t.x1r = 0.0f;
        if (wide) { t.xml = src[p - 1]; t.x2l = src[p + 2]; }
    }
    zl_mix_frame<MODE>(t, alpha, inb, wide, stereo, vc.lgain, vc.rgain, env, vc.clip_volume, vc.lpan, vc.rpan, lout, rout);
    pos_out = pos;
}

// One sample of a 16-bit WAV as the reference's recorder writes it (AudioLevels.cpp:53-58: a juce::WavAudioFormat writer, 16 bit,
// fed floats through AudioFormatWriter::ThreadedWriter): float -> 32-bit fixed point (clamped at +-1, times 0x7fffffff in double,
// round to nearest even) -> its upper 16 bits.  JUCE is not in the tree: restated from its public source (convertFloatsToInts +
// the Int32 -> Int16 little-endian write), version unpinned like the ADSR.  NaN, which JUCE leaves
// undefined, is written as 0.  The reader side (zl_libzl.cpp, libzl_wav_read) is the inverse convention.
ZL_HD inline int16_t zl_pcm16(float x)
{
    const double d = (double)x;
    int32_t q;
    if (d <= -1.0) q = INT32_MIN;
    else if (d >= 1.0) q = INT32_MAX;
    else if (d != d) q = 0;
    else q = (int32_t)rint(2147483647.0 * d);                   // |.| < 2^31: fits
    return (int16_t)(q >> 16);
}
