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

// Position and envelope of frame f of a planned block.
ZL_HD inline void zl_eval_control(const ZlBlockPlan &pl, const ZlSegment *extra, const double *ctlP,
                                  const float *ctlEnv, int f, double &P, float &env)
{
    if (pl.flags & ZL_PLAN_SLOW) {
        P = ctlP[f];
        env = ctlEnv[f];
        return;
    }
    env = pl.env;
    double P0 = pl.P0, step = pl.step;
    int n0 = 0;
    for (int i = 0; i + 1 < pl.nseg; ++i) {          // later segments override earlier ones
        if (f >= extra[i].n0) { P0 = extra[i].P0; step = extra[i].step; n0 = extra[i].n0; }
    }
    P = fma((double)(f - n0), step, P0);             // exact (see zl_plan.h)
}

ZL_HD inline float zl_hermite4(float y0, float y1, float y2, float y3, float a)
{
    const float c1 = 0.5f * (y2 - y0);
    const float c2 = (y0 + 2.0f * y2) - (0.5f * y3 + 2.5f * y1);
    const float c3 = (0.5f * y3 + 1.5f * y1) - (0.5f * y0 + 1.5f * y2);
    return y1 + a * (c1 + a * (c2 + a * c3));
}

// src points at the voice's source in the arena (interleaved stereo or mono).
// Returns the panned (l', r') of :210-211 and pos of :198.
template <uint32_t MODE>
ZL_HD inline void zl_render_frame(const ZlVoiceConst &vc, const float *src, double P, float env,
                                  float &lout, float &rout, int &pos_out)
{
    const int pos = (int)P;                                      // :198
    const float alpha = (float)(P - (double)pos);                // :199
    const float invAlpha = 1.0f - alpha;                         // :200
    const bool inb = vc.sample_duration > pos && pos >= 0;       // :204 guard (Q5); pos < 0 cannot occur for P >= 0
    const int p = inb ? pos : 0;
    const bool stereo = vc.channels > 1;
    float l, r;
    if (MODE & ZL_MODE_HERMITE) {
        const bool wide = inb && (pos - 1 >= 0) && (pos + 2 <= vc.sample_duration);
        float sl, sr;
        if (stereo) {
            const float x0l = src[2 * p], x0r = src[2 * p + 1], x1l = src[2 * p + 2], x1r = src[2 * p + 3];
            if (wide) {
                sl = zl_hermite4(src[2 * p - 2], x0l, x1l, src[2 * p + 4], alpha);
                sr = zl_hermite4(src[2 * p - 1], x0r, x1r, src[2 * p + 5], alpha);
            } else {
                sl = x0l * invAlpha + x1l * alpha;
                sr = x0r * invAlpha + x1r * alpha;
            }
        } else {
            const float x0 = src[p], x1 = src[p + 1];
            sl = wide ? zl_hermite4(src[p - 1], x0, x1, src[p + 2], alpha) : (x0 * invAlpha + x1 * alpha);
            sr = 0.0f;
        }
        l = sl * vc.lgain * env * vc.clip_volume;
        r = stereo ? (sr * vc.rgain * env * vc.clip_volume) : l;
        if (!inb) { l = 0.0f; r = 0.0f; }
    } else {
        float x0l, x0r, x1l, x1r;
        if (stereo) {
            // taps of both channels are 16 contiguous bytes in the interleaved arena
            x0l = src[2 * p]; x0r = src[2 * p + 1]; x1l = src[2 * p + 2]; x1r = src[2 * p + 3];
        } else {
            x0l = src[p]; x1l = src[p + 1]; x0r = 0.0f; x1r = 0.0f;
        }
        if (MODE & ZL_MODE_FIX_GAIN) {
            l = (x0l * invAlpha + x1l * alpha) * vc.lgain * env * vc.clip_volume;
            r = (x0r * invAlpha + x1r * alpha) * vc.rgain * env * vc.clip_volume;
        } else {
            // :204-205 -- quirk Q1: the gain chain multiplies only the second tap
            l = x0l * invAlpha + x1l * alpha * vc.lgain * env * vc.clip_volume;
            r = x0r * invAlpha + x1r * alpha * vc.rgain * env * vc.clip_volume;
        }
        if (!inb) l = 0.0f;
        if (!(stereo && inb)) r = l;                             // :205 mono / out-of-range: r = l
    }
    const float mSignal = 0.5f * (l + r);                        // :208 (0.5 * float in double, exact)
    const float sSignal = l - r;                                 // :209
    lout = vc.lpan * mSignal + sSignal;                          // :210
    rout = vc.rpan * mSignal - sSignal;                          // :211
    pos_out = pos;
}
