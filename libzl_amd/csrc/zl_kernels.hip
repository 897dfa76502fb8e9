// zl_kernels.hip -- HIP kernels of the sampler engine, written for gfx950 (CDNA4, wave64).
//
//   K0 zl_k0_apply_ops   device half of SamplerChannel::handleCommand (SamplerSynth.cpp:187-230)
//   K1 zl_k1_plan        per-voice control plan of SamplerSynthVoice::process (:174-270), zl_plan.h
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
#include "zl_types.h"
#include "zl_plan.h"
#include "zl_render.h"
#include "zl_kernels.h"

// ------------------------------------------------------------------------------------------------
__global__ void zl_k0_apply_ops(const ZlBatch A)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n_op_ranges) return;
    const ZlOpRange rg = A.op_ranges[i];
    ZlVoiceState st = A.voices[rg.voice];
    for (int j = 0; j < rg.count; ++j) zl_apply_op(st, A.ops[rg.first + j]);
    A.voices[rg.voice] = st;
}

// ------------------------------------------------------------------------------------------------
// K1: one lane per voice.  Steady-state cost is O(1) per block (one linear segment); see zl_plan.h.
__global__ void __launch_bounds__(64) zl_k1_plan(const ZlBatch A, int force_slow)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= A.V) return;
    ZlPlanStats s;
    zl_plan_voice(A, v, force_slow, s);
    if (A.stats) {
        if (s.source_bytes)  atomicAdd(&A.stats->source_bytes, s.source_bytes);
        if (s.slow_blocks)   atomicAdd(&A.stats->slow_blocks, s.slow_blocks);
        if (s.active_frames) atomicAdd(&A.stats->active_frames, s.active_frames);
    }
}

// ------------------------------------------------------------------------------------------------
// K2: one workgroup = one (bus, mix group, block); one lane = one output frame.  The voices of the
// group are walked sequentially in voice order, so the per-frame sum has the reference's order
// (SamplerSynth.cpp:136-140) and lives in two registers.  Every per-voice record is wave-uniform
// (scalar loads); the only vector memory traffic is the 16-byte two-tap stereo gather and the
// final coalesced store.
static __device__ __forceinline__ float zl_wave_max(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

template <uint32_t MODE>
__global__ void __launch_bounds__(256) zl_k2_render(const ZlBatch A)
{
    const int N = A.N, V = A.V;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;          // frame inside the block
    const int k = blockIdx.y;
    const int bus = blockIdx.z / A.groups;
    const int g   = blockIdx.z - bus * A.groups;
    const int v0 = bus * A.VPB + g * A.G;
    const int vend = (bus + 1) * A.VPB;
    const int v1 = (v0 + A.G < vend) ? v0 + A.G : vend;
    const bool wantPeak = (k == A.K - 1);

    float accL = 0.0f, accR = 0.0f;
    for (int v = v0; v < v1; ++v) {
        const size_t pidx = (size_t)k * V + v;
        const ZlBlockPlan pl = A.plans[pidx];
        if (!(pl.flags & ZL_PLAN_ACTIVE)) continue;               // !voice->isPlaying, SamplerSynth.cpp:137
        const ZlVoiceConst vc = A.vconst[v];
        const bool act = f < pl.n_active;
        double P; float env;
        zl_eval_control(pl, A.segs + pidx * (ZL_MAXSEG - 1), A.ctl_P + pidx * (size_t)N, A.ctl_env + pidx * (size_t)N,
                        act ? f : 0, P, env);
        float l, r; int pos;
        zl_render_frame<MODE>(vc, A.arena + vc.src_offset, P, env, l, r, pos);
        if (act) { accL += l; accR += r; }                        // :218-221 (index shift applied at the store)
        if (A.trace) A.pos_trace[pidx * (size_t)N + f] = act ? pos : -1;
        if (wantPeak) {                                           // :213-216, signed peak from 0 (Q6)
            const float ng = l + r;
            float pk = (act && ng > 0.0f) ? ng : 0.0f;
            pk = zl_wave_max(pk);
            if ((threadIdx.x & 63) == 0 && pk > 0.0f) atomicMax(&A.reports[v].peak_bits, __float_as_uint(pk));
        }
    }

    float *outL, *outR;
    if (A.groups == 1) {
        const size_t KN = (size_t)A.K * N;
        outL = A.bus + ((size_t)bus * 2) * KN + (size_t)k * N;
        outR = outL + KN;
    } else {
        outL = A.partials + ((((size_t)k * A.B + bus) * A.groups + g) * 2) * (size_t)N;
        outR = outL + N;
    }
    if (MODE & ZL_MODE_FIX_DELAY) {
        outL[f] = accL; outR[f] = accR;
    } else {
        // quirk Q2: the reference pre-increments its output pointers, so frame f lands in out[f+1],
        // out[0] stays 0 and the sample of the last frame falls outside the buffer (dropped)
        if (f + 1 < N) { outL[f + 1] = accL; outR[f + 1] = accR; }
        if (f == 0)    { outL[0] = 0.0f;    outR[0] = 0.0f; }
    }
}

// ------------------------------------------------------------------------------------------------
// K3: one workgroup per (block, bus).  Sums the mix-group partials in group order (when there are
// any), writes the bus, and scans it for the AudioLevels integer peak and the RMS extension.
static __device__ __forceinline__ int zl_sample_to_peak_int(float x)
{
    const float v = fabsf(131072.0f * x);                          // AudioLevels.cpp:356,367
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return 0x7fffffff;
    return (int)v;
}

__global__ void __launch_bounds__(256) zl_k3_finalize(const ZlBatch A, const float *bus_in)
{
    const int k = blockIdx.x, bus = blockIdx.y, N = A.N;
    const size_t KN = (size_t)A.K * N;
    float *outL = A.bus ? A.bus + ((size_t)bus * 2) * KN + (size_t)k * N : nullptr;
    float *outR = outL ? outL + KN : nullptr;
    const float *inL = bus_in ? bus_in + ((size_t)bus * 2) * KN + (size_t)k * N : outL;
    const float *inR = inL + KN;

    int pkL = 0, pkR = 0; float sqL = 0.0f, sqR = 0.0f;
    for (int f = threadIdx.x; f < N; f += blockDim.x) {
        float l, r;
        if (A.groups > 1 && !bus_in) {
            l = 0.0f; r = 0.0f;
            const float *p = A.partials + (((size_t)k * A.B + bus) * A.groups) * 2 * (size_t)N;
            for (int g = 0; g < A.groups; ++g) { l += p[f]; r += p[N + f]; p += 2 * (size_t)N; }
            outL[f] = l; outR[f] = r;
        } else {
            l = inL[f]; r = inR[f];
        }
        const int a = zl_sample_to_peak_int(l), b = zl_sample_to_peak_int(r);
        pkL = a > pkL ? a : pkL; pkR = b > pkR ? b : pkR;
        sqL += l * l; sqR += r * r;
    }
    __shared__ int   s_pk[2][4];
    __shared__ float s_sq[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int a = __shfl_xor(pkL, o, 64), b = __shfl_xor(pkR, o, 64);
        pkL = a > pkL ? a : pkL; pkR = b > pkR ? b : pkR;
        sqL += __shfl_xor(sqL, o, 64); sqR += __shfl_xor(sqR, o, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_pk[0][w] = pkL; s_pk[1][w] = pkR; s_sq[0][w] = sqL; s_sq[1][w] = sqR; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        ZlBlockLevels lv; lv.peak_l = 0; lv.peak_r = 0; lv.sumsq_l = 0.0f; lv.sumsq_r = 0.0f;
        for (int i = 0; i < nw; ++i) {
            lv.peak_l = s_pk[0][i] > lv.peak_l ? s_pk[0][i] : lv.peak_l;
            lv.peak_r = s_pk[1][i] > lv.peak_r ? s_pk[1][i] : lv.peak_r;
            lv.sumsq_l += s_sq[0][i]; lv.sumsq_r += s_sq[1][i];
        }
        A.levels[(size_t)k * A.B + bus] = lv;
    }
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

// report finalisation: gain = peakGain * 0.5f (SamplerSynthVoice.cpp:266)
__global__ void zl_k_reports(const ZlReport *reports, int V, float *gain_out)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    gain_out[v] = __uint_as_float(reports[v].peak_bits) * 0.5f;
}

// ------------------------------------------------------------------------------------------------
// JackPassthrough: in [B][2][frames] -> out [B][6][frames]
struct ZlPassParams { float dry, fx1, fx2, pan; int muted; };

__global__ void __launch_bounds__(256) zl_k_passthrough(const ZlPassParams *params, const float *in, float *out, long long frames)
{
    const int bus = blockIdx.y;
    const ZlPassParams p = params[bus];
    const float *inL = in + (size_t)bus * 2 * frames, *inR = inL + frames;
    float *o = out + (size_t)bus * 6 * frames;
    const float amounts[3] = { p.dry, p.fx1, p.fx2 };
    const float lm = fminf(1 - p.pan, 1.0f), rm = fminf(1 + p.pan, 1.0f);   // JackPassthrough.cpp:100-101
    for (long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x; f < frames; f += (long long)gridDim.x * blockDim.x) {
        const float sl = inL[f], sr = inR[f];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float ol, orr;
            if (p.muted)                               { ol = 0.0f; orr = 0.0f; }             // :55-61
            else if (p.pan == 0 && amounts[c] == 0)    { ol = 0.0f; orr = 0.0f; }             // memset fast path
            else if (p.pan == 0 && amounts[c] == 1)    { ol = sl;   orr = sr; }               // memcpy fast path
            else { ol = amounts[c] * sl * lm; orr = amounts[c] * sr * rm; }                   // :100-109
            o[(size_t)(2 * c) * frames + f] = ol;
            o[(size_t)(2 * c + 1) * frames + f] = orr;
        }
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
    if (A.n_op_ranges <= 0) return 0;
    hipLaunchKernelGGL(zl_k0_apply_ops, dim3((A.n_op_ranges + 63) / 64), dim3(64), 0, s, A);
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_plan(const ZlBatch &A, int force_slow, hipStream_t s)
{
    hipLaunchKernelGGL(zl_k1_plan, dim3((A.V + 63) / 64), dim3(64), 0, s, A, force_slow);
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_render(const ZlBatch &A, hipStream_t s)
{
    const int tpb = A.N < 256 ? A.N : 256;
    const dim3 grid(A.N / tpb, A.K, A.B * A.groups), block(tpb);
    switch (A.mode & 7u) {
#define ZL_CASE(M) case M: hipLaunchKernelGGL(zl_k2_render<M>, grid, block, 0, s, A); break;
        ZL_CASE(0) ZL_CASE(1) ZL_CASE(2) ZL_CASE(3) ZL_CASE(4) ZL_CASE(5) ZL_CASE(6) ZL_CASE(7)
#undef ZL_CASE
    }
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_finalize(const ZlBatch &A, const float *bus_in, hipStream_t s)
{
    hipLaunchKernelGGL(zl_k3_finalize, dim3(A.K, A.B), dim3(A.N < 256 ? A.N : 256), 0, s, A, bus_in);
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_reports(const ZlReport *reports, int V, float *gain_out, hipStream_t s)
{
    hipLaunchKernelGGL(zl_k_reports, dim3((V + 255) / 256), dim3(256), 0, s, reports, V, gain_out);
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_levels_tick(ZlLevelsState *state, const ZlBlockLevels *levels, int B, int N, int with_hold_bus, hipStream_t s)
{
    hipLaunchKernelGGL(zl_k_levels_tick, dim3((B + 63) / 64), dim3(64), 0, s, state, levels, B, N, with_hold_bus);
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_passthrough(const void *params_dev, const float *in, float *out, int B, long long frames, hipStream_t s)
{
    long long nb = (frames + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(zl_k_passthrough, dim3((unsigned)nb, B), dim3(256), 0, s, (const ZlPassParams *)params_dev, in, out, frames);
    ZL_LAUNCH_CHECK();
    return 0;
}

int zl_launch_interleave(const float *L, const float *R, float *dst, int length, int pad, hipStream_t s)
{
    hipLaunchKernelGGL(zl_k_interleave, dim3((length + pad + 255) / 256), dim3(256), 0, s, L, R, dst, length, pad);
    ZL_LAUNCH_CHECK();
    return 0;
}
