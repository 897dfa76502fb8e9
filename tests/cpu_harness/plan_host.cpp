// plan_host.cpp -- TEST HARNESS (not part of the product library).
//
// Compiles the engine's __host__ __device__ code (zl_plan.h: K1 control plan, zl_render.h: K2
// per-frame arithmetic, zl_host.h: command handling) with g++ and executes the kernels' work as
// plain loops, so the CPU-only test tier can check the planning logic and the summation order
// against the oracle bit for bit without a GPU.  libzl_amd never loads this library.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "zlhip.h"
#include "zl_host.h"
#include "zl_plan.h"
#include "zl_render.h"
#include "zl_types.h"

struct ZlSim {
    int B, VPB, V, G; uint32_t mode;
    ZlHostControl hc;
    std::vector<float> arena;
    std::vector<ZlSound> sounds;
    std::vector<ZlClip> clips;
    std::vector<ZlVoiceState> voices;
    std::vector<ZlPassCache> passCache;
    std::vector<ZlVoiceConst> vconst; std::vector<ZlRunList> runs; std::vector<ZlTSeg> tsegs;
    std::vector<ZlPlanHdr> planHdr; std::vector<ZlPlanSeg0> planSeg0; std::vector<ZlPlanSeg1> planSeg1;
    std::vector<double> ctlP; std::vector<float> ctlEnv; std::vector<ZlSimConst> simConst;
    unsigned long long ctlNext = 0, ctlBase = 0; int ctlSlots = -1;      // -1: a slot for every (block, voice); >= 0: the pool's size (tests of the exhausted pool)
    std::vector<ZlReport> reports;
    std::vector<int32_t> trace;
    std::vector<ZlBlockLevels> levels;
    std::vector<int32_t> nsegOut;
    ZlPlanStats stats{};
    int lastK = 0, lastN = 0; int expanded = 0;
    bool reportsFresh = false;
    void absorb() { if (reportsFresh) { hc.absorb_reports(reports.data()); reportsFresh = false; } }
};

template <uint32_t MODE>
static void render_all(ZlSim &S, const ZlBatch &A, float *bus)
{
    const int N = A.N, K = A.K, V = A.V;
    const size_t KN = (size_t)K * N;
    std::vector<float> part((size_t)A.groups * 2 * N);
    for (int k = 0; k < K; ++k)
        for (int b = 0; b < A.B; ++b) {
            for (int g = 0; g < A.groups; ++g) {
                const int v0 = b * A.VPB + g * A.G;
                const int vend = (b + 1) * A.VPB;
                const int v1 = v0 + A.G < vend ? v0 + A.G : vend;
                float *oL = &part[(size_t)g * 2 * N], *oR = oL + N;
                for (int f = 0; f < N; ++f) {
                    float accL = 0.0f, accR = 0.0f;
                    for (int v = v0; v < v1; ++v) {
                        const size_t pidx = (size_t)k * V + v;
                        const ZlBlockPlan pl = zl_plan_lookup(A, k, v, A.vconst[v].env);
                        if (!(pl.flags & ZL_PLAN_ACTIVE)) continue;
                        const ZlVoiceConst &vc = A.vconst[v];
                        const bool act = f < pl.n_active;
                        double P; float env;
                        if (pl.flags & ZL_PLAN_SLOW) zl_slow_control(A, pl, v, k, act ? f : 0, P, env);
                        else zl_eval_control(pl, act ? f : 0, P, env);
                        float l, r; int pos;
                        zl_render_frame<MODE>(vc, A.arena + vc.src_offset, P, env, l, r, pos);
                        if (act) { accL += l; accR += r; }
                        S.trace[pidx * (size_t)N + f] = act ? pos : -1;
                        if (k == K - 1) {
                            const float ng = l + r;
                            const float pk = (act && ng > 0.0f) ? ng : 0.0f;
                            const uint32_t bits = __builtin_bit_cast(uint32_t, pk);
                            if (bits > A.reports[v].peak_bits) A.reports[v].peak_bits = bits;
                        }
                    }
                    if (MODE & ZL_MODE_FIX_DELAY) { oL[f] = accL; oR[f] = accR; }
                    else { if (f + 1 < N) { oL[f + 1] = accL; oR[f + 1] = accR; } if (f == 0) { oL[0] = 0.0f; oR[0] = 0.0f; } }
                }
            }
            float *outL = bus + ((size_t)b * 2) * KN + (size_t)k * N, *outR = outL + KN;
            ZlBlockLevels lv{0, 0, 0.0f, 0.0f};
            for (int f = 0; f < N; ++f) {
                float l, r;
                if (A.groups > 1) { l = 0.0f; r = 0.0f; for (int g = 0; g < A.groups; ++g) { l += part[(size_t)g * 2 * N + f]; r += part[(size_t)g * 2 * N + N + f]; } }
                else { l = part[f]; r = part[N + f]; }
                outL[f] = l; outR[f] = r;
                auto pk = [](float x) { const float v = fabsf(131072.0f * x); if (!(v == v)) return 0; if (v >= 2147483648.0f) return 0x7fffffff; return (int)v; };
                const int a = pk(l), c = pk(r);
                lv.peak_l = a > lv.peak_l ? a : lv.peak_l; lv.peak_r = c > lv.peak_r ? c : lv.peak_r;
            }
            S.levels[(size_t)k * A.B + b] = lv;
        }
}

extern "C" {

ZlSim *zlsim_create(int B, int VPB, int max_sounds, double fs, uint32_t mode, int G)
{
    ZlSim *S = new ZlSim();
    S->B = B; S->VPB = VPB; S->V = B * VPB; S->G = G > 0 ? (G < VPB ? G : VPB) : VPB; S->mode = mode;
    S->hc.init(B, VPB, max_sounds, fs);
    S->sounds.assign((size_t)max_sounds, ZlSound{0, 0, 0, 0.0});
    S->clips.assign((size_t)max_sounds, ZlClip{});
    S->voices.assign((size_t)S->V, ZlVoiceState{});
    S->passCache.assign((size_t)S->V, ZlPassCache{});
    S->vconst.assign((size_t)S->V, ZlVoiceConst{}); S->runs.assign((size_t)S->V, ZlRunList{}); S->tsegs.assign((size_t)S->V * ZL_MAXTSEG, ZlTSeg{});
    S->reports.assign((size_t)S->V, ZlReport{});
    return S;
}

void zlsim_destroy(ZlSim *S) { delete S; }
void zlsim_set_ctl_slots(ZlSim *S, int slots) { S->ctlSlots = slots; }

int zlsim_clip_set(ZlSim *S, int id, const zlhip_clip_params *p)
{
    S->hc.clipParams[(size_t)id] = *p;
    ZlHostControl::fill_clip(S->clips[(size_t)id], *p);
    return 0;
}

int zlsim_sound_upload(ZlSim *S, const float *L, const float *R, int length, double sr)
{
    int id = -1;
    for (size_t i = 0; i < S->hc.soundUsed.size(); ++i) if (!S->hc.soundUsed[i]) { id = (int)i; break; }
    if (id < 0) return -1;
    const int ch = R ? 2 : 1;
    size_t floats = ((size_t)length + 8) * ch;
    floats = (floats + 3) & ~(size_t)3;
    ZlSound s; s.offset = S->arena.size(); s.length = length; s.channels = ch; s.sample_rate = sr;
    S->arena.resize(S->arena.size() + floats, 0.0f);
    float *dst = S->arena.data() + s.offset;
    if (R) for (int i = 0; i < length; ++i) { dst[2 * i] = L[i]; dst[2 * i + 1] = R[i]; }
    else std::memcpy(dst, L, (size_t)length * sizeof(float));
    S->sounds[(size_t)id] = s; S->hc.sounds[(size_t)id] = s; S->hc.soundUsed[(size_t)id] = 1;
    zlhip_clip_params p;
    ZlHostControl::default_clip_params(&p, (float)(length / sr));
    zlsim_clip_set(S, id, &p);
    return id;
}

int zlsim_handle_command(ZlSim *S, const zlhip_clip_command *c, uint64_t tick)
{
    S->absorb();
    return S->hc.handle_command(*c, tick);
}

int zlsim_start_voice(ZlSim *S, int bus, int slot, const zlhip_clip_command *c, uint64_t tick)
{
    S->absorb();
    return S->hc.handle_on_bus(bus, *c, tick, slot);
}

int zlsim_set_bus_enabled(ZlSim *S, int bus, int enabled)
{
    S->absorb();
    return S->hc.set_bus_enabled(bus, enabled != 0);
}

int zlsim_update_voice(ZlSim *S, int bus, int slot, const zlhip_clip_command *c)
{
    S->absorb();
    return S->hc.update_voice(bus, slot, *c);
}

int zlsim_stop_voice(ZlSim *S, int bus, int slot, int allow_tail_off)
{
    S->absorb();
    return S->hc.stop_voice(bus, slot, allow_tail_off != 0);
}

// bus: [B][2][K*N]
int zlsim_render_batch(ZlSim *S, int K, int N, const zlhip_clock *clocks, float *bus, int force_slow)
{
    const size_t V = (size_t)S->V;
    std::vector<ZlClock> ck((size_t)K);
    bool regular = true;
    for (int k = 0; k < K; ++k) {
        ZlHostControl::fill_clock(ck[(size_t)k], clocks[k], N);
        if (ck[(size_t)k].usecs_per_frame >= (1ull << 21) || ck[(size_t)k].usecs_per_frame != ck[0].usecs_per_frame
            || (k > 0 && ck[(size_t)k].current_usecs < ck[(size_t)k - 1].current_usecs)) regular = false;
    }
    S->planHdr.assign((size_t)K * V, ZlPlanHdr{}); S->planSeg0.assign((size_t)K * V, ZlPlanSeg0{}); S->planSeg1.assign((size_t)K * V, ZlPlanSeg1{});
    const size_t slots = S->ctlSlots >= 0 ? (size_t)S->ctlSlots : (size_t)K * V;
    S->ctlP.assign(slots * N + 1, 0.0); S->ctlEnv.assign(slots * N + 1, 0.0f); S->simConst.assign(V, ZlSimConst{});
    S->ctlBase = S->ctlNext + (unsigned long long)K * V + 1;            // past every value the counter can have reached (as the engine does)
    S->trace.assign((size_t)K * V * N, -1);
    S->levels.assign((size_t)K * S->B, ZlBlockLevels{});
    std::vector<ZlVoiceOp> ops; std::vector<ZlOpRange> ranges;
    S->absorb();
    S->hc.drain_ops(ops, ranges);

    ZlBatch A; std::memset(&A, 0, sizeof A);
    A.V = S->V; A.B = S->B; A.VPB = S->VPB; A.K = K; A.N = N; A.k0 = 0; A.Ktot = K; A.G = S->G; A.groups = (S->VPB + S->G - 1) / S->G; A.mode = S->mode; A.clocks_regular = regular ? 1 : 0;
    A.clocks = ck.data(); A.sounds = S->sounds.data(); A.clips = S->clips.data(); A.arena = S->arena.data();
    A.voices = S->voices.data(); A.pass_cache = S->passCache.data(); A.vconst = S->vconst.data(); A.runs = S->runs.data(); A.tsegs = S->tsegs.data(); A.plan_hdr = S->planHdr.data(); A.plan_seg0 = S->planSeg0.data(); A.plan_seg1 = S->planSeg1.data();
    A.ctl_P = S->ctlP.data(); A.ctl_env = S->ctlEnv.data(); A.reports = S->reports.data();
    A.ctl_next = &S->ctlNext; A.ctl_base = S->ctlBase; A.ctl_slots = (int)slots; A.sim_const = S->simConst.data();

    for (const ZlOpRange &rg : ranges) {                          // K0
        ZlVoiceState st = S->voices[(size_t)rg.voice];
        for (int j = 0; j < rg.count; ++j) zl_apply_op(st, ops[(size_t)(rg.first + j)]);
        S->voices[(size_t)rg.voice] = st;
    }
    S->stats = ZlPlanStats{0, 0, 0};
    for (int v = 0; v < S->V; ++v) {                              // K1
        ZlPlanStats st;
        zl_plan_voice(A, v, force_slow, st);
        S->stats.source_bytes += st.source_bytes; S->stats.slow_blocks += st.slow_blocks; S->stats.active_frames += st.active_frames;
    }
    S->expanded = 0;
    // K1c, as the kernel cuts it: a lane assembles 2 (8 from 512 voices on) consecutive blocks and finds its place in the voice's
    // segment stream by itself (ZlAssembler::begin -> locate, also inside the repetitions of a periodic pass)
    const int bpl = S->V >= 512 ? 8 : 2;
    for (int v = 0; v < S->V; ++v)
    for (int kbeg = 0; kbeg < K; kbeg += bpl) {
        ZlAssembler as;
        const int kend = kbeg + bpl < K ? kbeg + bpl : K;
        as.begin(A, v, kbeg, kend);
        for (int k = kbeg; k < kend; ++k) {
            int idx0 = 0, base0 = 0, n_active = 0;
            if (as.block(A, k, idx0, base0, n_active) <= 2) continue;
            ++S->expanded;
            const int slot = zl_expand_slot(A, (size_t)k * V + (size_t)v, idx0, base0);
            if (slot < 0) continue;                               // pool exhausted: the block was marked, rendering recomputes it
            for (int f = 0; f < N; ++f) {
                float env;
                S->ctlP[(size_t)slot * (size_t)N + f] = zl_expand_frame(as.ss, N, k, idx0, base0, f < n_active ? f : 0, env);
                S->ctlEnv[(size_t)slot * (size_t)N + f] = env;
            }
        }
    }
    switch (S->mode & 7u) {                                       // K2 + K3
#define C(M) case M: render_all<M>(*S, A, bus); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7)
#undef C
    }
    S->lastK = K; S->lastN = N; S->reportsFresh = true;
    return 0;
}

int zlsim_reports(ZlSim *S, zlhip_voice_report *out)
{
    for (int v = 0; v < S->V; ++v) {
        const ZlReport &r = S->reports[(size_t)v];
        out[v].playing = r.playing ? 1 : 0; out[v].valid = r.valid;
        out[v].gain = r.valid ? __builtin_bit_cast(float, r.peak_bits) * 0.5f : 0.0f;
        out[v].progress = r.progress; out[v].clip = r.clip; out[v].reserved = 0; out[v].source_sample_position = r.P;
    }
    return 0;
}

int zlsim_trace(ZlSim *S, int32_t *out) { std::memcpy(out, S->trace.data(), S->trace.size() * sizeof(int32_t)); return 0; }
int zlsim_block_peaks(ZlSim *S, int32_t *out)
{
    for (size_t i = 0; i < S->levels.size(); ++i) { out[2 * i] = S->levels[i].peak_l; out[2 * i + 1] = S->levels[i].peak_r; }
    return 0;
}
int zlsim_expanded_blocks(ZlSim *S) { return S->expanded; }
unsigned long long zlsim_slow_blocks(ZlSim *S) { return S->stats.slow_blocks; }
unsigned long long zlsim_source_bytes(ZlSim *S) { return S->stats.source_bytes; }
// segments used by (block, voice) of the last batch (0 for slow / inactive blocks)
int zlsim_nseg(ZlSim *S, int k, int v) { return S->planHdr[(size_t)k * S->V + v].nseg; }
int zlsim_num_runs(ZlSim *S, int v) { return S->runs[(size_t)v].n; }
int zlsim_num_tsegs(ZlSim *S, int v) { return S->runs[(size_t)v].nts; }
// segment j of voice v of the last window: out = {P, step, t, flags, E, estep}
void zlsim_get_tseg(ZlSim *S, int v, int j, double *out)
{
    const ZlTSeg &g = S->tsegs[(size_t)v * ZL_MAXTSEG + (size_t)j];
    out[0] = g.P; out[1] = g.step; out[2] = g.t; out[3] = g.flags; out[4] = g.E; out[5] = g.estep;
}
int zlsim_periodic_segments(ZlSim *S, int v) { return S->runs[(size_t)v].per_n; }
int zlsim_period_start(ZlSim *S, int v) { return S->runs[(size_t)v].per_t0; }   // < 0: the window replayed the voice's cached pass
int zlsim_plan_flags(ZlSim *S, int k, int v) { return S->planHdr[(size_t)k * S->V + v].flags; }

// ---- direct fuzz of the exact-linear-run machinery against the naive recurrence -----------------
// Walks `steps` additions P += r both ways; returns the index of the first mismatch or -1.
long long zlsim_check_linear_runs(double P0, double r, long long steps, long long *runs_out)
{
    double Pn = P0;      // naive
    double P = P0;       // run-based
    long long done = 0, runs = 0;
    while (done < steps) {
        double s; int L;
        zl_linear_run(P, r, 1.0 / r, s, L);
        ++runs;
        if (L == 0) {
            P = P + r; Pn = Pn + r; ++done;
            if (P != Pn) return done;
            continue;
        }
        long long take = L;
        if (take > steps - done) take = steps - done;
        // every intermediate value must match
        for (long long i = 1; i <= take; ++i) {
            Pn = Pn + r;
            const double Pi = fma((double)i, s, P);
            if (Pi != Pn) return done + i;
        }
        P = fma((double)take, s, P);
        done += take;
    }
    if (runs_out) *runs_out = runs;
    return -1;
}

// fp32 envelope ramps: walks e <- e + d both ways (naive float recurrence vs zl_env_linear_run + one real step between
// runs) until the ADSR event (d > 0: e >= limit, d < 0: e <= limit) or `steps`; returns the first mismatching step or -1.
long long zlsim_check_env_runs(float e0, float d, float limit, long long steps, long long *runs_out, long long *linear_out)
{
    float en = e0, e = e0;
    long long done = 0, runs = 0, lin = 0;
    auto event = [&](float x) { return d > 0.0f ? x >= limit : x <= limit; };
    while (done < steps) {
        float es; int Lc;
        zl_env_linear_run(e, d, limit, es, Lc);
        ++runs;
        long long take = Lc;
        if (take > steps - done) take = steps - done;
        for (long long i = 1; i <= take; ++i) {
            en = en + d;                                          // the recurrence, one fp32 addition per step
            const float ei = (float)fma((double)i, (double)es, (double)e);
            if (ei != en) return done + i;
            if (event(en)) return -(done + i) - 2;                // the run must stop before the event
        }
        lin += take;
        e = (float)fma((double)take, (double)es, (double)e);
        done += take;
        if (done >= steps) break;
        e = e + d; en = en + d; ++done;                           // the real step between two runs
        if (e != en) return done;
        if (event(en)) break;
    }
    if (runs_out) *runs_out = runs;
    if (linear_out) *linear_out = lin;
    return -1;
}

int zlsim_steps_to_reach(double P, double r, double X)
{
    double s; int L;
    zl_linear_run(P, r, 1.0 / r, s, L);
    return zl_steps_to_reach(P, s, 1.0 / r, L, X);
}

}  // extern "C"
