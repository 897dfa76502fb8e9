// rt_cycle_probe.cpp -- the PUBLISHED real-time figure: latency of one JACK cycle through the C-ABI (zlhip_render / zlhip_render_fanout:
// host buffers in and out, synchronous), measured from C++ -- no interpreter between the clock reads and the call -- for the launched
// path (three kernels + a completion event) and the resident kernel, back to back and paced at the JACK period, at periods of 128 ...
// 1024 frames, with and without the JackPassthrough fan-out.  Every cycle longer than a millisecond is ATTRIBUTED from the engine's own
// trace (zlhip_rt_last_cycle: host before the post / wait for the device / host after; the longest gap between two polls of the
// waiting thread and its involuntary context switches say whether the thread was off its core).
//   g++ -O2 -std=c++17 -I include scripts/probes/rt_cycle_probe.cpp -L libzl_amd/lib -lzlhip -Wl,-rpath,$PWD/libzl_amd/lib -lpthread -o scripts/probes/_build/rt_cycle_probe
//   usage: rt_cycle_probe [cycles = 6000] [--quick]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "zlhip.h"

using clk = std::chrono::steady_clock;

struct Row { int V, B, N; bool resident, paced, fan; int hogs; bool pinned = false; bool idle = false; };   // idle: no voice plays -- the cycle's floor (signalling only)   // pinned: the caller's buffers are page-locked (the kernels write them directly)   // hogs: busy threads next to the cycle thread (more than the box's CPU quota: the attribution's known answer)

static int run(const Row &r, int cycles)
{
    setenv("ZL_RT_PERSISTENT", r.resident ? "1" : "0", 1);
    setenv("ZL_RT_TRACE", "1", 1);
    const double fs = 48000.0;
    const int lf = 96000, vpb = r.V / r.B;
    zlhip_config cfg;
    zlhip_config_default(&cfg);
    cfg.num_buses = r.B; cfg.voices_per_bus = vpb; cfg.max_frames = r.N; cfg.max_batch_blocks = 4; cfg.max_sounds = r.V; cfg.playback_sample_rate = fs;
    cfg.sound_arena_bytes = (uint64_t)(lf + 16) * 8 * (uint64_t)r.V + (1u << 20);
    zlhip_engine *e = nullptr;
    if (zlhip_engine_create(&cfg, &e) != ZLHIP_OK) { std::fprintf(stderr, "no engine\n"); return 1; }
    std::vector<float> a((size_t)lf), b((size_t)lf);
    for (int v = 0; v < r.V; ++v) {
        for (int i = 0; i < lf; ++i) { a[(size_t)i] = (float)std::sin(0.001 * (i + 31 * v)); b[(size_t)i] = (float)std::cos(0.0013 * (i + 17 * v)); }
        int32_t id = -1;
        if (zlhip_sound_upload(e, a.data(), b.data(), lf, fs, &id) != ZLHIP_OK || id != v) { std::fprintf(stderr, "upload failed: %s\n", zlhip_last_error(e)); return 1; }
        zlhip_clip_params p;
        zlhip_clip_params_default(&p, (float)(lf / fs));
        p.length_in_beats = 3.5f; p.length_seconds = (float)((lf - 64 - v % 17) / fs);
        zlhip_clip_set(e, id, &p);
        zlhip_clip_command c;
        zlhip_clip_command_clear(&c);
        c.clip = id; c.midi_note = 60; c.midi_channel = v / vpb - 2; c.start_playback = 1; c.looping = 1; c.change_volume = 1; c.volume = 0.5f;
        if (!r.idle && zlhip_start_voice(e, v / vpb, v % vpb, &c, 0) != 1) { std::fprintf(stderr, "start failed\n"); return 1; }
    }
    std::vector<float> Lv((size_t)r.B * r.N), Rv((size_t)r.B * r.N), fanv(r.fan ? (size_t)r.B * 6 * r.N : 0);
    float *L = Lv.data(), *R = Rv.data(), *fan = fanv.data();
    void *pin = nullptr;
    if (r.pinned) {
        if (zlhip_host_alloc((size_t)r.B * r.N * 8 * sizeof(float), &pin) != ZLHIP_OK) { std::fprintf(stderr, "no pinned memory\n"); return 1; }
        L = (float *)pin; R = L + (size_t)r.B * r.N; fan = R + (size_t)r.B * r.N;
    }
    std::vector<zlhip_passthrough_params> pp((size_t)r.B);
    for (int i = 0; i < r.B; ++i) { zlhip_passthrough_params_default(&pp[(size_t)i]); pp[(size_t)i].dry_amount = 0.9f; pp[(size_t)i].wet_fx1_amount = 0.5f; pp[(size_t)i].pan_amount = 0.1f * (float)(i % 3 - 1); }
    const uint64_t period = (uint64_t)std::llround(1e6 * r.N / fs);
    const auto periodNs = std::chrono::nanoseconds((long long)std::llround(1e9 * r.N / fs));
    std::vector<double> us;
    us.reserve((size_t)cycles);
    struct Slow { int k; double harness; zlhip_rt_cycle_trace t; };
    std::vector<Slow> slow;
    double sumBefore = 0.0, sumWait = 0.0, sumAfter = 0.0;
    std::atomic<bool> stopHogs{false};
    std::vector<std::thread> hogs;
    for (int i = 0; i < r.hogs; ++i) hogs.emplace_back([&stopHogs] { volatile unsigned long long x = 0; while (!stopHogs.load(std::memory_order_relaxed)) x = x + 1; });
    auto tNext = clk::now();
    for (int k = 0; k < cycles + 50; ++k) {
        if (r.paced) { while (clk::now() < tNext) { } tNext += periodNs; }
        zlhip_clock c;
        c.current_usecs = (uint64_t)k * period; c.next_usecs = (uint64_t)(k + 1) * period; c.jack_playhead = 0; c.jack_playhead_usecs = 0; c.jack_subbeat_length_usecs = 5208;
        if (r.fan && (k % 200) == 100) pp[(size_t)(k / 200 % r.B)].pan_amount += 0.01f;      // a knob now and then
        const auto t0 = clk::now();
        const int rc = r.fan ? zlhip_render_fanout(e, r.N, &c, L, R, pp.data(), fan) : zlhip_render(e, r.N, &c, L, R);
        const auto t1 = clk::now();
        if (rc != ZLHIP_OK) { std::fprintf(stderr, "render failed: %d %s\n", rc, zlhip_last_error(e)); return 1; }
        if (k < 50) continue;
        const double d = std::chrono::duration<double, std::micro>(t1 - t0).count();
        us.push_back(d);
        { zlhip_rt_cycle_trace t; if (zlhip_rt_last_cycle(e, &t) == ZLHIP_OK) { sumBefore += t.before_post_us; sumWait += t.wait_us; sumAfter += t.after_us; } }
        if (d > 1000.0) { Slow s; s.k = k; s.harness = d; zlhip_rt_last_cycle(e, &s.t); slow.push_back(s); }
    }
    stopHogs.store(true);
    for (auto &h : hogs) h.join();
    uint64_t starts = 0, cyc = 0;
    zlhip_rt_stats(e, &starts, &cyc);
    std::vector<double> sorted = us;
    std::sort(sorted.begin(), sorted.end());
    std::printf("V=%4d B=%3d N=%4d %s %s %s: p50 %6.1f us  p99 %6.1f us  p99.9 %6.1f us  max %8.1f us  (period %5.0f us, %d cycles, resident launches %llu, cycles over 1 ms: %zu)\n",
                r.V, r.B, r.N, r.resident ? "resident" : "launched", r.paced ? "paced    " : "back2back", r.fan ? (r.pinned ? "fan, pinned" : "fan-out    ") : (r.hogs ? "+hogs      " : (r.idle ? "idle voices" : (r.pinned ? "pinned     " : "           "))),
                sorted[sorted.size() / 2], sorted[(size_t)((double)sorted.size() * 0.99)], sorted[(size_t)((double)sorted.size() * 0.999)], sorted.back(),
                1e6 * r.N / fs, cycles, (unsigned long long)starts, slow.size());
    if (std::getenv("ZL_PROBE_BREAKDOWN"))
        std::printf("    mean inside the engine: host before the post %.2f us, wait for the device %.2f us, host after %.2f us\n", sumBefore / us.size(), sumWait / us.size(), sumAfter / us.size());
    for (size_t si = 0; si < slow.size(); ++si) {
        const Slow &s = slow[si];
        if (si == 8) { std::printf("    ... and %zu more\n", slow.size() - 8); break; }
        std::printf("    cycle %6d: %8.0f us at the caller; inside the engine %8.0f = before the post %6.0f + wait %8.0f + after %5.0f; longest poll gap of the waiting thread %8.0f us, "
                    "involuntary context switches %lld -> %s\n", s.k, s.harness, s.t.total_us, s.t.before_post_us, s.t.wait_us, s.t.after_us, s.t.max_poll_gap_us,
                    (long long)s.t.involuntary_switches,
                    s.t.max_poll_gap_us > 0.5 * s.t.total_us ? "the waiting THREAD was off its core (host scheduler / cgroup quota), not the device"
                    : s.t.before_post_us > 0.5 * s.t.total_us ? "host side, before the cycle was posted (a HIP call: kernel restart / launch)"
                    : s.t.wait_us > 0.5 * s.t.total_us ? "the device (or its runtime) took the time" : "host side, after the device was done");
    }
    std::fflush(stdout);
    zlhip_engine_destroy(e);
    if (pin) zlhip_host_free(pin);
    return 0;
}

int main(int argc, char **argv)
{
    int cycles = 6000;
    bool quick = false;
    for (int i = 1; i < argc; ++i) { if (std::strcmp(argv[i], "--quick") == 0) quick = true; else cycles = std::atoi(argv[i]); }
    if (quick) cycles = std::min(cycles, 2000);
    std::vector<Row> rows;
    for (int res = 0; res < 2; ++res) {
        rows.push_back({96, 12, 256, res == 1, false, false, 0});      // the reference's own shape: 12 channels x 8 voices
        rows.push_back({64, 8, 256, res == 1, false, false, 0});       // BASELINE configs[1]
        rows.push_back({96, 12, 128, res == 1, false, false, 0});
        rows.push_back({96, 12, 512, res == 1, false, false, 0});
        rows.push_back({96, 12, 1024, res == 1, false, false, 0});
        if (!quick) { rows.push_back({96, 12, 2048, res == 1, false, false, 0}); rows.push_back({96, 12, 4096, res == 1, false, false, 0}); }
        rows.push_back({96, 12, 256, res == 1, false, true, 0});       // with the JackPassthrough fan-out (three more pairs per bus over PCIe)
    }
    // the caller's buffers page-locked: the kernels deliver straight into them (no host copy behind the cycle)
    for (int res = 0; res < 2; ++res) {
        Row a{96, 12, 256, res == 1, false, false, 0}; a.pinned = true; rows.push_back(a);
        Row b{96, 12, 256, res == 1, false, true, 0}; b.pinned = true; rows.push_back(b);
        Row c{96, 12, 1024, res == 1, false, false, 0}; c.pinned = true; rows.push_back(c);
    }
    // the floor: no voice plays -- what a cycle costs in signalling alone (mailbox, hand-off between workgroups, completion)
    for (int res = 0; res < 2; ++res) { Row f{96, 12, 256, res == 1, false, false, 0}; f.idle = true; f.pinned = true; rows.push_back(f); }
    for (int res = 0; res < 2; ++res) {
        rows.push_back({96, 12, 256, res == 1, true, false, 0});       // paced: one cycle per JACK period, as JACK runs it (the GPU idles in between)
        rows.push_back({96, 12, 256, res == 1, true, true, 0});
        { Row d{96, 12, 256, res == 1, true, true, 0}; d.pinned = true; rows.push_back(d); }
        if (!quick) { rows.push_back({96, 12, 512, res == 1, true, false, 0}); rows.push_back({96, 12, 1024, res == 1, true, false, 0}); }
    }
    // the attribution's known answer: far more busy threads than the box grants this job CPUs (cgroup quota) -- the cycle thread loses
    // its core every now and then, and the slow cycles must read "the waiting THREAD was off its core", not "the device"
    const int hw = (int)std::thread::hardware_concurrency();
    rows.push_back({96, 12, 256, true, false, false, std::max(24, std::min(hw, 64))});
    for (const Row &r : rows) {
        int n = cycles;
        if (r.paced) n = std::min(cycles, (int)(8.0 * 48000.0 / r.N));   // at most ~8 s per paced row
        if (run(r, n) != 0) return 1;
    }
    return 0;
}
