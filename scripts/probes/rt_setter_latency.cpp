// rt_setter_latency.cpp -- latency of the libzl-named real-time cycle (libzl_hotpath_process: 12 channels x 8 voices through the
// resident kernel) with and without a second thread that turns a pan knob 200 times per second (ClipAudioSource_setPan), the
// cycle paced at the JACK period.  VERDICT r2 item 4: a parameter edit must not make the JACK thread wait behind HIP calls nor
// evict the resident kernel -- p99 with the setter thread within 10 % of the quiet p99.
//   g++ -O2 -std=c++17 -I include scripts/probes/rt_setter_latency.cpp -L libzl_amd/lib -lzlhip -Wl,-rpath,$PWD/libzl_amd/lib -lpthread -o scripts/probes/_build/rt_setter_latency
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "libzl_hotpath.h"

using clk = std::chrono::steady_clock;

static void run(const char *label, std::vector<ClipAudioSource *> &clips, int cycles, int nframes, double fs, bool setter, uint64_t &block, bool commands = false)
{
    std::atomic<bool> stop{false};
    std::atomic<long> edits{0};
    std::thread th, th2;
    if (commands) {
        // a second caller thread: retriggers and stops clips 50 times a second through the request queue
        th2 = std::thread([&] {
            auto next = clk::now();
            long i = 0;
            while (!stop.load(std::memory_order_relaxed)) {
                next += std::chrono::microseconds(20000);
                std::this_thread::sleep_until(next);
                ClipAudioSource *c = clips[(size_t)((i * 7) % (long)clips.size())];
                if (i % 3 == 2) ClipAudioSource_stopOnChannel(c, (int)((i * 7) % (long)clips.size()) / 8 - 2);
                else ClipAudioSource_playOnChannel(c, true, (int)((i * 7) % (long)clips.size()) / 8 - 2);
                ++i;
            }
        });
    }
    if (setter) {
        th = std::thread([&] {
            auto next = clk::now();
            long i = 0;
            while (!stop.load(std::memory_order_relaxed)) {
                next += std::chrono::microseconds(5000);          // 200 Hz
                std::this_thread::sleep_until(next);
                ClipAudioSource_setPan(clips[(size_t)(i % (long)clips.size())], (float)std::sin(0.01 * (double)i));
                ++i;
            }
            edits.store(i);
        });
    }
    std::vector<float> L(12 * (size_t)nframes), R(12 * (size_t)nframes);
    std::vector<double> us;
    us.reserve((size_t)cycles);
    const uint64_t period = (uint64_t)std::llround(1e6 * nframes / fs);
    const auto periodNs = std::chrono::nanoseconds((long long)std::llround(1e9 * nframes / fs));
    auto tNext = clk::now();
    for (int k = 0; k < cycles + 50; ++k, ++block) {
        while (clk::now() < tNext) { }                              // paced: one cycle per JACK period
        tNext += periodNs;
        zlhip_clock c;
        c.current_usecs = block * period; c.next_usecs = (block + 1) * period; c.jack_playhead = 0; c.jack_playhead_usecs = 0; c.jack_subbeat_length_usecs = 5208;
        const auto t0 = clk::now();
        const int rc = libzl_hotpath_process((uint32_t)nframes, &c, L.data(), R.data());
        const auto t1 = clk::now();
        if (rc != 0) { std::fprintf(stderr, "process failed: %d\n", rc); std::exit(1); }
        if (k >= 50) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    stop.store(true);
    if (th.joinable()) th.join();
    if (th2.joinable()) th2.join();
    std::sort(us.begin(), us.end());
    uint64_t starts = 0, cyc = 0;
    zlhip_rt_stats(libzl_hotpath_engine(), &starts, &cyc);
    std::printf("%-28s p50 %7.1f us  p99 %7.1f us  p99.9 %7.1f us  max %7.1f us  (%d cycles of %d frames, %ld setPan calls, resident-kernel launches so far %llu)\n", label,
                us[us.size() / 2], us[(size_t)((double)us.size() * 0.99)], us[(size_t)((double)us.size() * 0.999)], us.back(), cycles, nframes, edits.load(), (unsigned long long)starts);
    std::fflush(stdout);
}

int main(int argc, char **argv)
{
    const int cycles = argc > 1 ? std::atoi(argv[1]) : 6000;
    const int nframes = 256;
    const double fs = 48000.0;
    initJuce();
    if (libzl_hotpath_status() != 0) { std::fprintf(stderr, "no engine\n"); return 1; }
    std::vector<ClipAudioSource *> clips;
    std::vector<float> a(96000), b(96000);
    for (size_t i = 0; i < a.size(); ++i) { a[i] = (float)std::sin(0.001 * (double)i); b[i] = (float)std::cos(0.0013 * (double)i); }
    for (int v = 0; v < 96; ++v) {
        ClipAudioSource *c = ClipAudioSource_newFromBuffer(a.data(), b.data(), 96000 - 64 - v, fs, "probe");
        if (!c) return 1;
        ClipAudioSource_setLength(c, 3.5f, 120);
        clips.push_back(c);
        ClipAudioSource_playOnChannel(c, true, v / 8 - 2);           // 8 voices on each of the 12 channels
    }
    uint64_t block = 0;
    run("warm-up", clips, 500, nframes, fs, false, block);
    for (int rep = 0; rep < 2; ++rep) {
        run("quiet", clips, cycles, nframes, fs, false, block);
        run("200 Hz setPan thread", clips, cycles, nframes, fs, true, block);
    }
    run("setPan + 50 Hz play/stop", clips, cycles, nframes, fs, true, block, true);
    for (ClipAudioSource *c : clips) ClipAudioSource_destroy(c);
    shutdownJuce();
    return 0;
}
