// vram_mailbox_probe.cpp -- can the HOST write device memory directly (large BAR), so that a mailbox could live in VRAM and the resident
// kernel would poll local memory instead of reading host memory over PCIe?  Tries fine-grained device memory (hipExtMallocWithFlags) and plain
// hipMalloc memory: a CPU store under a SIGSEGV / SIGBUS handler (device mappings are not inherited by a forked child), then a kernel reads the word back.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/vram_mailbox_probe.cpp -o scripts/probes/_build/vram_mailbox_probe
#include <hip/hip_runtime.h>
#include <setjmp.h>
#include <signal.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>

__global__ void read_word(const volatile unsigned long long *p, unsigned long long *out) { *out = *p; }
__global__ void spin_until(volatile unsigned long long *p, unsigned long long want, unsigned long long *ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load((unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want) { if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) break; }
    *ticks = __builtin_amdgcn_s_memrealtime() - t0;
}

static bool try_kind(const char *name, unsigned flags, bool ext)
{
    unsigned long long *d = nullptr, *out = nullptr, *hout = nullptr;
    hipError_t st = ext ? hipExtMallocWithFlags((void **)&d, 4096, flags) : hipMalloc((void **)&d, 4096);
    if (st != hipSuccess) { std::printf("%s: allocation failed: %s\n", name, hipGetErrorString(st)); (void)hipGetLastError(); return false; }
    (void)hipMemset(d, 0, 4096); (void)hipDeviceSynchronize();
    static sigjmp_buf jb;
    struct sigaction sa, old1, old2; std::memset(&sa, 0, sizeof sa);
    sa.sa_handler = [](int) { siglongjmp(jb, 1); };
    sigaction(SIGSEGV, &sa, &old1); sigaction(SIGBUS, &sa, &old2);
    bool faulted = false;
    if (sigsetjmp(jb, 1) == 0) { volatile unsigned long long *q = d; *q = 0x1234567812345678ull; } else faulted = true;
    sigaction(SIGSEGV, &old1, nullptr); sigaction(SIGBUS, &old2, nullptr);
    if (faulted) { std::printf("%s: the CPU cannot store to it (fault)\n", name); (void)hipFree(d); return false; }
    volatile unsigned long long *p = d; *p = 0xabcdef01ull; __sync_synchronize();
    (void)hipMalloc((void **)&out, 8); (void)hipHostMalloc((void **)&hout, 8);
    hipLaunchKernelGGL(read_word, dim3(1), dim3(1), 0, 0, d, out);
    (void)hipMemcpy(hout, out, 8, hipMemcpyDeviceToHost);
    std::printf("%s: CPU store works; the device reads back %llx (%s)\n", name, *hout, *hout == 0xabcdef01ull ? "visible" : "NOT visible");
    const bool ok = *hout == 0xabcdef01ull;
    if (ok) {
        // latency: a resident one-lane kernel spins on the word; the host stores the value after a pause; device ticks (100 MHz) from kernel start do not
        // measure the hand-off, so measure the host side: store -> completion of the kernel
        double best = 1e9;
        for (int rep = 0; rep < 20; ++rep) {
            *p = 0; __sync_synchronize();
            hipLaunchKernelGGL(spin_until, dim3(1), dim3(1), 0, 0, d, 77ull + rep, out);
            usleep(2000);
            const auto t0 = std::chrono::steady_clock::now();
            *p = 77ull + rep; __sync_synchronize();
            (void)hipDeviceSynchronize();
            best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        std::printf("%s: host store -> spinning kernel sees it and ends -> hipDeviceSynchronize returns: best %.1f us\n", name, best);
    }
    (void)hipFree(d);
    return ok;
}

// ping-pong: the kernel waits for word A to become k and answers by writing k into word B (mapped host memory); the host times
// store(A = k) -> B == k.  A in mapped host memory (the device polls over PCIe) against A in VRAM (the host stores over the BAR).
__global__ void pingpong(volatile unsigned long long *a, volatile unsigned long long *b, int rounds, int words, unsigned long long *sink)
{
    unsigned long long acc = 0;
    for (int k = 1; k <= rounds; ++k) {
        while (__hip_atomic_load((unsigned long long *)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned long long)k) { }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        for (int i = 1; i <= words; ++i) acc += a[i];              // the command's body: `words` more words from the same place
        __hip_atomic_store((unsigned long long *)b, (unsigned long long)k + (acc & 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    *sink = acc;
}

static void pingpong_row(const char *name, unsigned long long *a_host_view, unsigned long long *a_dev_view, int words)
{
    unsigned long long *b = nullptr, *bd = nullptr, *sink = nullptr;
    (void)hipHostMalloc((void **)&b, 64); (void)hipHostGetDevicePointer((void **)&bd, b, 0); (void)hipMalloc((void **)&sink, 8);
    const int rounds = 2000;
    for (int i = 0; i <= words; ++i) ((volatile unsigned long long *)a_host_view)[i] = 0;
    *b = 0; __sync_synchronize();
    hipLaunchKernelGGL(pingpong, dim3(1), dim3(1), 0, 0, a_dev_view, bd, rounds, words, sink);
    usleep(3000);
    std::vector<double> us;
    for (int k = 1; k <= rounds; ++k) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 1; i <= words; ++i) ((volatile unsigned long long *)a_host_view)[i] = (unsigned long long)k * 3 + i;
        __sync_synchronize();
        *(volatile unsigned long long *)a_host_view = (unsigned long long)k;
        __sync_synchronize();
        while (*(volatile unsigned long long *)b != (unsigned long long)k) { }
        us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    (void)hipDeviceSynchronize();
    std::sort(us.begin(), us.end());
    std::printf("ping-pong, command in %-20s (%2d words): p50 %.2f us  p99 %.2f us  min %.2f us\n", name, words, us[us.size() / 2], us[(size_t)(us.size() * 0.99)], us[0]);
    (void)hipHostFree(b); (void)hipFree(sink);
}

int main()
{
    {
        unsigned long long *h = nullptr, *hd = nullptr, *v = nullptr;
        (void)hipHostMalloc((void **)&h, 4096); (void)hipHostGetDevicePointer((void **)&hd, h, 0);
        (void)hipMalloc((void **)&v, 4096); (void)hipMemset(v, 0, 4096); (void)hipDeviceSynchronize();
        for (int words : {0, 24}) { pingpong_row("mapped host memory", h, hd, words); pingpong_row("VRAM (BAR store)", v, v, words); }
        (void)hipHostFree(h); (void)hipFree(v);
    }
    try_kind("fine-grained device memory (hipExtMallocWithFlags)", hipDeviceMallocFinegrained, true);
    try_kind("uncached device memory (hipExtMallocWithFlags)", hipDeviceMallocUncached, true);
    try_kind("plain hipMalloc", 0, false);
    // reference: the same hand-off through mapped HOST memory (what the mailbox uses today)
    unsigned long long *h = nullptr, *hd = nullptr, *out = nullptr;
    (void)hipHostMalloc((void **)&h, 4096); (void)hipHostGetDevicePointer((void **)&hd, h, 0); (void)hipMalloc((void **)&out, 8);
    double best = 1e9;
    for (int rep = 0; rep < 20; ++rep) {
        *h = 0; __sync_synchronize();
        hipLaunchKernelGGL(spin_until, dim3(1), dim3(1), 0, 0, hd, 77ull + rep, out);
        usleep(2000);
        const auto t0 = std::chrono::steady_clock::now();
        *(volatile unsigned long long *)h = 77ull + rep; __sync_synchronize();
        (void)hipDeviceSynchronize();
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::printf("mapped host memory: host store -> spinning kernel sees it and ends -> hipDeviceSynchronize returns: best %.1f us\n", best);
    return 0;
}
