// What puts D2H copies into the slow state: a large hipMalloc, the memset kernel, or the hipFree?  Does it recover?
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/_build/d2h_probe4 scripts/probes/d2h_probe4.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static char *host, *dev; static hipStream_t s; static const size_t bytes = 128u << 20;
static int copy(const char *name)
{
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipStreamSynchronize(s));
        const double t0 = now();
        for (size_t o = 0; o < bytes; o += (32u << 20)) CK(hipMemcpyAsync(host + o, dev, 32u << 20, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        best = std::min(best, now() - t0);
    }
    std::printf("%-70s %.1f GB/s\n", name, bytes / best / 1e9);
    return 0;
}
int main()
{
    CK(hipMalloc((void **)&dev, 64u << 20)); CK(hipMemset(dev, 1, 64u << 20));
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipHostMalloc((void **)&host, bytes)); std::memset(host, 0, bytes);
    if (copy("start")) return 1;
    char *big; CK(hipMalloc((void **)&big, 6ull << 30));
    if (copy("after hipMalloc(6 GB)")) return 1;
    CK(hipMemset(big, 0, 6ull << 30)); CK(hipDeviceSynchronize());
    if (copy("after hipMemset of it")) return 1;
    CK(hipFree(big));
    if (copy("after hipFree of it")) return 1;
    sleep(2);
    if (copy("2 s later")) return 1;
    char *sm; CK(hipMalloc((void **)&sm, 256u << 20)); CK(hipMemset(sm, 0, 256u << 20)); CK(hipDeviceSynchronize());
    if (copy("after a 256 MB malloc + memset")) return 1;
    CK(hipFree(sm));
    if (copy("after its free")) return 1;
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); hipStream_t keep = s; s = s2;
    if (copy("on a new stream")) return 1;
    s = keep;
    return 0;
}
