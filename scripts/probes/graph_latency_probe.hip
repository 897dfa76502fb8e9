// Launch-chain latency on this platform: N dependent tiny kernels + a host wait, issued (a) as N hipLaunchKernelGGL calls on one stream,
// (b) as one replay of a captured hipGraph (hipGraphLaunch), (c) as (b) with the kernel parameters of every node set before each replay
// (what a real-time cycle needs: its clock and operation counts travel in the kernel arguments).  The kernels do ~1 us of dependent work
// each; the host spins on an event (hipEventQuery) the way zlhip_render waits.   build: hipcc --offload-arch=gfx950 -O2 -o graph_latency_probe graph_latency_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void step(unsigned long long *p, unsigned long long add)
{
    unsigned long long v = *p;
    for (int i = 0; i < 64; ++i) v = v * 6364136223846793005ull + add;      // a short dependent chain
    *p = v;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int N = argc > 1 ? std::atoi(argv[1]) : 4, iters = 3000;
    unsigned long long *d = nullptr;
    CK(hipMalloc(&d, 8)); CK(hipMemset(d, 0, 8));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    auto wait = [&] { while (hipEventQuery(ev) == hipErrorNotReady) { } };
    auto report = [&](const char *name, std::vector<double> &t) {
        std::sort(t.begin(), t.end());
        std::printf("%-58s p50 %6.1f us  p99 %6.1f us\n", name, t[t.size() / 2], t[t.size() * 99 / 100]);
    };
    std::vector<double> t;
    // (a) direct launches
    for (int w = 0; w < 2; ++w) {
        t.clear();
        for (int i = 0; i < iters; ++i) {
            const double t0 = now_us();
            for (int k = 0; k < N; ++k) hipLaunchKernelGGL(step, dim3(1), dim3(64), 0, s, d, (unsigned long long)(i + k));
            (void)hipEventRecord(ev, s); wait();
            t.push_back(now_us() - t0);
        }
    }
    char name[128]; std::snprintf(name, sizeof name, "%d dependent kernels, direct launches", N); report(name, t);
    // (b) graph replay
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < N; ++k) hipLaunchKernelGGL(step, dim3(1), dim3(64), 0, s, d, (unsigned long long)k);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 2; ++w) {
        t.clear();
        for (int i = 0; i < iters; ++i) {
            const double t0 = now_us();
            (void)hipGraphLaunch(ge, s);
            (void)hipEventRecord(ev, s); wait();
            t.push_back(now_us() - t0);
        }
    }
    std::snprintf(name, sizeof name, "%d dependent kernels, one hipGraphLaunch", N); report(name, t);
    // (c) graph replay with fresh kernel parameters per node
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
    for (int w = 0; w < 2; ++w) {
        t.clear();
        for (int i = 0; i < iters; ++i) {
            const double t0 = now_us();
            for (size_t k = 0; k < nn; ++k) {
                unsigned long long add = (unsigned long long)(i + (int)k);
                void *args[2] = { &d, &add };
                hipKernelNodeParams kp; std::memset(&kp, 0, sizeof kp);
                kp.func = (void *)step; kp.gridDim = dim3(1); kp.blockDim = dim3(64); kp.sharedMemBytes = 0; kp.kernelParams = args; kp.extra = nullptr;
                (void)hipGraphExecKernelNodeSetParams(ge, nodes[k], &kp);
            }
            (void)hipGraphLaunch(ge, s);
            (void)hipEventRecord(ev, s); wait();
            t.push_back(now_us() - t0);
        }
    }
    std::snprintf(name, sizeof name, "%d dependent kernels, parameters set + hipGraphLaunch", N); report(name, t);
    // (d) one kernel, for scale
    t.clear();
    for (int i = 0; i < iters; ++i) {
        const double t0 = now_us();
        hipLaunchKernelGGL(step, dim3(1), dim3(64), 0, s, d, (unsigned long long)i);
        (void)hipEventRecord(ev, s); wait();
        t.push_back(now_us() - t0);
    }
    report("1 kernel, direct launch", t);
    return 0;
}
