// D2H strided copies into page-locked memory: hipMemcpy2DAsync vs one hipMemcpyAsync per row vs a kernel storing to mapped host memory.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/_build/d2h_probe scripts/probes/d2h_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_rows(const uint4 *src, uint4 *dst, long long rowVec, long long dpitchVec, int rows)
{
    const int r = blockIdx.y;
    const uint4 *s = src + (long long)r * rowVec; uint4 *d = dst + (long long)r * dpitchVec;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rowVec; i += (long long)gridDim.x * blockDim.x) d[i] = s[i];
}

int main()
{
    const int rows = 16; const size_t rowBytes = 2048 * 256 * 4, dpitch = rowBytes * 4;
    char *dev, *host; CK(hipMalloc((void **)&dev, rows * rowBytes)); CK(hipHostMalloc((void **)&host, rows * dpitch));
    CK(hipMemset(dev, 1, rows * rowBytes)); std::memset(host, 0, rows * dpitch);
    char *hdev; CK(hipHostGetDevicePointer((void **)&hdev, host, 0));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipStreamSynchronize(s));
            const double t0 = now();
            if (mode == 0) CK(hipMemcpy2DAsync(host, dpitch, dev, rowBytes, rowBytes, rows, hipMemcpyDeviceToHost, s));
            else if (mode == 1) { for (int r = 0; r < rows; ++r) CK(hipMemcpyAsync(host + r * dpitch, dev + r * rowBytes, rowBytes, hipMemcpyDeviceToHost, s)); }
            else if (mode == 2) hipLaunchKernelGGL(k_rows, dim3(64, rows), dim3(256), 0, s, (const uint4 *)dev, (uint4 *)hdev, (long long)(rowBytes / 16), (long long)(dpitch / 16), rows);
            else CK(hipMemcpyAsync(host, dev, rows * rowBytes, hipMemcpyDeviceToHost, s));
            const double t1 = now();
            CK(hipStreamSynchronize(s));
            const double t2 = now();
            if (rep) std::printf("mode %d (%s): enqueue %.3f ms, total %.3f ms, %.1f GB/s\n", mode,
                                 mode == 0 ? "memcpy2D" : mode == 1 ? "row memcpys" : mode == 2 ? "kernel to mapped host" : "one contiguous memcpy",
                                 (t1 - t0) * 1e3, (t2 - t0) * 1e3, rows * rowBytes / (t2 - t0) / 1e9);
        }
    }
    return 0;
}
