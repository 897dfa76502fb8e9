// glds_probe -- what does one LDS-DMA wave-instruction (global_load_lds_dwordx4) cost on gfx950 when only part of its
// lanes are active?  Every wave streams its own contiguous region of a large buffer (no byte is read twice) in pieces of
// `nl` active lanes x 16 bytes through a ring of D slots (counted s_waitcnt vmcnt), with no compute at all.  Compared with
// the same stream through registers (global_load_dwordx4).  Diagnostic only: not part of the library.
// build: hipcc --offload-arch=gfx950 -O3 -o glds_probe scripts/probes/glds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
extern __shared__ __attribute__((aligned(16))) char dyn[];
static __device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p; }
static __device__ __forceinline__ void piece(const char *g, uint32_t lane16, uint32_t nbytes, uint32_t dst)
{
    unsigned keep; unsigned long long save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\tv_cmp_lt_u32 vcc, %3, %4\n\ts_and_saveexec_b64 %1, vcc\n\t"
                 "global_load_lds_dwordx4 %2, off\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(save) : "v"(g), "v"(lane16), "v"(nbytes), "s"(dst) : "memory", "vcc", "scc");
}
__global__ void k_fill(float *p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f; }
template <int D> __global__ void __launch_bounds__(256) k_dma(const char *buf, size_t bytes_per_wave, int iters, int nl, float *out)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t gw = (size_t)blockIdx.x * 4 + wave;
    const char *base = buf + gw * bytes_per_wave + lane * 16;
    const uint32_t ring = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(dyn + wave * D * 1024));
    const uint32_t lane16 = lane * 16, nb = nl * 16;
    for (int i = 0; i < D; ++i) piece(base + (size_t)i * nb, lane16, nb, ring + i * 1024);
    float acc = 0;
    int slot = 0;
    for (int i = 0; i < iters; ++i) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D - 1) : "memory");
        acc += *reinterpret_cast<const float *>(dyn + wave * D * 1024 + slot * 1024 + lane * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        piece(base + (size_t)(i + D) * nb, lane16, nb, ring + slot * 1024);
        slot = slot + 1 == D ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane < nl) atomicAdd(out, acc);                            // every float of the buffer is 1.0f: the sum counts the pieces seen
}
template <int D> __global__ void __launch_bounds__(256) k_reg(const char *buf, size_t bytes_per_wave, int iters, int nl, float *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t gw = (size_t)blockIdx.x * 4 + wave;
    const char *base = buf + gw * bytes_per_wave + lane * 16;
    const uint32_t nb = nl * 16;
    float4 r[D];
    float acc = 0;
    for (int i = 0; i < iters; i += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) r[u] = lane < nl ? *reinterpret_cast<const float4 *>(base + (size_t)(i + u) * nb) : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < D; ++u) acc += r[u].x;
    }
    if (lane < nl) atomicAdd(out, acc);
}
int main()
{
    const size_t total = (size_t)6 << 30;                                   // 6 GiB: nothing fits a cache
    char *buf; float *out;
    if (hipMalloc(&buf, total + (1 << 20)) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, (float *)buf, total / 4);
    if (hipDeviceSynchronize() != hipSuccess) { printf("fill failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern, int D, int blocks, int nl, int lds) {
        const size_t waves = (size_t)blocks * 4, bpw = (total / waves) & ~(size_t)1023;
        const int iters = (int)(bpw / (nl * 16)) - D - 1;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(out, 0, 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, buf, bpw, iters, nl, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)waves * iters * nl * 16;
        float seen = 0; hipMemcpy(&seen, out, 4, hipMemcpyDeviceToHost);
        printf("%-8s D=%2d blocks/CU=%d lanes=%2d (%4d B/piece): %7.1f GB/s  %.1f ns per piece per CU  (sum %.0f, expected %.0f)\n", name, D, blocks / 256, nl, nl * 16,
               bytes / ms / 1e6, ms * 1e6 / ((double)waves * iters / 256), seen, (double)waves * iters * nl);
    };
    for (int nl : {64, 48, 34, 17}) {
        run("lds-dma", k_dma<8>, 8, 256 * 3, nl, 4 * 8 * 1024);
        run("lds-dma", k_dma<8>, 8, 256 * 4, nl, 4 * 8 * 1024);
        run("lds-dma", k_dma<16>, 16, 256 * 2, nl, 4 * 16 * 1024);
        run("lds-dma", k_dma<4>, 4, 256 * 8, nl, 4 * 4 * 1024);
        run("regs", k_reg<8>, 8, 256 * 4, nl, 0);
        run("regs", k_reg<8>, 8, 256 * 8, nl, 0);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
