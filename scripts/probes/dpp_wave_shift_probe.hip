#include <hip/hip_runtime.h>
__global__ void k(const int *in, int *out)
{
    const int v = in[threadIdx.x];
    out[threadIdx.x]       = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
    out[64 + threadIdx.x]  = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
}
int main()
{
    int h[64], o[128], *d, *e;
    for (int i = 0; i < 64; ++i) h[i] = i;
    hipMalloc(&d, sizeof h); hipMalloc(&e, sizeof o);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o, e, sizeof o, hipMemcpyDeviceToHost);
    printf("wave_shl:1 lanes 0,1,15,16,62,63 -> %d %d %d %d %d %d\n", o[0], o[1], o[15], o[16], o[62], o[63]);
    printf("wave_shr:1 lanes 0,1,15,16,62,63 -> %d %d %d %d %d %d\n", o[64], o[65], o[79], o[80], o[126], o[127]);
    return 0;
}
