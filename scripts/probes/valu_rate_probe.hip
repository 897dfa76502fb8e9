// Issue cost of the VALU operations the position arithmetic could be built from (gfx950): cycles per wave-instruction on one SIMD,
// measured with dependent-free streams of each op (8 independent chains per lane, 4 waves per SIMD so the pipeline is never starved).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/_build/valu_rate_probe scripts/probes/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void __launch_bounds__(256) k(double *out, int iters, double seed, unsigned long long *cycles)
{
    double d[8]; float fl[8]; uint32_t u[8]; uint64_t q[8]; int n[8];
    for (int i = 0; i < 8; ++i) { d[i] = seed + threadIdx.x * 0.37 + i; fl[i] = (float)d[i]; u[i] = (uint32_t)(threadIdx.x * 2654435761u + i); q[i] = ((uint64_t)u[i] << 20) + i; n[i] = i; }
    const double step = seed * 1.0000001;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(step));
            if (OP == 1) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(d[i]));
            if (OP == 2) asm volatile("v_fract_f64 %0, %1" : "=v"(d[i]) : "v"(d[i]));
            if (OP == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(fl[i]) : "v"(d[i]));
            if (OP == 4) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(step));
            if (OP == 5) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]) : "vcc");
            if (OP == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(n[i]));
            if (OP == 7) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(u[i]) : "v"(n[i]));
            if (OP == 8) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(fl[i]) : "v"(u[i]));
            if (OP == 9) asm volatile("v_lshrrev_b64 %0, 7, %0" : "+v"(q[i]));
            if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(fl[i]) : "v"(fl[(i + 1) & 7]));
            if (OP == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d[i]) : "v"(step));
            if (OP == 12) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(n[i]));
            if (OP == 13) asm volatile("v_floor_f64 %0, %1" : "=v"(d[i]) : "v"(d[i]));
            if (OP == 14) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u[i]) : "v"(d[i]));
            if (OP == 15) asm volatile("v_readfirstlane_b32 s20, %0" :: "v"(u[i]) : "s20");
            if (OP == 16) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(u[i]) : "v"(n[i]));
            if (OP == 17) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(step));
            if (OP == 18) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(n[i]));
            if (OP == 19) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(fl[i]) : "v"(n[i]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double acc = 0; for (int i = 0; i < 8; ++i) acc += d[i] + fl[i] + u[i] + (double)q[i] + n[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int OP> static int run(const char *name, double *out, unsigned long long *cyc)
{
    const int iters = 20000;
    // one workgroup of 256 lanes on one CU: its 4 waves sit one per SIMD -> every SIMD issues 8 * iters instructions of its wave
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256), 0, 0, out, iters, 1.25, cyc); CK(hipDeviceSynchronize()); }
    const double one = (double)*cyc / (8.0 * iters);
    // four workgroups on one CU would share SIMDs; instead 1024 lanes in one workgroup: 4 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(1024), 0, 0, out, iters, 1.25, cyc); CK(hipDeviceSynchronize());
    const double four = (double)*cyc / (8.0 * iters * 4);
    std::printf("%-18s %6.2f cycles per instruction with 1 wave per SIMD, %6.2f with 4 waves per SIMD (issue cost)\n", name, one, four);
    return 0;
}

int main()
{
    double *out; unsigned long long *cyc; CK(hipMalloc((void **)&out, 1024 * 8)); CK(hipHostMalloc((void **)&cyc, 8));
    int r = 0;
    r |= run<10>("v_fma_f32", out, cyc); r |= run<11>("v_pk_fma_f32", out, cyc); r |= run<0>("v_fma_f64", out, cyc); r |= run<4>("v_add_f64", out, cyc);
    r |= run<17>("v_mul_f64", out, cyc); r |= run<1>("v_cvt_i32_f64", out, cyc); r |= run<14>("v_cvt_u32_f64", out, cyc); r |= run<2>("v_fract_f64", out, cyc);
    r |= run<13>("v_floor_f64", out, cyc); r |= run<3>("v_cvt_f32_f64", out, cyc); r |= run<12>("v_cvt_f64_i32", out, cyc); r |= run<18>("v_ldexp_f64", out, cyc);
    r |= run<5>("v_mad_u64_u32", out, cyc); r |= run<6>("v_mul_lo_u32", out, cyc); r |= run<7>("v_mad_u32_u24", out, cyc); r |= run<8>("v_cvt_f32_u32", out, cyc);
    r |= run<19>("v_cvt_f32_i32", out, cyc); r |= run<9>("v_lshrrev_b64", out, cyc); r |= run<16>("v_lshl_add_u32", out, cyc); r |= run<15>("v_readfirstlane", out, cyc);
    return r;
}
