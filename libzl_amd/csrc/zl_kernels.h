// zl_kernels.h -- launchers of the HIP kernels in zl_kernels.hip (return 0 or a hipError_t value).
#pragma once
#include <hip/hip_runtime_api.h>
#include "zl_types.h"

int zl_launch_apply_ops(const ZlBatch &A, hipStream_t s);
int zl_launch_plan(const ZlBatch &A, int force_slow, hipStream_t s);
int zl_launch_assemble(const ZlBatch &A, hipStream_t s);
int zl_launch_render(const ZlBatch &A, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
int zl_launch_finalize(const ZlBatch &A, const float *bus_in, hipStream_t s);
int zl_launch_reports(const ZlReport *reports, int V, float *gain_out, ZlReport *host_reports, float *host_gain,
                      const ZlBatchStats *stats, ZlBatchStats *host_stats, hipStream_t s, hipEvent_t ev_done = nullptr);
int zl_launch_levels_tick(ZlLevelsState *state, const ZlBlockLevels *levels, int B, int N, int with_hold_bus, hipStream_t s);
int zl_launch_passthrough(const void *params_dev, const float *in, float *out, int B, long long frames, hipStream_t s);
int zl_launch_deliver(const float *bus, void *out, int pcm16, int B, long long in_stride, long long off, long long frames, long long total, hipStream_t s);
int zl_launch_interleave(const float *L, const float *R, float *dst, int length, int pad, hipStream_t s);
int zl_launch_reduce_scan(const float *pieces, int npieces, long long stride, long long units, int N, int off, float *out, ZlUnitLevels *lv, hipStream_t s);
int zl_launch_levels_import(const ZlUnitLevels *units, ZlBlockLevels *levels, int B, int K, hipStream_t s);
int zl_launch_rt_loop(const ZlBatch &A, void *mailbox_dev, void *dev_state, unsigned long long first_seq, unsigned long long idle_ticks, float *gain_out,
                      ZlReport *host_reports, float *host_gain, ZlOpRange *dev_ranges, int vw, int threads, hipStream_t s);
int zl_rt_loop_capacity(uint32_t mode, int wide, int threads, int device);
