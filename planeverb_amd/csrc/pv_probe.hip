// pv_probe.hip -- the shader clock the chip sustains right now, for bench.py's device record.
// One wave sleeps 100 x s_sleep 127 = 812 800 shader-clock cycles and times that with s_memrealtime (constant 100 MHz): the
// clock of the moment it runs in, whatever else the chip is doing (tools/clock_probe.hip, made callable).  Launched on a
// stream of its own so that it can sit beside the solvers' launches.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "pv_launch.h"

namespace pva {

namespace {
__global__ void pv_clock_probe_kernel(unsigned long long* out) {
    const unsigned long long c0 = clock64(), s0 = wall_clock64();
    for (int i = 0; i < 100; ++i) __builtin_amdgcn_s_sleep(127);
    out[0] = wall_clock64() - s0;  // 10 ns ticks
    out[1] = clock64() - c0;       // s_memtime ticks over the same span
}
}  // namespace

// The status words of a run -- error flag, the analysis' two cell counts, the resident kernel's claim counter -- written to
// pinned host memory by the LAST kernel of the run, so that Solver::sync() reads them after its one stream synchronisation
// instead of copying them back and synchronising a second time (~25-40 us of host latency per run: a tenth of a 70^2 run).
namespace {
__global__ void pv_run_status_kernel(int* err, int* counts, const unsigned* claims, int* out) {
    if (threadIdx.x == 0) {
        out[0] = *err;
        *err = 0;  // reported: the next run -- a resident run has no launch in front of its kernel that would -- starts clean
        out[1] = counts[0];
        out[2] = counts[1];
        out[3] = claims ? (int)*claims : -1;
        out[4] = counts[3];
        // the cell counters and the list of groups with work start the next run's analysis at zero (pv_onset_kernel adds to them)
        counts[1] = 0;
        counts[2] = 0;
        counts[3] = 0;
        counts[4] = 0;
    }
}
}  // namespace

void launchRunStatus(int* err, int* counts, const unsigned* claims, int* outHost, hipStream_t stream) {
    hipLaunchKernelGGL(pv_run_status_kernel, dim3(1), dim3(64), 0, stream, err, counts, claims, outHost);
}

// Do two idle streams take turns?  Streams are dealt hardware queues, and queues that sit on the same dispatch pipe are served
// one packet at a time: a kernel with more workgroups than the chip holds occupies its pipe until its LAST workgroup has been
// launched, and a kernel of another stream on that pipe starts only then -- two K-step stencil launches of 6 000 tiles each then
// run one after the other although they are "in flight together".  (A one-wave kernel does not show it: it is launched at once.)
// The probe: four rounds of workgroups that sleep 10 us each on `a`, one stamp on `b` right behind.  On separate pipes the
// stamp is taken while `a` still has workgroups to launch; on one pipe after the last of them has started.
// stamps: 4 words of pinned host memory.  (~0.1 ms; Solver::claimOwnQueue)
namespace {
__global__ __launch_bounds__(256) void pv_queue_sleep_kernel(unsigned long long* out) {
    const unsigned long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t0;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[1] = t0;  // the last workgroup has been launched
    while (wall_clock64() - t0 < 1000ull) __builtin_amdgcn_s_sleep(32);  // 10 us of the 100 MHz counter
}
__global__ void pv_queue_stamp_kernel(unsigned long long* out) { out[2] = wall_clock64(); }
}  // namespace

bool streamsShareQueue(hipStream_t a, hipStream_t b, unsigned long long* stamps) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    const unsigned blocks = 4u * 8u * (unsigned)prop.multiProcessorCount;  // four times what the chip holds at once
    int votes = 0;
    for (int i = 0; i < 3; ++i) {  // (two of three: a launch can be late for other reasons)
        stamps[1] = stamps[2] = 0;
        hipLaunchKernelGGL(pv_queue_sleep_kernel, dim3(blocks), dim3(256), 0, a, stamps);
        hipLaunchKernelGGL(pv_queue_stamp_kernel, dim3(1), dim3(64), 0, b, stamps);
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
        if (stamps[2] >= stamps[1]) ++votes;
        if (const char* e = getenv("PLANEVERB_AMD_QUEUE_PROBE"))
            if (atoi(e) == 3)
                std::fprintf(stderr, "[planeverb_amd] queue probe: stamp at %.1f us of a dispatch window of %.1f us\n",
                             ((double)stamps[2] - (double)stamps[0]) * 0.01, ((double)stamps[1] - (double)stamps[0]) * 0.01);
    }
    return votes >= 2;
}

// MHz by the s_sleep method (and by s_memtime in *byMemtime), or 0 on failure
float clockProbeMHz(int device, float* byMemtime) {
    if (hipSetDevice(device) != hipSuccess) return 0.f;
    static thread_local hipStream_t stream = nullptr;
    static thread_local unsigned long long* host = nullptr;
    if (!stream && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return 0.f;
    if (!host && hipHostMalloc((void**)&host, 16) != hipSuccess) return 0.f;
    host[0] = host[1] = 0;
    hipLaunchKernelGGL(pv_clock_probe_kernel, dim3(1), dim3(64), 0, stream, host);
    if (hipStreamSynchronize(stream) != hipSuccess || host[0] == 0) return 0.f;
    if (byMemtime) *byMemtime = (float)((double)host[1] / (double)host[0] * 100.0);
    return (float)(100.0 * 127.0 * 64.0 / (double)host[0] * 100.0);
}

// The box's own streaming bandwidth (SURVEY.md 8d "confirm on the box with a device-to-device copy micro-benchmark and report
// against both"): tools/hbm_calib.hip's legs made callable, for bench.py's roofline record.  1 GiB per buffer (four times the
// 256 MiB Infinity Cache), best of five launches each, HIP events on a stream of its own:
//   out[0] copy, 16 B per lane (bytes read + written per second)     out[1] copy, 4 B per lane
//   out[2] read only, 4 B per lane in 256-byte rows per wave (the stencil kernels' load pattern)     out[3] write only, likewise
namespace {
__global__ __launch_bounds__(256) void pv_bw_copy16_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void pv_bw_copy4_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void pv_bw_read_kernel(const float* __restrict__ src, float* sink, long long nfloat) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), rows = nfloat / 64;
    float acc = 0.f;
    for (long long r = wave * 40; r < rows; r += (long long)gridDim.x * 4 * 40) {
#pragma unroll
        for (int k = 0; k < 40; ++k)
            if (r + k < rows) acc += src[(r + k) * 64 + lane];
    }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void pv_bw_write_kernel(float* dst, long long nfloat) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), rows = nfloat / 64;
    for (long long r = wave * 40; r < rows; r += (long long)gridDim.x * 4 * 40) {
#pragma unroll
        for (int k = 0; k < 40; ++k)
            if (r + k < rows) dst[(r + k) * 64 + lane] = 1.0f;
    }
}
}  // namespace

bool bandwidthProbeGBs(int device, float out[4]) {
    if (hipSetDevice(device) != hipSuccess) return false;
    const long long bytes = 1ll << 30, nfloat = bytes / 4;
    float *a = nullptr, *b = nullptr, *sink = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = hipMalloc((void**)&a, bytes) == hipSuccess && hipMalloc((void**)&b, bytes) == hipSuccess &&
              hipMalloc((void**)&sink, 4) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess &&
              hipMemsetAsync(a, 0, bytes, st) == hipSuccess && hipMemsetAsync(b, 0, bytes, st) == hipSuccess;
    auto best = [&](double moved, auto&& launch) -> float {
        float bestMs = 1e30f;
        for (int i = 0; ok && i < 6; ++i) {  // (the first one warms)
            hipEventRecord(e0, st);
            launch();
            hipEventRecord(e1, st);
            float ms = 0.f;
            ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
            if (ok && i > 0 && ms < bestMs) bestMs = ms;
        }
        return ok ? (float)(moved / bestMs / 1e6) : 0.f;
    };
    if (ok) {
        out[0] = best(2.0 * bytes, [&] { hipLaunchKernelGGL(pv_bw_copy16_kernel, dim3(2048), dim3(256), 0, st, (const float4*)a, (float4*)b, nfloat / 4); });
        out[1] = best(2.0 * bytes, [&] { hipLaunchKernelGGL(pv_bw_copy4_kernel, dim3(2048), dim3(256), 0, st, a, b, nfloat); });
        out[2] = best((double)bytes, [&] { hipLaunchKernelGGL(pv_bw_read_kernel, dim3(2048), dim3(256), 0, st, a, sink, nfloat); });
        out[3] = best((double)bytes, [&] { hipLaunchKernelGGL(pv_bw_write_kernel, dim3(2048), dim3(256), 0, st, b, nfloat); });
    }
    if (st) hipStreamSynchronize(st);
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (st) hipStreamDestroy(st);
    for (float* p : {a, b, sink})
        if (p) hipFree(p);
    return ok;
}

}  // namespace pva
