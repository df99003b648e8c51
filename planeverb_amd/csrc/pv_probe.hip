// pv_probe.hip -- the shader clock the chip sustains right now, for bench.py's device record.
// One wave sleeps 100 x s_sleep 127 = 812 800 shader-clock cycles and times that with s_memrealtime (constant 100 MHz): the
// clock of the moment it runs in, whatever else the chip is doing (tools/clock_probe.hip, made callable).  Launched on a
// stream of its own so that it can sit beside the solvers' launches.
#include <hip/hip_runtime.h>

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
__global__ void pv_run_status_kernel(const int* err, int* counts, const unsigned* claims, int* out) {
    if (threadIdx.x == 0) {
        out[0] = *err;
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

void launchRunStatus(const int* err, int* counts, const unsigned* claims, int* outHost, hipStream_t stream) {
    hipLaunchKernelGGL(pv_run_status_kernel, dim3(1), dim3(64), 0, stream, err, counts, claims, outHost);
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

}  // namespace pva
