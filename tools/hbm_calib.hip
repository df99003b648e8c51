// tools/hbm_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access pattern the
// stencil kernels use (one dword per lane through a buffer descriptor, rows of 256 B per wave), plus the box's own
// streaming bandwidth (dword and dwordx4 copies).  Known byte counts: each kernel touches exactly `bytes`.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_calib.hip -o gpurun_out/hbm_calib && gpurun_out/hbm_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using rsrc_t = __amdgpu_buffer_rsrc_t;

__global__ void calib_read_dword(const float* src, float* sink, long long nfloat) {
    // each wave reads consecutive 256-byte rows, 40 rows per "tile", like pv_step_air_kernel's load phase
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long rows = nfloat / 64;
    float acc = 0.f;
    for (long long r = wave * 40; r < rows; r += (long long)gridDim.x * 4 * 40) {
#pragma unroll
        for (int k = 0; k < 40; ++k)
            if (r + k < rows) acc += src[(r + k) * 64 + lane];
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void calib_write_dword(float* dst, long long nfloat) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long rows = nfloat / 64;
    for (long long r = wave * 40; r < rows; r += (long long)gridDim.x * 4 * 40) {
#pragma unroll
        for (int k = 0; k < 40; ++k)
            if (r + k < rows) dst[(r + k) * 64 + lane] = 1.0f;
    }
}

__global__ void calib_copy_x4(const float4* src, float4* dst, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

__global__ void calib_copy_dword(const float* src, float* dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

int main() {
    const long long bytes = 1ll << 30;  // > 256 MiB Infinity Cache
    const long long nfloat = bytes / 4;
    float *a, *b, *sink;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMalloc(&sink, 4);
    hipMemset(a, 0, bytes);
    hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](const char* name, double moved, auto&& launch) {
        launch();
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int i = 0; i < 5; ++i) {
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-18s %.3f ms  %.1f GB/s (bytes moved %.0f)\n", name, best, moved / best / 1e6, moved);
    };
    time("read_dword", (double)bytes, [&] { hipLaunchKernelGGL(calib_read_dword, dim3(2048), dim3(256), 0, 0, a, sink, nfloat); });
    time("write_dword", (double)bytes, [&] { hipLaunchKernelGGL(calib_write_dword, dim3(2048), dim3(256), 0, 0, b, nfloat); });
    time("copy_dword", 2.0 * bytes, [&] { hipLaunchKernelGGL(calib_copy_dword, dim3(2048), dim3(256), 0, 0, a, b, nfloat); });
    time("copy_dwordx4", 2.0 * bytes, [&] { hipLaunchKernelGGL(calib_copy_x4, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, nfloat / 4); });
    return 0;
}
