// tools/hist_write_probe.hip -- which layout should the pressure history have?  The step kernels record, per sub-step and
// tile, RXI rows of WI = 40 floats (160 B).  Pattern A (the layout in use): plane[t][row][col] -- the 36 rows of a tile sit
// histPitch * 4 = 16 KB apart, the 12 sub-steps of a launch 69 MB apart.  Pattern B (tile-major): plane[t][tile][row][40] --
// a tile's 36 x 160 B of one sub-step are one contiguous 5.76 KB block.  Same bytes, same number of store instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/hist_write_probe.hip -o /tmp/hwp && /tmp/hwp
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int RXI = 36, WI = 40, K = 12;

// one wave per tile, K sub-steps, RXI row stores of WI active lanes each
template <bool TILE_MAJOR>
__global__ void write_hist(float* hist, long long plane, int pitch, int ntx, int nty) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntx * nty) return;
    const int ti = tile / nty, tj = tile - ti * nty;
    if (lane >= WI) return;
    for (int s = 0; s < K; ++s) {
        float* p = hist + (long long)s * plane;
#pragma unroll 4
        for (int r = 0; r < RXI; ++r) {
            if (TILE_MAJOR)
                p[((long long)tile * RXI + r) * WI + lane] = (float)(s + r);
            else
                p[(long long)(ti * RXI + r) * pitch + tj * WI + lane] = (float)(s + r);
        }
    }
}

int main() {
    const int ntx = 114, nty = 103, pitch = 4160;
    const long long plane = (long long)(ntx * RXI) * pitch;  // floats
    float* h;
    hipMalloc(&h, plane * 4 * K);
    hipMemset(h, 0, plane * 4 * K);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double bytes = (double)ntx * nty * RXI * WI * 4 * K;
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int i = 0; i < 6; ++i) {
            hipEventRecord(e0);
            if (mode)
                hipLaunchKernelGGL(write_hist<true>, dim3((ntx * nty + 3) / 4), dim3(256), 0, 0, h, plane, pitch, ntx, nty);
            else
                hipLaunchKernelGGL(write_hist<false>, dim3((ntx * nty + 3) / 4), dim3(256), 0, 0, h, plane, pitch, ntx, nty);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (i && ms < best) best = ms;
        }
        printf("%s: %.3f ms for %.0f MB = %.1f GB/s\n", mode ? "tile-major [t][tile][row][40]" : "row-major  [t][row][col]    ", best,
               bytes / 1e6, bytes / best / 1e6);
    }
    return 0;
}
