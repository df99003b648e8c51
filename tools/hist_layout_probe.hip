// tools/hist_layout_probe.hip -- what the LAYOUT of the pressure history costs its readers (round 5).
// Every analysis kernel walks a cell's samples through time; 64 consecutive cells of a tile are one wave.  With the history
// plane-major (plane[t][tile][row][col]) consecutive samples of a wave are a whole plane apart (1.1 MB at 512^2); tile-major in
// time (tile[t][row][col]) they are one tile apart (3.5 - 5.8 KB).  Each wave reads `nt` samples of its 256 bytes, `inflight`
// loads at a time, and adds them up; prints GB/s for both layouts at several numbers of waves (cells 0 .. 64 waves - 1 of the plane).
//   hipcc --offload-arch=gfx950 -O3 tools/hist_layout_probe.hip -o /tmp/hist_layout_probe && /tmp/hist_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int INF>
__global__ __launch_bounds__(256) void walk(const float* __restrict__ h, long long tStride, long long tileStride, int tileCells, int nt, float* out) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    const float* p = h + (g / tileCells) * tileStride + (g % tileCells);  // cell g of the plane: tile g / tileCells, offset g % tileCells
    float acc = 0.f;
    for (int t = 0; t < nt; t += INF) {
        float v[INF];
#pragma unroll
        for (int k = 0; k < INF; ++k) v[k] = p[(long long)(t + k) * tStride];
#pragma unroll
        for (int k = 0; k < INF; ++k) acc += v[k] * v[k];
    }
    if (acc == 12345.f) out[g] = acc;
}

int main() {
    const int T = 3179, tileCells = 880, tiles = 312;          // 512^2 Mode B: 26 x 12 tiles of 20 x 44
    const long long plane = (long long)tiles * tileCells;
    float* h; float* out;
    hipMalloc(&h, (size_t)plane * T * 4);
    hipMemset(h, 0, (size_t)plane * T * 4);
    hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::printf("# waves  inflight | plane-major GB/s | tile-time-major GB/s   (T = %d, %lld cells per plane)\n", T, plane);
    for (int waves : {1024, 2048, 4096}) {
        for (int inf : {8, 32}) {
            float ms[2];
            for (int layout = 0; layout < 2; ++layout) {
                // wave w reads cells [64 w, 64 w + 64): plane-major: base 64 w, stride plane; tile-time-major: the wave's 64 cells
                // sit in tile (64 w / tileCells) whose block is T * tileCells long; approximated by cellBase per wave
                // plane-major: plane[t][tile][cell]; tile-time-major: tile[t][cell] with T samples per tile
                const long long tStride = layout == 0 ? plane : tileCells;
                const long long tileStride = layout == 0 ? tileCells : (long long)T * tileCells;
                const float* base = h;
                auto launch = [&]() {
                    if (inf == 8) hipLaunchKernelGGL(walk<8>, dim3(waves / 4), dim3(256), 0, 0, base, tStride, tileStride, tileCells, T / 32 * 32, out);
                    else hipLaunchKernelGGL(walk<32>, dim3(waves / 4), dim3(256), 0, 0, base, tStride, tileStride, tileCells, T / 32 * 32, out);
                };
                launch(); hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int r = 0; r < 5; ++r) launch();
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[layout], e0, e1);
                ms[layout] /= 5;
            }
            const double bytes = (double)waves * 256.0 * (T / 32 * 32);
            std::printf("%6d  %4d | %8.0f (%.3f ms) | %8.0f (%.3f ms)\n", waves, inf, bytes / ms[0] / 1e6, ms[0], bytes / ms[1] / 1e6, ms[1]);
        }
    }
    return 0;
}
