// tools/tile_load_probe.hip -- does the WIDTH of a register tile's loads matter?  A wave may have 63 vector-memory
// operations in flight (vmcnt is 6 bits).  The air tile of pv_step_merged_kernel<12,36> is 60 rows x 64 lanes x 3 fields =
// 180 buffer_load_dword = three windows of 63, i.e. three memory round trips per tile whatever the bandwidth.  With the
// planes stored in groups of 4 interleaved rows ([row/4][col][4]) the same tile is 45 buffer_load_dwordx4 (and 27
// instead of 108 stores): one window.  This probe runs both forms of "load a 60 x 64 x 3 tile with its halo from three
// padded planes, spend `work` dependent-free VALU rounds on it, store the 36 x 40 x 3 interior", one wave per tile, two
// waves per SIMD, 11 742 tiles = one 4096^2 sweep, and prints the time per sweep.
//   hipcc --offload-arch=gfx950 -O3 tools/tile_load_probe.hip -o /tmp/tlp && /tmp/tlp
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int K = 12, RXI = 36, WI = 40, ROWS = RXI + 2 * K, NTX = 114, NTY = 103;
constexpr int G = 16, PITCH = 4160, PROWS = G + NTX * RXI + G;  // padded plane, rows a multiple of 4

typedef float v4f __attribute__((ext_vector_type(4)));

template <int WIDE>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ in, float* __restrict__ out, long long plane,
                                                int work, float seed) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= NTX * NTY) return;
    const int ti = tile / NTY, tj = tile - ti * NTY;
    const int row0 = G - K + ti * RXI, col0 = G - K + tj * WI;
    float f[3][ROWS];
    if (WIDE == 1) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) f[p][r] = in[p * plane + (long long)(row0 + r) * PITCH + col0 + lane];
    } else {  // rows interleaved in groups of 4: element (row, col) at ((row / 4) * PITCH + col) * 4 + row % 4
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int g = 0; g < ROWS / 4; ++g) {
                const v4f t = *reinterpret_cast<const v4f*>(in + p * plane + ((long long)(row0 / 4 + g) * PITCH + col0 + lane) * 4);
                f[p][4 * g] = t.x;
                f[p][4 * g + 1] = t.y;
                f[p][4 * g + 2] = t.z;
                f[p][4 * g + 3] = t.w;
            }
    }
    // "12 steps": work rounds of 2 VALU ops per value (the real tile: ~4400 packed instructions)
    for (int it = 0; it < work; ++it) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) f[p][r] = f[p][r] * 1.0001f + seed;
    }
    if (lane >= K && lane < 64 - K) {
        if (WIDE == 1) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int r = K; r < ROWS - K; ++r) out[p * plane + (long long)(row0 + r) * PITCH + col0 + lane] = f[p][r];
        } else {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int g = K / 4; g < (ROWS - K) / 4; ++g)
                    *reinterpret_cast<v4f*>(out + p * plane + ((long long)(row0 / 4 + g) * PITCH + col0 + lane) * 4) =
                        v4f{f[p][4 * g], f[p][4 * g + 1], f[p][4 * g + 2], f[p][4 * g + 3]};
        }
    }
}

int main() {
    const long long plane = (long long)PROWS * PITCH;
    float *a, *b;
    hipMalloc(&a, plane * 4 * 3);
    hipMalloc(&b, plane * 4 * 3);
    hipMemset(a, 0, plane * 4 * 3);
    hipMemset(b, 0, plane * 4 * 3);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const dim3 grid((NTX * NTY + 3) / 4), blk(256);
    for (int work : {0, 6, 12, 18}) {
        for (int wide : {1, 4}) {
            float best = 1e9f;
            for (int i = 0; i < 8; ++i) {
                hipEventRecord(e0);
                for (int rep = 0; rep < 4; ++rep) {  // ping-pong like the solver
                    if (wide == 1)
                        hipLaunchKernelGGL(probe<1>, grid, blk, 0, 0, rep & 1 ? b : a, rep & 1 ? a : b, plane, work, 0.f);
                    else
                        hipLaunchKernelGGL(probe<4>, grid, blk, 0, 0, rep & 1 ? b : a, rep & 1 ? a : b, plane, work, 0.f);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (i && ms / 4 < best) best = ms / 4;
            }
            printf("work %2d rounds (%5d VALU ops per wave), %s: %.1f us per sweep of 11742 tiles\n", work, work * 360,
                   wide == 1 ? "180 dword loads + 108 dword stores  " : " 45 dwordx4 loads + 27 dwordx4 stores", best * 1e3);
        }
    }
    return 0;
}
