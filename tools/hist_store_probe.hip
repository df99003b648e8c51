// tools/hist_store_probe.hip -- why does a RECORDING tile cost 3.5x a silent one (sparse-emitter / dense-history modes)?
// A wave may have 63 vector-memory operations in flight (vmcnt is 6 bits); a recording tile issues 36 dword stores per
// sub-step between ~1500 cycles of arithmetic.  Same bytes as 18 dwordx2 stores (the mirror-pair registers of the air
// tile ARE (row i, row 59-i) pairs: a pair-interleaved history layout [pair][col][2] would store them directly) or 9
// dwordx4 stores.  Each wave: 12 x { `work` rounds of VALU ; stores of a 36 x 40 block, W floats per lane and store },
// one wave per tile, two waves per SIMD, tile-major destination, 12 planes of 67.6 MB (the real launch's history traffic).
//   hipcc --offload-arch=gfx950 -O3 tools/hist_store_probe.hip -o /tmp/hsp && /tmp/hsp
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int RXI = 36, WI = 40, K = 12;

template <int W>  // floats per store instruction and lane; W = 0: no stores
__global__ __launch_bounds__(256, 2) void probe(float* hist, long long plane, int ntiles, int work, float seed) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    float v[3 * RXI];  // a register tile's worth of state, so that two waves per SIMD is what fits
#pragma unroll
    for (int r = 0; r < 3 * RXI; ++r) v[r] = seed + r + lane;
    for (int s = 0; s < K; ++s) {
        for (int it = 0; it < work; ++it) {
#pragma unroll
            for (int r = 0; r < 3 * RXI; ++r) v[r] = v[r] * 1.0001f + seed;
        }
        if (lane < WI) {
            float* p = hist + (long long)s * plane + (long long)tile * RXI * WI;
            if (W == 1) {
#pragma unroll
                for (int r = 0; r < RXI; ++r) p[r * WI + lane] = v[r];
            } else if (W == 2) {
#pragma unroll
                for (int r = 0; r < RXI; r += 2) *reinterpret_cast<float2*>(p + (r / 2 * WI + lane) * 2) = make_float2(v[r], v[r + 1]);
            } else if (W == 4) {
#pragma unroll
                for (int r = 0; r < RXI; r += 4)
                    *reinterpret_cast<float4*>(p + (r / 4 * WI + lane) * 4) = make_float4(v[r], v[r + 1], v[r + 2], v[r + 3]);
            }
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 3 * RXI; ++r) acc += v[r];
    if (acc == 12345.678f) hist[0] = acc;
}

int main() {
    const int ntiles = 11742;
    const long long plane = (long long)ntiles * RXI * WI;
    float* h;
    if (hipMalloc(&h, plane * 4 * K) != hipSuccess) return 1;
    (void)hipMemset(h, 0, plane * 4 * K);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int work : {0, 4, 8}) {
        for (int w : {0, 1, 2, 4}) {
            float best = 1e9f;
            for (int i = 0; i < 5; ++i) {
                (void)hipEventRecord(e0);
                const dim3 g((ntiles + 3) / 4), b(256);
                if (w == 0) hipLaunchKernelGGL(probe<0>, g, b, 0, 0, h, plane, ntiles, work, 1.f);
                if (w == 1) hipLaunchKernelGGL(probe<1>, g, b, 0, 0, h, plane, ntiles, work, 1.f);
                if (w == 2) hipLaunchKernelGGL(probe<2>, g, b, 0, 0, h, plane, ntiles, work, 1.f);
                if (w == 4) hipLaunchKernelGGL(probe<4>, g, b, 0, 0, h, plane, ntiles, work, 1.f);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (i && ms < best) best = ms;
            }
            printf("work %d rounds (%4d VALU ops per sub-step), %s: %.1f us per launch (%.0f MB of history)\n", work, work * 108,
                   w == 0 ? "no stores               " : w == 1 ? "36 dword stores / step  " : w == 2 ? "18 dwordx2 stores / step" : " 9 dwordx4 stores / step",
                   best * 1e3, w ? plane * 4.0 * K / 1e6 : 0.0);
        }
    }
    return 0;
}
