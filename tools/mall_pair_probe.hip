// tools/mall_pair_probe.hip -- a second level of temporal blocking through the 256 MiB Infinity Cache: what would it buy?
// pv_step_merged_kernel<12,36> is at 80 % of the floor its HBM-side bytes set (tools/tile_major_probe.hip).  Idea: advance the
// grid TWO sweeps (2 x 12 steps) per pass over HBM.  The grid is cut into B bands of tile rows; for every band
//     launch 1:  tile rows [r0 - 1, r1 + 1)  set A -> scratch S      (one tile row = 36 rows >= 2K more on either side)
//     launch 2:  tile rows [r0, r1)          scratch S -> set C
// S is as large as one extended band and is reused by every band, so it lives in the Infinity Cache: HBM sees A read once and C
// written once per TWO sweeps.  Price: 2 / rows-per-band more tile work on every first launch, 2B launches per pair of sweeps
// instead of 2, each a fraction of the 2048 resident waves' worth of tiles.
// This probe times exactly that launch structure with the tile kernel's memory pattern and instruction count (see
// tile_major_probe.hip: load 60 x 64 x 3, wait for all, `work` rounds of 180 fma, store 36 x 40 x 3), one "run" per stream, one
// or two streams in flight, against the plain two launches per pair of sweeps.
//   hipcc --offload-arch=gfx950 -O3 tools/mall_pair_probe.hip -o /tmp/mpp && /tmp/mpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int K = 12, RXI = 36, WI = 40, ROWS = RXI + 2 * K, NTX = 114, NTY = 103;
constexpr int G = 16, PITCH = 4160, PROWS = G + NTX * RXI + G + 32;
constexpr long long PLANE = (long long)PROWS * PITCH;

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t makeRsrc(const void* p, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bufLoadF(rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bufStoreF(float v, rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

// tile rows [t0, t0 + nr) of the grid; in / out: plane 0 of three planes `inPlane` / `outPlane` floats apart, addressed as if they
// were whole planes (the scratch is passed with its base moved up by the band's first row)
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ in, long long inPlane, float* __restrict__ out,
                                                long long outPlane, int t0, int nr, int work, float seed) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x, xcd = b & 7, q = (b >> 3) * 4 + wave;  // XCD x owns tile columns [x*cw, (x+1)*cw): row-major in its strip
    const int cw = (NTY + 7) >> 3, c0 = xcd * cw, w = min(cw, NTY - c0);
    if (w <= 0) return;
    const int r = q / w;
    if (r >= nr) return;
    const int ti = t0 + r, tj = c0 + (q - r * w);
    float f[3][ROWS];
    rsrc_t rin[3], rout[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        rin[p] = makeRsrc(in + p * inPlane, PLANE * 4);
        rout[p] = makeRsrc(out + p * outPlane, PLANE * 4);
    }
    const int row0 = G - K + ti * RXI, col0 = G - K + tj * WI;
    const int so0 = (row0 * PITCH + col0) * 4;
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr)
#pragma unroll
        for (int p = 0; p < 3; ++p) f[p][rr] = bufLoadF(rin[p], lane * 4, so0 + rr * PITCH * 4);
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) s += f[p][rr];
    const float c = (__ballot(s == 12345.f) != 0ull) ? 1.f : seed;
    for (int it = 0; it < work; ++it) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) f[p][rr] = __builtin_fmaf(f[p][(rr + 1) % ROWS], c, f[p][rr]);
    }
    if (lane >= K && lane < 64 - K) {
#pragma unroll
        for (int rr = K; rr < ROWS - K; ++rr)
#pragma unroll
            for (int p = 0; p < 3; ++p) bufStoreF(f[p][rr], rout[p], lane * 4, so0 + rr * PITCH * 4);
    }
}

static int blocksFor(int nr) { return 8 * ((nr * ((NTY + 7) / 8) + 3) / 4); }

struct Run {
    float *a, *c, *s;
    long long sPlane;
    hipStream_t st;
};

// one pair of sweeps of one run
static void pairPlain(const Run& r, int work) {
    hipLaunchKernelGGL(probe, dim3(blocksFor(NTX)), dim3(256), 0, r.st, r.a, PLANE, r.c, PLANE, 0, NTX, work, 0.f);
    hipLaunchKernelGGL(probe, dim3(blocksFor(NTX)), dim3(256), 0, r.st, r.c, PLANE, r.a, PLANE, 0, NTX, work, 0.f);
}
static void pairBanded(const Run& r, int B, int work, bool flip) {
    const float* src = flip ? r.c : r.a;
    float* dst = flip ? r.a : r.c;
    for (int b = 0; b < B; ++b) {
        const int r0 = NTX * b / B, r1 = NTX * (b + 1) / B;
        const int e0 = r0 > 0 ? r0 - 1 : 0, e1 = r1 < NTX ? r1 + 1 : NTX;
        float* sb = r.s - (long long)e0 * RXI * PITCH;  // tile row e0 lands on the scratch's first tile row
        hipLaunchKernelGGL(probe, dim3(blocksFor(e1 - e0)), dim3(256), 0, r.st, src, PLANE, sb, r.sPlane, e0, e1 - e0, work, 0.f);
        hipLaunchKernelGGL(probe, dim3(blocksFor(r1 - r0)), dim3(256), 0, r.st, (const float*)sb, r.sPlane, dst, PLANE, r0, r1 - r0, work, 0.f);
    }
}

int main() {
    const int maxBandRows = NTX / 2 + 3;
    const long long sPlane = (long long)(G + maxBandRows * RXI + G + 32) * PITCH;
    Run runs[2];
    for (auto& r : runs) {
        hipMalloc(&r.a, PLANE * 12);
        hipMalloc(&r.c, PLANE * 12);
        hipMalloc(&r.s, sPlane * 12);
        hipMemset(r.a, 0, PLANE * 12);
        hipMemset(r.c, 0, PLANE * 12);
        hipMemset(r.s, 0, sPlane * 12);
        r.sPlane = sPlane;
        hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int work : {0, 22}) {
        for (int nruns : {1, 2}) {
            for (int B : {0, 2, 4, 8}) {
                float best = 1e9f;
                const int pairs = 4;
                for (int i = 0; i < 6; ++i) {
                    hipDeviceSynchronize();
                    hipEventRecord(e0, runs[0].st);
                    for (int p = 0; p < pairs; ++p)
                        for (int k = 0; k < nruns; ++k) {
                            if (B == 0)
                                pairPlain(runs[k], work);
                            else
                                pairBanded(runs[k], B, work, p & 1);
                        }
                    hipDeviceSynchronize();
                    hipEventRecord(e1, runs[0].st);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    const float per = ms / (pairs * 2 * nruns);
                    if (i && per < best) best = per;
                }
                printf("work %2d, %d run(s) in flight, %s: %.1f us per sweep and run\n", work, nruns,
                       B == 0 ? "plain (2 launches per pair of sweeps)" : (B == 2 ? "2 bands x 2 launches through the scratch  " : (B == 4 ? "4 bands x 2 launches through the scratch  " : "8 bands x 2 launches through the scratch  ")), best * 1e3);
            }
        }
    }
    return 0;
}
