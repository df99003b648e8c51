// tools/tile_major_probe.hip -- would a TILE-MAJOR layout of the field planes shorten the memory phase of the register-tile
// stencil?  pv_step_merged_kernel<12,36> loads a 60 x 64 tile with its halo from three row-major padded planes: 180 segments
// of 256 B, 16.6 KB apart, and stores its 36 x 40 interior as 108 segments of 160 B, 16.6 KB apart.  The pressure HISTORY went
// tile-major in round 2 for the same reason (tools/hist_write_probe.hip: 5.2 instead of 3.2 TB/s of writes).  With planes laid out
// [tile row][tile column][36][40] a tile's interior is ONE run of 5760 B per plane; its halo comes from the 8 neighbouring tiles
// (rows above / below: runs of 160 B back to back; columns left / right: 48 B of every 160 B).  Same instruction count: the lane
// part of an address is a per-lane constant, the row part is wave-uniform.
// This probe times both layouts on "load tile + halo, `work` rounds of 180 fma, store interior", one wave per tile, two waves
// per SIMD, 11 742 tiles = one 4096^2 sweep, ping-pong between two buffer sets, linear tile order and XCD strips (the solver's
// order 3).   hipcc --offload-arch=gfx950 -O3 tools/tile_major_probe.hip -o /tmp/tmp_probe && /tmp/tmp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int K = 12, RXI = 36, WI = 40, ROWS = RXI + 2 * K, NTX = 114, NTY = 103;
constexpr int G = 16, PITCH = 4160, PROWS = G + NTX * RXI + G + 32;
constexpr long long PLANE_RM = (long long)PROWS * PITCH;
constexpr int TILE = RXI * WI;                                  // floats per tile and plane
constexpr long long PLANE_TM = (long long)(NTX + 2) * (NTY + 2) * TILE;  // one ring of ghost tiles

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

__device__ __forceinline__ bool tileOf(int order, int b, int wave, int* ti, int* tj) {
    if (order == 0) {
        const int t = b * 4 + wave;
        if (t >= NTX * NTY) return false;
        *ti = t / NTY;
        *tj = t - *ti * NTY;
        return true;
    }
    const int xcd = b & 7, q = (b >> 3) * 4 + wave;  // XCD x owns tile columns [x*cw, (x+1)*cw), walks its strip row-major
    const int cw = (NTY + 7) >> 3, c0 = xcd * cw, w = min(cw, NTY - c0);
    if (w <= 0) return false;
    const int r = q / w;
    if (r >= NTX) return false;
    *ti = r;
    *tj = c0 + (q - r * w);
    return true;
}

// LORDER (row-major planes only): 0 = the compiler's schedule of the 180 loads; 1 = rows top to bottom, pinned (a scheduling barrier
// per row); 2 = pinned and ALIGNED between vertical neighbours: a tile shares its first 2K rows with the tile above and its last 2K with
// the tile below, so odd tile rows load their last third first and their first third last -- both sharers of a row then ask for it at
// the same point of their load phases (does the XCD's 4 MiB L2 still have it?  11.8 MB of tiles are in flight per XCD)
template <int LAYOUT, int LORDER = 0>  // 0 = row-major padded planes, 1 = tile-major
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ in, float* __restrict__ out, int order, int work,
                                                float seed) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int ti, tj;
    if (!tileOf(order, blockIdx.x, wave, &ti, &tj)) return;
    float f[3][ROWS];
    constexpr long long PL = LAYOUT == 0 ? PLANE_RM : PLANE_TM;
    rsrc_t rin[3], rout[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        rin[p] = makeRsrc(in + p * PL, PL * 4);
        rout[p] = makeRsrc(out + p * PL, PL * 4);
    }
    if (LAYOUT == 0) {
        const int row0 = G - K + ti * RXI, col0 = G - K + tj * WI;
        const int so0 = (row0 * PITCH + col0) * 4;
        if (LORDER == 2 && (ti & 1)) {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                const int r = k < 2 * K ? ROWS - 2 * K + k : (k < ROWS - 2 * K ? k : k - (ROWS - 2 * K));
#pragma unroll
                for (int p = 0; p < 3; ++p) f[p][r] = bufLoadF(rin[p], lane * 4, so0 + r * PITCH * 4);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
#pragma unroll
                for (int p = 0; p < 3; ++p) f[p][r] = bufLoadF(rin[p], lane * 4, so0 + r * PITCH * 4);
                if (LORDER) __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        const int col = lane - K, dtj = col < 0 ? -1 : (col >= WI ? 1 : 0);
        const int lanePart = (dtj * TILE + (col - WI * dtj)) * 4;  // per-lane constant (may be negative: soffset covers it)
        const int t0 = ((ti + 1) * (NTY + 2) + tj + 1) * TILE;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int rr = r - K, dti = rr < 0 ? -1 : (rr >= RXI ? 1 : 0), rloc = rr - RXI * dti;  // compile-time
            const int so = (t0 + dti * (NTY + 2) * TILE + rloc * WI) * 4 - TILE * 4;  // wave-uniform; lanePart + TILE*4 >= 0
#pragma unroll
            for (int p = 0; p < 3; ++p) f[p][r] = bufLoadF(rin[p], lanePart + TILE * 4, so);
        }
    }
    // like the real tile: nothing is consumed before every load has landed (its non-zero test), and the rows depend on their
    // neighbours (no per-element fusion of load, work and store by the compiler)
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int r = 0; r < ROWS; ++r) s += f[p][r];
    const float c = (__ballot(s == 12345.f) != 0ull) ? 1.f : seed;
    for (int it = 0; it < work; ++it) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) f[p][r] = __builtin_fmaf(f[p][(r + 1) % ROWS], c, f[p][r]);
    }
    if (lane >= K && lane < 64 - K) {
        if (LAYOUT == 0) {
            const int row0 = G - K + ti * RXI, col0 = G - K + tj * WI;
            const int so0 = (row0 * PITCH + col0) * 4;
#pragma unroll
            for (int r = K; r < ROWS - K; ++r)
#pragma unroll
                for (int p = 0; p < 3; ++p) bufStoreF(f[p][r], rout[p], lane * 4, so0 + r * PITCH * 4);
        } else {
            const int t0 = ((ti + 1) * (NTY + 2) + tj + 1) * TILE;
#pragma unroll
            for (int r = K; r < ROWS - K; ++r)
#pragma unroll
                for (int p = 0; p < 3; ++p) bufStoreF(f[p][r], rout[p], (lane - K) * 4, (t0 + (r - K) * WI) * 4);
        }
    }
}

int main(int argc, char** argv) {
    const long long n = (PLANE_RM > PLANE_TM ? PLANE_RM : PLANE_TM) * 3;
    float *a, *b;
    hipMalloc(&a, n * 4);
    hipMalloc(&b, n * 4);
    hipMemset(a, 0, n * 4);
    hipMemset(b, 0, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const dim3 blk(256);
    const int only = argc > 1 ? atoi(argv[1]) : -1;  // >= 0: one variant only, 4 launches (for rocprofv3 --pmc)
    for (int order : {0, 3}) {
        const int blocks = order == 0 ? (NTX * NTY + 3) / 4 : 8 * ((NTX * ((NTY + 7) / 8) + 3) / 4);
        for (int work : {0, 12, 22}) {
            for (int variant : {0, 1, 2, 3}) {  // row-major, tile-major, row-major pinned top-down, row-major pinned + aligned
                if (only >= 0 && (variant != only || order != 3 || work != 22)) continue;
                if (variant >= 2 && order != 3) continue;
                float best = 1e9f;
                for (int i = 0; i < (only >= 0 ? 2 : 8); ++i) {
                    hipEventRecord(e0);
                    for (int rep = 0; rep < 4; ++rep) {  // ping-pong like the solver
                        const float* in = rep & 1 ? b : a;
                        float* out = rep & 1 ? a : b;
                        if (variant == 0)
                            hipLaunchKernelGGL((probe<0, 0>), dim3(blocks), blk, 0, 0, in, out, order, work, 0.f);
                        else if (variant == 1)
                            hipLaunchKernelGGL((probe<1, 0>), dim3(blocks), blk, 0, 0, in, out, order, work, 0.f);
                        else if (variant == 2)
                            hipLaunchKernelGGL((probe<0, 1>), dim3(blocks), blk, 0, 0, in, out, order, work, 0.f);
                        else
                            hipLaunchKernelGGL((probe<0, 2>), dim3(blocks), blk, 0, 0, in, out, order, work, 0.f);
                    }
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (i && ms / 4 < best) best = ms / 4;
                }
                static const char* names[] = {"row-major planes", "tile-major planes", "row-major, loads pinned top-down", "row-major, loads pinned + aligned between vertical neighbours"};
                printf("tile order %d, work %2d rounds (%4d fma per wave), %s: %.1f us per sweep of 11742 tiles\n", order, work, work * 180,
                       names[variant], best * 1e3);
            }
        }
    }
    return 0;
}
