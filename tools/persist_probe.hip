// tools/persist_probe.hip -- would overlapping the sweeps of ONE run pay?  (VERDICT r04 item 4 / DESIGN.md 8, "probe first")
//
// One run in flight is 37 dependent launches of pv_step_merged_kernel<12,36> at 4096^2: every launch ends in a tail (11 742 tiles on
// 2048 wave slots = 5.7 rounds, the last one part full), a ~2 us gap and a ramp; with TWO runs in flight the other run fills those
// holes (1.50e12 -> 1.78e12 cell-updates/s, profiles/r05_bench_inflight1.json / r05_bench.json).  The alternative for one run: a
// PERSISTENT launch whose waves draw (sweep, tile) items from per-XCD ticket counters in sweep-major order; an item of sweep n + 1
// may start when its own tile and its 8 neighbours have finished sweep n (they read what it overwrites and wrote what it reads).
// Tickets are handed out in order, so whoever waits, waits for items that running waves hold: no deadlock, no co-residency assumption.
//
// Timing only (results are not checked; every variant moves the same bytes and issues the same arithmetic): the tile body has the
// real kernel's structure -- 180 loads of 256 B pinned top to bottom, `work` rounds of 180 dependent FMAs, 108 stores of 160 B, one
// wave per tile, two waves per SIMD, XCD strips of tile columns, ping-pong between two buffer sets.
//   A   per-launch form, one stream, 37 launches back to back                       (= one run in flight)
//   A2  per-launch form, two independent chains on two streams                      (= two runs in flight; per-sweep time = total / 74)
//   B1  persistent, tickets only (no dependency waits, plain loads / stores)        (what the scheduling alone buys)
//   B2  persistent, + per-tile "sweeps done" words: wait for 9, s_waitcnt vmcnt(0) + agent-scope increment after the stores
//   B3  B2 + strip-boundary tiles load and store with sc1 (write-through / L2-bypassing loads: what crosses XCDs inside a launch)
//   C1  persistent, STATIC assignment (wave v of an XCD takes items v, v + 256, ...: no ticket counter), no waits
//   C3  C1 + dependency words + sc1 on strip-boundary tiles
//   hipcc --offload-arch=gfx950 -O3 tools/persist_probe.hip -o /tmp/persist_probe && /tmp/persist_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int K = 12, RXI = 36, WI = 40, ROWS = RXI + 2 * K, NTX = 114, NTY = 103;
constexpr int G = 16, PITCH = 4160, PROWS = G + NTX * RXI + G + 32;
constexpr long long PLANE = (long long)PROWS * PITCH;
constexpr int CW = (NTY + 7) / 8;  // tile columns per XCD strip
constexpr int SWEEPS = 37;

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t makeRsrc(const void* p, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ float bufLoadF(rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX));
}
template <int AUX>
__device__ __forceinline__ void bufStoreF(float v, rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, AUX);
}
constexpr int kSc1 = 2;  // aux bit 1 = sc1 on gfx940+: agent scope

// tile q of XCD strip `xcd` (row-major walk of the strip)
__device__ __forceinline__ bool stripTile(int xcd, int q, int* ti, int* tj) {
    const int c0 = xcd * CW, w = min(CW, NTY - c0);
    if (w <= 0) return false;
    const int r = q / w;
    if (r >= NTX) return false;
    *ti = r;
    *tj = c0 + (q - r * w);
    return true;
}
__host__ __device__ constexpr int stripTiles(int xcd) {
    const int c0 = xcd * CW, w = (CW < NTY - c0 ? CW : NTY - c0);
    return w > 0 ? w * NTX : 0;
}

template <int AUX>
__device__ __forceinline__ void tileBody(const float* __restrict__ in, float* __restrict__ out, int ti, int tj, int lane, int work,
                                         float seed) {
    float f[3][ROWS];
    rsrc_t rin[3], rout[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        rin[p] = makeRsrc(in + p * PLANE, PLANE * 4);
        rout[p] = makeRsrc(out + p * PLANE, PLANE * 4);
    }
    const int row0 = G - K + ti * RXI, col0 = G - K + tj * WI;
    const int so0 = (row0 * PITCH + col0) * 4;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
#pragma unroll
        for (int p = 0; p < 3; ++p) f[p][r] = bufLoadF<AUX>(rin[p], lane * 4, so0 + r * PITCH * 4);
        __builtin_amdgcn_sched_barrier(0);
    }
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
#pragma unroll
        for (int r = K; r < ROWS - K; ++r)
#pragma unroll
            for (int p = 0; p < 3; ++p) bufStoreF<AUX>(f[p][r], rout[p], lane * 4, so0 + r * PITCH * 4);
    }
}

// per-launch form: block b -> XCD b % 8, four tiles per block
__global__ __launch_bounds__(256, 2) void sweepKernel(const float* __restrict__ in, float* __restrict__ out, int work, float seed) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int ti, tj;
    if (!stripTile(blockIdx.x & 7, (blockIdx.x >> 3) * 4 + wave, &ti, &tj)) return;
    tileBody<0>(in, out, ti, tj, lane, work, seed);
}

struct Persist {
    float* buf[2];
    unsigned* tickets;  // [8]
    unsigned* done;     // [(NTX + 2) * (NTY + 2)]: sweeps finished by the tile, ring of ghost tiles pre-set to "all"
    int work, sweeps, mode;  // mode 1: tickets only, 2: + dependency words, 3: + sc1 on strip-boundary tiles
    float seed;
};

__device__ __forceinline__ unsigned loadWordAgent(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256, 2) void persistKernel(const Persist a) {
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7;  // (workgroups are dealt to the XCDs round-robin; all of them are resident)
    const int per = stripTiles(xcd);
    const unsigned total = (unsigned)per * (unsigned)a.sweeps;
    // static assignment (mode & 8): wave v of the XCD's 256 takes items v, v + 256, ... -- no ticket counter (every wave is resident)
    const unsigned wavesPerXcd = (gridDim.x >> 3) * 4u;
    unsigned mine = (blockIdx.x >> 3) * 4u + (threadIdx.x >> 6);
    while (true) {
        unsigned t = 0;
        if (a.mode & 8) {
            t = mine;
            mine += wavesPerXcd;
        } else {
            if (lane == 0) t = atomicAdd(a.tickets + xcd, 1u);
        }
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= total) break;
        const int sweep = (int)(t / (unsigned)per), q = (int)(t - (unsigned)sweep * (unsigned)per);
        int ti, tj;
        stripTile(xcd, q, &ti, &tj);
        const unsigned* dn = a.done + (ti + 1) * (NTY + 2) + (tj + 1);
        if ((a.mode & 7) >= 2 && sweep > 0) {
            // lanes 0..8: the tile itself and its 8 neighbours must have finished sweep - 1
            const int dr = lane / 3 - 1, dc = lane % 3 - 1;
            const unsigned* w = dn + dr * (NTY + 2) + dc;
            while (true) {
                const bool ok = lane >= 9 || loadWordAgent(w) >= (unsigned)sweep;
                if (__ballot(!ok) == 0ull) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        const float* in = a.buf[sweep & 1];
        float* out = a.buf[(sweep & 1) ^ 1];
        const int c0 = xcd * CW;
        const bool edge = (a.mode & 7) >= 3 && (tj == c0 || tj == min(c0 + CW, NTY) - 1);
        if (edge)
            tileBody<kSc1>(in, out, ti, tj, lane, a.work, a.seed);
        else
            tileBody<0>(in, out, ti, tj, lane, a.work, a.seed);
        if ((a.mode & 7) >= 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(const_cast<unsigned*>(dn), (unsigned)sweep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char** argv) {
    const long long n = PLANE * 3;
    float *a, *b, *c, *d;
    hipMalloc(&a, n * 4);
    hipMalloc(&b, n * 4);
    hipMalloc(&c, n * 4);
    hipMalloc(&d, n * 4);
    for (float* p : {a, b, c, d}) hipMemset(p, 0, n * 4);
    unsigned *tickets, *done;
    hipMalloc(&tickets, 8 * 4);
    hipMalloc(&done, (NTX + 2) * (NTY + 2) * 4);
    std::vector<unsigned> done0((NTX + 2) * (NTY + 2), 0xffffffffu);
    for (int i = 0; i < NTX; ++i)
        for (int j = 0; j < NTY; ++j) done0[(size_t)(i + 1) * (NTY + 2) + j + 1] = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    int dev = 0, cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int blocks = 8 * ((NTX * CW + 3) / 4);
    const int pblocks = cus * 2;  // two 4-wave workgroups per CU = two waves per SIMD
    printf("# 4096^2 (114 x 103 tiles of 36 x 40 cells, K = 12), %d sweeps, %d CUs; us per sweep, best of 6\n", SWEEPS, cus);
    for (int work : {22, 30, 36}) {
        float res[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int variant = 0; variant < 7; ++variant) {
            float best = 1e9f;
            for (int i = 0; i < 7; ++i) {
                if (variant >= 2) {
                    hipMemsetAsync(tickets, 0, 32, s1);
                    hipMemcpyAsync(done, done0.data(), done0.size() * 4, hipMemcpyHostToDevice, s1);
                    hipStreamSynchronize(s1);
                }
                hipEventRecord(e0, s1);
                if (variant == 0) {
                    for (int sw = 0; sw < SWEEPS; ++sw)
                        hipLaunchKernelGGL(sweepKernel, dim3(blocks), dim3(256), 0, s1, sw & 1 ? b : a, sw & 1 ? a : b, work, 0.f);
                } else if (variant == 1) {
                    hipStreamWaitEvent(s2, e0, 0);
                    for (int sw = 0; sw < SWEEPS; ++sw) {
                        hipLaunchKernelGGL(sweepKernel, dim3(blocks), dim3(256), 0, s1, sw & 1 ? b : a, sw & 1 ? a : b, work, 0.f);
                        hipLaunchKernelGGL(sweepKernel, dim3(blocks), dim3(256), 0, s2, sw & 1 ? d : c, sw & 1 ? c : d, work, 0.f);
                    }
                    hipEventRecord(e1, s2);
                    hipStreamWaitEvent(s1, e1, 0);
                } else {
                    Persist p{{a, b}, tickets, done, work, SWEEPS, variant <= 4 ? variant - 1 : (variant == 5 ? 8 + 1 : 8 + 3), 0.f};
                    hipLaunchKernelGGL(persistKernel, dim3(pblocks), dim3(256), 0, s1, p);
                }
                hipEventRecord(e1, s1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const float per = ms * 1e3f / (variant == 1 ? 2 * SWEEPS : SWEEPS);
                if (i && per < best) best = per;
            }
            res[variant] = best;
        }
        printf("work %2d rounds (%4d fma per wave): A one chain %.1f | A2 two chains %.1f | B1 persistent, tickets only %.1f | "
               "B2 + dependency words %.1f | B3 + sc1 on strip-boundary tiles %.1f | C1 static assignment, no waits %.1f | C3 static + words + sc1 %.1f"
               "   (A2/A %.3f, B3/A %.3f, C3/A %.3f)\n",
               work, work * 180, res[0], res[1], res[2], res[3], res[4], res[5], res[6], res[1] / res[0], res[4] / res[0], res[6] / res[0]);
    }
    if (hipGetLastError() != hipSuccess) printf("HIP error\n");
    return 0;
}
