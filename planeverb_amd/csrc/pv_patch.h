// pv_patch.h -- persistent per-CU form of the air-tile stencil with LDS-DMA run-ahead (included by pv_kernels.hip).
//
// Why (DESIGN.md 4.1 "where the time goes"): the one-wave-per-tile kernel spends 131 us per 12-step
// sweep of a 4096^2 grid although its arithmetic alone takes 92 us and its memory pattern alone 79 us -- a wave holds its
// whole 60 x 64 tile in 250 VGPRs, so it can only load OR compute, and the CU's 160 KiB of LDS sit idle.  Here the LDS is
// the landing zone of the NEXT tile while the current one computes:
//
//   * one 512-thread workgroup per CU, resident for the whole launch: 8 waves = two GROUPS of 4 (waves w and w + 4 share
//     a SIMD).  A group advances a PATCH of 4 horizontally adjacent tiles (one wave per tile), whose loaded regions overlap
//     by 2K columns: 60 rows x 184 columns per plane instead of 4 x (60 x 64) -- 28 % fewer bytes into the CU.
//   * a patch's three planes (138 KB with the 192-float row pitch) arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`,
//     1 KiB per wave-instruction, no VGPR destinations, one vmcnt window): 135 instructions per patch, ~34 per wave,
//     instead of the 181 dword loads per wave of the tile kernel (three vmcnt windows).
//   * ONE zone, time-shared by the two groups half a tile period apart (the whole-workgroup s_barrier is the clock):
//         half h      group X                                   group Y
//         ---------   ---------------------------------------   ---------------------------------------------
//         barrier 1   (its DMA has landed: vmcnt waited)
//                     ds_read its tile zone -> registers         advances step K/2 of its current tile
//         barrier 2   (zone free)
//                     advances steps 0 .. K/2-1                  issues the DMA of ITS next patch into the zone,
//                                                                advances steps K/2+1 .. K-1, waits vmcnt, stores
//         half h + 1  roles swapped
//     so each wave computes almost all the time, its next tile lands while it computes, and at any time the two waves of a
//     SIMD are in different phases (one of them may be reading LDS / storing while the other has the VALU).
//   * the tile arithmetic is the mirror-pair packed form of pv_kernels.hip, unchanged (same bits).
// General tiles (walls, grid edges, listener) keep their 4-wave blocks in a launch of their own (launchStep which = 16).
#pragma once

namespace pva {

template <int K, int RXI>
struct PatchGeom {
    static constexpr int ROWS = RXI + 2 * K;
    static constexpr int NP = ROWS / 2;
    static constexpr int WI = 64 - 2 * K;
    static constexpr int TPP = 4;                      // tiles per patch (= waves per group)
    static constexpr int COLS = (TPP - 1) * WI + 64;   // loaded columns of a patch
    static constexpr int PW = 192;                     // zone row pitch in floats: 4 rows = 3 DMA instructions of 1 KiB
    static constexpr int PLANE = ROWS * PW;            // floats per plane of the zone
    static constexpr int ZONE = 3 * PLANE;
    static constexpr int RB = ROWS / 4;                // 4-row blocks per plane
    static_assert(COLS <= PW, "patch wider than the zone rows");
    static_assert(ROWS % 4 == 0, "zone rows are filled in blocks of 4");
    static_assert(ROWS % 2 == 0, "mirror pairs");
    static_assert(K % 2 == 0, "the time loop is cut in two halves");
};

// one LDS-DMA instruction: 64 lanes x 16 B from buffer `r` at voff(lane) + soff -> LDS bytes [lds, lds + 1024), lane-linear
// (measured, tools/lds_dma_probe.hip: the dwordx4 form writes 16 B per lane densely and needs only 4-byte aligned sources;
// the dwordx3 form also strides the LDS by 16 B per lane, leaving 4-byte gaps; out-of-range lanes write zeros; LDS offsets
// beyond 64 KiB work).  A zone row is 768 B = 48 lanes' worth, so 3 instructions fill 4 rows: lane l of instruction j is
// 16-byte unit u = 64 j + l of the block, row u / 48, column (u % 48) * 4.
__device__ __forceinline__ void patchDma16(rsrc_t r, int voff, int soff, unsigned lds) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(r), "s"(soff), "s"(lds)
        : "memory");
}

// Per-tile bookkeeping reads (tile class, run parameters, first-active step) as SCALAR loads: written as plain loads they
// become vector loads followed by `s_waitcnt vmcnt(0)` -- and in this kernel a wave reaches them right after issuing the
// 108 result stores of its previous tile, so each of those waits drained the whole store queue (the first build: 169 us
// per sweep instead of the tile kernel's 139).  Constant-address-space loads with a uniform address are selected as
// s_load_* (lgkmcnt).  What they read is written by earlier launches only -- or, tileFirst, by this wave itself -- and
// the scalar cache is invalidated at every kernel start.
template <typename T>
__device__ __forceinline__ T patchSLoad(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "scalar loads are dword loads");
    struct Words {
        uint32_t w[sizeof(T) / 4];
    } v;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i)
        v.w[i] = *reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(a + 4 * i);
    return __builtin_bit_cast(T, v);
}
__device__ __forceinline__ uint32_t patchSLoadByte(const uint8_t* p) {  // (no sub-dword scalar loads on gfx9: containing dword)
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint32_t w = patchSLoad(reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3));
    return (w >> (8 * (uint32_t)(a & 3))) & 0xffu;
}

// steps [S, E) of the unrolled trapezoid (MirrorSteps of pv_kernels.hip, cut into ranges)
template <int K, int RXI, int S, int E>
struct PatchSteps {
    static constexpr int ROWS = RXI + 2 * K;
    static constexpr int NP = ROWS / 2;
    static __device__ __forceinline__ void run(v2f (&pr)[NP], v2f (&vx)[NP], v2f (&vy)[NP], float& vxS, const float C,
                                               const int nsteps, const bool recLane, const float* hplane0,
                                               const long long hstride, const int hvoff, const int hsoff0,
                                               const int hpitchB) {
        if constexpr (S < E) {
            if (S < nsteps) {
                leapfrogStepMirror<NP, PV_MIRROR_G, S>(pr, vx, vy, vxS, C);
                if (recLane) {  // pressure of this step, interior rows (air tiles never hold the listener)
                    const rsrc_t rH = makeRsrc(hplane0 + (long long)S * hstride, hstride * 4);
#pragma unroll
                    for (int r = K; r < ROWS - K; ++r)
                        bufStoreF(r < NP ? pr[r].x : pr[ROWS - 1 - r].y, rH, hvoff, hsoff0 + r * hpitchB);
                }
                __builtin_amdgcn_sched_barrier(0);
                PatchSteps<K, RXI, S + 1, E>::run(pr, vx, vy, vxS, C, nsteps, recLane, hplane0, hstride, hvoff, hsoff0, hpitchB);
            }
        }
    }
};

template <int K, int RXI>
__global__ __launch_bounds__(512, 2) void pv_step_patch_kernel(const StepArgs a) {
    using Gm = PatchGeom<K, RXI>;
    constexpr int ROWS = Gm::ROWS, NP = Gm::NP, WI = Gm::WI, PW = Gm::PW, PLANE = Gm::PLANE;
    constexpr int SLOTG = ROWS / 12;  // the ring of row slots advances by RXI = 36 rows per patch: everything moves in 12-row groups
    static_assert(ROWS % 12 == 0 && RXI % 12 == 0 && (2 * K) % 12 == 0, "ring arithmetic in groups of 12 rows");
    __shared__ __attribute__((aligned(1024))) float zone[Gm::ZONE];
    __shared__ int counters[2];  // [0] DMA shares landed, [1] tile reads done: 4 per patch each (one per wave of its group)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = wave >> 2, w = wave & 3;
    if (threadIdx.x == 0) counters[0] = counters[1] = 0;
    __syncthreads();

    // this block's share of the patches: a contiguous range of the COLUMN-major sequence (patch c = pj * ntx + ti), so that
    // consecutive patches are vertically adjacent; blocks of one XCD (block % 8, observed) take neighbouring ranges
    const int npy = (a.nty + Gm::TPP - 1) / Gm::TPP;
    const int npatch = a.ntx * npy;
    const int nb = (int)gridDim.x;
    const int q = (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3);
    const int c0 = (int)((long long)npatch * q / nb), c1 = (int)((long long)npatch * (q + 1) / nb);
    const int n = c1 - c0;  // patch m of the block (0 <= m < n) belongs to group m & 1 and sits in the zone after patch m - 1

    const int pitchB = a.pitch * 4;
    // measurement aid (tools/patch_decomp.sh): bits 8.. of patchStrip switch parts of the kernel off -- 1 steps, 2 DMA, 4 result
    // stores, 8 zone reads.  0 in every real run.
    const int dbg = a.patchStrip >> 8;
    // per-lane offsets are re-made from the lane index where they are used (opaque copies, so that they are not hoisted):
    // every value kept live across the unrolled steps is a VGPR the 180-register tile does not have
    auto laneCopy = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };
    const unsigned zoneB = (unsigned)(size_t)zone;
    // pr, vx, vy of a buffer set are ONE allocation (pv_solver.cpp): one descriptor reaches all three, the plane offset
    // rides in the scalar offset (4 SGPRs instead of 12; the host checks 3 * planeBytes <= INT_MAX)
    const int planeB = (int)a.planeBytes;
    const rsrc_t rIn = makeRsrc(a.prIn, a.inBytes ? 3 * planeB : 0);

    volatile int* const cnt = counters;
    auto bump = [&](int which) {  // one count per wave
        if (lane == 0) __hip_atomic_fetch_add(&counters[which], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto reached = [&](int which, int target) { return __builtin_amdgcn_readfirstlane(cnt[which]) >= target; };
    auto waitFor = [&](int which, int target) {
        while (!reached(which, target)) __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");
    };

    // DMA of patch m's NEW rows into the ring: all ROWS rows for the first patch of the block or of a grid column, else the
    // RXI rows below the 2K rows it shares with patch m - 1 (already in the zone: slots are never moved).  Row r of patch m
    // lives in slot (m * RXI + r) mod ROWS.  This wave's share: 4-row blocks w, w + 4, ... of the new rows.  The caller has
    // made sure that patch m - 1 has been read (counters[1] >= 4 m).
    auto issueDma = [&](int m) {
        if (dbg & 2) return;
        const int c = c0 + m;
        const int pj = c / a.ntx, ti = c - pj * a.ntx;
        const int r0 = (m > 0 && ti != 0) ? 2 * K : 0;
        const int row0 = a.G - K + ti * RXI, col0 = a.G - K + pj * Gm::TPP * WI;
        int pB = pitchB, plB = planeB;  // (opaque: no loop-invariant offset tables in SGPRs)
        asm volatile("" : "+s"(pB), "+s"(plB));
        int dvoff[3];
        {
            const int l = laneCopy();
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int u = j * 64 + l;
                dvoff[j] = (u / 48) * pB + (u % 48) * 16;
            }
        }
        const int slot0 = (m * (RXI / 4)) % (ROWS / 4);  // first 4-row slot block of the patch
#pragma unroll 1
        for (int rb = r0 / 4 + w; rb < ROWS / 4; rb += 4) {
            int sb = slot0 + rb;
            if (sb >= ROWS / 4) sb -= ROWS / 4;
            const int so = ((row0 + 4 * rb) * a.pitch + col0) * 4;
            const unsigned l0 = zoneB + (unsigned)(sb * 4 * PW * 4);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                patchDma16(rIn, dvoff[j], so, l0 + (unsigned)(j * 1024));
                patchDma16(rIn, dvoff[j], so + plB, l0 + (unsigned)(PLANE * 4 + j * 1024));
                patchDma16(rIn, dvoff[j], so + 2 * plB, l0 + (unsigned)(2 * PLANE * 4 + j * 1024));
            }
        }
    };

    // prologue: the group's first patch.  Group 1's (m = 1) may only overwrite the zone once group 0 has read patch 0.
    if (g < n) {
        if (g == 1) waitFor(1, 4);
        issueDma(g);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bump(0);
    }

    const float C = a.courant;
    const bool inCols = lane >= K && lane < 64 - K;
    // development aid: stamp `id` of this wave's tile number `it` (block 0 only, 16 stamps x 16 tiles x 8 waves)
    const bool tracing = a.patchTrace != nullptr && blockIdx.x == 0 && a.nsteps == K;
    auto stamp = [&](int it, int id) {
        if (tracing && it < 16) {
            const long long t = (long long)__builtin_amdgcn_s_memtime();
            if (lane == 0) a.patchTrace[(wave * 16 + it) * 16 + id] = t;
        }
    };

    v2f pr[NP], vx[NP], vy[NP];
    float vxS = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) pr[i] = vx[i] = vy[i] = v2f{0.f, 0.f};

#pragma unroll 1
    for (int m = g; m < n; m += 2) {
        const int c = c0 + m;
        const int pj = c / a.ntx, ti = c - pj * a.ntx;
        const int tj = pj * Gm::TPP + w;
        bool mine = tj < a.nty;
        const int tile = ti * a.nty + tj;
        const DynParams dyn = patchSLoad(a.dyn);  // (per tile: a copy kept live across the loop costs 10 SGPRs)
        if (mine) {
            if (patchSLoadByte(a.tileClass + tile) != 0) mine = false;  // general tile: its own launch
            if (a.withPulse) {                                          // so is the tile that holds the listener
                const int lr = dyn.lrow - (a.G - K + ti * RXI), lc = dyn.lcol - (a.G - K + tj * WI);
                if (lr >= 0 && lr < ROWS && lc >= 0 && lc < 64) mine = false;
            }
        }
        mine = __builtin_amdgcn_readfirstlane((int)mine) != 0;

        // ---- take the tile out of the zone (all four DMA shares of the patch have landed), steps [0, K/2)
        stamp(m >> 1, 0);
        waitFor(0, 4 * (m + 1));
        stamp(m >> 1, 1);
        bool recLane = false;
        long long hstride = a.histPlane;
        int ns = a.nsteps;
        // (opaque per tile: S * histPlane and the masks of S < nsteps for the 12 steps would be hoisted into 48 SGPRs)
        asm volatile("" : "+s"(hstride), "+s"(ns));
        const float* hplane = a.hist + (long long)a.histSlot * hstride;
        int hsoff0 = 0;
        const int hvoff = (laneCopy() - K) * 4, hpitchB = WI * 4;
        if (mine && !(dbg & 8)) {
            // row r of the patch sits in slot group (bq + r / 12) mod 5, row r % 12 of it
            const int bq = (m * (RXI / 12)) % SLOTG;
            const float* zl = zone + w * WI + laneCopy();
            const float* gp[SLOTG];
#pragma unroll
            for (int k = 0; k < SLOTG; ++k) {
                int sg = bq + k;
                if (sg >= SLOTG) sg -= SLOTG;
                gp[k] = zl + sg * 12 * PW;
            }
            auto at = [&](int plane, int r) { return gp[r / 12][plane * PLANE + (r % 12) * PW]; };
            vxS = at(1, NP);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                pr[i].x = at(0, i);
                pr[i].y = at(0, ROWS - 1 - i);
                vy[i].x = at(2, i);
                vy[i].y = at(2, ROWS - 1 - i);
                vx[i].x = at(1, i);
                vx[i].y = (i > 0) ? -at(1, ROWS - i) : 0.f;  // face ROWS-i; face ROWS is not in the tile
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bump(1);  // this wave is done with the zone
        stamp(m >> 1, 2);
        if (mine) {
            uint32_t nz = __float_as_uint(vxS);
#pragma unroll
            for (int i = 0; i < NP; ++i)
                nz |= __float_as_uint(pr[i].x) | __float_as_uint(pr[i].y) | __float_as_uint(vx[i].x) |
                      __float_as_uint(vx[i].y) | __float_as_uint(vy[i].x) | __float_as_uint(vy[i].y);
            const bool active = __ballot((nz & 0x7fffffffu) != 0u) != 0ull;
            if (lane == 0) a.nzOut[tile] = active ? 1 : 0;
            const int hti = ti - dyn.histTileX0, htj = tj - dyn.histTileY0;
            const bool inWin = hti >= 0 && hti < dyn.histTilesX && htj >= 0 && htj < dyn.histTilesY;
            const bool wasActive = a.record && patchSLoad(a.tileFirst + tile) != INT_MAX;
            const bool rec = a.record && inWin && (active || wasActive || a.dense);  // (no streaming analysis here: every
                                                                                      // history plane is wanted)
            if (a.record && active && !wasActive && lane == 0) atomicMin(&a.tileFirst[tile], a.t0);
            if (a.record && active && !inWin && lane == 0) atomicExch(a.errFlag, 1);
            recLane = rec && inCols;
            hsoff0 = ((hti * dyn.histTilesY + htj) * RXI - K) * hpitchB;
            if (!(dbg & 1))
                PatchSteps<K, RXI, 0, K / 2>::run(pr, vx, vy, vxS, C, ns, recLane, hplane, hstride, hvoff, hsoff0, hpitchB);
        }

        // ---- second half: fetch the group's next patch as soon as the other group has read ITS current one (asked after
        // every step, forced before the result stores), steps [K/2, K)
        stamp(m >> 1, 3);
        const bool more = m + 2 < n;
        bool issued = !more;
        auto tryIssue = [&](bool force) {
            if (issued) return;
            if (force)
                waitFor(1, 4 * (m + 2));
            else if (!reached(1, 4 * (m + 2)))
                return;
            issueDma(m + 2);
            issued = true;
        };
        tryIssue(false);
        if (mine && !(dbg & 1)) {
            PatchSteps<K, RXI, K / 2, K / 2 + 2>::run(pr, vx, vy, vxS, C, ns, recLane, hplane, hstride, hvoff, hsoff0, hpitchB);
            tryIssue(false);
            PatchSteps<K, RXI, K / 2 + 2, K / 2 + 4>::run(pr, vx, vy, vxS, C, ns, recLane, hplane, hstride, hvoff, hsoff0, hpitchB);
            tryIssue(false);
            PatchSteps<K, RXI, K / 2 + 4, K>::run(pr, vx, vy, vxS, C, ns, recLane, hplane, hstride, hvoff, hsoff0, hpitchB);
        }
        stamp(m >> 1, 4);
        tryIssue(true);
        stamp(m >> 1, 5);
        // the next patch's shares must have landed before this wave says so -- and waiting BEFORE the result stores are issued
        // keeps them out of the wait (vmcnt counts loads and stores in issue order)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (more) bump(0);
        stamp(m >> 1, 6);
        if (mine && inCols && !(dbg & 4)) {
            const int row0 = a.G - K + ti * RXI, col0 = a.G - K + tj * WI;
            int pB = pitchB, plB = planeB;  // (opaque: see issueDma)
            asm volatile("" : "+s"(pB), "+s"(plB));
            int so = (row0 * a.pitch + col0) * 4 + K * pB;
            const rsrc_t rOut = makeRsrc(a.prOut, 3 * planeB);
            const int voffLane = laneCopy() * 4;
#pragma unroll
            for (int r = K; r < ROWS - K; ++r) {
                bufStoreF(r < NP ? pr[r].x : pr[ROWS - 1 - r].y, rOut, voffLane, so);
                bufStoreF(r < NP ? vx[r].x : (r == NP ? vxS : -vx[ROWS - r].y), rOut, voffLane, so + plB);
                bufStoreF(r < NP ? vy[r].x : vy[ROWS - 1 - r].y, rOut, voffLane, so + 2 * plB);
                so += pB;
            }
        }
        stamp(m >> 1, 7);
    }
}

}  // namespace pva
