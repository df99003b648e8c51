// pv_seg.h -- row-streaming air segments: the air arm of pv_step_seg_kernel (included by pv_kernels.hip).
//
// The register tile of stepTileAirMirror advances a 60 x 64 block K = 12 steps to write back 36 x 40 cells: 53 % of its
// lanes advance halo cells, every wave loads 46 KB before it computes anything and a launch is a series of rounds of
// such waves.  Here a wave owns a COLUMN STRIP and STREAMS down it (time-skewed / "sliding window" temporal blocking):
//
//   * lane l holds NC ADJACENT columns (NC*l .. NC*l+NC-1) of the strip, one buffer_load_dwordx4 (NC = 4) per field
//     and row.  A strip is 64*NC columns wide, of which K on either side are the y halo: with NC = 4 and K = 12,
//     200 of the 256 columns (5 tiles of 40) are interior -- 78 % instead of 62.5 % of the lanes -- and three of
//     four y-neighbours are the lane's OWN registers (plain v_sub_f32, 2.7 cycles) instead of DPP reads (4.3-4.9
//     cycles, tools/valu_probe.hip).
//   * every iteration j loads ONE new row (time level 0) and advances the K rows above it by one level each: row
//     j-s-1 goes from level s to s+1, s = 0..K-1, shallowest first.  Level s+1 of row i needs level s of rows i, i+1
//     (the latter was advanced a moment ago in this iteration) and pr of row i-1 at level s+1 (advanced in the
//     previous iteration, not yet in this one): the update is IN PLACE.  A row is loaded once, lives in one ring slot
//     for K+1 iterations and leaves at level K: no x halo is recomputed except K rows at either end of the segment
//     (K * roundup(X + 2K, RS) row updates for K*X useful ones: the loop is branch-free, see `iteration`), and only
//     K+2 rows + the prefetched ones are in registers ((K+2+PF) * 3 * NC: 132 at K = 8, 192 at K = 12), whatever the
//     segment's length.
//   * loads (row j+PF) and stores (row j-K) are spread evenly through the wave's life: no load phase, no store phase.
//
// Measured: the loop runs at the VALU rate (3.6 cycles per instruction and SIMD at K = 8, two waves per
// SIMD) and executes 11-40 % fewer VALU instructions than the tile kernel, but at the segment lengths a 4096^2 or 8192^2
// grid allows it re-reads too much x halo from HBM: slower than the tile kernels there, off by default.
//
// The ring is indexed statically: the loop is unrolled RS = K+2+PF times (row i lives in slot i mod RS).
// Same arithmetic per cell, in the same order, as leapfrogStep<FAST> (FDTD.cpp:124-199 for air|air faces).
#pragma once

#ifndef PV_SEG_OPAQUE
#define PV_SEG_OPAQUE 1
#endif
#ifndef PV_SEG_SCHEDBAR
#define PV_SEG_SCHEDBAR 0
#endif
#ifndef PV_SEG_PF
#define PV_SEG_PF 2
#endif

#include <utility>

namespace pva {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

typedef float v4f __attribute__((ext_vector_type(4)));

// A row of one field in a lane: NC adjacent columns in ONE register tuple -- what buffer_load_dwordx4 / dwordx2 writes
// and buffer_store reads, so that rows enter and leave the ring without a copy.  The packed arithmetic works on its
// aligned 64-bit halves.
template <int NC>
struct SegVec;
template <>
struct SegVec<4> {
    using type = v4f;
};
template <>
struct SegVec<2> {
    using type = v2f;
};
template <int H>
__device__ __forceinline__ v2f segHalf(const v4f& v) {
    if constexpr (H == 0)
        return __builtin_shufflevector(v, v, 0, 1);
    else
        return __builtin_shufflevector(v, v, 2, 3);
}
template <int H>
__device__ __forceinline__ v2f segHalf(const v2f& v) {
    return v;
}
__device__ __forceinline__ v4f segJoin(const v2f lo, const v2f hi, const v4f*) {
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ v2f segJoin(const v2f lo, const v2f, const v2f*) { return lo; }

__device__ __forceinline__ v4f segLoadRow(rsrc_t r, int voff, int soff, const v4f*) {
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return __builtin_bit_cast(v4f, t);
}
__device__ __forceinline__ v2f segLoadRow(rsrc_t r, int voff, int soff, const v2f*) {
    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __builtin_bit_cast(v2f, t);
}
__device__ __forceinline__ void segStoreRow(const v4f d, rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d), r, voff, soff, 0);
}
__device__ __forceinline__ void segStoreRow(const v2f d, rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, d), r, voff, soff, 0);
}
__device__ __forceinline__ unsigned segBitsOr(const v4f d) {
    const u32x4 t = __builtin_bit_cast(u32x4, d);
    return t.x | t.y | t.z | t.w;
}
__device__ __forceinline__ unsigned segBitsOr(const v2f d) {
    const u32x2 t = __builtin_bit_cast(u32x2, d);
    return t.x | t.y;
}

// The y-direction differences are SCALAR subtracts on purpose: the DAG combiner would otherwise build each pair of
// them as shuffle (v_pk_mov_b32) + v_pk_add_f32, 9 cycles for what two plain v_sub_f32 do in 5.5
// (tools/valu_probe.hip).  The empty asm makes each difference an opaque scalar.
__device__ __forceinline__ float segOpaque(float x) {
#if PV_SEG_OPAQUE
    asm("" : "+v"(x));
#endif
    return x;
}
// a - (value of b in lane-1): v_subrev_f32_dpp computes src1 - dpp(src0).  The compiler leaves this operand order as
// v_mov_b32_dpp + v_sub_f32; gfx9-family ISAs need 2 wait states between a VALU write of a VGPR and a DPP read of it,
// which the hazard recogniser cannot see inside inline asm: hence the leading s_nop 1 and the early-clobber output
// (tools/check_dpp_hazard.py scans the generated assembly for this pattern).
__device__ __forceinline__ float segSubLanePrev(float a, float b) {
#if PV_USE_DPP
    float d;
    asm("s_nop 1\n\tv_subrev_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(d)
        : "v"(b), "v"(a));
    return d;
#else
    return a - lanePrev(b);
#endif
}

// d[c] = v[c+1] - v[c] over the lane's NC columns, as 64-bit halves; column NC of lane l is column 0 of lane l+1
__device__ __forceinline__ void segNextDiff(const v4f v, v2f& lo, v2f& hi) {
    lo = v2f{segOpaque(v.y - v.x), segOpaque(v.z - v.y)};
    hi = v2f{segOpaque(v.w - v.z), segOpaque(laneNext(v.x) - v.w)};
}
__device__ __forceinline__ void segNextDiff(const v2f v, v2f& lo, v2f& hi) {
    lo = v2f{segOpaque(v.y - v.x), segOpaque(laneNext(v.x) - v.y)};
    hi = lo;
}
// d[c] = v[c] - v[c-1]; column -1 of lane l is column NC-1 of lane l-1
__device__ __forceinline__ void segPrevDiff(const v4f v, v2f& lo, v2f& hi) {
    lo = v2f{segSubLanePrev(v.x, v.w), segOpaque(v.y - v.x)};
    hi = v2f{segOpaque(v.z - v.y), segOpaque(v.w - v.z)};
}
__device__ __forceinline__ void segPrevDiff(const v2f v, v2f& lo, v2f& hi) {
    lo = v2f{segSubLanePrev(v.x, v.y), segOpaque(v.y - v.x)};
    hi = lo;
}

// a lane offset beyond every buffer descriptor's extent: such a lane's loads return 0 and its stores are dropped, without
// a memory access
constexpr int kSegOob = 0x7ffff000;

template <class F, int... Us>
__device__ __forceinline__ void segUnrolled(F& f, const int j0, std::integer_sequence<int, Us...>) {
    (f(std::integral_constant<int, Us>{}, j0 + Us), ...);
}

template <int K, int NC>
struct SegGeom {
    static constexpr int WI = 64 - 2 * K;
    static constexpr int LH = K / NC;              // halo lanes on either side
    static constexpr int LT = WI / NC;             // lanes per tile column
    static constexpr int WMAX = (64 - 2 * LH) / LT;  // tile columns a strip can hold
    static constexpr int PF = PV_SEG_PF;           // rows in flight ahead of the one being consumed
    static constexpr int RS = K + 2 + PF;          // ring slots = unroll factor
    static constexpr int MAXTR = 8;                // tile rows a segment may touch (bit layout of the per-tile masks)
    static_assert(K % NC == 0 && WI % NC == 0, "halo and tile width must be whole lanes");
};

template <int K, int RXI, int NC>
__device__ __forceinline__ void stepSegment(const StepArgs& a, const SegDesc sd, const int lane) {
    using Sg = SegGeom<K, NC>;
    constexpr int WI = Sg::WI, LH = Sg::LH, LT = Sg::LT, PF = Sg::PF, RS = Sg::RS;
    const int X = sd.nrows;
    const int nin = X + 2 * K;                    // rows loaded
    const int nl = 2 * LH + sd.w * LT;            // lanes that hold columns of the strip
    const int row0 = a.G - K + sd.row0;           // first loaded row / column, padded coordinates
    const int col0 = a.G - K + sd.tj0 * WI;
    const int pitchB = a.pitch * 4;
    const int soff0 = (row0 * a.pitch + col0) * 4;
    const bool storeLane = lane >= LH && lane < LH + sd.w * LT;

    // ONE descriptor per buffer set: the host allocates pr, vx, vy of a set as one block (pv_solver.cpp: vx = pr + plane,
    // vy = pr + 2 planes) and the plane is selected by the scalar offset.  (Three descriptors per set are 24 SGPRs, and
    // this loop needs them.)  Lanes that must not access memory carry an out-of-range lane offset; a whole access is
    // switched off by a descriptor of extent 0 (rNone) -- an SGPR select, no branch.  (The scalar offset takes part in
    // the range check on gfx950, so the extent cannot serve as a lane mask.)
    const int planeB = (int)a.planeBytes;
    const int voff = lane < nl ? lane * (4 * NC) : kSegOob;      // lanes that hold columns of the strip
    const int voffSt = storeLane ? lane * (4 * NC) : kSegOob;    // lanes that write back: the strip's interior columns
    const rsrc_t rIn = makeRsrc(a.prIn, a.inBytes ? 3 * planeB : 0);   // extent 0 on the first launch of a run: zeros
    const rsrc_t rNone = makeRsrc(a.prIn, 0);
    const rsrc_t rOut = makeRsrc(a.prOut, 3 * planeB);

    // ---- which tiles of the segment record their pressure history in this launch ------------------------------
    // lane 8*tr + tc looks after tile (tiFirst + tr, tj0 + tc).  A tile is recorded once it, or one of its 8
    // neighbours, was non-zero at the end of the previous launch (a launch moves the field by K < tile cells).
    // (Evaluated before AND after the stream rather than kept in registers across it: the loop needs them all.)
    const DynParams dyn = *a.dyn;
    const int tiFirst = sd.row0 / RXI;
    const int rit0 = sd.row0 - tiFirst * RXI;      // row inside its tile of the first interior row
    struct MyTile {
        bool mine, inWin, was, near;
        int tile, ti, tj, bit;
    };
    auto myTile = [&]() __attribute__((always_inline)) {
        MyTile m;
        const int ntr = (rit0 + X + RXI - 1) / RXI;    // tile rows touched
        const int mtr = lane >> 3, mtc = lane & 7;
        m.ti = tiFirst + mtr;
        m.tj = sd.tj0 + mtc;
        m.bit = 8 * mtr + mtc;
        m.mine = mtr < ntr && mtc < sd.w && m.ti < a.ntx && m.tj < a.nty;
        m.tile = m.ti * a.nty + m.tj;
        const int hti = m.ti - dyn.histTileX0, htj = m.tj - dyn.histTileY0;
        m.inWin = hti >= 0 && hti < dyn.histTilesX && htj >= 0 && htj < dyn.histTilesY;
        m.was = false;   // the tile counted as non-zero before this launch (flags are monotone within a run)
        m.near = false;  // ... or one of its neighbours did
        if (m.mine) {
            for (int di = -1; di <= 1; ++di)
                for (int dj = -1; dj <= 1; ++dj) {
                    const int u = m.ti + di, v = m.tj + dj;
                    if (u >= 0 && u < a.ntx && v >= 0 && v < a.nty && a.nzIn[u * a.nty + v]) {
                        m.near = true;
                        m.was = m.was || (di == 0 && dj == 0);
                    }
                }
        }
        return m;
    };
    // bit 8*tr + tc: tile (tiFirst + tr, tj0 + tc) records in this launch (only tiles inside the history window)
    unsigned long long recBits = 0;
    if (a.record) {
        const MyTile m = myTile();
        bool rec = false;
        if (m.mine) {
            const bool act = m.near || a.dense || a.tileFirst[m.tile] != INT_MAX;
            if (act && a.tileFirst[m.tile] == INT_MAX) atomicMin(&a.tileFirst[m.tile], a.t0);
            rec = act && m.inWin && historyWanted(a, m.ti, m.tj);
        }
        recBits = __ballot(rec);
    }
    // History stores (tile-major planes, histOffset): the lane part of the address is the lane's tile column and the
    // column inside it (out of range for halo lanes and for tile columns outside the window), the row part is the
    // scalar offset.  A row is recorded when ANY tile of its tile row records (the row's other window tiles are
    // written too: harmless, the analysis reads a tile's history from tileFirst on).
    int hvoff = kSegOob;
    if (recBits != 0ull && storeLane) {
        const int tc = (lane - LH) / LT;
        const int htj = sd.tj0 + tc - dyn.histTileY0;
        if (htj >= 0 && htj < dyn.histTilesY) hvoff = (htj * (RXI * WI) + (lane - LH - tc * LT) * NC) * 4;
    }
    const int histExtent = (int)(a.histPlane * 4);
    const int htrow0 = (tiFirst - dyn.histTileX0) * dyn.histTilesY;  // window tile index of (tiFirst, window column 0)

    using VT = typename SegVec<NC>::type;
    constexpr VT* kVT = nullptr;
    const v2f c2 = {a.courant, a.courant};
    VT P[RS], Vx[RS], Vy[RS];
#pragma unroll
    for (int k = 0; k < RS; ++k) P[k] = Vx[k] = Vy[k] = VT(0.f);
    // running scalars (one SGPR each; written as chains so that the unrolled body does not turn them into RS separate
    // induction variables): offset of the next row to load / to store, row-in-tile and tile row of the next row to store
    int soLd = soff0, soSt = soff0 + K * pitchB, ritSt = rit0, trSt = 0;
    unsigned nzAcc = 0;                 // OR of the stored values of the current tile row (this lane's columns)
    unsigned long long nzTiles = 0;     // bit 8*tr + tc: tile has a non-zero output

    // One level-step: row i (slot sa) from level s to s+1, with row i+1 (slot sb, already at level s) and row i-1 (slot sc,
    // still at level s+1).  Slots are compile-time constants: fixed register sets, updated in place.
    auto levelStep = [&](const int sa, const int sb, const int sc) __attribute__((always_inline)) {
        v2f dyl, dyh, dpl, dph;
        segNextDiff(Vy[sa], dyl, dyh);
        // FDTD.cpp:124-141
        const v2f pl = segHalf<0>(P[sa]) - c2 * ((segHalf<0>(Vx[sb]) - segHalf<0>(Vx[sa])) + dyl);
        const v2f ph = segHalf<1>(P[sa]) - c2 * ((segHalf<1>(Vx[sb]) - segHalf<1>(Vx[sa])) + dyh);
        const VT pn = segJoin(pl, ph, kVT);
        // FDTD.cpp:143-170
        const v2f xl = segHalf<0>(Vx[sa]) - c2 * (pl - segHalf<0>(P[sc]));
        const v2f xh = segHalf<1>(Vx[sa]) - c2 * (ph - segHalf<1>(P[sc]));
        // FDTD.cpp:172-199
        segPrevDiff(pn, dpl, dph);
        const v2f yl = segHalf<0>(Vy[sa]) - c2 * dpl;
        const v2f yh = segHalf<1>(Vy[sa]) - c2 * dph;
        P[sa] = pn;
        Vx[sa] = segJoin(xl, xh, kVT);
        Vy[sa] = segJoin(yl, yh, kVT);
    };
    auto closeTileRow = [&]() __attribute__((always_inline)) {
        const unsigned long long bl = __ballot((nzAcc & 0x7fffffffu) != 0u) >> LH;
#pragma unroll
        for (int tc = 0; tc < Sg::WMAX; ++tc)
            if ((bl >> (tc * LT)) & ((1ull << LT) - 1ull)) nzTiles |= 1ull << (8 * trSt + tc);
        nzAcc = 0;
        ritSt = 0;
        ++trSt;
    };

    // ---- one iteration of the stream; U = j mod RS is a compile-time constant --------------------------------------
    // NO branch around a load or a final store and none per level: the s_waitcnt pass merges the outstanding-operation
    // counts of all paths into a branch target conservatively, and one skipped load turns every wait into vmcnt(0) --
    // no prefetch (measured: waves parked on s_waitcnt 47 % of their cycles).  So loads past the last row and stores
    // before the first finished row go through a descriptor of extent 0 (no memory access), and all K level-steps run
    // from the first iteration on: what they compute before their rows exist is garbage in slots nothing valid ever
    // reads (a level-step only touches the slots of rows j-K-1 .. j, the loads fill those of rows j+1 .. j+PF: disjoint
    // mod RS) -- K^2 wasted row updates per segment.  The loop runs whole chunks of RS iterations (up to RS-1 more
    // wasted iterations at the end).
    auto iteration = [&](auto uc, const int j) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        {
            constexpr int sl = (U + PF) % RS;
            const rsrc_t rl = (j + PF < nin) ? rIn : rNone;
            P[sl] = segLoadRow(rl, voff, soLd, kVT);
            Vx[sl] = segLoadRow(rl, voff, soLd + planeB, kVT);
            Vy[sl] = segLoadRow(rl, voff, soLd + 2 * planeB, kVT);
            soLd += pitchB;
            asm("" : "+s"(soLd));  // keep ONE running offset (the optimiser would precompute RS of them per chunk)
        }
#pragma unroll
        for (int s = 0; s < K; ++s) {
            levelStep(((U - s - 1) % RS + RS) % RS, ((U - s) % RS + RS) % RS, ((U - s - 2) % RS + RS) % RS);
            // (PV_SEG_SCHEDBAR: one level-step = one scheduling region; measured slower at K = 8: 6.6 vs 5.6 ms per run)
#if PV_SEG_SCHEDBAR
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        // Record: the rows j-1 .. j-K now hold levels 1 .. K = the pressure after steps t0 .. t0+K-1, before any pulse
        // (there is none in an air segment), FDTD.cpp:226-234.  Scalar bookkeeping + K stores, skipped by the segments
        // that record nothing (the only branch around memory operations: the s_waitcnt pass then assumes the count of
        // the path without the stores, which is the one that matters).
        if (recBits != 0ull) {
            int q = j - 1 - K;                 // interior row index of row j-1
            const int rr = rit0 + q;
            int tr = rr >= 0 ? rr / RXI : 0;
            int rit = rr - tr * RXI;
            // (opaque: the K plane descriptors would otherwise be hoisted out of the loop -- 48 SGPRs held for a block
            // most segments never enter)
            unsigned long long hp = (unsigned long long)(a.hist + (long long)a.histSlot * a.histPlane);
            asm volatile("" : "+s"(hp));
            const float* hplane = (const float*)hp;
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const bool ok = q >= 0 && q < X && ((recBits >> (8 * (tr & 7))) & 0xffull) != 0ull;
                const rsrc_t rH = makeRsrc(hplane, ok ? histExtent : 0);
                segStoreRow(P[((U - s - 1) % RS + RS) % RS], rH, hvoff, ((htrow0 + tr * dyn.histTilesY) * RXI + rit) * (WI * 4));
                --q;
                if (--rit < 0) {
                    rit = RXI - 1;
                    --tr;
                }
                hplane += a.histPlane;
            }
        }
        // the row that reached level K leaves (rows [K, K+X): 2K <= j < nin)
        constexpr int so = ((U - K) % RS + RS) % RS;
        const bool out = j >= 2 * K && j < nin;
        const rsrc_t rs = out ? rOut : rNone;
        segStoreRow(P[so], rs, voffSt, soSt);
        segStoreRow(Vx[so], rs, voffSt, soSt + planeB);
        segStoreRow(Vy[so], rs, voffSt, soSt + 2 * planeB);
        nzAcc |= (segBitsOr(P[so]) | segBitsOr(Vx[so]) | segBitsOr(Vy[so])) & (out ? 0x7fffffffu : 0u);
        soSt += out ? pitchB : 0;
        asm("" : "+s"(soSt));
        ritSt += out ? 1 : 0;
        if (ritSt == RXI || j == nin - 1) closeTileRow();  // last row of a tile row: close its flags
    };

    auto prologue = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            P[k] = segLoadRow(rIn, voff, soLd, kVT);
            Vx[k] = segLoadRow(rIn, voff, soLd + planeB, kVT);
            Vy[k] = segLoadRow(rIn, voff, soLd + 2 * planeB, kVT);
            soLd += pitchB;
        }
    };
    prologue();
#pragma unroll 1
    for (int j0 = 0; j0 < nin; j0 += RS) segUnrolled(iteration, j0, std::make_integer_sequence<int, RS>{});
    // per-tile non-zero flags (monotone within a run: written, never cleared) and the window check
    {
        const MyTile m = myTile();
        if (m.mine) {
            const bool nzNow = (nzTiles >> m.bit) & 1ull;
            if (nzNow || m.was) a.nzOut[m.tile] = 1;
            // a tile outside the history window must never become non-zero (the window covers the pulse's reach)
            if (a.record && nzNow && !m.inWin) atomicExch(a.errFlag, 1);
        }
    }
}

}  // namespace pva
