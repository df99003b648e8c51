// pv_analysis_dev.h -- the per-cell bodies of the impulse-response analysis (Analyzer.cpp:139-431) as device functions, shared
// by the one-pass-per-launch kernels (pv_kernels.hip, pv_rt60.hip) and the fused kernel of the small grids (pv_fused.hip).
// Moved here from those files in round 5; the arithmetic is unchanged.
//
// SC1 (template flag of the functions that exchange data BETWEEN workgroups inside one launch -- the fused kernel): the delay
// and occlusion maps and the direction table are then written with agent-scope (write-through, `sc1`) stores and read with
// agent-scope (L1-bypassing) loads, the hand-off form the resident kernel uses (MI355X_MICROARCH.md, "Workgroup dispatch, XCD
// placement & inter-workgroup visibility").  With SC1 = false they are plain accesses: a kernel boundary lies between writer
// and reader.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>

#include "pv_analysis.h"
#include "pv_device.h"
#include "pv_libm.h"
#include "pv_prims.h"

namespace pva {

template <bool SC1>
__device__ __forceinline__ float xLoadF(const float* p) {
    if constexpr (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool SC1>
__device__ __forceinline__ int xLoadI(const int* p) {
    if constexpr (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool SC1>
__device__ __forceinline__ void xStoreF(float* p, float v) {
    if constexpr (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool SC1>
__device__ __forceinline__ void xStoreI(int* p, int v) {
    if constexpr (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

__device__ __forceinline__ int waveMin(int v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v = min(v, __shfl_xor(v, off));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int waveMax(int v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v = max(v, __shfl_xor(v, off));
    return __builtin_amdgcn_readfirstlane(v);
}

// samples per memory round trip of the forward pass (three planes each)
#ifndef PV_ENCODE_CH
#define PV_ENCODE_CH 8
#endif

// ---------------------------------------------------------------------------------------------------------------
// dry gain, source directivity, low-pass cutoff (+ wet gain) of the 64 cells of one wave: Analyzer.cpp:170-247
// ---------------------------------------------------------------------------------------------------------------
// One lane per cell, lanes along the tile-major plane (planeCell); all sums are sequential float32 accumulations in the
// reference's order.  vx / vy are not stored: they are re-derived from the pressure history with the stencil's own
// recurrence (v_t = v_{t-1} - C (p_t[i] - p_t[n]) on air|air faces, k (p_i + p_n) otherwise), bit-identical to the values the
// step kernel held.
// TIME is wave-uniform: every load instruction reads ONE plane (the wave's 256 contiguous bytes of it), from the earliest first
// sample of the wave's cells to the latest last one, and a lane whose own window has not begun or is over loads nothing and adds
// nothing.  (Per-lane start and end times -- each lane walking its own [first, onset + N) -- put up to 64 planes into one load
// instruction: 0.65 ms instead of 0.07 for the 104 000 cells of the 512^2 / T = 3179 room.)
// Every lane of the wave must call (wave reductions inside); `live` = the lane's cell has an onset.  withWet (wave-uniform): the
// wet gain too (beside the lane-per-cell decay-time form and in the fused kernel; the sixteen- and four-lane decay-time forms
// compute their own).
template <bool SC1>
__device__ __forceinline__ void encodeWave(const AnalyzeArgs& a, const DynParams& dyn, const PlaneCell& pc0, const bool live,
                                           const int onsetIn, const bool withWet) {
    const int X = pc0.X, Y = pc0.Y;
    const int s = X * a.gy + Y;
    const int onset = live ? onsetIn : 0;
    const int T = a.T;
    constexpr int kOut = 0x7fffffff;  // a buffer offset beyond every extent: the load returns 0 without touching memory
    const int prow = X + a.G, pcol = Y + a.G;

    // The cell's own samples and those of its neighbours (X - 1, Y) and (X, Y - 1), for the velocity recurrence, are plane
    // offsets: inside a tile one row / one cell back, across a tile's first row / column into the tile above / to the left.  A
    // neighbour tile that became active later (or never) has unwritten history that is exactly zero by causality, and so has a
    // neighbour outside the window.  (Everything here is loads that do not depend on each other and a handful of compares: the
    // first version of this pass went through histOffset's divisions three times and four dependent round trips before its
    // first sample -- a third of the 36 us the kernel took at 70^2.)
    const int tileCells = a.rxi * a.wi;
    const bool hasX = pc0.hti > 0 || pc0.row > 0, hasY = pc0.htj > 0 || pc0.col > 0;
    const int gX = pc0.row > 0 ? pc0.g - a.wi : pc0.g - dyn.histTilesY * tileCells + (a.rxi - 1) * a.wi;
    const int gY = pc0.col > 0 ? pc0.g - 1 : pc0.g - tileCells + (a.wi - 1);
    const int tileX = pc0.row > 0 ? pc0.tile : pc0.tile - a.nty, tileY = pc0.col > 0 ? pc0.tile : pc0.tile - 1;
    int tFirst = T, tFx = INT_MAX, tFy = INT_MAX;
    FaceCoef fc{0.f, 0.f, 0.f};
    if (live) {
        tFirst = a.tileFirst[pc0.tile];
        if (hasX) tFx = a.tileFirst[tileX];
        if (hasY) tFy = a.tileFirst[tileY];
        fc = a.coef[(size_t)prow * a.pitch + pcol];
    }
    // first row of a slab: the row above lives in the neighbouring slab, as a dense [T][histPitch] array
    const bool above = live && X == 0 && a.histAbove != nullptr;
    if (above) tFx = 0;
    const float kx = fc.kx, ky = fc.ky;
    const bool airX = kx != kx, airY = ky != ky;
    const float C = a.courant;

    // The loop walks the history in chunks of CH samples: the CH loads are issued together (they do not depend on the
    // running sums), then consumed strictly in sample order, so the float32 accumulation order is the reference's while the
    // memory latency is paid once per chunk instead of once per sample.  Loads are buffer loads, one descriptor per plane built
    // on the scalar unit; a lane whose window does not hold the sample loads through kOut (no branch around any load).
    constexpr int CH = PV_ENCODE_CH;
    const int planeBytes = (int)(a.histPlane * 4);
    auto planeRsrc = [&](int t) { return makeRsrc(a.hist + (long long)t * a.histPlane, planeBytes); };
    const int vo = pc0.g * 4, voX = gX * 4, voY = gY * 4;

    // dry energy + flux, Analyzer.cpp:170-195: both sums run from sample 0 (samples before tFirst are zero) to the end of
    // their windows behind the onset; vx / vy by the stencil's own recurrence as long as the flux needs them
    const int sourceDirEnd = live ? onset + a.nDir : 0, directEnd = live ? min(onset + a.nDry, T) : 0;
    // (first sample that can be non-zero: the tile's first recorded step, and never before the pulse can have arrived through
    // the stencil -- one cell per step along one axis, so this cell's pressure is exactly zero up to step m and its neighbours'
    // up to step m - 1: pv_onset_kernel)
    const int m = abs(X - (dyn.lrow - a.G)) + abs(Y - (dyn.lcol - a.G));
    const int tBegin = max(tFirst, m - 1);
    const int tLo = waveMin(live ? tBegin : INT_MAX), tHi = waveMax(directEnd), tVHi = waveMax(sourceDirEnd);
    float Edry = 0.f, fluxX = 0.f, fluxY = 0.f, vx = 0.f, vy = 0.f;
#pragma unroll 1
    for (int t0 = tLo; t0 < tHi; t0 += CH) {
        float pc[CH], pxc[CH], pyc[CH];
        const bool needVChunk = t0 < tVHi;  // (wave-uniform)
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int t = t0 + k, tt = min(t, T - 1);
            const bool mine = t >= tBegin && t < directEnd, mineV = mine && t < sourceDirEnd;
            const rsrc_t rs = planeRsrc(tt);
            pc[k] = bufLoadF(rs, mine ? vo : kOut, 0);
            if (needVChunk) {
                pxc[k] = bufLoadF(rs, (mineV && hasX && tt >= tFx) ? voX : kOut, 0);
                pyc[k] = bufLoadF(rs, (mineV && hasY && tt >= tFy) ? voY : kOut, 0);
            } else {
                pxc[k] = pyc[k] = 0.f;
            }
        }
        if (a.histAbove != nullptr && needVChunk) {  // (slab groups only, wave-uniform)
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int t = t0 + k, tt = min(t, T - 1);
                const bool mineV = t >= tBegin && t < directEnd && t < sourceDirEnd;
                const float pa = bufLoadF(makeRsrc(a.histAbove + (long long)tt * a.histPitch, a.histPitch * 4),
                                          (above && mineV) ? (pcol - dyn.histCol0) * 4 : kOut, 0);
                pxc[k] = above ? pa : pxc[k];
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int t = t0 + k;
            const bool mine = t >= tBegin && t < directEnd, mineV = mine && t < sourceDirEnd;
            const float p = pc[k];  // (0 outside the lane's window: the sums below then stay as they are, bit for bit)
            const float pxn = pxc[k], pyn = pyc[k];
            const float ax = vx - C * (p - pxn), wx = kx * (p + pxn);
            const float ay = vy - C * (p - pyn), wy = ky * (p + pyn);
            vx = mineV ? (airX ? ax : wx) : vx;
            vy = mineV ? (airY ? ay : wy) : vy;
            Edry = mine ? Edry + p * p : Edry;
            fluxX = mineV ? fluxX + p * vx : fluxX;
            fluxY = mineV ? fluxY + p * vy : fluxY;
        }
    }
    CellHistory hc{a.hist + pc0.g, a.histPlane};

    // wet gain, Analyzer.cpp:235-247: forwards over [onset + N_dry + 1, + N_wet) ^ [0, T)
    float wet = 0.f;
    if (withWet) {
        constexpr int WCH = 32;  // one plane per sample: more of them per memory round trip
        const int wetBegin = live ? onset + a.nDry + 1 : INT_MAX, wetEnd = live ? min(wetBegin + a.nWet, T) : 0;
        const int wLo = waveMin(wetBegin), wHi = waveMax(wetEnd);
#pragma unroll 1
        for (int t0 = wLo; t0 < wHi; t0 += WCH) {
            float pw[WCH];
#pragma unroll
            for (int k = 0; k < WCH; ++k) pw[k] = (t0 + k >= wetBegin && t0 + k < wetEnd) ? hc.at(t0 + k) : 0.f;
#pragma unroll
            for (int k = 0; k < WCH; ++k) wet = wet + pw[k] * pw[k];  // + 0 outside the lane's window
        }
    }
    if (!live) return;

    // obstruction gain + source directivity, Analyzer.cpp:197-220
    const float EfreePr = efreePerR(a.efree, a.dx, a.lcx, a.lcy, X + a.x0, Y);
    const float occ = sqrtf(Edry / EfreePr);
    float norm = sqrtf(fluxX * fluxX + fluxY * fluxY);
    norm = -1.0f / (norm > 0.0f ? norm : 1.0f);
    const float sdx = norm * fluxX, sdy = norm * fluxY;

    // low-pass cutoff, Analyzer.cpp:227-230 (std::max(0.001f, g) == (0.001f < g) ? g : 0.001f)
    const float rr = 1.0f / ((0.001f < occ) ? occ : 0.001f);
    const float lowpass = -147.f + (18390.f) / (1.f + pvPowf(rr / 12.f, 0.8f));

    xStoreF<SC1>(a.out + s, occ);  // (read by the listener-direction pass, of OTHER cells)
    if (withWet) a.out[a.resN + s] = sqrtf(wet / a.efree);
    a.out[3 * a.resN + s] = lowpass;
    a.out[6 * a.resN + s] = sdx;
    a.out[7 * a.resN + s] = sdy;
}

// ---------------------------------------------------------------------------------------------------------------
// wet gain + decay time, SIXTEEN lanes (one DPP row) per cell, four cells per wave
// ---------------------------------------------------------------------------------------------------------------
// Lane j of a row holds sample i0 - j of a 16-sample chunk, walking backwards from T - 1; the chunk's 16 loads are one
// instruction and its 16 log10f evaluations run side by side.  The three running sums stay strictly sequential, in the
// reference's order: each is a chain of 16 steps per chunk in which lane j adds its addend to the value of lane j - 1, fetched by
// a DPP row rotation riding on the add (v_add_f32 row_ror:1); lane 0 thereby reads lane 15, which still holds the chain's value
// at the end of the previous chunk, so the carry between chunks needs no broadcast.  Lanes outside the regression range add
// +0.0f, which leaves a non-negative-zero float sum unchanged bit for bit.  Sixteen times the threads of one lane per cell, a
// sixteenth of its dependent work per thread: the form for the few thousand cells of a closed room.
__device__ __forceinline__ float rowRor1Add(float acc, float addend) {  // acc[lane - 1 in its row of 16] + addend
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x121, 0xf, 0xf,
                                                                  false)) + addend;
}

// One chunk of a sequential sum along a row of 16 lanes: lane j ends up having added a_0 ... a_j, in that order, to the value
// lane 15 carried in.  Every step is ONE instruction for the whole wave, acc = acc[lane - 1] + a (v_add_f32 row_ror:1), with no
// lane frozen: at step j lane j reads lane j - 1, which became final at step j - 1, and what a lane holds after its own step is
// never read again -- except lane 15's, the carry into the next chunk, and that one is final because its step is the last.
// (Rounds 3-4 froze every lane after its step with a select ON the chain -- add, select, two wait states for the DPP read of a
// just-written register -- 16 x ~16 cycles per chain and chunk for a lone wave; here the chain is the adds alone, and a lane that
// needs its own partial sum -- the energy per sample -- captures it with a select BESIDE the chain.)
template <bool CAPTURE>
__device__ __forceinline__ float rowScan(float& acc, const float a, const int sub) {
    float own = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        acc = rowRor1Add(acc, a);
        if (CAPTURE) own = sub == j ? acc : own;
    }
    return own;
}
// two independent sums at once: their instructions alternate, which covers the wait states between a write and its DPP read
__device__ __forceinline__ void rowScan2(float& accA, const float aA, float& accB, const float aB) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        accA = rowRor1Add(accA, aA);
        accB = rowRor1Add(accB, aB);
    }
}

#ifndef PV_RT60_AHEAD
#define PV_RT60_AHEAD 4
#endif
constexpr int kRt60Ahead = PV_RT60_AHEAD;  // chunks of history loads in flight per wave in the sixteen- and four-lane decay-time forms

// every lane of the wave must call; live / s / hc / startingPointIn are the values of the lane's cell (the same in the 16
// lanes of a row), sub = the lane's place in its row
// tab: the logarithm's table functor (LogTabLds: a per-lane index into a constant array is a global load at the head of every
// evaluation -- the one dependent memory round trip per chunk this form had left, round 6)
template <class TabF>
__device__ __forceinline__ void rt60WaveBody(const AnalyzeArgs& a, const TabF& tab, const int sub, const bool live, const int s,
                                             const CellHistory hc, const int startingPointIn) {
    const int T = a.T;
    const int endPoint = T - a.nCut;
    const int startingPoint = live ? startingPointIn : T;  // dead rows: no sample is in range
    const int lowest = min(startingPoint, endPoint);         // the pre-sum over [endPoint, T) is not bounded by the onset
    // wave-uniform trip count: the longest of the four cells
    int n = max(T - lowest, 0);
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) n = max(n, __shfl_xor(n, off));
    n = __builtin_amdgcn_readfirstlane(n);
    float edc = 0.f, xysum = 0.f, ysum = 0.f;  // lane 15 of the row carries them from chunk to chunk
    // kRt60Ahead chunks of loads in flight (round 6; one until then): a ring of fixed registers, the loop unrolled over it, every
    // load issued in consumption order (the wait counts then retire one chunk at a time).  Same additions in the same order.
    // Measured: no change -- the pass is bound by its dependent DPP chains and the logarithm, not by the loads (what did help is
    // the logarithm's table in LDS: profiles/r06_analysis_chain.txt); kept, it costs nothing and takes the loads off the table.
    float ring[kRt60Ahead];
#pragma unroll
    for (int b = 0; b < kRt60Ahead; ++b) {
        const int i = T - 1 - 16 * b - sub;
        ring[b] = (live && 16 * b < n && i >= lowest && i >= 0) ? hc.at(i) : 0.f;
    }
#pragma unroll 1
    for (int n0 = 0; n0 < n; n0 += 16 * kRt60Ahead) {
#pragma unroll
        for (int b = 0; b < kRt60Ahead; ++b) {
            const int c0 = n0 + 16 * b;
            if (c0 < n) {  // (scalar)
                const int i = T - 1 - c0 - sub;
                const float p = ring[b];
                {  // the slot's next occupant
                    const int in = i - 16 * kRt60Ahead;
                    ring[b] = (live && c0 + 16 * kRt60Ahead < n && in >= lowest && in >= 0) ? hc.at(in) : 0.f;
                }
                const float e = rowScan<true>(edc, p * p, sub);  // (p = 0 outside [lowest, T): edc + 0 = edc)
                const bool regress = i >= startingPoint && i < endPoint;
                const float y = 10.f * pvLog10fNonNegT(regress ? e : 1.f, tab);
                rowScan2(xysum, regress ? y * (float)(i - startingPoint) : 0.f, ysum, regress ? y : 0.f);
            }
        }
    }
    // wet gain (Analyzer.cpp:235-247): the same chain, forwards over [startingPoint, startingPoint + N_wet) ^ [0, T)
    const int wetEnd = min(startingPoint + a.nWet, T);
    int nw = max(wetEnd - startingPoint, 0);
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) nw = max(nw, __shfl_xor(nw, off));
    nw = __builtin_amdgcn_readfirstlane(nw);
    float wet = 0.f;
    float pNext = (live && startingPoint + sub < wetEnd) ? hc.at(startingPoint + sub) : 0.f;
#pragma unroll 1
    for (int j0 = 0; j0 < nw; j0 += 16) {
        const float p = pNext;
        const int jn = startingPoint + j0 + 16 + sub;
        pNext = (live && j0 + 16 < nw && jn < wetEnd) ? hc.at(jn) : 0.f;
        (void)rowScan<false>(wet, p * p, sub);
    }
    if (live && sub == 15) {
        a.out[a.resN + s] = sqrtf(wet / a.efree);
        a.out[2 * a.resN + s] = rt60FromSums(a, startingPointIn, xysum, ysum);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wet gain + decay time, blocked: L lanes per cell, each S consecutive samples of a chunk (pv_rt60.hip's header)
// ---------------------------------------------------------------------------------------------------------------
struct LogTabLds {
    const double* t;  // 48 x {invc, logc + kk ln2} in LDS, entry (kk + 1) * 16 + i
    __device__ __forceinline__ void operator()(int i, int kk, double* invc, double* y0) const {
        const int e = (kk + 1) * 16 + i;
        *invc = t[2 * e];
        *y0 = t[2 * e + 1];
    }
};
// the table: 96 doubles of LDS filled by the first 96 threads of a block of at least that many (the caller synchronises)
__device__ __forceinline__ void fillLogTab(double* tab, const int tid, const int nthreads) {
    for (int w = tid; w < 96; w += nthreads) {
        double invc, y0;
        const int e = w >> 1;
        PvLogTabConst{}(e & 15, (e >> 4) - 1, &invc, &y0);
        tab[w] = (w & 1) ? y0 : invc;
    }
}

// value of lane j - 1 of this lane's group of L (lane 0 reads lane L - 1)
template <int L>
__device__ __forceinline__ float prevInGroup(float v) {
    if constexpr (L == 16)  // DPP row_ror:1
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    else if constexpr (L == 4)  // DPP quad_perm:[3,0,1,2]
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x93, 0xf, 0xf, false));
    else
        return v;
}

// One chunk of a sequential sum shared by the L lanes of a group: acc = (...((acc + add_0[0]) + add_0[1]) ... + add_{L-1}[S-1]),
// lane j's addends after lane j-1's.  Every lane runs every step (SIMD), lane j keeps step j; the value a chunk ends with
// stays in lane L-1, which is where lane 0 of the next chunk fetches it from.  KEEP: also the S partial sums of the own step.
template <int L, int S, bool KEEP>
__device__ __forceinline__ void groupChain(float& acc, const float (&add)[S], const int sub, float (&part)[S]) {
#pragma unroll
    for (int j = 0; j < L; ++j) {
        float t = prevInGroup<L>(acc);
        float tmp[S];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            t = t + add[k];
            tmp[k] = t;
        }
        const bool mine = (L == 1) || sub == j;
        acc = mine ? t : acc;
        if (KEEP) {
#pragma unroll
            for (int k = 0; k < S; ++k) part[k] = mine ? tmp[k] : part[k];
        }
    }
}

// every lane of the wave must call; live / s / h0 / startingPointIn are the values of the lane's cell (the same in the L lanes
// of a group), sub = the lane's place in its group
template <int L, int S>
__device__ __forceinline__ void rt60BlockedBody(const AnalyzeArgs& a, const LogTabLds& ltab, const int sub, const bool live,
                                                const int s, const float* const h0In, const int startingPointIn) {
    const int T = a.T;
    const int endPoint = T - a.nCut;
    const int startingPoint = live ? startingPointIn : T;  // dead lanes: no sample is in range
    const int lowest = min(startingPoint, endPoint);         // the pre-sum over [endPoint, T) is not bounded by the onset
    const long long plane = a.histPlane;
    const float* const h0 = live ? h0In : a.hist;

    // ---- decay time: backwards from T - 1; wave-uniform trip count = the longest of the wave's cells ----
    int n = T - lowest;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) n = max(n, __shfl_xor(n, off));
    n = __builtin_amdgcn_readfirstlane(n);  // (wave-uniform by value: scalar loop counters and branches)
    float edc = 0.f, xysum = 0.f, ysum = 0.f;
    // kRt60Ahead chunks of loads in flight, as in rt60WaveBody (round 6; one until then).  Measured: no change -- 125 us for 36 000
    // cells at T = 1187 either way: 2 176 waves of ~21 000 dependent instructions on 1 024 SIMDs, two or three waves per SIMD decide
    float ring[kRt60Ahead][S];
#pragma unroll
    for (int b = 0; b < kRt60Ahead; ++b) {
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const int i = T - 1 - b * L * S - sub * S - k;
            ring[b][k] = (live && b * L * S < n && i >= lowest && i >= 0) ? h0[(long long)i * plane] : 0.f;
        }
    }
#pragma unroll 1
    for (int n0 = 0; n0 < n; n0 += kRt60Ahead * L * S) {
#pragma unroll
        for (int b = 0; b < kRt60Ahead; ++b) {
            const int c0 = n0 + b * L * S;
            if (c0 < n) {  // (scalar)
                const int iTop = T - 1 - c0 - sub * S;  // this lane's samples: iTop - k
                float q[S];
#pragma unroll
                for (int k = 0; k < S; ++k) q[k] = ring[b][k] * ring[b][k];  // 0 outside [lowest, T): edc + 0 = edc
                if (c0 + kRt60Ahead * L * S < n) {  // the slot's next occupant
#pragma unroll
                    for (int k = 0; k < S; ++k) {
                        const int i = iTop - kRt60Ahead * L * S - k;
                        ring[b][k] = (live && i >= lowest && i >= 0) ? h0[(long long)i * plane] : 0.f;
                    }
                }
                float e[S];
#pragma unroll
                for (int k = 0; k < S; ++k) e[k] = 0.f;
                groupChain<L, S, true>(edc, q, sub, e);
                float ax[S], ay[S];
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    const int i = iTop - k;
                    const bool regress = i >= startingPoint && i < endPoint;
                    const float y = 10.f * pvLog10fNonNegT(regress ? e[k] : 1.f, ltab);
                    ax[k] = regress ? y * (float)(i - startingPoint) : 0.f;
                    ay[k] = regress ? y : 0.f;
                }
                float unused[S];
                groupChain<L, S, false>(xysum, ax, sub, unused);
                groupChain<L, S, false>(ysum, ay, sub, unused);
            }
        }
    }

    // ---- wet gain (Analyzer.cpp:235-247): the same chain, forwards over [startingPoint, startingPoint + N_wet) ^ [0, T) ----
    const int wetEnd = min(startingPoint + a.nWet, T);
    int nw = max(wetEnd - startingPoint, 0);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) nw = max(nw, __shfl_xor(nw, off));
    float wet = 0.f;
    float pNext[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const int j = startingPoint + sub * S + k;
        pNext[k] = (live && j < wetEnd) ? h0[(long long)j * plane] : 0.f;
    }
#pragma unroll 1
    for (int j0 = 0; j0 < nw; j0 += L * S) {
        float q[S];
#pragma unroll
        for (int k = 0; k < S; ++k) q[k] = pNext[k] * pNext[k];
        if (j0 + L * S < nw) {
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const int j = startingPoint + j0 + L * S + sub * S + k;
                pNext[k] = (live && j < wetEnd) ? h0[(long long)j * plane] : 0.f;
            }
        }
        float unused[S];
        groupChain<L, S, false>(wet, q, sub, unused);
    }
    if (live && sub == L - 1) {
        a.out[a.resN + s] = sqrtf(wet / a.efree);
        a.out[2 * a.resN + s] = rt60FromSums(a, startingPointIn, xysum, ysum);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// dry gain, source directivity, low-pass cutoff with L LANES PER CELL (16 or 4): the small windows' form of encodeWave
// ---------------------------------------------------------------------------------------------------------------
// One lane per cell walks a cell's dry window sample by sample: ~50 instructions per sample, and at 70^2 ... 254^2 the kernel
// lasts as long as its longest walk (a cell behind a corner: 100-250 samples -> 27 us at 70^2, the longest pass of the analysis).
// Here the L lanes of a group take L consecutive samples of a chunk, forwards in time, and the five sequential sums of the
// reference's loop (Analyzer.cpp:170-195) are chains along the group like the decay-time forms' (rowScan): lane j adds its
// term to the value of lane j - 1, fetched by the DPP rotation riding on the add; the last lane carries into the next chunk.
//   vx, vy   on an air|air face the stencil's recurrence v_t = v_{t-1} - C (p_t[i] - p_t[n]) IS a running sum of the terms
//            -(C (p_t[i] - p_t[n])) (a - b and a + (-b) are the same float); every lane keeps the value of ITS sample (the flux
//            needs it).  On a wall face v_t = k (p_t[i] + p_t[n]) needs no chain.
//   Edry, fluxX, fluxY   plain sums; only their totals are wanted: the chain is the adds alone.
// Terms outside a cell's window are +0.0f, which leaves a float sum as it is (a sum that is -0.0f becomes +0.0f: the sign of a
// zero, which no comparison in the test suite or the reference's consumers sees).  Same additions in the same order: same bits.
template <int L, bool CAPTURE>
__device__ __forceinline__ float groupScan(float& acc, const float a, const int sub) {
    float own = 0.f;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        acc = prevInGroup<L>(acc) + a;
        if (CAPTURE) own = sub == j ? acc : own;
    }
    return own;
}

// every lane of the wave must call; gcell = plane offset of the lane's cell (the same in the L lanes of a group), sub = the lane's
// place in its group, live = the cell has an onset
template <int L, bool SC1>
__device__ __forceinline__ void encodeGroups(const AnalyzeArgs& a, const DynParams& dyn, const PlaneCell& pc0, const int sub,
                                             const bool live, const int onsetIn) {
    const int X = pc0.X, Y = pc0.Y;
    const int s = X * a.gy + Y;
    const int onset = live ? onsetIn : 0;
    const int T = a.T;
    const int prow = X + a.G, pcol = Y + a.G;
    const int tileCells = a.rxi * a.wi;
    const bool hasX = pc0.hti > 0 || pc0.row > 0, hasY = pc0.htj > 0 || pc0.col > 0;
    const int gX = pc0.row > 0 ? pc0.g - a.wi : pc0.g - dyn.histTilesY * tileCells + (a.rxi - 1) * a.wi;
    const int gY = pc0.col > 0 ? pc0.g - 1 : pc0.g - tileCells + (a.wi - 1);
    const int tileX = pc0.row > 0 ? pc0.tile : pc0.tile - a.nty, tileY = pc0.col > 0 ? pc0.tile : pc0.tile - 1;
    int tFirst = T, tFx = INT_MAX, tFy = INT_MAX;
    FaceCoef fc{0.f, 0.f, 0.f};
    if (live) {
        tFirst = a.tileFirst[pc0.tile];
        if (hasX) tFx = a.tileFirst[tileX];
        if (hasY) tFy = a.tileFirst[tileY];
        fc = a.coef[(size_t)prow * a.pitch + pcol];
    }
    const bool above = live && X == 0 && a.histAbove != nullptr;  // (first row of a slab: encodeWave)
    if (above) tFx = 0;
    const float kx = fc.kx, ky = fc.ky;
    const bool airX = kx != kx, airY = ky != ky;
    const float C = a.courant;
    const int sourceDirEnd = live ? onset + a.nDir : 0, directEnd = live ? min(onset + a.nDry, T) : 0;
    const int m = abs(X - (dyn.lrow - a.G)) + abs(Y - (dyn.lcol - a.G));
    const int tBegin = max(tFirst, m - 1);  // (first sample that can be non-zero: encodeWave)
    const int tLo = waveMin(live ? tBegin : INT_MAX), tHi = waveMax(directEnd);
    const float* const hbase = a.hist;
    float accVx = 0.f, accVy = 0.f, Edry = 0.f, fluxX = 0.f, fluxY = 0.f;  // the group's last lane carries them
#pragma unroll 1
    for (int t0 = tLo; t0 < tHi; t0 += L) {
        const int t = t0 + sub, tt = min(t, T - 1);
        const bool mine = t >= tBegin && t < directEnd, mineV = mine && t < sourceDirEnd;
        // (the L lanes of a group read L planes: per-lane addresses)
        const float* const hp = hbase + (long long)tt * a.histPlane;
        const float p = mine ? hp[pc0.g] : 0.f;
        float pxn = (mineV && hasX && tt >= tFx) ? hp[gX] : 0.f;
        const float pyn = (mineV && hasY && tt >= tFy) ? hp[gY] : 0.f;
        if (above && mineV) pxn = a.histAbove[(long long)tt * a.histPitch + (pcol - dyn.histCol0)];
        // the velocities of this lane's sample: chain on air faces, closed form on wall faces
        const float dX = (mineV && airX) ? -(C * (p - pxn)) : 0.f, dY = (mineV && airY) ? -(C * (p - pyn)) : 0.f;
        float vx = groupScan<L, true>(accVx, dX, sub), vy = groupScan<L, true>(accVy, dY, sub);
        vx = airX ? vx : kx * (p + pxn);
        vy = airY ? vy : ky * (p + pyn);
        (void)groupScan<L, false>(Edry, mine ? p * p : 0.f, sub);
        (void)groupScan<L, false>(fluxX, mineV ? p * vx : 0.f, sub);
        (void)groupScan<L, false>(fluxY, mineV ? p * vy : 0.f, sub);
    }
    if (!live || sub != L - 1) return;

    // obstruction gain + source directivity, Analyzer.cpp:197-220
    const float EfreePr = efreePerR(a.efree, a.dx, a.lcx, a.lcy, X + a.x0, Y);
    const float occ = sqrtf(Edry / EfreePr);
    float norm = sqrtf(fluxX * fluxX + fluxY * fluxY);
    norm = -1.0f / (norm > 0.0f ? norm : 1.0f);
    const float sdx = norm * fluxX, sdy = norm * fluxY;
    // low-pass cutoff, Analyzer.cpp:227-230 (std::max(0.001f, g) == (0.001f < g) ? g : 0.001f)
    const float rr = 1.0f / ((0.001f < occ) ? occ : 0.001f);
    const float lowpass = -147.f + (18390.f) / (1.f + pvPowf(rr / 12.f, 0.8f));
    xStoreF<SC1>(a.out + s, occ);  // (read by the listener-direction pass, of OTHER cells)
    a.out[3 * a.resN + s] = lowpass;
    a.out[6 * a.resN + s] = sdx;
    a.out[7 * a.resN + s] = sdy;
}

// ---------------------------------------------------------------------------------------------------------------
// listener direction, Analyzer::EncodeListenerDirection, Analyzer.cpp:332-431
// ---------------------------------------------------------------------------------------------------------------
// Analyzer.cpp:332-337
__device__ __forceinline__ int neighbourDr(int i) {
    constexpr int d[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
    return d[i];
}
__device__ __forceinline__ int neighbourDc(int i) {
    constexpr int d[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
    return d[i];
}

// direction = normalised (position of the walk's last cell - listener), Analyzer.cpp:415-428
__device__ __forceinline__ void storeDirection(const AnalyzeArgs& a, int index, int fin) {
    const int r = fin / a.gy, c = fin - r * a.gy;
    float ox = (float)r * a.dx - a.lx, oy = (float)c * a.dx - a.lz;
    float len = (ox * ox) + (oy * oy);
    if (len != 0.f) {
        len = sqrtf(len);
        ox /= len;
        oy /= len;
    }
    a.out[4 * a.resN + index] = ox;
    a.out[5 * a.resN + index] = oy;
}

// The walk by pointer jumping: its cost per cell grows with the path length (an open field at T = 435: up to
// 435 steps of 8 neighbour reads for each of 0.6 M cells; a 25 m room at 4096^2 cells: 191 ms), but where a walk
// goes from a cell does not depend on where it started, so "one step from here" is a functional graph and
// ceil(log2 T) rounds of J[p] = J[J[p]] resolve every walk.  kDirFinal marks entries that are already terminal.
constexpr int kDirFinal = (int)0x80000000;

template <bool SC1 = false>
__device__ __forceinline__ int dirBestNeighbour(const AnalyzeArgs& a, int cell, float* bestDelay) {
    const int r = cell / a.gy, c = cell - r * a.gy;
    int best = -1;
    float bd = FLT_MAX;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int nr = r + neighbourDr(i), nc = c + neighbourDc(i);
        if (nr < 0 || nc < 0 || nr >= a.gx || nc >= a.gy) continue;
        const int ni = nr * a.gy + nc;
        const float occ = xLoadF<SC1>(a.out + ni);
        const float d = xLoadF<SC1>(a.delay + ni);
        if (occ == 0.f) continue;               // Analyzer.cpp:372 (the (unsigned)delay test never fires)
        if (d < bd && occ > 0.f) {              // strict <: first neighbour wins ties
            best = ni;
            bd = d;
        }
    }
    *bestDelay = bd;
    return best;
}

__device__ __forceinline__ bool dirLineOfSight(const AnalyzeArgs& a, int cell, float d) {
    const float geodesic = kCDev * d / (float)a.fs;
    const int r = cell / a.gy, c = cell - r * a.gy;
    const float tx = (float)r * a.dx - a.lx, ty = (float)c * a.dx - a.lz;
    const float euclid = sqrtf((tx * tx) + (ty * ty));
    return fabsf(geodesic - euclid) < 0.3f * (kCDev / (float)a.res);
}

// window-local index of grid cell `cell` (= X*gy + Y), for the per-cell scratch of the direction kernels
__device__ __forceinline__ int analysisWindowIndex(const AnalyzeArgs& a, const DynParams& dyn, int cell) {
    const int r = cell / a.gy, c = cell - r * a.gy;
    return (r - (dyn.histRow0 - a.G)) * a.winCols + (c - (dyn.histCol0 - a.G));
}

// One launch (or phase) follows every unfinished walk for kDirHops hops through the CURRENT table: each hop reads either the
// old or an already-updated entry of the cell it stands on -- both lie further down the same walk -- so a pass
// multiplies the distance an entry spans by at least kDirHops + 1 whatever the interleaving of the threads, and
// ceil(log_{kDirHops+1} T) passes resolve every walk (2 at T = 435 instead of the 9 of hop-doubling).
constexpr int kDirHops = 20;
__host__ __device__ inline int dirJumpPasses(int T) {
    int rounds = 1;  // chains are shorter than T (delays are distinct integers < T)
    for (long long span = kDirHops + 1; span < T + 2; span *= kDirHops + 1) ++rounds;
    return rounds + 1;
}

// the three per-cell steps; J is indexed by window-local cell, its entries are GRID cell indices (| kDirFinal).  A hop always
// lands on a reached cell (finite delay), i.e. inside the window.
template <bool SC1>
__device__ __forceinline__ void dirInitCell(const AnalyzeArgs& a, const DynParams& dyn, int* J, const int p) {
    const float d = xLoadF<SC1>(a.delay + p), o = xLoadF<SC1>(a.out + p);
    int hop = p | kDirFinal;
    if (d > kDelayCloseDev && o < kDistanceGainDev) {
        float nd;
        const int n = dirBestNeighbour<SC1>(a, p, &nd);
        if (n >= 0) hop = (nd >= d || dirLineOfSight(a, n, nd)) ? (n | kDirFinal) : n;
    }
    xStoreI<SC1>(J + analysisWindowIndex(a, dyn, p), hop);
}
template <bool SC1>
__device__ __forceinline__ void dirJumpCell(const AnalyzeArgs& a, const DynParams& dyn, int* J, const int p) {
    const int wp = analysisWindowIndex(a, dyn, p);
    int h = xLoadI<SC1>(J + wp);
    if (h < 0) return;  // final
#pragma unroll 1
    for (int i = 0; i < kDirHops && h >= 0; ++i) h = xLoadI<SC1>(J + analysisWindowIndex(a, dyn, h));
    xStoreI<SC1>(J + wp, h);
}
template <bool SC1>
__device__ __forceinline__ void dirFinalCell(const AnalyzeArgs& a, const DynParams& dyn, const int* J, const int index) {
    int fin = index;
    if (xLoadF<SC1>(a.out + index) < kDistanceGainDev) {  // first iteration: delay = FLT_MAX, so only the loudness test applies
        float nd;
        const int n = dirBestNeighbour<SC1>(a, index, &nd);
        if (n >= 0) fin = dirLineOfSight(a, n, nd) ? n : (xLoadI<SC1>(J + analysisWindowIndex(a, dyn, n)) & ~kDirFinal);
    }
    storeDirection(a, index, fin);
}

}  // namespace pva
