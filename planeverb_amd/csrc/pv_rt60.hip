// pv_rt60.hip -- wet gain and decay time (Analyzer.cpp:235-247, 282-327) for ANY number of reached cells.
//
// Per cell and tail sample the reference evaluates one log10f and advances three float32 sums whose order is part of the
// result: the backward energy integral, sum(y x) and sum(y).  Rounds 1-3 had two forms: one thread per cell
// inside pv_encode_kernel (the "cell form": 636-940 cycles per wave and sample, its eight logarithms re-serialised behind
// compiler-inserted branches, each with its table load at the head of the chain: 27-33 % of a 254^2 / 382^2 run) and sixteen
// lanes per cell (pv_rt60_wave_kernel: parallel logarithms, but 96 of its ~160 instructions per 16 samples are the DPP
// chains that keep the sums sequential; up to 65 536 cells).  This is the blocked form for everything above a few thousand
// cells:
//   * L lanes share a cell, each holds S CONSECUTIVE samples of a chunk of L * S (lane j: samples i0 - j S - k, k < S);
//   * the energy integral is a chain over the L lanes -- lane j takes lane j-1's value by one DPP rotation, adds its S
//     squares one after the other, keeps the S partial sums -- so the order of additions is the reference's; lanes outside
//     a range add +0.0f (the identity of a sum that started at +0, bit for bit);
//   * the S logarithms of a lane are independent: branch-free glibc log10f (pv_libm.h) with its 16-entry table in LDS, one
//     basic block, interleaved by the compiler;
//   * the two regression sums are chains like the first.
// Instructions per cell and sample: ~0.9 at (L, S) = (4, 4) against ~2.5 for the sixteen-lane form; which form runs is
// chosen ON THE DEVICE from the number of cells of the window's ever-non-zero tiles (rt60LanesPerCell): a few thousand
// cells need sixteen lanes each for parallelism, everything above runs four.  Same bits in every form.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cstdint>

#include "pv_analysis.h"
#include "pv_analysis_dev.h"
#include "pv_device.h"
#include "pv_launch.h"
#include "pv_libm.h"
#include "pv_prims.h"

#ifndef PV_RT60_TILE_S
#define PV_RT60_TILE_S 8  // samples per chunk of the lane-per-cell form
#endif
#ifndef PV_RT60_TILE_NB
#define PV_RT60_TILE_NB 4  // chunks of loads in flight per wave
#endif
#ifndef PV_RT60_TILE_BLOCK
#define PV_RT60_TILE_BLOCK 256  // threads per workgroup of the lane-per-cell form
#endif

namespace pva {

namespace {

// The sixteen- and the four-lane form in ONE launch (round 6; they were two launches beside the lane-per-cell form's, two of the
// three leaving at once: 5-22 us each behind a run's stencil).  Which one runs is decided here, from the number of cells
// pv_onset_kernel found an onset in; both work through the list of 64-cell groups with work (AnalyzeArgs::unitList, like the
// encode pass beside them) instead of the window's rows and columns -- a closed room in a 4096^2 grid is ~100 groups of a
// 873 x 873-cell window, whose 12 000-48 000 workgroups took 57 us to find the 70 that had work.  The workgroups stride over the
// list: 16 cells x 16 lanes (a quarter of a group) or 64 cells x 4 lanes per turn.
__global__ __launch_bounds__(256) void pv_rt60_groups_kernel(const AnalyzeArgs a) {
    __shared__ double tab[96];
    if (analysisAborted(a)) return;
    // (the previous run's near box was read by pv_onset_kernel, a launch in front of this one: emptied here, it is the box of the
    // run after this one)
    if (a.prevBox && blockIdx.x == 0 && threadIdx.x == 0) {
        a.prevBox[0] = INT_MAX;
        a.prevBox[1] = INT_MAX;
        a.prevBox[2] = -1;
        a.prevBox[3] = -1;
    }
    const int lanes = rt60LanesPerCell(a, a.activeCount[1]);  // (grid-uniform)
    if (lanes == 1) return;  // (the lane-per-cell form's launch does the work)
    const DynParams dyn = *a.dyn;
    const int groups = a.activeCount[4];
    fillLogTab(tab, threadIdx.x, 256);
    __syncthreads();
    const LogTabLds ltab{tab};
    if (lanes == 16) {
        const int sub = threadIdx.x & 15;
        for (int q = blockIdx.x; q < 4 * groups; q += gridDim.x) {
            const long long g = (long long)a.unitList[q >> 2] * 64 + (q & 3) * 16 + (threadIdx.x >> 4);
            const PlaneCell pc = planeCell(a, dyn, g);
            const int s = pc.X * a.gy + pc.Y;
            const float d = pc.inGrid ? a.delay[s] : FLT_MAX;
            const bool live = d != FLT_MAX;
            // a wave skips only when none of its four cells has work: the DPP chains need whole rows, not whole waves, but
            // keeping the wave together costs nothing
            if (__ballot(live) == 0ull) continue;
            rt60WaveBody(a, ltab, sub, live, s, CellHistory{a.hist + g, a.histPlane}, (int)(live ? d : 0.f) + a.nDry + 1);
        }
        return;
    }
    const int sub = threadIdx.x & 3;
    for (int q = blockIdx.x; q < groups; q += gridDim.x) {
        const long long g = (long long)a.unitList[q] * 64 + (threadIdx.x >> 2);
        const PlaneCell pc = planeCell(a, dyn, g);
        const int s = pc.X * a.gy + pc.Y;
        const float d = pc.inGrid ? a.delay[s] : FLT_MAX;
        const bool live = d != FLT_MAX;
        if (__ballot(live) == 0ull) continue;
        rt60BlockedBody<4, 4>(a, ltab, sub, live, s, a.hist + g, (int)(live ? d : 0.f) + a.nDry + 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Lane-per-cell form over the TILE-MAJOR history (round 5): the form for everything above a few thousand cells.
// ---------------------------------------------------------------------------------------------------------------
// The blocked form above is bound by bytes it should not read (profiles/r04_analysis_pmc.md: 3 564 MB for a 1 328 MB
// history at 4.8 TB/s): its lanes run along WINDOW columns, so one load instruction touches four time planes x 16 floats,
// i.e. 64-byte pieces of 128-byte lines of tile rows that are 160-192 bytes long and not line-aligned -- the other halves
// arrive later, by other waves on other XCDs.  Here a lane's cell is its OFFSET g inside a history plane
// (plane[t][tile][row][col] is tile-major without padding, so hist + t * histPlane + g is the cell's sample t): the 64 lanes
// of a wave read 256 contiguous bytes of one plane per load instruction, every line is fetched by exactly one workgroup.
// One lane per cell also means no lane runs another lane's chain steps: per sample one multiply, three dependent adds and
// one logarithm (~45 instructions) instead of ~58 per sample and 64 lanes.
// Like pv_encode_kernel it reads the onset from the delay map (pv_onset_kernel) and nothing else of that kernel's, so the two
// run BESIDE each other on two streams -- the encode pass is bound by memory latency, this one by its instructions.
// Chunks of the backward pass (Analyzer.cpp:300-318) are of three wave-uniform kinds: in [endPoint, T) energy only; in [max of
// the wave's starting points, endPoint) every live lane regresses every sample -- no masks, branch-free logarithm for normal
// arguments (pvLog10fNormalT; a chunk in which some lane's running energy is zero, subnormal or not finite takes the general
// form); the rest masked per lane.  Lanes out of range load through an out-of-range buffer offset: the load returns 0
// without touching memory, and adding 0 * 0 leaves a non-negative sum as it is, bit for bit.  The wet gain of this form's
// cells is pv_encode_kernel's (a forward walk, like its dry gain).
template <int S, int NB>
__global__ __launch_bounds__(PV_RT60_TILE_BLOCK) void pv_rt60_tile_kernel(const AnalyzeArgs a) {
    __shared__ double tab[96];
    if (analysisAborted(a)) return;
    if (rt60LanesPerCell(a, a.activeCount[1]) != 1) return;  // (grid-uniform)
    fillLogTab(tab, threadIdx.x, PV_RT60_TILE_BLOCK);
    __syncthreads();
    const LogTabLds ltab{tab};
    const DynParams dyn = *a.dyn;
    const int T = a.T;
    constexpr int kOut = 0x7fffffff;  // >= every descriptor's extent: the load returns 0
    const long long plane = a.histPlane;
    const int planeBytes = (int)(plane * 4);

    // which cell: offset in a history plane -> window tile, row and column in the tile -> grid cell; its onset from the delay map
    const int unit = blockIdx.x * (PV_RT60_TILE_BLOCK / 64) + (int)(threadIdx.x >> 6);  // one wave per group of 64 cells with work
    if (unit >= a.activeCount[4]) return;
    const long long gl = (long long)a.unitList[unit] * 64 + (threadIdx.x & 63);
    const PlaneCell pc = planeCell(a, dyn, gl);
    const int cell = pc.X * a.gy + pc.Y;
    const float delay = pc.inGrid ? a.delay[cell] : FLT_MAX;
    const bool live = delay != FLT_MAX;
    if (__ballot(live) == 0ull) return;
    const int onset = live ? (int)delay : 0;
    const int voff = (int)gl * 4;
    auto planeRsrc = [&](int t) { return makeRsrc(a.hist + (long long)t * plane, planeBytes); };
    // a chunk's S planes through ONE descriptor (base = the chunk's lowest plane) and S scalar offsets that never change: a
    // descriptor per load is a 64-bit multiply-add and two merges on the scalar unit, 12 scalar instructions per load and a third
    // as many as the kernel's vector instructions (profiles/r05_rt60.txt).  The bounds check of a raw buffer is on vector +
    // scalar offset, so the descriptor's extent is the chunk's S planes, and lanes out of range still load through kOut
    // (S planes must stay below 2^31 bytes for that: rt60LanesPerCell).
    auto chunkRsrc = [&](int iTop) { return makeRsrc(a.hist + (long long)(iTop - S + 1) * plane, (long long)S * planeBytes); };
    auto chunkOff = [&](int k) { return (int)((unsigned)(S - 1 - k) * (unsigned)planeBytes); };

    const int endPoint = T - a.nCut;
    const int sp = onset + a.nDry + 1;  // startingPoint (live lanes)
    int spMax = live ? sp : INT_MIN, spMin = live ? sp : INT_MAX;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        spMax = max(spMax, __shfl_xor(spMax, off));
        spMin = min(spMin, __shfl_xor(spMin, off));
    }
    // (wave-uniform by value; said so to the compiler: the loops below then run on scalar counters and scalar descriptors)
    spMax = __builtin_amdgcn_readfirstlane(spMax);
    spMin = __builtin_amdgcn_readfirstlane(spMin);
    const int lvoff = live ? voff : kOut;
    // dead lanes carry an energy of 1 through the unmasked chunks (their loads return 0): a normal argument for the logarithm
    float edc = live ? 0.f : 1.f, xysum = 0.f, ysum = 0.f;

    // A software-pipelined loop over chunks of S samples, backwards from T - 1 (chunk c: samples i0 = T - 1 - c S ... i0 - S + 1).
    // NB chunks of loads are in flight per wave: a ring of NB x S registers, the loop unrolled NB times so that every slot is a
    // fixed set of registers, loads issued in consumption order everywhere (the compiler's wait-count bookkeeping joins
    // prologue and loop at the loop header; a reordered prologue made every pass through the header wait for ALL loads).
    // This many cells leave one or two waves per SIMD: with one chunk ahead the kernel waited a memory round trip per chunk
    // (2.1 TB/s, 0.68 ms where its instructions take 0.3: profiles/r05_rt60.txt).  Every chunk issues and consumes its S loads
    // whatever it is (the counts are the same on every path); what a chunk does with the energies is wave-uniform:
    //   entirely in [endPoint, T)      nothing (energy only);
    //   entirely in [spMax, endPoint)  every live lane regresses every sample: no masks, pvLog10fNormalT;
    //   otherwise                      masked per lane and sample, the general logarithm.
    const int lowest = live ? max(min(sp, endPoint), 0) : INT_MAX;  // this lane's energy integral covers [lowest, T)
    const int hiAll = max(min(spMax, endPoint), 0);                  // from here up every live lane is inside its integral
    const int lo = max(min(spMin, endPoint), 0);                     // nothing below is anybody's
    const int n1 = max(T - hiAll, 0) / S;                            // chunks that lie entirely in [hiAll, T): loads without masks
    int i0 = T - 1;                                                  // first sample of the chunk being consumed
    {
        float ring[NB][S];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int vo = (b < n1) ? lvoff : kOut;
            const rsrc_t rs = chunkRsrc(i0 - b * S);
#pragma unroll
            for (int k = 0; k < S; ++k) ring[b][k] = bufLoadF(rs, vo, chunkOff(k));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll 1
        for (int c0 = 0; c0 < n1; c0 += NB) {
#pragma unroll
            for (int b = 0; b < NB; ++b, i0 -= S) {
                float e[S];
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    edc = edc + ring[b][k] * ring[b][k];  // (+ 0 past the last chunk: those loads returned 0)
                    e[k] = edc;
                }
                {  // the slot's next occupant: chunk c0 + b + NB (past the last one: loads that return 0 untouched)
                    const int vo = (c0 + b + NB < n1) ? lvoff : kOut;
                    const rsrc_t rs = chunkRsrc(i0 - NB * S);
#pragma unroll
                    for (int k = 0; k < S; ++k) ring[b][k] = bufLoadF(rs, vo, chunkOff(k));
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (c0 + b < n1 && i0 < endPoint && i0 - S + 1 >= spMax) {
                    float y[S];
                    // the energy only grows: normal at both ends of the chunk = normal throughout (NaN fails the test)
#ifdef PV_RT60_PROBE_NOLOG  // timing probe only: what the kernel takes without its logarithms
                    if (true) {
#pragma unroll
                        for (int k = 0; k < S; ++k) y[k] = 10.f * e[k];
                    } else
#endif
                    if (__ballot(!(pvIsNormalPositive(e[0]) && pvIsNormalPositive(e[S - 1]))) == 0ull) {
                        pvLog10fNormalBatch(e, y, ltab);
#pragma unroll
                        for (int k = 0; k < S; ++k) y[k] = 10.f * y[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < S; ++k) y[k] = 10.f * pvLog10fNonNegT(e[k], ltab);
                    }
                    float xf = (float)(i0 - sp);  // (float)(i - startingPoint), counted down: exact (integers below 2^24)
#pragma unroll
                    for (int k = 0; k < S; ++k) {
                        xysum = xysum + y[k] * xf;
                        ysum = ysum + y[k];
                        xf = xf - 1.f;
                    }
                } else if (c0 + b < n1 && i0 - S + 1 < endPoint) {  // (the chunk that holds endPoint; late-onset waves)
#pragma unroll
                    for (int k = 0; k < S; ++k) {
                        const int ii = i0 - k;
                        const bool regress = live && ii >= sp && ii < endPoint;
                        const float y = 10.f * pvLog10fNonNegT(regress ? e[k] : 1.f, ltab);
                        xysum = regress ? xysum + y * (float)(ii - sp) : xysum;
                        ysum = regress ? ysum + y : ysum;
                    }
                }
            }
        }
        i0 = T - 1 - n1 * S;  // (the unrolled loop may have counted past the last chunk)
    }
    // ---- the rest, down to the wave's lowest starting point: masked per lane (a few chunks) ----
#pragma unroll 1
    for (; i0 >= lo; i0 -= S) {
        float p[S], e[S];
#pragma unroll
        for (int k = 0; k < S; ++k) p[k] = bufLoadF(planeRsrc(max(i0 - k, 0)), (i0 - k >= lowest) ? voff : kOut, 0);
#pragma unroll
        for (int k = 0; k < S; ++k) {
            edc = edc + p[k] * p[k];  // + 0 outside the lane's integral
            e[k] = edc;
        }
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const int ii = i0 - k;
            const bool regress = live && ii >= sp && ii < endPoint;
            const float y = 10.f * pvLog10fNonNegT(regress ? e[k] : 1.f, ltab);
            xysum = regress ? xysum + y * (float)(ii - sp) : xysum;
            ysum = regress ? ysum + y : ysum;
        }
    }
    // (wet gain, Analyzer.cpp:235-247: a forward walk like the dry gain's -- in this form pv_encode_kernel continues into it)
    if (live) a.out[2 * a.resN + cell] = rt60FromSums(a, sp, xysum, ysum);
}

}  // namespace

// Live module with two iterations in flight on two solvers (Solver::run's carryFrom): a cell in which this run found no onset
// keeps what the PREVIOUS iteration left there (the reference never touches m_results[s] then, Analyzer.cpp:160-165) -- the
// previous iteration ran on the other solver, so its six persistent planes are copied over for exactly those cells
namespace {
__global__ __launch_bounds__(256) void pv_carry_results_kernel(const AnalyzeArgs a, const float* __restrict__ src) {
    const DynParams dyn = *a.dyn;
    const int wc = blockIdx.x * blockDim.x + threadIdx.x, wr = blockIdx.y;
    if (wc >= a.winCols) return;
    const int X = dyn.histRow0 - a.G + wr, Y = dyn.histCol0 - a.G + wc;
    if (X >= a.gx || Y >= a.gy) return;
    const long long s = (long long)X * a.gy + Y;
    if (a.delay[s] != FLT_MAX) return;
    a.out[s] = src[s];
    a.out[a.resN + s] = src[a.resN + s];
    a.out[2 * a.resN + s] = src[2 * a.resN + s];
    a.out[3 * a.resN + s] = src[3 * a.resN + s];
    a.out[6 * a.resN + s] = src[6 * a.resN + s];
    a.out[7 * a.resN + s] = src[7 * a.resN + s];
}
}  // namespace

void launchCarryResults(const AnalyzeArgs& a, const float* srcOut, hipStream_t stream) {
    hipLaunchKernelGGL(pv_carry_results_kernel, dim3((a.winCols + 255) / 256, a.winRows), dim3(256), 0, stream, a, srcOut);
}

// One launch for the sixteen- and four-lane forms, one more for the lane-per-cell form where the caller expects it to be the one
// that runs (AnalyzeArgs::rt60Tile: the window can hold that many cells AND the previous run reached that many -- Solver::
// analyzeArgs): without it the four-lane form takes whatever number of cells there is.  A forced form (PVA_OPT_RT60_LANES)
// launches only itself.
void launchRt60Forms(const AnalyzeArgs& a, hipStream_t stream) {
    const long long groups = (a.histPlane + 63) / 64;
    // (the workgroups stride over the list: enough of them for either form's own range, kRt60WaveMaxCells / kRt60TileMinCells)
    hipLaunchKernelGGL(pv_rt60_groups_kernel, dim3((unsigned)std::min<long long>(4 * groups, a.rt60Lanes ? 8192 : 2048)), dim3(256), 0, stream, a);
    if (a.rt60Tile)
        hipLaunchKernelGGL((pv_rt60_tile_kernel<PV_RT60_TILE_S, PV_RT60_TILE_NB>),
                           dim3((unsigned)((a.histPlane + PV_RT60_TILE_BLOCK - 1) / PV_RT60_TILE_BLOCK)), dim3(PV_RT60_TILE_BLOCK), 0, stream, a);
}

}  // namespace pva
