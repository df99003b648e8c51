// pv_fused.hip -- the WHOLE impulse-response analysis of a small grid as one launch (round 5): an arm of the EXPERIMENTAL build
// (-DPV_EXPERIMENTAL, libplaneverb_amd_exp.so) -- built, bit-identical to the separate kernels, measured slower than them
// (docs/experiments/fused_analysis.md) and therefore not in the product library, which refuses PVA_OPT_FUSED_ANALYSIS = 1;
// its equivalence tests (tests/test_gpu_resident.py::test_fused_analysis_*) load the experimental build.  The run's LAST kernel
// (pv_run_finish_kernel, at the end of this file) is product code.
//
// The grids the reference ships (its resolution presets on a 25 m scene: 70^2 ... 254^2 cells, include/PvTypes.h:21-30) run
// their T stencil steps as ONE launch (pv_resident.hip), but the analysis behind it (Analyzer::AnalyzeResponses,
// Analyzer.cpp:48-104, straight behind GenerateResponse in the reference's loop, Context/PvContext.cpp:80-83) was a chain of
// 10-13 dependent launches -- far frame, onsets, encode, two decay-time forms, direction init, 3-4 jumps, final, output
// gather, status -- 5-8 us of launch gap each on 5 000-60 000 cells: 23-28 % of every live iteration (profiles/r04_presets.txt).
// Here it is one launch of WORKERS (workgroups of five waves) that draw ITEMS from a ticket counter:
//
//   phase 0   one item per group of CI window cells (CI = 16 with the sixteen-lane decay-time form, 64 with the four-lane one):
//             the onsets of the group (all five waves, parallel in time: pv_onset_kernel's search), then waves 0-3 the wet
//             gain / decay time of a quarter of the cells each (rt60WaveBody / rt60BlockedBody) while wave 4 walks the
//             dry window (encodeWave); no-onset cells get their "no onset" delay (and, two iterations in flight, the other
//             solver's record: Solver::run's carryFrom);
//   phase 1   listener direction, first hop of every window cell (dirInitCell);
//   phase 2.. kDirHops hops through the table per pass (dirJumpCell), dirJumpPasses(T) passes;
//   last      the walk's end -> direction (dirFinalCell).
//
// An item of phase k starts when ALL items of phase k - 1 are done (a counter per phase).  Tickets are handed out in phase
// order, so whoever waits, waits for items that running workers already hold: no deadlock whatever the number of resident
// workgroups, no assumption about dispatch order or placement, no co-residency budget.  What crosses workgroups inside the
// launch -- the delay and occlusion maps (phase 0 -> 1, last) and the hop table -- is written with agent-scope write-through
// stores, drained (s_waitcnt vmcnt(0)) before the phase counter is raised, and read with agent-scope loads: the hand-off
// form of the resident kernel (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility").
// Everything else is plain stores for the next launch.  The last worker to leave clears the counters for the next run.
// Bit-identical to the separate kernels: the per-cell bodies are the same functions (pv_analysis_dev.h).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>
#include <cstdint>
#include <cstdlib>

#include "pv_analysis.h"
#include "pv_analysis_dev.h"
#include "pv_device.h"
#include "pv_launch.h"
#include "pv_libm.h"
#include "pv_prims.h"

namespace pva {

#ifdef PV_EXPERIMENTAL
namespace {

constexpr int kFusedThreads = 320;      // waves 0-3: a quarter of the item's cells each (decay time), wave 4: their dry windows
constexpr int kFusedOnsetSC = 8;        // samples per lane and round of the onset search
constexpr unsigned kFusedSpinLimit = 1u << 20;  // polls (~1 us each with the s_sleep) before a worker gives the run up

struct FusedShared {
    double tab[96];  // log10f table of the four-lane decay-time form
    int found[64];   // onsets of the item's cells
    unsigned ticket;
    int active;      // cells of the window's ever-non-zero tiles
    int part[kFusedThreads];
};

// cells of the window's tiles that were ever non-zero (what pv_far_*_kernel's block 0 counts for the separate kernels): every
// worker computes it for itself -- it decides the decay-time form and with it the number of items
__device__ __forceinline__ int fusedActiveCells(const AnalyzeArgs& a, const DynParams& dyn, FusedShared& sh) {
    int n = 0;
    for (int i = threadIdx.x; i < dyn.histTilesX * dyn.histTilesY; i += kFusedThreads) {
        const int ti = dyn.histTileX0 + i / dyn.histTilesY, tj = dyn.histTileY0 + i % dyn.histTilesY;
        if (a.tileFirst[ti * a.nty + tj] < a.T) n += a.rxi * a.wi;
    }
    sh.part[threadIdx.x] = n;
    __syncthreads();
    if (threadIdx.x < 64) {
        int v = 0;
        for (int i = threadIdx.x; i < kFusedThreads; i += 64) v += sh.part[i];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off);
        if (threadIdx.x == 0) sh.active = v;
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(sh.active);  // (block-uniform; said so to the compiler: scalar branches below)
}

// phase 0, one group of CI cells (CI = 16: the 64 lanes of a wave are 16 cells x 4 time slots; CI = 64: 64 cells)
template <int CI>
__device__ __forceinline__ void fusedCells(const FusedArgs& f, const DynParams& dyn, FusedShared& sh, const int item) {
    const AnalyzeArgs& a = f.a;
    constexpr int Q = 64 / CI;  // time slots per wave of the onset search
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int T = a.T;
    const int ci = lane % CI, q = lane / CI;
    const long long g = (long long)item * CI + ci;
    const PlaneCell c = planeCell(a, dyn, g);
    int tF = INT_MAX;
    bool air = c.inGrid;
    if (air) {
        tF = a.tileFirst[c.tile];
        // never reached by the pulse, or a wall (beta = 0: pr is identically zero, FDTD.cpp:139): no onset
        air = tF < T && a.coef[(size_t)(c.X + a.G) * a.pitch + (c.Y + a.G)].beta != 0.f;
    }
    const bool airCell = air;  // (counts as silent without an onset)
    if (a.labels) {  // ... nor a cell that no chain of air cells joins to the listener's (AnalyzeArgs::labels): not scanned
        const int lX = dyn.lrow - a.G, lY = dyn.lcol - a.G;
        const int mineL = air ? a.labels[(size_t)c.X * a.labelNY + c.Y] : -1;
        const int theirs = (lX >= 0 && lX <= a.gx && lY >= 0 && lY < a.labelNY) ? a.labels[(size_t)lX * a.labelNY + lY] : -2;
        air = air && mineL == theirs;
    }
    if (threadIdx.x < 64) sh.found[threadIdx.x] = INT_MAX;
    __syncthreads();

    // ---- onsets (Analyzer.cpp:146-165), parallel in time: pv_onset_kernel's search with 5 Q slots of SC samples per round ----
    if (__ballot(air) != 0ull) {  // (the same lanes in every wave of the block: block-uniform)
        const int m = abs(c.X - (dyn.lrow - a.G)) + abs(c.Y - (dyn.lcol - a.G));
        const int tS = air ? min(max(tF, m), T) : INT_MAX;
        const int tBeg = waveMin(tS);
        constexpr int SC = kFusedOnsetSC, W = kFusedThreads / 64;
#pragma unroll 1
        for (int t0 = tBeg; t0 < T; t0 += W * Q * SC) {
            const int best = sh.found[ci];
            const bool go = __ballot(air && best > t0) != 0ull;  // else: nothing at or after t0 can be the first
            __syncthreads();  // every wave has read the table before anybody writes it again: `go` is the same in all of them
            if (!go) break;
            const int t = t0 + (wave * Q + q) * SC;
            float pc[SC];
#pragma unroll
            for (int k = 0; k < SC; ++k) {
                const int tt = t + k;
                const bool want = air && tt >= tS && tt < T && tt < best;  // (a tile's history starts at its first recorded step)
                pc[k] = want ? a.hist[(long long)tt * a.histPlane + g] : 0.f;
            }
            int hit = INT_MAX;
#pragma unroll
            for (int k = SC - 1; k >= 0; --k) hit = fabsf(pc[k]) > kAudibleThresholdDev ? t + k : hit;
            if (hit != INT_MAX) atomicMin(&sh.found[ci], hit);
            __syncthreads();  // the round's hits are in
        }
    }
    __syncthreads();
#ifdef PV_FUSED_DEBUG
    unsigned long long* const dbg = reinterpret_cast<unsigned long long*>(f.ctl + kFusedCtlWords);
    if (threadIdx.x == 0) atomicMax(dbg + 22, (unsigned long long)wall_clock64());
#endif
    const int onset = sh.found[ci];
    const bool live = air && onset != INT_MAX;
    const int s = c.X * a.gy + c.Y;

    if (wave == 4) {
        // ---- the cells' records: no onset -> "no onset" (and the other solver's record); onset -> dry window, wet gain apart ----
        const bool mine = q == 0;  // (CI = 16: lanes 16 .. 63 repeat the cells)
        if (mine && c.inGrid) {
            xStoreF<true>(a.delay + s, live ? (float)onset : FLT_MAX);
            if (!live && f.carrySrc) {  // Analyzer.cpp:160-165: the record of the previous iteration stays
                const float* src = f.carrySrc;
                xStoreF<true>(a.out + s, src[s]);
                a.out[a.resN + s] = src[a.resN + s];
                a.out[2 * a.resN + s] = src[2 * a.resN + s];
                a.out[3 * a.resN + s] = src[3 * a.resN + s];
                a.out[6 * a.resN + s] = src[6 * a.resN + s];
                a.out[7 * a.resN + s] = src[7 * a.resN + s];
            }
        }
        const unsigned long long mr = __ballot(mine && live), ms = __ballot(mine && airCell && !live);
        if (lane == 0) {
            if (mr) atomicAdd(a.activeCount + 1, __popcll(mr));
            if (ms) atomicAdd(a.activeCount + 3, __popcll(ms));
        }
#ifndef PV_FUSED_NO_ENCODE
        if (mr != 0ull) encodeWave<true>(a, dyn, c, mine && live, onset, false);
#endif
#ifdef PV_FUSED_DEBUG
        if (lane == 0) atomicMax(dbg + 24, (unsigned long long)wall_clock64());
#endif
    } else {
        // ---- wet gain + decay time: wave w takes cells [w CI / 4, (w + 1) CI / 4) with L = 64 / (CI / 4) lanes each ----
        constexpr int L = 256 / CI;  // 16 (CI = 16) or 4 (CI = 64)
        const int cj = wave * (CI / 4) + lane / L, sub = lane % L;
        const long long gj = (long long)item * CI + cj;
        const PlaneCell cc = planeCell(a, dyn, gj);
        const int on = sh.found[cj];
        // (found[] holds an onset only for air cells inside the grid)
        const bool lv = cc.inGrid && on != INT_MAX;
#ifdef PV_FUSED_NO_RT60
        if (false) {
#else
        if (__ballot(lv) != 0ull) {
#endif
            const int sj = cc.X * a.gy + cc.Y;
            const float* h0 = a.hist + gj;
            if constexpr (L == 16)
                rt60WaveBody(a, LogTabLds{sh.tab}, sub, lv, sj, CellHistory{h0, a.histPlane}, on + a.nDry + 1);
            else
                rt60BlockedBody<4, 4>(a, LogTabLds{sh.tab}, sub, lv, sj, h0, on + a.nDry + 1);
        }
#ifdef PV_FUSED_DEBUG
        if (lane == 0) atomicMax(dbg + 26, (unsigned long long)wall_clock64());
#endif
    }
}

}  // namespace

__global__ __launch_bounds__(kFusedThreads) void pv_analysis_fused_kernel(const FusedArgs f) {
    __shared__ FusedShared sh;
    const AnalyzeArgs& a = f.a;
    unsigned* const ctl = f.ctl;
    // (a run whose stencil was given up -- resident kernel, errFlag 3 / 4 -- has a half-written history: nothing to analyse, as in
    // every separate kernel.  Grid-uniform, and no ticket has been drawn: the control words stay at zero.)
    if (analysisAborted(a)) return;
    if (a.stamp && blockIdx.x == 0 && threadIdx.x == 0) a.stamp[1] = wall_clock64();
    const DynParams dyn = *a.dyn;
    fillLogTab(sh.tab, threadIdx.x, kFusedThreads);
    const int active = fusedActiveCells(a, dyn, sh);  // (synchronises: the table is in place)
    const int lanes = rt60LanesPerCell(a, active) == 16 ? 16 : 4;
    const int CI = lanes == 16 ? 16 : 64;
    const int nB = (int)((a.histPlane + CI - 1) / CI);
    const int winCells = a.winRows * a.winCols;
    const int nD = (winCells + kFusedThreads - 1) / kFusedThreads;
    const int passes = dirJumpPasses(a.T);
    const int nPhases = 1 + 1 + passes + 1;
    const unsigned total = (unsigned)nB + (unsigned)nD * (unsigned)(nPhases - 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.activeCount[0] = active;

#ifdef PV_FUSED_DEBUG
    if (threadIdx.x == 0) atomicMin(reinterpret_cast<unsigned long long*>(ctl + kFusedCtlWords), (unsigned long long)wall_clock64());
#endif
    // (One single-lane region per turn -- the previous item's phase counter and the next ticket together -- closed by a barrier
    // before anything else happens: with the counter raised at the END of a turn and the ticket drawn at the START of the next,
    // the compiler joined the two lane-0 regions across the loop's back edge into a loop of their own around the rest, and the
    // wave that holds thread 0 left the kernel while the other four went on taking the same ticket from LDS for ever.)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int prevPhase = -1;
    for (;;) {
        if (wave == 0) {
            unsigned v = 0;
            if (lane == 0) {
                if (prevPhase >= 0) __hip_atomic_fetch_add(ctl + 1 + prevPhase, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sh.ticket = v;
            }
        }
        __syncthreads();
        // (block-uniform, as a scalar: the phase branches and the barriers inside them are then not 'divergent' to the compiler)
        const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)sh.ticket);
        if (t >= total) break;
        const int phase = t < (unsigned)nB ? 0 : 1 + (int)((t - (unsigned)nB) / (unsigned)nD);
        const int item = phase == 0 ? (int)t : (int)((t - (unsigned)nB) % (unsigned)nD);
        if (phase > 0) {
            // every item of the previous phase is done (all of them are held by running workers: see the header)
            if (threadIdx.x == 0) {
                const unsigned need = phase == 1 ? (unsigned)nB : (unsigned)nD;
                unsigned spins = 0;
                while (__hip_atomic_load(ctl + phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                    // (a worker died: give the run up instead of hanging the device; once one waiter has, all do)
                    if (++spins > kFusedSpinLimit ||
                        ((spins & 1023u) == 0u && __hip_atomic_load(f.errFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 6)) {
                        atomicExch(f.errFlag, 6);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
        }
#ifdef PV_FUSED_DEBUG
        // development build: 100 MHz stamps -- dbg[0] = first worker in, dbg[1 + 2 p] / dbg[2 + 2 p] = first item of phase p begun /
        // last one ended (its wait included in "begun" - previous "ended")
        unsigned long long* const dbg = reinterpret_cast<unsigned long long*>(ctl + kFusedCtlWords);
        if (threadIdx.x == 0) atomicMin(dbg + 1 + 2 * phase, (unsigned long long)wall_clock64());
#endif
        if (phase == 0) {
            if (CI == 16)
                fusedCells<16>(f, dyn, sh, item);
            else
                fusedCells<64>(f, dyn, sh, item);
        } else {
            const int w = item * kFusedThreads + (int)threadIdx.x;
            if (w < winCells) {
                const int wr = w / a.winCols, wc = w - wr * a.winCols;
                const int X = dyn.histRow0 - a.G + wr, Y = dyn.histCol0 - a.G + wc;
                if (X < a.gx && Y < a.gy) {
                    const int p = X * a.gy + Y;
                    if (phase == 1)
                        dirInitCell<true>(a, dyn, a.dirScratch, p);
                    else if (phase < nPhases - 1)
                        dirJumpCell<true>(a, dyn, a.dirScratch, p);
                    else
                        dirFinalCell<true>(a, dyn, a.dirScratch, p);
                }
            }
        }
#ifdef PV_FUSED_DEBUG
        if (threadIdx.x == 0) atomicMax(dbg + 2 + 2 * phase, (unsigned long long)wall_clock64());
#endif
        // the item's write-through stores have left this CU before its phase counter moves
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // (also: every wave has read this turn's ticket before the next one is written)
        prevPhase = phase;
    }
    // the last worker to leave clears the counters: the next run finds them at zero
    if (threadIdx.x == 0) {
        const unsigned gone = __hip_atomic_fetch_add(ctl + kFusedCtlWords - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (after a run that was given up the words stay as they are, for Solver::sync's message; the host clears them)
        if (gone == gridDim.x - 1 && __hip_atomic_load(f.errFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 6)
            for (int i = 0; i < kFusedCtlWords; ++i) __hip_atomic_store(ctl + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

bool fusedAnalysisBuilt() { return true; }

bool fusedAnalysisOk(const AnalyzeArgs& a) {
    // phase counters: 1 ticket + (2 + passes + 1) phases + 1 exit word
    return 1 + 2 + dirJumpPasses(a.T) + 1 + 1 <= kFusedCtlWords;
}

void launchAnalysisFused(const FusedArgs& f, hipStream_t stream) {
    // enough workers to hold every item of the widest phase at once where the chip has room for them (tickets make any number
    // correct); a worker is five waves.  PLANEVERB_AMD_FUSED_WORKERS caps the number (the equivalence tests run the launch with
    // 1, 2 and 3 workers: "no deadlock whatever the number of resident workgroups").
    const long long items = (f.a.histPlane + 15) / 16;
    unsigned grid = (unsigned)std::min<long long>(std::max<long long>(items, 64), 1024);
    if (const char* e = std::getenv("PLANEVERB_AMD_FUSED_WORKERS"))
        if (std::atoi(e) > 0) grid = std::min(grid, (unsigned)std::atoi(e));
    hipLaunchKernelGGL(pv_analysis_fused_kernel, dim3(grid), dim3(kFusedThreads), 0, stream, f);
}
#else   // product build: the arm is not compiled in (Solver::init refuses the option)
bool fusedAnalysisBuilt() { return false; }
bool fusedAnalysisOk(const AnalyzeArgs&) { return false; }
void launchAnalysisFused(const FusedArgs&, hipStream_t) {}
#endif  // PV_EXPERIMENTAL

// ---------------------------------------------------------------------------------------------------------------
// last kernel of a run: the registered output queries (pv_gather_queries_kernel) and the status words (pv_run_status_kernel)
// in ONE launch, both into pinned host memory
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(512) void pv_run_finish_kernel(const float* __restrict__ res, long long n, const long long* cells, int nq,
                                                            float* out, const FarInfo f, int* err, int* counts,
                                                            const unsigned* claims, int* status, unsigned* zeroWords, int nZero,
                                                            unsigned long long* stamp) {
    if (stamp && threadIdx.x == 0) stamp[2] = wall_clock64();  // (the run's device work ends here: Solver::stampTimed_)
    const int q = threadIdx.x >> 3, k = threadIdx.x & 7;
    if (q < nq) {
        const long long c = cells[q];
        float v = c >= 0 ? res[k * n + c] : 0.f;
        if (c >= 0 && (k == 4 || k == 5) && isFarCell(f, c)) {  // a far cell: its direction in closed form
            float x, y;
            farDirectionOf(f, c, &x, &y);
            v = k == 4 ? x : y;
        }
        out[q * 8 + k] = v;
    }
    const int claimed = claims ? (int)*claims : -1;  // (before the words are cleared below: every thread reads, thread 0 uses)
    __syncthreads();
    // the resident kernel's flag words and the error flag start the NEXT run at zero (no begin-run launch in front of that kernel)
    for (int i = threadIdx.x; i < nZero; i += 512) zeroWords[i] = 0u;
    if (threadIdx.x == 0) {
        status[0] = *err;
        if (nZero > 0) *err = 0;
        status[1] = counts[0];
        status[2] = counts[1];
        status[3] = claimed;
        status[4] = counts[3];
        // (the analysis adds to the cell counters and the list of groups with work: they start every run at zero)
        counts[1] = 0;
        counts[2] = 0;
        counts[3] = 0;
        counts[4] = 0;
    }
}
}  // namespace

void launchRunFinish(const float* res, long long n, const long long* cellsHost, int nq, float* outHost, const FarInfo& far,
                     int* err, int* counts, const unsigned* claims, int* statusHost, unsigned* zeroWords, int nZero,
                     unsigned long long* stamp, hipStream_t stream) {
    hipLaunchKernelGGL(pv_run_finish_kernel, dim3(1), dim3(512), 0, stream, res, n, cellsHost, nq, outHost, far, err, counts, claims,
                       statusHost, zeroWords, nZero, stamp);
}

}  // namespace pva
