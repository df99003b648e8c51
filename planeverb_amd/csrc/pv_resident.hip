// pv_resident.hip -- ONE launch per run for the grids the reference itself ships: its resolution presets on a 25 m scene
// (pv_LowResolution ... pv_ExtremeResolution = 275 ... 750 Hz, include/PvTypes.h:21-30 -> 70^2 ... 191^2 cells, T = 435 ... 1187)
// as the background loop runs them (Context/PvContext.cpp:63-94).
//
// Why a kernel of its own: at these sizes a K-step launch of the tile kernels (pv_kernels.hip) lasts ~13 us for ~4 us of
// arithmetic -- dispatch, three dependent scalar round trips before the first tile load, the load, the store, the end-of-kernel
// write-back -- and a run is a chain of T / K of them.  Here every tile of the grid is a 256-thread workgroup that stays
// RESIDENT for all T steps:
//   * tile = (RXI + 2K) x 64 cells, rows split over the block's 4 waves exactly as in the tile kernels' general arm
//     (stepTileGeneral4Packed: lanes along y, y-neighbours by DPP, the two boundary faces of vx through LDS, one s_barrier per
//     step); the face coefficients are looked up ONCE per run and stay in registers;
//   * every K steps (an "epoch") a block publishes its RXI x (64 - 2K) interior into the ping-pong planes with write-through
//     (sc1) stores, drains them, and raises ITS OWN flag word to the epoch number; it then polls the flags of its <= 8
//     neighbours (one relaxed sc1 load by 9 lanes of wave 0), and reloads tile + halo with sc1 loads (which bypass the CU's
//     L1; the per-XCD L2s are not coherent with each other, the guide's R1 hand-off: MI355X_MICROARCH.md, "Workgroup
//     dispatch, XCD placement & inter-workgroup visibility") -- no grid-wide barrier, no kernel boundary;
//   * two buffer sets suffice: a block overwrites set s (epoch e + 2) only after every neighbour has raised its flag to
//     e + 1, i.e. after that neighbour has finished reading epoch e from set s.
// ONE-XCD MODE (a.xcdMode, grids of up to 32 tiles: the Sandbox's 71^2, 95^2): the hand-off above is two dependent trips
// through the fabric (flag, then data) plus the producer's drain, ~3 us per epoch, because sc1 stores leave the XCD's L2.
// Inside ONE XCD the L2 is the coherence point: plain stores stay in it, L1-bypassing loads are served from it.  The launch
// then has 8 x ntiles blocks; every block reads the XCD it actually runs on (HW_REG_XCC_ID), the blocks on XCD
// a.xcdTarget claim the tiles through a counter, the others leave.  Correctness never depends on where the hardware put a
// block -- only blocks that ARE on the target XCD take part -- and if fewer than ntiles of them turn up there (another
// dispatch pattern or partition mode) the claim check below gives the run up within milliseconds (errFlag 4) and the host
// repeats it in the placement-independent mode.
// No assumption about dispatch order or workgroup -> XCD placement is made; all blocks must be co-resident (the host caps
// the grid and keeps a per-device budget), and every wait is bounded: a block that waits longer than ~2 s raises the abort
// word, every other block leaves at its next wait, and the run fails with an error instead of hanging the device.
//
// Arithmetic (bit-identical to the tile kernels' and to the oracle's fields, sign of zero aside as everywhere): a lone wave
// per SIMD issues one VALU instruction per ~5 cycles whatever its kind, so the step is written for the fewest instructions.
// In a run that starts from zero fields the pressure of every non-air cell (wall, ghost row / column, guard band) stays
// exactly 0 (FDTD.cpp:139: beta = 0), so ALL four face forms of FDTD.cpp:143-223 are one expression
//     v = a * v - c * (p_i - p_n)        air|air: a = 1, c = C          (v - C (p_i - p_n), FDTD.cpp:162-163)
//                                        wall(n)|air(i): a = 0, c = Y_n  (p_n = 0:  -Y_n p_i,  FDTD.cpp:165-168)
//                                        air(n)|wall(i): a = 0, c = Y_i  (p_i = 0:  +Y_i p_n);   wall|wall: a = c = 0
// (the absorbing grid edges, FDTD.cpp:201-223, are the two wall forms with Y = 1), evaluated as d = p_i - p_n, m = c * d,
// v = fma(a, v, -m): a * v is exact for a in {0, 1}, so the fused form rounds exactly once like the reference's subtract,
// and c * d is the reference's product (up to the sign both carry).  Likewise p -= Cb * div with Cb = beta * C.
// 15 packed instructions per row pair and step instead of 27 (mask-select form of the tile kernels' general arm).
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdio>
#include <cstdint>

#include "pv_device.h"
#include "pv_launch.h"
#include "pv_prims.h"

namespace pva {

namespace {

typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u3v __attribute__((ext_vector_type(3)));

constexpr int kSc1 = 16;  // buffer aux bit: sc1 = agent scope (write-through stores, L1-bypassing loads)
constexpr unsigned kSpinLimit = 1u << 21;  // polls (~1 us each with the s_sleep) before a block gives the run up

// W windows of R rows with stride R - 2 cover W (R - 2) + 2 >= RXI + 2K rows (cf. GenStackGeom in pv_kernels.hip)
template <int K, int RXI, int W>
struct ResGeom {
    static constexpr int L0 = RXI + 2 * K;
    static constexpr int R = (L0 - 2 + W - 1) / W + 2;
    static constexpr int L = W * (R - 2) + 2;
    static constexpr int D = L - L0;
    static constexpr int NP = (R + 1) / 2;
    static constexpr int WI = 64 - 2 * K;
};

template <int W>
struct ResShared {
    float xch[2][W][2][64];  // [step parity][wave][0: vx[1] for the wave above, 1: vx[R-2] for the wave below][lane]
    int abort;
};

// STRIDED pairing: row r of a wave's window lives in pair r % NP, component r / NP (rows 0 .. NP-1 in the .x halves, rows
// NP .. 2 NP - 1 in the .y halves).  The x-neighbour of BOTH halves of pair i is then pair i +- 1 itself, so the row
// differences of the pressure and vx sweeps are whole-pair subtracts; only the seam (pair NP-1 -> pair 0) costs a shuffle.
// (Adjacent-row pairs (2i, 2i+1), as in the tile kernels' general arm, pay a shuffle for every pair and sweep.)
template <int NP>
__device__ __forceinline__ float rowGet(const v2f (&f)[NP], int r) { return (r >= NP) ? f[r - NP].y : f[r].x; }
template <int NP>
__device__ __forceinline__ void rowSet(v2f (&f)[NP], int r, float val) {
    if (r >= NP) f[r - NP].y = val; else f[r].x = val;
}
__device__ __forceinline__ v2f selMask(const u2 m, const v2f airv, const v2f wallv) {  // v_bfi_b32 x 2
    const u2 ua = __builtin_bit_cast(u2, airv), uw = __builtin_bit_cast(u2, wallv);
    return __builtin_bit_cast(v2f, (ua & m) | (uw & ~m));
}

// rows wave WV stores: its own rows 1 .. R-2 that lie in the tile's interior [K, K + RXI) -- a compile-time range per wave
// index: straight-line stores (a runtime range costs a scalar branch per row, and a taken branch ~20 cycles of a step's ~800)
template <int K, int RXI, int W, int WV, int AUX, int NP>
__device__ __forceinline__ void storeRows(const v2f (&f)[NP], const rsrc_t rs, const int voff, const int soff0,
                                          const int pitchB) {
    using Gm = ResGeom<K, RXI, W>;
    constexpr int R = Gm::R;
    constexpr int ws = WV * (R - 2);
    constexpr int rLo = (K - ws > 1) ? K - ws : 1, rHi = (K + RXI - ws < R - 1) ? K + RXI - ws : R - 1;
#pragma unroll
    for (int r = 1; r < R - 1; ++r)
        if (r >= rLo && r < rHi)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rowGet(f, r)), rs, voff, soff0 + r * pitchB, AUX);
}

// The K (or fewer) steps of an epoch for wave WV of the block: the step loop holds no load, no wait for memory and -- with
// the wave index a template parameter -- no branch but the loop's own and the (rarely taken) pulse injection.
template <int K, int RXI, int W, int WV, int NP>
__device__ __forceinline__ void residentSteps(const ResidentArgs& a, ResShared<W>& sh, v2f (&pr)[NP], v2f (&vx)[NP], v2f (&vy)[NP],
                                              const v2f (&cb)[NP], const v2f (&ax)[NP], const v2f (&cx)[NP],
                                              const v2f (&ay)[NP], const v2f (&cy)[NP], const int nsteps, const int t0,
                                              const int lane, const float pvec, const bool pulseWave, const bool lAir,
                                              const int lr, const int lc, const bool inCols, const int hvoff, const int hsoff0) {
    using Gm = ResGeom<K, RXI, W>;
    constexpr int R = Gm::R, WI = Gm::WI;
    constexpr int hpitchB = WI * 4;
    constexpr bool first = WV == 0, last = WV == W - 1;
    float* hplane = a.hist + (long long)t0 * a.histPlane;
    // the two faces of vx this wave receives from its neighbours each step travel in loop-carried registers and are put in
    // place at the top of the NEXT step: their LDS reads then complete under the first dozen instructions of the pressure sweep
    float in0 = rowGet(vx, 0), inL = rowGet(vx, R - 1);
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        rowSet(vx, 0, in0);
        rowSet(vx, R - 1, inL);
        // pressure sweep, FDTD.cpp:124-141 (row R-1 holds no live pressure; what is computed there is never read);
        // breadth-first over the pairs, so that consecutive instructions of the lone wave are independent
        {
            v2f dx[NP], dy[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i)  // vx of rows r+1 (the seam: row NP is pair 0's .y; row 2 NP does not exist)
                dx[i] = (i + 1 < NP) ? vx[i + 1] : v2f{vx[0].y, 0.f};
#pragma unroll
            for (int i = 0; i < NP; ++i) dy[i] = v2f{laneNext(vy[i].x) - vy[i].x, laneNext(vy[i].y) - vy[i].y};
#pragma unroll
            for (int i = 0; i < NP; ++i) dx[i] = dx[i] - vx[i];
#pragma unroll
            for (int i = 0; i < NP; ++i) dx[i] = dx[i] + dy[i];
#pragma unroll
            for (int i = 0; i < NP; ++i) dx[i] = cb[i] * dx[i];
#pragma unroll
            for (int i = 0; i < NP; ++i) pr[i] = pr[i] - dx[i];
        }
        // vx sweep, FDTD.cpp:143-170 (+ edges :201-223 through the coefficients): own rows only (1 .. R-2); row 0's and
        // row R-1's faces come from the neighbouring waves, and whatever is computed for them here is overwritten
        {
            const float vx0keep = rowGet(vx, 0), vxLkeep = rowGet(vx, R - 1);
            v2f d[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i)  // pressure of rows r-1 (the seam: row NP-1 is pair NP-1's .x; row -1 does not exist)
                d[i] = (i > 0) ? pr[i - 1] : v2f{pr[0].x, pr[NP - 1].x};
#pragma unroll
            for (int i = 0; i < NP; ++i) d[i] = pr[i] - d[i];
#pragma unroll
            for (int i = 0; i < NP; ++i) d[i] = cx[i] * d[i];
#pragma unroll
            for (int i = 0; i < NP; ++i) vx[i] = __builtin_elementwise_fma(ax[i], vx[i], -d[i]);
            rowSet(vx, 0, vx0keep);
            rowSet(vx, R - 1, vxLkeep);
        }
        sh.xch[s & 1][WV][0][lane] = rowGet(vx, 1);
        sh.xch[s & 1][WV][1][lane] = rowGet(vx, R - 2);
        // (LDS only: the history stores stay in flight across the barrier.)  The reads are issued at once and land under
        // the vy sweep, which needs neither
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!first) in0 = sh.xch[s & 1][WV - 1][1][lane];
        if (!last) inL = sh.xch[s & 1][WV + 1][0][lane];
        // vy sweep, FDTD.cpp:172-199
        {
            v2f d[NP];
#pragma unroll
            for (int i = 0; i < NP; i += 2) {
                if (i + 1 < NP) {
                    subLanePrev2(pr[i], pr[i + 1], d[i], d[i + 1]);
                } else {
                    v2f unused;
                    subLanePrev2(pr[i], pr[i], d[i], unused);
                }
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) d[i] = cy[i] * d[i];
#pragma unroll
            for (int i = 0; i < NP; ++i) vy[i] = __builtin_elementwise_fma(ay[i], vy[i], -d[i]);
        }
        // record the pressure of this step before the pulse is injected (FDTD.cpp:226-234); the stores are never waited for
        if (inCols) storeRows<K, RXI, W, WV, 0>(pr, makeRsrc(hplane, a.histPlane * 4), hvoff, hsoff0, hpitchB);
        hplane += a.histPlane;
        // soft source: p[listener] += pulse[t], FDTD.cpp:234.  (The pulse underflows to exactly +0 after a few dozen samples:
        // adding it then changes no bit but a zero's sign.)  One add, found by wave-uniform branches.
        if (pulseWave) {
            const float pvCur = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pvec), s));
            if (pvCur != 0.f && (lAir || t0 + s == a.T - 1)) {
                const float pv = (lane == lc) ? pvCur : 0.f;
#pragma unroll
                for (int r = 0; r < R - 1; ++r)
                    if (r == lr) rowSet(pr, r, rowGet(pr, r) + pv);
            }
        }
    }
    rowSet(vx, 0, in0);
    rowSet(vx, R - 1, inL);
}

template <int K, int RXI, int W, int WV, int NP>
__device__ __forceinline__ void publishRows(const int wave, const ResidentArgs& a, const int set, const v2f (&pr)[NP],
                                            const v2f (&vx)[NP], const v2f (&vy)[NP], const int voff, const int soff0,
                                            const int pitchB) {
    if constexpr (WV < W) {
        if (wave == WV) {
            if (a.xcdMode) {  // (one XCD: plain stores stay in the L2 every taking-part block shares)
                storeRows<K, RXI, W, WV, 0>(pr, makeRsrc(a.pr[set], a.planeBytes), voff, soff0, pitchB);
                storeRows<K, RXI, W, WV, 0>(vx, makeRsrc(a.vx[set], a.planeBytes), voff, soff0, pitchB);
                storeRows<K, RXI, W, WV, 0>(vy, makeRsrc(a.vy[set], a.planeBytes), voff, soff0, pitchB);
            } else {
                storeRows<K, RXI, W, WV, kSc1>(pr, makeRsrc(a.pr[set], a.planeBytes), voff, soff0, pitchB);
                storeRows<K, RXI, W, WV, kSc1>(vx, makeRsrc(a.vx[set], a.planeBytes), voff, soff0, pitchB);
                storeRows<K, RXI, W, WV, kSc1>(vy, makeRsrc(a.vy[set], a.planeBytes), voff, soff0, pitchB);
            }
        } else {
            publishRows<K, RXI, W, WV + 1>(wave, a, set, pr, vx, vy, voff, soff0, pitchB);
        }
    }
}

template <int K, int RXI, int W, int WV, typename... Ts>
__device__ __forceinline__ void stepsOfWave(const int wave, Ts&&... args) {
    if constexpr (WV < W) {
        if (wave == WV)
            residentSteps<K, RXI, W, WV>(args...);
        else
            stepsOfWave<K, RXI, W, WV + 1>(wave, args...);
    }
}

#ifdef PV_RESIDENT_TRACE  // development builds (make EXTRA=-DPV_RESIDENT_TRACE): 100 MHz stamps of one block's phases
constexpr int kTraceEpochs = 48, kTracePhases = 6;
__device__ long long g_resTrace[8][kTraceEpochs][kTracePhases];
#define PV_STAMP(ph) \
    do { \
        if (tile == a.ntiles / 2 && e < kTraceEpochs && lane == 0) g_resTrace[wave][e][ph] = (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define PV_STAMP(ph) \
    do { \
    } while (0)
#endif

}  // namespace

template <int K, int RXI, int W>
__global__ __launch_bounds__(64 * W) void pv_resident_kernel(const ResidentArgs a) {
    using Gm = ResGeom<K, RXI, W>;
    constexpr int R = Gm::R, NP = Gm::NP, WI = Gm::WI;
    __shared__ ResShared<W> sh;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int tile = blockIdx.x;
    if (a.xcdMode) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;  // HW_REG_XCC_ID[3:0]
        if ((int)xcc != a.xcdTarget) return;
        if (threadIdx.x == 0) sh.abort = (int)atomicAdd(a.flags + a.ntiles + 1, 1u);  // claim a tile
        __syncthreads();
        tile = sh.abort;
        __syncthreads();
        if (tile >= a.ntiles) return;
        // every tile must have found a block on this XCD before anybody waits for a neighbour: ~2 ms, then give the run up
        if (wave == 0) {
            bool bad = false;
            for (unsigned spins = 0;; ++spins) {
                const unsigned got = __hip_atomic_load(a.flags + a.ntiles + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (got >= (unsigned)a.ntiles) break;
                if (spins > 2000u || __hip_atomic_load(a.flags + a.ntiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    if (lane == 0) {
                        __hip_atomic_store(a.flags + a.ntiles, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        atomicExch(a.errFlag, 4);
                    }
                    bad = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if (lane == 0) sh.abort = bad ? 1 : 0;
        }
        __syncthreads();
        if (sh.abort) return;
    }
    const int ti = tile / a.nty;
    const int tj = tile - ti * a.nty;
    const int ws = wave * (R - 2);  // first row of this wave's window, in loaded-tile rows
    const int row0 = a.G - K + ti * RXI + ws;
    const int col0 = a.G - K + tj * WI;
    const int voff = lane * 4;
    const int pitchB = a.pitch * 4;
    const int soff0 = (row0 * a.pitch + col0) * 4;

    // coefficients of this wave's cells (see the header): once per run.  row r of the window: rowGet / rowSet (strided pairing)
    const float C = a.courant;
    v2f pr[NP], vx[NP], vy[NP], cb[NP], ax[NP], cx[NP], ay[NP], cy[NP];
    {
        const rsrc_t rCoef = makeRsrc(a.coef, a.planeBytes * 3);
#pragma unroll
        for (int r = 0; r < 2 * NP; ++r) {
            float kxv = 0.f, kyv = 0.f, btv = 0.f;  // the spare row of an odd window: wall|wall, never read by a live row
            if (r < R) {
                const u3v c = __builtin_amdgcn_raw_buffer_load_b96(rCoef, lane * 12, 3 * (soff0 + r * pitchB), 0);
                kxv = __uint_as_float(c.x);
                kyv = __uint_as_float(c.y);
                btv = __uint_as_float(c.z);
            }
            const bool air = btv != 0.f, airX = kxv != kxv, airY = kyv != kyv;
            rowSet(cb, r, air ? C : 0.f);
            rowSet(ax, r, airX ? 1.f : 0.f);
            rowSet(cx, r, airX ? C : (air ? -kxv : kxv));
            rowSet(ay, r, airY ? 1.f : 0.f);
            rowSet(cy, r, airY ? C : (air ? -kyv : kyv));
        }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) pr[i] = vx[i] = vy[i] = v2f{0.f, 0.f};  // a run starts from zero fields, FDTD.cpp:109-119

    const DynParams dyn = a.dynVal;
    if (tile == 0 && threadIdx.x == 0) {  // (the block that HOLDS tile 0: in the one-XCD mode block 0 may have left)
        *a.dynOut = a.dynVal;
        if (a.stamp) a.stamp[0] = wall_clock64();
    }
    if (threadIdx.x == 0) a.tileFirst[tile] = INT_MAX;  // (only this block ever touches the entry during the run)
    // listener row inside this wave's window: rows 0..R-2 hold a live pressure (row 0 = the copy of the previous wave's
    // last row), row R-1 does not
    const int lr = dyn.lrow - row0;
    const int lc = dyn.lcol - col0;
    const bool hasL = lr >= 0 && lr <= R - 2 && lc >= 0 && lc < 64;
    // a listener inside a wall: the reference adds the pulse to a pressure that its next sweep zeroes before anything reads it
    // (FDTD.cpp:139,234) -- only the LAST sample survives, in the final field; here a wall cell's pressure must stay 0
    bool lAir = false;
#pragma unroll
    for (int r = 0; r < R - 1; ++r)
        if (r == lr) lAir = rowGet(cb, r) != 0.f;
    const int lrT = dyn.lrow - (row0 - ws);
    const bool tileHasL = lrT >= 0 && lrT < Gm::L && lc >= 0 && lc < 64;
    const int hti = ti - dyn.histTileX0, htj = tj - dyn.histTileY0;  // (the window is the whole grid: host precondition)
    const bool inCols = lane >= K && lane < 64 - K;
    constexpr int hpitchB = WI * 4;
    const int hsoff0 = ((hti * dyn.histTilesY + htj) * RXI - K + ws) * hpitchB;
    const int hvoff = (lane - K) * 4;
    bool seen = false;  // tileFirst[tile] already holds an epoch of this run

    // the <= 8 neighbours whose flags this block polls (lanes 0..8 of wave 0, lane 4 = itself); lane 9 watches the abort word
    const int ni = ti + lane / 3 - 1, nj = tj + lane % 3 - 1;
    const bool need = lane < 9 && lane != 4 && ni >= 0 && ni < a.ntx && nj >= 0 && nj < a.nty;
    unsigned* const myFlag = a.flags + tile;
    unsigned* const abortWord = a.flags + a.ntiles;
    unsigned* const pollWord = lane == 9 ? abortWord : (need ? a.flags + (ni * a.nty + nj) : myFlag);

    const int nEpochs = (a.T + K - 1) / K;
#pragma unroll 1
    for (int e = 0; e < nEpochs; ++e) {
        const int t0 = e * K;
        const int nsteps = min(K, a.T - t0);
        PV_STAMP(0);
        if (e > 0) {
            // ---- wait until every neighbour has published epoch e, then reload tile + halo (state after e * K steps) ----
            if (wave == 0) {
                bool bad = false;
                for (unsigned spins = 0;; ++spins) {
                    const unsigned v = __hip_atomic_load(pollWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__any(lane == 9 && v != 0u)) {
                        bad = true;
                        break;
                    }
                    if (__all(!need || v >= (unsigned)e)) break;
                    if (spins > kSpinLimit) {
                        if (lane == 0) {
                            __hip_atomic_store(abortWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            atomicExch(a.errFlag, 3);
                        }
                        bad = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (lane == 0) sh.abort = bad ? 1 : 0;
            }
            __syncthreads();
            if (sh.abort) return;  // (block-uniform)
            PV_STAMP(1);
            const int s = e & 1;
            const rsrc_t rPr = makeRsrc(a.pr[s], a.planeBytes), rVx = makeRsrc(a.vx[s], a.planeBytes),
                         rVy = makeRsrc(a.vy[s], a.planeBytes);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int so = soff0 + r * pitchB;
                rowSet(pr, r, __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rPr, voff, so, kSc1)));
                rowSet(vx, r, __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rVx, voff, so, kSc1)));
                rowSet(vy, r, __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rVy, voff, so, kSc1)));
            }
        }
        // tileFirst[tile] = first step block in which the tile (halo included) was non-zero; the analysis reads its
        // history from there on (earlier samples are the exact zeros they are)
        if (!seen) {
            uint32_t nz = 0;
#pragma unroll
            for (int i = 0; i < NP; ++i)
                nz |= (__float_as_uint(pr[i].x) | __float_as_uint(pr[i].y) | __float_as_uint(vx[i].x) |
                       __float_as_uint(vx[i].y) | __float_as_uint(vy[i].x) | __float_as_uint(vy[i].y)) & 0x7fffffffu;
            if (tileHasL || __ballot(nz != 0u) != 0ull) {
                seen = true;
                if (lane == 0) atomicMin(&a.tileFirst[tile], t0);
            }
        }

        // the epoch's pulse samples: lane s holds pulse[t0 + s] (ONE load per epoch; the step loop contains no load and no wait
        // for memory at all -- with a per-step pulse load every step ended in s_waitcnt vmcnt(0), i.e. waited ~1 us for its own
        // history stores to drain: that, not the launch, was most of the tile kernels' 13 us per 12-step launch)
        const float pvec = a.pulse[min(t0 + lane, a.T - 1)];
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): tile + pulse have landed; the loop's stores are never waited for
        PV_STAMP(2);
        stepsOfWave<K, RXI, W, 0>(wave, a, sh, pr, vx, vy, cb, ax, cx, ay, cy, nsteps, t0, lane, pvec, hasL, lAir, lr, lc, inCols,
                                  hvoff, hsoff0);

        PV_STAMP(3);
        // ---- publish the interior (state after (e + 1) * K steps) into the other buffer set, write-through ----
        if (inCols) publishRows<K, RXI, W, 0>(wave, a, (e + 1) & 1, pr, vx, vy, voff, soff0, pitchB);
        PV_STAMP(4);
        if (e + 1 < nEpochs) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains its write-through stores ...
            __syncthreads();                                   // ... before ONE lane raises the block's flag
            if (threadIdx.x == 0) {
                if (a.xcdMode)  // (a plain store: the flag stays in the XCD's L2, where the neighbours' L1-bypassing polls find it)
                    __hip_atomic_store(myFlag, (unsigned)(e + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else
                    __hip_atomic_store(myFlag, (unsigned)(e + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        PV_STAMP(5);
    }
}

#ifdef PV_RESIDENT_TRACE
void residentDumpTrace() {
    static long long h[8][kTraceEpochs][kTracePhases];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_resTrace), sizeof(h)) != hipSuccess) return;
    std::fprintf(stderr, "# resident trace (middle tile), us relative to the epoch's first stamp of wave 0: wait-done, loaded, steps-done, stores-issued, published | epoch length\n");
    for (int e = 1; e < kTraceEpochs; ++e) {
        if (!h[0][e][0]) break;
        for (int w = 0; w < 4; ++w) {
            std::fprintf(stderr, "e %2d w %d:", e, w);
            for (int ph = 0; ph < kTracePhases; ++ph) std::fprintf(stderr, " %6.2f", (h[w][e][ph] - h[0][e][0]) * 0.01);
            std::fprintf(stderr, " | %6.2f\n", (h[w][e][0] - h[w][e - 1][0]) * 0.01);
        }
    }
}
#endif

// instantiations: (K, RXI) of the launch-bound grids' default tile
#define PV_RESIDENT_CONFIGS(X) X(12, 12, 4)

bool residentConfigOk(int K, int rxi) {
#define X(k, r, w) \
    if (K == k && rxi == r) return true;
    PV_RESIDENT_CONFIGS(X)
#undef X
    return false;
}

int residentExtraRows(int K, int rxi) {
#define X(k, r, w) \
    if (K == k && rxi == r) return ResGeom<k, r, w>::D;
    PV_RESIDENT_CONFIGS(X)
#undef X
    return 0;
}

// blocks of this configuration that can be co-resident on the device (0 = unknown configuration)
int residentMaxBlocks(int K, int rxi, int device) {
    int perCu = 0;
#define X(k, r, w) \
    if (K == k && rxi == r) \
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, reinterpret_cast<const void*>(pv_resident_kernel<k, r, w>), 64 * w, 0);
    PV_RESIDENT_CONFIGS(X)
#undef X
    hipDeviceProp_t prop;
    if (perCu <= 0 || hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
    return perCu * prop.multiProcessorCount;
}

void launchResident(int K, int rxi, const ResidentArgs& a, hipStream_t stream) {
#define X(k, r, w) \
    if (K == k && rxi == r) { \
        hipLaunchKernelGGL((pv_resident_kernel<k, r, w>), dim3(a.xcdMode ? 8 * a.ntiles : a.ntiles), dim3(64 * w), 0, stream, a); \
        return; \
    }
    PV_RESIDENT_CONFIGS(X)
#undef X
}

}  // namespace pva
