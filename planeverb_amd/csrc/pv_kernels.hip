// pv_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4): the Planeverb FDTD leapfrog stencil with
// K time steps fused per launch, pr-history recording, and the per-cell impulse-response analysis.
//
// Reference semantics implemented here (paths relative to /root/reference/ProjectPlaneverb):
//   pv_step_kernel      src/FDTD/FDTD.cpp:122-235   pressure / vx / vy sweeps, edge absorption, record, pulse
//   pv_coef_kernel      src/FDTD/FDTD.cpp:143-223 + src/FDTD/Grid.cpp:88-108   (beta, Y, edges -> face coefficients)
//   pv_encode_kernel    src/DSP/Analyzer.cpp:139-328  onset, dry gain, source direction, lowpass, wet, RT60
//   pv_direction_kernel src/DSP/Analyzer.cpp:340-431  listener direction by delay-map descent
//   pv_efree_kernel     src/FDTD/FreeGrid.cpp:96-110  free-field energy sum
//   pv_ir_kernel        src/FDTD/FDTD.cpp:60-79       impulse response (pr, vx, vy) of one cell
//
// Design (MI355X-first, not a translation of the reference's three AoS sweeps):
//   * memory-bound 5-point staggered stencil, 24 B of state per cell-step; no MFMA.
//   * one WAVE owns one (RXI+2K) x 64 tile held entirely in VGPRs (SoA pr/vx/vy, one cell column per lane, lanes
//     along the contiguous y axis); x-neighbours are the lane's own registers, y-neighbours come from lane+-1 by
//     DPP wave shifts.  No LDS traffic for the fields, no barriers inside the time loop.
//   * K leapfrog steps are advanced per launch on the tile+halo (overlapped temporal tiling): HBM traffic per
//     cell-step drops from 24 B to about (12*(1+halo)+12)/K B.
//   * beta, wall admittance and the absorbing grid edges are folded into one uint16 face-code per cell that indexes
//     a 256-entry coefficient LUT in LDS; tiles whose faces are all air|air take a branch-free fast path.
//   * float32 arithmetic in the reference's operation order; compiled with -ffp-contract=off so that
//     v - C*(p_i - p_n) stays a separate multiply and subtract (bit-identical fields).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <climits>
#include <cstdint>

#include "pv_device.h"
#include "pv_launch.h"
#include "pv_libm.h"
#include "pv_prims.h"
#include "pv_analysis.h"
#include "pv_analysis_dev.h"

namespace pva {

// self-test of the lane-shift primitives: out[i] = {laneNext(i), lanePrev(i)}
__global__ void pv_lane_selftest_kernel(float* out) {
    const int lane = threadIdx.x;
    out[lane] = laneNext((float)lane);
    out[64 + lane] = lanePrev((float)lane);
}

// ---------------------------------------------------------------------------------------------------------------
// face coefficients
// ---------------------------------------------------------------------------------------------------------------

// mat: (gx+1)*(gy+1) floats: NaN = air cell (beta = 1, Grid.cpp:88-108,229-246), else the admittance Y = (1 - R) / (1 + R) of
// the wall cell (FDTD.cpp:150,156; computed on the host in the reference's float arithmetic).
// One thread per padded cell.  Folds FDTD.cpp:143-223 into one coefficient per face (FaceCoef, pv_device.h):
//   air|air   : v = v - C*(p_i - p_n)                                   (FDTD.cpp:162-163, beta*beta_n = 1)
//   wall(n)|air(i): v = -Y_n * p_i ; air(n)|wall(i): v = +Y_i * p_n      (FDTD.cpp:165-168)
//   grid edges: vx[0,y] = -p[0,y], vx[gx,y] = p[gx-1,y], vy likewise     (FDTD.cpp:201-223)
__global__ void pv_coef_kernel(const float* __restrict__ mat, FaceCoef* __restrict__ coef, Geometry g) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (col >= g.pitch || row >= g.rows) return;
    // (x, y) in the WHOLE grid's cell array: a slab's guard rows hold its neighbours' true face coefficients
    const int x = row - g.G + g.x0, y = col - g.G;
    const float air = __uint_as_float(kAirFaceBits);
    float kx = 0.f, ky = 0.f;  // wall|wall
    bool bi = false;
    if (x >= 0 && x < g.NXg && y >= 0 && y < g.NY) {
        const bool ghost = (x == g.gxg) || (y == g.gy);
        const float mi = mat[(size_t)x * g.NY + y];
        bi = (mi != mi) && !ghost;
        const float Yi = (ghost || mi != mi) ? 1.f : mi;  // (the ghost row / column: R = 0)
        // x face: neighbour n = (x-1, y)
        if (x == 0) {
            kx = (bi && y < g.gy) ? -1.f : 0.f;
        } else if (x == g.gxg) {
            kx = (y < g.gy) ? 1.f : 0.f;
        } else {
            const float mn = mat[(size_t)(x - 1) * g.NY + y];
            const bool bn = (mn != mn) && (y != g.gy);
            const float Yn = (y == g.gy || mn != mn) ? 1.f : mn;
            kx = (bi && bn) ? air : bi ? -Yn : bn ? Yi : 0.f;
        }
        // y face: neighbour n = (x, y-1)
        if (y == 0) {
            ky = (bi && x < g.gxg) ? -1.f : 0.f;
        } else if (y == g.gy) {
            ky = (x < g.gxg) ? 1.f : 0.f;
        } else {
            const float mn = mat[(size_t)x * g.NY + (y - 1)];
            const bool bn = (mn != mn) && (x != g.gxg);
            const float Yn = (x == g.gxg || mn != mn) ? 1.f : mn;
            ky = (bi && bn) ? air : bi ? -Yn : bn ? Yi : 0.f;
        }
    }
    coef[(size_t)row * g.pitch + col] = FaceCoef{kx, ky, bi ? 1.f : 0.f};
}

// ---------------------------------------------------------------------------------------------------------------
// fused K-step stencil
// ---------------------------------------------------------------------------------------------------------------

// air rows only (every face air|air): FDTD.cpp:124-199 without coefficients
template <int ROWS>
__device__ __forceinline__ void leapfrogStep(float (&pr)[ROWS], float (&vx)[ROWS], float (&vy)[ROWS], const float C) {
    // pressure sweep, FDTD.cpp:124-141:  p = p - C * ((vx[x+1] - vx[x]) + (vy[y+1] - vy[y]))
#pragma unroll
    for (int r = 0; r < ROWS - 1; ++r) {
        const float vyR = laneNext(vy[r]);
        const float div = (vx[r + 1] - vx[r]) + (vyR - vy[r]);
        pr[r] = pr[r] - C * div;
    }
    // velocity sweeps, FDTD.cpp:143-199
#pragma unroll
    for (int r = ROWS - 1; r >= 1; --r) {
        const float pi = pr[r], pn = pr[r - 1];
        vx[r] = vx[r] - C * (pi - pn);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const float pi = pr[r];
        const float pn = lanePrev(pi);
        vy[r] = vy[r] - C * (pi - pn);
    }
}

// General tiles, coefficients in registers.  kx[r] / ky[r] are the face coefficients of this lane's cell in row r
// (NaN = air|air face, else v = k * (p_i + p_n), which covers walls, wall|wall (k = 0) and the absorbing grid edges),
// bt[r] = beta of the cell as 0.f / 1.f -- all looked up ONCE per launch.  The first version re-read the two LUT
// entries of every cell on every step (41 instructions per row and step, 6x an air row); with the coefficients
// resident a row costs 20.  pr = beta * (...) is the reference's own expression (FDTD.cpp:139).
template <int ROWS>
__device__ __forceinline__ void leapfrogStepCoef(float (&pr)[ROWS], float (&vx)[ROWS], float (&vy)[ROWS],
                                                 const float (&kx)[ROWS], const float (&ky)[ROWS],
                                                 const float (&bt)[ROWS], const float C) {
    // pressure sweep, FDTD.cpp:124-141
#pragma unroll
    for (int r = 0; r < ROWS - 1; ++r) {
        const float vyR = laneNext(vy[r]);
        const float div = (vx[r + 1] - vx[r]) + (vyR - vy[r]);
        pr[r] = bt[r] * (pr[r] - C * div);
    }
    // velocity sweeps, FDTD.cpp:143-199 (+ edges :201-223 through the coefficients)
#pragma unroll
    for (int r = ROWS - 1; r >= 1; --r) {
        const float pi = pr[r], pn = pr[r - 1];
        const float air = vx[r] - C * (pi - pn);
        const float wall = kx[r] * (pi + pn);
        vx[r] = (kx[r] != kx[r]) ? air : wall;
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const float pi = pr[r];
        const float pn = lanePrev(pi);
        const float air = vy[r] - C * (pi - pn);
        const float wall = ky[r] * (pi + pn);
        vy[r] = (ky[r] != ky[r]) ? air : wall;
    }
}

// face coefficients + beta of this lane's cell in N rows of a general tile (row r at byte offset soff0 + r * pitchB of a float
// plane: the FaceCoef plane has three times that pitch): ONE independent 12-byte load per row, issued with the field loads
template <int N>
__device__ __forceinline__ void loadFaceCoefs(const StepArgs& a, const int lane, const int soff0, const int pitchB,
                                              float (&kx)[N], float (&ky)[N], float (&bt)[N]) {
    typedef unsigned int u3v __attribute__((ext_vector_type(3)));
    const rsrc_t rCoef = makeRsrc(a.coef, a.planeBytes * 3);
#pragma unroll
    for (int r = 0; r < N; ++r) {
        const u3v c = __builtin_amdgcn_raw_buffer_load_b96(rCoef, lane * 12, 3 * (soff0 + r * pitchB), 0);
        kx[r] = __uint_as_float(c.x);
        ky[r] = __uint_as_float(c.y);
        bt[r] = __uint_as_float(c.z);
    }
}


// The pulse samples of a launch: lane s holds pulse[t0 + s] (ONE load, issued with the tile loads); step s takes its sample
// with v_readlane.  A load inside the step loop put an s_waitcnt vmcnt(0) behind it, i.e. the wave that holds the listener
// waited for ITS OWN history stores to drain on every step (~0.5 us each) -- and a launch-bound grid's launch lasts as long
// as its slowest tile (found with the resident kernel's phase stamps, round 4).  Used by the launch-bound grids' tile only
// (12 steps x 12 rows: the presets' replayed graph 10 % faster at 70^2, 5 % at 191^2); in the other instantiations the
// change moved the air arm's code (512^2 with four runs in flight 4 % slower) and is not what bounds them.
constexpr bool pulseLanesConfig(int K, int RXI) { return K == 12 && RXI == 12; }
__device__ __forceinline__ float loadPulseLanes(const StepArgs& a, const int lane) {
    return a.pulse[a.t0 + min(lane, a.nsteps - 1)];
}
__device__ __forceinline__ float pulseOfStep(const float pvec, const int s) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pvec), s));
}

// Streaming-analysis mode: a tile's pressure history is only consumed while one of its cells -- or a cell of the
// tile below / to the right, whose velocity reconstruction reads this tile's last row / column -- still has an open
// forward-analysis window, or while it holds a registered emitter.  Once all of that is closed (N_dry samples after
// the wave front passed), the tile stops writing history planes: in a 25 m scene at 4096^2 that is most tiles for
// most of the 25 432 steps.
__device__ __forceinline__ bool historyWanted(const StepArgs& a, int ti, int tj) {
    if (!a.tileOpen) return true;
    const int t = ti * a.nty + tj;
    if (a.tileOpen[t]) return true;
    if (ti + 1 < a.ntx && a.tileOpen[t + a.nty]) return true;
    if (tj + 1 < a.nty && a.tileOpen[t + 1]) return true;
    return false;
}

// Scalar tile body of the two-kernel form (pv_step_air_kernel with packed math off, pv_step_general_kernel).
// GENERAL = false: every face of the tile (halo included) is air|air and the listener is not inside it -- no
// coefficient reads, no pulse.  GENERAL = true: walls / grid edges / listener, coefficients resident in registers.
// A wave advances SUB interior rows (+ K halo rows either side) of tile `tile`; `part` selects which SUB-row slice
// of the tile's RXI rows (air tiles: SUB == RXI, part 0; general tiles are split over RXI/SUB waves to shorten the
// latency of that small, VALU-heavier kernel).
template <int K, int RXI, int SUB, bool GENERAL>
__device__ __forceinline__ void stepTile(const StepArgs& a, const int tile, const int part, const int lane) {
    constexpr int ROWS = SUB + 2 * K;
    constexpr int WI = 64 - 2 * K;
    const int ti = tile / a.nty;
    const int tj = tile - ti * a.nty;
    const int row0 = a.G - K + ti * RXI + part * SUB;  // first loaded row / column, padded coordinates
    const int col0 = a.G - K + tj * WI;
    const int voff = lane * 4;
    const int pitchB = a.pitch * 4;
    const int soff0 = (row0 * a.pitch + col0) * 4;

    const rsrc_t rPrIn = makeRsrc(a.prIn, a.inBytes), rVxIn = makeRsrc(a.vxIn, a.inBytes),
                 rVyIn = makeRsrc(a.vyIn, a.inBytes);

    float pr[ROWS], vx[ROWS], vy[ROWS];
    constexpr int CR = GENERAL ? ROWS : 1;
    float kx[CR], ky[CR], bt[CR];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int so = soff0 + r * pitchB;
        pr[r] = bufLoadF(rPrIn, voff, so);
        vx[r] = bufLoadF(rVxIn, voff, so);
        vy[r] = bufLoadF(rVyIn, voff, so);
    }
    if constexpr (GENERAL) loadFaceCoefs<CR>(a, lane, soff0, pitchB, kx, ky, bt);

    // is anything non-zero in the tile?  (sign bit ignored: -0 from v = -p at the grid edges is still zero)
    uint32_t nz = 0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
        nz |= (__float_as_uint(pr[r]) | __float_as_uint(vx[r]) | __float_as_uint(vy[r])) & 0x7fffffffu;
    bool active = __ballot(nz != 0u) != 0ull;

    const DynParams dyn = *a.dyn;
    const int lr = dyn.lrow - row0;
    const int lc = dyn.lcol - col0;
    const bool hasL = GENERAL && a.withPulse && lr >= 0 && lr < ROWS && lc >= 0 && lc < 64;
    active = active || hasL;
    // non-zero flag for the row-streaming kernel's recording decision in the next launch (conservative: general
    // tiles, whose slices are advanced by several waves, always count as non-zero)
    if (lane == 0 && (GENERAL || part == 0)) a.nzOut[tile] = (GENERAL || active) ? 1 : 0;

    const int hti = ti - dyn.histTileX0, htj = tj - dyn.histTileY0;
    const bool inWin = hti >= 0 && hti < dyn.histTilesX && htj >= 0 && htj < dyn.histTilesY;
    // tileFirst[tile] = first launch in which the tile was non-zero; the analysis reads its history from there on.
    // Air tiles are recorded from that launch to the end of the run.  General tiles are recorded on every step:
    // their slices are advanced by different waves, which could not otherwise agree on when recording starts.
    bool rec;
    if (GENERAL) {
        rec = a.record && inWin && historyWanted(a, ti, tj);
        if (a.record && active && lane == 0) atomicMin(&a.tileFirst[tile], a.t0);
    } else {
        const bool wasActive = a.record && a.tileFirst[tile] != INT_MAX;
        rec = a.record && inWin && (active || wasActive || a.dense) && historyWanted(a, ti, tj);
        if (a.record && active && !wasActive && lane == 0) atomicMin(&a.tileFirst[tile], a.t0);
    }
    if (a.record && active && !inWin && lane == 0) atomicExch(a.errFlag, 1);

    const float C = a.courant;
    const bool inCols = lane >= K && lane < 64 - K;
    const float* hplane = a.hist + (long long)a.histSlot * a.histPlane;
    // history window addressing (tile-major, see histOffset): soffset = tile + row part (>= 0 for every stored row),
    // voffset = column part
    const int hpitchB = WI * 4;
    const int hsoff0 = ((hti * a.dyn->histTilesY + htj) * RXI + part * SUB - K) * hpitchB;  // + r*hpitchB with r >= K
    const int hvoff = (lane - K) * 4;                   // >= 0 for the stored lanes (lane >= K)

#pragma unroll 1
    for (int s = 0; s < a.nsteps; ++s) {
        if constexpr (!GENERAL) {
            leapfrogStep<ROWS>(pr, vx, vy, C);
        } else {
            leapfrogStepCoef<ROWS>(pr, vx, vy, kx, ky, bt, C);
        }

        // record the pressure of this step before the pulse is injected (FDTD.cpp:226-234)
        if (rec) {
            const rsrc_t rH = makeRsrc(hplane, a.histPlane * 4);
            if (inCols) {
#pragma unroll
                for (int r = K; r < ROWS - K; ++r) bufStoreF(pr[r], rH, hvoff, hsoff0 + r * hpitchB);
            }
        }
        hplane += a.histPlane;

        if (GENERAL && hasL) {  // soft source: p[listener] += pulse[t], FDTD.cpp:234
            const float pv = (lane == lc) ? a.pulse[a.t0 + s] : 0.f;
            int lrS = lr;  // opaque per step, so the ROWS row-select masks are not hoisted and spilled
            asm volatile("" : "+s"(lrS));
#pragma unroll
            for (int r = 0; r < ROWS; ++r) pr[r] += (r == lrS) ? pv : 0.f;
        }
    }

    const rsrc_t rPrOut = makeRsrc(a.prOut, a.planeBytes), rVxOut = makeRsrc(a.vxOut, a.planeBytes),
                 rVyOut = makeRsrc(a.vyOut, a.planeBytes);
    if (inCols) {
#pragma unroll
        for (int r = K; r < ROWS - K; ++r) {
            const int so = soff0 + r * pitchB;
            bufStoreF(pr[r], rPrOut, voff, so);
            bufStoreF(vx[r], rVxOut, voff, so);
            bufStoreF(vy[r], rVyOut, voff, so);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// air tile with packed-f32 arithmetic
// ---------------------------------------------------------------------------------------------------------------
// PMC on the scalar form (SQ_ACTIVE_INST_VALU ~ 92 % of the SIMD issue slots at K = 8) shows the air kernel is bound
// by VALU ISSUE, not by HBM.  Packed ops (v_pk_add_f32 / v_pk_mul_f32, two floats per lane) halve the instruction
// count; measured later (tools/valu_probe.hip): a packed op takes 4.4-4.9 cycles per wave and SIMD against 2.75 for a
// plain two-source f32 op, so packing is worth 1.13-1.25x in VALU time (and a DPP read costs as much as a packed op),
// not the 2x this was written for.  (Original text: "only packed ops
// reach the full f32 rate.")  Here two ADJACENT ROWS of a lane's column live in one 64-bit register pair, so every
// y-direction difference and every multiply / subtract is one packed op per two cells; x-direction differences need
// the row-shifted pair (p[r-1], p[r]), built with one v_pk_mov-style shuffle per pair.  Per-element arithmetic and
// its order are unchanged (packed ops are IEEE per half), so the fields stay bit-identical.

// ILP: the three sweeps are written breadth-first over groups of G row pairs (all shuffles, then all differences,
// then all multiplies ...) so that consecutive instructions of a wave are independent; written row by row the
// compiler funnels every row through the same two temporaries and each instruction waits for its predecessor
// (SQ_WAIT_INST_ANY 44 % of wave cycles).
template <int NP, int G, int LO = 0, int HI = NP>
__device__ __forceinline__ void leapfrogStepPacked(v2f (&pr)[NP], v2f (&vx)[NP], v2f (&vy)[NP], const float C) {
    const v2f c2 = {C, C};
    // pressure sweep, FDTD.cpp:124-141
#pragma unroll
    for (int i0 = LO; i0 < HI; i0 += G) {
        v2f vxs[G], vyr[G], d[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = i0 + g;
            if (i < HI) {
                vxs[g] = (i + 1 < NP) ? __builtin_shufflevector(vx[i], vx[i + 1], 1, 2)
                                      : __builtin_shufflevector(vx[i], vx[i], 1, 1);  // last row: halo garbage
                // y-differences as two scalar subtracts: the lane shift folds into the subtract (v_sub_f32_dpp), one
                // instruction per cell pair less than shift + shift + packed subtract
                vyr[g].x = laneNext(vy[i].x) - vy[i].x;
                vyr[g].y = laneNext(vy[i].y) - vy[i].y;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < HI) vxs[g] = vxs[g] - vx[i0 + g];
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < HI) d[g] = vxs[g] + vyr[g];
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < HI) d[g] = c2 * d[g];
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < HI) pr[i0 + g] = pr[i0 + g] - d[g];
    }
    // vx sweep, FDTD.cpp:143-170 (air|air faces only in this kernel); descending so that pr[i-1] is still needed
#pragma unroll
    for (int i0 = LO; i0 < HI; i0 += G) {
        v2f t[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = i0 + g;
            if (i < HI)
                t[g] = (i > 0) ? __builtin_shufflevector(pr[i - 1], pr[i], 1, 2)
                               : __builtin_shufflevector(pr[i], pr[i], 0, 0);  // first row: halo garbage
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < HI) t[g] = pr[i0 + g] - t[g];
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < HI) t[g] = c2 * t[g];
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < HI) vx[i0 + g] = vx[i0 + g] - t[g];
    }
    // vy sweep, FDTD.cpp:172-199
#pragma unroll
    for (int i0 = LO; i0 < HI; i0 += G) {
        v2f t[G];
#pragma unroll
        for (int g = 0; g < G; g += 2) {
            const int i = i0 + g;
            if (i + 1 < HI) {
                subLanePrev2(pr[i], pr[i + 1], t[g], t[g + 1]);
            } else if (i < HI) {
                v2f unused;
                subLanePrev2(pr[i], pr[i], t[g], unused);
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < HI) t[g] = c2 * t[g];
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < HI) vy[i0 + g] = vy[i0 + g] - t[g];
    }
}

// The K steps of a launch, unrolled, each over the rows that are still VALID: after step s the outermost s+1 rows of
// the tile hold garbage (their stencil reached outside the loaded halo), so step s only advances rows
// [s, ROWS-s-1) -- a trapezoid in (row, time) that ends on the RXI interior rows.  17 % fewer VALU instructions at
// K = 10, ROWS = 60 than advancing every row every step; the lane (y) direction cannot be trimmed.
template <int K, int RXI, int S>
struct PackedSteps {
    static constexpr int ROWS = RXI + 2 * K;
    static constexpr int NP = ROWS / 2;
    static __device__ __forceinline__ void run(v2f (&pr)[NP], v2f (&vx)[NP], v2f (&vy)[NP], const float C,
                                               const StepArgs& a, const bool recLane, const float* hplane,
                                               const int hvoff, const int hsoff0, const int hpitchB) {
        if constexpr (S < K) {
            if (S < a.nsteps) {
                leapfrogStepPacked<NP, 4, S / 2, (ROWS - S) / 2>(pr, vx, vy, C);
                if (recLane) {  // pressure of this step, interior rows (air tiles never hold the listener)
                    const rsrc_t rH = makeRsrc(hplane, a.histPlane * 4);
#pragma unroll
                    for (int r = K; r < ROWS - K; ++r)
                        bufStoreF((r & 1) ? pr[r >> 1].y : pr[r >> 1].x, rH, hvoff, hsoff0 + r * hpitchB);
                }
#if PV_STEP_SCHEDBAR
                __builtin_amdgcn_sched_barrier(0);  // keep the steps apart: interleaving them only costs registers
#endif
                PackedSteps<K, RXI, S + 1>::run(pr, vx, vy, C, a, recLane, hplane + a.histPlane, hvoff, hsoff0,
                                                hpitchB);
            }
        }
    }
};

template <int K, int RXI>
__device__ __forceinline__ void stepTileAirPacked(const StepArgs& a, const int tile, const int lane) {
    constexpr int ROWS = RXI + 2 * K;
    static_assert(ROWS % 2 == 0, "packed air tile needs an even number of rows");
    constexpr int NP = ROWS / 2;
    constexpr int WI = 64 - 2 * K;
    const int ti = tile / a.nty;
    const int tj = tile - ti * a.nty;
    const int row0 = a.G - K + ti * RXI;
    const int col0 = a.G - K + tj * WI;
    const int voff = lane * 4;
    const int pitchB = a.pitch * 4;
    const int soff0 = (row0 * a.pitch + col0) * 4;

    const rsrc_t rPrIn = makeRsrc(a.prIn, a.inBytes), rVxIn = makeRsrc(a.vxIn, a.inBytes),
                 rVyIn = makeRsrc(a.vyIn, a.inBytes);
    v2f pr[NP], vx[NP], vy[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int so = soff0 + (2 * i) * pitchB;
        pr[i].x = bufLoadF(rPrIn, voff, so);
        pr[i].y = bufLoadF(rPrIn, voff, so + pitchB);
        vx[i].x = bufLoadF(rVxIn, voff, so);
        vx[i].y = bufLoadF(rVxIn, voff, so + pitchB);
        vy[i].x = bufLoadF(rVyIn, voff, so);
        vy[i].y = bufLoadF(rVyIn, voff, so + pitchB);
    }
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i)
        nz |= (__float_as_uint(pr[i].x) | __float_as_uint(pr[i].y) | __float_as_uint(vx[i].x) |
               __float_as_uint(vx[i].y) | __float_as_uint(vy[i].x) | __float_as_uint(vy[i].y)) &
              0x7fffffffu;
    const bool active = __ballot(nz != 0u) != 0ull;
    if (lane == 0) a.nzOut[tile] = active ? 1 : 0;

    const DynParams dyn = *a.dyn;
    const int hti = ti - dyn.histTileX0, htj = tj - dyn.histTileY0;
    const bool inWin = hti >= 0 && hti < dyn.histTilesX && htj >= 0 && htj < dyn.histTilesY;
    const bool wasActive = a.record && a.tileFirst[tile] != INT_MAX;
    const bool rec = a.record && inWin && (active || wasActive || a.dense) && historyWanted(a, ti, tj);
    if (a.record && active && !wasActive && lane == 0) atomicMin(&a.tileFirst[tile], a.t0);
    if (a.record && active && !inWin && lane == 0) atomicExch(a.errFlag, 1);

    const float C = a.courant;
    const bool inCols = lane >= K && lane < 64 - K;
    const float* hplane = a.hist + (long long)a.histSlot * a.histPlane;
    const int hpitchB = WI * 4;
    const int hsoff0 = ((hti * a.dyn->histTilesY + htj) * RXI - K) * hpitchB;
    const int hvoff = (lane - K) * 4;

    PackedSteps<K, RXI, 0>::run(pr, vx, vy, C, a, rec && inCols, hplane, hvoff, hsoff0, hpitchB);

    const rsrc_t rPrOut = makeRsrc(a.prOut, a.planeBytes), rVxOut = makeRsrc(a.vxOut, a.planeBytes),
                 rVyOut = makeRsrc(a.vyOut, a.planeBytes);
    if (inCols) {
#pragma unroll
        for (int r = K; r < ROWS - K; ++r) {
            const int so = soff0 + r * pitchB;
            bufStoreF((r & 1) ? pr[r >> 1].y : pr[r >> 1].x, rPrOut, voff, so);
            bufStoreF((r & 1) ? vx[r >> 1].y : vx[r >> 1].x, rVxOut, voff, so);
            bufStoreF((r & 1) ? vy[r >> 1].y : vy[r >> 1].x, rVyOut, voff, so);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// air tile, MIRROR-packed
// ---------------------------------------------------------------------------------------------------------------
// The row-pair packing above spends 2 of its 15 VALU instructions per cell pair and step on v_pk_mov shuffles: the
// x-differences vx[r+1]-vx[r] and pr[r]-pr[r-1] straddle register pairs.  Here a pair holds row i of the tile's TOP
// half in .x and the MIRRORED row ROWS-1-i of its BOTTOM half in .y, and the bottom half's vx is stored negated and
// shifted by one face (vx[i].y = -vx_row[ROWS-i]).  Reflecting the staggered grid about the tile's centre line maps
// the leapfrog update onto itself when vx changes sign, so both halves of every pair obey the SAME recurrence with
// the SAME neighbour pair (i+1 for the pressure sweep, i-1 for the vx sweep): every x-difference is one packed
// subtract of two whole register pairs, no shuffles -- 13 VALU instructions per cell pair and step.  IEEE negation,
// a-b = -(b-a) and round-to-nearest are sign-symmetric, so the fields keep the reference's bits (sign of zero
// aside, as everywhere in this file).  The two halves meet at face vx_row[NP], kept in one extra register per lane
// (`vxS`) and advanced by three scalar ops per step.  The trapezoid of still-valid rows is [s, ROWS-s-1) = pairs
// [s, NP) (.y of pair s is scratch), the same number of pairs per step as in the row-pair form.
template <int NP, int G, int LO>
__device__ __forceinline__ void mirrorPressureSweep(v2f (&pr)[NP], const v2f (&vx)[NP], const v2f (&vy)[NP],
                                                    const float vxS, const float C) {
    const v2f c2 = {C, C};
    // FDTD.cpp:124-141
#pragma unroll
    for (int i0 = LO; i0 < NP; i0 += G) {
        v2f xd[G], yd[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = i0 + g;
            if (i < NP) {
                const v2f nxt = (i + 1 < NP) ? vx[i + 1] : v2f{vxS, -vxS};
                xd[g] = nxt - vx[i];
                yd[g].x = laneNext(vy[i].x) - vy[i].x;
                yd[g].y = laneNext(vy[i].y) - vy[i].y;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < NP) xd[g] = xd[g] + yd[g];
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < NP) xd[g] = c2 * xd[g];
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < NP) pr[i0 + g] = pr[i0 + g] - xd[g];
    }
}

// FDTD.cpp:143-170 (air|air faces only); the seam face between the halves first
template <int NP, int G, int LO>
__device__ __forceinline__ void mirrorVxSweep(const v2f (&pr)[NP], v2f (&vx)[NP], float& vxS, const float C) {
    const v2f c2 = {C, C};
    {
        const float t = C * (pr[NP - 1].y - pr[NP - 1].x);
        vxS = vxS - t;
    }
#pragma unroll
    for (int i0 = LO; i0 < NP; i0 += G) {
        v2f t[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = i0 + g;
            if (i < NP) t[g] = pr[i] - pr[i > 0 ? i - 1 : 0];  // pair 0: tile-edge rows, never valid
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < NP) t[g] = c2 * t[g];
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < NP) vx[i0 + g] = vx[i0 + g] - t[g];
    }
}

// FDTD.cpp:172-199
template <int NP, int G, int LO>
__device__ __forceinline__ void mirrorVySweep(const v2f (&pr)[NP], v2f (&vy)[NP], const float C) {
    const v2f c2 = {C, C};
#pragma unroll
    for (int i0 = LO; i0 < NP; i0 += G) {
        v2f t[G];
#pragma unroll
        for (int g = 0; g < G; g += 2) {
            const int i = i0 + g;
            if (i + 1 < NP) {
                subLanePrev2(pr[i], pr[i + 1], t[g], t[g + 1]);
            } else if (i < NP) {
                v2f unused;
                subLanePrev2(pr[i], pr[i], t[g], unused);
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < NP) t[g] = c2 * t[g];
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < NP) vy[i0 + g] = vy[i0 + g] - t[g];
    }
}

template <int NP, int G, int LO>
__device__ __forceinline__ void leapfrogStepMirror(v2f (&pr)[NP], v2f (&vx)[NP], v2f (&vy)[NP], float& vxS,
                                                   const float C) {
    mirrorPressureSweep<NP, G, LO>(pr, vx, vy, vxS, C);
    mirrorVxSweep<NP, G, LO>(pr, vx, vxS, C);
    mirrorVySweep<NP, G, LO>(pr, vy, C);
}

#ifndef PV_AIR_MIRROR
#define PV_AIR_MIRROR 1
#endif
#ifndef PV_EDGE_TILES
#define PV_EDGE_TILES 0
#endif
// Row offsets of the air arm (round 5): the history rows' offsets ride in the stores' immediate fields on two scalar bases, and the
// output rows' offsets are formed behind the K steps from a base the compiler cannot identify with the loads' -- so no row offset
// is alive during the steps.  Rounds 3-4 kept ~100 of them alive through the kernel; the register allocator parked some in VGPR
// lanes, WHICH ones depended on everything else in the kernel, and a never-read pointer in the kernel's arguments
// (StepArgs::layoutPad) steered it away from reloading them on every step (237 v_readlane inside the arithmetic, 3-4 % slower at
// 4096^2).  0 = that form (without the pad: the bad allocation), for A/B builds.
#ifndef PV_ROWOFF2
#define PV_ROWOFF2 1
#endif
#ifndef PV_LOAD_FENCE
#define PV_LOAD_FENCE 0
#endif
#ifndef PV_PROBE_NOMEM
#define PV_PROBE_NOMEM 0
#endif
#ifndef PV_MIRROR_G
#define PV_MIRROR_G 4  // row pairs per interleaved group of the mirror sweeps
#endif
#ifndef PV_STEP_SCHEDBAR
#define PV_STEP_SCHEDBAR 1  // scheduling barrier between the steps of an air tile
#endif
#ifndef PV_LATE_ARGS
#define PV_LATE_ARGS 0
#endif
#ifndef PV_LOAD_PIN
#define PV_LOAD_PIN 2  // issue order of the mirror-pair air tile's loads: 0 = the compiler's (grouped by plane), 2 = pinned row by row,
                       // top to bottom: 4-5 % faster at 4096^2, 6-9 % at 8192^2 (profiles/r04_load_order.txt)
#endif
#ifndef PV_LOAD_PIN_ROWS
#define PV_LOAD_PIN_ROWS 2  // rows per scheduling barrier of the pinned row order
#endif
#ifndef PV_AIR_LOAD_AUX
#define PV_AIR_LOAD_AUX 0   // cache-policy word of the mirror-pair air tile's field loads / stores (measurement builds)
#endif
#ifndef PV_AIR_STORE_AUX
#define PV_AIR_STORE_AUX 0
#endif

// Edge tiles (tile class 2): tiles of an otherwise EMPTY region that touch the grid's x = 0, y = 0 or y = gy edge.
// Their only non-air faces are the absorbing edge itself (FDTD.cpp:201-223; face coefficients -1 / +1, i.e.
// Y = 1) and the dead cells outside the grid, so they run the air tile's code plus three overrides per step instead
// of the general path (which costs 3.7 air tiles):
//   x = 0    (row K of the first tile row):      vx[0, y] = -p[0, y]                    after the vx sweep
//   y = 0    (lane K of the first tile column):  vy[x, 0] = -p[x, 0]                    after the vy sweep
//   y = gy   (the ghost column, lane eR):        p[x, gy] = 0;  vy[x, gy] = p[x, gy-1]  after the pressure / vy sweep
// (k * (p_i + p_n) with k = -1 or +1 and the outside neighbour's p = 0, bit for bit).  Registers that hold cells
// OUTSIDE the grid evolve as if they were air; nothing inside the grid reads them (every face between inside and
// outside is one of the overridden ones), they are reloaded as zeros by the next launch and stored as zeros.
// The ghost ROW x = gx sits at a run-time row of the tile, which register-resident rows cannot index: tiles that see
// it stay on the general path.
struct EdgeInfo {
    bool top, left;
    int eR;  // lane of the ghost column y = gy, or -1
};

// one step of an edge tile: every row pair on every step (the time loop of the edge tiles is a real loop: a second
// copy of the 12 unrolled trapezoid steps in the same kernel pushes the air tiles' code out of the instruction
// cache and slows ALL tiles by 25 %)
template <int NP, int K>
__device__ __forceinline__ void edgeStepMirror(v2f (&pr)[NP], v2f (&vx)[NP], v2f (&vy)[NP], float& vxS, const float C,
                                               const EdgeInfo& e, const int lane) {
    mirrorPressureSweep<NP, 4, 0>(pr, vx, vy, vxS, C);
    const bool ghost = lane == e.eR;  // eR = -1: no lane
    if (e.eR >= 0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) pr[i] = v2f{ghost ? 0.f : pr[i].x, ghost ? 0.f : pr[i].y};
    }
    mirrorVxSweep<NP, 4, 0>(pr, vx, vxS, C);
    if (e.top) vx[K].x = -pr[K].x;
    mirrorVySweep<NP, 4, 0>(pr, vy, C);
    if (e.left) {
        const bool edge = lane == K;
#pragma unroll
        for (int i = 0; i < NP; ++i) vy[i] = v2f{edge ? -pr[i].x : vy[i].x, edge ? -pr[i].y : vy[i].y};
    }
    if (e.eR >= 0) {  // (selects, not a divergent branch: the lane shift must run with every lane on)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float px = lanePrev(pr[i].x), py = lanePrev(pr[i].y);
            vy[i] = v2f{ghost ? px : vy[i].x, ghost ? py : vy[i].y};
        }
    }
}

// The y-trapezoid's dead LANES switched off (round 6).  A tile's 64 lanes are 64 - 2K interior columns + K halo columns on each
// side; for the interior [K, 63 - K] to be right after step K - 1, step s needs its pressure on lanes [s, 62 - s] and its
// velocities on [s + 1, 62 - s] (the pressure sweep reads vy of lane + 1, the vy sweep pr of lane - 1: one lane lost per side and
// step).  The x-trapezoid skips its dead ROWS outright; the dead lanes kept executing all 13 packed operations per row pair and
// step on garbage -- at full toggle power on non-zero fields (roofline.dense: 2.03 instead of 2.40 GHz).
// PV_EXEC_TRAPEZOID = N > 0 narrows EXEC in front of the steps s = N, 2N, ... to lanes [s - PV_EXEC_MARGIN, 63 - s] and restores
// it behind the last step.  (63 - s, not 62 - s: a DPP read of a switched-off lane returns zero, bound_ctrl, and the pressure
// sweep of lane 62 - s reads vy of lane 63 - s, which is still valid data.  A lane that reads a switched-off neighbour on the
// other side is itself outside the next step's range, so no valid cell sees it; the bit-exact suite is the proof.)  Same
// instruction count per lane that stays on: what changes is the lanes that toggle.  Measured: profiles/r06_exec_mask.txt.
#ifndef PV_EXEC_TRAPEZOID
#ifdef PV_EXPERIMENTAL
#define PV_EXEC_TRAPEZOID 1  // (the experimental build carries it: tests/test_gpu_parity.py::test_exec_trapezoid_equals_product)
#else
#define PV_EXEC_TRAPEZOID 0  // off: measured, profiles/r06_exec_mask.txt (any partial mask costs ~1.6 % of the launch; the clock it buys back is worth +2 %)
#endif
#endif
#ifndef PV_EXEC_MARGIN
#define PV_EXEC_MARGIN 0
#endif
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // ("exec" on the clobber list is the point: the compiler must know)
template <unsigned long long MASK>
__device__ __forceinline__ void setExecMask() {
    asm volatile("s_mov_b32 exec_lo, %0\n\ts_mov_b32 exec_hi, %1" ::"i"((int)(unsigned)(MASK & 0xffffffffull)), "i"((int)(unsigned)(MASK >> 32))
                 : "exec");
}
#pragma clang diagnostic pop
template <int S>
__device__ __forceinline__ void narrowExecToStep() {
#if PV_EXEC_TRAPEZOID == 63  // (measurement build: lane 63 alone goes off at step 1 -- does ANY partial mask cost time?)
    if constexpr (S == 1) setExecMask<(~0ull >> 1)>();
#elif PV_EXEC_TRAPEZOID
    if constexpr (S > 0 && S % PV_EXEC_TRAPEZOID == 0) {
        constexpr int lo = S - PV_EXEC_MARGIN < 0 ? 0 : S - PV_EXEC_MARGIN;
        constexpr int hi = 63 - S;
        constexpr unsigned long long upTo = hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull);
        setExecMask<(upTo & ~((1ull << lo) - 1ull))>();
    }
#endif
}
__device__ __forceinline__ void restoreExec() {
#if PV_EXEC_TRAPEZOID
    setExecMask<~0ull>();
#endif
}

template <int K, int RXI, int S>
struct MirrorSteps {
    static constexpr int ROWS = RXI + 2 * K;
    static constexpr int NP = ROWS / 2;
    static __device__ __forceinline__ void run(v2f (&pr)[NP], v2f (&vx)[NP], v2f (&vy)[NP], float& vxS,
                                               const float C, const StepArgs& a, const bool recLane,
                                               const float* hplane, const int hvoff, const int hsoff0,
                                               const int hpitchB) {
        if constexpr (S < K) {
            if (S < a.nsteps) {
                narrowExecToStep<S>();
                leapfrogStepMirror<NP, PV_MIRROR_G, S>(pr, vx, vy, vxS, C);
                if (recLane) {  // pressure of this step, interior rows (air tiles never hold the listener)
                    const rsrc_t rH = makeRsrc(hplane, a.histPlane * 4);
#if PV_ROWOFF2
                    // the rows of a tile's history block are a compile-time pitch apart (tile-major planes): their offsets ride in
                    // the store's 12-bit immediate field on TWO scalar bases, instead of one scalar register per row that must
                    // stay alive through all K steps (36 of the ~100 row offsets the register allocator used to park in VGPR lanes)
                    const int hb1 = hsoff0 + K * hpitchB, hb2 = hb1 + 4096;
#pragma unroll
                    for (int r = K; r < ROWS - K; ++r) {
                        const int off = (r - K) * hpitchB;
                        const float v = r < NP ? pr[r].x : pr[ROWS - 1 - r].y;
                        if (off < 4096)
                            bufStoreF(v, rH, hvoff + off, hb1);
                        else
                            bufStoreF(v, rH, hvoff + (off - 4096), hb2);
                    }
#else
#pragma unroll
                    for (int r = K; r < ROWS - K; ++r)
                        bufStoreF(r < NP ? pr[r].x : pr[ROWS - 1 - r].y, rH, hvoff, hsoff0 + r * hpitchB);
#endif
                }
#if PV_STEP_SCHEDBAR
                __builtin_amdgcn_sched_barrier(0);  // keep the steps apart: interleaving them only costs registers
#endif
                MirrorSteps<K, RXI, S + 1>::run(pr, vx, vy, vxS, C, a, recLane, hplane + a.histPlane, hvoff,
                                                hsoff0, hpitchB);
            }
        }
    }
};

template <int K, int RXI, bool EDGE = false>
__device__ __forceinline__ void stepTileAirMirror(const StepArgs& a, const int tile, const int lane) {
    constexpr int ROWS = RXI + 2 * K;
    static_assert(ROWS % 2 == 0, "packed air tile needs an even number of rows");
    constexpr int NP = ROWS / 2;
    constexpr int WI = 64 - 2 * K;
    const int ti = tile / a.nty;
    const int tj = tile - ti * a.nty;
    const int row0 = a.G - K + ti * RXI;
    const int col0 = a.G - K + tj * WI;
    const int voff = lane * 4;
    const int pitchB = a.pitch * 4;
    const int soff0 = (row0 * a.pitch + col0) * 4;
    EdgeInfo e{false, false, -1};
    if constexpr (EDGE) {
        e.top = ti == 0;
        e.left = tj == 0;
        const int er = a.gy - (tj * WI - K);
        e.eR = (er >= 0 && er < 64) ? er : -1;
    }

    const rsrc_t rPrIn = makeRsrc(a.prIn, a.inBytes), rVxIn = makeRsrc(a.vxIn, a.inBytes),
                 rVyIn = makeRsrc(a.vyIn, a.inBytes);
    v2f pr[NP], vx[NP], vy[NP];
#if PV_PROBE_NOMEM == 1 || PV_PROBE_NOMEM == 2  // measurement builds only: the arithmetic of an air tile without its loads and stores
    float vxS = (float)lane * 1e-6f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        pr[i] = v2f{vxS + i, vxS - i};
        vy[i] = v2f{vxS * 2.f + i, vxS * 3.f - i};
        vx[i] = v2f{vxS * 5.f + i, vxS * 7.f - i};
    }
#else
#if PV_LOAD_PIN >= 2
    // rows top to bottom, a scheduling barrier every PV_LOAD_PIN_ROWS rows: the order the loads are ISSUED in (the compiler otherwise groups them by
    // plane; measured with tools/tile_major_probe.hip: the pinned row order is 5-7 % faster on the same bytes)
    // (plane by plane, bottom to top, pair by pair outside-in, the planes a third of the tile apart, stores pinned as well: all
    // measured, none better -- profiles/r04_load_order.txt)
    float vxS = 0.f;
    auto loadRow = [&](const int r) {
        const int so = soff0 + r * pitchB;
        if (r < NP) {
            pr[r].x = bufLoadFA<PV_AIR_LOAD_AUX>(rPrIn, voff, so);
            vy[r].x = bufLoadFA<PV_AIR_LOAD_AUX>(rVyIn, voff, so);
            vx[r].x = bufLoadFA<PV_AIR_LOAD_AUX>(rVxIn, voff, so);
        } else {
            const int i = ROWS - 1 - r;
            pr[i].y = bufLoadFA<PV_AIR_LOAD_AUX>(rPrIn, voff, so);
            vy[i].y = bufLoadFA<PV_AIR_LOAD_AUX>(rVyIn, voff, so);
            if (r == NP)
                vxS = bufLoadFA<PV_AIR_LOAD_AUX>(rVxIn, voff, so);
            else
                vx[i + 1].y = bufLoadFA<PV_AIR_LOAD_AUX>(rVxIn, voff, so);  // face r belongs to pair ROWS - r (negated below)
        }
        if (r % PV_LOAD_PIN_ROWS == PV_LOAD_PIN_ROWS - 1) __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int r = 0; r < ROWS; ++r) loadRow(r);
    vx[0].y = 0.f;  // face ROWS is not in the tile
#pragma unroll
    for (int i = 1; i < NP; ++i) vx[i].y = -vx[i].y;
#else
    float vxS = bufLoadFA<PV_AIR_LOAD_AUX>(rVxIn, voff, soff0 + NP * pitchB);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int soT = soff0 + i * pitchB, soB = soff0 + (ROWS - 1 - i) * pitchB;
        pr[i].x = bufLoadFA<PV_AIR_LOAD_AUX>(rPrIn, voff, soT);
        pr[i].y = bufLoadFA<PV_AIR_LOAD_AUX>(rPrIn, voff, soB);
        vy[i].x = bufLoadFA<PV_AIR_LOAD_AUX>(rVyIn, voff, soT);
        vy[i].y = bufLoadFA<PV_AIR_LOAD_AUX>(rVyIn, voff, soB);
        vx[i].x = bufLoadFA<PV_AIR_LOAD_AUX>(rVxIn, voff, soT);
        vx[i].y = (i > 0) ? -bufLoadFA<PV_AIR_LOAD_AUX>(rVxIn, voff, soB + pitchB) : 0.f;  // face ROWS-i; face ROWS is not in the tile
    }
#endif
#endif
    // All 3*ROWS loads are in flight before anything consumes one.  Without this fence the schedule depends on what
    // ELSE is in the kernel: with more code (an extra tile variant, even one that never runs) the scheduler feeds the
    // non-zero test below a few loads at a time, s_waitcnt vmcnt(5) after every group -- ten memory round trips per
    // tile instead of one, every tile 30 % slower (SQ_WAIT_ANY x2.3, same instruction counts, same I-cache hits).
#if PV_LOAD_FENCE
    __builtin_amdgcn_sched_barrier(0);
#endif
    uint32_t nz = __float_as_uint(vxS);
#pragma unroll
    for (int i = 0; i < NP; ++i)
        nz |= __float_as_uint(pr[i].x) | __float_as_uint(pr[i].y) | __float_as_uint(vx[i].x) |
              __float_as_uint(vx[i].y) | __float_as_uint(vy[i].x) | __float_as_uint(vy[i].y);
    const bool active = __ballot((nz & 0x7fffffffu) != 0u) != 0ull;
    if (lane == 0) a.nzOut[tile] = active ? 1 : 0;

    const DynParams dyn = *a.dyn;
    const int hti = ti - dyn.histTileX0, htj = tj - dyn.histTileY0;
    const bool inWin = hti >= 0 && hti < dyn.histTilesX && htj >= 0 && htj < dyn.histTilesY;
    const bool wasActive = a.record && a.tileFirst[tile] != INT_MAX;
    const bool rec = a.record && inWin && (active || wasActive || a.dense) && historyWanted(a, ti, tj);
    if (a.record && active && !wasActive && lane == 0) atomicMin(&a.tileFirst[tile], a.t0);
    if (a.record && active && !inWin && lane == 0) atomicExch(a.errFlag, 1);

    const float C = a.courant;
    const bool inCols = lane >= K && lane < 64 - K;
    const float* hplane = a.hist + (long long)a.histSlot * a.histPlane;
    const int hpitchB = WI * 4;
    const int hsoff0 = ((hti * a.dyn->histTilesY + htj) * RXI - K) * hpitchB;
    const int hvoff = (lane - K) * 4;

    if constexpr (EDGE) {
        const bool recLane = rec && inCols;
#pragma unroll 1
        for (int st = 0; st < a.nsteps; ++st) {
            edgeStepMirror<NP, K>(pr, vx, vy, vxS, C, e, lane);
            if (recLane) {
                const rsrc_t rH = makeRsrc(hplane, a.histPlane * 4);
#pragma unroll
                for (int r = K; r < ROWS - K; ++r)
                    bufStoreF(r < NP ? pr[r].x : pr[ROWS - 1 - r].y, rH, hvoff, hsoff0 + r * hpitchB);
            }
            hplane += a.histPlane;
        }
    } else {
        MirrorSteps<K, RXI, 0>::run(pr, vx, vy, vxS, C, a, rec && inCols, hplane, hvoff, hsoff0, hpitchB);
        restoreExec();
    }

#if PV_LATE_ARGS
    const StepArgs* late = (const StepArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(late));
#else
    const StepArgs* late = &a;
#endif
    const rsrc_t rPrOut = makeRsrc(late->prOut, late->planeBytes), rVxOut = makeRsrc(late->vxOut, late->planeBytes),
                 rVyOut = makeRsrc(late->vyOut, late->planeBytes);
    // cells past the ghost column are outside the grid: stored as the zeros they are in memory; the ghost column's
    // own vx is zero (wall|wall face)
    const bool outP = EDGE && e.eR >= 0 && lane > e.eR, outX = EDGE && e.eR >= 0 && lane >= e.eR;
#if PV_PROBE_NOMEM == 1 || PV_PROBE_NOMEM == 2
    if (inCols && pr[3].x == 12345.678f) {  // (never true: keeps the arithmetic alive without the stores)
#else
    if (inCols) {
#endif
#if PV_ROWOFF2
        // the output rows' offsets are formed HERE, from a base and a pitch the compiler cannot identify with the ones the loads
        // used: nothing of them is alive during the K steps
        int soffS = soff0, pitchS = pitchB;
        asm volatile("" : "+s"(soffS), "+s"(pitchS));
#else
        const int soffS = soff0, pitchS = pitchB;
#endif
#pragma unroll
        for (int r = K; r < ROWS - K; ++r) {
            const int so = soffS + r * pitchS;
            const float p = r < NP ? pr[r].x : pr[ROWS - 1 - r].y;
            const float x = r < NP ? vx[r].x : (r == NP ? vxS : -vx[ROWS - r].y);
            const float y = r < NP ? vy[r].x : vy[ROWS - 1 - r].y;
            bufStoreFA<PV_AIR_STORE_AUX>(outP ? 0.f : p, rPrOut, voff, so);
            bufStoreFA<PV_AIR_STORE_AUX>(outX ? 0.f : x, rVxOut, voff, so);
            bufStoreFA<PV_AIR_STORE_AUX>(outP ? 0.f : y, rVyOut, voff, so);
        }
    }
}

// does a (K, RXI) configuration have the edge-tile path?  (the mirror-pair tiles)
template <int K, int RXI>
constexpr bool edgeTilesOk() {
    return PV_AIR_MIRROR >= 1 && (RXI + 2 * K) % 2 == 0 && RXI + 2 * K >= 48;
}

// Which packed form a configuration uses: the mirror pairs win on the tall tiles that run at 2 waves/SIMD (-1..2 %
// at 4096^2 / 8192^2 with the 60-row tile) and lose 1-2 % on the 40-row tiles at 3 waves/SIMD (measured, K = 8).
// -DPV_AIR_MIRROR=0 / =2 force one form for A/B builds.
template <int K, int RXI, bool EDGE_ARM = (PV_EDGE_TILES != 0)>
__device__ __forceinline__ void airTilePacked(const StepArgs& a, const int tile, const int lane, const bool edge) {
    if constexpr (PV_AIR_MIRROR == 2 || (PV_AIR_MIRROR == 1 && RXI + 2 * K >= 48)) {
        if constexpr (EDGE_ARM && edgeTilesOk<K, RXI>()) {
            if (edge)
                stepTileAirMirror<K, RXI, true>(a, tile, lane);
            else
                stepTileAirMirror<K, RXI, false>(a, tile, lane);
        } else {
            stepTileAirMirror<K, RXI, false>(a, tile, lane);
        }
    } else {
        stepTileAirPacked<K, RXI>(a, tile, lane);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// general tile: 4 waves share the tile's rows, boundary faces exchanged through LDS
// ---------------------------------------------------------------------------------------------------------------
// Walls, grid edges, listener.  A general row needs its two face coefficients and beta beside the three fields
// (6 registers per row instead of 3) and 20 instructions per step instead of 6.5, so a whole (RXI+2K)-row tile does
// not fit one wave.  The first design cut the tile into RXI/4-row slices advanced by independent waves, each
// recomputing its own 2K-row halo: 4 x 33 rows for a 60-row tile, and the ~4 % of tiles on the border of an open
// 4096^2 grid took 22 % of the run.  Here the block's 4 waves own CONSECUTIVE windows of the tile's rows and keep
// each other's boundary faces current through LDS (the scheme of the stacked air tile below, scalar layout):
//   wave w holds rows [w(R-2), w(R-2)+R) of the loaded tile; row 0 of its window is the previous wave's last owned
//   row, whose pr and vy it advances redundantly; rows 1..R-2 are its own; of row R-1 it only needs vx.
//   Every step it publishes vx[1] (the wave above's vx[R-1]) and vx[R-2] (the wave below's vx[0]) and receives
//   the two counterparts: one 8-byte LDS write per lane, one workgroup barrier, two LDS reads.
// 4 x 17 rows for the 60-row tile, no halo recomputed between the waves.
template <int K, int RXI>
struct GenStackGeom {
    static constexpr int L0 = RXI + 2 * K;
    static constexpr int R = (L0 + 6 + 3) / 4;   // 4 windows of R rows, stride R-2, cover 4R-6 >= L0 rows
    static constexpr int L = 4 * R - 6;           // rows the block loads
    static constexpr int D = L - L0;              // rows beyond the tile's own K-row halo at the bottom (0..3)
};

struct GenShared {
    float xch[2][4][2][64];  // [step parity][wave][0: vx[1] for the wave above, 1: vx[R-2] for the wave below][lane]
};

template <int K, int RXI>
__device__ __forceinline__ void stepTileGeneral4Scalar(const StepArgs& a, const int tile, const int wave, const int lane, GenShared& sh) {
    using Gm = GenStackGeom<K, RXI>;
    constexpr int R = Gm::R;
    constexpr int WI = 64 - 2 * K;
    const int ti = tile / a.nty;
    const int tj = tile - ti * a.nty;
    const int ws = wave * (R - 2);                   // first row of this wave's window, in loaded-tile rows
    const int row0 = a.G - K + ti * RXI + ws;
    const int col0 = a.G - K + tj * WI;
    const int voff = lane * 4;
    const int pitchB = a.pitch * 4;
    const int soff0 = (row0 * a.pitch + col0) * 4;
    const bool first = wave == 0, last = wave == 3;

    const rsrc_t rPrIn = makeRsrc(a.prIn, a.inBytes), rVxIn = makeRsrc(a.vxIn, a.inBytes),
                 rVyIn = makeRsrc(a.vyIn, a.inBytes);
    float pr[R], vx[R], vy[R], kx[R], ky[R], bt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int so = soff0 + r * pitchB;
        pr[r] = bufLoadF(rPrIn, voff, so);
        vx[r] = bufLoadF(rVxIn, voff, so);
        vy[r] = bufLoadF(rVyIn, voff, so);
    }
    loadFaceCoefs<R>(a, lane, soff0, pitchB, kx, ky, bt);

    const DynParams dyn = *a.dyn;
    // listener row inside this wave's window: rows 0..R-2 hold a live pressure (row 0 = the copy of the previous
    // wave's last row), row R-1 does not
    const int lr = dyn.lrow - row0;
    const int lc = dyn.lcol - col0;
    const bool hasL = a.withPulse && lr >= 0 && lr <= R - 2 && lc >= 0 && lc < 64;
    float pvec = 0.f;
    if constexpr (pulseLanesConfig(K, RXI)) pvec = hasL ? loadPulseLanes(a, lane) : 0.f;
    const int lrT = dyn.lrow - (row0 - ws);  // listener row in loaded-tile rows: is it anywhere in the block?
    const bool tileHasL = a.withPulse && lrT >= 0 && lrT < Gm::L && lc >= 0 && lc < 64;
    // general tiles are recorded on every step
    const int hti = ti - dyn.histTileX0, htj = tj - dyn.histTileY0;
    const bool inWin = hti >= 0 && hti < dyn.histTilesX && htj >= 0 && htj < dyn.histTilesY;
    const bool rec = a.record && inWin && historyWanted(a, ti, tj);
    if (a.record) {
        uint32_t nz = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
            nz |= (__float_as_uint(pr[r]) | __float_as_uint(vx[r]) | __float_as_uint(vy[r])) & 0x7fffffffu;
        const bool active = tileHasL || __ballot(nz != 0u) != 0ull;
        if (active && lane == 0) {
            atomicMin(&a.tileFirst[tile], a.t0);
            if (!inWin) atomicExch(a.errFlag, 1);
        }
    }

    // rows this wave stores: its own rows 1..R-2 that lie in the tile's interior
    const int rLo = max(1, K - ws), rHi = min(R - 1, K + RXI - ws);
    const float C = a.courant;
    const bool inCols = lane >= K && lane < 64 - K;
    const float* hplane = a.hist + (long long)a.histSlot * a.histPlane;
    const int hpitchB = WI * 4;
    const int hsoff0 = ((hti * a.dyn->histTilesY + htj) * RXI - K + ws) * hpitchB;
    const int hvoff = (lane - K) * 4;

#pragma unroll 1
    for (int s = 0; s < a.nsteps; ++s) {
        // pressure sweep, FDTD.cpp:124-141 (row R-1 holds no live pressure)
#pragma unroll
        for (int r = 0; r < R - 1; ++r) {
            const float vyR = laneNext(vy[r]);
            const float div = (vx[r + 1] - vx[r]) + (vyR - vy[r]);
            pr[r] = bt[r] * (pr[r] - C * div);
        }
        // vx sweep, FDTD.cpp:143-170 (+ edges :201-223 through the coefficients): own rows only
#pragma unroll
        for (int r = R - 2; r >= 1; --r) {
            const float pi = pr[r], pn = pr[r - 1];
            const float air = vx[r] - C * (pi - pn);
            const float wall = kx[r] * (pi + pn);
            vx[r] = (kx[r] != kx[r]) ? air : wall;
        }
        sh.xch[s & 1][wave][0][lane] = vx[1];
        sh.xch[s & 1][wave][1][lane] = vx[R - 2];
        // vy sweep, FDTD.cpp:172-199
#pragma unroll
        for (int r = 0; r < R - 1; ++r) {
            const float pi = pr[r];
            const float pn = lanePrev(pi);
            const float air = vy[r] - C * (pi - pn);
            const float wall = ky[r] * (pi + pn);
            vy[r] = (ky[r] != ky[r]) ? air : wall;
        }
        // LDS only: the history stores of earlier steps stay in flight across the barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!first) vx[0] = sh.xch[s & 1][wave - 1][1][lane];
        if (!last) vx[R - 1] = sh.xch[s & 1][wave + 1][0][lane];

        // record the pressure of this step before the pulse is injected (FDTD.cpp:226-234)
        if (rec && inCols) {
            const rsrc_t rH = makeRsrc(hplane, a.histPlane * 4);
#pragma unroll
            for (int r = 1; r < R - 1; ++r)
                if (r >= rLo && r < rHi) bufStoreF(pr[r], rH, hvoff, hsoff0 + r * hpitchB);
        }
        hplane += a.histPlane;

        if (hasL) {  // soft source: p[listener] += pulse[t], FDTD.cpp:234
            float pv;
            if constexpr (pulseLanesConfig(K, RXI))
                pv = (lane == lc) ? pulseOfStep(pvec, s) : 0.f;
            else
                pv = (lane == lc) ? a.pulse[a.t0 + s] : 0.f;
#pragma unroll
            for (int r = 0; r < R - 1; ++r) pr[r] += (r == lr) ? pv : 0.f;
        }
    }

    // per-tile non-zero flag of the state this launch leaves behind (halo included: conservative), for the row-streaming
    // segments' recording decision in the next launch.  Monotone within a run: written, never cleared
    // (pv_begin_run_kernel zeroes both flag planes).
    if (a.record) {
        uint32_t nzE = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
            nzE |= (__float_as_uint(pr[r]) | __float_as_uint(vx[r]) | __float_as_uint(vy[r])) & 0x7fffffffu;
        if ((__ballot(nzE != 0u) != 0ull || a.nzIn[tile]) && lane == 0) a.nzOut[tile] = 1;
    }
    const rsrc_t rPrOut = makeRsrc(a.prOut, a.planeBytes), rVxOut = makeRsrc(a.vxOut, a.planeBytes),
                 rVyOut = makeRsrc(a.vyOut, a.planeBytes);
    if (inCols) {
#pragma unroll
        for (int r = 1; r < R - 1; ++r) {
            if (r >= rLo && r < rHi) {
                const int so = soff0 + r * pitchB;
                bufStoreF(pr[r], rPrOut, voff, so);
                bufStoreF(vx[r], rVxOut, voff, so);
                bufStoreF(vy[r], rVyOut, voff, so);
            }
        }
    }
}

// PV_GENERAL_PACKED (round 2): the same block, the same windows, the same exchange -- with two ADJACENT rows of a lane's
// column per 64-bit register pair and v_pk_* arithmetic, as in stepTileAirPacked.  The x-differences need the row-shifted
// pair (one v_pk_mov-style shuffle per pair and sweep); the air / wall choice of a face is a bit select (v_bfi_b32) with
// a mask made once per launch from the face coefficient (NaN = air|air) instead of a compare + v_cndmask per face and
// step: 12.5 instead of 21 VALU instructions per row and step.  Same operations per cell in the same order (packed ops
// are IEEE per half; the wall value of an air face is computed and discarded, as before): same bits.
#ifndef PV_GEN_LOAD_PIN
#define PV_GEN_LOAD_PIN 0  // 1: the packed general arm's loads issued row by row (measurement: profiles/r06_general_arm.txt)
#endif
#ifndef PV_GENERAL_PACKED
#define PV_GENERAL_PACKED 1  // 1 = where it pays (K < 12), 2 = everywhere, 0 = nowhere
#endif
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

template <int K, int RXI>
__device__ __forceinline__ void stepTileGeneral4Packed(const StepArgs& a, const int tile, const int wave, const int lane, GenShared& sh) {
    using Gm = GenStackGeom<K, RXI>;
    constexpr int R = Gm::R;
    constexpr int NP = (R + 1) / 2;  // row pairs (R odd: the last pair's second row is a spare row below the window)
    constexpr int WI = 64 - 2 * K;
    const int ti = tile / a.nty;
    const int tj = tile - ti * a.nty;
    const int ws = wave * (R - 2);                   // first row of this wave's window, in loaded-tile rows
    const int row0 = a.G - K + ti * RXI + ws;
    const int col0 = a.G - K + tj * WI;
    const int voff = lane * 4;
    const int pitchB = a.pitch * 4;
    const int soff0 = (row0 * a.pitch + col0) * 4;
    const bool first = wave == 0, last = wave == 3;

    const rsrc_t rPrIn = makeRsrc(a.prIn, a.inBytes), rVxIn = makeRsrc(a.vxIn, a.inBytes),
                 rVyIn = makeRsrc(a.vyIn, a.inBytes);
    // row r of the window lives in pair r / 2, component r % 2
    v2f pr[NP], vx[NP], vy[NP], kx[NP], ky[NP], bt[NP];
    u2 mx[NP], my[NP];  // all-ones where the face is air|air
    auto getc = [](const v2f& v, int r) { return (r & 1) ? v.y : v.x; };
    auto setc = [](v2f& v, int r, float val) {
        if (r & 1) v.y = val; else v.x = val;
    };
    float kxr[R], kyr[R], btr[R];
#if PV_GEN_LOAD_PIN
    {  // rows top to bottom, the row's coefficients with its fields, a scheduling barrier every two rows (the air arm's order: PV_LOAD_PIN)
        typedef unsigned int u3v __attribute__((ext_vector_type(3)));
        const rsrc_t rCoef = makeRsrc(a.coef, a.planeBytes * 3);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int so = soff0 + r * pitchB;
            setc(pr[r / 2], r, bufLoadF(rPrIn, voff, so));
            setc(vy[r / 2], r, bufLoadF(rVyIn, voff, so));
            setc(vx[r / 2], r, bufLoadF(rVxIn, voff, so));
            const u3v c = __builtin_amdgcn_raw_buffer_load_b96(rCoef, lane * 12, 3 * so, 0);
            kxr[r] = __uint_as_float(c.x);
            kyr[r] = __uint_as_float(c.y);
            btr[r] = __uint_as_float(c.z);
            if (r % 2 == 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
#else
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int so = soff0 + r * pitchB;
        setc(pr[r / 2], r, bufLoadF(rPrIn, voff, so));
        setc(vx[r / 2], r, bufLoadF(rVxIn, voff, so));
        setc(vy[r / 2], r, bufLoadF(rVyIn, voff, so));
    }
    loadFaceCoefs<R>(a, lane, soff0, pitchB, kxr, kyr, btr);
#endif
#pragma unroll
    for (int r = 0; r < 2 * NP; ++r) {
        if (r < R) {
            const float kxv = kxr[r < R ? r : 0], kyv = kyr[r < R ? r : 0];
            setc(kx[r / 2], r, kxv);
            setc(ky[r / 2], r, kyv);
            setc(bt[r / 2], r, btr[r < R ? r : 0]);
            if (r & 1) {
                mx[r / 2].y = (kxv != kxv) ? 0xffffffffu : 0u;
                my[r / 2].y = (kyv != kyv) ? 0xffffffffu : 0u;
            } else {
                mx[r / 2].x = (kxv != kxv) ? 0xffffffffu : 0u;
                my[r / 2].x = (kyv != kyv) ? 0xffffffffu : 0u;
            }
        } else {  // the spare row: zeros, wall|wall faces (never read by a live row)
            setc(pr[r / 2], r, 0.f);
            setc(vx[r / 2], r, 0.f);
            setc(vy[r / 2], r, 0.f);
            setc(kx[r / 2], r, 0.f);
            setc(ky[r / 2], r, 0.f);
            setc(bt[r / 2], r, 0.f);
            if (r & 1) mx[r / 2].y = my[r / 2].y = 0u; else mx[r / 2].x = my[r / 2].x = 0u;
        }
    }

    const DynParams dyn = *a.dyn;
    // listener row inside this wave's window: rows 0..R-2 hold a live pressure (row 0 = the copy of the previous
    // wave's last row), row R-1 does not
    const int lr = dyn.lrow - row0;
    const int lc = dyn.lcol - col0;
    const bool hasL = a.withPulse && lr >= 0 && lr <= R - 2 && lc >= 0 && lc < 64;
    float pvec = 0.f;
    if constexpr (pulseLanesConfig(K, RXI)) pvec = hasL ? loadPulseLanes(a, lane) : 0.f;
    const int lrT = dyn.lrow - (row0 - ws);  // listener row in loaded-tile rows: is it anywhere in the block?
    const bool tileHasL = a.withPulse && lrT >= 0 && lrT < Gm::L && lc >= 0 && lc < 64;
    // general tiles are recorded on every step
    const int hti = ti - dyn.histTileX0, htj = tj - dyn.histTileY0;
    const bool inWin = hti >= 0 && hti < dyn.histTilesX && htj >= 0 && htj < dyn.histTilesY;
    const bool rec = a.record && inWin && historyWanted(a, ti, tj);
    if (a.record) {
        uint32_t nz = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            nz |= (__float_as_uint(pr[i].x) | __float_as_uint(pr[i].y) | __float_as_uint(vx[i].x) |
                   __float_as_uint(vx[i].y) | __float_as_uint(vy[i].x) | __float_as_uint(vy[i].y)) & 0x7fffffffu;
        const bool active = tileHasL || __ballot(nz != 0u) != 0ull;
        if (active && lane == 0) {
            atomicMin(&a.tileFirst[tile], a.t0);
            if (!inWin) atomicExch(a.errFlag, 1);
        }
    }

    // rows this wave stores: its own rows 1..R-2 that lie in the tile's interior
    const int rLo = max(1, K - ws), rHi = min(R - 1, K + RXI - ws);
    const float C = a.courant;
    const v2f c2 = {C, C};
    const bool inCols = lane >= K && lane < 64 - K;
    const float* hplane = a.hist + (long long)a.histSlot * a.histPlane;
    const int hpitchB = WI * 4;
    const int hsoff0 = ((hti * a.dyn->histTilesY + htj) * RXI - K + ws) * hpitchB;
    const int hvoff = (lane - K) * 4;
    auto sel = [](const u2 m, const v2f airv, const v2f wallv) {  // air where the mask is set, else wall: v_bfi_b32 x 2
        const u2 ua = __builtin_bit_cast(u2, airv), uw = __builtin_bit_cast(u2, wallv);
        return __builtin_bit_cast(v2f, (ua & m) | (uw & ~m));
    };

#pragma unroll 1
    for (int s = 0; s < a.nsteps; ++s) {
        // pressure sweep, FDTD.cpp:124-141 (row R-1 holds no live pressure; what is computed there is never read)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const v2f vxn = (i + 1 < NP) ? v2f{vx[i].y, vx[i + 1].x} : v2f{vx[i].y, 0.f};  // vx of rows r+1
            const v2f dyv = v2f{laneNext(vy[i].x) - vy[i].x, laneNext(vy[i].y) - vy[i].y};
            const v2f div = (vxn - vx[i]) + dyv;
            pr[i] = bt[i] * (pr[i] - c2 * div);
        }
        // vx sweep, FDTD.cpp:143-170 (+ edges :201-223 through the coefficients): own rows only (1 .. R-2); row 0's
        // and row R-1's faces come from the neighbouring waves below, and whatever is computed for them here is
        // overwritten
        const float vx0keep = vx[0].x, vxLkeep = getc(vx[(R - 1) / 2], R - 1);
#pragma unroll
        for (int i = NP - 1; i >= 0; --i) {
            const v2f pnv = (i > 0) ? v2f{pr[i - 1].y, pr[i].x} : v2f{pr[0].x, pr[0].x};  // pressure of rows r-1
            const v2f airv = vx[i] - c2 * (pr[i] - pnv);
            const v2f wallv = kx[i] * (pr[i] + pnv);
            vx[i] = sel(mx[i], airv, wallv);
        }
        vx[0].x = vx0keep;
        setc(vx[(R - 1) / 2], R - 1, vxLkeep);
        sh.xch[s & 1][wave][0][lane] = vx[0].y;                          // vx[1]
        sh.xch[s & 1][wave][1][lane] = getc(vx[(R - 2) / 2], R - 2);      // vx[R-2]
        // vy sweep, FDTD.cpp:172-199
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const v2f pnv = v2f{lanePrev(pr[i].x), lanePrev(pr[i].y)};
            const v2f airv = vy[i] - c2 * (pr[i] - pnv);
            const v2f wallv = ky[i] * (pr[i] + pnv);
            vy[i] = sel(my[i], airv, wallv);
        }
        // LDS only: the history stores of earlier steps stay in flight across the barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!first) vx[0].x = sh.xch[s & 1][wave - 1][1][lane];
        if (!last) setc(vx[(R - 1) / 2], R - 1, sh.xch[s & 1][wave + 1][0][lane]);

        // record the pressure of this step before the pulse is injected (FDTD.cpp:226-234)
        if (rec && inCols) {
            const rsrc_t rH = makeRsrc(hplane, a.histPlane * 4);
#pragma unroll
            for (int r = 1; r < R - 1; ++r)
                if (r >= rLo && r < rHi) bufStoreF(getc(pr[r / 2], r), rH, hvoff, hsoff0 + r * hpitchB);
        }
        hplane += a.histPlane;

        if (hasL) {  // soft source: p[listener] += pulse[t], FDTD.cpp:234
            float pv;
            if constexpr (pulseLanesConfig(K, RXI))
                pv = (lane == lc) ? pulseOfStep(pvec, s) : 0.f;
            else
                pv = (lane == lc) ? a.pulse[a.t0 + s] : 0.f;
#pragma unroll
            for (int r = 0; r < R - 1; ++r) setc(pr[r / 2], r, getc(pr[r / 2], r) + ((r == lr) ? pv : 0.f));
        }
    }

    // per-tile non-zero flag of the state this launch leaves behind (halo included: conservative), for the row-streaming
    // segments' recording decision in the next launch.  Monotone within a run: written, never cleared
    // (pv_begin_run_kernel zeroes both flag planes).
    if (a.record) {
        uint32_t nzE = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            nzE |= (__float_as_uint(pr[i].x) | __float_as_uint(pr[i].y) | __float_as_uint(vx[i].x) |
                    __float_as_uint(vx[i].y) | __float_as_uint(vy[i].x) | __float_as_uint(vy[i].y)) & 0x7fffffffu;
        if ((__ballot(nzE != 0u) != 0ull || a.nzIn[tile]) && lane == 0) a.nzOut[tile] = 1;
    }
    const rsrc_t rPrOut = makeRsrc(a.prOut, a.planeBytes), rVxOut = makeRsrc(a.vxOut, a.planeBytes),
                 rVyOut = makeRsrc(a.vyOut, a.planeBytes);
    if (inCols) {
#pragma unroll
        for (int r = 1; r < R - 1; ++r) {
            if (r >= rLo && r < rHi) {
                const int so = soff0 + r * pitchB;
                bufStoreF(getc(pr[r / 2], r), rPrOut, voff, so);
                bufStoreF(getc(vx[r / 2], r), rVxOut, voff, so);
                bufStoreF(getc(vy[r / 2], r), rVyOut, voff, so);
            }
        }
    }
}

template <int K, int RXI, bool GP = false>
__device__ __forceinline__ void stepTileGeneral4(const StepArgs& a, const int tile, const int wave, const int lane, GenShared& sh) {
    // Measured (profiles/r02_ab_general_packed.txt): +9-10 % at 512^2 (4 runs in flight, K = 8 tiles: a quarter of all
    // tiles are general), +1.5 % at 2048^2 (K = 10) -- and 3 % SLOWER at 4096^2 / 8192^2, where 4 % of the tiles are
    // general and the changed arm disturbs the register allocation of the air arm that shares its kernel (DESIGN.md 8.4
    // has two more cases of that).  So the large-grid tile (K = 12) keeps the scalar form -- unless a scene has MANY general
    // tiles (GP: a second instantiation of the merged kernel, chosen per geometry by Solver::applyGeometry when >= 8 % of the
    // tiles are general -- the 25 m rooms at 4096^2 / 8192^2, Mode B: 13-16 %, profiles/r05_modeB_general.txt).
    if constexpr (GP || (PV_GENERAL_PACKED == 1 ? K < 12 : PV_GENERAL_PACKED != 0))
        stepTileGeneral4Packed<K, RXI>(a, tile, wave, lane, sh);
    else
        stepTileGeneral4Scalar<K, RXI>(a, tile, wave, lane, sh);
}

// ---------------------------------------------------------------------------------------------------------------
// stacked air tile: W waves share one tall tile, x halo paid once per block
// ---------------------------------------------------------------------------------------------------------------
// The single-wave tile re-loads and re-computes a K-row halo above and below its RXI interior rows (36 of 60 rows are
// interior at K = 12), and the kernel is bound by the bytes its CUs move.  Here the W waves of a block
// own CONSECUTIVE row windows of one X-row tile and keep each other's boundary faces current through LDS, so the K-row
// x halo is loaded once per block (X = 189 of 213 rows interior at K = 12, NP = 27, W = 4): 2.8 B of CU-level traffic
// per cell-step instead of 3.7.
//
// Rows of the block, counted from the first loaded row (tile interior = rows [K, K+X)): wave w holds the R = 2*NP
// rows [w(R-1), w(R-1)+R) in the mirror-pair layout above, plus face vx_row[R] in the spare slot vx[0].y.  Its first
// row is the previous wave's last one: pr and vy of that row are advanced redundantly by both waves, so that the ONLY
// values a wave needs from its neighbours each step are two faces of vx --
//     vx_row[0]  (from the wave above: its vx_row[R-1])    and    vx_row[R]  (from the wave below: its vx_row[1]),
// both of which live in the sender's pair vx[1] and both of which land in the receiver's pair vx[0].  Per step and
// wave: one 8-byte LDS write, one workgroup barrier, two 4-byte LDS reads (slots double-buffered by step parity).
// The first wave's top K rows and the last wave's bottom K (+D) rows are the tile's x halo; what they hold decays
// one row per step exactly as in the single-wave tile.  Every wave advances all of its rows on every step (no
// trapezoid), so the waves of a block stay balanced and the time loop is a real loop: the kernel is a tenth of the
// unrolled tile kernel's code.
template <int K, int NP, int W, int X>
struct StackGeom {
    static constexpr int R = 2 * NP;
    static constexpr int L = W * (R - 1) + 1;   // rows loaded by the block
    static constexpr int D = L - X - 2 * K;     // halo rows beyond K at the bottom (X need not use every row)
    static constexpr int WI = 64 - 2 * K;
    static_assert(D >= 0, "tile taller than the block's windows");
    static_assert(R - 1 >= K + D + 1, "halo must fit the end waves");
};

// row r of a wave's window (LO <= r < HI, compile-time) -> register
template <int NP>
__device__ __forceinline__ float mirrorRow(const v2f (&f)[NP], int r) {
    return r < NP ? f[r].x : f[2 * NP - 1 - r].y;
}
template <int NP>
__device__ __forceinline__ float mirrorVxRow(const v2f (&vx)[NP], float vxS, int r) {
    return r < NP ? vx[r].x : (r == NP ? vxS : -vx[2 * NP - r].y);
}

template <int NP, int LO, int HI>
__device__ __forceinline__ void stackRecord(const v2f (&pr)[NP], rsrc_t rH, int hvoff, int hsoff, int hpitchB) {
#pragma unroll
    for (int r = LO; r < HI; ++r) bufStoreF(mirrorRow<NP>(pr, r), rH, hvoff, hsoff + r * hpitchB);
}

template <int NP, int LO, int HI>
__device__ __forceinline__ void stackStore(const v2f (&pr)[NP], const v2f (&vx)[NP], const v2f (&vy)[NP],
                                           float vxS, rsrc_t rPr, rsrc_t rVx, rsrc_t rVy, int voff, int soff,
                                           int pitchB) {
#pragma unroll
    for (int r = LO; r < HI; ++r) {
        const int so = soff + r * pitchB;
        bufStoreF(mirrorRow<NP>(pr, r), rPr, voff, so);
        bufStoreF(mirrorVxRow<NP>(vx, vxS, r), rVx, voff, so);
        bufStoreF(mirrorRow<NP>(vy, r), rVy, voff, so);
    }
}

// LDS of one block: exchange slots [parity][wave][lane] + one activity flag per wave
template <int W>
struct StackShared {
    v2f xch[2][W][64];
    int nz[W];
};

template <int K, int NP, int W, int X>
__device__ __forceinline__ void stepTileStack(const StepArgs& a, const int tile, const int wave, const int lane,
                                              StackShared<W>& sh) {
    using Gm = StackGeom<K, NP, W, X>;
    constexpr int R = Gm::R, D = Gm::D, WI = Gm::WI;
    const int ti = tile / a.nty;
    const int tj = tile - ti * a.nty;
    const int row0 = a.G - K + ti * X;
    const int col0 = a.G - K + tj * WI;
    const int ws = wave * (R - 1);  // first row of this wave's window
    const int voff = lane * 4;
    const int pitchB = a.pitch * 4;
    const int soffW = ((row0 + ws) * a.pitch + col0) * 4;
    const bool first = wave == 0, last = wave == W - 1;

    const rsrc_t rPrIn = makeRsrc(a.prIn, a.inBytes), rVxIn = makeRsrc(a.vxIn, a.inBytes),
                 rVyIn = makeRsrc(a.vyIn, a.inBytes);
    v2f pr[NP], vx[NP], vy[NP];
    float vxS = bufLoadF(rVxIn, voff, soffW + NP * pitchB);
    float faceR = 0.f;  // vx_row[R]: the next wave's vx_row[1]; below the last wave it is outside the block
    if (!last) faceR = bufLoadF(rVxIn, voff, soffW + R * pitchB);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int soT = soffW + i * pitchB, soB = soffW + (R - 1 - i) * pitchB;
        pr[i].x = bufLoadF(rPrIn, voff, soT);
        pr[i].y = bufLoadF(rPrIn, voff, soB);
        vy[i].x = bufLoadF(rVyIn, voff, soT);
        vy[i].y = bufLoadF(rVyIn, voff, soB);
        vx[i].x = bufLoadF(rVxIn, voff, soT);
        if (i > 0) vx[i].y = -bufLoadF(rVxIn, voff, soB + pitchB);
    }
    vx[0].y = -faceR;

    uint32_t nz = __float_as_uint(vxS) | __float_as_uint(faceR);
#pragma unroll
    for (int i = 0; i < NP; ++i)
        nz |= __float_as_uint(pr[i].x) | __float_as_uint(pr[i].y) | __float_as_uint(vx[i].x) |
              __float_as_uint(vx[i].y) | __float_as_uint(vy[i].x) | __float_as_uint(vy[i].y);
    const bool activeW = __ballot((nz & 0x7fffffffu) != 0u) != 0ull;
    if (lane == 0) sh.nz[wave] = activeW ? 1 : 0;
    __syncthreads();
    bool active = false;
#pragma unroll
    for (int w = 0; w < W; ++w) active = active || sh.nz[w] != 0;
    const bool writer = first && lane == 0;
    if (writer) a.nzOut[tile] = active ? 1 : 0;

    const DynParams dyn = *a.dyn;
    const int hti = ti - dyn.histTileX0, htj = tj - dyn.histTileY0;
    const bool inWin = hti >= 0 && hti < dyn.histTilesX && htj >= 0 && htj < dyn.histTilesY;
    const bool wasActive = a.record && a.tileFirst[tile] != INT_MAX;
    const bool rec = a.record && inWin && (active || wasActive || a.dense) && historyWanted(a, ti, tj);
    // tileFirst is read by every wave above and written here: the barrier inside the first step separates the two
    const bool setFirst = a.record && active && !wasActive && writer;
    if (a.record && active && !inWin && writer) atomicExch(a.errFlag, 1);

    const float C = a.courant;
    const bool inCols = lane >= K && lane < 64 - K;
    const bool recLane = rec && inCols;
    const float* hplane = a.hist + (long long)a.histSlot * a.histPlane;
    const int hpitchB = WI * 4;
    const int hsoffW = ((hti * a.dyn->histTilesY + htj) * X - K + ws) * hpitchB;  // + r*hpitchB; stored rows have -K + ws + r >= 0
    const int hvoff = (lane - K) * 4;

#pragma unroll 1
    for (int s = 0; s < a.nsteps; ++s) {
        mirrorPressureSweep<NP, 4, 0>(pr, vx, vy, vxS, C);
        mirrorVxSweep<NP, 4, 1>(pr, vx, vxS, C);  // pair 0 of vx is received, not computed
        sh.xch[s & 1][wave][lane] = vx[1];
        mirrorVySweep<NP, 4, 0>(pr, vy, C);
        // LDS only: the history stores of earlier steps stay in flight across the barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s == 0 && setFirst) atomicMin(&a.tileFirst[tile], a.t0);
        const float up = first ? 0.f : sh.xch[s & 1][first ? 0 : wave - 1][lane].y;
        const float dn = last ? 0.f : sh.xch[s & 1][last ? 0 : wave + 1][lane].x;
        vx[0] = v2f{-up, -dn};
        if (recLane) {  // pressure of this step, interior rows (air tiles never hold the listener)
            const rsrc_t rH = makeRsrc(hplane, a.histPlane * 4);
            if (first)
                stackRecord<NP, K, R>(pr, rH, hvoff, hsoffW, hpitchB);
            else if (last)
                stackRecord<NP, 1, R - K - D>(pr, rH, hvoff, hsoffW, hpitchB);
            else
                stackRecord<NP, 1, R>(pr, rH, hvoff, hsoffW, hpitchB);
        }
        hplane += a.histPlane;
    }

    const rsrc_t rPrOut = makeRsrc(a.prOut, a.planeBytes), rVxOut = makeRsrc(a.vxOut, a.planeBytes),
                 rVyOut = makeRsrc(a.vyOut, a.planeBytes);
    if (inCols) {
        if (first)
            stackStore<NP, K, R>(pr, vx, vy, vxS, rPrOut, rVxOut, rVyOut, voff, soffW, pitchB);
        else if (last)
            stackStore<NP, 1, R - K - D>(pr, vx, vy, vxS, rPrOut, rVxOut, rVyOut, voff, soffW, pitchB);
        else
            stackStore<NP, 1, R>(pr, vx, vy, vxS, rPrOut, rVxOut, rVyOut, voff, soffW, pitchB);
    }
}

// Position q inside an XCD's band of `bandRows` tile rows -> (row in band, tile column).
//   order 1      : row-major over the whole band
//   order 2      : column-major over the whole band
//   order H >= 4 : the band is cut into sub-bands of H tile rows, each walked column by column.  The ~256 tiles an
//                  XCD has in flight then form a compact H x (256/H) patch: a tile's vertical AND horizontal halo
//                  neighbours are in flight with it or a few tiles away, instead of a whole tile row (13 MB of
//                  traffic at 8192^2, three times the 4 MiB L2) away.
__device__ __forceinline__ bool bandPosition(const StepArgs& a, int q, int bandRows, int* r, int* tj) {
    if (a.tileOrder <= 1) {
        *r = q / a.nty;
        *tj = q - *r * a.nty;
        return *r < bandRows;
    }
    if (a.tileOrder == 2) {
        *tj = q / bandRows;
        *r = q - *tj * bandRows;
        return *tj < a.nty;
    }
    const int H = max(a.tileOrder, 4);
    const int per = H * a.nty;
    const int sb = q / per;
    const int r0 = sb * H;
    if (r0 >= bandRows) return false;
    const int h = min(H, bandRows - r0);
    const int qq = q - sb * per;
    *tj = qq / h;
    *r = r0 + (qq - *tj * h);
    return *tj < a.nty;
}

// block b (of the air part of a launch), wave w -> tile.  XCD x = b % 8 owns a contiguous span of the row-major tile
// sequence: ntiles/8 tiles each (order 1, balanced to one tile), or a band of whole tile rows (orders 2, >= 4).
__device__ __forceinline__ bool xcdTileAt(const StepArgs& a, int xcd, int q, int* ti, int* tj) {
    if (a.tileOrder == 3) {  // column strips: XCD x owns tile columns [x*cw, (x+1)*cw) and walks its strip row-major, so
                             // that a tile's vertical neighbours are cw tiles -- not a whole tile row of the grid -- away
        if (a.sweepReverse & 2) {
            // 2 x 4 regions instead of 8 strips (PVA_OPT_XCD_REGIONS): XCD x owns tile rows [x / 4 * rh, ...) x tile columns
            // [x % 4 * cw, ...) and walks its region row-major.  Twice as wide, a region has half as many lines on its sides that
            // the neighbouring XCD fetches too (11 % of all reads with strips of 13 tiles), and a vertical neighbour is still only
            // 26 tiles away: 4096^2 +0.5-4 %, 8192^2 +2 % (profiles/r04_xcd_regions.txt; 4 x 2 regions: slower)
            const int cw = (a.nty + 3) >> 2, rh = (a.ntx + 1) >> 1;
            const int c0 = (xcd & 3) * cw, w = min(cw, a.nty - c0);
            const int r0 = (xcd >> 2) * rh, h = min(rh, a.ntx - r0);
            if (w <= 0 || h <= 0) return false;
            const int r = q / w;
            if (r >= h) return false;
            *ti = r0 + ((a.sweepReverse & 1) ? h - 1 - r : r);
            *tj = c0 + (q - r * w);
            return true;
        }
        const int cw = (a.nty + 7) >> 3;
        const int c0 = xcd * cw, w = min(cw, a.nty - c0);
        if (w <= 0) return false;
        const int r = q / w;
        if (r >= a.ntx) return false;
        *ti = (a.sweepReverse & 1) ? a.ntx - 1 - r : r;
        *tj = c0 + (q - r * w);
        return true;
    }
    if (a.tileOrder <= 1) {
        const int per = (a.ntiles + 7) >> 3;
        const int t = xcd * per + q;
        if (q >= per || t >= a.ntiles) return false;
        *ti = t / a.nty;
        *tj = t - *ti * a.nty;
        return true;
    }
    const int ti0 = xcd * a.bandRows;
    const int bandRows = min(a.bandRows, a.ntx - ti0);
    if (bandRows <= 0) return false;
    int r;
    if (!bandPosition(a, q, bandRows, &r, tj)) return false;
    *ti = ti0 + r;
    return true;
}
__device__ __forceinline__ bool xcdTile(const StepArgs& a, int b, int wave, int* ti, int* tj) {
    return xcdTileAt(a, b & 7, (b >> 3) * 4 + wave, ti, tj);  // position inside the XCD's share
}

// air tiles: one wave per tile, 4 tiles per 256-thread block; tiles of the other class exit immediately.
// XCD-aware tile order: workgroup b is dispatched to XCD b % 8 (observed, speed only), and each XCD has a private
// 4 MiB L2.  XCD x therefore owns a contiguous band of tile rows and walks it column by column, so the tiles that
// share halo rows / columns are processed close together in time ON THE SAME L2 instead of being re-fetched
// through the fabric by eight different L2s.
template <int K, int RXI, int WPS, bool PACKED>
__global__ __launch_bounds__(256, WPS) void pv_step_air_kernel(const StepArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int ti, tj;
    if (a.tileOrder == 0) {  // linear: block b = tiles 4b..4b+3 in row-major order
        const int t = blockIdx.x * 4 + wave;
        if (t >= a.ntiles) return;
        ti = t / a.nty;
        tj = t - ti * a.nty;
    } else {
        if (!xcdTile(a, blockIdx.x, wave, &ti, &tj)) return;
    }
    const int tile = ti * a.nty + tj;
    const int cls = a.tileClass[tile];
    if (cls == 1) return;
    if (a.withPulse) {  // the tile(s) holding the listener are on the general kernel's list
        const int lr = a.dyn->lrow - (a.G - K + ti * RXI), lc = a.dyn->lcol - (a.G - K + tj * (64 - 2 * K));
        if (lr >= 0 && lr < RXI + 2 * K && lc >= 0 && lc < 64) return;
    }
    if constexpr (PACKED && (RXI + 2 * K) % 2 == 0) {
        airTilePacked<K, RXI>(a, tile, lane, cls == 2);
    } else {
        stepTile<K, RXI, RXI, false>(a, tile, 0, lane);
    }
}

// wall / edge / listener tiles, taken from a compact list; each tile is split over RXI/SUB waves
template <int K, int RXI, int SUB>
__global__ __launch_bounds__(256, 2) void pv_step_general_kernel(const StepArgs a) {
    constexpr int S = RXI / SUB;
    static_assert(S * SUB == RXI, "general-tile split must divide the tile");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int idx = blockIdx.x * 4 + wave;
    if (idx >= a.dyn->numGeneral * S) return;
    // few, long, latency-bound waves that run beside the air kernel: let them win VALU/issue arbitration
    __builtin_amdgcn_s_setprio(3);
    const int tile = __builtin_amdgcn_readfirstlane(a.generalList[idx / S]);
    stepTile<K, RXI, SUB, true>(a, tile, idx % S, lane);
}

// Merged form: ONE launch per K steps.  The first numGeneral blocks advance one general tile each (4 waves sharing
// its rows, stepTileGeneral4), every other block four air tiles (one wave each).  Versus the two-kernel / two-stream
// form this removes the cross-stream event hand-shake between every pair of launches; the general arm (~110 VGPRs)
// fits inside the air arm's register budget.  SUB is the slice height of the two-kernel form and unused here.
template <int K, int RXI, int WPS, int SUB, bool GP = false>
__global__ __launch_bounds__(256, WPS) void pv_step_merged_kernel(const StepArgs a) {
    __shared__ GenShared gsh;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gblocks = a.numGeneral;  // one block per general tile
    if ((int)blockIdx.x < gblocks) {
#if PV_PROBE_NOMEM >= 2  // (measurement builds: air tiles only; 3 = with their loads and stores)
        return;
#endif
        if ((int)blockIdx.x >= a.dyn->numGeneral) return;
        const int tile = __builtin_amdgcn_readfirstlane(a.generalList[blockIdx.x]);
        if (deadTileSkippable<K, RXI>(a, tile)) return;  // (block-uniform)
        // (s_setprio(3) for these few, long, barrier-bound waves: +2 % with one run in flight, -3 % with two -- round 5, not kept)
        stepTileGeneral4<K, RXI, GP>(a, tile, wave, lane, gsh);
        return;
    }
    const int b = blockIdx.x - gblocks;
    int ti, tj;
    if (!xcdTile(a, b, wave, &ti, &tj)) return;
    const int tile = ti * a.nty + tj;
    const int cls = a.tileClass[tile];
    if (cls == 1) return;
    if (a.withPulse) {
        const int lr = a.dyn->lrow - (a.G - K + ti * RXI), lc = a.dyn->lcol - (a.G - K + tj * (64 - 2 * K));
        if (lr >= 0 && lr < RXI + 2 * K && lc >= 0 && lc < 64) return;
    }
    if constexpr ((RXI + 2 * K) % 2 == 0) {
        airTilePacked<K, RXI>(a, tile, lane, cls == 2);
    } else {
        stepTile<K, RXI, RXI, false>(a, tile, 0, lane);
    }
}

// Batched form of the merged launch: blockIdx.y selects one of up to kBatchMax independent runs (identically
// configured solvers on one device, each with its own planes), so that B runs cost ONE launch per K steps instead
// of B launches on B streams.  Launch-bound grids (<= 2048^2) gain the most: a 512^2 run is 55 dependent launches of
// ~250 waves each, and the command processor serialises the dispatches of concurrent streams.  The body is the
// merged kernel's; the general-tile block count is padded to a multiple of 8 so that block b - gblocks still sits on
// XCD (b - gblocks) % 8.
template <int K, int RXI, int WPS, int SUB>
__global__ __launch_bounds__(256, WPS) void pv_step_batch_kernel(const BatchArgs ba) {
    __shared__ GenShared gsh;
    const StepArgs& a = ba.a[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gblocks = ba.gblocks;
    if ((int)blockIdx.x < gblocks) {
        if ((int)blockIdx.x >= a.dyn->numGeneral) return;
        const int tile = __builtin_amdgcn_readfirstlane(a.generalList[blockIdx.x]);
        if (deadTileSkippable<K, RXI>(a, tile)) return;  // (block-uniform)
        stepTileGeneral4<K, RXI>(a, tile, wave, lane, gsh);
        return;
    }
    const int b = blockIdx.x - gblocks;
    int ti, tj;
    if (!xcdTile(a, b, wave, &ti, &tj)) return;
    const int tile = ti * a.nty + tj;
    const int cls = a.tileClass[tile];
    if (cls == 1) return;
    if (a.withPulse) {
        const int lr = a.dyn->lrow - (a.G - K + ti * RXI), lc = a.dyn->lcol - (a.G - K + tj * (64 - 2 * K));
        if (lr >= 0 && lr < RXI + 2 * K && lc >= 0 && lc < 64) return;
    }
    if constexpr ((RXI + 2 * K) % 2 == 0) {
        airTilePacked<K, RXI, true>(a, tile, lane, cls == 2);
    } else {
        stepTile<K, RXI, RXI, false>(a, tile, 0, lane);
    }
}

// Stacked form of the merged launch: the first blocks advance the general tiles in SUB-row slices (one wave each,
// as above), every other block advances ONE air tile of X rows with its W = 4 waves (stepTileStack).
template <int K, int NP, int X, int SUB>
__global__ __launch_bounds__(256, 2) void pv_step_stack_kernel(const StepArgs a) {
    constexpr int W = 4;
    using Gm = StackGeom<K, NP, W, X>;
    __shared__ StackShared<W> sh;
    constexpr int S = X / SUB;
    static_assert(S * SUB == X, "general-tile split must divide the tile");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gblocks = (a.numGeneral * S + 3) / 4;
    if ((int)blockIdx.x < gblocks) {
        const int idx = blockIdx.x * 4 + wave;
        if (idx >= a.dyn->numGeneral * S) return;
        const int tile = __builtin_amdgcn_readfirstlane(a.generalList[idx / S]);
        stepTile<K, X, SUB, true>(a, tile, idx % S, lane);
        return;
    }
    const int b = blockIdx.x - gblocks;
    int ti, tj;
    if (!xcdTileAt(a, b & 7, b >> 3, &ti, &tj)) return;
    const int tile = ti * a.nty + tj;
    if (a.tileClass[tile] != 0) return;
    if (a.withPulse) {  // the tile(s) holding the listener are on the general list
        const int lr = a.dyn->lrow - (a.G - K + ti * X), lc = a.dyn->lcol - (a.G - K + tj * Gm::WI);
        if (lr >= 0 && lr < Gm::L && lc >= 0 && lc < 64) return;
    }
    stepTileStack<K, NP, W, X>(a, tile, wave, lane, sh);
}

// Per-tile class: 0 = every face code in the tile's loaded region is air|air, 1 = needs the general kernel.
// One wave per tile.  Tiles of class 1 are also appended to generalList (order irrelevant).
// face coefficients of cell (x, y) in an EMPTY grid (pv_coef_kernel with every cell air), as bits: what an edge tile must hold
__device__ __forceinline__ void emptyGridCoef(int x, int y, const Geometry& g, uint32_t* kxBits, uint32_t* kyBits) {
    constexpr uint32_t neg1 = 0xbf800000u, pos1 = 0x3f800000u, wall = 0u, air = kAirFaceBits;
    if (x < 0 || x >= g.NX || y < 0 || y >= g.NY) {
        *kxBits = *kyBits = wall;
        return;
    }
    if (x == 0)
        *kxBits = y < g.gy ? neg1 : wall;
    else if (x == g.gx)
        *kxBits = y < g.gy ? pos1 : wall;
    else
        *kxBits = y != g.gy ? air : wall;
    if (y == 0)
        *kyBits = x < g.gx ? neg1 : wall;
    else if (y == g.gy)
        *kyBits = x < g.gx ? pos1 : wall;
    else
        *kyBits = x != g.gx ? air : wall;
}

// Per-tile class.  0 = every face code in the tile's loaded region is air|air (air path).  2 = edge tile (EDGE
// configurations only): the codes are exactly those of an empty grid and the region does not reach the ghost row
// x = gx -- air path plus the edge overrides (stepTileAirMirror<EDGE>).  1 = everything else: general path; these
// are also appended to generalList (order irrelevant).  One wave per tile.
template <int K, int RXI, int ROWS = RXI + 2 * K, bool EDGE = false>
__global__ __launch_bounds__(256) void pv_tileclass_kernel(const FaceCoef* coef, uint8_t* tileClass,
                                                           int* generalList, int* generalCount, Geometry g,
                                                           int allowEdge) {
    constexpr int WI = 64 - 2 * K;
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= g.ntx * g.nty) return;
    const int ti = tile / g.nty, tj = tile - ti * g.nty;
    const int x0 = ti * RXI - K, y = tj * WI - K + lane;  // grid coordinates of the first loaded row / this lane
    const size_t base = (size_t)(g.G + x0) * g.pitch + (g.G + y);
    uint32_t any = 0, diff = 0;  // any: some face of the region is not air|air
    for (int r = 0; r < ROWS; ++r) {
        const FaceCoef c = coef[base + (size_t)r * g.pitch];
        const uint32_t bx = __float_as_uint(c.kx), by = __float_as_uint(c.ky);
        any |= (bx ^ kAirFaceBits) | (by ^ kAirFaceBits);
        if (EDGE) {
            uint32_t ex, ey;
            emptyGridCoef(x0 + r, y, g, &ex, &ey);
            diff |= (bx ^ ex) | (by ^ ey);
        }
    }
    const bool air = __ballot(any != 0u) == 0ull;
    const bool edge =
        EDGE && allowEdge && !air && __ballot(diff != 0u) == 0ull && !(x0 <= g.gx && g.gx < x0 + ROWS);
    if (lane == 0) {
        tileClass[tile] = air ? 0 : edge ? 2 : 1;
        if (!air && !edge) generalList[atomicAdd(generalCount, 1)] = tile;
    }
}

// Dead tiles: every interior cell is a wall cell whose x and y face coefficients are +0 (wall|wall, or a wall of admittance 0
// against air: the same update) or whose neighbour across the face is a wall cell as well.  Then pr = beta * (...)
// = 0 (FDTD.cpp:139) and both velocities are 0 (FDTD.cpp:165-168 with beta = beta_n = 0) whatever the halo holds, so a
// run that starts from zero fields never has to touch the tile: both buffer sets keep their zeros there.  One wave per tile.
__global__ __launch_bounds__(256) void pv_tiledead_kernel(const FaceCoef* coef, uint8_t* dead, int* count, Geometry g,
                                                          int K) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= g.ntx * g.nty) return;
    const int WI = 64 - 2 * K;
    const int ti = tile / g.nty, tj = tile - ti * g.nty;
    const size_t base = (size_t)(g.G + ti * g.rxi) * g.pitch + (g.G + tj * WI - K + lane);
    // Round 5: a face with a non-zero wall coefficient is as good as a +0 one when the cell across it is a wall cell too
    // (v = k * (0 + 0)): the ghost row / column of the grid (k = 1, FDTD.cpp:201-223) inside a thick outer wall.
    bool ok = true;
    if (lane >= K && lane < 64 - K)
        for (int r = 0; r < g.rxi; ++r) {
            const size_t at = base + (size_t)r * g.pitch;
            const FaceCoef c = coef[at];
            const bool solid = c.beta == 0.f;
            const bool fx = __float_as_uint(c.kx) == 0u || (c.kx == c.kx && coef[at - g.pitch].beta == 0.f);
            const bool fy = __float_as_uint(c.ky) == 0u || (c.ky == c.ky && coef[at - 1].beta == 0.f);
            ok = ok && solid && fx && fy;
        }
    const bool d = __ballot(!ok) == 0ull;
    if (lane == 0) {
        dead[tile] = d ? 1 : 0;
        if (d) atomicAdd(count, 1);
    }
}

void launchTileDead(const FaceCoef* coef, uint8_t* dead, int* count, const Geometry& g, int K, hipStream_t stream) {
    hipLaunchKernelGGL(pv_tiledead_kernel, dim3((g.ntx * g.nty + 3) / 4), dim3(256), 0, stream, coef, dead, count, g, K);
}

// general arm: may this block leave at once?  (a dead tile, unless the listener sits in its loaded region: the pulse is
// injected into the wall cell and swallowed there, but the final field still shows its last sample, FDTD.cpp:234)
template <int K, int RXI>
__device__ __forceinline__ bool deadTileSkippable(const StepArgs& a, int tile) {
    if (!a.tileDead || !a.tileDead[tile]) return false;
    if (!a.withPulse) return true;
    const int ti = tile / a.nty, tj = tile - ti * a.nty;
    const int lr = a.dyn->lrow - (a.G - K + ti * RXI), lc = a.dyn->lcol - (a.G - K + tj * (64 - 2 * K));
    return !(lr >= 0 && lr < RXI + 2 * K + 8 && lc >= 0 && lc < 64);
}

}  // namespace pva
// The arms that were built, shown bit-exact, measured and switched off -- row-streaming air segments (pv_seg.h), the
// persistent patch kernel (pv_patch.h), stacked tiles (PV_STACK_CONFIGS) -- are compiled only into the EXPERIMENTAL build
// of the library (make EXTRA=-DPV_EXPERIMENTAL BUILD=build_exp OUT=../libplaneverb_amd_exp.so), which their equivalence
// tests load; the product library carries the kernels its defaults and tuning options can reach.
#ifdef PV_EXPERIMENTAL
#include "pv_seg.h"
#include "pv_patch.h"
#endif
#include "pv_stream.h"
namespace pva {

// configurations whose sparse-emitter mode advances the forward sums inside the stencil (pv_stream.h)
#define PV_OPEN_CONFIGS(X) X(12, 36) X(10, 36) X(8, 24)

// the unpacked air kernel (PVA_OPT_PACKED_MATH = 0) is a validation form: experimental build only
bool unpackedAirOk() {
#ifdef PV_EXPERIMENTAL
    return true;
#else
    return false;
#endif
}

bool openConfigOk(int K, int rxi) {
#define X(k, r) \
    if (K == k && rxi == r) return true;
    PV_OPEN_CONFIGS(X)
#undef X
    return false;
}

void launchStreamClassify(const ClassifyArgs& c, hipStream_t stream) {
    const int n = c.ntx * c.nty;
    hipLaunchKernelGGL(pv_stream_classify_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, c);
}

void launchStreamIdle(const ClassifyArgs& c, int* idleHost, hipStream_t stream) {
    hipLaunchKernelGGL(pv_stream_idle_kernel, dim3(1), dim3(256), 0, stream, c, idleHost);
}

// open tiles of one sweep: one wave per listed half tile; the grid is sized for the capacity (2 halves x ntiles), the live
// count is read on the device
void launchStepOpen(int K, int rxi, const StepArgs& a, const OpenArgs& o, hipStream_t stream) {
    const int blocks = (2 * a.ntiles + 3) / 4;
#define X(k, r) \
    if (K == k && rxi == r) { \
        hipLaunchKernelGGL((pv_step_open_kernel<k, r>), dim3(blocks), dim3(256), 0, stream, a, o); \
        return; \
    }
    PV_OPEN_CONFIGS(X)
#undef X
}

#ifdef PV_EXPERIMENTAL
// configurations with a persistent patch kernel (pv_patch.h): the large-grid tile
#define PV_PATCH_CONFIGS(X) X(12, 36)

bool patchConfigOk(int K, int rxi) {
#define X(k, r) \
    if (K == k && rxi == r) return true;
    PV_PATCH_CONFIGS(X)
#undef X
    return false;
}

// the air tiles of one K-step sweep: `blocks` resident 512-thread workgroups (one per CU, a multiple of 8)
void launchStepPatch(int K, int rxi, const StepArgs& a, int blocks, hipStream_t stream) {
#define X(k, r) \
    if (K == k && rxi == r) { \
        hipLaunchKernelGGL((pv_step_patch_kernel<k, r>), dim3(blocks), dim3(512), 0, stream, a); \
        return; \
    }
    PV_PATCH_CONFIGS(X)
#undef X
}

// Segment form of the merged launch: the first blocks advance one general tile each (stepTileGeneral4, as in
// pv_step_merged_kernel), every other block four row-streaming air segments, one per wave (pv_seg.h).  XCD x = block % 8
// takes a contiguous eighth of the segment list, which the host sorts by (first row, tile column): segments that share
// y-halo columns stream down side by side on the same L2.
template <int K, int RXI, int NC, int WPS>
__global__ __launch_bounds__(256, WPS) void pv_step_seg_kernel(const StepArgs a) {
    __shared__ GenShared gsh;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gblocks = a.numGeneral;  // one block per general tile
    if ((int)blockIdx.x < gblocks) {
        if ((int)blockIdx.x >= a.dyn->numGeneral) return;
        const int tile = __builtin_amdgcn_readfirstlane(a.generalList[blockIdx.x]);
        if (deadTileSkippable<K, RXI>(a, tile)) return;  // (block-uniform)
        stepTileGeneral4<K, RXI>(a, tile, wave, lane, gsh);
        return;
    }
    const int b = blockIdx.x - gblocks;
    const int nseg = a.dyn->numSeg;
    const int per = (nseg + 7) >> 3;
    const int q = (b >> 3) * 4 + wave;
    const int idx = (b & 7) * per + q;
    if (q >= per || idx >= nseg) return;
    SegDesc sd = a.segList[idx];
    sd.row0 = __builtin_amdgcn_readfirstlane(sd.row0);
    sd.nrows = __builtin_amdgcn_readfirstlane(sd.nrows);
    sd.tj0 = __builtin_amdgcn_readfirstlane(sd.tj0);
    sd.w = __builtin_amdgcn_readfirstlane(sd.w);
    stepSegment<K, RXI, NC>(a, sd, lane);
}

// configurations with a segment kernel: (K, tile rows, columns per lane, waves per SIMD).  K = 8: 11 ring slots, 199
// VGPRs, two waves per SIMD.  K = 12: 16 ring slots need ~290 registers -- one wave per SIMD (256 VGPRs + AGPRs).
#define PV_SEG_CONFIGS(X) X(8, 40, 4, 2) X(12, 36, 4, 1)

int segConfigColumns(int K, int rxi) {
#define X(k, r, nc, w) \
    if (K == k && rxi == r) return nc;
    PV_SEG_CONFIGS(X)
#undef X
    return 0;
}

int segConfigMaxTileColumns(int K, int rxi) {
#define X(k, r, nc, w) \
    if (K == k && rxi == r) return SegGeom<k, nc>::WMAX;
    PV_SEG_CONFIGS(X)
#undef X
    return 0;
}

void launchStepSeg(int K, int rxi, const StepArgs& a, hipStream_t stream) {
    const int per = (a.numSeg + 7) / 8;
    const int blocks = a.numGeneral + 8 * ((per + 3) / 4);
#define X(k, r, nc, w) \
    if (K == k && rxi == r) { \
        hipLaunchKernelGGL((pv_step_seg_kernel<k, r, nc, w>), dim3(blocks), dim3(256), 0, stream, a); \
        return; \
    }
    PV_SEG_CONFIGS(X)
#undef X
}

#else  // product build: the experimental arms do not exist
bool patchConfigOk(int, int) { return false; }
void launchStepPatch(int, int, const StepArgs&, int, hipStream_t) {}
int segConfigColumns(int, int) { return 0; }
int segConfigMaxTileColumns(int, int) { return 0; }
void launchStepSeg(int, int, const StepArgs&, hipStream_t) {}
#endif

// positions an XCD's band needs under the chosen order (sub-bands are padded to whole multiples of H rows)
static int bandPositions(const StepArgs& a) {
    if (a.tileOrder <= 1) return (a.ntiles + 7) / 8;
    if (a.tileOrder == 3) return max(a.ntx * ((a.nty + 7) / 8), ((a.ntx + 1) / 2) * ((a.nty + 3) / 4));  // (8 strips / 2 x 4 regions)
    if (a.tileOrder < 4) return a.bandRows * a.nty;
    const int H = a.tileOrder;
    return ((a.bandRows + H - 1) / H) * H * a.nty;
}

template <int K, int RXI, int WPS, int SUB>
static void launchStepT(const StepArgs& a, hipStream_t stream, int which, hipStream_t stream2) {
    if (which == 16) {  // the general tiles of the merged launch alone (the air tiles go through pv_step_patch_kernel)
        if (a.numGeneral > 0)
            hipLaunchKernelGGL((pv_step_merged_kernel<K, RXI, WPS, SUB>), dim3(a.numGeneral), dim3(256), 0, stream, a);
        return;
    }
    if ((which & ~kStepGeneralPacked) == 4) {  // merged single launch: one block per general tile, then 4 air tiles per block
        // (All general blocks FIRST.  Spreading them over the launch in groups of eight, so that no round of the launch holds
        // general blocks only, was measured in round 5: 8-10 % slower at 4096^2 / 8192^2 in Mode B's geometry, 3-5 % in Mode A's.)
        // (PV_PROBE_GENERAL_ONLY: measurement aid -- the launch ends behind its general blocks, the air tiles are not advanced)
        static const bool genOnly = getenv("PV_PROBE_GENERAL_ONLY") != nullptr;
        const int blocks = a.numGeneral + (genOnly ? 0 : 8 * ((bandPositions(a) + 3) / 4));
        // (PV_PROBE_LDS = bytes of dynamic LDS per block: measurement aid, limits the blocks resident per CU)
        static const int probeLds = getenv("PV_PROBE_LDS") ? atoi(getenv("PV_PROBE_LDS")) : 0;
        if constexpr (K >= 12 && PV_GENERAL_PACKED == 1) {  // (smaller K: the packed general arm is the only one)
            if (which & kStepGeneralPacked) {
                hipLaunchKernelGGL((pv_step_merged_kernel<K, RXI, WPS, SUB, true>), dim3(blocks), dim3(256), probeLds, stream, a);
                return;
            }
        }
        hipLaunchKernelGGL((pv_step_merged_kernel<K, RXI, WPS, SUB>), dim3(blocks), dim3(256), probeLds, stream, a);
        return;
    }
    if (which & 1) {
        const int blocks = a.tileOrder == 0 ? (a.ntiles + 3) / 4 : 8 * ((bandPositions(a) + 3) / 4);
#ifdef PV_EXPERIMENTAL
        if (!a.packed)  // (the unpacked form: PVA_OPT_PACKED_MATH = 0, refused by Solver::init in the product build)
            hipLaunchKernelGGL((pv_step_air_kernel<K, RXI, WPS, false>), dim3(blocks), dim3(256), 0, stream, a);
        else
#endif
            hipLaunchKernelGGL((pv_step_air_kernel<K, RXI, WPS, true>), dim3(blocks), dim3(256), 0, stream, a);
    }
    if ((which & 2) && a.numGeneral > 0) {
        const int gblocks = (a.numGeneral * (RXI / SUB) + 3) / 4;
        hipLaunchKernelGGL((pv_step_general_kernel<K, RXI, SUB>), dim3(gblocks), dim3(256), 0,
                           stream2 ? stream2 : stream, a);
    }
}

template <int K, int RXI, int WPS, int SUB>
static void launchBatchT(const BatchArgs& ba, hipStream_t stream) {
    const int blocks = ba.gblocks + 8 * ((bandPositions(ba.a[0]) + 3) / 4);
    hipLaunchKernelGGL((pv_step_batch_kernel<K, RXI, WPS, SUB>), dim3(blocks, ba.n), dim3(256), 0, stream, ba);
}

template <int K, int RXI, int ROWS = RXI + 2 * K, bool EDGE = false>
static void launchTileClassT(const FaceCoef* coef, uint8_t* tileClass, int* list, int* count, const Geometry& g,
                             hipStream_t stream, int allowEdge) {
    const int blocks = (g.ntx * g.nty + 3) / 4;
    hipLaunchKernelGGL((pv_tileclass_kernel<K, RXI, ROWS, EDGE>), dim3(blocks), dim3(256), 0, stream, coef,
                       tileClass, list, count, g, allowEdge);
}

template <int K, int NP, int X, int SUB>
static void launchStackT(const StepArgs& a, hipStream_t stream) {
    const int gblocks = (a.numGeneral * (X / SUB) + 3) / 4;
    const int blocks = gblocks + 8 * bandPositions(a);  // one block per air tile
    hipLaunchKernelGGL((pv_step_stack_kernel<K, NP, X, SUB>), dim3(blocks), dim3(256), 0, stream, a);
}

// (K steps per launch, interior rows per tile, waves/SIMD bound of the air kernel, rows per general-tile slice)
// (K steps per launch, interior rows per tile, waves per SIMD, rows per slice of the two-kernel form's general tiles).  The
// PRODUCT library carries the tiles its defaults choose by grid size (pv_solver.cpp: (12, 36), (10, 36), (8, 24), (10, 20),
// (12, 12)) and the batched launches' tile (8, 40); every other (K, rows) -- tuning experiments of rounds 1-3, all bit-exact
// -- and the stacked tiles live in the experimental build only.
#define PV_PRODUCT_STEP_CONFIGS(X) X(8, 24, 3, 12) X(10, 36, 2, 9) X(12, 36, 2, 9) X(8, 40, 2, 10) X(12, 12, 3, 6) X(10, 20, 3, 10)
#ifdef PV_EXPERIMENTAL
#define PV_STEP_CONFIGS(X) \
    PV_PRODUCT_STEP_CONFIGS(X) \
    X(4, 32, 3, 8) X(4, 24, 4, 6) X(2, 28, 4, 7) X(1, 30, 4, 15) X(6, 28, 3, 14) X(3, 26, 4, 13) \
    X(8, 48, 2, 12) X(12, 40, 2, 10) X(12, 32, 2, 8) X(8, 44, 2, 11) X(10, 40, 2, 10) \
    X(9, 42, 2, 14) X(11, 36, 2, 9) X(9, 40, 2, 10)
// stacked tiles (K steps per launch, row pairs per wave, interior rows per tile, rows per general-tile slice); the
// tile's interior height doubles as the configuration's `rxi`
#define PV_STACK_CONFIGS(X) \
    X(12, 28, 196, 7) X(12, 30, 210, 10)
#define PV_BATCH_CONFIGS(X) PV_PRODUCT_STEP_CONFIGS(X) X(6, 28, 3, 14)
#else
#define PV_STEP_CONFIGS(X) PV_PRODUCT_STEP_CONFIGS(X)
#define PV_STACK_CONFIGS(X)
#define PV_BATCH_CONFIGS(X) PV_PRODUCT_STEP_CONFIGS(X)
#endif

void launchStep(int K, int rxi, const StepArgs& a, hipStream_t stream, int which, hipStream_t stream2) {
#define X(k, np, x, sub) \
    if (K == k && rxi == x) return launchStackT<k, np, x, sub>(a, stream);
    PV_STACK_CONFIGS(X)
#undef X
#define X(k, r, w, sub) \
    if (K == k && rxi == r) return launchStepT<k, r, w, sub>(a, stream, which, stream2);
    PV_STEP_CONFIGS(X)
#undef X
}

// configurations with a batched kernel: the defaults of every grid-size class (pv_solver.cpp) and the other
// merged-launch tiles of the tuning sweeps

bool batchConfigOk(int K, int rxi) {
#define X(k, r, w, sub) \
    if (K == k && rxi == r) return true;
    PV_BATCH_CONFIGS(X)
#undef X
    return false;
}

// configurations whose batched kernel has the edge-tile arm (the mirror-pair tiles)
bool edgeConfigOk(int K, int rxi) {
#define X(k, r, w, sub) \
    if (K == k && rxi == r) return edgeTilesOk<k, r>();
    PV_BATCH_CONFIGS(X)
#undef X
    return false;
}

void launchBatch(int K, int rxi, const BatchArgs& ba, hipStream_t stream) {
#define X(k, r, w, sub) \
    if (K == k && rxi == r) return launchBatchT<k, r, w, sub>(ba, stream);
    PV_BATCH_CONFIGS(X)
#undef X
}

void launchTileClass(int K, int rxi, const FaceCoef* coef, uint8_t* tileClass, int* list, int* count,
                     const Geometry& g, hipStream_t stream, bool allowEdge) {
#define X(k, np, x, sub) \
    if (K == k && rxi == x) \
        return launchTileClassT<k, x, StackGeom<k, np, 4, x>::L>(coef, tileClass, list, count, g, stream, 0);
    PV_STACK_CONFIGS(X)
#undef X
#define X(k, r, w, sub) \
    if (K == k && rxi == r) \
        return launchTileClassT<k, r, r + 2 * k, edgeTilesOk<k, r>()>(coef, tileClass, list, count, g, stream, \
                                                                      allowEdge ? 1 : 0);
    PV_STEP_CONFIGS(X)
#undef X
}

// rows a configuration's blocks load beyond rxi + 2K at the bottom (general-tile blocks: 0..3; stacked air tiles:
// their D); the guard band covers them
int stepConfigExtraRows(int K, int rxi) {
#define X(k, np, x, sub) \
    if (K == k && rxi == x) return StackGeom<k, np, 4, x>::D;
    PV_STACK_CONFIGS(X)
#undef X
#define X(k, r, w, sub) \
    if (K == k && rxi == r) return GenStackGeom<k, r>::D;
    PV_STEP_CONFIGS(X)
#undef X
    return 0;
}

bool stepConfigStacked(int K, int rxi) {
#define X(k, np, x, sub) \
    if (K == k && rxi == x) return true;
    PV_STACK_CONFIGS(X)
#undef X
    return false;
}

// configurations whose merged (single-launch) kernel allocates without (or with a handful of) spills in the air-tile
// arm; the others keep the two-kernel form
bool mergedConfigOk(int K, int rxi) {
    if (stepConfigStacked(K, rxi)) return true;
    return (K == 8 && rxi == 24) || (K == 4 && rxi == 32) || (K == 6 && rxi == 28) || K >= 10 || rxi >= 40;
}

bool stepConfigSupported(int K, int rxi) {
    if (stepConfigStacked(K, rxi)) return true;
#define X(k, r, w, sub) \
    if (K == k && rxi == r) return true;
    PV_STEP_CONFIGS(X)
#undef X
    return false;
}

// First node of every run: per-run parameters (listener cell, history window, general-tile list) from pinned host
// memory into HBM, per-tile bookkeeping back to "never non-zero".  Doing this in a kernel keeps the whole run a
// chain of kernel nodes (no DMA copy or memset node whose ordering against a graph replay would matter).
__global__ void pv_begin_run_kernel(BeginArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = min(a.dynHost->numGeneral, a.listCap);
    if (i < n) a.list[i] = a.listHost[i];
    if (a.segHost && i < min(a.dynHost->numSeg, a.segCap)) a.seg[i] = a.segHost[i];
    if (i == 0) {
        *a.dyn = *a.dynHost;
        *a.errFlag = 0;
    }
    if (i < a.nbands) a.dynBands[i] = a.dynBandsHost[i];
    if (i < a.nZero) a.zeroWords[i] = 0u;
    if (a.tileFirst && i < a.ntiles) {
        a.tileFirst[i] = a.tileFirstInit;
        a.nz0[i] = 0;
        a.nz1[i] = 0;
        if (a.tileOpen) a.tileOpen[i] = 1;
    }
}

__global__ void pv_zero_kernel(float4* p, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
void launchZero(float* p, long long n, hipStream_t stream) {
    const long long n4 = n / 4;
    hipLaunchKernelGGL(pv_zero_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, (float4*)p, n4);
}

void launchBeginRun(const BeginArgs& a, hipStream_t stream) {
    int n = a.ntiles > a.listCap ? a.ntiles : a.listCap;
    if (a.segHost && a.segCap > n) n = a.segCap;
    if (a.nZero > n) n = a.nZero;
    hipLaunchKernelGGL(pv_begin_run_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a);
}

void launchCoefs(const float* mat, FaceCoef* coef, const Geometry& g, hipStream_t stream) {
    dim3 grid((g.pitch + 255) / 256, g.rows);
    hipLaunchKernelGGL(pv_coef_kernel, grid, dim3(256), 0, stream, mat, coef, g);
}

void launchLaneSelfTest(float* out, hipStream_t stream) {
    hipLaunchKernelGGL(pv_lane_selftest_kernel, dim3(1), dim3(64), 0, stream, out);
}

// ---------------------------------------------------------------------------------------------------------------
// whole-grid-resident stencil for small grids
// ---------------------------------------------------------------------------------------------------------------

// Grids up to ~110^2 cells (the sandbox default is 71^2, the Unity demo 38^2) fit one CU: the three fields live in
// LDS for the whole run (160 KiB per CU on MI355X), each of the 1024 threads owns up to CPT cells whose state and
// face coefficients stay in registers, and ALL T time steps run inside one launch with two workgroup barriers per
// step.  Per step a cell reads its four neighbour values from LDS (vx[x+1], vy[y+1] for the pressure update,
// p[x-1], p[y-1] for the velocity update) and writes its three new values back; the pressure history row is stored
// to HBM as it is produced.  This replaces ~110 launch-bound kernel launches per run by one.
constexpr int kSmallThreads = 1024;
constexpr int kSmallCpt = 12;
constexpr int kSmallLdsBytes = 160 * 1024;

__host__ __device__ inline int smallLdsFloats(int NX, int NY) { return 3 * ((NX + 2) * NY + 2); }

bool smallGridFits(int NX, int NY) {
    return NX * NY <= kSmallThreads * kSmallCpt && smallLdsFloats(NX, NY) * 4 <= kSmallLdsBytes;
}

__global__ __launch_bounds__(kSmallThreads) void pv_small_grid_kernel(const SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NX = a.NX, NY = a.NY;
    const int cells = NX * NY;
    const int plane = (NX + 2) * NY + 2;  // one guard row above and below, +2 slack for the y+1 read of the last cell
    float* sp = smem + NY;                // cell (x, y) at sp[x*NY + y]; sp[-NY .. -1] is the zero guard row
    float* sx = sp + plane;
    float* sy = sx + plane;
    for (int i = threadIdx.x; i < 3 * plane; i += kSmallThreads) smem[i] = 0.f;

    const DynParams dyn = *a.dyn;
    const float C = a.courant;
    const int lcell = (dyn.lrow >= a.G && dyn.lcol >= a.G) ? (dyn.lrow - a.G) * NY + (dyn.lcol - a.G) : -1;

    float p[kSmallCpt], vx[kSmallCpt], vy[kSmallCpt], kx[kSmallCpt], ky[kSmallCpt];
    int hoff[kSmallCpt];
    unsigned beta = 0;
#pragma unroll
    for (int j = 0; j < kSmallCpt; ++j) {
        const int i = threadIdx.x + j * kSmallThreads;
        p[j] = vx[j] = vy[j] = 0.f;
        kx[j] = ky[j] = 0.f;
        hoff[j] = 0;
        if (i < cells) {
            const int x = i / NY, y = i - x * NY;
            const size_t ci = (size_t)(x + a.G) * a.pitch + (y + a.G);
            const FaceCoef c = a.coef[ci];
            kx[j] = c.kx;
            ky[j] = c.ky;
            if (c.beta != 0.f) beta |= 1u << j;
            hoff[j] = (int)histOffset(x + a.G - dyn.histRow0, y + a.G - dyn.histCol0, a.rxi, a.wi, dyn.histTilesY);
        }
    }
    __syncthreads();

    float* hplane = a.hist;
#pragma unroll 1
    for (int t = 0; t < a.T; ++t) {
        // pressure, FDTD.cpp:124-141 (reads past the last row / element hit the zero guard and are times beta = 0)
#pragma unroll
        for (int j = 0; j < kSmallCpt; ++j) {
            const int i = threadIdx.x + j * kSmallThreads;
            if (i < cells) {
                const float div = (sx[i + NY] - vx[j]) + (sy[i + 1] - vy[j]);
                const float pn = p[j] - C * div;
                p[j] = (beta >> j) & 1u ? pn : 0.f;
                sp[i] = p[j];
            }
        }
        __syncthreads();
        // velocities, FDTD.cpp:143-223 through the face coefficients; record, FDTD.cpp:226-230
#pragma unroll
        for (int j = 0; j < kSmallCpt; ++j) {
            const int i = threadIdx.x + j * kSmallThreads;
            if (i < cells) {
                const float pi = p[j];
                const float pxn = sp[i - NY];
                const float pyn = (i > 0) ? sp[i - 1] : 0.f;
                const float ax = vx[j] - C * (pi - pxn), wx = kx[j] * (pi + pxn);
                const float ay = vy[j] - C * (pi - pyn), wy = ky[j] * (pi + pyn);
                vx[j] = (kx[j] != kx[j]) ? ax : wx;
                vy[j] = (ky[j] != ky[j]) ? ay : wy;
                sx[i] = vx[j];
                sy[i] = vy[j];
                if (a.record) hplane[hoff[j]] = pi;
                if (i == lcell) p[j] = pi + a.pulse[t];  // soft source after the record, FDTD.cpp:234
            }
        }
        hplane += a.histPlane;
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < kSmallCpt; ++j) {
        const int i = threadIdx.x + j * kSmallThreads;
        if (i < cells) {
            const int x = i / NY, y = i - x * NY;
            const size_t o = (size_t)(x + a.G) * a.pitch + (y + a.G);
            a.prOut[o] = p[j];
            a.vxOut[o] = vx[j];
            a.vyOut[o] = vy[j];
        }
    }
}

void launchSmallGrid(const SmallArgs& a, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(pv_small_grid_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kSmallLdsBytes);
        attr = true;
    }
    const size_t lds = (size_t)smallLdsFloats(a.NX, a.NY) * 4;
    hipLaunchKernelGGL(pv_small_grid_kernel, dim3(1), dim3(kSmallThreads), lds, stream, a);
}

// ---------------------------------------------------------------------------------------------------------------
// impulse-response analysis
// ---------------------------------------------------------------------------------------------------------------

// pvLog10f / pvPowf: glibc-2.35-exact log10f and powf, see pv_libm.h

// CH samples (i0, i0-1, ..., i0-CH+1; those below startingPoint are skipped) of the backward Schroeder integration
// + regression sums, Analyzer.cpp:300-318.  Three passes over the chunk: the running energy (sequential, the
// reference's order), 10*log10f of each partial sum (independent of each other: branch-free, so the compiler
// interleaves the CH evaluations -- with one thread per cell and few waves this loop is bound by the latency of one
// evaluation, ~1800 cycles per sample when they ran one after the other), the two regression sums (sequential).
template <int CH>
__device__ __forceinline__ void rt60Chunk(const float (&pc)[CH], const int i0, const int startingPoint, float& edc,
                                          float& xysum, float& ysum) {
    float e[CH], y[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        edc = (i0 - k >= startingPoint) ? edc + pc[k] * pc[k] : edc;
        e[k] = edc;
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) y[k] = 10.f * pvLog10fNonNeg(e[k]);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const bool v = i0 - k >= startingPoint;
        xysum = v ? xysum + y[k] * (float)(i0 - k - startingPoint) : xysum;
        ysum = v ? ysum + y[k] : ysum;
    }
}

// One thread per result cell (X, Y); lanes along Y so every history read is a coalesced row segment of one
// recorded plane.  All sums are sequential float32 accumulations in the reference's order.
// vx / vy are not stored: they are re-derived from the pressure history with the stencil's own recurrence
// (v_t = v_{t-1} - C*(p_t[i] - p_t[n]) on air|air faces, k*(p_i + p_n) otherwise), bit-identical to the values
// the step kernel held.
// The analysis kernels run over the HISTORY WINDOW only (winRows x winCols cells from the window's origin, which
// lives in dyn so that the launch grid does not depend on the listener): a cell outside it cannot have been reached
// by the pulse.  What the reference computes for such a cell -- delay = FLT_MAX, direction = the unit vector from
// the listener to the cell itself -- is written for the whole map by pv_far_cells_kernel first.  (The first version
// launched one thread per grid cell: 67 M threads at 8192^2 to find the 0.6 M reached cells.)
__device__ __forceinline__ bool analysisWindowCell(const AnalyzeArgs& a, const DynParams& dyn, int* X, int* Y) {
    const int wc = blockIdx.x * blockDim.x + threadIdx.x, wr = blockIdx.y;
    if (wc >= a.winCols) return false;
    *X = dyn.histRow0 - a.G + wr;
    *Y = dyn.histCol0 - a.G + wc;
    return *X < a.gx && *Y < a.gy;
}

// Onset of every window cell (Analyzer.cpp:146-165: first sample whose |pressure| exceeds the audible threshold), round 5.
// Rounds 1-4 found it inside pv_encode_kernel, one thread per cell walking forward through time: the kernel lasted as long as
// its slowest thread, and the slowest threads were the SILENT cells -- air cells of a reached tile that never become audible
// (the air outside a closed room, in the tiles its walls cross): all T samples, 16 per memory round trip, 280 us of a 370 us
// kernel at 512^2 / T = 3179 -- and the decay-time pass could only start behind it.  Here the search is parallel IN TIME: a
// block is 64 consecutive cells of the tile-major plane (planeCell: one coalesced 256-byte read per plane) x 16 waves, wave w
// scans the samples t = tBeg + (16 j + w) 16 + k; the earliest hit per cell is kept with an LDS minimum, and a wave stops once
// no cell of the block can improve.  A silent cell costs T / 256 round trips instead of T / 16.  pv_encode_kernel and the
// decay-time kernels read the onset from the delay map, side by side on two streams.
#ifndef PV_ONSET_WAVES
#define PV_ONSET_WAVES 4
#endif
#ifndef PV_ONSET_SC
#define PV_ONSET_SC 32
#endif
constexpr int kOnsetWaves = PV_ONSET_WAVES, kOnsetSC = PV_ONSET_SC;
__global__ __launch_bounds__(64 * kOnsetWaves) void pv_onset_kernel(const AnalyzeArgs a) {
    __shared__ int found[64];
    if (analysisAborted(a)) return;
    const DynParams dyn = *a.dyn;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long g = (long long)blockIdx.x * 64 + lane;
    const PlaneCell c = planeCell(a, dyn, g);
    const int T = a.T;
    int tF = INT_MAX;
    bool live = c.inGrid;
    if (live) {
        tF = a.tileFirst[c.tile];
        // never reached by the pulse, or a wall (beta = 0: pr is identically zero, FDTD.cpp:139): no onset
        live = tF < T && a.coef[(size_t)(c.X + a.G) * a.pitch + (c.Y + a.G)].beta != 0.f;
    }
    const bool air = live;  // (an air cell of a reached tile: without an onset it counts as silent)
    if (a.labels) {
        // ... nor has a cell that no chain of air cells joins to the listener's (AnalyzeArgs::labels): not scanned
        const int lX = dyn.lrow - a.G, lY = dyn.lcol - a.G;
        const int mine = live ? a.labels[(size_t)c.X * a.labelNY + c.Y] : -1;
        const int theirs = (lX >= 0 && lX <= a.gx && lY >= 0 && lY < a.labelNY) ? a.labels[(size_t)lX * a.labelNY + lY] : -2;
        live = live && mine == theirs;
    }
    if (wave == 0) found[lane] = INT_MAX;
    if (a.stamp && blockIdx.x == 0 && threadIdx.x == 0) a.stamp[1] = wall_clock64();
    __syncthreads();
    // near box (AnalyzeArgs::box): the cells the PREVIOUS run reached get "no onset" back unless this run reaches them again --
    // every other cell of the map holds it already.  A cell of this window is its own lane's business (below, and where the
    // search ends); the previous box's cells OUTSIDE this window (the listener moved far) are dealt over the workgroups here.
    bool inPrev = false;
    if (a.box) {
        const int p0 = a.prevBox[0], p1 = a.prevBox[1], p2 = min(a.prevBox[2], a.gx - 1), p3 = min(a.prevBox[3], a.gy - 1);
        inPrev = c.inGrid && c.X >= p0 && c.X <= p2 && c.Y >= p1 && c.Y <= p3;
        if (p2 >= p0 && p3 >= p1) {  // (empty: INT_MAX / -1 -- no arithmetic on those)
            const int wr0 = dyn.histRow0 - a.G, wc0 = dyn.histCol0 - a.G;
            const int wr1 = wr0 + min(a.winRows, a.gx - wr0) - 1, wc1 = wc0 + min(a.winCols, a.gy - wc0) - 1;
            if (p0 < wr0 || p2 > wr1 || p1 < wc0 || p3 > wc1)  // (block-uniform)
                for (int r = p0 + (int)blockIdx.x; r <= p2; r += (int)gridDim.x)
                    for (int cc = p1 + (int)threadIdx.x; cc <= p3; cc += (int)blockDim.x)
                        if (r < wr0 || r > wr1 || cc < wc0 || cc > wc1) a.delay[r * a.gy + cc] = FLT_MAX;
        }
    }
    if (a.wholeWindow || a.box) {
        // no far-frame launch in front of this one: "no onset" for the cells that will not get one, and the count of active cells
        // (what pv_far_frame_kernel's first block does; the run's last kernel has left the other counters at zero)
        if (wave == 0 && c.inGrid && !live && (a.wholeWindow || inPrev)) a.delay[c.X * a.gy + c.Y] = FLT_MAX;
        if (blockIdx.x == 0 && wave == 1) {
            int n = 0;
            for (int i = lane; i < dyn.histTilesX * dyn.histTilesY; i += 64) {
                const int ti = dyn.histTileX0 + i / dyn.histTilesY, tj = dyn.histTileY0 + i % dyn.histTilesY;
                if (a.tileFirst[ti * a.nty + tj] < T) n += a.rxi * a.wi;
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) n += __shfl_xor(n, off);
            if (lane == 0) a.activeCount[0] = n;
        }
    }
    if (__ballot(live) == 0ull) {  // (the same lanes in every wave of the block: block-uniform)
        const unsigned long long ms = __ballot(air);
        if (wave == 0 && lane == 0 && ms) atomicAdd(a.activeCount + 3, __popcll(ms));
        return;
    }
    // A run starts from zero fields and the stencil moves a value by one cell per step along one axis (FDTD.cpp:124-199): the
    // recorded pressure of a cell at Manhattan distance m from the listener is exactly zero up to and including step m,
    // whatever the geometry.  The search starts there (cells far from the listener: half the samples between the tile's first
    // recorded step and the onset).  No listener in the grid: no pulse, no onset.
    const int m = abs(c.X - (dyn.lrow - a.G)) + abs(c.Y - (dyn.lcol - a.G));
    const int tS = live ? min(max(tF, m), T) : INT_MAX;
    int tBeg = tS;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) tBeg = min(tBeg, __shfl_xor(tBeg, off));
    tBeg = __builtin_amdgcn_readfirstlane(tBeg);  // (wave-uniform by value: scalar loop counter, scalar descriptors)
    const int voff = (int)g * 4;
    const int planeBytes = (int)(a.histPlane * 4);
    constexpr int SC = kOnsetSC;
#pragma unroll 1
    for (int t = tBeg + wave * SC; t < T; t += kOnsetWaves * SC) {
        const int best = found[lane];
        if (__ballot(live && best > t) == 0ull) break;  // nothing at or after t can be the first
        float pc[SC];
#pragma unroll
        for (int k = 0; k < SC; ++k) {
            const int tt = t + k;
            const bool want = live && tt >= tS && tt < T && tt < best;  // (a tile's history starts at its first recorded step)
            pc[k] = bufLoadF(makeRsrc(a.hist + (long long)min(tt, T - 1) * a.histPlane, planeBytes), want ? voff : 0x7fffffff, 0);
        }
        int hit = INT_MAX;
#pragma unroll
        for (int k = SC - 1; k >= 0; --k) hit = fabsf(pc[k]) > kAudibleThresholdDev ? t + k : hit;
        if (hit != INT_MAX) atomicMin(&found[lane], hit);
    }
    __syncthreads();
    if (wave != 0) return;
    const int onset = found[lane];
    if (live) {
        if (onset != INT_MAX) a.delay[c.X * a.gy + c.Y] = (float)onset;
        else if (a.wholeWindow || inPrev) a.delay[c.X * a.gy + c.Y] = FLT_MAX;
    }
    // reached cells of this run (bench / PvAmdTimings.reachedCells) and silent ones: one atomic each per block
    const bool reached = live && onset != INT_MAX;
    const unsigned long long mr = __ballot(reached), ms = __ballot(air && !reached);
    if (a.box && mr) {  // the reached cells' bounding box (AnalyzeArgs::box): four atomics per group with work
        int r0 = reached ? c.X : INT_MAX, c0 = reached ? c.Y : INT_MAX, r1 = reached ? c.X : -1, c1 = reached ? c.Y : -1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            r0 = min(r0, __shfl_xor(r0, off));
            c0 = min(c0, __shfl_xor(c0, off));
            r1 = max(r1, __shfl_xor(r1, off));
            c1 = max(c1, __shfl_xor(c1, off));
        }
        // (an atomic only where the group can still move a bound: in an open field 4 300 groups have work, and four atomics each on
        // the same four words took 330 us of the kernel -- same-address atomics are served one after the other; the plain reads
        // may be stale, which only means an atomic that changes nothing)
        if (lane == 0) {
            const int b0 = __hip_atomic_load(a.box + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int b1 = __hip_atomic_load(a.box + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int b2 = __hip_atomic_load(a.box + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int b3 = __hip_atomic_load(a.box + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (r0 < b0) atomicMin(a.box + 0, r0);
            if (c0 < b1) atomicMin(a.box + 1, c0);
            if (r1 > b2) atomicMax(a.box + 2, r1);
            if (c1 > b3) atomicMax(a.box + 3, c1);
        }
    }
    if (lane == 0) {
        if (mr) {
            atomicAdd(a.activeCount + 1, __popcll(mr));
            a.unitList[atomicAdd(a.activeCount + 4, 1)] = (int)blockIdx.x;  // this group of 64 cells has work for the passes behind
        }
        if (ms) atomicAdd(a.activeCount + 3, __popcll(ms));
    }
}

// dry gain, source directivity, low-pass cutoff (+ wet gain beside the lane-per-cell decay-time form) of the cells that have an
// onset, behind pv_onset_kernel: one wave per entry of the list of 64-cell groups with work (encodeWave, pv_analysis_dev.h)
__global__ __launch_bounds__(256) void pv_encode_kernel(const AnalyzeArgs a) {
    if (analysisAborted(a)) return;
    const DynParams dyn = *a.dyn;
    const int unit = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (unit >= a.activeCount[4]) return;
    const PlaneCell pc0 = planeCell(a, dyn, (long long)a.unitList[unit] * 64 + (threadIdx.x & 63));
    const float delay = pc0.inGrid ? a.delay[pc0.X * a.gy + pc0.Y] : FLT_MAX;
    const bool live = delay != FLT_MAX;  // no onset (Analyzer.cpp:160-165): the result record stays as it is
    if (__ballot(live) == 0ull) return;
    encodeWave<false>(a, dyn, pc0, live, live ? (int)delay : 0, rt60LanesPerCell(a, a.activeCount[1]) == 1);
}

// the same pass with L lanes per cell (encodeGroups, pv_analysis_dev.h): the small windows, where the longest walk is the kernel.
// A 64-cell group of the list is L waves: L / 4 workgroups
template <int L>
__global__ __launch_bounds__(256) void pv_encode_groups_kernel(const AnalyzeArgs a) {
    if (analysisAborted(a)) return;
    const DynParams dyn = *a.dyn;
    constexpr int BPU = L / 4;  // workgroups per group of 64 cells
    const int unit = blockIdx.x / BPU;
    if (unit >= a.activeCount[4]) return;
    const int ww = (blockIdx.x % BPU) * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const PlaneCell pc0 = planeCell(a, dyn, (long long)a.unitList[unit] * 64 + ww * (64 / L) + lane / L);
    const float delay = pc0.inGrid ? a.delay[pc0.X * a.gy + pc0.Y] : FLT_MAX;
    const bool live = delay != FLT_MAX;
    if (__ballot(live) == 0ull) return;
    encodeGroups<L, false>(a, dyn, pc0, lane % L, live, live ? (int)delay : 0);
}

// every cell of the map: no onset (Analyzer.cpp:64-68), listener direction = towards the cell itself (a walk that
// finds no neighbour with a smaller delay stays where it is, Analyzer.cpp:365-391).  The window's cells are
// overwritten by the kernels that follow.
// Upper bound of the cells the pulse reached: the cells of the window's tiles that were ever non-zero.  Computed by
// block 0 of pv_far_cells_kernel (the first launch of the analysis); its result chooses, on the device, between the
// cell form (inside pv_encode_kernel) and the wave form of the wet gain / decay time.
__device__ __forceinline__ void countActiveCells(const AnalyzeArgs& a) {
    __shared__ int part[256];
    const DynParams dyn = *a.dyn;
    int n = 0;
    for (int i = threadIdx.x; i < dyn.histTilesX * dyn.histTilesY; i += 256) {
        const int ti = dyn.histTileX0 + i / dyn.histTilesY, tj = dyn.histTileY0 + i % dyn.histTilesY;
        if (a.tileFirst[ti * a.nty + tj] < a.T) n += a.rxi * a.wi;
    }
    part[threadIdx.x] = n;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.activeCount[0] = part[0];
        a.activeCount[1] = 0;
        a.activeCount[3] = 0;
        a.activeCount[4] = 0;
    }
}

__global__ __launch_bounds__(256) void pv_far_cells_kernel(const AnalyzeArgs a) {
    if (analysisAborted(a)) return;  // (grid-uniform)
    if (blockIdx.x == 0 && a.tileFirst) countActiveCells(a);  // (before any thread leaves: the reduction has barriers)
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= a.gx * a.gy) return;
    a.delay[index] = FLT_MAX;
    storeDirection(a, index, index);
}

// The far-cell pass restricted to where something can have changed: blockIdx.z = 0 the previous run's window block,
// 1 this run's.  Every cell of both gets "no onset" and the default direction; the kernels that follow overwrite this
// run's reached cells.  All other cells of the map keep delay = FLT_MAX from the solver's creation / their own last reset,
// and their direction is made on demand (pv_far_dir_kernel, farDirectionOf).
__global__ __launch_bounds__(256) void pv_far_frame_kernel(const AnalyzeArgs a) {
    if (analysisAborted(a)) return;
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && a.tileFirst) countActiveCells(a);
    const DynParams dyn = *a.dyn;
    int r0, c0, nr, nc;
    if (blockIdx.z == 0) {
        r0 = a.prevR0, c0 = a.prevC0, nr = a.prevNR, nc = a.prevNC;
    } else {
        r0 = dyn.histRow0 - a.G, c0 = dyn.histCol0 - a.G;
        nr = min(a.winRows, a.gx - r0), nc = min(a.winCols, a.gy - c0);
    }
    const int wc = blockIdx.x * blockDim.x + threadIdx.x, wr = blockIdx.y;
    if (wc >= nc || wr >= nr) return;
    const int index = (r0 + wr) * a.gy + (c0 + wc);
    a.delay[index] = FLT_MAX;
    storeDirection(a, index, index);
}

// the cells a listener-direction pass covers: one thread per cell of the window block -- with a near box only the cells inside
// it (the workgroups of the other rows and column blocks leave at once: a closed room's box is ~90 of the window's 3 500
// workgroups.  A bounded grid whose workgroups stride over the box was measured first: fine for that room, but an open field's
// box IS the window, and three cells per thread one after the other tripled these latency-bound passes -- 8192^2 open field
// analysis 0.41 -> 0.75 ms)
template <class F>
__device__ __forceinline__ void forDirectionCells(const AnalyzeArgs& a, const DynParams& dyn, F&& f) {
    int X, Y;
    if (!analysisWindowCell(a, dyn, &X, &Y)) return;
    if (a.box && !nearBoxOf(a, dyn).holds(X, Y)) return;
    f(X * a.gy + Y);
}

// (farDirectionOf / isFarCell: pv_analysis.h)

// materialise the direction planes of the far cells (whole-map read-backs)
__global__ __launch_bounds__(256) void pv_far_dir_kernel(float* __restrict__ dirX, float* __restrict__ dirY, long long n,
                                                         const FarInfo f) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !isFarCell(f, i)) return;
    float ox, oy;
    farDirectionOf(f, i, &ox, &oy);
    dirX[i] = ox;
    dirY[i] = oy;
}

void launchFarDirections(float* res, long long n, const FarInfo& f, hipStream_t stream) {
    hipLaunchKernelGGL(pv_far_dir_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, res + 4 * n, res + 5 * n, n, f);
}

void launchFillDelay(float* delay, long long n, hipStream_t stream);

__device__ __forceinline__ void directionWalkCell(const AnalyzeArgs& a, const int index) {
    float loudness = a.out[index];
    int cur = index;
    float delay = FLT_MAX;
    const float samplingRate = (float)a.fs;
    const float wavelength = kCDev / (float)a.res;
    const float thresholdDist = 0.3f * wavelength;

    while (delay > kDelayCloseDev && loudness < kDistanceGainDev) {
        const int r = cur / a.gy, c = cur - r * a.gy;
        float bestLoud = 0.f, bestDelay = FLT_MAX;
        for (int i = 0; i < 8; ++i) {
            const int nr = r + neighbourDr(i), nc = c + neighbourDc(i);
            if (nr < 0 || nc < 0 || nr >= a.gx || nc >= a.gy) continue;
            const int ni = nr * a.gy + nc;
            const float occ = a.out[ni];
            const float d = a.delay[ni];
            if (occ == 0.f) continue;               // Analyzer.cpp:372 (the (unsigned)delay test never fires)
            if (d < bestDelay && occ > 0.f) {       // strict <: first neighbour wins ties
                bestLoud = occ;
                cur = ni;                           // kept even if the step is rejected below (Analyzer.cpp:377)
                bestDelay = d;
            }
        }
        if (bestDelay == FLT_MAX || bestDelay >= delay) break;
        delay = bestDelay;
        loudness = bestLoud;
        // line-of-sight test, Analyzer.cpp:393-411
        const float geodesic = kCDev * bestDelay / samplingRate;
        const int r2 = cur / a.gy, c2 = cur - r2 * a.gy;
        const float tx = (float)r2 * a.dx - a.lx, ty = (float)c2 * a.dx - a.lz;
        const float euclid = sqrtf((tx * tx) + (ty * ty));
        if (fabsf(geodesic - euclid) < thresholdDist) break;
    }
    storeDirection(a, index, cur);
}
__global__ __launch_bounds__(256) void pv_direction_kernel(const AnalyzeArgs a) {
    if (analysisAborted(a)) return;
    const DynParams dyn = *a.dyn;
    forDirectionCells(a, dyn, [&](const int index) { directionWalkCell(a, index); });
}

// listener direction by pointer jumping: the per-cell steps are dirInitCell / dirJumpCell / dirFinalCell (pv_analysis_dev.h)
__global__ __launch_bounds__(256) void pv_dir_init_kernel(const AnalyzeArgs a, int* J) {
    if (analysisAborted(a)) return;
    const DynParams dyn = *a.dyn;
    forDirectionCells(a, dyn, [&](const int p) { dirInitCell<false>(a, dyn, J, p); });
}

__global__ __launch_bounds__(256) void pv_dir_jump_kernel(const AnalyzeArgs a, int* J) {
    if (analysisAborted(a)) return;
    const DynParams dyn = *a.dyn;
    forDirectionCells(a, dyn, [&](const int p) { dirJumpCell<false>(a, dyn, J, p); });
}

__global__ __launch_bounds__(256) void pv_dir_final_kernel(const AnalyzeArgs a, const int* J) {
    if (analysisAborted(a)) return;
    const DynParams dyn = *a.dyn;
    forDirectionCells(a, dyn, [&](const int p) { dirFinalCell<false>(a, dyn, J, p); });
}

static dim3 analysisWindowGrid(const AnalyzeArgs& a) { return dim3((a.winCols + 255) / 256, a.winRows); }

static void launchDirectionJump(const AnalyzeArgs& a, int* J, hipStream_t stream) {
    const dim3 grid = analysisWindowGrid(a), block(256);
    hipLaunchKernelGGL(pv_dir_init_kernel, grid, block, 0, stream, a, J);
    for (int i = 0; i < dirJumpPasses(a.T); ++i) hipLaunchKernelGGL(pv_dir_jump_kernel, grid, block, 0, stream, a, J);
    hipLaunchKernelGGL(pv_dir_final_kernel, grid, block, 0, stream, a, J);
}

__global__ void pv_fill_delay_kernel(float* delay, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) delay[i] = FLT_MAX;  // Analyzer.cpp:64-68
}

// SoA result planes -> the reference's array of PlaneverbOutput structs (AnalyzerResult, Analyzer.h:11-22)
__global__ void pv_pack_results_kernel(const float* __restrict__ res, long long n, float* __restrict__ res8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 lo = make_float4(res[i], res[n + i], res[2 * n + i], res[3 * n + i]);
    float4 hi = make_float4(res[4 * n + i], res[5 * n + i], res[6 * n + i], res[7 * n + i]);
    reinterpret_cast<float4*>(res8)[2 * i] = lo;
    reinterpret_cast<float4*>(res8)[2 * i + 1] = hi;
}

// the nr x nc block of the result map whose first cell is (r0, c0) as AoS records (the live module publishes only
// the history window's block of every iteration: everything outside it is stale values + a closed-form direction)
__global__ void pv_pack_window_kernel(const float* __restrict__ res, long long n, int gy, int r0, int c0, int nr, int nc,
                                      float* __restrict__ out8, const FarInfo f) {
    const int wc = blockIdx.x * blockDim.x + threadIdx.x, wr = blockIdx.y;
    if (wc >= nc || wr >= nr) return;
    const long long i = (long long)(r0 + wr) * gy + (c0 + wc);
    const long long o = (long long)wr * nc + wc;
    float4 lo = make_float4(res[i], res[n + i], res[2 * n + i], res[3 * n + i]);
    float4 hi = make_float4(res[4 * n + i], res[5 * n + i], res[6 * n + i], res[7 * n + i]);
    if (isFarCell(f, i)) farDirectionOf(f, i, &hi.x, &hi.y);  // (a far cell: its direction is not in the planes)
    reinterpret_cast<float4*>(out8)[2 * o] = lo;
    reinterpret_cast<float4*>(out8)[2 * o + 1] = hi;
}

void launchPackWindow(const float* res, long long n, int gy, int r0, int c0, int nr, int nc, float* out8, const FarInfo& far,
                      hipStream_t stream) {
    if (nr <= 0 || nc <= 0) return;
    hipLaunchKernelGGL(pv_pack_window_kernel, dim3((unsigned)((nc + 255) / 256), (unsigned)nr), dim3(256), 0, stream, res,
                       n, gy, r0, c0, nr, nc, out8, far);
}

// one cell of the result map -> 8 floats in pinned host memory (Analyzer::GetResponseResult, Analyzer.cpp:106-116)
__global__ void pv_gather_output_kernel(const float* __restrict__ res, long long n, long long cell, float* out8,
                                        const FarInfo f) {
    if (threadIdx.x < 8) {
        float v = res[threadIdx.x * n + cell];
        if ((threadIdx.x == 4 || threadIdx.x == 5) && isFarCell(f, cell)) {  // a far cell: its direction is not in the planes
            float ox, oy;
            farDirectionOf(f, cell, &ox, &oy);
            v = threadIdx.x == 4 ? ox : oy;
        }
        out8[threadIdx.x] = v;
    }
}

void launchGatherOutput(const float* res, long long n, long long cell, float* out8Host, const FarInfo& f, hipStream_t stream) {
    hipLaunchKernelGGL(pv_gather_output_kernel, dim3(1), dim3(64), 0, stream, res, n, cell, out8Host, f);
}

// the registered output queries of a run (PvAmdSetOutputQueries): nq result cells -> nq x 8 floats in pinned host
// memory, enqueued behind the analysis so that the caller's one stream sync also delivers the outputs.
// cells[] lives in pinned host memory too (cell < 0: position outside the result map, left to the host's sentinel)
__global__ void pv_gather_queries_kernel(const float* __restrict__ res, long long n, const long long* cells, int nq,
                                         float* out, const FarInfo f) {
    const int q = threadIdx.x >> 3, k = threadIdx.x & 7;
    if (q < nq) {
        const long long c = cells[q];
        float v = c >= 0 ? res[k * n + c] : 0.f;
        if (c >= 0 && (k == 4 || k == 5) && isFarCell(f, c)) {
            float ox, oy;
            farDirectionOf(f, c, &ox, &oy);
            v = k == 4 ? ox : oy;
        }
        out[q * 8 + k] = v;
    }
}

void launchGatherQueries(const float* res, long long n, const long long* cellsHost, int nq, float* outHost, const FarInfo& f,
                         hipStream_t stream) {
    hipLaunchKernelGGL(pv_gather_queries_kernel, dim3(1), dim3(512), 0, stream, res, n, cellsHost, nq, outHost, f);
}

void launchPackResults(const float* res, long long n, float* res8, hipStream_t stream) {
    hipLaunchKernelGGL(pv_pack_results_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, res, n, res8);
}

// One row of the history window over all T steps as a dense [T][histPitch] array (zeros where the tile had not been
// reached yet): what the slab BELOW this one needs for the vx recurrence of its first row (AnalyzeArgs::histAbove).
__global__ void pv_hist_row_kernel(const AnalyzeArgs a, int X, float* __restrict__ out) {
    const int wc = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (wc >= a.histPitch || t >= a.T) return;
    const DynParams dyn = *a.dyn;
    const int pcol = dyn.histCol0 + wc, prow = X + a.G;
    float v = 0.f;
    const int ti = X / a.rxi, tj = (pcol - a.G) / a.wi;
    const int wti = ti - dyn.histTileX0, wtj = tj - dyn.histTileY0;
    if (wc < a.winCols && pcol >= a.G && wti >= 0 && wti < dyn.histTilesX && wtj >= 0 && wtj < dyn.histTilesY &&
        t >= a.tileFirst[ti * a.nty + tj])
        v = a.hist[(long long)t * a.histPlane + histOffset(prow - dyn.histRow0, wc, a.rxi, a.wi, dyn.histTilesY)];
    out[(long long)t * a.histPitch + wc] = v;
}

void launchHistRow(const AnalyzeArgs& a, int X, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(pv_hist_row_kernel, dim3((a.histPitch + 255) / 256, a.T), dim3(256), 0, stream, a, X, out);
}

// nplanes planes: the nr x nc block at (sr0, sc0) of src planes (row pitch spitch, plane stride sstride) into the block at
// (dr0, dc0) of dst planes -- a slab's part of the history window into the whole grid's result / delay maps
__global__ void pv_copy_block_kernel(const float* __restrict__ src, long long sstride, int spitch, int sr0, int sc0,
                                     float* __restrict__ dst, long long dstride, int dpitch, int dr0, int dc0, int nr,
                                     int nc, const int* srcPlanes, const int* dstPlanes, const unsigned* abortWord) {
    if (abortWord && *abortWord != 0u) return;  // (AnalyzeArgs::abortWord)
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    const int ks = srcPlanes ? srcPlanes[blockIdx.z] : blockIdx.z, kd = dstPlanes ? dstPlanes[blockIdx.z] : blockIdx.z;
    if (c >= nc || r >= nr) return;
    dst[kd * dstride + (long long)(dr0 + r) * dpitch + dc0 + c] = src[ks * sstride + (long long)(sr0 + r) * spitch + sc0 + c];
}

void launchCopyBlock(const float* src, long long sstride, int spitch, int sr0, int sc0, float* dst, long long dstride,
                     int dpitch, int dr0, int dc0, int nr, int nc, int nplanes, const int* srcPlanesDev,
                     const int* dstPlanesDev, hipStream_t stream, const unsigned* abortWord) {
    if (nr <= 0 || nc <= 0) return;
    hipLaunchKernelGGL(pv_copy_block_kernel, dim3((unsigned)((nc + 255) / 256), (unsigned)nr, (unsigned)nplanes), dim3(256),
                       0, stream, src, sstride, spitch, sr0, sc0, dst, dstride, dpitch, dr0, dc0, nr, nc, srcPlanesDev,
                       dstPlanesDev, abortWord);
}

// Slab decomposition (pv_slabs.cpp): after a K-step launch a slab PUSHES the K rows next to each of its boundaries into the
// neighbour's guard band -- one launch for all six blocks (3 planes x 2 directions) instead of six hipMemcpyAsync on the
// receiver's stream (each a 5 us copy kernel of its own: 20-85 us per sweep in round 2).  Every block is K whole padded
// rows = a contiguous run of floats (a multiple of 64); the destination may live on another device (peer access is
// enabled by the group: plain stores over xGMI).
struct HaloPushArgs {
    const float* src[6];
    float* dst[6];
    long long n;  // floats per block
    HaloHandoff h;
};
// Device-side hand-off between slabs on ONE device (HaloHandoff, pv_device.h).  A cross-queue event wait per slab and sweep
// costs ~50 us on this runtime (the waiting queue is parked until the command processor looks at it again: a 2048^2 run with two
// slabs took 95 us per sweep for 37 us of stencil, profiles/r04_slabs.txt); a word in memory costs 2-3 us.  So the push kernel
// of sweep li (a) copies its rows, (b) fence + count: the block that completes the count raises the words its neighbours
// look at to li + 1, (c) that same block then waits until the slab's own words -- raised by the neighbours' pushes of the
// same sweep -- have reached li + 1.  The slab's next step launch follows in stream order: behind the neighbours' halos,
// without an event.  (Write-after-read: a neighbour's push of sweep li comes behind its step li, which came behind its wait
// for THIS slab's push li - 1, which came behind this slab's step li - 1 -- the last reader of the guard rows it overwrites.)
// One wave spins, bounded; the others leave: the neighbours' kernels never lack a place to run.
__global__ __launch_bounds__(256) void pv_halo_push_kernel(const HaloPushArgs h) {
    typedef unsigned int v4u __attribute__((__vector_size__(16)));
    const int b = blockIdx.y;
    const bool hand = h.h.count != nullptr;
    if (h.dst[b]) {
        // (with the hand-off the rows are written THROUGH to memory -- sc1 -- so that nothing has to be flushed before the word
        // is raised: a release fence here writes back and invalidates the whole L2 under the other slab's running step kernel,
        // measured 65 us per sweep)
        const rsrc_t rs = makeRsrc(h.src[b], h.n * 4), rd = makeRsrc(h.dst[b], h.n * 4);
        const int n4 = (int)(h.n >> 2);
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, 0);
            if (hand)
                __builtin_amdgcn_raw_buffer_store_b128(v, rd, i * 16, 0, 16 /* sc1 */);
            else
                __builtin_amdgcn_raw_buffer_store_b128(v, rd, i * 16, 0, 0);
        }
    }
    if (!hand) return;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this thread's rows are in memory
    __syncthreads();
    if (threadIdx.x != 0) return;
    const unsigned nblocks = gridDim.x * gridDim.y;
    if (__hip_atomic_fetch_add(h.h.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u != h.h.seq * nblocks) return;
    for (int i = 0; i < 2; ++i)
        if (h.h.raise[i]) __hip_atomic_store(h.h.raise[i], h.h.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = 0; i < 2; ++i) {
        if (!h.h.await[i]) continue;
        int spins = 0;
        while (__hip_atomic_load(h.h.await[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < h.h.seq) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 19)) {  // (~0.5 s: a neighbour whose launches do not run beside this one must not hang the device;
                                        // SlabGroup::run then repeats the run with events -- the abort word keeps this run's
                                        // analysis away from the result maps)
                atomicExch(h.h.err, 5);
                if (h.h.abortWord) __hip_atomic_store(h.h.abortWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
    }
}

void launchHaloPush(const float* const src[6], float* const dst[6], long long n, const HaloHandoff& hand, hipStream_t stream) {
    HaloPushArgs h;
    for (int i = 0; i < 6; ++i) {
        h.src[i] = src[i];
        h.dst[i] = dst[i];
    }
    h.n = n;
    h.h = hand;
    const unsigned bx = (unsigned)std::min<long long>(64, (n / 4 + 255) / 256);
    hipLaunchKernelGGL(pv_halo_push_kernel, dim3(bx, 6), dim3(256), 0, stream, h);
}

// far cells of the whole map (delay = FLT_MAX, default listener direction): the first analysis launch
void launchFarCells(const AnalyzeArgs& a, hipStream_t stream) {
    const int n = a.gx * a.gy;
    hipLaunchKernelGGL(pv_far_cells_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a);
}

// onset, dry gain, source directivity, lowpass, wet gain, decay time of the window's cells (everything but the listener
// direction, which needs the delay / occlusion maps of the WHOLE window: launchAnalysisDirection)
void launchOnset(const AnalyzeArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(pv_onset_kernel, dim3((unsigned)((a.histPlane + 63) / 64)), dim3(64 * kOnsetWaves), 0, stream, a);
}
void launchEncode(const AnalyzeArgs& a, hipStream_t stream) {
    // lanes per cell by the window's size (an upper bound of the cells with work): sixteen where a cell's walk IS the kernel's
    // duration, four up to the sizes the four-lane decay-time form serves, one (which then also takes the wet gain beside the
    // lane-per-cell decay-time form) above
    const unsigned units = (unsigned)((a.histPlane + 63) / 64);
    if (a.rt60Lanes != 1 && a.histPlane <= 16384)
        hipLaunchKernelGGL(pv_encode_groups_kernel<16>, dim3(units * 4), dim3(256), 0, stream, a);
    else if (a.rt60Lanes != 1 && a.histPlane <= kRt60TileMinCells)
        hipLaunchKernelGGL(pv_encode_groups_kernel<4>, dim3(units), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(pv_encode_kernel, dim3((unsigned)((a.histPlane + 255) / 256)), dim3(256), 0, stream, a);
}
// wet gain + decay time (pv_rt60.hip): the form is decided on the device from the number of cells with an onset
void launchRt60(const AnalyzeArgs& a, hipStream_t stream) { launchRt60Forms(a, stream); }

void launchAnalysisCells(const AnalyzeArgs& a, hipStream_t stream) {
    launchOnset(a, stream);
    launchEncode(a, stream);
    launchRt60(a, stream);
}

void launchAnalysisDirection(const AnalyzeArgs& a, hipStream_t stream) {
    const dim3 grid = analysisWindowGrid(a);
    // (the whole pass in ONE workgroup with the table in LDS -- no kernel boundaries between init, jumps and final -- was built and
    // measured for the windows of up to 16 384 cells: 25-80 us slower, one CU's memory pipeline does the 17 loads per cell of
    // init and final for everybody: docs/experiments/fused_analysis.md)
    if (a.dirJump)
        launchDirectionJump(a, a.dirScratch, stream);
    else
        hipLaunchKernelGGL(pv_direction_kernel, grid, dim3(256), 0, stream, a);
}

void launchFillDelay(float* delay, long long n, hipStream_t stream) {
    hipLaunchKernelGGL(pv_fill_delay_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, delay, (int)n);
}

void launchAnalysisFar(const AnalyzeArgs& a, hipStream_t stream) {
    if (a.wholeWindow || a.box) return;  // (no far-cell pass: pv_onset_kernel does what is left of it)
    const int n = a.gx * a.gy;
    if (a.lazyFar) {
        const int nr = max(a.prevNR, min(a.winRows, a.gx)), nc = max(a.prevNC, min(a.winCols, a.gy));
        hipLaunchKernelGGL(pv_far_frame_kernel, dim3((unsigned)((max(nc, 1) + 255) / 256), (unsigned)max(nr, 1), 2), dim3(256), 0,
                           stream, a);
    } else {
        hipLaunchKernelGGL(pv_far_cells_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a);
    }
}

void launchAnalysis(const AnalyzeArgs& a, hipStream_t stream) {
    launchAnalysisFar(a, stream);
    launchAnalysisCells(a, stream);
    // listener direction: the plain walk where walks are short (small windows: rooms, the sandbox's grids), pointer
    // jumping where a window is wide enough for hundreds of steps (a dozen tiny launches, path-length independent)
    launchAnalysisDirection(a, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// streaming analysis (sparse-emitter mode)
// ---------------------------------------------------------------------------------------------------------------
// With T in the tens of thousands (a 25 m scene at a 4096^2 grid: T = 25432) the full pressure history cannot be
// kept.  What needs ALL of a cell's samples is only the wet gain and the RT60 regression; onset, dry energy and
// flux are forward sums that close N_dry samples after the onset.  So the history becomes a ring of `ring` planes;
// after every `ring` steps pv_stream_accum_kernel advances the forward sums of every open cell in the reference's
// sample order (state carried in per-cell planes), and the pressure of the REGISTERED emitter cells is copied to
// per-emitter traces, from which wet gain and RT60 are computed at the end exactly as pv_encode_kernel does.

__global__ __launch_bounds__(256) void pv_stream_accum_kernel(const AnalyzeArgs a) {
    int X, Y;
    if (a.ringList) {
        // forward sums of the air tiles inside the stencil (pv_stream.h): this pass serves only the listed tiles (walls, grid
        // edges, listener, registered emitters) -- blockIdx.x = list entry * chunks + 256-cell chunk of the tile (a grid's y
        // extent stops at 65 535: a 16k^2 scene with many wall tiles has more list entries than that)
        const int chunks = (a.rxi * a.wi + 255) / 256;
        const int entry = blockIdx.x / chunks;
        const int t = a.ringList[entry];
        const int idx = (blockIdx.x - entry * chunks) * blockDim.x + threadIdx.x;
        if (idx >= a.rxi * a.wi) return;
        const int ti = t / a.nty, r = idx / a.wi;
        X = ti * a.rxi + r;
        Y = (t - ti * a.nty) * a.wi + (idx - r * a.wi);
    } else {
        Y = blockIdx.x * blockDim.x + threadIdx.x;
        X = blockIdx.y;
    }
    if (Y >= a.gy || X >= a.gx) return;
    const int s = X * a.gy + Y;
    const DynParams dyn = *a.dyn;
    const int tile = (X / a.rxi) * a.nty + (Y / a.wi);
    // tiles whose sums advance inside the stencil (pv_stream.h) are not this pass's business
    if (a.fuseClass && !a.ringList &&
        fusedTile(a.fuseClass, a.fuseEmit, dyn, X / a.rxi, Y / a.wi, a.nty, a.G, a.fuseK, a.rxi, a.wi, a.rxi + 2 * a.fuseK, 1))
        return;
    const int tFirst = a.tileFirst[tile];
    if (tFirst == INT_MAX || tFirst >= a.tB) return;
    int onset = a.sOnset[s];
    int sourceDirEnd = onset >= 0 ? onset + a.nDir : INT_MAX;
    int directEnd = onset >= 0 ? onset + a.nDry : INT_MAX;
    if (a.tA >= directEnd) return;  // this cell's dry window is closed
    // a wall cell's pressure is identically zero: it never has an onset and must not keep its tile recording
    if (a.coef[(size_t)(X + a.G) * a.pitch + (Y + a.G)].beta == 0.f) return;

    const int prow = X + a.G, pcol = Y + a.G;
    const int hr = prow - dyn.histRow0, hcol = pcol - dyn.histCol0;
    const long long hoff = histOffset(hr, hcol, a.rxi, a.wi, dyn.histTilesY);
    int tFx = INT_MAX, tFy = INT_MAX;
    if (X > 0 && prow - 1 >= dyn.histRow0) tFx = a.tileFirst[((X - 1) / a.rxi) * a.nty + (Y / a.wi)];
    if (Y > 0 && pcol - 1 >= dyn.histCol0) tFy = a.tileFirst[(X / a.rxi) * a.nty + ((Y - 1) / a.wi)];
    const float* hc = a.hist + hoff;
    const float* hx = a.hist + (tFx != INT_MAX ? histOffset(hr - 1, hcol, a.rxi, a.wi, dyn.histTilesY) : hoff);
    const float* hy = a.hist + (tFy != INT_MAX ? histOffset(hr, hcol - 1, a.rxi, a.wi, dyn.histTilesY) : hoff);
    const FaceCoef fc = a.coef[(size_t)prow * a.pitch + pcol];
    const float kx = fc.kx, ky = fc.ky;
    const bool airX = kx != kx, airY = ky != ky;
    const float C = a.courant;

    float Edry = a.sEdry[s], fluxX = a.sFx[s], fluxY = a.sFy[s], vx = a.sVx[s], vy = a.sVy[s];
    constexpr int CH = 8;
    bool done = false;
    const int tEnd = min(a.tB, a.T);
    for (int t0 = max(a.tA, tFirst); t0 < tEnd && !done; t0 += CH) {
        float pc[CH], pxc[CH], pyc[CH];
        const bool needVChunk = t0 < sourceDirEnd;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int tt = min(t0 + k, tEnd - 1);
            const long long o = (long long)(tt % a.ring) * a.histPlane;
            pc[k] = hc[o];
            pxc[k] = (needVChunk && tt >= tFx) ? hx[o] : 0.f;
            pyc[k] = (needVChunk && tt >= tFy) ? hy[o] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int t = t0 + k;
            if (done || t >= tEnd || t >= directEnd) {
                done = done || t >= directEnd;
                continue;
            }
            const float p = pc[k];
            if (t < sourceDirEnd) {
                const float pxn = pxc[k], pyn = pyc[k];
                const float ax = vx - C * (p - pxn), wx = kx * (p + pxn);
                const float ay = vy - C * (p - pyn), wy = ky * (p + pyn);
                vx = airX ? ax : wx;
                vy = airY ? ay : wy;
            }
            if (onset < 0 && fabsf(p) > kAudibleThresholdDev) {
                onset = t;
                sourceDirEnd = t + a.nDir;
                directEnd = t + a.nDry;
                if (t >= directEnd) {
                    done = true;
                    continue;
                }
            }
            Edry += p * p;
            if (t < sourceDirEnd) {
                fluxX += p * vx;
                fluxY += p * vy;
            }
        }
    }
    // still open after this pass?  (no onset yet, or the dry window reaches past tB) -> keep the tile recording
    if (onset < 0 || a.tB < directEnd) a.tileOpenOut[tile] = 1;
    a.sOnset[s] = onset;
    a.sEdry[s] = Edry;
    a.sFx[s] = fluxX;
    a.sFy[s] = fluxY;
    a.sVx[s] = vx;
    a.sVy[s] = vy;
}

// pressure of the registered emitter cells for steps [tA, tB): ring -> per-emitter trace
__global__ void pv_stream_trace_kernel(const AnalyzeArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = a.tB - a.tA;
    if (i >= a.numEmitters * n) return;
    const int e = i / n, t = a.tA + (i - e * n);
    if (t >= a.T) return;
    const DynParams dyn = *a.dyn;
    const int cell = a.emCells[e];
    const int X = cell / a.gy, Y = cell - X * a.gy;
    const int tFirst = a.tileFirst[(X / a.rxi) * a.nty + (Y / a.wi)];
    float v = 0.f;
    if (t >= tFirst)
        v = a.hist[(long long)(t % a.ring) * a.histPlane +
                   histOffset(X + a.G - dyn.histRow0, Y + a.G - dyn.histCol0, a.rxi, a.wi, dyn.histTilesY)];
    a.emTrace[(size_t)e * a.T + t] = v;
}

// end of run: onset map + the outputs that come from the forward sums (Analyzer.cpp:197-230), every cell
__global__ __launch_bounds__(256) void pv_stream_finalize_kernel(const AnalyzeArgs a) {
    const int Y = blockIdx.x * blockDim.x + threadIdx.x;
    const int X = blockIdx.y;
    if (Y >= a.gy || X >= a.gx) return;
    const int s = X * a.gy + Y;
    const int onset = a.sOnset[s];
    if (onset < 0) {
        a.delay[s] = FLT_MAX;
        return;
    }
    a.delay[s] = (float)onset;
    const float Edry = a.sEdry[s], fluxX = a.sFx[s], fluxY = a.sFy[s];
    const float EfreePr = efreePerR(a.efree, a.dx, a.lcx, a.lcy, X, Y);
    const float occ = sqrtf(Edry / EfreePr);
    float norm = sqrtf(fluxX * fluxX + fluxY * fluxY);
    norm = -1.0f / (norm > 0.0f ? norm : 1.0f);
    const float rr = 1.0f / ((0.001f < occ) ? occ : 0.001f);
    a.out[s] = occ;
    a.out[3 * a.resN + s] = -147.f + (18390.f) / (1.f + pvPowf(rr / 12.f, 0.8f));
    a.out[6 * a.resN + s] = norm * fluxX;
    a.out[7 * a.resN + s] = norm * fluxY;
}

// wet gain + RT60 of the registered emitter cells from their traces (Analyzer.cpp:235-327); one thread each
__global__ void pv_stream_emitter_kernel(const AnalyzeArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.numEmitters) return;
    const int s = a.emCells[e];
    const int onset = a.sOnset[s];
    if (onset < 0) return;
    const float* tr = a.emTrace + (size_t)e * a.T;
    const int T = a.T;
    const int directEnd = onset + a.nDry;
    float wetEnergy = 0.f;
    {
        int end = directEnd + 1 + a.nWet;
        if (T < end) end = T;
        for (int j = directEnd + 1; j < end; ++j) wetEnergy += tr[j] * tr[j];
    }
    const int startingPoint = directEnd + 1;
    const int endPoint = T - a.nCut;
    const float rn = (float)(endPoint - startingPoint);
    const float xmean = (rn - 1.0f) * 0.5f;
    const float xsum = rn * xmean;
    const float denominator = (1.0f / 12.0f) * rn * (rn * rn - 1.0f);
    float edc = 0.f, xysum = 0.f, ysum = 0.f;
    for (int i = T - 1; i >= endPoint && i >= 0; --i) edc += tr[i] * tr[i];
    constexpr int CH = 8;
    for (int i0 = endPoint - 1; i0 >= startingPoint; i0 -= CH) {
        float pc[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) pc[k] = tr[max(i0 - k, 0)];
        rt60Chunk<CH>(pc, i0, startingPoint, edc, xysum, ysum);
    }
    const float ymean = ysum / rn;
    const float numerator = xysum - ymean * xsum - xmean * ysum + rn * xmean * ymean;
    const float slopePerSec = (numerator / denominator) * (float)a.fs;
    a.out[a.resN + s] = sqrtf(wetEnergy / a.efree);
    a.out[2 * a.resN + s] = -60.f / slopePerSec;
}

// tileOpen[tile] for the next launches: some cell marked it open in this pass, or the wave has not reached it yet,
// or it holds a registered emitter (its whole trace is needed)
__global__ void pv_stream_tilegate_kernel(const uint8_t* marks, const uint8_t* hasEmitter, const int* tileFirst, int tB,
                                          uint8_t* tileOpen, int ntiles) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ntiles) tileOpen[t] = (marks[t] || hasEmitter[t] || tileFirst[t] >= tB) ? 1 : 0;
}

void launchStreamAccum(const AnalyzeArgs& a, const uint8_t* hasEmitter, uint8_t* tileOpen, int ntiles,
                       hipStream_t stream) {
    dim3 grid((a.gy + 255) / 256, a.gx);
    if (a.ringList) grid = dim3((unsigned)((a.rxi * a.wi + 255) / 256) * (unsigned)(a.numRing > 0 ? a.numRing : 1), 1);
    hipMemsetAsync(a.tileOpenOut, 0, (size_t)ntiles, stream);
    if (!a.ringList || a.numRing > 0) hipLaunchKernelGGL(pv_stream_accum_kernel, grid, dim3(256), 0, stream, a);
    hipLaunchKernelGGL(pv_stream_tilegate_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, stream, a.tileOpenOut,
                       hasEmitter, a.tileFirst, a.tB, tileOpen, ntiles);
    const int n = a.numEmitters * (a.tB - a.tA);
    if (n > 0) hipLaunchKernelGGL(pv_stream_trace_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a);
}

void launchStreamFinalize(const AnalyzeArgs& a, hipStream_t stream) {
    dim3 grid((a.gy + 255) / 256, a.gx);
    const int n = a.gx * a.gy;
    hipLaunchKernelGGL(pv_fill_delay_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a.delay, n);
    hipLaunchKernelGGL(pv_stream_finalize_kernel, grid, dim3(256), 0, stream, a);
    if (a.numEmitters > 0)
        hipLaunchKernelGGL(pv_stream_emitter_kernel, dim3((a.numEmitters + 63) / 64), dim3(64), 0, stream, a);
    // T is large in this mode: resolve the delay-map descent by pointer jumping (the window is the whole grid here)
    launchDirectionJump(a, a.dirScratch, stream);
}

// FreeGrid::CalculateEFree + SimulateFreeFieldEnergy tail, FreeGrid.cpp:86-110: sequential float sum of p^2 over
// the first n samples at one cell, times the discrete distance r.
__global__ void pv_efree_kernel(const float* hist, long long plane, long long cellOff, int n, float r,
                                float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float e = 0.f;
    for (int i = 0; i < n; ++i) {
        const float p = hist[(long long)i * plane + cellOff];
        e += p * p;
    }
    out[0] = e * r;
}

void launchEfree(const float* hist, long long plane, long long cellOff, int n, float r, float* out,
                 hipStream_t stream) {
    hipLaunchKernelGGL(pv_efree_kernel, dim3(1), dim3(64), 0, stream, hist, plane, cellOff, n, r, out);
}

// Grid::GetResponse, FDTD.cpp:74-79: the (pr, vx, vy) impulse response of one array cell, rebuilt from the
// pressure history (see pv_encode_kernel).  out = T x 3 floats.
__global__ void pv_ir_kernel(const AnalyzeArgs a, int X, int Y, float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const DynParams dyn = *a.dyn;
    const int prow = X + a.G, pcol = Y + a.G;
    const int tj = (Y / a.wi), ti = (X / a.rxi);
    const int wti = ti - dyn.histTileX0, wtj = tj - dyn.histTileY0;
    const bool inWin = wti >= 0 && wti < dyn.histTilesX && wtj >= 0 && wtj < dyn.histTilesY;
    const int tFirst = inWin ? a.tileFirst[ti * a.nty + tj] : INT_MAX;
    const long long hoff = inWin ? histOffset(prow - dyn.histRow0, pcol - dyn.histCol0, a.rxi, a.wi, dyn.histTilesY) : 0;
    int tFx = INT_MAX, tFy = INT_MAX;
    if (X > 0 && prow - 1 >= dyn.histRow0) tFx = a.tileFirst[((X - 1) / a.rxi) * a.nty + tj];
    if (Y > 0 && pcol - 1 >= dyn.histCol0) tFy = a.tileFirst[ti * a.nty + ((Y - 1) / a.wi)];
    const long long hoffX = tFx != INT_MAX ? histOffset(prow - 1 - dyn.histRow0, pcol - dyn.histCol0, a.rxi, a.wi, dyn.histTilesY) : 0;
    const long long hoffY = tFy != INT_MAX ? histOffset(prow - dyn.histRow0, pcol - 1 - dyn.histCol0, a.rxi, a.wi, dyn.histTilesY) : 0;
    const FaceCoef fc = a.coef[(size_t)prow * a.pitch + pcol];
    const float kx = fc.kx, ky = fc.ky;
    const bool airX = kx != kx, airY = ky != ky;
    const bool above = X == 0 && a.histAbove;  // first row of a slab: the row above lives in the neighbouring slab
    float vx = 0.f, vy = 0.f;
    for (int t = 0; t < a.T; ++t) {
        float p = 0.f;
        if (t >= tFirst) {
            p = a.hist[(long long)t * a.histPlane + hoff];
            const float pxn = above ? a.histAbove[(long long)t * a.histPitch + (pcol - dyn.histCol0)]
                                    : (t >= tFx) ? a.hist[(long long)t * a.histPlane + hoffX] : 0.f;
            const float pyn = (t >= tFy) ? a.hist[(long long)t * a.histPlane + hoffY] : 0.f;
            const float ax = vx - a.courant * (p - pxn), wx = kx * (p + pxn);
            const float ay = vy - a.courant * (p - pyn), wy = ky * (p + pyn);
            vx = airX ? ax : wx;
            vy = airY ? ay : wy;
        }
        out[3 * t + 0] = p;
        out[3 * t + 1] = vx;
        out[3 * t + 2] = vy;
    }
}

void launchIr(const AnalyzeArgs& a, int X, int Y, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(pv_ir_kernel, dim3(1), dim3(64), 0, stream, a, X, Y, out);
}

// gather / scatter between the reference's dense (gx+1)x(gy+1) order and the padded device planes
__global__ void pv_unpad_kernel(const float* padded, float* dense, Geometry g) {
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    const int x = blockIdx.y;
    if (y < g.NY && x < g.NX) dense[(size_t)x * g.NY + y] = padded[(size_t)(x + g.G) * g.pitch + (y + g.G)];
}
__global__ void pv_pad_kernel(const float* dense, float* padded, Geometry g) {
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    const int x = blockIdx.y;
    if (y < g.NY && x < g.NX) padded[(size_t)(x + g.G) * g.pitch + (y + g.G)] = dense[(size_t)x * g.NY + y];
}
// recorded pressure plane t -> dense order, zero where nothing was stored
__global__ void pv_histplane_kernel(const AnalyzeArgs a, int t, float* dense, int NX, int NY, int histRows) {
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    const int x = blockIdx.y;
    if (y >= NY || x >= NX) return;
    const DynParams dyn = *a.dyn;
    const int hr = x + a.G - dyn.histRow0, hcn = y + a.G - dyn.histCol0;
    float v = 0.f;
    if (hr >= 0 && hr < histRows && hcn >= 0 && hcn < dyn.histTilesY * a.wi) {
        const int tF = a.tileFirst[(x / a.rxi) * a.nty + (y / a.wi)];
        if (t >= tF) v = a.hist[(long long)t * a.histPlane + histOffset(hr, hcn, a.rxi, a.wi, dyn.histTilesY)];
    }
    dense[(size_t)x * NY + y] = v;
}

void launchUnpad(const float* padded, float* dense, const Geometry& g, hipStream_t stream) {
    dim3 grid((g.NY + 255) / 256, g.NX);
    hipLaunchKernelGGL(pv_unpad_kernel, grid, dim3(256), 0, stream, padded, dense, g);
}
void launchPad(const float* dense, float* padded, const Geometry& g, hipStream_t stream) {
    dim3 grid((g.NY + 255) / 256, g.NX);
    hipLaunchKernelGGL(pv_pad_kernel, grid, dim3(256), 0, stream, dense, padded, g);
}
void launchHistPlane(const AnalyzeArgs& a, int t, float* dense, int NX, int NY, int histRows,
                     hipStream_t stream) {
    dim3 grid((NY + 255) / 256, NX);
    hipLaunchKernelGGL(pv_histplane_kernel, grid, dim3(256), 0, stream, a, t, dense, NX, NY, histRows);
}

}  // namespace pva
