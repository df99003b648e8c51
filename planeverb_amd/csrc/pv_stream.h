// pv_stream.h -- sparse-emitter mode: forward sums of the analysis INSIDE the stencil (included by
// pv_kernels.hip).
//
// Why: with T in the tens of thousands the mode keeps a ring of pressure planes and
// a separate pass (pv_stream_accum_kernel) walks it for every cell whose window is open -- onset, dry energy and the flux of
// the source direction (Analyzer.cpp:146-247).  While the wave front crosses the grid that is 0.8 GB of ring writes plus
// 0.8 GB of ring reads per 12-step launch beside 0.5 GB of fields: the run is HBM-bound at 0.75 of its own stencil.  But the
// step kernel HOLDS what those sums need -- p, vx, vy of every interior cell after every step -- so for AIR tiles (every face
// air|air, no listener, no registered emitter) the sums move into the stencil:
//
//   * pv_stream_classify_kernel (one tiny launch per K steps) splits the air tiles: a tile whose cells can still change their
//     sums -- some window open (cellsOpen), and something non-zero within reach (its own or a neighbour's non-zero flag of the
//     previous launch: a superset of "its loaded region is non-zero now") -- goes on the OPEN list and gets class 1 in the
//     launch's copy of the tile classes, so that the merged step kernel leaves it alone without a line of it changing;
//     everything else stays with the merged kernel.  It also writes the plane the step kernels read as `tileOpen`: 1 only for
//     RING tiles (general tiles, emitter tiles) that are open, so that an air tile records pressure history only while a ring
//     tile below / right of it still reads its last row / column (historyWanted).
//   * pv_step_open_kernel: one wave per HALF of an open tile (RXI/2 interior rows + the K-row halos: 42 rows of 3 fields =
//     126 registers, which leaves room for 4 state values of each of its 18 x 1 interior cells per lane).  Mirror-pair stencil
//     as in the air tile; after every step the interior cells' sums advance exactly as pv_stream_accum_kernel advances them
//     from the ring (same operations in the same order: Edry += p * p; flux += p * v; first |p| above the threshold = onset),
//     with vx, vy taken from the registers instead of being re-derived.  State lives in the planes the finalize pass reads
//     (sOnset, sEdry, sFx, sFy): 16 B read + 16 B written per open cell and launch instead of 48 B + 48 B of ring traffic.
//   * ring tiles keep the ring and the accumulate pass (which skips the cells of fused tiles).
#pragma once

namespace pva {

// is `tile` one whose forward sums live in the stencil?  (air class, no registered emitter, not the listener's tile -- that
// one is on the general list for the run)
__device__ __forceinline__ bool fusedTile(const uint8_t* tileClass, const uint8_t* tileEmit, const DynParams& dyn, int ti,
                                          int tj, int nty, int G, int K, int rxi, int wi, int rows, int withPulse) {
    const int t = ti * nty + tj;
    if (tileClass[t] != 0 || tileEmit[t]) return false;
    if (withPulse) {
        const int lr = dyn.lrow - (G - K + ti * rxi), lc = dyn.lcol - (G - K + tj * wi);
        if (lr >= 0 && lr < rows && lc >= 0 && lc < 64) return false;
    }
    return true;
}

__global__ __launch_bounds__(256) void pv_stream_classify_kernel(const ClassifyArgs c) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= c.ntx * c.nty) return;
    const DynParams dyn = *c.dyn;
    const int ti = t / c.nty, tj = t - ti * c.nty;
    const bool fused = fusedTile(c.tileClass, c.tileEmit, dyn, ti, tj, c.nty, c.G, c.K, c.rxi, c.wi, c.rows, c.withPulse);
    uint8_t cls = c.tileClass[t];
    c.ringOpenOut[t] = (!fused && c.tileOpenRing[t]) ? 1 : 0;
    if (fused && (c.cellsOpen2[2 * t] || c.cellsOpen2[2 * t + 1])) {
        bool near = false;
        for (int di = -1; di <= 1; ++di)
            for (int dj = -1; dj <= 1; ++dj) {
                const int i = ti + di, j = tj + dj;
                if (i >= 0 && i < c.ntx && j >= 0 && j < c.nty && c.nzPrev[i * c.nty + j]) near = true;
            }
        if (near) {
            cls = 1;
            c.nzNext[t] = 0;
            const int at = atomicAdd(c.openCount, 2);
            c.openList[at] = 2 * t;
            c.openList[at + 1] = 2 * t + 1;
        }
    }
    c.classOut[t] = cls;
}

// Has every fused tile closed for good?  (a half's flag only ever goes 1 -> 0: once all of its cells have had their onset and
// their dry windows are over; a tile the sound never reaches keeps it at 1.)  One block; the answer goes to pinned host
// memory, where Solver::enqueueRun reads it two passes later and stops launching the classify pass and the open-tile kernel.
__global__ __launch_bounds__(256) void pv_stream_idle_kernel(const ClassifyArgs c, int* idleHost) {
    const DynParams dyn = *c.dyn;
    int busy = 0;
    for (int t = threadIdx.x; t < c.ntx * c.nty; t += 256) {
        const int ti = t / c.nty, tj = t - ti * c.nty;
        if ((c.cellsOpen2[2 * t] || c.cellsOpen2[2 * t + 1]) &&
            fusedTile(c.tileClass, c.tileEmit, dyn, ti, tj, c.nty, c.G, c.K, c.rxi, c.wi, c.rows, c.withPulse))
            busy = 1;
    }
    busy = __syncthreads_or(busy);
    if (threadIdx.x == 0) *idleHost = busy ? 0 : 1;
}

template <int K, int RXI>
struct OpenGeom {
    static constexpr int HX = RXI / 2;        // interior rows of a half tile
    static constexpr int ROWS = HX + 2 * K;
    static constexpr int NP = ROWS / 2;
    static constexpr int NPI = HX / 2;        // interior row PAIRS: row K + i (top half) with row ROWS - 1 - K - i (bottom)
    static constexpr int WI = 64 - 2 * K;
    static_assert(RXI % 4 == 0, "a half tile is a whole number of mirror pairs");
};

// the per-cell forward sums of one interior row pair after step t (pv_stream_accum_kernel's loop body, Analyzer.cpp:146-247):
// oe = onset step (INT_MAX = none yet); the dry window is open while t - oe < nDry, the direction window while t - oe < nDir
__device__ __forceinline__ void openAccumulate(const v2f p, const v2f vx, const v2f vy, const int t, const int nDir,
                                               const int nDry, int (&oe)[2], v2f& E, v2f& fx, v2f& fy) {
    const float pc[2] = {p.x, p.y}, xc[2] = {vx.x, vx.y}, yc[2] = {vy.x, vy.y};
    float e[2] = {E.x, E.y}, ax[2] = {fx.x, fx.y}, ay[2] = {fy.x, fy.y};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const bool open = (t - oe[k]) < nDry;  // (oe = INT_MAX: hugely negative -> open)
        if (open && oe[k] == INT_MAX && fabsf(pc[k]) > kAudibleThresholdDev) oe[k] = t;
        const bool dir = open && (t - oe[k]) < nDir;
        const float pp = pc[k] * pc[k], px = pc[k] * xc[k], py = pc[k] * yc[k];
        e[k] = e[k] + (open ? pp : 0.f);
        ax[k] = ax[k] + (dir ? px : 0.f);
        ay[k] = ay[k] + (dir ? py : 0.f);
    }
    E = v2f{e[0], e[1]};
    fx = v2f{ax[0], ax[1]};
    fy = v2f{ay[0], ay[1]};
}

template <int K, int RXI, int S>
struct OpenSteps {
    using Gm = OpenGeom<K, RXI>;
    static constexpr int ROWS = Gm::ROWS, NP = Gm::NP, NPI = Gm::NPI;
    static __device__ __forceinline__ void run(v2f (&pr)[NP], v2f (&vx)[NP], v2f (&vy)[NP], float& vxS, const float C,
                                               const int nsteps, const int t0, const int nDir, const int nDry,
                                               int (&oe)[NPI][2], v2f (&E)[NPI], v2f (&fx)[NPI], v2f (&fy)[NPI],
                                               const bool recLane, const float* hplane0, const long long hstride,
                                               const int hvoff, const int hsoff0, const int hpitchB) {
        if constexpr (S < K) {
            if (S < nsteps) {
                leapfrogStepMirror<NP, PV_MIRROR_G, S>(pr, vx, vy, vxS, C);
                if (recLane) {  // (only while a ring tile below / right of this one reads its last row / column)
                    const rsrc_t rH = makeRsrc(hplane0 + (long long)S * hstride, hstride * 4);
#pragma unroll
                    for (int r = K; r < ROWS - K; ++r)
                        bufStoreF(r < NP ? pr[r].x : pr[ROWS - 1 - r].y, rH, hvoff, hsoff0 + r * hpitchB);
                }
#pragma unroll
                for (int i = 0; i < NPI; ++i) {
                    // row K + i and its mirror ROWS - 1 - K - i; the mirrored half stores vx negated and shifted by one face
                    const v2f vxp = v2f{vx[K + i].x, (i == NPI - 1) ? vxS : -vx[K + i + 1].y};
                    openAccumulate(pr[K + i], vxp, vy[K + i], t0 + S, nDir, nDry, oe[i], E[i], fx[i], fy[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
                OpenSteps<K, RXI, S + 1>::run(pr, vx, vy, vxS, C, nsteps, t0, nDir, nDry, oe, E, fx, fy, recLane, hplane0,
                                              hstride, hvoff, hsoff0, hpitchB);
            }
        }
    }
};

template <int K, int RXI>
__global__ __launch_bounds__(256, 2) void pv_step_open_kernel(const StepArgs a, const OpenArgs o) {
    using Gm = OpenGeom<K, RXI>;
    constexpr int ROWS = Gm::ROWS, NP = Gm::NP, NPI = Gm::NPI, HX = Gm::HX, WI = Gm::WI;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int idx = blockIdx.x * 4 + wave;
    if (blockIdx.x == 0 && threadIdx.x == 0) *o.nextCount = 0;  // the classify pass of the NEXT launch counts from zero
    if (idx >= *o.openCount) return;
    const int entry = __builtin_amdgcn_readfirstlane(o.openList[idx]);
    const int tile = entry >> 1, half = entry & 1;
    const int ti = tile / a.nty, tj = tile - ti * a.nty;
    const int row0 = a.G - K + ti * RXI + half * HX;
    const int col0 = a.G - K + tj * WI;
    const int voff = lane * 4;
    const int pitchB = a.pitch * 4;
    const int soff0 = (row0 * a.pitch + col0) * 4;

    const rsrc_t rPrIn = makeRsrc(a.prIn, a.inBytes), rVxIn = makeRsrc(a.vxIn, a.inBytes),
                 rVyIn = makeRsrc(a.vyIn, a.inBytes);
    v2f pr[NP], vx[NP], vy[NP];
    float vxS = bufLoadF(rVxIn, voff, soff0 + NP * pitchB);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int soT = soff0 + i * pitchB, soB = soff0 + (ROWS - 1 - i) * pitchB;
        pr[i].x = bufLoadF(rPrIn, voff, soT);
        pr[i].y = bufLoadF(rPrIn, voff, soB);
        vy[i].x = bufLoadF(rVyIn, voff, soT);
        vy[i].y = bufLoadF(rVyIn, voff, soB);
        vx[i].x = bufLoadF(rVxIn, voff, soT);
        vx[i].y = (i > 0) ? -bufLoadF(rVxIn, voff, soB + pitchB) : 0.f;  // face ROWS-i; face ROWS is not in the half tile
    }
    // the interior cells' sums: result cell (X, Y) = (ti * RXI + half * HX + (r - K), tj * WI + lane - K)
    const bool inCols = lane >= K && lane < 64 - K;
    const long long resN = (long long)o.gxRes * o.gyRes;
    const rsrc_t rOn = makeRsrc(o.sOnset, resN * 4), rE = makeRsrc(o.sEdry, resN * 4), rFx = makeRsrc(o.sFx, resN * 4),
                 rFy = makeRsrc(o.sFy, resN * 4);
    const int X0 = ti * RXI + half * HX, Y = tj * WI + lane - K;
    const int svoff = Y * 4;  // (used by the interior lanes only: the halo lanes hold no result cell of this tile)
    int oe[NPI][2];
    v2f E[NPI], fx[NPI], fy[NPI];
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        oe[i][0] = oe[i][1] = INT_MAX;
        E[i] = fx[i] = fy[i] = v2f{0.f, 0.f};
    }
    if (inCols) {
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int soT = (X0 + i) * o.gyRes * 4, soB = (X0 + HX - 1 - i) * o.gyRes * 4;
            const int onT = (int)__builtin_amdgcn_raw_buffer_load_b32(rOn, svoff, soT, 0);
            const int onB = (int)__builtin_amdgcn_raw_buffer_load_b32(rOn, svoff, soB, 0);
            oe[i][0] = onT >= 0 ? onT : INT_MAX;
            oe[i][1] = onB >= 0 ? onB : INT_MAX;
            E[i] = v2f{bufLoadF(rE, svoff, soT), bufLoadF(rE, svoff, soB)};
            fx[i] = v2f{bufLoadF(rFx, svoff, soT), bufLoadF(rFx, svoff, soB)};
            fy[i] = v2f{bufLoadF(rFy, svoff, soT), bufLoadF(rFy, svoff, soB)};
        }
    }

    uint32_t nz = __float_as_uint(vxS);
#pragma unroll
    for (int i = 0; i < NP; ++i)
        nz |= __float_as_uint(pr[i].x) | __float_as_uint(pr[i].y) | __float_as_uint(vx[i].x) |
              __float_as_uint(vx[i].y) | __float_as_uint(vy[i].x) | __float_as_uint(vy[i].y);
    const bool active = __ballot((nz & 0x7fffffffu) != 0u) != 0ull;
    // (the classify pass cleared this launch's flag of the tile: its two halves only ever set it)
    if (active && lane == 0) a.nzOut[tile] = 1;

    const DynParams dyn = *a.dyn;
    const int hti = ti - dyn.histTileX0, htj = tj - dyn.histTileY0;
    const bool inWin = hti >= 0 && hti < dyn.histTilesX && htj >= 0 && htj < dyn.histTilesY;
    // Both halves must take the same recording decision, whatever each of them holds: a tile on the open list counts as
    // reached from this launch on (recording zeros earlier than needed is harmless: the readers treat what lies before
    // tileFirst as the zeros it is).
    if (a.record && lane == 0) atomicMin(&a.tileFirst[tile], a.t0);
    const bool rec = a.record && inWin && historyWanted(a, ti, tj);
    if (a.record && active && !inWin && lane == 0) atomicExch(a.errFlag, 1);

    const float C = a.courant;
    const float* hplane = a.hist + (long long)a.histSlot * a.histPlane;
    const int hpitchB = WI * 4;
    const int hsoff0 = ((hti * dyn.histTilesY + htj) * RXI + half * HX - K) * hpitchB;
    const int hvoff = (lane - K) * 4;

    OpenSteps<K, RXI, 0>::run(pr, vx, vy, vxS, C, a.nsteps, a.t0, o.nDir, o.nDry, oe, E, fx, fy, rec && inCols, hplane,
                              a.histPlane, hvoff, hsoff0, hpitchB);

    // any interior cell whose dry window is still open after this launch?
    const int tLast = a.t0 + a.nsteps - 1;
    bool open = false;
#pragma unroll
    for (int i = 0; i < NPI; ++i) open = open || (tLast + 1 - oe[i][0]) < o.nDry || (tLast + 1 - oe[i][1]) < o.nDry;
    const bool anyOpen = __ballot(open && inCols) != 0ull;
    if (lane == 0) o.cellsOpen2[entry] = anyOpen ? 1 : 0;

    const rsrc_t rPrOut = makeRsrc(a.prOut, a.planeBytes), rVxOut = makeRsrc(a.vxOut, a.planeBytes),
                 rVyOut = makeRsrc(a.vyOut, a.planeBytes);
    if (inCols) {
#pragma unroll
        for (int r = K; r < ROWS - K; ++r) {
            const int so = soff0 + r * pitchB;
            bufStoreF(r < NP ? pr[r].x : pr[ROWS - 1 - r].y, rPrOut, voff, so);
            bufStoreF(r < NP ? vx[r].x : (r == NP ? vxS : -vx[ROWS - r].y), rVxOut, voff, so);
            bufStoreF(r < NP ? vy[r].x : vy[ROWS - 1 - r].y, rVyOut, voff, so);
        }
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int soT = (X0 + i) * o.gyRes * 4, soB = (X0 + HX - 1 - i) * o.gyRes * 4;
            __builtin_amdgcn_raw_buffer_store_b32((uint32_t)(oe[i][0] == INT_MAX ? -1 : oe[i][0]), rOn, svoff, soT, 0);
            __builtin_amdgcn_raw_buffer_store_b32((uint32_t)(oe[i][1] == INT_MAX ? -1 : oe[i][1]), rOn, svoff, soB, 0);
            bufStoreF(E[i].x, rE, svoff, soT);
            bufStoreF(E[i].y, rE, svoff, soB);
            bufStoreF(fx[i].x, rFx, svoff, soT);
            bufStoreF(fx[i].y, rFx, svoff, soB);
            bufStoreF(fy[i].x, rFy, svoff, soT);
            bufStoreF(fy[i].y, rFy, svoff, soB);
        }
    }
}

}  // namespace pva
