// pv_analysis.h -- device helpers shared by the analysis kernels of pv_kernels.hip and pv_rt60.hip (moved here unchanged)
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>

#include "pv_device.h"
#include "pv_prims.h"

namespace pva {

// (slab groups: see AnalyzeArgs::abortWord)
__device__ __forceinline__ bool analysisAborted(const AnalyzeArgs& a) { return a.abortWord && *a.abortWord != 0u; }

// The near box (AnalyzeArgs::box / FarInfo::box): the reached cells' bounding box grown by one cell, inside the window block
// [wr0, wr0 + wnr) x [wc0, wc0 + wnc).  Inclusive bounds; empty when r1 < r0 or c1 < c0.
struct NearBox {
    int r0, c0, r1, c1;
    __device__ __forceinline__ bool holds(int r, int c) const { return r >= r0 && r <= r1 && c >= c0 && c <= c1; }
};
__device__ __forceinline__ NearBox nearBoxOf(const int* box, int wr0, int wc0, int wnr, int wnc) {
    NearBox b;
    b.r0 = max(box[0] - 1, wr0);  // (an empty box holds INT_MAX / -1: no overflow either way)
    b.c0 = max(box[1] - 1, wc0);
    b.r1 = min(box[2] + 1, wr0 + wnr - 1);
    b.c1 = min(box[3] + 1, wc0 + wnc - 1);
    return b;
}
__device__ __forceinline__ NearBox nearBoxOf(const AnalyzeArgs& a, const DynParams& dyn) {
    const int wr0 = dyn.histRow0 - a.G, wc0 = dyn.histCol0 - a.G;
    return nearBoxOf(a.box, wr0, wc0, min(a.winRows, a.gx - wr0), min(a.winCols, a.gy - wc0));
}

// closed-form listener direction of a far cell: what storeDirection(a, index, index) writes (a walk that stays put)
__device__ __forceinline__ void farDirectionOf(const FarInfo& f, long long cell, float* ox, float* oy) {
    const int r = (int)(cell / f.gy), c = (int)(cell - (long long)r * f.gy);
    float x = (float)r * f.dx - f.lx, y = (float)c * f.dx - f.lz;
    float len = (x * x) + (y * y);
    if (len != 0.f) {
        len = sqrtf(len);
        x /= len;
        y /= len;
    }
    *ox = x;
    *oy = y;
}
// is the direction of this cell NOT in the result planes?
__device__ __forceinline__ bool isFarCell(const FarInfo& f, long long cell) {
    if (!f.on) return false;
    const int r = (int)(cell / f.gy), c = (int)(cell - (long long)r * f.gy);
    if (f.box) return !nearBoxOf(f.box, f.r0, f.c0, f.nr, f.nc).holds(r, c);
    return r < f.r0 || r >= f.r0 + f.nr || c < f.c0 || c >= f.c0 + f.nc;
}

struct CellHistory {
    const float* h;     // this cell, step 0
    long long plane;
    __device__ __forceinline__ float at(int t) const { return h[(long long)t * plane]; }
};

// The history planes are tile-major without padding (pv_prims.h histOffset), so an OFFSET g inside a plane names a window cell:
// tile g / (rxi wi), row and column inside it.  Analysis kernels whose lanes run along g read 256 contiguous bytes of one
// plane per load instruction and wave, and every line of the history is fetched by exactly one workgroup (lanes along WINDOW
// columns read 160-192-byte pieces of tile rows that are not line-aligned: 2.7 x the bytes, profiles/r04_analysis_pmc.md).
struct PlaneCell {
    bool inGrid;   // a result cell (X < gx, Y < gy)
    int X, Y;      // array cell (row of a slab: local)
    int tile;      // grid tile index ti * nty + tj
    int row, col;  // inside the tile
    int g;         // the offset itself
    int hti, htj;  // the tile's place in the window
};
__device__ __forceinline__ PlaneCell planeCell(const AnalyzeArgs& a, const DynParams& dyn, long long g) {
    PlaneCell c{false, 0, 0, 0, 0, 0, 0, 0, 0};
    if (g >= a.histPlane) return c;
    const int gi = (int)g, tileCells = a.rxi * a.wi;
    const int wt = gi / tileCells, f = gi - wt * tileCells;
    c.row = f / a.wi;
    c.col = f - c.row * a.wi;
    const int hti = wt / dyn.histTilesY, htj = wt - hti * dyn.histTilesY;
    c.g = gi;
    c.hti = hti;
    c.htj = htj;
    const int ti = dyn.histTileX0 + hti, tj = dyn.histTileY0 + htj;
    c.tile = ti * a.nty + tj;
    c.X = ti * a.rxi + c.row;
    c.Y = tj * a.wi + c.col;
    c.inGrid = c.X < a.gx && c.Y < a.gy;
    return c;
}

// FreeGrid::GetEFreePerR, FreeGrid.cpp:41-59
__device__ __forceinline__ float efreePerR(float efree, float dx, int lX, int lY, int eX, int eY) {
    const float lx = (float)lX * dx, ly = (float)lY * dx;
    const float ex = (float)eX * dx, ey = (float)eY * dx;
    const float r = sqrtf((ex - lx) * (ex - lx) + (ey - ly) * (ey - ly));
    if (r == 0.f) return efree;
    return efree / r;
}

// ---------------------------------------------------------------------------------------------------------------
// RT60: backward Schroeder integration + linear regression (Analyzer.cpp:282-327): forms with the same bits
// ---------------------------------------------------------------------------------------------------------------
// All forms keep the three running sums (energy decay, sum y x, sum y) strictly sequential in the reference's order; they
// differ in how many LANES share a cell (pv_rt60.hip).  Which one runs is decided on the device from the number of cells
// of the window's ever-non-zero tiles (an upper bound of the reached cells, counted by block 0 of the far-cells pass):
// few cells -> sixteen lanes per cell (parallelism), more -> four (fewer instructions per sample).
__device__ __forceinline__ int rt60LanesPerCell(const AnalyzeArgs& a, int activeCells) {
    if (a.rt60Lanes) return a.rt60Lanes;  // (PVA_OPT_RT60_LANES: validation / measurement)
    // (measured on MI355X, profiles/r04_rt60.txt: 70^2 0.093 / 0.098 ms, 127^2 0.156 / 0.148 ms for sixteen / four lanes; round 5:
    // one lane per cell over the tile-major history replaces the four-lane form, profiles/r05_rt60.txt)
    // sixteen lanes per cell for a few thousand cells (parallelism), one lane per cell over the tile-major plane from ~100 000 (fewest
    // instructions and bytes per sample; it needs two waves per SIMD worth of cells), four in between: profiles/r05_rt60.txt
    // (round 6: counted on the cells WITH AN ONSET -- pv_onset_kernel runs in front of every pass that asks -- instead of the cells
    // of the window's ever-non-zero tiles: a closed room of 4 400 reached cells sat in 13 000 cells of tiles and took the four-lane form)
    if (activeCells <= kRt60WaveMaxCells) return 16;
    if (activeCells <= kRt60TileMinCells || !a.rt60Tile) return 4;  // (no launch of the lane-per-cell form: AnalyzeArgs::rt60Tile)
    return a.histPlane * 4 * 16 < (1ll << 31) ? 1 : 4;  // (the lane-per-cell form reaches a chunk's planes through one descriptor and scalar offsets)
}

struct Rt60Cell {
    int s;              // result index, < 0: nothing to do
    CellHistory hc;
    int startingPoint;  // onset + N_dry + 1
};

// the part shared by both forms: which cell, its history, its onset (read back from the delay map)
__device__ __forceinline__ Rt60Cell rt60Cell(const AnalyzeArgs& a, const DynParams& dyn, int X, int Y) {
    Rt60Cell c{-1, {nullptr, 0}, 0};
    if (X >= a.gx || Y >= a.gy) return c;
    const int s = X * a.gy + Y;
    const float d = a.delay[s];
    if (d == FLT_MAX) return c;
    c.s = s;
    c.hc = CellHistory{a.hist + histOffset(X + a.G - dyn.histRow0, Y + a.G - dyn.histCol0, a.rxi, a.wi, dyn.histTilesY),
                       a.histPlane};
    c.startingPoint = (int)d + a.nDry + 1;
    return c;
}

__device__ __forceinline__ float rt60FromSums(const AnalyzeArgs& a, int startingPoint, float xysum, float ysum) {
    const int endPoint = a.T - a.nCut;
    const int regressN = endPoint - startingPoint;
    const float rn = (float)regressN;
    const float xmean = (rn - 1.0f) * 0.5f;
    const float xsum = rn * xmean;
    const float denominator = (1.0f / 12.0f) * rn * (rn * rn - 1.0f);
    const float ymean = ysum / rn;
    const float numerator = xysum - ymean * xsum - xmean * ysum + rn * xmean * ymean;
    const float slopePerSample = numerator / denominator;
    const float slopePerSec = slopePerSample * (float)a.fs;
    return -60.f / slopePerSec;
}


}  // namespace pva
