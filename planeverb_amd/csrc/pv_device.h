// pv_device.h -- device-side data layout shared by the HIP kernels (pv_kernels.hip) and the host solver.
//
// HBM layout (all planes share one padded geometry, SoA):
//   rows  = G + ntx*RXI + G,  pitch = roundup(G + nty*WI + G, 64) floats
//   array cell (x, y) of the reference's (gx+1) x (gy+1) grid lives at padded (x + G, y + G);
//   y is the contiguous dimension (reference index = x*(gy+1) + y, FDTD.cpp:99) so wave lanes run along y.
//   pr/vx/vy : float32, ping-pong pair (a fused K-step launch reads one set and writes the other)
//   coef     : three float32 per cell: the coefficients of its x and y face + its beta (FaceCoef below)
//              (read by general tiles and the analysis only)
//   hist     : float32 pr[t][window tile][row in tile][col in tile] over a tile-aligned window around the listener
//              (tile-major: a tile's block of one step is one contiguous chunk; cells that the pulse cannot have
//              reached are exactly zero and are not stored)
// The guard band (G cells on every side, plus tile overhang) holds zeros / wall codes and is never written
// with anything else, so no kernel needs a bounds check.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pva {

// analysis constants (reference: include/PvTypes.h:83-101), usable in device code
constexpr float kCDev = 343.21f;
constexpr float kAudibleThresholdDev = 0.00000316f;
constexpr float kDistanceGainDev = 0.891251f;
constexpr float kDelayCloseDev = 5.f;

// Face coefficients.  beta, the wall admittance Y = (1 - R) / (1 + R) and the absorbing grid edges (FDTD.cpp:143-223) fold
// into ONE float per face, stored per cell for its x face (neighbour (x-1, y)) and its y face (neighbour (x, y-1)):
//   air|air        : NaN (kAirFaceBits)  ->  v = v - C * (p_i - p_n)
//   wall(n)|air(i) : -Y_n                ->  v = k * (p_i + p_n)   (p_n = 0 inside the wall)
//   air(n)|wall(i) : +Y_i
//   wall|wall      : +0
//   grid edges     : the same two forms with Y = 1
// Rounds 1-2 stored 8-bit indices into a 256-entry table of such values (127 absorption values alive at once), round 3 first
// 16-bit ones; the values themselves need no table, no palette and no limit (the reference has none), and a general tile's
// load phase loses its dependent table look-up.  beta of the cell (1.f = air) rides along as a third float: it cannot be told
// from the coefficients when an absorption above 1 makes Y negative, and one 12-byte load per row keeps a general tile at four
// loads per row (a fifth passed the 63 loads a wave can have in flight: 2048^2 BigRoom.pv 2.6 % slower).
struct FaceCoef {
    float kx, ky, beta;
};
constexpr uint32_t kAirFaceBits = 0x7fc00000u;  // the quiet NaN pv_coef_kernel writes for an air|air face
// material plane (host -> pv_coef_kernel): NaN = air cell (beta 1), else the cell's admittance Y (beta 0)

struct Geometry {
    int gx, gy, NX, NY;
    int G;           // guard width
    int rxi, wi;     // tile interior rows / columns
    int ntx, nty;    // tiles
    int rows, pitch; // padded plane
    // row slab of a larger grid (single-grid decomposition, pv_slabs.cpp): local array row 0 is row x0 of the whole
    // grid, whose cell array has NXg = gxg + 1 rows.  A whole grid has x0 = 0, NXg = NX, gxg = gx.
    int x0, NXg, gxg;
};

// per-run parameters that change with the listener; lives in device memory so a captured graph can be replayed
struct DynParams {
    int lrow, lcol;          // listener cell, padded coordinates
    int histRow0, histCol0;  // padded coordinates of the history window origin (tile aligned)
    int histTileX0, histTileY0, histTilesX, histTilesY;
    int numGeneral;          // live entries of generalList (the launch grid is sized for its capacity)
    int numSeg;              // live entries of segList (row-streaming air segments, pv_seg.h)
};

// A row-streaming air segment (pv_seg.h): interior rows [row0, row0 + nrows) of the array x tile columns
// [tj0, tj0 + w); every face in its loaded region (K cells around) is air|air and the listener is not inside.
struct SegDesc {
    int row0, nrows, tj0, w;
};

struct StepArgs {
    const float* prIn;
    const float* vxIn;
    const float* vyIn;
    float* prOut;
    float* vxOut;
    float* vyOut;
    const FaceCoef* coef;  // face coefficients + beta of every padded cell (general tiles only)
    const float* pulse;    // T floats
    float* hist;           // window base, plane stride histPlane
    int* tileFirst;        // per tile: first step block in which the tile was non-zero (INT_MAX = never)
    const uint8_t* tileClass;   // per tile: 0 = all faces air|air (air kernel), 1 = general kernel
    const uint8_t* tileDead;    // per tile (NULL = feature off): 1 = every interior cell is wall with wall|wall faces, so
                                // its pr, vx, vy are identically zero in a run that starts from zero fields: the general
                                // arm skips it (thick walls of a 25 m scene at fine resolution: 17 % of all tiles)
    const int* generalList;     // tiles for the general kernel: class-1 tiles + tiles holding the listener
    int numGeneral;             // capacity of generalList used to size the grid; live count is dyn->numGeneral
    const SegDesc* segList;     // row-streaming air segments of this run (pv_step_seg_kernel; NULL = tile kernels)
    int numSeg;                 // capacity used to size the grid; live count is dyn->numSeg
    const DynParams* dyn;
    int* errFlag;
    long long histPlane;   // floats per recorded step (= window tiles x RXI x WI: tile-major, no padding)
    long long planeBytes;  // bytes of one padded float plane
    int histPitch;         // pitch of the slabs' boundary-row history arrays only ([T][histPitch]); planes are tile-major
    int pitch;
    int G;
    int ntx, nty, ntiles;
    int gx, gy;            // the reference's grid size (cells are 0..gx x 0..gy): edge tiles locate the ghost column
    int bandRows;          // tile rows per XCD band = ceil(ntx / 8)
    int packed;            // air kernel: packed-f32 (v_pk_*) arithmetic variant
    const uint8_t* tileOpen;    // streaming analysis only (else NULL): per tile, 1 while any of its cells still has an
                                // open forward-analysis window, or the tile holds a registered emitter
    const uint8_t* nzIn;   // per tile: non-zero at the end of the previous launch (conservative)
    uint8_t* nzOut;        // per tile: non-zero at the end of this launch
    int tileOrder;         // air-kernel block -> tile mapping (0 linear, 1 XCD band row-major, 2 band column-major)
    int patchStrip;        // persistent patch kernel (pv_patch.h): bits 8.. = measurement switches (0 in real runs)
    long long* patchTrace; // development aid (PV_PATCH_TRACE=1): s_memtime stamps of the phases of block 0's waves, or NULL
    int t0;                // first global step of this launch
    int histSlot;          // history plane index of step t0 (= t0, or t0 % ring length in streaming mode)
    int nsteps;            // steps in this launch (<= K)
    int withPulse;
    int record;            // write pr history
    int dense;             // record even all-zero tiles
    float courant;
    int inBytes;           // extent of the INPUT planes' buffer descriptors: planeBytes, or 0 for the first launch of a
                           // run -- every field load is then out of range and returns 0 without touching memory,
                           // which is the run's zero initial state (no reset pass over the planes)
    int sweepReverse;      // tile order 3 only.  bit 0: the XCD's tiles are walked from its last tile row to its first -- odd launches of
                           // a run set it (PVA_OPT_ALTERNATE_SWEEPS): a launch then reads first what the previous launch wrote last,
                           // i.e. what is still in the 256 MiB Infinity Cache, instead of streaming through it.  bit 1: 2 x 4 regions
                           // instead of 8 strips of tile columns (PVA_OPT_XCD_REGIONS)
};

// Batched launch (pv_step_batch_kernel): up to kBatchMax independent runs of identically configured solvers advance
// in ONE launch, blockIdx.y = run.  The whole table travels by value in the kernarg segment (scalar loads).
constexpr int kBatchMax = 8;
struct BatchArgs {
    int n;        // live runs (= gridDim.y)
    int gblocks;  // general-tile blocks per run: max over the runs' capacities, rounded up to a multiple of 8
    StepArgs a[kBatchMax];
};

// pv_begin_run_kernel: the per-run parameters travel from pinned host memory to HBM inside a kernel (a node of the
// captured run graph like every other launch) together with the per-tile bookkeeping resets
struct BeginArgs {
    const DynParams* dynHost;  // pinned, device-visible
    DynParams* dyn;
    const int* listHost;       // pinned, dynHost->numGeneral live entries
    int* list;
    const SegDesc* segHost;    // pinned, dynHost->numSeg live entries (NULL = no segments)
    SegDesc* seg;
    int segCap;
    int* tileFirst;            // NULL = leave the per-tile state alone (stencil-only stepping)
    uint8_t* nz0;
    uint8_t* nz1;
    uint8_t* tileOpen;         // streaming analysis only (else NULL): all tiles start open
    int* errFlag;
    int ntiles;
    int tileFirstInit;         // INT_MAX (recorded from the first non-zero launch) or 0 (dense history)
    int listCap;
    // row bands (Solver::enqueueSteps): each band of tile rows is launched with its own view of the run parameters
    // (listener row, window origin and general-tile count relative to the band's first tile row)
    const DynParams* dynBandsHost;  // pinned, nbands entries (NULL when the run is not banded)
    DynParams* dynBands;
    int nbands;
    unsigned* zeroWords;   // words to clear (the resident kernel's flags: no memset node beside the launch), or NULL / 0
    int nZero;
};

// whole-grid-resident kernel for grids that fit one CU's LDS (pv_small_grid_kernel)
struct SmallArgs {
    float* prOut;
    float* vxOut;
    float* vyOut;
    const FaceCoef* coef;
    const float* pulse;
    float* hist;
    const DynParams* dyn;
    long long histPlane;
    int histPitch;
    int pitch, G;
    int NX, NY;
    int T;
    int record;
    float courant;
    int rxi, wi;  // tile interior (the history planes are tile-major: histOffset)
};

// resident kernel (pv_resident.hip): every tile a workgroup that stays on its CU for all T steps of a run; epochs of K
// steps, neighbours hand their interiors over through the two buffer sets + one flag word per tile
struct ResidentArgs {
    float* pr[2];
    float* vx[2];
    float* vy[2];          // the two buffer sets: epoch e reads set e & 1 (nothing for e = 0) and publishes into the other
    const FaceCoef* coef;
    const float* pulse;    // >= T floats
    float* hist;           // window base (the window is the whole grid), plane stride histPlane
    int* tileFirst;        // per tile: first step block in which the tile was non-zero (every block resets its own tile's entry)
    DynParams dynVal;      // the run's parameters, by value: no begin-run launch in front of this kernel (it read them from pinned
                           // host memory, 7-8 us of a 0.3 ms run); block 0 leaves a copy at dynOut for the analysis kernels
    DynParams* dynOut;
    int* errFlag;          // 3 = a block gave up waiting for a neighbour (every block then leaves: the run failed)
    unsigned* flags;       // ntiles epoch counters + 1 abort word + 1 claim counter (one-XCD mode), zero before every launch (the
                           // previous run's last kernel clears them: launchRunFinish)
    int xcdMode;           // 1: the blocks that run on XCD xcdTarget claim the tiles; hand-off through that XCD's L2
    int xcdTarget;
    long long histPlane;   // floats per recorded step
    long long planeBytes;  // bytes of one padded float plane
    int pitch, G;
    int ntx, nty, ntiles;
    int T;
    float courant;
    unsigned long long* stamp;  // pinned host words, or NULL: stamp[0] = the 100 MHz counter when the block that holds tile 0 starts
                                // (Solver::stampTimed_: a run's timings without event packets between its kernels)
};

// slab decomposition, neighbours on one device: words in device memory instead of cross-queue events (pv_halo_push_kernel)
struct HaloHandoff {
    unsigned* count;           // blocks of this slab's push launches that have finished, since the run began (NULL: hand-off by events)
    unsigned* raise[2];        // the words the upper / lower neighbour waits on (NULL: no such neighbour, or it is on another device)
    const unsigned* await[2];  // this slab's own words, raised by the upper / lower neighbour
    unsigned seq;              // sweep index + 1
    int* err;                  // this slab's error flag (5 = a neighbour's halo never arrived)
    unsigned* abortWord;       // the group's abort word (AnalyzeArgs::abortWord), raised together with err
};

// sparse-emitter mode with the forward sums inside the stencil (pv_stream.h)
struct OpenArgs {
    int* sOnset;            // per result cell: onset step, -1 = none yet
    float* sEdry;
    float* sFx;
    float* sFy;
    uint8_t* cellsOpen2;    // per tile half: some interior cell's dry window is still open
    const int* openList;    // tile * 2 + half
    const int* openCount;
    int* nextCount;         // the other of the two counters (they alternate from launch to launch)
    int nDir, nDry;
    int gxRes, gyRes;       // result map
};

struct ClassifyArgs {
    const uint8_t* tileClass;   // static: 0 air, 1 general, 2 edge
    const uint8_t* tileEmit;    // holds a registered emitter
    const uint8_t* tileOpenRing;  // accumulate pass: ring tile still has an open window (or has not been reached)
    const uint8_t* nzPrev;      // per tile: non-zero at the end of the previous launch (conservative)
    const uint8_t* cellsOpen2;
    const DynParams* dyn;
    uint8_t* classOut;          // per-launch classes for the merged kernel
    uint8_t* ringOpenOut;       // the step kernels' `tileOpen`
    uint8_t* nzNext;            // this launch's flag plane: cleared for open tiles (their two halves only ever set it)
    int* openList;
    int* openCount;             // zeroed by the caller before the launch
    int ntx, nty, G, K, rxi, wi, rows;
    int withPulse;
};

// cells with an onset up to which the decay-time pass runs sixteen lanes per cell / from which it runs one lane per cell (four in
// between): profiles/r06_rt60.txt -- 95^2 (8 617 cells) 0.066 / 0.074 ms with sixteen / four lanes, 127^2 (15 310) 0.107 / 0.101;
// 254^2 (61 231) 0.317 / 0.298 with four / one, 318^2 (96 059) 0.481 / 0.366.  (Rounds 4-5: 8 192 and 98 304.)
constexpr int kRt60WaveMaxCells = 12288;
constexpr int kRt60TileMinCells = 49152;

// where the far cells of the last run begin, and what their listener direction is (output gathers, pv_far_dir_kernel)
struct FarInfo {
    int on;              // 0: every cell's direction is in the result planes
    int r0, c0, nr, nc;  // the last run's window block of the result map: directions inside it are in the planes
    int gy;
    float lx, lz, dx;
    const int* box;      // device words {r0, c0, r1, c1}: bounding box of the last run's REACHED cells (AnalyzeArgs::box), or NULL.
                         // With it, only the box grown by one cell (inside the window block) has its directions in the planes
};

struct AnalyzeArgs {
    // slab groups with hand-off words: raised by a push kernel that waited for a neighbour in vain -- the run's fields are
    // not a run's fields then, and the analysis must leave the result maps (whose no-onset cells carry over to later runs) as
    // they are; SlabGroup::run repeats the run with stream events.  NULL everywhere else
    const unsigned* abortWord;
    const float* hist;
    const FaceCoef* coef;
    const int* tileFirst;
    const DynParams* dyn;
    float* out;    // 8 planes of gx*gy floats (SoA): occlusion, wet gain, RT60, lowpass, direction x/y, source
                   // direction x/y -- plane k of cell s at out[k*resN + s].  The AoS view the C-ABI hands out
                   // (PlaneverbOutput per cell) is packed on demand (pv_pack_results_kernel)
    float* delay;  // gx*gy
    long long resN;  // gx*gy
    long long histPlane;
    int histPitch;
    int pitch, G;
    int gx, gy;
    int rxi, wi, nty;
    int winRows, winCols;  // extent of the history window in cells: the analysis kernels' launch grid
    int* activeCount;      // [0]: cells of the window's ever-non-zero tiles (an upper bound of the reached cells: chooses the
                           // decay-time form on the device), [1]: cells with an onset, [3]: silent air cells, [4]: entries of
                           // unitList (all counted by pv_onset_kernel and reset by the first launch of the analysis)
    int* unitList;         // the 64-cell groups of the history plane (group u = plane offsets 64 u .. 64 u + 63) that hold a cell
                           // with an onset, in no particular order: what pv_encode_kernel and pv_rt60_tile_kernel work on, one
                           // wave per entry -- consecutive workgroups then hold equal amounts of work wherever the reached cells
                           // lie in the plane (launched over the plane, a third of the SIMDs got three waves of it, most one)
    int* dirScratch;       // winRows x winCols ints for the listener-direction pointer jumping
    int dirJump;           // listener direction by pointer jumping (wide windows) instead of the plain walk
    int rt60Lanes;         // 0 = by the number of reached cells (rt60LanesPerCell); 16 / 4 / 1 = that form of the decay-time pass
    int rt60Tile;          // the lane-per-cell form of the decay-time pass is launched (launchRt60Forms): it may be chosen
    int T;
    int nDir, nDry, nWet, nCut;
    unsigned fs;
    int res;
    float dx;
    float courant;
    float efree;
    float lx, lz;        // listener, metres
    int lcx, lcy;        // listener cell by reciprocal multiply (Analyzer.cpp:200-201)
    // row slab of a larger grid: result row X of this map is row X + x0 of the whole grid (position arithmetic), and the
    // pressure history of the row above the slab's first row -- needed by the vx recurrence there -- comes from the
    // neighbouring slab as a dense [T][histPitch] array (window columns; zeros where nothing was recorded)
    int x0;
    const float* histAbove;
    // Far cells (cells outside the history window: no onset, listener direction = unit vector listener -> cell) are not
    // rewritten for the whole map on every run: lazyFar = 1 resets only the cells of the PREVIOUS run's window block
    // [prevR0, +prevNR) x [prevC0, +prevNC) and of this run's (pv_far_frame_kernel); the direction of the other far cells
    // is materialised when a whole-map reader asks (pv_far_dir_kernel) or computed in closed form by the output gathers.
    int lazyFar;
    int prevR0, prevC0, prevNR, prevNC;
    // Round 6: the window block is an upper bound of what a run can reach ((2T + 3)^2 cells around the listener: 760 000 at
    // T = 435), a closed room reaches a few thousand of them -- and the far frame and the listener-direction passes moved 62 MB
    // per run for them (profiles/r05_analysis_pmc.md).  box = four device words {r0, c0, r1, c1}, the inclusive bounding box of
    // the cells pv_onset_kernel finds an onset in (atomics; empty = {INT_MAX, INT_MAX, -1, -1}); prevBox = the same of the
    // previous analysed run of this solver.  With them there is no far-frame launch: pv_onset_kernel itself gives "no onset" back
    // to prevBox's cells that this run does not reach (every other cell of the map holds it already); the direction passes cover
    // the box grown by one cell (a cell further away has no neighbour with an onset: its walk stays put, Analyzer.cpp:365-391 --
    // the closed form), and every other cell of the window is a far cell like the ones outside it (FarInfo::box).  The decay-time
    // launch behind the onsets (pv_rt60_groups_kernel) empties prevBox: it is the box of the run after this one.  NULL: the
    // window-wide passes (slabs, whole-grid windows, the experimental one-launch analysis, PLANEVERB_AMD_NEAR_BOX=0).
    int* box;
    int* prevBox;
    unsigned long long* stamp;  // pinned host words, or NULL: stamp[1] = the 100 MHz counter when the analysis' first kernel starts
    const int* labels;   // per array cell ((gx + 1) x (gy + 1), index x * labelNY + y): its 4-connected AIR component, -1 for a wall
                         // cell -- or NULL (large grids, slabs).  Pressure never crosses a wall cell (beta = 0 keeps it at zero,
                         // FDTD.cpp:139, and a wall|air face's velocity is a multiple of the AIR cell's pressure, :165-168): a cell of
                         // another component than the listener's is exactly zero for the whole run.  pv_onset_kernel leaves such
                         // cells without reading their history (the air outside a closed room, in the tiles its walls cross: every
                         // one of them cost a scan of all T samples, 100 of the kernel's 140 us at 512^2 / T = 3179)
    int labelNY;
    int wholeWindow;     // the history window is the whole grid (the reference's presets): no far cells at all -- pv_onset_kernel writes
                         // "no onset" itself and counts the active cells, the direction pass covers every cell: no far-frame launch
    // streaming analysis (sparse-emitter mode): the history is a ring of `ring` planes and the forward sums of
    // every cell are carried in per-cell state planes between passes
    int ring;            // 0 = full history (plane index = t), else plane index = t % ring
    int tA, tB;          // step range of this accumulate pass
    int* sOnset;
    float* sEdry;
    float* sFx;
    float* sFy;
    float* sVx;
    float* sVy;
    uint8_t* tileOpenOut; // per tile: set to 1 by any cell whose window is still open after this pass
    // forward sums inside the stencil (pv_stream.h): the accumulate pass skips the cells of fused tiles.  NULL = off
    const uint8_t* fuseClass;  // static tile classes
    const uint8_t* fuseEmit;   // per tile: holds a registered emitter
    int fuseK;
    const int* ringList;       // the tiles this pass serves when the others are fused (NULL: every tile, addressed by cell)
    int numRing;
    const int* emCells;  // registered emitter cells: X*gy + Y
    float* emTrace;      // numEmitters x T pressure traces
    int numEmitters;
};

// fused analysis of the small grids (pv_fused.hip)
constexpr int kFusedCtlWords = 16;  // ticket, one counter per phase, the workers that have left
struct FusedArgs {
    AnalyzeArgs a;
    unsigned* ctl;          // kFusedCtlWords words, zero before the launch (the launch leaves them at zero)
    const float* carrySrc;  // two iterations in flight: the other solver's result planes (no-onset cells take their record), or NULL
    int* errFlag;           // 6 = a worker waited for a phase in vain
};

}  // namespace pva
