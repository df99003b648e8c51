/* planeverb_amd.h -- C-ABI of libplaneverb_amd.so (MI355X / gfx950 build of Planeverb's FDTD + IR-analysis path)
 *
 * Part 1 is the reference's own flat C-ABI, symbol for symbol -- the functions Unity P/Invokes from
 * ProjectPlaneverbUnityPlugin (reference: ProjectPlaneverb/PlaneverbUnityPluginAPI/PlaneverbUnity.cpp:12-135,
 * C# mirror PlaneverbContext.cs:25-60).  A build of the Acoustics module that loads this library instead of
 * ProjectPlaneverbUnityPlugin.dll needs no source change.  threadExecutionType 0 (pv_CPU) and 1 (pv_GPU, the value
 * the C# enum already defines, PlaneverbConfig.cs:23-29) are accepted alike and BOTH run on the HIP device: this
 * library has no CPU path and fails loudly without a HIP device.
 *
 * Part 2 (PvAmd*) is an extension: a handle-based, synchronous batch interface to the same solver, used by the
 * benchmarks, the parity tests and the multi-GPU sharding layer.  It replaces nothing in the reference; it exposes
 * what the reference's classes Grid / FreeGrid / Analyzer expose to its own Context (PvContext.cpp:63-94).
 *
 * Plain C types only.  No C++ exception crosses this boundary: errors are reported by return code
 * (0 = ok) and PvAmdLastError().
 */
#ifndef PLANEVERB_AMD_H
#define PLANEVERB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define PVA_EXPORT __attribute__((visibility("default")))
#else
#define PVA_EXPORT
#endif

/* ------------------------------------------------------------------------------------------------------------
 * Part 1 -- reference C-ABI
 * ---------------------------------------------------------------------------------------------------------- */

/* PlaneverbUnity.cpp:66-76 (returned BY VALUE, 8 floats) */
typedef struct PlaneverbOutput {
    float occlusion;        /* dry/obstruction gain; -1 (PV_INVALID_DRY_GAIN) = invalid, PvTypes.h:80 */
    float wetGain;
    float rt60;
    float lowpass;
    float directionX;
    float directionY;
    float sourceDirectionX;
    float sourceDirectionY;
} PlaneverbOutput;

/* PlaneverbUnity.cpp:12-20 (no-ops) */
PVA_EXPORT void UnityPluginLoad(void* unityInterfaces);
PVA_EXPORT void UnityPluginUnload(void);

/* PlaneverbUnity.cpp:25-40 -> Planeverb::Init (PvContext.cpp:25-32).  Invalid config (res < 275, size 0,
 * tempFileDir NULL: PvContext.cpp:101-107) leaves the module un-initialised instead of throwing. */
PVA_EXPORT void PlaneverbInit(float gridSizeX, float gridSizeY, int gridResolution, int gridBoundaryType,
                              char* tempFileDir, int maxThreadUsage, int threadExecutionType);
/* PlaneverbUnity.cpp:42-46 */
PVA_EXPORT void PlaneverbExit(void);
/* PlaneverbUnity.cpp:48-52 ; -1 when the module is not initialised (EmissionManager.cpp:13) */
PVA_EXPORT int PlaneverbEmit(float x, float y, float z);
/* PlaneverbUnity.cpp:54-58 */
PVA_EXPORT void PlaneverbUpdateEmission(int id, float x, float y, float z);
/* PlaneverbUnity.cpp:60-64 */
PVA_EXPORT void PlaneverbEndEmission(int id);
/* PlaneverbUnity.cpp:78-92 -> Planeverb::GetOutput (FDTD.cpp:16-58); O(1), no lock, no device access (lock-free: a
 * reader repeats its 32-byte copy only if the once-per-iteration publish step ran meanwhile).
 * Sparse-emitter mode (configurations whose T-step pressure history does not fit the device, e.g. 25 m at 16 kHz; chosen
 * by PlaneverbInit by itself, forced by the environment variable PLANEVERB_AMD_LIVE_STREAMING=1): occlusion, lowpass and
 * both directions are valid for every cell as usual; wetGain and rt60 are computed for the cells of the emitters that
 * existed when the iteration started (an emitter added or moved to another cell gets them one iteration later). */
PVA_EXPORT PlaneverbOutput PlaneverbGetOutput(int emissionID);
/* PlaneverbUnity.cpp:94-107 ; -1 when not initialised (GeometryManager.cpp:20) */
PVA_EXPORT int PlaneverbAddGeometry(float posX, float posY, float width, float height, float absorption);
/* PlaneverbUnity.cpp:109-123 */
PVA_EXPORT void PlaneverbUpdateGeometry(int id, float posX, float posY, float width, float height,
                                        float absorption);
/* PlaneverbUnity.cpp:125-129 */
PVA_EXPORT void PlaneverbRemoveGeometry(int id);
/* PlaneverbUnity.cpp:131-135 */
PVA_EXPORT void PlaneverbSetListenerPosition(float x, float y, float z);

/* Extensions to the live module (not in the reference's flat ABI) */
/* One sample of an impulse response as the reference stores it (Cell, PvTypes.h:106-121: 16 bytes) */
typedef struct PlaneverbCell {
    float pr, vx, vy;
    short b;  /* beta of the cell during the run (0 = wall / ghost row or column) */
    short by; /* never read by the solver; carried for layout compatibility (Grid.cpp:93-108,241-242,281-290) */
} PlaneverbCell;
/* Planeverb::GetImpulseResponse (Planeverb.h:47, FDTD.cpp:60-79) for the live module: the impulse response of the
 * last COMPLETED iteration at the cell holding world position (x, z) -- (int)(x/dx), (int)(z/dx) -- as reference
 * Cells.  Writes min(capacity, T) cells to `out` and returns T (the response length; call with capacity 0 to size the
 * buffer), 0 for a position outside the cell array, -1 when the module is not initialised / on error.  Waits for the
 * iteration in flight (the reference reads the cube while the worker rewrites it); a debugging call, like upstream. */
PVA_EXPORT int PlaneverbGetImpulseResponse(float x, float y, float z, PlaneverbCell* out, int capacity);
/* Load a .pv scene (PlaneverbSandbox/src/Editor/Editor.cpp:245-281) into the live module; returns #boxes or <0 */
PVA_EXPORT int PlaneverbLoadScene(const char* pvPath);
/* Number of completed simulation iterations since Init (an iteration = FDTD + analysis, PvContext.cpp:74-93) */
PVA_EXPORT long long PlaneverbIterationCount(void);
/* Block until at least `count` iterations have completed, or timeoutMs elapsed; returns the iteration count */
PVA_EXPORT long long PlaneverbWaitIterations(long long count, int timeoutMs);
/* 1 if the module is initialised and its simulation worker is alive (0 after a worker error: PvAmdLastError) */
PVA_EXPORT int PlaneverbIsRunning(void);
/* why the simulation worker stopped ("" while it runs / when the module is down); valid until this thread's next call */
PVA_EXPORT const char* PlaneverbWorkerError(void);
/* 1 if the live module runs in the sparse-emitter mode (see PlaneverbGetOutput), else 0 */
PVA_EXPORT int PlaneverbIsStreaming(void);

/* ------------------------------------------------------------------------------------------------------------
 * Part 2 -- batch solver handle (extension)
 * ---------------------------------------------------------------------------------------------------------- */

typedef struct PvAmdSolver PvAmdSolver;

typedef struct PvAmdInfo {
    int gx, gy;             /* (int)m_gridSize: result map is gx*gy, cell array (gx+1)*(gy+1)  (Grid.cpp:48-53) */
    int T;                  /* response length in samples (Grid.cpp:55) */
    int fs;                 /* sampling rate (Grid.cpp:390-396) */
    int res;
    float dx, dt;
    float efree;            /* FreeGrid energy at 1 m (FreeGrid.cpp:71-94) */
    int device;
    int stepsPerLaunch;     /* K: time steps fused per kernel launch */
    int tileRows, tileCols; /* interior cells per wave tile */
    int pitch, rows;        /* padded device array geometry (floats per row, rows) */
    int histRows, histPitch;/* history window geometry */
    int numGeometry;
    long long deviceBytes;  /* bytes of HBM held by this solver */
    int streamFuse;         /* streaming analysis: 1 = the forward sums of air tiles advance inside the step kernel (PVA_OPT_STREAM_FUSE as resolved for this grid and tile) */
    int residentKernel;     /* 1 = runs go out as ONE launch of the resident kernel (PVA_OPT_RESIDENT_KERNEL as resolved for this grid and tile) */
} PvAmdInfo;

typedef struct PvAmdTimings {
    float fdtdMs;           /* HIP-event time of the T-step loop (incl. IR record) of the last run */
    float analysisMs;       /* HIP-event time of both analysis kernels of the last run */
    float geometryMs;       /* material upload + face-code build, last time it ran */
    float stepKernelMs;     /* fdtdMs / number of step launches */
    int stepLaunches;
    float airKernelMs;      /* mean duration of one air-tile step-kernel launch (PVA_OPT_TIME_KERNELS) */
    float generalKernelMs;  /* mean duration of one general-tile step-kernel launch */
    int airLaunches, generalLaunches;
    float stepLoopMs;       /* HIP-event time of the back-to-back step launches alone (fdtdMs minus the field reset);
                               0 when the run was replayed from a hipGraph */
    int reachedCells;       /* cells with an onset in the last run's analysis (Analyzer.cpp:146-165): the impulse responses that
                               were actually analysed -- the others leave at once */
    int activeCells;        /* cells of the history window's tiles that ever held a non-zero value: what the analysis kernels
                               look at (an upper bound of reachedCells) */
    int silentCells;        /* air cells among them whose whole history stayed below the audible threshold (no onset: Analyzer.cpp:160-165);
                               many of them make the next run's analysis look for an audible sample before anything else */
} PvAmdTimings;

/* option keys for PvAmdSetOption (must be set before the first run) */
enum {
    PVA_OPT_DENSE_HISTORY = 1, /* 1 = record every tile every step (no zero-tile skipping) */
    PVA_OPT_NUM_STEPS = 2,     /* override T (extension); 0 = reference value */
    PVA_OPT_SKIP_ANALYSIS = 3, /* 1 = PvAmdRun does the FDTD loop only */
    PVA_OPT_USE_GRAPH = 4,     /* replay a run from a captured hipGraph: 0 = auto (launch-bound small grids), 1 = always, 2 = never */
    PVA_OPT_STEPS_PER_LAUNCH = 5, /* K: time steps fused per kernel launch (tuning) */
    PVA_OPT_TILE_ROWS = 6,     /* interior rows of a wave tile (tuning; must pair with a compiled K) */
    PVA_OPT_NO_FREE_GRID = 7,  /* 1 = skip the free-field run (efree = 0; stencil-only use) */
    PVA_OPT_TIME_KERNELS = 8,  /* N > 0: HIP events around every Nth step-kernel launch (per-kernel durations) */
    PVA_OPT_TILE_ORDER = 9,    /* air-kernel workgroup->tile map: 0 linear, 1 XCD band of tile rows walked row-major, 2 the band column-major, 3 XCD strip of tile columns walked row-major (vertical halo neighbours stay in the XCD's L2), >= 4 sub-bands of that many tile rows; default: 3 for grids of >= 4500 tiles (3072^2 and up), else 1 */
    PVA_OPT_SMALL_GRID_KERNEL = 10, /* the kernel that keeps the whole grid in one CU's LDS for all T steps: 0 = auto (grids of up to 1536 array cells, where it beats the replayed tile-kernel graph: 28^2 ... 38^2), 1 = whenever the grid fits one CU (up to ~110^2), 2 = never */
    PVA_OPT_PACKED_MATH = 11,  /* air-tile kernel arithmetic: 1 = packed f32 (default), 0 = scalar f32 */
    PVA_OPT_STREAMING_ANALYSIS = 12, /* 1 = sparse-emitter mode: ring history + incremental analysis (see PvAmdSetEmitters) */
    PVA_OPT_STREAM_ROWS = 13,  /* N > 0: the air part of the grid is advanced by about N row-streaming segments per sweep (a wave streams down a 256-column strip, K time levels in flight) instead of one wave per air tile; tile configurations (8, 40) and (12, 36) only, ignored elsewhere and with slabs / row bands / graphs / streaming analysis.  Bit-identical; experimental: slower than the tile kernels at 4096^2.  Default 0 = off */
    PVA_OPT_MERGED_LAUNCH = 14, /* 1 (default) = general + air tiles in one launch per K steps; 0 = two kernels, two streams */
    PVA_OPT_ROW_BANDS = 16,    /* B > 1: every K-step sweep is launched as B bands of tile rows on B HIP streams; band b of sweep n+1 waits only for bands b-1, b, b+1 of sweep n, so consecutive sweeps of ONE run overlap (no chip-wide drain between launches).  Bit-identical; on the current runtime the cross-queue event waits cost more than the overlap gains (4096^2, one run in flight: 1.52e12 with one launch per sweep, 1.34e12 with two bands, 1.13e12 with three -- round 5), so 0 = auto means 1 = one launch per sweep.  Large grids with the merged kernel only; ignored elsewhere (streaming analysis, graphs, batched runs) */
    PVA_OPT_PATCH_KERNEL = 17, /* air tiles by the persistent per-CU kernel with LDS-DMA run-ahead (csrc/pv_patch.h: one 512-thread workgroup per CU, the next 4-tile patch lands in LDS while the current one computes) instead of one wave per tile; general tiles in a launch of their own.  Only the large-grid tile (steps per launch 12, tile rows 36) has the kernel; ignored elsewhere and with streaming analysis, slabs, edge tiles, kernel timing.  -1 = default for the configuration, 0 = off, 1 = on */
    PVA_OPT_LAZY_FAR_CELLS = 19, /* 1 (default): a run resets "no onset" / the default listener direction only in the previous and the current history-window block of the result map; the listener direction of the other far cells (unit vector listener -> cell, Analyzer.cpp:365-391,415-428) is materialised when a whole-map read-back asks for it and computed in closed form by PvAmdGetOutput / the output queries.  0: rewrite every far cell on every run (201 MB at 4096^2), the form of rounds 1-2 (validation) */
    PVA_OPT_STREAM_FUSE = 20,  /* streaming analysis only: the forward sums of the analysis (onset, dry energy, source-direction flux) of AIR tiles advance inside the step kernel (csrc/pv_stream.h: open half tiles with the sums in registers); the ring of pressure planes and the accumulate pass then serve only tiles with walls, grid edges, the listener or a registered emitter.  -1 (default): by grid size (on from 6000 tiles, where the ring traffic binds); 0: ring + accumulate pass for every tile (round 2's form); 1: on */
    PVA_OPT_AUX_STREAMS = 21,  /* HIP streams the solver creates beside its own two and never launches on (default 0).  The streams of a process share a handful of hardware queues, handed out by creation order, and two step loops that land on one queue run their launches strictly one after the other.  Measured on MI355X / ROCm 7.0 at 512^2: with 1, the two batch groups' step loops overlap (batched launches 33 instead of 63 us: 3.5e11 instead of 2.1e11 cell-updates/s); with 0, four pipelined single runs keep 2.8e11 instead of 2.5e11.  api.batch_solver_options sets 1 */
    PVA_OPT_RESIDENT_KERNEL = 22, /* the resident kernel (csrc/pv_resident.hip): ONE launch per run, every tile a workgroup that stays on its CU for all T steps and hands its interior to its neighbours every K steps through write-through stores + one flag word per tile (no kernel boundary, no grid barrier) -- for the grids the reference ships (its 275 ... 750 Hz presets on a 25 m scene: 70^2 ... 191^2) and everything else whose history window is the whole grid and whose tiles fit the chip at once.  0 (default) = auto: where the solver runs its default tile for launch-bound grids; 1 = also with an explicitly chosen (steps per launch, tile rows) = (12, 12); 2 = never (the replayed graph of tile-kernel launches) */
    PVA_OPT_RT60_LANES = 23,   /* wet gain + decay time (Analyzer.cpp:235-247,282-327): lanes that share a cell -- 16 (DPP row: few cells, parallel logarithms), 1 (lane per cell over the tile-major history, csrc/pv_rt60.hip: fewest instructions and bytes per sample) or 4 (round 4's blocked form, kept for validation); same bits in every form.  0 (default) = 16 or 1, chosen on the device from the number of reachable cells */
    PVA_OPT_FUSED_ANALYSIS = 29, /* grids whose pressure history covers the whole grid, up to 98 304 cells (the reference's presets): 1 = the whole analysis of a run is ONE launch (csrc/pv_fused.hip: onsets, dry / wet gain, decay time, listener direction as phases of workers that draw items from a ticket counter) -- an arm of the EXPERIMENTAL build of the library (bit-identical, measured slower); the product build refuses it.  -1 / 0 (default) = the separate kernels.  Results do not depend on it */
    PVA_OPT_ANALYSIS_FORK = 28, /* 1 (default) = on windows of 4 096 cells and more (PV_ANALYSIS_FORK_CELLS, csrc/pv_solver.cpp) the wet gain / decay-time pass runs on the solver's second stream beside the dry-gain pass (both only read the onsets); 0 = behind it; 2 = beside it on every window.  Results do not depend on it */
    PVA_OPT_STREAM_PRIORITY = 25, /* 1 = the solver's main stream is created with the device's highest priority.  Streams of different priorities never share a hardware queue (the runtime multiplexes the streams of a process on a handful of them, and launches of streams that share one run one after the other): give every other solver of a group that is meant to run side by side -- two runs in flight on one GPU -- this option.  Results do not depend on it.  Default 0 */
    PVA_OPT_ALTERNATE_SWEEPS = 26, /* tile order 3 only: 1 = odd launches of a run walk every XCD's strip of tiles from its last tile row to its first, so that a launch reads first what the previous launch wrote last (still in the 256 MiB Infinity Cache) instead of streaming through the cache in the order that evicts everything before it is read again.  Results do not depend on it.  -1 = default, 0 = off */
    PVA_OPT_XCD_REGIONS = 27, /* tile order 3 only: 1 = every XCD owns one of 2 x 4 regions of the grid (walked row-major) instead of one of 8 strips of tile columns: a region is twice as wide, so half as many cache lines on its sides are fetched by two XCDs, and a tile's vertical neighbours are still close enough for the XCD's L2.  Results do not depend on it.  -1 = default (on), 0 = strips */
    PVA_OPT_DEBUG_LOSE_FIRST_CAPTURE = 24, /* validation: 1 = the solver's first run-graph capture counts as lost (what a legacy-stream operation of another host thread does to it): that run goes out as plain launches, the next one captures again (tests/test_gpu_parity.py) */
    PVA_OPT_PATCH_STRIP = 18,  /* patch columns per strip of the patch kernel's walk over the grid (development; default 3) */
    PVA_OPT_EDGE_TILES = 15    /* 1 = tiles whose only non-air faces are the grid's absorbing edges run the air-tile code + edge overrides (tile class 2) instead of the general path.  Only the batched kernels of the mirror-pair tiles (K, rows = (8,40), (10,36), (12,36)) have that arm -- inside the merged kernel it slows the air tiles by 25-40 %, DESIGN.md 8.4 -- so every run of such a solver goes through PvAmdRunBatch's kernel (PvAmdRun = a batch of one) and PvAmdRunSteps is refused; ignored for other configurations.  Default 0 */
};

PVA_EXPORT int PvAmdDeviceCount(void);
PVA_EXPORT const char* PvAmdLastError(void);
PVA_EXPORT const char* PvAmdVersion(void);

/* Create the grid for a config (Grid::Grid, Grid.cpp:30-117, + FreeGrid, FreeGrid.cpp:6-34) on HIP device
 * `device`.  PlaneverbCreateGrid is the same function under the name BASELINE.json uses. */
PVA_EXPORT PvAmdSolver* PvAmdCreate(float gridSizeX, float gridSizeY, int gridResolution, int device);
PVA_EXPORT PvAmdSolver* PlaneverbCreateGrid(float gridSizeX, float gridSizeY, int gridResolution, int device);
/* SURVEY.md 8f N4 -- ONE grid decomposed into `nslabs` row slabs (whole tile rows), slab i on HIP device devices[i] (all
 * equal: several slabs on one GPU).  Each slab holds planes, pressure history and result maps for its own rows only;
 * per K-step launch the K boundary rows of pr, vx, vy travel into the neighbours' guard bands, per run the boundary rows'
 * pressure histories and the window block of the per-slab result maps.  The reference has no
 * counterpart (its loop advances the whole grid, PvContext.cpp:63-94); results are bit-identical to PvAmdCreate's.
 * The handle supports: SetOption (tile / step options), GetInfo, Add / Update / RemoveGeometry, LoadScene, Run,
 * GetOutput, CopyResults, CopyFields, CopyHistoryPlane, GetImpulseResponse, CopyPulse, CopyMaterial, GetTimings,
 * GetSlabInfo, Destroy; the other PvAmd* calls return -1 for it. */
PVA_EXPORT PvAmdSolver* PvAmdCreateSlabs(float gridSizeX, float gridSizeY, int gridResolution, const int* devices,
                                         int nslabs);
typedef struct PvAmdSlabInfo {
    int nslabs;
    int row0[16], rows[16], device[16];  /* cell-array rows [row0, row0 + rows) of slab i */
    long long haloBytesPerLaunch;        /* pr, vx, vy boundary rows moved between slabs per K-step launch */
    long long exchangeBytesPerRun;       /* boundary histories + result blocks of the last run */
    long long deviceBytes[16];           /* HBM held by slab i (whole-grid result maps: see PvAmdGetInfo) */
    int handoffWords;                    /* 1: slabs of one device hand over through words in device memory, 0: stream events */
    int streamRedeals;                   /* times every slab was given another stream at creation: the hand-off's dry run timed out or was slow */
    float dryRunUsPerSweep;              /* how long a sweep of that dry run took (launches and sync included; ~4 us per slab when the streams run beside each other) */
} PvAmdSlabInfo;
PVA_EXPORT int PvAmdGetSlabInfo(PvAmdSolver* s, PvAmdSlabInfo* out);
/* The same decomposition with the slabs in DIFFERENT PROCESSES (one rank per GPU): a rank creates ITS slab, the host
 * language moves the halos between ranks (planeverb_amd/dist_slabs.py: torch.distributed send / recv, RCCL on GPU ranks),
 * rank 0 additionally holds the whole-grid maps (PvAmdSlabRoot*).  Per run and rank:
 *   SlabBegin; for li < SlabNumLaunches: { SlabLaunch(li); SlabExportHalo(side) -> neighbour -> SlabImportHalo(side) };
 *   SlabExportEdgeHistory -> rank below -> SlabImportAboveHistory; SlabAnalyze; SlabWindowBlock -> rank 0 ->
 *   SlabRootImportBlock; rank 0: SlabRootBegin before the blocks, SlabRootFinish after them, then SlabRootGetOutput.
 * side 0 = towards the slab above (smaller rows), 1 = towards the slab below.  Buffers are host memory. */
PVA_EXPORT PvAmdSolver* PvAmdCreateSlabRank(float gridSizeX, float gridSizeY, int gridResolution, int device,
                                            int slabIndex, int slabCount);
/* FreeGrid energy (FreeGrid.cpp:71-110) of a config: computed once (rank 0) and given to every slab rank */
PVA_EXPORT int PvAmdComputeEfree(float gridSizeX, float gridSizeY, int gridResolution, int device, float* efree);
PVA_EXPORT int PvAmdSlabSetEfree(PvAmdSolver* slab, float efree);
PVA_EXPORT int PvAmdSlabBegin(PvAmdSolver* slab, float lx, float ly, float lz);
PVA_EXPORT int PvAmdSlabNumLaunches(PvAmdSolver* slab);
PVA_EXPORT int PvAmdSlabLaunch(PvAmdSolver* slab, int li);
PVA_EXPORT int PvAmdSlabHaloFloats(PvAmdSolver* slab);    /* 3 x K x pitch */
/* (the buffers of the next four calls may be HOST or DEVICE memory of the slab's device -- e.g. the device tensors an RCCL
 * transport sends and receives: nothing is staged through the host then) */
PVA_EXPORT int PvAmdSlabExportHalo(PvAmdSolver* slab, int side, float* host);
PVA_EXPORT int PvAmdSlabImportHalo(PvAmdSolver* slab, int side, const float* host);
PVA_EXPORT int PvAmdSlabHistoryFloats(PvAmdSolver* slab); /* T x histPitch */
PVA_EXPORT int PvAmdSlabExportEdgeHistory(PvAmdSolver* slab, float* host);
PVA_EXPORT int PvAmdSlabImportAboveHistory(PvAmdSolver* slab, const float* host);
PVA_EXPORT int PvAmdSlabAnalyze(PvAmdSolver* slab);
/* info4 = {first whole-grid row, first column, rows, columns} of the block; returns the floats needed (7 planes), and
 * fills `host` when capacity suffices; < 0 on error */
PVA_EXPORT long long PvAmdSlabWindowBlock(PvAmdSolver* slab, int* info4, float* host, long long capacityFloats);
typedef struct PvAmdSlabRoot PvAmdSlabRoot;
PVA_EXPORT PvAmdSlabRoot* PvAmdSlabRootCreate(PvAmdSolver* anySlab, int device);
PVA_EXPORT void PvAmdSlabRootDestroy(PvAmdSlabRoot* r);
PVA_EXPORT int PvAmdSlabRootBegin(PvAmdSlabRoot* r, float lx, float ly, float lz);
PVA_EXPORT int PvAmdSlabRootImportBlock(PvAmdSlabRoot* r, const int* info4, const float* host);
PVA_EXPORT int PvAmdSlabRootFinish(PvAmdSlabRoot* r);
PVA_EXPORT int PvAmdSlabRootGetOutput(PvAmdSlabRoot* r, float ex, float ey, float ez, PlaneverbOutput* out);
PVA_EXPORT int PvAmdSlabRootCopyResults(PvAmdSlabRoot* r, float* res8, float* delay);
PVA_EXPORT void PvAmdDestroy(PvAmdSolver* s);
PVA_EXPORT int PvAmdSetOption(PvAmdSolver* s, int key, long long value);
PVA_EXPORT int PvAmdGetInfo(PvAmdSolver* s, PvAmdInfo* out);

/* Geometry (Grid::AddAABB / RemoveAABB / UpdateAABB, Grid.cpp:136-303); applied before the next run */
PVA_EXPORT int PvAmdAddGeometry(PvAmdSolver* s, float posX, float posY, float width, float height,
                                float absorption);
PVA_EXPORT int PvAmdUpdateGeometry(PvAmdSolver* s, int id, float posX, float posY, float width, float height,
                                   float absorption);
PVA_EXPORT int PvAmdRemoveGeometry(PvAmdSolver* s, int id);
PVA_EXPORT int PvAmdLoadScene(PvAmdSolver* s, const char* pvPath);
/* Write the current boxes as a .pv file (Editor.cpp:219-243) */
PVA_EXPORT int PvAmdSaveScene(PvAmdSolver* s, const char* pvPath);

/* One iteration of the reference's background loop (PvContext.cpp:80-83): GenerateResponse + AnalyzeResponses
 * for a listener position; synchronous. */
PVA_EXPORT int PvAmdRun(PvAmdSolver* s, float lx, float ly, float lz);
/* Enqueue the same work on the solver's stream without waiting; PvAmdSync waits. */
PVA_EXPORT int PvAmdRunAsync(PvAmdSolver* s, float lx, float ly, float lz);
/* Two solvers taking turns on ONE sequence of iterations (what the live module does for small grids: two iterations in
 * flight): a run of `s` that continues `prev`'s result map -- the cells in which this run finds no onset keep the values the
 * run enqueued LAST on `prev` left there (the reference never touches them, Analyzer.cpp:160-165, and its listener-direction
 * walk reads them), carried over on the device behind prev's analysis.  Same grid, same device; asynchronous like
 * PvAmdRunAsync (PvAmdSync(s) waits for it; prev's run need not have finished when this is called). */
PVA_EXPORT int PvAmdRunAsyncAfter(PvAmdSolver* s, PvAmdSolver* prev, float lx, float ly, float lz);
PVA_EXPORT int PvAmdSync(PvAmdSolver* s);
/* n (1..8) independent runs -- one per solver, listener i at listenersXYZ[3i..3i+2] -- advanced together by ONE
 * kernel launch per K steps instead of n launches on n streams (the reference would make these n iterations of its
 * background loop one after the other, PvContext.cpp:74-93).  The solvers must sit on one device and share grid
 * size, resolution and tile configuration (scenes may differ); streaming analysis and kernel timing are excluded.
 * Afterwards every solver holds its own run's results exactly as after PvAmdRun (same bits).  wait = 0 returns after
 * enqueueing; PvAmdSync each solver before reading results. */
PVA_EXPORT int PvAmdRunBatch(PvAmdSolver* const* solvers, int n, const float* listenersXYZ, int wait);
/* The shader clock (MHz) the device sustains at this moment: one wave sleeps a known number of shader-clock cycles and times
 * them against the constant 100 MHz counter, on a stream of its own (so it can run beside solvers at work).  *byMemtimeMHz
 * (optional) = the same span by s_memtime.  0 on failure.  bench.py's device record. */
PVA_EXPORT float PvAmdClockProbe(int device, float* byMemtimeMHz);
/* The device's own streaming bandwidth, GB/s ("confirm on the box with a device-to-device copy micro-benchmark
 * and report against both"): gbPerS4[0] = device-to-device copy, 16 B per lane (bytes read + written per second), [1] = the
 * same with 4 B per lane, [2] = read only and [3] = write only, 4 B per lane in 256-byte rows per wave (the stencil kernels'
 * pattern).  1 GiB per buffer (2 GiB of device memory while it runs), best of five launches each, ~60 ms, on a stream of its
 * own; call it on an idle device.  0, or -1 on failure.  bench.py's roofline record. */
PVA_EXPORT int PvAmdBandwidthProbe(int device, float* gbPerS4);
PVA_EXPORT int PvAmdGetTimings(PvAmdSolver* s, PvAmdTimings* out);

/* Streaming-analysis (sparse-emitter) mode only -- SURVEY.md 8f N3.  Registers the emitter positions (n x {x,y,z})
 * whose wet gain and RT60 the next runs compute; onset, occlusion, lowpass, source directivity and listener direction
 * are still produced for EVERY cell, wet gain / RT60 only at these cells (0 elsewhere).  This removes the
 * T x cells pressure history, which is what makes T ~ 25 000 (a 25 m scene at 4096^2) possible at all. */
PVA_EXPORT int PvAmdSetEmitters(PvAmdSolver* s, const float* xyz, int n);
/* Analyzer::GetResponseResult + Planeverb::GetOutput (Analyzer.cpp:106-116, FDTD.cpp:16-58) */
PVA_EXPORT int PvAmdGetOutput(PvAmdSolver* s, float ex, float ey, float ez, PlaneverbOutput* out);
/* Output queries: the emitter positions (n <= 64, n x {x,y,z}) whose PlaneverbOutput every FOLLOWING run gathers into
 * pinned host memory with one kernel behind its analysis -- the batch-API counterpart of the reference's registered
 * emitters (Emit / GetOutput(id), EmissionManager.cpp:11-38, FDTD.cpp:16-58).  After PvAmdSync,
 * PvAmdGetQueriedOutputs returns them without any further GPU work or stream synchronisation (PvAmdGetOutput costs a
 * launch and a sync per emitter).  n must equal the registered count; positions outside the grid give the
 * reference's sentinel (occlusion = -1, the rest 0).  Setting queries waits for a run in flight. */
PVA_EXPORT int PvAmdSetOutputQueries(PvAmdSolver* s, const float* xyz, int n);
PVA_EXPORT int PvAmdGetQueriedOutputs(PvAmdSolver* s, PlaneverbOutput* out, int n);
/* Whole result map: res8 = gx*gy*8 floats in AnalyzerResult order (Analyzer.h:13-21), delay = gx*gy */
PVA_EXPORT int PvAmdCopyResults(PvAmdSolver* s, float* res8, float* delay);
/* the same for the block of result cells [r0, r0 + nr) x [c0, c0 + nc): nr x nc records / onsets, row-major (either may be NULL) */
PVA_EXPORT int PvAmdCopyResultsBlock(PvAmdSolver* s, int r0, int c0, int nr, int nc, float* res8, float* delay);
/* Planeverb::GetImpulseResponse (FDTD.cpp:60-70): T x {pr, vx, vy} at array cell (cx, cy) */
PVA_EXPORT int PvAmdGetImpulseResponse(PvAmdSolver* s, int cx, int cy, float* out3T);
/* the same as T reference Cells (pr, vx, vy + the cell's b / by), the layout Planeverb::GetImpulseResponse hands out */
PVA_EXPORT int PvAmdGetImpulseResponseCells(PvAmdSolver* s, int cx, int cy, PlaneverbCell* outT);
/* Final fields of the last run, (gx+1)*(gy+1) each, reference order (x*(gy+1)+y) */
PVA_EXPORT int PvAmdCopyFields(PvAmdSolver* s, float* pr, float* vx, float* vy);
/* Recorded pressure plane of step t (zeros where the history was provably zero and not stored) */
PVA_EXPORT int PvAmdCopyHistoryPlane(PvAmdSolver* s, int t, float* pr);
/* Gaussian pulse table (Grid.cpp:12-27), T floats */
PVA_EXPORT int PvAmdCopyPulse(PvAmdSolver* s, float* out);
/* Material planes after rasterisation: beta (uint8) and R (float), (gx+1)*(gy+1) each */
PVA_EXPORT int PvAmdCopyMaterial(PvAmdSolver* s, uint8_t* beta, float* R);
/* Overwrite the fields the NEXT PvAmdRunSteps starts from (test / benchmark hook; reference order) */
PVA_EXPORT int PvAmdSetFields(PvAmdSolver* s, const float* pr, const float* vx, const float* vy);
/* Advance `nsteps` time steps from the current fields without resetting them, without pulse when
 * withPulse == 0 and without recording history: the raw stencil (used for roofline measurements and for
 * linearity / equivalence property tests). */
PVA_EXPORT int PvAmdRunSteps(PvAmdSolver* s, int nsteps, int withPulse, float lx, float lz);

/* ------------------------------------------------------------------------------------------------------------
 * Part 3 -- independent runs sharded over the GPUs of one node, C++ host side
 * A "run" = one listener position on one scene (one iteration of the reference's loop, PvContext.cpp:74-93).  Runs share
 * nothing: run k belongs to rank k mod world (a rank = one process, normally one GPU); inside a rank the runs go round-
 * robin over the rank's solvers, kept busy through their own HIP streams.  The only exchange is ONE all-gather of the
 * per-emitter records, RCCL (ncclAllGather over xGMI) when the ranks span processes.  RCCL is bound at run time
 * (dlopen): a single-GPU user of this library needs no RCCL.
 * ---------------------------------------------------------------------------------------------------------- */
/* the plan itself (no device needed): fills runIdx[i] / solverIdx[i] for this rank's i-th run, returns their number
 * (<= cap entries written) */
PVA_EXPORT int PvAmdShardPlan(int nRuns, int world, int rank, int nLocalSolvers, int* runIdx, int* solverIdx, int cap);
/* the segment plan of PVA_OPT_STREAM_ROWS (no device needed; test hook): air[ti * nty + tj] != 0 marks the air tiles;
 * fills seg4[4 i .. 4 i + 3] = {first array row, rows, first tile column, tile columns} for up to cap segments, returns
 * their number.  Every air tile is covered by exactly one segment. */
PVA_EXPORT int PvAmdPlanSegments(const unsigned char* air, int ntx, int nty, int tileRows, int maxTileColumns, int target,
                                 int* seg4, int cap);
typedef struct PvAmdComm PvAmdComm;
/* rank 0 creates the 128-byte id (ncclGetUniqueId) and hands it to every rank by whatever bootstrap the host has
 * (a file, MPI, torch.distributed's store); every rank then joins (ncclCommInitRank) with its HIP device */
PVA_EXPORT int PvAmdCommUniqueId(char id128[128]);
PVA_EXPORT PvAmdComm* PvAmdCommCreate(const char id128[128], int rank, int world, int device);
PVA_EXPORT void PvAmdCommDestroy(PvAmdComm* c);
/* every rank contributes countPerRank floats; all = world * countPerRank floats, rank-major, on every rank */
PVA_EXPORT int PvAmdCommAllGather(PvAmdComm* c, const float* mine, int countPerRank, float* all);
/* Simulate nRuns listener positions (listenersXYZ[3k..]) with emittersPerRun emitter positions each
 * (emittersXYZ[(k*emittersPerRun + e)*3 ..]) on this rank's `solvers` (identically configured, scenes loaded by the
 * caller; two per GPU keep two runs in flight) and gather all records: out[k*emittersPerRun + e] on every
 * rank.  world == 1: comm may be NULL.  Returns 0, or -1 with PvAmdLastError. */
PVA_EXPORT int PvAmdRunSharded(PvAmdSolver* const* solvers, int nSolvers, const float* listenersXYZ, int nRuns,
                               const float* emittersXYZ, int emittersPerRun, int rank, int world, PvAmdComm* comm,
                               PlaneverbOutput* out);

/* Host-side pieces of the path that need no device (grid arithmetic, pulse table, rasteriser, .pv parser); the
 * solver uses exactly these internally.  Exposed so that they can be checked without a GPU. */
/* Grid.cpp:390-396,46-55: fills gx, gy, T, fs, res, dx, dt (other fields 0) */
PVA_EXPORT int PvAmdHostGridInfo(float gridSizeX, float gridSizeY, int gridResolution, PvAmdInfo* out);
/* Grid.cpp:12-27: T floats */
PVA_EXPORT int PvAmdHostPulse(float gridSizeX, float gridSizeY, int gridResolution, float* out);
/* 1 if this host's expf reproduces five samples of the reference's 275 Hz pulse table bit for bit (the pulse is made
 * with the host libm, as in the reference: Grid.cpp:12-27), else 0.  The solver checks this once per process by itself
 * and warns on stderr. */
PVA_EXPORT int PvAmdHostPulseSelfCheck(void);
/* Grid::AddAABB / RemoveAABB on a fresh grid: ops[i] = +1 add / -1 remove of boxes5[5*i..]; beta, R are
 * (gx+1)*(gy+1) */
PVA_EXPORT int PvAmdHostRasterize(float gridSizeX, float gridSizeY, int gridResolution, const float* boxes5,
                                  const int* ops, int n, uint8_t* beta, float* R);
/* Editor.cpp:245-281: returns the number of boxes (<= maxBoxes written as 5 floats each) or -1 */
PVA_EXPORT int PvAmdHostLoadPv(const char* pvPath, float* boxes5, int maxBoxes);
/* Editor.cpp:219-243: writes `n` boxes (ids[i] or i when ids == NULL) in the .pv text format; 0 = ok */
PVA_EXPORT int PvAmdHostSavePv(const char* pvPath, const float* boxes5, const int* ids, int n);
/* FDTD.cpp:97-98 listener cell and Analyzer.cpp:106-116 result cell (valid = 0 where GetOutput returns -1) */
PVA_EXPORT int PvAmdHostCells(float gridSizeX, float gridSizeY, int gridResolution, float x, float z, int* listenerCx,
                              int* listenerCy, int* resultCx, int* resultCy, int* resultValid);

/* PlaneverbDSP reverb-bus split of wetGain by rt60 (PlaneverbDSP/src/PvDSPContext.cpp:165-228) */
PVA_EXPORT void PvAmdReverbBusGains(float rt60, float wetGain, float* a, float* b, float* c);

#ifdef __cplusplus
}
#endif
#endif /* PLANEVERB_AMD_H */
