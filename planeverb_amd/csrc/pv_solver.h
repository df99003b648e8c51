// pv_solver.h -- host side of the MI355X solver: owns the HBM planes, the HIP stream and the launch schedule.
//
// One Solver = the reference's Grid + FreeGrid + Analyzer for one config on one GPU
// (ProjectPlaneverb/src/FDTD/Grid.h:25-74, FreeGrid.h:7-24, DSP/Analyzer.h:24-50), without any thread: run()
// enqueues  [geometry deltas] -> reset -> T/K fused step launches -> analysis  on the solver's stream, which is
// what one iteration of Context's background loop does (Context/PvContext.cpp:74-93).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "pv_core.h"
#include "pv_device.h"

namespace pva {

struct SolverOptions {
    int K = 0;            // steps per launch (0 = default)
    int rxi = 0;          // interior rows per tile (0 = default)
    bool denseHistory = false;
    int numSteps = 0;     // override T (0 = reference value)
    bool skipAnalysis = false;
    int useGraph = 0;     // 0 = auto (small grids), 1 = always, 2 = never: replay the run from a captured hipGraph
    bool withFreeGrid = true;
    int tileOrder = -1;   // air-kernel block->tile map: 0 = linear, 1 = XCD band of tile rows, row-major (1-5 % faster than 0),
                          // 2 / >= 4 = compact variants of it (fewer bytes, slower), 3 = XCD strip of tile COLUMNS, row-major:
                          // a tile's vertical halo neighbours are nty/8 tiles away instead of nty, so they are still in the
                          // 4 MiB L2 (HBM reads 316 -> 268 MB per launch at 4096^2, 1400 -> 948 MB at 8192^2; +2.5 % / +4 %
                          // cell-updates/s; -3 % at 2048^2 and below).  -1 = by grid size: 3 for the large-grid tile, else 1
    int merged = 1;       // 1 = general + air tiles in one launch per K steps (where it compiles spill-free), 0 = two kernels on two streams
    int segments = 0;     // N > 0: row-streaming air segments (pv_seg.h) instead of one wave per air tile, about N per
                          // sweep; only the (K, rows) = (8, 40) and (12, 36) configurations have the kernel.  0 = off
    bool streaming = false;  // sparse-emitter mode: ring history + incremental forward analysis
    bool autoStreaming = false;  // switch to the sparse-emitter mode when the T-step pressure history does not fit the
                                 // device (the live module: Planeverb::Init accepts any resolution, PvContext.cpp:101-107)
    bool packed = true;   // packed-f32 arithmetic in the air-tile kernel (VALU-issue bound otherwise)
    bool edgeTiles = false;  // grid-edge tiles of empty regions on the air path + overrides (tile class 2): only the
                             // batched kernel has that arm, so every run of such a solver goes through it
    int smallGrid = 0;    // 0 = auto: grids that fit one CU's LDS run in the whole-grid-resident kernel; 2 = never
    int timeKernels = 0;  // N > 0: HIP events around every Nth step-kernel launch (bench / roofline)
    // Row slab of a larger grid (single-grid domain decomposition, pv_slabs.h): this solver owns the tile rows
    // [ntx * slabIndex / slabCount, ntx * (slabIndex + 1) / slabCount) of the grid, allocates planes for those rows only
    // and keeps the K rows its neighbours own next to them current in its guard band (SlabGroup exchanges them after
    // every launch).  slabCount = 1: a whole grid.
    int slabIndex = 0, slabCount = 1;
    int auxStreams = 0;   // streams created and never used (PVA_OPT_AUX_STREAMS: hardware-queue placement of the solvers' main streams)
    int streamFuse = -1;  // sparse-emitter mode: forward sums of air tiles inside the stencil (pv_stream.h): -1 = by grid size (on from 6000 tiles: it costs two more launches per sweep and pays where the ring traffic binds), 0 = ring + accumulate pass for every tile (round 2's form), 1 = on
    bool lazyFar = true;  // far cells of the result map lazily (see Solver::lazyFar_)
    int patch = -1;       // air tiles by the persistent patch kernel (pv_patch.h): -1 = default of the configuration, 0 off, 1 on
    int patchStrip = 3;   // patch columns per strip of its walk
    int streamPriority = 0;  // PVA_OPT_STREAM_PRIORITY: 1 = main stream on the highest priority (a hardware queue apart from the default-priority streams)
    bool debugLoseFirstCapture = false;  // PVA_OPT_DEBUG_LOSE_FIRST_CAPTURE: the solver's first graph capture counts as lost
    int fusedAnalysis = -1; // 1 = grids whose history window is the whole grid (up to 98 304 cells) run their analysis as ONE launch (pv_fused.hip: EXPERIMENTAL build only, refused by the product build); -1 / 0 (default) = the separate kernels
    int analysisFork = 1; // 0: wet gain / decay time behind the encode pass instead of beside it (measurements)
    int rt60Lanes = 0;    // decay-time pass: lanes per cell (pv_rt60.hip): 0 = by the number of reachable cells, 16 / 4 / 1 = forced
    int resident = 0;     // resident kernel (pv_resident.hip: one launch per run, every tile a workgroup that stays on its CU for
                          // all T steps): 0 = auto (default tile of the launch-bound grids, whole-grid history window, all
                          // blocks co-resident), 1 = also with an explicitly chosen tile, 2 = never
    int xcdRegions = -1;       // tile order 3: 1 = every XCD owns one of 2 x 4 regions of the grid instead of one of 8 strips of tile
                               // columns (half as many boundary lines fetched by two XCDs); -1 = default (on), 0 = strips
    int alternateSweeps = -1;  // tile order 3: odd launches of a run walk the XCD strips backwards (StepArgs::sweepReverse):
                               // -1 = default, 0 = off, 1 = on
    int rowBands = 0;     // B > 1: each sweep = B launches (bands of tile rows, one stream each) with 3-point
                          // dependencies between consecutive sweeps; 0 = auto, 1 = off
};

struct SolverTimings {
    float fdtdMs = 0, analysisMs = 0, geometryMs = 0;
    float stepLoopMs = 0;  // the back-to-back step launches alone (fdtdMs minus the field reset); 0 when replayed from a graph
    float airKernelMs = 0, generalKernelMs = 0;  // mean duration per launch (timeKernels only)
    int airLaunches = 0, generalLaunches = 0;
    int stepLaunches = 0;
    int reachedCells = 0;  // cells with an onset in the last analysed run (whole grids, full-history analysis)
    int activeCells = 0;   // cells of the window's ever-non-zero tiles
    int silentCells = 0;   // air cells among them whose whole history stayed below the audible threshold
};

class SlabGroup;
class SlabRoot;
struct SlabRankOps;

// A stream with a hardware queue apart from every other claimed stream of its device.  The runtime multiplexes a process's streams
// on a small pool of hardware queues (four per priority by default), dealt by the number of streams that already use each; packets
// of two streams on one queue run one after the other, and a waiting packet parks what sits behind it.  claim() probes the stream
// against every claimed, idle stream of the device (streamsShareQueue, pv_probe.hip) and replaces it while it shares -- the
// replaced ones stay alive ("parked") until release(), so that the pool deals the next one elsewhere.
struct QueueClaim {
    std::vector<hipStream_t> parked;
    hipStream_t claimed = nullptr;
    int redeals = 0;
    // The probe of a NEW solver launches on the claimed streams of the live ones, from the creating thread.  `use` serialises it
    // with the stream's owner: the owner holds it while it enqueues on (or captures) its stream, the prober only tries it and
    // leaves a stream whose owner is at work alone (round 5's hipStreamQuery alone left a window in which the owner could begin
    // a graph capture under the sleepers: ADVICE r05).
    std::shared_ptr<std::recursive_mutex> use;
    std::unique_lock<std::recursive_mutex> lockUse() const { return use ? std::unique_lock<std::recursive_mutex>(*use) : std::unique_lock<std::recursive_mutex>(); }  // (recursive: a run that falls back to the replayed graph re-enters enqueueRun from sync)
    enum { kNormal = 0, kHigh = 1, kLow = 2 };  // the runtime keeps a pool of hardware queues per priority
    bool claim(int device, hipStream_t* stream, int priority, const void* owner);
    // another stream of the same priority in place of the claimed one (which stays parked, so that the pool deals elsewhere): a
    // slab group whose hand-off's dry run did not come through on the streams it was dealt (SlabGroup::init)
    bool replace(hipStream_t* stream);
    int priorityClass = kNormal;
    void release();  // before the stream itself is destroyed
};

class Solver {
    friend class SlabGroup;
    friend class SlabRoot;
    friend struct SlabRankOps;

public:
    static Solver* create(const GridSpec& spec, int device, const SolverOptions& opt, std::string* err);
    ~Solver();

    const GridSpec& spec() const { return g_; }
    const Geometry& geometry() const { return geo_; }
    int device() const { return device_; }
    int K() const { return K_; }
    float efree() const { return efree_; }
    void setEfree(float e) { efree_ = e; }  // slab ranks: computed once for the whole grid
    int T() const { return T_; }
    long long deviceBytes() const { return deviceBytes_; }
    int histRows() const { return histRows_; }
    int histPitch() const { return histPitch_; }
    const std::string& lastError() const { return err_; }
    bool streamFuse() const { return streamFuse_; }
    bool residentKernel() const { return useResident_; }  // runs of this solver go through pv_resident_kernel (when the
                                                          // device's resident-block budget allows: else the replayed graph)
    SolverOptions& options() { return opt_; }

    // geometry table with the reference's id recycling (Geometry/GeometryManager.cpp:67-121)
    int addBox(const Box& b);
    bool updateBox(int id, const Box& b);
    bool removeBox(int id);
    int numBoxes() const;
    std::vector<std::pair<int, Box>> boxes() const;
    // raw rasteriser access (Grid::AddAABB / RemoveAABB), used by the live context's change queue
    void rasterAdd(const Box& b) { mat_.add(b); }
    void rasterRemove(const Box& b) { mat_.remove(b); }

    // carryFrom (live module, two iterations in flight on two solvers): the solver that ran the PREVIOUS iteration.  The cells
    // in which this run finds no onset take their occlusion / wet gain / decay time / lowpass / source direction from that
    // solver's maps, on the device, behind that solver's analysis and in front of this run's listener-direction pass -- what one
    // solver's persistent result map does by itself (Analyzer.cpp:160-165 leaves m_results untouched there, and the direction
    // walk reads those values, :340-431)
    bool run(float lx, float ly, float lz, bool wait, Solver* carryFrom = nullptr);
    bool runCells(int lcx, int lcy, float lx, float lz, bool wait);
    bool sync();
    // n <= 8 identically configured solvers of one device: n independent runs (listeners lxyz[3n]) advanced by one
    // launch per K steps (pv_step_batch_kernel); each solver then holds its run's results as after run()
    static bool runBatch(Solver* const* s, int n, const float* lxyz, bool wait, std::string* err);
    bool runSteps(int nsteps, bool withPulse, float lx, float lz);
    const SolverTimings& timings() const { return tim_; }

    bool getOutput(float ex, float ey, float ez, float out8[8], bool* valid);
    // output queries: up to kMaxQueries emitter positions whose 8 outputs every following run leaves in pinned host
    // memory (one gather kernel behind the analysis): after sync() they are read without any further GPU work
    static constexpr int kMaxQueries = 64;
    bool setOutputQueries(const float* xyz, int n);
    bool queriedOutputs(float* out8n, unsigned char* valid, int n);
    bool copyResults(float* res8, float* delay);
    // the block [r0, r0 + nr) x [c0, c0 + nc) of the result map (AoS records) and of the onset map, row-major nr x nc
    bool copyResultsBlock(int r0, int c0, int nr, int nc, float* res8, float* delay);
    // device -> caller-provided (pinned) host buffers, asynchronously on the solver's stream
    bool copyResultsAsync(float* res8Host);
    // The block of the result map the LAST run could have changed: the history window, clipped to the map (everything
    // outside it keeps its earlier values, except the listener direction, which is the unit vector from the listener
    // to the cell: Analyzer.cpp:64-68,365-391,415-428).  publishWindowAsync packs that block as nr*nc AoS records
    // and copies it to hostDst (pinned, >= windowCapacity() records) on the solver's stream.
    struct WindowBlock {
        int r0 = 0, c0 = 0, nr = 0, nc = 0;  // result-map cells [r0, r0+nr) x [c0, c0+nc)
        float lx = 0, lz = 0;                // the run's listener position, metres
    };
    size_t windowCapacity() const;  // records: upper bound of nr*nc for any listener position
    // overlap = true: only the pack runs on the solver's stream; the device -> host copy goes to a second stream, so that the
    // NEXT run's launches need not wait for it (the live module's loop); waitPublish() then waits for that copy alone.
    bool publishWindowAsync(float* hostDst, WindowBlock* info, bool overlap = false);
    bool waitPublish();
    // pinned host memory for the buffers above
    static void* hostAlloc(size_t bytes);
    static void hostFree(void* p);
    bool impulseResponse(int cx, int cy, float* out3T);
    // the same as T reference Cells {f32 pr, vx, vy; i16 b, by} (PvTypes.h:106-121; recorded at FDTD.cpp:226-230 with
    // the b / by the cell had during the run)
    bool impulseResponseCells(int cx, int cy, void* out16T);
    bool copyFields(float* pr, float* vx, float* vy);
    bool setFields(const float* pr, const float* vx, const float* vy);
    bool copyHistoryPlane(int t, float* pr);
    bool copyPulse(float* out);
    bool copyMaterial(uint8_t* beta, float* R);
    bool freeFieldEnergyAt(int cellX, int cellY, int n, float r, float* out);
    // streaming mode: the cells whose wet gain / RT60 are wanted (world positions -> result cells)
    bool setEmitters(const float* xyz, int n);

private:
    Solver() = default;
    bool init(const GridSpec& spec, int device, const SolverOptions& opt);
    bool applyGeometry();
    bool computeEfree();
    bool enqueueRun(int lcx, int lcy, float lx, float lz);
    bool enqueueSteps(int firstStep, int nsteps, bool withPulse, bool record, bool fromZero = false);
    StepArgs baseStepArgs(bool withPulse, bool record) const;
    void setLaunchArgs(StepArgs& a, int t0, int k, bool firstOfRun, int li) const;
    bool prepareDyn(int lcx, int lcy, bool withPulse, bool banded);
    void enqueueBeginRun(bool resetTiles);
    bool zeroPlanesIfNeeded();
    AnalyzeArgs analyzeArgs(float lx, float lz) const;
    // Far cells lazily: a run resets only the previous and the current window block; the far cells'
    // listener direction is materialised for whole-map read-backs and computed in closed form by the output gathers
    bool lazyFar_ = false;
    struct Block {
        int r0 = 0, c0 = 0, nr = 0, nc = 0;
    };
    Block curWindow() const;        // the history-window block of the result map of the run prepared last
    Block farWin_;                  // window block of the last ANALYSED run (the far cells lie outside it)
    bool farDirValid_ = true;       // the direction planes hold the far cells' directions of that run
    // Near box (AnalyzeArgs::box): two sets of four device words, the reached cells' bounding box of the run analysed last
    // (nearBox_ + 4 nearPar_) and of the one before it; PLANEVERB_AMD_NEAR_BOX=0 keeps the window-wide passes
    int* nearBox_ = nullptr;
    int nearPar_ = 0;
    bool useNearBox_ = false;
    bool nearBoxValid_ = false;     // the last analysed run went through the near-box passes
    // Timings of a resident-kernel run without event packets between its kernels (an event record on the stream is a barrier
    // packet: ~7 us of bubble in front of the next kernel, three of them per 0.3 ms run): the run's first kernel, the analysis'
    // first kernel and the run's last kernel write the 100 MHz counter into three pinned words, sync() takes the differences
    unsigned long long* stampsHost_ = nullptr;
    bool stampTimed_ = false;       // the run in flight is timed that way
    bool labelsValid_ = false;      // labelDev_ holds the air components of the current material plane
    int lastReached_ = -1;          // cells with an onset in the last run read back (-1: none yet): AnalyzeArgs::rt60Tile
    FarInfo farInfo() const;
    bool ensureFarDirections();
    void enqueueAnalysis(float lx, float lz);
    Solver* carryFrom_ = nullptr;  // for the run being enqueued
    bool fail(const std::string& what);
    bool hipOk(hipError_t e, const char* what);
    template <typename Tp>
    bool dalloc(Tp** p, size_t count, bool zero);

    GridSpec g_;         // always the WHOLE grid
    // what this solver owns of it (a whole grid: x0_ = 0, lNX_ = g_.NX, lgx_ = g_.gx)
    int x0_ = 0;         // first cell-array row
    int tileRow0_ = 0;   // = x0_ / rxi_
    int lNX_ = 0;        // cell-array rows owned (the last slab also owns the ghost row)
    int lgx_ = 0;        // result-map rows owned
    int ntxG_ = 0, histTilesXG_ = 0;  // the whole grid's tile rows / history-window tile rows
    bool isSlab() const { return opt_.slabCount > 1; }
    int globalWindowTileRow0(int lcx) const;
    float* histAbove_ = nullptr;  // [T][histPitch]: pressure history of the row above this slab (from the neighbour)
    float* histEdge_ = nullptr;   // [T][histPitch]: this slab's last row, for the neighbour below
    Geometry geo_{};
    SolverOptions opt_;
    int device_ = 0;
    int K_ = 4, rxi_ = 32, wi_ = 56;
    int T_ = 0;
    hipStream_t stream_ = nullptr;
    hipStream_t stream2_ = nullptr;        // general-tile kernels run here, concurrently with the air kernel
    bool lastRunBatched_ = false;          // the last run was a member of a batch of several
    std::vector<hipStream_t> auxStreams_;  // PVA_OPT_AUX_STREAMS
    std::vector<hipEvent_t> airDone_, genDone_;  // per-launch cross-stream dependencies (no timing)
    int stepWhich_ = 4;                    // launchStep's `which` of the merged launch (applyGeometry: | kStepGeneralPacked)
    QueueClaim queue_;                     // stream_'s hardware queue, apart from the other solvers' (Solver::init)
    bool redealMainStream() { return queue_.replace(&stream_); }  // (before any run: nothing of the old stream is in flight)
    hipEvent_t forkEv_ = nullptr;
    hipEvent_t anaEv_[2] = {nullptr, nullptr};  // enqueueAnalysis: onsets known -> stream2_, decay times done -> stream_
    // captured launch schedule of one run (reset + all step launches on both streams), replayed per run
    hipGraph_t graph_ = nullptr;
    hipGraphExec_t graphExec_ = nullptr;
    int graphCap_ = -1;  // general-list capacity the graph was captured with
    bool captureLossInjected_ = false;  // (test hook, buildGraph)
    bool buildGraph(int cap);
    void dropGraph();
    bool enqueueResetAndSteps();
    hipEvent_t ev_[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pubEv_[2] = {nullptr, nullptr};  // window block packed / copied to the host (publishWindowAsync, overlap form)
    bool pubCopyPending_ = false;
    long long deviceBytes_ = 0;
    std::string err_;

    // HBM
    float* pr_[2] = {nullptr, nullptr};
    float* vx_[2] = {nullptr, nullptr};
    float* vy_[2] = {nullptr, nullptr};
    int cur_ = 0;  // which set holds the current fields
    FaceCoef* coef_ = nullptr;      // face coefficients + beta of every padded cell (pv_device.h)
    float* matDev_ = nullptr;       // material plane: NaN = air cell, else the wall cell's admittance Y
    float* pulseDev_ = nullptr;
    float* hist_ = nullptr;
    long long histPlane_ = 0;
    int histRows_ = 0, histPitch_ = 0, histTilesX_ = 0, histTilesY_ = 0;
    uint8_t* nz_[2] = {nullptr, nullptr};  // per-tile non-zero flags, ping-pong per launch
    int* tileFirst_ = nullptr;
    uint8_t* tileClass_ = nullptr;
    uint8_t* tileDead_ = nullptr;   // per tile: all-wall interior (skipped by runs that start from zero fields)
    int* deadCount_ = nullptr;
    int numDead_ = 0;
    bool planesDirty_ = false;      // the field planes may hold non-zero values inside dead tiles
    int* generalList_ = nullptr;
    int* generalCount_ = nullptr;
    DynParams* dynDev_ = nullptr;
    int* errFlag_ = nullptr;
    int residentBudget_ = 0;        // blocks of the resident kernel this device holds at once x 3/4 (init)
    int* labelDev_ = nullptr;       // AnalyzeArgs::labels (grids of up to kLabelMaxCells array cells), re-made by applyGeometry
    std::vector<int> labelHost_;
    bool makeLabels();
    unsigned* fusedCtl_ = nullptr;  // FusedArgs::ctl
    bool useFused_ = false;
    int* unitList_ = nullptr;     // AnalyzeArgs::unitList: histPlane / 64 + 1 ints
    int* activeCount_ = nullptr;  // cells with an onset in the last analysis (chooses the RT60 kernel on the device)
    float* res_ = nullptr;   // 8 SoA result planes (see AnalyzeArgs::res)
    float* res8_ = nullptr;  // AoS copy for whole-map read-backs, allocated and packed on demand
    float* win8_ = nullptr;  // AoS staging of the window block (publishWindowAsync), allocated on demand
    bool packResults();
    float* delay_ = nullptr;
    // streaming analysis state
    int ring_ = 0;             // history planes allocated (T_ when not streaming)
    int halfRing_ = 0;         // streaming: steps per accumulate pass (the ring holds two such halves)
    hipEvent_t streamEv_[4] = {nullptr, nullptr, nullptr, nullptr};  // stepDone[2], accDone[2]
    hipStream_t openStream_ = nullptr;  // sparse-emitter mode: the open half tiles of a sweep, beside the merged launch
    hipEvent_t openEv_[4] = {nullptr, nullptr, nullptr, nullptr};  // classified[2], openDone[2] (alternating sweeps)
    bool openPending_ = false;
    void joinOpen(int li);
    int* sOnset_ = nullptr;
    float* sState_[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // Edry, fluxX, fluxY, vx, vy
    uint8_t* tileOpen_ = nullptr;   // per tile: history still wanted (streaming mode)
    uint8_t* tileMarks_ = nullptr;  // scratch of one accumulate pass
    uint8_t* tileEmit_ = nullptr;   // per tile: holds a registered emitter
    // forward sums inside the stencil (pv_stream.h)
    bool streamFuse_ = false;
    uint8_t* classStream_ = nullptr;   // per-launch tile classes (open air tiles masked out of the merged launch)
    uint8_t* ringOpen_ = nullptr;      // per-launch `tileOpen` of the step kernels: open RING tiles only
    uint8_t* cellsOpen2_ = nullptr;    // per tile half: a dry window is still open
    int* openList_ = nullptr;
    int* idleHost_ = nullptr;          // pinned, 2 ints (one per ring half): every fused tile has closed for good
    bool fuseIdle_ = false;            // ... as the host has seen it: the rest of the run is plain merged launches
    ClassifyArgs classifyArgs(const StepArgs& a, int li, bool withPulse) const;
    int* ringList_ = nullptr;          // the tiles the accumulate pass still serves (general list + emitter tiles), per run
    int* ringHost_ = nullptr;          // pinned staging of it
    int numRing_ = 0;
    std::vector<uint8_t> emTilesHost_; // host copy of tileEmit_
    int* openCount_ = nullptr;         // two counters, alternating from launch to launch
    int* emCells_ = nullptr;
    float* emTrace_ = nullptr;
    int numEmitters_ = 0, emCap_ = 0;
    // resident kernel (pv_resident.hip)
    bool useResident_ = false;
    unsigned* resFlags_ = nullptr;     // ntiles epoch counters + 1 abort word
    int residentHeld_ = 0;             // blocks of the device's resident budget held by the run in flight
    // one-XCD mode of the resident kernel (grids of up to kResidentXcdMaxTiles tiles): hand-off through one XCD's L2
    bool xcdOk_ = true;                // false once a launch found fewer blocks on its XCD than tiles (errFlag 4)
    int xcdTarget_ = 0;                // this solver's XCD (solvers take turns, so that pipelined solvers do not share one)
    int xcdHeld_ = 0;                  // blocks of that XCD's budget held by the run in flight
    bool lastRunXcd_ = false;          // the run in flight went out in the one-XCD mode
    int lastLcx_ = 0, lastLcy_ = 0;    // (a run given up by the claim check is repeated in the placement-independent mode)
    void releaseResident();
    float* scratch_ = nullptr;  // max(3T, NX*NY) floats
    size_t scratchCount_ = 0;

    // pinned host staging
    DynParams* dynHost_ = nullptr;
    float* outHost_ = nullptr;  // 8 floats the output-gather kernel writes straight into host memory
    int* statusHost_ = nullptr; // the words the last kernel of a run leaves here: error flag, two cell counts, resident claims, silent cells
    bool statusQueued_ = false; // ... for the run in flight (else sync() copies them back)
    void enqueueRunStatus();
    long long* qCellsHost_ = nullptr;  // kMaxQueries result-cell indices (-1 = outside the map), device-visible
    float* qOutHost_ = nullptr;        // kMaxQueries x 8 floats
    int numQueries_ = 0;
    void enqueueQueries();
    int* listHost_ = nullptr;
    int listCap_ = 0;
    // row-streaming air segments (pv_seg.h): rebuilt for every run (they avoid the tiles around the listener)
    bool usePatch_ = false;    // air tiles go through the persistent patch kernel (pv_patch.h)
    int patchBlocks_ = 0;      // its grid: one workgroup per CU, a multiple of 8
    long long* patchTrace_ = nullptr;  // development aid (PV_PATCH_TRACE)
    bool useSeg_ = false;      // this solver's configuration and options allow them
    bool segActive_ = false;   // the run being enqueued uses them
    int segWMax_ = 0;          // tile columns a segment can span
    SegDesc* segHost_ = nullptr;  // pinned
    SegDesc* segList_ = nullptr;  // device
    int segCap_ = 0, numSeg_ = 0;
    void buildSegments(int listed);

    // host state
    MaterialPlane mat_;
    std::vector<float> matHost_;     // host copy of matDev_
    std::vector<uint8_t> betaHost_;  // Cell::b as of the last applyGeometry() (= during the last run)
    std::vector<uint8_t> byHost_;  // Cell::by as of the last applyGeometry() (= during the last run)
    std::vector<float> pulse_;
    std::vector<int> wallTiles_;
    std::vector<uint8_t> tileClassHost_;
    int numGeneral_ = 0;
    int launchCap_ = 0;  // general-list entries the general kernel's grid is sized for
    bool geometryDirty_ = true;
    float efree_ = 0.f;
    DynParams dynCur_{};
    bool dynValid_ = false;
    float lastLx_ = 0, lastLz_ = 0;

    // geometry table (GeometryManager.cpp)
    std::vector<Box> boxTable_;
    std::vector<uint8_t> boxUsed_;
    std::vector<int> boxFree_;

    // row bands: band b covers tile rows [bandRow_[b], bandRow_[b+1])
    int nb_ = 1;
    std::vector<int> bandRow_;
    std::vector<hipStream_t> bandStream_;  // [0] = stream_
    std::vector<hipEvent_t> bandEv_;       // 2 per band: sweep parity
    std::vector<int> bandListOff_, bandListCount_;
    DynParams* dynBandsDev_ = nullptr;
    DynParams* dynBandsHost_ = nullptr;
    bool bandedRun_ = false;  // the run prepared last is launched band by band
    bool bandsActive() const;
    StepArgs bandStepArgs(const StepArgs& a, int b) const;

    std::vector<hipEvent_t> kev_;  // 3 events per step launch when opt_.timeKernels
    int kevUsed_ = 0;
    SolverTimings tim_;
    bool pendingTimings_ = false;
    bool loopTimed_ = false;  // ev_[3] was recorded between the reset and the first step launch of this run
};

}  // namespace pva
