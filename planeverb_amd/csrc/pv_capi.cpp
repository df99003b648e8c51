// pv_capi.cpp -- extern "C" boundary of libplaneverb_amd.so (declared in include/planeverb_amd.h).
// No C++ exception leaves this file; the reference's sentinels are kept (-1 ids, occlusion = -1).
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>

#include "../../include/planeverb_amd.h"
#include "pv_context.h"
#include "pv_core.h"
#ifndef PVA_HOST_TEST  // (tests/host/: HIP-less sanitizer build of the live module against a fake Solver)
#include "pv_shard.h"
#include "pv_slabs.h"
#include "pv_launch.h"
#include "pv_solver.h"
#endif

using namespace pva;

static thread_local std::string g_lastError;

// ---------------------------------------------------------------------------------------------------------------
// The exception barrier (the reference throws from Init, PvContext.cpp:106,123, Grid.cpp:69; a C-ABI must
// not).  EVERY extern "C" body below is a function-try-block closed by one of these macros: whatever is thrown underneath
// (std::bad_alloc from a table growing, std::system_error from a thread or a mutex, anything from the HIP runtime's C++
// side) ends here, the call returns its failure sentinel and PvAmdLastError() names the function and the exception.
// tests/host/alloc_fault.cpp drives every allocation of the Part 1 calls into failure, one at a time.
// ---------------------------------------------------------------------------------------------------------------
static void noteException(const char* fn) noexcept {
    try {
        std::string what = "unknown exception";
        try {
            throw;
        } catch (const std::bad_alloc&) {
            what = "out of memory (std::bad_alloc)";
        } catch (const std::exception& e) {
            what = e.what();
        } catch (...) {
        }
        g_lastError = std::string("exception in ") + fn + ": " + what;
    } catch (...) {  // (not even the message could be built)
        g_lastError.clear();
    }
}
static PlaneverbOutput invalidOutput() noexcept {
    PlaneverbOutput o;
    std::memset(&o, 0, sizeof(o));
    o.occlusion = kInvalidDryGain;  // FDTD.cpp:22-26
    return o;
}
#define PV_API_CATCH(sentinel) catch (...) { noteException(__func__); return sentinel; }
#define PV_API_CATCH_VOID      catch (...) { noteException(__func__); }
#define PV_API_CATCH_OUTPUT    catch (...) { noteException(__func__); return invalidOutput(); }

#ifndef PVA_HOST_TEST
struct PvAmdSolver {
    Solver* s = nullptr;
    SlabGroup* g = nullptr;     // a slab group instead of one solver (PvAmdCreateSlabs)
    std::vector<int> slabDevices;
    SolverOptions opt;
    GridSpec spec;
    int device = 0;
    ~PvAmdSolver() {
        delete s;
        delete g;
    }
};

// creates the solver / slab group on first use (options come first); `slabsOk` = the call is implemented for groups
static bool ensure(PvAmdSolver* h, bool slabsOk = false) {
    if (!h) {
        g_lastError = "null solver handle";
        return false;
    }
    if (!h->slabDevices.empty()) {
        if (!slabsOk) {
            g_lastError = "not available for a slab group (PvAmdCreateSlabs)";
            return false;
        }
        if (!h->g) h->g = SlabGroup::create(h->spec, h->slabDevices, h->opt, &g_lastError);
        return h->g != nullptr;
    }
    if (h->s) return true;
    h->s = Solver::create(h->spec, h->device, h->opt, &g_lastError);
    return h->s != nullptr;
}

// A handle from PvAmdCreateSlabRank holds ONE slab of the grid (no FreeGrid, no halo exchange of its own): the whole-grid
// entry points (run / outputs / result maps) would silently work on a partial grid, so they refuse it.
static bool wholeGrid(PvAmdSolver* h) {
    if (h && h->opt.slabCount > 1) {
        g_lastError = "slab rank handle: use the PvAmdSlab* / PvAmdSlabRoot* primitives (PvAmdCreateSlabRank)";
        return false;
    }
    return true;
}

static int ret(PvAmdSolver* h, bool ok) {
    if (ok) return 0;
    if (h && h->s && !h->s->lastError().empty()) g_lastError = h->s->lastError();
    if (h && h->g && !h->g->lastError().empty()) g_lastError = h->g->lastError();
    return -1;
}
#endif  // !PVA_HOST_TEST

extern "C" {

// ---------------------------------------------------------------------------------------------------------------
// Part 1: reference C-ABI
// ---------------------------------------------------------------------------------------------------------------

void UnityPluginLoad(void*) try {} PV_API_CATCH_VOID
void UnityPluginUnload(void) try {} PV_API_CATCH_VOID

void PlaneverbInit(float gridSizeX, float gridSizeY, int gridResolution, int gridBoundaryType, char* tempFileDir,
                   int maxThreadUsage, int threadExecutionType) try {
    LiveConfig c;
    c.sizeX = gridSizeX;
    c.sizeY = gridSizeY;
    c.res = gridResolution;
    c.boundaryType = gridBoundaryType;
    c.tempDir = tempFileDir;
    c.maxThreads = maxThreadUsage;
    c.executionType = threadExecutionType;  // 0 (pv_CPU) and 1 (pv_GPU) both run on the HIP device here
    std::string err;
    if (!Context::init(c, &err)) {
        g_lastError = err;
        std::fprintf(stderr, "[planeverb_amd] PlaneverbInit failed: %s\n", err.c_str());
    }
} PV_API_CATCH_VOID

void PlaneverbExit(void) try {
    Context::exit();
} PV_API_CATCH_VOID

int PlaneverbEmit(float x, float y, float z) try {
    Context::Ref c;
    return c ? c->emit(x, y, z) : -1;
} PV_API_CATCH(-1)

void PlaneverbUpdateEmission(int id, float x, float y, float z) try {
    Context::Ref c;
    if (c) c->updateEmission(id, x, y, z);
} PV_API_CATCH_VOID

void PlaneverbEndEmission(int id) try {
    Context::Ref c;
    if (c) c->endEmission(id);
} PV_API_CATCH_VOID

PlaneverbOutput PlaneverbGetOutput(int emissionID) try {
    PlaneverbOutput o;
    std::memset(&o, 0, sizeof(o));
    Context::Ref c;
    if (!c) {  // FDTD.cpp:22-26
        o.occlusion = kInvalidDryGain;
        return o;
    }
    const Out8 r = c->getOutput(emissionID);
    std::memcpy(&o, r.v, sizeof(o));
    return o;
} PV_API_CATCH_OUTPUT

int PlaneverbAddGeometry(float posX, float posY, float width, float height, float absorption) try {
    Context::Ref c;
    return c ? c->addGeometry(Box{posX, posY, width, height, absorption}) : -1;
} PV_API_CATCH(-1)

void PlaneverbUpdateGeometry(int id, float posX, float posY, float width, float height, float absorption) try {
    Context::Ref c;
    if (c) c->updateGeometry(id, Box{posX, posY, width, height, absorption});
} PV_API_CATCH_VOID

void PlaneverbRemoveGeometry(int id) try {
    Context::Ref c;
    if (c) c->removeGeometry(id);
} PV_API_CATCH_VOID

void PlaneverbSetListenerPosition(float x, float y, float z) try {
    Context::Ref c;
    if (c) c->setListener(x, y, z);
} PV_API_CATCH_VOID

int PlaneverbLoadScene(const char* pvPath) try {
    Context::Ref c;
    if (!c || !pvPath) return -1;
    std::vector<Box> boxes;
    if (!loadPv(pvPath, &boxes, &g_lastError)) return -1;
    for (const Box& b : boxes) c->addGeometry(b);
    return (int)boxes.size();
} PV_API_CATCH(-1)

long long PlaneverbIterationCount(void) try {
    Context::Ref c;
    return c ? c->iterations() : 0;
} PV_API_CATCH(0)

long long PlaneverbWaitIterations(long long count, int timeoutMs) try {
    Context::Ref c;
    return c ? c->waitIterations(count, timeoutMs) : 0;
} PV_API_CATCH(0)

// 0 also when the simulation worker has stopped on an error (PvAmdLastError then says why)
int PlaneverbIsRunning(void) try {
    Context::Ref c;
    return (c && !c->failed()) ? 1 : 0;
} PV_API_CATCH(0)

int PlaneverbIsStreaming(void) try {
    Context::Ref c;
    return (c && c->streaming()) ? 1 : 0;
} PV_API_CATCH(0)

int PlaneverbGetImpulseResponse(float x, float y, float z, PlaneverbCell* out, int capacity) try {
    Context::Ref c;
    if (!c || capacity < 0) return -1;
    const int n = c->impulseResponse(x, y, z, out, capacity);
    if (n < 0) g_lastError = "impulse response not available (no completed iteration yet, or solver error)";
    return n;
} PV_API_CATCH(-1)

// ---------------------------------------------------------------------------------------------------------------
// Part 2: batch solver handle
// ---------------------------------------------------------------------------------------------------------------

const char* PvAmdLastError(void) try {
    // A live module whose worker died reports that -- ONCE per failed context and thread, so that the errors of later,
    // unrelated calls on this thread (batch solver, slabs, communicator) stay readable.  PlaneverbWorkerError() always
    // has the worker's reason.
    {
        static thread_local unsigned long long reported = 0;  // (a generation number: addresses are re-used)
        Context::Ref c;
        if (c && c->failed() && reported != c->generation()) {
            reported = c->generation();
            g_lastError = "simulation worker stopped: " + c->workerError();
        }
    }
    return g_lastError.c_str();
} PV_API_CATCH("")
const char* PlaneverbWorkerError(void) try {
    static thread_local std::string w;
    Context::Ref c;
    w = (c && c->failed()) ? c->workerError() : std::string();
    return w.c_str();
} PV_API_CATCH("")
const char* PvAmdVersion(void) try { return "planeverb_amd 0.2 (gfx950)"; } PV_API_CATCH("")

#ifndef PVA_HOST_TEST
int PvAmdDeviceCount(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
} PV_API_CATCH(0)

PvAmdSolver* PvAmdCreate(float gridSizeX, float gridSizeY, int gridResolution, int device) try {
    if (gridResolution < kLowResolution || gridSizeX == 0.f || gridSizeY == 0.f) {
        g_lastError = "invalid config (pv_InvalidConfig)";  // PvContext.cpp:101-107
        return nullptr;
    }
    int n = PvAmdDeviceCount();
    if (n <= 0) {
        g_lastError = "no HIP device: libplaneverb_amd has no CPU path";
        return nullptr;
    }
    if (device < 0 || device >= n) {
        g_lastError = "HIP device index out of range";
        return nullptr;
    }
    std::unique_ptr<PvAmdSolver> h(new PvAmdSolver());
    h->spec = makeGridSpec(gridSizeX, gridSizeY, gridResolution);
    h->device = device;
    if (h->spec.gx != h->spec.gy) {
        static bool warned = false;
        if (!warned)
            std::fprintf(stderr, "[planeverb_amd] warning: %d x %d grid -- the reference indexes non-square grids "
                                 "inconsistently (SURVEY.md Q1); results there are defined by this library (cell array stride "
                                 "gy+1 throughout), not by the reference\n", h->spec.gx, h->spec.gy);
        warned = true;
    }
    return h.release();
} PV_API_CATCH(nullptr)

PvAmdSolver* PlaneverbCreateGrid(float gridSizeX, float gridSizeY, int gridResolution, int device) try {
    return PvAmdCreate(gridSizeX, gridSizeY, gridResolution, device);
} PV_API_CATCH(nullptr)

PvAmdSolver* PvAmdCreateSlabs(float gridSizeX, float gridSizeY, int gridResolution, const int* devices, int nslabs) try {
    if (!devices || nslabs < 2 || nslabs > 16) {
        g_lastError = "PvAmdCreateSlabs: 2..16 slabs and their devices";
        return nullptr;
    }
    std::unique_ptr<PvAmdSolver> h(PvAmdCreate(gridSizeX, gridSizeY, gridResolution, devices[0]));
    if (!h) return nullptr;
    const int n = PvAmdDeviceCount();
    for (int i = 0; i < nslabs; ++i) {
        if (devices[i] < 0 || devices[i] >= n) {
            g_lastError = "HIP device index out of range";
            return nullptr;
        }
        h->slabDevices.push_back(devices[i]);
    }
    return h.release();
} PV_API_CATCH(nullptr)

int PvAmdGetSlabInfo(PvAmdSolver* h, PvAmdSlabInfo* out) try {
    if (!out || !ensure(h, true) || !h->g) return -1;
    std::memset(out, 0, sizeof(*out));
    out->nslabs = h->g->numSlabs();
    for (int i = 0; i < out->nslabs; ++i) {
        out->row0[i] = h->g->slabRow0(i);
        out->rows[i] = h->g->slabRows(i);
        out->device[i] = h->g->slab(i)->device();
        out->deviceBytes[i] = h->g->slab(i)->deviceBytes();
    }
    out->haloBytesPerLaunch = h->g->haloBytesPerLaunch();
    out->exchangeBytesPerRun = h->g->exchangeBytesPerRun();
    out->handoffWords = h->g->handoffWords() ? 1 : 0;
    out->streamRedeals = h->g->streamRedeals();
    out->dryRunUsPerSweep = h->g->dryRunUsPerSweep();
    return 0;
} PV_API_CATCH(-1)

PvAmdSolver* PvAmdCreateSlabRank(float gridSizeX, float gridSizeY, int gridResolution, int device, int slabIndex,
                                 int slabCount) try {
    if (slabCount < 2 || slabIndex < 0 || slabIndex >= slabCount) {
        g_lastError = "PvAmdCreateSlabRank: 0 <= slabIndex < slabCount, slabCount >= 2";
        return nullptr;
    }
    PvAmdSolver* h = PvAmdCreate(gridSizeX, gridSizeY, gridResolution, device);
    if (!h) return nullptr;
    h->opt.slabIndex = slabIndex;
    h->opt.slabCount = slabCount;
    return h;
} PV_API_CATCH(nullptr)

int PvAmdComputeEfree(float gridSizeX, float gridSizeY, int gridResolution, int device, float* efree) try {
    if (!efree) return -1;
    std::unique_ptr<PvAmdSolver> h(PvAmdCreate(gridSizeX, gridSizeY, gridResolution, device));
    if (!h) return -1;
    // a throw-away solver of a tiny grid would have another centre cell: the free-field run depends on the grid size
    // (FreeGrid.cpp:78-84), so the real config is used; its planes are what the windowed FreeGrid run needs anyway
    const bool ok = ensure(h.get());
    if (ok) *efree = h->s->efree();
    return ok ? 0 : -1;
} PV_API_CATCH(-1)

static Solver* slabOf(PvAmdSolver* h) {
    if (!ensure(h)) return nullptr;
    if (h->opt.slabCount < 2) {
        g_lastError = "not a slab rank handle (PvAmdCreateSlabRank)";
        return nullptr;
    }
    return h->s;
}

int PvAmdSlabSetEfree(PvAmdSolver* h, float efree) try {
    Solver* s = slabOf(h);
    if (!s) return -1;
    s->setEfree(efree);
    return 0;
} PV_API_CATCH(-1)
int PvAmdSlabBegin(PvAmdSolver* h, float lx, float ly, float lz) try {
    Solver* s = slabOf(h);
    return s ? ret(h, SlabRankOps::begin(*s, lx, ly, lz)) : -1;
} PV_API_CATCH(-1)
int PvAmdSlabNumLaunches(PvAmdSolver* h) try {
    Solver* s = slabOf(h);
    return s ? SlabRankOps::numLaunches(*s) : -1;
} PV_API_CATCH(-1)
int PvAmdSlabLaunch(PvAmdSolver* h, int li) try {
    Solver* s = slabOf(h);
    return s ? ret(h, SlabRankOps::launch(*s, li)) : -1;
} PV_API_CATCH(-1)
int PvAmdSlabHaloFloats(PvAmdSolver* h) try {
    Solver* s = slabOf(h);
    return s ? SlabRankOps::haloFloats(*s) : -1;
} PV_API_CATCH(-1)
int PvAmdSlabExportHalo(PvAmdSolver* h, int side, float* host) try {
    Solver* s = slabOf(h);
    return (s && host) ? ret(h, SlabRankOps::exportHalo(*s, side, host)) : -1;
} PV_API_CATCH(-1)
int PvAmdSlabImportHalo(PvAmdSolver* h, int side, const float* host) try {
    Solver* s = slabOf(h);
    return (s && host) ? ret(h, SlabRankOps::importHalo(*s, side, host)) : -1;
} PV_API_CATCH(-1)
int PvAmdSlabHistoryFloats(PvAmdSolver* h) try {
    Solver* s = slabOf(h);
    return s ? SlabRankOps::historyFloats(*s) : -1;
} PV_API_CATCH(-1)
int PvAmdSlabExportEdgeHistory(PvAmdSolver* h, float* host) try {
    Solver* s = slabOf(h);
    return (s && host) ? ret(h, SlabRankOps::exportEdgeHistory(*s, host)) : -1;
} PV_API_CATCH(-1)
int PvAmdSlabImportAboveHistory(PvAmdSolver* h, const float* host) try {
    Solver* s = slabOf(h);
    return (s && host) ? ret(h, SlabRankOps::importAboveHistory(*s, host)) : -1;
} PV_API_CATCH(-1)
int PvAmdSlabAnalyze(PvAmdSolver* h) try {
    Solver* s = slabOf(h);
    return s ? ret(h, SlabRankOps::analyze(*s)) : -1;
} PV_API_CATCH(-1)
long long PvAmdSlabWindowBlock(PvAmdSolver* h, int* info4, float* host, long long cap) try {
    Solver* s = slabOf(h);
    if (!s || !info4) return -1;
    const long long n = SlabRankOps::windowBlock(*s, &info4[0], &info4[1], &info4[2], &info4[3], host, cap);
    if (n < 0) ret(h, false);
    return n;
} PV_API_CATCH(-1)

struct PvAmdSlabRoot {
    SlabRoot* r = nullptr;
    ~PvAmdSlabRoot() { delete r; }
};
PvAmdSlabRoot* PvAmdSlabRootCreate(PvAmdSolver* anySlab, int device) try {
    Solver* s = slabOf(anySlab);
    if (!s) return nullptr;
    std::unique_ptr<PvAmdSlabRoot> h(new PvAmdSlabRoot());
    h->r = SlabRoot::create(*s, device, &g_lastError);
    return h->r ? h.release() : nullptr;
} PV_API_CATCH(nullptr)
void PvAmdSlabRootDestroy(PvAmdSlabRoot* h) try {
    delete h;
} PV_API_CATCH_VOID
static int rootRet(PvAmdSlabRoot* h, bool ok) {
    if (!ok && h && h->r) g_lastError = h->r->lastError();
    return ok ? 0 : -1;
}
int PvAmdSlabRootBegin(PvAmdSlabRoot* h, float lx, float ly, float lz) try {
    return (h && h->r) ? rootRet(h, h->r->begin(lx, ly, lz)) : -1;
} PV_API_CATCH(-1)
int PvAmdSlabRootImportBlock(PvAmdSlabRoot* h, const int* info4, const float* host) try {
    return (h && h->r && info4 && (host || info4[2] * info4[3] == 0))
               ? rootRet(h, h->r->importBlock(info4[0], info4[1], info4[2], info4[3], host))
               : -1;
} PV_API_CATCH(-1)
int PvAmdSlabRootFinish(PvAmdSlabRoot* h) try { return (h && h->r) ? rootRet(h, h->r->finish()) : -1; } PV_API_CATCH(-1)
int PvAmdSlabRootGetOutput(PvAmdSlabRoot* h, float ex, float ey, float ez, PlaneverbOutput* out) try {
    if (!h || !h->r || !out) return -1;
    float v[8];
    bool valid = false;
    if (!h->r->getOutput(ex, ey, ez, v, &valid)) return rootRet(h, false);
    std::memset(out, 0, sizeof(*out));
    if (valid)
        std::memcpy(out, v, sizeof(*out));
    else
        out->occlusion = kInvalidDryGain;
    return 0;
} PV_API_CATCH(-1)
int PvAmdSlabRootCopyResults(PvAmdSlabRoot* h, float* res8, float* delay) try {
    return (h && h->r) ? rootRet(h, h->r->copyResults(res8, delay)) : -1;
} PV_API_CATCH(-1)

void PvAmdDestroy(PvAmdSolver* h) try {
    delete h;
} PV_API_CATCH_VOID

int PvAmdSetOption(PvAmdSolver* h, int key, long long value) try {
    if (!h) return -1;
    if (h->s || h->g) {
        g_lastError = "options must be set before the solver is first used";
        return -1;
    }
    switch (key) {
        case PVA_OPT_DENSE_HISTORY: h->opt.denseHistory = value != 0; break;
        case PVA_OPT_NUM_STEPS: h->opt.numSteps = (int)value; break;
        case PVA_OPT_SKIP_ANALYSIS: h->opt.skipAnalysis = value != 0; break;
        case PVA_OPT_USE_GRAPH: h->opt.useGraph = (int)value; break;
        case PVA_OPT_STEPS_PER_LAUNCH: h->opt.K = (int)value; break;
        case PVA_OPT_TILE_ROWS: h->opt.rxi = (int)value; break;
        case PVA_OPT_NO_FREE_GRID: h->opt.withFreeGrid = value == 0; break;
        case PVA_OPT_TIME_KERNELS: h->opt.timeKernels = (int)value; break;
        case PVA_OPT_TILE_ORDER: h->opt.tileOrder = (int)value; break;
        case PVA_OPT_SMALL_GRID_KERNEL: h->opt.smallGrid = (int)value; break;
        case PVA_OPT_PACKED_MATH: h->opt.packed = value != 0; break;
        case PVA_OPT_STREAMING_ANALYSIS: h->opt.streaming = value != 0; break;
        case PVA_OPT_STREAM_ROWS: h->opt.segments = (int)value; break;
        case PVA_OPT_MERGED_LAUNCH: h->opt.merged = (int)value; break;
        case PVA_OPT_EDGE_TILES: h->opt.edgeTiles = value != 0; break;
        case PVA_OPT_ROW_BANDS: h->opt.rowBands = (int)value; break;
        case PVA_OPT_PATCH_KERNEL: h->opt.patch = (int)value; break;
        case PVA_OPT_LAZY_FAR_CELLS: h->opt.lazyFar = value != 0; break;
        case PVA_OPT_STREAM_FUSE: h->opt.streamFuse = (int)value; break;
        case PVA_OPT_AUX_STREAMS: h->opt.auxStreams = (int)std::max<long long>(0, std::min<long long>(value, 8)); break;
        case PVA_OPT_PATCH_STRIP: h->opt.patchStrip = (int)value; break;
        case PVA_OPT_RESIDENT_KERNEL: h->opt.resident = (int)value; break;
        case PVA_OPT_RT60_LANES: h->opt.rt60Lanes = (int)value; break;
        case PVA_OPT_ANALYSIS_FORK: h->opt.analysisFork = (int)value; break;
        case PVA_OPT_FUSED_ANALYSIS: h->opt.fusedAnalysis = (int)value; break;
        case PVA_OPT_DEBUG_LOSE_FIRST_CAPTURE: h->opt.debugLoseFirstCapture = value != 0; break;
        case PVA_OPT_STREAM_PRIORITY: h->opt.streamPriority = (int)value; break;
        case PVA_OPT_ALTERNATE_SWEEPS: h->opt.alternateSweeps = (int)value; break;
        case PVA_OPT_XCD_REGIONS: h->opt.xcdRegions = (int)value; break;
        default: g_lastError = "unknown option"; return -1;
    }
    return 0;
} PV_API_CATCH(-1)

int PvAmdGetInfo(PvAmdSolver* h, PvAmdInfo* out) try {
    if (!out || !ensure(h, true)) return -1;
    if (h->g) {
        const GridSpec& g = h->g->spec();
        const Solver* s0 = h->g->slab(0);
        std::memset(out, 0, sizeof(*out));
        out->gx = g.gx;
        out->gy = g.gy;
        out->T = h->g->T();
        out->fs = (int)g.fs;
        out->res = g.res;
        out->dx = g.dx;
        out->dt = g.dt;
        out->efree = h->g->efree();
        out->device = s0->device();
        out->stepsPerLaunch = s0->K();
        out->tileRows = s0->geometry().rxi;
        out->tileCols = s0->geometry().wi;
        out->pitch = s0->geometry().pitch;
        out->rows = s0->geometry().rows;
        out->histRows = s0->histRows();
        out->histPitch = s0->histPitch();
        out->numGeometry = h->g->numBoxes();
        out->deviceBytes = h->g->deviceBytes();
        return 0;
    }
    const GridSpec& g = h->s->spec();
    const Geometry& geo = h->s->geometry();
    std::memset(out, 0, sizeof(*out));
    out->gx = g.gx;
    out->gy = g.gy;
    out->T = h->s->T();
    out->fs = (int)g.fs;
    out->res = g.res;
    out->dx = g.dx;
    out->dt = g.dt;
    out->efree = h->s->efree();
    out->device = h->s->device();
    out->stepsPerLaunch = h->s->K();
    out->tileRows = geo.rxi;
    out->tileCols = geo.wi;
    out->pitch = geo.pitch;
    out->rows = geo.rows;
    out->histRows = h->s->histRows();
    out->histPitch = h->s->histPitch();
    out->numGeometry = h->s->numBoxes();
    out->deviceBytes = h->s->deviceBytes();
    out->streamFuse = h->s->streamFuse() ? 1 : 0;
    out->residentKernel = h->s->residentKernel() ? 1 : 0;
    return 0;
} PV_API_CATCH(-1)

int PvAmdAddGeometry(PvAmdSolver* h, float posX, float posY, float width, float height, float absorption) try {
    if (!ensure(h, true)) return -1;
    if (h->g) return h->g->addBox(Box{posX, posY, width, height, absorption});
    return h->s->addBox(Box{posX, posY, width, height, absorption});
} PV_API_CATCH(-1)

int PvAmdUpdateGeometry(PvAmdSolver* h, int id, float posX, float posY, float width, float height,
                        float absorption) try {
    if (!ensure(h, true)) return -1;
    if (h->g) return ret(h, h->g->updateBox(id, Box{posX, posY, width, height, absorption}));
    return ret(h, h->s->updateBox(id, Box{posX, posY, width, height, absorption}));
} PV_API_CATCH(-1)

int PvAmdRemoveGeometry(PvAmdSolver* h, int id) try {
    if (!ensure(h, true)) return -1;
    if (h->g) return ret(h, h->g->removeBox(id));
    return ret(h, h->s->removeBox(id));
} PV_API_CATCH(-1)

int PvAmdLoadScene(PvAmdSolver* h, const char* pvPath) try {
    if (!ensure(h, true) || !pvPath) return -1;
    std::vector<Box> boxes;
    if (!loadPv(pvPath, &boxes, &g_lastError)) return -1;
    for (const Box& b : boxes) {
        if (h->g)
            h->g->addBox(b);
        else
            h->s->addBox(b);
    }
    return (int)boxes.size();
} PV_API_CATCH(-1)

int PvAmdSaveScene(PvAmdSolver* h, const char* pvPath) try {
    if (!ensure(h) || !pvPath) return -1;
    return savePv(pvPath, h->s->boxes(), &g_lastError) ? 0 : -1;
} PV_API_CATCH(-1)

int PvAmdRun(PvAmdSolver* h, float lx, float ly, float lz) try {
    if (!wholeGrid(h) || !ensure(h, true)) return -1;
    if (h->g) return ret(h, h->g->run(lx, ly, lz));
    return ret(h, h->s->run(lx, ly, lz, true));
} PV_API_CATCH(-1)

int PvAmdRunAsync(PvAmdSolver* h, float lx, float ly, float lz) try {
    if (!wholeGrid(h) || !ensure(h)) return -1;
    return ret(h, h->s->run(lx, ly, lz, false));
} PV_API_CATCH(-1)

int PvAmdRunAsyncAfter(PvAmdSolver* h, PvAmdSolver* prev, float lx, float ly, float lz) try {
    if (!wholeGrid(h) || !ensure(h) || !prev || !wholeGrid(prev) || !ensure(prev)) return -1;
    if (prev->s->spec().gx != h->s->spec().gx || prev->s->spec().gy != h->s->spec().gy || prev->s->device() != h->s->device()) {
        g_lastError = "PvAmdRunAsyncAfter: the two solvers must have the same grid and device";
        return -1;
    }
    return ret(h, h->s->run(lx, ly, lz, false, prev->s));
} PV_API_CATCH(-1)

int PvAmdRunBatch(PvAmdSolver* const* hs, int n, const float* listenersXYZ, int wait) try {
    if (!hs || !listenersXYZ || n < 1 || n > kBatchMax) {
        g_lastError = "PvAmdRunBatch: 1..8 solvers and their listener positions";
        return -1;
    }
    Solver* s[kBatchMax];
    for (int i = 0; i < n; ++i) {
        if (!wholeGrid(hs[i]) || !ensure(hs[i])) return -1;
        s[i] = hs[i]->s;
    }
    std::string err;
    if (Solver::runBatch(s, n, listenersXYZ, wait != 0, &err)) return 0;
    g_lastError = err.empty() ? "batched run failed" : err;
    return -1;
} PV_API_CATCH(-1)

int PvAmdSync(PvAmdSolver* h) try {
    if (!wholeGrid(h) || !ensure(h)) return -1;
    return ret(h, h->s->sync());
} PV_API_CATCH(-1)

float PvAmdClockProbe(int device, float* byMemtimeMHz) try {
    return clockProbeMHz(device, byMemtimeMHz);
} PV_API_CATCH(0.f)

int PvAmdBandwidthProbe(int device, float* gbPerS4) try {
    if (!gbPerS4) return -1;
    if (bandwidthProbeGBs(device, gbPerS4)) return 0;
    g_lastError = "PvAmdBandwidthProbe: allocation or launch failed (2 GiB of device memory are needed)";
    return -1;
} PV_API_CATCH(-1)

int PvAmdGetTimings(PvAmdSolver* h, PvAmdTimings* out) try {
    if (!out || !ensure(h, true)) return -1;
    const SolverTimings& t = h->g ? h->g->timings() : h->s->timings();
    out->fdtdMs = t.fdtdMs;
    out->analysisMs = t.analysisMs;
    out->geometryMs = t.geometryMs;
    out->stepLaunches = t.stepLaunches;
    out->stepKernelMs = t.stepLaunches ? t.fdtdMs / (float)t.stepLaunches : 0.f;
    out->airKernelMs = t.airKernelMs;
    out->generalKernelMs = t.generalKernelMs;
    out->airLaunches = t.airLaunches;
    out->generalLaunches = t.generalLaunches;
    out->stepLoopMs = t.stepLoopMs;
    out->reachedCells = t.reachedCells;
    out->activeCells = t.activeCells;
    out->silentCells = t.silentCells;
    return 0;
} PV_API_CATCH(-1)

int PvAmdSetEmitters(PvAmdSolver* h, const float* xyz, int n) try {
    if (!wholeGrid(h) || !ensure(h) || (n > 0 && !xyz)) return -1;
    return ret(h, h->s->setEmitters(xyz, n));
} PV_API_CATCH(-1)

int PvAmdGetOutput(PvAmdSolver* h, float ex, float ey, float ez, PlaneverbOutput* out) try {
    if (!wholeGrid(h) || !out || !ensure(h, true)) return -1;
    float v[8];
    bool valid = false;
    if (!(h->g ? h->g->getOutput(ex, ey, ez, v, &valid) : h->s->getOutput(ex, ey, ez, v, &valid))) return ret(h, false);
    if (!valid) {
        std::memset(out, 0, sizeof(*out));
        out->occlusion = kInvalidDryGain;
        return 0;
    }
    std::memcpy(out, v, sizeof(*out));
    return 0;
} PV_API_CATCH(-1)

int PvAmdSetOutputQueries(PvAmdSolver* h, const float* xyz, int n) try {
    if (!wholeGrid(h) || (n > 0 && !xyz) || !ensure(h)) return -1;
    return ret(h, h->s->setOutputQueries(xyz, n));
} PV_API_CATCH(-1)

int PvAmdGetQueriedOutputs(PvAmdSolver* h, PlaneverbOutput* out, int n) try {
    if (n < 0 || n > Solver::kMaxQueries || (n > 0 && !out) || !wholeGrid(h) || !ensure(h)) return -1;
    float v[Solver::kMaxQueries * 8];
    unsigned char valid[Solver::kMaxQueries];
    if (!h->s->queriedOutputs(v, valid, n)) return ret(h, false);
    for (int i = 0; i < n; ++i) {
        if (valid[i]) {
            std::memcpy(&out[i], v + 8 * i, sizeof(PlaneverbOutput));
        } else {  // the reference's sentinel for a position outside the grid (FDTD.cpp:19-47)
            std::memset(&out[i], 0, sizeof(PlaneverbOutput));
            out[i].occlusion = kInvalidDryGain;
        }
    }
    return 0;
} PV_API_CATCH(-1)

int PvAmdCopyResults(PvAmdSolver* h, float* res8, float* delay) try {
    if (!wholeGrid(h) || !ensure(h, true)) return -1;
    return ret(h, h->g ? h->g->copyResults(res8, delay) : h->s->copyResults(res8, delay));
} PV_API_CATCH(-1)

int PvAmdCopyResultsBlock(PvAmdSolver* h, int r0, int c0, int nr, int nc, float* res8, float* delay) try {
    if (!wholeGrid(h) || !ensure(h)) return -1;
    return ret(h, h->s->copyResultsBlock(r0, c0, nr, nc, res8, delay));
} PV_API_CATCH(-1)

int PvAmdGetImpulseResponse(PvAmdSolver* h, int cx, int cy, float* out3T) try {
    if (!out3T || !ensure(h, true)) return -1;
    return ret(h, h->g ? h->g->impulseResponse(cx, cy, out3T) : h->s->impulseResponse(cx, cy, out3T));
} PV_API_CATCH(-1)

int PvAmdGetImpulseResponseCells(PvAmdSolver* h, int cx, int cy, PlaneverbCell* outT) try {
    if (!outT || !ensure(h)) return -1;
    static_assert(sizeof(PlaneverbCell) == 16, "PvTypes.h:106-121");
    return ret(h, h->s->impulseResponseCells(cx, cy, outT));
} PV_API_CATCH(-1)

int PvAmdCopyFields(PvAmdSolver* h, float* pr, float* vx, float* vy) try {
    if (!ensure(h, true)) return -1;
    return ret(h, h->g ? h->g->copyFields(pr, vx, vy) : h->s->copyFields(pr, vx, vy));
} PV_API_CATCH(-1)

int PvAmdCopyHistoryPlane(PvAmdSolver* h, int t, float* pr) try {
    if (!pr || !ensure(h, true)) return -1;
    return ret(h, h->g ? h->g->copyHistoryPlane(t, pr) : h->s->copyHistoryPlane(t, pr));
} PV_API_CATCH(-1)

int PvAmdCopyPulse(PvAmdSolver* h, float* out) try {
    if (!out || !ensure(h, true)) return -1;
    return ret(h, h->g ? h->g->copyPulse(out) : h->s->copyPulse(out));
} PV_API_CATCH(-1)

int PvAmdCopyMaterial(PvAmdSolver* h, uint8_t* beta, float* R) try {
    if (!ensure(h, true)) return -1;
    return ret(h, h->g ? h->g->copyMaterial(beta, R) : h->s->copyMaterial(beta, R));
} PV_API_CATCH(-1)

int PvAmdSetFields(PvAmdSolver* h, const float* pr, const float* vx, const float* vy) try {
    if (!ensure(h)) return -1;
    return ret(h, h->s->setFields(pr, vx, vy));
} PV_API_CATCH(-1)

int PvAmdRunSteps(PvAmdSolver* h, int nsteps, int withPulse, float lx, float lz) try {
    if (!wholeGrid(h) || !ensure(h)) return -1;
    return ret(h, h->s->runSteps(nsteps, withPulse != 0, lx, lz));
} PV_API_CATCH(-1)

// ---------------------------------------------------------------------------------------------------------------
// Part 3: sharded runs + RCCL gather
// ---------------------------------------------------------------------------------------------------------------

struct PvAmdComm {
    Comm* c = nullptr;
    ~PvAmdComm() { delete c; }
};

int PvAmdShardPlan(int nRuns, int world, int rank, int nLocalSolvers, int* runIdx, int* solverIdx, int cap) try {
    const std::vector<ShardItem> plan = shardPlan(nRuns, world, rank, nLocalSolvers);
    for (int i = 0; i < (int)plan.size() && i < cap; ++i) {
        if (runIdx) runIdx[i] = plan[(size_t)i].run;
        if (solverIdx) solverIdx[i] = plan[(size_t)i].solver;
    }
    return (int)plan.size();
} PV_API_CATCH(-1)

int PvAmdPlanSegments(const unsigned char* air, int ntx, int nty, int tileRows, int maxTileColumns, int target, int* seg4,
                      int cap) try {
    if (!air || ntx < 1 || nty < 1) return 0;
    const std::vector<SegRect> segs = planSegments(air, ntx, nty, tileRows, maxTileColumns, target);
    for (int i = 0; i < (int)segs.size() && i < cap && seg4; ++i) {
        seg4[4 * i] = segs[(size_t)i].row0;
        seg4[4 * i + 1] = segs[(size_t)i].nrows;
        seg4[4 * i + 2] = segs[(size_t)i].tj0;
        seg4[4 * i + 3] = segs[(size_t)i].w;
    }
    return (int)segs.size();
} PV_API_CATCH(0)

int PvAmdCommUniqueId(char id128[128]) try {
    if (!id128) return -1;
    return Comm::uniqueId(id128, &g_lastError) ? 0 : -1;
} PV_API_CATCH(-1)

PvAmdComm* PvAmdCommCreate(const char id128[128], int rank, int world, int device) try {
    if (!id128) return nullptr;
    std::unique_ptr<PvAmdComm> h(new PvAmdComm());
    h->c = Comm::create(id128, rank, world, device, &g_lastError);
    return h->c ? h.release() : nullptr;
} PV_API_CATCH(nullptr)

void PvAmdCommDestroy(PvAmdComm* h) try {
    delete h;
} PV_API_CATCH_VOID

int PvAmdCommAllGather(PvAmdComm* h, const float* mine, int countPerRank, float* all) try {
    if (!h || !h->c || !mine || !all) return -1;
    return h->c->allGather(mine, countPerRank, all, &g_lastError) ? 0 : -1;
} PV_API_CATCH(-1)

int PvAmdRunSharded(PvAmdSolver* const* hs, int nSolvers, const float* listenersXYZ, int nRuns, const float* emittersXYZ,
                    int E, int rank, int world, PvAmdComm* comm, PlaneverbOutput* out) try {
    if (!hs || nSolvers < 1 || !listenersXYZ || nRuns < 0 || (E > 0 && !emittersXYZ) || E < 0 || E > Solver::kMaxQueries ||
        !out || world < 1 || rank < 0 || rank >= world) {
        g_lastError = "PvAmdRunSharded: invalid arguments";
        return -1;
    }
    if (world > 1 && (!comm || !comm->c || comm->c->world() != world || comm->c->rank() != rank)) {
        g_lastError = "PvAmdRunSharded: ranks span processes, a matching PvAmdComm is required";
        return -1;
    }
    std::vector<Solver*> sv;
    for (int i = 0; i < nSolvers; ++i) {
        if (!wholeGrid(hs[i]) || !ensure(hs[i])) return -1;
        sv.push_back(hs[i]->s);
    }
    const std::vector<ShardItem> plan = shardPlan(nRuns, world, rank, nSolvers);
    const int perRank = (nRuns + world - 1) / world;
    const size_t rec = (size_t)E * 8;
    std::vector<float> mine((size_t)perRank * rec, 0.f);
    std::vector<int> pending((size_t)nSolvers, -1);
    auto collect = [&](int s) -> bool {  // wait for solver s's run, take its records (gathered behind the analysis)
        const int j = pending[(size_t)s];
        pending[(size_t)s] = -1;
        float v[Solver::kMaxQueries * 8];
        unsigned char valid[Solver::kMaxQueries];
        if (!sv[(size_t)s]->sync() || !sv[(size_t)s]->queriedOutputs(v, valid, E)) return false;
        for (int e = 0; e < E; ++e) {
            float* dst = mine.data() + (size_t)j * rec + (size_t)e * 8;
            if (valid[e]) {
                std::memcpy(dst, v + 8 * e, 32);
            } else {  // the reference's sentinel for a position outside the grid (FDTD.cpp:19-47)
                std::memset(dst, 0, 32);
                dst[0] = kInvalidDryGain;
            }
        }
        return true;
    };
    for (size_t j = 0; j < plan.size(); ++j) {
        const int s = plan[j].solver, k = plan[j].run;
        if (pending[(size_t)s] >= 0 && !collect(s)) return ret(hs[s], false);  // the other solvers keep the GPU busy
        if (!sv[(size_t)s]->setOutputQueries(emittersXYZ + (size_t)k * E * 3, E) ||
            !sv[(size_t)s]->run(listenersXYZ[3 * k], listenersXYZ[3 * k + 1], listenersXYZ[3 * k + 2], /*wait=*/false))
            return ret(hs[s], false);
        pending[(size_t)s] = (int)j;
    }
    for (int s = 0; s < nSolvers; ++s)
        if (pending[(size_t)s] >= 0 && !collect(s)) return ret(hs[s], false);
    float* o = reinterpret_cast<float*>(out);
    if (world == 1) {
        for (int k = 0; k < nRuns; ++k) std::memcpy(o + (size_t)k * rec, mine.data() + (size_t)k * rec, rec * 4);
        return 0;
    }
    std::vector<float> all((size_t)world * perRank * rec);
    if (rec > 0 && !comm->c->allGather(mine.data(), (int)((size_t)perRank * rec), all.data(), &g_lastError)) return -1;
    for (int k = 0; k < nRuns; ++k)  // run k = the (k / world)-th run of rank k mod world
        std::memcpy(o + (size_t)k * rec, all.data() + ((size_t)(k % world) * perRank + (size_t)(k / world)) * rec, rec * 4);
    return 0;
} PV_API_CATCH(-1)

#endif  // !PVA_HOST_TEST

int PvAmdHostGridInfo(float sx, float sy, int res, PvAmdInfo* out) try {
    if (!out || res < kLowResolution) return -1;
    const GridSpec g = makeGridSpec(sx, sy, res);
    std::memset(out, 0, sizeof(*out));
    out->gx = g.gx;
    out->gy = g.gy;
    out->T = g.T;
    out->fs = (int)g.fs;
    out->res = g.res;
    out->dx = g.dx;
    out->dt = g.dt;
    return 0;
} PV_API_CATCH(-1)

int PvAmdHostPulseSelfCheck(void) try { return pulseMatchesReferenceLibm() ? 1 : 0; } PV_API_CATCH(0)

int PvAmdHostPulse(float sx, float sy, int res, float* out) try {
    if (!out || res < kLowResolution) return -1;
    const GridSpec g = makeGridSpec(sx, sy, res);
    const std::vector<float> p = gaussianPulse(g);
    std::memcpy(out, p.data(), p.size() * sizeof(float));
    return 0;
} PV_API_CATCH(-1)

int PvAmdHostRasterize(float sx, float sy, int res, const float* b5, const int* ops, int n, uint8_t* beta,
                       float* R) try {
    if (res < kLowResolution) return -1;
    const GridSpec g = makeGridSpec(sx, sy, res);
    MaterialPlane m;
    m.init(g);
    for (int i = 0; i < n; ++i) {
        const Box b{b5[5 * i], b5[5 * i + 1], b5[5 * i + 2], b5[5 * i + 3], b5[5 * i + 4]};
        if (ops && ops[i] < 0)
            m.remove(b);
        else
            m.add(b);
    }
    const size_t cells = (size_t)g.NX * g.NY;
    if (beta) std::memcpy(beta, m.beta().data(), cells);
    if (R) std::memcpy(R, m.R().data(), cells * sizeof(float));
    return 0;
} PV_API_CATCH(-1)

int PvAmdHostLoadPv(const char* path, float* b5, int maxBoxes) try {
    if (!path) return -1;
    std::vector<Box> boxes;
    if (!loadPv(path, &boxes, &g_lastError)) return -1;
    for (int i = 0; i < (int)boxes.size() && i < maxBoxes; ++i) {
        b5[5 * i] = boxes[(size_t)i].x;
        b5[5 * i + 1] = boxes[(size_t)i].y;
        b5[5 * i + 2] = boxes[(size_t)i].w;
        b5[5 * i + 3] = boxes[(size_t)i].h;
        b5[5 * i + 4] = boxes[(size_t)i].R;
    }
    return (int)boxes.size();
} PV_API_CATCH(-1)

int PvAmdHostSavePv(const char* path, const float* b5, const int* ids, int n) try {
    if (!path || (n > 0 && !b5)) return -1;
    std::vector<std::pair<int, Box>> boxes;
    for (int i = 0; i < n; ++i)
        boxes.emplace_back(ids ? ids[i] : i, Box{b5[5 * i], b5[5 * i + 1], b5[5 * i + 2], b5[5 * i + 3], b5[5 * i + 4]});
    return savePv(path, boxes, &g_lastError) ? 0 : -1;
} PV_API_CATCH(-1)

int PvAmdHostCells(float sx, float sy, int res, float x, float z, int* lcx, int* lcy, int* rcx, int* rcy,
                   int* rvalid) try {
    if (res < kLowResolution) return -1;
    const GridSpec g = makeGridSpec(sx, sy, res);
    listenerCell(g, x, z, lcx, lcy);
    int cx = -1, cy = -1;
    *rvalid = resultCell(g, x, z, &cx, &cy) ? 1 : 0;
    *rcx = cx;
    *rcy = cy;
    return 0;
} PV_API_CATCH(-1)

void PvAmdReverbBusGains(float rt60, float wetGain, float* a, float* b, float* c) try {
    reverbBusGains(rt60, wetGain, a, b, c);
} PV_API_CATCH_VOID

}  // extern "C"
