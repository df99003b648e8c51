// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Direct-drive harness around the UNMODIFIED reference classes (harness style 1):
// Grid (ProjectPlaneverb/src/FDTD/Grid.h:25-74), FreeGrid (FDTD/FreeGrid.h:7-24) and Analyzer
// (DSP/Analyzer.h:24-50) are constructed and called exactly as Context::Context does
// (Context/PvContext.cpp:135-155) but without the background thread, so a run is deterministic.
// Built by `make -C oracle ref` into oracle/_ref/libpvref.so; the reference sources are compiled
// where they lie under /root/reference and are never copied into this repository.
//
// `#define private public` is used ONLY in this translation unit to read the reference's private
// result arrays (material plane, pulse table, result map, delay map, EFree).  It does not change
// any class layout or any arithmetic.
#define private public
#include <FDTD\Grid.h>
#include <FDTD\FreeGrid.h>
#include <DSP\Analyzer.h>
#undef private
#include <Planeverb.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>

using namespace Planeverb;

namespace {
double now_s() {
    using clk = std::chrono::steady_clock;
    return std::chrono::duration<double>(clk::now().time_since_epoch()).count();
}
}  // namespace

struct PvRef {
    PlaneverbConfig cfg;
    char* gridPool = nullptr;
    char* anPool = nullptr;
    Grid* grid = nullptr;
    FreeGrid* freeGrid = nullptr;
    Analyzer* analyzer = nullptr;
    double ctorGrid = 0, ctorFree = 0;
};

extern "C" {

// aabbs: n x {posX, posY, width, height, absorption} (the .pv record minus its id, Editor.cpp:262-279)
PvRef* pvref_create(float sizeX, float sizeY, int res, const float* aabbs, int n, int withFreeGrid) {
    PvRef* r = new PvRef();
    r->cfg.gridSizeInMeters = vec2(sizeX, sizeY);
    r->cfg.gridResolution = res;
    r->cfg.gridBoundaryType = pv_AbsorbingBoundary;
    r->cfg.tempFileDirectory = ".";
    r->cfg.maxThreadUsage = 1;
    r->cfg.threadExecutionType = pv_CPU;
    r->cfg.gridWorldOffset = vec2(0.f, 0.f);

    double t0 = now_s();
    unsigned gsz = Grid::GetMemoryRequirement(&r->cfg);
    r->gridPool = new char[gsz];
    r->grid = new Grid(&r->cfg, r->gridPool);
    r->ctorGrid = now_s() - t0;
    for (int i = 0; i < n; ++i) {
        AABB a;
        a.position = vec2(aabbs[5 * i + 0], aabbs[5 * i + 1]);
        a.width = aabbs[5 * i + 2];
        a.height = aabbs[5 * i + 3];
        a.absorption = aabbs[5 * i + 4];
        r->grid->AddAABB(&a);
    }
    if (withFreeGrid) {
        t0 = now_s();
        r->freeGrid = new FreeGrid(&r->cfg, nullptr);
        r->ctorFree = now_s() - t0;
        unsigned asz = Analyzer::GetMemoryRequirement(&r->cfg);
        r->anPool = new char[asz];
        std::memset(r->anPool, 0, asz);  // Context zeroes its pool: PvContext.cpp:132
        r->analyzer = new Analyzer(r->grid, r->freeGrid, r->anPool);
    }
    return r;
}

void pvref_destroy(PvRef* r) {
    if (!r) return;
    delete r->analyzer;
    delete r->freeGrid;
    delete r->grid;
    delete[] r->anPool;
    delete[] r->gridPool;
    delete r;
}

void pvref_info(PvRef* r, int* gx, int* gy, int* T, int* fs, float* dx, float* dt, float* efree,
                double* ctorGrid, double* ctorFree) {
    *gx = (int)r->grid->m_gridSize.x;
    *gy = (int)r->grid->m_gridSize.y;
    *T = (int)r->grid->m_responseLength;
    *fs = (int)r->grid->m_samplingRate;
    *dx = r->grid->m_dx;
    *dt = r->grid->m_dt;
    *efree = r->freeGrid ? r->freeGrid->m_EFree : 0.f;
    *ctorGrid = r->ctorGrid;
    *ctorFree = r->ctorFree;
}

void pvref_pulse(PvRef* r, float* out) {
    std::memcpy(out, r->grid->m_pulse, sizeof(float) * r->grid->m_responseLength);
}

// b and R planes, (gx+1)*(gy+1) each, in the reference's own linear order
void pvref_material(PvRef* r, short* b, float* R) {
    int n = ((int)r->grid->m_gridSize.x + 1) * ((int)r->grid->m_gridSize.y + 1);
    for (int i = 0; i < n; ++i) {
        b[i] = r->grid->m_grid[i].b;
        R[i] = r->grid->m_boundaries[i].absorption;
    }
}

void pvref_add_aabb(PvRef* r, const float* a5) {
    AABB a;
    a.position = vec2(a5[0], a5[1]);
    a.width = a5[2];
    a.height = a5[3];
    a.absorption = a5[4];
    r->grid->AddAABB(&a);
}

void pvref_remove_aabb(PvRef* r, const float* a5) {
    AABB a;
    a.position = vec2(a5[0], a5[1]);
    a.width = a5[2];
    a.height = a5[3];
    a.absorption = a5[4];
    r->grid->RemoveAABB(&a);
}

double pvref_generate(PvRef* r, float lx, float ly, float lz) {
    double t0 = now_s();
    r->grid->GenerateResponse(vec3(lx, ly, lz));
    return now_s() - t0;
}

double pvref_analyze(PvRef* r, float lx, float ly, float lz) {
    double t0 = now_s();
    r->analyzer->AnalyzeResponses(vec3(lx, ly, lz));
    return now_s() - t0;
}

// res8: gx*gy x {occlusion, wetGain, rt60, lowpass, dirX, dirY, srcDirX, srcDirY}; delay: gx*gy
void pvref_results(PvRef* r, float* res8, float* delay) {
    int n = (int)r->analyzer->m_gridX * (int)r->analyzer->m_gridY;
    std::memcpy(res8, r->analyzer->m_results, sizeof(AnalyzerResult) * n);
    std::memcpy(delay, r->analyzer->m_delaySamples, sizeof(float) * n);
}

// IR at cube cell (cx, cy) through Grid::GetResponse (FDTD.cpp:74-79): out = T x {pr, vx, vy}
void pvref_ir(PvRef* r, int cx, int cy, float* out) {
    const Cell* c = r->grid->GetResponse(vec2((float)cx, (float)cy));
    unsigned T = r->grid->GetResponseSize();
    for (unsigned t = 0; t < T; ++t) {
        out[3 * t + 0] = c[t].pr;
        out[3 * t + 1] = c[t].vx;
        out[3 * t + 2] = c[t].vy;
    }
}

// the same IR as raw 16-byte reference Cells {pr, vx, vy, short b, short by} (PvTypes.h:106-121): what
// Planeverb::GetImpulseResponse hands its caller (FDTD.cpp:60-70)
void pvref_ir_cells(PvRef* r, int cx, int cy, void* out16T) {
    const Cell* c = r->grid->GetResponse(vec2((float)cx, (float)cy));
    static_assert(sizeof(Cell) == 16, "reference Cell is 16 bytes");
    std::memcpy(out16T, c, sizeof(Cell) * r->grid->GetResponseSize());
}

// recorded fields of step t for every cell of the (gx+1)x(gy+1) cube, reference linear order
void pvref_snapshot(PvRef* r, int t, float* pr, float* vx, float* vy) {
    int n = ((int)r->grid->m_gridSize.x + 1) * ((int)r->grid->m_gridSize.y + 1);
    for (int i = 0; i < n; ++i) {
        const Cell& c = r->grid->m_pulseResponse[i][t];
        pr[i] = c.pr;
        vx[i] = c.vx;
        vy[i] = c.vy;
    }
}

// Analyzer::GetResponseResult (Analyzer.cpp:106-116) for a world-space emitter position
int pvref_output(PvRef* r, float ex, float ey, float ez, float* out8) {
    const AnalyzerResult* a = r->analyzer->GetResponseResult(vec3(ex, ey, ez));
    if (!a) return 0;
    std::memcpy(out8, a, sizeof(AnalyzerResult));
    return 1;
}

}  // extern "C"
