// pv_slabs.h -- ONE grid split into row slabs (single-grid domain decomposition).
//
// The reference has no counterpart: its one loop advances the whole grid (Context/PvContext.cpp:63-94).  Here a grid
// of ntx tile rows is cut into S slabs of whole tile rows; slab s is a Solver that allocates planes, history and
// result maps for ITS rows only (HBM per device falls with 1/S) and runs the unchanged step kernels on them.  What
// couples the slabs:
//   * per K-step launch: the K rows of pr, vx, vy next to a slab boundary travel into the neighbour's guard band
//     (3 contiguous blocks of K x pitch floats each way; same device: a copy on the receiver's stream, other device:
//     a peer copy over xGMI) -- ordered by one event per slab and launch, nothing ever waits for the whole grid;
//   * per run: the pressure history of each slab's LAST row goes to the slab below (the analysis re-derives vx from the
//     pressure history, and vx of a slab's first row needs the row above): T x histPitch floats;
//   * per run: every slab analyses its own cells (onset, gains, decay time, source directivity); the window block of
//     those maps is gathered into whole-grid maps on the first slab's device, where the listener-direction descent --
//     a walk over the delay / occlusion maps that crosses slab boundaries -- runs once for the whole window.
// Results are bit-identical to one Solver on the whole grid (tests/test_gpu_slabs.py).
#pragma once

#include <string>
#include <vector>

#include "pv_solver.h"

namespace pva {

class SlabGroup {
public:
    // devices[s] = HIP device of slab s (all equal: S slabs on one GPU)
    static SlabGroup* create(const GridSpec& spec, const std::vector<int>& devices, const SolverOptions& opt,
                             std::string* err);
    ~SlabGroup();
    static constexpr int kProbeSweeps = 16;  // sweeps of the hand-off's dry run (probeHandoff)
    float probeUsPerSweep_ = 0.f;            // how long one of them took, launches and sync included
    int redeals_ = 0;                        // times every slab was given another stream because the dry run was slow or timed out

    const GridSpec& spec() const { return g_; }
    int numSlabs() const { return (int)slabs_.size(); }
    bool handoffWords() const { return handoff_ != nullptr; }
    int streamRedeals() const { return redeals_; }
    float dryRunUsPerSweep() const { return probeUsPerSweep_; }
    const Solver* slab(int s) const { return slabs_[(size_t)s]; }
    int slabRow0(int s) const { return slabs_[(size_t)s]->x0_; }
    int slabRows(int s) const { return slabs_[(size_t)s]->lNX_; }
    float efree() const { return efree_; }
    int T() const { return T_; }
    long long deviceBytes() const;
    const std::string& lastError() const { return err_; }
    const SolverTimings& timings() const { return tim_; }
    // bytes of pr / vx / vy halo rows moved between slabs per launch, and of boundary history + result blocks per run
    long long haloBytesPerLaunch() const;
    long long exchangeBytesPerRun() const { return exchangePerRun_; }

    int addBox(const Box& b);
    bool updateBox(int id, const Box& b);
    bool removeBox(int id);
    int numBoxes() const { return slabs_[0]->numBoxes(); }

    bool run(float lx, float ly, float lz);
    bool getOutput(float ex, float ey, float ez, float out8[8], bool* valid);
    bool copyResults(float* res8, float* delay);
    bool copyFields(float* pr, float* vx, float* vy);
    bool copyHistoryPlane(int t, float* pr);
    bool impulseResponse(int cx, int cy, float* out3T);
    bool copyMaterial(uint8_t* beta, float* R) { return slabs_[0]->copyMaterial(beta, R); }
    bool copyPulse(float* out) { return slabs_[0]->copyPulse(out); }

private:
    SlabGroup() = default;
    bool init(const GridSpec& spec, const std::vector<int>& devices, const SolverOptions& opt);
    bool fail(const std::string& what);
    bool hipOk(hipError_t e, const char* what);
    bool slabFailed(int s);
    AnalyzeArgs rootArgs(float lx, float lz) const;
    bool probeHandoff();

    GridSpec g_;
    std::vector<Solver*> slabs_;
    std::vector<int> devices_;
    bool pushHalos_ = true;  // halos are pushed by pv_halo_push_kernel (peer stores); false: pulled with hipMemcpyAsync
    std::vector<hipEvent_t> stepEv_;   // 2 per slab (launch parity)
    std::vector<hipEvent_t> miscEv_;   // 1 per slab: boundary history ready / cell analysis done
    int K_ = 0, rxi_ = 0, wi_ = 0, T_ = 0;
    float efree_ = 0.f;
    std::string err_;
    SolverTimings tim_;
    long long exchangePerRun_ = 0;

    // whole-grid maps on devices_[0]
    int rootDevice_ = 0;
    hipStream_t rootStream_ = nullptr;
    QueueClaim rootQueue_;
    hipEvent_t rootEv_[3] = {nullptr, nullptr, nullptr};
    float* res_ = nullptr;    // 8 planes x gx*gy
    float* res8_ = nullptr;   // AoS, on demand
    float* delay_ = nullptr;  // gx*gy
    int* dirScratch_ = nullptr;
    int* planesDev_ = nullptr;  // {0, 1, 2, 3, 6, 7}: everything a slab computes (the direction is the root's)
    DynParams* dynDev_ = nullptr;
    unsigned* handoff_ = nullptr;  // 3 words per slab: device-side hand-off between slabs of one device (pv_halo_push_kernel)
    DynParams* dynHost_ = nullptr;  // pinned
    float* outHost_ = nullptr;      // pinned, 8 floats
    int winRows_ = 0, winCols_ = 0;
    float lastLx_ = 0, lastLz_ = 0;
    bool ran_ = false;
};

// ---- the same decomposition with the slabs in DIFFERENT PROCESSES (one rank per GPU, torch.distributed / RCCL or any
// other transport in the host language): the group's steps as per-rank primitives.  A rank owns one slab Solver
// (SolverOptions::slabIndex / slabCount); what crosses ranks is handed over as plain host buffers, so the transport is
// the caller's (planeverb_amd/dist_slabs.py: send / recv over torch.distributed).
struct SlabRankOps {
    static bool begin(Solver& v, float lx, float ly, float lz);  // per-run set-up (listener, window, begin-run kernel)
    static bool launch(Solver& v, int li);                        // K-step launch li of this slab's rows
    static int numLaunches(const Solver& v);
    static int haloFloats(const Solver& v);                       // 3 planes x K rows x pitch
    // side 0 = towards the slab ABOVE (smaller rows), 1 = towards the slab BELOW
    static bool exportHalo(Solver& v, int side, float* host);       // my K boundary rows of the set just written
    static bool importHalo(Solver& v, int side, const float* host); // the neighbour's rows into my guard band
    static int historyFloats(const Solver& v);                    // T x histPitch
    static bool exportEdgeHistory(Solver& v, float* host);        // my last row's pressure history (for the slab below)
    static bool importAboveHistory(Solver& v, const float* host);
    static bool analyze(Solver& v);                               // far cells + cell analysis of my rows
    // the window block of my maps: whole-grid row of its first row, first column, extent; 7 planes (result planes
    // 0,1,2,3,6,7 + delay) of nr x nc floats into host (capacity in floats; returns the floats needed, < 0 on error)
    static long long windowBlock(Solver& v, int* r0g, int* c0, int* nr, int* nc, float* host, long long cap);
};

// whole-grid result / delay maps of a decomposed grid whose slabs live in other processes: far cells, the slabs' blocks,
// then the listener-direction descent (what SlabGroup does on its first device)
class SlabRoot {
public:
    static SlabRoot* create(const Solver& anySlab, int device, std::string* err);  // geometry taken from a slab
    ~SlabRoot();
    bool begin(float lx, float ly, float lz);
    bool importBlock(int r0g, int c0, int nr, int nc, const float* host7);
    bool finish();
    bool getOutput(float ex, float ey, float ez, float out8[8], bool* valid);
    bool copyResults(float* res8, float* delay);
    const std::string& lastError() const { return err_; }
    const GridSpec& spec() const { return g_; }

private:
    SlabRoot() = default;
    bool fail(const std::string& w) {
        err_ = w;
        return false;
    }
    GridSpec g_;
    int device_ = 0, G_ = 0, rxi_ = 0, wi_ = 0, nty_ = 0, T_ = 0, K_ = 0;
    int ntxG_ = 0, histTilesXG_ = 0, histTilesY_ = 0;
    float efree_ = 0;
    hipStream_t stream_ = nullptr;
    float* res_ = nullptr;
    float* res8_ = nullptr;
    float* delay_ = nullptr;
    float* stage_ = nullptr;  // one block, 7 planes
    size_t stageCap_ = 0;
    int* dirScratch_ = nullptr;
    int* planesDev_ = nullptr;
    DynParams* dynDev_ = nullptr;
    float* outHost_ = nullptr;
    int winRows_ = 0, winCols_ = 0;
    float lx_ = 0, lz_ = 0;
    std::string err_;
    AnalyzeArgs args() const;
};

}  // namespace pva
