// pv_context.h -- the live acoustics module behind the reference's C-ABI (PlaneverbInit ... PlaneverbExit).
//
// Mirrors ProjectPlaneverb/src/Context/PvContext.{h,cpp}: a singleton that owns the grid, the geometry and
// emission tables, and one library-owned worker that loops
//     GenerateResponse(listener) -> AnalyzeResponses(listener) -> PushGeometryChanges() -> latch listener
// (PvContext.cpp:63-94).  Here the worker owns a Solver (one HIP stream on one MI355X) instead of running the
// sweeps itself, and publishes what each iteration can have changed -- the history-window block of the result map
// -- into a pinned host double buffer, so that GetOutput stays O(1), never touches the device and never takes a
// lock (the reference reads the result grid unsynchronised while it is rewritten).
//
// Threading contract (checked under ThreadSanitizer / AddressSanitizer by tests/host/, a HIP-less build of this
// file and pv_core.cpp against a fake Solver):
//   * any API function may be called from any thread at any time, including concurrently with Exit / re-Init:
//     entry points pin the context with Context::Ref (one fetch_add + one load, wait-free); Exit unpublishes the
//     pointer, marks the context as retiring (which wakes the callers that block inside it: WaitIterations) and waits
//     for the pinned callers to leave before it deletes anything;
//   * GetOutput never takes a lock and never touches the device.  It reads one consistent iteration through a
//     sequence lock around the publish step (a few stores): a reader repeats only if a publish happened while it
//     copied its 32 bytes -- LOCK-FREE, not wait-free (a reader can in principle lose against every publish; publishes
//     are one per iteration, milliseconds apart).
#pragma once

#include <atomic>
#include <condition_variable>
#include <deque>
#include <cstdint>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pv_core.h"
#ifdef PVA_HOST_TEST
#include "fake_solver.h"  // tests/host/: same interface, no HIP
#else
#include "pv_solver.h"
#endif

namespace pva {

struct LiveConfig {
    float sizeX = 0, sizeY = 0;
    int res = 0;
    int boundaryType = 0;
    const char* tempDir = nullptr;
    int maxThreads = 0;
    int executionType = 0;
};

struct Out8 {
    float v[8];
};

class Context {
public:
    // Planeverb::Init / Exit (PvContext.cpp:25-44).  init() returns false (and leaves no context) where the
    // reference throws pv_InvalidConfig / pv_NotEnoughMemory.
    static bool init(const LiveConfig& cfg, std::string* err);
    static void exit();

    // Pins the current context for the duration of one API call (see the threading contract above).
    class Ref {
    public:
        Ref();
        ~Ref();
        Ref(const Ref&) = delete;
        Ref& operator=(const Ref&) = delete;
        explicit operator bool() const { return c_ != nullptr; }
        Context* operator->() const { return c_; }
        Context* get() const { return c_; }

    private:
        Context* c_;
    };

    // EmissionManager (Emissions/EmissionManager.cpp:37-75)
    int emit(float x, float y, float z);
    void updateEmission(int id, float x, float y, float z);
    void endEmission(int id);
    // Planeverb::GetOutput (FDTD.cpp:16-58)
    Out8 getOutput(int id);
    // the published result record of result cell (cx, cy) (what GetOutput reads after its position -> cell step)
    Out8 outputAt(int cx, int cy);

    // GeometryManager (Geometry/GeometryManager.cpp:67-152)
    int addGeometry(const Box& b);
    void updateGeometry(int id, const Box& b);
    void removeGeometry(int id);

    void setListener(float x, float y, float z);  // PvContext.cpp:50-56
    // Planeverb::GetImpulseResponse (FDTD.cpp:60-70): the IR of the last COMPLETED iteration at a world position as
    // reference Cells (16 B each).  Waits for the iteration in flight.  Returns the response length T (cells16 gets
    // min(cap, T) cells), 0 for a position outside the cell array, -1 on error.
    int impulseResponse(float x, float y, float z, void* cells16, int cap);

    long long iterations() const { return iterations_.load(std::memory_order_acquire); }
    // a number no other context of this process has had (a new context can land on a retired one's address)
    unsigned long long generation() const { return generation_; }
    long long waitIterations(long long count, int timeoutMs);
    const GridSpec& spec() const { return solver_->spec(); }
    // the worker stops for good on a solver error: then this is true, workerError() says why and IsRunning reports 0
    bool failed() const { return failed_.load(std::memory_order_acquire); }
    std::string workerError();
    // true when the solver runs in the sparse-emitter mode (the T-step history did not fit, or PLANEVERB_AMD_LIVE_STREAMING=1):
    // the worker registers the emitters alive at the start of every iteration, wet gain / RT60 exist for their cells
    bool streaming() const { return streaming_; }

private:
    Context() = default;
    ~Context();
    void workerMain();
    void workerLoop();
    // Pipelined iterations (small grids): TWO solvers, two iterations in flight.  Iteration i + 1 starts -- with the listener
    // latched and the geometry pushed at that moment, as in the reference's loop (PvContext.cpp:86-89) -- while iteration i is
    // still running on the other solver: every result still answers the listener its run started with one run time later,
    // but results arrive twice as often.  What an iteration leaves untouched (cells without an onset keep the previous
    // iteration's values, Analyzer.cpp:160-165) is carried from the other solver's maps on the device (Solver::run's
    // carryFrom), so the published records are those of the one-solver loop.
    void workerLoopPipelined();
    void pushGeometryChanges();
    void applyPending(int k);
    bool beginPublish();
    bool finishPublish();
    void publishSlot(int back, const Solver::WindowBlock& w);
    bool pendPublish_ = false;
    int pendBack_ = 0;
    Solver::WindowBlock pendWin_;
    void beginRetire();
    bool registerEmitters();
    friend void retireContext(Context*);

    unsigned long long generation_ = 0;
    Solver* solver_ = nullptr;
    Solver* solver2_ = nullptr;           // pipelined mode only
    Solver* lastSolver_ = nullptr;        // the solver that ran the last PUBLISHED iteration (GetImpulseResponse reads it)
    std::thread worker_;
    std::atomic<bool> running_{false};
    std::atomic<bool> retiring_{false};
    std::atomic<bool> failed_{false};
    bool streaming_ = false;
    std::atomic<long long> iterations_{0};
    std::mutex iterMutex_;
    std::condition_variable iterCv_;
    std::mutex errMutex_;
    std::string workerErr_;
    // the worker holds this while it uses the solver (an iteration + the geometry push); GetImpulseResponse takes it
    std::mutex solverMutex_;
    std::atomic<int> solverWaiters_{0};

    // listener (a plain vec3 in the reference, PvContext.h:40, where a reader can see x of one SetListenerPosition and
    // z of the next).  The simulation uses x and z only (world y is ignored, FDTD.cpp:97-98): both live in ONE 64-bit
    // atomic, so the worker always latches a position some caller actually set.
    std::atomic<uint64_t> lxz_{0};
    std::atomic<float> ly_{0.f};
    static uint64_t packXZ(float x, float z);
    static void unpackXZ(uint64_t v, float* x, float* z);

    // emitters: fixed-capacity chunks so readers never race a reallocation
    struct Emitter {
        std::atomic<float> x{0.f}, y{0.f}, z{0.f};
    };
    static constexpr int kChunk = 1024, kMaxChunks = 1024;
    std::atomic<Emitter*> chunks_[kMaxChunks] = {};
    std::atomic<int> emitterCount_{0};
    std::vector<int> emitterFree_;
    std::mutex emitMutex_;
    Emitter* emitterAt(int id);

    // geometry table + change queue (GeometryManager.h:28-46)
    struct Change {
        bool add;
        Box box;
    };
    std::vector<Box> geometry_;
    std::vector<int> geometryFree_;
    std::vector<Change> changes_;
    std::vector<Change> pending_[2];  // pipelined mode: drained from the queue, not yet rasterised into solver k
    std::mutex geomMutex_;

    // Published results.  Only the block of the map an iteration can have changed (the history window) crosses PCIe:
    // two pinned slots of windowCapacity() records, `front_` selects the readable one, `pubSeq_` is the sequence lock
    // of the publish step (odd while the worker flips front_ and the slot's block description).  Cells outside the
    // front block keep the values they had when they were last inside one -- `base_`, a full-size map allocated the
    // first time the window moves (all zeros before: a fresh context's pool, PvContext.cpp:132) -- except the listener
    // direction, which the reference recomputes for every cell on every iteration and which is the unit vector
    // listener -> cell there (Analyzer.cpp:365-391,415-428).
    struct Slot {
        float* data = nullptr;
        std::atomic<int> r0{0}, c0{0}, nr{0}, nc{0};
        std::atomic<float> lx{0.f}, lz{0.f};
    };
    Slot slots_[3];  // one readable + one per iteration in flight
    std::atomic<int> front_{-1};
    std::atomic<uint64_t> pubSeq_{0};
    std::atomic<float*> base_{nullptr};
};

}  // namespace pva
