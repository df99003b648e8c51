// pv_context.h -- the live acoustics module behind the reference's C-ABI (PlaneverbInit ... PlaneverbExit).
//
// Mirrors ProjectPlaneverb/src/Context/PvContext.{h,cpp}: a singleton that owns the grid, the geometry and
// emission tables, and one library-owned worker that loops
//     GenerateResponse(listener) -> AnalyzeResponses(listener) -> PushGeometryChanges() -> latch listener
// (PvContext.cpp:63-94).  Here the worker owns a Solver (one HIP stream on one MI355X) instead of running the
// sweeps itself, and publishes each finished result map into a pinned host double buffer so that GetOutput stays
// O(1) and never touches the device (the reference reads the result grid unsynchronised while it is rewritten).
#pragma once

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pv_core.h"
#include "pv_solver.h"

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
    static Context* get();

    // EmissionManager (Emissions/EmissionManager.cpp:37-75)
    int emit(float x, float y, float z);
    void updateEmission(int id, float x, float y, float z);
    void endEmission(int id);
    // Planeverb::GetOutput (FDTD.cpp:16-58)
    Out8 getOutput(int id);

    // GeometryManager (Geometry/GeometryManager.cpp:67-152)
    int addGeometry(const Box& b);
    void updateGeometry(int id, const Box& b);
    void removeGeometry(int id);

    void setListener(float x, float y, float z);  // PvContext.cpp:50-56
    long long iterations() const { return iterations_.load(std::memory_order_acquire); }
    long long waitIterations(long long count, int timeoutMs);
    const GridSpec& spec() const { return solver_->spec(); }
    const std::string& workerError() const { return workerErr_; }

private:
    Context() = default;
    ~Context();
    void workerLoop();
    void pushGeometryChanges();

    Solver* solver_ = nullptr;
    std::thread worker_;
    std::atomic<bool> running_{false};
    std::atomic<long long> iterations_{0};
    std::mutex iterMutex_;
    std::condition_variable iterCv_;
    std::string workerErr_;

    // listener (plain fields in the reference, PvContext.h:40)
    std::atomic<float> lx_{0.f}, ly_{0.f}, lz_{0.f};

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
    std::mutex geomMutex_;

    // published results: two pinned host maps, front_ selects the readable one
    float* resHost_[2] = {nullptr, nullptr};
    std::atomic<int> front_{0};
    std::atomic<bool> published_{false};
};

}  // namespace pva
