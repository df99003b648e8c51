// pv_context.cpp -- see pv_context.h
#include "pv_context.h"

#include <chrono>
#include <cstdlib>
#include <cstring>

namespace pva {

static Context* g_context = nullptr;
static std::mutex g_contextMutex;

Context* Context::get() { return g_context; }

bool Context::init(const LiveConfig& cfg, std::string* err) {
    std::lock_guard<std::mutex> lock(g_contextMutex);
    if (g_context) {  // Init while running = Exit + Init, PvContext.cpp:27-31
        delete g_context;
        g_context = nullptr;
    }
    // PvContext.cpp:101-107
    if (cfg.res < kLowResolution || cfg.sizeX == 0.f || cfg.sizeY == 0.f || cfg.tempDir == nullptr ||
        cfg.maxThreads < 0) {
        if (err) *err = "invalid config (pv_InvalidConfig)";
        return false;
    }
    int device = 0;
    if (const char* e = std::getenv("PLANEVERB_AMD_DEVICE")) device = std::atoi(e);
    GridSpec spec = makeGridSpec(cfg.sizeX, cfg.sizeY, cfg.res);
    SolverOptions opt;
    Context* c = new Context();
    c->solver_ = Solver::create(spec, device, opt, err);
    if (!c->solver_) {
        delete c;
        return false;
    }
    const size_t bytes = (size_t)spec.gx * spec.gy * 32;
    for (int i = 0; i < 2; ++i) {
        if (hipHostMalloc((void**)&c->resHost_[i], bytes) != hipSuccess) {
            if (err) *err = "hipHostMalloc failed for the published result map";
            delete c;
            return false;
        }
        std::memset(c->resHost_[i], 0, bytes);  // fresh context: all-zero results, PvContext.cpp:132
    }
    c->running_.store(true);
    c->worker_ = std::thread(&Context::workerLoop, c);  // PvContext.cpp:160
    g_context = c;
    return true;
}

void Context::exit() {
    std::lock_guard<std::mutex> lock(g_contextMutex);
    if (g_context) {
        delete g_context;
        g_context = nullptr;
    }
}

Context::~Context() {
    running_.store(false);  // PvContext.cpp:166-167
    if (worker_.joinable()) worker_.join();
    delete solver_;
    for (float* p : resHost_)
        if (p) hipHostFree(p);
    for (auto& c : chunks_) {
        Emitter* p = c.load();
        delete[] p;
    }
}

// PvContext.cpp:63-94
void Context::workerLoop() {
    float lx = lx_.load(), ly = ly_.load(), lz = lz_.load();
    while (running_.load(std::memory_order_acquire)) {
        bool ok = solver_->run(lx, ly, lz, /*wait=*/false);
        const int back = front_.load() ^ 1;
        ok = ok && solver_->copyResultsAsync(resHost_[back]) && solver_->sync();
        if (!ok) {
            workerErr_ = solver_->lastError();
            running_.store(false);
            break;
        }
        front_.store(back, std::memory_order_release);
        published_.store(true, std::memory_order_release);
        {
            std::lock_guard<std::mutex> lock(iterMutex_);
            iterations_.fetch_add(1, std::memory_order_acq_rel);
        }
        iterCv_.notify_all();
        pushGeometryChanges();  // PvContext.cpp:86
        lx = lx_.load();        // PvContext.cpp:89
        ly = ly_.load();
        lz = lz_.load();
    }
    iterCv_.notify_all();
}

long long Context::waitIterations(long long count, int timeoutMs) {
    std::unique_lock<std::mutex> lock(iterMutex_);
    iterCv_.wait_for(lock, std::chrono::milliseconds(timeoutMs),
                     [&] { return iterations_.load() >= count || !running_.load(); });
    return iterations_.load();
}

void Context::setListener(float x, float y, float z) {
    lx_.store(x);
    ly_.store(y);
    lz_.store(z);
}

// ----------------------------------------------------------------------------------------------------------------
// emissions
// ----------------------------------------------------------------------------------------------------------------

Context::Emitter* Context::emitterAt(int id) {
    if (id < 0 || id >= emitterCount_.load(std::memory_order_acquire)) return nullptr;
    Emitter* c = chunks_[id / kChunk].load(std::memory_order_acquire);
    return c ? &c[id % kChunk] : nullptr;
}

int Context::emit(float x, float y, float z) {
    std::lock_guard<std::mutex> lock(emitMutex_);
    int id;
    if (!emitterFree_.empty()) {  // EmissionManager.cpp:40-46
        id = emitterFree_.back();
        emitterFree_.pop_back();
    } else {  // EmissionManager.cpp:48-53
        id = emitterCount_.load();
        if (id >= kChunk * kMaxChunks) return -1;
        if (!chunks_[id / kChunk].load()) chunks_[id / kChunk].store(new Emitter[kChunk], std::memory_order_release);
    }
    Emitter* e = &chunks_[id / kChunk].load()[id % kChunk];
    e->x.store(x);
    e->y.store(y);
    e->z.store(z);
    if (id == emitterCount_.load()) emitterCount_.store(id + 1, std::memory_order_release);
    return id;
}

void Context::updateEmission(int id, float x, float y, float z) {
    if (Emitter* e = emitterAt(id)) {  // EmissionManager.cpp:56-61
        e->x.store(x);
        e->y.store(y);
        e->z.store(z);
    }
}

void Context::endEmission(int id) {
    // EmissionManager.cpp:63-67 pushes any id unchecked (an out-of-range id later corrupts memory there);
    // ids that were never handed out are ignored here, a double end is kept (the id is then handed out twice).
    if (id < 0 || id >= emitterCount_.load()) return;
    std::lock_guard<std::mutex> lock(emitMutex_);
    emitterFree_.push_back(id);
}

Out8 Context::getOutput(int id) {
    Out8 o;
    std::memset(&o, 0, sizeof(o));
    Emitter* e = emitterAt(id);
    if (!e) {  // FDTD.cpp:34-38
        o.v[0] = kInvalidDryGain;
        return o;
    }
    int cx, cy;
    if (!resultCell(solver_->spec(), e->x.load(), e->z.load(), &cx, &cy)) {  // FDTD.cpp:43-47
        o.v[0] = kInvalidDryGain;
        return o;
    }
    const float* map = resHost_[front_.load(std::memory_order_acquire)];
    std::memcpy(o.v, map + 8 * ((size_t)cx * solver_->spec().gy + cy), 32);
    return o;
}

// ----------------------------------------------------------------------------------------------------------------
// geometry
// ----------------------------------------------------------------------------------------------------------------

int Context::addGeometry(const Box& b) {
    std::lock_guard<std::mutex> lock(geomMutex_);
    int id;
    if (geometryFree_.empty()) {  // GeometryManager.cpp:70-79
        id = (int)geometry_.size();
        geometry_.push_back(b);
    } else {  // :81-92
        id = geometryFree_.back();
        geometryFree_.pop_back();
        geometry_[(size_t)id] = b;
    }
    changes_.push_back({true, b});
    return id;
}

void Context::removeGeometry(int id) {
    std::lock_guard<std::mutex> lock(geomMutex_);
    if (id < 0 || id >= (int)geometry_.size()) return;
    changes_.push_back({false, geometry_[(size_t)id]});  // GeometryManager.cpp:101-110
    geometry_[(size_t)id] = Box{0, 0, 0, 0, 0};
    geometryFree_.push_back(id);
}

void Context::updateGeometry(int id, const Box& b) {
    std::lock_guard<std::mutex> lock(geomMutex_);
    if (id < 0 || id >= (int)geometry_.size()) return;
    changes_.push_back({false, geometry_[(size_t)id]});  // GeometryManager.cpp:112-121
    geometry_[(size_t)id] = b;
    changes_.push_back({true, b});
}

void Context::pushGeometryChanges() {
    std::vector<Change> q;
    {
        std::lock_guard<std::mutex> lock(geomMutex_);
        q.swap(changes_);
    }
    for (const Change& c : q) {  // GeometryManager.cpp:123-152, applied in queue order
        if (c.add)
            solver_->rasterAdd(c.box);
        else
            solver_->rasterRemove(c.box);
    }
}

}  // namespace pva
