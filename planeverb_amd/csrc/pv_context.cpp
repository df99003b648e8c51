// pv_context.cpp -- see pv_context.h
#include "pv_context.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace pva {

// ----------------------------------------------------------------------------------------------------------------
// lifetime: one published pointer, pinned by every API call for its duration
// ----------------------------------------------------------------------------------------------------------------

static std::atomic<Context*> g_context{nullptr};
static std::atomic<int> g_pinned{0};  // callers between Ref() and ~Ref() that hold a non-null context
static std::mutex g_lifeMutex;        // serialises Init / Exit

Context::Ref::Ref() {
    // Load, pin, re-check: once Exit has unpublished the pointer no new caller touches the counter, so the counter
    // only drains and Exit's wait ends.  A caller that loses the race between its two loads backs off without
    // having dereferenced anything.
    Context* c = g_context.load(std::memory_order_seq_cst);
    if (c) {
        g_pinned.fetch_add(1, std::memory_order_seq_cst);
        if (g_context.load(std::memory_order_seq_cst) != c) {
            g_pinned.fetch_sub(1, std::memory_order_seq_cst);
            c = nullptr;
        }
    }
    c_ = c;
}

Context::Ref::~Ref() {
    if (c_) g_pinned.fetch_sub(1, std::memory_order_seq_cst);
}

void retireContext(Context* c) {
    if (!c) return;
    // wake the callers that block INSIDE the context (WaitIterations) and let the worker finish its iteration, so that
    // the pins drain at once instead of after the callers' own timeouts.  The wake-up is repeated while pins remain: a
    // waiter that tested its predicate just before retiring_ was set and blocks just after the first notify is caught by
    // the next one (no mutex is taken here, so nothing of the context is held when it is deleted).
    c->beginRetire();
    while (g_pinned.load(std::memory_order_seq_cst) != 0) {  // callers still inside
        c->iterCv_.notify_all();
        std::this_thread::yield();
    }
    delete c;
}

void Context::beginRetire() {
    running_.store(false);
    retiring_.store(true, std::memory_order_seq_cst);
    iterCv_.notify_all();
}

uint64_t Context::packXZ(float x, float z) {
    uint32_t a, b;
    std::memcpy(&a, &x, 4);
    std::memcpy(&b, &z, 4);
    return ((uint64_t)a << 32) | b;
}
void Context::unpackXZ(uint64_t v, float* x, float* z) {
    const uint32_t a = (uint32_t)(v >> 32), b = (uint32_t)v;
    std::memcpy(x, &a, 4);
    std::memcpy(z, &b, 4);
}

bool Context::init(const LiveConfig& cfg, std::string* err) {
    std::lock_guard<std::mutex> lock(g_lifeMutex);
    // Init while running = Exit + Init, PvContext.cpp:27-31
    retireContext(g_context.exchange(nullptr, std::memory_order_seq_cst));
    // PvContext.cpp:101-107
    if (cfg.res < kLowResolution || cfg.sizeX == 0.f || cfg.sizeY == 0.f || cfg.tempDir == nullptr ||
        cfg.maxThreads < 0) {
        if (err) *err = "invalid config (pv_InvalidConfig)";
        return false;
    }
    int device = 0;
    if (const char* e = std::getenv("PLANEVERB_AMD_DEVICE")) device = std::atoi(e);
    GridSpec spec = makeGridSpec(cfg.sizeX, cfg.sizeY, cfg.res);
    if (spec.gx != spec.gy)
        std::fprintf(stderr, "[planeverb_amd] warning: %d x %d grid -- the reference indexes non-square grids "
                             "inconsistently (SURVEY.md Q1); results there are defined by this library (cell array "
                             "stride gy+1 throughout), not by the reference\n", spec.gx, spec.gy);
    // Planeverb::Init takes any resolution >= 275 (PvContext.cpp:101-107).  Where the T-step pressure history does not fit
    // the device (25 m at 16 kHz: T = 25 432 on a 4096^2 grid) the solver falls back to the sparse-emitter mode by itself;
    // PLANEVERB_AMD_LIVE_STREAMING=1 forces that mode (tests, memory-tight hosts).
    SolverOptions opt;
    opt.autoStreaming = true;
    if (const char* e = std::getenv("PLANEVERB_AMD_LIVE_STREAMING")) opt.streaming = std::atoi(e) != 0;
    // (owned until it is published: an exception from underneath -- a table or the worker thread that cannot be created --
    // unwinds through ~Context, which stops the worker and releases the solvers, and ends at the C-ABI's barrier, pv_capi.cpp)
    struct Owner {
        Context* c;
        ~Owner() { delete c; }
    } own{new Context()};
    Context* const c = own.c;
    static std::atomic<unsigned long long> generations{0};
    c->generation_ = generations.fetch_add(1) + 1;
    c->solver_ = Solver::create(spec, device, opt, err);
    if (!c->solver_) return false;
    c->streaming_ = c->solver_->options().streaming;
    c->lastSolver_ = c->solver_;
    // Two iterations in flight where an iteration leaves most of the chip idle: the grids the resident kernel serves (the
    // reference's presets).  PLANEVERB_AMD_LIVE_PIPELINE=1 / 2 forces one / two.
    int pipeline = (!c->streaming_ && c->solver_->residentKernel()) ? 2 : 1;
    if (const char* e = std::getenv("PLANEVERB_AMD_LIVE_PIPELINE")) pipeline = std::atoi(e) >= 2 && !c->streaming_ ? 2 : 1;
    if (pipeline == 2) {
        c->solver2_ = Solver::create(spec, device, opt, err);
        if (!c->solver2_) return false;
    }
    if (c->streaming_)
        std::fprintf(stderr, "[planeverb_amd] %d x %d grid, T = %d: sparse-emitter mode (wet gain / RT60 for the cells of the "
                             "emitters registered at the start of an iteration)\n", spec.gx, spec.gy, c->solver_->T());
    const size_t bytes = c->solver_->windowCapacity() * 32;
    for (Slot& s : c->slots_) {
        s.data = static_cast<float*>(Solver::hostAlloc(bytes));
        if (!s.data) {
            if (err) *err = "pinned host allocation failed for the published result block";
            return false;
        }
    }
    c->running_.store(true);
    c->worker_ = std::thread(&Context::workerMain, c);  // PvContext.cpp:160
    own.c = nullptr;
    g_context.store(c, std::memory_order_seq_cst);
    return true;
}

void Context::exit() {
    std::lock_guard<std::mutex> lock(g_lifeMutex);
    retireContext(g_context.exchange(nullptr, std::memory_order_seq_cst));
}

Context::~Context() {
    running_.store(false);  // PvContext.cpp:166-167
    if (worker_.joinable()) worker_.join();
    delete solver2_;
    delete solver_;
    for (Slot& s : slots_) Solver::hostFree(s.data);
    std::free(base_.load());
    for (auto& c : chunks_) {
        Emitter* p = c.load();
        delete[] p;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// worker: PvContext.cpp:63-94
// ----------------------------------------------------------------------------------------------------------------

// sparse-emitter mode: the emitters alive now are the cells whose wet gain / RT60 this iteration computes
// (EmissionManager.cpp:37-54 keeps the same table; ended ids are skipped)
bool Context::registerEmitters() {
    std::vector<float> xyz;
    {
        std::lock_guard<std::mutex> lock(emitMutex_);
        const int n = emitterCount_.load(std::memory_order_acquire);
        std::vector<uint8_t> ended((size_t)n, 0);
        for (int id : emitterFree_)
            if (id >= 0 && id < n) ended[(size_t)id] = 1;
        for (int id = 0; id < n; ++id) {
            if (ended[(size_t)id]) continue;
            Emitter* e = emitterAt(id);
            if (!e) continue;
            xyz.push_back(e->x.load());
            xyz.push_back(e->y.load());
            xyz.push_back(e->z.load());
        }
    }
    return solver_->setEmitters(xyz.data(), (int)(xyz.size() / 3));
}

// The thread's entry: no exception leaves the worker (std::terminate would take the host process -- the game -- with it).
// One that reaches this point stops the worker like a solver error does: IsRunning reports 0, PvAmdLastError /
// PlaneverbWorkerError carry the reason, GetOutput keeps serving the last published iteration.
void Context::workerMain() {
    try {
        if (solver2_)
            workerLoopPipelined();
        else
            workerLoop();
        return;
    } catch (const std::exception& e) {
        try {
            std::lock_guard<std::mutex> lock(errMutex_);
            workerErr_ = std::string("exception in the simulation worker: ") + e.what();
        } catch (...) {
        }
    } catch (...) {
        try {
            std::lock_guard<std::mutex> lock(errMutex_);
            workerErr_ = "unknown exception in the simulation worker";
        } catch (...) {
        }
    }
    std::fprintf(stderr, "[planeverb_amd] simulation worker stopped on an exception\n");
    failed_.store(true, std::memory_order_release);
    running_.store(false);
    iterCv_.notify_all();
}

void Context::workerLoop() {
    float lx, lz, ly = ly_.load();
    unpackXZ(lxz_.load(), &lx, &lz);
    std::unique_lock<std::mutex> solverLock(solverMutex_);
    auto stop = [&]() {
        finishPublish();  // (the last COMPLETED iteration's block may still be on its way: it is valid, make it visible)
        // The worker stops; the host can see it: IsRunning reports 0, PvAmdLastError carries the reason, GetOutput keeps
        // serving the last published iteration.
        std::string e = solver_->lastError();
        std::fprintf(stderr, "[planeverb_amd] simulation worker stopped: %s\n", e.c_str());
        {
            std::lock_guard<std::mutex> lock(errMutex_);
            workerErr_ = std::move(e);
        }
        failed_.store(true, std::memory_order_release);
        running_.store(false);
    };
    while (running_.load(std::memory_order_acquire)) {
        // Iteration i: enqueue FDTD + analysis; make iteration i - 1 visible (its result block has been travelling to the
        // host on the copy stream since it was packed); queue this iteration's block behind its analysis; wait for the
        // DEVICE work of iteration i -- not for its copy, which overlaps the next iteration's first launches.
        // (sparse-emitter mode: enqueueing a run BLOCKS for the front phase of the run -- the host paces the fused forward sums
        // -- so the previous iteration is made visible first instead of ~100 ms late)
        if (!((!streaming_ || (finishPublish() && registerEmitters())) && solver_->run(lx, ly, lz, /*wait=*/false) && finishPublish() &&
              beginPublish() && solver_->sync())) {
            stop();
            break;
        }
        pushGeometryChanges();            // PvContext.cpp:86: after the iteration's analysis
        unpackXZ(lxz_.load(), &lx, &lz);  // PvContext.cpp:89
        ly = ly_.load();
        if (solverWaiters_.load(std::memory_order_acquire) > 0) {  // a GetImpulseResponse call wants the solver
            // (it reads the iteration just completed: publish it first, so that "the last completed iteration" and the
            // published outputs agree)
            if (!finishPublish()) {
                stop();
                break;
            }
            solverLock.unlock();
            while (solverWaiters_.load(std::memory_order_acquire) > 0) std::this_thread::yield();
            solverLock.lock();
        }
    }
    if (!failed_.load()) finishPublish();  // the last iteration's block
    solverLock.unlock();
    iterCv_.notify_all();
}

void Context::workerLoopPipelined() {
    float lx, lz, ly = ly_.load();
    unpackXZ(lxz_.load(), &lx, &lz);
    std::unique_lock<std::mutex> solverLock(solverMutex_);
    Solver* const sv[2] = {solver_, solver2_};
    struct Flight {
        int k, slot;
        Solver::WindowBlock win;
    };
    std::deque<Flight> flights;
    auto fail = [&](Solver* s) {
        std::string e = s->lastError();
        std::fprintf(stderr, "[planeverb_amd] simulation worker stopped: %s\n", e.c_str());
        {
            std::lock_guard<std::mutex> lock(errMutex_);
            workerErr_ = std::move(e);
        }
        failed_.store(true, std::memory_order_release);
        running_.store(false);
    };
    // the oldest iteration in flight: wait for its device work and its block's copy, make it visible
    auto complete = [&]() -> bool {
        const Flight f = flights.front();
        flights.pop_front();
        if (!(sv[f.k]->sync() && sv[f.k]->waitPublish())) {
            fail(sv[f.k]);
            return false;
        }
        publishSlot(f.slot, f.win);
        lastSolver_ = sv[f.k];
        pushGeometryChanges();            // PvContext.cpp:86: after an iteration's analysis
        unpackXZ(lxz_.load(), &lx, &lz);  // PvContext.cpp:89
        ly = ly_.load();
        return true;
    };
    long long started = 0;
    int prevK = -1;
    bool ok = true;
    while (ok && running_.load(std::memory_order_acquire)) {
        const int k = (int)(started & 1);
        while (ok && !flights.empty() && (flights.size() >= 2 || flights.front().k == k)) ok = complete();
        if (!ok) break;
        applyPending(k);
        // a slot that is neither readable nor the target of an iteration in flight
        int slot = 0;
        for (; slot < 3; ++slot) {
            bool used = slot == front_.load(std::memory_order_relaxed);
            for (const Flight& f : flights) used = used || f.slot == slot;
            if (!used) break;
        }
        Flight f{k, slot, {}};
        if (!(sv[k]->run(lx, ly, lz, /*wait=*/false, prevK >= 0 ? sv[prevK] : nullptr) &&
              sv[k]->publishWindowAsync(slots_[slot].data, &f.win, false))) {
            // (the iteration still in flight on the other solver is valid: make it visible first, as the one-solver loop does)
            while (!flights.empty() && complete()) {
            }
            if (!failed_.load()) fail(sv[k]);
            break;
        }
        flights.push_back(f);
        prevK = k;
        // (the first iteration runs alone: the reference pushes the geometry its caller added after Init behind it,
        // PvContext.cpp:80-86)
        if (started++ == 0) ok = complete();
        if (ok && solverWaiters_.load(std::memory_order_acquire) > 0) {  // a GetImpulseResponse call wants the solvers
            while (ok && !flights.empty()) ok = complete();
            if (!ok) break;
            solverLock.unlock();
            while (solverWaiters_.load(std::memory_order_acquire) > 0) std::this_thread::yield();
            solverLock.lock();
        }
    }
    while (!failed_.load() && !flights.empty() && complete()) {
    }
    solverLock.unlock();
    iterCv_.notify_all();
}

std::string Context::workerError() {
    std::lock_guard<std::mutex> lock(errMutex_);
    return workerErr_;
}

long long Context::waitIterations(long long count, int timeoutMs) {
    std::unique_lock<std::mutex> lock(iterMutex_);
    // (system_clock: pthread_cond_timedwait, which ThreadSanitizer models; wait_for's pthread_cond_clockwait it does not)
    iterCv_.wait_until(lock, std::chrono::system_clock::now() + std::chrono::milliseconds(timeoutMs),
                     [&] { return iterations_.load() >= count || !running_.load() || retiring_.load(); });
    return iterations_.load();
}

void Context::setListener(float x, float y, float z) {
    lxz_.store(packXZ(x, z));
    ly_.store(y);
}

// ----------------------------------------------------------------------------------------------------------------
// publish / read
// ----------------------------------------------------------------------------------------------------------------

namespace {
// 32-bit relaxed atomic accesses: readers may overlap a writer (they then discard what they read, see pubSeq_)
inline void storeRelaxed(float* dst, const float* src, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t v;
        std::memcpy(&v, src + i, 4);
        __atomic_store_n(reinterpret_cast<uint32_t*>(dst) + i, v, __ATOMIC_RELAXED);
    }
}
inline void loadRelaxed(float* dst, const float* src, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        const uint32_t v = __atomic_load_n(reinterpret_cast<const uint32_t*>(src) + i, __ATOMIC_RELAXED);
        std::memcpy(dst + i, &v, 4);
    }
}
}  // namespace

// Copy the block the finished iteration can have changed into the back slot, keep what leaves the window, flip.
// Two halves, so that the device -> host copy of iteration i overlaps iteration i + 1's first launches: beginPublish is
// enqueued right behind iteration i's analysis (pack on the solver's stream, copy on a second stream into the BACK slot);
// finishPublish -- called once iteration i + 1 has been enqueued -- waits for that copy alone, does the host bookkeeping and
// flips the slots.  Readers only ever see the front slot, which no copy is writing.
bool Context::beginPublish() {
    const int f = front_.load(std::memory_order_relaxed);
    pendBack_ = f < 0 ? 0 : f ^ 1;
    // (small blocks -- the Sandbox's 157 kB -- ride on the solver's own stream: the cross-stream hand-over costs more than
    // their copy, measured 0.77 against 0.71 ms per iteration at 71^2)
    const bool overlap = solver_->windowCapacity() * 32 > (size_t)(1u << 20);
    if (!solver_->publishWindowAsync(slots_[pendBack_].data, &pendWin_, overlap)) return false;
    pendPublish_ = true;
    return true;
}

bool Context::finishPublish() {
    if (!pendPublish_) return true;
    pendPublish_ = false;
    if (!solver_->waitPublish()) return false;
    publishSlot(pendBack_, pendWin_);
    return true;
}

// host part of a publish: the block has landed in slots_[back]
void Context::publishSlot(const int back, const Solver::WindowBlock& w) {
    const int f = front_.load(std::memory_order_relaxed);
    if (f >= 0) {
        // Cells of the old block outside the new one keep the old block's values from now on (the reference leaves
        // m_results untouched where an iteration finds no onset, Analyzer.cpp:160-165).  Nobody can be reading these
        // cells from base_ now: a reader that started after the last publish takes them from the front slot, and one
        // that started before it repeats.
        const Slot& o = slots_[f];
        const int or0 = o.r0.load(std::memory_order_relaxed), oc0 = o.c0.load(std::memory_order_relaxed);
        const int onr = o.nr.load(std::memory_order_relaxed), onc = o.nc.load(std::memory_order_relaxed);
        const bool same = or0 == w.r0 && oc0 == w.c0 && onr == w.nr && onc == w.nc;
        if (!same && onr > 0 && onc > 0) {
            const int gy = solver_->spec().gy;
            float* b = base_.load(std::memory_order_relaxed);
            if (!b) {
                b = static_cast<float*>(std::calloc((size_t)solver_->spec().gx * gy, 32));
                if (!b) std::abort();  // (a map of 32 bytes per cell on the host)
                base_.store(b, std::memory_order_release);
            }
            for (int r = or0; r < or0 + onr; ++r) {
                const bool rowInside = r >= w.r0 && r < w.r0 + w.nr;
                for (int c = oc0; c < oc0 + onc;) {
                    if (rowInside && c >= w.c0 && c < w.c0 + w.nc) {
                        c = w.c0 + w.nc;  // the part of the row the new block covers
                        continue;
                    }
                    const int end = (rowInside && c < w.c0) ? std::min(oc0 + onc, w.c0) : oc0 + onc;
                    storeRelaxed(b + ((size_t)r * gy + c) * 8, o.data + ((size_t)(r - or0) * onc + (c - oc0)) * 8,
                                 (size_t)(end - c) * 8);
                    c = end;
                }
            }
        }
    }
    pubSeq_.fetch_add(1, std::memory_order_seq_cst);  // odd: publish in progress
    Slot& s = slots_[back];
    s.r0.store(w.r0, std::memory_order_relaxed);
    s.c0.store(w.c0, std::memory_order_relaxed);
    s.nr.store(w.nr, std::memory_order_relaxed);
    s.nc.store(w.nc, std::memory_order_relaxed);
    s.lx.store(w.lx, std::memory_order_relaxed);
    s.lz.store(w.lz, std::memory_order_relaxed);
    front_.store(back, std::memory_order_relaxed);
    pubSeq_.fetch_add(1, std::memory_order_seq_cst);  // even
    {
        std::lock_guard<std::mutex> lock(iterMutex_);
        iterations_.fetch_add(1, std::memory_order_acq_rel);
    }
    iterCv_.notify_all();
}

Out8 Context::outputAt(int cx, int cy) {
    const GridSpec& g = solver_->spec();
    for (;;) {
        const uint64_t s1 = pubSeq_.load(std::memory_order_acquire);
        if (s1 & 1) continue;  // the worker is between its two stores
        Out8 o;
        std::memset(&o, 0, sizeof(o));  // nothing published yet: the zeroed pool of a fresh context, PvContext.cpp:132
        const int f = front_.load(std::memory_order_relaxed);
        if (f >= 0) {
            const Slot& s = slots_[f];
            const int r0 = s.r0.load(std::memory_order_relaxed), c0 = s.c0.load(std::memory_order_relaxed);
            const int nr = s.nr.load(std::memory_order_relaxed), nc = s.nc.load(std::memory_order_relaxed);
            if (cx >= r0 && cx < r0 + nr && cy >= c0 && cy < c0 + nc) {
                loadRelaxed(o.v, s.data + ((size_t)(cx - r0) * nc + (cy - c0)) * 8, 8);
            } else {
                if (const float* b = base_.load(std::memory_order_acquire))
                    loadRelaxed(o.v, b + ((size_t)cx * g.gy + cy) * 8, 8);
                // no neighbour of a cell out here has an onset, so the listener-direction walk stays put
                // (Analyzer.cpp:365-391) and the direction is the normalised (cellPos - listener), :415-428
                float ox = (float)cx * g.dx - s.lx.load(std::memory_order_relaxed);
                float oy = (float)cy * g.dx - s.lz.load(std::memory_order_relaxed);
                float len = (ox * ox) + (oy * oy);
                if (len != 0.f) {
                    len = std::sqrt(len);
                    ox /= len;
                    oy /= len;
                }
                o.v[4] = ox;
                o.v[5] = oy;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (pubSeq_.load(std::memory_order_relaxed) == s1) return o;
    }
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
    return outputAt(cx, cy);
}

// FDTD.cpp:60-79: cell = ((int)(x / dx), (int)(z / dx)) of the (gx+1) x (gy+1) array
int Context::impulseResponse(float x, float y, float z, void* cells16, int cap) {
    (void)y;
    const GridSpec& g = solver_->spec();
    const int cx = (int)(x / g.dx), cy = (int)(z / g.dx);
    if (cx < 0 || cx > g.gx || cy < 0 || cy > g.gy) return 0;  // (the reference indexes past its array here)
    solverWaiters_.fetch_add(1, std::memory_order_acq_rel);
    std::unique_lock<std::mutex> lock(solverMutex_);
    solverWaiters_.fetch_sub(1, std::memory_order_acq_rel);
    const int T = solver_->T();
    std::vector<char> buf((size_t)T * 16);
    if (!lastSolver_->impulseResponseCells(cx, cy, buf.data())) return -1;
    lock.unlock();
    if (cells16 && cap > 0) std::memcpy(cells16, buf.data(), (size_t)std::min(cap, T) * 16);
    return T;
}

// ----------------------------------------------------------------------------------------------------------------
// geometry
// ----------------------------------------------------------------------------------------------------------------

namespace {
// Room for k more elements BEFORE a table operation commits anything: the only step that can throw (std::bad_alloc) then
// happens while the tables are still what they were, and the push_backs behind it cannot fail (geometric growth kept).
template <class V>
void ensureRoom(V& v, size_t k) {
    if (v.capacity() - v.size() < k) v.reserve(std::max(v.capacity() * 2, v.size() + k));
}
}  // namespace

int Context::addGeometry(const Box& b) {
    // (a non-finite absorption is refused: NaN marks air in the material and coefficient planes, pv_solver.cpp addBox)
    if (!std::isfinite(b.R)) return -1;
    std::lock_guard<std::mutex> lock(geomMutex_);
    int id;
    ensureRoom(changes_, 1);
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
    ensureRoom(changes_, 1);
    ensureRoom(geometryFree_, 1);
    changes_.push_back({false, geometry_[(size_t)id]});  // GeometryManager.cpp:101-110
    geometry_[(size_t)id] = Box{0, 0, 0, 0, 0};
    geometryFree_.push_back(id);
}

void Context::updateGeometry(int id, const Box& b) {
    if (!std::isfinite(b.R)) return;
    std::lock_guard<std::mutex> lock(geomMutex_);
    if (id < 0 || id >= (int)geometry_.size()) return;
    ensureRoom(changes_, 2);
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
    if (solver2_) {  // pipelined: every solver rasterises every change, in queue order, before ITS next iteration
        for (auto& pend : pending_) pend.insert(pend.end(), q.begin(), q.end());
        return;
    }
    for (const Change& c : q) {  // GeometryManager.cpp:123-152, applied in queue order
        if (c.add)
            solver_->rasterAdd(c.box);
        else
            solver_->rasterRemove(c.box);
    }
}

void Context::applyPending(const int k) {
    Solver* s = k == 0 ? solver_ : solver2_;
    for (const Change& c : pending_[k]) {
        if (c.add)
            s->rasterAdd(c.box);
        else
            s->rasterRemove(c.box);
    }
    pending_[k].clear();
}

}  // namespace pva
