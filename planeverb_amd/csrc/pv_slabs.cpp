// pv_slabs.cpp -- see pv_slabs.h
#include "pv_slabs.h"

#include <cstdio>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstring>
#include <memory>

#include "pv_launch.h"

namespace pva {

namespace {
inline int ceilDiv(int a, int b) { return (a + b - 1) / b; }
}  // namespace

bool SlabGroup::fail(const std::string& what) {
    err_ = what;
    return false;
}

bool SlabGroup::hipOk(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    err_ = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

bool SlabGroup::slabFailed(int s) {
    err_ = "slab " + std::to_string(s) + ": " + slabs_[(size_t)s]->lastError();
    return false;
}

SlabGroup* SlabGroup::create(const GridSpec& spec, const std::vector<int>& devices, const SolverOptions& opt,
                             std::string* err) {
    std::unique_ptr<SlabGroup> g(new SlabGroup());  // (owned across init: see Solver::create)
    if (!g->init(spec, devices, opt)) {
        if (err) *err = g->err_;
        return nullptr;
    }
    return g.release();
}

bool SlabGroup::init(const GridSpec& spec, const std::vector<int>& devices, const SolverOptions& opt) {
    g_ = spec;
    devices_ = devices;
    const int S = (int)devices.size();
    if (S < 2) return fail("a slab group needs at least two slabs");
    for (int s = 0; s < S; ++s) {
        SolverOptions o = opt;
        o.slabIndex = s;
        o.slabCount = S;
        std::string e;
        Solver* sv = Solver::create(spec, devices[(size_t)s], o, &e);
        if (!sv) return fail("slab " + std::to_string(s) + ": " + e);
        slabs_.push_back(sv);
    }
    const Solver& a = *slabs_[0];
    K_ = a.K_;
    rxi_ = a.rxi_;
    wi_ = a.wi_;
    T_ = a.T_;
    for (const Solver* sv : slabs_)
        if (sv->geo_.pitch != a.geo_.pitch || sv->geo_.G != a.geo_.G || sv->histPitch_ != a.histPitch_)
            return fail("slabs disagree on the plane geometry");
    // peer access between the devices of adjacent slabs (halo rows) and of every slab with the root (result blocks)
    for (int s = 0; s < S; ++s)
        for (int t : {s - 1, s + 1, 0}) {
            if (t < 0 || t >= S || devices_[(size_t)t] == devices_[(size_t)s]) continue;
            hipSetDevice(devices_[(size_t)s]);
            const hipError_t e = hipDeviceEnablePeerAccess(devices_[(size_t)t], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                (void)hipGetLastError();  // copies then stage
                if (t != 0 || t == s - 1 || t == s + 1) pushHalos_ = false;  // no peer stores into that neighbour's guard band
            }
        }
    rootDevice_ = devices_[0];
    if (!hipOk(hipSetDevice(rootDevice_), "hipSetDevice")) return false;
    {
        // The root stream WAITS for the slabs' events: on a hardware queue of one of theirs it would park that slab's launches.
        // Even slabs run on normal-priority streams, odd ones on high-priority streams (Solver::init); the root is claimed apart
        // from the even ones.  (PLANEVERB_AMD_SLAB_ROOT_PRIORITY=1 puts it on a low-priority stream, a third pool of queues:
        // measured 15 % slower at 4096^2, S = 2, whenever another solver had lived in the process -- profiles/r05_slabs.txt.)
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        const char* er = getenv("PLANEVERB_AMD_SLAB_ROOT_PRIORITY");
        if (!(er && atoi(er) == 1)) lo = 0;
        if (!hipOk(hipStreamCreateWithPriority(&rootStream_, hipStreamNonBlocking, lo), "hipStreamCreate")) return false;
        if (!rootQueue_.claim(rootDevice_, &rootStream_, lo != 0 ? QueueClaim::kLow : QueueClaim::kNormal, this)) return fail("hipStreamCreate");
    }
    for (auto& e : rootEv_)
        if (!hipOk(hipEventCreate(&e), "hipEventCreate")) return false;
    stepEv_.resize((size_t)2 * S);
    miscEv_.resize((size_t)S);
    for (int s = 0; s < S; ++s) {
        hipSetDevice(devices_[(size_t)s]);
        for (int k = 0; k < 2; ++k)
            if (!hipOk(hipEventCreateWithFlags(&stepEv_[(size_t)2 * s + k], hipEventDisableTiming), "hipEventCreate"))
                return false;
        if (!hipOk(hipEventCreateWithFlags(&miscEv_[(size_t)s], hipEventDisableTiming), "hipEventCreate")) return false;
    }
    hipSetDevice(rootDevice_);
    {
        // hand-off words for neighbours that share a device (PLANEVERB_AMD_SLAB_HANDOFF=0: events everywhere, as in round 3)
        bool any = false;
        for (int s = 0; s + 1 < S; ++s) any = any || devices_[(size_t)s] == devices_[(size_t)s + 1];
        bool oneDev = true;  // (the words live on the root's device: groups that span devices keep their events)
        for (int s = 0; s < S; ++s) oneDev = oneDev && devices_[(size_t)s] == rootDevice_;
        // A push kernel WAITS for its neighbours' pushes: every slab's stream needs a hardware queue to itself -- the runtime
        // multiplexes a process's streams on GPU_MAX_HW_QUEUES (default 4) of them by creation order, and a waiting kernel
        // parks whatever sits behind it in its queue (the neighbour's launches, if they share it: the wait then ends in its
        // time-out).  So: only if the streams of each priority (even slabs and the root: normal, odd slabs: high) are no more than
        // that number, after QueueClaim has dealt them apart, and after a dry run of the hand-off on this group's own streams has
        // come through (probeHandoff).
        const char* e = getenv("PLANEVERB_AMD_SLAB_HANDOFF");
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        const int queues = q && atoi(q) > 0 ? atoi(q) : 4;
        int prLo = 0, prHi = 0;
        hipDeviceGetStreamPriorityRange(&prLo, &prHi);
        const char* er = getenv("PLANEVERB_AMD_SLAB_ROOT_PRIORITY");
        const bool lowIsNormal = prLo == 0 || !(er && atoi(er) == 1);  // (no low priority on this device: the root shares the even slabs' pool)
        const char* ep = getenv("PLANEVERB_AMD_SLAB_PRIORITY");
        const int normalStreams = ((ep && atoi(ep) == 0) ? S : (S + 1) / 2) + (lowIsNormal ? 1 : 0);
        const bool forced = e && atoi(e) == 2;  // (2: whatever the queue count says -- the stress tests of the fallback)
        if (any && oneDev && pushHalos_ && !(e && atoi(e) == 0) && (normalStreams <= queues || forced)) {
            if (!hipOk(hipMalloc((void**)&handoff_, sizeof(unsigned) * (3 * (size_t)S + 1)), "hipMalloc")) return false;
            // (round 6: a dry run that times out is answered by other streams for every slab, up to three times, before the group falls
            // back to stream events -- one creation in ten came up that way at 5.6 instead of 2.2 ms per run at 2048^2, always behind
            // other groups of the same process: profiles/r05_slabs.txt)
            // ... and so is a dry run that comes through SLOWLY: sixteen sweeps of push kernels that move no rows take ~4 us per slab and
            // sweep where the slabs' streams run beside each other, and 2-4 x that where they only take turns -- the group then runs at
            // 5.0-7.3 instead of 2.6-6.2 ms (profiles/r06_slabs.txt; the pairwise sleeper / stamp probe of QueueClaim sees nothing there)
            bool ok = forced || probeHandoff();
            const float slowUs = 6.5f * (float)S;
            const char* fr = getenv("PLANEVERB_AMD_SLAB_REDEAL");  // (tests: 1 = one re-deal whatever the dry run said)
            const bool forceRedeal = fr && atoi(fr) == 1;
            for (int attempt = 0; !forced && (!ok || probeUsPerSweep_ > slowUs || (forceRedeal && attempt == 0)) && attempt < 3; ++attempt) {
                ++redeals_;
                bool dealt = true;
                for (Solver* sv : slabs_) {
                    hipSetDevice(sv->device_);
                    dealt = sv->redealMainStream() && dealt;
                }
                hipSetDevice(rootDevice_);
                if (!dealt) break;
                ok = probeHandoff();
                const char* dbg = getenv("PLANEVERB_AMD_QUEUE_PROBE");
                if (dbg && atoi(dbg) >= 2)
                    std::fprintf(stderr, "[planeverb_amd] slab group %p: hand-off dry run after re-deal %d: %s, %.1f us per sweep\n", (void*)this, attempt + 1, ok ? "ok" : "timed out", probeUsPerSweep_);
            }
            if (!ok) {
                hipFree(handoff_);
                handoff_ = nullptr;
            }
        }
        const char* dbg = getenv("PLANEVERB_AMD_QUEUE_PROBE");
        if (dbg && atoi(dbg) >= 2)
            std::fprintf(stderr, "[planeverb_amd] slab group %p: %d slabs, hand-off words %s (dry run: %.1f us per sweep)\n", (void*)this, S,
                         handoff_ ? "on" : (any && oneDev ? "OFF (stream events)" : "not applicable"), probeUsPerSweep_);
    }
    const size_t n = (size_t)g_.gx * g_.gy;
    winRows_ = a.histTilesXG_ * rxi_;
    winCols_ = a.histTilesY_ * wi_;
    if (!hipOk(hipMalloc((void**)&res_, n * 8 * 4), "hipMalloc") || !hipOk(hipMalloc((void**)&delay_, n * 4), "hipMalloc") ||
        !hipOk(hipMalloc((void**)&dirScratch_, (size_t)winRows_ * winCols_ * 4), "hipMalloc") ||
        !hipOk(hipMalloc((void**)&planesDev_, 6 * sizeof(int)), "hipMalloc") ||
        !hipOk(hipMalloc((void**)&dynDev_, sizeof(DynParams)), "hipMalloc"))
        return false;
    if (!hipOk(hipMemsetAsync(res_, 0, n * 8 * 4, rootStream_), "memset")) return false;  // zeroed pool: PvContext.cpp:132
    if (!hipOk(hipMemsetAsync(delay_, 0, n * 4, rootStream_), "memset")) return false;
    const int planes[6] = {0, 1, 2, 3, 6, 7};
    // (never the legacy stream: see Solver::applyGeometry)
    if (!hipOk(hipMemcpyAsync(planesDev_, planes, sizeof(planes), hipMemcpyHostToDevice, rootStream_), "planes upload") ||
        !hipOk(hipStreamSynchronize(rootStream_), "planes upload"))
        return false;
    if (!hipOk(hipHostMalloc((void**)&dynHost_, sizeof(DynParams)), "hipHostMalloc")) return false;
    if (!hipOk(hipHostMalloc((void**)&outHost_, 8 * sizeof(float)), "hipHostMalloc")) return false;
    if (!hipOk(hipStreamSynchronize(rootStream_), "init sync")) return false;

    // FreeGrid (FreeGrid.cpp:6-34,71-110): one windowed run on an empty grid, shared by all slabs
    if (opt.withFreeGrid) {
        Solver probe;
        probe.g_ = g_;
        probe.device_ = rootDevice_;
        if (!probe.computeEfree()) return fail(probe.err_);
        efree_ = probe.efree_;
        hipSetDevice(rootDevice_);
    }
    for (Solver* sv : slabs_) sv->efree_ = efree_;
    return true;
}

// Dry run of the hand-off on the group's own streams: three sweeps of push kernels that move no rows.  False if a wait timed
// out (the slabs' streams do not run beside each other here): the group then keeps round 3's stream events.
bool SlabGroup::probeHandoff() {
    const int S = (int)slabs_.size();
    hipSetDevice(rootDevice_);
    if (hipMemsetAsync(handoff_, 0, sizeof(unsigned) * (3 * (size_t)S + 1), rootStream_) != hipSuccess ||
        hipStreamSynchronize(rootStream_) != hipSuccess)
        return false;
    const float* src[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float* dst[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const auto t0 = std::chrono::steady_clock::now();
    for (int li = 0; li < kProbeSweeps; ++li)
        for (int s = 0; s < S; ++s) {
            Solver& v = *slabs_[(size_t)s];
            HaloHandoff hand{};
            hand.count = handoff_ + 3 * s;
            hand.seq = (unsigned)li + 1u;
            hand.err = reinterpret_cast<int*>(handoff_ + 3 * S);  // (the probe's only error word: the abort word itself)
            hand.abortWord = handoff_ + 3 * S;
            if (s > 0) {
                hand.raise[0] = handoff_ + 3 * (s - 1) + 2;
                hand.await[0] = handoff_ + 3 * s + 1;
            }
            if (s + 1 < S) {
                hand.raise[1] = handoff_ + 3 * (s + 1) + 1;
                hand.await[1] = handoff_ + 3 * s + 2;
            }
            launchHaloPush(src, dst, 1024, hand, v.stream_);
        }
    bool ok = hipGetLastError() == hipSuccess;
    for (int s = 0; s < S; ++s) ok = (hipStreamSynchronize(slabs_[(size_t)s]->stream_) == hipSuccess) && ok;
    probeUsPerSweep_ = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count() / (float)kProbeSweeps;
    unsigned word = 1;  // (never the legacy stream: see Solver::applyGeometry)
    ok = ok && hipMemcpyAsync(&word, handoff_ + 3 * S, sizeof(word), hipMemcpyDeviceToHost, rootStream_) == hipSuccess &&
         hipStreamSynchronize(rootStream_) == hipSuccess && word == 0u;
    return ok;
}

SlabGroup::~SlabGroup() {
    for (Solver* sv : slabs_) delete sv;
    for (auto& e : stepEv_)
        if (e) hipEventDestroy(e);
    for (auto& e : miscEv_)
        if (e) hipEventDestroy(e);
    hipSetDevice(rootDevice_);
    if (rootStream_) hipStreamSynchronize(rootStream_);
    for (void* p : {(void*)res_, (void*)res8_, (void*)delay_, (void*)dirScratch_, (void*)planesDev_, (void*)dynDev_, (void*)handoff_})
        if (p) hipFree(p);
    if (dynHost_) hipHostFree(dynHost_);
    if (outHost_) hipHostFree(outHost_);
    for (auto& e : rootEv_)
        if (e) hipEventDestroy(e);
    rootQueue_.release();
    if (rootStream_) hipStreamDestroy(rootStream_);
}

long long SlabGroup::deviceBytes() const {
    long long b = (long long)g_.gx * g_.gy * 36 + (long long)winRows_ * winCols_ * 4;
    for (const Solver* sv : slabs_) b += sv->deviceBytes();
    return b;
}

long long SlabGroup::haloBytesPerLaunch() const {
    return (long long)(slabs_.size() - 1) * 2 * 3 * K_ * slabs_[0]->geo_.pitch * 4;
}

int SlabGroup::addBox(const Box& b) {
    int id = -1;
    for (Solver* sv : slabs_) id = sv->addBox(b);  // every slab rasterises the whole plane: identical id sequences
    return id;
}

bool SlabGroup::updateBox(int id, const Box& b) {
    for (size_t s = 0; s < slabs_.size(); ++s)
        if (!slabs_[s]->updateBox(id, b)) return slabFailed((int)s);
    return true;
}

bool SlabGroup::removeBox(int id) {
    for (size_t s = 0; s < slabs_.size(); ++s)
        if (!slabs_[s]->removeBox(id)) return slabFailed((int)s);
    return true;
}

AnalyzeArgs SlabGroup::rootArgs(float lx, float lz) const {
    const Solver& s0 = *slabs_[0];
    AnalyzeArgs a{};
    a.dyn = dynDev_;
    a.out = res_;
    a.resN = (long long)g_.gx * g_.gy;
    a.delay = delay_;
    a.G = s0.geo_.G;
    a.gx = g_.gx;
    a.gy = g_.gy;
    a.rxi = rxi_;
    a.wi = wi_;
    a.nty = s0.geo_.nty;
    a.winRows = winRows_;
    a.winCols = winCols_;
    a.dirScratch = dirScratch_;
    a.dirJump = (a.winRows > 256 && a.winCols > 256) ? 1 : 0;
    a.T = T_;
    a.fs = g_.fs;
    a.res = g_.res;
    a.dx = g_.dx;
    a.courant = g_.courant;
    a.efree = efree_;
    a.lx = lx;
    a.lz = lz;
    listenerCellRecip(g_, lx, lz, &a.lcx, &a.lcy);
    return a;
}

// One iteration of the reference's loop (PvContext.cpp:80-83) on the decomposed grid.
bool SlabGroup::run(float lx, float ly, float lz) {
    (void)ly;
    const int S = (int)slabs_.size();
    int lcx, lcy;
    listenerCell(g_, lx, lz, &lcx, &lcy);
    hipSetDevice(rootDevice_);
    if (handoff_ && !hipOk(hipMemsetAsync(handoff_, 0, sizeof(unsigned) * (3 * (size_t)S + 1), rootStream_), "hand-off words")) return false;
    hipEventRecord(rootEv_[0], rootStream_);
    DynParams d{};  // the WHOLE grid's history window (filled in below, once slab 0 has placed the columns)
    for (int s = 0; s < S; ++s) {
        Solver& v = *slabs_[(size_t)s];
        if (!hipOk(hipSetDevice(v.device_), "hipSetDevice")) return false;
        if (v.pendingTimings_ && !v.sync()) return slabFailed(s);
        if (!v.applyGeometry() || !v.prepareDyn(lcx, lcy, true, false) || !v.zeroPlanesIfNeeded()) return slabFailed(s);
        v.lastLx_ = lx;
        v.lastLz_ = lz;
        v.tim_.stepLaunches = 0;
        v.kevUsed_ = 0;
        v.loopTimed_ = false;
        v.cur_ = 0;
        v.launchCap_ = v.numGeneral_;
        hipStreamWaitEvent(v.stream_, rootEv_[0], 0);  // (the previous run's gathers have read this slab's maps)
        v.enqueueBeginRun(true);
    }
    {
        d = slabs_[0]->dynCur_;
        const int gtx0 = slabs_[0]->globalWindowTileRow0(lcx);
        d.lrow = lcx + slabs_[0]->geo_.G;
        d.histTileX0 = gtx0;
        d.histTilesX = slabs_[0]->histTilesXG_;
        d.histRow0 = slabs_[0]->geo_.G + gtx0 * rxi_;
        d.numGeneral = 0;
        hipSetDevice(rootDevice_);
        *dynHost_ = d;  // (the previous run ended with a full synchronisation: the pinned copy is free)
        if (!hipOk(hipMemcpyAsync(dynDev_, dynHost_, sizeof(DynParams), hipMemcpyHostToDevice, rootStream_), "dyn upload"))
            return false;
    }
    const int nl = ceilDiv(T_, K_);
    const size_t haloFloats = (size_t)K_ * slabs_[0]->geo_.pitch;
    if (pushHalos_) {
        // T steps, K per launch: every slab advances its rows, then PUSHES the K rows next to each boundary into its neighbours'
        // guard bands (one small launch behind its step launch: pv_halo_push_kernel; peer stores when the neighbour is on
        // another device).  Launch li + 1 of slab s is ordered behind [step + push] li of slabs s - 1, s, s + 1 only: one event
        // per slab and launch, nothing waits for the whole grid.  (Hazards: a push writes guard rows of the set the neighbour's
        // launch li did not touch and its launch li + 1 will read -- after the event; the neighbour's launch li, which READ that
        // set's guard rows of the previous round, finished before this slab's launch li started.)
        for (int li = 0; li < nl; ++li) {
            const int k = std::min(K_, T_ - li * K_);
            for (int s = 0; s < S; ++s) {
                Solver& v = *slabs_[(size_t)s];
                hipSetDevice(v.device_);
                // (neighbours on the same device hand over through words in memory, inside the push kernel: sameDev)
                const bool upSame = s > 0 && handoff_ && devices_[(size_t)s - 1] == devices_[(size_t)s];
                const bool downSame = s + 1 < S && handoff_ && devices_[(size_t)s + 1] == devices_[(size_t)s];
                if (li > 0) {
                    if (s > 0 && !upSame) hipStreamWaitEvent(v.stream_, stepEv_[(size_t)2 * (s - 1) + ((li - 1) & 1)], 0);
                    if (s + 1 < S && !downSame) hipStreamWaitEvent(v.stream_, stepEv_[(size_t)2 * (s + 1) + ((li - 1) & 1)], 0);
                }
                if (!v.enqueueSteps(li * K_, k, true, true, li == 0)) return slabFailed(s);
                const int set = v.cur_;  // the set launch li wrote (enqueueSteps has toggled cur_)
                const size_t G = (size_t)v.geo_.G, pitch = (size_t)v.geo_.pitch;
                const size_t myRows = (size_t)v.geo_.ntx * rxi_;
                const float* mine[3] = {v.pr_[set], v.vx_[set], v.vy_[set]};
                const float* src[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
                float* dst[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
                if (s > 0) {  // my first K rows -> the rows just below the upper neighbour's last row
                    Solver& u = *slabs_[(size_t)s - 1];
                    float* theirs[3] = {u.pr_[set], u.vx_[set], u.vy_[set]};
                    const size_t uRows = (size_t)u.geo_.ntx * rxi_;
                    for (int f = 0; f < 3; ++f) {
                        src[f] = mine[f] + G * pitch;
                        dst[f] = theirs[f] + (G + uRows) * pitch;
                    }
                }
                if (s + 1 < S) {  // my last K rows -> the rows just above the lower neighbour's first row
                    Solver& d = *slabs_[(size_t)s + 1];
                    float* theirs[3] = {d.pr_[set], d.vx_[set], d.vy_[set]};
                    for (int f = 0; f < 3; ++f) {
                        src[3 + f] = mine[f] + (G + myRows - K_) * pitch;
                        dst[3 + f] = theirs[f] + (G - K_) * pitch;
                    }
                }
                // words of slab s: [3 s] count, [3 s + 1] raised by the upper neighbour, [3 s + 2] raised by the lower neighbour
                HaloHandoff hand{};
                if (upSame || downSame) {
                    hand.count = handoff_ + 3 * s;
                    hand.seq = (unsigned)li + 1u;
                    hand.err = v.errFlag_;
                    hand.abortWord = handoff_ + 3 * S;
                    if (upSame) {
                        hand.raise[0] = handoff_ + 3 * (s - 1) + 2;
                        hand.await[0] = handoff_ + 3 * s + 1;
                    }
                    if (downSame) {
                        hand.raise[1] = handoff_ + 3 * (s + 1) + 1;
                        hand.await[1] = handoff_ + 3 * s + 2;
                    }
                }
                launchHaloPush(src, dst, (long long)haloFloats, hand, v.stream_);
                // (the event: cross-device neighbours, and the root's join behind the last launch)
                if (li == nl - 1 || (s > 0 && !upSame) || (s + 1 < S && !downSame)) hipEventRecord(stepEv_[(size_t)2 * s + (li & 1)], v.stream_);
            }
        }
    } else {
        // (no peer access between two adjacent slabs' devices: the receiver pulls with hipMemcpyAsync, which stages)
        // T steps, K per launch: every slab advances its rows, then takes its neighbours' K boundary rows of the set just
        // written into its guard band.  Launch li + 1 of slab s is ordered behind launch li of s - 1, s, s + 1 only.
        for (int li = 0; li < nl; ++li) {
            const int k = std::min(K_, T_ - li * K_);
            for (int s = 0; s < S; ++s) {
                Solver& v = *slabs_[(size_t)s];
                hipSetDevice(v.device_);
                if (!v.enqueueSteps(li * K_, k, true, true, li == 0)) return slabFailed(s);
                hipEventRecord(stepEv_[(size_t)2 * s + (li & 1)], v.stream_);
            }
            for (int s = 0; s < S; ++s) {
                Solver& v = *slabs_[(size_t)s];
                hipSetDevice(v.device_);
                const int set = v.cur_;  // the set launch li wrote (all slabs toggle together)
                float* mine[3] = {v.pr_[set], v.vx_[set], v.vy_[set]};
                const size_t G = (size_t)v.geo_.G, pitch = (size_t)v.geo_.pitch;
                const size_t myRows = (size_t)v.geo_.ntx * rxi_;
                if (s > 0) {  // rows just above my first row = the upper neighbour's last K rows
                    const Solver& u = *slabs_[(size_t)s - 1];
                    hipStreamWaitEvent(v.stream_, stepEv_[(size_t)2 * (s - 1) + (li & 1)], 0);
                    const float* theirs[3] = {u.pr_[set], u.vx_[set], u.vy_[set]};
                    const size_t uRows = (size_t)u.geo_.ntx * rxi_;
                    for (int f = 0; f < 3; ++f)
                        if (!hipOk(hipMemcpyAsync(mine[f] + (G - K_) * pitch, theirs[f] + (G + uRows - K_) * pitch,
                                                  haloFloats * 4, hipMemcpyDefault, v.stream_), "halo copy"))
                            return false;
                }
                if (s + 1 < S) {  // rows just below my last row = the lower neighbour's first K rows
                    const Solver& d = *slabs_[(size_t)s + 1];
                    hipStreamWaitEvent(v.stream_, stepEv_[(size_t)2 * (s + 1) + (li & 1)], 0);
                    const float* theirs[3] = {d.pr_[set], d.vx_[set], d.vy_[set]};
                    for (int f = 0; f < 3; ++f)
                        if (!hipOk(hipMemcpyAsync(mine[f] + (G + myRows) * pitch, theirs[f] + G * pitch, haloFloats * 4,
                                                  hipMemcpyDefault, v.stream_), "halo copy"))
                            return false;
                }
            }
        }
    }
    // the stencil part is over when every slab's stream has drained its last launch
    hipSetDevice(rootDevice_);
    for (int s = 0; s < S; ++s) hipStreamWaitEvent(rootStream_, stepEv_[(size_t)2 * s + ((nl - 1) & 1)], 0);
    hipEventRecord(rootEv_[1], rootStream_);

    // pressure history of every slab's last row -> the slab below (vx recurrence of its first row)
    exchangePerRun_ = 0;
    for (int s = 0; s + 1 < S; ++s) {
        Solver& v = *slabs_[(size_t)s];
        hipSetDevice(v.device_);
        launchHistRow(v.analyzeArgs(lx, lz), v.lNX_ - 1, v.histEdge_, v.stream_);
        hipEventRecord(miscEv_[(size_t)s], v.stream_);
    }
    for (int s = 1; s < S; ++s) {
        Solver& v = *slabs_[(size_t)s];
        hipSetDevice(v.device_);
        hipStreamWaitEvent(v.stream_, miscEv_[(size_t)s - 1], 0);
        const size_t bytes = (size_t)T_ * v.histPitch_ * 4;
        if (!hipOk(hipMemcpyAsync(v.histAbove_, slabs_[(size_t)s - 1]->histEdge_, bytes, hipMemcpyDefault, v.stream_),
                   "boundary history copy"))
            return false;
        exchangePerRun_ += (long long)bytes;
    }
    // whole-grid maps: delay = FLT_MAX and the default direction everywhere (Analyzer.cpp:64-68,415-428) ...
    hipSetDevice(rootDevice_);
    AnalyzeArgs ra = rootArgs(lx, lz);
    ra.abortWord = handoff_ ? handoff_ + 3 * S : nullptr;
    launchFarCells(ra, rootStream_);
    // ... every slab analyses its own cells ...
    for (int s = 0; s < S; ++s) {
        Solver& v = *slabs_[(size_t)s];
        hipSetDevice(v.device_);
        AnalyzeArgs a = v.analyzeArgs(lx, lz);
        a.abortWord = handoff_ ? handoff_ + 3 * S : nullptr;
        launchFarCells(a, v.stream_);
        launchAnalysisCells(a, v.stream_);
        hipEventRecord(miscEv_[(size_t)s], v.stream_);
        v.enqueueRunStatus();  // (the slab's error flag into pinned memory: read below without a copy and a second synchronisation)
    }
    // ... the window block of each slab's maps goes into the whole-grid maps, where the direction descent runs once
    hipSetDevice(rootDevice_);
    const int c0 = d.histCol0 - slabs_[0]->geo_.G;
    const int nc = std::max(0, std::min(winCols_, g_.gy - c0));
    for (int s = 0; s < S; ++s) {
        const Solver& v = *slabs_[(size_t)s];
        hipStreamWaitEvent(rootStream_, miscEv_[(size_t)s], 0);
        const int lr0 = v.dynCur_.histRow0 - v.geo_.G;  // first window row, slab-local
        const int nr = std::max(0, std::min(v.histTilesX_ * rxi_, v.lgx_ - lr0));
        if (nr == 0 || nc == 0) continue;
        const long long lresN = (long long)std::max(v.lgx_, 1) * g_.gy;
        if (v.device_ == rootDevice_) {
            launchCopyBlock(v.res_, lresN, g_.gy, lr0, c0, res_, ra.resN, g_.gy, lr0 + v.x0_, c0, nr, nc, 6, planesDev_,
                            planesDev_, rootStream_, ra.abortWord);
            launchCopyBlock(v.delay_, 0, g_.gy, lr0, c0, delay_, 0, g_.gy, lr0 + v.x0_, c0, nr, nc, 1, nullptr, nullptr,
                            rootStream_, ra.abortWord);
        } else {
            const int planes[6] = {0, 1, 2, 3, 6, 7};
            for (int k : planes)
                if (!hipOk(hipMemcpy2DAsync(res_ + k * ra.resN + (long long)(lr0 + v.x0_) * g_.gy + c0, (size_t)g_.gy * 4,
                                            v.res_ + k * lresN + (long long)lr0 * g_.gy + c0, (size_t)g_.gy * 4,
                                            (size_t)nc * 4, (size_t)nr, hipMemcpyDefault, rootStream_), "result gather"))
                    return false;
            if (!hipOk(hipMemcpy2DAsync(delay_ + (long long)(lr0 + v.x0_) * g_.gy + c0, (size_t)g_.gy * 4,
                                        v.delay_ + (long long)lr0 * g_.gy + c0, (size_t)g_.gy * 4, (size_t)nc * 4,
                                        (size_t)nr, hipMemcpyDefault, rootStream_), "delay gather"))
                return false;
        }
        exchangePerRun_ += (long long)nr * nc * 7 * 4;
    }
    launchAnalysisDirection(ra, rootStream_);
    hipEventRecord(rootEv_[2], rootStream_);
    if (!hipOk(hipGetLastError(), "slab run launch")) return false;
    if (!hipOk(hipStreamSynchronize(rootStream_), "slab run sync")) return false;
    for (int s = 0; s < S; ++s) {
        Solver& v = *slabs_[(size_t)s];
        hipSetDevice(v.device_);
        v.pendingTimings_ = false;
        if (!hipOk(hipStreamSynchronize(v.stream_), "slab sync")) return false;
        const int flag = v.statusHost_[0];
        v.statusQueued_ = false;
        if (flag == 5 && handoff_) {
            // A push kernel gave up waiting for a neighbour's word: the neighbour's launches did not run beside it (streams that
            // share one hardware queue can be serialised by the runtime).  From now on the cross-queue events of round 3; the run
            // is repeated with them.  (Every slab's stream is drained first; the begin-run kernels clear the flags.)
            for (int t = 0; t < S; ++t) {
                hipSetDevice(slabs_[(size_t)t]->device_);
                hipStreamSynchronize(slabs_[(size_t)t]->stream_);
            }
            hipSetDevice(rootDevice_);
            hipFree(handoff_);
            handoff_ = nullptr;
            std::fprintf(stderr, "[planeverb_amd] slab group: hand-off words not usable on this device / runtime (a halo push waited "
                                 "in vain); falling back to stream events\n");
            return run(lx, ly, lz);
        }
        if (flag) return fail("slab " + std::to_string(s) + ": pressure history window overflow");
    }
    hipSetDevice(rootDevice_);
    hipEventElapsedTime(&tim_.fdtdMs, rootEv_[0], rootEv_[1]);
    hipEventElapsedTime(&tim_.analysisMs, rootEv_[1], rootEv_[2]);
    tim_.stepLaunches = nl;
    tim_.stepLoopMs = tim_.fdtdMs;
    lastLx_ = lx;
    lastLz_ = lz;
    ran_ = true;
    return true;
}

bool SlabGroup::getOutput(float ex, float ey, float ez, float out8[8], bool* valid) {
    (void)ey;
    int cx, cy;
    *valid = resultCell(g_, ex, ez, &cx, &cy);
    if (!*valid) return true;
    if (!hipOk(hipSetDevice(rootDevice_), "hipSetDevice")) return false;
    launchGatherOutput(res_, (long long)g_.gx * g_.gy, (long long)cx * g_.gy + cy, outHost_, FarInfo{}, rootStream_);
    if (!hipOk(hipStreamSynchronize(rootStream_), "output sync")) return false;
    for (int k = 0; k < 8; ++k) out8[k] = outHost_[k];
    return true;
}

bool SlabGroup::copyResults(float* res8, float* delay) {
    if (!hipOk(hipSetDevice(rootDevice_), "hipSetDevice")) return false;
    const size_t n = (size_t)g_.gx * g_.gy;
    if (res8) {
        if (!res8_ && !hipOk(hipMalloc((void**)&res8_, n * 32), "hipMalloc")) return false;
        launchPackResults(res_, (long long)n, res8_, rootStream_);
        if (!hipOk(hipMemcpyAsync(res8, res8_, n * 32, hipMemcpyDeviceToHost, rootStream_), "results copy")) return false;
    }
    if (delay && !hipOk(hipMemcpyAsync(delay, delay_, n * 4, hipMemcpyDeviceToHost, rootStream_), "delay copy")) return false;
    return hipOk(hipStreamSynchronize(rootStream_), "results sync");
}

bool SlabGroup::copyFields(float* pr, float* vx, float* vy) {
    for (size_t s = 0; s < slabs_.size(); ++s) {
        Solver& v = *slabs_[s];
        const size_t off = (size_t)v.x0_ * g_.NY;
        if (!v.copyFields(pr ? pr + off : nullptr, vx ? vx + off : nullptr, vy ? vy + off : nullptr))
            return slabFailed((int)s);
    }
    return true;
}

bool SlabGroup::copyHistoryPlane(int t, float* pr) {
    for (size_t s = 0; s < slabs_.size(); ++s) {
        Solver& v = *slabs_[s];
        if (!v.copyHistoryPlane(t, pr + (size_t)v.x0_ * g_.NY)) return slabFailed((int)s);
    }
    return true;
}

bool SlabGroup::impulseResponse(int cx, int cy, float* out3T) {
    for (size_t s = 0; s < slabs_.size(); ++s) {
        Solver& v = *slabs_[s];
        if (cx >= v.x0_ && cx < v.x0_ + v.lNX_) {
            if (!v.impulseResponse(cx - v.x0_, cy, out3T)) return slabFailed((int)s);
            return true;
        }
    }
    return fail("cell outside the grid");
}

// ----------------------------------------------------------------------------------------------------------------
// per-rank primitives (slabs in different processes)
// ----------------------------------------------------------------------------------------------------------------

bool SlabRankOps::begin(Solver& v, float lx, float ly, float lz) {
    (void)ly;
    if (!v.isSlab()) return v.fail("not a slab");
    if (!v.hipOk(hipSetDevice(v.device_), "hipSetDevice")) return false;
    int lcx, lcy;
    listenerCell(v.g_, lx, lz, &lcx, &lcy);
    if (v.pendingTimings_ && !v.sync()) return false;
    if (!v.applyGeometry() || !v.prepareDyn(lcx, lcy, true, false) || !v.zeroPlanesIfNeeded()) return false;
    v.lastLx_ = lx;
    v.lastLz_ = lz;
    v.tim_.stepLaunches = 0;
    v.kevUsed_ = 0;
    v.loopTimed_ = false;
    v.cur_ = 0;
    v.launchCap_ = v.numGeneral_;
    v.enqueueBeginRun(true);
    return v.hipOk(hipGetLastError(), "begin run");
}

int SlabRankOps::numLaunches(const Solver& v) { return ceilDiv(v.T_, v.K_); }

bool SlabRankOps::launch(Solver& v, int li) {
    if (li < 0 || li >= numLaunches(v)) return v.fail("launch index out of range");
    hipSetDevice(v.device_);
    return v.enqueueSteps(li * v.K_, std::min(v.K_, v.T_ - li * v.K_), true, true, li == 0);
}

int SlabRankOps::haloFloats(const Solver& v) { return 3 * v.K_ * v.geo_.pitch; }

// (the `host` buffers of the four transfers below may be host OR device memory -- hipMemcpyDefault: a rank that talks RCCL
// hands in the device tensors it sends / receives, dist_slabs.TorchTransport)
bool SlabRankOps::exportHalo(Solver& v, int side, float* host) {
    hipSetDevice(v.device_);
    const size_t pitch = (size_t)v.geo_.pitch, n = (size_t)v.K_ * pitch;
    const size_t row = side == 0 ? (size_t)v.geo_.G : (size_t)v.geo_.G + (size_t)v.geo_.ntx * v.rxi_ - v.K_;
    const float* src[3] = {v.pr_[v.cur_], v.vx_[v.cur_], v.vy_[v.cur_]};
    for (int f = 0; f < 3; ++f)
        if (!v.hipOk(hipMemcpyAsync(host + f * n, src[f] + row * pitch, n * 4, hipMemcpyDefault, v.stream_), "halo export"))
            return false;
    return v.hipOk(hipStreamSynchronize(v.stream_), "halo export sync");
}

bool SlabRankOps::importHalo(Solver& v, int side, const float* host) {
    hipSetDevice(v.device_);
    const size_t pitch = (size_t)v.geo_.pitch, n = (size_t)v.K_ * pitch;
    const size_t row = side == 0 ? (size_t)v.geo_.G - v.K_ : (size_t)v.geo_.G + (size_t)v.geo_.ntx * v.rxi_;
    float* dst[3] = {v.pr_[v.cur_], v.vx_[v.cur_], v.vy_[v.cur_]};
    for (int f = 0; f < 3; ++f)
        if (!v.hipOk(hipMemcpyAsync(dst[f] + row * pitch, host + f * n, n * 4, hipMemcpyDefault, v.stream_), "halo import"))
            return false;
    return v.hipOk(hipStreamSynchronize(v.stream_), "halo import sync");  // (the host buffer may be reused at once)
}

int SlabRankOps::historyFloats(const Solver& v) { return v.T_ * v.histPitch_; }

bool SlabRankOps::exportEdgeHistory(Solver& v, float* host) {
    if (!v.histEdge_) return v.fail("the last slab has no slab below it");
    hipSetDevice(v.device_);
    launchHistRow(v.analyzeArgs(v.lastLx_, v.lastLz_), v.lNX_ - 1, v.histEdge_, v.stream_);
    if (!v.hipOk(hipMemcpyAsync(host, v.histEdge_, (size_t)historyFloats(v) * 4, hipMemcpyDefault, v.stream_), "history export"))
        return false;
    return v.hipOk(hipStreamSynchronize(v.stream_), "history export sync");
}

bool SlabRankOps::importAboveHistory(Solver& v, const float* host) {
    if (!v.histAbove_) return v.fail("the first slab has no slab above it");
    hipSetDevice(v.device_);
    if (!v.hipOk(hipMemcpyAsync(v.histAbove_, host, (size_t)historyFloats(v) * 4, hipMemcpyDefault, v.stream_), "history import"))
        return false;
    return v.hipOk(hipStreamSynchronize(v.stream_), "history import sync");
}

bool SlabRankOps::analyze(Solver& v) {
    hipSetDevice(v.device_);
    const AnalyzeArgs a = v.analyzeArgs(v.lastLx_, v.lastLz_);
    launchFarCells(a, v.stream_);
    launchAnalysisCells(a, v.stream_);
    if (!v.hipOk(hipGetLastError(), "slab analysis")) return false;
    if (!v.hipOk(hipStreamSynchronize(v.stream_), "slab analysis sync")) return false;
    int flag = 0;
    if (!v.hipOk(hipMemcpyAsync(&flag, v.errFlag_, sizeof(int), hipMemcpyDeviceToHost, v.stream_), "errFlag copy") ||
        !v.hipOk(hipStreamSynchronize(v.stream_), "errFlag sync"))
        return false;
    if (flag) return v.fail("pressure history window overflow (a tile outside the window became non-zero)");
    return true;
}

long long SlabRankOps::windowBlock(Solver& v, int* r0g, int* c0, int* nr, int* nc, float* host, long long cap) {
    hipSetDevice(v.device_);
    const int lr0 = v.dynCur_.histRow0 - v.geo_.G;
    *r0g = lr0 + v.x0_;
    *c0 = v.dynCur_.histCol0 - v.geo_.G;
    *nr = std::max(0, std::min(v.histTilesX_ * v.rxi_, v.lgx_ - lr0));
    *nc = std::max(0, std::min(v.histTilesY_ * v.wi_, v.g_.gy - *c0));
    const long long need = 7LL * *nr * *nc;
    if (!host || cap < need || need == 0) return need;
    const int planes[6] = {0, 1, 2, 3, 6, 7};
    const long long lresN = (long long)std::max(v.lgx_, 1) * v.g_.gy;
    const size_t blk = (size_t)*nr * *nc;
    for (int k = 0; k < 7; ++k) {
        const float* src = k < 6 ? v.res_ + planes[k] * lresN : v.delay_;
        if (!v.hipOk(hipMemcpy2DAsync(host + k * blk, (size_t)*nc * 4, src + (long long)lr0 * v.g_.gy + *c0, (size_t)v.g_.gy * 4,
                                      (size_t)*nc * 4, (size_t)*nr, hipMemcpyDeviceToHost, v.stream_), "window block"))
            return -1;
    }
    if (!v.hipOk(hipStreamSynchronize(v.stream_), "window block sync")) return -1;
    return need;
}

SlabRoot* SlabRoot::create(const Solver& a, int device, std::string* err) {
    std::unique_ptr<SlabRoot> r(new SlabRoot());
    r->g_ = a.g_;
    r->device_ = device;
    r->G_ = a.geo_.G;
    r->rxi_ = a.rxi_;
    r->wi_ = a.wi_;
    r->nty_ = a.geo_.nty;
    r->T_ = a.T_;
    r->K_ = a.K_;
    r->ntxG_ = a.ntxG_;
    r->histTilesXG_ = a.histTilesXG_;
    r->histTilesY_ = a.histTilesY_;
    r->efree_ = a.efree_;
    r->winRows_ = a.histTilesXG_ * a.rxi_;
    r->winCols_ = a.histTilesY_ * a.wi_;
    const size_t n = (size_t)r->g_.gx * r->g_.gy;
    const int planes[6] = {0, 1, 2, 3, 6, 7};
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&r->stream_, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void**)&r->res_, n * 32) == hipSuccess && hipMalloc((void**)&r->delay_, n * 4) == hipSuccess &&
              hipMalloc((void**)&r->dirScratch_, (size_t)r->winRows_ * r->winCols_ * 4) == hipSuccess &&
              hipMalloc((void**)&r->planesDev_, sizeof(planes)) == hipSuccess &&
              hipMalloc((void**)&r->dynDev_, sizeof(DynParams)) == hipSuccess &&
              hipHostMalloc((void**)&r->outHost_, 32) == hipSuccess &&
              hipMemsetAsync(r->res_, 0, n * 32, r->stream_) == hipSuccess &&
              hipMemsetAsync(r->delay_, 0, n * 4, r->stream_) == hipSuccess &&
              hipMemcpyAsync(r->planesDev_, planes, sizeof(planes), hipMemcpyHostToDevice, r->stream_) == hipSuccess &&
              hipStreamSynchronize(r->stream_) == hipSuccess;  // (never the legacy stream: see Solver::applyGeometry)
    if (!ok) {
        if (err) *err = "slab root: allocation failed";
        return nullptr;
    }
    return r.release();
}

SlabRoot::~SlabRoot() {
    hipSetDevice(device_);
    if (stream_) hipStreamSynchronize(stream_);
    for (void* p : {(void*)res_, (void*)res8_, (void*)delay_, (void*)stage_, (void*)dirScratch_, (void*)planesDev_, (void*)dynDev_})
        if (p) hipFree(p);
    if (outHost_) hipHostFree(outHost_);
    if (stream_) hipStreamDestroy(stream_);
}

AnalyzeArgs SlabRoot::args() const {
    AnalyzeArgs a{};
    a.dyn = dynDev_;
    a.out = res_;
    a.resN = (long long)g_.gx * g_.gy;
    a.delay = delay_;
    a.G = G_;
    a.gx = g_.gx;
    a.gy = g_.gy;
    a.rxi = rxi_;
    a.wi = wi_;
    a.nty = nty_;
    a.winRows = winRows_;
    a.winCols = winCols_;
    a.dirScratch = dirScratch_;
    a.dirJump = (winRows_ > 256 && winCols_ > 256) ? 1 : 0;
    a.T = T_;
    a.fs = g_.fs;
    a.res = g_.res;
    a.dx = g_.dx;
    a.courant = g_.courant;
    a.efree = efree_;
    a.lx = lx_;
    a.lz = lz_;
    listenerCellRecip(g_, lx_, lz_, &a.lcx, &a.lcy);
    return a;
}

bool SlabRoot::begin(float lx, float ly, float lz) {
    (void)ly;
    lx_ = lx;
    lz_ = lz;
    int lcx, lcy;
    listenerCell(g_, lx, lz, &lcx, &lcy);
    // the whole grid's history window: the same placement every slab derives (Solver::prepareDyn)
    const int reach = T_ + 2 + K_;
    auto floorDiv = [](int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); };
    int gtx0 = 0, ty0 = 0;
    if (histTilesXG_ < ntxG_)
        gtx0 = std::min(std::max(floorDiv(std::min(std::max(lcx, 0), g_.gx) - reach, rxi_), 0), ntxG_ - histTilesXG_);
    if (histTilesY_ < nty_)
        ty0 = std::min(std::max(floorDiv(std::min(std::max(lcy, 0), g_.gy) - reach, wi_), 0), nty_ - histTilesY_);
    DynParams d{};
    d.lrow = lcx + G_;
    d.lcol = lcy + G_;
    d.histTileX0 = gtx0;
    d.histTileY0 = ty0;
    d.histTilesX = histTilesXG_;
    d.histTilesY = histTilesY_;
    d.histRow0 = G_ + gtx0 * rxi_;
    d.histCol0 = G_ + ty0 * wi_;
    if (hipSetDevice(device_) != hipSuccess || hipMemcpyAsync(dynDev_, &d, sizeof(d), hipMemcpyHostToDevice, stream_) != hipSuccess ||
        hipStreamSynchronize(stream_) != hipSuccess)
        return fail("slab root: dyn upload failed");
    launchFarCells(args(), stream_);
    return hipGetLastError() == hipSuccess ? true : fail("slab root: far cells launch failed");
}

bool SlabRoot::importBlock(int r0g, int c0, int nr, int nc, const float* host7) {
    if (nr <= 0 || nc <= 0) return true;
    if (r0g < 0 || c0 < 0 || r0g + nr > g_.gx || c0 + nc > g_.gy) return fail("slab root: block outside the map");
    hipSetDevice(device_);
    const size_t blk = (size_t)nr * nc;
    if (blk * 7 > stageCap_) {
        if (stage_) hipFree(stage_);
        stage_ = nullptr;
        if (hipMalloc((void**)&stage_, blk * 7 * 4) != hipSuccess) return fail("slab root: staging allocation failed");
        stageCap_ = blk * 7;
    }
    if (hipMemcpyAsync(stage_, host7, blk * 7 * 4, hipMemcpyHostToDevice, stream_) != hipSuccess)
        return fail("slab root: block upload failed");
    // staged planes 0..5 -> result planes {0, 1, 2, 3, 6, 7}; staged plane 6 -> the delay map
    launchCopyBlock(stage_, (long long)blk, nc, 0, 0, res_, (long long)g_.gx * g_.gy, g_.gy, r0g, c0, nr, nc, 6, nullptr,
                    planesDev_, stream_);
    launchCopyBlock(stage_ + 6 * blk, 0, nc, 0, 0, delay_, 0, g_.gy, r0g, c0, nr, nc, 1, nullptr, nullptr, stream_);
    return hipGetLastError() == hipSuccess ? true : fail("slab root: block copy failed");
}

bool SlabRoot::finish() {
    hipSetDevice(device_);
    launchAnalysisDirection(args(), stream_);
    if (hipGetLastError() != hipSuccess) return fail("slab root: direction launch failed");
    return hipStreamSynchronize(stream_) == hipSuccess ? true : fail("slab root: sync failed");
}

bool SlabRoot::getOutput(float ex, float ey, float ez, float out8[8], bool* valid) {
    (void)ey;
    int cx, cy;
    *valid = resultCell(g_, ex, ez, &cx, &cy);
    if (!*valid) return true;
    hipSetDevice(device_);
    launchGatherOutput(res_, (long long)g_.gx * g_.gy, (long long)cx * g_.gy + cy, outHost_, FarInfo{}, stream_);
    if (hipStreamSynchronize(stream_) != hipSuccess) return fail("slab root: output sync failed");
    for (int k = 0; k < 8; ++k) out8[k] = outHost_[k];
    return true;
}

bool SlabRoot::copyResults(float* res8, float* delay) {
    hipSetDevice(device_);
    const size_t n = (size_t)g_.gx * g_.gy;
    if (res8) {
        if (!res8_ && hipMalloc((void**)&res8_, n * 32) != hipSuccess) return fail("slab root: allocation failed");
        launchPackResults(res_, (long long)n, res8_, stream_);
        if (hipMemcpyAsync(res8, res8_, n * 32, hipMemcpyDeviceToHost, stream_) != hipSuccess) return fail("results copy");
    }
    if (delay && hipMemcpyAsync(delay, delay_, n * 4, hipMemcpyDeviceToHost, stream_) != hipSuccess) return fail("delay copy");
    return hipStreamSynchronize(stream_) == hipSuccess ? true : fail("slab root: sync failed");
}

}  // namespace pva
