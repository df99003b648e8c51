// pv_slabs.cpp -- see pv_slabs.h
#include "pv_slabs.h"

#include <algorithm>
#include <climits>
#include <cstring>

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
    SlabGroup* g = new SlabGroup();
    if (!g->init(spec, devices, opt)) {
        if (err) *err = g->err_;
        delete g;
        return nullptr;
    }
    return g;
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
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();  // copies then stage
        }
    rootDevice_ = devices_[0];
    if (!hipOk(hipSetDevice(rootDevice_), "hipSetDevice")) return false;
    if (!hipOk(hipStreamCreateWithFlags(&rootStream_, hipStreamNonBlocking), "hipStreamCreate")) return false;
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
    if (!hipOk(hipMemcpy(planesDev_, planes, sizeof(planes), hipMemcpyHostToDevice), "planes upload")) return false;
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

SlabGroup::~SlabGroup() {
    for (Solver* sv : slabs_) delete sv;
    for (auto& e : stepEv_)
        if (e) hipEventDestroy(e);
    for (auto& e : miscEv_)
        if (e) hipEventDestroy(e);
    hipSetDevice(rootDevice_);
    if (rootStream_) hipStreamSynchronize(rootStream_);
    for (void* p : {(void*)res_, (void*)res8_, (void*)delay_, (void*)dirScratch_, (void*)planesDev_, (void*)dynDev_})
        if (p) hipFree(p);
    if (dynHost_) hipHostFree(dynHost_);
    if (outHost_) hipHostFree(outHost_);
    for (auto& e : rootEv_)
        if (e) hipEventDestroy(e);
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
    hipEventRecord(rootEv_[0], rootStream_);
    DynParams d{};  // the WHOLE grid's history window (filled in below, once slab 0 has placed the columns)
    for (int s = 0; s < S; ++s) {
        Solver& v = *slabs_[(size_t)s];
        if (!hipOk(hipSetDevice(v.device_), "hipSetDevice")) return false;
        if (v.pendingTimings_ && !v.sync()) return slabFailed(s);
        if (!v.applyGeometry() || !v.prepareDyn(lcx, lcy, true, false)) return slabFailed(s);
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
    // T steps, K per launch: every slab advances its rows, then takes its neighbours' K boundary rows of the set just
    // written into its guard band.  Launch li + 1 of slab s is ordered behind launch li of s - 1, s, s + 1 only.
    const int nl = ceilDiv(T_, K_);
    const size_t haloFloats = (size_t)K_ * slabs_[0]->geo_.pitch;
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
    const AnalyzeArgs ra = rootArgs(lx, lz);
    launchFarCells(ra, rootStream_);
    // ... every slab analyses its own cells ...
    for (int s = 0; s < S; ++s) {
        Solver& v = *slabs_[(size_t)s];
        hipSetDevice(v.device_);
        const AnalyzeArgs a = v.analyzeArgs(lx, lz);
        launchFarCells(a, v.stream_);
        launchAnalysisCells(a, v.stream_);
        hipEventRecord(miscEv_[(size_t)s], v.stream_);
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
                            rootStream_);
            launchCopyBlock(v.delay_, 0, g_.gy, lr0, c0, delay_, 0, g_.gy, lr0 + v.x0_, c0, nr, nc, 1, nullptr, rootStream_);
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
        int flag = 0;
        if (!hipOk(hipMemcpy(&flag, v.errFlag_, sizeof(int), hipMemcpyDeviceToHost), "errFlag copy")) return false;
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
    launchGatherOutput(res_, (long long)g_.gx * g_.gy, (long long)cx * g_.gy + cy, outHost_, rootStream_);
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

}  // namespace pva
