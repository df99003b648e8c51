// pv_solver.cpp -- see pv_solver.h
#include "pv_solver.h"

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>

#include "pv_launch.h"

namespace pva {

#ifndef PV_LABEL_MAX_CELLS
#define PV_LABEL_MAX_CELLS (1100 * 1100)  // array cells up to which the air components are labelled on the host (~15 ns per cell)
#endif
#ifndef PV_ANALYSIS_FORK_CELLS
#define PV_ANALYSIS_FORK_CELLS 4096
#endif
namespace {
// Row bands per sweep when the option is left at "auto": 1 = off.  Measured on MI355X (profiles/r02_bands.txt): the
// banded sweeps are bit-exact but 6-35 % SLOWER than one launch per sweep at 4096^2 and 8192^2 for every band count --
// two cross-stream event waits per band and sweep cost more than the chip-wide drain they remove.
constexpr int kAutoRowBands = 1;
constexpr int kDefaultPatch = 0;  // persistent patch kernel for the (12, 36) tile: off until measured faster (PVA_OPT_PATCH_KERNEL)
constexpr long long kAnalysisForkCells = PV_ANALYSIS_FORK_CELLS;  // window cells from which the decay-time pass runs beside the encode pass (enqueueAnalysis)
constexpr int kMinGuard = 8;  // guard width = max(this, K): a tile's halo never leaves the allocation
inline int roundUp(int v, int m) { return (v + m - 1) / m * m; }
inline int ceilDiv(int a, int b) { return (a + b - 1) / b; }
inline int floorDiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
}  // namespace

// resident kernel (pv_resident.hip): blocks of runs in flight per device, all solvers of the process
static constexpr int kResidentMaxTiles = 512;
static int kResidentXcdMaxTiles = 32;  // one workgroup per CU of one XCD (PLANEVERB_AMD_RESIDENT_XCD=N: N > 1 sets it, measurements)
static std::atomic<int>& residentXcdInFlight(int device, int xcd) {
    static std::atomic<int> n[64][8];
    return n[device & 63][xcd & 7];
}
static std::atomic<int>& residentInFlight(int device) {
    static std::atomic<int> n[64];
    return n[device & 63];
}

void Solver::releaseResident() {
    if (residentHeld_ > 0) residentInFlight(device_).fetch_sub(residentHeld_);
    residentHeld_ = 0;
    if (xcdHeld_ > 0) residentXcdInFlight(device_, xcdTarget_).fetch_sub(xcdHeld_);
    xcdHeld_ = 0;
}

bool Solver::fail(const std::string& what) {
    err_ = what;
    return false;
}

bool Solver::hipOk(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    err_ = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

template <typename Tp>
bool Solver::dalloc(Tp** p, size_t count, bool zero) {
    const size_t bytes = count * sizeof(Tp);
    if (!hipOk(hipMalloc((void**)p, bytes), "hipMalloc")) return false;
    deviceBytes_ += (long long)bytes;
    if (zero && !hipOk(hipMemsetAsync(*p, 0, bytes, stream_), "hipMemsetAsync")) return false;
    return true;
}

// claimed streams (the live solvers' main streams, slab groups' root streams), per device (QueueClaim)
namespace {
struct ClaimedStream {
    int device, priority;
    hipStream_t stream;
    std::shared_ptr<std::recursive_mutex> use;  // QueueClaim::use of the owner
};
std::mutex g_streamRegistryMutex;
std::vector<ClaimedStream> g_claimedStreams;

hipError_t createStream(hipStream_t* s, int priority) {  // priority: QueueClaim's classes
    int lo = 0, hi = 0;  // numerically lowest = highest priority
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (priority == QueueClaim::kNormal) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, priority == QueueClaim::kHigh ? hi : lo);
}
}  // namespace

bool QueueClaim::claim(int device, hipStream_t* stream, int priority, const void* owner) {
    const char* e = getenv("PLANEVERB_AMD_QUEUE_PROBE");
    if (e && atoi(e) == 0) return true;
    std::lock_guard<std::mutex> lk(g_streamRegistryMutex);
    unsigned long long* stamps = nullptr;
    size_t peers = 0;
    for (int attempt = 0; attempt < 6; ++attempt) {
        bool shared = false;
        peers = 0;
        for (const auto& other : g_claimedStreams) {
            // Streams of different priorities never share a queue (a pool per priority) -- and the probe cannot tell: a
            // high-priority sleeper keeps a normal stream's stamp kernel waiting from ANOTHER queue just as well.
            if (other.device != device || other.priority != priority) continue;
            ++peers;
            // (the owner is enqueueing or capturing right now, or its stream is busy: left alone)
            std::unique_lock<std::recursive_mutex> owner(*other.use, std::try_to_lock);
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (!owner.owns_lock() || hipStreamIsCapturing(other.stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone ||
                hipStreamQuery(other.stream) != hipSuccess)
                continue;
            if (!stamps && hipHostMalloc((void**)&stamps, 4 * sizeof(unsigned long long)) != hipSuccess) return true;
            if (streamsShareQueue(other.stream, *stream, stamps)) {
                shared = true;
                break;
            }
        }
        if (!shared) break;
        parked.push_back(*stream);  // (alive until the owner goes: the pool then deals the next stream another queue)
        *stream = nullptr;
        if (createStream(stream, priority) != hipSuccess) {
            if (stamps) hipHostFree(stamps);
            return false;
        }
        ++redeals;
    }
    if (stamps) hipHostFree(stamps);
    if (e && atoi(e) >= 2)
        std::fprintf(stderr, "[planeverb_amd] stream of %p (priority class %d): %d re-deal(s), %zu claimed stream(s) of its class, device %d\n",
                     owner, priority, redeals, peers, device);
    use = std::make_shared<std::recursive_mutex>();
    g_claimedStreams.push_back(ClaimedStream{device, priority, *stream, use});
    claimed = *stream;
    priorityClass = priority;
    return true;
}

bool QueueClaim::replace(hipStream_t* stream) {
    if (!claimed || *stream != claimed) return false;  // (not a claimed stream: PLANEVERB_AMD_QUEUE_PROBE=0)
    std::lock_guard<std::mutex> lk(g_streamRegistryMutex);
    hipStream_t fresh = nullptr;
    if (createStream(&fresh, priorityClass) != hipSuccess) return false;
    for (auto& c : g_claimedStreams)
        if (c.stream == claimed) c.stream = fresh;
    parked.push_back(claimed);
    claimed = fresh;
    *stream = fresh;
    ++redeals;
    return true;
}

void QueueClaim::release() {
    if (claimed) {
        std::lock_guard<std::mutex> lk(g_streamRegistryMutex);
        for (size_t i = 0; i < g_claimedStreams.size(); ++i)
            if (g_claimedStreams[i].stream == claimed) {
                g_claimedStreams.erase(g_claimedStreams.begin() + (long)i);
                break;
            }
        claimed = nullptr;
    }
    for (hipStream_t x : parked) hipStreamDestroy(x);
    parked.clear();
}

Solver* Solver::create(const GridSpec& spec, int device, const SolverOptions& opt, std::string* err) {
    // (owned across init: an exception from underneath -- a table that cannot grow -- unwinds through ~Solver, which releases
    // whatever init had created, and leaves through the C-ABI's barrier in pv_capi.cpp)
    std::unique_ptr<Solver> s(new Solver());
    if (!s->init(spec, device, opt)) {
        if (err) *err = s->err_;
        return nullptr;
    }
    return s.release();
}

bool Solver::init(const GridSpec& spec, int device, const SolverOptions& opt) {
    g_ = spec;
    opt_ = opt;
    device_ = device;
    if (spec.gx < 1 || spec.gy < 1) return fail("grid has no cells");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail("no HIP device: libplaneverb_amd has no CPU path");
    if (device < 0 || device >= ndev) return fail("HIP device index out of range");
    if (!hipOk(hipSetDevice(device), "hipSetDevice")) return false;
    {
        // (slab groups: the odd slabs' streams get the high priority -- streams of different priorities never share a hardware
        // queue, and two neighbouring slabs whose streams do share one run their launches one after the other:
        // PLANEVERB_AMD_SLAB_PRIORITY=0 switches it off, profiles/r04_slabs.txt)
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        const char* e = getenv("PLANEVERB_AMD_SLAB_PRIORITY");
        const bool high = opt.streamPriority > 0 || (opt.slabCount > 1 && (opt.slabIndex & 1) && !(e && atoi(e) == 0));
        if (!hipOk(high ? hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, hi)
                        : hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate"))
            return false;
        // A hardware queue apart from the other solvers' for the main stream: the runtime multiplexes a process's streams on a
        // small pool of hardware queues (four per priority by default), dealt by the number of streams that already use each,
        // and launches of two streams that share one run one after the other -- two runs "in flight" on two solvers then take as
        // long as one after the other (1.47-1.59e12 instead of 1.70e12 at 4096^2, by how many streams the process had created
        // before: profiles/r04_placement.txt; round 4 repaired that in bench.py, for itself).  Checked here, once, against every
        // other live solver's main stream on the device; a stream that shares a queue is kept (parked, so that the pool deals
        // the next one elsewhere) and replaced.  (A stream with a CU mask does not get a queue of its own either: measured.)
        // Slab groups: every slab's stream, whatever its priority (a push kernel of the hand-off WAITS for the neighbour's).
        if (!opt.skipAnalysis && (!high || opt.slabCount > 1) && !queue_.claim(device_, &stream_, high ? QueueClaim::kHigh : QueueClaim::kNormal, this))
            return fail("hipStreamCreate");  // (not the free-grid child: it runs alone, once)
    }
    {
        int lo = 0, hi = 0;  // numerically lowest = highest priority
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (!hipOk(hipStreamCreateWithPriority(&stream2_, hipStreamNonBlocking, hi), "hipStreamCreate")) return false;
    }
    for (int i = 0; i < opt.auxStreams; ++i) {
        hipStream_t x = nullptr;
        if (!hipOk(hipStreamCreateWithFlags(&x, hipStreamNonBlocking), "hipStreamCreate")) return false;
        auxStreams_.push_back(x);
    }
    if (!hipOk(hipEventCreateWithFlags(&forkEv_, hipEventDisableTiming), "hipEventCreate")) return false;
    for (auto& e : anaEv_)
        if (!hipOk(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) return false;
    for (auto& e : ev_)
        if (!hipOk(hipEventCreate(&e), "hipEventCreate")) return false;

    // Default tile = fastest measured on MI355X for the grid's size class (tools/gpu_tune.py, profiles/r01_sizes.txt).
    // Large grids take the 60-row tile at 2 waves/SIMD with K = 12 (36 x 40 interior cells of the 60 x 64 loaded:
    // 3.7 B of CU-level traffic per cell-step instead of 4.8 for the 40-row tile at K = 8); it needs several
    // thousand tiles to fill 256 CUs x 8 waves a few times over.  Around 2048^2 the 56-row tile at K = 10 fills the
    // chip more evenly; smaller grids are bound by launch latency and wave quantisation and prefer many small tiles.
    const long long tiles36 = (long long)ceilDiv(g_.NX, 36) * ceilDiv(g_.NY, 40);
    if (opt.K > 0 || opt.rxi > 0) {
        K_ = opt.K > 0 ? opt.K : 8;
        rxi_ = opt.rxi > 0 ? opt.rxi : 24;
    } else if (tiles36 >= 4500) {
        K_ = 12;
        rxi_ = 36;
    } else if (tiles36 >= 1500) {
        K_ = 10;
        rxi_ = 36;
    } else if (tiles36 >= 1000) {
        K_ = 8;
        rxi_ = 24;
    } else if (tiles36 >= 150) {
        // launch-bound grids (450^2 ... 1150^2): the chip is far from full, so a deeper K on a smaller tile trades halo
        // recomputation nobody waits for against fewer dependent launches: +7-15 % at 512^2 ... 1024^2 with four runs in flight
        // (profiles/r03_small_tiles.txt)
        K_ = 10;
        rxi_ = 20;
    } else {
        // the reference's own presets (70^2 ... 382^2: a handful of general tiles, one run at a time): 36 loaded rows over four
        // waves instead of 40, 12 steps per launch instead of 8: 7-19 % off an iteration
        K_ = 12;
        rxi_ = 12;
    }
    if (!stepConfigSupported(K_, rxi_)) return fail("unsupported (stepsPerLaunch, tileRows) configuration");
    if (opt_.tileOrder < 0) opt_.tileOrder = tiles36 >= 4500 ? 3 : 1;  // (measured: profiles/r02_tile_order.txt)
    // odd launches walk the strips backwards: what the previous launch wrote last is read first, out of the Infinity Cache
    // (4096^2: +2 % with two runs in flight, +4-5 % with one; profiles/r04_alternate_sweeps.txt)
    if (opt_.alternateSweeps < 0) opt_.alternateSweeps = opt_.tileOrder == 3 ? 1 : 0;
    // tile order 3: 2 x 4 regions instead of 8 strips of tile columns (xcdTileAt; profiles/r04_xcd_regions.txt)
    // (where the strips are narrow: 4096^2 has 13 tile columns per strip, 3072^2 10: +2-4 %; 8192^2 has 26 and loses 1-2 %)
    if (opt_.xcdRegions < 0) opt_.xcdRegions = (opt_.tileOrder == 3 && ceilDiv(ceilDiv(g_.NY, 64 - 2 * K_), 8) < 20) ? 1 : 0;
    // edge tiles are an "allow": only the batched kernels of the mirror-pair tiles have that arm
    if (!opt_.packed && !unpackedAirOk())
        return fail("PVA_OPT_PACKED_MATH = 0 (the unpacked air kernel) is a validation form of the experimental build of the library");
    if (opt_.edgeTiles && !(edgeConfigOk(K_, rxi_) && opt_.packed && opt_.merged == 1 && !opt_.streaming &&
                            opt_.timeKernels == 0))
        opt_.edgeTiles = false;
    wi_ = 64 - 2 * K_;
    T_ = opt.numSteps > 0 ? opt.numSteps : g_.T;
    // The live module must come up at any resolution the reference accepts (PvContext.cpp:101-107): where the T-step
    // history window does not fit beside the planes, fall back to the sparse-emitter mode (a 2 x 8-launch ring; the
    // caller then registers its emitters before every run: Context::workerLoop).
    if (opt_.autoStreaming && !opt_.streaming && !opt_.denseHistory && opt_.slabCount == 1) {
        const int reach = T_ + 2 + K_;
        const long long wtx = std::min(ceilDiv(g_.NX, rxi_), ceilDiv(2 * reach + 1, rxi_) + 1);
        const long long wty = std::min(ceilDiv(g_.NY, wi_), ceilDiv(2 * reach + 1, wi_) + 1);
        const long long histBytes = wtx * wty * rxi_ * wi_ * 4 * (long long)T_;
        const long long planeBytes = (long long)(g_.NX + 64) * (g_.NY + 128) * (24 + 2 + 36 + 12);  // fields, codes, maps, scratch
        // (decided on the device's TOTAL memory: the mode decides which outputs exist -- wet gain / RT60 for registered
        // emitters only -- and must not flip with what other processes hold at this moment; a history that then does not fit
        // what is free fails the creation with a message, below)
        size_t freeB = 0, totalB = 0;
        hipMemGetInfo(&freeB, &totalB);
        if ((unsigned long long)(histBytes + planeBytes) + (8ull << 30) > totalB || wtx * wty * rxi_ * wi_ * 4 > (long long)INT_MAX)
            opt_.streaming = true;
    }
    if (opt_.streaming && opt_.edgeTiles) opt_.edgeTiles = false;

    ntxG_ = ceilDiv(g_.NX, rxi_);
    if (opt_.slabCount < 1 || opt_.slabIndex < 0 || opt_.slabIndex >= opt_.slabCount) return fail("invalid slab spec");
    if (isSlab()) {
        if (opt_.slabCount > ntxG_ / 2) return fail("too many slabs: every slab needs at least two tile rows");
        if (opt_.streaming || opt_.edgeTiles || stepConfigStacked(K_, rxi_) || opt_.merged != 1 ||
            !mergedConfigOk(K_, rxi_))
            return fail("slabs need the default merged step kernel (no streaming analysis, stacked or edge tiles)");
        opt_.rowBands = 1;
        opt_.useGraph = 2;
        opt_.smallGrid = 2;
        opt_.withFreeGrid = false;  // the group computes EFree once
    }
    tileRow0_ = (int)((long long)ntxG_ * opt_.slabIndex / opt_.slabCount);
    const int tileRow1 = (int)((long long)ntxG_ * (opt_.slabIndex + 1) / opt_.slabCount);
    x0_ = tileRow0_ * rxi_;
    lNX_ = std::min(tileRow1 * rxi_, g_.NX) - x0_;
    lgx_ = std::max(0, std::min(tileRow1 * rxi_, g_.gx) - x0_);
    geo_.gx = lgx_;
    geo_.gy = g_.gy;
    geo_.NX = lNX_;
    geo_.NY = g_.NY;
    geo_.x0 = x0_;
    geo_.NXg = g_.NX;
    geo_.gxg = g_.gx;
    const int kGuard = std::max(kMinGuard, K_ + stepConfigExtraRows(K_, rxi_));
    geo_.G = kGuard;
    geo_.rxi = rxi_;
    geo_.wi = wi_;
    geo_.ntx = tileRow1 - tileRow0_;
    geo_.nty = ceilDiv(g_.NY, wi_);
    geo_.rows = kGuard + geo_.ntx * rxi_ + kGuard;
    geo_.pitch = roundUp(kGuard + geo_.nty * wi_ + kGuard, 64);
    const size_t plane = (size_t)geo_.rows * geo_.pitch;
    // (the face-coefficient plane is 12 B per cell and is reached through one buffer descriptor with 32-bit offsets, like the
    // three-plane span of the segment and patch kernels: ~13 400^2 padded cells)
    if (plane * 12 > (size_t)INT_MAX) return fail("grid too large for 32-bit plane offsets (12 bytes per cell of face coefficients)");
    const int ntiles = geo_.ntx * geo_.nty;

    // lane-shift self test: the stencil relies on DPP wave shifts moving data by exactly one lane
    {
        float* d = nullptr;
        if (!dalloc(&d, 128, true)) return false;
        launchLaneSelfTest(d, stream_);
        float h[128];
        if (!hipOk(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, stream_), "selftest copy")) return false;
        if (!hipOk(hipStreamSynchronize(stream_), "selftest sync")) return false;
        hipFree(d);
        deviceBytes_ -= 128 * 4;
        for (int i = 0; i < 63; ++i)
            if (h[i] != (float)(i + 1)) return fail("DPP wave_shl self-test failed");
        for (int i = 1; i < 64; ++i)
            if (h[64 + i] != (float)(i - 1)) return fail("DPP wave_shr self-test failed");
    }

    // pr, vx, vy of a buffer set are ONE allocation: the segment kernel reaches the three planes of a set through a
    // single buffer descriptor (pv_seg.h)
    for (int i = 0; i < 2; ++i) {
        if (!dalloc(&pr_[i], 3 * plane, true)) return false;
        vx_[i] = pr_[i] + plane;
        vy_[i] = pr_[i] + 2 * plane;
    }
    if (!dalloc(&coef_, plane, true)) return false;
    if (!dalloc(&matDev_, (size_t)g_.NX * g_.NY, true)) return false;
    if (!dalloc(&pulseDev_, (size_t)std::max(T_, g_.T), true)) return false;
    if (!dalloc(&tileFirst_, (size_t)ntiles, true)) return false;
    if (!dalloc(&tileClass_, (size_t)ntiles, true)) return false;
    if (!dalloc(&tileDead_, (size_t)ntiles, true) || !dalloc(&deadCount_, 1, true)) return false;
    if (!dalloc(&nz_[0], (size_t)ntiles, true) || !dalloc(&nz_[1], (size_t)ntiles, true)) return false;
    listCap_ = ntiles;
    if (!dalloc(&generalList_, (size_t)listCap_, true)) return false;
    if (!dalloc(&generalCount_, 1, true)) return false;
    if (!dalloc(&dynDev_, 1, true)) return false;
    if (!dalloc(&errFlag_, 1, true)) return false;
    if (!dalloc(&activeCount_, 8, true)) return false;  // AnalyzeArgs::activeCount
    if (!dalloc(&res_, (size_t)std::max(lgx_, 1) * g_.gy * 8, true)) return false;  // zeroed pool: PvContext.cpp:132
    if (!dalloc(&delay_, (size_t)std::max(lgx_, 1) * g_.gy, true)) return false;
    // far cells lazily: whole grids with a windowed history (a slab group / the streaming mode run their own passes)
    lazyFar_ = opt_.lazyFar && !isSlab() && !opt_.streaming;
    if (lazyFar_) launchFillDelay(delay_, (long long)std::max(lgx_, 1) * g_.gy, stream_);  // "no onset", Analyzer.cpp:64-68
    {
        const char* e = getenv("PLANEVERB_AMD_NEAR_BOX");
        useNearBox_ = lazyFar_ && !(e && atoi(e) == 0);
        if (useNearBox_) {
            if (!dalloc(&nearBox_, 8, false)) return false;
            const int empty[8] = {INT_MAX, INT_MAX, -1, -1, INT_MAX, INT_MAX, -1, -1};
            // (pageable source: the copy is staged before the call returns)
            if (!hipOk(hipMemcpyAsync(nearBox_, empty, sizeof(empty), hipMemcpyHostToDevice, stream_), "near box")) return false;
        }
    }
    scratchCount_ = std::max<size_t>({(size_t)3 * std::max(T_, g_.T), (size_t)lNX_ * g_.NY * 3,
                                     (size_t)geo_.ntx * rxi_ * geo_.nty * wi_});  // the last: direction scratch
    if (!dalloc(&scratch_, scratchCount_, true)) return false;

    // history window: the pulse moves at most one cell per step along each axis, so after T steps everything
    // farther than T cells from the listener is still exactly zero and needs no storage
    {
        // + K: a tile counts as active as soon as its K-cell halo is touched
        const int reach = T_ + 2 + K_;
        int wtx = geo_.ntx, wty = geo_.nty;
        histTilesXG_ = ntxG_;
        if (!opt_.denseHistory && !opt_.streaming) {
            histTilesXG_ = std::min(ntxG_, ceilDiv(2 * reach + 1, rxi_) + 1);
            wtx = std::min(geo_.ntx, histTilesXG_);  // (a slab records its part of the whole grid's window)
            wty = std::min(geo_.nty, ceilDiv(2 * reach + 1, wi_) + 1);
        }
        histTilesX_ = wtx;
        histTilesY_ = wty;
        histRows_ = wtx * rxi_;
        histPitch_ = roundUp(wty * wi_, 64);
        histPlane_ = (long long)wtx * wty * rxi_ * wi_;  // tile-major planes: [tile][row][col], no padding
        // streaming mode keeps a ring of 8 launches' worth of planes instead of all T
        // streaming: TWO half rings of 8 launches' planes -- the forward sums of one half advance on a second stream while
        // the step kernels fill the other (the accumulate pass is latency / bandwidth work, the stencil VALU work)
        halfRing_ = std::min(roundUp(T_, K_), 8 * K_);
        ring_ = opt_.streaming ? 2 * halfRing_ : T_;
        const long long bytes = histPlane_ * 4 * (long long)ring_;
        if (histPlane_ * 4 > (long long)INT_MAX) return fail("history plane too large for 32-bit offsets");
        size_t freeB = 0, totalB = 0;
        hipMemGetInfo(&freeB, &totalB);
        if ((unsigned long long)bytes + (1ull << 30) > freeB)
            return fail("not enough HBM for the pressure history (" + std::to_string(bytes >> 20) +
                        " MiB for " + std::to_string(ring_) + " steps): use the sparse-emitter mode "
                        "(PVA_OPT_STREAMING_ANALYSIS + PvAmdSetEmitters), which keeps a 64-step ring instead");
        if (!hipOk(hipMalloc((void**)&hist_, (size_t)bytes), "hipMalloc history")) return false;
        deviceBytes_ += bytes;
        if (!dalloc(&unitList_, (size_t)(histPlane_ / 64 + 1), true)) return false;
        if (isSlab()) {
            if (opt_.slabIndex > 0 && !dalloc(&histAbove_, (size_t)T_ * histPitch_, true)) return false;
            if (opt_.slabIndex + 1 < opt_.slabCount && !dalloc(&histEdge_, (size_t)T_ * histPitch_, true)) return false;
        }
    }

    if (opt_.streaming) {
        const size_t nres = (size_t)g_.gx * g_.gy;
        if (!dalloc(&sOnset_, nres, true)) return false;
        for (auto& p : sState_)
            if (!dalloc(&p, nres, true)) return false;
        if (!dalloc(&tileOpen_, (size_t)ntiles, true) || !dalloc(&tileMarks_, (size_t)ntiles, true) ||
            !dalloc(&tileEmit_, (size_t)ntiles, true))
            return false;
        // measured on MI355X (profiles/r03_modeB_fuse.txt): +44 % at 8192^2, +10-14 % at 4096^2 (11 742 tiles), +2 % at 3059^2
        // (6545 tiles), -2 % at 2048^2, -17 / -28 % at 1024^2 / 512^2, where a sweep is launch-bound and the classify +
        // open-tile launches of the front phase weigh more than the ring traffic they save
        const bool fuseWanted = opt_.streamFuse > 0 || (opt_.streamFuse < 0 && ntiles >= 6000);
        streamFuse_ = fuseWanted && openConfigOk(K_, rxi_) && opt_.packed && opt_.merged == 1 && mergedConfigOk(K_, rxi_) &&
                      !stepConfigStacked(K_, rxi_) && opt_.timeKernels == 0;
        if (streamFuse_ && (!dalloc(&classStream_, (size_t)ntiles, true) || !dalloc(&ringOpen_, (size_t)ntiles, true) ||
                            !dalloc(&cellsOpen2_, (size_t)2 * ntiles, true) || !dalloc(&openList_, (size_t)2 * ntiles, true) ||
                            !dalloc(&openCount_, 2, true) || !dalloc(&ringList_, (size_t)ntiles, true) ||
                            !hipOk(hipHostMalloc((void**)&ringHost_, sizeof(int) * (size_t)ntiles), "hipHostMalloc") ||
                            !hipOk(hipHostMalloc((void**)&idleHost_, 2 * sizeof(int)), "hipHostMalloc")))
            return false;
        const char* os = getenv("PLANEVERB_AMD_OPEN_STREAM");  // development knob: 0 = the open half tiles behind the merged launch
        if (streamFuse_ && !(os && atoi(os) == 0)) {
            if (!hipOk(hipStreamCreateWithFlags(&openStream_, hipStreamNonBlocking), "hipStreamCreate")) return false;
            for (auto& e : openEv_)
                if (!hipOk(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) return false;
        }
    }
    if (!hipOk(hipHostMalloc((void**)&dynHost_, sizeof(DynParams)), "hipHostMalloc")) return false;
    if (!hipOk(hipHostMalloc((void**)&outHost_, 8 * sizeof(float)), "hipHostMalloc")) return false;
    if (!hipOk(hipHostMalloc((void**)&statusHost_, 8 * sizeof(int)), "hipHostMalloc")) return false;
    {
        const char* e = getenv("PLANEVERB_AMD_STAMP_TIMINGS");  // 0: HIP events around the stencil and the analysis of every run
        if (!(e && atoi(e) == 0) && !hipOk(hipHostMalloc((void**)&stampsHost_, 4 * sizeof(unsigned long long)), "hipHostMalloc")) return false;
    }
    if (!hipOk(hipHostMalloc((void**)&qCellsHost_, kMaxQueries * sizeof(long long)), "hipHostMalloc")) return false;
    if (!hipOk(hipHostMalloc((void**)&qOutHost_, kMaxQueries * 8 * sizeof(float)), "hipHostMalloc")) return false;
    if (!hipOk(hipHostMalloc((void**)&listHost_, sizeof(int) * (size_t)listCap_), "hipHostMalloc")) return false;
    // row-streaming air segments (opt-in: measured slower than the tile kernels at 4096^2): grids whose
    // configuration has a segment kernel; not with the modes whose launches are cut differently (slabs, row bands,
    // batched / edge-tile kernel, graphs) or record every tile (sparse-emitter ring)
    segWMax_ = segConfigMaxTileColumns(K_, rxi_);
    useSeg_ = segWMax_ > 0 && opt_.segments > 0 && !opt_.streaming && !isSlab() && !opt_.edgeTiles && opt_.merged == 1 &&
              opt_.timeKernels == 0 && opt_.useGraph != 1 && (opt_.useGraph == 2 || ntiles > 4096) &&
              plane * 12 <= (size_t)INT_MAX;
    if (useSeg_) {
        segCap_ = 2 * ntiles + 8;
        if (!dalloc(&segList_, (size_t)segCap_, true)) return false;
        if (!hipOk(hipHostMalloc((void**)&segHost_, sizeof(SegDesc) * (size_t)segCap_), "hipHostMalloc")) return false;
    }

    // Persistent patch kernel for the air tiles (pv_patch.h): the large-grid tile only; one descriptor spans a buffer
    // set's three planes, so they must fit 31 bits
    {
        const bool mergedLaunch = !stepConfigStacked(K_, rxi_) && opt_.merged == 1 && mergedConfigOk(K_, rxi_);
        int want = opt_.patch;
        if (want < 0) want = kDefaultPatch;
        usePatch_ = want > 0 && patchConfigOk(K_, rxi_) && mergedLaunch && opt_.packed && !opt_.streaming && !isSlab() &&
                    !opt_.edgeTiles && opt_.timeKernels == 0 && !useSeg_ && plane * 12 <= (size_t)INT_MAX;
        if (usePatch_) {
            hipDeviceProp_t prop;
            if (!hipOk(hipGetDeviceProperties(&prop, device_), "hipGetDeviceProperties")) return false;
            patchBlocks_ = std::max(8, prop.multiProcessorCount / 8 * 8);
            if ((opt_.patchStrip & 0xff) < 1) opt_.patchStrip |= 1;
            if (std::getenv("PV_PATCH_TRACE") && !dalloc(&patchTrace_, (size_t)8 * 16 * 16, true)) return false;
            opt_.rowBands = 1;
        }
    }

    // Row bands (see enqueueSteps): worth it where a sweep is thousands of tiles; measured on MI355X at 4096^2 / 8192^2
    {
        const bool mergedLaunch = !stepConfigStacked(K_, rxi_) && opt_.merged == 1 && mergedConfigOk(K_, rxi_);
        int want = opt_.rowBands;
        if (want == 0) want = kAutoRowBands;
        if (!mergedLaunch || opt_.streaming || opt_.edgeTiles || opt_.timeKernels > 0) want = 1;
        // a band must be tall enough that only ADJACENT bands share halos: >= 2 tile rows each
        want = std::max(1, std::min(want, geo_.ntx / 2));
        nb_ = want;
        bandRow_.resize((size_t)nb_ + 1);
        for (int b = 0; b <= nb_; ++b) bandRow_[(size_t)b] = (int)((long long)geo_.ntx * b / nb_);
        bandStream_.assign((size_t)nb_, stream_);
        for (int b = 1; b < nb_; ++b)
            if (!hipOk(hipStreamCreateWithFlags(&bandStream_[(size_t)b], hipStreamNonBlocking), "hipStreamCreate"))
                return false;
        bandEv_.resize((size_t)2 * nb_);
        for (auto& e : bandEv_)
            if (!hipOk(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) return false;
        bandListOff_.assign((size_t)nb_ + 1, 0);
        bandListCount_.assign((size_t)nb_, 0);
        if (!dalloc(&dynBandsDev_, (size_t)nb_, true)) return false;
        if (!hipOk(hipHostMalloc((void**)&dynBandsHost_, sizeof(DynParams) * (size_t)nb_), "hipHostMalloc")) return false;
    }

    // Resident kernel (pv_resident.hip): the launch-bound grids whose history window is the whole grid -- the reference's own
    // presets -- run as ONE launch of ntiles workgroups that hand their interiors to their neighbours every K steps.  All of
    // them must be co-resident (they wait for each other inside the launch): the grid is capped by the occupancy query, and
    // concurrent runs of several solvers share a per-device budget (enqueueRun).
    {
        const bool explicitTile = opt.K > 0 || opt.rxi > 0;
        const bool wanted = opt_.resident == 1 || (opt_.resident == 0 && !explicitTile && opt_.useGraph != 1);
        useResident_ = wanted && residentConfigOk(K_, rxi_) && !opt_.streaming && !isSlab() && opt_.timeKernels == 0 &&
                       !opt_.denseHistory && !opt_.edgeTiles && opt_.merged == 1 && histTilesX_ == geo_.ntx &&
                       histTilesY_ == geo_.nty && kGuard >= K_ + residentExtraRows(K_, rxi_) && T_ >= 1;
        if (useResident_) {
            const int cap = residentMaxBlocks(K_, rxi_, device_);
            residentBudget_ = cap * 3 / 4;  // (an occupancy query and the device properties: once, not per 0.3 ms run)
            if (ntiles > std::min(residentBudget_, kResidentMaxTiles)) useResident_ = false;  // (= the run-time budget of enqueueRun)
        }
        if (useResident_ && !dalloc(&resFlags_, (size_t)ntiles + 2, true)) return false;
        if (useResident_) {
            static std::atomic<int> turn{0};
            xcdTarget_ = turn.fetch_add(1) & 7;
            // (validation: an XCD that does not exist -- the first run is then given up by the claim check and repeated)
            if (const char* e = std::getenv("PLANEVERB_AMD_RESIDENT_XCD_TARGET")) xcdTarget_ = std::atoi(e);
            if (const char* e = std::getenv("PLANEVERB_AMD_RESIDENT_XCD")) {  // 0: never the one-XCD mode
                xcdOk_ = std::atoi(e) != 0;
                if (std::atoi(e) > 1) kResidentXcdMaxTiles = std::min(std::atoi(e), 128);  // (measurements: profiles/r06_resident_two_tiles.txt)
            }
        }
    }

    // Fused analysis (pv_fused.hip; experimental build only): the grids whose history window is the whole grid, up to the cell
    // count the four-lane decay-time form serves
    {
        const bool wanted = opt_.fusedAnalysis > 0;  // (opt-in: measured slower than the separate kernels, docs/experiments/fused_analysis.md)
        if (wanted && !fusedAnalysisBuilt())
            return fail("PVA_OPT_FUSED_ANALYSIS = 1 (the one-launch analysis) is an arm of the experimental build of the library");
        useFused_ = wanted && !opt_.streaming && !isSlab() && !opt_.denseHistory && histTilesX_ == geo_.ntx &&
                    histTilesY_ == geo_.nty && histPlane_ <= 98304 && (opt_.rt60Lanes == 0 || opt_.rt60Lanes == 16 || opt_.rt60Lanes == 4) &&
                    fusedAnalysisOk(analyzeArgs(0.f, 0.f));
        // (+ 32 64-bit phase stamps behind the words: development builds, -DPV_FUSED_DEBUG + PLANEVERB_AMD_FUSED_DEBUG=2)
        if (useFused_ && !dalloc(&fusedCtl_, (size_t)kFusedCtlWords + 64, true)) return false;
    }

    warnIfPulseDiffers();
    pulse_ = gaussianPulse(g_);
    pulse_.resize((size_t)std::max(T_, g_.T), 0.f);  // an extended run (numSteps > T) injects nothing after T
    if (!hipOk(hipMemcpyAsync(pulseDev_, pulse_.data(), pulse_.size() * 4, hipMemcpyHostToDevice, stream_),
               "pulse upload"))
        return false;

    mat_.init(g_);
    matHost_.assign((size_t)g_.NX * g_.NY, 0.f);
    betaHost_.assign((size_t)g_.NX * g_.NY, 0);
    byHost_.assign((size_t)g_.NX * g_.NY, 0);
    geometryDirty_ = true;
    if (!applyGeometry()) return false;
    if (!hipOk(hipStreamSynchronize(stream_), "init sync")) return false;

    if (opt.withFreeGrid) {
        if (!computeEfree()) return false;
    }
    // One-XCD hand-off (pv_resident.hip): whether the blocks of a launch are spread over the XCDs as that mode needs is a property of
    // the device / partition mode, found out by a run that fails its claim check (errFlag 4) and is repeated in sync().  Find it out
    // HERE, with a throw-away run of the stencil alone: the live module publishes a run's results before it calls sync(), and its
    // first iteration on each solver would otherwise publish the aborted run's (silent) maps.
    if (useResident_ && xcdOk_ && ntiles <= kResidentXcdMaxTiles && !opt_.skipAnalysis) {
        opt_.skipAnalysis = true;
        const bool ok = enqueueRun(g_.gx / 2, g_.gy / 2, 0.f, 0.f) && sync();
        opt_.skipAnalysis = false;
        if (!ok) return false;
        // (it was not a run: a fresh solver answers "no simulation has run yet" to whoever asks for a window, an impulse response
        // or a history plane, and its timings are empty)
        dynValid_ = false;
        tim_ = SolverTimings{};
        lastReached_ = -1;
    }
    return true;
}

Solver::~Solver() {
    if (stream_) hipStreamSynchronize(stream_);
    if (patchTrace_) {  // development aid: phase stamps of the LAST launch (block 0), cycles relative to the first stamp
        std::vector<long long> t((size_t)8 * 16 * 16);
        hipMemcpyAsync(t.data(), patchTrace_, t.size() * 8, hipMemcpyDeviceToHost, stream_);
        hipStreamSynchronize(stream_);
        long long t0 = 0;
        for (long long v : t)
            if (v && (!t0 || v < t0)) t0 = v;
        for (int w = 0; w < 8; ++w)
            for (int it = 0; it < 16; ++it) {
                const long long* r = &t[(size_t)(w * 16 + it) * 16];
                if (!r[0]) continue;
                std::fprintf(stderr, "trace wave %d tile %2d:", w, it);
                for (int k = 0; k < 8; ++k) std::fprintf(stderr, " %8lld", r[k] ? r[k] - t0 : -1);
                std::fprintf(stderr, "\n");
            }
        hipFree(patchTrace_);
    }
    if (stream2_) hipStreamSynchronize(stream2_);
    for (int i = 0; i < 2; ++i) {
        if (pr_[i]) hipFree(pr_[i]);  // (vx, vy live in the same allocation)
    }
    for (uint8_t* p : nz_)
        if (p) hipFree(p);
    for (float* p : sState_)
        if (p) hipFree(p);
    if (sOnset_) hipFree(sOnset_);
    if (tileOpen_) hipFree(tileOpen_);
    if (tileMarks_) hipFree(tileMarks_);
    if (tileEmit_) hipFree(tileEmit_);
    for (void* p : {(void*)classStream_, (void*)ringOpen_, (void*)cellsOpen2_, (void*)openList_, (void*)openCount_, (void*)ringList_})
        if (p) hipFree(p);
    if (ringHost_) hipHostFree(ringHost_);
    if (idleHost_) hipHostFree(idleHost_);
    releaseResident();
#ifdef PV_RESIDENT_TRACE
    if (useResident_) residentDumpTrace();
#endif
    if (resFlags_) hipFree(resFlags_);
    if (emCells_) hipFree(emCells_);
    if (emTrace_) hipFree(emTrace_);
    void* ptrs[] = {coef_,      matDev_, pulseDev_, hist_,  tileFirst_, tileClass_, generalList_,
                    generalCount_, dynDev_, errFlag_, res8_,     delay_, scratch_, res_, activeCount_, win8_, unitList_, fusedCtl_, labelDev_, nearBox_,
                    histAbove_, histEdge_, tileDead_, deadCount_};
    for (void* p : ptrs)
        if (p) hipFree(p);
    for (size_t b = 1; b < bandStream_.size(); ++b)
        if (bandStream_[b]) {
            hipStreamSynchronize(bandStream_[b]);
            hipStreamDestroy(bandStream_[b]);
        }
    for (auto& e : bandEv_)
        if (e) hipEventDestroy(e);
    for (auto& e : streamEv_)
        if (e) hipEventDestroy(e);
    if (dynBandsDev_) hipFree(dynBandsDev_);
    if (dynBandsHost_) hipHostFree(dynBandsHost_);
    if (dynHost_) hipHostFree(dynHost_);
    if (outHost_) hipHostFree(outHost_);
    if (statusHost_) hipHostFree(statusHost_);
    if (stampsHost_) hipHostFree(stampsHost_);
    if (qCellsHost_) hipHostFree(qCellsHost_);
    if (qOutHost_) hipHostFree(qOutHost_);
    if (listHost_) hipHostFree(listHost_);
    if (segHost_) hipHostFree(segHost_);
    if (segList_) hipFree(segList_);
    for (auto& e : ev_)
        if (e) hipEventDestroy(e);
    for (auto& e : pubEv_)
        if (e) hipEventDestroy(e);
    dropGraph();
    for (auto& e : kev_) hipEventDestroy(e);
    for (auto& e : airDone_) hipEventDestroy(e);
    for (auto& e : genDone_) hipEventDestroy(e);
    if (forkEv_) hipEventDestroy(forkEv_);
    for (auto& e : anaEv_)
        if (e) hipEventDestroy(e);
    for (hipStream_t x : auxStreams_) hipStreamDestroy(x);
    for (hipEvent_t e : openEv_)
        if (e) hipEventDestroy(e);
    if (openStream_) hipStreamDestroy(openStream_);
    queue_.release();
    if (stream2_) hipStreamDestroy(stream2_);
    if (stream_) hipStreamDestroy(stream_);
}

// ----------------------------------------------------------------------------------------------------------------
// geometry
// ----------------------------------------------------------------------------------------------------------------

// A box whose absorption is not finite is refused: NaN is what marks an air cell in the material plane and an air|air face
// in the coefficient plane, so such a wall would silently become air (the reference would spread NaN through its fields
// instead, FDTD.cpp:150-168: neither is a result).
static bool finiteBox(const Box& b) { return std::isfinite(b.R); }

int Solver::addBox(const Box& b) {
    if (!finiteBox(b)) {
        err_ = "geometry with a non-finite absorption";
        return -1;
    }
    int id;
    if (boxFree_.empty()) {
        id = (int)boxTable_.size();
        boxTable_.push_back(b);
        boxUsed_.push_back(1);
    } else {
        id = boxFree_.back();  // LIFO recycling, GeometryManager.cpp:83-85
        boxFree_.pop_back();
        boxTable_[(size_t)id] = b;
        boxUsed_[(size_t)id] = 1;
    }
    mat_.add(b);
    geometryDirty_ = true;
    return id;
}

bool Solver::updateBox(int id, const Box& b) {
    if (id < 0 || id >= (int)boxTable_.size()) return fail("invalid geometry id");
    if (!finiteBox(b)) return fail("geometry with a non-finite absorption");
    mat_.remove(boxTable_[(size_t)id]);  // UpdateObject = Remove(old) then Add(new), GeometryManager.cpp:112-121
    boxTable_[(size_t)id] = b;
    mat_.add(b);
    geometryDirty_ = true;
    return true;
}

bool Solver::removeBox(int id) {
    if (id < 0 || id >= (int)boxTable_.size()) return fail("invalid geometry id");
    mat_.remove(boxTable_[(size_t)id]);
    boxTable_[(size_t)id] = Box{0, 0, 0, 0, 0};  // GeometryManager.cpp:108
    boxUsed_[(size_t)id] = 0;
    boxFree_.push_back(id);
    geometryDirty_ = true;
    return true;
}

int Solver::numBoxes() const {
    int n = 0;
    for (uint8_t u : boxUsed_) n += u;
    return n;
}

std::vector<std::pair<int, Box>> Solver::boxes() const {
    std::vector<std::pair<int, Box>> out;
    for (size_t i = 0; i < boxTable_.size(); ++i)
        if (boxUsed_[i]) out.emplace_back((int)i, boxTable_[i]);
    return out;
}

bool Solver::applyGeometry() {
    if (!geometryDirty_ && mat_.dirtyLo() >= mat_.dirtyHi()) return true;
    const auto t0 = std::chrono::steady_clock::now();
    int lo = mat_.dirtyLo(), hi = mat_.dirtyHi();
    bool airChanged = geometryDirty_;  // did any cell change between air and wall?  (else the air components stand: makeLabels)
    if (lo < hi) {
        const auto& beta = mat_.beta();
        const auto& R = mat_.R();
        // material of every cell of the dirty rows: NaN = air (beta 1), else the wall's admittance Y = (1 - R) / (1 + R)
        // (FDTD.cpp:150,156, in the reference's float arithmetic) -- any number of absorption values, as in the reference
        // (rounds 1-2 palettised them: 127 alive at once)
        for (int x = lo; x < hi; ++x)
            for (int y = 0; y < g_.NY; ++y) {
                const size_t i = (size_t)x * g_.NY + y;
                const float Rv = R[i];
                matHost_[i] = beta[i] ? std::numeric_limits<float>::quiet_NaN() : (1.f - Rv) / (1.f + Rv);
                airChanged = airChanged || betaHost_[i] != (beta[i] ? 1 : 0);
                betaHost_[i] = beta[i] ? 1 : 0;
                byHost_[i] = mat_.by()[i];
            }
        if (!hipOk(hipMemcpyAsync(matDev_ + (size_t)lo * g_.NY, matHost_.data() + (size_t)lo * g_.NY,
                                  (size_t)(hi - lo) * g_.NY * sizeof(float), hipMemcpyHostToDevice, stream_),
                   "material upload"))
            return false;
    }
    launchCoefs(matDev_, coef_, geo_, stream_);
    if (!hipOk(hipMemsetAsync(generalCount_, 0, sizeof(int), stream_), "memset")) return false;
    launchTileClass(K_, rxi_, coef_, tileClass_, generalList_, generalCount_, geo_, stream_,
                    opt_.edgeTiles);
    if (!hipOk(hipMemsetAsync(deadCount_, 0, sizeof(int), stream_), "memset")) return false;
    launchTileDead(coef_, tileDead_, deadCount_, geo_, K_, stream_);
    int count = 0;
    if (!hipOk(hipMemcpyAsync(&numDead_, deadCount_, sizeof(int), hipMemcpyDeviceToHost, stream_), "count copy"))
        return false;
    if (!hipOk(hipMemcpyAsync(&count, generalCount_, sizeof(int), hipMemcpyDeviceToHost, stream_), "count copy"))
        return false;
    if (!hipOk(hipStreamSynchronize(stream_), "geometry sync")) return false;
    // (copies on the solver's own stream: a synchronous hipMemcpy runs on the LEGACY stream, which the runtime refuses while
    // any other host thread is capturing a run graph -- "operation would make the legacy stream depend on a capturing blocking
    // stream" -- and which invalidates that thread's capture: the live module's worker beside a caller's own solver)
    wallTiles_.resize((size_t)count);
    if (count > 0 &&
        !hipOk(hipMemcpyAsync(wallTiles_.data(), generalList_, sizeof(int) * (size_t)count, hipMemcpyDeviceToHost, stream_),
               "list copy"))
        return false;
    tileClassHost_.resize((size_t)geo_.ntx * geo_.nty);
    if (!hipOk(hipMemcpyAsync(tileClassHost_.data(), tileClass_, tileClassHost_.size(), hipMemcpyDeviceToHost, stream_),
               "class copy") ||
        !hipOk(hipStreamSynchronize(stream_), "geometry sync"))
        return false;
    std::sort(wallTiles_.begin(), wallTiles_.end());
    {
        // Scenes with many wall tiles (>= 8 % general: the 25 m rooms at 4096^2 / 8192^2 have 13-16 %) take the merged kernel
        // whose general arm is the packed one also at K = 12 (launchStep, kStepGeneralPacked); decided per geometry, so a
        // captured run graph never mixes the two.  PLANEVERB_AMD_GENERAL_PACKED = 0 / 1: never / always.
        const char* e = getenv("PLANEVERB_AMD_GENERAL_PACKED");
        const long long nt = (long long)geo_.ntx * geo_.nty;
        stepWhich_ = 4 | (((e ? atoi(e) != 0 : (long long)count * 100 >= 8 * nt)) ? kStepGeneralPacked : 0);
    }
    // (an update that only changes absorption values -- or re-adds a box where it was -- leaves the air components as they are: no
    // flood fill over the grid, no upload: 15-20 ms of a 1024^2 live iteration with moving absorbers)
    if ((airChanged || !labelsValid_) && !makeLabels()) return false;
    mat_.clearDirty();
    geometryDirty_ = false;
    planesDirty_ = true;  // a tile that is dead now may hold an earlier scene's fields
    dynValid_ = false;
    dropGraph();  // tile classes / list capacity may have changed
    tim_.geometryMs =
        std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

// 4-connected components of the air cells (AnalyzeArgs::labels), on the host from the rasteriser's own beta plane: a flood fill
// per component.  Small grids only (the reference's presets, the 512^2 configurations): O(cells) per geometry change.
bool Solver::makeLabels() {
    const size_t n = (size_t)g_.NX * g_.NY;
    if (isSlab() || opt_.streaming || opt_.skipAnalysis || n > (size_t)PV_LABEL_MAX_CELLS) return labelsValid_ = true;
    if (!labelDev_ && !dalloc(&labelDev_, n, false)) return false;
    labelHost_.assign(n, -1);
    std::vector<int> stack;
    int next = 0;
    const int NX = g_.NX, NY = g_.NY;
    for (size_t seed = 0; seed < n; ++seed) {
        if (!betaHost_[seed] || labelHost_[seed] >= 0) continue;
        labelHost_[seed] = next;
        stack.push_back((int)seed);
        while (!stack.empty()) {
            const int i = stack.back();
            stack.pop_back();
            const int x = i / NY, y = i - x * NY;
            const int nb[4] = {x > 0 ? i - NY : -1, x + 1 < NX ? i + NY : -1, y > 0 ? i - 1 : -1, y + 1 < NY ? i + 1 : -1};
            for (int j : nb)
                if (j >= 0 && betaHost_[(size_t)j] && labelHost_[(size_t)j] < 0) {
                    labelHost_[(size_t)j] = next;
                    stack.push_back(j);
                }
        }
        ++next;
    }
    labelsValid_ = hipOk(hipMemcpyAsync(labelDev_, labelHost_.data(), n * sizeof(int), hipMemcpyHostToDevice, stream_), "label upload") &&
                   hipOk(hipStreamSynchronize(stream_), "label sync");
    return labelsValid_;
}

// ----------------------------------------------------------------------------------------------------------------
// FreeGrid
// ----------------------------------------------------------------------------------------------------------------

bool Solver::freeFieldEnergyAt(int cellX, int cellY, int n, float r, float* out) {
    const int hr = cellX + geo_.G - dynCur_.histRow0, hc = cellY + geo_.G - dynCur_.histCol0;
    const int hti = hr / rxi_, htj = hc / wi_;  // tile-major history planes (pv_kernels.hip: histOffset)
    const long long off = ((long long)(hti * histTilesY_ + htj) * rxi_ + (hr - hti * rxi_)) * wi_ + (hc - htj * wi_);
    launchEfree(hist_, histPlane_, off, n, r, scratch_, stream_);
    if (!hipOk(hipMemcpyAsync(out, scratch_, 4, hipMemcpyDeviceToHost, stream_), "efree copy")) return false;
    return hipOk(hipStreamSynchronize(stream_), "efree sync");
}

// FreeGrid::FreeGrid / SimulateFreeFieldEnergy (FreeGrid.cpp:6-34,71-94): a run on an EMPTY grid of the same
// config, source at the centre cell, energy of the first nFree samples at the cell (int)(1/dx) to its +x side.
// Only nFree samples are read, so by causality any sub-grid whose edges are more than nFree cells from both cells
// gives bit-identical samples; large grids therefore simulate a window instead of the whole plane.
bool Solver::computeEfree() {
    const int lx0 = g_.gx / 2, ly0 = g_.gy / 2;         // FreeGrid.cpp:78-79
    const int ex = lx0 + (int)(1.f / g_.dx), ey = ly0;  // FreeGrid.cpp:80-81
    // the listener is passed in metres and truncated again by GenerateResponse (FreeGrid.cpp:84, FDTD.cpp:97-98)
    int lcx, lcy;
    listenerCell(g_, (float)lx0 * g_.dx, (float)ly0 * g_.dx, &lcx, &lcy);
    const int n = g_.nFree;
    const float r = (float)(ex - lx0) * g_.dx;  // FreeGrid.cpp:89
    if (n > g_.T) return fail("free-field window longer than the impulse response");

    const int margin = n + 8;
    const int spanX = ex - lcx, spanY = ey - lcy;  // >= 0; the re-truncation can move the source by one cell
    const bool window = (lcx - margin > 0) && (ex + margin < g_.gx) && (lcy - margin > 0) && (ey + margin < g_.gy) &&
                        spanX >= 0 && spanY >= 0;
    GridSpec fs;
    int sx, sy, qx, qy;
    if (window) {
        fs = makeGridSpecCells(2 * margin + spanX + 1, 2 * margin + spanY + 1, g_.res);
        sx = margin;
        sy = margin;
        qx = margin + spanX;
        qy = margin + spanY;
    } else {
        fs = makeGridSpecCells(g_.gx, g_.gy, g_.res);
        sx = lcx;
        sy = lcy;
        qx = ex;
        qy = ey;
    }
    SolverOptions o;
    o.withFreeGrid = false;  // (tile configuration: the child's own default for its size)
    o.skipAnalysis = true;
    o.numSteps = roundUp(n, 8);
    std::string e;
    Solver* f = Solver::create(fs, device_, o, &e);
    if (!f) return fail("free grid: " + e);
    bool ok = f->runCells(sx, sy, 0.f, 0.f, true) && f->freeFieldEnergyAt(qx, qy, n, r, &efree_);
    if (!ok) err_ = "free grid: " + f->err_;
    delete f;
    hipSetDevice(device_);
    return ok;
}

// ----------------------------------------------------------------------------------------------------------------
// run
// ----------------------------------------------------------------------------------------------------------------

// first tile row of the WHOLE grid's history window for a listener in cell row lcx
int Solver::globalWindowTileRow0(int lcx) const {
    const int reach = T_ + 2 + K_;
    if (histTilesXG_ >= ntxG_) return 0;
    return std::min(std::max(floorDiv(std::min(std::max(lcx, 0), g_.gx) - reach, rxi_), 0), ntxG_ - histTilesXG_);
}

bool Solver::prepareDyn(int lcx, int lcy, bool withPulse, bool banded) {
    bandedRun_ = banded && nb_ > 1;
    DynParams d{};
    const bool inside = withPulse && lcx >= 0 && lcx <= g_.gx && lcy >= 0 && lcy <= g_.gy;
    d.lrow = inside ? lcx - x0_ + geo_.G : -100000;  // (a slab: possibly far outside its own rows)
    d.lcol = inside ? lcy + geo_.G : -100000;
    // history window in tiles, centred on the listener and clamped to the grid
    // first window tile = the tile holding the lowest reachable row / column (init() sized the window so that
    // histTiles * tile >= 2*reach + tile, i.e. it then also covers listener + reach), clamped into the grid
    int tx0 = 0, ty0 = 0;
    const int reach = T_ + 2 + K_;
    {
        // the WHOLE grid's window first; this solver records its part of it (a slab's own rows always hold everything
        // of the window that falls into the slab: init() gave it min(own tile rows, window tile rows) planes)
        tx0 = std::min(std::max(globalWindowTileRow0(lcx) - tileRow0_, 0), geo_.ntx - histTilesX_);
    }
    if (histTilesY_ < geo_.nty)
        ty0 = std::min(std::max(floorDiv(std::min(std::max(lcy, 0), g_.gy) - reach, wi_), 0), geo_.nty - histTilesY_);
    d.histTileX0 = tx0;
    d.histTileY0 = ty0;
    d.histTilesX = histTilesX_;
    d.histTilesY = histTilesY_;
    d.histRow0 = geo_.G + tx0 * rxi_;
    d.histCol0 = geo_.G + ty0 * wi_;
    dynCur_ = d;
    *dynHost_ = d;

    // general-kernel work list = wall/edge tiles + every tile whose loaded region holds the listener
    const int rowsT = rxi_ + 2 * K_ + (stepConfigStacked(K_, rxi_) ? stepConfigExtraRows(K_, rxi_) : 0);
    int n = 0;
    for (int t : wallTiles_) listHost_[n++] = t;
    if (inside) {
        const int a0 = d.lrow - (geo_.G - K_), b0 = d.lcol - (geo_.G - K_);
        const int tiLo = std::max(0, floorDiv(a0 - rowsT, rxi_) + 1), tiHi = std::min(geo_.ntx - 1, floorDiv(a0, rxi_));
        const int tjLo = std::max(0, floorDiv(b0 - 64, wi_) + 1), tjHi = std::min(geo_.nty - 1, floorDiv(b0, wi_));
        for (int ti = tiLo; ti <= tiHi; ++ti)
            for (int tj = tjLo; tj <= tjHi; ++tj) {
                const int t = ti * geo_.nty + tj;
                if (tileClassHost_[(size_t)t] != 1) listHost_[n++] = t;  // air and edge tiles holding the listener
            }
    }
    numGeneral_ = n;
    dynCur_.numGeneral = n;
    dynHost_->numGeneral = n;
    if (const char* v = getenv("PLANEVERB_AMD_VERBOSE"); v && atoi(v) > 0)
        std::fprintf(stderr, "[planeverb_amd] %d x %d tiles (K %d, %d rows): %d general, %d dead\n", geo_.ntx, geo_.nty, K_, rxi_, n, numDead_);
    segActive_ = useSeg_ && !bandedRun_;
    numSeg_ = 0;
    if (segActive_) buildSegments(n);
    segActive_ = segActive_ && numSeg_ > 0;
    dynCur_.numSeg = numSeg_;
    dynHost_->numSeg = numSeg_;
    if (bandedRun_) {
        // the list, band by band, as LOCAL tile ids of each band's own tile-row range; each band gets its own view of
        // the run parameters (the kernels then see a grid that starts at the band's first tile row)
        std::vector<int> all(listHost_, listHost_ + n);
        std::stable_sort(all.begin(), all.end());
        int pos = 0, b = 0;
        bandListOff_[0] = 0;
        for (int t : all) {
            const int ti = t / geo_.nty;
            while (ti >= bandRow_[(size_t)b + 1]) {
                ++b;
                bandListOff_[(size_t)b] = pos;
            }
            listHost_[pos++] = (ti - bandRow_[(size_t)b]) * geo_.nty + (t - ti * geo_.nty);
        }
        while (b < nb_) bandListOff_[(size_t)++b] = pos;
        for (int k = 0; k < nb_; ++k) {
            bandListCount_[(size_t)k] = bandListOff_[(size_t)k + 1] - bandListOff_[(size_t)k];
            DynParams v = d;
            const int r0 = bandRow_[(size_t)k];
            if (inside) v.lrow -= r0 * rxi_;
            v.histTileX0 -= r0;
            v.histRow0 -= r0 * rxi_;
            v.numGeneral = bandListCount_[(size_t)k];
            dynBandsHost_[k] = v;
        }
    }
    dynValid_ = true;
    return true;
}

// Row-streaming air segments of this run (pv_seg.h): every air tile that is not on the general list (walls, edges, the
// tiles around the listener: listHost_[0, listed)) is covered by exactly one segment (planSegments, pv_core.cpp);
// pv_step_seg_kernel gives each XCD a contiguous eighth of the sorted list.
void Solver::buildSegments(int listed) {
    const int ntx = geo_.ntx, nty = geo_.nty;
    std::vector<uint8_t> air((size_t)ntx * nty);
    for (size_t t = 0; t < air.size(); ++t) air[t] = tileClassHost_[t] == 0;
    for (int i = 0; i < listed; ++i) air[(size_t)listHost_[i]] = 0;
    const std::vector<SegRect> segs = planSegments(air.data(), ntx, nty, rxi_, segWMax_, opt_.segments > 0 ? opt_.segments : 1024);
    int n = 0;
    for (const SegRect& r : segs) {
        if (n >= segCap_) break;
        segHost_[n++] = SegDesc{r.row0, r.nrows, r.tj0, r.w};
    }
    numSeg_ = n;
}

// prepareDyn() left the run's parameters in pinned host memory; this launch moves them to HBM and resets the
// per-tile bookkeeping (see pv_begin_run_kernel).  It is captured into the run graph with the step launches.
// Dead tiles are never written by a run, so both buffer sets must hold zeros there: re-established here whenever the
// planes may hold anything else (new geometry, setFields / raw stepping).  Scenes without dead tiles pay nothing.
bool Solver::zeroPlanesIfNeeded() {
    if (numDead_ == 0 || !planesDirty_) return true;
    const size_t bytes = (size_t)geo_.rows * geo_.pitch * 4;
    for (int i = 0; i < 2; ++i)
        if (!hipOk(hipMemsetAsync(pr_[i], 0, bytes, stream_), "plane clear") ||
            !hipOk(hipMemsetAsync(vx_[i], 0, bytes, stream_), "plane clear") ||
            !hipOk(hipMemsetAsync(vy_[i], 0, bytes, stream_), "plane clear"))
            return false;
    planesDirty_ = false;
    return true;
}

void Solver::enqueueBeginRun(bool resetTiles) {
    BeginArgs b{};
    b.dynHost = dynHost_;
    b.dyn = dynDev_;
    b.listHost = listHost_;
    b.list = generalList_;
    b.tileFirst = resetTiles ? tileFirst_ : nullptr;
    b.nz0 = nz_[0];
    b.nz1 = nz_[1];
    b.tileOpen = opt_.streaming ? tileOpen_ : nullptr;
    b.errFlag = errFlag_;
    b.ntiles = geo_.ntx * geo_.nty;
    b.tileFirstInit = opt_.denseHistory ? 0 : INT_MAX;
    b.listCap = listCap_;
    b.segHost = segActive_ ? segHost_ : nullptr;
    b.seg = segList_;
    b.segCap = segCap_;
    b.dynBandsHost = bandedRun_ ? dynBandsHost_ : nullptr;
    b.dynBands = dynBandsDev_;
    b.nbands = bandedRun_ ? nb_ : 0;
    b.zeroWords = resFlags_;
    b.nZero = resFlags_ ? geo_.ntx * geo_.nty + 2 : 0;
    launchBeginRun(b, stream_);
}

// the part of a launch's arguments that does not change from launch to launch
StepArgs Solver::baseStepArgs(bool withPulse, bool record) const {
    StepArgs a{};
    a.coef = coef_;
    a.pulse = pulseDev_;
    a.hist = hist_;
    a.tileFirst = tileFirst_;
    a.tileClass = tileClass_;
    // dead tiles are skipped only by runs that start from zero fields (record = a run; raw stepping starts from the
    // caller's fields, which may be anything inside a wall)
    a.tileDead = (record && numDead_ > 0) ? tileDead_ : nullptr;
    a.generalList = generalList_;
    a.numGeneral = launchCap_;
    a.segList = segActive_ ? segList_ : nullptr;
    a.numSeg = segActive_ ? numSeg_ : 0;
    a.dyn = dynDev_;
    a.tileOpen = opt_.streaming ? tileOpen_ : nullptr;
    a.errFlag = errFlag_;
    a.histPlane = histPlane_;
    a.planeBytes = (long long)geo_.rows * geo_.pitch * 4;
    a.histPitch = histPitch_;
    a.pitch = geo_.pitch;
    a.G = geo_.G;
    a.ntx = geo_.ntx;
    a.nty = geo_.nty;
    a.ntiles = geo_.ntx * geo_.nty;
    a.gx = g_.gx;
    a.gy = g_.gy;
    a.bandRows = ceilDiv(geo_.ntx, 8);
    a.tileOrder = opt_.tileOrder;
    a.patchStrip = opt_.patchStrip;
    a.patchTrace = patchTrace_;
    a.packed = opt_.packed ? 1 : 0;
    a.withPulse = withPulse ? 1 : 0;
    a.record = record ? 1 : 0;
    a.dense = opt_.denseHistory ? 1 : 0;
    a.courant = g_.courant;
    return a;
}

// the per-launch part: buffer sets, step range, per-tile flag planes (li = index of the launch within the run)
void Solver::setLaunchArgs(StepArgs& a, int t0, int k, bool firstOfRun, int li) const {
    a.prIn = pr_[cur_];
    a.vxIn = vx_[cur_];
    a.vyIn = vy_[cur_];
    a.prOut = pr_[cur_ ^ 1];
    a.vxOut = vx_[cur_ ^ 1];
    a.vyOut = vy_[cur_ ^ 1];
    a.t0 = t0;
    a.histSlot = opt_.streaming ? t0 % ring_ : t0;
    a.nsteps = k;
    a.inBytes = firstOfRun ? 0 : (int)a.planeBytes;
    a.nzIn = nz_[li & 1];
    a.nzOut = nz_[(li & 1) ^ 1];
    // bit 0: this launch walks the XCDs' tiles backwards (odd launches of a run); bit 1: 2 x 4 regions instead of 8 strips
    a.sweepReverse = opt_.tileOrder == 3 ? (((opt_.alternateSweeps == 1) ? (li & 1) : 0) | (opt_.xcdRegions == 1 ? 2 : 0)) : 0;
}

bool Solver::bandsActive() const { return bandedRun_; }

// the launch arguments of band b: the same kernels, shown a grid that starts at the band's first tile row
StepArgs Solver::bandStepArgs(const StepArgs& a, int b) const {
    StepArgs v = a;
    const int r0 = bandRow_[(size_t)b], nr = bandRow_[(size_t)b + 1] - r0;
    const long long off = (long long)r0 * rxi_ * geo_.pitch;  // floats
    v.prIn += off;
    v.vxIn += off;
    v.vyIn += off;
    v.prOut += off;
    v.vxOut += off;
    v.vyOut += off;
    v.coef += off;
    v.tileFirst += (long long)r0 * geo_.nty;
    v.tileClass += (long long)r0 * geo_.nty;
    if (v.tileDead) v.tileDead += (long long)r0 * geo_.nty;
    if (v.tileOpen) v.tileOpen += (long long)r0 * geo_.nty;
    v.nzIn += (long long)r0 * geo_.nty;
    v.nzOut += (long long)r0 * geo_.nty;
    v.generalList += bandListOff_[(size_t)b];
    v.numGeneral = bandListCount_[(size_t)b];
    v.dyn = dynBandsDev_ + b;
    v.ntx = nr;
    v.ntiles = nr * geo_.nty;
    v.bandRows = ceilDiv(nr, 8);
    v.planeBytes = a.planeBytes - off * 4;
    if (v.inBytes != 0) v.inBytes = (int)v.planeBytes;
    return v;
}

bool Solver::enqueueSteps(int firstStep, int nsteps, bool withPulse, bool record, bool fromZero) {
    StepArgs a = baseStepArgs(withPulse, record);
    // Two streams: the air-tile kernel (the bulk of the grid) on stream_, the general-tile kernel (walls, edges,
    // listener: few tiles, latency-bound) concurrently on stream2_.  Both read buffer set `cur` and write disjoint
    // tiles of the other set, so launch i+1 of EITHER kernel must wait for launch i of BOTH (RAW on the halos it
    // reads, WAR on the tiles it overwrites): one event per kernel per launch.
    // merged: one launch per K steps on one stream (no cross-stream hand-shake); not with the streaming kernel
    const bool mergedLaunch =
        stepConfigStacked(K_, rxi_) || (opt_.merged == 1 && mergedConfigOk(K_, rxi_));
    const bool two = launchCap_ > 0 && !mergedLaunch;
    const int nl = ceilDiv(nsteps, K_);
    if (two) {
        auto grow = [&](std::vector<hipEvent_t>& v) {
            while ((int)v.size() < nl) {
                hipEvent_t e;
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
                v.push_back(e);
            }
            return true;
        };
        if (!grow(airDone_) || !grow(genDone_)) return fail("hipEventCreate failed");
        hipEventRecord(forkEv_, stream_);  // everything enqueued so far (reset, dyn upload, list upload)
        hipStreamWaitEvent(stream2_, forkEv_, 0);
    }
    // A run starts from zero fields.  The first launch gets zero-extent input descriptors (its loads return 0) and
    // overwrites the other buffer set completely, so no reset pass over the planes exists.  (No hipMemsetAsync here on
    // purpose: as nodes of a replayed graph the three large plane memsets were observed to be skipped after a
    // hipDeviceSynchronize on ROCm 7.0's runtime.)
    (void)fromZero;
    if (mergedLaunch && bandsActive()) {
        // Row bands: sweep n of band b reads buffer set A (its own rows + K halo rows of bands b-1, b+1) and writes its
        // rows of set B.  So sweep n+1 of band b -- which reads set B around band b and overwrites set A's rows of band b
        // -- may start as soon as sweep n of bands b-1, b, b+1 is complete: its own stream orders it behind band b, two
        // event waits behind the neighbours.  Nothing ever waits for a whole sweep, so the chip does not drain between
        // the launches of one run.  (Bands two apart never touch each other's rows: a band is >= 2 tile rows >= K rows.)
        hipEventRecord(forkEv_, stream_);  // begin-run kernel, dyn / list upload
        for (int b = 1; b < nb_; ++b) hipStreamWaitEvent(bandStream_[(size_t)b], forkEv_, 0);
        int done = 0, li = 0;
        while (done < nsteps) {
            const int k = std::min(K_, nsteps - done);
            setLaunchArgs(a, firstStep + done, k, fromZero && done == 0, li);
            for (int b = 0; b < nb_; ++b) {
                hipStream_t sb = bandStream_[(size_t)b];
                if (li > 0) {
                    if (b > 0) hipStreamWaitEvent(sb, bandEv_[(size_t)2 * (b - 1) + ((li - 1) & 1)], 0);
                    if (b + 1 < nb_) hipStreamWaitEvent(sb, bandEv_[(size_t)2 * (b + 1) + ((li - 1) & 1)], 0);
                }
                launchStep(K_, rxi_, bandStepArgs(a, b), sb, stepWhich_);
                hipEventRecord(bandEv_[(size_t)2 * b + (li & 1)], sb);
            }
            cur_ ^= 1;
            done += k;
            ++li;
            ++tim_.stepLaunches;
        }
        for (int b = 1; b < nb_; ++b) hipStreamWaitEvent(stream_, bandEv_[(size_t)2 * b + ((li - 1) & 1)], 0);  // join
        return hipOk(hipGetLastError(), "step launch");
    }
    int done = 0, li = 0;
    while (done < nsteps) {
        const int k = std::min(K_, nsteps - done);
        setLaunchArgs(a, firstStep + done, k, fromZero && done == 0, li);
        if (opt_.timeKernels > 0) {  // 4 timing events per sampled launch: air begin/end on stream_, general begin/end
            while ((int)kev_.size() < kevUsed_ + 4) {
                hipEvent_t e;
                if (!hipOk(hipEventCreate(&e), "hipEventCreate")) return false;
                kev_.push_back(e);
            }
        }
        // only full K-step launches are sampled, so that duration and algorithmic bytes refer to the same work
        hipEvent_t* te = (opt_.timeKernels > 0 && k == K_ && li % opt_.timeKernels == 0) ? &kev_[(size_t)kevUsed_] : nullptr;
        if (mergedLaunch) {
            if (te) hipEventRecord(te[0], stream_);
            // the segment kernel always advances exactly K levels; a run's short last launch takes the tile kernel
            if (a.segList && k == K_) {
                launchStepSeg(K_, rxi_, a, stream_);
            } else if (streamFuse_ && record && !fuseIdle_) {
                // sparse-emitter mode, forward sums inside the stencil (pv_stream.h): classify, the merged launch without
                // the open air tiles, the open half tiles
                // The open half tiles go out on a stream of their own, beside the merged launch of the same sweep (they touch
                // disjoint tiles of the same buffer sets): behind it they are one to four rounds of 246-register waves with
                // nothing to cover their load and store phases.  The next sweep's classify pass waits for them.
                const ClassifyArgs c = classifyArgs(a, li, withPulse);
                if (openPending_) joinOpen(li - 1);
                launchStreamClassify(c, stream_);
                if (openStream_) {
                    hipEventRecord(openEv_[li & 1], stream_);
                    hipStreamWaitEvent(openStream_, openEv_[li & 1], 0);
                }
                StepArgs a2 = a;
                a2.tileClass = classStream_;
                a2.tileOpen = ringOpen_;
                launchStep(K_, rxi_, a2, stream_, stepWhich_);
                OpenArgs o{};
                o.sOnset = sOnset_;
                o.sEdry = sState_[0];
                o.sFx = sState_[1];
                o.sFy = sState_[2];
                o.cellsOpen2 = cellsOpen2_;
                o.openList = openList_;
                o.openCount = openCount_ + (li & 1);
                o.nextCount = openCount_ + ((li & 1) ^ 1);
                o.nDir = g_.nDir;
                o.nDry = g_.nDry;
                o.gxRes = g_.gx;
                o.gyRes = g_.gy;
                launchStepOpen(K_, rxi_, a2, o, openStream_ ? openStream_ : stream_);
                if (openStream_) {
                    hipEventRecord(openEv_[2 + (li & 1)], openStream_);
                    openPending_ = true;
                }
            } else if (usePatch_) {
                // general tiles in their 4-wave blocks, then the air tiles by the resident workgroups (disjoint tiles of
                // the same output set; neither reads what the other writes)
                launchStep(K_, rxi_, a, stream_, 16);
                launchStepPatch(K_, rxi_, a, patchBlocks_, stream_);
            } else {
                launchStep(K_, rxi_, a, stream_, stepWhich_);
            }
            if (te) {
                hipEventRecord(te[1], stream_);
                hipEventRecord(te[2], stream_);
                hipEventRecord(te[3], stream_);
            }
        } else if (two) {
            if (li > 0) {
                hipStreamWaitEvent(stream_, genDone_[(size_t)li - 1], 0);
                hipStreamWaitEvent(stream2_, airDone_[(size_t)li - 1], 0);
            }
            // air first, then general (launching the general kernel first was measured 10 % slower)
            if (te) hipEventRecord(te[0], stream_);
            launchStep(K_, rxi_, a, stream_, 1);
            if (te) hipEventRecord(te[1], stream_);
            hipEventRecord(airDone_[(size_t)li], stream_);
            if (te) hipEventRecord(te[2], stream2_);
            launchStep(K_, rxi_, a, stream_, 2, stream2_);
            if (te) hipEventRecord(te[3], stream2_);
            hipEventRecord(genDone_[(size_t)li], stream2_);
        } else {
            if (te) hipEventRecord(te[0], stream_);
            launchStep(K_, rxi_, a, stream_, 1);
            if (te) {
                hipEventRecord(te[1], stream_);
                hipEventRecord(te[2], stream_);
                hipEventRecord(te[3], stream_);
            }
        }
        if (te) kevUsed_ += 4;
        cur_ ^= 1;
        done += k;
        ++li;
        ++tim_.stepLaunches;
    }
    if (two) hipStreamWaitEvent(stream_, genDone_[(size_t)li - 1], 0);  // join
    if (openPending_) joinOpen(li - 1);  // the last sweep's open half tiles
    return hipOk(hipGetLastError(), "step launch");
}

// the stream of the merged launches waits for sweep li's open half tiles
void Solver::joinOpen(int li) {
    hipStreamWaitEvent(stream_, openEv_[2 + (li & 1)], 0);
    openPending_ = false;
}

ClassifyArgs Solver::classifyArgs(const StepArgs& a, int li, bool withPulse) const {
    ClassifyArgs c{};
    c.tileClass = tileClass_;
    c.tileEmit = tileEmit_;
    c.tileOpenRing = tileOpen_;
    c.nzPrev = a.nzIn;
    c.cellsOpen2 = cellsOpen2_;
    c.dyn = dynDev_;
    c.classOut = classStream_;
    c.ringOpenOut = ringOpen_;
    c.nzNext = a.nzOut;
    c.openList = openList_;
    c.openCount = openCount_ + (li & 1);
    c.ntx = geo_.ntx;
    c.nty = geo_.nty;
    c.G = geo_.G;
    c.K = K_;
    c.rxi = rxi_;
    c.wi = wi_;
    c.rows = rxi_ + 2 * K_;
    c.withPulse = withPulse ? 1 : 0;
    return c;
}

AnalyzeArgs Solver::analyzeArgs(float lx, float lz) const {
    AnalyzeArgs a{};
    a.hist = hist_;
    a.coef = coef_;
    a.tileFirst = tileFirst_;
    a.dyn = dynDev_;
    a.out = res_;
    a.resN = (long long)std::max(lgx_, 1) * g_.gy;
    a.delay = delay_;
    a.histPlane = histPlane_;
    a.histPitch = histPitch_;
    a.pitch = geo_.pitch;
    a.G = geo_.G;
    a.gx = lgx_;
    a.gy = g_.gy;
    a.x0 = x0_;
    a.histAbove = histAbove_;
    a.rxi = rxi_;
    a.wi = wi_;
    a.nty = geo_.nty;
    a.winRows = histTilesX_ * rxi_;
    a.winCols = histTilesY_ * wi_;
    a.activeCount = activeCount_;
    a.unitList = unitList_;
    a.dirScratch = reinterpret_cast<int*>(scratch_);
    // Listener direction: pointer jumping (six tiny launches at T = 1187, path-length independent) wherever a walk can be
    // long or launches are cheap -- wide windows, and every grid small enough to run as one replayed graph, where the plain
    // walk was 10-13 % of a run (191^2, T = 1187: 241 us of 1.83 ms; profiles/r03_presets.txt).  A small room inside a
    // large grid keeps the plain walk (short walks, one launch instead of five queued behind another run's stencil), and
    // so does the dense-history (validation) mode.
    const bool smallWindow = a.winRows <= 256 && a.winCols <= 256;
    a.dirJump = (!opt_.denseHistory && !(smallWindow && geo_.ntx * geo_.nty > 4096)) ? 1 : 0;
    a.rt60Lanes = (opt_.rt60Lanes == 16 || opt_.rt60Lanes == 4 || opt_.rt60Lanes == 1) ? opt_.rt60Lanes : 0;
    // the lane-per-cell form of the decay-time pass gets its launch where it can be the one that runs: a window that holds that
    // many cells, and -- once a run of this solver has been read back -- a run that reached half as many (a closed room in a large
    // grid never does: one launch less behind every run; without the launch the four-lane form takes any number of cells)
    a.rt60Tile = (a.rt60Lanes == 1 || (a.rt60Lanes == 0 && histPlane_ > kRt60TileMinCells &&
                                       (lastReached_ < 0 || lastReached_ > kRt60TileMinCells / 2))) ? 1 : 0;
    a.T = T_;
    a.nDir = g_.nDir;
    a.nDry = g_.nDry;
    a.nWet = g_.nWet;
    a.nCut = g_.nCut;
    a.fs = g_.fs;
    a.res = g_.res;
    a.dx = g_.dx;
    a.courant = g_.courant;
    a.efree = efree_;
    a.lx = lx;
    a.lz = lz;
    listenerCellRecip(g_, lx, lz, &a.lcx, &a.lcy);
    a.lazyFar = lazyFar_ ? 1 : 0;
    a.labels = labelDev_;
    a.stamp = stampTimed_ ? stampsHost_ : nullptr;
    a.labelNY = g_.NY;
    a.wholeWindow = (!isSlab() && !opt_.streaming && histTilesX_ == geo_.ntx && histTilesY_ == geo_.nty) ? 1 : 0;
    a.prevR0 = farWin_.r0;
    a.prevC0 = farWin_.c0;
    a.prevNR = farWin_.nr;
    a.prevNC = farWin_.nc;
    if (useNearBox_ && !a.wholeWindow && !useFused_) {  // (nearPar_ names the box of the run analysed LAST: this run takes the other)
        a.box = nearBox_ + 4 * (nearPar_ ^ 1);
        a.prevBox = nearBox_ + 4 * nearPar_;
    }
    a.ring = opt_.streaming ? ring_ : 0;
    a.sOnset = sOnset_;
    a.sEdry = sState_[0];
    a.sFx = sState_[1];
    a.sFy = sState_[2];
    a.sVx = sState_[3];
    a.sVy = sState_[4];
    a.fuseClass = streamFuse_ ? tileClass_ : nullptr;
    a.fuseEmit = tileEmit_;
    a.fuseK = K_;
    a.ringList = streamFuse_ ? ringList_ : nullptr;
    a.numRing = numRing_;
    a.emCells = emCells_;
    a.emTrace = emTrace_;
    a.numEmitters = numEmitters_;
    a.tileOpenOut = tileMarks_;
    return a;
}

Solver::Block Solver::curWindow() const {
    Block b;
    b.r0 = dynCur_.histRow0 - geo_.G;
    b.c0 = dynCur_.histCol0 - geo_.G;
    b.nr = std::max(0, std::min(histTilesX_ * rxi_, g_.gx - b.r0));
    b.nc = std::max(0, std::min(histTilesY_ * wi_, g_.gy - b.c0));
    return b;
}

FarInfo Solver::farInfo() const {
    FarInfo f{};
    f.on = (lazyFar_ && dynValid_ && !farDirValid_) ? 1 : 0;
    f.r0 = farWin_.r0;
    f.c0 = farWin_.c0;
    f.nr = farWin_.nr;
    f.nc = farWin_.nc;
    f.box = (f.on && nearBoxValid_) ? nearBox_ + 4 * nearPar_ : nullptr;
    f.gy = g_.gy;
    f.lx = lastLx_;
    f.lz = lastLz_;
    f.dx = g_.dx;
    return f;
}

// whole-map readers: give the far cells of the last run their listener direction (Analyzer.cpp:365-391,415-428)
bool Solver::ensureFarDirections() {
    if (!lazyFar_ || farDirValid_ || !dynValid_) return true;
    launchFarDirections(res_, (long long)g_.gx * g_.gy, farInfo(), stream_);
    farDirValid_ = true;
    return hipOk(hipGetLastError(), "far directions");
}

// the analysis of the run enqueued on stream_ (history recorded, dynCur_ = its parameters)
void Solver::enqueueAnalysis(float lx, float lz) {
    const AnalyzeArgs a = analyzeArgs(lx, lz);
    const bool carry = carryFrom_ && carryFrom_ != this && carryFrom_->device_ == device_ && !opt_.streaming && !isSlab();
    // Two iterations in flight on two solvers: the OTHER solver's iteration reads this solver's result maps in its carry pass,
    // so nothing of this analysis may write them before that iteration's analysis (and its own carry) is complete
    if (carry) hipStreamWaitEvent(stream_, carryFrom_->ev_[2], 0);
    if (useFused_) {
        FusedArgs f{};
        f.a = a;
        f.ctl = fusedCtl_;
        f.carrySrc = carry ? carryFrom_->res_ : nullptr;
        f.errFlag = errFlag_;
        launchAnalysisFused(f, stream_);
        if (lazyFar_) {
            farWin_ = curWindow();
            farDirValid_ = false;
        }
        nearBoxValid_ = false;
        return;
    }
    launchAnalysisFar(a, stream_);
    launchOnset(a, stream_);
    // Wet gain / decay time on the second stream, beside everything else behind the onsets: the encode pass AND the listener-
    // direction passes (which need the onsets and the occlusion map, nothing of the decay-time pass).  Worth two cross-stream
    // waits where the passes are long (windows of tens of thousands of cells) -- or always: PVA_OPT_ANALYSIS_FORK = 2
    const bool fork = opt_.analysisFork == 2 || (opt_.analysisFork != 0 && histPlane_ >= kAnalysisForkCells);
    if (fork) {
        hipEventRecord(anaEv_[0], stream_);
        hipStreamWaitEvent(stream2_, anaEv_[0], 0);
        launchRt60(a, stream2_);
        hipEventRecord(anaEv_[1], stream2_);
        launchEncode(a, stream_);
    } else {
        launchEncode(a, stream_);
        launchRt60(a, stream_);
    }
    if (carry) launchCarryResults(a, carryFrom_->res_, stream_);
    launchAnalysisDirection(a, stream_);
    if (fork) hipStreamWaitEvent(stream_, anaEv_[1], 0);
    if (lazyFar_) {
        farWin_ = curWindow();
        farDirValid_ = false;
    }
    if (a.box) nearPar_ ^= 1;  // this run's box is now "the box of the run analysed last"
    nearBoxValid_ = a.box != nullptr;
}

bool Solver::enqueueRun(int lcx, int lcy, float lx, float lz) {
    const auto inUse = queue_.lockUse();  // (against another solver's creation probing this stream: QueueClaim::use)
    if (!hipOk(hipSetDevice(device_), "hipSetDevice")) return false;
    // one run in flight at a time: the pinned staging of the per-run parameters is reused
    if (pendingTimings_ && !sync()) return false;
    if (!applyGeometry()) return false;
    const int ntiles = geo_.ntx * geo_.nty;
    const bool graph = !opt_.timeKernels && !opt_.streaming &&
                       (opt_.useGraph == 1 || (opt_.useGraph == 0 && ntiles <= 4096));
    if (!prepareDyn(lcx, lcy, true, /*banded=*/!graph)) return false;
    if (!zeroPlanesIfNeeded()) return false;
    lastLx_ = lx;
    lastLz_ = lz;
    tim_.stepLaunches = 0;
    kevUsed_ = 0;
    loopTimed_ = false;
    cur_ = 0;  // the reset clears set 0; a run never depends on the previous run's fields
    stampTimed_ = false;
    if (opt_.streaming) hipEventRecord(ev_[0], stream_);  // (the other paths: below, once it is known whether the run is a resident one)
    // (an explicit tile configuration means "use the tile kernels")
    // The whole-grid-resident kernel wins where a run is a chain of tiny launches even as a replayed graph: measured on
    // MI355X (profiles/r03_presets.txt) 0.47 vs 0.62 ms at 28^2 and 0.67 vs 0.82 ms at 38^2 -- but 0.81 vs 0.69 ms at 39^2,
    // 0.88 vs 0.71 ms at the Sandbox's 70^2 and 1.60 vs 0.97 ms at 95^2, where its two barriers per step over 5-9 cells per
    // thread are slower than the 4-wave general tiles.  auto = up to 1536 array cells; 1 = whenever the grid fits one CU.
    const bool smallWanted = opt_.smallGrid == 1 || (opt_.smallGrid == 0 && (long long)g_.NX * g_.NY <= 1536);
    const bool small = smallWanted && opt_.K == 0 && opt_.rxi == 0 && !opt_.timeKernels &&
                       opt_.useGraph != 1 && !opt_.streaming && smallGridFits(g_.NX, g_.NY) &&
                       histTilesX_ == geo_.ntx && histTilesY_ == geo_.nty;
    if (opt_.streaming) {
        // sparse-emitter mode: ring history; forward sums advanced after every `ring_` steps
        const size_t nres = (size_t)g_.gx * g_.gy;
        if (!hipOk(hipMemsetAsync(sOnset_, 0xff, nres * 4, stream_), "state reset")) return false;  // onset = -1
        for (float* p : sState_)
            if (!hipOk(hipMemsetAsync(p, 0, nres * 4, stream_), "state reset")) return false;
        if (numEmitters_ > 0 &&
            !hipOk(hipMemsetAsync(emTrace_, 0, (size_t)numEmitters_ * T_ * 4, stream_), "trace reset"))
            return false;
        if (streamFuse_ && (!hipOk(hipMemsetAsync(cellsOpen2_, 1, (size_t)2 * ntiles, stream_), "open flags") ||
                            !hipOk(hipMemsetAsync(openCount_, 0, 2 * sizeof(int), stream_), "open count")))
            return false;
        launchCap_ = numGeneral_;
        enqueueBeginRun(true);
        if (streamFuse_) {
            // what is left to the ring and the accumulate pass: the run's general list (walls, edges, the tiles whose loaded
            // region holds the listener: prepareDyn) and the tiles of the registered emitters -- the complement of
            // fusedTile() (pv_stream.h)
            std::vector<uint8_t> in((size_t)ntiles, 0);
            int n = 0;
            auto add = [&](int t) {
                if (!in[(size_t)t]) {
                    in[(size_t)t] = 1;
                    ringHost_[n++] = t;
                }
            };
            for (int i = 0; i < numGeneral_; ++i) add(listHost_[i]);
            for (int t = 0; t < ntiles; ++t)
                if (tileClassHost_[(size_t)t] != 0 || (t < (int)emTilesHost_.size() && emTilesHost_[(size_t)t])) add(t);
            numRing_ = n;
            if (n > 0 && !hipOk(hipMemcpyAsync(ringList_, ringHost_, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, stream_),
                                "ring list"))
                return false;
        }
        AnalyzeArgs aa = analyzeArgs(lx, lz);
        if (streamEv_[0] == nullptr)
            for (auto& e : streamEv_)
                if (!hipOk(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) return false;
        // Half h of the ring: [steps fill it on stream_] -> stepDone[h] -> [accumulate pass on stream2_] -> accDone[h] ->
        // [steps may overwrite it].  The passes stay in order on stream2_ (they carry per-cell state from one to the next);
        // the per-tile "history still wanted" flags they produce reach the step kernels half a ring later than in a serial
        // schedule, which only means a tile may record a few planes nobody reads.
        int pass = 0;
        fuseIdle_ = false;
        if (streamFuse_) idleHost_[0] = idleHost_[1] = 0;
        for (int tA = 0; tA < T_; tA += halfRing_, ++pass) {
            const int n = std::min(halfRing_, T_ - tA), h = pass & 1;
            if (streamFuse_ && !fuseIdle_ && pass >= 2) {
                // Feedback from the device, two passes old (the GPU keeps pass - 1 queued while the host waits here): once
                // every fused tile has closed for good -- the wave front has passed and the dry windows behind it are over,
                // a third of the way into a Mode B run -- the rest of the run needs neither the classify pass nor the
                // open-tile kernel, i.e. one launch per sweep instead of three.
                if (!hipOk(hipEventSynchronize(streamEv_[2 + h]), "pass sync")) return false;
                fuseIdle_ = idleHost_[h] != 0;
            }
            if (pass >= 2) hipStreamWaitEvent(stream_, streamEv_[2 + h], 0);
            if (!enqueueSteps(tA, n, true, true, tA == 0)) return false;
            hipEventRecord(streamEv_[h], stream_);
            hipStreamWaitEvent(stream2_, streamEv_[h], 0);
            aa.tA = tA;
            aa.tB = tA + n;
            launchStreamAccum(aa, tileEmit_, tileOpen_, ntiles, stream2_);
            if (streamFuse_ && !fuseIdle_) launchStreamIdle(classifyArgs(baseStepArgs(true, true), 0, true), idleHost_ + h, stream2_);
            hipEventRecord(streamEv_[2 + h], stream2_);
        }
        hipStreamWaitEvent(stream_, streamEv_[2], 0);
        if (pass >= 2) hipStreamWaitEvent(stream_, streamEv_[3], 0);
        hipEventRecord(ev_[1], stream_);
        launchStreamFinalize(aa, stream_);
        hipEventRecord(ev_[2], stream_);
        lastRunBatched_ = false;
        enqueueQueries();
        pendingTimings_ = true;
        return hipOk(hipGetLastError(), "run launch");
    }
    // resident kernel: one launch per run.  Its blocks wait for each other, so they must all be on the chip at once: a run
    // takes its blocks out of the device's budget until sync(); when concurrent runs of other solvers have used the budget
    // up, this run goes out as the replayed graph instead (never a wait, never a deadlock)
    bool resident = useResident_ && !(small && opt_.resident != 1);
    releaseResident();  // (a run enqueued without a sync() behind the previous one: its reservation goes back first)
    if (resident) {
        const int budget = residentBudget_;
        std::atomic<int>& inFlight = residentInFlight(device_);
        if (inFlight.fetch_add(ntiles) + ntiles > budget) {
            inFlight.fetch_sub(ntiles);
            resident = false;
        } else {
            residentHeld_ = ntiles;
        }
    }
    stampTimed_ = resident && stampsHost_ != nullptr;
    if (!stampTimed_) hipEventRecord(ev_[0], stream_);
    if (resident) {
        // one-XCD mode where the grid fits one XCD's CUs and that XCD is not taken by another solver's run
        bool xcd = xcdOk_ && ntiles <= kResidentXcdMaxTiles;
        if (xcd) {
            std::atomic<int>& onXcd = residentXcdInFlight(device_, xcdTarget_);
            if (onXcd.fetch_add(ntiles) + ntiles > kResidentXcdMaxTiles) {
                onXcd.fetch_sub(ntiles);
                xcd = false;
            } else {
                xcdHeld_ = ntiles;
            }
        }
        lastLcx_ = lcx;
        lastLcy_ = lcy;
        // (no begin-run launch: the run's parameters travel in the kernel's arguments, every block resets its own tile's entry, and
        // the flag words and the error flag were cleared by the previous run's last kernel -- or by the allocation)
        ResidentArgs ra{};
        for (int i = 0; i < 2; ++i) {
            ra.pr[i] = pr_[i];
            ra.vx[i] = vx_[i];
            ra.vy[i] = vy_[i];
        }
        ra.coef = coef_;
        ra.pulse = pulseDev_;
        ra.hist = hist_;
        ra.tileFirst = tileFirst_;
        ra.dynVal = dynCur_;
        ra.dynOut = dynDev_;
        ra.errFlag = errFlag_;
        ra.flags = resFlags_;
        ra.xcdMode = xcd ? 1 : 0;
        ra.xcdTarget = xcdTarget_;
        lastRunXcd_ = xcd;
        ra.histPlane = histPlane_;
        ra.planeBytes = (long long)geo_.rows * geo_.pitch * 4;
        ra.pitch = geo_.pitch;
        ra.G = geo_.G;
        ra.ntx = geo_.ntx;
        ra.nty = geo_.nty;
        ra.ntiles = ntiles;
        ra.T = T_;
        ra.courant = g_.courant;
        ra.stamp = stampTimed_ ? stampsHost_ : nullptr;
        if (stampTimed_) stampsHost_[0] = stampsHost_[1] = stampsHost_[2] = 0ull;  // (the previous run has been synced: enqueueRun's head)
        launchResident(K_, rxi_, ra, stream_);
        tim_.stepLaunches = ceilDiv(T_, K_);
        cur_ = tim_.stepLaunches & 1;
    } else if (small) {
        // the whole grid lives in one CU's LDS for all T steps: one launch, every cell recorded from step 0
        enqueueBeginRun(false);
        if (!hipOk(hipMemsetAsync(tileFirst_, 0, sizeof(int) * (size_t)ntiles, stream_), "tileFirst")) return false;
        SmallArgs sa{};
        sa.prOut = pr_[0];
        sa.vxOut = vx_[0];
        sa.vyOut = vy_[0];
        sa.coef = coef_;
        sa.pulse = pulseDev_;
        sa.hist = hist_;
        sa.dyn = dynDev_;
        sa.histPlane = histPlane_;
        sa.histPitch = histPitch_;
        sa.pitch = geo_.pitch;
        sa.G = geo_.G;
        sa.NX = g_.NX;
        sa.NY = g_.NY;
        sa.T = T_;
        sa.record = 1;
        sa.courant = g_.courant;
        sa.rxi = rxi_;
        sa.wi = wi_;
        launchSmallGrid(sa, stream_);
        tim_.stepLaunches = 1;
    } else if (graph) {
        // the grid of the general kernel is captured for a capacity; the live count is read from dyn on the device
        // (+ the tiles whose loaded region can hold the listener: ceil(loaded rows / tile rows) x ceil(64 / tile columns) --
        // 2 x 2 for every tile of rounds 1-2, 3 x 2 for the 12-row tile at K = 12, whose two extra listener tiles fell off a
        // capacity of "+ 4" and were advanced by nobody: tools/gpu_fuzz.py seeds 30051 / 30098 / 30172)
        const int rowsL = rxi_ + 2 * K_ + (stepConfigStacked(K_, rxi_) ? stepConfigExtraRows(K_, rxi_) : 0);
        const int cap = (int)wallTiles_.size() + ceilDiv(rowsL, rxi_) * ceilDiv(64, wi_);
        if ((!graphExec_ || graphCap_ != cap) && !buildGraph(cap)) {
            // The capture did not survive: some legacy-stream operation of another host thread while this stream was capturing
            // invalidates it (this library issues none any more -- see applyGeometry -- but a host application may).  This
            // run goes out as plain launches; the next one captures again.
            (void)hipGetLastError();
            err_.clear();
            launchCap_ = numGeneral_;
            if (!enqueueResetAndSteps()) return false;
        } else {
            if (!hipOk(hipGraphLaunch(graphExec_, stream_), "hipGraphLaunch")) return false;
            tim_.stepLaunches = ceilDiv(T_, K_);
            cur_ = tim_.stepLaunches & 1;
        }
    } else {
        launchCap_ = numGeneral_;
        if (!enqueueResetAndSteps()) return false;
    }
    if (!stampTimed_) hipEventRecord(ev_[1], stream_);
    if (!opt_.skipAnalysis) enqueueAnalysis(lx, lz);
    if (!stampTimed_) hipEventRecord(ev_[2], stream_);
    lastRunBatched_ = false;
    // last kernel of the run: the registered queries' outputs and the status words, both into pinned memory
    launchRunFinish(res_, (long long)g_.gx * g_.gy, qCellsHost_, opt_.skipAnalysis ? 0 : numQueries_, qOutHost_, farInfo(), errFlag_,
                    activeCount_, lastRunXcd_ ? resFlags_ + geo_.ntx * geo_.nty + 1 : nullptr, statusHost_, resFlags_,
                    resFlags_ ? geo_.ntx * geo_.nty + 2 : 0, stampTimed_ ? stampsHost_ : nullptr, stream_);
    // (ev_[2] is also what the OTHER solver of a pipelined pair waits for before its carry pass reads this solver's maps: behind
    // the last kernel here, where it delays nothing of this run)
    if (stampTimed_) hipEventRecord(ev_[2], stream_);
    statusQueued_ = true;
    pendingTimings_ = true;
    return hipOk(hipGetLastError(), "run launch");
}

// reset pr / vx / vy (FDTD.cpp:109-119) + the T-step loop; this is what a captured graph contains
bool Solver::enqueueResetAndSteps() {
    enqueueBeginRun(true);
    // graph capture cannot hold timing events; plain launches get one more event so that the launch loop is timed alone
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    hipStreamIsCapturing(stream_, &cs);
    if (cs == hipStreamCaptureStatusNone) {
        hipEventRecord(ev_[3], stream_);
        loopTimed_ = true;
    }
    return enqueueSteps(0, T_, true, true, true);
}

void Solver::dropGraph() {
    if (graphExec_) hipGraphExecDestroy(graphExec_);
    if (graph_) hipGraphDestroy(graph_);
    graphExec_ = nullptr;
    graph_ = nullptr;
    graphCap_ = -1;
}

// Stream-capture the whole run schedule (memset nodes + every air / general launch with their cross-stream event
// edges) once per geometry; later runs replay it with one hipGraphLaunch.  Launch-bound small grids gain the most
// (the sandbox's 71^2 grid: 110 kernel launches + 220 event operations per run otherwise).
bool Solver::buildGraph(int cap) {
    dropGraph();
    launchCap_ = cap;
    const int savedCur = cur_;
    const int savedLaunches = tim_.stepLaunches;
    if (!hipOk(hipStreamBeginCapture(stream_, hipStreamCaptureModeRelaxed), "hipStreamBeginCapture")) return false;
    bool ok = enqueueResetAndSteps();
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(stream_, &g);
    // test hook (tests/test_gpu_parity.py::test_run_survives_a_lost_graph_capture): the solver's first capture counts as lost
    if (!captureLossInjected_ && opt_.debugLoseFirstCapture) {
        captureLossInjected_ = true;
        ok = false;
    }
    cur_ = savedCur;
    tim_.stepLaunches = savedLaunches;
    if (!ok || e != hipSuccess || !g) {  // (the caller falls back to plain launches)
        if (g) hipGraphDestroy(g);
        return false;
    }
    graph_ = g;
    if (hipGraphInstantiate(&graphExec_, graph_, nullptr, nullptr, 0) != hipSuccess) {
        graphExec_ = nullptr;
        dropGraph();
        return false;
    }
    graphCap_ = cap;
    return true;
}

// B independent runs of identically configured solvers, stepped in lock-step by ONE launch per K steps
// (pv_step_batch_kernel, blockIdx.y = run) on the first solver's stream.  Each solver prepares its run on its own
// stream (begin-run kernel) and analyses its own history there afterwards, so the analyses of the batch overlap.
// Timings: every solver's fdtdMs spans the whole batch's step loop (the B runs share it).
bool Solver::runBatch(Solver* const* s, int n, const float* lxyz, bool wait, std::string* err) {
    auto bad = [&](const char* what) {
        if (err) *err = what;
        return false;
    };
    if (n < 1 || n > kBatchMax) return bad("batch size must be 1..8");
    std::unique_lock<std::recursive_mutex> inUse[kBatchMax];  // (every member's stream, in the batch's order)
    for (int i = 0; i < n; ++i) inUse[i] = s[i]->queue_.lockUse();
    Solver& lead = *s[0];
    for (int i = 0; i < n; ++i) {
        Solver& v = *s[i];
        for (int j = 0; j < i; ++j)
            if (s[j] == s[i]) return bad("a solver appears twice in the batch");
        if (v.device_ != lead.device_ || v.K_ != lead.K_ || v.rxi_ != lead.rxi_ || v.T_ != lead.T_ ||
            v.geo_.rows != lead.geo_.rows || v.geo_.pitch != lead.geo_.pitch || v.geo_.ntx != lead.geo_.ntx ||
            v.geo_.nty != lead.geo_.nty || v.g_.gx != lead.g_.gx || v.g_.gy != lead.g_.gy ||
            v.opt_.tileOrder != lead.opt_.tileOrder)
            return bad("batched solvers must share device, grid and tile configuration");
        if (v.opt_.streaming || v.opt_.timeKernels > 0 || v.opt_.merged != 1 ||
            !v.opt_.packed || !batchConfigOk(v.K_, v.rxi_))
            return bad("batched runs need the default merged packed-math kernel of a batch configuration, without "
                       "streaming analysis or kernel timing");
    }
    if (hipSetDevice(lead.device_) != hipSuccess) return bad("hipSetDevice failed");
    int gcap = 0;
    for (int i = 0; i < n; ++i) {
        Solver& v = *s[i];
        const float lx = lxyz[3 * i], lz = lxyz[3 * i + 2];
        int lcx, lcy;
        listenerCell(v.g_, lx, lz, &lcx, &lcy);
        if ((v.pendingTimings_ && !v.sync()) || !v.applyGeometry() || !v.prepareDyn(lcx, lcy, true, false) ||
            !v.zeroPlanesIfNeeded()) {
            if (err) *err = v.err_;
            return false;
        }
        v.lastLx_ = lx;
        v.lastLz_ = lz;
        v.tim_.stepLaunches = 0;
        v.kevUsed_ = 0;
        v.loopTimed_ = false;
        v.cur_ = 0;
        v.launchCap_ = v.numGeneral_;
        gcap = std::max(gcap, v.numGeneral_);
        v.stampTimed_ = false;
        hipEventRecord(v.ev_[0], v.stream_);
        v.enqueueBeginRun(true);
        if (i > 0) {  // the shared step loop starts when every run's parameters are on the device
            hipEventRecord(v.forkEv_, v.stream_);
            hipStreamWaitEvent(lead.stream_, v.forkEv_, 0);
        }
    }
    BatchArgs ba{};
    ba.n = n;
    ba.gblocks = (gcap + 7) & ~7;
    for (int i = 0; i < n; ++i) ba.a[i] = s[i]->baseStepArgs(true, true);
    const int T = lead.T_, K = lead.K_;
    int li = 0;
    for (int done = 0; done < T; done += K, ++li) {
        const int k = std::min(K, T - done);
        for (int i = 0; i < n; ++i) {
            s[i]->setLaunchArgs(ba.a[i], done, k, done == 0, li);
            s[i]->cur_ ^= 1;
            ++s[i]->tim_.stepLaunches;
        }
        launchBatch(K, lead.rxi_, ba, lead.stream_);
    }
    hipEventRecord(lead.forkEv_, lead.stream_);
    for (int i = 0; i < n; ++i) {
        Solver& v = *s[i];
        if (i > 0) hipStreamWaitEvent(v.stream_, lead.forkEv_, 0);
        hipEventRecord(v.ev_[1], v.stream_);
        if (!v.opt_.skipAnalysis) v.enqueueAnalysis(v.lastLx_, v.lastLz_);
        hipEventRecord(v.ev_[2], v.stream_);
        v.lastRunBatched_ = n > 1;
        v.enqueueQueries();
        v.enqueueRunStatus();
        v.pendingTimings_ = true;
    }
    if (hipGetLastError() != hipSuccess) return bad("batched run launch failed");
    if (wait)
        for (int i = 0; i < n; ++i)
            if (!s[i]->sync()) {
                if (err) *err = s[i]->err_;
                return false;
            }
    return true;
}

bool Solver::runCells(int lcx, int lcy, float lx, float lz, bool wait) {
    if (!enqueueRun(lcx, lcy, lx, lz)) return false;
    return wait ? sync() : true;
}

bool Solver::run(float lx, float ly, float lz, bool wait, Solver* carryFrom) {
    (void)ly;  // world y is ignored: grid-x = world x, grid-y = world z (FDTD.cpp:97-98)
    struct Scope {
        Solver*& p;
        ~Scope() { p = nullptr; }
    } scope{carryFrom_};
    carryFrom_ = carryFrom;
    if (opt_.edgeTiles) {  // tile class 2 exists only in the batched kernel: a batch of one
        Solver* self = this;
        const float xyz[3] = {lx, ly, lz};
        std::string e;
        if (runBatch(&self, 1, xyz, wait, &e)) return true;
        return err_.empty() ? fail(e) : false;
    }
    int lcx, lcy;
    listenerCell(g_, lx, lz, &lcx, &lcy);
    return runCells(lcx, lcy, lx, lz, wait);
}

bool Solver::sync() {
    if (!hipOk(hipSetDevice(device_), "hipSetDevice")) return false;
    if (useFused_ && std::getenv("PLANEVERB_AMD_FUSED_DEBUG")) {  // development aid: a run that does not end within 3 s
        for (int i = 0; i < 3000 && hipStreamQuery(stream_) == hipErrorNotReady; ++i) usleep(1000);
        if (hipStreamQuery(stream_) == hipErrorNotReady) {
            unsigned w[kFusedCtlWords] = {};
            hipStream_t s2;
            hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
            hipMemcpyAsync(w, fusedCtl_, sizeof(w), hipMemcpyDeviceToHost, s2);
            hipStreamSynchronize(s2);
            std::fprintf(stderr, "[planeverb_amd] fused analysis stuck; control words:");
            for (unsigned v : w) std::fprintf(stderr, " %u", v);
            std::fprintf(stderr, "\n");
            _exit(3);
        }
    }
    if (!hipOk(hipStreamSynchronize(stream_), "stream sync")) return false;
    if (useFused_) {
        static const char* dbgEnv = std::getenv("PLANEVERB_AMD_FUSED_DEBUG");
        if (dbgEnv && std::atoi(dbgEnv) == 2 && pendingTimings_) {
            unsigned long long st[32];
            hipMemcpy(st, fusedCtl_ + kFusedCtlWords, sizeof(st), hipMemcpyDeviceToHost);
            std::fprintf(stderr, "[fused] us since the first worker: ");
            for (int p = 0; p < 10; ++p) {
                if (st[1 + 2 * p] == ~0ull) break;
                std::fprintf(stderr, " p%d %.1f-%.1f", p, (double)(st[1 + 2 * p] - st[0]) * 0.01, (double)(st[2 + 2 * p] - st[0]) * 0.01);
            }
            std::fprintf(stderr, " | last onset %.1f encode %.1f decay %.1f\n", (double)(st[22] - st[0]) * 0.01, (double)(st[24] - st[0]) * 0.01,
                         (double)(st[26] - st[0]) * 0.01);
            hipMemset(fusedCtl_ + kFusedCtlWords + 2 * 22, 0, 8 * 6);
            hipMemset(fusedCtl_ + kFusedCtlWords, 0xff, 8 * 21);
            for (int p = 0; p < 10; ++p) hipMemset(fusedCtl_ + kFusedCtlWords + 2 * (2 + 2 * p), 0, 8);
        }
    }
    releaseResident();
    if (pendingTimings_) {
        pendingTimings_ = false;
        if (stampTimed_) {  // 100 MHz ticks: the run's first kernel, the analysis' first kernel (0: none ran), the run's last kernel
            const unsigned long long s0 = stampsHost_[0], s2 = stampsHost_[2], s1 = stampsHost_[1] ? stampsHost_[1] : s2;
            tim_.fdtdMs = (s0 && s1 > s0) ? (float)((double)(s1 - s0) * 1e-5) : 0.f;
            tim_.analysisMs = (s2 > s1) ? (float)((double)(s2 - s1) * 1e-5) : 0.f;
        } else {
            hipEventElapsedTime(&tim_.fdtdMs, ev_[0], ev_[1]);
            hipEventElapsedTime(&tim_.analysisMs, ev_[1], ev_[2]);
        }
        tim_.stepLoopMs = 0.f;
        if (loopTimed_) hipEventElapsedTime(&tim_.stepLoopMs, ev_[3], ev_[1]);
        if (opt_.timeKernels && kevUsed_ > 0) {
            double air = 0, gen = 0;
            const int n = kevUsed_ / 4;
            for (int i = 0; i < n; ++i) {
                float a = 0, g = 0;
                hipEventElapsedTime(&a, kev_[(size_t)4 * i], kev_[(size_t)4 * i + 1]);
                hipEventElapsedTime(&g, kev_[(size_t)4 * i + 2], kev_[(size_t)4 * i + 3]);
                air += a;
                gen += g;
            }
            tim_.airKernelMs = (float)(air / n);
            tim_.generalKernelMs = (float)(gen / n);
            tim_.airLaunches = n;
            tim_.generalLaunches = numGeneral_ > 0 ? n : 0;
        }
        int flag = 0, counts[2] = {0, 0};
        if (statusQueued_) {
            // the run's last kernel left its status words in pinned memory: no copy, no second synchronisation
            statusQueued_ = false;
            flag = statusHost_[0];
            counts[0] = statusHost_[1];
            counts[1] = statusHost_[2];
            tim_.silentCells = statusHost_[4];
            if (lastRunXcd_ && statusHost_[3] >= 0 && statusHost_[3] < geo_.ntx * geo_.nty && flag == 0) flag = 4;
            lastRunXcd_ = false;
        } else {
            // (raw stepping: no status kernel.)  The error flag, never through the legacy stream (applyGeometry), and by the kind
            // of run -- read behind stream_, the batched launches of a batch member's NEXT step loop took twice as long
            // (profiles/r03_ab_errflag.txt) -- through the solver's second stream, which the batched mode does not use.
            lastRunXcd_ = false;
            hipStream_t fs = lastRunBatched_ ? stream2_ : stream_;  // (stream2_ is idle in the batched mode)
            if (!hipOk(hipMemcpyAsync(&flag, errFlag_, sizeof(int), hipMemcpyDeviceToHost, fs), "errFlag copy") ||
                !hipOk(hipMemcpyAsync(counts, activeCount_, sizeof(counts), hipMemcpyDeviceToHost, fs), "count copy") ||
                !hipOk(hipStreamSynchronize(fs), "errFlag sync"))
                return false;
            if (flag) hipMemsetAsync(errFlag_, 0, sizeof(int), fs);  // (reported: the next run starts clean)
        }
        tim_.reachedCells = counts[1];
        lastReached_ = counts[1];
        tim_.activeCells = counts[0];
        if (flag == 4 && xcdOk_) {
            // one-XCD mode: fewer workgroups than tiles turned up on this solver's XCD (another dispatch pattern / partition
            // mode than the one observed).  Nothing was computed; from now on the placement-independent hand-off, and the run
            // is repeated in it.
            xcdOk_ = false;
            std::fprintf(stderr, "[planeverb_amd] resident kernel: one-XCD mode not available on this device (workgroups are "
                                 "not spread over the XCDs as expected); using the placement-independent hand-off\n");
            return enqueueRun(lastLcx_, lastLcy_, lastLx_, lastLz_) && sync();
        }
        if (flag == 3 && useResident_) {
            // A workgroup waited ~2 s for a neighbour: the blocks were not all on the chip (another process, or a long kernel
            // of the host application, holds CUs -- the per-process budget cannot see those).  The replayed-graph path needs no
            // co-residency and gives the same bits: this solver uses it from now on, and the run is repeated in it.
            useResident_ = false;
            std::fprintf(stderr, "[planeverb_amd] resident kernel: a workgroup gave up waiting for its neighbours (the device is "
                                 "shared?); this solver runs its steps as a replayed graph from now on\n");
            return enqueueRun(lastLcx_, lastLcy_, lastLx_, lastLz_) && sync();
        }
        if (flag == 3 || flag == 4) return fail("resident kernel: a workgroup gave up waiting for its neighbours (run aborted)");
        if (flag == 5) return fail("slab decomposition: a neighbour's halo rows never arrived (run aborted)");
        if (flag == 6) {
            // (on the solver's own stream, never the legacy stream: see applyGeometry)
            unsigned w[kFusedCtlWords] = {};
            if (fusedCtl_) {
                hipMemcpyAsync(w, fusedCtl_, sizeof(w), hipMemcpyDeviceToHost, stream_);
                hipMemsetAsync(fusedCtl_, 0, sizeof(w), stream_);
            }
            hipMemsetAsync(errFlag_, 0, sizeof(int), stream_);
            hipStreamSynchronize(stream_);
            std::string m = "fused analysis: a worker waited for a phase in vain (run aborted); ticket and phase counters:";
            for (unsigned v : w) m += " " + std::to_string(v);
            return fail(m);
        }
        if (flag) return fail("pressure history window overflow (a tile outside the window became non-zero)");
    }
    return true;
}

bool Solver::runSteps(int nsteps, bool withPulse, float lx, float lz) {
    if (opt_.edgeTiles) return fail("stencil-only stepping is not available with edge tiles (batched kernel only)");
    const auto inUse = queue_.lockUse();
    if (!hipOk(hipSetDevice(device_), "hipSetDevice")) return false;
    if (!applyGeometry()) return false;
    int lcx, lcy;
    listenerCell(g_, lx, lz, &lcx, &lcy);
    if (!prepareDyn(lcx, lcy, withPulse, true)) return false;
    tim_.stepLaunches = 0;
    kevUsed_ = 0;
    loopTimed_ = false;
    launchCap_ = numGeneral_;
    planesDirty_ = true;
    enqueueBeginRun(false);
    stampTimed_ = false;
    hipEventRecord(ev_[0], stream_);
    if (!enqueueSteps(0, nsteps, withPulse, false)) return false;
    hipEventRecord(ev_[1], stream_);
    hipEventRecord(ev_[2], stream_);
    pendingTimings_ = true;
    return sync();
}

// ----------------------------------------------------------------------------------------------------------------
// read-back
// ----------------------------------------------------------------------------------------------------------------

bool Solver::getOutput(float ex, float ey, float ez, float out8[8], bool* valid) {
    (void)ey;
    int cx, cy;
    *valid = resultCell(g_, ex, ez, &cx, &cy);
    if (!*valid) return true;
    if (!hipOk(hipSetDevice(device_), "hipSetDevice")) return false;
    const size_t idx = (size_t)cx * g_.gy + cy;
    // 8 planes -> 8 floats: a one-wave kernel writes them straight into pinned host memory (a strided
    // hipMemcpy2DAsync of 8 x 4 bytes costs 0.4 ms on this runtime)
    launchGatherOutput(res_, (long long)g_.gx * g_.gy, (long long)idx, outHost_, farInfo(), stream_);
    if (!hipOk(hipStreamSynchronize(stream_), "output sync")) return false;
    for (int k = 0; k < 8; ++k) out8[k] = outHost_[k];
    return true;
}

bool Solver::setOutputQueries(const float* xyz, int n) {
    if (n < 0 || n > kMaxQueries) return fail("at most 64 output queries");
    if (pendingTimings_ && !sync()) return false;  // the gather of a run in flight still reads the cell table
    for (int i = 0; i < n; ++i) {
        int cx, cy;
        qCellsHost_[i] = resultCell(g_, xyz[3 * i], xyz[3 * i + 2], &cx, &cy) ? (long long)cx * g_.gy + cy : -1;
    }
    numQueries_ = n;
    return true;
}

// last kernel of a run: its status words into pinned memory (sync() then needs no copy and no second synchronisation)
void Solver::enqueueRunStatus() {
    launchRunStatus(errFlag_, activeCount_, lastRunXcd_ ? resFlags_ + geo_.ntx * geo_.nty + 1 : nullptr, statusHost_, stream_);
    statusQueued_ = true;
}

void Solver::enqueueQueries() {
    if (numQueries_ > 0 && !opt_.skipAnalysis)
        launchGatherQueries(res_, (long long)g_.gx * g_.gy, qCellsHost_, numQueries_, qOutHost_, farInfo(), stream_);
}

// after sync(): the registered queries' outputs of the last run, straight from pinned memory
bool Solver::queriedOutputs(float* out8n, unsigned char* valid, int n) {
    if (n != numQueries_) return fail("queriedOutputs: count differs from the registered queries");
    if (pendingTimings_ && !sync()) return false;
    for (int i = 0; i < n; ++i) {
        valid[i] = qCellsHost_[i] >= 0;
        for (int k = 0; k < 8; ++k) out8n[8 * i + k] = qOutHost_[8 * i + k];
    }
    return true;
}

// the reference's result map is an array of 8-float structs; ours is 8 planes.  Whole-map read-backs (tests, the live
// module's host copy) get the AoS form from a pack kernel into a buffer that exists only once somebody asked.
bool Solver::packResults() {
    if (!ensureFarDirections()) return false;
    const size_t n = (size_t)g_.gx * g_.gy;
    if (!res8_) {
        if (!dalloc(&res8_, n * 8, false)) return false;
    }
    launchPackResults(res_, (long long)n, res8_, stream_);
    return hipOk(hipGetLastError(), "pack results");
}

bool Solver::copyResults(float* res8, float* delay) {
    if (!hipOk(hipSetDevice(device_), "hipSetDevice")) return false;
    const size_t n = (size_t)g_.gx * g_.gy;
    if (res8 && !packResults()) return false;
    if (res8 && !hipOk(hipMemcpyAsync(res8, res8_, n * 32, hipMemcpyDeviceToHost, stream_), "results copy"))
        return false;
    if (delay && !hipOk(hipMemcpyAsync(delay, delay_, n * 4, hipMemcpyDeviceToHost, stream_), "delay copy"))
        return false;
    return hipOk(hipStreamSynchronize(stream_), "results sync");
}

bool Solver::copyResultsBlock(int r0, int c0, int nr, int nc, float* res8, float* delay) {
    if (r0 < 0 || c0 < 0 || nr < 1 || nc < 1 || r0 + nr > g_.gx || c0 + nc > g_.gy) return fail("result block outside the map");
    if (isSlab()) return fail("copyResultsBlock is not available on a slab");
    if (!hipOk(hipSetDevice(device_), "hipSetDevice")) return false;
    const size_t cells = (size_t)nr * nc;
    // (far cells -- outside the last run's window, or outside its near box -- get their direction in closed form inside the pack)
    if (res8) {
        float* tmp = nullptr;
        if (!hipOk(hipMalloc((void**)&tmp, cells * 32), "hipMalloc result block")) return false;
        launchPackWindow(res_, (long long)g_.gx * g_.gy, g_.gy, r0, c0, nr, nc, tmp, farInfo(), stream_);
        const bool ok = hipOk(hipMemcpyAsync(res8, tmp, cells * 32, hipMemcpyDeviceToHost, stream_), "block copy") &&
                        hipOk(hipStreamSynchronize(stream_), "block sync");
        hipFree(tmp);
        if (!ok) return false;
    }
    if (delay && !hipOk(hipMemcpy2DAsync(delay, (size_t)nc * 4, delay_ + (size_t)r0 * g_.gy + c0, (size_t)g_.gy * 4,
                                          (size_t)nc * 4, (size_t)nr, hipMemcpyDeviceToHost, stream_), "delay block copy"))
        return false;
    return hipOk(hipStreamSynchronize(stream_), "block sync");
}

bool Solver::copyResultsAsync(float* res8Host) {
    const size_t n = (size_t)g_.gx * g_.gy;
    if (!packResults()) return false;
    return hipOk(hipMemcpyAsync(res8Host, res8_, n * 32, hipMemcpyDeviceToHost, stream_), "results copy");
}

size_t Solver::windowCapacity() const {
    return (size_t)std::min(histTilesX_ * rxi_, g_.gx) * (size_t)std::min(histTilesY_ * wi_, g_.gy);
}

bool Solver::publishWindowAsync(float* hostDst, WindowBlock* info, bool overlap) {
    if (!dynValid_) return fail("no simulation has run yet");
    WindowBlock w;
    const Block b = curWindow();
    w.r0 = b.r0;
    w.c0 = b.c0;
    w.nr = b.nr;
    w.nc = b.nc;
    w.lx = lastLx_;
    w.lz = lastLz_;
    if (!win8_ && !dalloc(&win8_, windowCapacity() * 8, false)) return false;
    if (overlap && !pubEv_[0]) {
        for (auto& e : pubEv_)
            if (!hipOk(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) return false;
    }
    if (pubCopyPending_) hipStreamWaitEvent(stream_, pubEv_[1], 0);  // the staging block is free again
    launchPackWindow(res_, (long long)g_.gx * g_.gy, g_.gy, w.r0, w.c0, w.nr, w.nc, win8_, farInfo(), stream_);
    if (!hipOk(hipGetLastError(), "pack window")) return false;
    *info = w;
    pubCopyPending_ = false;
    if (w.nr == 0 || w.nc == 0) return true;
    if (!overlap)
        return hipOk(hipMemcpyAsync(hostDst, win8_, (size_t)w.nr * w.nc * 32, hipMemcpyDeviceToHost, stream_), "window copy");
    hipEventRecord(pubEv_[0], stream_);
    hipStreamWaitEvent(stream2_, pubEv_[0], 0);
    if (!hipOk(hipMemcpyAsync(hostDst, win8_, (size_t)w.nr * w.nc * 32, hipMemcpyDeviceToHost, stream2_), "window copy")) return false;
    hipEventRecord(pubEv_[1], stream2_);
    pubCopyPending_ = true;
    return true;
}

bool Solver::waitPublish() {
    if (!pubCopyPending_) return true;
    return hipOk(hipEventSynchronize(pubEv_[1]), "window copy sync");
}

void* Solver::hostAlloc(size_t bytes) {
    void* p = nullptr;
    return hipHostMalloc(&p, bytes) == hipSuccess ? p : nullptr;
}

void Solver::hostFree(void* p) {
    if (p) hipHostFree(p);
}

bool Solver::setEmitters(const float* xyz, int n) {
    if (!opt_.streaming) return fail("emitters are registered only in streaming-analysis mode");
    if (!hipOk(hipSetDevice(device_), "hipSetDevice")) return false;
    std::vector<int> cells;
    for (int i = 0; i < n; ++i) {
        int cx, cy;
        if (resultCell(g_, xyz[3 * i], xyz[3 * i + 2], &cx, &cy)) cells.push_back(cx * g_.gy + cy);
    }
    if (!hipOk(hipStreamSynchronize(stream_), "sync")) return false;
    if ((int)cells.size() > emCap_) {
        if (emCells_) hipFree(emCells_);
        if (emTrace_) hipFree(emTrace_);
        emCells_ = nullptr;
        emTrace_ = nullptr;
        emCap_ = (int)cells.size();
        if (!dalloc(&emCells_, (size_t)emCap_, true) || !dalloc(&emTrace_, (size_t)emCap_ * T_, true)) return false;
    }
    numEmitters_ = (int)cells.size();
    // The uploads go through stream_, behind the zero fill dalloc() queued there (a synchronous hipMemcpy is not ordered with
    // a non-blocking stream: the fill could land AFTER it and leave every emitter at cell 0 -- seen as wet gain / RT60 = 0 in
    // about one process out of four).
    std::vector<uint8_t> te((size_t)geo_.ntx * geo_.nty, 0);
    for (int c : cells) te[(size_t)((c / g_.gy) / rxi_) * geo_.nty + (c % g_.gy) / wi_] = 1;
    emTilesHost_ = te;
    if (!hipOk(hipMemcpyAsync(tileEmit_, te.data(), te.size(), hipMemcpyHostToDevice, stream_), "emitter tiles")) return false;
    if (numEmitters_ > 0 &&
        !hipOk(hipMemcpyAsync(emCells_, cells.data(), cells.size() * 4, hipMemcpyHostToDevice, stream_), "emitters"))
        return false;
    return hipOk(hipStreamSynchronize(stream_), "emitter upload");
}

bool Solver::impulseResponse(int cx, int cy, float* out3T) {
    if (opt_.streaming) return fail("the full pressure history is not kept in streaming-analysis mode");
    if (cx < 0 || cx >= lNX_ || cy < 0 || cy > g_.gy) return fail("cell outside the grid");
    if (!dynValid_) return fail("no simulation has run yet");
    if (!hipOk(hipSetDevice(device_), "hipSetDevice")) return false;
    launchIr(analyzeArgs(lastLx_, lastLz_), cx, cy, scratch_, stream_);
    if (!hipOk(hipMemcpyAsync(out3T, scratch_, (size_t)3 * T_ * 4, hipMemcpyDeviceToHost, stream_), "ir copy"))
        return false;
    return hipOk(hipStreamSynchronize(stream_), "ir sync");
}

bool Solver::impulseResponseCells(int cx, int cy, void* out16T) {
    std::vector<float> f((size_t)3 * T_);
    if (!impulseResponse(cx, cy, f.data())) return false;
    struct RefCell {
        float pr, vx, vy;
        short b, by;
    };
    static_assert(sizeof(RefCell) == 16, "PvTypes.h:106-121");
    const size_t i = (size_t)(cx + x0_) * g_.NY + cy;
    const short b = (short)betaHost_[i], by = (short)byHost_[i];
    RefCell* out = static_cast<RefCell*>(out16T);
    for (int t = 0; t < T_; ++t) out[t] = RefCell{f[(size_t)3 * t], f[(size_t)3 * t + 1], f[(size_t)3 * t + 2], b, by};
    return true;
}

bool Solver::copyFields(float* pr, float* vx, float* vy) {
    if (!hipOk(hipSetDevice(device_), "hipSetDevice")) return false;
    const size_t n = (size_t)lNX_ * g_.NY;  // (a slab: its own rows)
    const float* src[3] = {pr_[cur_], vx_[cur_], vy_[cur_]};
    float* dst[3] = {pr, vx, vy};
    for (int i = 0; i < 3; ++i) {
        if (!dst[i]) continue;
        launchUnpad(src[i], scratch_, geo_, stream_);
        if (!hipOk(hipMemcpyAsync(dst[i], scratch_, n * 4, hipMemcpyDeviceToHost, stream_), "field copy")) return false;
        if (!hipOk(hipStreamSynchronize(stream_), "field sync")) return false;
    }
    return true;
}

bool Solver::setFields(const float* pr, const float* vx, const float* vy) {
    if (isSlab()) return fail("setFields is not available on a slab");
    planesDirty_ = true;
    if (!hipOk(hipSetDevice(device_), "hipSetDevice")) return false;
    const size_t n = (size_t)g_.NX * g_.NY;
    const float* src[3] = {pr, vx, vy};
    float* dst[3] = {pr_[cur_], vx_[cur_], vy_[cur_]};
    const size_t planeBytes = (size_t)geo_.rows * geo_.pitch * 4;
    for (int i = 0; i < 3; ++i) {
        if (!hipOk(hipMemsetAsync(dst[i], 0, planeBytes, stream_), "field clear")) return false;
        if (!src[i]) continue;
        if (!hipOk(hipMemcpyAsync(scratch_, src[i], n * 4, hipMemcpyHostToDevice, stream_), "field upload")) return false;
        launchPad(scratch_, dst[i], geo_, stream_);
        if (!hipOk(hipStreamSynchronize(stream_), "field sync")) return false;
    }
    return true;
}

bool Solver::copyHistoryPlane(int t, float* pr) {
    if (opt_.streaming) return fail("the full pressure history is not kept in streaming-analysis mode");
    if (t < 0 || t >= T_) return fail("step outside the recorded range");
    if (!dynValid_) return fail("no simulation has run yet");
    if (!hipOk(hipSetDevice(device_), "hipSetDevice")) return false;
    launchHistPlane(analyzeArgs(lastLx_, lastLz_), t, scratch_, lNX_, g_.NY, histRows_, stream_);
    if (!hipOk(hipMemcpyAsync(pr, scratch_, (size_t)lNX_ * g_.NY * 4, hipMemcpyDeviceToHost, stream_), "plane copy"))
        return false;
    return hipOk(hipStreamSynchronize(stream_), "plane sync");
}

bool Solver::copyPulse(float* out) {
    std::memcpy(out, pulse_.data(), (size_t)g_.T * 4);
    return true;
}

bool Solver::copyMaterial(uint8_t* beta, float* R) {
    const size_t n = (size_t)g_.NX * g_.NY;
    if (beta) std::memcpy(beta, mat_.beta().data(), n);
    if (R) std::memcpy(R, mat_.R().data(), n * 4);
    return true;
}

}  // namespace pva
