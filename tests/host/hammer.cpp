// tests/host/hammer.cpp -- TEST INFRASTRUCTURE: hammers the live module's C-ABI (include/planeverb_amd.h, Part 1) from
// several threads in a HIP-less build (pv_core.cpp + pv_context.cpp + pv_capi.cpp against tests/host/fake_solver.h)
// under -fsanitize=thread and -fsanitize=address,undefined.  What the reference leaves racy (PvContext.h:34-42: plain
// fields shared by the game thread, the audio thread and the worker; Exit deleting what GetOutput reads) must be
// clean here, and every record GetOutput returns must come from ONE iteration and belong to the cell asked for.
//
//   hammer [seconds]      exit code 0 = all invariants held (sanitizer reports make the process fail on their own)
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "planeverb_amd.h"
#include "fake_solver.h"

static std::atomic<bool> g_stop{false};
// even = the module's state is stable, odd = main is inside Init / Exit.  Emitter ids restart with every context, so a
// getter's id (and therefore its cell check) is only meaningful within one stable generation.
static std::atomic<unsigned> g_gen{0};
static std::atomic<long long> g_bad{0}, g_reads{0}, g_stale{0}, g_window{0}, g_sentinel{0};
static const float kSize = 25.0f;
static const int kRes = 275;
static float g_dx = 0.f;

static void fail(const char* what, const PlaneverbOutput& o, int cx, int cy) {
    if (g_bad.fetch_add(1) < 10)
        std::fprintf(stderr, "BAD %s: cell (%d,%d) got {%g %g %g %g %g %g %g %g}\n", what, cx, cy, o.occlusion, o.wetGain,
                     o.rt60, o.lowpass, o.directionX, o.directionY, o.sourceDirectionX, o.sourceDirectionY);
}

// one record must be: the sentinel; or all-zero history (+ a unit / zero direction); or a record of ONE iteration
// (tag in occlusion == tag in sourceDirectionX) that belongs to this cell
static void check(const PlaneverbOutput& o, int cx, int cy) {
    g_reads.fetch_add(1, std::memory_order_relaxed);
    if (o.occlusion == -1.f) {
        g_sentinel.fetch_add(1, std::memory_order_relaxed);
        if (o.wetGain != 0.f || o.rt60 != 0.f || o.lowpass != 0.f) fail("sentinel with payload", o, cx, cy);
        return;
    }
    const bool inWindow = o.directionX == 0.25f && o.directionY == -0.5f;
    if (!inWindow) {
        const float n = o.directionX * o.directionX + o.directionY * o.directionY;
        if (!(n == 0.f || std::fabs(n - 1.f) < 1e-4f)) fail("direction of an out-of-window cell is not a unit vector", o, cx, cy);
        g_stale.fetch_add(1, std::memory_order_relaxed);
    } else {
        g_window.fetch_add(1, std::memory_order_relaxed);
    }
    const bool zero = o.occlusion == 0.f && o.wetGain == 0.f && o.rt60 == 0.f && o.lowpass == 0.f &&
                      o.sourceDirectionX == 0.f && o.sourceDirectionY == 0.f;
    if (zero) {
        if (inWindow) fail("zero record with an in-window direction", o, cx, cy);
        return;
    }
    if (o.occlusion != o.sourceDirectionX) fail("torn record (two iterations)", o, cx, cy);
    if (o.wetGain != (float)cx || o.rt60 != (float)cy || o.sourceDirectionY != (float)(cx + cy))
        fail("record of another cell", o, cx, cy);
}

static void getterThread(unsigned seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> pos(0.2f, kSize - 0.5f);
    int id = -1;
    unsigned idGen = 1;
    while (!g_stop.load(std::memory_order_relaxed)) {
        const unsigned g1 = g_gen.load();
        if (id < 0 || idGen != g1) {
            id = PlaneverbEmit(1.f, 0.f, 1.f);
            idGen = g1;
        }
        const float x = pos(rng), z = pos(rng);
        const int cx = (int)(unsigned)(x / g_dx), cy = (int)(unsigned)(z / g_dx);
        PlaneverbUpdateEmission(id, x, 0.f, z);
        PlaneverbOutput o[8];
        for (int k = 0; k < 8; ++k) o[k] = PlaneverbGetOutput(id);  // (also while Init / Exit run: must not crash)
        if ((g1 & 1) == 0 && g_gen.load() == g1 && idGen == g1)     // same stable context throughout: id was ours
            for (int k = 0; k < 8; ++k) check(o[k], cx, cy);
        if ((rng() & 1023) == 0 && g_gen.load() == idGen) {
            PlaneverbEndEmission(id);
            id = -1;
        }
    }
}

static void churnThread(unsigned seed) {  // emitter and geometry tables
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> pos(0.f, kSize);
    std::vector<int> em, geo;
    unsigned gen = 1;
    while (!g_stop.load(std::memory_order_relaxed)) {
        const unsigned g1 = g_gen.load();
        if (g1 != gen) {  // ids of an earlier context: ending them now would free ids that belong to the getters
            em.clear();   // (EndEmission does not validate, like EmissionManager.cpp:63-67)
            geo.clear();
            gen = g1;
        }
        if (g1 & 1) {  // Init / Exit in progress: only calls that cannot hand out or free ids
            PlaneverbGetOutput((int)(rng() % 64));
            PlaneverbAddGeometry(pos(rng), pos(rng), 2.f, 0.5f, 0.9f);
            continue;
        }
        switch (rng() % 6) {
            case 0: em.push_back(PlaneverbEmit(pos(rng), 0.f, pos(rng))); break;
            case 1:
                if (!em.empty() && g_gen.load() == g1) {
                    PlaneverbEndEmission(em.back());
                    em.pop_back();
                }
                break;
            case 2: geo.push_back(PlaneverbAddGeometry(pos(rng), pos(rng), 2.f, 0.5f, 0.9f)); break;
            case 3:
                if (!geo.empty()) PlaneverbUpdateGeometry(geo[rng() % geo.size()], pos(rng), pos(rng), 1.f, 3.f, 0.7f);
                break;
            case 4:
                if (geo.size() > 8) {
                    PlaneverbRemoveGeometry(geo.back());
                    geo.pop_back();
                }
                break;
            default: PlaneverbGetOutput((int)(rng() % 64)); break;
        }
        if (em.size() > 200) em.clear();
    }
}

static void listenerThread(unsigned seed) {  // moves the listener (so the published window moves) and asks for IRs
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> pos(-1.f, kSize + 1.f);
    std::vector<PlaneverbCell> ir;
    while (!g_stop.load(std::memory_order_relaxed)) {
        PlaneverbSetListenerPosition(pos(rng), 0.f, pos(rng));
        if ((rng() & 15) == 0) {
            const float x = pos(rng), z = pos(rng);
            const int T = PlaneverbGetImpulseResponse(x, 0.f, z, nullptr, 0);
            if (T > 0) {
                ir.resize((size_t)T);
                const int got = PlaneverbGetImpulseResponse(x, 0.f, z, ir.data(), T);
                if (got == T && (ir[1].pr != 1.f || ir[(size_t)T - 1].pr != (float)(T - 1))) g_bad.fetch_add(1);
            }
        }
        std::this_thread::sleep_for(std::chrono::microseconds(300));
    }
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? std::atof(argv[1]) : 2.0;
    PvAmdInfo info;
    if (PvAmdHostGridInfo(kSize, kSize, kRes, &info) != 0) return 2;
    g_dx = info.dx;
    char dir[] = ".";

    // phase 1: steady hammering
    PlaneverbInit(kSize, kSize, kRes, 0, dir, 0, 1);
    if (!PlaneverbIsRunning()) return 3;
    std::vector<std::thread> th;
    th.emplace_back(getterThread, 1u);
    th.emplace_back(getterThread, 2u);
    th.emplace_back(churnThread, 3u);
    th.emplace_back(listenerThread, 4u);
    // phase 2 (same threads keep going): Exit / re-Init racing every call above
    const auto t0 = std::chrono::steady_clock::now();
    int cycles = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        std::this_thread::sleep_for(std::chrono::milliseconds(60));
        g_gen.fetch_add(1);
        if (cycles % 3 == 2) {
            PlaneverbExit();
            std::this_thread::sleep_for(std::chrono::milliseconds(5));  // every call meets a null context for a while
        }
        PlaneverbInit(kSize, kSize, kRes, 0, dir, 0, 1);  // Init while running = Exit + Init (PvContext.cpp:27-31)
        g_gen.fetch_add(1);
        ++cycles;
    }
    g_stop.store(true);
    for (auto& t : th) t.join();
    PlaneverbExit();
    PlaneverbExit();

    // phase 3: a worker error must be visible to the host, and the module must still answer and shut down cleanly
    pva::Solver::failAfterRuns().store(5);
    PlaneverbInit(kSize, kSize, kRes, 0, dir, 0, 1);
    const int e = PlaneverbEmit(5.f, 0.f, 6.f);
    for (int i = 0; i < 2000 && PlaneverbIsRunning(); ++i) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    int bad3 = 0;
    if (PlaneverbIsRunning()) bad3 |= 1;
    if (!std::strstr(PvAmdLastError(), "injected failure")) bad3 |= 2;
    {  // (two iterations in flight on two solvers: each fake solver fails on its own sixth run)
        const char* pl = std::getenv("PLANEVERB_AMD_LIVE_PIPELINE");
        if (PlaneverbIterationCount() != ((pl && std::atoi(pl) >= 2) ? 10 : 5)) bad3 |= 4;
    }
    if (!std::strstr(PlaneverbWorkerError(), "injected failure")) bad3 |= 16;
    // the worker's failure is reported once per thread: a later, unrelated error on this thread stays readable
    if (PvAmdHostLoadPv("/nonexistent/scene.pv", nullptr, 0) >= 0) bad3 |= 32;
    if (std::strstr(PvAmdLastError(), "injected failure") || !PvAmdLastError()[0]) bad3 |= 64;
    (void)PlaneverbGetOutput(e);
    PlaneverbExit();
    if (PlaneverbWorkerError()[0]) bad3 |= 128;
    pva::Solver::failAfterRuns().store(-1);
    if (pva::Solver::liveInstances().load() != 0) bad3 |= 8;

    // phase 4: Exit must not wait for a caller's own timeout -- a thread blocked in WaitIterations(far future, 30 s) holds a
    // pin; retiring the context wakes it
    int bad4 = 0;
    {
        PlaneverbInit(kSize, kSize, kRes, 0, dir, 0, 1);
        std::atomic<bool> entered{false};
        std::thread waiter([&] {
            entered.store(true);
            (void)PlaneverbWaitIterations(1LL << 40, 30000);
        });
        while (!entered.load()) std::this_thread::yield();
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        const auto a = std::chrono::steady_clock::now();
        PlaneverbExit();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
        waiter.join();
        if (ms > 5000.0) bad4 |= 1;
        if (pva::Solver::liveInstances().load() != 0) bad4 |= 2;
    }

    // phase 5: sparse-emitter mode of the live module (forced through the environment, as on the GPU): the worker registers
    // the emitters alive at the start of each iteration -- ended ids are left out
    int bad5 = 0;
    {
        setenv("PLANEVERB_AMD_LIVE_STREAMING", "1", 1);
        PlaneverbInit(kSize, kSize, kRes, 0, dir, 0, 1);
        if (PlaneverbIsStreaming() != 1) bad5 |= 1;
        const int a = PlaneverbEmit(5.f, 0.f, 6.f), b = PlaneverbEmit(7.f, 0.f, 6.f), c = PlaneverbEmit(9.f, 0.f, 6.f);
        (void)a;
        (void)c;
        PlaneverbWaitIterations(PlaneverbIterationCount() + 2, 5000);
        if (pva::Solver::lastEmitters().load() != 3) bad5 |= 2;
        PlaneverbEndEmission(b);
        PlaneverbWaitIterations(PlaneverbIterationCount() + 2, 5000);
        if (pva::Solver::lastEmitters().load() != 2) bad5 |= 4;
        PlaneverbExit();
        unsetenv("PLANEVERB_AMD_LIVE_STREAMING");
        PlaneverbInit(kSize, kSize, kRes, 0, dir, 0, 1);
        if (PlaneverbIsStreaming() != 0) bad5 |= 8;
        PlaneverbExit();
    }
    bad3 |= (bad4 << 8) | (bad5 << 12);

    std::printf("hammer: %lld reads (%lld in-window, %lld out-of-window, %lld sentinel), %d init/exit cycles, %lld bad, "
                "phase3 flags %d\n", g_reads.load(), g_window.load(), g_stale.load(), g_sentinel.load(), cycles,
                g_bad.load(), bad3);
    return (g_bad.load() == 0 && bad3 == 0 && g_reads.load() > 1000 && g_window.load() > 0 && g_stale.load() > 0) ? 0 : 1;
}
