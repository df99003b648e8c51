// tests/host/alloc_fault.cpp -- TEST INFRASTRUCTURE: the C-ABI's exception barrier under allocation failure.
//
// the reference throws from Init (PvContext.cpp:106,123, Grid.cpp:69); "a C-ABI must not leak C++
// exceptions -- catch at the boundary".  This program replaces the global operator new of a HIP-less build of the live
// module (pv_core.cpp + pv_context.cpp + pv_capi.cpp against tests/host/fake_solver.h) and, for every Part 1 entry point
// that can allocate, fails the 0th, 1st, 2nd ... allocation the call makes, one at a time, until the call gets through
// without reaching the armed allocation.  After every injected failure:
//   * the call has RETURNED (an exception that crossed extern "C" would have ended the process: std::terminate),
//   * with the function's failure sentinel, and PvAmdLastError() names the function ("exception in <name>: ..."),
//   * the tables are what they were (the next un-faulted call hands out the id the failed one would have),
//   * and the module keeps serving GetOutput.
// Then the same for an allocation on the WORKER thread: the worker stops like on a solver error (IsRunning 0,
// PlaneverbWorkerError names the exception), the host is not taken down, Exit + Init bring the module back.
// Built with -fsanitize=address,undefined: the unwinding paths must not leak or touch freed memory either.
//
//   alloc_fault           exit code 0 = every sweep held
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>

#include "planeverb_amd.h"
#include "fake_solver.h"

// ---- the allocator hook ------------------------------------------------------------------------------------------
static thread_local long t_failAt = -1;       // >= 0: fail the allocation that finds this at 0 (this thread), once
static thread_local long t_count = 0;         // allocations made by this thread (for "GetOutput allocates nothing")
static thread_local bool t_fired = false;
static thread_local bool t_isMain = false;
static std::atomic<long> g_workerFailAt{-1};  // the same for any thread that is not main
static std::atomic<bool> g_workerFired{false};

static void* hookedAlloc(std::size_t n) {
    ++t_count;
    if (t_failAt >= 0 && t_failAt-- == 0) {
        t_fired = true;
        throw std::bad_alloc();
    }
    if (!t_isMain && g_workerFailAt.load(std::memory_order_relaxed) >= 0 && g_workerFailAt.fetch_sub(1) == 0) {
        g_workerFired.store(true);
        throw std::bad_alloc();
    }
    void* p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void* operator new(std::size_t n) { return hookedAlloc(n); }
void* operator new[](std::size_t n) { return hookedAlloc(n); }
void operator delete(void* p) noexcept { std::free(p); }
void operator delete[](void* p) noexcept { std::free(p); }
void operator delete(void* p, std::size_t) noexcept { std::free(p); }
void operator delete[](void* p, std::size_t) noexcept { std::free(p); }

// ---- the sweeps --------------------------------------------------------------------------------------------------
static int g_failures = 0;
#define EXPECT(cond, ...)                                 \
    do {                                                  \
        if (!(cond)) {                                    \
            ++g_failures;                                 \
            std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__);            \
            std::fprintf(stderr, "\n");                   \
        }                                                 \
    } while (0)

static const float kSize = 25.f;
static const int kRes = 275;
static char g_tmp[] = "";

static bool lastErrorNames(const char* fn) {
    const char* e = PvAmdLastError();
    return e && std::strstr(e, "exception in ") && std::strstr(e, fn) && std::strstr(e, "bad_alloc");
}

// per entry point: calls swept, allocations failed in all
struct Tally {
    const char* what;
    long calls, faults;
};
static Tally g_tally[32];
static int g_ntally = 0;
static void tally(const char* what, int faults) {
    int i = 0;
    while (i < g_ntally && std::strcmp(g_tally[i].what, what)) ++i;
    if (i == g_ntally) g_tally[g_ntally++] = Tally{what, 0, 0};
    ++g_tally[i].calls;
    g_tally[i].faults += faults;
}

// fails allocation k = 0, 1, 2 ... of `call` until it runs through; `after(faulted, k)` checks the outcome of every attempt
template <class Call, class After>
static int sweep(const char* what, Call call, After after, int limit = 4000) {
    int k = 0;
    for (; k < limit; ++k) {
        t_fired = false;
        t_failAt = k;
        call();
        t_failAt = -1;
        const bool faulted = t_fired;
        after(faulted, k);
        if (!faulted) break;
    }
    EXPECT(k < limit, "%s: still faulting after %d allocations", what, limit);
    tally(what, k);
    return k;
}

static bool waitRunning(long long iters) {
    PlaneverbWaitIterations(iters, 5000);
    return PlaneverbIsRunning() == 1 && PlaneverbIterationCount() >= iters;
}

int main() {
    t_isMain = true;
    const char* pipeline = std::getenv("PLANEVERB_AMD_LIVE_PIPELINE");
    std::printf("alloc_fault: pipeline=%s\n", pipeline ? pipeline : "(default)");

    // 1. PlaneverbInit: every allocation on the way up (context, solver(s), material planes, result slots, worker thread)
    int inits = sweep("PlaneverbInit", [] { PlaneverbInit(kSize, kSize, kRes, 0, g_tmp, 0, 0); },
                      [](bool faulted, int k) {
                          if (faulted) {
                              EXPECT(PlaneverbIsRunning() == 0, "Init faulted at allocation %d but the module runs", k);
                              EXPECT(lastErrorNames("PlaneverbInit"), "Init, allocation %d: last error '%s'", k, PvAmdLastError());
                              EXPECT(pva::Solver::liveInstances().load() == 0, "Init, allocation %d: %lld solver(s) left behind", k,
                                     pva::Solver::liveInstances().load());
                              EXPECT(PlaneverbGetOutput(0).occlusion == -1.f, "no module: GetOutput must return the sentinel");
                              EXPECT(PlaneverbEmit(1, 0, 1) == -1, "no module: Emit must return -1");
                          }
                      });
    EXPECT(inits > 3, "PlaneverbInit made only %d allocations: the hook is not in the path", inits);
    EXPECT(waitRunning(1), "module did not come up after the Init sweep: %s", PvAmdLastError());

    // 2. AddGeometry: the id a faulted call would have handed out goes to the next one
    int nextId = 0;
    for (int round = 0; round < 40; ++round) {  // (40 boxes: the tables grow several times)
        int got = -2;
        sweep("PlaneverbAddGeometry", [&] { got = PlaneverbAddGeometry(5.f + 0.1f * round, 5.f, 1.f, 1.f, 0.5f); },
              [&](bool faulted, int k) {
                  if (faulted) {
                      EXPECT(got == -1, "AddGeometry faulted at allocation %d and returned %d", k, got);
                      EXPECT(lastErrorNames("PlaneverbAddGeometry"), "AddGeometry: last error '%s'", PvAmdLastError());
                  } else {
                      EXPECT(got == nextId, "AddGeometry returned id %d, expected %d (a faulted call changed the table)", got, nextId);
                  }
              });
        ++nextId;
    }
    // 3. Update / Remove: void calls -- they return, the module runs on, a removed id is re-used exactly once
    for (int id = 0; id < 12; ++id) {
        std::this_thread::sleep_for(std::chrono::milliseconds(2));  // (the worker swaps the change queue out: room is needed again)
        sweep("PlaneverbUpdateGeometry", [&] { PlaneverbUpdateGeometry(id, 6.f, 6.f, 2.f, 1.f, 0.25f); },
              [&](bool faulted, int) {
                  if (faulted) EXPECT(lastErrorNames("PlaneverbUpdateGeometry"), "UpdateGeometry: last error '%s'", PvAmdLastError());
              });
    }
    {
        bool removed = false;
        sweep("PlaneverbRemoveGeometry", [&] { PlaneverbRemoveGeometry(7); },
              [&](bool faulted, int) {
                  if (faulted)
                      EXPECT(lastErrorNames("PlaneverbRemoveGeometry"), "RemoveGeometry: last error '%s'", PvAmdLastError());
                  else
                      removed = true;
              });
        EXPECT(removed, "RemoveGeometry never got through");
        const int a = PlaneverbAddGeometry(1, 1, 1, 1, 0.5f), b = PlaneverbAddGeometry(2, 2, 1, 1, 0.5f);
        EXPECT(a == 7 && b == nextId, "after Remove(7): ids %d, %d (expected 7, %d)", a, b, nextId);
        ++nextId;
    }
    // 4. Emit / EndEmission
    int nextEmit = 0;
    for (int round = 0; round < 300; ++round) {  // (crosses an emitter chunk boundary)
        int got = -2;
        sweep("PlaneverbEmit", [&] { got = PlaneverbEmit(3.f, 0.f, 4.f); },
              [&](bool faulted, int) {
                  if (faulted) {
                      EXPECT(got == -1, "Emit faulted and returned %d", got);
                      EXPECT(lastErrorNames("PlaneverbEmit"), "Emit: last error '%s'", PvAmdLastError());
                  } else {
                      EXPECT(got == nextEmit, "Emit returned id %d, expected %d", got, nextEmit);
                  }
              }, 64);
        ++nextEmit;
    }
    for (int id = 0; id < 40; ++id)
        sweep("PlaneverbEndEmission", [&] { PlaneverbEndEmission(id); }, [&](bool faulted, int) {
            if (faulted) EXPECT(lastErrorNames("PlaneverbEndEmission"), "EndEmission: last error '%s'", PvAmdLastError());
        }, 64);
    // 5. GetOutput / UpdateEmission / SetListenerPosition: the per-frame calls allocate NOTHING (so they cannot fail that way)
    {
        const long before = t_count;
        for (int i = 0; i < 1000; ++i) {
            PlaneverbUpdateEmission(100 + (i % 50), 3.f + 0.01f * i, 0.f, 4.f);
            PlaneverbSetListenerPosition(5.f, 0.f, 5.f + 0.001f * i);
            const PlaneverbOutput o = PlaneverbGetOutput(100 + (i % 50));
            (void)o;
        }
        EXPECT(t_count == before, "the per-frame calls made %ld allocation(s)", t_count - before);
    }
    // 6. GetImpulseResponse (a T x 16 byte staging buffer) and LoadScene (a file, a vector of boxes)
    {
        static PlaneverbCell cells[8];
        int got = -2;
        sweep("PlaneverbGetImpulseResponse", [&] { got = PlaneverbGetImpulseResponse(3.f, 0.f, 4.f, cells, 8); },
              [&](bool faulted, int) {
                  if (faulted) {
                      EXPECT(got == -1, "GetImpulseResponse faulted and returned %d", got);
                      EXPECT(lastErrorNames("PlaneverbGetImpulseResponse"), "GetImpulseResponse: last error '%s'", PvAmdLastError());
                  } else {
                      EXPECT(got > 0, "GetImpulseResponse returned %d", got);
                  }
              });
        EXPECT(PlaneverbIsRunning() == 1, "worker stopped after the GetImpulseResponse sweep: %s", PlaneverbWorkerError());
    }
    if (const char* scene = std::getenv("PV_TEST_SCENE")) {
        int got = -2, faults = 0;
        sweep("PlaneverbLoadScene", [&] { got = PlaneverbLoadScene(scene); },
              [&](bool faulted, int) {
                  if (faulted) {
                      ++faults;
                      // (boxes added before the fault stay added -- each AddGeometry is atomic, the scene as a whole is not)
                      EXPECT(got == -1, "LoadScene faulted and returned %d", got);
                      // (an allocation that fails inside the iostream is swallowed there -- badbit -- and comes out as the
                      // loader's own "truncated scene file"; one in the box table comes out of the barrier)
                      EXPECT(lastErrorNames("PlaneverbLoadScene") || std::strstr(PvAmdLastError(), "scene file"),
                             "LoadScene: last error '%s'", PvAmdLastError());
                  } else {
                      EXPECT(got > 0, "LoadScene returned %d (%s)", got, PvAmdLastError());
                  }
              });
        EXPECT(faults > 0, "LoadScene made no allocation");
    }
    EXPECT(waitRunning(PlaneverbIterationCount() + 2), "module stopped during the main-thread sweeps: %s", PlaneverbWorkerError());

    // 7. the worker thread: its next allocation fails (the geometry queue it swaps out, the emitter list) -- it must stop
    //    like on a solver error, not std::terminate the host
    {
        g_workerFired.store(false);
        g_workerFailAt.store(0);
        for (int i = 0; i < 2000 && !g_workerFired.load(); ++i) {
            PlaneverbAddGeometry(8.f, 8.f, 0.5f, 0.5f, 0.5f);  // (gives the worker a change queue to allocate for)
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        g_workerFailAt.store(-1);
        if (g_workerFired.load()) {
            for (int i = 0; i < 2000 && PlaneverbIsRunning(); ++i) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            EXPECT(PlaneverbIsRunning() == 0, "a worker allocation failed and the worker still reports running");
            const std::string w = PlaneverbWorkerError();
            EXPECT(w.find("exception") != std::string::npos, "worker error '%s'", w.c_str());
            (void)PlaneverbGetOutput(150);  // the last published iteration is still served
            std::printf("  worker stopped with: %s\n", w.c_str());
        } else {
            std::printf("  (the worker made no allocation in 2 s: nothing to fail)\n");
        }
        PlaneverbExit();
        EXPECT(pva::Solver::liveInstances().load() == 0, "Exit left %lld solver(s)", pva::Solver::liveInstances().load());
        PlaneverbInit(kSize, kSize, kRes, 0, g_tmp, 0, 0);
        EXPECT(waitRunning(2), "module did not come back after the worker fault: %s", PvAmdLastError());
    }
    // 7b. the same through an exception thrown by the solver itself inside an iteration (what a C++ runtime underneath may do)
    {
        pva::Solver::throwInRun().store(true);
        for (int i = 0; i < 5000 && PlaneverbIsRunning(); ++i) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        EXPECT(PlaneverbIsRunning() == 0, "the solver threw inside an iteration and the worker still reports running");
        const std::string w = PlaneverbWorkerError();
        EXPECT(w.find("injected exception") != std::string::npos, "worker error '%s'", w.c_str());
        EXPECT(std::strstr(PvAmdLastError(), "simulation worker stopped") != nullptr, "last error '%s'", PvAmdLastError());
        (void)PlaneverbGetOutput(150);  // the last published iteration is still served
        EXPECT(PlaneverbAddGeometry(1, 1, 1, 1, 0.5f) >= 0, "tables must stay usable beside a stopped worker");
        PlaneverbExit();
        PlaneverbInit(kSize, kSize, kRes, 0, g_tmp, 0, 0);
        EXPECT(waitRunning(2), "module did not come back after the worker's exception: %s", PvAmdLastError());
    }
    // 8. Exit under fault, then a clean end
    sweep("PlaneverbExit", [] { PlaneverbExit(); }, [](bool, int) {});
    PlaneverbExit();
    EXPECT(PlaneverbIsRunning() == 0 && pva::Solver::liveInstances().load() == 0, "module still up after Exit");

    for (int i = 0; i < g_ntally; ++i)
        std::printf("  %-28s %5ld call(s) swept, %6ld allocation(s) failed one at a time\n", g_tally[i].what, g_tally[i].calls,
                    g_tally[i].faults);
    std::printf("alloc_fault: %d failure(s)\n", g_failures);
    return g_failures ? 1 : 0;
}
