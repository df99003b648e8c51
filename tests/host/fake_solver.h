// tests/host/fake_solver.h -- TEST INFRASTRUCTURE: a HIP-less stand-in for pva::Solver with the interface
// pv_context.cpp uses, so that the live module's HOST logic (context lifetime, emitter / geometry tables, the worker
// loop, the publish step and its sequence lock, pv_core.cpp's rasteriser) runs under ThreadSanitizer and
// AddressSanitizer in the CPU container ("run host code under TSan/ASan"; the GPU pool has no
// sanitizer support).  It computes nothing acoustic: every published record encodes (iteration, cell) so that the
// hammer can tell a torn or mis-attributed read from a good one.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "pv_core.h"

namespace pva {

struct SolverOptions {
    bool streaming = false, autoStreaming = false;
};

class Solver {
public:
    struct WindowBlock {
        int r0 = 0, c0 = 0, nr = 0, nc = 0;
        float lx = 0, lz = 0;
    };
    static constexpr int kWin = 24;  // the fake's "history window": kWin x kWin cells around the listener cell

    // test hooks
    static std::atomic<int>& failAfterRuns() {
        static std::atomic<int> v{-1};
        return v;
    }
    static std::atomic<int>& lastEmitters() {
        static std::atomic<int> v{-1};
        return v;
    }
    static std::atomic<bool>& throwInRun() {  // the next run() throws (the worker's exception barrier, alloc_fault.cpp)
        static std::atomic<bool> v{false};
        return v;
    }
    static std::atomic<long long>& liveInstances() {
        static std::atomic<long long> v{0};
        return v;
    }

    static Solver* create(const GridSpec& spec, int, const SolverOptions& o, std::string* err) {
        if (spec.gx < 1 || spec.gy < 1) {
            if (err) *err = "grid has no cells";
            return nullptr;
        }
        std::unique_ptr<Solver> s(new Solver());  // (owned across init, as the real Solver::create: alloc_fault.cpp)
        s->opt_ = o;
        s->g_ = spec;
        s->mat_.init(spec);
        return s.release();
    }
    ~Solver() { liveInstances().fetch_sub(1); }

    const GridSpec& spec() const { return g_; }
    int T() const { return g_.T; }
    const std::string& lastError() const { return err_; }
    SolverOptions& options() { return opt_; }
    bool setEmitters(const float*, int n) {
        emitters_ = n;
        lastEmitters().store(n);
        return true;
    }
    void rasterAdd(const Box& b) { mat_.add(b); }
    void rasterRemove(const Box& b) { mat_.remove(b); }

    bool residentKernel() const { return false; }
    bool run(float lx, float, float lz, bool, Solver* carryFrom = nullptr) {
        (void)carryFrom;
        if (throwInRun().exchange(false)) throw std::runtime_error("fake solver: injected exception");
        const int fa = failAfterRuns().load();
        if (fa >= 0 && runs_ >= fa) {
            err_ = "fake solver: injected failure";
            return false;
        }
        ++runs_;
        lx_ = lx;
        lz_ = lz;
        int cx, cy;
        listenerCell(g_, lx, lz, &cx, &cy);
        cx = std::min(std::max(cx, 0), g_.gx - 1);
        cy = std::min(std::max(cy, 0), g_.gy - 1);
        r0_ = std::min(std::max(cx - kWin / 2, 0), std::max(0, g_.gx - kWin));
        c0_ = std::min(std::max(cy - kWin / 2, 0), std::max(0, g_.gy - kWin));
        mat_.clearDirty();
        std::this_thread::sleep_for(std::chrono::microseconds(200));  // "the GPU is busy"
        return true;
    }
    bool sync() { return true; }
    size_t windowCapacity() const { return (size_t)std::min(kWin, g_.gx) * (size_t)std::min(kWin, g_.gy); }

    // record of cell (cx, cy) published by run number `it`
    static void record(long long it, int cx, int cy, float out[8]) {
        out[0] = (float)(it % 1000000);
        out[1] = (float)cx;
        out[2] = (float)cy;
        out[3] = (float)((it * 7 + cx) % 1000);
        out[4] = 0.25f;
        out[5] = -0.5f;
        out[6] = (float)(it % 1000000);
        out[7] = (float)(cx + cy);
    }

    bool waitPublish() { return true; }
    bool publishWindowAsync(float* dst, WindowBlock* info, bool = false) {
        WindowBlock w;
        w.r0 = r0_;
        w.c0 = c0_;
        w.nr = std::min(kWin, g_.gx - r0_);
        w.nc = std::min(kWin, g_.gy - c0_);
        w.lx = lx_;
        w.lz = lz_;
        // what the DMA engine does on the GPU box; relaxed atomics because a late reader may still be looking
        // at this slot (it then discards what it read, pv_context.cpp's sequence lock)
        for (int r = 0; r < w.nr; ++r)
            for (int c = 0; c < w.nc; ++c) {
                float v[8];
                record(runs_, w.r0 + r, w.c0 + c, v);
                for (int k = 0; k < 8; ++k) {
                    uint32_t u;
                    std::memcpy(&u, &v[k], 4);
                    __atomic_store_n(reinterpret_cast<uint32_t*>(dst) + ((size_t)r * w.nc + c) * 8 + k, u,
                                     __ATOMIC_RELAXED);
                }
            }
        *info = w;
        return true;
    }

    bool impulseResponseCells(int cx, int cy, void* out16T) {
        if (runs_ == 0) {
            err_ = "no simulation has run yet";
            return false;
        }
        struct RefCell {
            float pr, vx, vy;
            short b, by;
        };
        RefCell* o = static_cast<RefCell*>(out16T);
        const size_t i = (size_t)cx * g_.NY + cy;
        for (int t = 0; t < g_.T; ++t)
            o[t] = RefCell{(float)t, (float)cx, (float)cy, (short)mat_.beta()[i], (short)mat_.by()[i]};
        return true;
    }

    static void* hostAlloc(size_t bytes) { return std::malloc(bytes); }
    static void hostFree(void* p) { std::free(p); }

private:
    Solver() { liveInstances().fetch_add(1); }
    GridSpec g_;
    SolverOptions opt_;
    int emitters_ = 0;
    MaterialPlane mat_;
    std::string err_;
    long long runs_ = 0;
    float lx_ = 0, lz_ = 0;
    int r0_ = 0, c0_ = 0;
};

}  // namespace pva
