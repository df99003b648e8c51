// pv_shard.cpp -- see pv_shard.h
#include "pv_shard.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

namespace pva {

std::vector<ShardItem> shardPlan(int nRuns, int world, int rank, int nLocalSolvers) {
    std::vector<ShardItem> plan;
    if (world < 1 || rank < 0 || rank >= world || nLocalSolvers < 1) return plan;
    int j = 0;
    for (int k = rank; k < nRuns; k += world, ++j) plan.push_back(ShardItem{k, j % nLocalSolvers});
    return plan;
}

// ----------------------------------------------------------------------------------------------------------------
// RCCL, bound at run time
// ----------------------------------------------------------------------------------------------------------------
namespace {

struct NcclId {
    char internal[128];  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
};
using ncclComm_t = void*;
constexpr int kNcclFloat32 = 7;  // rccl.h:466

struct Rccl {
    void* lib = nullptr;
    int (*getUniqueId)(NcclId*) = nullptr;
    int (*commInitRank)(ncclComm_t*, int, NcclId, int) = nullptr;
    int (*commDestroy)(ncclComm_t) = nullptr;
    int (*allGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*errorString)(int) = nullptr;
    int (*getVersion)(int*) = nullptr;
    std::string why;
    std::string bound;  // which library was bound, and the version it reports: part of every error message
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // 1. the librccl this process already carries (PyTorch ships its own next to its HIP runtime: mixing it with the
        //    system's would put two HIP runtimes behind one communicator), found in the process's memory map
        std::string path;
        if (const char* e = std::getenv("PLANEVERB_AMD_RCCL")) path = e;
        if (path.empty()) {
            if (FILE* f = std::fopen("/proc/self/maps", "r")) {
                char line[1024];
                while (std::fgets(line, sizeof(line), f)) {
                    const char* p = std::strstr(line, "librccl");
                    if (!p) continue;
                    const char* s = std::strchr(line, '/');
                    if (!s) continue;
                    path.assign(s);
                    while (!path.empty() && (path.back() == '\n' || path.back() == ' ')) path.pop_back();
                    break;
                }
                std::fclose(f);
            }
        }
        // dlerror() hands out its message ONCE and clears it: read it right after each failed dlopen, keep the last
        std::string lastErr;
        auto tryOpen = [&](const char* name) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!r.lib) {
                const char* e = dlerror();
                lastErr = e ? e : "?";
            } else {
                r.bound = name;
            }
        };
        if (!path.empty()) tryOpen(path.c_str());
        // 2. the system's
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if (!r.lib) tryOpen(name);
        if (!r.lib) {
            r.why = std::string("librccl not found: ") + (lastErr.empty() ? "?" : lastErr);
            return;
        }
        r.getUniqueId = reinterpret_cast<decltype(r.getUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.commInitRank = reinterpret_cast<decltype(r.commInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.commDestroy = reinterpret_cast<decltype(r.commDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.allGather = reinterpret_cast<decltype(r.allGather)>(dlsym(r.lib, "ncclAllGather"));
        r.errorString = reinterpret_cast<decltype(r.errorString)>(dlsym(r.lib, "ncclGetErrorString"));
        r.getVersion = reinterpret_cast<decltype(r.getVersion)>(dlsym(r.lib, "ncclGetVersion"));
        // The four entry points above are declared HERE (no rccl.h at build time: the library is bound at run time), with the
        // layout of ncclUniqueId (128 bytes, by value) and ncclFloat32 = 7 of the RCCL this was written against (2.26).  Should a
        // later RCCL change either, a first multi-rank contact fails inside ncclCommInitRank / ncclAllGather: every error message
        // therefore says which library was bound and which version it reports.
        int v = 0;
        if (r.getVersion && r.getVersion(&v) == 0)
            r.bound += " (ncclGetVersion " + std::to_string(v) + " = " + std::to_string(v / 10000) + "." + std::to_string(v / 100 % 100) + "." +
                       std::to_string(v % 100) + "; written against 2.26)";
        else
            r.bound += " (no ncclGetVersion)";
        if (!r.getUniqueId || !r.commInitRank || !r.commDestroy || !r.allGather) r.why = "librccl lacks an nccl* entry point: " + r.bound;
    });
    return r;
}

bool ncclOk(int rc, const char* what, std::string* err) {
    if (rc == 0) return true;
    Rccl& r = rccl();
    if (err)
        *err = std::string(what) + ": " + (r.errorString ? r.errorString(rc) : "RCCL error " + std::to_string(rc)) + " [bound " + r.bound + "]";
    return false;
}

}  // namespace

bool Comm::uniqueId(char out[128], std::string* err) {
    Rccl& r = rccl();
    if (!r.why.empty()) {
        if (err) *err = r.why;
        return false;
    }
    NcclId id;
    if (!ncclOk(r.getUniqueId(&id), "ncclGetUniqueId", err)) return false;
    std::memcpy(out, id.internal, 128);
    return true;
}

Comm* Comm::create(const char idBytes[128], int rank, int world, int device, std::string* err) {
    Rccl& r = rccl();
    if (!r.why.empty()) {
        if (err) *err = r.why;
        return nullptr;
    }
    if (world < 1 || rank < 0 || rank >= world) {
        if (err) *err = "invalid rank / world size";
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        if (err) *err = "hipSetDevice failed";
        return nullptr;
    }
    std::unique_ptr<Comm> c(new Comm());
    c->rank_ = rank;
    c->world_ = world;
    c->device_ = device;
    NcclId id;
    std::memcpy(id.internal, idBytes, 128);
    if (hipStreamCreateWithFlags(&c->stream_, hipStreamNonBlocking) != hipSuccess ||
        !ncclOk(r.commInitRank(&c->comm_, world, id, rank), "ncclCommInitRank", err)) {
        if (err && err->empty()) *err = "hipStreamCreate failed";
        return nullptr;
    }
    // first contact: one float per rank through the very collective of the data path.  ncclFloat32 and the by-value ncclUniqueId
    // are declared by hand above; a librccl that disagrees with them shows here, at creation, with a message -- not as garbage
    // records (or a hang) in the middle of a sharded job
    std::vector<float> all((size_t)world, -1.f);
    const float mine = (float)(rank + 1);
    std::string e;
    bool ok = c->allGather(&mine, 1, all.data(), &e);
    for (int i = 0; ok && i < world; ++i) ok = all[(size_t)i] == (float)(i + 1);
    if (!ok) {
        if (err) *err = "RCCL self-test at communicator creation failed (" + (e.empty() ? std::string("wrong values gathered") : e) + ") [bound " + r.bound + "]";
        return nullptr;
    }
    return c.release();
}

Comm::~Comm() {
    hipSetDevice(device_);
    if (stream_) hipStreamSynchronize(stream_);
    if (comm_) rccl().commDestroy(comm_);
    if (send_) hipFree(send_);
    if (recv_) hipFree(recv_);
    if (stream_) hipStreamDestroy(stream_);
}

bool Comm::allGather(const float* mine, int countPerRank, float* all, std::string* err) {
    auto bad = [&](const char* what) {
        if (err) *err = what;
        return false;
    };
    // a collective: every rank must enter it with the same count.  A rank that skipped it for "nothing to send" would
    // leave the others blocked inside ncclAllGather, so an empty contribution is an error here and the callers pad
    // (dist.gather_outputs_native, PvAmdRunSharded: per_rank * n_emitters * 8 floats on every rank, zero-filled).
    if (countPerRank <= 0) return bad("PvAmdCommAllGather: countPerRank must be > 0 and equal on every rank");
    if (hipSetDevice(device_) != hipSuccess) return bad("hipSetDevice failed");
    const size_t n = (size_t)countPerRank;
    if (n > cap_) {
        if (send_) hipFree(send_);
        if (recv_) hipFree(recv_);
        send_ = recv_ = nullptr;
        if (hipMalloc((void**)&send_, n * 4) != hipSuccess || hipMalloc((void**)&recv_, n * 4 * (size_t)world_) != hipSuccess)
            return bad("hipMalloc failed for the gather buffers");
        cap_ = n;
    }
    if (hipMemcpyAsync(send_, mine, n * 4, hipMemcpyHostToDevice, stream_) != hipSuccess) return bad("gather upload failed");
    if (!ncclOk(rccl().allGather(send_, recv_, n, kNcclFloat32, comm_, stream_), "ncclAllGather", err)) return false;
    if (hipMemcpyAsync(all, recv_, n * 4 * (size_t)world_, hipMemcpyDeviceToHost, stream_) != hipSuccess)
        return bad("gather download failed");
    if (hipStreamSynchronize(stream_) != hipSuccess) return bad("gather sync failed");
    return true;
}

}  // namespace pva
