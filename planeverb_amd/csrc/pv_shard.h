// pv_shard.h -- independent runs sharded over the GPUs of a node, host side in C++.
//
// A "run" is one listener position on one scene (one pass of the reference's background loop,
// Context/PvContext.cpp:74-93).  Runs share nothing: run k belongs to rank k mod W (a rank = one process, normally one
// GPU), inside a rank the runs go round-robin over the rank's solvers, which the calling thread keeps busy through
// their own HIP streams (RunAsync / Sync).  The only exchange is one all-gather of the per-emitter PlaneverbOutput
// records -- RCCL (ncclAllGather) over xGMI when the ranks span processes.  RCCL is bound at run time (dlopen of the
// librccl the process already carries, else the system's): the library itself does not link against it, so the
// single-GPU drop-in has no RCCL dependency.
#pragma once

#include <string>
#include <vector>

#include <hip/hip_runtime.h>

namespace pva {

// which runs a rank simulates and on which of its solvers: run k -> rank k mod W; the rank's j-th run -> solver j mod n
struct ShardItem {
    int run;
    int solver;
};
std::vector<ShardItem> shardPlan(int nRuns, int world, int rank, int nLocalSolvers);

class Comm {
public:
    static bool uniqueId(char out[128], std::string* err);  // rank 0 creates it, the caller hands it to every rank
    static Comm* create(const char id[128], int rank, int world, int device, std::string* err);
    ~Comm();
    int rank() const { return rank_; }
    int world() const { return world_; }
    // every rank contributes `countPerRank` floats; all = world * countPerRank floats (rank-major), on every rank
    bool allGather(const float* mine, int countPerRank, float* all, std::string* err);

private:
    Comm() = default;
    void* comm_ = nullptr;  // ncclComm_t
    int rank_ = 0, world_ = 1, device_ = 0;
    hipStream_t stream_ = nullptr;
    float* send_ = nullptr;
    float* recv_ = nullptr;
    size_t cap_ = 0;
};

}  // namespace pva
