// pv_rt60.hip -- wet gain and decay time (Analyzer.cpp:235-247, 282-327) for ANY number of reached cells.
//
// Per cell and tail sample the reference evaluates one log10f and advances three float32 sums whose order is part of the
// result: the backward energy integral, sum(y x) and sum(y).  Rounds 1-3 had two forms: one thread per cell
// inside pv_encode_kernel (the "cell form": 636-940 cycles per wave and sample, its eight logarithms re-serialised behind
// compiler-inserted branches, each with its table load at the head of the chain: 27-33 % of a 254^2 / 382^2 run) and sixteen
// lanes per cell (pv_rt60_wave_kernel: parallel logarithms, but 96 of its ~160 instructions per 16 samples are the DPP
// chains that keep the sums sequential; up to 65 536 cells).  This is the blocked form for everything above a few thousand
// cells:
//   * L lanes share a cell, each holds S CONSECUTIVE samples of a chunk of L * S (lane j: samples i0 - j S - k, k < S);
//   * the energy integral is a chain over the L lanes -- lane j takes lane j-1's value by one DPP rotation, adds its S
//     squares one after the other, keeps the S partial sums -- so the order of additions is the reference's; lanes outside
//     a range add +0.0f (the identity of a sum that started at +0, bit for bit);
//   * the S logarithms of a lane are independent: branch-free glibc log10f (pv_libm.h) with its 16-entry table in LDS, one
//     basic block, interleaved by the compiler;
//   * the two regression sums are chains like the first.
// Instructions per cell and sample: ~0.9 at (L, S) = (4, 4) against ~2.5 for the sixteen-lane form; which form runs is
// chosen ON THE DEVICE from the number of cells of the window's ever-non-zero tiles (rt60LanesPerCell): a few thousand
// cells need sixteen lanes each for parallelism, everything above runs four.  Same bits in every form.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "pv_analysis.h"
#include "pv_device.h"
#include "pv_launch.h"
#include "pv_libm.h"
#include "pv_prims.h"

namespace pva {

namespace {

struct LogTabLds {
    const double* t;  // 48 x {invc, logc + kk ln2} in LDS, entry (kk + 1) * 16 + i
    __device__ __forceinline__ void operator()(int i, int kk, double* invc, double* y0) const {
        const int e = (kk + 1) * 16 + i;
        *invc = t[2 * e];
        *y0 = t[2 * e + 1];
    }
};

// value of lane j - 1 of this lane's group of L (lane 0 reads lane L - 1)
template <int L>
__device__ __forceinline__ float prevInGroup(float v) {
    if constexpr (L == 16)  // DPP row_ror:1
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    else if constexpr (L == 4)  // DPP quad_perm:[3,0,1,2]
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x93, 0xf, 0xf, false));
    else
        return v;
}

// One chunk of a sequential sum shared by the L lanes of a group: acc = (...((acc + add_0[0]) + add_0[1]) ... + add_{L-1}[S-1]),
// lane j's addends after lane j-1's.  Every lane runs every step (SIMD), lane j keeps step j; the value a chunk ends with
// stays in lane L-1, which is where lane 0 of the next chunk fetches it from.  KEEP: also the S partial sums of the own step.
template <int L, int S, bool KEEP>
__device__ __forceinline__ void groupChain(float& acc, const float (&add)[S], const int sub, float (&part)[S]) {
#pragma unroll
    for (int j = 0; j < L; ++j) {
        float t = prevInGroup<L>(acc);
        float tmp[S];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            t = t + add[k];
            tmp[k] = t;
        }
        const bool mine = (L == 1) || sub == j;
        acc = mine ? t : acc;
        if (KEEP) {
#pragma unroll
            for (int k = 0; k < S; ++k) part[k] = mine ? tmp[k] : part[k];
        }
    }
}

template <int L, int S>
__global__ __launch_bounds__(256) void pv_rt60_blocked_kernel(const AnalyzeArgs a) {
    __shared__ double tab[96];
    if (analysisAborted(a)) return;
    if (rt60LanesPerCell(a, *a.activeCount) != L) return;  // (grid-uniform: the other instantiations' launches do the work)
    if (threadIdx.x < 96) {
        double invc, y0;
        const int e = (int)threadIdx.x >> 1;
        PvLogTabConst{}(e & 15, (e >> 4) - 1, &invc, &y0);
        tab[threadIdx.x] = (threadIdx.x & 1) ? y0 : invc;
    }
    __syncthreads();
    const LogTabLds ltab{tab};
    const DynParams dyn = *a.dyn;
    constexpr int CPB = 256 / L;  // cells per block, along the window's columns; one window row per blockIdx.y
    const int sub = threadIdx.x % L;
    const int wc = blockIdx.x * CPB + threadIdx.x / L, wr = blockIdx.y;
    Rt60Cell c{-1, {nullptr, 0}, 0};
    if (wc < a.winCols) c = rt60Cell(a, dyn, dyn.histRow0 - a.G + wr, dyn.histCol0 - a.G + wc);
    if (__ballot(c.s >= 0) == 0ull) return;  // a wave leaves only when none of its cells has work
    const bool live = c.s >= 0;
    const int T = a.T;
    const int endPoint = T - a.nCut;
    const int startingPoint = live ? c.startingPoint : T;  // dead lanes: no sample is in range
    const int lowest = min(startingPoint, endPoint);         // the pre-sum over [endPoint, T) is not bounded by the onset
    const long long plane = a.histPlane;
    const float* const h0 = live ? c.hc.h : a.hist;

    // ---- decay time: backwards from T - 1; wave-uniform trip count = the longest of the wave's cells ----
    int n = T - lowest;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) n = max(n, __shfl_xor(n, off));
    float edc = 0.f, xysum = 0.f, ysum = 0.f;
    float pNext[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const int i = T - 1 - sub * S - k;
        pNext[k] = (live && i >= lowest && i >= 0) ? h0[(long long)i * plane] : 0.f;
    }
#pragma unroll 1
    for (int n0 = 0; n0 < n; n0 += L * S) {
        const int iTop = T - 1 - n0 - sub * S;  // this lane's samples: iTop - k
        float q[S];
#pragma unroll
        for (int k = 0; k < S; ++k) q[k] = pNext[k] * pNext[k];  // 0 outside [lowest, T): edc + 0 = edc
        if (n0 + L * S < n) {  // the next chunk's loads are in flight while this chunk's chains and logarithms run
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const int i = iTop - L * S - k;
                pNext[k] = (live && i >= lowest && i >= 0) ? h0[(long long)i * plane] : 0.f;
            }
        }
        float e[S];
#pragma unroll
        for (int k = 0; k < S; ++k) e[k] = 0.f;
        groupChain<L, S, true>(edc, q, sub, e);
        float ax[S], ay[S];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const int i = iTop - k;
            const bool regress = i >= startingPoint && i < endPoint;
            const float y = 10.f * pvLog10fNonNegT(regress ? e[k] : 1.f, ltab);
            ax[k] = regress ? y * (float)(i - startingPoint) : 0.f;
            ay[k] = regress ? y : 0.f;
        }
        float unused[S];
        groupChain<L, S, false>(xysum, ax, sub, unused);
        groupChain<L, S, false>(ysum, ay, sub, unused);
    }

    // ---- wet gain (Analyzer.cpp:235-247): the same chain, forwards over [startingPoint, startingPoint + N_wet) ^ [0, T) ----
    const int wetEnd = min(startingPoint + a.nWet, T);
    int nw = max(wetEnd - startingPoint, 0);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) nw = max(nw, __shfl_xor(nw, off));
    float wet = 0.f;
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const int j = startingPoint + sub * S + k;
        pNext[k] = (live && j < wetEnd) ? h0[(long long)j * plane] : 0.f;
    }
#pragma unroll 1
    for (int j0 = 0; j0 < nw; j0 += L * S) {
        float q[S];
#pragma unroll
        for (int k = 0; k < S; ++k) q[k] = pNext[k] * pNext[k];
        if (j0 + L * S < nw) {
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const int j = startingPoint + j0 + L * S + sub * S + k;
                pNext[k] = (live && j < wetEnd) ? h0[(long long)j * plane] : 0.f;
            }
        }
        float unused[S];
        groupChain<L, S, false>(wet, q, sub, unused);
    }
    if (live && sub == L - 1) {
        a.out[a.resN + c.s] = sqrtf(wet / a.efree);
        a.out[2 * a.resN + c.s] = rt60FromSums(a, c.startingPoint, xysum, ysum);
    }
}

}  // namespace

// Live module with two iterations in flight on two solvers (Solver::run's carryFrom): a cell in which this run found no onset
// keeps what the PREVIOUS iteration left there (the reference never touches m_results[s] then, Analyzer.cpp:160-165) -- the
// previous iteration ran on the other solver, so its six persistent planes are copied over for exactly those cells
namespace {
__global__ __launch_bounds__(256) void pv_carry_results_kernel(const AnalyzeArgs a, const float* __restrict__ src) {
    const DynParams dyn = *a.dyn;
    const int wc = blockIdx.x * blockDim.x + threadIdx.x, wr = blockIdx.y;
    if (wc >= a.winCols) return;
    const int X = dyn.histRow0 - a.G + wr, Y = dyn.histCol0 - a.G + wc;
    if (X >= a.gx || Y >= a.gy) return;
    const long long s = (long long)X * a.gy + Y;
    if (a.delay[s] != FLT_MAX) return;
    a.out[s] = src[s];
    a.out[a.resN + s] = src[a.resN + s];
    a.out[2 * a.resN + s] = src[2 * a.resN + s];
    a.out[3 * a.resN + s] = src[3 * a.resN + s];
    a.out[6 * a.resN + s] = src[6 * a.resN + s];
    a.out[7 * a.resN + s] = src[7 * a.resN + s];
}
}  // namespace

void launchCarryResults(const AnalyzeArgs& a, const float* srcOut, hipStream_t stream) {
    hipLaunchKernelGGL(pv_carry_results_kernel, dim3((a.winCols + 255) / 256, a.winRows), dim3(256), 0, stream, a, srcOut);
}

// the blocked form; the sixteen-lane form (pv_rt60_wave_kernel, pv_kernels.hip) is launched beside them by launchAnalysis
void launchRt60Blocked(const AnalyzeArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL((pv_rt60_blocked_kernel<4, 4>), dim3((a.winCols + 63) / 64, a.winRows), dim3(256), 0, stream, a);
    // (one lane per cell -- <1, 4>, <1, 8> -- was measured too: slower than four lanes at every size, 0.27 vs 0.15 ms at 127^2,
    // 1.93 vs 1.70 ms at 512^2 / T = 3179; <4, 8> and <4, 2> are within 3 % of <4, 4>: profiles/r04_rt60.txt)
}

}  // namespace pva
