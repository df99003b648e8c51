// pv_prims.h -- device primitives shared by the kernel translation units (pv_kernels.hip, pv_resident.hip): lane shifts
// by DPP, buffer (SRSRC) addressing, the tile-major history offset.  Moved here unchanged from pv_kernels.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pva {

#ifndef PV_USE_DPP
#define PV_USE_DPP 1
#endif

typedef float v2f __attribute__((ext_vector_type(2)));

// The s_nop in front of the inline-asm DPP subtracts (subLanePrev2).  The product build's kernels never need it -- the pressure
// rows a vy sweep shifts were written a whole vx sweep earlier -- and tools/check_dpp_hazard.py checks that for the build at hand:
// the Makefile runs it on the assembly of the very flags the objects are built with, along every path into every such
// instruction, and a hazard fails the build (tests/test_host_cpu.py runs it too); 273 s_nop fewer per air tile are worth ~3 % at
// 4096^2 (profiles/r04_load_order.txt).  The experimental build keeps it: three of its instantiations do read a register written
// one wait state earlier.
#ifndef PV_DPP_ASM_NOP
#ifdef PV_EXPERIMENTAL
#define PV_DPP_ASM_NOP "s_nop 1\n\t"
#else
#define PV_DPP_ASM_NOP ""
#endif
#endif

// The pressure history is TILE-MAJOR: plane[t][window tile][row in tile][column in tile], RXI x WI floats per tile, no
// padding.  A step kernel records a tile's RXI x WI block of one sub-step as ONE contiguous chunk (5.76 KB for the 36 x 40
// tile) instead of RXI segments of 160 B that sit a whole plane row apart: 5.2 instead of 3.2 TB/s of history writes on
// MI355X (tools/hist_write_probe.hip), which is what the dense-history and the sparse-emitter (ring) modes are bound by.
// Offset (floats) of window cell (hr, hc) -- window row / column, both >= 0 -- inside one plane:
__device__ __forceinline__ long long histOffset(int hr, int hc, int rxi, int wi, int tilesY) {
    const int ti = hr / rxi, tj = hc / wi;
    return ((long long)(ti * tilesY + tj) * rxi + (hr - ti * rxi)) * wi + (hc - tj * wi);
}

// value held by lane+1 (lane 63 receives an unspecified value; it is always a halo lane)
__device__ __forceinline__ float laneNext(float v) {
#if PV_USE_DPP
    // DPP wave_shl:1 -- dst[i] = src[i+1] across the whole 64-lane wavefront (gfx9 DPP_WF_SL1 = 0x130)
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
#else
    return __shfl_down(v, 1);
#endif
}

// value held by lane-1 (lane 0 receives an unspecified value)
__device__ __forceinline__ float lanePrev(float v) {
#if PV_USE_DPP
    // DPP wave_shr:1 -- dst[i] = src[i-1] (DPP_WF_SR1 = 0x138)
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
#else
    return __shfl_up(v, 1);
#endif
}

// t = v - (value held by lane-1) for two row pairs (4 floats) in 4 instructions: v_subrev_f32_dpp computes
// src1 - dpp(src0), the lane shift rides on the subtract.  The compiler folds laneNext(v) - v like this by itself
// but leaves this operand order as shift + subtract, hence the asm.  gfx9-family ISAs need 2 wait states between a
// VALU write of a VGPR and a DPP read of it, and the compiler's hazard recogniser cannot see into inline asm: the
// leading PV_DPP_ASM_NOP (above) covers the inputs where they need it, and the outputs are early-clobber so they never alias a
// later input.
__device__ __forceinline__ void subLanePrev2(const v2f a, const v2f b, v2f& ta, v2f& tb) {
#if PV_USE_DPP
    float t0, t1, t2, t3;
    asm(PV_DPP_ASM_NOP
        "v_subrev_f32_dpp %0, %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_subrev_f32_dpp %1, %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_subrev_f32_dpp %2, %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_subrev_f32_dpp %3, %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));
    ta = v2f{t0, t1};
    tb = v2f{t2, t3};
#else
    ta = v2f{a.x - lanePrev(a.x), a.y - lanePrev(a.y)};
    tb = v2f{b.x - lanePrev(b.x), b.y - lanePrev(b.y)};
#endif
}

// Buffer (SRSRC) addressing: every plane is reached through a 128-bit descriptor built from kernel arguments;
// the per-lane part of an address is the constant lane*4 in voffset and everything wave-uniform (tile origin,
// row) goes into the scalar soffset, so the 3*ROWS loads / stores of a tile cost no address VGPRs.
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t makeRsrc(const void* p, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bufLoadF(rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bufStoreF(float v, rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
// with a cache-policy word (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX>
__device__ __forceinline__ float bufLoadFA(rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX));
}
template <int AUX>
__device__ __forceinline__ void bufStoreFA(float v, rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, AUX);
}

}  // namespace pva
