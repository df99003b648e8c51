// tools/valu_probe.hip -- what does one wave64 VALU instruction cost on gfx950?  (plain f32, packed f32, DPP forms;
// independent streams and dependent chains; 1 and 2 waves per SIMD).  The register-tile stencil's instruction mix is
// 9 v_pk_* : 4 v_sub_f32_dpp per cell pair and step; whether a DPP / plain op occupies the pipe for as long as a packed
// one decides what "VALU-bound" means for it.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long* cyc, float* sink, int iters, float seed) {
    float a[16];
    v2f p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + i + threadIdx.x;
        p[i] = v2f{seed + i, seed - i};
    }
    const float c = seed * 0.999f;
    const v2f c2 = v2f{c, c};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (KIND == 0) {  // plain v_mul_f32, 16 independent streams
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                REP16(X)
#undef X
            } else if (KIND == 1) {  // v_pk_mul_f32
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                REP16(X)
#undef X
            } else if (KIND == 2) {  // v_sub_f32_dpp wave_shr:1 (reads another stream's register: no RAW on itself)
#define X(i) asm volatile("v_sub_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 8) & 15]));
                REP16(X)
#undef X
            } else if (KIND == 3) {  // v_pk_add_f32
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                REP16(X)
#undef X
            } else if (KIND == 4) {  // v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                REP16(X)
#undef X
            } else if (KIND == 5) {  // v_pk_fma_f32
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
                REP16(X)
#undef X
            } else if (KIND == 6) {  // dependent chain of v_pk_mul_f32
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[0]) : "v"(c2));
                REP16(X)
#undef X
            } else if (KIND == 7) {  // dependent chain of v_mul_f32
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
                REP16(X)
#undef X
            } else if (KIND == 8) {  // v_mov_b32_dpp wave_shl:1
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 8) & 15]));
                REP16(X)
#undef X
            } else if (KIND == 9) {  // the stencil's mix: 9 packed : 4 dpp (here 12 : 4 -> close), interleaved
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                REP16(X)
#undef X
#define X(i) if ((i) < 7) asm volatile("v_sub_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 8) & 15]));
                REP16(X)
#undef X
            } else if (KIND == 10) {  // plain v_sub_f32 (no dpp), reading another stream
#define X(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 8) & 15]));
                REP16(X)
#undef X
            } else if (KIND == 11) {  // dependent chain: pk_mul -> pk_add alternating (the stencil's mul + sub)
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1" : "+v"(p[0]) : "v"(c2));
                REP16(X)
#undef X
            } else if (KIND == 13) {  // three-address v_pk_add_f32 with neg modifiers (the stencil's form), sources from two other streams
#define X(i) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(p[i]) : "v"(p[(i + 5) & 15]), "v"(p[(i + 10) & 15]));
                REP16(X)
#undef X
            } else if (KIND == 14) {  // v_pk_mul_f32 with an SGPR-pair source (the stencil's C * x)
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(p[i]) : "s"(c2), "v"(p[(i + 5) & 15]));
                REP16(X)
#undef X
            } else if (KIND == 15) {  // explicit registers: both source pairs in the same VGPR bank pair (v[100:101], v[104:105])
                asm volatile(
                    "v_pk_add_f32 v[108:109], v[100:101], v[104:105]\n\tv_pk_add_f32 v[112:113], v[100:101], v[104:105]\n\t"
                    "v_pk_add_f32 v[116:117], v[100:101], v[104:105]\n\tv_pk_add_f32 v[120:121], v[100:101], v[104:105]\n\t"
                    "v_pk_add_f32 v[108:109], v[100:101], v[104:105]\n\tv_pk_add_f32 v[112:113], v[100:101], v[104:105]\n\t"
                    "v_pk_add_f32 v[116:117], v[100:101], v[104:105]\n\tv_pk_add_f32 v[120:121], v[100:101], v[104:105]\n\t"
                    "v_pk_add_f32 v[108:109], v[100:101], v[104:105]\n\tv_pk_add_f32 v[112:113], v[100:101], v[104:105]\n\t"
                    "v_pk_add_f32 v[116:117], v[100:101], v[104:105]\n\tv_pk_add_f32 v[120:121], v[100:101], v[104:105]\n\t"
                    "v_pk_add_f32 v[108:109], v[100:101], v[104:105]\n\tv_pk_add_f32 v[112:113], v[100:101], v[104:105]\n\t"
                    "v_pk_add_f32 v[116:117], v[100:101], v[104:105]\n\tv_pk_add_f32 v[120:121], v[100:101], v[104:105]"
                    ::: "v100", "v101", "v104", "v105", "v108", "v109", "v112", "v113", "v116", "v117", "v120", "v121");
            } else if (KIND == 16) {  // explicit registers: source pairs in different banks (v[100:101], v[106:107])
                asm volatile(
                    "v_pk_add_f32 v[108:109], v[100:101], v[106:107]\n\tv_pk_add_f32 v[112:113], v[100:101], v[106:107]\n\t"
                    "v_pk_add_f32 v[116:117], v[100:101], v[106:107]\n\tv_pk_add_f32 v[120:121], v[100:101], v[106:107]\n\t"
                    "v_pk_add_f32 v[108:109], v[100:101], v[106:107]\n\tv_pk_add_f32 v[112:113], v[100:101], v[106:107]\n\t"
                    "v_pk_add_f32 v[116:117], v[100:101], v[106:107]\n\tv_pk_add_f32 v[120:121], v[100:101], v[106:107]\n\t"
                    "v_pk_add_f32 v[108:109], v[100:101], v[106:107]\n\tv_pk_add_f32 v[112:113], v[100:101], v[106:107]\n\t"
                    "v_pk_add_f32 v[116:117], v[100:101], v[106:107]\n\tv_pk_add_f32 v[120:121], v[100:101], v[106:107]\n\t"
                    "v_pk_add_f32 v[108:109], v[100:101], v[106:107]\n\tv_pk_add_f32 v[112:113], v[100:101], v[106:107]\n\t"
                    "v_pk_add_f32 v[116:117], v[100:101], v[106:107]\n\tv_pk_add_f32 v[120:121], v[100:101], v[106:107]"
                    ::: "v100", "v101", "v106", "v107", "v108", "v109", "v112", "v113", "v116", "v117", "v120", "v121");
            } else if (KIND == 12) {  // row_shr:1 dpp sub (row-local shift) for comparison with the wave shift
#define X(i) asm volatile("v_sub_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 8) & 15]));
                REP16(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int perIter) {
    unsigned long long* cyc;
    float* sink;
    hipMalloc(&cyc, 8 * 4096 * 4);
    hipMalloc(&sink, 64);
    const int iters = 2000;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float best = 1e9f;
        for (int r = 0; r < 4; ++r) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, cyc, sink, iters, 1.0001f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (r && ms < best) best = ms;
        }
        static unsigned long long h[4096 * 4];
        hipMemcpy(h, cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        for (int i = 0; i < blocks * 4; ++i) sum += (double)h[i];
        const double n = (double)iters * 4 * perIter;  // instructions per wave
        // s_memtime counts at a constant 100 MHz on gfx9 (readcyclecounter = s_memtime); wall time is what we trust:
        // instructions per SIMD = n * wps; cycles per instruction at 2.4 GHz nominal = ms * 2.4e6 / (n * wps)
        printf("%-44s wps %d: %.3f ms  -> %.2f cyc/instr/SIMD @2.4GHz (counter ticks per instr per wave %.3f)\n", name, wps,
               best, best * 2.4e6 / (n * wps), sum / (blocks * 4) / n);
    }
    hipFree(cyc);
    hipFree(sink);
}

int main() {
    run<0>("v_mul_f32 x16 independent", 16);
    run<10>("v_sub_f32 (two vgpr sources) x16", 16);
    run<4>("v_fma_f32 x16 independent", 16);
    run<1>("v_pk_mul_f32 x16 independent", 16);
    run<3>("v_pk_add_f32 x16 independent", 16);
    run<5>("v_pk_fma_f32 x16 independent", 16);
    run<2>("v_sub_f32_dpp wave_shr:1 x16", 16);
    run<12>("v_sub_f32_dpp row_shr:1 x16", 16);
    run<8>("v_mov_b32_dpp wave_shl:1 x16", 16);
    run<9>("mix 16 v_pk_mul + 7 v_sub_dpp", 23);
    run<7>("v_mul_f32 dependent chain", 16);
    run<6>("v_pk_mul_f32 dependent chain", 16);
    run<11>("pk_mul -> pk_add dependent chain", 32);
    run<13>("v_pk_add_f32 three-address, neg modifiers", 16);
    run<14>("v_pk_mul_f32 with SGPR-pair source", 16);
    run<15>("v_pk_add_f32 sources v[100:101], v[104:105] (same banks)", 16);
    run<16>("v_pk_add_f32 sources v[100:101], v[106:107] (other banks)", 16);
    return 0;
}
