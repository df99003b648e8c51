// tools/f64_probe.hip -- what the instructions of the glibc-exact log10f cost on gfx950 (round 5): cycles per wave64 instruction
// for 8 independent streams, one wave per SIMD and two.  The decay-time kernel evaluates per sample 8 double-precision
// operations (cvt, fma x 4, mul, add, cvt) and ~28 single-precision / integer ones.
//   hipcc --offload-arch=gfx950 -O3 tools/f64_probe.hip -o /tmp/f64_probe && /tmp/f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long* cyc, double* sink, int iters, double seed) {
    double d[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        d[i] = seed + i + threadIdx.x;
        f[i] = (float)d[i];
    }
    const double c = seed * 0.999;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            if (KIND == 0) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 1) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 2) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 3) {
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
                REP8(X)
#undef X
            } else if (KIND == 4) {
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
                REP8(X)
#undef X
            } else if (KIND == 5) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"((float)c));
                REP8(X)
#undef X
            } else if (KIND == 6) {
#define X(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[i]) : "v"(__float_as_int(f[(i + 1) & 7])));
                REP8(X)
#undef X
            } else if (KIND == 7) {  // dependent chain of v_fma_f64
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[0]) : "v"(c));
                REP8(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += d[i] + f[i];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (s == 1.2345) sink[0] = s;
}

template <int KIND>
static void run(const char* name) {
    unsigned long long* cyc; double* sink;
    hipMalloc(&cyc, 8 * 4096); hipMalloc(&sink, 8);
    const int iters = 2000;
    for (int wavesPerSimd : {1, 2, 4}) {
        const int blocks = 256 * wavesPerSimd;  // 4 waves per block = one per SIMD of a CU
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, cyc, sink, iters, 1.0001);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, cyc, sink, iters, 1.0001);
        hipDeviceSynchronize();
        unsigned long long h[2048];
        hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
        double avg = 0;
        for (int i = 0; i < blocks; ++i) avg += (double)h[i];
        avg /= blocks;
        // s_memtime counts at 100 MHz; report shader cycles per instruction and wave assuming 2.4 GHz
        std::printf("%-18s %d wave(s)/SIMD: %7.2f cycles per instruction and wave (x %d waves = %.2f per SIMD slot)\n", name, wavesPerSimd,
                    avg * 24.0 / (iters * 64.0), wavesPerSimd, avg * 24.0 / (iters * 64.0) / wavesPerSimd);
    }
}

int main() {
    run<5>("v_fma_f32");
    run<0>("v_fma_f64");
    run<1>("v_mul_f64");
    run<2>("v_add_f64");
    run<3>("v_cvt_f64_f32");
    run<4>("v_cvt_f32_f64");
    run<6>("v_cvt_f32_i32");
    run<7>("v_fma_f64 chain");
    return 0;
}
