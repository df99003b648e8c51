// Runs ON THE GPU BOX beside a load (e.g. bench.py in the background): the shader clock the chip sustains, from
// s_memtime (shader clock cycles) against s_memrealtime (constant 100 MHz), sampled by one wave for ~20 us.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
__global__ void probe(unsigned long long* out) {
    const unsigned long long r0 = wall_clock64(), c0 = clock64();
    unsigned long long r1 = r0;
    while (r1 - r0 < 2000) r1 = wall_clock64();  // 2000 ticks of 100 MHz = 20 us
    const unsigned long long c1 = clock64();
    out[0] = c1 - c0;
    out[1] = r1 - r0;
    // second estimate, independent of s_memtime: 100 x s_sleep 127 = 100 x 127 x 64 shader clock cycles of sleep
    const unsigned long long s0 = wall_clock64();
    for (int i = 0; i < 100; ++i) __builtin_amdgcn_s_sleep(127);
    out[2] = wall_clock64() - s0;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 20;
    unsigned long long* d;
    hipHostMalloc((void**)&d, 32);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        printf("shader clock %.0f MHz by s_memtime (%llu cycles in %llu x 10 ns); %.0f MHz by s_sleep (812800 cycles in %llu x 10 ns)\n",
               (double)d[0] / (double)d[1] * 100.0, d[0], d[1], 812800.0 / (double)d[2] * 100.0, d[2]);
        usleep(300000);
    }
    return 0;
}
