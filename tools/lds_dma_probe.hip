// tools/lds_dma_probe.hip -- what does `buffer_load_dword{,x3,x4} ... offen lds` write where on gfx950?  (development aid for
// csrc/pv_patch.h)    hipcc --offload-arch=gfx950 -O2 tools/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

using rsrc_t = __amdgpu_buffer_rsrc_t;

template <int W>
__device__ __forceinline__ void dma(rsrc_t r, int voff, int soff, unsigned lds) {
    unsigned keep;
    if constexpr (W == 1)
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(r), "s"(soff), "s"(lds) : "memory");
    else if constexpr (W == 3)
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx3 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(r), "s"(soff), "s"(lds) : "memory");
    else
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(r), "s"(soff), "s"(lds) : "memory");
}

// one wave; zone of 36000 floats; DMA at LDS byte offset `ldsOff`; dump 512 floats from there
template <int W>
__global__ void probe(const float* src, int bytes, int soff, int ldsOff, float* out) {
    __shared__ float zone[36000];
    for (int i = threadIdx.x; i < 36000; i += 64) zone[i] = -1.f;
    __syncthreads();
    const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
    const unsigned base = (unsigned)(size_t)zone + (unsigned)ldsOff;
    dma<W>(r, (int)threadIdx.x * 4 * W, soff, base);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = zone[ldsOff / 4 + i];
}

int main() {
    const int N = 1 << 16;
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, N * 4);
    hipMalloc(&o, 512 * 4);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<float> r(512);
    auto show = [&](const char* tag) {
        hipDeviceSynchronize();
        hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
        std::printf("%s\n  ", tag);
        for (int i = 0; i < 260; ++i) std::printf("%g%s", r[i], (i % 20 == 19) ? "\n  " : " ");
        std::printf("\n");
    };
    hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, d, N * 4, 4000, 1024, o);
    show("dword, soff 4000 B (element 1000), LDS offset 1024");
    hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, d, N * 4, 4000, 1024, o);
    show("dwordx3, voff = lane*12");
    hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, d, N * 4, 4000, 1024, o);
    show("dwordx4, voff = lane*16");
    hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, d, N * 4, 4000, 100 * 1024, o);
    show("dwordx3 at LDS offset 100 KiB");
    hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, d, 0, 4000, 1024, o);
    show("dwordx3, zero-extent descriptor (out of range)");
    hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, d, N * 4, 4004, 1024, o);
    show("dwordx4, source 4-byte aligned only (soff 4004)");
    return 0;
}
