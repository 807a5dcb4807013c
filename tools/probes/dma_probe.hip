// Probe (lab tool, not part of the library): what does an LDS-DMA buffer load write for an out-of-range lane?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/dma_probe.hip -o gpurun_out/dma_probe && gpurun_out/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* src, unsigned bytes, float* out, int mode) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 2];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 4 * 2; i += 64) lds[i] = -7.0f;           // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)bytes, 0x00020000);
    unsigned voff = (unsigned)lane * 16u;
    if (lane % 4 == 1) voff = 0xC0000000u;                               // out of range: expect zeros (or an untouched sentinel?)
    if (mode == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, 0, 0, 0);
    } else {
        // exec-masked lanes: are they skipped?
        if (lane % 4 != 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    __syncthreads();
    for (int i = lane; i < 64 * 4; i += 64) out[i] = lds[i];
}

__global__ void swap_probe(int* out) {
    int a = threadIdx.x, b = 100 + threadIdx.x;
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
#else
    out[threadIdx.x] = -1; out[64 + threadIdx.x] = -1;
#endif
}

int main() {
    float *src, *out; int* iout;
    hipMalloc(&src, 4096); hipMalloc(&out, 64 * 4 * 4); hipMalloc(&iout, 128 * 4);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)(i + 1);
    hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, 4096u, out, mode);
        float o[256]; hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
        printf("mode %d (%s): lane0 %g %g | lane1(OOB) %g %g %g %g | lane2 %g | lane5(OOB) %g\n", mode, mode ? "exec-masked" : "OOB voffset",
               o[0], o[1], o[4], o[5], o[6], o[7], o[8], o[20]);
    }
    hipLaunchKernelGGL(swap_probe, dim3(1), dim3(64), 0, 0, iout);
    int io[128]; hipMemcpy(io, iout, sizeof(io), hipMemcpyDeviceToHost);
    printf("permlane32_swap(a=lane, b=100+lane): r0[0]=%d r0[32]=%d r1[0]=%d r1[32]=%d\n", io[0], io[32], io[64], io[96]);
    return 0;
}
