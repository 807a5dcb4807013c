// Probe (lab tool, not part of the library): cycles per ds_read_b128 / ds_write_b128 for the LDS address patterns of the bf16 conv
// kernels - which of them bank-conflict?  One wave, 256 back-to-back accesses per pattern, s_memtime around them.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_bank_probe.hip -o gpurun_out/lds_bank_probe && gpurun_out/lds_bank_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

// pattern -> byte address of this lane's 16-byte access for k-step ks
__device__ int addr_of(int pat, int lane, int ks, int base) {
    const int khalf = lane >> 5, r = lane & 31;
    if (pat == 0) {                       // linear-tile kernel: row = r, slot = octet ^ ((row >> 1) & 7)
        const int row = base + r;
        return row * 128 + (((2 * ks + khalf) ^ ((row >> 1) & 7)) << 4);
    }
    if (pat == 1 || pat == 2) {           // halo kernel: two image rows of 16 pixels, 18 halo pixels per row; base even (1) / odd (2)
        const int hp = base + (r & 15) + 18 * (r >> 4) + (pat == 2 ? 1 : 0);
        return hp * 128 + (((2 * ks + khalf) ^ ((hp >> 1) & 7)) << 4);
    }
    if (pat == 3) return lane * 16 + ks * 1024;                  // lane-linear (the register-staged halo store / the DMA image)
    if (pat == 4) {                       // no swizzle at all: 32 rows x 128 B, same octet -> 16-way conflict expected
        return (base + r) * 128 + ((2 * ks + khalf) << 4);
    }
    // pat 5: halo pattern with the swizzle keyed on the pixel's position inside its 16-pixel row instead of hp (candidate fix)
    const int hp = base + (r & 15) + 18 * (r >> 4);
    return hp * 128 + (((2 * ks + khalf) ^ ((r >> 1) & 7)) << 4);
}

__global__ void probe(long long* cycles, unsigned* sink, int pat, int write, int base) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x;
    for (int i = lane; i < 65536 / 16; i += 64) reinterpret_cast<uintx4*>(lds)[i] = uintx4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    int a[4];
    for (int ks = 0; ks < 4; ++ks) a[ks] = addr_of(pat, lane, ks, base);
    uintx4 acc = {0u, 0u, 0u, 0u};
    __builtin_amdgcn_s_waitcnt(0);
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (write) *reinterpret_cast<uintx4*>(lds + a[ks]) = acc;
            else { const uintx4 v = *reinterpret_cast<const uintx4*>(lds + a[ks]); acc += v; }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[0] = t1 - t0;
    sink[lane] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    long long* cyc; unsigned* sink;
    hipMalloc(&cyc, 8); hipMalloc(&sink, 256);
    const char* names[6] = {"linear tile (row swizzle)", "halo, even base", "halo, odd base", "lane-linear", "no swizzle", "halo, swizzle on x"};
    for (int write = 0; write < 2; ++write)
        for (int pat = 0; pat < 6; ++pat)
            for (int base = 0; base < (pat == 1 || pat == 5 ? 3 : 1); ++base) {
                long long best = 1LL << 60;
                for (int rep = 0; rep < 5; ++rep) {
                    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, cyc, sink, pat, write, 20 + 20 * base);
                    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
                    if (c < best) best = c;
                }
                printf("%s %-28s base %2d: %6.1f cycles per b128 access (256 accesses)\n", write ? "write" : "read ", names[pat], 20 + 20 * base, best / 256.0);
            }
    return 0;
}
