// Probe (lab tool): LDS read THROUGHPUT of one CU with eight waves hammering it - full ds_read_b128 vs the same instruction with
// only a few lanes enabled (exec-masked).  Question behind it: can the 3x3 conv kernels read the centre fragment once, build the
// dx = -1 / +1 fragments with DPP row shifts and fetch only the two edge columns with an exec-masked read?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_tput_probe.hip -o build/lds_tput_probe && build/lds_tput_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(long long* cycles, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 65536 / 16; i += 512) reinterpret_cast<uintx4*>(lds)[i] = uintx4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    // conflict-free pattern of the halo kernels: pixel = lane & 31 (+ wave offset), slot swizzled on the column
    const int r = lane & 31, khalf = lane >> 5;
    int a[4];
    for (int ks = 0; ks < 4; ++ks) a[ks] = ((wid * 40 + 20 + (r & 15) + 18 * (r >> 4)) * 128 + (((2 * ks + khalf) ^ ((r >> 1) & 7)) << 4)) & 65535;
    const bool edge = (lane & 15) == 0 || (lane & 15) == 15;
    uintx4 acc[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (MODE == 0) {                                   // full read
                acc[ks] ^= *reinterpret_cast<const uintx4*>(lds + a[ks]);
            } else if (MODE == 1) {                            // 8 of 64 lanes
                if (edge) acc[ks] ^= *reinterpret_cast<const uintx4*>(lds + a[ks]);
            } else if (MODE == 2) {                            // full + masked: the DPP scheme's LDS traffic per three taps
                acc[ks] ^= *reinterpret_cast<const uintx4*>(lds + a[ks]);
                if (edge) acc[(ks + 1) & 3] ^= *reinterpret_cast<const uintx4*>(lds + ((a[ks] + 2304) & 65535));
            } else {                                           // three full reads: today's traffic per three taps
                acc[ks] ^= *reinterpret_cast<const uintx4*>(lds + a[ks]);
                acc[(ks + 1) & 3] ^= *reinterpret_cast<const uintx4*>(lds + ((a[ks] + 128) & 65535));
                acc[(ks + 2) & 3] ^= *reinterpret_cast<const uintx4*>(lds + ((a[ks] + 256) & 65535));
            }
        }
    }
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[0] = t1 - t0;
    sink[tid] = acc[0][0] ^ acc[1][1] ^ acc[2][2] ^ acc[3][3];
}

template <int MODE>
static void run(const char* name, int per_iter) {
    long long* cyc; unsigned* sink;
    hipMalloc(&cyc, 8); hipMalloc(&sink, 4096);
    long long best = 1LL << 60;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(512), 0, 0, cyc, sink);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        if (c < best) best = c;
    }
    printf("%-44s %8lld cycles for 8 waves x %d instruction groups -> %.1f cycles per group per wave-slot\n", name, best, 256, best / 256.0 / 8.0);
    hipFree(cyc); hipFree(sink);
}

int main() {
    run<0>("full ds_read_b128", 1);
    run<1>("exec-masked ds_read_b128 (8 of 64 lanes)", 1);
    run<2>("1 full + 1 masked (DPP scheme, 3 taps)", 2);
    run<3>("3 full (today, 3 taps)", 3);
    return 0;
}
