// ABI bookkeeping for liblwg_hip.so (the kernels' extern "C" entry points live next to the kernels).
#include "lwg_common.h"
#include "lwg_conv_args.h"
#include "lwg_conv_slices.h"

extern "C" int lwg_abi_version(void) { return LWG_ABI_VERSION; }

// Number of compute units of the current device (256 on MI355X) - lets the host size split factors.
extern "C" int lwg_device_cu_count(void) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return cus;
}

// How many kernel launches the convolution entry points make for this description: 1, or the number of batch slices when a gathered input
// exceeds the kernels' 32-bit buffer range (lwg_conv_slices.h); 0 = a single frame does not fit (the entry points reject it).  For callers
// that account launches (bench.py brackets entry-point calls with events and reports per-kernel-launch averages).
extern "C" int lwg_conv_slice_count(const LwgConvArgs* pa) {
    if (!pa || pa->B <= 0) return 0;
    const int nbs = lwg_conv_slice_frames(*pa);
    if (nbs == 0) return 1;
    if (nbs < 0) return 0;
    return (pa->B + nbs - 1) / nbs;
}
