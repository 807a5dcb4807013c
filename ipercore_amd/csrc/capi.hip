// ABI bookkeeping for liblwg_hip.so (the kernels' extern "C" entry points live next to the kernels).
#include "lwg_common.h"
#include "lwg_conv_args.h"

extern "C" int lwg_abi_version(void) { return LWG_ABI_VERSION; }

// Number of compute units of the current device (256 on MI355X) - lets the host size split factors.
extern "C" int lwg_device_cu_count(void) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return cus;
}
