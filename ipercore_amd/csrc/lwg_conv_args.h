// The kernel-side view of the C ABI structs is the ABI header itself.
#pragma once
#include "../../include/lwg_hip.h"
