// josefine_gpu.hip — C ABI (include/josefine_gpu.h) of the MI355X batched
// Chained-Raft engine: host-side marshalling around the gfx950 kernels in
// jg_kernels.h.  There is deliberately no CPU implementation in this library:
// every entry point that computes does so on the device or fails with
// JG_EDEVICE.  What the host does do: validate arguments, bucket command rows by
// group (a stable radix sort of row indices — marshalling, not Raft), move bytes,
// and launch.
#include <cstring>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "jg_kernels.h"
#include "jg_route.h"
#include "jg_follower.h"
#include "jg_node.h"

// The host side, by entry-point family (ONE translation unit: the pieces see each other's internals, in this order)
#include "jg_api_core.h"
#include "jg_api_drains.h"
#include "jg_api_engine.h"
#include "jg_api_node.h"
#include "jg_api_cluster.h"
#include "jg_api_routed.h"
#include "jg_api_misc.h"
