/*
  bwtm_api.hip -- implementation of the C ABI declared in include/bwtm.h: host-side
  orchestration of the gfx950 kernels in bwtm_kernels.hip.h.  No CPU fallback exists: every
  entry point fails with BWTM_ENODEV when no HIP device is usable.

    api/context.hip.h   contexts (device, streams, memory pool), errors, launch macros, profiling, scans
    api/index.hip.h     device index: pipelined upload + transcode, canonical encoder + pipelined download, queries
    api/search.hip.h    rank array: frontier search / per-chain walk, finalize, downloads
    api/fslice.hip.h    one GPU's state of the sliced frontier search (only with -DBWTM_EXPERIMENTAL; include/bwtm_experimental.h)
    api/partition.hip.h the merge over partitioned records: windows of indexes and rank arrays, fixed cuts, routed node phase (same build only)
    api/merge.hip.h     interleave, whole-path entry points (device-resident, consuming, host-to-host)
    api/slices.hip.h    output-range-sharded interleave + encode (one slice per GPU)
*/
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bwtm.h"
#include "bwtm_kernels.hip.h"

using namespace bwtm;

#include "api/context.hip.h"
#include "api/index.hip.h"
#include "api/search.hip.h"
#ifdef BWTM_EXPERIMENTAL
#include "../../include/bwtm_experimental.h"
#include "api/fslice.hip.h"
#include "api/partition.hip.h"
#endif
#include "api/merge.hip.h"
#include "api/slices.hip.h"
#include "api/ingest.hip.h"
