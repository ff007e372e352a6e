/*
  bwtm_api.hip -- implementation of the C ABI declared in include/bwtm.h: host-side
  orchestration of the gfx950 kernels in bwtm_kernels.hip.h.  No CPU fallback exists: every
  entry point fails with BWTM_ENODEV when no HIP device is usable.

    api/context.hip.h   contexts (device, streams, memory pool), errors, launch macros, profiling, scans
    api/index.hip.h     device index: pipelined upload + transcode, canonical encoder + pipelined download, queries
    api/search.hip.h    rank array: frontier search / per-chain walk, finalize, downloads
    api/merge.hip.h     interleave, whole-path entry points (device-resident, consuming, host-to-host)
    api/slices.hip.h    output-range-sharded interleave + encode (one slice per GPU)
    api/group.hip.h     the parts of a multi-GPU merge: shared control block (barrier, small all-gathers), exported arenas (raw pointer / HIP IPC)
    api/pmerge.hip.h    the merge over PARTITIONED records, one part per GPU: windows from byte shares, cuts, the routed search, the second half
    api/fslice.hip.h    one GPU's state of the sliced frontier search (only with -DBWTM_EXPERIMENTAL; include/bwtm_experimental.h)
    api/partition.hip.h the first, host-driven form of the partitioned search (same build only; kept for its tests and A/B measurements)
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
#include "api/merge.hip.h"
#include "api/slices.hip.h"
#include "api/group.hip.h"
#include "api/pmerge.hip.h"
#ifdef BWTM_EXPERIMENTAL
#include "../../include/bwtm_experimental.h"
#include "api/fslice.hip.h"
#include "api/partition.hip.h"
#endif
#include "api/ingest.hip.h"
