"""MI355X-native rank-array / interleave path of bwt-merge.

The product is the C-ABI library libbwtm.so (csrc/, include/bwtm.h) plus the C++ facade and
the bwt_merge CLI in csrc/host; this Python package only binds the C ABI for tests and bench.
"""
from . import build as _build          # noqa: F401
from . import capi                      # noqa: F401
from .capi import (Builder, BwtmError, Context, Group, HostBuffer, Index, Part, host_index, partition_cuts_host, window_blocks, RankArray, Slice, device_bytes_peak, fold_offsets, merged_records, slice_bounds, slice_bounds_equal, init, interleave,  # noqa: F401
                   make_default_current, merge, merge_consume, merge_host, merge_host_pipelined, upload_begin, RESULT_ON_DEVICE, pool_stats, profile_enable, profile_only, profile_read, profile_reset,
                   ra_buffer_bytes, synchronize, trim, tune)

# (the bindings of include/bwtm_experimental.h live in bwt_merge_amd.experimental, importable only next to libbwtm_experimental.so)


def build(force=False, verbose=False, experimental=False):
    return _build.build(force=force, verbose=verbose, experimental=experimental)
