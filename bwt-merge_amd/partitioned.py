"""Drivers of the merge over PARTITIONED records (include/bwtm.h: bwtm_group_*, bwtm_part_*).

The library runs the whole merge of one part -- windows from byte shares, the routed search in lock step with the other parts, the
second half -- in four calls; what is left for a host is to say who the parts are:

  merge_part         one part, in the calling thread / process (bench.py --gpus N --search partitioned: one process per GPU)
  merge_parts        `parts` threads of THIS process, one library context each (contexts of one GPU stand in for GPUs on a one-GPU box;
                     the C++ host does the same with one thread per device: csrc/host/multi_gpu.h)
  unique_group_name  a shared-memory name for a group

Inputs are host-resident like the reference's loaded FMI: native bytes + the cumulative sample arrays (capi.host_index).
"""
import os
import threading

import numpy as np

from . import capi

_counter = [0]


def unique_group_name(tag=""):
    _counter[0] += 1
    return "/bwtm-%d-%d%s" % (os.getpid(), _counter[0], ("-" + str(tag)) if tag else "")


def merge_part(group, a, b, cuts=None, kmer=0, shares=None, profile=None):
    """One part's merge.  a, b: capi.host_index objects; cuts = (cut_a, cut_b) or None (computed here: every part gets the same answer);
    shares = ((ptr, nbytes, first_position, counts_before), (...)) of device-resident byte shares (bench.py's HBM-resident inputs) instead of
    uploads from the host arrays.  Returns (Slice, stats dict)."""
    if cuts is None:
        cuts = capi.partition_cuts_host(a, b, group.parts, kmer)
    P = capi.Part(group, a, b, cuts[0], cuts[1])
    try:
        for which, x in ((0, a), (1, b)):
            if shares is not None:
                ptr, nbytes, fp, before = shares[which]
                P.upload(which, ptr, nbytes, fp, before, on_device=True)
            else:
                P.upload_host(which, x)
        if profile:
            profile("transcode")
        P.search()
        if profile:
            profile("search")
        S = P.finish()
        if profile:
            profile("finish")
        stats = P.stats()
    except Exception:
        group.abort()
        raise
    finally:
        P.free()
    return S, stats


def merge_parts(pkg, a, b, parts, device=0, kmer=0, cuts=None, profile=False, collect=None):
    """The whole merge with `parts` threads of this process, one context of `device` each.  Returns a dict: slices (in order; their contexts
    stay alive until release() is called), stats, cuts, and with profile = True the kernel milliseconds of every part by phase and kernel (phases[g][phase][kernel]).
    collect(g, slice) (optional) runs in part g's thread after its merge (download its bytes / samples there)."""
    if cuts is None:
        cuts = capi.partition_cuts_host(a, b, parts, kmer)
    name = unique_group_name() if parts > 1 else None
    ctxs = [pkg.Context(device) for _ in range(parts)]
    slices, stats, errors, phases, collected = [None] * parts, [None] * parts, [None] * parts, [dict() for _ in range(parts)], [None] * parts

    def worker(g):
        group = None
        try:
            ctxs[g].make_current()
            group = capi.Group(name, g, parts)

            def prof(phase):
                pkg.synchronize()
                phases[g][phase] = {name: v[0] for name, v in pkg.profile_read().items()}      # kernel -> ms of this phase
                pkg.profile_reset()
            if profile:
                pkg.profile_only(None); pkg.profile_reset(); pkg.profile_enable(True)
            slices[g], stats[g] = merge_part(group, a, b, cuts=cuts, profile=(prof if profile else None))
            if profile:
                pkg.profile_enable(False)
            if collect is not None:
                collected[g] = collect(g, slices[g])
        except Exception as e:                       # noqa: BLE001 -- reported to the caller below
            errors[g] = e
            if group is not None:
                group.abort()
        finally:
            if group is not None:
                group.free()

    threads = [threading.Thread(target=worker, args=(g,)) for g in range(parts)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()

    def release():
        for g in range(parts):
            ctxs[g].make_current()
            if slices[g] is not None:
                slices[g].free()
        pkg.make_default_current()
        for c in ctxs:
            c.destroy()

    bad = [e for e in errors if e is not None]
    if bad:
        release()
        own = [e for e in bad if "bwtm error 5" not in str(e)]          # a part's own failure says more than its peers' BWTM_EPEER
        raise (own[0] if own else bad[0])
    return dict(slices=slices, stats=stats, cuts=cuts, phases=phases, collected=collected, release=release, contexts=ctxs)


def slice_arrays(s):
    """(data, block_end, cum) of an encoded part's slice (call in the thread whose context the slice lives in)."""
    data = s.data()
    be, cum = s.samples(s.next_block_start)
    return data, be, cum
