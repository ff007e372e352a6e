"""The per-step exchange of the partitioned search between PROCESSES (one per GPU, torch.distributed: RCCL on GPUs, gloo on CPU) -- the
multi-process form of experimental.search_partitioned, which drives contexts of one GPU from one thread.

After a step every rank holds its outputs as ONE dense array in logical order: class after class, increasing position inside a class, and
knows per class how many of them lie below every cut (bwtm_fslice_view.below).  Class c's range of that array is therefore already ordered
by destination, and what a rank receives for class c, source after source, is increasing position again: the exchange is one
all_to_all_single per class on the send buffer as it is, and the concatenation of the five results is the rank's next input.

  exchange_plan         the split sizes of those calls from the all-gathered counts
  all_to_all_classes    the five calls
  search_partitioned_dist   the element steps of the search on this rank's windows (needs libbwtm_experimental.so and a GPU)
The plan and the calls are plain tensor code: tests/test_partition_dist_host.py runs them over gloo with 2 and 3 ranks on the CPU.
"""


def exchange_plan(below_all, rank, world):
    """below_all[h][c][k] = elements of rank h's class c below cut k (k = 0 .. world; below[..][0] = 0, below[..][world] = all of the class).
    Returns (send[c][k] = this rank's elements of class c that go to rank k, recv[c][h] = elements of class c it gets from rank h)."""
    send = [[int(below_all[rank][c][k + 1]) - int(below_all[rank][c][k]) for k in range(world)] for c in range(5)]
    recv = [[int(below_all[h][c][rank + 1]) - int(below_all[h][c][rank]) for h in range(world)] for c in range(5)]
    return send, recv


def all_to_all_classes(dist, send_buf, class_first, send, recv, recv_buf, units=1):
    """Five all_to_all_single calls, one per class: class c's elements leave from send_buf[class_first[c] ...] (ordered by destination) and
    arrive behind the classes before it, source after source.  units = tensor items per element (the 2-byte high parts travel as byte
    pairs: RCCL has no 16-bit integer type).  Returns the number of elements received."""
    off = 0
    for c in range(5):
        n_send, n_recv = sum(send[c]), sum(recv[c])
        first = int(class_first[c])
        dist.all_to_all_single(recv_buf[units * off: units * (off + n_recv)], send_buf[units * first: units * (first + n_send)],
                               output_split_sizes=[units * x for x in recv[c]], input_split_sizes=[units * x for x in send[c]])
        off += n_recv
    return off


class _DeviceArray:
    """A device pointer of the library as something torch.as_tensor takes without a copy."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_tensor(torch, device, ptr, count, typestr):
    return torch.as_tensor(_DeviceArray(ptr, count, typestr), device=device)


def _gather_counts(dist, torch, device, below, world):
    mine = torch.tensor([[int(below[c][k]) for k in range(world + 1)] for c in range(5)], dtype=torch.int64, device=device)
    everyone = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    return [t.cpu().numpy() for t in everyone]


def search_partitioned_dist(pkg, wa, wb, ra, sequences, r_cuts, rank, world, dist, torch, device, capacity=None, node_ratio=8, node_capacity=None):
    """The partitioned search on this rank (GPU `device`): wa / wb = its windows, ra = its rank array, r_cuts = the B ranks of the cuts
    (world + 1 of them).  The first levels run on trie nodes (while a level has at most sequences / node_ratio of them; 0: elements from the
    roots on) when all roots lie on one rank -- cuts at k-mer boundaries: the first --, the children exchanged like the elements, with three
    arrays.  Returns (element steps, node levels)."""
    from . import experimental as X
    cap = int(capacity) if capacity else int(sequences) + 1
    fs = X.FSlice(wa, wb, ra, cap, world)
    fs.set_cuts(r_cuts)
    first, last = min(int(r_cuts[rank]), sequences), min(int(r_cuts[rank + 1]), sequences)
    limit = sequences // node_ratio if node_ratio > 0 else 0
    owners = torch.tensor([1 if last > first else 0], dtype=torch.int64, device=device)
    dist.all_reduce(owners)
    levels = 0
    view = X.FSliceView()
    if limit >= 1 and int(owners.item()) == 1:
        ncap = int(node_capacity) if node_capacity else min(5 * limit, sequences) + 1
        fs.nodes_begin(first, last - first, ncap)
        nview = X.FSliceNodesView()
        sp_in, r_in, cnt_in, _ = fs.nodes_input_buffers()
        recv = [device_tensor(torch, device, p, ncap, "<i8") for p in (sp_in, r_in, cnt_in)]
        level_nodes = 1
        while 0 < level_nodes <= limit:
            fs.nodes_step(nview)                                     # synchronizes: the children are complete
            below_all = _gather_counts(dist, torch, device, nview.below, world)
            send, rcv = exchange_plan(below_all, rank, world)
            n_in = sum(sum(x) for x in rcv)
            if n_in > ncap:
                raise X.BwtmError("%d nodes fall into rank %d's range, capacity %d" % (n_in, rank, ncap))
            held = int(nview.class_first[5])
            class_first = [int(nview.class_first[c]) for c in range(6)]
            for src, dst in zip((nview.sp, nview.r, nview.count), recv):
                all_to_all_classes(dist, device_tensor(torch, device, src, max(held, 1), "<i8"), class_first, send, rcv, dst)
            torch.cuda.synchronize(device)
            dist.barrier()
            fs.nodes_set_input(n_in)
            level_nodes = sum(int(b[c][world]) for b in below_all for c in range(5))
            levels += 1
        fs.nodes_expand()
    else:
        fs.seed(first, last - first)
    lo_in, hi_in, in_cap = fs.input_buffers()
    recv_lo = device_tensor(torch, device, lo_in, in_cap, "<i8")
    recv_hi = device_tensor(torch, device, hi_in, 2 * in_cap, "|u1") if hi_in else None
    steps = 0
    while True:
        fs.export(view)                                              # synchronizes the library's stream: the dense outputs are complete
        below_all = _gather_counts(dist, torch, device, view.below, world)
        if sum(int(b[c][world]) for b in below_all for c in range(5)) == 0:
            break
        send, recv = exchange_plan(below_all, rank, world)
        n_in = sum(sum(x) for x in recv)
        if n_in > cap:
            raise X.BwtmError("%d elements fall into rank %d's range, capacity %d" % (n_in, rank, cap))
        held = sum(int(view.totals[c]) for c in range(5))
        class_first = [int(view.class_first[c]) for c in range(6)]
        send_lo = device_tensor(torch, device, view.dense_lo, max(held, 1), "<i8")
        all_to_all_classes(dist, send_lo, class_first, send, recv, recv_lo)
        if recv_hi is not None:
            all_to_all_classes(dist, device_tensor(torch, device, view.dense_hi, 2 * max(held, 1), "|u1"), class_first, send, recv, recv_hi, units=2)
        torch.cuda.synchronize(device)                               # the collectives ran on torch's stream
        dist.barrier()                                               # every rank has received: the send buffers may be overwritten
        fs.set_input(n_in)
        fs.advance()
        steps += 1
    fs.finish()
    fs.free()
    return steps, levels


def merge_partitioned_dist(pkg, a, b, cuts, rank, world, dist, torch, device, times=None):
    """The whole partitioned merge as ONE rank of `world` processes sees it (experimental.merge_partitioned is the one-thread form over
    contexts): its windows from its byte shares, its range of the bitvector, the search with the exchange above, the earlier ranks' bits inside its first output segment (one all-gather of 8 KiB per pair), then the product's range finalize /
    interleave / encode with the product's own small exchanges (dist.combine_range_counts, dist.exchange_encoder_carries).
    Returns the encoded pkg.Slice of this rank's range (total_nbytes = the size of the whole merged stream), what to free, and (element
    steps, node levels).  times (a dict, optional) receives this rank's wall milliseconds by phase."""
    import time as _time
    from . import experimental as X
    t0 = _time.perf_counter()
    from .dist import combine_range_counts, exchange_encoder_carries
    I, R = cuts
    na, nb = int(a.bases), int(b.bases)
    nrecs = ((na + nb) >> 7) + 1
    P = [I[g] + R[g] for g in range(world + 1)]
    M = X.MERGE_MARGIN
    wa = X.index_upload_window(a.data, a.samples[1], na, a.sequences, max(0, I[rank] - M), min(na, I[rank + 1] + M), starts=X._starts_of(a))
    wb = X.index_upload_window(b.data, b.samples[1], nb, b.sequences, max(0, R[rank] - M), min(nb, R[rank + 1] + M), starts=X._starts_of(b))
    ra = X.rank_array_range(wa, wb, P[rank], P[rank + 1])
    pkg.synchronize()
    t1 = _time.perf_counter()
    steps, levels = search_partitioned_dist(pkg, wa, wb, ra, int(b.sequences), R, rank, world, dist, torch, device)
    pkg.synchronize()
    t2 = _time.perf_counter()
    seg = [0] + [P[g] // 65536 for g in range(1, world)]
    bounds = [(min(nrecs, seg[g] * 512), nrecs if g == world - 1 else min(nrecs, seg[g + 1] * 512)) for g in range(world)]
    # boundary bits: row k of `mine` = this rank's bits inside rank k's first segment; after the all-gather rank k ORs its column
    mine = torch.zeros((world, 1024), dtype=torch.int64, device=device)
    torch.cuda.synchronize(device)                                   # the zero-fill ran on torch's stream; the library's streams do not order behind it
    for k in range(rank + 1, world):
        if max(seg[k] * 65536, P[rank]) < min(P[k], P[rank + 1]):
            X.ra_read_words(ra, seg[k] * 65536, seg[k] * 65536 + 65536, mine[k].data_ptr())
    everyone = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    torch.cuda.synchronize(device)
    for h in range(rank):
        X.ra_or_words(ra, seg[rank] * 65536, seg[rank] * 65536 + 65536, everyone[h][rank].data_ptr())
    rec_first, rec_last = bounds[rank]
    ones, local, tail = ra.range_counts(rec_first, rec_last)
    before, total, super_boff, halo = combine_range_counts(ones, local, tail, bounds, rank, world, dist, torch, device)
    if int(total) != nb:
        raise X.BwtmError("the ranks' ranges hold %d set bits, b has %d positions" % (int(total), nb))
    ra.finalize_range(rec_first, rec_last, before, total, super_boff, halo)
    S = pkg.Slice(wa, wb, ra, rec_first, rec_last)
    _, offset, total_bytes = exchange_encoder_carries(S.lasthead(), S.size_table, rank, world, dist, torch, device)
    S.encode(offset)
    S.total_nbytes = total_bytes
    pkg.synchronize()
    if times is not None:
        t3 = _time.perf_counter()
        for k, v in (("ms_windows_from_host_bytes", t1 - t0), ("ms_search", t2 - t1), ("ms_finalize_interleave_encode", t3 - t2)):
            times[k] = times.get(k, 0.0) + v * 1e3
        times["merges"] = times.get("merges", 0) + 1
    return S, (ra, wa, wb), (steps, levels)
