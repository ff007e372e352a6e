/*
  api/ingest.hip.h -- reads -> index: leaves by suffix sort (kernels/ingest.hip.h), grown into one
  index by a merge tree that uses the merger itself (search + interleave on records, no native form in between).
  The reference has no such component: it merges BWTs that RopeBWT / SGA built (README.md:5,20); this is what stands in
  for them, and what bench.py builds its inputs with.  Part of bwtm_api.hip.
*/
#pragma once

// Streaming builder: reads come in batches, in collection order; the stack holds indexes of adjacent read ranges whose
// levels decrease towards the top (a binary counter: a leaf is level 0, a merge of two level-k indexes is level k + 1).
struct bwtm_builder
{
  bwtm_context* ctx = nullptr;
  u64 leaf_reads = 0;
  u64 reads = 0;
  std::vector<std::pair<int, bwtm_index*>> stack;
};

namespace
{

// One leaf: `m` reads (rows of `stride` bytes on the device, `width` columns, optional lengths) -> records.
int leaf_from_reads(const u8* reads, u64 m, u32 width, u64 stride, const u32* lengths, bwtm_index** out)
{
  const u64 w1 = (u64)width + 1;
  if(m * w1 >= (1ull << 32)) { return fail(BWTM_EINVAL, "ingest: a leaf of %llu reads x %u columns exceeds 32-bit suffix ids", (unsigned long long)m, width); }
  DevBuf first, flags;
  TRY(flags.alloc(sizeof(u32), true));
  u64 n = m * w1;
  if(lengths)
  {
    TRY(first.alloc((m + 1) * sizeof(u64)));
    LAUNCH("ingest_lens", k_ingest_lens, div_up(m + 1, BLOCK_THREADS), BLOCK_THREADS, lengths, m, width, first.as<u64>());
    TRY(device_scan<0>(first.as<u64>(), first.as<u64>(), m + 1));
    TRY(fetch_u64(first.as<u64>() + m, 0));
    HIP_TRY(hipStreamSynchronize(CTX.stream));
    n = CTX.host_scratch[0];
  }
  const u64 ntiles = div_up(n, ING_TILE);
  DevBuf ids[2], keys[2], hist;
  for(int k = 0; k < 2; k++) { TRY(ids[k].alloc(n * sizeof(u32))); TRY(keys[k].alloc(n * sizeof(u64))); }
  TRY(hist.alloc(256 * ntiles * sizeof(u64)));
  if(m > 0)
  {
    LAUNCH("ingest_init", k_ingest_init, div_up(m * w1, BLOCK_THREADS), BLOCK_THREADS, reads, stride, lengths,
      lengths ? first.as<const u64>() : (const u64*)nullptr, m, width, ids[0].as<u32>(), flags.as<u32>());
  }
  const u32 nwords = (u32)div_up(w1, ING_SYMS);
  // a key word may start at any offset <= width of the last word: rows are padded so that both loads stay inside the row
  const u32 row_words = (u32)div_up(3 * (w1 + (u64)nwords * ING_SYMS), 64) + 1;
  DevBuf packed; TRY(packed.alloc(m * row_words * sizeof(u64)));
  LAUNCH("ingest_pack", k_ingest_pack, div_up(m * row_words, BLOCK_THREADS), BLOCK_THREADS, reads, stride, lengths, m, width, row_words, packed.as<u64>());
  int cur = 0;
  for(u32 word = nwords; n > 0 && word-- > 0; )
  {
    LAUNCH("ingest_keys", k_ingest_keys, div_up(n, BLOCK_THREADS), BLOCK_THREADS, packed.as<const u64>(), row_words, width,
      ids[cur].as<const u32>(), n, word, keys[cur].as<u64>());
    // Columns >= width are zero for every suffix: the low bits of the last word that only they can reach need no pass.
    const u32 last_col = (width > 0 ? width - 1 : 0);                                   // last column that can hold a symbol
    const u32 last_sym = (last_col >= word * ING_SYMS ? std::min(last_col - word * ING_SYMS, ING_SYMS - 1) : 0);
    const u32 zero_bits = 3 * (ING_SYMS - 1 - last_sym);
    for(u32 shift = 0; shift < 3 * ING_SYMS; shift += 8)
    {
      if(shift + 8 <= zero_bits) { continue; }
      LAUNCH("ingest_hist", k_ingest_hist, ntiles, WAVE, keys[cur].as<const u64>(), n, shift, ntiles, hist.as<u64>());
      TRY(device_scan<0>(hist.as<u64>(), hist.as<u64>(), 256 * ntiles));
      LAUNCH("ingest_scatter", k_ingest_scatter, ntiles, WAVE, keys[cur].as<const u64>(), ids[cur].as<const u32>(), n, shift, ntiles,
        hist.as<const u64>(), keys[cur ^ 1].as<u64>(), ids[cur ^ 1].as<u32>());
      cur ^= 1;
    }
  }
  DevBuf bad;
  if(g_tune.ingest_verify && n > 1)
  {
    TRY(bad.alloc(sizeof(u64), true));
    LAUNCH("ingest_verify", k_ingest_verify, div_up(n, BLOCK_THREADS), BLOCK_THREADS, reads, stride, lengths, width, ids[cur].as<const u32>(), n, bad.as<unsigned long long>());
    TRY(fetch_u64(bad.as<u64>(), 9));
  }
  DevBuf sym; TRY(sym.alloc(n));
  if(n > 0) { LAUNCH("ingest_symbols", k_ingest_symbols, div_up(n, BLOCK_THREADS), BLOCK_THREADS, reads, stride, width, ids[cur].as<const u32>(), n, sym.as<u8>()); }
  HIP_TRY(hipMemcpyAsync(CTX.host_scratch + 8, flags.p, sizeof(u32), hipMemcpyDeviceToHost, CTX.stream));
  bwtm_index* x = new bwtm_index();
  x->ctx = t_ctx;
  int rc = index_from_symbols(sym.as<const u8>(), n, x);               // synchronizes
  if(rc == BWTM_OK)
  {
    const u32 f = *(const u32*)(CTX.host_scratch + 8);
    if(f & 1u) { rc = fail(BWTM_EINVAL, "ingest: a read holds a value outside 1..5"); }
    else if(f & 2u) { rc = fail(BWTM_EINVAL, "ingest: a read is longer than the row width %u", width); }
    else if(x->m != m) { rc = fail(BWTM_EINVAL, "ingest: built %llu sequences from %llu reads", (unsigned long long)x->m, (unsigned long long)m); }
    else if(bad.p && CTX.host_scratch[9] != 0) { rc = fail(BWTM_ENODEV, "ingest: the suffix sort left %llu adjacent pairs out of order", (unsigned long long)CTX.host_scratch[9]); }
  }
  if(rc != BWTM_OK) { (void)hipStreamSynchronize(CTX.stream); delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

int builder_push(bwtm_builder* b, bwtm_index* leaf)
{
  b->stack.push_back(std::make_pair(0, leaf));
  while(b->stack.size() >= 2 && b->stack[b->stack.size() - 1].first == b->stack[b->stack.size() - 2].first)
  {
    bwtm_index* right = b->stack.back().second; b->stack.pop_back();
    bwtm_index* left = b->stack.back().second; const int level = b->stack.back().first; b->stack.pop_back();
    bwtm_index* merged = nullptr;
    int rc = merge_records(left, right, true, &merged);
    if(rc != BWTM_OK) { return rc; }
    b->stack.push_back(std::make_pair(level + 1, merged));
  }
  return BWTM_OK;
}

void builder_destroy(bwtm_builder* b)
{
  if(!b) { return; }
  for(auto& e : b->stack) { index_destroy(e.second); }
  delete b;
}

} // namespace

extern "C" int bwtm_builder_create(uint64_t leaf_reads, bwtm_builder** out)
{
  ENTER(nullptr);
  if(!out) { return fail(BWTM_EINVAL, "bwtm_builder_create: null argument"); }
  bwtm_builder* b = new bwtm_builder();
  b->ctx = t_ctx;
  b->leaf_reads = (leaf_reads == 0 ? (u64)1 << 19 : leaf_reads);
  *out = b;
  return BWTM_OK;
}

extern "C" int bwtm_builder_add(bwtm_builder* b, const uint8_t* reads, uint64_t nreads, uint32_t width, uint64_t stride,
  const uint32_t* lengths, int on_device)
{
  if(!b) { return fail(BWTM_EINVAL, "bwtm_builder_add: null builder"); }
  ENTER(b->ctx);
  if(nreads == 0) { return BWTM_OK; }
  if(!reads || stride < width) { return fail(BWTM_EINVAL, "bwtm_builder_add: null reads or stride < width"); }
  const u64 w1 = (u64)width + 1;
  const u64 leaf = std::max<u64>(1, std::min<u64>(b->leaf_reads, ((1ull << 32) - 1) / w1));
  for(u64 first = 0; first < nreads; first += leaf)
  {
    const u64 m = std::min(leaf, nreads - first);
    const u8* d_reads = reads + first * stride;
    const u32* d_len = (lengths ? lengths + first : nullptr);
    DevBuf staged, staged_len;
    if(!on_device)
    {
      TRY(staged.alloc(m * stride));
      HIP_TRY(hipMemcpyAsync(staged.p, d_reads, m * stride, hipMemcpyHostToDevice, CTX.stream));
      d_reads = staged.as<const u8>();
      if(lengths)
      {
        TRY(staged_len.alloc(m * sizeof(u32)));
        HIP_TRY(hipMemcpyAsync(staged_len.p, d_len, m * sizeof(u32), hipMemcpyHostToDevice, CTX.stream));
        d_len = staged_len.as<const u32>();
      }
    }
    bwtm_index* x = nullptr;
    TRY(leaf_from_reads(d_reads, m, width, stride, d_len, &x));
    b->reads += m;
    TRY(builder_push(b, x));
  }
  HIP_TRY(hipStreamSynchronize(CTX.stream));           // the caller may release `reads` on return
  return BWTM_OK;
}

extern "C" uint64_t bwtm_builder_reads(const bwtm_builder* b) { return b ? b->reads : 0; }

extern "C" int bwtm_builder_finish(bwtm_builder* b, bwtm_index** out)
{
  if(!b || !out) { return fail(BWTM_EINVAL, "bwtm_builder_finish: null argument"); }
  ENTER(b->ctx);
  int rc = BWTM_OK;
  while(rc == BWTM_OK && b->stack.size() >= 2)
  {
    bwtm_index* right = b->stack.back().second; const int lr = b->stack.back().first; b->stack.pop_back();
    bwtm_index* left = b->stack.back().second; const int ll = b->stack.back().first; b->stack.pop_back();
    bwtm_index* merged = nullptr;
    rc = merge_records(left, right, true, &merged);
    if(rc == BWTM_OK) { b->stack.push_back(std::make_pair(std::max(ll, lr) + 1, merged)); }
  }
  bwtm_index* x = nullptr;
  if(rc == BWTM_OK)
  {
    if(b->stack.empty())
    {
      x = new bwtm_index(); x->ctx = t_ctx;
      rc = index_from_symbols(nullptr, 0, x);
      if(rc != BWTM_OK) { delete x; x = nullptr; }
    }
    else { x = b->stack.back().second; b->stack.pop_back(); }
  }
  if(rc == BWTM_OK) { rc = (hipStreamSynchronize(CTX.stream) == hipSuccess ? BWTM_OK : fail(BWTM_ENODEV, "bwtm_builder_finish: synchronize failed")); }
  builder_destroy(b);
  if(rc != BWTM_OK) { if(x) { index_destroy(x); } return rc; }
  *out = x;
  return BWTM_OK;
}

extern "C" void bwtm_builder_free(bwtm_builder* b)
{
  if(!b) { return; }
  Scope scope(b->ctx);
  builder_destroy(b);
}
