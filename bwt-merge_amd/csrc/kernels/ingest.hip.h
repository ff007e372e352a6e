/*
  kernels/ingest.hip.h -- reads -> BWT symbols of a leaf collection (the ingest the reference leaves to
  RopeBWT / SGA, README.md:5,20; PlainData::read, formats.cpp:133-161, is the text form of what comes in).
  Part of bwtm_kernels.hip.h (included there, inside namespace bwtm); gfx950 only.

  The suffixes of the leaf are sorted by LSD radix sort.  A suffix is named by id = read * (width + 1) + offset (32 bits;
  offset == length names the read's endmarker suffix).  The sort key is the suffix text with the endmarker smallest and equal
  suffixes in read order: the text is cut into words of 21 symbols (3 bits each, zeros from the endmarker on), the words are
  processed last to first, and each word is sorted by eight stable 8-bit counting passes over (key, id) pairs:

    k_ingest_pack     the reads at 3 bits per symbol, so that a key word of any suffix is two 8-byte loads
    k_ingest_keys     the current word of every suffix, gathered in the current order
    k_ingest_hist     per-tile digit histograms (one wave per tile, LDS counters)
    (exclusive scan of the bin-major histogram = first output slot of every (digit, tile))
    k_ingest_scatter  stable scatter: a lane's slot = its tile's next slot for the digit + the number of lower lanes of the
                      wave with the same digit (nine ballots)
    k_ingest_symbols  BWT symbol of every suffix in final order (the symbol before it, endmarker for offset 0)
*/
#pragma once

constexpr int ING_ROUNDS = 64;
constexpr u32 ING_TILE = WAVE * ING_ROUNDS;      // suffixes per single-wave workgroup
constexpr u32 ING_SYMS = 21;                     // symbols per 63-bit key word

// first[k] (before the scan) = number of suffixes of read k.
__global__ void __launch_bounds__(BLOCK_THREADS) k_ingest_lens(const u32* lengths, u64 m, u32 width, u64* first)
{
  const u64 s = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(s > m) { return; }
  first[s] = (s < m ? (u64)(lengths[s] < width ? lengths[s] : width) + 1 : 0);
}

// ids in text order (read-major), only the suffixes that exist; flags bit 0: a symbol outside 1..5, bit 1: a length > width.
__global__ void __launch_bounds__(BLOCK_THREADS) k_ingest_init(const u8* reads, u64 stride, const u32* lengths, const u64* first,
  u64 m, u32 width, u32* ids, u32* flags)
{
  const u32 w1 = width + 1;
  const u64 t = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(t >= m * w1) { return; }
  const u32 s = (u32)(t / w1), o = (u32)(t - (u64)s * w1);
  u32 len = width;
  if(lengths) { len = lengths[s]; if(len > width) { if(o == 0) { atomicOr(flags, 2u); } len = width; } }
  if(o > len) { return; }
  if(o < len) { const u32 c = reads[(u64)s * stride + o]; if(c < 1 || c > 5) { atomicOr(flags, 1u); } }
  ids[(first ? first[s] : (u64)s * w1) + o] = (u32)t;
}

// The reads as rows of `row_words` u64 holding 3 bits per symbol, MSB first (symbol j of a read in stream bits [3 j, 3 j + 3)
// counted from the top bit of word 0), zeros from the endmarker on: a 21-symbol key word of any suffix is then two loads.
__global__ void __launch_bounds__(BLOCK_THREADS) k_ingest_pack(const u8* reads, u64 stride, const u32* lengths, u64 m, u32 width,
  u32 row_words, u64* packed)
{
  const u64 t = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(t >= m * row_words) { return; }
  const u64 s = t / row_words; const u32 k = (u32)(t - s * row_words);
  u32 len = (lengths ? lengths[s] : width); if(len > width) { len = width; }
  const u8* row = reads + s * stride;
  u64 out = 0;
  const u32 j_lo = (64 * k) / 3, j_hi = (64 * k + 63) / 3;                 // symbols that overlap stream bits [64 k, 64 k + 64)
  for(u32 j = j_lo; j <= j_hi && j < len; j++)
  {
    const u64 v = row[j] & 7u;
    const int top = 64 * (int)k + 63 - 3 * (int)j;                          // bit of this word that holds the symbol's high bit ...
    const int sh = top - 2;                                                // ... and its low bit
    out |= (sh >= 0 ? (sh < 64 ? v << sh : 0ull) : v >> (-sh));
  }
  packed[t] = out;
}

// keys[i] = word `word` of the suffix ids[i]: symbol j of the word in bits [3 (20 - j), 3 (20 - j) + 3).
// Both packed words come in ONE 16-byte load (8-byte aligned).  The obvious form -- two 8-byte loads, the second one under
// `if(r != 0)` -- compiled to `global_load_dwordx2 v[4:5], v[4:5], off offset:8` (destination = its own address registers,
// issued while the first load from the same registers was in flight) and returned a garbled second word for about 2 of
// 5.3e7 lanes per launch on gfx950 / ROCm 7.2, in 15 of 25 leaf builds (found by k_ingest_verify; memory was intact before
// and after).  This form and a byte-wise gather never failed; see DESIGN.md section 7.
__global__ void __launch_bounds__(BLOCK_THREADS) k_ingest_keys(const u64* packed, u32 row_words, u32 width,
  const u32* ids, u64 n, u32 word, u64* keys)
{
  const u64 i = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(i >= n) { return; }
  const u32 w1 = width + 1, id = ids[i];
  const u32 s = id / w1, o = id - s * w1;
  const u32 bit = 3 * (o + word * ING_SYMS);
  const u32* w = (const u32*)(packed + (u64)s * row_words + (bit >> 6));
  const u32 r = bit & 63u;
  const u32 a = w[1], b = w[0], c = w[3], d = w[2];
  const u64 hi = ((u64)a << 32) | b, lo = ((u64)c << 32) | d;
  const u64 v = (hi << r) | ((lo >> 1) >> (63 - r));                   // r = 0: the second term is 0
  keys[i] = v >> 1;
}

// hist[d * ntiles + t] = number of keys of tile t whose digit (bits [shift, shift + 8)) is d.
// The keys of ING_BATCH rounds are loaded together, unconditionally (clamped index): a predicated load per round made every
// round a full memory round trip (64 in a row per wave).
constexpr int ING_BATCH = 8;

__global__ void __launch_bounds__(WAVE) k_ingest_hist(const u64* keys, u64 n, u32 shift, u64 ntiles, u64* hist)
{
  __shared__ u32 cnt[256];
  const u32 lane = lane_id();
#pragma unroll
  for(u32 k = 0; k < 4; k++) { cnt[k * 64 + lane] = 0; }
  wave_sync_lds();
  const u64 base = (u64)blockIdx.x * ING_TILE;
  for(int r0 = 0; r0 < ING_ROUNDS && base + (u64)r0 * WAVE < n; r0 += ING_BATCH)
  {
    u64 key[ING_BATCH];
#pragma unroll
    for(int j = 0; j < ING_BATCH; j++) { const u64 i = base + (u64)(r0 + j) * WAVE + lane; key[j] = keys[i < n ? i : n - 1]; }
#pragma unroll
    for(int j = 0; j < ING_BATCH; j++)
    {
      const u64 i = base + (u64)(r0 + j) * WAVE + lane;
      if(i < n) { atomicAdd(&cnt[(u32)(key[j] >> shift) & 255u], 1u); }
    }
  }
  wave_sync_lds();
#pragma unroll
  for(u32 k = 0; k < 4; k++) { hist[(u64)(k * 64 + lane) * ntiles + blockIdx.x] = cnt[k * 64 + lane]; }
}

// Stable scatter by the same digit; `first_slot` is the exclusive scan of the histogram.
__global__ void __launch_bounds__(WAVE) k_ingest_scatter(const u64* keys_in, const u32* ids_in, u64 n, u32 shift, u64 ntiles,
  const u64* first_slot, u64* keys_out, u32* ids_out)
{
  __shared__ u32 next[256];
  const u32 lane = lane_id();
#pragma unroll
  for(u32 k = 0; k < 4; k++) { next[k * 64 + lane] = (u32)first_slot[(u64)(k * 64 + lane) * ntiles + blockIdx.x]; }
  wave_sync_lds();
  const u64 base = (u64)blockIdx.x * ING_TILE;
  const u64 below = (1ull << lane) - 1ull;
  for(int r0 = 0; r0 < ING_ROUNDS && base + (u64)r0 * WAVE < n; r0 += ING_BATCH)
  {
    u64 kbuf[ING_BATCH]; u32 ibuf[ING_BATCH];
#pragma unroll
    for(int j = 0; j < ING_BATCH; j++)
    {
      const u64 i = base + (u64)(r0 + j) * WAVE + lane;
      const u64 ic = (i < n ? i : n - 1);
      kbuf[j] = keys_in[ic]; ibuf[j] = ids_in[ic];
    }
#pragma unroll
    for(int j = 0; j < ING_BATCH; j++)
    {
      const u64 i = base + (u64)(r0 + j) * WAVE + lane;
      const bool valid = (i < n);
      const u64 key = kbuf[j];
      const u32 d = (valid ? (u32)(key >> shift) & 255u : 256u);     // lanes past the end form a group of their own
      u64 same = ~0ull;
#pragma unroll
      for(u32 b = 0; b < 9; b++)
      {
        const bool bit = ((d >> b) & 1u) != 0;
        const u64 bal = __ballot(bit);
        same &= (bit ? bal : ~bal);
      }
      const u32 rank = (u32)__popcll(same & below), total = (u32)__popcll(same);
      const u32 slot = (valid ? next[d] + rank : 0u);
      wave_sync_lds();                                                // every lane has read its counter
      if(valid && rank + 1 == total) { next[d] += total; }
      wave_sync_lds();
      if(valid) { keys_out[slot] = key; ids_out[slot] = ibuf[j]; }
    }
  }
}

// sym[i] = symbol before the suffix ids[i] (0 = endmarker for the suffix that is a whole read).
__global__ void __launch_bounds__(BLOCK_THREADS) k_ingest_symbols(const u8* reads, u64 stride, u32 width, const u32* ids, u64 n, u8* sym)
{
  const u64 i = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(i >= n) { return; }
  const u32 w1 = width + 1, id = ids[i];
  const u32 s = id / w1, o = id - s * w1;
  sym[i] = (o > 0 ? reads[(u64)s * stride + o - 1] : (u8)0);
}

// Verification (bwtm_tune("ingest_verify", 1); the GPU tests run with it): adjacent suffixes of the final order compared
// symbol by symbol from the reads themselves; bad[0] += pairs that are out of order or equal.
__global__ void __launch_bounds__(BLOCK_THREADS) k_ingest_verify(const u8* reads, u64 stride, const u32* lengths, u32 width, const u32* ids, u64 n,
  unsigned long long* bad)
{
  const u64 i = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(i + 1 >= n) { return; }
  const u32 w1 = width + 1;
  const u32 a = ids[i], b = ids[i + 1];
  const u32 sa = a / w1, oa = a - sa * w1, sb = b / w1, ob = b - sb * w1;
  const u32 la = (lengths ? (lengths[sa] < width ? lengths[sa] : width) : width), lb = (lengths ? (lengths[sb] < width ? lengths[sb] : width) : width);
  const u8* ra = reads + (u64)sa * stride; const u8* rb = reads + (u64)sb * stride;
  int cmp = 0;
  for(u32 k = 0; k <= width && cmp == 0; k++)
  {
    const u32 ca = (oa + k < la ? ra[oa + k] : 0), cb = (ob + k < lb ? rb[ob + k] : 0);
    if(ca != cb) { cmp = (ca < cb ? -1 : 1); }
    else if(ca == 0) { break; }
  }
  if(cmp == 0) { cmp = (a < b ? -1 : 1); }                             // equal suffixes: read order
  if(cmp > 0) { atomicAdd(&bad[0], 1ull); }
}
