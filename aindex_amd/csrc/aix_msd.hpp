// aix_msd.hpp — the two-level most-significant-digit partition shared by K1 (aix_k1.hip: distinct k-mers) and A2
// (aix_a2msd.hip: positions fill): 64-bit elements are split by their top 11 bits into chunk lists (level 1), every level-1
// partition is split again by the next D2 bits into exactly sized, contiguous buckets (level 2: count, scan, scatter).
// The final per-bucket stage differs (hash aggregation for K1, counting sort + rank by offset for A2) and lives with its caller.
#pragma once
#include <cstdint>

#include "aix_internal.hpp"

namespace aix {

static constexpr int K1_PBITS = 11;
static constexpr int K1_P = 1 << K1_PBITS;            // level-1 partitions
static constexpr int K1_TB = 1024;                    // threads of the split / count / scatter workgroups
static constexpr int K1_WPT = 16;                     // codes per lane and tile
static constexpr int K1_TILE = K1_TB * K1_WPT;        // 16384 codes per tile (128 KiB of u64 in LDS)
static constexpr int K1_CH = 128;                     // entries per chunk (1 KiB)
static constexpr int K1_FILLBITS = 8;                 // cursor = (chunk << 8) | fill, fill in [0, 128]
static constexpr int K1_DUMMY = 64;
static constexpr unsigned K1_MAXGRID = 512;
static constexpr int K1_FB = 256;                     // threads of the per-bucket workgroup
static constexpr int K1_CAP = 2048;                   // hash slots per bucket
static constexpr int K1_DMAX = 1536;                  // distinct remainders a bucket may hold (load <= 0.75)
static constexpr int K1_PRE = 4;                      // entries per lane fetched ahead for the NEXT bucket (256 x 4 = 1024 entries)
static constexpr uint64_t K1_INVALID = ~0ull;

// ---------------------------------------------------------------------------------------------
// the tile counting sort shared by the two levels: P = 2048 counters, two per lane
// ---------------------------------------------------------------------------------------------
struct TileLds {
    uint32_t* hist;      // [P + DUMMY] tile-local count per digit; level 1 re-uses it for the first NEW chunk of the partition
    uint32_t* loc_off;   // [P] exclusive scan of hist
    uint32_t* cursor;    // [P] level 1: (current chunk << 8) | fill; level 2: running position inside the bucket
    uint32_t* wsum;      // [16]
    uint64_t* sorted;    // [TILE]
};
__device__ __forceinline__ TileLds tile_lds(uint8_t* smem) {
    TileLds l;
    l.hist = (uint32_t*)smem;
    l.loc_off = l.hist + K1_P + K1_DUMMY;
    l.cursor = l.loc_off + K1_P;
    l.wsum = l.cursor + K1_P;
    l.sorted = (uint64_t*)(l.wsum + 16);
    return l;
}
static constexpr size_t K1_TILE_LDS = 4 * (3 * K1_P + K1_DUMMY + 16) + 8 * (size_t)K1_TILE;   // 155 968 B

// block-wide exclusive scan of one 32-bit value per lane (1024 lanes); `all` = the total. Two barriers.
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t mine, uint32_t* wsum, uint32_t& all) {
    const int t = threadIdx.x;
    uint32_t s = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(s, d);
        if ((t & 63) >= d) s += y;
    }
    if ((t & 63) == 63) wsum[t >> 6] = s;
    __syncthreads();
    uint32_t off = 0;
    all = 0;
    for (int w = 0; w < K1_TB / 64; ++w) {
        const uint32_t x = wsum[w];
        if (w < (t >> 6)) off += x;
        all += x;
    }
    __syncthreads();
    return off + s - mine;
}

// ---------------------------------------------------------------------------------------------
// level 1
// ---------------------------------------------------------------------------------------------
template <class SRC>
__global__ void __launch_bounds__(K1_TB) k_k1_split(const SRC src, uint32_t s1, uint64_t ntiles, uint32_t region,
                                                   uint16_t* __restrict__ dir_part, uint16_t* __restrict__ dir_cnt, uint64_t* __restrict__ parts,
                                                   uint32_t* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const TileLds L = tile_lds(smem);
    const int t = threadIdx.x;
    constexpr uint32_t FILLMASK = (1u << K1_FILLBITS) - 1;
    const uint32_t region_base = blockIdx.x * region;
    uint32_t next_chunk = 0;                                    // same value in every lane
    L.cursor[2 * t] = K1_CH;
    L.cursor[2 * t + 1] = K1_CH;
    L.hist[2 * t] = 0;
    L.hist[2 * t + 1] = 0;
    if (t < K1_DUMMY) L.hist[K1_P + t] = 0;
    __syncthreads();
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint64_t c[K1_WPT];
        uint32_t rank[K1_WPT / 2];
        const uint32_t dummy = K1_P + (t & (K1_DUMMY - 1));
        src.load(tile, t, c);
#pragma unroll
        for (int j = 0; j < K1_WPT; ++j) {
            const bool ok = c[j] != K1_INVALID;
            const uint32_t r = atomicAdd(&L.hist[ok ? (uint32_t)(c[j] >> s1) : dummy], 1u);
            if (j & 1) rank[j >> 1] |= r << 16; else rank[j >> 1] = r;
        }
        __syncthreads();
        // this lane owns partitions 2t and 2t+1: entries a, b; fresh chunks k0, k1 = by how much they overflow the current chunk
        const uint32_t a = L.hist[2 * t], b = L.hist[2 * t + 1];
        const uint32_t c0 = L.cursor[2 * t], c1 = L.cursor[2 * t + 1];
        const uint32_t tot0 = (c0 & FILLMASK) + a, tot1 = (c1 & FILLMASK) + b;
        const uint32_t k0 = tot0 > (uint32_t)K1_CH ? (tot0 - 1) / K1_CH : 0u;
        const uint32_t k1 = tot1 > (uint32_t)K1_CH ? (tot1 - 1) / K1_CH : 0u;
        uint32_t all;
        const uint32_t excl = block_scan_excl((a + b) | ((k0 + k1) << 16), L.wsum, all);   // entries <= 16384 and fresh chunks <= 2176 per tile: one scan
        const uint32_t e0 = excl & 0xFFFFu, nb0 = next_chunk + (excl >> 16), nb1 = nb0 + k0;
        L.loc_off[2 * t] = e0;
        L.loc_off[2 * t + 1] = e0 + a;
        L.hist[2 * t] = nb0;
        L.hist[2 * t + 1] = nb1;
        for (uint32_t i = 0; i < k0; ++i) { if (nb0 + i < region) dir_part[region_base + nb0 + i] = (uint16_t)(2 * t); else *err = 1u; }
        for (uint32_t i = 0; i < k1; ++i) { if (nb1 + i < region) dir_part[region_base + nb1 + i] = (uint16_t)(2 * t + 1); else *err = 1u; }
        const uint32_t entries = all & 0xFFFFu;
        next_chunk += all >> 16;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < K1_WPT; ++j) {
            if (c[j] != K1_INVALID) {
                const uint32_t r = (j & 1) ? (rank[j >> 1] >> 16) : (rank[j >> 1] & 0xFFFFu);
                L.sorted[L.loc_off[(uint32_t)(c[j] >> s1)] + r] = c[j];
            }
        }
        __syncthreads();
        for (uint32_t i = t; i < entries; i += K1_TB) {
            const uint64_t e = L.sorted[i];
            const uint32_t p = (uint32_t)(e >> s1);
            const uint32_t cu = L.cursor[p];
            const uint32_t pos = (cu & FILLMASK) + (i - L.loc_off[p]);
            uint32_t chunk, o;
            if (pos < (uint32_t)K1_CH) { chunk = cu >> K1_FILLBITS; o = pos; }
            else { chunk = L.hist[p] + (pos - K1_CH) / K1_CH; o = (pos - K1_CH) % K1_CH; }
            if (chunk < region) parts[(uint64_t)(region_base + chunk) * K1_CH + o] = e; else *err = 1u;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {                            // advance the two cursors this lane owns
            const uint32_t p = 2 * t + q, cu = L.cursor[p], tot = (cu & FILLMASK) + (q ? b : a);
            if (tot > (uint32_t)K1_CH) {
                const uint32_t k = (tot - 1) / K1_CH;
                L.cursor[p] = ((L.hist[p] + k - 1) << K1_FILLBITS) | (tot - k * K1_CH);
            } else {
                L.cursor[p] = (cu & ~FILLMASK) | tot;
            }
            L.hist[p] = 0;
        }
        if (t < K1_DUMMY) L.hist[K1_P + t] = 0;
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {                                // the chunk each (workgroup, partition) pair was still filling
        const uint32_t cu = L.cursor[2 * t + q], fill = cu & FILLMASK, chunk = cu >> K1_FILLBITS;
        if (fill < (uint32_t)K1_CH) { if (chunk < region) dir_cnt[region_base + chunk] = (uint16_t)fill; else *err = 1u; }
    }
}

// chunk range [lo, hi) of partition p in the directory sorted by partition
__device__ __forceinline__ void partition_range(const uint16_t* __restrict__ spart, uint32_t cap, uint32_t p, uint32_t* range /* LDS[2] */) {
    if (threadIdx.x < 64) {                                      // wave 0; the others wait at the barrier
        const uint32_t r = wave_lower_bound_pair(spart, cap, p);
        if ((threadIdx.x & 31u) == 0) range[threadIdx.x >> 5] = r;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// level 2, pass 1: bucket sizes of one level-1 partition per workgroup
// ---------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(K1_TB) k_k1_count(const uint64_t* __restrict__ parts, const uint16_t* __restrict__ spart, const uint64_t* __restrict__ sdesc, uint32_t cap,
                                                   uint32_t s2, uint32_t nb2, uint32_t* __restrict__ bucket_cnt) {
    __shared__ uint32_t h2[K1_P];
    __shared__ uint32_t range[2];
    const uint32_t p = blockIdx.x, t = threadIdx.x;
    h2[2 * t] = 0;
    h2[2 * t + 1] = 0;
    partition_range(spart, cap, p, range);
    const uint32_t lo = range[0], hi = range[1], dmask = nb2 - 1;
    // 64 chunks per pass of the workgroup (128 lanes per chunk, one entry each, eight chunks per lane): the eight descriptor loads and
    // then the eight entry loads of a lane are independent, so a pass costs two memory latencies instead of sixteen
    constexpr uint32_t U = 8, LANES_CH = K1_TB / K1_CH;
    for (uint32_t c0 = lo; c0 < hi; c0 += U * LANES_CH) {
        uint64_t desc[U], e[U];
        const uint32_t o = t & (K1_CH - 1);
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) { const uint32_t ci = c0 + u * LANES_CH + (t >> 7); desc[u] = ci < hi ? sdesc[ci] : 0ull; }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) e[u] = o < (uint32_t)(desc[u] >> 32) ? parts[(uint64_t)(uint32_t)desc[u] * K1_CH + o] : ~0ull;      // fill = 0 for chunks past hi
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) if (e[u] != ~0ull) atomicAdd(&h2[(uint32_t)(e[u] >> s2) & dmask], 1u);
    }
    __syncthreads();
    for (uint32_t d = t; d < nb2; d += K1_TB) bucket_cnt[(uint64_t)p * nb2 + d] = h2[d];
}

// ---------------------------------------------------------------------------------------------
// level 2, pass 2: tile sort by the D2 bits, runs written at the bucket bases as 32-bit remainders
// ---------------------------------------------------------------------------------------------
template <typename REM /* uint32_t (K1: remainders below 2^32) or uint64_t (A2: slot remainder | window offset) */>
__global__ void __launch_bounds__(K1_TB) k_k1_scatter(const uint64_t* __restrict__ parts, const uint16_t* __restrict__ spart, const uint64_t* __restrict__ sdesc, uint32_t cap,
                                                     uint32_t s2, uint32_t nb2, const uint32_t* __restrict__ bucket_base, REM* __restrict__ out32) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t range[2];
    const TileLds L = tile_lds(smem);
    const uint32_t p = blockIdx.x, t = threadIdx.x;
    const uint32_t dmask = nb2 - 1;
    const uint64_t remmask = (1ull << s2) - 1;
    for (uint32_t d = t; d < (uint32_t)K1_P; d += K1_TB) {
        L.cursor[d] = d < nb2 ? bucket_base[(uint64_t)p * nb2 + d] : 0u;
        L.hist[d] = 0;
    }
    if (t < (uint32_t)K1_DUMMY) L.hist[K1_P + t] = 0;
    partition_range(spart, cap, p, range);
    const uint32_t lo = range[0], hi = range[1];
    const uint32_t dummy = K1_P + (t & (K1_DUMMY - 1));
    for (uint32_t c0 = lo; c0 < hi; c0 += K1_TILE / K1_CH) {      // a tile = up to 128 chunks of the partition
        uint64_t c[K1_WPT];
        uint32_t rank[K1_WPT / 2];
#pragma unroll
        for (int j = 0; j < K1_WPT; ++j) {                        // code j of this lane: chunk c0 + 8 j + t / 128, entry t % 128
            const uint32_t ci = c0 + 8u * j + (t >> 7);
            c[j] = K1_INVALID;
            if (ci < hi) {
                const uint64_t desc = sdesc[ci];
                const uint32_t id = (uint32_t)desc, fill = (uint32_t)(desc >> 32), o = t & (K1_CH - 1);
                if (o < fill) c[j] = parts[(uint64_t)id * K1_CH + o];
            }
        }
#pragma unroll
        for (int j = 0; j < K1_WPT; ++j) {
            const bool ok = c[j] != K1_INVALID;
            const uint32_t r = atomicAdd(&L.hist[ok ? ((uint32_t)(c[j] >> s2) & dmask) : dummy], 1u);
            if (j & 1) rank[j >> 1] |= r << 16; else rank[j >> 1] = r;
        }
        __syncthreads();
        const uint32_t a = L.hist[2 * t], b = L.hist[2 * t + 1];
        uint32_t all;
        const uint32_t excl = block_scan_excl(a + b, L.wsum, all);
        L.loc_off[2 * t] = excl;
        L.loc_off[2 * t + 1] = excl + a;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < K1_WPT; ++j) {
            if (c[j] != K1_INVALID) {
                const uint32_t r = (j & 1) ? (rank[j >> 1] >> 16) : (rank[j >> 1] & 0xFFFFu);
                L.sorted[L.loc_off[(uint32_t)(c[j] >> s2) & dmask] + r] = c[j];
            }
        }
        __syncthreads();
        for (uint32_t i = t; i < all; i += K1_TB) {
            const uint64_t e = L.sorted[i];
            const uint32_t d = (uint32_t)(e >> s2) & dmask;
            out32[L.cursor[d] + (i - L.loc_off[d])] = (REM)(e & remmask);
        }
        __syncthreads();
        L.cursor[2 * t] += a;
        L.cursor[2 * t + 1] += b;
        L.hist[2 * t] = 0;
        L.hist[2 * t + 1] = 0;
        if (t < (uint32_t)K1_DUMMY) L.hist[K1_P + t] = 0;
        __syncthreads();
    }
}

// directory sort values: chunk id | entries << 32, generated on the fly from the counting iterator
struct K1Desc {
    const uint16_t* dir_cnt;
    __host__ __device__ uint64_t operator()(uint32_t i) const { return (uint64_t)i | ((uint64_t)dir_cnt[i] << 32); }
};

}  // namespace aix
