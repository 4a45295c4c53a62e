// aix_a2msd.hip — A2 row (lu_compressed_worker, src/hash.cpp:960-1060) without a full-width library sort.
//
// After the probe every window holds a bucket h (its MPHF slot) or none. The reference hands a bucket's slots out in arrival
// order, i.e. ascending file offsets with one worker: positions[indices[h] + r] = offset + 1 for the r-th window of h, r < tf[h].
// That is a sort of (h, offset) pairs — but only the GROUPING by h has to be computed, the order inside a group is known from
// the offsets themselves. So the pairs (h << 32 | offset) go through the two-level MSD partition of aix_msd.hpp on the bits of
// h (unstable tile counting sorts, no order kept), which leaves buckets of ~10^3 pairs that span 2^rbits consecutive slots, and
// one workgroup per bucket finishes inside LDS:
//     counting sort by the slot remainder (one returning LDS atomic per pair: its arrival number is its place in the slot's run; runs
//     start at multiples of four entries and are padded with +infinity), then the lanes walk the GROUPED array, so that neighbouring
//     lanes hold pairs of one slot and read its 16-byte record and its run at the same addresses (broadcasts): every pair ranks
//     itself among the offsets of its slot (a slot holds tf[h] ~ coverage-many pairs) and writes its position. What a bucket needs
//     from memory (bounds, first 1 024 pairs, indices / filled / tf of its slots) is fetched while the previous bucket is processed.
// Slots, fill counters and output ranges of a bucket are contiguous, so the per-bucket global reads and the position writes
// are dense. A slot with more than A2_HEAVY pairs is sorted in place by the whole workgroup. A bucket that does not fit (> A2_CAP
// pairs: a k-mer that occurs thousands of times in one piece) is set aside; those buckets — and only those — are gathered and go
// through the radix sort + run placement the whole input used to take. A piece whose workspace cannot be allocated takes that
// path as a whole (positions_fill).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "aix_internal.hpp"
#include "aix_msd.hpp"

namespace aix {

static constexpr int A2_FB = 256;                     // threads of the per-bucket workgroup
static constexpr int A2_CAP = 4096;                   // pairs a bucket may hold in LDS
static constexpr int A2_PRE = 4;                      // pairs per lane held in registers (256 x 4 = 1024: the usual bucket)
static constexpr int A2_HEAVY = 128;                   // pairs of one slot above which its run is sorted cooperatively instead of ranked pair by pair
static constexpr int A2_RBITS_MAX = 8;
static constexpr int A2_R = 1 << A2_RBITS_MAX;        // slots per bucket at most

// level-1 source: the probe's output, keys[i] = bucket of window i (>= nslots: none)
struct A2Keys {
    const uint32_t* keys;
    uint64_t n;
    uint32_t nslots;
    __device__ __forceinline__ void load(uint64_t tile, int t, uint64_t (&c)[K1_WPT]) const {
        const uint64_t base = tile * K1_TILE + (uint64_t)t;   // window j of this lane = base + j * 1024: coalesced 4 KiB per load instruction
#pragma unroll
        for (int j = 0; j < K1_WPT; ++j) {
            const uint64_t i = base + (uint64_t)j * K1_TB;
            const uint32_t h = i < n ? keys[i] : 0xFFFFFFFFu;
            c[j] = h < nslots ? ((uint64_t)h << 32) | (uint32_t)i : K1_INVALID;
        }
    }
};

struct A2Over {                                       // head of the workspace
    uint32_t err;                                     // a chunk id left its region (cannot happen while the region bound is right): the call fails
    uint32_t n;                                       // buckets set aside
    unsigned long long elems;                         // pairs in them
};

// ---------------------------------------------------------------------------------------------
// per bucket: counting sort by slot + rank by offset, all in LDS
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(A2_FB, 6) k_a2_final(const IndexDev ix, const uint64_t* __restrict__ rem, const uint32_t* __restrict__ bucket_base, uint32_t nbuckets,
                                                   uint32_t rbits, uint32_t cap_e, uint64_t piece_first, uint32_t* __restrict__ filled, int advance,
                                                   const uint64_t* __restrict__ indices, uint64_t* __restrict__ positions, A2Over* __restrict__ over,
                                                   uint32_t* __restrict__ over_b, uint64_t* __restrict__ over_off) {
    __shared__ uint32_t hist2[2][A2_R];               // pairs per slot of the bucket; two copies: the one of the NEXT bucket is cleared while this one is read
    __shared__ uint32_t cursor[A2_R];                 // exclusive scan of the padded counts: where every slot's run starts (a bucket of more than 1024 pairs advances it while grouping)
    __shared__ uint4 slotrec[A2_R];                   // per slot, one 16-byte read: x = start of its run in offs | pairs << 16, y = slots still free in its
                                                      // positions range (tf[h] - filled[h]), z:w = indices[h] + filled[h]
    __shared__ __attribute__((aligned(16))) uint32_t offs[A2_CAP + 4];      // window offsets grouped by slot
    __shared__ uint16_t grp2slot[A2_CAP / 4 + 2];     // slot of every aligned group of four entries of offs (a run starts at a multiple of four); 0xFFFF: placed by the heavy path
    __shared__ uint32_t wsum[A2_FB / 64];
    __shared__ uint32_t heavy_n[2];                   // slots of the bucket with more than A2_HEAVY pairs (cleared with hist2)
    __shared__ uint32_t heavy_s[2][A2_CAP / A2_HEAVY];
    const uint32_t t = threadIdx.x, R = 1u << rbits, n = (uint32_t)ix.n;

    constexpr int SPT = A2_R / A2_FB;                 // slots a lane owns in the scan
    // A bucket is ~10^3 pairs and its global reads form a chain (bucket bounds -> pairs; slots -> output ranges) of HBM latencies
    // that would dominate the few microseconds of LDS work: everything the NEXT bucket needs from memory is therefore fetched
    // while this one is processed (its bounds, its first 1024 pairs, and indices / filled / tf of the slots this lane owns).
    uint32_t n_lo = 0, n_n = 0, n_fl[SPT];
    uint64_t n_e[A2_PRE], n_ind[SPT], n_tf[SPT];
    auto prefetch = [&](uint32_t b) {
        n_n = 0;
        if (b < nbuckets) {
            n_lo = bucket_base[b];
            n_n = bucket_base[b + 1] - n_lo;
#pragma unroll
            for (int q = 0; q < A2_PRE; ++q) { const uint32_t i = t + q * A2_FB; n_e[q] = i < n_n ? rem[n_lo + i] : ~0ull; }
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const uint32_t s = SPT * t + j;
                const uint64_t h = ((uint64_t)b << rbits) | s;
                n_ind[j] = 0; n_fl[j] = 0; n_tf[j] = 0;
                if (s < R && h < n && n_n) {
                    n_ind[j] = indices[h];
                    n_fl[j] = filled[h];
                    n_tf[j] = indices[h + 1] - n_ind[j];             // tf[h]: indices is its prefix sum (n + 1 entries); no second table is read
                }
            }
        }
    };
    prefetch(blockIdx.x);
    for (uint32_t i = t; i < R; i += A2_FB) { hist2[0][i] = 0; hist2[1][i] = 0; }
    if (t < 2) heavy_n[t] = 0;
    __syncthreads();
    uint32_t par = 0;                                 // which copy of hist this bucket uses (uniform)
    for (uint32_t b = blockIdx.x; b < nbuckets; b += gridDim.x) {
        const uint32_t lo = n_lo, n_e_cnt = n_n;
        uint64_t e[A2_PRE], s_ind[SPT], s_tf[SPT];
        uint32_t s_fl[SPT];
#pragma unroll
        for (int q = 0; q < A2_PRE; ++q) e[q] = n_e[q];
#pragma unroll
        for (int j = 0; j < SPT; ++j) { s_ind[j] = n_ind[j]; s_fl[j] = n_fl[j]; s_tf[j] = n_tf[j]; }
        prefetch(b + gridDim.x);
        const uint32_t n_e = n_e_cnt;
        if (n_e == 0) continue;
        if (n_e > cap_e) {                                         // set aside (the caller sorts these buckets by themselves)
            if (t == 0) {
                const uint32_t idx = atomicAdd(&over->n, 1u);
                over_b[idx] = b;
                over_off[idx] = atomicAdd(&over->elems, (unsigned long long)n_e);
            }
            continue;
        }
        uint32_t* hist = hist2[par];
        // the usual bucket (<= 1024 pairs) lives in registers; the tail of a fuller one is re-read (it is in L2: level 2 has just written it)
        uint32_t r[A2_PRE];                                        // arrival number of a pair among the pairs of its slot: its place inside the slot's run
#pragma unroll
        for (int q = 0; q < A2_PRE; ++q)
            if (e[q] != ~0ull) r[q] = atomicAdd(&hist[(uint32_t)(e[q] >> 32)], 1u);
        for (uint32_t i = t + A2_PRE * A2_FB; i < n_e; i += A2_FB) atomicAdd(&hist[(uint32_t)(rem[lo + i] >> 32)], 1u);
        __syncthreads();                                           // also: every wave has left the previous bucket
        // scan: a lane owns SPT consecutive slots and publishes their output ranges. Every slot's run starts at a multiple of four
        // entries and is padded with +infinity, so that the ranking below reads whole 16-byte vectors and needs no index masks
        uint32_t c4[SPT], sum = 0;
#pragma unroll
        for (int j = 0; j < SPT; ++j) { const uint32_t s = SPT * t + j; c4[j] = s < R ? hist[s] : 0u; sum += (c4[j] + 3u) & ~3u; }
        uint32_t sc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(sc, d); if ((t & 63) >= d) sc += y; }
        if ((t & 63) == 63) wsum[t >> 6] = sc;
        // the other copy of the histogram was last read while the previous bucket was placed: clear it for the next one
        for (uint32_t i = t; i < R; i += A2_FB) hist2[par ^ 1][i] = 0;
        if (t == 0) heavy_n[par ^ 1] = 0;
        __syncthreads();
        uint32_t padded = 0;
#pragma unroll
        for (int w = 0; w < A2_FB / 64; ++w) padded += wsum[w];
        if (padded > (uint32_t)A2_CAP) {                           // the padding does not fit (many sparsely used slots in a full bucket): set aside
            if (t == 0) {
                const uint32_t idx = atomicAdd(&over->n, 1u);
                over_b[idx] = b;
                over_off[idx] = atomicAdd(&over->elems, (unsigned long long)n_e);
            }
            par ^= 1;
            continue;
        }
        {
            uint32_t off = sc - sum;
            for (uint32_t w = 0; w < (t >> 6); ++w) off += wsum[w];
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const uint32_t s = SPT * t + j;
                const uint64_t h = ((uint64_t)b << rbits) | s;
                if (s < R) {
                    cursor[s] = off;
                    const uint32_t c = c4[j], cr = (c + 3u) & ~3u;
                    for (uint32_t i = c; i < cr; ++i) offs[off + i] = 0xFFFFFFFFu;
                    uint32_t lim = 0;
                    uint64_t base = 0;
                    if (c && h < n) {
                        const uint32_t fl = s_fl[j];
                        base = s_ind[j] + fl;
                        const uint64_t room = s_tf[j] > fl ? s_tf[j] - fl : 0ull;
                        lim = room > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)room;
                        if (advance) { const uint64_t tot = (uint64_t)fl + c; filled[h] = tot > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tot; }
                        if (c > (uint32_t)A2_HEAVY) heavy_s[par][atomicAdd(&heavy_n[par], 1u)] = s;
                    }
                    slotrec[s] = make_uint4(off | (c << 16), lim, (uint32_t)base, (uint32_t)(base >> 32));
                    if (c <= (uint32_t)A2_HEAVY) for (uint32_t g = off >> 2; g < ((off + cr) >> 2); ++g) grp2slot[g] = (uint16_t)s;
                    off += cr;
                }
            }
        }
        __syncthreads();
        // grouping: any order inside a slot's run (the order is recovered from the offsets below)
        if (n_e <= (uint32_t)(A2_PRE * A2_FB)) {                   // the usual bucket: every pair kept its arrival number, no second atomic
#pragma unroll
            for (int q = 0; q < A2_PRE; ++q)
                if (e[q] != ~0ull) offs[cursor[(uint32_t)(e[q] >> 32)] + r[q]] = (uint32_t)e[q];
        } else {
#pragma unroll
            for (int q = 0; q < A2_PRE; ++q)
                if (e[q] != ~0ull) offs[atomicAdd(&cursor[(uint32_t)(e[q] >> 32)], 1u)] = (uint32_t)e[q];
            for (uint32_t i = t + A2_PRE * A2_FB; i < n_e; i += A2_FB) { const uint64_t x = rem[lo + i]; offs[atomicAdd(&cursor[(uint32_t)(x >> 32)], 1u)] = (uint32_t)x; }
        }
        __syncthreads();
        // a slot with many pairs (a k-mer repeated hundreds of times inside this piece): its run is sorted in place by the whole
        // workgroup — a bitonic network whose compare-exchanges all point upwards, so that runs of any length need no padding
        // (a partner index past the run is a virtual +infinity that would never move) — and placed straight from the sorted run
        const uint32_t nheavy = heavy_n[par];                      // uniform
        for (uint32_t hi = 0; hi < nheavy; ++hi) {
            const uint32_t s = heavy_s[par][hi];
            const uint4 rec = slotrec[s];
            const uint32_t cnt = rec.x >> 16;
            uint32_t* run = offs + (rec.x & 0xFFFFu);
            for (uint32_t g = t; g < ((cnt + 3u) >> 2); g += A2_FB) grp2slot[((rec.x & 0xFFFFu) >> 2) + g] = 0xFFFFu;      // visible after the first barrier below
            uint32_t half_p = 1;
            while (2 * half_p < cnt) half_p <<= 1;                 // pairs per step = P / 2, P = the power of two >= cnt
            auto cmpx = [&](uint32_t i, uint32_t x) {
                if (x < cnt) { const uint32_t a = run[i], c = run[x]; if (a > c) { run[i] = c; run[x] = a; } }
            };
            for (uint32_t k = 2; k <= 2 * half_p; k <<= 1) {
                const uint32_t hk = k >> 1;
                for (uint32_t p = t; p < half_p; p += A2_FB) {     // first step of a merge: mirror pairs inside every k-block
                    const uint32_t blk = p / hk * k, pos = p & (hk - 1);
                    cmpx(blk + pos, blk + k - 1 - pos);
                }
                __syncthreads();
                for (uint32_t j = k >> 2; j > 0; j >>= 1) {
                    for (uint32_t p = t; p < half_p; p += A2_FB) {
                        const uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                        cmpx(i, i + j);
                    }
                    __syncthreads();
                }
            }
            const uint32_t lim = min(cnt, rec.y);
            const uint64_t base = ((uint64_t)rec.w << 32) | rec.z;
            for (uint32_t i = t; i < lim; i += A2_FB) positions[base + i] = piece_first + run[i] + 1;
        }
        // every other pair ranks itself among the offsets of its slot, four per LDS read (ascending offsets = the reference's arrival
        // order). The lanes walk the GROUPED array: neighbouring lanes hold pairs of the same slot, so the slot record and the run are
        // read at a handful of distinct addresses per wave instruction (broadcasts) instead of 64 scattered ones
        for (uint32_t i = t; i < padded; i += A2_FB) {
            const uint32_t off = offs[i];
            const uint32_t s = grp2slot[i >> 2];
            if (off == 0xFFFFFFFFu || s == 0xFFFFu) continue;                                      // padding / placed by the heavy path
            const uint4 rec = slotrec[s];
            const uint32_t cnt = rec.x >> 16, first = rec.x & 0xFFFFu, end = first + cnt;
            uint32_t rank = 0;
            for (uint32_t q = first; q < end; q += 4) {                                            // `first` is a multiple of four, the run is padded with +infinity
                const uint4 v = *reinterpret_cast<const uint4*>(&offs[q]);
                rank += (v.x < off ? 1u : 0u) + (v.y < off ? 1u : 0u) + (v.z < off ? 1u : 0u) + (v.w < off ? 1u : 0u);
            }
            if (rank < rec.y) positions[(((uint64_t)rec.w << 32) | rec.z) + rank] = piece_first + off + 1;      // hash.cpp:1037-1040, 1-based
        }
        par ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------
// the buckets that were set aside: gathered with their slot restored, sorted as 64-bit keys, placed run by run
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_a2_over_gather(const uint64_t* __restrict__ rem, const uint32_t* __restrict__ bucket_base, const uint32_t* __restrict__ over_b,
                                                       const uint64_t* __restrict__ over_off, uint32_t n_over, uint32_t s2, uint64_t* __restrict__ out) {
    for (uint32_t i = blockIdx.x; i < n_over; i += gridDim.x) {
        const uint32_t b = over_b[i], lo = bucket_base[b], n_e = bucket_base[b + 1] - lo;
        const uint64_t dst = over_off[i], prefix = (uint64_t)b << s2;
        for (uint32_t j = threadIdx.x; j < n_e; j += 256) out[dst + j] = prefix | rem[lo + j];
    }
}
__global__ void __launch_bounds__(256) k_a2_first64(const uint64_t* __restrict__ sorted, uint64_t m, uint32_t* __restrict__ first) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += stride) {
        const uint32_t h = (uint32_t)(sorted[j] >> 32);
        if (j == 0 || (uint32_t)(sorted[j - 1] >> 32) != h) first[h] = (uint32_t)j;
    }
}
__global__ void __launch_bounds__(256) k_a2_place64(const IndexDev ix, const uint64_t* __restrict__ sorted, uint64_t m, uint64_t piece_first, const uint32_t* __restrict__ first,
                                                   const uint32_t* __restrict__ filled, const uint64_t* __restrict__ indices, uint64_t* __restrict__ positions) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;

    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += stride) {
        const uint64_t e = sorted[j];
        const uint32_t h = (uint32_t)(e >> 32);
        const uint64_t rank = (uint64_t)filled[h] + (j - first[h]);
        const uint64_t base = indices[h], tf = indices[h + 1] - base;    // tf[h] from its prefix sums
        if (rank < tf) positions[base + rank] = piece_first + (uint32_t)e + 1;
    }
}
__global__ void __launch_bounds__(256) k_a2_advance64(const uint64_t* __restrict__ sorted, uint64_t m, const uint32_t* __restrict__ first, uint32_t* __restrict__ filled) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += stride) {
        const uint32_t h = (uint32_t)(sorted[j] >> 32);
        if (j + 1 < m && (uint32_t)(sorted[j + 1] >> 32) == h) continue;                            // one writer per slot: the end of its run
        const uint64_t tot = (uint64_t)filled[h] + (j - first[h] + 1);
        filled[h] = tot > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tot;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static inline uint64_t up256(uint64_t x) { return (x + 255) / 256 * 256; }
static inline unsigned grid256(uint64_t work) { return (unsigned)std::min<uint64_t>(std::max<uint64_t>((work + 255) / 256, 1), 8192); }

bool a2_msd_eligible(uint64_t nwin, uint64_t n) {
    if (const char* e = getenv("AIX_A2_MSD")) return atoi(e) != 0 && nwin > 0 && nwin <= (1ull << 31) && n > 0 && n <= (1ull << 30);      // A/B and test switch
    return nwin >= (1ull << 22) && nwin <= (1ull << 31) && n > 0 && n <= (1ull << 30);                                                    // short buffers: fewer launches win
}

// keys: the probe's output for the nwin windows of this piece (bucket, or >= n for none). filled: u32[n] occurrences of every
// bucket in the pieces / shards before this one; advanced by this piece's tallies when `advance`.
hipError_t a2_msd_place(const IndexDev& ix, const uint32_t* keys, uint64_t nwin, uint64_t piece_first, uint32_t* filled, bool advance, const uint64_t* d_indices,
                        uint64_t* d_positions, hipStream_t s, bool* untouched) {
    if (untouched) *untouched = false;
    const uint64_t n = ix.n;
    uint32_t Bh = 12;
    while (Bh < 32 && ((n - 1) >> Bh)) ++Bh;                   // slots are < 2^Bh
    // a bucket spans 2^rbits slots and should hold ~10^3 pairs (at most nwin / n pairs per slot on average)
    uint32_t rbits = 0;
    uint64_t target = 1024;
    if (const char* e = getenv("AIX_A2_TARGET")) { const long v = atol(e); if (v > 0) target = (uint64_t)v; }       // A/B switch
    while (rbits < (uint32_t)A2_RBITS_MAX && ((nwin << (rbits + 1)) / n) <= target) ++rbits;
    int D2 = (int)Bh - K1_PBITS - (int)rbits;
    if (D2 > 11) { D2 = 11; rbits = Bh - K1_PBITS - 11; }
    if (D2 < 1) { D2 = 1; rbits = Bh - K1_PBITS - 1; }
    const uint32_t s2 = 32 + rbits, s1 = s2 + (uint32_t)D2, nb2 = 1u << D2;
    const uint32_t nbuckets = (uint32_t)K1_P * nb2;
    uint32_t cap_e = A2_CAP;
    if (const char* e = getenv("AIX_A2_TEST_CAP")) { const long v = atol(e); if (v > 0 && v < A2_CAP) cap_e = (uint32_t)v; }   // test hook: force the set-aside path
    const uint64_t ntiles = (nwin + K1_TILE - 1) / K1_TILE;
    const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, K1_MAXGRID);
    const uint32_t region = (uint32_t)((ntiles + grid - 1) / grid * (K1_TILE / K1_CH) + K1_P);
    const uint32_t cap = grid * region;
    size_t sort_tmp = 0, scan_tmp = 0;
    {
        auto vals = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), K1Desc{nullptr});
        (void)rocprim::radix_sort_pairs(nullptr, sort_tmp, (const uint16_t*)nullptr, (uint16_t*)nullptr, vals, (uint64_t*)nullptr, (size_t)cap, 0u, 12u, s);
        (void)rocprim::exclusive_scan(nullptr, scan_tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)nbuckets + 1, rocprim::plus<uint32_t>(), s);
    }
    // one block: head | dir_part | dir_cnt | spart | sdesc | sort / scan temp | bucket_cnt | bucket_base | over_b | over_off | rem (8 B per window) | parts
    const uint64_t o_dirp = 256, o_dirc = o_dirp + up256(2ull * cap), o_spart = o_dirc + up256(2ull * cap), o_sdesc = o_spart + up256(2ull * cap),
                   o_stmp = o_sdesc + up256(8ull * cap), o_bcnt = o_stmp + up256(std::max(sort_tmp, scan_tmp)), o_bbase = o_bcnt + up256(4ull * (nbuckets + 1)),
                   o_ob = o_bbase + up256(4ull * (nbuckets + 1)), o_oo = o_ob + up256(4ull * nbuckets), o_rem = o_oo + up256(8ull * nbuckets),
                   o_parts = o_rem + up256(8ull * nwin), total = o_parts + 8ull * cap * K1_CH;
    uint8_t* w = nullptr;
    hipError_t e = getenv("AIX_A2_TEST_NOMEM") ? hipErrorOutOfMemory : pool_alloc((void**)&w, total);      // test hook: the workspace "does not fit"
    if (e != hipSuccess) { if (untouched) *untouched = true; return e; }     // the workspace did not fit: the caller may take the sort path, nothing was written
    A2Over* over = (A2Over*)w;
    uint16_t* dir_part = (uint16_t*)(w + o_dirp);
    uint16_t* dir_cnt = (uint16_t*)(w + o_dirc);
    uint16_t* spart = (uint16_t*)(w + o_spart);
    uint64_t* sdesc = (uint64_t*)(w + o_sdesc);
    void* tmp = w + o_stmp;
    uint32_t* bucket_cnt = (uint32_t*)(w + o_bcnt);
    uint32_t* bucket_base = (uint32_t*)(w + o_bbase);
    uint32_t* over_b = (uint32_t*)(w + o_ob);
    uint64_t* over_off = (uint64_t*)(w + o_oo);
    uint64_t* rem = (uint64_t*)(w + o_rem);
    uint64_t* parts = (uint64_t*)(w + o_parts);
    A2Over head{0, 0, 0};
    uint64_t *gathered = nullptr, *sorted = nullptr;
    uint32_t* first = nullptr;
    void* tmp2 = nullptr;
    do {
        e = hipFuncSetAttribute((const void*)k_k1_split<A2Keys>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K1_TILE_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_k1_scatter<uint64_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K1_TILE_LDS);
        if (e == hipSuccess) e = hipMemsetAsync(w, 0, 256, s);
        if (e == hipSuccess) e = hipMemsetD16Async((hipDeviceptr_t)dir_part, (unsigned short)K1_P, cap, s);      // "no partition": sorts behind every real one
        if (e == hipSuccess) e = hipMemsetD16Async((hipDeviceptr_t)dir_cnt, (unsigned short)K1_CH, cap, s);      // chunks are full unless the split says otherwise
        if (e == hipSuccess) e = hipMemsetAsync(bucket_cnt, 0, 4ull * (nbuckets + 1), s);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_k1_split<A2Keys>, dim3(grid), dim3(K1_TB), K1_TILE_LDS, s, A2Keys{keys, nwin, (uint32_t)std::min<uint64_t>(n, 0xFFFFFFFFull)}, s1, ntiles,
                           region, dir_part, dir_cnt, parts, &over->err);
        auto vals = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), K1Desc{dir_cnt});
        size_t tb = sort_tmp;
        e = rocprim::radix_sort_pairs(tmp, tb, (const uint16_t*)dir_part, spart, vals, sdesc, (size_t)cap, 0u, 12u, s);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_k1_count, dim3(K1_P), dim3(K1_TB), 0, s, (const uint64_t*)parts, (const uint16_t*)spart, (const uint64_t*)sdesc, cap, s2, nb2, bucket_cnt);
        tb = scan_tmp;
        e = rocprim::exclusive_scan(tmp, tb, bucket_cnt, bucket_base, 0u, (size_t)nbuckets + 1, rocprim::plus<uint32_t>(), s);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_k1_scatter<uint64_t>, dim3(K1_P), dim3(K1_TB), K1_TILE_LDS, s, (const uint64_t*)parts, (const uint16_t*)spart, (const uint64_t*)sdesc, cap, s2,
                           nb2, (const uint32_t*)bucket_base, rem);
        int per_cu = 4, cus = 256;                                // persistent grid: exactly what is resident at once
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_a2_final, A2_FB, 0);
        { int dev = 0; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); }
        const uint32_t fgrid = std::min<uint32_t>(nbuckets, (uint32_t)std::max(per_cu, 1) * (uint32_t)std::max(cus, 1));
        hipLaunchKernelGGL(k_a2_final, dim3(fgrid), dim3(A2_FB), 0, s, ix, (const uint64_t*)rem, (const uint32_t*)bucket_base, nbuckets,
                           rbits, cap_e, piece_first, filled, advance ? 1 : 0, d_indices, d_positions, over, over_b, over_off);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&head, over, sizeof(head), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) break;
        if (head.err) { e = hipErrorAssert; break; }            // chunk region exhausted: never silent
        if (head.n == 0) break;
        // the buckets that did not fit: radix sort of their pairs (slot in the high word, offset in the low word), then run placement
        const uint64_t m = head.elems;
        e = pool_alloc((void**)&gathered, 8 * m);
        if (e == hipSuccess) e = pool_alloc((void**)&sorted, 8 * m);
        if (e == hipSuccess) e = pool_alloc((void**)&first, 4 * n);
        size_t tb2 = 0;
        if (e == hipSuccess) e = rocprim::radix_sort_keys(nullptr, tb2, gathered, sorted, (size_t)m, 0u, 32u + Bh, s);
        if (e == hipSuccess) e = pool_alloc(&tmp2, tb2 ? tb2 : 1);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_a2_over_gather, dim3(std::min<uint32_t>(head.n, 4096u)), dim3(256), 0, s, (const uint64_t*)rem, (const uint32_t*)bucket_base,
                           (const uint32_t*)over_b, (const uint64_t*)over_off, head.n, s2, gathered);
        e = rocprim::radix_sort_keys(tmp2, tb2, gathered, sorted, (size_t)m, 0u, 32u + Bh, s);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_a2_first64, dim3(grid256(m)), dim3(256), 0, s, (const uint64_t*)sorted, m, first);
        hipLaunchKernelGGL(k_a2_place64, dim3(grid256(m)), dim3(256), 0, s, ix, (const uint64_t*)sorted, m, piece_first, (const uint32_t*)first, (const uint32_t*)filled,
                           d_indices, d_positions);
        if (advance) hipLaunchKernelGGL(k_a2_advance64, dim3(grid256(m)), dim3(256), 0, s, (const uint64_t*)sorted, m, (const uint32_t*)first, filled);
        e = hipGetLastError();
    } while (false);
    { const hipError_t es = hipStreamSynchronize(s); if (e == hipSuccess) e = es; }
    if (tmp2) pool_free(tmp2);
    if (first) pool_free(first);
    if (sorted) pool_free(sorted);
    if (gathered) pool_free(gathered);
    pool_free(w);
    return e;
}

}  // namespace aix
