// aix_k1.hip — K1 row (kmer_counter: count_kmers.cpp:297-341, distinct canonical k-mers with their counts) without a
// full-width library sort. The window codes (k_window_codes, 2k <= 46 bits) are partitioned most-significant-digit first
// in two levels, and every final bucket is counted and ordered inside LDS:
//
//   k_k1_split    level 1: 16 384-code tiles are counting-sorted in LDS by the top 11 bits (rank = one returning LDS atomic
//                 per code) and each partition's run is appended to the workgroup's current 128-entry chunk of that partition
//                 (chunk ids from a per-workgroup region, as in aix_count13.hip: nothing is sized beforehand)
//   directory     (partition, fill) of every chunk, sorted by partition (a few 10^5 u16 keys)
//   k_k1_count    level 2, pass 1: one workgroup per level-1 partition histograms the next D2 bits -> bucket sizes
//   scan          bucket sizes -> bucket bases (exact: level 2 writes contiguous buckets, no chunk lists)
//   k_k1_scatter  level 2, pass 2: the partition's chunks are tile-sorted in LDS by those D2 bits and written out as runs of
//                 32-bit remainders at the bucket bases
//   k_k1_final    one workgroup per bucket (~10^3 codes): open-addressing hash table in LDS (a k-mer that occurs 10^5 times is
//                 ONE slot and 10^5 LDS adds — heavy hitters cost no capacity), distinct keys compacted and bitonic-sorted in
//                 LDS, (key, count) written at the bucket base
//   k_k1_gather   buckets closed up into the final arrays (bucket order = key order)
//
// D2 is chosen so that a bucket holds ~10^3 codes. A bucket with more than K1_DMAX distinct remainders (possible only when
// the codes are far from uniform in their top 11 + D2 bits) raises a flag and the caller falls back to the radix-sort path:
// the result never depends on which path ran.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "aix_internal.hpp"
#include "aix_msd.hpp"

namespace aix {

// ---------------------------------------------------------------------------------------------
// level 1 sources: sixteen codes per lane and tile
// ---------------------------------------------------------------------------------------------
struct K1Codes {                                      // an array of window codes (~0 = no k-mer), e.g. written by k_window_codes
    const uint64_t* codes;
    uint64_t n;
    __device__ __forceinline__ void load(uint64_t tile, int t, uint64_t (&c)[K1_WPT]) const {
        const uint64_t base = tile * K1_TILE + (uint64_t)t;   // code j of this lane = base + j * 1024: coalesced 8 KiB per load instruction
#pragma unroll
        for (int j = 0; j < K1_WPT; ++j) {
            const uint64_t i = base + (uint64_t)j * K1_TB;
            c[j] = i < n ? codes[i] : K1_INVALID;
        }
    }
};
// The windows of a PLAIN buffer, encoded here (count_kmers.cpp:93-136,297-308: A/C/G/T/U either case, canonical form by
// canon_mode): a lane owns 16 consecutive window starts, its 16 + k - 1 bytes are turned ONCE into a 2-bit stream (4 bytes per
// step) and every window is a shift of it — no 8-byte code per window is written to and read back from HBM.
struct K1Plain {
    const uint8_t* buf;
    uint64_t len;         // bytes
    uint64_t n;           // windows = len - k + 1
    int k, canon_mode;
    __device__ __forceinline__ void load(uint64_t tile, int t, uint64_t (&c)[K1_WPT]) const {
        constexpr int ND = 10;                                // 40 bytes >= 16 + 23 - 1
        const uint64_t S = tile * K1_TILE + (uint64_t)t * K1_WPT;
#pragma unroll
        for (int j = 0; j < K1_WPT; ++j) c[j] = K1_INVALID;
        if (S >= n) return;
        uint32_t e[ND];
        {
            const uint32_t o = (uint32_t)((uintptr_t)(buf + S) & 3);
            const uint32_t* q = (const uint32_t*)(buf + S - o);
            const int64_t first = (int64_t)S - o, end = (int64_t)len;
            uint32_t d[ND + 1];
#pragma unroll
            for (int i = 0; i <= ND; ++i) d[i] = (first + 4 * i < end) ? q[i] : 0x0A0A0A0Au;   // never read a dword past the buffer
            const uint32_t sh = o * 8;
#pragma unroll
            for (int i = 0; i < ND; ++i) e[i] = __funnelshift_r(d[i], d[i + 1], sh);
        }
        uint32_t w[3] = {0, 0, 0};
        uint64_t vmask = 0;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            uint32_t x = e[i] & 0xDFDFDFDFu;
            {   // 'U' -> 'T' (count_kmers.cpp:71-88)
                const uint32_t z = x ^ 0x55555555u;
                const uint32_t nz = (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;
                x ^= (~nz & 0x80808080u) >> 7;
            }
            const uint32_t v = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
            const uint32_t diff = x ^ lut4(v, AIX_LUT_ACGT);
            const uint32_t z = ~(((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff | 0x7F7F7F7Fu);    // 0x80 in every byte where diff == 0
            const uint32_t f = z >> 7;
            const uint32_t nib = (f | (f >> 7) | (f >> 14) | (f >> 21)) & 0xFu;
            const uint32_t q2 = ((v << 6) & 0xC0u) | ((v >> 4) & 0x30u) | ((v >> 14) & 0x0Cu) | (v >> 24);
            w[i >> 2] |= q2 << (24 - 8 * (i & 3));
            vmask |= (uint64_t)nib << (4 * i);
        }
        const uint64_t avail = len - S;                        // bytes from S to the end of the buffer
        if (avail < 40) vmask &= (1ull << avail) - 1;
        // window j holds a k-mer iff k consecutive mask bits are set
        uint64_t run = vmask;
        for (int i = 1; i < k; ++i) run &= vmask >> i;
        const uint64_t hi = ((uint64_t)w[0] << 32) | w[1], lo = (uint64_t)w[2] << 32;
        const uint64_t kmask = (1ULL << (2 * k)) - 1;
#pragma unroll
        for (int j = 0; j < K1_WPT; ++j) {
            if (!((run >> j) & 1ull) || S + j >= n) continue;
            const int sh = 128 - 2 * j - 2 * k;                // >= 52
            uint64_t code = (sh >= 64 ? (hi >> (sh - 64)) : ((hi << (64 - sh)) | (lo >> sh))) & kmask;
            if (canon_mode == 1) { const uint64_t x = revcomp_refx86(code, k); code = code < x ? code : x; }
            else if (canon_mode == 2) { const uint64_t x = revcomp(code, k); code = code < x ? code : x; }
            c[j] = code;
        }
    }
};


// ---------------------------------------------------------------------------------------------
// per bucket: hash aggregation + sort of the distinct remainders, all in LDS
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(K1_FB) k_k1_final(const uint32_t* __restrict__ in32, const uint32_t* __restrict__ bucket_base, uint32_t nbuckets, uint32_t s2,
                                                   uint64_t* __restrict__ keys_out, uint32_t* __restrict__ cnt_out, uint32_t* __restrict__ distinct_m,
                                                   uint32_t* __restrict__ overflow) {
    __shared__ uint32_t hk[K1_CAP];          // remainder + 1 (0 = empty)
    __shared__ uint32_t hc[K1_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t dk[K1_DMAX + 4];      // + the +infinity padding of the rank pass
    __shared__ uint32_t dc[K1_DMAX];
    __shared__ uint32_t m_out;
    const uint32_t t = threadIdx.x;
    // A bucket is ~10^3 entries: its global loads (two bases, then the entries) are a dependent chain of HBM latencies that
    // would dominate the few microseconds of LDS work. The base pair and the first 1024 entries of the NEXT bucket are therefore
    // fetched into registers while this one is processed.
    uint32_t nlo = 0, nn = 0, pre[K1_PRE];
    auto prefetch = [&](uint32_t b) {
        nn = 0;
        if (b < nbuckets) {
            nlo = bucket_base[b];
            nn = bucket_base[b + 1] - nlo;
#pragma unroll
            for (int q = 0; q < K1_PRE; ++q) { const uint32_t i = t + q * K1_FB; pre[q] = i < nn ? in32[nlo + i] : 0u; }
        }
    };
    prefetch(blockIdx.x);
    // invariant at the top of every bucket: dk[] holds +infinity everywhere, so the rank pass can read whole 16-byte vectors past the
    // last distinct key; the entries a bucket has used are reset when the next one begins (every wave has passed the closing barrier)
    for (uint32_t i = t; i < (uint32_t)K1_DMAX + 4u; i += K1_FB) dk[i] = 0xFFFFFFFFu;
    uint32_t used = 0;                                          // entries of dk the previous bucket wrote (uniform)
    for (uint32_t b = blockIdx.x; b < nbuckets; b += gridDim.x) {
        const uint32_t lo = nlo, n_e = nn;
        uint32_t cur[K1_PRE];
#pragma unroll
        for (int q = 0; q < K1_PRE; ++q) cur[q] = pre[q];
        prefetch(b + gridDim.x);
        if (n_e == 0) { if (t == 0) distinct_m[b] = 0; continue; }
        // table size: a power of two >= 2 x (the most distinct remainders this bucket can be allowed to hold)
        uint32_t cap = 64;
        const uint32_t want = 2u * min(n_e, (uint32_t)K1_DMAX);
        while (cap < want && cap < (uint32_t)K1_CAP) cap <<= 1;
        for (uint32_t i = t; i < cap; i += K1_FB) { hk[i] = 0; hc[i] = 0; }
        for (uint32_t i = t; i < used; i += K1_FB) dk[i] = 0xFFFFFFFFu;
        used = 0;
        if (t == 0) m_out = 0;
        __syncthreads();
        auto insert = [&](uint32_t v) {
            const uint32_t key = v + 1u;                         // remainders are < 2^31 (s2 <= 31): + 1 cannot wrap to the empty marker
            uint32_t h = (key * 0x9E3779B1u) >> 7;
            for (uint32_t probes = 0; probes < cap; ++probes) {
                h &= cap - 1;
                const uint32_t old = atomicCAS(&hk[h], 0u, key);
                if (old == 0u || old == key) { atomicAdd(&hc[h], 1u); break; }
                ++h;
            }
        };
#pragma unroll
        for (int q = 0; q < K1_PRE; ++q) if (t + q * K1_FB < n_e) insert(cur[q]);
        for (uint32_t i = t + K1_PRE * K1_FB; i < n_e; i += K1_FB) insert(in32[lo + i]);      // the tail of an oversized bucket
        __syncthreads();
        // compaction of the occupied slots: one LDS atomic per WAVE (ballot + prefix popcount), not one per distinct key on one address
        for (uint32_t i0 = 0; i0 < cap; i0 += K1_FB) {
            const uint32_t i = i0 + t;
            const uint32_t k = i < cap ? hk[i] : 0u;
            const uint64_t mask = __ballot(k != 0u);
            uint32_t base = 0;
            if ((t & 63u) == 0u && mask) base = atomicAdd(&m_out, (uint32_t)__popcll(mask));
            base = __shfl(base, 0);
            if (k) {
                const uint32_t idx = base + (uint32_t)__popcll(mask & ((1ull << (t & 63u)) - 1ull));
                if (idx < (uint32_t)K1_DMAX) { dk[idx] = k - 1u; dc[idx] = hc[i]; }
            }
        }
        __syncthreads();
        const uint32_t m = m_out;
        used = min(m, (uint32_t)K1_DMAX);
        if (m > (uint32_t)K1_DMAX) {                             // too many distinct remainders for the LDS arrays (or a full table dropped some)
            if (t == 0) { *overflow = 1u; distinct_m[b] = 0; }
            __syncthreads();
            continue;
        }
        const uint64_t prefix = (uint64_t)b << s2;               // bucket id = the top 11 + D2 bits of the code
        if (m <= 384u) {
            // few distinct keys (the usual case: a bucket's ~10^3 entries are ~10^2 distinct k-mers at sequencing depth): every lane
            // ranks its key by counting the smaller ones (broadcast reads, no conflicts) and stores it straight at its place — no
            // further barrier, instead of the dozens a bitonic network needs
            for (uint32_t i = t; i < m; i += K1_FB) {
                const uint32_t key = dk[i];
                uint32_t r = 0;
                const uint32_t m16 = (m + 15u) & ~15u;                  // dk is +infinity past m (at least up to the next multiple of 16: K1_DMAX + 4 entries, m <= 384)
                for (uint32_t j = 0; j < m16; j += 16) {                // four independent 16-byte reads in flight
                    const uint4 v0 = *reinterpret_cast<const uint4*>(&dk[j]), v1 = *reinterpret_cast<const uint4*>(&dk[j + 4]);
                    const uint4 v2 = *reinterpret_cast<const uint4*>(&dk[j + 8]), v3 = *reinterpret_cast<const uint4*>(&dk[j + 12]);
                    r += (v0.x < key ? 1u : 0u) + (v0.y < key ? 1u : 0u) + (v0.z < key ? 1u : 0u) + (v0.w < key ? 1u : 0u);
                    r += (v1.x < key ? 1u : 0u) + (v1.y < key ? 1u : 0u) + (v1.z < key ? 1u : 0u) + (v1.w < key ? 1u : 0u);
                    r += (v2.x < key ? 1u : 0u) + (v2.y < key ? 1u : 0u) + (v2.z < key ? 1u : 0u) + (v2.w < key ? 1u : 0u);
                    r += (v3.x < key ? 1u : 0u) + (v3.y < key ? 1u : 0u) + (v3.z < key ? 1u : 0u) + (v3.w < key ? 1u : 0u);
                }
                keys_out[lo + r] = prefix | key;
                cnt_out[lo + r] = dc[i];
            }
        } else {
            // many distinct keys: bitonic sort of (dk, dc) by dk inside LDS. Every compare-exchange points upwards (the first step of a
            // merge pairs mirror positions), so m needs no rounding up to a power of two: a partner index at or past m is a virtual
            // +infinity that would never move — nothing is read or written past the m entries
            uint32_t half_p = 1;
            while (2 * half_p < m) half_p <<= 1;                 // pairs per step = P / 2, P = the power of two >= m
            auto cmpx = [&](uint32_t a, uint32_t x) {
                if (x < m) {
                    const uint32_t ka = dk[a], kb = dk[x];
                    if (ka > kb) { dk[a] = kb; dk[x] = ka; const uint32_t ca = dc[a]; dc[a] = dc[x]; dc[x] = ca; }
                }
            };
            for (uint32_t k = 2; k <= 2 * half_p; k <<= 1) {
                const uint32_t hk2 = k >> 1;
                for (uint32_t p = t; p < half_p; p += K1_FB) {
                    const uint32_t blk = p / hk2 * k, pos = p & (hk2 - 1);
                    cmpx(blk + pos, blk + k - 1 - pos);
                }
                __syncthreads();
                for (uint32_t j = k >> 2; j > 0; j >>= 1) {
                    for (uint32_t p = t; p < half_p; p += K1_FB) {
                        const uint32_t a = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                        cmpx(a, a + j);
                    }
                    __syncthreads();
                }
            }
            for (uint32_t i = t; i < m; i += K1_FB) {
                keys_out[lo + i] = prefix | dk[i];
                cnt_out[lo + i] = dc[i];
            }
        }
        if (t == 0) distinct_m[b] = m;
        __syncthreads();
    }
}

// buckets closed up: bucket b's m entries move from its base to dm_off[b]; one wave per bucket
// CNT = uint32_t (the ABI's per-piece result) or uint64_t (what the merge of pieces and kmer_counter's output carry: written here, no widening pass)
template <typename CNT>
__global__ void __launch_bounds__(256) k_k1_gather(const uint64_t* __restrict__ keys_st, const uint32_t* __restrict__ cnt_st, const uint32_t* __restrict__ bucket_base,
                                                  const uint32_t* __restrict__ dm_off, uint32_t nbuckets, uint64_t* __restrict__ keys, CNT* __restrict__ counts) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t nw = (uint64_t)gridDim.x * 4;
    for (uint64_t b = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < nbuckets; b += nw) {
        const uint32_t src = bucket_base[b], dst = dm_off[b], m = dm_off[b + 1] - dst;
        for (uint32_t i = lane; i < m; i += 64) { keys[dst + i] = keys_st[src + i]; counts[dst + i] = (CNT)cnt_st[src + i]; }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static inline uint64_t up256(uint64_t x) { return (x + 255) / 256 * 256; }

bool k1_msd_eligible(uint64_t nwin, int k) {
    if (getenv("AIX_K1_ROCPRIM") != nullptr) return false;      // A/B switch: the radix-sort path
    return k >= 9 && k <= 23 && nwin > 0 && nwin <= (1ull << 31);
}

// d_codes: nwin window codes (~0 = no k-mer), clobbered (re-used as staging). Outputs are pool blocks (caller releases).
// *fell_back = true: a bucket overflowed, nothing was produced (the caller runs the radix-sort path on d_codes — intact in that case
// only if it re-creates the codes, so the caller keeps a way to regenerate them).
// d_plain != nullptr: the windows are encoded inside level 1 (no code array is read; d_codes is then only the 8 B-per-window
// scratch the later stages re-use). plen = bytes of the PLAIN buffer.
hipError_t distinct_from_codes_msd(uint64_t* d_codes, uint64_t nwin, int k, uint64_t** d_keys_out, uint32_t** d_counts_out, uint64_t* n_out, bool* fell_back,
                                   hipStream_t s, const uint8_t* d_plain, uint64_t plen, int canon_mode, uint64_t** d_counts64_out, K1Scratch* scratch) {
    *d_keys_out = nullptr; *d_counts_out = nullptr; *n_out = 0; *fell_back = false;
    if (d_counts64_out) *d_counts64_out = nullptr;
    const uint32_t B = 2u * (uint32_t)k, s1 = B - K1_PBITS;
    // D2: ~768 codes per bucket on average; the remainder must fit 32 bits
    uint32_t D2 = 3;
    while (D2 < 11 && (nwin >> (K1_PBITS + D2)) > 900) ++D2;
    while (s1 - D2 > 31) ++D2;                                  // remainder + 1 must not wrap (0 marks an empty hash slot)
    if (D2 > s1) D2 = s1;
    const uint32_t s2 = s1 - D2, nb2 = 1u << D2;
    const uint32_t nbuckets = (uint32_t)K1_P * nb2;
    const uint64_t ntiles = (nwin + K1_TILE - 1) / K1_TILE;
    const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, K1_MAXGRID);
    uint32_t region = (uint32_t)((ntiles + grid - 1) / grid * (K1_TILE / K1_CH) + K1_P);
    if (const char* e = getenv("AIX_K1_TEST_REGION")) { const long v = atol(e); if (v > 0) region = (uint32_t)v; }   // test hook: an undersized region must fail loudly
    const uint32_t cap = grid * region;
    // one block: err | overflow | dir_part | dir_cnt | spart | sdesc | sort temp | bucket_cnt (nbuckets + 1) | bucket_base | distinct_m | dm_off | out32 | parts
    size_t sort_tmp = 0;
    {
        auto vals = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), K1Desc{nullptr});
        (void)rocprim::radix_sort_pairs(nullptr, sort_tmp, (const uint16_t*)nullptr, (uint16_t*)nullptr, vals, (uint64_t*)nullptr, (size_t)cap, 0u, 12u, s);
    }
    size_t scan_tmp = 0;
    (void)rocprim::exclusive_scan(nullptr, scan_tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)nbuckets + 1, rocprim::plus<uint32_t>(), s);
    const uint64_t o_flags = 0, o_dirp = 256, o_dirc = o_dirp + up256(2ull * cap), o_spart = o_dirc + up256(2ull * cap), o_sdesc = o_spart + up256(2ull * cap),
                   o_stmp = o_sdesc + up256(8ull * cap), o_bcnt = o_stmp + up256(std::max(sort_tmp, scan_tmp)), o_bbase = o_bcnt + up256(4ull * (nbuckets + 1)),
                   o_dm = o_bbase + up256(4ull * (nbuckets + 1)), o_dmo = o_dm + up256(4ull * (nbuckets + 1)), o_parts = o_dmo + up256(4ull * (nbuckets + 1)),
                   total = o_parts + 8ull * cap * K1_CH;
    uint8_t* w = nullptr;
    hipError_t e = scratch ? scratch->need(&scratch->work, &scratch->work_bytes, total) : pool_alloc((void**)&w, total);
    if (e != hipSuccess) return e;
    if (scratch) w = (uint8_t*)scratch->work;
    uint32_t* err = (uint32_t*)(w + o_flags);
    uint32_t* overflow = err + 1;
    uint16_t* dir_part = (uint16_t*)(w + o_dirp);
    uint16_t* dir_cnt = (uint16_t*)(w + o_dirc);
    uint16_t* spart = (uint16_t*)(w + o_spart);
    uint64_t* sdesc = (uint64_t*)(w + o_sdesc);
    void* tmp = w + o_stmp;
    uint32_t* bucket_cnt = (uint32_t*)(w + o_bcnt);
    uint32_t* bucket_base = (uint32_t*)(w + o_bbase);
    uint32_t* distinct_m = (uint32_t*)(w + o_dm);
    uint32_t* dm_off = (uint32_t*)(w + o_dmo);
    // the code array (8 B per window) is free once level 1 has read it: its first half takes the 32-bit remainders of level 2,
    // its second half the per-bucket counts; the chunk array (>= 8 B per window) is free after level 2 and takes the per-bucket keys
    uint32_t* out32 = (uint32_t*)d_codes;
    uint32_t* cnt_st = (uint32_t*)d_codes + nwin;
    uint64_t* parts = (uint64_t*)(w + o_parts);
    uint64_t* keys_st = parts;
    uint64_t* keys = nullptr;
    uint32_t* counts = nullptr;
    uint32_t flags[2] = {0, 0};
    uint32_t total_distinct = 0;
    do {
        e = hipFuncSetAttribute((const void*)k_k1_split<K1Codes>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K1_TILE_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_k1_split<K1Plain>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K1_TILE_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_k1_scatter<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K1_TILE_LDS);
        if (e == hipSuccess) e = hipMemsetAsync(w, 0, 256, s);
        if (e == hipSuccess) e = hipMemsetD16Async((hipDeviceptr_t)dir_part, (unsigned short)K1_P, cap, s);      // "no partition": sorts behind every real one
        if (e == hipSuccess) e = hipMemsetD16Async((hipDeviceptr_t)dir_cnt, (unsigned short)K1_CH, cap, s);      // chunks are full unless the split says otherwise
        if (e == hipSuccess) e = hipMemsetAsync(bucket_cnt, 0, 4ull * (nbuckets + 1), s);
        if (e == hipSuccess) e = hipMemsetAsync(distinct_m, 0, 4ull * (nbuckets + 1), s);
        if (e != hipSuccess) break;
        if (d_plain) hipLaunchKernelGGL(k_k1_split<K1Plain>, dim3(grid), dim3(K1_TB), K1_TILE_LDS, s, K1Plain{d_plain, plen, nwin, k, canon_mode}, s1, ntiles, region,
                                        dir_part, dir_cnt, parts, err);
        else hipLaunchKernelGGL(k_k1_split<K1Codes>, dim3(grid), dim3(K1_TB), K1_TILE_LDS, s, K1Codes{(const uint64_t*)d_codes, nwin}, s1, ntiles, region, dir_part,
                                dir_cnt, parts, err);
        auto vals = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), K1Desc{dir_cnt});
        size_t tb = sort_tmp;
        e = rocprim::radix_sort_pairs(tmp, tb, (const uint16_t*)dir_part, spart, vals, sdesc, (size_t)cap, 0u, 12u, s);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_k1_count, dim3(K1_P), dim3(K1_TB), 0, s, (const uint64_t*)parts, (const uint16_t*)spart, (const uint64_t*)sdesc, cap, s2, nb2, bucket_cnt);
        tb = scan_tmp;
        e = rocprim::exclusive_scan(tmp, tb, bucket_cnt, bucket_base, 0u, (size_t)nbuckets + 1, rocprim::plus<uint32_t>(), s);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_k1_scatter<uint32_t>, dim3(K1_P), dim3(K1_TB), K1_TILE_LDS, s, (const uint64_t*)parts, (const uint16_t*)spart, (const uint64_t*)sdesc, cap, s2, nb2,
                           (const uint32_t*)bucket_base, out32);
        hipLaunchKernelGGL(k_k1_final, dim3(std::min<uint32_t>(nbuckets, 256u * 5u)), dim3(K1_FB), 0, s, (const uint32_t*)out32, (const uint32_t*)bucket_base, nbuckets, s2,
                           keys_st, cnt_st, distinct_m, overflow);
        tb = scan_tmp;
        e = rocprim::exclusive_scan(tmp, tb, distinct_m, dm_off, 0u, (size_t)nbuckets + 1, rocprim::plus<uint32_t>(), s);
        if (e == hipSuccess) e = hipMemcpyAsync(flags, w, 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&total_distinct, dm_off + nbuckets, 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) break;
        if (flags[0]) { e = hipErrorAssert; break; }             // chunk region exhausted: never silent (the caller reports it)
        if (flags[1]) { *fell_back = true; break; }
        if (total_distinct == 0) break;
        e = pool_alloc((void**)&keys, 8ull * total_distinct);
        if (e == hipSuccess) e = pool_alloc((void**)&counts, (d_counts64_out ? 8ull : 4ull) * total_distinct);
        if (e != hipSuccess) break;
        const dim3 ggrid(std::min<uint32_t>((nbuckets + 3) / 4, 1u << 16));
        if (d_counts64_out) hipLaunchKernelGGL(k_k1_gather<uint64_t>, ggrid, dim3(256), 0, s, (const uint64_t*)keys_st, (const uint32_t*)cnt_st, (const uint32_t*)bucket_base,
                                               (const uint32_t*)dm_off, nbuckets, keys, (uint64_t*)counts);
        else hipLaunchKernelGGL(k_k1_gather<uint32_t>, ggrid, dim3(256), 0, s, (const uint64_t*)keys_st, (const uint32_t*)cnt_st, (const uint32_t*)bucket_base,
                                (const uint32_t*)dm_off, nbuckets, keys, counts);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    } while (false);
    { const hipError_t es = hipStreamSynchronize(s); if (e == hipSuccess) e = es; }
    if (!scratch) pool_free(w);
    if (e != hipSuccess || *fell_back) {
        if (keys) pool_free(keys);
        if (counts) pool_free(counts);
        return e;
    }
    *d_keys_out = keys; *n_out = total_distinct;
    if (d_counts64_out) { *d_counts64_out = (uint64_t*)counts; *d_counts_out = nullptr; } else *d_counts_out = counts;
    return hipSuccess;
}

}  // namespace aix
