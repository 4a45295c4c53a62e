// aix_count13.hip — 13-mer dense counting without global atomics (K13 row; count_kmers13.cpp:131-161).
//
// The 4^13 counter table (256 MiB as u32) is 1600x the LDS of a CU and 1.4e9 scattered memory-side atomics
// run at ~23 G/s on MI355X (60 ms for 10 M reads). This path streams instead:
//   k_c13_split_chunked : rolling 2-bit encode (24 or 32 window starts per lane, 4-byte SWAR); a 12 288- or 32 768-window tile is
//                         counting-sorted inside LDS by the top 11 code bits (rank from one ds_add_rtn per window) and each
//                         partition's run of low-15-bit payloads (u16) is appended to the workgroup's current 256-entry chunk
//                         of that partition; chunk ids come from a per-workgroup region, handed out by the tile's block scan
//   rocPRIM radix sort  : the chunk directory (partition of every chunk) -> per-partition chunk lists (a few M u16 keys)
//   k_c13_hist_chunked  : one workgroup per partition: 32 768 u32 counters in 128 KiB of LDS, ds_add per element, counters
//                         written through the code->mphf permutation — every bin of the output has exactly one writer
// Nothing is sized before the split, so there is no separate counting pass over the input.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "aix_internal.hpp"

namespace aix {

static constexpr int C13_PBITS = 11;
static constexpr int C13_P = 1 << C13_PBITS;          // partitions
static constexpr int C13_BINBITS = 26 - C13_PBITS;    // 15
static constexpr int C13_BINS = 1 << C13_BINBITS;     // 32768 bins per partition
static constexpr int C13_TB = 1024;                   // threads per workgroup of the histogram kernel
static constexpr int C13_DUMMY = 64;                  // scratch histogram bins for windows that do not count

// The split kernel comes in two shapes (template parameters TB = threads per workgroup, WPT = window starts per lane):
//   1024 x 32: a 32 768-window tile, 152 KiB of LDS, ONE workgroup per CU — the default;
//    512 x 24: a 12 288-window tile,  72 KiB of LDS, TWO workgroups per CU (AIX_C13_SHAPE=small). The kernel alternates VALU-bound
//              phases (encode, write-out) with LDS-bound ones (the returning atomic per window, the tile scatter) between barriers,
//              and PMC says the two add up (27 K + 16 K cycles per 32 K windows), so a second resident workgroup was
//              expected to run its phases in the gaps. Measured on one box, alternating: 3.84 against 3.09 ms per 10 M reads for
//              the 13-mer source, 1.60 against 1.17 ms per 2^28 slots: the per-tile work that does not shrink with the tile (a scan
//              and three words for each of the 2 048 partitions, four per lane instead of two) costs more than the overlap returns.

// Encode the WPT windows whose starts are [S, S+WPT): the packed 2-bit stream (Run13::code<j>() = 26 bits) and a validity bit mask.
// Bytes at positions >= len count as separators. Upper-casing and the ACGT test follow
// normalize_sequence / is_valid_kmer (count_kmers13.cpp:100-126).
struct Run13 {
    uint32_t w[3];        // 2-bit stream of the WPT + 12 bytes (<= 44), base 0 in the top bits of w[0]
    uint32_t valid;       // bit j: window j (bytes j..j+12) consists of bases only
    // 26-bit code of window j = stream bits [70 - 2j, 95 - 2j] of the 96-bit big-endian stream {w0, w1, w2}; two VALU
    // operations, so the windows are re-extracted where they are needed instead of being kept in 32 registers
    template <int J>
    __device__ __forceinline__ uint32_t code() const {
        constexpr int sh = 70 - 2 * J;
        uint32_t c;
        if (sh >= 64) c = w[0] >> (sh - 64);
        else if (sh >= 32) c = __funnelshift_r(w[1], w[0], sh - 32);
        else c = __funnelshift_r(w[2], w[1], sh);
        return c & 0x3FFFFFFu;
    }
    template <int J>
    __device__ __forceinline__ void get(uint32_t& c, uint32_t& ok) const { c = code<J>(); ok = (valid >> J) & 1u; }
};
template <int WPT, int J, typename RUN, typename F>
__device__ __forceinline__ void for_each_window13(const RUN& r, F&& f) {
    if constexpr (J < WPT) {
        uint32_t code, ok;
        r.template get<J>(code, ok);
        f(J, code, ok);
        for_each_window13<WPT, J + 1>(r, f);
    }
}

// The WPT + 12 bytes behind window start S as aligned dwords (bytes at and past `len` read as separators); fetched on their own so
// that the split kernel can have the NEXT tile's loads in flight while it writes the current one out.
template <int WPT>
struct Raw13 {
    uint32_t e[(WPT + 12) / 4];
    uint64_t limit;                                   // bytes available from S (0: nothing)
};
template <int WPT>
__device__ __forceinline__ Raw13<WPT> fetch_raw13(const uint8_t* __restrict__ buf, uint64_t len, uint64_t S) {
    static_assert((WPT + 12) % 4 == 0 && WPT + 12 <= 48 && WPT <= 32, "the stream is three dwords, the validity mask one");
    constexpr int ND = (WPT + 12) / 4;
    Raw13<WPT> r;
    r.limit = S < len ? len - S : 0;
#pragma unroll
    for (int k = 0; k < ND; ++k) r.e[k] = 0x0A0A0A0Au;
    if (S >= len) return r;
    const uint32_t o = (uint32_t)((uintptr_t)(buf + S) & 3);
    const uint32_t* q = (const uint32_t*)(buf + S - o);           // pointer arithmetic keeps these global_load (not flat_load)
    const int64_t first = (int64_t)S - o, end = (int64_t)len;     // byte offset of d[0] from buf (>= -3)
    uint32_t d[ND + 1];
#pragma unroll
    for (int k = 0; k <= ND; ++k) d[k] = (first + 4 * k < end) ? q[k] : 0x0A0A0A0Au;   // never read a dword past the buffer
    const uint32_t sh = o * 8;
#pragma unroll
    for (int k = 0; k < ND; ++k) r.e[k] = __funnelshift_r(d[k], d[k + 1], sh);
    return r;
}
template <int WPT>
__device__ __forceinline__ Run13 encode_raw13(const Raw13<WPT>& raw) {
    constexpr int NB = WPT + 12;
    constexpr int ND = NB / 4;
    Run13 r{{0, 0, 0}, 0};
    if (raw.limit == 0) return r;
    // Four bytes at a time (the VALU issues one wave instruction per clock per CU, and the byte-serial form of this loop
    // was ~2/3 of the kernel): x = upper-cased bytes; v = 2-bit value of every byte; a byte is a base iff it equals the
    // letter its own 2-bit value maps back to (v_perm_b32 as the 4-entry table); q = the four values packed first-base-high.
    uint64_t vmask = 0;                               // bit i: byte i is A/C/G/T (either case)
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        const uint32_t x = raw.e[k] & 0xDFDFDFDFu;
        const uint32_t v = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
        const uint32_t diff = x ^ lut4(v, AIX_LUT_ACGT);
        const uint32_t z = ~(((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff | 0x7F7F7F7Fu);      // 0x80 in every byte where diff == 0
        const uint32_t f = z >> 7;
        const uint32_t nib = (f | (f >> 7) | (f >> 14) | (f >> 21)) & 0xFu;
        const uint32_t q = ((v << 6) & 0xC0u) | ((v >> 4) & 0x30u) | ((v >> 14) & 0x0Cu) | (v >> 24);
        r.w[k >> 2] |= q << (24 - 8 * (k & 3));
        vmask |= (uint64_t)nib << (4 * k);
    }
    if (raw.limit < (uint64_t)NB) vmask &= (1ull << raw.limit) - 1;   // bytes at and past the end of the buffer are separators
    // window j (bytes j..j+12) is countable iff 13 consecutive mask bits are set
    const uint64_t m2 = vmask & (vmask >> 1), m4 = m2 & (m2 >> 2), m8 = m4 & (m4 >> 4);
    uint64_t ok = m8 & (m4 >> 8) & (vmask >> 12);
    if (WPT < 32) ok &= (1ull << WPT) - 1;
    r.valid = (uint32_t)ok;
    return r;
}

// What the split kernel sorts: WPT keys of at most 26 bits per lane and tile, with a validity bit each.
struct Src13 {                                        // the 13-mer windows of a PLAIN reads buffer (count_kmers13)
    const uint8_t* buf;
    uint64_t len;
    template <int TB, int WPT>
    __device__ __forceinline__ Raw13<WPT> fetch(uint64_t tile, int t) const { return fetch_raw13<WPT>(buf, len, tile * (uint64_t)(TB * WPT) + (uint64_t)t * WPT); }
    template <int TB, int WPT>
    __device__ __forceinline__ Run13 decode(const Raw13<WPT>& raw) const { return encode_raw13<WPT>(raw); }
};
template <int TB>
struct RunSlots {                                     // element j of this lane = p[j * TB]: every load instruction of the workgroup reads 4 * TB bytes of
    const uint32_t* p;                                // consecutive slots. Nothing is kept in registers between the two passes of the tile sort
    uint32_t limit;                                   // (WPT more live registers spill): the second pass re-reads the tile through L2
    uint32_t base, range;                             // this pass adds the slots of [base, base + range), range <= 2^26; the others are somebody else's
    template <int J>
    __device__ __forceinline__ void get(uint32_t& c, uint32_t& ok) const {
        const uint32_t v = (uint32_t)(J * TB) < limit ? p[J * TB] : 0xFFFFFFFFu;
        const uint32_t r = v - base;
        c = r & 0x3FFFFFFu;
        ok = (v != 0xFFFFFFFFu && r < range) ? 1u : 0u;
    }
};
struct SrcSlots {                                     // a stream of MPHF slots in HBM, 0xFFFFFFFF = nothing to count (count23); one pass per 2^26 slots of the key set
    const uint32_t* slots;
    uint64_t n;
    uint32_t base, range;
    template <int TB, int WPT>
    __device__ __forceinline__ RunSlots<TB> fetch(uint64_t tile, int t) const {  // nothing is fetched ahead: the passes read the stream themselves
        const uint64_t base = tile * (uint64_t)(TB * WPT) + (uint64_t)t;
        RunSlots<TB> r;
        r.p = slots + base;
        r.limit = base < n ? (uint32_t)min((uint64_t)(TB * WPT), n - base) : 0u;  // elements [0, limit) of p are inside the stream
        r.base = this->base;
        r.range = range;
        return r;
    }
    template <int TB, int WPT>
    __device__ __forceinline__ RunSlots<TB> decode(const RunSlots<TB>& r) const { return r; }
};

// The same stream with the tile held in registers between the two passes of the tile sort — the default since round 3 (AIX_C23_SLOTS_REGS=0: the
// re-reading source above, A/B). One read of the stream instead of two: 1.54 against 3.53 ms per 6.9e8 slots on one box (the second pass's 32
// strided loads per lane came from L2 / MALL at best — 256 resident tiles are 32 MiB — and every one was waited for); 128 VGPRs with 4 spilled words.
template <int TB, int WPT>
struct RunSlotsR {
    uint32_t v[WPT];                                  // element j of this lane = slots[tile * TB * WPT + j * TB + t], already reduced: slot - base, or 0xFFFFFFFF
    template <int J>
    __device__ __forceinline__ void get(uint32_t& c, uint32_t& ok) const {
        c = v[J] & 0x3FFFFFFu;
        ok = v[J] != 0xFFFFFFFFu ? 1u : 0u;
    }
};
struct SrcSlotsR {
    const uint32_t* slots;
    uint64_t n;
    uint32_t base, range;
    template <int TB, int WPT>
    __device__ __forceinline__ RunSlotsR<TB, WPT> fetch(uint64_t tile, int t) const {
        const uint64_t first = tile * (uint64_t)(TB * WPT) + (uint64_t)t;
        const uint32_t limit = first < n ? (uint32_t)min((uint64_t)(TB * WPT), n - first) : 0u;
        const uint32_t* p = slots + first;
        RunSlotsR<TB, WPT> r;
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const uint32_t x = (uint32_t)(j * TB) < limit ? p[j * TB] : 0xFFFFFFFFu;
            const uint32_t d = x - base;
            r.v[j] = (x != 0xFFFFFFFFu && d < range) ? d : 0xFFFFFFFFu;
        }
        return r;
    }
    template <int TB, int WPT>
    __device__ __forceinline__ RunSlotsR<TB, WPT> decode(const RunSlotsR<TB, WPT>& r) const { return r; }
};

// ---------------------------------------------------------------------------------------------
// Chunked partitions: a workgroup's share of a partition is a list of 256-entry chunks, so nothing has to be sized
// before the split (an earlier version ran a separate sizing pass — encode + 1.4e9 LDS atomics + a column scan, 1.2 ms per
// 10 M reads — to give every workgroup an exact, contiguous segment of every partition). A directory (partition, fill)
// per chunk, sorted by partition with one small radix sort, tells the histogram kernel which chunks belong to it.
// ---------------------------------------------------------------------------------------------
static constexpr int C13_CH = 256;                    // entries per chunk (512 B)
static constexpr int C13_INFLIGHT = 8;                // chunk loads a wave of the histogram kernel keeps in flight
static constexpr int C13_FILLBITS = 9;                // cursor = (chunk << 9) | fill, fill in [0, 256]; chunk = index inside the workgroup's region (< 2^23)

// Chunk ids need no global allocator: a workgroup that sorts T tiles fills at most T * TILE / 256 + 2048 chunks (entries / 256
// plus one partly filled chunk per partition), so workgroup b owns chunk ids [b * region, (b + 1) * region) and hands them
// out with the same block scan that orders the tile (the per-partition chunk demand rides in the high half of the scan).
// Chunk ids are handed out contiguously, so "every fresh chunk of this tile lies inside the region" is ONE uniform test per tile;
// a tile that fails it is dropped whole AND `err` is raised: the call fails with AIX_ERR_UNSUPPORTED instead of returning short counts.
template <class SRC, int TB, int WPT, bool PREFETCH>
__global__ void __launch_bounds__(TB) k_c13_split_chunked(const SRC src, uint64_t ntiles, uint32_t region,
                                                         uint16_t* __restrict__ dir_part, uint16_t* __restrict__ dir_cnt, uint16_t* __restrict__ parts,
                                                         uint32_t* __restrict__ err) {
    constexpr int TILE = TB * WPT;
    constexpr int PPL = C13_P / TB;                             // partitions a lane owns: PPL * t .. PPL * t + PPL - 1
    static_assert(C13_P % TB == 0 && TB % 64 == 0 && TILE <= 32768 && TILE / C13_CH + C13_P < 65536 && WPT % 2 == 0, "one packed 16 + 16 bit scan per tile");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* hist = (uint32_t*)smem;                           // [P + DUMMY] tile-local count per partition; after the scan: write-out delta A of the partition
    uint32_t* loc_t = hist + C13_P + C13_DUMMY;                 // [P] exclusive scan of hist (low half) | first tile index that no longer fits the current chunk (high half)
    uint32_t* delta_b = loc_t + C13_P;                          // [P] write-out delta B of the partition (entries that spill into its fresh chunks)
    uint32_t* wsum = delta_b + C13_P;                           // [16] per-wave totals: entries (low 16 bits, <= 32768), new chunks (high 16 bits, <= 2176)
    uint32_t* sorted = wsum + 16;                               // [TILE] (partition << 16) | low 15 bits, grouped by partition
    const int t = threadIdx.x;
    constexpr uint32_t FILLMASK = (1u << C13_FILLBITS) - 1;
    const uint32_t region_base = blockIdx.x * region;           // chunk ids below are relative to it
    uint32_t next_chunk = 0;                                    // same value in every lane
    // the cursors of the partitions this lane owns, (current chunk << 9) | entries used in it (256 = none / full), never leave its registers
    uint32_t cur[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) { cur[q] = C13_CH; hist[PPL * t + q] = 0; }
    if (t < C13_DUMMY) hist[C13_P + t] = 0;
    __syncthreads();
    auto raw = src.template fetch<TB, WPT>(blockIdx.x, t);
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t rank[WPT / 2];                                  // two 16-bit ranks per register
        if constexpr (!PREFETCH) raw = src.template fetch<TB, WPT>(tile, t);
        const auto run = src.template decode<TB, WPT>(raw);
        const uint32_t dummy = C13_P + (t & (C13_DUMMY - 1));
        for_each_window13<WPT, 0>(run, [&](int j, uint32_t code, uint32_t ok) {
            const uint32_t r = atomicAdd(&hist[ok ? code >> C13_BINBITS : dummy], 1u);
            if (j & 1) rank[j >> 1] |= r << 16; else rank[j >> 1] = r;
        });
        __syncthreads();
        // entries a[q] of the partitions this lane owns; fresh chunks k[q] = by how much they overflow the current chunk
        uint32_t a[PPL], f[PPL], tot[PPL], k[PPL], s = 0;
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            a[q] = hist[PPL * t + q];
            f[q] = cur[q] & FILLMASK;
            tot[q] = f[q] + a[q];
            k[q] = tot[q] > (uint32_t)C13_CH ? (tot[q] - 1) / C13_CH : 0u;       // = ceil((tot - CH) / CH)
            s += a[q] | (k[q] << 16);                            // one scan for both: a tile has <= 32768 entries and needs <= 2176 fresh chunks
        }
        const uint32_t mine = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(s, d);
            if ((t & 63) >= d) s += y;
        }
        if ((t & 63) == 63) wsum[t >> 6] = s;
        __syncthreads();                                         // every lane has read its counters: hist may be overwritten
        uint32_t off = 0, all = 0;
#pragma unroll
        for (int w = 0; w < TB / 64; ++w) {
            const uint32_t x = wsum[w];
            if (w < (t >> 6)) off += x;
            all += x;
        }
        const uint32_t entries = all & 0xFFFFu;
        const bool fits = next_chunk + (all >> 16) <= region;
        if (!fits && t == 0) *err = 1u;
        uint32_t excl = off + s - mine;
        // where the entries of a partition go, as two deltas to their index i inside the sorted tile: i < T lands in the current
        // chunk at A + i, the rest in the fresh chunks (consecutive ids) at B + i
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const uint32_t p = PPL * t + q, e0 = excl & 0xFFFFu, nb = next_chunk + (excl >> 16);
            loc_t[p] = e0 | ((e0 + C13_CH - f[q]) << 16);        // T <= 32768 + 256
            hist[p] = (cur[q] >> C13_FILLBITS) * C13_CH + f[q] - e0;
            delta_b[p] = nb * C13_CH + f[q] - e0 - C13_CH;
            if (fits) {
                for (uint32_t i = 0; i < k[q]; ++i) dir_part[region_base + nb + i] = (uint16_t)p;      // fill stays at the pre-set 256 unless it ends up the last one
                cur[q] = k[q] ? ((nb + k[q] - 1) << C13_FILLBITS) | (tot[q] - k[q] * C13_CH) : (cur[q] & ~FILLMASK) | tot[q];
            }
            excl += a[q] | (k[q] << 16);
        }
        if (fits) next_chunk += all >> 16;
        __syncthreads();                                         // the per-partition words are complete
        for_each_window13<WPT, 0>(run, [&](int j, uint32_t code, uint32_t ok) {
            if (ok) {
                const uint32_t p = code >> C13_BINBITS;
                const uint32_t r = (j & 1) ? (rank[j >> 1] >> 16) : (rank[j >> 1] & 0xFFFFu);
                sorted[(loc_t[p] & 0xFFFFu) + r] = (p << 16) | (code & (C13_BINS - 1));
            }
        });
        __syncthreads();
        if constexpr (PREFETCH) { if (tile + gridDim.x < ntiles) raw = src.template fetch<TB, WPT>(tile + gridDim.x, t); }      // in flight while this tile is written out
        if (fits) {
            uint16_t* const out = parts + (uint64_t)region_base * C13_CH;
            for (uint32_t i = t; i < entries; i += TB) {
                const uint32_t e = sorted[i];
                const uint32_t p = e >> 16;
                const uint32_t d = i < (loc_t[p] >> 16) ? hist[p] : delta_b[p];
                out[d + i] = (uint16_t)e;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PPL; ++q) hist[PPL * t + q] = 0;
        if (t < C13_DUMMY) hist[C13_P + t] = 0;
        __syncthreads();
    }
    // the chunk each (workgroup, partition) pair was still filling is the only one that is not full
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        const uint32_t fl = cur[q] & FILLMASK;
        if (fl < (uint32_t)C13_CH) dir_cnt[region_base + (cur[q] >> C13_FILLBITS)] = (uint16_t)fl;
    }
}

// One workgroup per partition: its chunks are sdesc[lo, hi) of the directory sorted by partition (descriptor = chunk id in
// the low half, entries in the high half). A wave fetches 64 descriptors with one coalesced load and then walks them with
// v_readlane, eight 512-byte chunk loads in flight (64 KiB per CU), so no load depends on another one inside the loop.
__global__ void __launch_bounds__(C13_TB) k_c13_hist_chunked(const uint16_t* __restrict__ parts, const uint16_t* __restrict__ spart /* sorted partition ids */,
                                                            const uint64_t* __restrict__ sdesc, uint32_t cap, unsigned long long* __restrict__ table,
                                                            const uint32_t* __restrict__ perm, uint64_t* __restrict__ out_mphf, int accumulate,
                                                            uint32_t* __restrict__ out32 = nullptr, uint64_t n32 = 0) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t range[2];
    uint32_t* h = (uint32_t*)smem;                              // [BINS]
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t p = blockIdx.x; p < (uint32_t)C13_P; p += gridDim.x) {
        if (wave == 0) {                                        // every other wave of the workgroup waits for this at the barrier below
            const uint32_t r = wave_lower_bound_pair(spart, cap, p);
            if ((lane & 31u) == 0) range[lane >> 5] = r;
        }
        for (int i = threadIdx.x; i < C13_BINS; i += C13_TB) h[i] = 0;
        __syncthreads();
        const uint32_t lo = range[0], hi = range[1];
        for (uint32_t c0 = lo + wave * 64; c0 < hi; c0 += (C13_TB / 64) * 64) {
            const uint64_t desc = (c0 + lane < hi) ? sdesc[c0 + lane] : 0ull;
            const uint32_t d_id = (uint32_t)desc, d_n = (uint32_t)(desc >> 32);
            const uint32_t nch = min(64u, hi - c0);
            for (uint32_t u0 = 0; u0 < nch; u0 += C13_INFLIGHT) {
                uint2 x[C13_INFLIGHT];
                uint32_t n[C13_INFLIGHT];
#pragma unroll
                for (int u = 0; u < C13_INFLIGHT; ++u) {
                    const uint32_t j = min(u0 + u, 63u);
                    const uint32_t id = __builtin_amdgcn_readlane(d_id, j);
                    n[u] = (u0 + u < nch) ? (uint32_t)__builtin_amdgcn_readlane(d_n, j) : 0u;      // lanes past hi hold n = 0 anyway
                    x[u] = ((const uint2*)(parts + (uint64_t)id * C13_CH))[lane];                  // id = 0 for padding: a valid chunk, ignored through n = 0
                }
#pragma unroll
                for (int u = 0; u < C13_INFLIGHT; ++u) {
                    const uint32_t base = lane * 4;
                    if (base + 0 < n[u]) atomicAdd(&h[x[u].x & 0xFFFFu], 1u);
                    if (base + 1 < n[u]) atomicAdd(&h[x[u].x >> 16], 1u);
                    if (base + 2 < n[u]) atomicAdd(&h[x[u].y & 0xFFFFu], 1u);
                    if (base + 3 < n[u]) atomicAdd(&h[x[u].y >> 16], 1u);
                }
            }
        }
        __syncthreads();
        const uint64_t base = (uint64_t)p * C13_BINS;
        if (out32) {                                            // count23: bins are MPHF slots, tf_out[slot] += count (one writer per slot)
            for (int i = threadIdx.x; i < C13_BINS; i += C13_TB) {
                const uint32_t c = h[i];
                if (c && base + i < n32) out32[base + i] += c;
            }
        } else if (perm) {
            for (int i = threadIdx.x; i < C13_BINS; i += C13_TB) {
                const uint32_t c = h[i];
                if (c) { const uint32_t slot = perm[base + i]; if (slot < 67108864u) out_mphf[slot] = accumulate ? out_mphf[slot] + c : (uint64_t)c; }
            }
        } else {
            for (int i = threadIdx.x; i < C13_BINS; i += C13_TB) table[base + i] = accumulate ? table[base + i] + h[i] : (unsigned long long)h[i];
        }
        __syncthreads();
    }
}

// directory sort values: chunk id | entries << 32, generated on the fly from the counting iterator
struct ChunkDesc {
    const uint16_t* dir_cnt;
    __host__ __device__ uint64_t operator()(uint32_t i) const { return (uint64_t)i | ((uint64_t)dir_cnt[i] << 32); }
};

// shape of the split kernel for this process (A/B switch AIX_C13_SHAPE=big|small): tile size, workgroups that are resident at once
struct C13Shape { int tb, wpt; unsigned maxgrid; };
static inline C13Shape c13_shape() {
    static const C13Shape sh = [] {
        C13Shape v{1024, 32, 256};                                // one 152 KiB workgroup per CU, one per CU in the grid: measured faster than two smaller ones and than 512..1024 workgroups (4.23 / 4.28 / 4.32 / 4.33 ms at 256 / 512 / 768 / 1024)
        if (const char* e = getenv("AIX_C13_SHAPE")) { if (e[0] == 's') v = C13Shape{512, 24, 1024}; }
        if (const char* e = getenv("AIX_C13_GRID")) { const int g = atoi(e); if (g >= 1 && g <= 4096) v.maxgrid = (unsigned)g; }      // A/B switch: workgroups of the split
        return v;
    }();
    return sh;
}

static inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }
// chunk ids one workgroup can need: TILE / 256 per tile it sorts + one partly filled chunk per partition
static inline uint32_t chunk_region(uint64_t nwin) {
    if (const char* e = getenv("AIX_COUNT13_TEST_REGION")) { const long v = atol(e); if (v > 0) return (uint32_t)v; }   // test hook: an undersized region must fail loudly
    const C13Shape sh = c13_shape();
    const uint64_t tile = (uint64_t)sh.tb * sh.wpt;
    const uint64_t ntiles = (nwin + tile - 1) / tile;
    const uint64_t grid = std::min<uint64_t>(ntiles ? ntiles : 1, sh.maxgrid);
    return (uint32_t)((ntiles + grid - 1) / grid * ((tile + C13_CH - 1) / C13_CH) + C13_P);
}
static inline uint32_t chunk_capacity(uint64_t nwin) {
    const C13Shape sh = c13_shape();
    const uint64_t tile = (uint64_t)sh.tb * sh.wpt;
    const uint64_t ntiles = (nwin + tile - 1) / tile;
    return (uint32_t)(std::min<uint64_t>(ntiles ? ntiles : 1, sh.maxgrid) * chunk_region(nwin));
}
static size_t dir_sort_temp_bytes(uint32_t cap) {
    size_t bytes = 0;
    auto vals = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), ChunkDesc{nullptr});
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint16_t*)nullptr, (uint16_t*)nullptr, vals, (uint64_t*)nullptr, (size_t)cap, 0u, 12u, (hipStream_t)0);
    return bytes;
}

// workspace (chunked path): error word (first 4 of 256 B) | dir_part u16[cap] | dir_cnt u16[cap] | spart u16[cap] | sdesc u64[cap] | sort temp | parts u16[cap * 256]
// `len` = bytes handed to one launch (<= 2^31 + 12)
uint64_t count13_workspace_bytes(uint64_t len) {
    const uint64_t nwin = len >= 13 ? len - 12 : 0;
    const uint64_t cap = chunk_capacity(nwin);
    return 256 + 3 * align_up(2 * cap, 256) + align_up(8 * cap, 256) + align_up(dir_sort_temp_bytes((uint32_t)cap), 256) + 2 * cap * C13_CH + 256;
}

template <class SRC, int TB, int WPT>
static hipError_t launch_split(const SRC& src, uint64_t ntiles, unsigned grid, uint32_t region, uint16_t* dir_part, uint16_t* dir_cnt, uint16_t* parts, uint32_t* err,
                               hipStream_t s) {
    constexpr size_t lds = 4 * (3 * C13_P + C13_DUMMY + 16) + 4 * (size_t)TB * WPT;       // 155 968 B (1024 x 32) / 74 048 B (512 x 24)
    // the next tile's loads in flight during the write-out: measured SLOWER for the 13-mer source (3.45 against 3.10 ms per 10 M reads: eleven more
    // live registers put the kernel at its 128-VGPR limit), so it is off unless asked for (A/B switch); requested right after the decode instead
    // (older than every store of the write-out): 4.27-4.42 against 4.11-4.19 ms per call, round 3. The slot source has nothing to fetch ahead
    bool prefetch = false;
    if (const char* pf = getenv("AIX_C13_PREFETCH")) prefetch = atoi(pf) != 0;
    // > 64 KiB of dynamic LDS needs the attribute; set per call (cheap, and correct for every device / thread)
    hipError_t e = prefetch ? hipFuncSetAttribute((const void*)k_c13_split_chunked<SRC, TB, WPT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                            : hipFuncSetAttribute((const void*)k_c13_split_chunked<SRC, TB, WPT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (prefetch) hipLaunchKernelGGL((k_c13_split_chunked<SRC, TB, WPT, true>), dim3(grid), dim3(TB), lds, s, src, ntiles, region, dir_part, dir_cnt, parts, err);
    else hipLaunchKernelGGL((k_c13_split_chunked<SRC, TB, WPT, false>), dim3(grid), dim3(TB), lds, s, src, ntiles, region, dir_part, dir_cnt, parts, err);
    return hipGetLastError();
}

template <class SRC>
static hipError_t partitioned_histogram(const SRC& src, uint64_t nwin, void* workspace, unsigned long long* table, const uint32_t* perm, uint64_t* out_mphf,
                                        int accumulate, uint32_t* out32, uint64_t n32, hipStream_t s) {
    if (nwin > (1ull << 31)) return hipErrorInvalidValue;
    const uint32_t cap = chunk_capacity(nwin);
    uint8_t* w = (uint8_t*)workspace;
    uint32_t* err = (uint32_t*)w;                             // sticky inside one call: the caller zeroes it before the first piece and reads it after the last
    w += 256;
    uint16_t* dir_part = (uint16_t*)w;                        w += align_up(2ull * cap, 256);
    uint16_t* dir_cnt = (uint16_t*)w;                         w += align_up(2ull * cap, 256);
    uint16_t* spart = (uint16_t*)w;                           w += align_up(2ull * cap, 256);
    uint64_t* sdesc = (uint64_t*)w;                           w += align_up(8ull * cap, 256);
    size_t tmp_bytes = dir_sort_temp_bytes(cap);
    void* tmp = w;                                            w += align_up(tmp_bytes, 256);
    uint16_t* parts = (uint16_t*)w;
    const C13Shape sh = c13_shape();
    const uint64_t tile = (uint64_t)sh.tb * sh.wpt;
    const uint64_t ntiles = (nwin + tile - 1) / tile;
    const unsigned grid = (unsigned)std::min<uint64_t>(ntiles ? ntiles : 1, sh.maxgrid);
    const size_t hist_lds = 4 * C13_BINS;                                       // 131 072 B
    hipError_t e = hipFuncSetAttribute((const void*)k_c13_hist_chunked, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_lds);
    if (e == hipSuccess) e = hipMemsetD16Async((hipDeviceptr_t)dir_part, (unsigned short)C13_P, cap, s);     // "no partition": sorts behind every real one
    if (e == hipSuccess) e = hipMemsetD16Async((hipDeviceptr_t)dir_cnt, (unsigned short)C13_CH, cap, s);     // chunks are full unless the split says otherwise
    if (e != hipSuccess) return e;
    if (sh.tb == 512) e = launch_split<SRC, 512, 24>(src, ntiles, grid, chunk_region(nwin), dir_part, dir_cnt, parts, err, s);
    else e = launch_split<SRC, 1024, 32>(src, ntiles, grid, chunk_region(nwin), dir_part, dir_cnt, parts, err, s);
    if (e != hipSuccess) return e;
    auto vals = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), ChunkDesc{dir_cnt});
    e = rocprim::radix_sort_pairs(tmp, tmp_bytes, (const uint16_t*)dir_part, spart, vals, sdesc, (size_t)cap, 0u, 12u, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_c13_hist_chunked, dim3(C13_P), dim3(C13_TB), hist_lds, s, parts, spart, sdesc, cap, table, perm, out_mphf, accumulate, out32, n32);
    return hipGetLastError();
}

hipError_t launch_count13_partitioned(const uint8_t* buf, uint64_t len, void* workspace, unsigned long long* table, const uint32_t* perm,
                                      uint64_t* out_mphf, int accumulate, hipStream_t s) {
    const uint64_t nwin = len >= 13 ? len - 12 : 0;
    return partitioned_histogram(Src13{buf, len}, nwin, workspace, table, perm, out_mphf, accumulate, nullptr, 0, s);
}

// count23 back end: tf_out[slot] += occurrences of slot in d_slots[0, nslots) (0xFFFFFFFF entries are skipped). The partitions hold 26-bit
// values (2 048 partitions of 32 768 bins), so a key set of more than `range` (2^26) slots takes one pass over the slot stream per range of
// slots — the stream is 4 bytes per window and a pass only sorts the windows of its own range, so P passes cost little more than one.
// Same workspace layout as the 13-mer counter (count13_workspace_bytes(nslots + 12)). *passes_out: passes made.
hipError_t launch_histogram_slots(const uint32_t* d_slots, uint64_t nslots, void* workspace, uint32_t* tf_out, uint64_t n, hipStream_t s, uint32_t range_bits,
                                  uint32_t* passes_out) {
    if (passes_out) *passes_out = 0;
    if (nslots == 0 || n == 0) return hipSuccess;
    if (range_bits < 4 || range_bits > 26) range_bits = 26;
    const uint64_t range = 1ull << range_bits;
    for (uint64_t base = 0; base < n; base += range) {
        const uint64_t m = std::min(range, n - base);
        static const int in_regs = [] { const char* e = getenv("AIX_C23_SLOTS_REGS"); return e ? atoi(e) : 1; }();
        const hipError_t e = in_regs ? partitioned_histogram(SrcSlotsR{d_slots, nslots, (uint32_t)base, (uint32_t)m}, nslots, workspace, nullptr, nullptr, nullptr, 1, tf_out + base, m, s)
                                     : partitioned_histogram(SrcSlots{d_slots, nslots, (uint32_t)base, (uint32_t)m}, nslots, workspace, nullptr, nullptr, nullptr, 1, tf_out + base, m, s);
        if (e != hipSuccess) return e;
        if (passes_out) ++*passes_out;
    }
    return hipSuccess;
}

}  // namespace aix
