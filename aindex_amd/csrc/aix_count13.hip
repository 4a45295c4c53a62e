// aix_count13.hip — 13-mer dense counting without global atomics (K13 row; count_kmers13.cpp:131-161).
//
// The 4^13 counter table (256 MiB as u32) is 1600x the LDS of a CU and 1.4e9 scattered memory-side atomics
// run at ~23 G/s on MI355X (60 ms for 10 M reads). This path is HBM-streaming bound instead:
//   P1 k_c13_sizes   : rolling 2-bit encode (32 window starts per lane), LDS histogram of the top 11 code
//                      bits -> windows per (workgroup, partition); k_c13_colscan / k_c13_scan turn them into a
//                      private, contiguous segment of every partition for every workgroup
//   P2 k_c13_split   : same encode; a 32 768-window tile is counting-sorted by partition inside LDS and appended to
//                      the workgroup's segment of each partition as coalesced runs of the low 15 code bits (u16)
//   P3 k_c13_hist    : one workgroup per partition: 32 768 u32 counters in 128 KiB of LDS, ds_add per element,
//                      counters stored (u64) to the code-ordered table — every bin written exactly once
// then k_scatter13 permutes the table into the reference's mphf order.
#include <algorithm>

#include "aix_internal.hpp"

namespace aix {

static constexpr int C13_PBITS = 11;
static constexpr int C13_P = 1 << C13_PBITS;          // partitions
static constexpr int C13_BINBITS = 26 - C13_PBITS;    // 15
static constexpr int C13_BINS = 1 << C13_BINBITS;     // 32768 bins per partition
static constexpr int C13_TB = 1024;                   // threads per workgroup
static constexpr int C13_WPT = 32;                    // window starts per lane
static constexpr int C13_TILE = C13_TB * C13_WPT;     // 32768 window starts per tile
static constexpr int C13_DUMMY = 64;                  // scratch histogram bins for windows that do not count

// Encode the C13_WPT windows whose starts are [S, S+WPT): the packed 2-bit stream (Run13::code<j>() = 26 bits) and a validity bit mask.
// Bytes at positions >= len count as separators. Upper-casing and the ACGT test follow
// normalize_sequence / is_valid_kmer (count_kmers13.cpp:100-126).
struct Run13 {
    uint32_t w[3];        // 2-bit stream of the 44 bytes, base 0 in the top bits of w[0]; w[2] holds bases 32..43 in its top 24 bits
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
};
template <int J, typename F>
__device__ __forceinline__ void for_each_window13(const Run13& r, F&& f) {
    if constexpr (J < C13_WPT) {
        f(J, r.template code<J>(), (r.valid >> J) & 1u);
        for_each_window13<J + 1>(r, f);
    }
}

__device__ __forceinline__ Run13 encode_run13(const uint8_t* __restrict__ buf, uint64_t len, uint64_t S) {
    constexpr int NB = C13_WPT + 12;                  // 44 bytes
    constexpr int ND = NB / 4;                        // 11 aligned dwords
    Run13 r{{0, 0, 0}, 0};
    if (S >= len) return r;
    const uint64_t limit = len - S;                   // bytes available from S
    uint32_t e[ND];
    {
        const uint32_t o = (uint32_t)((uintptr_t)(buf + S) & 3);
        const uint32_t* q = (const uint32_t*)(buf + S - o);           // pointer arithmetic keeps these global_load (not flat_load)
        const int64_t first = (int64_t)S - o, end = (int64_t)len;     // byte offset of d[0] from buf (>= -3)
        uint32_t d[ND + 1];
#pragma unroll
        for (int k = 0; k <= ND; ++k) d[k] = (first + 4 * k < end) ? q[k] : 0x0A0A0A0Au;   // never read a dword past the buffer
        const uint32_t sh = o * 8;
#pragma unroll
        for (int k = 0; k < ND; ++k) e[k] = __funnelshift_r(d[k], d[k + 1], sh);
    }
    // Four bytes at a time (the VALU issues one wave instruction per clock per CU, and the byte-serial form of this loop
    // was ~2/3 of the kernel): x = upper-cased bytes; v = 2-bit value of every byte; a byte is a base iff it equals the
    // letter its own 2-bit value maps back to (v_perm_b32 as the 4-entry table); q = the four values packed first-base-high.
    uint64_t vmask = 0;                               // bit i: byte i is A/C/G/T (either case)
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        const uint32_t x = e[k] & 0xDFDFDFDFu;
        const uint32_t v = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
        const uint32_t diff = x ^ lut4(v, AIX_LUT_ACGT);
        const uint32_t z = ~(((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff | 0x7F7F7F7Fu);      // 0x80 in every byte where diff == 0
        const uint32_t f = z >> 7;
        const uint32_t nib = (f | (f >> 7) | (f >> 14) | (f >> 21)) & 0xFu;
        const uint32_t q = ((v << 6) & 0xC0u) | ((v >> 4) & 0x30u) | ((v >> 14) & 0x0Cu) | (v >> 24);
        r.w[k >> 2] |= q << (24 - 8 * (k & 3));
        vmask |= (uint64_t)nib << (4 * k);
    }
    if (limit < (uint64_t)NB) vmask &= (1ull << limit) - 1;           // bytes at and past the end of the buffer are separators
    // window j (bytes j..j+12) is countable iff 13 consecutive mask bits are set
    const uint64_t m2 = vmask & (vmask >> 1), m4 = m2 & (m2 >> 2), m8 = m4 & (m4 >> 4);
    r.valid = (uint32_t)(m8 & (m4 >> 8) & (vmask >> 12));
    return r;
}

// P1: per-(workgroup, partition) window counts. Workgroup b owns tiles b, b+G, b+2G, ... in P1 and in P2 alike.
__global__ void __launch_bounds__(C13_TB) k_c13_sizes(const uint8_t* __restrict__ buf, uint64_t len, uint64_t ntiles, uint32_t* __restrict__ cnt /* [G][P] */) {
    __shared__ uint32_t hist[C13_P + C13_DUMMY];      // windows that do not count go to one of 64 scratch bins (no branch per window)
    for (int i = threadIdx.x; i < C13_P + C13_DUMMY; i += C13_TB) hist[i] = 0;
    __syncthreads();
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const Run13 run = encode_run13(buf, len, t * C13_TILE + (uint64_t)threadIdx.x * C13_WPT);
        const uint32_t dummy = C13_P + (threadIdx.x & (C13_DUMMY - 1));
        for_each_window13<0>(run, [&](int, uint32_t code, uint32_t ok) { atomicAdd(&hist[ok ? code >> C13_BINBITS : dummy], 1u); });
    }
    __syncthreads();
    uint32_t* row = cnt + (uint64_t)blockIdx.x * C13_P;
    for (int i = threadIdx.x; i < C13_P; i += C13_TB) row[i] = hist[i];
}

// column scan: cnt[b][p] -> exclusive prefix over b (in place), part_count[p] = column total. One lane per partition.
__global__ void __launch_bounds__(C13_TB) k_c13_colscan(uint32_t* __restrict__ cnt, uint32_t G, unsigned long long* __restrict__ part_count) {
    const uint32_t p = blockIdx.x * C13_TB + threadIdx.x;
    if (p >= (uint32_t)C13_P) return;
    uint32_t run = 0;
    for (uint32_t b = 0; b < G; ++b) {
        const uint32_t c = cnt[(uint64_t)b * C13_P + p];
        cnt[(uint64_t)b * C13_P + p] = run;
        run += c;
    }
    part_count[p] = run;
}

// part_base[0..P] = exclusive scan of part_count. One workgroup.
__global__ void __launch_bounds__(C13_TB) k_c13_scan(const unsigned long long* __restrict__ part_count, unsigned long long* __restrict__ part_base) {
    __shared__ unsigned long long wsum[C13_TB / 64];
    const int t = threadIdx.x;
    const unsigned long long a = part_count[2 * t], b = part_count[2 * t + 1];
    unsigned long long s = a + b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long y = __shfl_up(s, d);
        if ((t & 63) >= d) s += y;
    }
    if ((t & 63) == 63) wsum[t >> 6] = s;
    __syncthreads();
    unsigned long long off = 0;
    for (int w = 0; w < (t >> 6); ++w) off += wsum[w];
    const unsigned long long excl = off + s - (a + b);
    part_base[2 * t] = excl;
    part_base[2 * t + 1] = excl + a;
    if (t == C13_TB - 1) part_base[C13_P] = off + s;
}

// P2: every workgroup appends to its own private segment of every partition (bases from P1), so no global atomics
// are needed. A 32 768-window tile is counting-sorted by partition inside LDS (rank from one ds_add_rtn, packed
// {partition, low 15 bits} entries), then written out as coalesced u16 runs. (Direct 2-byte scattered stores from the
// lanes, without the LDS sort, were measured at 12.1 ms for this kernel against 4.4 ms with it.)
__global__ void __launch_bounds__(C13_TB) k_c13_split(const uint8_t* __restrict__ buf, uint64_t len, uint64_t ntiles, const unsigned long long* __restrict__ part_base,
                                                     const uint32_t* __restrict__ cnt /* [G][P] exclusive over workgroups */, uint16_t* __restrict__ parts) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* hist = (uint32_t*)smem;                           // [P] tile-local count per partition
    uint32_t* loc_off = hist + C13_P + C13_DUMMY;               // [P] exclusive scan of hist (hist has C13_DUMMY scratch bins behind it)
    uint32_t* cursor = loc_off + C13_P;                         // [P] next free slot of this workgroup's segment (absolute)
    uint32_t* wsum = cursor + C13_P;                            // [16]
    uint32_t* sorted = wsum + 16;                               // [TILE] (partition << 16) | low 15 bits, grouped by partition
    const int t = threadIdx.x;
    {
        const uint32_t* row = cnt + (uint64_t)blockIdx.x * C13_P;
        cursor[2 * t] = (uint32_t)part_base[2 * t] + row[2 * t];            // nwin < 2^32
        cursor[2 * t + 1] = (uint32_t)part_base[2 * t + 1] + row[2 * t + 1];
        hist[2 * t] = 0;
        hist[2 * t + 1] = 0;
        if (t < C13_DUMMY) hist[C13_P + t] = 0;
    }
    __syncthreads();
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t rank[C13_WPT / 2];                              // two 16-bit ranks per register
        const Run13 run = encode_run13(buf, len, tile * C13_TILE + (uint64_t)t * C13_WPT);
        const uint32_t dummy = C13_P + (t & (C13_DUMMY - 1));    // windows that do not count take a rank from a scratch bin: no branch
        for_each_window13<0>(run, [&](int j, uint32_t code, uint32_t ok) {
            const uint32_t r = atomicAdd(&hist[ok ? code >> C13_BINBITS : dummy], 1u);   // < 2^16 (scratch bins: <= 512 per tile)
            if (j & 1) rank[j >> 1] |= r << 16; else rank[j >> 1] = r;
        });
        __syncthreads();
        const uint32_t a = hist[2 * t], b = hist[2 * t + 1];
        {   // exclusive scan of hist[2048]
            uint32_t s = a + b;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(s, d);
                if ((t & 63) >= d) s += y;
            }
            if ((t & 63) == 63) wsum[t >> 6] = s;
            __syncthreads();
            uint32_t off = 0;
            for (int w = 0; w < (t >> 6); ++w) off += wsum[w];
            const uint32_t excl = off + s - (a + b);
            loc_off[2 * t] = excl;
            loc_off[2 * t + 1] = excl + a;
        }
        __syncthreads();
        for_each_window13<0>(run, [&](int j, uint32_t code, uint32_t ok) {
            if (ok) {
                const uint32_t p = code >> C13_BINBITS;
                const uint32_t r = (j & 1) ? (rank[j >> 1] >> 16) : (rank[j >> 1] & 0xFFFFu);
                sorted[loc_off[p] + r] = (p << 16) | (code & (C13_BINS - 1));
            }
        });
        __syncthreads();
        const uint32_t total = loc_off[C13_P - 1] + hist[C13_P - 1];
        for (uint32_t i = t; i < total; i += C13_TB) {
            const uint32_t e = sorted[i];
            const uint32_t p = e >> 16;
            parts[cursor[p] + (i - loc_off[p])] = (uint16_t)e;
        }
        __syncthreads();
        cursor[2 * t] += a;                                      // this lane owns partitions 2t, 2t+1
        cursor[2 * t + 1] += b;
        hist[2 * t] = 0;
        hist[2 * t + 1] = 0;
        if (t < C13_DUMMY) hist[C13_P + t] = 0;
        __syncthreads();
    }
}

// One workgroup per partition. `perm` != nullptr: fused permutation — non-zero counters go straight to the
// mphf-ordered output (pre-zeroed by the caller): out[perm[code]] = count. Otherwise every bin of the
// code-ordered table is stored.
__global__ void __launch_bounds__(C13_TB) k_c13_hist(const uint16_t* __restrict__ parts, const unsigned long long* __restrict__ part_base,
                                                    unsigned long long* __restrict__ table, const uint32_t* __restrict__ perm, uint64_t* __restrict__ out_mphf,
                                                    int accumulate) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* h = (uint32_t*)smem;                              // [BINS]
    for (uint32_t p = blockIdx.x; p < (uint32_t)C13_P; p += gridDim.x) {
        for (int i = threadIdx.x; i < C13_BINS; i += C13_TB) h[i] = 0;
        __syncthreads();
        const unsigned long long lo = part_base[p], hi = part_base[p + 1];
        // head up to a 16-byte boundary, then 8 codes per lane per load, then the tail
        const unsigned long long lo8 = (lo + 7) & ~7ull, hi8 = hi & ~7ull;
        if (lo8 < hi8) {
            for (unsigned long long i = lo + threadIdx.x; i < lo8; i += C13_TB) atomicAdd(&h[parts[i]], 1u);
            const uint4* v = (const uint4*)(parts + lo8);
            const unsigned long long nv = (hi8 - lo8) >> 3;
            auto add8 = [&](const uint4& x) {
                atomicAdd(&h[x.x & 0xFFFFu], 1u); atomicAdd(&h[x.x >> 16], 1u);
                atomicAdd(&h[x.y & 0xFFFFu], 1u); atomicAdd(&h[x.y >> 16], 1u);
                atomicAdd(&h[x.z & 0xFFFFu], 1u); atomicAdd(&h[x.z >> 16], 1u);
                atomicAdd(&h[x.w & 0xFFFFu], 1u); atomicAdd(&h[x.w >> 16], 1u);
            };
            // one workgroup per CU: four 16-byte loads per lane in flight (64 KiB per CU) to cover the HBM latency
            unsigned long long i = threadIdx.x;
            for (; i + 3 * C13_TB < nv; i += 4 * C13_TB) {
                const uint4 x0 = v[i], x1 = v[i + C13_TB], x2 = v[i + 2 * C13_TB], x3 = v[i + 3 * C13_TB];
                add8(x0); add8(x1); add8(x2); add8(x3);
            }
            for (; i < nv; i += C13_TB) add8(v[i]);
            for (unsigned long long i = hi8 + threadIdx.x; i < hi; i += C13_TB) atomicAdd(&h[parts[i]], 1u);
        } else {
            for (unsigned long long i = lo + threadIdx.x; i < hi; i += C13_TB) atomicAdd(&h[parts[i]], 1u);
        }
        __syncthreads();
        const uint64_t base = (uint64_t)p * C13_BINS;
        if (perm) {
            for (int i = threadIdx.x; i < C13_BINS; i += C13_TB) {
                const uint32_t c = h[i];
                // every bin has exactly one writer per launch, launches of one call are ordered on the stream: plain += is exact
                if (c) { const uint32_t slot = perm[base + i]; if (slot < 67108864u) out_mphf[slot] = accumulate ? out_mphf[slot] + c : (uint64_t)c; }
            }
        } else {
            for (int i = threadIdx.x; i < C13_BINS; i += C13_TB) table[base + i] = accumulate ? table[base + i] + h[i] : (unsigned long long)h[i];
        }
        __syncthreads();
    }
}

static constexpr unsigned C13_MAXGRID = 512;

// workspace: part_count[P] u64 | part_base[P+1] u64 | cnt[G][P] u32 | parts u16[nwin]
uint64_t count13_workspace_bytes(uint64_t len) {
    const uint64_t nwin = len >= 13 ? len - 12 : 0;
    return 8ull * C13_P + 8ull * (C13_P + 1) + 4ull * C13_P * C13_MAXGRID + 2ull * nwin + 64;
}

hipError_t launch_count13_partitioned(const uint8_t* buf, uint64_t len, void* workspace, unsigned long long* table, const uint32_t* perm,
                                      uint64_t* out_mphf, int accumulate, hipStream_t s) {
    const uint64_t nwin = len >= 13 ? len - 12 : 0;
    unsigned long long* part_count = (unsigned long long*)workspace;
    unsigned long long* part_base = part_count + C13_P;
    uint32_t* cnt = (uint32_t*)(part_base + C13_P + 1);
    uint16_t* parts = (uint16_t*)(((uintptr_t)(cnt + (uint64_t)C13_P * C13_MAXGRID) + 15) & ~(uintptr_t)15);   // 16-byte aligned for uint4 loads
    const uint64_t ntiles = (nwin + C13_TILE - 1) / C13_TILE;
    const unsigned grid = (unsigned)std::min<uint64_t>(ntiles ? ntiles : 1, C13_MAXGRID);
    const size_t hist_lds = 4 * C13_BINS;                                   // 131 072 B
    const size_t split_lds = 4 * (3 * C13_P + C13_DUMMY + 16) + 4 * C13_TILE;   // 155 968 B
    {   // > 64 KiB of dynamic LDS needs the attribute; set per call (cheap, and correct for every device / thread)
        hipError_t e = hipFuncSetAttribute((const void*)k_c13_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_lds);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)k_c13_split, hipFuncAttributeMaxDynamicSharedMemorySize, (int)split_lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_c13_sizes, dim3(grid), dim3(C13_TB), 0, s, buf, len, ntiles, cnt);
    hipLaunchKernelGGL(k_c13_colscan, dim3(C13_P / C13_TB), dim3(C13_TB), 0, s, cnt, grid, part_count);
    hipLaunchKernelGGL(k_c13_scan, dim3(1), dim3(C13_TB), 0, s, part_count, part_base);
    hipLaunchKernelGGL(k_c13_split, dim3(grid), dim3(C13_TB), split_lds, s, buf, len, ntiles, part_base, cnt, parts);
    hipLaunchKernelGGL(k_c13_hist, dim3(C13_P), dim3(C13_TB), hist_lds, s, parts, part_base, table, perm, out_mphf, accumulate);
    return hipGetLastError();
}

}  // namespace aix
