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

// Encode the C13_WPT windows whose starts are [S, S+WPT): code[j] (26 bits) and a validity bit mask.
// Bytes at positions >= len count as separators. Upper-casing and the ACGT test follow
// normalize_sequence / is_valid_kmer (count_kmers13.cpp:100-126).
__device__ __forceinline__ uint32_t encode_run13(const uint8_t* __restrict__ buf, uint64_t len, uint64_t S, uint32_t (&code)[C13_WPT]) {
    constexpr int NB = C13_WPT + 12;                  // 44 bytes
    constexpr int ND = NB / 4;                        // 11 aligned dwords
    uint32_t e[ND];
    uint32_t validmask = 0;
    if (S >= len) {
#pragma unroll
        for (int j = 0; j < C13_WPT; ++j) code[j] = 0;
        return 0;
    }
    const uint64_t limit = len - S;                   // bytes available from S
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
    uint32_t c26 = 0, run = 0;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const uint32_t c = (e[i >> 2] >> (8 * (i & 3))) & 0xDFu;
        const uint32_t v = ((c >> 1) ^ (c >> 2)) & 3u;
        const bool ok = (c == ((AIX_LUT_ACGT >> (8 * v)) & 0xFFu)) && ((uint64_t)i < limit);
        c26 = ((c26 << 2) | v) & 0x3FFFFFFu;
        run = ok ? run + 1 : 0;
        if (i >= 12) {
            code[i - 12] = c26;
            if (run >= 13) validmask |= 1u << (i - 12);
        }
    }
    return validmask;
}

// P1: per-(workgroup, partition) window counts. Workgroup b owns tiles b, b+G, b+2G, ... in P1 and in P2 alike.
__global__ void __launch_bounds__(C13_TB) k_c13_sizes(const uint8_t* __restrict__ buf, uint64_t len, uint64_t ntiles, uint32_t* __restrict__ cnt /* [G][P] */) {
    __shared__ uint32_t hist[C13_P];
    for (int i = threadIdx.x; i < C13_P; i += C13_TB) hist[i] = 0;
    __syncthreads();
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        uint32_t code[C13_WPT];
        const uint32_t vm = encode_run13(buf, len, t * C13_TILE + (uint64_t)threadIdx.x * C13_WPT, code);
#pragma unroll
        for (int j = 0; j < C13_WPT; ++j)
            if (vm & (1u << j)) atomicAdd(&hist[code[j] >> C13_BINBITS], 1u);
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
    uint32_t* loc_off = hist + C13_P;                           // [P] exclusive scan of hist
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
    }
    __syncthreads();
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t code[C13_WPT];
        uint32_t rank[C13_WPT / 2];                              // two 16-bit ranks per register
        const uint32_t vm = encode_run13(buf, len, tile * C13_TILE + (uint64_t)t * C13_WPT, code);
#pragma unroll
        for (int j = 0; j < C13_WPT; ++j) {
            uint32_t r = 0;
            if (vm & (1u << j)) r = atomicAdd(&hist[code[j] >> C13_BINBITS], 1u);
            if (j & 1) rank[j >> 1] |= r << 16; else rank[j >> 1] = r;
        }
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
#pragma unroll
        for (int j = 0; j < C13_WPT; ++j) {
            if (vm & (1u << j)) {
                const uint32_t p = code[j] >> C13_BINBITS;
                const uint32_t r = (j & 1) ? (rank[j >> 1] >> 16) : (rank[j >> 1] & 0xFFFFu);
                sorted[loc_off[p] + r] = (p << 16) | (code[j] & (C13_BINS - 1));
            }
        }
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
    const size_t split_lds = 4 * (3 * C13_P + 16) + 4 * C13_TILE;           // 155 712 B
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
