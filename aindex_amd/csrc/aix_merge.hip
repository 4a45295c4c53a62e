// aix_merge.hip — K1 beyond one piece: sorted (key, count) runs are merged, never re-sorted.
//
// The reference merges its per-thread maps once (count_kmers.cpp:334-341). Here a buffer of more than 2^31 windows is counted piece by
// piece (aix_k1.hip: every piece comes out as keys ascending + counts), and across GPUs every rank receives one sorted run per peer
// (aindex_amd/dist.py). Both used to concatenate and run a full-width radix sort + reduce-by-key per merge; what is needed is a TWO-WAY
// MERGE WITH SUMMATION of two runs that are each sorted and distinct:
//   k_merge_split : merge-path split points, one per tile of 2 048 merged entries (binary search on the diagonal, ties = A first);
//                   a split that would separate A[i] == B[j] moves B[j] into the earlier tile, so equal keys always meet in one tile
//   k_merge_tile  : both slices into LDS, every lane merge-path-searches its own 8 entries, merges them in registers, writes the merged
//                   tile back to LDS; an entry equal to its left neighbour is the B half of a pair and folds its count into the A half;
//                   pass 1 counts the survivors per tile, a scan gives the tile bases, pass 2 writes keys and summed counts
// One read of the keys per pass, one of the counts, one write of the result: ~40 B per entry against twelve radix passes of 16 B.
// Sizes are 64-bit throughout (the 2^32 bound of the old path is gone).
#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "aix_internal.hpp"

namespace aix {

static constexpr int kB = 256;
static inline unsigned grid_of(uint64_t work) {
    uint64_t b = (work + kB - 1) / kB;
    if (b > 65536) b = 65536;
    if (b == 0) b = 1;
    return (unsigned)b;
}

static constexpr int MG_TB = 256, MG_IPT = 8, MG_T = MG_TB * MG_IPT;

__global__ void __launch_bounds__(kB) k_merge_split(const uint64_t* __restrict__ A, uint64_t na, const uint64_t* __restrict__ B, uint64_t nb, uint64_t ntiles,
                                                    uint64_t* __restrict__ split /* [ntiles + 1][2] */) {
    const uint64_t t = (uint64_t)blockIdx.x * kB + threadIdx.x;
    if (t > ntiles) return;
    const uint64_t d = std::min<uint64_t>(t * (uint64_t)MG_T, na + nb);
    uint64_t lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
    while (lo < hi) {                                            // entries of A among the first d merged ones, equal keys: A first
        const uint64_t mid = (lo + hi) >> 1;
        if (A[mid] <= B[d - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    uint64_t ai = lo, bi = d - lo;
    if (ai > 0 && bi < nb && A[ai - 1] == B[bi]) ++bi;           // keep the pair in one tile (tiles hold MG_T + 1 entries at most)
    split[2 * t] = ai;
    split[2 * t + 1] = bi;
}

template <bool WRITE>
__global__ void __launch_bounds__(MG_TB) k_merge_tile(const uint64_t* __restrict__ Ak, const uint64_t* __restrict__ Ac, const uint64_t* __restrict__ Bk,
                                                      const uint64_t* __restrict__ Bc, const uint64_t* __restrict__ split, uint64_t ntiles,
                                                      uint64_t* __restrict__ tile_count /* pass 1: out */, const uint64_t* __restrict__ tile_base /* pass 2: in */,
                                                      uint64_t* __restrict__ out_k, uint64_t* __restrict__ out_c) {
    __shared__ uint64_t sk[MG_T + 2];
    __shared__ uint64_t sc[WRITE ? MG_T + 2 : 1];
    __shared__ uint32_t wsum[MG_TB / 64];
    const int tid = threadIdx.x;
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint64_t a0 = split[2 * t], b0 = split[2 * t + 1], a1 = split[2 * t + 2], b1 = split[2 * t + 3];
        const uint32_t na_t = (uint32_t)(a1 - a0), nb_t = (uint32_t)(b1 - b0), m = na_t + nb_t;
        for (uint32_t i = tid; i < m; i += MG_TB) {
            sk[i] = i < na_t ? Ak[a0 + i] : Bk[b0 + (i - na_t)];
            if (WRITE) sc[i] = i < na_t ? Ac[a0 + i] : Bc[b0 + (i - na_t)];
        }
        __syncthreads();
        // lane i merges entries [8 i, 8 i + 8); the last lane also takes the 2 049th entry of a tile that was handed a pair's second half
        const uint32_t d0 = std::min<uint32_t>((uint32_t)tid * MG_IPT, m), d1 = tid == MG_TB - 1 ? m : std::min<uint32_t>(d0 + MG_IPT, m);
        uint32_t lo = d0 > nb_t ? d0 - nb_t : 0u, hi = d0 < na_t ? d0 : na_t;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (sk[mid] <= sk[na_t + d0 - 1 - mid]) lo = mid + 1; else hi = mid;
        }
        uint32_t ai = lo, bi = d0 - lo;
        uint64_t rk[MG_IPT + 1], rc[MG_IPT + 1];
#pragma unroll
        for (int j = 0; j <= MG_IPT; ++j) {
            rk[j] = 0; rc[j] = 0;
            if (d0 + j < d1) {
                const bool take_a = ai < na_t && (bi >= nb_t || sk[ai] <= sk[na_t + bi]);
                const uint32_t idx = take_a ? ai : na_t + bi;
                rk[j] = sk[idx];
                if (WRITE) rc[j] = sc[idx];
                if (take_a) ++ai; else ++bi;
            }
        }
        __syncthreads();                                         // the two slices have been read by everybody: the merged tile replaces them
#pragma unroll
        for (int j = 0; j <= MG_IPT; ++j)
            if (d0 + j < d1) { sk[d0 + j] = rk[j]; if (WRITE) sc[d0 + j] = rc[j]; }
        __syncthreads();
        uint32_t kept = 0;
#pragma unroll
        for (int j = 0; j <= MG_IPT; ++j) {
            const uint32_t i = d0 + j;
            if (i < d1 && (i == 0 || sk[i] != sk[i - 1])) ++kept;
        }
        uint32_t incl = kept;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(incl, d);
            if ((tid & 63) >= d) incl += y;
        }
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        uint32_t off = 0, all = 0;
#pragma unroll
        for (int w = 0; w < MG_TB / 64; ++w) { const uint32_t x = wsum[w]; if (w < (tid >> 6)) off += x; all += x; }
        if (!WRITE) {
            if (tid == 0) tile_count[t] = all;
        } else {
            uint64_t o = tile_base[t] + off + incl - kept;
#pragma unroll
            for (int j = 0; j <= MG_IPT; ++j) {
                const uint32_t i = d0 + j;
                if (i < d1 && (i == 0 || sk[i] != sk[i - 1])) {
                    out_k[o] = sk[i];
                    out_c[o] = sc[i] + ((i + 1 < m && sk[i + 1] == sk[i]) ? sc[i + 1] : 0ull);
                    ++o;
                }
            }
        }
        __syncthreads();                                         // sk / wsum are reused by the next tile
    }
}

// (A, B) -> keys ascending, equal keys summed. A and B are each sorted and free of repeats. Outputs from the scratch pool (caller frees);
// *n_out is known when this returns (the stream is synchronised).
hipError_t merge_sum_runs(const uint64_t* ak, const uint64_t* ac, uint64_t na, const uint64_t* bk, const uint64_t* bc, uint64_t nb, uint64_t** d_keys_out,
                          uint64_t** d_counts_out, uint64_t* n_out, hipStream_t s) {
    *d_keys_out = nullptr; *d_counts_out = nullptr; *n_out = 0;
    const uint64_t tot = na + nb;
    if (tot == 0) return hipSuccess;
    const uint64_t ntiles = (tot + MG_T - 1) / MG_T;
    DevArr split(s), counts(s), bases(s), tmp(s), ok(s), oc(s);
    hipError_t e = split.alloc(16 * (ntiles + 1));
    if (e == hipSuccess) e = counts.alloc(8 * (ntiles + 1));
    if (e == hipSuccess) e = bases.alloc(8 * (ntiles + 1));
    if (e != hipSuccess) return e;
    size_t tb = 0;
    e = rocprim::exclusive_scan(nullptr, tb, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, (size_t)(ntiles + 1), rocprim::plus<uint64_t>(), s);
    if (e == hipSuccess) e = tmp.alloc(tb ? tb : 1);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, 1u << 20);
    hipLaunchKernelGGL(k_merge_split, dim3((unsigned)((ntiles + 1 + kB - 1) / kB)), dim3(kB), 0, s, ak, na, bk, nb, ntiles, (uint64_t*)split.p);
    e = hipMemsetAsync((uint64_t*)counts.p + ntiles, 0, 8, s);   // the scan runs over ntiles + 1 words: its last output is the total
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_merge_tile<false>), dim3(grid), dim3(MG_TB), 0, s, ak, ac, bk, bc, (const uint64_t*)split.p, ntiles, (uint64_t*)counts.p,
                       (const uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t*)nullptr);
    e = hipGetLastError();
    if (e == hipSuccess) e = rocprim::exclusive_scan(tmp.p, tb, (uint64_t*)counts.p, (uint64_t*)bases.p, (uint64_t)0, (size_t)(ntiles + 1), rocprim::plus<uint64_t>(), s);
    uint64_t total = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&total, (uint64_t*)bases.p + ntiles, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    e = ok.alloc(8 * (total ? total : 1));
    if (e == hipSuccess) e = oc.alloc(8 * (total ? total : 1));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_merge_tile<true>), dim3(grid), dim3(MG_TB), 0, s, ak, ac, bk, bc, (const uint64_t*)split.p, ntiles, (uint64_t*)nullptr,
                       (const uint64_t*)bases.p, (uint64_t*)ok.p, (uint64_t*)oc.p);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    *d_keys_out = (uint64_t*)ok.release(); *d_counts_out = (uint64_t*)oc.release(); *n_out = total;
    return hipSuccess;
}

__global__ void __launch_bounds__(kB) k_widen_counts(const uint32_t* __restrict__ c32, uint64_t n, uint64_t* __restrict__ c64) {
    const uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n; i += stride) c64[i] = c32[i];
}
__global__ void __launch_bounds__(kB) k_flag_min64(const uint64_t* __restrict__ counts, uint64_t n, uint64_t min_count, uint8_t* __restrict__ flags) {
    const uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n; i += stride) flags[i] = counts[i] >= min_count ? 1 : 0;
}

// keep the entries with count >= min_count (order kept). In: pool blocks k / c of n entries (freed here when replaced).
static hipError_t filter_min_count(uint64_t** k, uint64_t** c, uint64_t* n, uint64_t min_count, hipStream_t s) {
    if (min_count <= 1 || *n == 0) return hipSuccess;
    DevArr flags(s), fk(s), fc(s), d_sel(s), tmp(s);
    hipError_t e = flags.alloc(*n);
    if (e == hipSuccess) e = fk.alloc(8 * *n);
    if (e == hipSuccess) e = fc.alloc(8 * *n);
    if (e == hipSuccess) e = d_sel.alloc(8);
    size_t tb = 0;
    if (e == hipSuccess) { hipLaunchKernelGGL(k_flag_min64, dim3(grid_of(*n)), dim3(kB), 0, s, (const uint64_t*)*c, *n, min_count, (uint8_t*)flags.p); e = hipGetLastError(); }
    if (e == hipSuccess) e = rocprim::select(nullptr, tb, *k, (uint8_t*)flags.p, (uint64_t*)fk.p, (uint64_t*)d_sel.p, (size_t)*n, s);
    if (e == hipSuccess) e = tmp.alloc(tb ? tb : 1);
    if (e == hipSuccess) e = rocprim::select(tmp.p, tb, *k, (uint8_t*)flags.p, (uint64_t*)fk.p, (uint64_t*)d_sel.p, (size_t)*n, s);
    if (e == hipSuccess) e = rocprim::select(tmp.p, tb, *c, (uint8_t*)flags.p, (uint64_t*)fc.p, (uint64_t*)d_sel.p, (size_t)*n, s);
    uint64_t kept = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&kept, d_sel.p, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    pool_free(*k); pool_free(*c);
    *k = (uint64_t*)fk.release(); *c = (uint64_t*)fc.release(); *n = kept;
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------------
// DistinctAcc: the distinct set of everything added so far (K1 for buffers / files of any length). A piece = every k-window of one
// PLAIN buffer of at most 2^31 windows; its distinct set (aix_k1.hip, or the radix-sort path) is merged into the accumulated one.
// ---------------------------------------------------------------------------------------------
DistinctAcc::DistinctAcc(int kk, int canon, hipStream_t st) : k(kk), canon_mode(canon), s(st) { scratch.s = st; }
DistinctAcc::~DistinctAcc() {
    if (acc_k || acc_c) (void)hipStreamSynchronize(s);
    if (acc_k) pool_free(acc_k);
    if (acc_c) pool_free(acc_c);
}

hipError_t DistinctAcc::add_plain(const uint8_t* d_plain, uint64_t plen) {
    if (plen < (uint64_t)k) return hipSuccess;
    const uint64_t nwin = plen - k + 1;
    if (nwin > (1ull << 31)) return hipErrorInvalidValue;
    ++pieces;
    // the 8 B-per-window staging and the partition workspace stay with the accumulator from piece to piece (K1Scratch)
    struct { void* p; } codes{nullptr};
    hipError_t e = scratch.need(&scratch.codes, &scratch.codes_bytes, 8 * nwin);
    codes.p = scratch.codes;
    uint64_t* pk = nullptr; uint32_t* pc = nullptr; uint64_t pm = 0;
    uint64_t* pc64 = nullptr;                                             // the MSD path hands its counts over as u64 (no widening pass)
    bool sorted_path = !k1_msd_eligible(nwin, k);
    if (e == hipSuccess && sorted_path) e = launch_window_codes(d_plain, plen, k, canon_mode, (uint64_t*)codes.p, s);
    if (e == hipSuccess && !sorted_path) {                                // MSD partition + per-bucket LDS hash / sort (aix_k1.hip); windows encoded inside level 1
        bool fell_back = false;
        e = distinct_from_codes_msd((uint64_t*)codes.p, nwin, k, &pk, &pc, &pm, &fell_back, s, d_plain, plen, canon_mode, &pc64, &scratch);
        if (e == hipSuccess && fell_back) {                               // a bucket too rich for LDS: the codes were used as staging, make them again
            sorted_path = true;
            e = launch_window_codes(d_plain, plen, k, canon_mode, (uint64_t*)codes.p, s);
        }
    }
    if (e == hipSuccess && sorted_path) e = distinct_from_codes((uint64_t*)codes.p, nwin, k, 1, &pk, &pc, &pm, s);
    DevArr hold_k(s), hold_c(s), hold_c64(s); hold_k.p = pk; hold_c.p = pc; hold_c64.p = pc64;
    if (e != hipSuccess || pm == 0) return e;
    if (!pc64) {                                                          // the sort path counts in 32 bits
        e = hold_c64.alloc(8 * pm);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_widen_counts, dim3(grid_of(pm)), dim3(kB), 0, s, pc, pm, (uint64_t*)hold_c64.p);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (acc_n == 0) {                                                     // the first non-empty piece: sorted, distinct, u64 counts — taken as it is
        (void)hipStreamSynchronize(s);
        acc_k = (uint64_t*)hold_k.release(); acc_c = (uint64_t*)hold_c64.release(); acc_n = pm;
        return hipSuccess;
    }
    uint64_t *mk = nullptr, *mc = nullptr, mn = 0;
    e = merge_sum_runs(acc_k, acc_c, acc_n, (const uint64_t*)hold_k.p, (const uint64_t*)hold_c64.p, pm, &mk, &mc, &mn, s);
    if (e != hipSuccess) return e;
    ++merges;
    pool_free(acc_k); pool_free(acc_c);
    acc_k = mk; acc_c = mc; acc_n = mn;
    return hipSuccess;
}

hipError_t DistinctAcc::finish(uint64_t min_count, uint64_t** d_keys_out, uint64_t** d_counts_out, uint64_t* n_out) {
    *d_keys_out = nullptr; *d_counts_out = nullptr; *n_out = 0;
    if (acc_n == 0) return hipSuccess;
    const hipError_t e = filter_min_count(&acc_k, &acc_c, &acc_n, min_count, s);
    if (e != hipSuccess) return e;
    *d_keys_out = acc_k; *d_counts_out = acc_c; *n_out = acc_n;
    acc_k = nullptr; acc_c = nullptr; acc_n = 0;
    return hipSuccess;
}

hipError_t distinct_from_plain(const uint8_t* d_plain, uint64_t plen, int k, int canon_mode, uint64_t min_count, uint64_t piece, uint64_t** d_keys_out,
                               uint64_t** d_counts_out, uint64_t* n_out, hipStream_t s) {
    *d_keys_out = nullptr; *d_counts_out = nullptr; *n_out = 0;
    if (plen < (uint64_t)k) return hipSuccess;
    const uint64_t nwin_all = plen - k + 1;
    // 32-bit window indices inside a piece allow 2^31 windows; the default is 2^30: the two big temporaries of a piece are then 8 and 8.5 GiB
    // and stay in the block cache between calls (16 + 17 GiB did not: every call paid ~0.75 s of hipMalloc / hipFree), at the price of one more
    // 2.4 ms merge per 2^30 windows
    if (piece == 0 || piece > (1ull << 31)) piece = 1ull << 30;
    DistinctAcc acc(k, canon_mode, s);
    for (uint64_t w0 = 0; w0 < nwin_all; w0 += piece) {                      // a window belongs to the piece that holds its first byte
        const uint64_t nwin = std::min(piece, nwin_all - w0);
        const hipError_t e = acc.add_plain(d_plain + w0, nwin + k - 1);
        if (e != hipSuccess) return e;
    }
    return acc.finish(min_count, d_keys_out, d_counts_out, n_out);
}

// (key, count) pairs in `nruns` runs, run r = entries [offs[r], offs[r + 1]), each run sorted by key and free of repeats (what a rank
// holds after the K1 exchange: one run per peer) -> keys ascending, counts of equal keys summed, counts >= min_count. A tree of
// two-way merges: every entry is read and written log2(nruns) times, nothing is sorted. Inputs are not modified.
hipError_t merge_runs(const uint64_t* d_keys, const uint64_t* d_counts, const uint64_t* offs, uint32_t nruns, uint64_t min_count, uint64_t** d_keys_out,
                      uint64_t** d_counts_out, uint64_t* n_out, hipStream_t s) {
    *d_keys_out = nullptr; *d_counts_out = nullptr; *n_out = 0;
    struct Run { const uint64_t* k; const uint64_t* c; uint64_t n; bool owned; };
    std::vector<Run> cur;
    for (uint32_t r = 0; r < nruns; ++r)
        if (offs[r + 1] > offs[r]) cur.push_back(Run{d_keys + offs[r], d_counts + offs[r], offs[r + 1] - offs[r], false});
    auto drop = [&](std::vector<Run>& v) { for (Run& x : v) if (x.owned) { pool_free((void*)x.k); pool_free((void*)x.c); } v.clear(); };
    hipError_t e = hipSuccess;
    while (cur.size() > 1 && e == hipSuccess) {
        std::vector<Run> nxt;
        for (size_t i = 0; i + 1 < cur.size() && e == hipSuccess; i += 2) {
            uint64_t *mk = nullptr, *mc = nullptr, mn = 0;
            e = merge_sum_runs(cur[i].k, cur[i].c, cur[i].n, cur[i + 1].k, cur[i + 1].c, cur[i + 1].n, &mk, &mc, &mn, s);
            if (e == hipSuccess) nxt.push_back(Run{mk, mc, mn, true});
        }
        if (e == hipSuccess && (cur.size() & 1)) { nxt.push_back(cur.back()); cur.back().owned = false; }
        drop(cur);
        cur.swap(nxt);
    }
    if (e != hipSuccess) { drop(cur); return e; }
    if (cur.empty()) return hipSuccess;
    uint64_t *k = nullptr, *c = nullptr, n = cur[0].n;
    if (cur[0].owned) { k = (uint64_t*)cur[0].k; c = (uint64_t*)cur[0].c; }
    else {                                                                  // a single run: the result is a copy (inputs stay the caller's)
        DevArr ck(s), cc(s);
        e = ck.alloc(8 * n);
        if (e == hipSuccess) e = cc.alloc(8 * n);
        if (e == hipSuccess) e = hipMemcpyAsync(ck.p, cur[0].k, 8 * n, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(cc.p, cur[0].c, 8 * n, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        k = (uint64_t*)ck.release(); c = (uint64_t*)cc.release();
    }
    e = filter_min_count(&k, &c, &n, min_count, s);
    if (e != hipSuccess) { pool_free(k); pool_free(c); return e; }
    *d_keys_out = k; *d_counts_out = c; *n_out = n;
    return hipSuccess;
}

// the same for pairs in ANY order (repeated keys anywhere): sort by key + reduce-by-key. Kept for callers that cannot name their runs.
hipError_t merge_counts(const uint64_t* d_keys, const uint64_t* d_counts, uint64_t n, uint64_t min_count, uint64_t** d_keys_out, uint64_t** d_counts_out,
                        uint64_t* n_out, hipStream_t s) {
    *d_keys_out = nullptr; *d_counts_out = nullptr; *n_out = 0;
    if (n == 0) return hipSuccess;
    DevArr srt_k(s), srt_c(s), out_k(s), out_c(s), d_n(s), tmp(s);
    hipError_t e = srt_k.alloc(8 * n);
    if (e == hipSuccess) e = srt_c.alloc(8 * n);
    size_t tb = 0;
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, tb, d_keys, (uint64_t*)srt_k.p, d_counts, (uint64_t*)srt_c.p, (size_t)n, 0u, 64u, s);
    if (e == hipSuccess) e = tmp.alloc(tb);
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(tmp.p, tb, d_keys, (uint64_t*)srt_k.p, d_counts, (uint64_t*)srt_c.p, (size_t)n, 0u, 64u, s);
    if (e == hipSuccess) e = out_k.alloc(8 * n);
    if (e == hipSuccess) e = out_c.alloc(8 * n);
    if (e == hipSuccess) e = d_n.alloc(8);
    tb = 0;
    if (e == hipSuccess) e = rocprim::reduce_by_key(nullptr, tb, (uint64_t*)srt_k.p, (uint64_t*)srt_c.p, (size_t)n, (uint64_t*)out_k.p, (uint64_t*)out_c.p, (uint64_t*)d_n.p, rocprim::plus<uint64_t>(), rocprim::equal_to<uint64_t>(), s);
    if (e == hipSuccess) e = tmp.alloc(tb);
    if (e == hipSuccess) e = rocprim::reduce_by_key(tmp.p, tb, (uint64_t*)srt_k.p, (uint64_t*)srt_c.p, (size_t)n, (uint64_t*)out_k.p, (uint64_t*)out_c.p, (uint64_t*)d_n.p, rocprim::plus<uint64_t>(), rocprim::equal_to<uint64_t>(), s);
    uint64_t merged = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&merged, d_n.p, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    uint64_t *k = (uint64_t*)out_k.release(), *c = (uint64_t*)out_c.release();
    e = filter_min_count(&k, &c, &merged, min_count, s);
    if (e != hipSuccess) { pool_free(k); pool_free(c); return e; }
    *d_keys_out = k; *d_counts_out = c; *n_out = merged;
    return hipSuccess;
}

}  // namespace aix
