// aix_positions.hip — A1/A2 rows: the positions index ("aindex") of a reads buffer.
//   A1 AIndexCompressed ctor   src/hash.hpp:365-399      indices = exclusive prefix sum of tf
//   A2 lu_compressed_worker    src/hash.cpp:960-1060     positions[indices[h] + slot] = offset + 1, slot < tf[h]
//
// The reference assigns slots with fetch_add in thread-arrival order (ascending offsets with one thread; SURVEY
// §8a-A2). Here every window's bucket is computed in parallel, then a STABLE radix sort by bucket (rocPRIM) of the
// window offsets (a counting iterator, i.e. already ascending) yields each bucket's offsets in ascending order —
// exactly the 1-thread reference result, deterministically — and the first tf[h] of them are placed.
#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "aix_internal.hpp"

namespace aix {

static constexpr int kB = 256;
static inline unsigned grid_of(uint64_t work) {
    uint64_t b = (work + kB - 1) / kB;
    if (b > 8192) b = std::max<uint64_t>(8192, std::min<uint64_t>(b / 4, 65536));     // as grid_for of aix_kernels.hip: finer workgroups for long buffers
    if (b == 0) b = 1;
    return (unsigned)b;
}

// any byte of x (restricted to `mask` bytes) equal to c ?
__device__ __forceinline__ bool has_byte(uint64_t x, uint8_t c, uint64_t mask) {
    const uint64_t z = (x ^ (0x0101010101010101ULL * c)) | ~mask;
    return (((z - 0x0101010101010101ULL) & ~z) & 0x8080808080808080ULL) != 0;
}

__global__ void __launch_bounds__(kB) k_tf_u64(const IndexDev ix, uint64_t n, uint64_t* __restrict__ tf64) {
    const uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i <= n; i += stride) tf64[i] = i < n ? (uint64_t)key_at(ix, i).tf : 0ull;
}

// bucket of every window (key = n for "no bucket"): hash.cpp:1004-1052. Wave-uniform loop: the verification-table probe is
// wave-cooperative (aix_device.hpp: bucket_probe_wave); a forward-strand window with bytes outside ACGT hashes its RAW bytes
// (:1032-1040), which the table cannot answer, and goes through the MPHF.
template <int LPP>
__global__ void __launch_bounds__(kB) k_a2_probe(const IndexDev ix, const uint8_t* __restrict__ buf, uint64_t nwin, uint64_t start, uint32_t* __restrict__ keys) {
    const uint64_t stride = (uint64_t)gridDim.x * kB;
    const uint64_t FULL = ~0ULL, LAST7 = 0x00FFFFFFFFFFFFFFULL;
    for (uint64_t base = (uint64_t)blockIdx.x * kB + (threadIdx.x & ~63u); base < nwin; base += stride) {
        const uint64_t i = base + (threadIdx.x & 63u);
        const bool in = i < nwin;
        uint32_t key = (uint32_t)ix.n;
        uint64_t w0 = 0, w1 = 0, w2 = 0;
        bool probe = false;
        if (in && i >= start) {
            load23(buf + i, w0, w1, w2);
            const bool skip = has_byte(w0, '\n', FULL) || has_byte(w1, '\n', FULL) || has_byte(w2, '\n', LAST7) ||
                              has_byte(w0, '~', FULL) || has_byte(w1, '~', FULL) || has_byte(w2, '~', LAST7) ||
                              has_byte(w0, 'N', FULL) || has_byte(w1, 'N', FULL) || has_byte(w2, 'N', LAST7);      // :1006-1011
            probe = !skip;
        }
        const Enc23 e = encode23_words(w0, w1, w2);                            // get_dna23_bitset: non-ACGT -> 0
        const uint64_t r = revcomp(e.code, 23);
        const bool fwd = e.code <= r;                                           // :1032: probe the numerically smaller strand only
        const uint64_t want = fwd ? e.code : r;
        uint64_t x0 = w0, x1 = w1, x2 = w2;                                     // forward: the raw bytes of the window
        if (!fwd) ascii23_of_rc(e.code, x0, x1, x2);
        const bool tab = probe && (e.valid || !fwd);                            // the hashed bytes are the ASCII of `want`
        const bool rest = probe;
        uint64_t a = 0, b = 0, c = 0;
        if (rest) jenkins23(x0, x1, x2, ix.m.seed, a, b, c);
        bool mphf = rest;
        if (ix.bk) {
            const bool use = rest && tab;
            const BkRes k = bucket_probe_wave<LPP>(ix.bk, ix.nb, use, a, want);
            if (use) {
                if (k.found) key = k.slot;
                mphf = !k.found && k.overflow;
            }
        }
        if (mphf) {
            const uint64_t h = mphf_from_hash(ix.m, a, b, c);
            if (h < ix.n && key_at(ix, h).code == want) key = (uint32_t)h;
        }
        if (in) keys[i] = key;
    }
}

__global__ void __launch_bounds__(kB) k_a2_first(const uint32_t* __restrict__ skeys, uint64_t nwin, uint32_t n, uint32_t* __restrict__ first) {
    const uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t j = (uint64_t)blockIdx.x * kB + threadIdx.x; j < nwin; j += stride) {
        const uint32_t h = skeys[j];
        if (h < n && (j == 0 || skeys[j - 1] != h)) first[h] = (uint32_t)j;
    }
}
// `filled[h]` = occurrences of bucket h in the pieces before this one (the reference's ppositions[h] counter, hash.cpp:1037)
__global__ void __launch_bounds__(kB) k_a2_place(const IndexDev ix, const uint32_t* __restrict__ skeys, const uint32_t* __restrict__ svals, uint64_t nwin,
                                                uint64_t piece_first, const uint32_t* __restrict__ first, const uint32_t* __restrict__ filled,
                                                const uint64_t* __restrict__ indices, uint64_t* __restrict__ positions) {
    const uint64_t stride = (uint64_t)gridDim.x * kB;
    const uint32_t n = (uint32_t)ix.n;

    for (uint64_t j = (uint64_t)blockIdx.x * kB + threadIdx.x; j < nwin; j += stride) {
        const uint32_t h = skeys[j];
        if (h >= n) continue;
        const uint64_t rank = (uint64_t)filled[h] + (j - first[h]);
        const uint64_t base = indices[h], tf = indices[h + 1] - base;                             // tf[h] (13-mer: the u64 table of count_kmers13) from its prefix sums
        if (rank < tf) positions[base + rank] = piece_first + svals[j] + 1;                       // :1037-1040 / compute_aindex13.cpp:205-211, 1-based offsets
    }
}
// 13-mer probe (compute_aindex13.cpp:163-204): a window counts iff its 13 bytes are upper-case A/C/G/T; forward strand only;
// bucket = mphf(window) = perm13[code] (the MPHF over all 13-mers, tabulated at open)
__global__ void __launch_bounds__(kB) k_a2_probe13(const uint32_t* __restrict__ perm13, const uint8_t* __restrict__ buf, uint64_t nwin, uint64_t start,
                                                  uint32_t* __restrict__ keys) {
    const uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < nwin; i += stride) {
        uint32_t key = 67108864u;
        if (i >= start) {
            uint64_t w0, w1;
            load13(buf + i, w0, w1);
            const Enc13 e = encode13_words(w0, w1);
            if (e.valid) { const uint32_t h = perm13[e.code]; if (h < 67108864u) key = h; }
        }
        keys[i] = key;
    }
}
__global__ void __launch_bounds__(kB) k_tf13_copy(const uint64_t* __restrict__ tf, uint64_t n, uint64_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i <= n; i += stride) out[i] = i < n ? tf[i] : 0ull;
}
// after a piece has been placed: filled[h] += occurrences of h in the piece (saturating). The counters are u32: for 23-mers tf is 32 bits
// (.tf.bin), so a saturated counter can never admit another offset; a 13-mer table is u64, and positions_fill REFUSES a table with an entry
// above 2^32 - 1 (hipErrorNotSupported -> AIX_ERR_UNSUPPORTED) instead of letting a saturated counter overwrite earlier slots.
// One writer per bucket: the lane that holds the last element of h's run.
__global__ void __launch_bounds__(kB) k_tf13_above_u32(const uint64_t* __restrict__ tf, uint64_t n, uint32_t* __restrict__ flag) {
    const uint64_t stride = (uint64_t)gridDim.x * kB;
    bool any = false;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < n; i += stride) any |= (tf[i] >> 32) != 0;
    if (any) *flag = 1u;
}
__global__ void __launch_bounds__(kB) k_a2_advance(const uint32_t* __restrict__ skeys, uint64_t nwin, uint32_t n, const uint32_t* __restrict__ first,
                                                  uint32_t* __restrict__ filled) {
    const uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t j = (uint64_t)blockIdx.x * kB + threadIdx.x; j < nwin; j += stride) {
        const uint32_t h = skeys[j];
        if (h >= n || (j + 1 < nwin && skeys[j + 1] == h)) continue;
        const uint64_t tot = (uint64_t)filled[h] + (j - first[h] + 1);
        filled[h] = tot > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tot;
    }
}

// occurrences per bucket (u64, += ): the shard-local part of the reference's ppositions[h] counters
__global__ void __launch_bounds__(kB) k_a2_tally(const uint32_t* __restrict__ keys, uint64_t nwin, uint32_t n, unsigned long long* __restrict__ counts) {
    const uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < nwin; i += stride) {
        const uint32_t h = keys[i];
        if (h < n) atomicAdd(&counts[h], 1ull);
    }
}

static void launch_a2_probe(const IndexDev& ix, const uint8_t* d_reads, uint64_t nwin, uint64_t start, uint32_t* keys, hipStream_t s) {
    if (ix.k == 13) hipLaunchKernelGGL(k_a2_probe13, dim3(grid_of(nwin)), dim3(kB), 0, s, ix.perm13, d_reads, nwin, start, keys);
    else {
        // lanes that share one bucket line (IndexDev::bk_lpp; the ABI hands in 2 unless the caller chose a width: like count23's slot probe,
        // nothing but a 4-byte slot leaves this kernel, and two lanes per line measured 28.5-29.9 against 29.6-30.7 ms per 5 M reads with eight,
        // three alternating repetitions on one box). AIX_A2_PROBE_LANES: A/B switch
        static const int forced = [] { const char* e = getenv("AIX_A2_PROBE_LANES"); const int v = e ? atoi(e) : 0; return v == 2 || v == 4 || v == 8 ? v : 0; }();
        const int lanes = forced ? forced : (int)ix.bk_lpp;
        if (lanes == 2) hipLaunchKernelGGL(k_a2_probe<2>, dim3(grid_of(nwin)), dim3(kB), 0, s, ix, d_reads, nwin, start, keys);
        else if (lanes == 4) hipLaunchKernelGGL(k_a2_probe<4>, dim3(grid_of(nwin)), dim3(kB), 0, s, ix, d_reads, nwin, start, keys);
        else hipLaunchKernelGGL(k_a2_probe<8>, dim3(grid_of(nwin)), dim3(kB), 0, s, ix, d_reads, nwin, start, keys);
    }
}

hipError_t exclusive_scan_u32(const uint32_t* d_in, uint32_t* d_out, uint64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    size_t tb = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, tb, d_in, d_out, 0u, (size_t)n, rocprim::plus<uint32_t>(), s);
    void* tmp = nullptr;
    if (e == hipSuccess) e = pool_alloc(&tmp, tb ? tb : 1);
    if (e == hipSuccess) e = rocprim::exclusive_scan(tmp, tb, d_in, d_out, 0u, (size_t)n, rocprim::plus<uint32_t>(), s);
    { const hipError_t es = hipStreamSynchronize(s); if (e == hipSuccess) e = es; }
    if (tmp) pool_free(tmp);
    return e;
}

// indices (device, n+1 entries). Returns hip error.
hipError_t positions_indices(const IndexDev& ix, uint64_t* d_indices, hipStream_t s) {
    const uint64_t n = ix.n;
    uint64_t* tf64 = nullptr;
    hipError_t e = pool_alloc((void**)&tf64, 8 * (n + 1));
    if (e != hipSuccess) return e;
    if (ix.k == 13) hipLaunchKernelGGL(k_tf13_copy, dim3(grid_of(n + 1)), dim3(kB), 0, s, ix.tf13_mphf, n, tf64);   // compute_aindex13.cpp:57-63
    else hipLaunchKernelGGL(k_tf_u64, dim3(grid_of(n + 1)), dim3(kB), 0, s, ix, n, tf64);
    size_t tmp_bytes = 0;
    e = rocprim::exclusive_scan(nullptr, tmp_bytes, tf64, d_indices, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), s);
    void* tmp = nullptr;
    if (e == hipSuccess) e = pool_alloc(&tmp, tmp_bytes ? tmp_bytes : 1);
    if (e == hipSuccess) e = rocprim::exclusive_scan(tmp, tmp_bytes, tf64, d_indices, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), s);
    { const hipError_t es = hipStreamSynchronize(s); if (e == hipSuccess) e = es; }     // also on error paths: pool blocks are released idle
    if (tmp) pool_free(tmp);
    pool_free(tf64);
    return e;
}

// positions (device, pre-zeroed, indices[n] entries) for a reads buffer in HBM. Windows are indexed with 32 bits inside a
// piece; longer buffers go piece by piece in ascending order, the per-bucket fill counters carried from piece to piece, which
// keeps every bucket's offsets ascending (the reference's 1-thread order) for any buffer length.
// d_counts (device, u64[n], pre-zeroed): how often every bucket occurs in the buffer under A2's window rules
hipError_t positions_bucket_counts(const IndexDev& ix, const uint8_t* d_reads, uint64_t len, uint64_t start, unsigned long long* d_counts, hipStream_t s) {
    if (len < ix.k || ix.n == 0) return hipSuccess;
    const uint64_t nwin_all = len - (ix.k - 1);
    const uint64_t pw = std::min<uint64_t>(1ull << 30, nwin_all);
    uint32_t* keys = nullptr;
    hipError_t e = pool_alloc((void**)&keys, 4 * pw);
    for (uint64_t w0 = 0; e == hipSuccess && w0 < nwin_all; w0 += pw) {
        const uint64_t nwin = std::min(pw, nwin_all - w0);
        launch_a2_probe(ix, d_reads + w0, nwin, start > w0 ? start - w0 : 0, keys, s);
        hipLaunchKernelGGL(k_a2_tally, dim3(grid_of(nwin)), dim3(kB), 0, s, keys, nwin, (uint32_t)ix.n, d_counts);
        e = hipGetLastError();
    }
    { const hipError_t es = hipStreamSynchronize(s); if (e == hipSuccess) e = es; }
    if (keys) pool_free(keys);
    return e;
}

// filled_init (device, u32[n], may be null = zeros): occurrences of every bucket in the shards BEFORE this buffer;
// base_offset: byte offset of this buffer inside the whole reads file (offsets are reported file-relative).
hipError_t positions_fill(const IndexDev& ix, const uint8_t* d_reads, uint64_t len, uint64_t start, const uint64_t* d_indices, uint64_t* d_positions,
                          uint64_t piece, const uint32_t* filled_init, uint64_t base_offset, hipStream_t s, uint32_t* backend_out) {
    if (backend_out) *backend_out = 0;
    if (len < ix.k || ix.n == 0) return hipSuccess;
    const uint64_t nwin_all = len - (ix.k - 1);
    if (piece == 0 || piece > (1ull << 31)) piece = 1ull << 30;
    const uint64_t pw = std::min(piece, nwin_all);
    if (ix.k == 13) {                                                           // the fill counters are 32 bits wide (see k_a2_advance)
        uint32_t* d_flag = nullptr;
        uint32_t flag = 0;
        hipError_t ef = pool_alloc((void**)&d_flag, 4);
        if (ef == hipSuccess) ef = hipMemsetAsync(d_flag, 0, 4, s);
        if (ef == hipSuccess) { hipLaunchKernelGGL(k_tf13_above_u32, dim3(grid_of(ix.n)), dim3(kB), 0, s, ix.tf13_mphf, ix.n, d_flag); ef = hipGetLastError(); }
        if (ef == hipSuccess) ef = hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, s);
        { const hipError_t es = hipStreamSynchronize(s); if (ef == hipSuccess) ef = es; }
        if (d_flag) pool_free(d_flag);
        if (ef != hipSuccess) return ef;
        if (flag) return hipErrorNotSupported;
    }
    uint32_t *keys = nullptr, *skeys = nullptr, *svals = nullptr, *first = nullptr, *filled = nullptr;
    void* tmp = nullptr;
    hipError_t e = pool_alloc((void**)&keys, 4 * pw);
    if (e == hipSuccess) e = pool_alloc((void**)&filled, 4 * ix.n);
    if (e == hipSuccess) e = filled_init ? hipMemcpyAsync(filled, filled_init, 4 * ix.n, hipMemcpyDeviceToDevice, s) : hipMemsetAsync(filled, 0, 4 * ix.n, s);
    unsigned end_bit = 1;
    while (end_bit < 32 && (ix.n >> end_bit)) ++end_bit;                        // keys are in [0, n]
    size_t tmp_bytes = 0;
    rocprim::counting_iterator<uint32_t> iota(0);
    for (uint64_t w0 = 0; e == hipSuccess && w0 < nwin_all; w0 += pw) {
        const uint64_t nwin = std::min(pw, nwin_all - w0);
        const uint64_t rel_start = start > w0 ? start - w0 : 0;               // windows before `start` get no bucket (hash.cpp:973-986)
        const bool more = w0 + pw < nwin_all;
        launch_a2_probe(ix, d_reads + w0, nwin, rel_start, keys, s);
        e = hipGetLastError();
        if (e != hipSuccess) break;
        if (a2_msd_eligible(nwin, ix.n)) {                                      // grouping by MSD partition + per-bucket LDS stage (aix_a2msd.hip)
            bool untouched = false;
            e = a2_msd_place(ix, keys, nwin, base_offset + w0, filled, more, d_indices, d_positions, s, &untouched);
            if (e == hipSuccess && backend_out) *backend_out |= 2u;
            if (e == hipSuccess || !untouched) continue;
            (void)hipGetLastError();                                            // its workspace (16 B per window + chunk slack) did not fit: nothing was written,
            e = hipSuccess;                                                     // the sort path below needs about half of that
        }
        if (backend_out) *backend_out |= 1u;
        if (!skeys) {                                                           // short buffers: one stable radix sort of (bucket, offset)
            e = pool_alloc((void**)&skeys, 4 * pw);
            if (e == hipSuccess) e = pool_alloc((void**)&svals, 4 * pw);
            if (e == hipSuccess) e = pool_alloc((void**)&first, 4 * ix.n);
            if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, skeys, iota, svals, (size_t)pw, 0u, end_bit, s);
            if (e == hipSuccess) e = pool_alloc(&tmp, tmp_bytes ? tmp_bytes : 1);
            if (e != hipSuccess) break;
        }
        size_t tb = tmp_bytes;                                                  // sized for pw >= nwin elements
        e = rocprim::radix_sort_pairs(tmp, tb, keys, skeys, iota, svals, (size_t)nwin, 0u, end_bit, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_a2_first, dim3(grid_of(nwin)), dim3(kB), 0, s, skeys, nwin, (uint32_t)ix.n, first);
            hipLaunchKernelGGL(k_a2_place, dim3(grid_of(nwin)), dim3(kB), 0, s, ix, skeys, svals, nwin, base_offset + w0, first, filled, d_indices, d_positions);
            if (more) hipLaunchKernelGGL(k_a2_advance, dim3(grid_of(nwin)), dim3(kB), 0, s, skeys, nwin, (uint32_t)ix.n, first, filled);
            e = hipGetLastError();
        }
    }
    { const hipError_t es = hipStreamSynchronize(s); if (e == hipSuccess) e = es; }
    if (tmp) pool_free(tmp);
    if (keys) pool_free(keys);
    if (skeys) pool_free(skeys);
    if (svals) pool_free(svals);
    if (first) pool_free(first);
    if (filled) pool_free(filled);
    return e;
}

// ---------------------------------------------------------------------------------------------
// K1 back end: distinct canonical k-mers and their counts = sort + run-length of the window codes.
// codes: nwin entries from k_window_codes (invalid windows = ~0). Outputs are hipMalloc'd device arrays.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kB) k_mark_invalid(uint64_t* __restrict__ codes, uint64_t nwin, uint64_t sentinel) {
    const uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < nwin; i += stride)
        if (codes[i] == ~0ULL) codes[i] = sentinel;
}
__global__ void __launch_bounds__(kB) k_flag_min(const uint32_t* __restrict__ counts, uint64_t runs, const uint64_t* __restrict__ keys, uint64_t sentinel,
                                                uint32_t min_count, uint8_t* __restrict__ flags) {
    const uint64_t stride = (uint64_t)gridDim.x * kB;
    for (uint64_t i = (uint64_t)blockIdx.x * kB + threadIdx.x; i < runs; i += stride)
        flags[i] = (keys[i] != sentinel && counts[i] >= min_count) ? 1 : 0;
}

hipError_t distinct_from_codes(uint64_t* d_codes /* clobbered */, uint64_t nwin, int k, uint64_t min_count, uint64_t** d_keys_out, uint32_t** d_counts_out,
                               uint64_t* n_out, hipStream_t s) {
    *d_keys_out = nullptr; *d_counts_out = nullptr; *n_out = 0;
    if (nwin == 0) return hipSuccess;
    const uint64_t sentinel = 1ULL << (2 * k);                 // k <= 31: one bit above every valid code
    hipLaunchKernelGGL(k_mark_invalid, dim3(grid_of(nwin)), dim3(kB), 0, s, d_codes, nwin, sentinel);
    uint64_t *sorted = nullptr, *ukeys = nullptr, *fkeys = nullptr;
    uint32_t *ucnt = nullptr, *fcnt = nullptr;
    uint64_t* d_runs = nullptr;
    uint8_t* flags = nullptr;
    void* tmp = nullptr;
    size_t tb = 0;
    hipError_t e = pool_alloc((void**)&sorted, 8 * nwin);
    if (e == hipSuccess) e = rocprim::radix_sort_keys(nullptr, tb, d_codes, sorted, (size_t)nwin, 0u, (unsigned)(2 * k + 1), s);
    if (e == hipSuccess) e = pool_alloc(&tmp, tb ? tb : 1);
    if (e == hipSuccess) e = rocprim::radix_sort_keys(tmp, tb, d_codes, sorted, (size_t)nwin, 0u, (unsigned)(2 * k + 1), s);
    if (tmp) { (void)hipStreamSynchronize(s); pool_free(tmp); tmp = nullptr; }
    // run-length encode into the (now free) input buffer as unique keys, plus counts
    ukeys = d_codes;
    if (e == hipSuccess) e = pool_alloc((void**)&ucnt, 4 * nwin);
    if (e == hipSuccess) e = pool_alloc((void**)&d_runs, 8);
    tb = 0;
    if (e == hipSuccess) e = rocprim::run_length_encode(nullptr, tb, sorted, (unsigned int)nwin, ukeys, ucnt, d_runs, s);   // size query
    if (e == hipSuccess) e = pool_alloc(&tmp, tb ? tb : 1);
    uint64_t runs = 0;
    if (e == hipSuccess) {
        // rocPRIM takes the input size as unsigned int: encode in slices of < 2^31 and stitch equal boundary keys on the host
        // side is avoided by requiring nwin < 2^32 here (the ABI checks len < 2^32)
        e = rocprim::run_length_encode(tmp, tb, sorted, (unsigned int)nwin, ukeys, ucnt, d_runs, s);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&runs, d_runs, 8, hipMemcpyDeviceToHost, s);
    { const hipError_t es = hipStreamSynchronize(s); if (e == hipSuccess) e = es; }
    if (tmp) { pool_free(tmp); tmp = nullptr; }
    if (sorted) { pool_free(sorted); sorted = nullptr; }
    // filter: drop the sentinel run and runs below min_count
    uint64_t* d_sel = nullptr;
    uint64_t kept = 0;
    if (e == hipSuccess && runs) {
        e = pool_alloc((void**)&flags, runs);
        if (e == hipSuccess) e = pool_alloc((void**)&fkeys, 8 * runs);
        if (e == hipSuccess) e = pool_alloc((void**)&fcnt, 4 * runs);
        if (e == hipSuccess) e = pool_alloc((void**)&d_sel, 8);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_flag_min, dim3(grid_of(runs)), dim3(kB), 0, s, ucnt, runs, ukeys, sentinel,
                               (uint32_t)(min_count > 0xFFFFFFFFull ? 0xFFFFFFFFull : min_count), flags);
            tb = 0;
            e = rocprim::select(nullptr, tb, ukeys, flags, fkeys, d_sel, (size_t)runs, s);
        }
        if (e == hipSuccess) e = pool_alloc(&tmp, tb ? tb : 1);
        if (e == hipSuccess) e = rocprim::select(tmp, tb, ukeys, flags, fkeys, d_sel, (size_t)runs, s);
        if (e == hipSuccess) e = rocprim::select(tmp, tb, ucnt, flags, fcnt, d_sel, (size_t)runs, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&kept, d_sel, 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    (void)hipStreamSynchronize(s);
    if (tmp) pool_free(tmp);
    if (flags) pool_free(flags);
    if (d_sel) pool_free(d_sel);
    if (d_runs) pool_free(d_runs);
    if (ucnt) pool_free(ucnt);
    if (e != hipSuccess) {
        if (fkeys) pool_free(fkeys);
        if (fcnt) pool_free(fcnt);
        return e;
    }
    *d_keys_out = fkeys; *d_counts_out = fcnt; *n_out = kept;
    return hipSuccess;
}

}  // namespace aix
