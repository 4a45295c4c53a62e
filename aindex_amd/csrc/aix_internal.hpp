// aix_internal.hpp — host-side view of an index resident in HBM + kernel launcher prototypes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aix_device.hpp"

namespace aix {

// what kernels receive by value
struct IndexDev {
    MphfDev m;
    const KeyRec* keys;       // 23-mer: n records {code, tf} — only while an index is being built, or when no verification table was built
    // 23-mer with a verification table: checker[h] / tf[h] are NOT stored a second time. side[h] says where slot h's key lives: an entry of
    // the table (filed keys: 99 %), or — top bit set — a record of `unfiled` (keys beyond the eighth of their bucket, keys that are not in
    // their own MPHF slot). key_at() below is the one accessor.
    const uint32_t* side;
    const BkEntry* bk_store;  // the table's storage (also when the table is switched off for probing)
    const KeyRec* unfiled;
    uint64_t n;               // 23-mer: number of keys; 13-mer: 4^13
    const uint64_t* tf13_code;  // 13-mer: tf in 2-bit-code order (u64[4^13])
    const uint64_t* tf13_mphf;  // 13-mer: tf in mphf order (the file's order)
    const uint32_t* perm13;     // 13-mer: 2-bit code -> mphf slot
    uint32_t canonical_only;  // 23-mer: every stored code <= its reverse complement
    uint32_t k;
    uint32_t use_fp;          // 23-mer: the fingerprint nibbles of the MPHF records are populated
    uint32_t early_exit;      // 23-mer: presence masks populated and the early-exit walk enabled
    const BkEntry* bk;        // 23-mer: verification table (nb buckets of 8 entries), nullptr when not built / switched off
    uint32_t nb;
    uint32_t bk_lpp;          // lanes that share one bucket read (8, 4, 2 or 1); launch-time choice
    const uint64_t* bloom;    // 23-mer: absence filter in front of the table (nbloom 64-bit words), nullptr when off
    uint32_t nbloom;
    const BkEntry* mk;        // 23-mer: minimizer-keyed copy of the table for the streaming counter: the entries of bucket b are mk[mk_off[b] .. mk_off[b + 1]); nullptr when off
    uint32_t nbm;
    const uint32_t* mk_off;   // nbm + 1 offsets into mk (a bucket = all filed keys whose minimizer hashes to it: no bucket overflows at build time)
    uint32_t mk_cap;          // entries of a bucket a lane reads (<= AIX_MK_ENTRIES); a longer bucket leaves its windows to the hash-keyed table
};

#define AIX_SIDE_UNFILED 0x80000000u
// checker[h] and tf[h] of PHASH_MAP (hash.hpp:82-121) for slot h < n
__device__ __forceinline__ KeyRec key_at(const IndexDev& ix, uint64_t h) {
    if (ix.keys) return ix.keys[h];
    const uint32_t s = ix.side[h];
    if (s & AIX_SIDE_UNFILED) return ix.unfiled[s & ~AIX_SIDE_UNFILED];
    const BkEntry e = ix.bk_store[s];
    KeyRec r;
    r.code = (uint64_t)e.code_lo | ((uint64_t)(e.code_hi & 0x3FFFu) << 32);
    r.tf = e.tf;
    r.pad = 0;
    return r;
}

enum LookupMode { MODE_TF = 0, MODE_HASH = 1, MODE_KIDSTRAND = 2, MODE_BOTH = 3, MODE_TOTAL = 4, MODE_LINES = 5 };

struct LookupOut {
    uint32_t* tf;       // MODE_TF
    uint64_t* u64a;     // HASH: hash; KIDSTRAND: kid; BOTH: fwd; TOTAL: sum
    uint64_t* u64b;     // BOTH: rc
    uint8_t* strand;    // KIDSTRAND
};

// scratch pool (aix_pool.hip): cached device blocks for per-call temporaries; release only after the using stream is synchronised
hipError_t pool_alloc(void** out, size_t bytes);
void pool_free(void* p);
void pool_trim();

// launchers (aix_kernels.hip). All asynchronous on `stream`; return hipGetLastError().
hipError_t launch_lookup23_ascii(const IndexDev& ix, const uint8_t* q, uint64_t N, int mode, LookupOut out, hipStream_t s);
hipError_t launch_lookup23_codes(const IndexDev& ix, const uint64_t* codes, uint64_t N, uint32_t* out, hipStream_t s);
hipError_t launch_selftest_lower_bound(const uint16_t* a, uint32_t n, const uint32_t* keys, uint32_t nkeys, uint32_t* out, hipStream_t s);
hipError_t launch_lookup23_ragged(const IndexDev& ix, const uint8_t* bytes, const uint64_t* offs, uint64_t N, uint32_t* out, hipStream_t s);
hipError_t launch_lookup13_ascii(const IndexDev& ix, const uint8_t* q, uint64_t N, int mode, LookupOut out, hipStream_t s);
hipError_t launch_lookup13_ragged(const IndexDev& ix, const uint8_t* bytes, const uint64_t* offs, uint64_t N, uint32_t* out, hipStream_t s);
hipError_t launch_coverage(const IndexDev& ix, const uint8_t* seqs, const uint64_t* offs, uint64_t M, uint64_t total_bytes,
                           uint32_t cutoff, uint32_t* out, const uint64_t* out_offs, hipStream_t s);

// index construction helpers
hipError_t launch_build_keyrecs(const uint64_t* checker, const uint32_t* tf, uint64_t n, KeyRec* recs, uint32_t* noncanon_count, hipStream_t s);
// fp nibbles of the MPHF records (do_fp) and / or the presence masks of the early-exit table (ee_rw non-null, initialised here); keys through key_at(ix, .)
hipError_t launch_set_fingerprints(const IndexDev& ix, BvRec* recs_rw, EeRec* ee_rw, bool do_fp, hipStream_t s);
// verification table: bk (nb * 8 entries) is initialised and filled from the keys that sit in their own MPHF slot;
// fill = nb zeroed u32 counters (scratch)
hipError_t launch_build_buckets(const MphfDev& m, const KeyRec* keys, uint64_t n, BkEntry* bk, uint32_t nb, uint32_t* fill, uint64_t* bloom /* zeroed, nullable */,
                                uint32_t nbloom, uint32_t nbm /* 0: no minimizer-keyed copy */, uint32_t* mfill /* nbm zeroed words: keys per minimizer bucket */,
                                uint32_t* side /* n words: entry index of every filed key, 0xFFFFFFFF for the others */, hipStream_t s);
// the keys the table does not hold, closed up: unfiled[idx] = keys[i], side[i] = AIX_SIDE_UNFILED | idx for every side[i] == 0xFFFFFFFF; *counter = zeroed word
hipError_t launch_side_unfiled(const KeyRec* keys, uint64_t n, uint32_t* side, KeyRec* unfiled, uint32_t* counter, hipStream_t s);
hipError_t launch_count_unfiled(const uint32_t* side, uint64_t n, uint32_t* counter, hipStream_t s);
// second pass of the minimizer-keyed copy: entries written at mk[off[bucket] + arrival]; mcur = nbm zeroed words
hipError_t launch_fill_minimizer_table(const MphfDev& m, const KeyRec* keys, uint64_t n, BkEntry* mk, const uint32_t* off, uint32_t nbm, uint32_t* mcur, hipStream_t s);
hipError_t launch_scatter23(const MphfDev& m, uint64_t n, uint64_t nslots, const uint8_t* keys_ascii /* or */, const uint64_t* codes, const uint32_t* counts,
                            uint64_t* checker, uint32_t* tf, uint32_t* occupied_bits, uint32_t* conflict, hipStream_t s);
hipError_t launch_perm13(const MphfDev& m, uint32_t* perm /* [4^13]: code -> mphf index */, hipStream_t s);
hipError_t launch_tf13_to_code_order(const uint32_t* perm, const uint64_t* tf_mphf, uint64_t* tf_code, hipStream_t s);
hipError_t launch_extract_tf(const IndexDev& ix, uint32_t* tf, uint64_t* checker, hipStream_t s);

// counting
hipError_t launch_count13_plain(const uint8_t* buf, uint64_t len, unsigned long long* table_code /* [4^13] */, hipStream_t s);
uint64_t count13_workspace_bytes(uint64_t len);
// perm/out_mphf set: counters are written straight into the (pre-zeroed) mphf-ordered output; else into table_code
hipError_t launch_count13_partitioned(const uint8_t* buf, uint64_t len, void* workspace, unsigned long long* table_code, const uint32_t* perm,
                                      uint64_t* out_mphf, int accumulate, hipStream_t s);
// count23 without global atomics: tf_out[slot] += occurrences of slot in the stream (0xFFFFFFFF = skip); one pass per 2^26 slots of the key set
hipError_t launch_histogram_slots(const uint32_t* d_slots, uint64_t nslots, void* workspace /* count13_workspace_bytes(nslots + 12) */, uint32_t* tf_out,
                                  uint64_t n, hipStream_t s, uint32_t range_bits = 26 /* slots per pass = 2^range_bits <= 2^26 */, uint32_t* passes_out = nullptr);
// the same slot stream from the minimizer-keyed table (aix_stream23.hip): a lane owns 32 consecutive windows; needs ix.mk
hipError_t launch_stream23_slots(const IndexDev& ix, const uint8_t* buf, uint64_t len, int canon_mode, uint32_t* slots /* [len - 22] */,
                                 uint32_t* flag /* zeroed word: set when a window stayed undecided */, hipStream_t s);
hipError_t launch_probe23_slots(const IndexDev& ix, const uint8_t* buf, uint64_t len, int canon_mode, uint32_t* slots /* [len - 22] */, hipStream_t s);
// the same slot stream with a run of w (16 / 32) consecutive windows per lane (aix_stream23.hip: the bytes are encoded once per run); needs ix.bk
hipError_t launch_run23_slots(const IndexDev& ix, const uint8_t* buf, uint64_t len, int canon_mode, uint32_t* slots, int w, hipStream_t s);
hipError_t launch_scatter13_to_mphf(const uint32_t* perm, const unsigned long long* table_code, uint64_t* out_mphf, int add, hipStream_t s);
hipError_t launch_perm13_check(const uint32_t* perm, uint32_t* bits /* 4^13 / 32 zeroed words */, uint32_t* bad /* zeroed */, hipStream_t s);
hipError_t launch_count23_fixed(const IndexDev& ix, const uint8_t* buf, uint64_t len, int canon_mode, uint32_t* tf_out, hipStream_t s);
hipError_t launch_add_counts23(const IndexDev& ix, const uint64_t* keys /* distinct, in the call's canonical form */, const uint64_t* counts, uint64_t n, uint32_t* tf_out,
                               hipStream_t s);

hipError_t launch_window_codes(const uint8_t* buf, uint64_t len, int k, int canon_mode, uint64_t* out /* [len-k+1] */, hipStream_t s);

hipError_t launch_gather(const uint8_t* table, uint64_t n_elems, int elem_bytes, int unroll, uint64_t n_access, uint64_t seed, uint64_t* sink, hipStream_t s);

// out[i] = in[0] + ... + in[i - 1] (device arrays of n words; synchronises the stream)
hipError_t exclusive_scan_u32(const uint32_t* d_in, uint32_t* d_out, uint64_t n, hipStream_t s);

// positions index (aix_positions.hip)
hipError_t positions_indices(const IndexDev& ix, uint64_t* d_indices /* n+1 */, hipStream_t s);
hipError_t positions_fill(const IndexDev& ix, const uint8_t* d_reads, uint64_t len, uint64_t start, const uint64_t* d_indices, uint64_t* d_positions,
                          uint64_t piece, const uint32_t* filled_init, uint64_t base_offset, hipStream_t s,
                          uint32_t* backend_out = nullptr /* bit 0: a piece went through the stable radix sort, bit 1: through the MSD partition */);
// grouping of the probe's (bucket, offset) pairs without a full-width sort (aix_a2msd.hip): two-level MSD partition + per-bucket LDS stage
bool a2_msd_eligible(uint64_t nwin, uint64_t n);
hipError_t a2_msd_place(const IndexDev& ix, const uint32_t* keys, uint64_t nwin, uint64_t piece_first, uint32_t* filled, bool advance, const uint64_t* d_indices,
                        uint64_t* d_positions, hipStream_t s, bool* untouched = nullptr /* set when the call failed before writing anything (workspace allocation) */);
hipError_t positions_bucket_counts(const IndexDev& ix, const uint8_t* d_reads, uint64_t len, uint64_t start, unsigned long long* d_counts, hipStream_t s);

// distinct k-mers = sort + run-length of window codes; outputs hipMalloc'd (caller frees), d_codes is clobbered
hipError_t distinct_from_codes(uint64_t* d_codes, uint64_t nwin, int k, uint64_t min_count, uint64_t** d_keys_out, uint32_t** d_counts_out, uint64_t* n_out,
                               hipStream_t s);
// the same without a full-width sort (aix_k1.hip): MSD partition in two levels + per-bucket LDS hash / sort. d_codes is clobbered.
// *fell_back: a bucket held too many distinct remainders; nothing was produced and the caller takes the radix-sort path.
bool k1_msd_eligible(uint64_t nwin, int k);
// The two big temporaries of a K1 piece (8 B per window of code / remainder staging, ~8 B per window of partition chunks: 16 + 17 GiB for a
// 2^31-window piece), kept by whoever counts piece after piece: a hipMalloc / hipFree pair of that size costs ~0.2 s, more than the kernels
// of the piece (measured: 60 M reads in 5 pieces took 2.5 s of wall clock around 0.16 s of kernels before the blocks were kept).
struct K1Scratch {
    void* codes = nullptr;
    size_t codes_bytes = 0;
    void* work = nullptr;
    size_t work_bytes = 0;
    hipStream_t s = nullptr;
    K1Scratch() = default;
    K1Scratch(const K1Scratch&) = delete;
    K1Scratch& operator=(const K1Scratch&) = delete;
    hipError_t need(void** p, size_t* have, size_t bytes) {
        if (bytes <= *have) return hipSuccess;
        if (*p) { (void)hipStreamSynchronize(s); pool_free(*p); *p = nullptr; *have = 0; }
        const hipError_t e = pool_alloc(p, bytes);              // from the block cache: a process that counts buffer after buffer finds them there again
        if (e == hipSuccess) *have = bytes;
        return e;
    }
    ~K1Scratch() {
        if (codes || work) (void)hipStreamSynchronize(s);
        if (codes) pool_free(codes);
        if (work) pool_free(work);
    }
};
hipError_t distinct_from_codes_msd(uint64_t* d_codes, uint64_t nwin, int k, uint64_t** d_keys_out, uint32_t** d_counts_out, uint64_t* n_out, bool* fell_back,
                                   hipStream_t s, const uint8_t* d_plain = nullptr /* encode inside level 1 */, uint64_t plen = 0, int canon_mode = 0,
                                   uint64_t** d_counts64_out = nullptr /* non-null: the counts come back as u64 here, *d_counts_out stays null */,
                                   K1Scratch* scratch = nullptr /* non-null: the partition workspace lives there, kept for the next piece */);
hipError_t distinct_from_plain(const uint8_t* d_plain, uint64_t plen, int k, int canon_mode, uint64_t min_count, uint64_t piece, uint64_t** d_keys_out,
                               uint64_t** d_counts_out, uint64_t* n_out, hipStream_t s);
hipError_t merge_counts(const uint64_t* d_keys, const uint64_t* d_counts, uint64_t n, uint64_t min_count, uint64_t** d_keys_out, uint64_t** d_counts_out,
                        uint64_t* n_out, hipStream_t s);

// sorted (key, count) runs merged without a sort (aix_merge.hip)
struct DevArr {                       // RAII holder of a pool block used on stream `st`: the stream is synchronised before the block goes back
    void* p = nullptr;
    hipStream_t st = nullptr;
    DevArr() = default;
    explicit DevArr(hipStream_t s) : st(s) {}
    DevArr(const DevArr&) = delete;
    DevArr& operator=(const DevArr&) = delete;
    void drop() { if (p) { (void)hipStreamSynchronize(st); pool_free(p); p = nullptr; } }
    ~DevArr() { drop(); }
    hipError_t alloc(size_t bytes) { drop(); return pool_alloc(&p, bytes ? bytes : 1); }
    void* release() { void* q = p; p = nullptr; return q; }
};
hipError_t merge_sum_runs(const uint64_t* ak, const uint64_t* ac, uint64_t na, const uint64_t* bk, const uint64_t* bc, uint64_t nb, uint64_t** d_keys_out,
                          uint64_t** d_counts_out, uint64_t* n_out, hipStream_t s);
hipError_t merge_runs(const uint64_t* d_keys, const uint64_t* d_counts, const uint64_t* offs /* host, nruns + 1 */, uint32_t nruns, uint64_t min_count,
                      uint64_t** d_keys_out, uint64_t** d_counts_out, uint64_t* n_out, hipStream_t s);
// the distinct k-mers of everything added so far; a piece = all k-windows of one PLAIN buffer (<= 2^31 windows)
class DistinctAcc {
public:
    DistinctAcc(int k, int canon_mode, hipStream_t s);
    ~DistinctAcc();
    DistinctAcc(const DistinctAcc&) = delete;
    DistinctAcc& operator=(const DistinctAcc&) = delete;
    hipError_t add_plain(const uint8_t* d_plain, uint64_t plen);
    hipError_t finish(uint64_t min_count, uint64_t** d_keys_out, uint64_t** d_counts_out, uint64_t* n_out);   // pool blocks, the caller frees
    uint64_t size() const { return acc_n; }
    uint64_t pieces = 0, merges = 0;
private:
    int k, canon_mode;
    hipStream_t s;
    K1Scratch scratch;
    uint64_t* acc_k = nullptr;
    uint64_t* acc_c = nullptr;
    uint64_t acc_n = 0;
};

// record normalisation on the device (aix_normalize.hip); d_out holds len + 1 bytes
hipError_t normalise_device(const uint8_t* d_raw, uint64_t len, int format, int fasta_mode, uint8_t* d_out, uint64_t* out_len, hipStream_t s);
// one part of an input normalised piece by piece: *state_io = reader state handed from part to part (AIX_NORM_START before the first)
static constexpr uint32_t AIX_NORM_START = 0xFFFFFFFFu;
hipError_t normalise_device_part(const uint8_t* d_raw, uint64_t len, int format, int fasta_mode, uint8_t* d_out, uint64_t* out_len, uint32_t* state_io, int last,
                                 hipStream_t s);

// synthetic generators
hipError_t launch_synth_genome(uint64_t seed, uint64_t length, uint8_t* out, hipStream_t s);
hipError_t launch_synth_kmers(uint64_t seed, uint64_t first, uint64_t N, int k, uint8_t* out, hipStream_t s);
hipError_t launch_synth_mix23(uint64_t seed, const uint8_t* genome, uint64_t glen, uint64_t first, uint64_t N, uint8_t* out, hipStream_t s);
hipError_t launch_synth_reads(uint64_t seed, const uint8_t* genome, uint64_t genome_len, uint64_t first_read, uint64_t n_reads,
                              uint32_t read_len, int rc_half, uint32_t n_rate_ppm, uint8_t* out, hipStream_t s);

}  // namespace aix
