/* include/aindex_hip.h — C ABI of libaindex_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the k-mer count / perfect-hash lookup path of ad3002/aindex.
 * Every entry point names the reference interface it replaces (file:line under /root/reference).
 * Plain pointers and sizes only; no C++/torch types. All functions return an int status
 * (AIX_OK = 0, negative = error, see aix_strerror) and never abort the process — the reference's
 * std::terminate()/exit(10|12) paths (python_wrapper.cpp:265,413,1118; hash.cpp:37,129,153) become
 * error codes. Queries are thread-safe per handle. The library fails loudly (AIX_ERR_HIP) when no
 * HIP device is usable: there is NO CPU fallback behind this ABI.
 *
 * Naming: `*_dev` entry points take DEVICE pointers and a hipStream_t (passed as void*) and are
 * asynchronous on that stream; the un-suffixed twins take HOST pointers, stage through HBM and
 * return when the result is in the caller's buffer.
 */
#ifndef AINDEX_HIP_H
#define AINDEX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIX_OK               0
#define AIX_ERR_ARG         -1   /* bad argument / NULL handle                                  */
#define AIX_ERR_IO          -2   /* file missing or short (reference: std::terminate / exit(10)) */
#define AIX_ERR_FORMAT      -3   /* malformed .pf / size mismatch                                */
#define AIX_ERR_NOMEM       -4
#define AIX_ERR_HIP         -5   /* HIP runtime error or no device                               */
#define AIX_ERR_UNSUPPORTED -6   /* e.g. n >= 2^32 keys                                          */
#define AIX_ERR_MODE        -7   /* 13-mer call on a 23-mer handle or vice versa                 */
#define AIX_ERR_CONFLICT   -12   /* index scatter collision (reference: exit(12), hash.cpp:708)  */

#define AIX_TOTAL_13MERS 67108864ull   /* 4^13, python_wrapper.cpp:141 */

/* input formats of the counters (count_kmers13.cpp:187-206) */
#define AIX_FMT_AUTO  -1
#define AIX_FMT_PLAIN  0
#define AIX_FMT_FASTA  1
#define AIX_FMT_FASTQ  2

/* canonical form used when counting 23-mers */
#define AIX_CANON_NONE    0
#define AIX_CANON_REF_X86 1   /* bit-exact with kmer_counter's defective rc, count_kmers.cpp:116-136 */
#define AIX_CANON_TRUE_RC 2   /* true reverse complement (tests/analyze_kmers.py:19-80)              */

typedef struct aix_index aix_index_t;   /* opaque: MPHF + tf/checker resident in HBM */

typedef struct {
    uint32_t k;               /* 23 or 13                                                     */
    uint32_t device;          /* HIP device ordinal                                            */
    uint64_t n;               /* keys: size(.kmers.bin)/8 (23) or 4^13 (13)                    */
    uint64_t mphf_n;          /* n stored in the .pf                                           */
    uint64_t hash_domain;     /* m_hash_domain, mphf.hpp:26                                    */
    uint64_t seed;            /* jenkins64 seed                                                */
    uint64_t bitpairs;        /* 3 * hash_domain                                               */
    uint64_t device_bytes;    /* HBM held by this handle                                       */
    uint32_t canonical_only;  /* 1 if every stored 23-mer code <= its reverse complement       */
    uint32_t bucket_table;    /* 1 if the verification table is built and switched on          */
    uint64_t buckets;         /* its 128-byte buckets (0: not built)                           */
    uint64_t bucket_unfiled_keys; /* keys beyond the eighth of their bucket (MPHF path)        */
    uint32_t bucket_lanes;    /* lanes that share one bucket read (8, 4, 2 or 1)               */
    uint32_t absence_filter_words; /* 64-bit words of the absence filter in front of the table (0: off) */
    uint64_t minimizer_lines;     /* 128-byte lines of the minimizer-keyed copy of the table (0: off)   */
    uint64_t minimizer_unfiled_keys; /* keys whose chain of lines was full (answered by the hash-keyed table) */
    uint32_t count23_backend; /* the last aix_count23_fixed* call on this handle: 0 none yet, 1 memory-side atomics (short buffers,
                                 AIX_COUNT23_ATOMICS=1), 2 slot stream + LDS histogram, 3 distinct k-mers of the reads first (K1), one
                                 probe per distinct k-mer (buffers of >= 2^29 windows holding >= 8 windows per key; AIX_COUNT23_VIA_K1) */
    uint32_t count23_passes;  /* back end 2: passes over the slot stream = ceil(n / 2^26)                                           */
    uint32_t positions_backend; /* the last aix_positions_fill* call: bit 0 = a piece was grouped by the stable radix sort (short buffers, more than
                                   2^30 slots, workspace did not fit), bit 1 = by the MSD partition; 0 none yet                    */
    uint32_t reserved0;
} aix_info_t;

const char* aix_version(void);
const char* aix_strerror(int status);
int aix_device_count(int* count);
/* Calls that need multi-GB temporaries (K1, A1/A2, I1, host-buffer staging) take them from a per-device cache of device
 * blocks (a hipMalloc of that size costs ~20 ms). AIX_SCRATCH_CACHE_GB (default 40) bounds the cache; this and
 * aix_index_close() return it to the driver. */
void aix_scratch_trim(void);                     /* AIX_ERR_HIP if the runtime is unusable  */
/* Page-locked host memory for the buffers a binding hands to the host-pointer entry points (queries in, answers out): a pinned buffer
 * crosses the link at ~48 GB/s instead of ~32 GB/s through the library's own staging, and one kept between calls is not page-faulted again
 * (a fresh 460 MB buffer written by 32 threads costs more than the lookup it feeds). This is where the reference's binding holds the
 * std::vector<std::string> / std::vector<uint32_t> of get_tf_values (python_wrapper.cpp:653-664). Free with aix_host_free only. */
int aix_host_alloc(uint64_t bytes, void** out);
int aix_host_free(void* p);

/* ------------------------------------------------------------------------------------------
 * Index lifecycle.
 * replaces AindexWrapper::load / load_hash_file / load_from_prefix_23mer
 *          (python_wrapper.cpp:228-258,1103-1132) + load_hash (hash.cpp:367-450)
 *          AindexWrapper::load_13mer_index / load_from_prefix_13mer (python_wrapper.cpp:404-437,1162-1188)
 * File layouts are the reference's: .pf = mphf::save (mphf.hpp:99-105), .kmers.bin = u64[n],
 * .tf.bin = u32[n] (23-mer, compute_index.cpp:59-67) or u64[4^13] (13-mer, count_kmers13.cpp:358-388).
 * ------------------------------------------------------------------------------------------ */
/* Validate a .pf image on the HOST (no device needed): header u64 n, hash_domain, seed, bitpairs (mphf.hpp:99-113,
 * base_hash.hpp:116-119, ranked_bitpair_vector.hpp:78-84), bitpairs == 3 * hash_domain without wrap-around, node ids
 * within 32 bits, n <= bitpairs, and the image long enough for the bit-pair words and block ranks. Every open / create /
 * scatter entry point applies the same check before anything is uploaded: AIX_ERR_FORMAT instead of an out-of-bounds
 * read on the device (the reference trusts the header). hdr_out (nullable) receives the four header words. */
int aix_pf_check(const void* pf_bytes, uint64_t pf_len, uint64_t hdr_out[4]);
int aix_index_open_23(const char* pf_path, const char* tf_bin_path, const char* kmers_bin_path,
                      int device, aix_index_t** out);
int aix_index_open_13(const char* pf_path, const char* tf_bin_path /* NULL: all-zero table */,
                      int device, aix_index_t** out);
/* same, from caller memory (host pointers) */
int aix_index_create_23(const void* pf_bytes, uint64_t pf_len, const uint64_t* checker,
                        const uint32_t* tf, uint64_t n, int device, aix_index_t** out);
int aix_index_create_13(const void* pf_bytes, uint64_t pf_len, const uint64_t* tf /* 4^13 or NULL */,
                        int device, aix_index_t** out);
int aix_index_close(aix_index_t* h);                  /* ~AindexWrapper, python_wrapper.cpp:185-226 */
int aix_index_info(const aix_index_t* h, aix_info_t* info);
/* force the two-probe path even on an all-canonical index (A/B measurements) */
int aix_index_set_canonical_fastpath(aix_index_t* h, int enabled);
/* switch the 4-bit fingerprint filter of the MPHF records off/on (A/B measurements; answers are identical) */
int aix_index_set_fingerprint_filter(aix_index_t* h, int enabled);
/* switch the early-exit MPHF walk (presence masks: an absent key usually costs one record read) off/on */
int aix_index_set_early_exit(aix_index_t* h, int enabled);
/* Verification table of a 23-mer handle (built at open unless AIX_BUCKET_TABLE=0): one 128-byte line per probe holding
 * {code, tf, slot} of the keys filed under it, compared in-line — the hit of get_tf_value_23mer / get_kid_by_kmer
 * (python_wrapper.cpp:610-627, hash.hpp:700-716) and of lu_compressed_worker's probe (hash.cpp:993-1054) in ONE read instead of
 * three MPHF records + a key record; answers are identical with it on or off (A/B measurements, tests).
 * lanes: how many lanes share one bucket read (8, 4, 2, 1; 0 = keep). */
int aix_index_set_bucket_table(aix_index_t* h, int enabled, int lanes);
/* Absence filter in front of the verification table (lookups, coverage): a blocked Bloom filter of the filed keys, one cached
 * 8-byte read that answers most absent keys before the table is touched (AIX_BLOOM_BITS bits per key at open, default 16,
 * 0 = none). Off / on for A/B measurements; answers are identical. */
int aix_index_set_absence_filter(aix_index_t* h, int enabled);
/* Minimizer-keyed copy of the verification table, used by the streaming form of aix_count23_fixed*: every filed key once more,
 * grouped by the bucket of the minimizer of its 23-mer (the 15-mer with the smallest hash over both strands; offsets + entries,
 * 16 B per key + 4 B per bucket), so the ~7 consecutive windows of a read that share a minimizer read ONE bucket. EXPERIMENTAL:
 * built at open only with AIX_MINIMIZER_TABLE=1 (AIX_MINIMIZER_LOAD = mean keys per bucket, default 2); measured 6-12 % faster than
 * the hash-keyed probe at one key per bucket and slower at four, so it is not the default; off / on per handle for
 * A/B measurements; answers are identical (verification in the bucket; undecided windows are settled through the hash-keyed table). */
int aix_index_set_minimizer_table(aix_index_t* h, int enabled);
/* replace the tf table of a 13-mer handle (u64[4^13], mphf order, HOST pointer) */
int aix_index_set_tf_13(aix_index_t* h, const uint64_t* tf);
/* copy tf out (HOST pointer): 23 -> u32[n]; 13 -> u64[4^13] in mphf order
 * replaces get_13mer_tf_array / get_tf_by_index_13mer (python_wrapper.cpp:983-998) */
int aix_index_get_tf(const aix_index_t* h, void* out, uint64_t out_bytes);
/* copy the checker (stored 2-bit codes, u64[n]) out; get_kmer_by_kid / get_kmer_info (:718-755) */
int aix_index_get_checker(const aix_index_t* h, uint64_t* out, uint64_t n);

/* I1: replaces `compute_index <dat> <pf> <prefix> <threads> <mock>` (src/compute_index.cpp:34-72,
 * index_hash_pp / worker_for_fill_index src/hash.cpp:671-723,779-881): for every key i,
 * checker_out[mphf(key_i)] = 2-bit code, tf_out[mphf(key_i)] = counts[i] (counts NULL = mock, tf 0).
 * keys = n*23 ASCII bytes (the first column of the .dat), outputs are the .kmers.bin / .tf.bin images.
 * AIX_ERR_CONFLICT when two keys land in one slot or a slot >= n (reference: exit(12) / OOB write). */
int aix_index_scatter(const void* pf_bytes, uint64_t pf_len, const char* keys, const uint32_t* counts, uint64_t n,
                      int device, uint64_t* checker_out, uint32_t* tf_out);
/* the same for a SHARD of the key set (multi-GPU index construction): n_keys keys scatter into full-size
 * arrays of n_slots entries (zero where this shard wrote nothing); occupied_out = ceil(n_slots/32) words, bit h set
 * <=> slot h was written. The caller merges shards with sum(tf), max(checker) and detects cross-shard collisions as
 * overlapping occupied bits. AIX_ERR_CONFLICT (collision inside the shard) still fills the outputs. */
int aix_index_scatter_shard(const void* pf_bytes, uint64_t pf_len, const char* keys, const uint32_t* counts, uint64_t n_keys,
                            uint64_t n_slots, int device, uint64_t* checker_out, uint32_t* tf_out, uint32_t* occupied_out);
/* the same scatter from 2-bit codes already in HBM, keeping the result resident as a 23-mer handle */
int aix_index_build_23_codes_dev(const void* pf_bytes, uint64_t pf_len, const uint64_t* d_codes,
                                 const uint32_t* d_counts /* nullable */, uint64_t n, int device, void* stream,
                                 aix_index_t** out);

/* ------------------------------------------------------------------------------------------
 * Batch tf queries. `kmers` is N*k contiguous ASCII bytes (k = 23 or 13 per the handle).
 * 23-mer handle: AindexWrapper::get_tf_values / get_tf_values_23mer / get_tf_value_23mer
 *   (python_wrapper.cpp:610-627,653-664,1219-1228) bit-exact incl. the raw-bytes forward probe /
 *   sanitised reverse probe asymmetry for non-ACGT bytes and forward-strand precedence.
 * 13-mer handle: get_tf_values_13mer / get_tf_value_13mer (python_wrapper.cpp:482-503,938-980):
 *   strict upper-case ACGT else 0, forward strand only, u64 -> u32 truncation.
 * ------------------------------------------------------------------------------------------ */
int aix_tf_batch_ascii(aix_index_t* h, const char* kmers, uint64_t N, uint32_t* out);
int aix_tf_batch_ascii_dev(aix_index_t* h, const char* d_kmers, uint64_t N, uint32_t* d_out, void* stream);
/* pre-encoded 2-bit codes (first base most significant; ACGT only), same answers as the ASCII call */
int aix_tf_batch_codes(aix_index_t* h, const uint64_t* codes, uint64_t N, uint32_t* out);
int aix_tf_batch_codes_dev(aix_index_t* h, const uint64_t* d_codes, uint64_t N, uint32_t* d_out, void* stream);
/* variable-length queries: query i = bytes[offsets[i] .. offsets[i+1]). Mirrors what the reference
 * does with a std::string of any length (hash of ALL bytes, code of the first k); lengths < k give 0
 * (the reference reads past the string there). 23-mer handles only for len != k semantics; a 13-mer
 * handle returns 0 unless len == 13 (python_wrapper.cpp:943-946). */
int aix_tf_batch_ragged(aix_index_t* h, const char* bytes, const uint64_t* offsets, uint64_t N, uint32_t* out);
int aix_tf_batch_ragged_dev(aix_index_t* h, const char* d_bytes, const uint64_t* d_offsets, uint64_t N,
                            uint32_t* d_out, void* stream);

/* instrumentation for the roofline accounting: d_out[i] = MPHF + key records that tf query i reads (23-mer handles) */
int aix_lines_batch_ascii_dev(aix_index_t* h, const char* d_kmers, uint64_t N, uint32_t* d_out, void* stream);
/* get_hash_values / get_hash_value (python_wrapper.cpp:629-642): raw mphf::lookup of the bytes. On a 13-mer handle: the
 * value of hasher_13mer.lookup (:1087, used by get_positions_13mer) — the reference crashes there (hash_map is null). */
int aix_hash_batch_ascii(aix_index_t* h, const char* kmers, uint64_t N, uint64_t* out);
int aix_hash_batch_ascii_dev(aix_index_t* h, const char* d_kmers, uint64_t N, uint64_t* d_out, void* stream);
/* get_kid_by_kmer (:700-716; 0 when absent) and get_strand (:726-742; 0 absent,1 fwd,2 rc) */
int aix_kid_strand_batch_ascii(aix_index_t* h, const char* kmers, uint64_t N, uint64_t* kid_out /* nullable */,
                               uint8_t* strand_out /* nullable */);
int aix_kid_strand_batch_ascii_dev(aix_index_t* h, const char* d_kmers, uint64_t N, uint64_t* d_kid,
                                   uint8_t* d_strand, void* stream);
/* get_tf_both_directions_{23,13}mer_batch (:594-608,1259-1286) and, summed, get_total_tf_values_*
 * (:548-562,1230-1257). Both directions are written as u64; either pointer may be NULL. */
int aix_tf_both_batch_ascii(aix_index_t* h, const char* kmers, uint64_t N, uint64_t* fwd_out, uint64_t* rc_out);
int aix_tf_both_batch_ascii_dev(aix_index_t* h, const char* d_kmers, uint64_t N, uint64_t* d_fwd,
                                uint64_t* d_rc, void* stream);
int aix_tf_total_batch_ascii(aix_index_t* h, const char* kmers, uint64_t N, uint64_t* out);
int aix_tf_total_batch_ascii_dev(aix_index_t* h, const char* d_kmers, uint64_t N, uint64_t* d_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Coverage. replaces AIndex.get_sequence_coverage (aindex/core/aindex.py:314-322) for M sequences:
 * sequence s = seqs[offs[s] .. offs[s+1]); for every window i of length k (k = handle's k):
 * out[out_offs[s] + i] = tf >= cutoff ? tf : 0, tf as aix_tf_batch_ascii would give for that window.
 * out_offs[s+1]-out_offs[s] must be >= max(0, len_s - k + 1).
 * ------------------------------------------------------------------------------------------ */
int aix_coverage_batch(aix_index_t* h, const char* seqs, const uint64_t* offs, uint64_t M, uint32_t cutoff,
                       uint32_t* out, const uint64_t* out_offs);
int aix_coverage_batch_dev(aix_index_t* h, const char* d_seqs, const uint64_t* d_offs, uint64_t M,
                           uint64_t total_bytes, uint32_t cutoff, uint32_t* d_out, const uint64_t* d_out_offs,
                           void* stream);

/* ------------------------------------------------------------------------------------------
 * Counting.
 * aix_count13: replaces Kmer13Counter (count_kmers13.cpp:113-161,194-272,358-388): every length-13
 *   window of upper-cased ACGT in every sequence adds 1 to counts[mphf13(window)] (forward strand).
 *   tf_out = u64[4^13] in mphf order — byte-identical to the reference's output file. The handle's
 *   own tf table is NOT modified. `_dev`: d_buf must be in PLAIN form (one sequence per line, any
 *   non-ACGT byte breaks a window); use aix_normalize_reads for FASTA/FASTQ.
 * aix_count23_fixed: histogram of 23-mer windows against the handle's fixed key set (what
 *   kmer_counter -> compute_index would store for these reads, restricted to keys in the index):
 *   window chars valid per count_kmers.cpp:71-88 (ACGTU any case), key = canonical per canon_mode,
 *   tf_out[slot] += 1 when checker[slot] == key. tf_out = u32[n]. Multi-GPU: each rank counts its
 *   shard, then all-reduce(sum) tf_out (aindex_amd/dist.py).
 * ------------------------------------------------------------------------------------------ */
int aix_count13(aix_index_t* h, const char* buf, uint64_t len, int format, uint64_t* tf_out);
int aix_count13_dev(aix_index_t* h, const char* d_plain, uint64_t len, uint64_t* d_tf_out, void* stream);
int aix_count23_fixed(aix_index_t* h, const char* buf, uint64_t len, int format, int canon_mode, uint32_t* tf_out);
int aix_count23_fixed_dev(aix_index_t* h, const char* d_plain, uint64_t len, int canon_mode,
                          uint32_t* d_tf_out /* accumulated into, caller zeroes */, void* stream);
/* Streaming ingestion: the counters fed from a FILE, as the reference's tools are (count_kmers13.cpp:166-183,211-272,277-350: a reader
 * thread pushes sequences while the workers count). The file is read part by part (AIX_INGEST_PART_MB, default 256) by several host
 * threads into pinned staging, crosses the link on a copy stream while the previous part is normalised (FASTA / FASTQ readers as above,
 * their state handed from part to part, so parts are cut at any byte) and counted; HBM use is O(part), the host never holds the file.
 * Results are identical to the buffer forms. The host-buffer forms above (aix_count13, aix_count23_fixed, aix_count_distinct) run the same
 * pipeline from caller memory.
 * aix_count13_file: tf (u64[4^13], mphf order) is written to out_path (the file `count_kmers13 <in> <pf> <out>` writes,
 *   count_kmers13.cpp:358-388) and / or copied to tf_out (either may be NULL, not both).
 * stats (nullable) reports what the call did. */
typedef struct {
    uint64_t bytes_in;         /* bytes read from the file / buffer                                        */
    uint64_t plain_bytes;      /* bytes of PLAIN form that were counted                                    */
    uint64_t parts;            /* parts the input was cut into                                             */
    uint64_t part_bytes;       /* size of a part                                                           */
    uint64_t pieces;           /* K1: pieces whose distinct sets were merged                               */
    uint64_t pinned_bytes;     /* pinned host staging held by the call                                     */
    uint64_t device_bytes;     /* HBM held by the call: part staging + PLAIN part / piece + result (NOT the index, NOT the workspace)   */
    uint64_t workspace_bytes;  /* the handle's counting workspace after the call (grow-only, sized by the largest part it has seen)      */
    double seconds_total;      /* wall clock of the call                                                   */
    double seconds_read;       /* file / page cache -> pinned staging (sum over parts, overlapped)         */
    double seconds_wait;       /* the consumer waited for a part to arrive                                 */
    double seconds_h2d;        /* the parts on the link (HIP events around every copy, summed; the last three parts are not counted) */
    double seconds_normalise;  /* FASTA / FASTQ parts: the transducer pass (waits for the part to arrive included) */
    double seconds_compute;    /* counting as seen by the host (includes the waits inside)                 */
    double seconds_output;     /* result download + file write                                             */
} aix_ingest_stats_t;
/* Pin the staging blocks of a streaming call in the background and return at once (optional): a process that streams ONE file — a tool run —
 * calls this before it opens its index, so that the ~100 ms of page pinning overlap the index load instead of delaying the first part.
 * Staging blocks are kept between calls (AIX_PINNED_CACHE_MB, default 1024; aix_scratch_trim() frees them). */
int aix_ingest_warm(int device);
int aix_count13_file(aix_index_t* h, const char* path, int format, const char* out_path /* nullable */, uint64_t* tf_out /* nullable */,
                     aix_ingest_stats_t* stats);
int aix_count23_fixed_file(aix_index_t* h, const char* path, int format, int canon_mode, uint32_t* tf_out /* u32[n] */, aix_ingest_stats_t* stats);
/* K1 front end: replaces OptimizedKmerCounter's window loop (count_kmers.cpp:93-136,297-308) on a
 * PLAIN buffer in HBM: d_codes[p] = canonical 2-bit code of the k-window starting at byte p
 * (chars valid per count_kmers.cpp:71-88: ACGTU any case; canon_mode as above), or ~0 when the
 * window contains any other byte. d_codes holds len-k+1 entries. Distinct k-mers and their counts are
 * the sort + run-length of the valid entries (aindex_amd/counting.py). 1 <= k <= 32. */
int aix_window_codes_dev(const char* d_plain, uint64_t len, int k, int canon_mode, uint64_t* d_codes, void* stream);
/* A1 + A2: replaces `compute_aindex <reads> <pf> <prefix> <threads> 23 <tf> <kmers.bin> <kmers.txt>`
 * (src/compute_aindex.cpp:28-116): indices_out[n+1] = exclusive prefix sum of tf (AIndexCompressed ctor,
 * src/hash.hpp:365-399) = the .indices.bin image; positions_out[indices[n]] = the .index.bin image:
 * for every 23-byte window of the `.reads` buffer without '\n', '~', 'N', probing only the numerically smaller
 * strand, positions[indices[h] + slot] = offset + 1 for the first tf[h] occurrences in ascending offset order —
 * the result of the reference run with ONE thread (lu_compressed_worker, src/hash.cpp:960-1060; its multi-thread
 * slot order is schedule dependent). positions_out may be NULL to query *total_out = indices[n] first.
 * Any length: buffers of more than 2^30 windows are filled piece by piece, per-bucket fill counters carried over.
 *
 * 13-mer handles (N3: replaces `compute_aindex13 <reads> <pf> <tf> <prefix> <threads>`, src/compute_aindex13.cpp:36-71,
 * 109-226,298-321): n = 4^13 buckets in MPHF order, tf = the handle's u64[4^13] table (what count_kmers13 writes), a window
 * counts iff its 13 bytes are upper-case A/C/G/T, forward strand only, same slot rule and start adjustment. The reference tool
 * itself reads the u64 tf file as u32[4^13] (compute_aindex13.cpp:46-47) and therefore indexes a scrambled table; handing this
 * entry point that misread view (widened to u64) reproduces the reference's files bit for bit, which is how the path is pinned
 * (tests/golden/aindex13). All positions entry points below accept 13-mer handles in the same way; a 13-mer table with an entry
 * above 2^32 - 1 is AIX_ERR_UNSUPPORTED (the per-bucket fill counters are 32 bits wide; the reference's are u64 fetch_adds). */
int aix_positions_fill(aix_index_t* h, const char* reads, uint64_t len, uint64_t* indices_out, uint64_t* positions_out,
                       uint64_t positions_cap, uint64_t* total_out);
/* device-resident twin (reads, indices and positions in HBM; returns after the fill has completed on `stream`):
 * aix_positions_total gives indices[n] = the size of d_positions_out; `start` is aix_positions_start() of the buffer
 * (only the caller holds its head in host memory). */
int aix_positions_total(aix_index_t* h, uint64_t* total_out);
int aix_positions_fill_dev(aix_index_t* h, const char* d_reads, uint64_t len, uint64_t start, uint64_t* d_indices_out,
                           uint64_t* d_positions_out, uint64_t positions_cap, void* stream);
/* A2 over SHARDS of the reads file (multi-GPU). Shards are cut after '\n' (no window spans a cut).
 * aix_positions_bucket_counts: counts_out[h] (u64[n]) = windows of this shard that fall into bucket h under A2's rules;
 * aix_positions_fill_shard: the fill of this shard alone, with bucket h's slot numbering starting at filled_init[h]
 * (= occurrences in all EARLIER shards, clamped to 2^32-1; NULL = zeros) and offsets reported as base_offset + local
 * offset + 1. positions_out is the FULL-size array (indices[n] entries); entries this shard does not own are zero, so
 * the shards combine by addition. first_shard != 0 applies the reference's start adjustment (hash.cpp:973-986), which
 * only the beginning of the file sees. The union over shards equals aix_positions_fill of the whole file. */
int aix_positions_bucket_counts(aix_index_t* h, const char* reads, uint64_t len, int first_shard, uint64_t* counts_out);
/* the start adjustment itself (host only): first window offset the reference's single worker looks at. A shard whose
 * adjusted start is >= its window count has no clean window, and the adjustment carries on into the next shard. */
int aix_positions_start(const char* reads, uint64_t len, uint64_t* start_out);
int aix_positions_start_k(const char* reads, uint64_t len, int k /* 23 or 13 */, uint64_t* start_out);
int aix_positions_fill_shard(aix_index_t* h, const char* reads, uint64_t len, int first_shard, uint64_t base_offset,
                             const uint32_t* filled_init, uint64_t* positions_out, uint64_t positions_cap);
/* Device-resident twins of the SHARD entry points (one process per GPU; the partial results stay in HBM for the RCCL collectives
 * of aindex_amd/dist.py: scatter_sharded_t, positions_fill_sharded_t). d_* are device pointers; `start` is
 * aix_positions_start_k() of the shard's head (0 unless the shard is the first one that holds a clean window);
 * aix_positions_fill_shard_dev writes into a caller-zeroed FULL-size positions array (indices[n] entries). */
int aix_index_scatter_shard_codes_dev(const void* pf_bytes, uint64_t pf_len, const uint64_t* d_codes, const uint32_t* d_counts /* nullable */,
                                      uint64_t n_keys, uint64_t n_slots, int device, void* stream, uint64_t* d_checker_out, uint32_t* d_tf_out,
                                      uint32_t* d_occupied_out /* ceil(n_slots / 32) words */);
int aix_positions_indices_dev(aix_index_t* h, uint64_t* d_indices_out /* n + 1 */, void* stream);
int aix_positions_bucket_counts_dev(aix_index_t* h, const char* d_reads, uint64_t len, uint64_t start, uint64_t* d_counts_out /* n, zeroed here */,
                                    void* stream);
int aix_positions_fill_shard_dev(aix_index_t* h, const char* d_reads, uint64_t len, uint64_t start, uint64_t base_offset,
                                 const uint32_t* d_filled_init /* n, nullable */, const uint64_t* d_indices /* n + 1 */,
                                 uint64_t* d_positions /* indices[n], zeroed by the caller */, void* stream);
/* K1 complete: replaces `kmer_counter <in.fa> <k> <out> [-t N] [-m min]` (src/count_kmers.cpp:235-382): the set of
 * (canonical k-mer code, count) with count >= min_count, sorted by code ascending (the reference sorts by count with
 * unspecified tie order; parity is on the set). *keys_out / *counts_out are malloc'd (aix_free). format as for the
 * counters (FASTA records follow count_kmers.cpp:250-295). 1 <= k <= 31; any length (buffers of more
 * than 2^30 windows are counted piece by piece and the sorted distinct sets MERGED, never re-sorted — count_kmers.cpp:334-341 merges its
 * per-thread maps once; counts and sizes are 64-bit). */
int aix_count_distinct(const char* buf, uint64_t len, int format, int k, int canon_mode, uint64_t min_count, int device,
                       uint64_t** keys_out, uint64_t** counts_out, uint64_t* n_out);
/* the same from a file, streamed (see aix_count13_file): pieces of 2^30 windows are counted as the parts arrive, their sorted sets merged */
int aix_count_distinct_file(const char* path, int format, int k, int canon_mode, uint64_t min_count, int device, uint64_t** keys_out,
                            uint64_t** counts_out, uint64_t* n_out, aix_ingest_stats_t* stats);
/* device-resident twin: d_plain is a PLAIN buffer in HBM; the (key, count) arrays stay in HBM inside *out until the caller
 * has sized its own arrays (aix_distinct_size) and copied them (aix_distinct_copy_dev: u64 keys ascending, u64 counts). */
typedef struct aix_distinct aix_distinct_t;
int aix_count_distinct_dev(const char* d_plain, uint64_t len, int k, int canon_mode, uint64_t min_count, int device, void* stream,
                           aix_distinct_t** out);
/* K1 across GPUs: (key, count) pairs with repeated keys (what a rank holds after the all-to-all by key owner) -> the same kind
 * of result object: keys ascending, counts of equal keys summed, counts >= min_count. Pairs in any order (sort + reduce-by-key). */
int aix_merge_counts_dev(const uint64_t* d_keys, const uint64_t* d_counts, uint64_t n, uint64_t min_count, int device, void* stream,
                         aix_distinct_t** out);
/* the same when the caller can name its runs: run r = entries [run_offsets[r], run_offsets[r + 1]) (HOST array of nruns + 1 offsets), each run
 * sorted by key and free of repeats — after the exchange a rank holds one such run per peer, as the reference's merge holds one map per
 * thread (count_kmers.cpp:334-341). A tree of two-way merges with summation; nothing is sorted, any size. */
int aix_merge_runs_dev(const uint64_t* d_keys, const uint64_t* d_counts, const uint64_t* run_offsets, uint32_t nruns, uint64_t min_count, int device,
                       void* stream, aix_distinct_t** out);
int aix_distinct_size(const aix_distinct_t* r, uint64_t* n_out);
int aix_distinct_copy_dev(const aix_distinct_t* r, uint64_t* d_keys, uint64_t* d_counts, void* stream);
void aix_distinct_free(aix_distinct_t* r);
/* Host-side record normalisation to PLAIN form (readers of count_kmers13.cpp:211-272 /
 * count_kmers.cpp:250-295): out must hold len+1 bytes; *out_len receives the normalised length.
 * fasta_mode: 0 = count_kmers13 rules, 1 = kmer_counter rules ('>' anywhere starts a record). */
int aix_normalize_reads(const char* buf, uint64_t len, int format, int fasta_mode, char* out, uint64_t* out_len);
int aix_detect_format(const char* buf, uint64_t len);  /* count_kmers13.cpp:194-206 */
/* Replaces the binary `compute_reads <file1> <file2|-> <fastq|fasta|se|reads> <prefix>` (src/compute_reads.cpp:20-216), host side (text
 * reformatting bound by file I/O: no device work, runs without a GPU): writes <prefix>.reads (a pair as R1~revcomp(R2), revcomp =
 * get_revcomp of kmers.cpp:310-330: anything but ACGT becomes N), <prefix>.ridx ("rid\tstart\tend") and, for FASTA, <prefix>.header
 * ("name\tstart\tlength"); mode "reads" only indexes an existing reads file. file2 is read in mode "fastq" only.
 * AIX_ERR_ARG: unknown mode; AIX_ERR_IO: a file cannot be read / written. */
int aix_compute_reads(const char* file1, const char* file2 /* nullable */, const char* mode, const char* prefix);
/* The tools' text files, host side (native: the files hold 10^7..10^9 lines).
 * aix_dat_load: the .dat of compute_index ("kmer<ws>tf" per line; worker_for_fill_index, src/hash.cpp:681-702: a missing or unreadable
 *   count reads as 0, one beyond u32 as its maximum; mock != 0: k-mers only). Every k-mer must be 23 characters (AIX_ERR_FORMAT), empty
 *   lines are skipped. *keys_out = n * 23 bytes, *tf_out = n counts (not written when mock); malloc'd, release with aix_free.
 * aix_pf_build_file: compute_mphf_seq <keys.txt> (compute_mphf_generic.hpp:21-30): one key per line -> aix_pf_build_ragged.
 * aix_kmers_write_text: the list kmer_counter writes (count_kmers.cpp:362-382), "KMER\tcount\n" per entry in the order given. */
int aix_dat_load(const char* path, int mock, uint64_t* n_out, char** keys_out, uint32_t** tf_out /* nullable when mock */);
int aix_pf_build_file(const char* keys_path, void** pf_out, uint64_t* pf_len);
int aix_kmers_write_text(const char* path, const uint64_t* keys, const uint64_t* counts, uint64_t n, int k);
/* A binary image to a file (the .index.bin / .indices.bin / .kmers.bin / .tf.bin the tools write: compute_aindex.cpp hash.hpp:470-486,
 * compute_index.cpp:59-67), through a mapping filled by several host threads. AIX_ERR_IO when the file cannot be created or written. */
int aix_file_write(const char* path, const void* data, uint64_t bytes);
/* The .ridx file ("rid\tstart\tend" per read) that compute_reads writes and AindexWrapper::load_reads_index reads back
 * (python_wrapper.cpp:261-279: `fin >> rid >> start >> end` until it fails). *out = 3 * n values, malloc'd (aix_free). */
int aix_ridx_load(const char* path, uint64_t* n_out, uint64_t** out);
/* The same normalisation for a buffer already in HBM (byte-identical output; the readers are finite-state transducers,
 * resolved with a parallel scan of per-chunk transition functions). format must be PLAIN, FASTA or FASTQ; d_out holds
 * len+1 bytes; *out_len is a HOST pointer; the call synchronises the stream. */
int aix_normalize_reads_dev(const char* d_raw, uint64_t len, int format, int fasta_mode, char* d_out, uint64_t* out_len, void* stream);

/* ------------------------------------------------------------------------------------------
 * Synthetic inputs generated directly in HBM (mirrored by aindex_amd/synth.py).
 * ------------------------------------------------------------------------------------------ */
int aix_synth_genome_dev(uint64_t seed, uint64_t length, char* d_out, void* stream);
int aix_synth_kmers_dev(uint64_t seed, uint64_t first, uint64_t N, int k, char* d_out, void* stream);
int aix_synth_mix23_dev(uint64_t seed, const char* d_genome, uint64_t genome_len, uint64_t first, uint64_t N,
                        char* d_out /* N*23 */, void* stream);   /* Q_mix: 50 % genome windows on a random strand, 50 % random */
int aix_synth_reads_dev(uint64_t seed, const char* d_genome, uint64_t genome_len, uint64_t first_read,
                        uint64_t n_reads, uint32_t read_len, int rc_half, uint32_t n_rate_ppm,
                        char* d_out /* n_reads*(read_len+1) */, void* stream);

/* ------------------------------------------------------------------------------------------
 * MWHC builder (host code, like the reference's). replaces `compute_mphf_seq <keys.txt> <out.pf>`
 * (src/emphf/compute_mphf_generic.hpp:19-61, mphf.hpp:21-67, hypergraph_sorter_seq.hpp:29-102):
 * bit-identical .pf image for the same key list (same mt19937_64(37) seed stream, peeling order
 * and value assignment). *pf_out is malloc'd; release with aix_free. AIX_ERR_CONFLICT when the key
 * list is not peelable after 64 seeds (duplicate keys; the reference would loop forever).
 * ------------------------------------------------------------------------------------------ */
int aix_pf_build(const char* keys /* n*key_len bytes */, uint64_t n, uint32_t key_len, void** pf_out, uint64_t* pf_len);
int aix_pf_build_ragged(const char* bytes, const uint64_t* offsets /* n+1 */, uint64_t n, void** pf_out, uint64_t* pf_len);
int aix_pf_build_codes(const uint64_t* codes, uint64_t n, int k, void** pf_out, uint64_t* pf_len); /* keys = ASCII of 2-bit codes */
int aix_pf_build_all_13mers(void** pf_out, uint64_t* pf_len);   /* generate_all_13mers + build_13mer_hash */
/* The same construction on the GPU for keys (2-bit codes, ASCII-hashed) already in HBM: same seed stream, hash domain
 * and hypergraph, parallel peeling — a valid emphf .pf that the reference loads and evaluates, but NOT byte-identical
 * to compute_mphf_seq's (the bit-pair values depend on the peeling order). n and 3*hash_domain must be < 2^32. */
int aix_pf_build_codes_dev(const uint64_t* d_codes, uint64_t n, int k, int device, void* stream, void** pf_out, uint64_t* pf_len);
void aix_free(void* p);

/* Roofline probe: n_access uniform-random reads of elem_bytes (4, 8, 16; 32, 64, 128 with unroll 1: the whole element) over a table
 * of n_elems elements in HBM, `unroll` (1 or 4) independent reads in flight per lane. Measurement aid only. */
int aix_bench_gather_dev(const void* d_table, uint64_t n_elems, int elem_bytes, int unroll, uint64_t n_access,
                         uint64_t seed, uint64_t* d_sink, void* stream);

/* Diagnostics of the placement experiments (scripts/gpu_r3_relocate.py, gpu_r3_bloomlot.py): move the verification table of a
 * 23-mer handle into d_dst (nb * 128 bytes, caller-owned, kept alive by the caller) or, with NULL, into a block allocated now; move the
 * absence filter into a block allocated now (behind `pad_bytes` of padding; the old block and the padding are deliberately not freed, so
 * that successive moves land on different pages). Answers are unchanged. Not for production use. */
int aix_debug_relocate_table(aix_index_t* h, void* d_dst);
int aix_debug_relocate_bloom(aix_index_t* h, uint64_t pad_bytes);
int aix_debug_pointers(const aix_index_t* h, uint64_t out[5]);   /* device addresses: MPHF records, side index, unfiled keys, table, absence filter */
int aix_debug_rehome(aix_index_t* h, uint32_t mask);   /* 1 MPHF records, 2 side index, 4 unfiled keys, 8 table, 16 absence filter -> fresh blocks */

/* self-test hook of the GPU suite: lower bounds of keys[i] and keys[i] + 1 in a sorted u16 array of n entries, computed by the wave-wide search with
 * which the partition kernels find a partition's chunks; out[2 i], out[2 i + 1]. Device pointers. */
int aix_selftest_lower_bound_dev(const uint16_t* d_sorted, uint32_t n, const uint32_t* d_keys, uint32_t nkeys, uint32_t* d_out, void* stream);

/* self-test hook for the CPU test-suite: the exact-modulo used by the kernels, run on the host */
uint64_t aix_selftest_mod(uint64_t h, uint64_t d);
uint64_t aix_selftest_revcomp(uint64_t code, int k);

#ifdef __cplusplus
}
#endif
#endif /* AINDEX_HIP_H */
