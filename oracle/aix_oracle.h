/* oracle/aix_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C11) of the reference algorithms on the hot path named in
 * BASELINE.json:north_star. Every function cites the reference file:line it follows
 * (paths relative to /root/reference). Parity is PINNED: tests/test_oracle_golden.py checks
 * this library against tests/golden/ fixtures that were produced by the compiled reference
 * itself (oracle/_ref, see tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (aindex_amd/, libaindex_hip.so) never links, imports or calls it.
 */
#ifndef AIX_ORACLE_H
#define AIX_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- H1: Jenkins lookup8 triple hash, src/emphf/base_hash.hpp:38-91, mix :127-145 ---- */
void aixo_jenkins64(const uint8_t* s, uint64_t len, uint64_t seed, uint64_t out[3]);

/* ---- H5: .pf container, src/emphf/mphf.hpp:107-113 etc. ---- */
typedef struct {
    uint64_t n;      /* number of keys                    mphf.hpp m_n            */
    uint64_t D;      /* m_hash_domain                                              */
    uint64_t seed;   /* jenkins64_hasher::m_seed                                   */
    uint64_t B;      /* bit-pair count = 3*D              bitpair_vector m_size    */
    uint64_t W;      /* ceil(B/32) words                                           */
    uint64_t R;      /* ceil(B/512) block ranks                                    */
    uint64_t* words; /* owned */
    uint64_t* ranks; /* owned */
} aixo_mphf;

int  aixo_mphf_load(const char* path, aixo_mphf* f);                 /* 0 ok, <0 error */
int  aixo_mphf_from_bytes(const uint8_t* buf, uint64_t len, aixo_mphf* f);
void aixo_mphf_free(aixo_mphf* f);
/* H2-H4: mphf::lookup, src/emphf/mphf.hpp:79-89; rank ranked_bitpair_vector.hpp:47-62 */
uint64_t aixo_mphf_lookup(const aixo_mphf* f, const uint8_t* s, uint64_t len);

/* ---- C1/C2: codec, src/kmers.cpp ---- */
uint64_t aixo_encode23(const char* s);            /* kmers.cpp:12-40  */
uint32_t aixo_encode13(const char* s);            /* kmers.cpp:42-55  */
void     aixo_decode23(uint64_t x, char* out23);  /* kmers.cpp:89-114 (k=23) */
void     aixo_decode13(uint32_t x, char* out13);
uint64_t aixo_revdna23(uint64_t x);               /* kmers.cpp:355-363,376-381 */
uint32_t aixo_revdna13(uint32_t x);               /* kmers.cpp:365-373,383-388 */

/* ---- P1/P2: 23-mer index (PHASH_MAP), src/hash.hpp:82-121, src/hash.cpp:367-450 ---- */
typedef struct {
    aixo_mphf f;
    uint64_t  n;        /* = size(.kmers.bin)/8 */
    uint64_t* checker;  /* owned */
    uint32_t* tf;       /* owned */
} aixo_index23;

int  aixo_index23_load(const char* pf, const char* tf_bin, const char* kmers_bin, aixo_index23* ix);
void aixo_index23_free(aixo_index23* ix);

/* ---- Q1/Q2/Q4 23-mer queries, src/python_wrapper.cpp:610-627,700-742,1219-1286 ----
 * `s` points at `len` raw query bytes. The reference hashes the whole string but encodes the
 * first 23 chars; len < 23 is UB there and returns 0 here. */
uint32_t aixo_tf23(const aixo_index23* ix, const char* s, uint64_t len);
uint64_t aixo_kid23(const aixo_index23* ix, const char* s, uint64_t len);     /* :700-716 */
uint64_t aixo_strand23(const aixo_index23* ix, const char* s, uint64_t len);  /* :726-742 */
uint64_t aixo_total23(const aixo_index23* ix, const char* s, uint64_t len);   /* :1230-1246 */
void     aixo_both23(const aixo_index23* ix, const char* s, uint64_t len, uint32_t* fwd, uint32_t* rc); /* :1259-1275 */
uint64_t aixo_hash23(const aixo_index23* ix, const char* s, uint64_t len);    /* :638-642 */
/* fixed-stride batches (N x 23 contiguous bytes) */
void aixo_tf23_batch(const aixo_index23* ix, const char* kmers, uint64_t N, uint32_t* out);
void aixo_tf23_batch_mt(const aixo_index23* ix, const char* kmers, uint64_t N, uint32_t* out, int nthreads);
void aixo_hash23_batch(const aixo_index23* ix, const char* kmers, uint64_t N, uint64_t* out);
void aixo_info23_batch(const aixo_index23* ix, const char* kmers, uint64_t N, uint64_t* kid, uint8_t* strand, uint64_t* total, uint32_t* fwd, uint32_t* rc);

/* ---- Q3/Q4 13-mer queries, src/python_wrapper.cpp:482-503,522-608,938-980 ---- */
typedef struct {
    aixo_mphf f;
    const uint64_t* tf;   /* 4^13 u64, NOT owned (caller's buffer / mmap) */
} aixo_index13;
#define AIXO_TOTAL_13MERS 67108864ull
uint32_t aixo_tf13(const aixo_index13* ix, const char* s, uint64_t len);            /* strict, u32 trunc */
uint64_t aixo_total13(const aixo_index13* ix, const char* s, uint64_t len);         /* no validation; OOB -> 0 */
void     aixo_both13(const aixo_index13* ix, const char* s, uint64_t len, uint64_t* fwd, uint64_t* rc);
void     aixo_tf13_batch(const aixo_index13* ix, const char* kmers, uint64_t N, uint32_t* out);
void     aixo_tf13_batch_mt(const aixo_index13* ix, const char* kmers, uint64_t N, uint32_t* out, int nthreads);

/* ---- Q5 coverage, aindex/core/aindex.py:314-322 over Q1 (k=23) or Q3 (k=13) ---- */
void aixo_coverage23(const aixo_index23* ix, const char* seq, uint64_t len, uint32_t cutoff, uint32_t* out /* len-22 */);
void aixo_coverage13(const aixo_index13* ix, const char* seq, uint64_t len, uint32_t cutoff, uint32_t* out /* len-12 */);

/* ---- K13: count_kmers13, src/count_kmers13.cpp:113-161,194-272 ----
 * format: 0 plain, 1 fasta, 2 fastq, -1 auto-detect (:194-206). counts: 4^13 u64, zeroed by caller. */
int aixo_detect_format(const char* buf, uint64_t len);
int aixo_count13(const aixo_mphf* f, const char* buf, uint64_t len, int format, uint64_t* counts);
int aixo_count13_mt(const aixo_mphf* f, const char* buf, uint64_t len, int format, uint64_t* counts, int nthreads);

/* ---- K1: kmer_counter, src/count_kmers.cpp:93-136,250-308,362-382 ----
 * canon_mode: 0 none, 1 REF_X86 (defective rc, :116-130), 2 TRUE_RC. Returns number of distinct
 * keys with count >= min_count; `keys`/`counts` are malloc'd, sorted by key ascending (the reference's
 * tie order is unspecified -> compare as sets). */
int64_t aixo_count_distinct(const char* fasta, uint64_t len, int k, int canon_mode, uint64_t min_count,
                            uint64_t** keys, uint64_t** counts);
uint64_t aixo_rc_refx86(uint64_t code, int k);   /* :116-130 */
uint64_t aixo_rc_true(uint64_t code, int k);
/* Same counting semantics but histogrammed against a fixed 23-mer index (config 4):
 * tf_out[h] += 1 for every valid window whose canonical code is stored at slot h. */
int aixo_count23_fixed(const aixo_index23* ix, const char* fasta_or_lines, uint64_t len, int is_fasta,
                       int canon_mode, uint32_t* tf_out);

/* ---- I1: compute_index scatter, src/hash.cpp:671-723 ----
 * keys: n x 23 bytes. returns 0, or -12 on conflict (reference exit(12)). */
int aixo_index_scatter(const aixo_mphf* f, const char* keys, const uint32_t* tfs, uint64_t n,
                       uint64_t* checker_out, uint32_t* tf_out);

/* ---- A1/A2: positions index, src/hash.hpp:365-399, src/hash.cpp:960-1060 (1 thread) ---- */
void aixo_indices_prefix(const uint32_t* tf, uint64_t n, uint64_t* indices /* n+1 */);
void aixo_positions_fill(const aixo_index23* ix, const char* reads, uint64_t len,
                         const uint64_t* indices, uint64_t* positions /* zeroed, indices[n] */);
/* ---- N3: 13-mer positions index, src/compute_aindex13.cpp:36-71,109-226 (1 thread), tf as u64[4^13] ---- */
void aixo_indices_prefix64(const uint64_t* tf, uint64_t n, uint64_t* indices /* n + 1 */);
void aixo_positions_fill13(const aixo_mphf* f, const char* reads, uint64_t len, const uint64_t* indices /* 4^13 + 1 */,
                           uint64_t* positions /* zeroed, indices[4^13] */);

void aixo_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
