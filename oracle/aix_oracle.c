/* oracle/aix_oracle.c — TEST INFRASTRUCTURE ONLY. See aix_oracle.h for the contract.
 *
 * Plain-C restatement of the reference's CPU algorithms (ad3002/aindex v1.4.4). It follows the
 * reference's *structure* (per-k-mer ASCII hashing, `%` by the hash domain, broadword popcount,
 * forward-then-reverse-complement probing) so that it is also a faithful CPU baseline.
 * Parity pinned by tests/test_oracle_golden.py against fixtures made by the compiled reference.
 */
#define _GNU_SOURCE
#include "aix_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* =====================================================================================
 * H1  jenkins64_hasher::operator()(byte_range)      src/emphf/base_hash.hpp:38-91
 *     mix()                                          src/emphf/base_hash.hpp:127-145
 * ===================================================================================== */
static inline uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

#define JMIX(a, b, c)                      \
    do {                                   \
        a -= b; a -= c; a ^= (c >> 43);    \
        b -= c; b -= a; b ^= (a << 9);     \
        c -= a; c -= b; c ^= (b >> 8);     \
        a -= b; a -= c; a ^= (c >> 38);    \
        b -= c; b -= a; b ^= (a << 23);    \
        c -= a; c -= b; c ^= (b >> 5);     \
        a -= b; a -= c; a ^= (c >> 35);    \
        b -= c; b -= a; b ^= (a << 49);    \
        c -= a; c -= b; c ^= (b >> 11);    \
        a -= b; a -= c; a ^= (c >> 12);    \
        b -= c; b -= a; b ^= (a << 18);    \
        c -= a; c -= b; c ^= (b >> 22);    \
    } while (0)

void aixo_jenkins64(const uint8_t* s, uint64_t len, uint64_t seed, uint64_t out[3]) {
    uint64_t a = seed, b = seed, c = 0x9e3779b97f4a7c13ULL;
    const uint8_t* cur = s;
    uint64_t rem = len;
    while (rem >= 24) {               /* :46-55 */
        a += ld64(cur); b += ld64(cur + 8); c += ld64(cur + 16);
        cur += 24; rem -= 24;
        JMIX(a, b, c);
    }
    c += len;                         /* :57 */
    /* :59-88 tail: bytes 0-7 -> a, 8-15 -> b, 16-22 -> c shifted one byte up (low byte = len) */
    for (uint64_t i = 0; i < rem; ++i) {
        uint64_t v = (uint64_t)cur[i];
        if (i < 8)       a += v << (8 * i);
        else if (i < 16) b += v << (8 * (i - 8));
        else             c += v << (8 * (i - 16 + 1));
    }
    JMIX(a, b, c);                    /* :90 */
    out[0] = a; out[1] = b; out[2] = c;
}

/* =====================================================================================
 * H5  .pf container  mphf::load mphf.hpp:107-113 -> jenkins64_hasher::load base_hash.hpp:116-119
 *     -> ranked_bitpair_vector::load :78-84 -> bitpair_vector::load bitpair_vector.hpp:102-107
 *     layout (LE): u64 n; u64 D; u64 seed; u64 B; u64 words[ceil(B/32)]; u64 ranks[ceil(B/512)]
 * ===================================================================================== */
int aixo_mphf_from_bytes(const uint8_t* buf, uint64_t len, aixo_mphf* f) {
    memset(f, 0, sizeof(*f));
    if (len < 32) return -1;
    memcpy(&f->n, buf, 8); memcpy(&f->D, buf + 8, 8); memcpy(&f->seed, buf + 16, 8); memcpy(&f->B, buf + 24, 8);
    f->W = (f->B + 31) / 32;
    f->R = (f->B + 511) / 512;
    if (len < 32 + 8 * (f->W + f->R)) return -2;
    f->words = (uint64_t*)malloc(8 * (f->W ? f->W : 1));
    f->ranks = (uint64_t*)malloc(8 * (f->R ? f->R : 1));
    if (!f->words || !f->ranks) return -3;
    memcpy(f->words, buf + 32, 8 * f->W);
    memcpy(f->ranks, buf + 32 + 8 * f->W, 8 * f->R);
    return 0;
}

static int read_file(const char* path, uint8_t** out, uint64_t* len) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return -1;
    fseek(fp, 0, SEEK_END);
    long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    uint8_t* b = (uint8_t*)malloc(sz > 0 ? (size_t)sz : 1);
    if (!b) { fclose(fp); return -3; }
    if (sz > 0 && fread(b, 1, (size_t)sz, fp) != (size_t)sz) { fclose(fp); free(b); return -2; }
    fclose(fp);
    *out = b; *len = (uint64_t)sz;
    return 0;
}

int aixo_mphf_load(const char* path, aixo_mphf* f) {
    uint8_t* b; uint64_t len;
    int rc = read_file(path, &b, &len);
    if (rc) return rc;
    rc = aixo_mphf_from_bytes(b, len, f);
    free(b);
    return rc;
}

void aixo_mphf_free(aixo_mphf* f) { free(f->words); free(f->ranks); memset(f, 0, sizeof(*f)); }

/* nonzero_pairs, EMPHF_USE_POPCOUNT=0 branch: ranked_bitpair_vector.hpp:92-106 */
static inline uint64_t nonzero_pairs(uint64_t x) {
    const uint64_t ones4 = 0x1111111111111111ULL, ones8 = 0x0101010101010101ULL;
    x = (x | (x >> 1)) & (0x5 * ones4);
    x = (x & 3 * ones4) + ((x >> 2) & 3 * ones4);
    x = (x + (x >> 4)) & 0x0f * ones8;
    return (x * ones8) >> 56;
}

/* bitpair_vector::operator[]  bitpair_vector.hpp:46-49 */
static inline uint64_t bv_get(const aixo_mphf* f, uint64_t pos) {
    return (f->words[pos / 32] >> ((pos % 32) * 2)) & 3;
}

/* ranked_bitpair_vector::rank  ranked_bitpair_vector.hpp:47-62 */
static inline uint64_t bv_rank(const aixo_mphf* f, uint64_t pos) {
    uint64_t word_idx = pos / 32, word_offset = pos % 32, block = pos / 512;
    uint64_t r = f->ranks[block];
    for (uint64_t w = block * 512 / 32; w < word_idx; ++w) r += nonzero_pairs(f->words[w]);
    uint64_t mask = ((uint64_t)1 << (word_offset * 2)) - 1;
    r += nonzero_pairs(f->words[word_idx] & mask);
    return r;
}

/* mphf::lookup  mphf.hpp:79-89 */
uint64_t aixo_mphf_lookup(const aixo_mphf* f, const uint8_t* s, uint64_t len) {
    uint64_t h[3];
    aixo_jenkins64(s, len, f->seed, h);
    uint64_t nodes[3] = {h[0] % f->D, f->D + (h[1] % f->D), 2 * f->D + (h[2] % f->D)};
    uint64_t hidx = (bv_get(f, nodes[0]) + bv_get(f, nodes[1]) + bv_get(f, nodes[2])) % 3;
    return bv_rank(f, nodes[hidx]);
}

/* =====================================================================================
 * C1/C2 codec  src/kmers.cpp
 * ===================================================================================== */
uint64_t aixo_encode23(const char* s) {   /* kmers.cpp:12-25: anything but A/C/G/T adds 0 */
    uint64_t num = 0;
    for (int n = 0; n < 23; n++) {
        num <<= 2;
        if (s[n] == 'C') num += 1;
        if (s[n] == 'G') num += 2;
        if (s[n] == 'T') num += 3;
    }
    return num;
}
uint32_t aixo_encode13(const char* s) {   /* kmers.cpp:42-55 */
    uint32_t num = 0;
    for (int n = 0; n < 13; n++) {
        num <<= 2;
        if (s[n] == 'C') num += 1;
        if (s[n] == 'G') num += 2;
        if (s[n] == 'T') num += 3;
    }
    return num;
}
void aixo_decode23(uint64_t x, char* out) {  /* kmers.cpp:89-114 */
    static const char L[4] = {'A', 'C', 'G', 'T'};
    for (int i = 22; i >= 0; i--) { out[i] = L[x & 3]; x >>= 2; }
}
void aixo_decode13(uint32_t x, char* out) {
    static const char L[4] = {'A', 'C', 'G', 'T'};
    for (int i = 12; i >= 0; i--) { out[i] = L[x & 3]; x >>= 2; }
}
/* _reversePairs + reverseDNA  kmers.cpp:355-363,376-381: reverse the 32 bit-pairs of the
 * word, complement, drop the 18 low bits that came from the 9 unused top pairs. */
uint64_t aixo_revdna23(uint64_t num) {
    uint64_t count = 62, rev = num;
    for (num >>= 2; num; num >>= 2) { rev <<= 2; rev |= num & 3; count -= 2; }
    rev <<= count;
    return (~rev) >> 18;
}
uint32_t aixo_revdna13(uint32_t num) {     /* kmers.cpp:365-373,383-388 */
    uint32_t count = 30, rev = num;
    for (num >>= 2; num; num >>= 2) { rev <<= 2; rev |= num & 3; count -= 2; }
    rev <<= count;
    return (~rev) >> 6;
}

/* =====================================================================================
 * P1/P2  PHASH_MAP + load_hash   src/hash.hpp:82-121, src/hash.cpp:367-450
 * ===================================================================================== */
int aixo_index23_load(const char* pf, const char* tf_bin, const char* kmers_bin, aixo_index23* ix) {
    memset(ix, 0, sizeof(*ix));
    int rc = aixo_mphf_load(pf, &ix->f);
    if (rc) return rc;
    uint8_t* b; uint64_t len;
    rc = read_file(kmers_bin, &b, &len);
    if (rc) return rc - 10;
    ix->n = len / 8;                           /* hash.cpp:393-397 */
    ix->checker = (uint64_t*)b;
    uint8_t* t; uint64_t tlen;
    rc = read_file(tf_bin, &t, &tlen);
    if (rc) return rc - 20;
    ix->tf = (uint32_t*)calloc(ix->n ? ix->n : 1, 4);   /* hash.cpp:431-444 reads until EOF */
    memcpy(ix->tf, t, (tlen / 4 < ix->n ? tlen / 4 : ix->n) * 4);
    free(t);
    return 0;
}
void aixo_index23_free(aixo_index23* ix) {
    aixo_mphf_free(&ix->f); free(ix->checker); free(ix->tf); memset(ix, 0, sizeof(*ix));
}

/* =====================================================================================
 * Q1  get_tf_value_23mer  src/python_wrapper.cpp:610-627  (shared shape with :700-742)
 * returns: which = 0 not found, 1 forward, 2 reverse; *slot = hash slot
 * ===================================================================================== */
static inline int probe23(const aixo_index23* ix, const char* s, uint64_t len, uint64_t* slot) {
    if (len < 23) return 0;                                  /* reference: UB */
    uint64_t u = aixo_encode23(s);                           /* :611 */
    uint64_t h1 = aixo_mphf_lookup(&ix->f, (const uint8_t*)s, len);  /* :612 raw bytes, full length */
    if (h1 >= ix->n || ix->checker[h1] != u) {               /* :613 */
        char rev[23];
        uint64_t r = aixo_revdna23(u);                       /* :615 */
        aixo_decode23(r, rev);                               /* :616 */
        uint64_t h2 = aixo_mphf_lookup(&ix->f, (const uint8_t*)rev, 23);  /* :617 */
        if (h2 >= ix->n || ix->checker[h2] != r) return 0;   /* :618-619 */
        *slot = h2; return 2;
    }
    *slot = h1; return 1;
}
uint32_t aixo_tf23(const aixo_index23* ix, const char* s, uint64_t len) {
    uint64_t slot; return probe23(ix, s, len, &slot) ? ix->tf[slot] : 0;
}
uint64_t aixo_kid23(const aixo_index23* ix, const char* s, uint64_t len) {
    uint64_t slot; return probe23(ix, s, len, &slot) ? slot : 0;      /* :700-716: 0 for not found */
}
uint64_t aixo_strand23(const aixo_index23* ix, const char* s, uint64_t len) {
    uint64_t slot; return (uint64_t)probe23(ix, s, len, &slot);      /* :726-742 */
}
uint64_t aixo_hash23(const aixo_index23* ix, const char* s, uint64_t len) {
    return aixo_mphf_lookup(&ix->f, (const uint8_t*)s, len);         /* :638-642 */
}
/* Q4  get_total_tf_value_23mer :1230-1246 ; get_tf_both_directions_23mer :1259-1275 */
void aixo_both23(const aixo_index23* ix, const char* s, uint64_t len, uint32_t* fwd, uint32_t* rc) {
    if (len != 23) { *fwd = 0; *rc = 0; return; }
    *fwd = aixo_tf23(ix, s, 23);
    char rev[23];
    aixo_decode23(aixo_revdna23(aixo_encode23(s)), rev);
    *rc = aixo_tf23(ix, rev, 23);
}
uint64_t aixo_total23(const aixo_index23* ix, const char* s, uint64_t len) {
    uint32_t a, b; aixo_both23(ix, s, len, &a, &b); return (uint64_t)a + (uint64_t)b;
}
void aixo_tf23_batch(const aixo_index23* ix, const char* kmers, uint64_t N, uint32_t* out) {
    for (uint64_t i = 0; i < N; ++i) out[i] = aixo_tf23(ix, kmers + 23 * i, 23);   /* :653-664 loop */
}
void aixo_hash23_batch(const aixo_index23* ix, const char* kmers, uint64_t N, uint64_t* out) {
    for (uint64_t i = 0; i < N; ++i) out[i] = aixo_hash23(ix, kmers + 23 * i, 23); /* :629-636 */
}
/* the remaining single-k-mer queries over a batch (plain loops over the functions above; the full-size config-3 test asks 10^6 of each) */
void aixo_info23_batch(const aixo_index23* ix, const char* kmers, uint64_t N, uint64_t* kid, uint8_t* strand, uint64_t* total, uint32_t* fwd, uint32_t* rc) {
    for (uint64_t i = 0; i < N; ++i) {
        const char* s = kmers + 23 * i;
        kid[i] = aixo_kid23(ix, s, 23);
        strand[i] = (uint8_t)aixo_strand23(ix, s, 23);
        total[i] = aixo_total23(ix, s, 23);
        aixo_both23(ix, s, 23, fwd + i, rc + i);
    }
}

typedef struct { const void* ix; const char* kmers; uint64_t lo, hi; uint32_t* out; int k; } mt_job;
static void* tf23_worker(void* p) {
    mt_job* j = (mt_job*)p;
    for (uint64_t i = j->lo; i < j->hi; ++i) j->out[i] = aixo_tf23((const aixo_index23*)j->ix, j->kmers + 23 * i, 23);
    return NULL;
}
static void* tf13_worker(void* p) {
    mt_job* j = (mt_job*)p;
    for (uint64_t i = j->lo; i < j->hi; ++i) j->out[i] = aixo_tf13((const aixo_index13*)j->ix, j->kmers + 13 * i, 13);
    return NULL;
}
static void run_mt(void* (*fn)(void*), const void* ix, const char* kmers, uint64_t N, uint32_t* out, int nt) {
    if (nt < 1) nt = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nt);
    mt_job* jobs = (mt_job*)malloc(sizeof(mt_job) * nt);
    uint64_t per = (N + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        uint64_t lo = per * t, hi = lo + per;
        if (lo > N) lo = N;
        if (hi > N) hi = N;
        jobs[t] = (mt_job){ix, kmers, lo, hi, out, 0};
        pthread_create(&th[t], NULL, fn, &jobs[t]);
    }
    for (int t = 0; t < nt; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
}
void aixo_tf23_batch_mt(const aixo_index23* ix, const char* kmers, uint64_t N, uint32_t* out, int nt) {
    run_mt(tf23_worker, ix, kmers, N, out, nt);
}

/* =====================================================================================
 * Q3  get_tf_value_13mer :482-503 / get_tf_values_13mer :938-980  (strict, forward, u32 trunc)
 * Q4  get_total_tf_value_13mer :522-543, get_tf_both_directions_13mer :567-588
 *     (no character validation; the reference indexes tf[h] unguarded — h==n would be out of
 *      bounds; we return 0 for that slot)
 * ===================================================================================== */
uint32_t aixo_tf13(const aixo_index13* ix, const char* s, uint64_t len) {
    if (len != 13) return 0;
    for (int i = 0; i < 13; ++i)
        if (s[i] != 'A' && s[i] != 'T' && s[i] != 'G' && s[i] != 'C') return 0;
    uint64_t h = aixo_mphf_lookup(&ix->f, (const uint8_t*)s, 13);
    return h < AIXO_TOTAL_13MERS ? (uint32_t)ix->tf[h] : 0;
}
/* get_reverse_complement_13mer :505-517: reverse; A<->T, C<->G; any other byte unchanged */
static void rc13_ascii(const char* s, char* out) {
    for (int i = 0; i < 13; ++i) {
        char c = s[12 - i];
        switch (c) { case 'A': c = 'T'; break; case 'T': c = 'A'; break; case 'G': c = 'C'; break; case 'C': c = 'G'; break; default: break; }
        out[i] = c;
    }
}
void aixo_both13(const aixo_index13* ix, const char* s, uint64_t len, uint64_t* fwd, uint64_t* rc) {
    *fwd = 0; *rc = 0;
    if (len != 13) return;
    uint64_t h = aixo_mphf_lookup(&ix->f, (const uint8_t*)s, 13);
    if (h < AIXO_TOTAL_13MERS) *fwd = ix->tf[h];
    char r[13];
    rc13_ascii(s, r);
    uint64_t h2 = aixo_mphf_lookup(&ix->f, (const uint8_t*)r, 13);
    if (h2 < AIXO_TOTAL_13MERS) *rc = ix->tf[h2];
}
uint64_t aixo_total13(const aixo_index13* ix, const char* s, uint64_t len) {
    uint64_t a, b; aixo_both13(ix, s, len, &a, &b); return a + b;
}
void aixo_tf13_batch(const aixo_index13* ix, const char* kmers, uint64_t N, uint32_t* out) {
    for (uint64_t i = 0; i < N; ++i) out[i] = aixo_tf13(ix, kmers + 13 * i, 13);
}
void aixo_tf13_batch_mt(const aixo_index13* ix, const char* kmers, uint64_t N, uint32_t* out, int nt) {
    run_mt(tf13_worker, ix, kmers, N, out, nt);
}

/* =====================================================================================
 * Q5  AIndex.get_sequence_coverage  aindex/core/aindex.py:314-322
 *     cov[i] = tf if tf >= cutoff else 0 for i in range(len-k+1), tf = self[seq[i:i+k]]
 * ===================================================================================== */
void aixo_coverage23(const aixo_index23* ix, const char* seq, uint64_t len, uint32_t cutoff, uint32_t* out) {
    if (len < 23) return;
    for (uint64_t i = 0; i + 23 <= len; ++i) { uint32_t t = aixo_tf23(ix, seq + i, 23); out[i] = t >= cutoff ? t : 0; }
}
void aixo_coverage13(const aixo_index13* ix, const char* seq, uint64_t len, uint32_t cutoff, uint32_t* out) {
    if (len < 13) return;
    for (uint64_t i = 0; i + 13 <= len; ++i) { uint32_t t = aixo_tf13(ix, seq + i, 13); out[i] = t >= cutoff ? t : 0; }
}

/* =====================================================================================
 * K13  count_kmers13   src/count_kmers13.cpp
 *   detect_format :194-206; read_fasta_file :211-235; read_fastq_file :240-257;
 *   read_plain_file :262-272; normalize_sequence :113-126; process_sequence :131-161
 * ===================================================================================== */
int aixo_detect_format(const char* buf, uint64_t len) {
    if (len == 0) return 0;
    if (buf[0] == '\n') return 0;         /* empty first line -> PLAIN */
    if (buf[0] == '>') return 1;
    if (buf[0] == '@') return 2;
    return 0;
}

static void count13_sequence(const aixo_mphf* f, const char* seq, uint64_t n, uint64_t* counts, int atomic) {
    if (n < 13) return;                                   /* :132 */
    char win[13];
    /* normalize (:113-126): toupper; non-ACGT -> 'N'. validity = run of >= 13 ACGT */
    uint64_t run = 0;
    for (uint64_t i = 0; i < n; ++i) {
        char c = seq[i];
        if (c >= 'a' && c <= 'z') c = (char)(c - 32);
        int ok = (c == 'A' || c == 'C' || c == 'G' || c == 'T');
        run = ok ? run + 1 : 0;
        if (run >= 13) {
            for (int j = 0; j < 13; ++j) { char d = seq[i - 12 + j]; if (d >= 'a' && d <= 'z') d = (char)(d - 32); win[j] = d; }
            uint64_t h = aixo_mphf_lookup(f, (const uint8_t*)win, 13);   /* :146 */
            if (h < AIXO_TOTAL_13MERS) {                                  /* :149 */
                if (atomic) __atomic_fetch_add(&counts[h], 1, __ATOMIC_RELAXED);
                else counts[h] += 1;
            }
        }
    }
}

/* Walk the buffer as std::getline would; hand each logical sequence to cb. */
typedef void (*seq_cb)(void* ctx, const char* s, uint64_t n);
static void for_each_sequence(const char* buf, uint64_t len, int format, seq_cb cb, void* ctx) {
    uint64_t pos = 0, line_no = 0;
    char* acc = NULL; uint64_t acc_n = 0, acc_cap = 0;
    while (pos < len) {
        const char* nl = (const char*)memchr(buf + pos, '\n', len - pos);
        uint64_t end = nl ? (uint64_t)(nl - buf) : len;
        const char* line = buf + pos; uint64_t n = end - pos;
        if (format == 1) {                                  /* FASTA :211-235 */
            if (n != 0) {
                if (line[0] == '>') {
                    if (acc_n) { cb(ctx, acc, acc_n); acc_n = 0; }
                } else {
                    if (acc_n + n > acc_cap) { acc_cap = (acc_n + n) * 2 + 64; acc = (char*)realloc(acc, acc_cap); }
                    memcpy(acc + acc_n, line, n); acc_n += n;
                }
            }
        } else if (format == 2) {                           /* FASTQ :240-257 */
            if (line_no % 4 == 1 && n != 0) cb(ctx, line, n);
        } else {                                            /* PLAIN :262-272 */
            if (n != 0) cb(ctx, line, n);
        }
        line_no++;
        pos = end + 1;
    }
    if (format == 1 && acc_n) cb(ctx, acc, acc_n);
    free(acc);
}

typedef struct { const aixo_mphf* f; uint64_t* counts; } c13_ctx;
static void c13_cb(void* ctx, const char* s, uint64_t n) { c13_ctx* c = (c13_ctx*)ctx; count13_sequence(c->f, s, n, c->counts, 0); }

int aixo_count13(const aixo_mphf* f, const char* buf, uint64_t len, int format, uint64_t* counts) {
    if (format < 0) format = aixo_detect_format(buf, len);
    c13_ctx c = {f, counts};
    for_each_sequence(buf, len, format, c13_cb, &c);
    return 0;
}

/* threaded variant (CPU baseline on all cores): sequences collected, then range-split */
typedef struct { const char** s; uint64_t* n; uint64_t cnt, cap; char* owned; } seq_list;
static void list_cb(void* ctx, const char* s, uint64_t n) {
    seq_list* l = (seq_list*)ctx;
    if (l->cnt == l->cap) { l->cap = l->cap ? l->cap * 2 : 1024; l->s = (const char**)realloc(l->s, 8 * l->cap); l->n = (uint64_t*)realloc(l->n, 8 * l->cap); }
    l->s[l->cnt] = s; l->n[l->cnt] = n; l->cnt++;
}
typedef struct { const aixo_mphf* f; seq_list* l; uint64_t lo, hi; uint64_t* counts; } c13_job;
static void* c13_worker(void* p) {
    c13_job* j = (c13_job*)p;
    for (uint64_t i = j->lo; i < j->hi; ++i) count13_sequence(j->f, j->l->s[i], j->l->n[i], j->counts, 1);
    return NULL;
}
int aixo_count13_mt(const aixo_mphf* f, const char* buf, uint64_t len, int format, uint64_t* counts, int nt) {
    if (format < 0) format = aixo_detect_format(buf, len);
    if (format == 1) return aixo_count13(f, buf, len, format, counts);  /* FASTA accumulates into a temp */
    if (nt < 1) nt = 1;
    seq_list l = {0};
    for_each_sequence(buf, len, format, list_cb, &l);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nt);
    c13_job* jobs = (c13_job*)malloc(sizeof(c13_job) * nt);
    uint64_t per = (l.cnt + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        uint64_t lo = per * t, hi = lo + per;
        if (lo > l.cnt) lo = l.cnt;
        if (hi > l.cnt) hi = l.cnt;
        jobs[t] = (c13_job){f, &l, lo, hi, counts};
        pthread_create(&th[t], NULL, c13_worker, &jobs[t]);
    }
    for (int t = 0; t < nt; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs); free(l.s); free(l.n);
    return 0;
}

/* =====================================================================================
 * K1  kmer_counter   src/count_kmers.cpp
 *   char table :71-88 (A/a C/c G/g T/t U/u); string_to_kmer_fast :93-113;
 *   reverse_complement_fast :116-130 (DEFECTIVE: the first stage swaps the two bits *inside*
 *   every base, so the net map is reverse + A<->T only); get_canonical :132-136;
 *   record walk :250-295; window loop :297-308; filter/sort :362-382
 * ===================================================================================== */
static inline int k1_bits(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: return 4;
    }
}
uint64_t aixo_rc_refx86(uint64_t kmer, int k) {
    kmer = ((kmer & 0xAAAAAAAAAAAAAAAAULL) >> 1) | ((kmer & 0x5555555555555555ULL) << 1);
    kmer = ((kmer & 0xCCCCCCCCCCCCCCCCULL) >> 2) | ((kmer & 0x3333333333333333ULL) << 2);
    kmer = ((kmer & 0xF0F0F0F0F0F0F0F0ULL) >> 4) | ((kmer & 0x0F0F0F0F0F0F0F0FULL) << 4);
    kmer = ((kmer & 0xFF00FF00FF00FF00ULL) >> 8) | ((kmer & 0x00FF00FF00FF00FFULL) << 8);
    kmer = ((kmer & 0xFFFF0000FFFF0000ULL) >> 16) | ((kmer & 0x0000FFFF0000FFFFULL) << 16);
    kmer = (kmer >> 32) | (kmer << 32);
    kmer = ~kmer;
    return kmer >> (64 - 2 * k);
}
uint64_t aixo_rc_true(uint64_t code, int k) {
    uint64_t r = 0;
    for (int i = 0; i < k; ++i) { r = (r << 2) | (3 - (code & 3)); code >>= 2; }
    return r;
}
static inline uint64_t canon(uint64_t code, int k, int mode) {
    if (mode == 0) return code;
    uint64_t rc = mode == 1 ? aixo_rc_refx86(code, k) : aixo_rc_true(code, k);
    return code < rc ? code : rc;
}

typedef void (*code_cb)(void* ctx, uint64_t code);
/* record walk of count_kmers.cpp:250-308 over a FASTA buffer */
static void k1_walk_fasta(const char* buf, uint64_t len, int k, int mode, code_cb cb, void* ctx) {
    char* seq = (char*)malloc(len + 1);
    uint64_t i = 0;
    while (i < len && buf[i] != '>') i++;                  /* records start at '>' (:251-256) */
    while (i < len) {
        uint64_t start = i, end = i + 1;
        while (end < len && buf[end] != '>') end++;        /* next '>' anywhere */
        uint64_t j = start;
        while (j < end && buf[j] != '\n') j++;             /* skip header (:277) */
        j++;
        uint64_t n = 0;
        for (; j < end; ++j) { char c = buf[j]; if (c != '\n' && c != '\r') seq[n++] = c; }   /* :284-290 */
        if (n >= (uint64_t)k) {
            for (uint64_t pos = 0; pos + k <= n; ++pos) {   /* :301-307, full re-encode per window */
                uint64_t code = 0; int ok = 1;
                for (int t = 0; t < k; ++t) { int b = k1_bits((unsigned char)seq[pos + t]); if (b == 4) { ok = 0; break; } code = (code << 2) | (uint64_t)b; }
                if (ok) cb(ctx, canon(code, k, mode));
            }
        }
        i = end;
    }
    free(seq);
}
/* same window rule over newline-separated reads (one sequence per line; '~' etc. are invalid chars) */
static void k1_walk_lines(const char* buf, uint64_t len, int k, int mode, code_cb cb, void* ctx) {
    uint64_t pos = 0;
    while (pos < len) {
        const char* nl = (const char*)memchr(buf + pos, '\n', len - pos);
        uint64_t end = nl ? (uint64_t)(nl - buf) : len;
        uint64_t n = end - pos; const char* s = buf + pos;
        if (n >= (uint64_t)k) {
            for (uint64_t p = 0; p + k <= n; ++p) {
                uint64_t code = 0; int ok = 1;
                for (int t = 0; t < k; ++t) { int b = k1_bits((unsigned char)s[p + t]); if (b == 4) { ok = 0; break; } code = (code << 2) | (uint64_t)b; }
                if (ok) cb(ctx, canon(code, k, mode));
            }
        }
        pos = end + 1;
    }
}

typedef struct { uint64_t* v; uint64_t n, cap; } u64vec;
static void push_cb(void* ctx, uint64_t code) {
    u64vec* v = (u64vec*)ctx;
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : (1u << 16); v->v = (uint64_t*)realloc(v->v, 8 * v->cap); }
    v->v[v->n++] = code;
}
static int cmp_u64(const void* a, const void* b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }

int64_t aixo_count_distinct(const char* fasta, uint64_t len, int k, int canon_mode, uint64_t min_count,
                            uint64_t** keys, uint64_t** counts) {
    u64vec v = {0};
    k1_walk_fasta(fasta, len, k, canon_mode, push_cb, &v);
    qsort(v.v, v.n, 8, cmp_u64);
    uint64_t* K = (uint64_t*)malloc(8 * (v.n ? v.n : 1));
    uint64_t* C = (uint64_t*)malloc(8 * (v.n ? v.n : 1));
    uint64_t m = 0;
    for (uint64_t i = 0; i < v.n;) {
        uint64_t j = i;
        while (j < v.n && v.v[j] == v.v[i]) j++;
        if (j - i >= min_count) { K[m] = v.v[i]; C[m] = j - i; m++; }   /* :365-369 */
        i = j;
    }
    free(v.v);
    *keys = K; *counts = C;
    return (int64_t)m;
}

typedef struct { const aixo_index23* ix; uint32_t* tf; } c23_ctx;
static void c23_cb(void* ctx, uint64_t code) {
    c23_ctx* c = (c23_ctx*)ctx;
    char s[23];
    aixo_decode23(code, s);
    uint64_t h = aixo_mphf_lookup(&c->ix->f, (const uint8_t*)s, 23);
    if (h < c->ix->n && c->ix->checker[h] == code) c->tf[h] += 1;
}
int aixo_count23_fixed(const aixo_index23* ix, const char* buf, uint64_t len, int is_fasta, int canon_mode, uint32_t* tf_out) {
    c23_ctx c = {ix, tf_out};
    if (is_fasta) k1_walk_fasta(buf, len, 23, canon_mode, c23_cb, &c);
    else k1_walk_lines(buf, len, 23, canon_mode, c23_cb, &c);
    return 0;
}

/* =====================================================================================
 * I1  worker_for_fill_index  src/hash.cpp:671-723 (single range); arrays zeroed :836-844
 * ===================================================================================== */
int aixo_index_scatter(const aixo_mphf* f, const char* keys, const uint32_t* tfs, uint64_t n,
                       uint64_t* checker_out, uint32_t* tf_out) {
    memset(checker_out, 0, 8 * n);
    memset(tf_out, 0, 4 * n);
    for (uint64_t i = 0; i < n; ++i) {
        const char* s = keys + 23 * i;
        uint64_t h = aixo_mphf_lookup(f, (const uint8_t*)s, 23);       /* :706 */
        if (h >= n) return -12;                                         /* reference: OOB write */
        if (tf_out[h] != 0) return -12;                                 /* :708-713 exit(12) */
        checker_out[h] = aixo_encode23(s);                              /* :715 */
        tf_out[h] = tfs ? tfs[i] : 0;                                   /* :716 (mock -> 0) */
    }
    return 0;
}

/* =====================================================================================
 * A1  AIndexCompressed ctor  src/hash.hpp:365-399 (exclusive prefix sum over tf)
 * A2  lu_compressed_worker   src/hash.cpp:960-1060, one worker over [0, len)
 * ===================================================================================== */
void aixo_indices_prefix(const uint32_t* tf, uint64_t n, uint64_t* indices) {
    indices[0] = 0;
    for (uint64_t i = 1; i < n + 1; ++i) indices[i] = indices[i - 1] + tf[i - 1];
}
void aixo_positions_fill(const aixo_index23* ix, const char* c, uint64_t len, const uint64_t* indices, uint64_t* positions) {
    const uint64_t k = 23;
    if (len < k) return;
    uint64_t start = 0, end = len;
    uint64_t* pp = (uint64_t*)calloc(ix->n ? ix->n : 1, 8);           /* ppositions */
    while (start < end - k + 1) {                                     /* :973-986 */
        int found = 0;
        for (uint64_t i = start; i < start + k; ++i)
            if (c[i] == '\n' || c[i] == '~' || c[i] == '?') { start = i + 1; found = 1; break; }
        if (!found) break;
    }
    for (uint64_t i = start; i < end - k + 1; ++i) {                   /* :992 */
        int skip = 0;
        for (uint64_t j = 0; j < k; ++j)
            if (c[i + j] == '\n' || c[i + j] == '~' || c[i + j] == 'N') { skip = 1; break; }   /* :1006-1011 */
        if (skip) continue;
        uint64_t u = aixo_encode23(c + i), r = aixo_revdna23(u);
        uint64_t h, key;
        if (u <= r) { h = aixo_mphf_lookup(&ix->f, (const uint8_t*)(c + i), 23); key = u; }      /* :1032-1033 */
        else { char rev[23]; aixo_decode23(r, rev); h = aixo_mphf_lookup(&ix->f, (const uint8_t*)rev, 23); key = r; }  /* :1043 */
        if (h >= ix->n || ix->checker[h] != key) continue;
        uint64_t slot = pp[h]++;
        if (slot >= ix->tf[h]) continue;
        positions[indices[h] + slot] = i + 1;
    }
    free(pp);
}

/* =====================================================================================
 * N3: 13-mer positions index — src/compute_aindex13.cpp:36-71 (indices), :109-226 (worker, one thread).
 * tf is the u64[4^13] table count_kmers13 writes. The reference reads that file as u32[4^13]
 * (compute_aindex13.cpp:46-47); its behaviour is reproduced by handing THIS function the misread
 * view widened to u64 (tests/golden/make_golden.py does so), which is how the restatement is pinned.
 * ===================================================================================== */
void aixo_indices_prefix64(const uint64_t* tf, uint64_t n, uint64_t* indices) {           /* :57-63 */
    indices[0] = 0;
    for (uint64_t i = 1; i < n + 1; ++i) indices[i] = indices[i - 1] + tf[i - 1];
}
void aixo_positions_fill13(const aixo_mphf* f, const char* c, uint64_t len, const uint64_t* indices, uint64_t* positions) {
    const uint64_t k = 13, N13 = 67108864ull;
    if (len < k) return;
    uint64_t start = 0, end = len;
    const uint64_t total = indices[N13];
    uint64_t* pp = (uint64_t*)calloc(N13, 8);                          /* ppositions, :66-71 */
    while (start < end - k + 1) {                                     /* :135-148 */
        int found = 0;
        for (uint64_t i = start; i < start + k; ++i)
            if (c[i] == '\n' || c[i] == '~' || c[i] == '?') { start = i + 1; found = 1; break; }
        if (!found) break;
    }
    for (uint64_t i = start; i < end - k + 1; ++i) {                   /* :163 */
        int skip = 0;
        for (uint64_t j = 0; j < k; ++j) {                            /* :183-190: anything but upper-case ACGT skips the window */
            const char ch = c[i + j];
            if (ch != 'A' && ch != 'T' && ch != 'G' && ch != 'C') { skip = 1; break; }
        }
        if (skip) continue;
        const uint64_t h = aixo_mphf_lookup(f, (const uint8_t*)(c + i), 13);   /* :203, forward strand only (lookup never throws) */
        if (h >= N13) continue;
        const uint64_t slot = pp[h]++;                                /* :205 */
        const uint64_t idx = indices[h] + slot;
        if (idx < total && slot < indices[h + 1] - indices[h]) positions[idx] = i + 1;   /* :208-211, 1-based */
    }
    free(pp);
}

void aixo_free(void* p) { free(p); }
