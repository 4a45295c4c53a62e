// aix_handle.hpp — private to the library: the index handle resident in HBM, the error / device / scratch helpers shared by the
// translation units that implement the C ABI (aix_api.hip, aix_ingest.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

#include "../../include/aindex_hip.h"
#include "aix_internal.hpp"

using namespace aix;

void set_last_error(const std::string& s);      // thread-local text behind aix_strerror(AIX_ERR_HIP)

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e));                \
            return AIX_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)


struct DevGuard {   // switch to the handle's device for the duration of a call, then restore
    int prev = -1;
    bool ok = false;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct DevBuf {                     // per-call temporary from the scratch pool, used on ONE stream (default: the null stream)
    void* p = nullptr;
    hipStream_t st = nullptr;
    DevBuf() = default;
    explicit DevBuf(hipStream_t s) : st(s) {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    bool pooled = true;
    hipError_t alloc(uint64_t bytes) { return pool_alloc(&p, bytes ? bytes : 1); }
    // one-shot staging (index load): straight from the driver and straight back, never parked in the scratch cache
    hipError_t alloc_once(uint64_t bytes) { pooled = false; return hipMalloc(&p, bytes ? bytes : 1); }
    // a block goes back to the pool idle: wait for the stream it was used on (not for the whole device — other handles and
    // other host threads keep running; error paths leave through here too)
    ~DevBuf() { if (p) { (void)hipStreamSynchronize(st); if (pooled) pool_free(p); else (void)hipFree(p); } }
};

struct aix_index {
    int device = 0;
    uint32_t k = 0;
    uint64_t n = 0;
    // .pf header
    uint64_t mphf_n = 0, D = 0, seed = 0, B = 0, W = 0;
    // HBM
    BvRec* recs = nullptr;
    EeRec* ee = nullptr;                       // early-exit table (23-mer handles with keys)
    KeyRec* keys = nullptr;                    // checker[] / tf[] interleaved: while the handle is built, and afterwards only when no verification table was built
    uint32_t* side = nullptr;                  // with a table: slot -> table entry | AIX_SIDE_UNFILED + index into `unfiled` (4 B per key instead of 16)
    KeyRec* unfiled = nullptr;                 // the keys the table does not hold (~1 %)
    uint64_t n_unfiled = 0;
    BkEntry* bk = nullptr;                     // verification table: nb buckets of eight {code, tf, slot} entries (one 128-byte line each)
    uint32_t nb = 0;
    bool bk_borrowed = false;                  // the table lives in a caller's block (aix_debug_relocate_table)
    uint32_t bk_lpp = 8;                       // lanes that share one bucket read
    bool bk_lpp_set = false;                   // chosen by the caller (AIX_BUCKET_LANES / aix_index_set_bucket_table): then every consumer uses it
    uint64_t bk_unfiled = 0;                   // keys beyond the eighth of their bucket (answered through the MPHF)
    bool bk_enabled = true;
    BkEntry* mk = nullptr;                     // minimizer-keyed copy of the table for the streaming counter: entries grouped by minimizer bucket
    uint32_t* mk_off = nullptr;                // nbm + 1 offsets into mk
    uint32_t mk_cap = 16;                      // entries of a bucket a lane of the streaming counter reads
    uint32_t nbm = 0;
    uint64_t mk_unfiled = 0;
    bool mk_enabled = true;
    uint64_t* bloom = nullptr;                 // absence filter in front of the table (lookups / coverage)
    uint32_t nbloom = 0;
    bool bloom_enabled = true;
    uint64_t* tf13_mphf = nullptr;
    uint64_t* tf13_code = nullptr;
    uint32_t* perm13 = nullptr;
    unsigned long long* scratch13 = nullptr;   // code-ordered count table, lazily allocated
    void* work13 = nullptr;                    // partition workspace of the atomic-free counter (grow-only)
    uint64_t work13_bytes = 0;
    // tiny host batches (a Python loop over index[kmer]): pinned, device-mapped staging so that a call is one memcpy into
    // host memory, one launch and one synchronise — no hipMemcpy round trips
    void* pin_in = nullptr;
    void* pin_out[3] = {nullptr, nullptr, nullptr};
    std::mutex small_mutex;
    hipStream_t small_stream = nullptr;
    void* pin_cov = nullptr;                   // pinned, device-mapped staging of small coverage requests (kCovPin bytes)
    struct HostPipe* pipe = nullptr;           // pinned staging + streams of the large host-buffer batches (lazily built)
    std::mutex pipe_mutex;
    hipEvent_t work13_done = nullptr;          // recorded behind every counting call: the next one (any stream) waits for it before touching the workspace
    hipStream_t hist_stream = nullptr;         // count23, AIX_COUNT23_HIST_CUS only: the partition + histogram kernels on their own CUs (CU-masked stream)
    hipStream_t probe_stream = nullptr;        // count23: the slot probe of piece i + 1 runs here while piece i is partitioned and added on the caller's stream
    hipEvent_t probe_ev[2] = {nullptr, nullptr}, hist_ev[2] = {nullptr, nullptr}, start_ev = nullptr;
    uint64_t device_bytes = 0;
    bool perm13_bijective = false;             // 13-mer: code -> mphf slot is a bijection of [0, 4^13) (true for the all-13-mers .pf)
    bool canonical_only = false;
    bool canonical_fastpath = true;
    bool has_fp = false;
    bool fp_filter = true;
    bool early_exit = true;                    // (used when the early-exit table exists: built at open only without a verification table, else on request)
    std::mutex count_mutex;
    uint64_t pos_total = 0;                        // aix_positions_total, once computed (sum of tf[])
    bool pos_total_known = false;
    uint32_t a2_backend = 0;                       // the last positions fill: bit 0 = stable radix sort, bit 1 = MSD partition (per piece)
    uint32_t c23_backend = 0, c23_passes = 0;     // the last aix_count23_fixed*: 1 = memory-side atomics, 2 = slot stream + LDS histogram; passes over the slot stream
    bool c13_atomics = false, c13_added = false;   // state of a 13-mer count in progress (between count13_begin_locked and count13_end_locked)

    // the slot-stream consumers (count23's histogram path, the positions probe): two lanes per bucket line unless the caller chose a width
    IndexDev dev_slots() const {
        IndexDev d = dev();
        if (!bk_lpp_set) d.bk_lpp = 2;
        return d;
    }
    IndexDev dev() const {
        IndexDev d{};
        d.m.recs = recs;
        d.m.ee = ee;
        d.m.D = D;
        d.m.seed = seed;
        d.m.nrecs = (B + 15) / 16;
        d.m.fm = make_fastmod(D);
        d.keys = keys;
        d.side = side;
        d.bk_store = bk;
        d.unfiled = unfiled;
        d.n = n;
        d.tf13_code = tf13_code;
        d.tf13_mphf = tf13_mphf;
        d.perm13 = perm13;
        d.canonical_only = (canonical_only && canonical_fastpath) ? 1u : 0u;
        d.k = k;
        d.use_fp = (has_fp && fp_filter) ? 1u : 0u;
        d.early_exit = (has_fp && ee && early_exit) ? 1u : 0u;
        d.bk = (bk && bk_enabled) ? bk : nullptr;
        d.nb = nb;
        d.bk_lpp = bk_lpp;
        d.bloom = (d.bk && bloom && bloom_enabled) ? bloom : nullptr;
        d.nbloom = nbloom;
        d.mk = (d.bk && mk && mk_enabled) ? mk : nullptr;
        d.nbm = nbm;
        d.mk_off = mk_off;
        d.mk_cap = mk_cap;
        return d;
    }
};

// K13 in steps (aix_api.hip); the caller holds h->count_mutex from begin to end
int count13_begin_locked(aix_index* h, uint64_t* d_tf_out, hipStream_t s);
int count13_add_locked(aix_index* h, const char* d_plain, uint64_t len, uint64_t* d_tf_out, hipStream_t s);
int count13_end_locked(aix_index* h, uint64_t* d_tf_out, hipStream_t s);
