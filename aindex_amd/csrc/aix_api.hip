// aix_api.hip — the C ABI of libaindex_hip.so (include/aindex_hip.h): index lifecycle in HBM,
// host<->device staging, launch glue. No CPU compute path exists behind these entry points.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/aindex_hip.h"
#include "aix_internal.hpp"

#include "aix_handle.hpp"
#include "aix_ingest.hpp"

static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }


// ---------------------------------------------------------------------------------------------
extern "C" const char* aix_version(void) { return "aindex_hip 0.1.0 (gfx950)"; }

extern "C" const char* aix_strerror(int st) {
    switch (st) {
        case AIX_OK: return "ok";
        case AIX_ERR_ARG: return "invalid argument";
        case AIX_ERR_IO: return "file missing, unreadable or short";
        case AIX_ERR_FORMAT: return "malformed index file";
        case AIX_ERR_NOMEM: return "out of memory";
        case AIX_ERR_HIP: return g_last_error.empty() ? "HIP runtime error / no device" : g_last_error.c_str();
        case AIX_ERR_UNSUPPORTED: return "unsupported size or configuration";
        case AIX_ERR_MODE: return "call does not match the handle's k-mer mode";
        case AIX_ERR_CONFLICT: return "hash conflict while scattering (key not in the MPHF set)";
        default: return "unknown status";
    }
}

extern "C" int aix_device_count(int* count) {
    if (!count) return AIX_ERR_ARG;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0) {
        *count = 0;
        set_last_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
        return AIX_ERR_HIP;
    }
    *count = c;
    return AIX_OK;
}

extern "C" void aix_scratch_trim(void) { pool_trim(); pinned_trim(); }

extern "C" int aix_host_alloc(uint64_t bytes, void** out) {
    if (!out) return AIX_ERR_ARG;
    *out = nullptr;
    if (bytes == 0) return AIX_OK;
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); set_last_error("aix_host_alloc: out of pinned memory"); return AIX_ERR_NOMEM; }
    HIPCHK(e);
    *out = p;
    return AIX_OK;
}
extern "C" int aix_host_free(void* p) {
    if (!p) return AIX_OK;
    HIPCHK(hipHostFree(p));
    return AIX_OK;
}

extern "C" uint64_t aix_selftest_mod(uint64_t h, uint64_t d) { return fastmod(h, make_fastmod(d)); }
extern "C" uint64_t aix_selftest_revcomp(uint64_t code, int k) { return revcomp(code, k); }

// ---------------------------------------------------------------------------------------------
// files
// ---------------------------------------------------------------------------------------------
struct MappedFile {
    const uint8_t* p = nullptr;
    uint64_t len = 0;
    int open(const char* path) {
        int fd = ::open(path, O_RDONLY);
        if (fd < 0) return AIX_ERR_IO;
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); return AIX_ERR_IO; }
        len = (uint64_t)st.st_size;
        if (len) {
            void* m = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); return AIX_ERR_IO; }
            p = (const uint8_t*)m;
        }
        ::close(fd);
        return AIX_OK;
    }
    ~MappedFile() { if (p) munmap((void*)p, len); }
};

// .pf header checks shared by every entry point that takes a .pf image (host only)
extern "C" int aix_pf_check(const void* pf_bytes, uint64_t pf_len, uint64_t hdr_out[4]) {
    if (!pf_bytes) return AIX_ERR_ARG;
    if (pf_len < 32) return AIX_ERR_FORMAT;
    uint64_t hdr[4];
    memcpy(hdr, pf_bytes, 32);
    const uint64_t n = hdr[0], D = hdr[1], B = hdr[3];
    // mphf.hpp:26,99-113: B = 3 * hash_domain bit-pairs. A header whose product wraps (D = 0x5555555555555556, B = 2) or whose
    // domain needs more than 32-bit node ids (the builder's own limit) would index the record table far out of bounds on the
    // device, so it is refused here, before anything is uploaded.
    // (D = 0 is the MPHF of an empty key set: mphf.hpp:26 gives (ceil(0 * 1.23) + 2) / 3 = 0; nothing is ever evaluated on it.)
    if (D > 0xFFFFFFFFull / 3) return AIX_ERR_FORMAT;
    if (B != 3 * D) return AIX_ERR_FORMAT;
    if (n > B) return AIX_ERR_FORMAT;
    const uint64_t W = (B + 31) / 32, R = (B + 511) / 512;
    if (pf_len < 32 + 8 * (W + R)) return AIX_ERR_FORMAT;
    if (hdr_out) memcpy(hdr_out, hdr, 32);
    return AIX_OK;
}

// parse a .pf image (mphf.hpp:99-113) and lay it out as BvRec records in HBM
static int upload_mphf(aix_index* h, const uint8_t* pf, uint64_t len) {
    uint64_t hdr[4];
    const int chk = aix_pf_check(pf, len, hdr);
    if (chk) return chk;
    h->mphf_n = hdr[0]; h->D = hdr[1]; h->seed = hdr[2]; h->B = hdr[3];
    h->W = (h->B + 31) / 32;
    if (h->mphf_n >> 32) return AIX_ERR_UNSUPPORTED;          // 32-bit rank prefixes
    const uint64_t* words = (const uint64_t*)(pf + 32);
    const uint64_t nrec = (h->B + 15) / 16;                      // two records per 64-bit word
    std::vector<BvRec> recs;
    try { recs.resize(nrec ? nrec : 1); } catch (const std::bad_alloc&) { return AIX_ERR_NOMEM; }
    uint64_t run = 0;
    for (uint64_t i = 0; i < nrec; ++i) {
        uint64_t w;
        memcpy(&w, words + (i >> 1), 8);
        const uint32_t half = (uint32_t)(w >> (32 * (i & 1)));
        recs[i].pairs = half;
        recs[i].prefix = (uint32_t)run;
        recs[i].fp = 0;
        run += (uint32_t)__builtin_popcount((half | (half >> 1)) & 0x55555555u);
    }
    if (run >> 32) return AIX_ERR_UNSUPPORTED;
    const uint64_t bytes = sizeof(BvRec) * recs.size();
    HIPCHK(hipMalloc((void**)&h->recs, bytes));
    h->device_bytes += bytes;
    HIPCHK(hipMemcpy(h->recs, recs.data(), bytes, hipMemcpyHostToDevice));
    return AIX_OK;
}

static int check_device(int device) {
    int c = 0;
    int st = aix_device_count(&c);
    if (st) return st;
    if (device < 0 || device >= c) return AIX_ERR_ARG;
    return AIX_OK;
}

static void free_host_pipe(struct HostPipe* p);

static void destroy(aix_index* h) {
    if (!h) return;
    DevGuard g(h->device);
    if (h->recs) (void)hipFree(h->recs);
    if (h->ee) (void)hipFree(h->ee);
    if (h->keys) (void)hipFree(h->keys);
    if (h->side) (void)hipFree(h->side);
    if (h->unfiled) (void)hipFree(h->unfiled);
    if (h->bk && !h->bk_borrowed) (void)hipFree(h->bk);
    if (h->bloom) (void)hipFree(h->bloom);
    if (h->mk) (void)hipFree(h->mk);
    if (h->mk_off) (void)hipFree(h->mk_off);
    if (h->tf13_mphf) (void)hipFree(h->tf13_mphf);
    if (h->tf13_code) (void)hipFree(h->tf13_code);
    if (h->perm13) (void)hipFree(h->perm13);
    if (h->scratch13) (void)hipFree(h->scratch13);
    if (h->work13) (void)hipFree(h->work13);
    if (h->pipe) { free_host_pipe(h->pipe); h->pipe = nullptr; }
    if (h->work13_done) (void)hipEventDestroy(h->work13_done);
    for (int i = 0; i < 2; ++i) { if (h->probe_ev[i]) (void)hipEventDestroy(h->probe_ev[i]); if (h->hist_ev[i]) (void)hipEventDestroy(h->hist_ev[i]); }
    if (h->start_ev) (void)hipEventDestroy(h->start_ev);
    if (h->probe_stream) (void)hipStreamDestroy(h->probe_stream);
    if (h->hist_stream) (void)hipStreamDestroy(h->hist_stream);
    if (h->small_stream) (void)hipStreamDestroy(h->small_stream);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->pin_cov) (void)hipHostFree(h->pin_cov);
    for (void* p : h->pin_out) if (p) (void)hipHostFree(p);
    delete h;
}

// Verification table: n / load buckets of one 128-byte line. AIX_BUCKET_LOAD = mean keys per 8-entry bucket
// (default 4: 32 B of HBM per key, 2 % of the buckets overflow and 0.4 % of the keys stay with the MPHF path);
// AIX_BUCKET_TABLE=0 skips it (every probe through the MPHF records + key records, as in round 1).
static int build_bucket_table(aix_index* h, hipStream_t s) {
    if (h->n == 0) return AIX_OK;
    if (const char* e = getenv("AIX_BUCKET_TABLE")) { if (atoi(e) == 0) return AIX_OK; }
    double load = 4.0;
    if (const char* e = getenv("AIX_BUCKET_LOAD")) { const double v = atof(e); if (v >= 0.25 && v <= 8.0) load = v; }
    if (const char* e = getenv("AIX_BUCKET_LANES")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) { h->bk_lpp = (uint32_t)v; h->bk_lpp_set = true; } }
    uint64_t nb = (uint64_t)((double)h->n / load) + 1;
    if (nb > 0x0FFFFFF0ull) nb = 0x0FFFFFF0ull;                                // entry indices (8 per bucket) share a word with the "unfiled" flag of the side index
    const uint64_t bytes = nb * 8 * sizeof(BkEntry);
    DevBuf fill(s);
    HIPCHK(fill.alloc_once(4 * nb));
    HIPCHK(hipMalloc((void**)&h->bk, bytes));
    h->nb = (uint32_t)nb;
    h->device_bytes += bytes;
    HIPCHK(hipMemsetAsync(fill.p, 0, 4 * nb, s));
    // absence filter: AIX_BLOOM_BITS bits per key (default 16: 2 B of Infinity-Cache-resident filter per key, < 1 % of the absent
    // keys pass; 0 = no filter)
    double bloom_bits = 16.0;
    if (const char* e = getenv("AIX_BLOOM_BITS")) { const double v = atof(e); if (v == 0.0 || (v >= 4.0 && v <= 64.0)) bloom_bits = v; }
    if (bloom_bits > 0) {
        uint64_t nw = (uint64_t)((double)h->n * bloom_bits / 64.0) + 1;
        if (nw > 0xFFFFFFF0ull) nw = 0xFFFFFFF0ull;
        HIPCHK(hipMalloc((void**)&h->bloom, 8 * nw));
        h->nbloom = (uint32_t)nw;
        h->device_bytes += 8 * nw;
        HIPCHK(hipMemsetAsync(h->bloom, 0, 8 * nw, s));
    }
    // minimizer-keyed copy for the streaming counter (aix_stream23.hip): built only on request (AIX_MINIMIZER_TABLE=1). Every filed key
    // once more, grouped by the bucket of its minimizer (offsets + entries: a bucket is as long as its content, 16 B per key + 4 B per
    // bucket); AIX_MINIMIZER_LOAD = mean keys per bucket (default 2: the offsets of 5e7 keys are 100 MB, Infinity-Cache sized).
    bool want_mk = false;
    if (const char* e = getenv("AIX_MINIMIZER_TABLE")) want_mk = atoi(e) != 0;
    uint64_t nbm = 0;
    if (want_mk) {
        double mload = 2.0;
        if (const char* e = getenv("AIX_MINIMIZER_LOAD")) { const double v = atof(e); if (v >= 0.25 && v <= 16.0) mload = v; }
        nbm = (uint64_t)((double)h->n / mload) + 1;
        if (nbm > 0xFFFFFFF0ull) nbm = 0xFFFFFFF0ull;
    }
    DevBuf mfill(s);
    HIPCHK(mfill.alloc_once(4 * (nbm + 1)));
    HIPCHK(hipMemsetAsync(mfill.p, 0, 4 * (nbm + 1), s));
    HIPCHK(hipMalloc((void**)&h->side, 4 * h->n));
    h->device_bytes += 4 * h->n;
    HIPCHK(launch_build_buckets(h->dev().m, h->keys, h->n, h->bk, h->nb, (uint32_t*)fill.p, h->bloom, h->nbloom, (uint32_t)nbm, (uint32_t*)mfill.p, h->side, s));
    h->mk_cap = AIX_MK_ENTRIES;
    if (const char* e = getenv("AIX_MINIMIZER_CAP")) { const int v = atoi(e); if (v >= 1 && v <= AIX_MK_ENTRIES) h->mk_cap = (uint32_t)v; }   // test hook: short buckets -> many undecided windows
    if (want_mk) {
        // offsets = exclusive scan of the bucket sizes (a bucket holds < 2^32 keys in total: n < 2^32), then the entries
        HIPCHK(hipMalloc((void**)&h->mk_off, 4 * (nbm + 1)));
        h->device_bytes += 4 * (nbm + 1);
        HIPCHK(exclusive_scan_u32((const uint32_t*)mfill.p, h->mk_off, nbm + 1, s));
        uint32_t filed = 0;
        HIPCHK(hipMemcpyAsync(&filed, h->mk_off + nbm, 4, hipMemcpyDeviceToHost, s));
        // keys the streaming counter cannot answer from their bucket (longer than the cap): host side, once per open
        std::vector<uint32_t> mf;
        try { mf.resize(nbm); } catch (const std::bad_alloc&) { return AIX_ERR_NOMEM; }
        HIPCHK(hipMemcpyAsync(mf.data(), mfill.p, 4 * nbm, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        uint64_t left = 0;
        for (uint64_t i = 0; i < nbm; ++i) if (mf[i] > h->mk_cap) left += mf[i];
        h->mk_unfiled = left;
        HIPCHK(hipMalloc((void**)&h->mk, (uint64_t)(filed ? filed : 1) * sizeof(BkEntry)));
        h->device_bytes += (uint64_t)filed * sizeof(BkEntry);
        h->nbm = (uint32_t)nbm;
        HIPCHK(hipMemsetAsync(mfill.p, 0, 4 * (nbm + 1), s));
        HIPCHK(launch_fill_minimizer_table(h->dev().m, h->keys, h->n, h->mk, h->mk_off, h->nbm, (uint32_t*)mfill.p, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    // keys left to the MPHF path: sum over buckets of max(fill - 8, 0) (host side: once per open, nb words)
    std::vector<uint32_t> f;
    try { f.resize(nb); } catch (const std::bad_alloc&) { return AIX_ERR_NOMEM; }
    HIPCHK(hipMemcpyAsync(f.data(), fill.p, 4 * nb, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    uint64_t unfiled = 0;
    for (uint64_t i = 0; i < nb; ++i) if (f[i] > 8) unfiled += f[i] - 8;
    h->bk_unfiled = unfiled;
    // The keys the table does not hold (beyond the eighth of their bucket, or not in their own MPHF slot) are closed up into `unfiled`; with
    // the side index every slot's {code, tf} is then reachable without the 16 B-per-key record array, which goes back to the driver.
    {
        DevBuf cnt(s);
        HIPCHK(cnt.alloc_once(8));
        HIPCHK(hipMemsetAsync(cnt.p, 0, 8, s));
        HIPCHK(launch_count_unfiled(h->side, h->n, (uint32_t*)cnt.p, s));
        uint32_t nu = 0;
        HIPCHK(hipMemcpyAsync(&nu, cnt.p, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipMalloc((void**)&h->unfiled, sizeof(KeyRec) * (uint64_t)(nu ? nu : 1)));
        h->n_unfiled = nu;
        h->device_bytes += sizeof(KeyRec) * (uint64_t)nu;
        HIPCHK(launch_side_unfiled(h->keys, h->n, h->side, h->unfiled, (uint32_t*)cnt.p + 1, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    return AIX_OK;
}

// presence masks of the early-exit MPHF walk (aix_device.hpp: EeRec), from the handle's keys
static int build_early_exit_table(aix_index* h, hipStream_t s) {
    if (h->ee || h->n == 0) return AIX_OK;
    const uint64_t nrec = (h->B + 15) / 16;
    HIPCHK(hipMalloc((void**)&h->ee, sizeof(EeRec) * (nrec ? nrec : 1)));
    h->device_bytes += sizeof(EeRec) * nrec;
    EeRec* ee = h->ee;
    h->ee = nullptr;                                                            // not visible to dev() until it is complete
    const hipError_t e = launch_set_fingerprints(h->dev(), h->recs, ee, false, s);
    const hipError_t e2 = hipStreamSynchronize(s);
    h->ee = ee;
    HIPCHK(e);
    HIPCHK(e2);
    return AIX_OK;
}

// interleave device-resident checker[]/tf[] into KeyRec records and detect an all-canonical key set
static int adopt_device_arrays(aix_index* h, const uint64_t* d_checker, const uint32_t* d_tf, uint64_t n, hipStream_t s) {
    if (n == 0) return AIX_OK;
    uint32_t* d_flag = nullptr;
    HIPCHK(hipMalloc((void**)&h->keys, sizeof(KeyRec) * n));
    h->device_bytes += sizeof(KeyRec) * n;
    HIPCHK(hipMalloc((void**)&d_flag, 4));
    hipError_t e = hipMemsetAsync(d_flag, 0, 4, s);
    if (e == hipSuccess) e = launch_build_keyrecs(d_checker, d_tf, n, h->keys, d_flag, s);
    uint32_t flag = 1;
    if (e == hipSuccess) e = hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_flag);
    HIPCHK(e);
    h->canonical_only = (flag == 0);
    // The early-exit table (32 B per 16 bit-pairs: 2.5 B per key) serves the MPHF walk; with a verification table in front that walk only runs
    // behind overflowed buckets, so the table is then built on request (aix_index_set_early_exit) instead of at every open.
    bool want_table = true;
    if (const char* e = getenv("AIX_BUCKET_TABLE")) want_table = atoi(e) != 0;
    if (!want_table) { const int st = build_early_exit_table(h, s); if (st) return st; }
    HIPCHK(launch_set_fingerprints(h->dev(), h->recs, nullptr, true, s));
    HIPCHK(hipStreamSynchronize(s));
    h->has_fp = true;
    const int st = build_bucket_table(h, s);
    if (st) return st;
    if (h->bk && h->side) {                                                     // every key is reachable through the table / the unfiled list: drop the duplicate
        (void)hipFree(h->keys);
        h->keys = nullptr;
        h->device_bytes -= sizeof(KeyRec) * n;
    }
    return AIX_OK;
}

extern "C" int aix_index_create_23(const void* pf_bytes, uint64_t pf_len, const uint64_t* checker, const uint32_t* tf, uint64_t n, int device,
                                   aix_index_t** out) {
    if (!pf_bytes || !out || (n && (!checker || !tf))) return AIX_ERR_ARG;
    *out = nullptr;
    int st = check_device(device);
    if (st) return st;
    if (n >> 32) return AIX_ERR_UNSUPPORTED;
    aix_index* h = new (std::nothrow) aix_index();
    if (!h) return AIX_ERR_NOMEM;
    h->device = device; h->k = 23; h->n = n;
    DevGuard g(device);
    st = upload_mphf(h, (const uint8_t*)pf_bytes, pf_len);
    if (!st && n) {
        DevBuf dc, dt;
        hipError_t e = dc.alloc_once(8 * n);
        if (e == hipSuccess) e = dt.alloc_once(4 * n);
        if (e == hipSuccess) e = hipMemcpy(dc.p, checker, 8 * n, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dt.p, tf, 4 * n, hipMemcpyHostToDevice);
        if (e != hipSuccess) { set_last_error(std::string("index upload: ") + hipGetErrorString(e)); st = AIX_ERR_HIP; }
        else st = adopt_device_arrays(h, (const uint64_t*)dc.p, (const uint32_t*)dt.p, n, 0);
    }
    if (st) { destroy(h); return st; }
    *out = h;
    return AIX_OK;
}

// I1 on the device: scatter (key, count) pairs through the MPHF. Exactly one of d_keys / d_codes is set.
// n keys into nslots slots (nslots == n for a whole key set; a shard of the keys scatters into the full-size arrays).
// occ_out (optional, device, ceil(nslots/32) words): bit h set <=> slot h was written by this call.
static int scatter_device(aix_index* h, uint64_t n, uint64_t nslots, const uint8_t* d_keys, const uint64_t* d_codes, const uint32_t* d_counts,
                          uint64_t* d_checker, uint32_t* d_tf, uint32_t* occ_out, hipStream_t s) {
    DevBuf occ(s), flag(s);
    const uint64_t occ_bytes = 4 * ((nslots + 31) / 32);
    HIPCHK(occ.alloc(occ_bytes));
    HIPCHK(flag.alloc(4));
    HIPCHK(hipMemsetAsync(occ.p, 0, occ_bytes, s));
    HIPCHK(hipMemsetAsync(flag.p, 0, 4, s));
    HIPCHK(hipMemsetAsync(d_checker, 0, 8 * nslots, s));       // hash.cpp:836-844: arrays start zeroed
    HIPCHK(hipMemsetAsync(d_tf, 0, 4 * nslots, s));
    const IndexDev d = h->dev();
    HIPCHK(launch_scatter23(d.m, n, nslots, d_keys, d_codes, d_counts, d_checker, d_tf, (uint32_t*)occ.p, (uint32_t*)flag.p, s));
    uint32_t conflicts = 0;
    HIPCHK(hipMemcpyAsync(&conflicts, flag.p, 4, hipMemcpyDeviceToHost, s));
    if (occ_out) HIPCHK(hipMemcpyAsync(occ_out, occ.p, occ_bytes, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return conflicts ? AIX_ERR_CONFLICT : AIX_OK;
}

static int scatter_host(const void* pf_bytes, uint64_t pf_len, const char* keys, const uint32_t* counts, uint64_t n, uint64_t nslots, int device,
                        uint64_t* checker_out, uint32_t* tf_out, uint32_t* occupied_out) {
    int st = check_device(device);
    if (st) return st;
    aix_index tmp;
    tmp.device = device; tmp.k = 23; tmp.n = nslots;
    DevGuard g(device);
    st = upload_mphf(&tmp, (const uint8_t*)pf_bytes, pf_len);
    if (!st) {
        DevBuf dk, dcnt, dc, dt, docc;
        const uint64_t occ_bytes = 4 * ((nslots + 31) / 32);
        hipError_t e = dk.alloc(23 * n + 8);
        if (e == hipSuccess) e = dc.alloc(8 * nslots);
        if (e == hipSuccess) e = dt.alloc(4 * nslots);
        if (e == hipSuccess && occupied_out) e = docc.alloc(occ_bytes);
        if (e == hipSuccess && counts && n) e = dcnt.alloc(4 * n);
        if (e == hipSuccess && n) e = hipMemcpy(dk.p, keys, 23 * n, hipMemcpyHostToDevice);
        if (e == hipSuccess && counts && n) e = hipMemcpy(dcnt.p, counts, 4 * n, hipMemcpyHostToDevice);
        if (e != hipSuccess) { set_last_error(std::string("scatter staging: ") + hipGetErrorString(e)); st = AIX_ERR_HIP; }
        if (!st) st = scatter_device(&tmp, n, nslots, (const uint8_t*)dk.p, nullptr, (counts && n) ? (const uint32_t*)dcnt.p : nullptr, (uint64_t*)dc.p,
                                     (uint32_t*)dt.p, occupied_out ? (uint32_t*)docc.p : nullptr, 0);
        if (!st || st == AIX_ERR_CONFLICT) {                       // a shard reports its conflict AND hands back what it wrote
            const int keep = st;
            e = hipMemcpy(checker_out, dc.p, 8 * nslots, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(tf_out, dt.p, 4 * nslots, hipMemcpyDeviceToHost);
            if (e == hipSuccess && occupied_out) e = hipMemcpy(occupied_out, docc.p, occ_bytes, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { set_last_error(std::string("scatter readback: ") + hipGetErrorString(e)); st = AIX_ERR_HIP; }
            else st = keep;
        }
    }
    if (tmp.recs) (void)hipFree(tmp.recs);
    tmp.recs = nullptr;
    return st;
}

extern "C" int aix_index_scatter(const void* pf_bytes, uint64_t pf_len, const char* keys, const uint32_t* counts, uint64_t n, int device,
                                 uint64_t* checker_out, uint32_t* tf_out) {
    if (!pf_bytes || !keys || !checker_out || !tf_out || n == 0) return AIX_ERR_ARG;
    return scatter_host(pf_bytes, pf_len, keys, counts, n, n, device, checker_out, tf_out, nullptr);
}

extern "C" int aix_index_scatter_shard(const void* pf_bytes, uint64_t pf_len, const char* keys, const uint32_t* counts, uint64_t n_keys, uint64_t n_slots,
                                       int device, uint64_t* checker_out, uint32_t* tf_out, uint32_t* occupied_out) {
    if (!pf_bytes || (n_keys && !keys) || !checker_out || !tf_out || !occupied_out || n_slots == 0 || n_keys > n_slots) return AIX_ERR_ARG;
    if (n_slots >> 32) return AIX_ERR_UNSUPPORTED;
    return scatter_host(pf_bytes, pf_len, keys, counts, n_keys, n_slots, device, checker_out, tf_out, occupied_out);
}

extern "C" int aix_index_build_23_codes_dev(const void* pf_bytes, uint64_t pf_len, const uint64_t* d_codes, const uint32_t* d_counts, uint64_t n,
                                            int device, void* stream, aix_index_t** out) {
    if (!pf_bytes || !d_codes || !out || n == 0) return AIX_ERR_ARG;
    *out = nullptr;
    int st = check_device(device);
    if (st) return st;
    if (n >> 32) return AIX_ERR_UNSUPPORTED;
    aix_index* h = new (std::nothrow) aix_index();
    if (!h) return AIX_ERR_NOMEM;
    h->device = device; h->k = 23; h->n = n;
    DevGuard g(device);
    st = upload_mphf(h, (const uint8_t*)pf_bytes, pf_len);
    if (!st) {
        DevBuf dc((hipStream_t)stream), dt((hipStream_t)stream);
        hipError_t e = dc.alloc(8 * n);
        if (e == hipSuccess) e = dt.alloc(4 * n);
        if (e != hipSuccess) { set_last_error(std::string("index build: ") + hipGetErrorString(e)); st = AIX_ERR_HIP; }
        if (!st) st = scatter_device(h, n, n, nullptr, d_codes, d_counts, (uint64_t*)dc.p, (uint32_t*)dt.p, nullptr, (hipStream_t)stream);
        if (!st) st = adopt_device_arrays(h, (const uint64_t*)dc.p, (const uint32_t*)dt.p, n, (hipStream_t)stream);
    }
    if (st) { destroy(h); return st; }
    *out = h;
    return AIX_OK;
}

extern "C" int aix_index_open_23(const char* pf, const char* tf_bin, const char* kmers_bin, int device, aix_index_t** out) {
    if (!pf || !tf_bin || !kmers_bin || !out) return AIX_ERR_ARG;
    MappedFile fpf, ftf, fk;
    if (fpf.open(pf) || ftf.open(tf_bin) || fk.open(kmers_bin)) return AIX_ERR_IO;
    const uint64_t n = fk.len / 8;                              // hash.cpp:393-397: n = size(.kmers.bin)/8
    std::vector<uint32_t> tfpad;
    const uint32_t* tfp = (const uint32_t*)ftf.p;
    if (ftf.len / 4 < n) {                                      // hash.cpp:431-444 reads until EOF; rest stays 0
        tfpad.assign(n, 0);
        memcpy(tfpad.data(), ftf.p, (ftf.len / 4) * 4);
        tfp = tfpad.data();
    }
    return aix_index_create_23(fpf.p, fpf.len, (const uint64_t*)fk.p, tfp, n, device, out);
}

static int build_13_tables(aix_index* h, const uint64_t* tf_host) {
    const uint64_t N13 = AIX_TOTAL_13MERS;
    HIPCHK(hipMalloc((void**)&h->tf13_mphf, 8 * N13));
    HIPCHK(hipMalloc((void**)&h->tf13_code, 8 * N13));
    HIPCHK(hipMalloc((void**)&h->perm13, 4 * N13));
    h->device_bytes += 20 * N13;
    if (tf_host) HIPCHK(hipMemcpy(h->tf13_mphf, tf_host, 8 * N13, hipMemcpyHostToDevice));
    else HIPCHK(hipMemset(h->tf13_mphf, 0, 8 * N13));
    const IndexDev d = h->dev();
    HIPCHK(launch_perm13(d.m, h->perm13, 0));
    HIPCHK(launch_tf13_to_code_order(h->perm13, h->tf13_mphf, h->tf13_code, 0));
    // The streaming counter writes each bin's total to out[perm[code]] with a plain store: right only when code -> slot is a
    // bijection, which holds for the all-13-mers .pf and not for a foreign one. Checked once here; a handle that fails the check
    // counts through the atomics path, which adds (the reference's fetch_add at mphf(window), count_kmers13.cpp:147-152).
    {
        DevBuf bits, bad;
        HIPCHK(bits.alloc_once(AIX_TOTAL_13MERS / 8));
        HIPCHK(bad.alloc_once(4));
        HIPCHK(hipMemsetAsync(bits.p, 0, AIX_TOTAL_13MERS / 8, 0));
        HIPCHK(hipMemsetAsync(bad.p, 0, 4, 0));
        HIPCHK(launch_perm13_check(h->perm13, (uint32_t*)bits.p, (uint32_t*)bad.p, 0));
        uint32_t nbad = 1;
        HIPCHK(hipMemcpy(&nbad, bad.p, 4, hipMemcpyDeviceToHost));
        h->perm13_bijective = (nbad == 0);
    }
    HIPCHK(hipStreamSynchronize(0));
    return AIX_OK;
}

extern "C" int aix_index_create_13(const void* pf_bytes, uint64_t pf_len, const uint64_t* tf, int device, aix_index_t** out) {
    if (!pf_bytes || !out) return AIX_ERR_ARG;
    *out = nullptr;
    int st = check_device(device);
    if (st) return st;
    aix_index* h = new (std::nothrow) aix_index();
    if (!h) return AIX_ERR_NOMEM;
    h->device = device; h->k = 13; h->n = AIX_TOTAL_13MERS;
    DevGuard g(device);
    st = upload_mphf(h, (const uint8_t*)pf_bytes, pf_len);
    if (!st) st = build_13_tables(h, tf);
    if (st) { destroy(h); return st; }
    *out = h;
    return AIX_OK;
}

extern "C" int aix_index_open_13(const char* pf, const char* tf_bin, int device, aix_index_t** out) {
    if (!pf || !out) return AIX_ERR_ARG;
    MappedFile fpf, ftf;
    if (fpf.open(pf)) return AIX_ERR_IO;
    const uint64_t* tf = nullptr;
    if (tf_bin) {
        if (ftf.open(tf_bin)) return AIX_ERR_IO;
        if (ftf.len < 8 * AIX_TOTAL_13MERS) return AIX_ERR_FORMAT;   // reference mmaps 4^13*8 bytes (:425)
        tf = (const uint64_t*)ftf.p;
    }
    return aix_index_create_13(fpf.p, fpf.len, tf, device, out);
}

extern "C" int aix_index_close(aix_index_t* h) {
    if (!h) return AIX_ERR_ARG;
    destroy(h);
    pool_trim();                      // scratch blocks cached for this handle's calls go back to the driver with it
    return AIX_OK;
}

extern "C" int aix_index_info(const aix_index_t* h, aix_info_t* info) {
    if (!h || !info) return AIX_ERR_ARG;
    memset(info, 0, sizeof(*info));
    info->k = h->k; info->device = (uint32_t)h->device; info->n = h->n; info->mphf_n = h->mphf_n;
    info->hash_domain = h->D; info->seed = h->seed; info->bitpairs = h->B; info->device_bytes = h->device_bytes;
    info->canonical_only = h->canonical_only ? 1 : 0;
    info->bucket_table = (h->bk && h->bk_enabled) ? 1 : 0;
    info->bucket_lanes = h->bk_lpp;
    info->buckets = h->bk ? h->nb : 0;
    info->bucket_unfiled_keys = h->bk_unfiled;
    info->absence_filter_words = (h->bk && h->bk_enabled && h->bloom && h->bloom_enabled) ? h->nbloom : 0;
    info->minimizer_lines = (h->bk && h->bk_enabled && h->mk && h->mk_enabled) ? h->nbm : 0;
    info->minimizer_unfiled_keys = h->mk_unfiled;
    info->count23_backend = h->c23_backend;
    info->count23_passes = h->c23_passes;
    info->positions_backend = h->a2_backend;
    return AIX_OK;
}

extern "C" int aix_index_set_canonical_fastpath(aix_index_t* h, int enabled) {
    if (!h) return AIX_ERR_ARG;
    h->canonical_fastpath = enabled != 0;
    return AIX_OK;
}

extern "C" int aix_index_set_fingerprint_filter(aix_index_t* h, int enabled) {
    if (!h) return AIX_ERR_ARG;
    h->fp_filter = enabled != 0;
    return AIX_OK;
}

extern "C" int aix_index_set_early_exit(aix_index_t* h, int enabled) {
    if (!h) return AIX_ERR_ARG;
    h->early_exit = enabled != 0;
    if (enabled && h->k == 23 && h->has_fp && !h->ee) {                         // first request on a handle that was opened with a verification table
        DevGuard g(h->device);
        const int st = build_early_exit_table(h, 0);
        if (st) return st;
    }
    return AIX_OK;
}

extern "C" int aix_index_set_minimizer_table(aix_index_t* h, int enabled) {
    if (!h) return AIX_ERR_ARG;
    h->mk_enabled = enabled != 0;
    return AIX_OK;
}

extern "C" int aix_index_set_absence_filter(aix_index_t* h, int enabled) {
    if (!h) return AIX_ERR_ARG;
    h->bloom_enabled = enabled != 0;
    return AIX_OK;
}

extern "C" int aix_index_set_bucket_table(aix_index_t* h, int enabled, int lanes) {
    if (!h) return AIX_ERR_ARG;
    if (lanes != 0 && lanes != 1 && lanes != 2 && lanes != 4 && lanes != 8) return AIX_ERR_ARG;
    h->bk_enabled = enabled != 0;
    if (lanes) { h->bk_lpp = (uint32_t)lanes; h->bk_lpp_set = true; }
    return AIX_OK;
}

// Move the verification table of a 23-mer handle into another block of HBM: d_dst (nb * 128 bytes, caller-owned and kept alive by the caller
// for the life of the handle) or, with d_dst == NULL, a block allocated now. Placement experiments only (scripts/gpu_r3_relocate.py).
extern "C" int aix_debug_relocate_table(aix_index_t* h, void* d_dst) {
    if (!h || h->k != 23 || !h->bk) return AIX_ERR_ARG;
    DevGuard g(h->device);
    const uint64_t bytes = (uint64_t)h->nb * 8 * sizeof(BkEntry);
    BkEntry* nb_ = (BkEntry*)d_dst;
    if (!nb_) HIPCHK(hipMalloc((void**)&nb_, bytes));
    HIPCHK(hipMemcpy(nb_, h->bk, bytes, hipMemcpyDeviceToDevice));
    HIPCHK(hipDeviceSynchronize());
    if (!h->bk_borrowed) (void)hipFree(h->bk);
    h->bk = nb_;
    h->bk_borrowed = d_dst != nullptr;
    return AIX_OK;
}

// the same for the absence filter; the old block is NOT freed (the next candidate must come from other physical pages): experiments only
extern "C" int aix_debug_relocate_bloom(aix_index_t* h, uint64_t pad_bytes) {
    if (!h || h->k != 23 || !h->bloom) return AIX_ERR_ARG;
    DevGuard g(h->device);
    void* pad = nullptr;
    if (pad_bytes) HIPCHK(hipMalloc(&pad, pad_bytes));               // leaked on purpose: shifts where the next block lands
    uint64_t* nb_ = nullptr;
    HIPCHK(hipMalloc((void**)&nb_, 8ull * h->nbloom));
    HIPCHK(hipMemcpy(nb_, h->bloom, 8ull * h->nbloom, hipMemcpyDeviceToDevice));
    HIPCHK(hipDeviceSynchronize());
    h->bloom = nb_;
    return AIX_OK;
}

// Move any subset of a 23-mer handle's device arrays into freshly allocated blocks (mask: 1 MPHF records, 2 side index, 4 unfiled keys, 8 verification
// table, 16 absence filter); the old blocks are NOT freed, so that the new ones come from other memory. Placement experiments only
// (scripts/experiments/r3/rehome.py).
extern "C" int aix_debug_rehome(aix_index_t* h, uint32_t mask) {
    if (!h || h->k != 23) return AIX_ERR_ARG;
    DevGuard g(h->device);
    auto move = [&](void** field, uint64_t bytes) -> int {
        if (!*field || !bytes) return AIX_OK;
        void* nb_ = nullptr;
        HIPCHK(hipMalloc(&nb_, bytes));
        HIPCHK(hipMemcpy(nb_, *field, bytes, hipMemcpyDeviceToDevice));
        *field = nb_;
        return AIX_OK;
    };
    int st = AIX_OK;
    if (!st && (mask & 1)) st = move((void**)&h->recs, (uint64_t)((h->B + 15) / 16) * sizeof(BvRec));
    if (!st && (mask & 2)) st = move((void**)&h->side, 4ull * h->n);
    if (!st && (mask & 4)) st = move((void**)&h->unfiled, (uint64_t)h->n_unfiled * sizeof(KeyRec));
    if (!st && (mask & 8) && !h->bk_borrowed) st = move((void**)&h->bk, (uint64_t)h->nb * 8 * sizeof(BkEntry));
    if (!st && (mask & 16)) st = move((void**)&h->bloom, 8ull * h->nbloom);
    HIPCHK(hipDeviceSynchronize());
    return st;
}

// GPU self-test hook of the suite: lower bounds of keys[i] and keys[i] + 1 in a sorted u16 array, by the wave-wide search the partition kernels
// use to find a partition's chunks (out[2 i], out[2 i + 1]); all pointers are device pointers
extern "C" int aix_selftest_lower_bound_dev(const uint16_t* d_sorted, uint32_t n, const uint32_t* d_keys, uint32_t nkeys, uint32_t* d_out, void* stream) {
    if ((n && !d_sorted) || (nkeys && (!d_keys || !d_out))) return AIX_ERR_ARG;
    HIPCHK(launch_selftest_lower_bound(d_sorted, n, d_keys, nkeys, d_out, (hipStream_t)stream));
    return AIX_OK;
}

// device addresses of a handle's arrays (same order as the mask bits of aix_debug_rehome): experiments only
extern "C" int aix_debug_pointers(const aix_index_t* h, uint64_t out[5]) {
    if (!h || !out) return AIX_ERR_ARG;
    out[0] = (uint64_t)(uintptr_t)h->recs; out[1] = (uint64_t)(uintptr_t)h->side; out[2] = (uint64_t)(uintptr_t)h->unfiled;
    out[3] = (uint64_t)(uintptr_t)h->bk; out[4] = (uint64_t)(uintptr_t)h->bloom;
    return AIX_OK;
}

extern "C" int aix_index_set_tf_13(aix_index_t* h, const uint64_t* tf) {
    if (!h || !tf) return AIX_ERR_ARG;
    if (h->k != 13) return AIX_ERR_MODE;
    DevGuard g(h->device);
    HIPCHK(hipMemcpy(h->tf13_mphf, tf, 8 * AIX_TOTAL_13MERS, hipMemcpyHostToDevice));
    h->pos_total_known = false;
    HIPCHK(launch_tf13_to_code_order(h->perm13, h->tf13_mphf, h->tf13_code, 0));
    HIPCHK(hipStreamSynchronize(0));
    return AIX_OK;
}

extern "C" int aix_index_get_tf(const aix_index_t* h, void* out, uint64_t out_bytes) {
    if (!h || !out) return AIX_ERR_ARG;
    DevGuard g(h->device);
    if (h->k == 13) {
        if (out_bytes < 8 * AIX_TOTAL_13MERS) return AIX_ERR_ARG;
        HIPCHK(hipMemcpy(out, h->tf13_mphf, 8 * AIX_TOTAL_13MERS, hipMemcpyDeviceToHost));
        return AIX_OK;
    }
    if (out_bytes < 4 * h->n) return AIX_ERR_ARG;
    if (h->n == 0) return AIX_OK;
    uint32_t* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, 4 * h->n));
    hipError_t e = launch_extract_tf(h->dev(), d, nullptr, 0);
    if (e == hipSuccess) e = hipMemcpy(out, d, 4 * h->n, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIPCHK(e);
    return AIX_OK;
}

extern "C" int aix_index_get_checker(const aix_index_t* h, uint64_t* out, uint64_t n) {
    if (!h || !out) return AIX_ERR_ARG;
    if (h->k != 23) return AIX_ERR_MODE;
    if (n < h->n) return AIX_ERR_ARG;
    if (h->n == 0) return AIX_OK;
    DevGuard g(h->device);
    uint64_t* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, 8 * h->n));
    hipError_t e = launch_extract_tf(h->dev(), nullptr, d, 0);
    if (e == hipSuccess) e = hipMemcpy(out, d, 8 * h->n, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIPCHK(e);
    return AIX_OK;
}

// ---------------------------------------------------------------------------------------------
// device-pointer entry points
// ---------------------------------------------------------------------------------------------
static int lookup_ascii_dev(aix_index_t* h, const char* d_kmers, uint64_t N, int mode, LookupOut o, void* stream) {
    if (!h || (N && !d_kmers)) return AIX_ERR_ARG;
    if (N == 0) return AIX_OK;
    DevGuard g(h->device);
    const IndexDev d = h->dev();
    if (h->k == 23) {
        if (h->n == 0) {                                       // empty index: every answer is 0
            return AIX_ERR_UNSUPPORTED;
        }
        HIPCHK(launch_lookup23_ascii(d, (const uint8_t*)d_kmers, N, mode, o, (hipStream_t)stream));
    } else {
        if (mode == MODE_KIDSTRAND) return AIX_ERR_MODE;       // hash_map is null in 13-mer mode (kid / strand need the checker)
        HIPCHK(launch_lookup13_ascii(d, (const uint8_t*)d_kmers, N, mode, o, (hipStream_t)stream));
    }
    return AIX_OK;
}

extern "C" int aix_tf_batch_ascii_dev(aix_index_t* h, const char* d_kmers, uint64_t N, uint32_t* d_out, void* stream) {
    if (N && !d_out) return AIX_ERR_ARG;
    LookupOut o{};
    o.tf = d_out;
    return lookup_ascii_dev(h, d_kmers, N, MODE_TF, o, stream);
}
// instrumentation: d_out[i] = number of MPHF + key records the tf query i reads under the handle's current settings
extern "C" int aix_lines_batch_ascii_dev(aix_index_t* h, const char* d_kmers, uint64_t N, uint32_t* d_out, void* stream) {
    if (N && !d_out) return AIX_ERR_ARG;
    if (h && h->k != 23) return AIX_ERR_MODE;
    LookupOut o{};
    o.tf = d_out;
    return lookup_ascii_dev(h, d_kmers, N, MODE_LINES, o, stream);
}
extern "C" int aix_hash_batch_ascii_dev(aix_index_t* h, const char* d_kmers, uint64_t N, uint64_t* d_out, void* stream) {
    if (N && !d_out) return AIX_ERR_ARG;
    LookupOut o{};
    o.u64a = d_out;
    return lookup_ascii_dev(h, d_kmers, N, MODE_HASH, o, stream);
}
extern "C" int aix_kid_strand_batch_ascii_dev(aix_index_t* h, const char* d_kmers, uint64_t N, uint64_t* d_kid, uint8_t* d_strand, void* stream) {
    LookupOut o{};
    o.u64a = d_kid; o.strand = d_strand;
    return lookup_ascii_dev(h, d_kmers, N, MODE_KIDSTRAND, o, stream);
}
extern "C" int aix_tf_both_batch_ascii_dev(aix_index_t* h, const char* d_kmers, uint64_t N, uint64_t* d_fwd, uint64_t* d_rc, void* stream) {
    LookupOut o{};
    o.u64a = d_fwd; o.u64b = d_rc;
    return lookup_ascii_dev(h, d_kmers, N, MODE_BOTH, o, stream);
}
extern "C" int aix_tf_total_batch_ascii_dev(aix_index_t* h, const char* d_kmers, uint64_t N, uint64_t* d_out, void* stream) {
    if (N && !d_out) return AIX_ERR_ARG;
    LookupOut o{};
    o.u64a = d_out;
    return lookup_ascii_dev(h, d_kmers, N, MODE_TOTAL, o, stream);
}
extern "C" int aix_tf_batch_codes_dev(aix_index_t* h, const uint64_t* d_codes, uint64_t N, uint32_t* d_out, void* stream) {
    if (!h || (N && (!d_codes || !d_out))) return AIX_ERR_ARG;
    if (h->k != 23) return AIX_ERR_MODE;
    if (N == 0) return AIX_OK;
    if (h->n == 0) return AIX_ERR_UNSUPPORTED;
    DevGuard g(h->device);
    HIPCHK(launch_lookup23_codes(h->dev(), d_codes, N, d_out, (hipStream_t)stream));
    return AIX_OK;
}
extern "C" int aix_tf_batch_ragged_dev(aix_index_t* h, const char* d_bytes, const uint64_t* d_offs, uint64_t N, uint32_t* d_out, void* stream) {
    if (!h || (N && (!d_offs || !d_out))) return AIX_ERR_ARG;
    if (N == 0) return AIX_OK;
    DevGuard g(h->device);
    if (h->k == 23) {
        if (h->n == 0) return AIX_ERR_UNSUPPORTED;
        HIPCHK(launch_lookup23_ragged(h->dev(), (const uint8_t*)d_bytes, d_offs, N, d_out, (hipStream_t)stream));
    } else {
        HIPCHK(launch_lookup13_ragged(h->dev(), (const uint8_t*)d_bytes, d_offs, N, d_out, (hipStream_t)stream));
    }
    return AIX_OK;
}
extern "C" int aix_coverage_batch_dev(aix_index_t* h, const char* d_seqs, const uint64_t* d_offs, uint64_t M, uint64_t total_bytes, uint32_t cutoff,
                                      uint32_t* d_out, const uint64_t* d_out_offs, void* stream) {
    if (!h || (M && (!d_seqs || !d_offs || !d_out || !d_out_offs))) return AIX_ERR_ARG;
    if (M == 0 || total_bytes == 0) return AIX_OK;
    if (h->k == 23 && h->n == 0) return AIX_ERR_UNSUPPORTED;
    DevGuard g(h->device);
    HIPCHK(launch_coverage(h->dev(), (const uint8_t*)d_seqs, d_offs, M, total_bytes, cutoff, d_out, d_out_offs, (hipStream_t)stream));
    return AIX_OK;
}

static int ensure_count_workspace(aix_index* h, uint64_t need, hipStream_t s);

// K13 in three steps, so that a file can be counted part by part (aix_ingest.hip) as well as in one call: begin() orders the call behind
// the previous user of the handle's workspace and zeroes the output, add() counts one PLAIN buffer into it (any number of times; every
// window of the concatenated stream must lie inside exactly one of the buffers), end() finishes the table. The caller holds count_mutex
// from begin to end.
int count13_begin_locked(aix_index* h, uint64_t* d_tf_out, hipStream_t s) {
    // the scratch table / partition workspace belong to the handle: calls are ordered behind one another even when they come
    // in on different streams (the mutex only orders the enqueueing)
    if (h->work13_done) HIPCHK(hipStreamWaitEvent(s, h->work13_done, 0));
    else HIPCHK(hipEventCreateWithFlags(&h->work13_done, hipEventDisableTiming));
    h->c13_atomics = getenv("AIX_COUNT13_ATOMICS") != nullptr || !h->perm13_bijective;   // env: A/B switch for measurements / tests
    h->c13_added = false;
    if (h->c13_atomics) {
        if (!h->scratch13) {
            HIPCHK(hipMalloc((void**)&h->scratch13, 8 * AIX_TOTAL_13MERS));
            h->device_bytes += 8 * AIX_TOTAL_13MERS;
        }
        HIPCHK(hipMemsetAsync(h->scratch13, 0, 8 * AIX_TOTAL_13MERS, s));
    }
    HIPCHK(hipMemsetAsync(d_tf_out, 0, 8 * AIX_TOTAL_13MERS, s));
    return AIX_OK;
}

int count13_add_locked(aix_index* h, const char* d_plain, uint64_t len, uint64_t* d_tf_out, hipStream_t s) {
    if (h->c13_atomics) {
        // scattered u64 memory-side atomics into the code-ordered table (slower; kept as the independent cross-check)
        HIPCHK(launch_count13_plain((const uint8_t*)d_plain, len, h->scratch13, s));
        h->c13_added = true;
        return AIX_OK;
    }
    // The partitioned path indexes windows and chunks with 32 bits: buffers are cut into pieces of at most `piece` (2^31) window starts.
    // A window belongs to the piece that holds its first byte; a piece is handed its 12 following bytes as well, so the
    // cut needs no record boundary and every window is counted exactly once. Pieces after the first add to the table.
    uint64_t piece = 1ull << 31;
    if (const char* e = getenv("AIX_COUNT13_PIECE")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 1 && v <= (1ull << 31)) piece = v; }
    const uint64_t nwin = len >= 13 ? len - 12 : 0;
    if (nwin == 0) return AIX_OK;
    for (;;) {                                                               // a workspace that does not fit halves the piece (down to 4096 windows)
        const uint64_t pw = std::min(nwin, piece);
        const int stw = ensure_count_workspace(h, count13_workspace_bytes(pw + 12), s);
        if (stw == AIX_OK) break;
        if (stw != AIX_ERR_NOMEM || pw <= 4096) return stw;
        piece = pw / 2;
    }
    HIPCHK(hipMemsetAsync(h->work13, 0, 4, s));                               // the error word of the workspace
    for (uint64_t first = 0; first < nwin; first += piece) {
        const uint64_t w = std::min(piece, nwin - first);
        HIPCHK(launch_count13_partitioned((const uint8_t*)d_plain + first, w + 12, h->work13, nullptr, h->perm13, d_tf_out, h->c13_added ? 1 : 0, s));   // fused permutation
        h->c13_added = true;
    }
    // a chunk id outside a workgroup's region would have dropped counts: the kernels raise the error word instead of
    // staying silent, and the call fails (costs one stream wait per call; the table is complete when this returns)
    uint32_t dropped = 0;
    HIPCHK(hipMemcpyAsync(&dropped, h->work13, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (dropped) { set_last_error("count13: chunk region exhausted (partition workspace undersized)"); return AIX_ERR_UNSUPPORTED; }
    return AIX_OK;
}

int count13_end_locked(aix_index* h, uint64_t* d_tf_out, hipStream_t s) {
    int st = AIX_OK;
    if (h->c13_atomics) {
        const hipError_t e = launch_scatter13_to_mphf(h->perm13, h->scratch13, d_tf_out, h->perm13_bijective ? 0 : 1, s);
        if (e != hipSuccess) { set_last_error(std::string("count13 scatter: ") + hipGetErrorString(e)); st = AIX_ERR_HIP; }
    }
    (void)hipEventRecord(h->work13_done, s);
    return st;
}

extern "C" int aix_count13_dev(aix_index_t* h, const char* d_plain, uint64_t len, uint64_t* d_tf_out, void* stream) {
    if (!h || !d_tf_out || (len && !d_plain)) return AIX_ERR_ARG;
    if (h->k != 13) return AIX_ERR_MODE;
    DevGuard g(h->device);
    std::lock_guard<std::mutex> lk(h->count_mutex);
    hipStream_t s = (hipStream_t)stream;
    int st = count13_begin_locked(h, d_tf_out, s);
    if (st) return st;
    st = count13_add_locked(h, d_plain, len, d_tf_out, s);
    const int st2 = count13_end_locked(h, d_tf_out, s);
    return st ? st : st2;
}

// grow-only per-handle workspace of the counting paths (13-mer partitions; 23-mer slot stream + partitions). Calls are ordered
// behind one another even when they come in on different streams (the mutex only orders the enqueueing).
static int ensure_count_workspace(aix_index* h, uint64_t need, hipStream_t s) {
    if (need <= h->work13_bytes) return AIX_OK;
    HIPCHK(hipStreamSynchronize(s));
    if (h->work13) { (void)hipFree(h->work13); h->device_bytes -= h->work13_bytes; h->work13 = nullptr; h->work13_bytes = 0; }
    // AIX_COUNT_TEST_WORKSPACE_MAX: test hook, requests above this many bytes "do not fit"
    const char* lim = getenv("AIX_COUNT_TEST_WORKSPACE_MAX");
    const hipError_t e = (lim && need > strtoull(lim, nullptr, 10)) ? hipErrorOutOfMemory : hipMalloc(&h->work13, need);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();                                               // the failed request is not a sticky error
        h->work13 = nullptr;
        set_last_error("count workspace of " + std::to_string(need) + " bytes does not fit in device memory");
        return AIX_ERR_NOMEM;
    }
    HIPCHK(e);
    h->work13_bytes = need;
    h->device_bytes += need;
    return AIX_OK;
}

extern "C" int aix_count23_fixed_dev(aix_index_t* h, const char* d_plain, uint64_t len, int canon_mode, uint32_t* d_tf_out, void* stream) {
    if (!h || !d_tf_out || (len && !d_plain)) return AIX_ERR_ARG;
    if (h->k != 23) return AIX_ERR_MODE;
    if (canon_mode < 0 || canon_mode > 2) return AIX_ERR_ARG;
    DevGuard g(h->device);
    hipStream_t s = (hipStream_t)stream;
    const uint64_t nwin = len >= 23 ? len - 22 : 0;
    if (nwin == 0 || h->n == 0) return AIX_OK;
    // Two back ends, same result. (a) one memory-side atomic per found window: ~23 G scattered atomics/s on MI355X, which bounds the
    // kernel once a probe costs a single line. (b) the slots are streamed out (4 B per window) and added into tf[] by the
    // chunked-partition + LDS histogram of the 13-mer counter: no global atomics at all. (b) handles 2^26 slots per pass over the slot
    // stream (2048 partitions of 32 768 bins; a larger key set takes ceil(n / 2^26) passes) and pays a fixed cost, so short buffers keep
    // (a). Which one ran is reported by aix_index_info (count23_backend / count23_passes). AIX_COUNT23_ATOMICS=1 forces (a); AIX_COUNT23_HIST_MIN moves
    // the threshold (tests run (b) on small inputs).
    uint64_t hist_min = 1ull << 22;
    if (const char* e = getenv("AIX_COUNT23_HIST_MIN")) hist_min = strtoull(e, nullptr, 10);
    const bool use_hist = nwin >= hist_min && getenv("AIX_COUNT23_ATOMICS") == nullptr;
    if (!use_hist) {
        HIPCHK(launch_count23_fixed(h->dev(), (const uint8_t*)d_plain, len, canon_mode, d_tf_out, s));
        h->c23_backend = 1; h->c23_passes = 0;
        return AIX_OK;
    }
    // (c) counting the distinct k-mers of the reads first (K1: MSD partition + per-bucket LDS hash, pieces merged) and probing each of them ONCE:
    // K1's LDS-bound 21 ps per window beat a 128-byte line per window (24 - 27 ps) as soon as the distinct k-mers — at most n — are few against the
    // windows: 34.7 / 69.8 / 138 / 272 ms against 43.0 / 85.7 / 171 / 341 ms at 32 / 64 / 128 / 256 windows per key (5e7 keys, one handle), config 4
    // 531 against 617 - 679 ms; the extra probe per distinct k-mer (28 ps) is paid back from ~6 windows per key. Taken from 8 windows per key and
    // 2^29 windows up (AIX_COUNT23_VIA_K1=0 / 1 forces); if K1's scratch does not fit the call goes on with (b). Same histogram: every window is
    // counted under the same canonical form, skipped for the same bytes, and two distinct k-mers never share a slot.
    bool via_k1 = nwin >= (1ull << 29) && nwin / 8 >= h->n;
    if (const char* e = getenv("AIX_COUNT23_VIA_K1")) via_k1 = atoi(e) != 0;
    if (via_k1) {
        uint64_t *dk = nullptr, *dc = nullptr, dn = 0;
        uint64_t piece = 0;
        if (const char* e = getenv("AIX_DISTINCT_PIECE")) piece = strtoull(e, nullptr, 10);
        const hipError_t e = distinct_from_plain((const uint8_t*)d_plain, len, 23, canon_mode, 1, piece, &dk, &dc, &dn, s);
        if (e == hipSuccess) {
            const hipError_t e2 = launch_add_counts23(h->dev_slots(), dk, dc, dn, d_tf_out, s);
            const hipError_t e3 = hipStreamSynchronize(s);                     // the K1 result goes back to the block cache idle
            if (dk) pool_free(dk);
            if (dc) pool_free(dc);
            HIPCHK(e2);
            HIPCHK(e3);
            h->c23_backend = 3; h->c23_passes = 0;
            return AIX_OK;
        }
        (void)hipGetLastError();
        if (e != hipErrorOutOfMemory) { set_last_error(std::string("count23 through K1: ") + hipGetErrorString(e)); return AIX_ERR_HIP; }
    }
    uint32_t range_bits = 26;                                                  // AIX_COUNT23_TEST_RANGE_BITS: test hook, several slot ranges on a small key set
    if (const char* e = getenv("AIX_COUNT23_TEST_RANGE_BITS")) { const int v = atoi(e); if (v >= 4 && v <= 26) range_bits = (uint32_t)v; }
    std::lock_guard<std::mutex> lk(h->count_mutex);
    if (h->work13_done) HIPCHK(hipStreamWaitEvent(s, h->work13_done, 0));
    else HIPCHK(hipEventCreateWithFlags(&h->work13_done, hipEventDisableTiming));
    struct RecordOnExit {
        hipEvent_t ev; hipStream_t st;
        ~RecordOnExit() { (void)hipEventRecord(ev, st); }
    } record_on_exit{h->work13_done, s};
    // windows per pass. Every pass pays for its chunk directory (sort, clears) and leaves one partly filled 512-byte chunk per (workgroup,
    // partition) pair — up to 2^20 of them — for the histogram kernel to read, so long passes win: 200 M reads in 809 / 788 / 775 / 772 ms with
    // 2^28 / 2^29 / 2^30 / 2^31 windows per pass (same box). 2^30: 4 GiB of slots (twice with the second stream) + ~2.6 GiB of partitions.
    uint64_t piece = 1ull << 30;
    if (const char* e = getenv("AIX_COUNT23_PIECE")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 1 && v <= (1ull << 31)) piece = v; }
    const IndexDev d = h->dev();
    // More than one piece: the probe of piece i + 1 (HBM lines + hash arithmetic, no LDS) runs on a second stream while piece i is
    // partitioned and added on the caller's stream (LDS-bound, one 152 KiB workgroup per CU) — two slot buffers, one partition
    // workspace, events both ways. AIX_COUNT23_OVERLAP=0 keeps everything on the caller's stream (A/B switch).
    // A workspace that does not fit (2^30 windows: ~10.6 GiB) halves the pass instead of failing the call (down to 4096 windows).
    uint64_t pw, part_bytes, slot_bytes;
    bool overlap;
    for (;;) {
        pw = std::min(piece, nwin);
        part_bytes = (count13_workspace_bytes(pw + 12) + 255) / 256 * 256;
        overlap = !d.mk && nwin > pw;
        if (const char* e = getenv("AIX_COUNT23_OVERLAP")) overlap = overlap && atoi(e) != 0;
        slot_bytes = (4 * pw + 255) / 256 * 256;
        const int st = ensure_count_workspace(h, part_bytes + (overlap ? 2 : 1) * slot_bytes, s);
        if (st == AIX_OK) break;
        if (st != AIX_ERR_NOMEM || pw <= 4096) return st;
        piece = pw / 2;
    }
    uint32_t* slot_buf[2] = {(uint32_t*)((uint8_t*)h->work13 + part_bytes), (uint32_t*)((uint8_t*)h->work13 + part_bytes + (overlap ? slot_bytes : 0))};
    // an error in the middle of the loop would return with the probe stream still writing into the workspace, and the event recorded on
    // exit (caller's stream only) would let the next counting call in under it: a failing call waits for the probe stream first
    struct ProbeDrainOnError {
        aix_index* h;
        bool armed = true;
        ~ProbeDrainOnError() { if (armed && h->probe_stream) (void)hipStreamSynchronize(h->probe_stream); if (armed && h->hist_stream) (void)hipStreamSynchronize(h->hist_stream); }
    } probe_drain{h};
    HIPCHK(hipMemsetAsync(h->work13, 0, 4, s));                               // the error word of the partition workspace
    // AIX_COUNT23_HIST_CUS=n[,style] (A/B switch, read when the streams are first made): the partition + histogram kernels get n of the 256 CUs
    // and the probe the others, through CU-masked streams. The split kernel takes a whole CU (152 KiB of LDS, 16 waves of 128 VGPRs), so on
    // shared CUs the two kernels alternate workgroup by workgroup instead of running side by side. style 0: the low n bits of the mask, 1: every
    // (256 / n)-th bit.
    hipStream_t hs = s;
    if (overlap) {
        if (!h->probe_stream) {
            int hist_cus = 0, style = 0;
            if (const char* e = getenv("AIX_COUNT23_HIST_CUS")) { hist_cus = atoi(e); if (const char* c = strchr(e, ',')) style = atoi(c + 1); }
            if (hist_cus >= 8 && hist_cus <= 224) {
                uint32_t mh[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mp[8];
                const int step = style ? 256 / hist_cus : 1;
                for (int i = 0, c = 0; c < hist_cus && i < 256; i += step, ++c) mh[i >> 5] |= 1u << (i & 31);
                for (int w = 0; w < 8; ++w) mp[w] = ~mh[w];
                HIPCHK(hipExtStreamCreateWithCUMask(&h->probe_stream, 8, mp));
                HIPCHK(hipExtStreamCreateWithCUMask(&h->hist_stream, 8, mh));
            } else
            HIPCHK(hipStreamCreateWithFlags(&h->probe_stream, hipStreamNonBlocking));
            for (int i = 0; i < 2; ++i) {
                HIPCHK(hipEventCreateWithFlags(&h->probe_ev[i], hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&h->hist_ev[i], hipEventDisableTiming));
            }
            HIPCHK(hipEventCreateWithFlags(&h->start_ev, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(h->start_ev, s));                                // the reads (and whatever else the caller queued) are ready when the first probe starts
        HIPCHK(hipStreamWaitEvent(h->probe_stream, h->start_ev, 0));
        if (h->hist_stream) { hs = h->hist_stream; HIPCHK(hipStreamWaitEvent(hs, h->start_ev, 0)); }
    }
    // the slot-stream probe of the counter runs best with two lanes per bucket line (38.7-40.4 against 42.5-42.7 ms per 10 M reads with
    // eight, same box): nothing but the 4-byte slot leaves the kernel, so fewer, wider reads per probe win; lookups keep eight
    const IndexDev dc = h->dev_slots();
    // probe kernel of the slot stream: one window per lane (k_probe23_slots) or a run of 16 / 32 windows per lane (k_run23_slots: the bytes are
    // encoded once per run). AIX_COUNT23_RUN=0 / 16 / 32 (A/B switch).
    int run_w = 0;
    if (const char* e = getenv("AIX_COUNT23_RUN")) { const int v = atoi(e); if (v == 16 || v == 32) run_w = v; }
    auto probe = [&](const uint8_t* p, uint64_t n, uint32_t* out, hipStream_t st) {
        return (run_w && dc.bk) ? launch_run23_slots(dc, p, n, canon_mode, out, run_w, st) : launch_probe23_slots(dc, p, n, canon_mode, out, st);
    };
    uint64_t ip = 0;
    for (uint64_t first = 0; first < nwin; first += pw, ++ip) {
        const uint64_t w = std::min(pw, nwin - first);
        uint32_t* slots = slot_buf[overlap ? (ip & 1) : 0];
        if (d.mk) {                                                            // 32 consecutive windows per lane; word 1 of the workspace = "undecided windows" flag
            HIPCHK(hipMemsetAsync((uint32_t*)h->work13 + 1, 0, 4, s));
            HIPCHK(launch_stream23_slots(d, (const uint8_t*)d_plain + first, w + 22, canon_mode, slots, (uint32_t*)h->work13 + 1, s));
        } else if (overlap) {
            const int b = (int)(ip & 1);
            if (ip >= 2) HIPCHK(hipStreamWaitEvent(h->probe_stream, h->hist_ev[b], 0));      // piece ip - 2 has been read out of this buffer
            HIPCHK(probe((const uint8_t*)d_plain + first, w + 22, slots, h->probe_stream));
            HIPCHK(hipEventRecord(h->probe_ev[b], h->probe_stream));
            HIPCHK(hipStreamWaitEvent(hs, h->probe_ev[b], 0));
        } else {
            HIPCHK(probe((const uint8_t*)d_plain + first, w + 22, slots, s));
        }
        uint32_t passes = 0;
        HIPCHK(launch_histogram_slots(slots, w, h->work13, d_tf_out, h->n, hs, range_bits, &passes));
        h->c23_backend = 2; h->c23_passes = passes;
        if (overlap) HIPCHK(hipEventRecord(h->hist_ev[ip & 1], hs));
    }
    if (hs != s && ip) HIPCHK(hipStreamWaitEvent(s, h->hist_ev[(ip - 1) & 1], 0));
    uint32_t dropped = 0;
    HIPCHK(hipMemcpyAsync(&dropped, h->work13, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (dropped) { set_last_error("count23: chunk region exhausted (partition workspace undersized)"); return AIX_ERR_UNSUPPORTED; }
    probe_drain.armed = false;                                                 // every probe has been waited for by a histogram on the caller's stream
    return AIX_OK;
}

extern "C" int aix_window_codes_dev(const char* d_plain, uint64_t len, int k, int canon_mode, uint64_t* d_codes, void* stream) {
    if ((len && !d_plain) || k < 1 || k > 32 || canon_mode < 0 || canon_mode > 2) return AIX_ERR_ARG;
    if (len < (uint64_t)k) return AIX_OK;
    if (!d_codes) return AIX_ERR_ARG;
    HIPCHK(launch_window_codes((const uint8_t*)d_plain, len, k, canon_mode, d_codes, (hipStream_t)stream));
    return AIX_OK;
}

extern "C" int aix_bench_gather_dev(const void* d_table, uint64_t n_elems, int elem_bytes, int unroll, uint64_t n_access, uint64_t seed, uint64_t* d_sink,
                                    void* stream) {
    if (!d_table || !d_sink || n_elems == 0 || (n_elems >> 32)) return AIX_ERR_ARG;
    HIPCHK(launch_gather((const uint8_t*)d_table, n_elems, elem_bytes, unroll, n_access, seed, d_sink, (hipStream_t)stream));
    return AIX_OK;
}

extern "C" int aix_synth_genome_dev(uint64_t seed, uint64_t length, char* d_out, void* stream) {
    if (length && !d_out) return AIX_ERR_ARG;
    HIPCHK(launch_synth_genome(seed, length, (uint8_t*)d_out, (hipStream_t)stream));
    return AIX_OK;
}
extern "C" int aix_synth_kmers_dev(uint64_t seed, uint64_t first, uint64_t N, int k, char* d_out, void* stream) {
    if ((N && !d_out) || k < 1 || k > 32) return AIX_ERR_ARG;
    HIPCHK(launch_synth_kmers(seed, first, N, k, (uint8_t*)d_out, (hipStream_t)stream));
    return AIX_OK;
}
extern "C" int aix_synth_mix23_dev(uint64_t seed, const char* d_genome, uint64_t genome_len, uint64_t first, uint64_t N, char* d_out, void* stream) {
    if (!d_genome || (N && !d_out) || genome_len < 23) return AIX_ERR_ARG;
    HIPCHK(launch_synth_mix23(seed, (const uint8_t*)d_genome, genome_len, first, N, (uint8_t*)d_out, (hipStream_t)stream));
    return AIX_OK;
}
extern "C" int aix_synth_reads_dev(uint64_t seed, const char* d_genome, uint64_t genome_len, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                                   int rc_half, uint32_t n_rate_ppm, char* d_out, void* stream) {
    if (!d_genome || (n_reads && !d_out) || read_len == 0 || genome_len < read_len) return AIX_ERR_ARG;
    HIPCHK(launch_synth_reads(seed, (const uint8_t*)d_genome, genome_len, first_read, n_reads, read_len, rc_half, n_rate_ppm, (uint8_t*)d_out,
                              (hipStream_t)stream));
    return AIX_OK;
}

// ---------------------------------------------------------------------------------------------
// host memory <-> pinned staging with several threads (one thread moves ~10 GB/s; PCIe 5 x16 wants ~50)
// ---------------------------------------------------------------------------------------------
namespace {
class CopyPool {
    struct Job { char* dst; const char* src; size_t bytes; };
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<Job> jobs;
    size_t pending = 0;
    bool stop = false;
    void run() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !jobs.empty(); });
                if (stop && jobs.empty()) return;
                j = jobs.back();
                jobs.pop_back();
            }
            memcpy(j.dst, j.src, j.bytes);
            std::lock_guard<std::mutex> lk(mu);
            if (--pending == 0) cv_done.notify_all();
        }
    }

public:
    explicit CopyPool(unsigned n) { for (unsigned i = 0; i < n; ++i) workers.emplace_back([this] { run(); }); }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_work.notify_all();
        for (auto& t : workers) t.join();
    }
    unsigned size() const { return (unsigned)workers.size(); }
    // one copy at a time per pool user (callers hold their handle's pipe mutex; the pool itself serialises with `busy`)
    std::mutex busy;
    void copy(void* dst, const void* src, size_t bytes) {
        const size_t parts = workers.size() + 1;
        if (bytes < (4u << 20) || parts == 1) { memcpy(dst, src, bytes); return; }
        std::lock_guard<std::mutex> only(busy);
        const size_t slice = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
        size_t off = slice;                                  // the caller takes the first slice itself
        {
            std::lock_guard<std::mutex> lk(mu);
            for (; off < bytes; off += slice) { jobs.push_back(Job{(char*)dst + off, (const char*)src + off, std::min(slice, bytes - off)}); ++pending; }
        }
        cv_work.notify_all();
        memcpy(dst, src, std::min(slice, bytes));
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};
CopyPool& copy_pool() {
    static CopyPool pool([] {
        unsigned n = std::thread::hardware_concurrency();
        n = n ? std::min(8u, std::max(1u, n / 2)) : 4u;          // measured on the MI355X host: 8 threads feed ~45 GB/s, more only contend
        if (const char* e = getenv("AIX_HOST_COPY_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) n = (unsigned)v; }
        return n - 1;                                        // + the calling thread
    }());
    return pool;
}
}  // namespace

// Large host-buffer batches: three staging sets (pinned host + device buffers + a stream + an event each). While set b's
// chunk is on the wire / in the kernel, the host threads fill the next set and drain the one before: H2D, kernel, D2H and the
// two host copies of different chunks overlap. (Round 1 staged synchronously from pageable memory: 25-35 GB/s in.)
struct HostPipe {
    static constexpr int S = 3;
    static constexpr uint64_t kChunkQ = 2ull << 20;          // queries per chunk
    static constexpr uint64_t kInBytes = kChunkQ * 23 + 64, kOutBytes = kChunkQ * 8;
    void* hin[S] = {};
    void* din[S] = {};
    void* hout[S][3] = {};
    void* dout[S][3] = {};
    hipStream_t st[S] = {};
    hipEvent_t ev[S] = {};
    bool ok = false;
    int init() {
        for (int b = 0; b < S; ++b) {
            if (hipHostMalloc(&hin[b], kInBytes, hipHostMallocDefault) != hipSuccess) return AIX_ERR_NOMEM;
            if (hipMalloc(&din[b], kInBytes) != hipSuccess) return AIX_ERR_NOMEM;
            if (hipStreamCreateWithFlags(&st[b], hipStreamNonBlocking) != hipSuccess) return AIX_ERR_HIP;
            if (hipEventCreateWithFlags(&ev[b], hipEventDisableTiming) != hipSuccess) return AIX_ERR_HIP;
        }
        ok = true;
        return AIX_OK;
    }
    int need_out(int j) {
        for (int b = 0; b < S; ++b) {
            if (hout[b][j]) continue;
            if (hipHostMalloc(&hout[b][j], kOutBytes, hipHostMallocDefault) != hipSuccess) return AIX_ERR_NOMEM;
            if (hipMalloc(&dout[b][j], kOutBytes) != hipSuccess) return AIX_ERR_NOMEM;
        }
        return AIX_OK;
    }
    ~HostPipe() {
        for (int b = 0; b < S; ++b) {
            if (st[b]) (void)hipStreamSynchronize(st[b]);
            if (hin[b]) (void)hipHostFree(hin[b]);
            if (din[b]) (void)hipFree(din[b]);
            for (int j = 0; j < 3; ++j) { if (hout[b][j]) (void)hipHostFree(hout[b][j]); if (dout[b][j]) (void)hipFree(dout[b][j]); }
            if (ev[b]) (void)hipEventDestroy(ev[b]);
            if (st[b]) (void)hipStreamDestroy(st[b]);
        }
    }
};

static void free_host_pipe(HostPipe* p) { delete p; }

static uint64_t pipe_fail_chunk() {                       // test hook, read once: the chunk of a large host batch whose launch "fails"
    static const uint64_t c = [] { const char* e = getenv("AIX_PIPE_TEST_FAIL_CHUNK"); return e ? strtoull(e, nullptr, 10) : ~0ull; }();
    return c;
}

// in: N elements of in_elem bytes each (host); outs[j]: N elements of out_elem[j] bytes (host, nullable). call(d_in, m, d_out0..2, stream).
template <typename F>
static int pipelined_host_batch(aix_index_t* h, const char* in, uint32_t in_elem, uint64_t N, const uint32_t out_elem[3], void* const outs[3], F&& call) {
    std::lock_guard<std::mutex> lk(h->pipe_mutex);          // one large host batch per handle at a time (they would share the wire anyway)
    if (!h->pipe) {
        h->pipe = new (std::nothrow) HostPipe();
        if (!h->pipe) return AIX_ERR_NOMEM;
        const int st = h->pipe->init();
        if (st) { delete h->pipe; h->pipe = nullptr; (void)hipGetLastError(); return st; }
    }
    HostPipe& P = *h->pipe;
    for (int j = 0; j < 3; ++j)
        if (outs[j]) { const int st = P.need_out(j); if (st) { (void)hipGetLastError(); return st; } }
    // every exit but the last one leaves with copies / kernels possibly queued on the pipe's streams; they touch the handle's staging
    // buffers (and caller-pinned outputs), which the next call reuses at once: a failing call drains the streams before it returns
    struct DrainOnError {
        HostPipe& P;
        bool armed = true;
        ~DrainOnError() { if (armed) for (int i = 0; i < HostPipe::S; ++i) (void)hipStreamSynchronize(P.st[i]); }
    } drain_on_error{P};
    CopyPool& pool = copy_pool();
    // buffers the caller has already pinned (hipHostMalloc / hipHostRegister, e.g. torch pinned tensors) go over the wire as they
    // are; only pageable memory is staged through the pipe's own pinned buffers
    auto pinned = [](const void* p) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    };
    const bool in_pinned = pinned(in);
    bool out_pinned[3] = {false, false, false};
    for (int j = 0; j < 3; ++j) out_pinned[j] = outs[j] && pinned(outs[j]);
    const uint64_t chunk = HostPipe::kChunkQ;
    const uint64_t nchunks = (N + chunk - 1) / chunk;
    auto drain = [&](uint64_t c) -> int {                    // chunk c has completed on the device: hand its answers to the caller
        const int b = (int)(c % HostPipe::S);
        HIPCHK(hipEventSynchronize(P.ev[b]));
        const uint64_t lo = c * chunk, m = std::min(chunk, N - lo);
        for (int j = 0; j < 3; ++j)
            if (outs[j] && !out_pinned[j]) pool.copy((char*)outs[j] + lo * out_elem[j], P.hout[b][j], m * out_elem[j]);
        return AIX_OK;
    };
    for (uint64_t c = 0; c < nchunks; ++c) {
        const int b = (int)(c % HostPipe::S);
        if (c >= (uint64_t)HostPipe::S) { const int st = drain(c - HostPipe::S); if (st) return st; }
        const uint64_t lo = c * chunk, m = std::min(chunk, N - lo);
        if (in_pinned) {
            HIPCHK(hipMemcpyAsync(P.din[b], in + lo * in_elem, m * in_elem, hipMemcpyHostToDevice, P.st[b]));
        } else {
            pool.copy(P.hin[b], in + lo * in_elem, m * in_elem);
            HIPCHK(hipMemcpyAsync(P.din[b], P.hin[b], m * in_elem, hipMemcpyHostToDevice, P.st[b]));
        }
        int st = call((const char*)P.din[b], m, P.dout[b][0], P.dout[b][1], P.dout[b][2], (void*)P.st[b]);
        if (!st && c == pipe_fail_chunk()) st = AIX_ERR_HIP;                  // AIX_PIPE_TEST_FAIL_CHUNK: fault injection (tests)
        if (st) return st;
        for (int j = 0; j < 3; ++j)
            if (outs[j]) HIPCHK(hipMemcpyAsync(out_pinned[j] ? (void*)((char*)outs[j] + lo * out_elem[j]) : P.hout[b][j], P.dout[b][j], m * out_elem[j],
                                               hipMemcpyDeviceToHost, P.st[b]));
        HIPCHK(hipEventRecord(P.ev[b], P.st[b]));
    }
    for (uint64_t c = nchunks > (uint64_t)HostPipe::S ? nchunks - HostPipe::S : 0; c < nchunks; ++c) { const int st = drain(c); if (st) return st; }
    drain_on_error.armed = false;
    return AIX_OK;
}

// ---------------------------------------------------------------------------------------------
// host-pointer twins: stage through HBM in bounded chunks, run the same kernels, copy back
// ---------------------------------------------------------------------------------------------
static constexpr uint64_t kSmall = 4096;         // up to here a host batch goes through the pinned, device-mapped staging of the handle

static int ensure_pinned(aix_index_t* h) {
    if (h->pin_in) return AIX_OK;
    void* in = nullptr;
    if (hipHostMalloc(&in, kSmall * 23 + 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return AIX_ERR_NOMEM; }
    for (int j = 0; j < 3; ++j)
        if (hipHostMalloc(&h->pin_out[j], kSmall * 8, hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            for (int i = 0; i < j; ++i) { (void)hipHostFree(h->pin_out[i]); h->pin_out[i] = nullptr; }
            (void)hipHostFree(in);
            return AIX_ERR_NOMEM;
        }
    memset(in, '\n', kSmall * 23 + 64);
    h->pin_in = in;
    return AIX_OK;
}

template <typename F>
static int chunked_ascii(aix_index_t* h, const char* kmers, uint64_t N, const uint32_t out_elem_bytes[3], void* const outs[3], F&& call) {
    if (!h || (N && !kmers)) return AIX_ERR_ARG;
    if (N == 0) return AIX_OK;
    DevGuard g(h->device);
    if (N <= kSmall) {                                     // latency path: the kernel reads the queries from, and writes the answers to, host memory
        std::lock_guard<std::mutex> lk(h->small_mutex);
        if (ensure_pinned(h) == AIX_OK) {
            void *din = nullptr, *dout[3] = {nullptr, nullptr, nullptr};
            hipError_t e = hipHostGetDevicePointer(&din, h->pin_in, 0);
            for (int j = 0; j < 3 && e == hipSuccess; ++j) e = hipHostGetDevicePointer(&dout[j], h->pin_out[j], 0);
            if (e == hipSuccess) {
                memcpy(h->pin_in, kmers, N * h->k);
                if (!h->small_stream && hipStreamCreateWithFlags(&h->small_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); h->small_stream = nullptr; }
                int st = call((const char*)din, N, dout[0], dout[1], dout[2], (void*)h->small_stream);   // own stream: no implicit ordering with the null stream
                if (st) return st;
                HIPCHK(hipStreamSynchronize(h->small_stream));
                for (int j = 0; j < 3; ++j)
                    if (outs[j]) memcpy(outs[j], h->pin_out[j], N * out_elem_bytes[j]);
                return AIX_OK;
            }
            (void)hipGetLastError();
        }
    }
    return pipelined_host_batch(h, kmers, h->k, N, out_elem_bytes, outs, call);
}

static bool empty23(const aix_index_t* h) { return h && h->k == 23 && h->n == 0; }

extern "C" int aix_tf_batch_ascii(aix_index_t* h, const char* kmers, uint64_t N, uint32_t* out) {
    if (N && !out) return AIX_ERR_ARG;
    if (empty23(h)) { memset(out, 0, 4 * N); return AIX_OK; }
    uint32_t eb[3] = {4, 0, 0};
    void* outs[3] = {out, nullptr, nullptr};
    return chunked_ascii(h, kmers, N, eb, outs, [&](const char* dq, uint64_t m, void* a, void*, void*, void* st) {
        return aix_tf_batch_ascii_dev(h, dq, m, (uint32_t*)a, st);
    });
}
extern "C" int aix_hash_batch_ascii(aix_index_t* h, const char* kmers, uint64_t N, uint64_t* out) {
    if (N && !out) return AIX_ERR_ARG;
    if (empty23(h)) return AIX_ERR_UNSUPPORTED;
    uint32_t eb[3] = {8, 0, 0};
    void* outs[3] = {out, nullptr, nullptr};
    return chunked_ascii(h, kmers, N, eb, outs, [&](const char* dq, uint64_t m, void* a, void*, void*, void* st) {
        return aix_hash_batch_ascii_dev(h, dq, m, (uint64_t*)a, st);
    });
}
extern "C" int aix_kid_strand_batch_ascii(aix_index_t* h, const char* kmers, uint64_t N, uint64_t* kid_out, uint8_t* strand_out) {
    if (h && h->k != 23) return AIX_ERR_MODE;
    if (empty23(h)) {
        if (kid_out) memset(kid_out, 0, 8 * N);
        if (strand_out) memset(strand_out, 0, N);
        return AIX_OK;
    }
    uint32_t eb[3] = {8, 1, 0};
    void* outs[3] = {kid_out, strand_out, nullptr};
    return chunked_ascii(h, kmers, N, eb, outs, [&](const char* dq, uint64_t m, void* a, void* b, void*, void* st) {
        return aix_kid_strand_batch_ascii_dev(h, dq, m, kid_out ? (uint64_t*)a : nullptr, strand_out ? (uint8_t*)b : nullptr, st);
    });
}
extern "C" int aix_tf_both_batch_ascii(aix_index_t* h, const char* kmers, uint64_t N, uint64_t* fwd_out, uint64_t* rc_out) {
    if (empty23(h)) {
        if (fwd_out) memset(fwd_out, 0, 8 * N);
        if (rc_out) memset(rc_out, 0, 8 * N);
        return AIX_OK;
    }
    uint32_t eb[3] = {8, 8, 0};
    void* outs[3] = {fwd_out, rc_out, nullptr};
    return chunked_ascii(h, kmers, N, eb, outs, [&](const char* dq, uint64_t m, void* a, void* b, void*, void* st) {
        return aix_tf_both_batch_ascii_dev(h, dq, m, fwd_out ? (uint64_t*)a : nullptr, rc_out ? (uint64_t*)b : nullptr, st);
    });
}
extern "C" int aix_tf_total_batch_ascii(aix_index_t* h, const char* kmers, uint64_t N, uint64_t* out) {
    if (N && !out) return AIX_ERR_ARG;
    if (empty23(h)) { memset(out, 0, 8 * N); return AIX_OK; }
    uint32_t eb[3] = {8, 0, 0};
    void* outs[3] = {out, nullptr, nullptr};
    return chunked_ascii(h, kmers, N, eb, outs, [&](const char* dq, uint64_t m, void* a, void*, void*, void* st) {
        return aix_tf_total_batch_ascii_dev(h, dq, m, (uint64_t*)a, st);
    });
}

extern "C" int aix_tf_batch_codes(aix_index_t* h, const uint64_t* codes, uint64_t N, uint32_t* out) {
    if (!h || (N && (!codes || !out))) return AIX_ERR_ARG;
    if (h->k != 23) return AIX_ERR_MODE;
    if (N == 0) return AIX_OK;
    if (h->n == 0) { memset(out, 0, 4 * N); return AIX_OK; }
    DevGuard g(h->device);
    const uint32_t eb[3] = {4, 0, 0};
    void* const outs[3] = {out, nullptr, nullptr};
    return pipelined_host_batch(h, (const char*)codes, 8, N, eb, outs, [&](const char* dq, uint64_t m, void* a, void*, void*, void* st) {
        return aix_tf_batch_codes_dev(h, (const uint64_t*)dq, m, (uint32_t*)a, st);
    });
}

extern "C" int aix_tf_batch_ragged(aix_index_t* h, const char* bytes, const uint64_t* offsets, uint64_t N, uint32_t* out) {
    if (!h || (N && (!offsets || !out))) return AIX_ERR_ARG;
    if (N == 0) return AIX_OK;
    if (empty23(h)) { memset(out, 0, 4 * N); return AIX_OK; }
    const uint64_t base = offsets[0], total = offsets[N] - base;
    if (total && !bytes) return AIX_ERR_ARG;
    DevGuard g(h->device);
    DevBuf db, doffs, dout;
    HIPCHK(db.alloc(total + 8));
    HIPCHK(doffs.alloc((N + 1) * 8));
    HIPCHK(dout.alloc(N * 4));
    std::vector<uint64_t> rel(N + 1);
    for (uint64_t i = 0; i <= N; ++i) rel[i] = offsets[i] - base;
    if (total) HIPCHK(hipMemcpy(db.p, bytes + base, total, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(doffs.p, rel.data(), (N + 1) * 8, hipMemcpyHostToDevice));
    int st = aix_tf_batch_ragged_dev(h, (const char*)db.p, (const uint64_t*)doffs.p, N, (uint32_t*)dout.p, nullptr);
    if (st) return st;
    HIPCHK(hipStreamSynchronize(0));
    HIPCHK(hipMemcpy(out, dout.p, N * 4, hipMemcpyDeviceToHost));
    return AIX_OK;
}

extern "C" int aix_coverage_batch(aix_index_t* h, const char* seqs, const uint64_t* offs, uint64_t M, uint32_t cutoff, uint32_t* out,
                                  const uint64_t* out_offs) {
    if (!h || (M && (!seqs || !offs || !out || !out_offs))) return AIX_ERR_ARG;
    if (M == 0) return AIX_OK;
    const uint64_t base = offs[0], total = offs[M] - base, obase = out_offs[0], ototal = out_offs[M] - obase;
    if (ototal == 0) return AIX_OK;
    if (empty23(h)) { memset(out + obase, 0, 4 * ototal); return AIX_OK; }
    DevGuard g(h->device);
    // latency path (one read, one contig window...): sequences, offsets and the profile live in pinned, device-mapped memory
    constexpr uint64_t kCovSeq = 128u << 10, kCovM = 1024, kCovPin = kCovSeq + 64 + 2 * 8 * (kCovM + 1) + 4 * kCovSeq;
    if (total <= kCovSeq && M <= kCovM && ototal <= kCovSeq) {
        std::lock_guard<std::mutex> lk(h->small_mutex);
        if (!h->pin_cov && hipHostMalloc(&h->pin_cov, kCovPin, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); h->pin_cov = nullptr; }
        void* dbase = nullptr;
        if (h->pin_cov && hipHostGetDevicePointer(&dbase, h->pin_cov, 0) == hipSuccess) {
            if (!h->small_stream && hipStreamCreateWithFlags(&h->small_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); h->small_stream = nullptr; }
            char* hp = (char*)h->pin_cov;
            uint64_t* hoffs = (uint64_t*)(hp + kCovSeq + 64);
            uint64_t* hooffs = hoffs + (kCovM + 1);
            uint32_t* hout = (uint32_t*)(hooffs + (kCovM + 1));
            memcpy(hp, seqs + base, total);
            memset(hp + total, '\n', 8);
            for (uint64_t i = 0; i <= M; ++i) { hoffs[i] = offs[i] - base; hooffs[i] = out_offs[i] - obase; }
            memset(hout, 0, 4 * ototal);
            char* dp = (char*)dbase;
            int st = aix_coverage_batch_dev(h, dp, (const uint64_t*)(dp + ((char*)hoffs - hp)), M, total, cutoff, (uint32_t*)(dp + ((char*)hout - hp)),
                                            (const uint64_t*)(dp + ((char*)hooffs - hp)), (void*)h->small_stream);
            if (st) return st;
            HIPCHK(hipStreamSynchronize(h->small_stream));
            memcpy(out + obase, hout, 4 * ototal);
            return AIX_OK;
        }
        (void)hipGetLastError();
    }
    DevBuf ds, doffs, dooffs, dout;
    HIPCHK(ds.alloc(total + 8));
    HIPCHK(doffs.alloc((M + 1) * 8));
    HIPCHK(dooffs.alloc((M + 1) * 8));
    HIPCHK(dout.alloc(ototal * 4));
    std::vector<uint64_t> rel(M + 1), orel(M + 1);
    for (uint64_t i = 0; i <= M; ++i) { rel[i] = offs[i] - base; orel[i] = out_offs[i] - obase; }
    if (total) HIPCHK(hipMemcpy(ds.p, seqs + base, total, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(doffs.p, rel.data(), (M + 1) * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dooffs.p, orel.data(), (M + 1) * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dout.p, 0, ototal * 4));
    int st = aix_coverage_batch_dev(h, (const char*)ds.p, (const uint64_t*)doffs.p, M, total, cutoff, (uint32_t*)dout.p, (const uint64_t*)dooffs.p, nullptr);
    if (st) return st;
    HIPCHK(hipStreamSynchronize(0));
    HIPCHK(hipMemcpy(out + obase, dout.p, ototal * 4, hipMemcpyDeviceToHost));
    return AIX_OK;
}

// ---------------------------------------------------------------------------------------------
// record normalisation (host): readers of count_kmers13.cpp:211-272 and count_kmers.cpp:250-295
// ---------------------------------------------------------------------------------------------
extern "C" int aix_detect_format(const char* buf, uint64_t len) {
    if (!buf || len == 0) return AIX_FMT_PLAIN;
    if (buf[0] == '\n') return AIX_FMT_PLAIN;
    if (buf[0] == '>') return AIX_FMT_FASTA;
    if (buf[0] == '@') return AIX_FMT_FASTQ;
    return AIX_FMT_PLAIN;
}

extern "C" int aix_normalize_reads(const char* buf, uint64_t len, int format, int fasta_mode, char* out, uint64_t* out_len) {
    if (!out || !out_len || (len && !buf)) return AIX_ERR_ARG;
    if (format == AIX_FMT_AUTO) format = aix_detect_format(buf, len);
    uint64_t o = 0;
    if (format == AIX_FMT_PLAIN) {
        memcpy(out, buf, len);
        o = len;
    } else if (format == AIX_FMT_FASTQ) {                      // count_kmers13.cpp:240-257: line 4i+1
        uint64_t pos = 0, line_no = 0;
        while (pos < len) {
            const char* nl = (const char*)memchr(buf + pos, '\n', len - pos);
            const uint64_t end = nl ? (uint64_t)(nl - buf) : len;
            if ((line_no & 3) == 1 && end > pos) { memcpy(out + o, buf + pos, end - pos); o += end - pos; out[o++] = '\n'; }
            line_no++;
            pos = end + 1;
        }
    } else if (format == AIX_FMT_FASTA && fasta_mode == 0) {  // count_kmers13.cpp:211-235
        uint64_t pos = 0;
        bool open = false;
        while (pos < len) {
            const char* nl = (const char*)memchr(buf + pos, '\n', len - pos);
            const uint64_t end = nl ? (uint64_t)(nl - buf) : len;
            if (end > pos) {
                if (buf[pos] == '>') { if (open) { out[o++] = '\n'; open = false; } }
                else { memcpy(out + o, buf + pos, end - pos); o += end - pos; open = true; }
            }
            pos = end + 1;
        }
        if (open) out[o++] = '\n';
    } else if (format == AIX_FMT_FASTA) {                      // count_kmers.cpp:250-295: '>' anywhere opens a record
        uint64_t i = 0;
        while (i < len && buf[i] != '>') i++;
        while (i < len) {
            uint64_t end = i + 1;
            while (end < len && buf[end] != '>') end++;
            uint64_t j = i;
            while (j < end && buf[j] != '\n') j++;
            j++;
            for (; j < end; ++j) { const char c = buf[j]; if (c != '\n' && c != '\r') out[o++] = c; }
            out[o++] = '\n';
            i = end;
        }
    } else {
        return AIX_ERR_ARG;
    }
    *out_len = o;
    return AIX_OK;
}

extern "C" int aix_normalize_reads_dev(const char* d_raw, uint64_t len, int format, int fasta_mode, char* d_out, uint64_t* out_len, void* stream) {
    if (!out_len || (len && (!d_raw || !d_out))) return AIX_ERR_ARG;
    if (format != AIX_FMT_PLAIN && format != AIX_FMT_FASTA && format != AIX_FMT_FASTQ) return AIX_ERR_ARG;   // no auto-detect on device buffers
    if (format == AIX_FMT_PLAIN) {
        if (len) HIPCHK(hipMemcpyAsync(d_out, d_raw, len, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        *out_len = len;
        return AIX_OK;
    }
    HIPCHK(normalise_device((const uint8_t*)d_raw, len, format, fasta_mode, (uint8_t*)d_out, out_len, (hipStream_t)stream));
    return AIX_OK;
}

// ---------------------------------------------------------------------------------------------
// A1/A2: positions index
// ---------------------------------------------------------------------------------------------
// host buffer -> HBM through the pinned, multi-threaded staging pipeline (aix_ingest.hip); returns when the bytes are on the device
static int upload_host(const char* buf, uint64_t len, uint8_t* d_dst, int device) {
    if (len == 0) return AIX_OK;
    ByteSource src;
    src.set_memory(buf, len);
    const int st = upload_pipelined(src, d_dst, device, 0);
    if (st) return st;
    HIPCHK(hipStreamSynchronize(0));
    return AIX_OK;
}

// positions_fill refuses a 13-mer tf table with an entry above 2^32 - 1 (32-bit fill counters, aix_positions.hip): that is a status, not a HIP failure
#define POSCHK(expr)                                                                                                                     \
    do {                                                                                                                                 \
        hipError_t _e = (expr);                                                                                                          \
        if (_e == hipErrorNotSupported) { (void)hipGetLastError(); set_last_error("positions: a 13-mer tf above 2^32 - 1"); return AIX_ERR_UNSUPPORTED; } \
        if (_e != hipSuccess) { set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e)); return AIX_ERR_HIP; }                  \
    } while (0)

// first window the reference's single worker looks at (hash.cpp:973-986): the start is pushed past any
// '\n', '~' or '?' found in the first k bytes, repeatedly
static uint64_t a2_start(const char* c, uint64_t len, uint64_t k = 23) {
    if (len < k) return 0;
    uint64_t start = 0;
    const uint64_t end = len;
    while (start < end - k + 1) {
        bool found = false;
        for (uint64_t i = start; i < start + k; ++i)
            if (c[i] == '\n' || c[i] == '~' || c[i] == '?') { start = i + 1; found = true; break; }
        if (!found) break;
    }
    return start;
}

extern "C" int aix_positions_fill(aix_index_t* h, const char* reads, uint64_t len, uint64_t* indices_out, uint64_t* positions_out, uint64_t positions_cap,
                                  uint64_t* total_out) {
    if (!h || !indices_out || (len && !reads)) return AIX_ERR_ARG;
    DevGuard g(h->device);
    const uint64_t n = h->n;
    uint64_t piece = 0;                                                        // 0: default (2^30 windows per sort)
    if (const char* e = getenv("AIX_POSITIONS_PIECE")) piece = strtoull(e, nullptr, 10);   // test hook: exercise the piece logic at small sizes
    DevBuf dind;
    HIPCHK(dind.alloc(8 * (n + 1)));
    if (n) HIPCHK(positions_indices(h->dev(), (uint64_t*)dind.p, 0));
    else HIPCHK(hipMemset(dind.p, 0, 8));
    { const int ds = download_to_host(indices_out, dind.p, 8 * (n + 1), 0); if (ds) return ds; }
    const uint64_t total = indices_out[n];
    if (total_out) *total_out = total;
    if (!positions_out) return AIX_OK;
    if (positions_cap < total) return AIX_ERR_ARG;
    if (total == 0) return AIX_OK;
    DevBuf dreads, dpos;
    HIPCHK(dreads.alloc(len + 8));
    HIPCHK(dpos.alloc(8 * total));
    HIPCHK(hipMemsetAsync(dpos.p, 0, 8 * total, 0));
    { const int us = upload_host(reads, len, (uint8_t*)dreads.p, h->device); if (us) return us; }
    POSCHK(positions_fill(h->dev_slots(), (const uint8_t*)dreads.p, len, a2_start(reads, len, h->k), (const uint64_t*)dind.p, (uint64_t*)dpos.p, piece, nullptr, 0, 0, &h->a2_backend));
    { const int ds = download_to_host(positions_out, dpos.p, 8 * total, 0); if (ds) return ds; }
    return AIX_OK;
}

// device-resident twin: reads already in HBM, indices (n+1) and positions (indices[n], caller-sized through
// aix_positions_total) written in HBM. `start` = aix_positions_start of the buffer's head (the caller holds the bytes).
extern "C" int aix_positions_total(aix_index_t* h, uint64_t* total_out) {
    if (!h || !total_out) return AIX_ERR_ARG;
    DevGuard g(h->device);
    *total_out = 0;
    if (h->n == 0) return AIX_OK;
    if (h->pos_total_known) { *total_out = h->pos_total; return AIX_OK; }       // sum of tf[]: fixed for the life of a 23-mer handle, reset by aix_index_set_tf_13
    DevBuf dind;
    HIPCHK(dind.alloc(8 * (h->n + 1)));
    HIPCHK(positions_indices(h->dev(), (uint64_t*)dind.p, 0));
    HIPCHK(hipMemcpy(total_out, (const uint64_t*)dind.p + h->n, 8, hipMemcpyDeviceToHost));
    h->pos_total = *total_out;
    h->pos_total_known = true;
    return AIX_OK;
}

extern "C" int aix_positions_fill_dev(aix_index_t* h, const char* d_reads, uint64_t len, uint64_t start, uint64_t* d_indices_out, uint64_t* d_positions_out,
                                      uint64_t positions_cap, void* stream) {
    if (!h || !d_indices_out || (len && !d_reads)) return AIX_ERR_ARG;
    DevGuard g(h->device);
    hipStream_t s = (hipStream_t)stream;
    const uint64_t n = h->n;
    uint64_t piece = 0;
    if (const char* e = getenv("AIX_POSITIONS_PIECE")) piece = strtoull(e, nullptr, 10);
    if (n) HIPCHK(positions_indices(h->dev(), d_indices_out, s));           // synchronises the stream
    else HIPCHK(hipMemsetAsync(d_indices_out, 0, 8, s));
    if (n == 0) return AIX_OK;
    uint64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, d_indices_out + n, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (total == 0) return AIX_OK;
    if (!d_positions_out || positions_cap < total) return AIX_ERR_ARG;
    HIPCHK(hipMemsetAsync(d_positions_out, 0, 8 * total, s));
    POSCHK(positions_fill(h->dev_slots(), (const uint8_t*)d_reads, len, start, d_indices_out, d_positions_out, piece, nullptr, 0, s, &h->a2_backend));
    return AIX_OK;
}

extern "C" int aix_positions_start(const char* reads, uint64_t len, uint64_t* start_out) {
    if (!start_out || (len && !reads)) return AIX_ERR_ARG;
    *start_out = a2_start(reads, len);
    return AIX_OK;
}
extern "C" int aix_positions_start_k(const char* reads, uint64_t len, int k, uint64_t* start_out) {
    if (!start_out || (len && !reads) || (k != 13 && k != 23)) return AIX_ERR_ARG;
    *start_out = a2_start(reads, len, (uint64_t)k);
    return AIX_OK;
}

// A2 over shards of the reads file (multi-GPU): tally, then fill with the counters of the earlier shards
extern "C" int aix_positions_bucket_counts(aix_index_t* h, const char* reads, uint64_t len, int first_shard, uint64_t* counts_out) {
    if (!h || !counts_out || (len && !reads)) return AIX_ERR_ARG;
    DevGuard g(h->device);
    const uint64_t n = h->n;
    if (n == 0) return AIX_OK;
    DevBuf dreads, dcnt;
    HIPCHK(dreads.alloc(len + 8));
    HIPCHK(dcnt.alloc(8 * n));
    { const int us = upload_host(reads, len, (uint8_t*)dreads.p, h->device); if (us) return us; }
    HIPCHK(hipMemset(dcnt.p, 0, 8 * n));
    HIPCHK(positions_bucket_counts(h->dev_slots(), (const uint8_t*)dreads.p, len, first_shard ? a2_start(reads, len, h->k) : 0, (unsigned long long*)dcnt.p, 0));
    { const int ds = download_to_host(counts_out, dcnt.p, 8 * n, 0); if (ds) return ds; }
    return AIX_OK;
}

extern "C" int aix_positions_fill_shard(aix_index_t* h, const char* reads, uint64_t len, int first_shard, uint64_t base_offset, const uint32_t* filled_init,
                                        uint64_t* positions_out, uint64_t positions_cap) {
    if (!h || !positions_out || (len && !reads)) return AIX_ERR_ARG;
    DevGuard g(h->device);
    const uint64_t n = h->n;
    if (n == 0) return AIX_OK;
    uint64_t piece = 0;
    if (const char* e = getenv("AIX_POSITIONS_PIECE")) piece = strtoull(e, nullptr, 10);
    DevBuf dind, dreads, dpos, dfill;
    HIPCHK(dind.alloc(8 * (n + 1)));
    HIPCHK(positions_indices(h->dev(), (uint64_t*)dind.p, 0));
    uint64_t total = 0;
    HIPCHK(hipMemcpy(&total, (const uint64_t*)dind.p + n, 8, hipMemcpyDeviceToHost));
    if (positions_cap < total) return AIX_ERR_ARG;
    if (total == 0) return AIX_OK;
    HIPCHK(dreads.alloc(len + 8));
    HIPCHK(dpos.alloc(8 * total));
    if (filled_init) {
        HIPCHK(dfill.alloc(4 * n));
        HIPCHK(hipMemcpy(dfill.p, filled_init, 4 * n, hipMemcpyHostToDevice));
    }
    { const int us = upload_host(reads, len, (uint8_t*)dreads.p, h->device); if (us) return us; }
    HIPCHK(hipMemset(dpos.p, 0, 8 * total));
    POSCHK(positions_fill(h->dev_slots(), (const uint8_t*)dreads.p, len, first_shard ? a2_start(reads, len, h->k) : 0, (const uint64_t*)dind.p, (uint64_t*)dpos.p, piece,
                          filled_init ? (const uint32_t*)dfill.p : nullptr, base_offset, 0, &h->a2_backend));
    { const int ds = download_to_host(positions_out, dpos.p, 8 * total, 0); if (ds) return ds; }
    return AIX_OK;
}

// ---------------------------------------------------------------------------------------------
// device-resident twins of the SHARD entry points (multi-GPU): a rank's share is already in HBM, the partial
// results stay in HBM for the collectives (RCCL), nothing crosses PCIe
// ---------------------------------------------------------------------------------------------
extern "C" int aix_index_scatter_shard_codes_dev(const void* pf_bytes, uint64_t pf_len, const uint64_t* d_codes, const uint32_t* d_counts, uint64_t n_keys,
                                                 uint64_t n_slots, int device, void* stream, uint64_t* d_checker_out, uint32_t* d_tf_out, uint32_t* d_occupied_out) {
    if (!pf_bytes || (n_keys && !d_codes) || !d_checker_out || !d_tf_out || !d_occupied_out || n_slots == 0 || n_keys > n_slots) return AIX_ERR_ARG;
    if (n_slots >> 32) return AIX_ERR_UNSUPPORTED;
    int st = check_device(device);
    if (st) return st;
    aix_index tmp;
    tmp.device = device; tmp.k = 23; tmp.n = n_slots;
    DevGuard g(device);
    st = upload_mphf(&tmp, (const uint8_t*)pf_bytes, pf_len);
    if (!st) st = scatter_device(&tmp, n_keys, n_slots, nullptr, d_codes, d_counts, d_checker_out, d_tf_out, d_occupied_out, (hipStream_t)stream);
    if (tmp.recs) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(tmp.recs); }
    tmp.recs = nullptr;
    return st;
}

extern "C" int aix_positions_indices_dev(aix_index_t* h, uint64_t* d_indices_out, void* stream) {
    if (!h || !d_indices_out) return AIX_ERR_ARG;
    DevGuard g(h->device);
    if (h->n == 0) { HIPCHK(hipMemsetAsync(d_indices_out, 0, 8, (hipStream_t)stream)); return AIX_OK; }
    HIPCHK(positions_indices(h->dev(), d_indices_out, (hipStream_t)stream));
    return AIX_OK;
}

extern "C" int aix_positions_bucket_counts_dev(aix_index_t* h, const char* d_reads, uint64_t len, uint64_t start, uint64_t* d_counts_out, void* stream) {
    if (!h || !d_counts_out || (len && !d_reads)) return AIX_ERR_ARG;
    DevGuard g(h->device);
    if (h->n == 0) return AIX_OK;
    HIPCHK(hipMemsetAsync(d_counts_out, 0, 8 * h->n, (hipStream_t)stream));
    HIPCHK(positions_bucket_counts(h->dev_slots(), (const uint8_t*)d_reads, len, start, (unsigned long long*)d_counts_out, (hipStream_t)stream));
    return AIX_OK;
}

extern "C" int aix_positions_fill_shard_dev(aix_index_t* h, const char* d_reads, uint64_t len, uint64_t start, uint64_t base_offset, const uint32_t* d_filled_init,
                                            const uint64_t* d_indices, uint64_t* d_positions, void* stream) {
    if (!h || !d_indices || !d_positions || (len && !d_reads)) return AIX_ERR_ARG;
    DevGuard g(h->device);
    if (h->n == 0) return AIX_OK;
    uint64_t piece = 0;
    if (const char* e = getenv("AIX_POSITIONS_PIECE")) piece = strtoull(e, nullptr, 10);
    POSCHK(positions_fill(h->dev_slots(), (const uint8_t*)d_reads, len, start, d_indices, d_positions, piece, d_filled_init, base_offset, (hipStream_t)stream, &h->a2_backend));
    return AIX_OK;
}

// ---------------------------------------------------------------------------------------------
// K1: distinct canonical k-mers of a sequence file (kmer_counter replacement)
// ---------------------------------------------------------------------------------------------
// device-resident twin: the PLAIN buffer is already in HBM; the result stays in HBM inside an opaque object until the caller
// has copied it into arrays of its own (two steps because the number of distinct k-mers is only known afterwards)
struct aix_distinct { uint64_t* keys = nullptr; uint64_t* counts = nullptr; uint64_t n = 0; int device = 0; };

extern "C" int aix_count_distinct_dev(const char* d_plain, uint64_t len, int k, int canon_mode, uint64_t min_count, int device, void* stream,
                                      aix_distinct_t** out) {
    if (!out || (len && !d_plain) || k < 1 || k > 31 || canon_mode < 0 || canon_mode > 2) return AIX_ERR_ARG;
    *out = nullptr;
    int st = check_device(device);
    if (st) return st;
    DevGuard g(device);
    aix_distinct* r = new (std::nothrow) aix_distinct();
    if (!r) return AIX_ERR_NOMEM;
    r->device = device;
    uint64_t piece = 0;
    if (const char* e = getenv("AIX_DISTINCT_PIECE")) piece = strtoull(e, nullptr, 10);
    hipError_t e = distinct_from_plain((const uint8_t*)d_plain, len, k, canon_mode, min_count ? min_count : 1, piece, &r->keys, &r->counts, &r->n, (hipStream_t)stream);
    if (e != hipSuccess) { delete r; set_last_error(std::string("count_distinct: ") + hipGetErrorString(e)); return AIX_ERR_HIP; }
    *out = r;
    return AIX_OK;
}
// K1 across GPUs: after the exchange a rank holds (key, count) pairs from every rank; equal keys are summed here
extern "C" int aix_merge_counts_dev(const uint64_t* d_keys, const uint64_t* d_counts, uint64_t n, uint64_t min_count, int device, void* stream,
                                    aix_distinct_t** out) {
    if (!out || (n && (!d_keys || !d_counts))) return AIX_ERR_ARG;
    *out = nullptr;
    int st = check_device(device);
    if (st) return st;
    DevGuard g(device);
    aix_distinct* r = new (std::nothrow) aix_distinct();
    if (!r) return AIX_ERR_NOMEM;
    r->device = device;
    hipError_t e = merge_counts(d_keys, d_counts, n, min_count ? min_count : 1, &r->keys, &r->counts, &r->n, (hipStream_t)stream);
    if (e != hipSuccess) { delete r; set_last_error(std::string("merge_counts: ") + hipGetErrorString(e)); return AIX_ERR_HIP; }
    *out = r;
    return AIX_OK;
}
// the same when the caller knows its runs: run r = entries [run_offsets[r], run_offsets[r + 1]), each sorted by key and free of repeats
// (after the K1 exchange a rank holds one such run per peer): a tree of two-way merges, nothing is sorted
extern "C" int aix_merge_runs_dev(const uint64_t* d_keys, const uint64_t* d_counts, const uint64_t* run_offsets, uint32_t nruns, uint64_t min_count, int device,
                                  void* stream, aix_distinct_t** out) {
    if (!out || !run_offsets || nruns == 0) return AIX_ERR_ARG;
    *out = nullptr;
    for (uint32_t r = 0; r < nruns; ++r) if (run_offsets[r + 1] < run_offsets[r]) return AIX_ERR_ARG;
    if (run_offsets[nruns] > run_offsets[0] && (!d_keys || !d_counts)) return AIX_ERR_ARG;
    int st = check_device(device);
    if (st) return st;
    DevGuard g(device);
    aix_distinct* r = new (std::nothrow) aix_distinct();
    if (!r) return AIX_ERR_NOMEM;
    r->device = device;
    hipError_t e = merge_runs(d_keys, d_counts, run_offsets, nruns, min_count ? min_count : 1, &r->keys, &r->counts, &r->n, (hipStream_t)stream);
    if (e != hipSuccess) { delete r; set_last_error(std::string("merge_runs: ") + hipGetErrorString(e)); return AIX_ERR_HIP; }
    *out = r;
    return AIX_OK;
}
extern "C" int aix_distinct_size(const aix_distinct_t* r, uint64_t* n_out) {
    if (!r || !n_out) return AIX_ERR_ARG;
    *n_out = r->n;
    return AIX_OK;
}
extern "C" int aix_distinct_copy_dev(const aix_distinct_t* r, uint64_t* d_keys, uint64_t* d_counts, void* stream) {
    if (!r || (r->n && (!d_keys || !d_counts))) return AIX_ERR_ARG;
    DevGuard g(r->device);
    if (r->n) {
        HIPCHK(hipMemcpyAsync(d_keys, r->keys, 8 * r->n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        HIPCHK(hipMemcpyAsync(d_counts, r->counts, 8 * r->n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    }
    return AIX_OK;
}
extern "C" void aix_distinct_free(aix_distinct_t* r) {
    if (!r) return;
    DevGuard g(r->device);
    if (r->keys) pool_free(r->keys);
    if (r->counts) pool_free(r->counts);
    delete r;
}
