// aix_kernels.hip — hand-written HIP kernels for gfx950 (MI355X): batch MPHF lookups (23-/13-mer),
// per-position coverage, k-mer counting, index re-layout, synthetic generators.
//
// All kernels are HBM/latency-bound integer work (no MFMA): one query / window per lane, three
// independent 16-byte MPHF record reads + one 16-byte key record read per probe, grid-stride
// launches of >= 8 workgroups per CU so that ~2k probes per CU are in flight.
#include "aix_internal.hpp"

namespace aix {

static constexpr int kBlock = 256;
static inline unsigned grid_for(uint64_t work, unsigned per_block = kBlock) {
    // Grid-stride kernels. Up to 8192 workgroups one trip each; beyond that a quarter of the trips' worth of workgroups — a wave keeps at
    // least four trips, which is what the lookups' FilterGauge needs to adapt — up to 256 per CU: finer workgroups balance the tail better
    // than 32 per CU did (100 M lookups 2.116 against 2.16 ms, coverage 29.9 against 31.4 ms, count23 38.3 against 38.7 ms, one box).
    static const uint64_t per_cu = [] { const char* e = getenv("AIX_GRID_PER_CU"); const long v = e ? atol(e) : 0; return (uint64_t)(v >= 1 && v <= 1024 ? v : 256); }();   // A/B switch
    const uint64_t need = (work + per_block - 1) / per_block, cap = 256ull * per_cu;
    uint64_t b = std::max(std::min<uint64_t>(need, 8192), need / 4);
    if (b > cap) b = cap;
    if (b == 0) b = 1;
    return (unsigned)b;
}

// ---------------------------------------------------------------------------------------------
// probes
// ---------------------------------------------------------------------------------------------
struct Probe {
    uint64_t slot;
    uint32_t tf;
    uint32_t lines;     // instrumentation for MODE_LINES: MPHF records read | 16 per key record read | 256 per completed evaluation | 4096 per filter word | 65536 per bucket line
    bool found;
};
// MPHF path: evaluate the MPHF on the hash (a, b, c) of the probed bytes, verify against the stored code.
// `filters`: the hashed bytes are exactly the ASCII of `code`. Only then do a stored key's fingerprint / presence
// bits (computed from ITS ASCII) say anything about this probe; the reference's forward probe of a query with
// non-ACGT bytes hashes the raw bytes but compares the sanitised code (python_wrapper.cpp:611-613) and can — by a
// 1-in-n coincidence of slots — match a stored key whose hash is different, so that probe runs unfiltered.
__device__ __forceinline__ Probe probe23_mphf(const IndexDev& ix, uint64_t a, uint64_t b, uint64_t c, uint64_t code, bool filters) {
    Probe r;
    r.found = false;
    r.tf = 0;
    r.slot = 0;
    r.lines = 3;
    if (!filters) {
        r.lines += 256;
        r.slot = mphf_from_hash(ix.m, a, b, c);
    } else if (ix.early_exit) {
        if (!mphf_probe_early_exit(ix.m, a, b, c, r.slot, r.lines)) return r;   // a node lacks one of the key's presence bits
        r.lines += 256;
    } else if (ix.use_fp) {
        r.lines += 256;
        uint32_t fps;
        uint64_t node;
        r.slot = mphf_from_hash_fp(ix.m, a, b, c, fps, node);
        if (fps != fp_of_hash(a, b, c)) return r;  // the key assigned to this node (if any) is a different key
    } else {
        r.lines += 256;
        r.slot = mphf_from_hash(ix.m, a, b, c);
    }
    if (r.slot < ix.n) {                           // python_wrapper.cpp:613 `h1 >= n ||`
        const KeyRec k = key_at(ix, r.slot);
        r.lines += 16;
        if (k.code == code) { r.found = true; r.tf = k.tf; }
    }
    return r;
}
// one lane on its own (ragged lengths, index construction): hash the 23 ASCII bytes in (w0,w1,w2), MPHF path
__device__ __forceinline__ Probe probe23(const IndexDev& ix, uint64_t w0, uint64_t w1, uint64_t w2, uint64_t code, bool filters = true) {
    uint64_t a, b, c;
    jenkins23(w0, w1, w2, ix.m.seed, a, b, c);
    return probe23_mphf(ix, a, b, c, code, filters);
}
// same, but the forward hash (a,b,c) was computed elsewhere (ragged lengths)
__device__ __forceinline__ Probe probe23_hashed(const IndexDev& ix, uint64_t a, uint64_t b, uint64_t c, uint64_t code) {
    return probe23_mphf(ix, a, b, c, code, false);
}

// The probe of the batch kernels. EVERY lane of the wave calls it (want = false: this lane has nothing to probe); with the
// verification table on, a probe whose hashed bytes are the ASCII of `code` (filters) is answered from its bucket line, all
// others — and an unmatched probe of an overflowed bucket — by the MPHF path.
// `absence`: consult the absence filter first (wave-uniform; the callers switch it per loop trip, see FilterGauge).
template <int LPP>
__device__ __forceinline__ Probe probe23_wave(const IndexDev& ix, bool want, uint64_t w0, uint64_t w1, uint64_t w2, uint64_t code, bool filters = true,
                                              bool absence = true) {
    Probe r;
    r.found = false; r.tf = 0; r.slot = 0; r.lines = 0;
    const bool rest = want;
    uint64_t a = 0, b = 0, c = 0;
    if (rest) jenkins23(w0, w1, w2, ix.m.seed, a, b, c);
    bool mphf = rest;
    if (ix.bk) {
        bool use = rest && filters;
        if (ix.bloom && absence && use) {                       // absent from the filter = not a filed key; an unfiled key (overflow) is in it too
            r.lines += 4096;
            const uint64_t m = bloom_mask(c);
            if ((ix.bloom[bloom_word(b, ix.nbloom)] & m) != m) { use = false; mphf = false; }
        }
        const BkRes k = bucket_probe_wave<LPP>(ix.bk, ix.nb, use, a, code);
        if (use) {
            r.lines += 65536;
            if (k.found) { r.found = true; r.tf = k.tf; r.slot = k.slot; }
            mphf = !k.found && k.overflow;
        }
    }
    if (mphf) {
        const Probe q = probe23_mphf(ix, a, b, c, code, filters);
        r.found = q.found; r.tf = q.tf; r.slot = q.slot; r.lines += q.lines;
    }
    return r;
}

// The absence filter pays when most probes are absent keys (one cached 8-byte read instead of a 128-byte line from HBM) and
// costs when most are present (one more read each). A wave decides trip by trip from what it has just seen: the filter is
// consulted in the next trip iff fewer than a quarter of this trip's queries were found. The answers do not depend on it.
struct FilterGauge {
    bool on = true;
    __device__ __forceinline__ void seen(bool active, bool found) {
        const uint32_t a = (uint32_t)__popcll(__ballot(active)), f = (uint32_t)__popcll(__ballot(active && found));
        if (a) on = 4u * f < a;
    }
};

// Result of get_tf_value_23mer-style probing (python_wrapper.cpp:610-627): strand 0 absent, 1 fwd, 2 rc
struct Q23 {
    uint64_t slot;
    uint32_t tf;
    uint32_t strand;
    uint32_t lines;
};
// wave-cooperative (all lanes call it; `active` = this lane holds a query)
// `reload(a, b, c)` fetches the query's 23 bytes again: the rare lane-by-lane path of a canonical index takes them from memory a second time instead
// of holding the original words, their code and its reverse complement (10 VGPRs) across the wave's probe for a case that almost never comes
template <bool CANON, int LPP, class RELOAD>
__device__ __forceinline__ Q23 query23(const IndexDev& ix, bool active, uint64_t w0, uint64_t w1, uint64_t w2, FilterGauge& fg, RELOAD&& reload) {
    const Enc23 e = encode23_words(w0, w1, w2);
    const uint64_t r = revcomp(e.code, 23);
    Q23 out;
    out.slot = 0; out.tf = 0; out.strand = 0; out.lines = 0;
    if (CANON) {
        // every stored code is canonical: only the canonical strand of a pure-ACGT query can match. ONE probe call site: with the
        // strands chosen per lane first, a wave runs Jenkins + the probe once, not once per strand with half its lanes idle
        const bool fwd = e.code <= r;
        const bool clean = active && e.valid, other = active && !e.valid;
        uint64_t x0 = w0, x1 = w1, x2 = w2;
        if (!fwd) ascii23_of_rc(e.code, x0, x1, x2);
        const Probe p = probe23_wave<LPP>(ix, clean, x0, x1, x2, fwd ? e.code : r, true, fg.on);
        fg.seen(clean, p.found);
        if (clean) {
            out.lines = p.lines;
            if (p.found) { out.slot = p.slot; out.tf = p.tf; out.strand = fwd ? 1u : 2u; }
        }
        if (other) {                                            // other bytes: the reference's two probes, lane by lane (rare)
            uint64_t v0, v1, v2;
            reload(v0, v1, v2);
            const Enc23 e2 = encode23_words(v0, v1, v2);
            uint64_t a, b, c;
            jenkins23(v0, v1, v2, ix.m.seed, a, b, c);
            const Probe f = probe23_mphf(ix, a, b, c, e2.code, false);   // raw bytes hashed, sanitised code compared
            out.lines = f.lines;
            if (f.found) { out.slot = f.slot; out.tf = f.tf; out.strand = 1; }
            else {
                uint64_t r0, r1, r2;
                ascii23_of_rc(e2.code, r0, r1, r2);             // decode(reverseDNA(u)), :615-616
                const Probe g = probe23(ix, r0, r1, r2, revcomp(e2.code, 23));
                out.lines += g.lines;
                if (g.found) { out.slot = g.slot; out.tf = g.tf; out.strand = 2; }
            }
        }
        return out;
    }
    const Probe f = probe23_wave<LPP>(ix, active, w0, w1, w2, e.code, e.valid, fg.on);   // raw bytes hashed, sanitised code compared
    uint64_t r0, r1, r2;
    ascii23_of_rc(e.code, r0, r1, r2);                          // decode(reverseDNA(u)), :615-616
    const Probe g = probe23_wave<LPP>(ix, active && !f.found, r0, r1, r2, r, true, fg.on);
    fg.seen(active, f.found || g.found);
    if (active) {
        out.lines = f.lines;
        if (f.found) { out.slot = f.slot; out.tf = f.tf; out.strand = 1; }
        else {
            out.lines += g.lines;
            if (g.found) { out.slot = g.slot; out.tf = g.tf; out.strand = 2; }
        }
    }
    return out;
}
template <bool CANON, int LPP>
__device__ __forceinline__ Q23 query23(const IndexDev& ix, bool active, uint64_t w0, uint64_t w1, uint64_t w2, FilterGauge& fg) {
    return query23<CANON, LPP>(ix, active, w0, w1, w2, fg, [&](uint64_t& a, uint64_t& b, uint64_t& c) { a = w0; b = w1; c = w2; });
}

// get_tf_both_directions_23mer (python_wrapper.cpp:1259-1275): Q1(q) and Q1(decode(rc(q))); wave-cooperative like query23
template <bool CANON, int LPP>
__device__ __forceinline__ void both23(const IndexDev& ix, bool active, uint64_t w0, uint64_t w1, uint64_t w2, uint32_t& fwd, uint32_t& rc, FilterGauge& fg) {
    const Enc23 e = encode23_words(w0, w1, w2);
    const uint64_t r = revcomp(e.code, 23);
    fwd = 0; rc = 0;
    const bool fast = CANON && e.valid;                         // canonical index, pure-ACGT query: both directions see the one canonical key
    const Q23 q = query23<CANON, LPP>(ix, active && fast, w0, w1, w2, fg);
    if (active && fast) { fwd = q.tf; rc = q.tf; }
    const bool slow = active && !fast;
    if (CANON) {
        if (slow) {                                             // non-ACGT bytes on a canonical index: lane by lane, MPHF path
            uint64_t r0, r1, r2, s0, s1, s2;
            ascii23_of_rc(e.code, r0, r1, r2);
            ascii23_of_rc(r, s0, s1, s2);
            uint64_t a, b, c;
            jenkins23(w0, w1, w2, ix.m.seed, a, b, c);
            const Probe F = probe23_mphf(ix, a, b, c, e.code, false);
            const Probe R = probe23(ix, r0, r1, r2, r);
            const Probe S = probe23(ix, s0, s1, s2, e.code);    // second call's fallback decodes the sanitised code
            fwd = F.found ? F.tf : (R.found ? R.tf : 0u);
            rc = R.found ? R.tf : (S.found ? S.tf : 0u);
        }
        return;
    }
    uint64_t r0, r1, r2, s0, s1, s2;
    ascii23_of_rc(e.code, r0, r1, r2);
    ascii23_of_rc(r, s0, s1, s2);
    const Probe F = probe23_wave<LPP>(ix, slow, w0, w1, w2, e.code, e.valid, fg.on);
    const Probe R = probe23_wave<LPP>(ix, slow, r0, r1, r2, r, true, fg.on);
    const Probe S2 = probe23_wave<LPP>(ix, slow && !e.valid, s0, s1, s2, e.code, true, fg.on);
    fg.seen(slow, F.found || R.found);
    if (slow) {
        const Probe S = e.valid ? F : S2;
        fwd = F.found ? F.tf : (R.found ? R.tf : 0u);
        rc = R.found ? R.tf : (S.found ? S.tf : 0u);
    }
}

// wave-uniform grid-stride loops: every lane of a wave runs the same number of trips (the probes are wave-cooperative)
#define AIX_WAVE_LOOP(i, N)                                                                                    \
    for (uint64_t i##_base = (uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u), i = i##_base + (threadIdx.x & 63u); i##_base < (N); \
         i##_base += (uint64_t)gridDim.x * kBlock, i += (uint64_t)gridDim.x * kBlock)

template <int MODE, bool CANON, int LPP>
__global__ void __launch_bounds__(kBlock) k_lookup23_ascii(const IndexDev ix_, const uint8_t* __restrict__ q, uint64_t N, LookupOut out) {
    const IndexDev& ix = ix_;
    FilterGauge fg;
    AIX_WAVE_LOOP(i, N) {
        const bool in = i < N;
        uint64_t w0 = 0, w1 = 0, w2 = 0;
        if (in) load23(q + 23 * i, w0, w1, w2);
        auto again = [&](uint64_t& a, uint64_t& b, uint64_t& c) { load23(q + 23 * i, a, b, c); };
        if (MODE == MODE_TF) {
            const uint32_t v = query23<CANON, LPP>(ix, in, w0, w1, w2, fg, again).tf;
            if (in) out.tf[i] = v;
        } else if (MODE == MODE_LINES) {
            const uint32_t v = query23<CANON, LPP>(ix, in, w0, w1, w2, fg).lines;   // instrumentation: records read for this query
            if (in) out.tf[i] = v;
        } else if (MODE == MODE_HASH) {
            if (in) {
                uint64_t a, b, c;
                jenkins23(w0, w1, w2, ix.m.seed, a, b, c);
                out.u64a[i] = mphf_from_hash(ix.m, a, b, c);
            }
        } else if (MODE == MODE_KIDSTRAND) {
            const Q23 r = query23<CANON, LPP>(ix, in, w0, w1, w2, fg, again);
            if (in) {
                if (out.u64a) out.u64a[i] = r.slot;             // get_kid_by_kmer: 0 when absent (:700-716)
                if (out.strand) out.strand[i] = (uint8_t)r.strand;
            }
        } else {
            uint32_t f, r;
            both23<CANON, LPP>(ix, in, w0, w1, w2, f, r, fg);
            if (in) {
                if (MODE == MODE_TOTAL) {
                    out.u64a[i] = (uint64_t)f + (uint64_t)r;
                } else {
                    if (out.u64a) out.u64a[i] = f;
                    if (out.u64b) out.u64b[i] = r;
                }
            }
        }
    }
}

template <bool CANON, int LPP>
__global__ void __launch_bounds__(kBlock) k_lookup23_codes(const IndexDev ix_, const uint64_t* __restrict__ codes, uint64_t N, uint32_t* __restrict__ out) {
    const IndexDev& ix = ix_;
    FilterGauge fg;
    AIX_WAVE_LOOP(i, N) {
        const bool in = i < N;
        const uint64_t u = in ? (codes[i] & ((1ULL << 46) - 1)) : 0ull;
        const uint64_t r = revcomp(u, 23);
        uint64_t w0, w1, w2;
        uint32_t tf = 0;
        if (CANON) {
            const uint64_t key = u <= r ? u : r;
            ascii23_of_rc(u <= r ? r : u, w0, w1, w2);           // string of `key`
            const Probe p = probe23_wave<LPP>(ix, in, w0, w1, w2, key, true, fg.on);
            fg.seen(in, p.found);
            tf = p.found ? p.tf : 0u;
        } else {
            ascii23_of_rc(r, w0, w1, w2);
            const Probe f = probe23_wave<LPP>(ix, in, w0, w1, w2, u, true, fg.on);
            ascii23_of_rc(u, w0, w1, w2);
            const Probe g = probe23_wave<LPP>(ix, in && !f.found, w0, w1, w2, r, true, fg.on);
            fg.seen(in, f.found || g.found);
            tf = f.found ? f.tf : (g.found ? g.tf : 0u);
        }
        if (in) out[i] = tf;
    }
}

// variable-length queries: the reference hashes ALL bytes of the std::string but encodes the first 23
__global__ void __launch_bounds__(kBlock) k_lookup23_ragged(const IndexDev ix, const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offs,
                                                           uint64_t N, uint32_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += stride) {
        const uint64_t lo = offs[i], len = offs[i + 1] - lo;
        uint32_t tf = 0;
        if (len >= 23) {
            uint64_t w0, w1, w2;
            load23(bytes + lo, w0, w1, w2);
            if (len == 23) {                                     // lane by lane (lengths differ inside a wave): the MPHF path, both probes
                const Enc23 e = encode23_words(w0, w1, w2);
                const Probe f = probe23(ix, w0, w1, w2, e.code, e.valid);
                if (f.found) tf = f.tf;
                else {
                    uint64_t r0, r1, r2;
                    ascii23_of_rc(e.code, r0, r1, r2);
                    const Probe g = probe23(ix, r0, r1, r2, revcomp(e.code, 23));
                    tf = g.found ? g.tf : 0u;
                }
            } else {
                const Enc23 e = encode23_words(w0, w1, w2);
                uint64_t a, b, c;
                jenkins_bytes(bytes + lo, len, ix.m.seed, a, b, c);
                const Probe f = probe23_hashed(ix, a, b, c, e.code);
                if (f.found) tf = f.tf;
                else {
                    uint64_t r0, r1, r2;
                    ascii23_of_rc(e.code, r0, r1, r2);
                    const Probe g = probe23(ix, r0, r1, r2, revcomp(e.code, 23));
                    tf = g.found ? g.tf : 0u;
                }
            }
        }
        out[i] = tf;
    }
}

// ---------------------------------------------------------------------------------------------
// 13-mer mode: the MPHF over all 4^13 13-mers is a bijection code -> slot, so the table is kept in
// 2-bit-code order in HBM (tf13_code) and a valid query is one 8-byte read. Queries with bytes
// outside ACGT take the reference's raw-bytes MPHF path against the mphf-ordered copy.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t tf13_raw(const IndexDev& ix, uint64_t w0, uint64_t w1) {
    uint64_t a, b, c;
    jenkins13(w0, w1, ix.m.seed, a, b, c);
    const uint64_t h = mphf_from_hash(ix.m, a, b, c);
    return h < 67108864ull ? ix.tf13_mphf[h] : 0ull;            // reference indexes unguarded (:534); OOB -> 0
}
// get_reverse_complement_13mer (python_wrapper.cpp:505-517) on raw bytes: reverse; A<->T, C<->G; others kept
__device__ __forceinline__ void rc13_raw(uint64_t w0, uint64_t w1, uint64_t& r0, uint64_t& r1) {
    r0 = 0; r1 = 0;
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int j = 12 - i;
        uint32_t ch = (uint32_t)((j < 8 ? (w0 >> (8 * j)) : (w1 >> (8 * (j - 8)))) & 0xff);
        ch = ch == 'A' ? 'T' : ch == 'T' ? 'A' : ch == 'G' ? 'C' : ch == 'C' ? 'G' : ch;
        if (i < 8) r0 |= (uint64_t)ch << (8 * i);
        else r1 |= (uint64_t)ch << (8 * (i - 8));
    }
}

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_lookup13_ascii(const IndexDev ix, const uint8_t* __restrict__ q, uint64_t N, LookupOut out) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += stride) {
        uint64_t w0, w1;
        load13(q + 13 * i, w0, w1);
        const Enc13 e = encode13_words(w0, w1);
        if (MODE == MODE_TF) {                                  // get_tf_values_13mer :938-980: strict, u32 truncation
            out.tf[i] = e.valid ? (uint32_t)ix.tf13_code[e.code] : 0u;
        } else if (MODE == MODE_HASH) {                         // hasher_13mer.lookup(kmer) (:1087): the tabulated slot, or the MPHF of the raw bytes
            uint64_t hval;
            if (e.valid) hval = ix.perm13[e.code];
            else { uint64_t a, b, c; jenkins13(w0, w1, ix.m.seed, a, b, c); hval = mphf_from_hash(ix.m, a, b, c); }
            out.u64a[i] = hval;
        } else {
            uint64_t f, r;
            if (e.valid) {
                f = ix.tf13_code[e.code];
                r = ix.tf13_code[(uint32_t)revcomp(e.code, 13)];
            } else {                                            // :522-543 has no validation: raw bytes are hashed
                uint64_t r0, r1;
                rc13_raw(w0, w1, r0, r1);
                f = tf13_raw(ix, w0, w1);
                r = tf13_raw(ix, r0, r1);
            }
            if (MODE == MODE_TOTAL) out.u64a[i] = f + r;
            else {
                if (out.u64a) out.u64a[i] = f;
                if (out.u64b) out.u64b[i] = r;
            }
        }
    }
}

__global__ void __launch_bounds__(kBlock) k_lookup13_ragged(const IndexDev ix, const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ offs,
                                                           uint64_t N, uint32_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += stride) {
        const uint64_t lo = offs[i], len = offs[i + 1] - lo;
        uint32_t tf = 0;
        if (len == 13) {                                        // is_13mer(): length check first (:943)
            uint64_t w0, w1;
            load13(bytes + lo, w0, w1);
            const Enc13 e = encode13_words(w0, w1);
            if (e.valid) tf = (uint32_t)ix.tf13_code[e.code];
        }
        out[i] = tf;
    }
}

// ---------------------------------------------------------------------------------------------
// coverage: one lane per byte position of the concatenated sequences (aindex.py:314-322)
// ---------------------------------------------------------------------------------------------
// (eight waves per SIMD: 64 VGPRs with two spilled words instead of 69 and seven waves — 268-270 against 280-282 ms per 10^10 positions on one box;
// the same limit on k_lookup23_ascii costs 5-7 %, so it is set here only)
template <bool CANON, int LPP>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) k_coverage(const IndexDev ix_, const uint8_t* __restrict__ seqs, const uint64_t* __restrict__ offs, uint64_t M,
                                                    uint64_t total, double seqs_per_byte, uint32_t cutoff, uint32_t* __restrict__ out,
                                                    const uint64_t* __restrict__ out_offs) {
    const IndexDev& ix = ix_;
    const uint32_t k = ix.k;
    FilterGauge fg;
    AIX_WAVE_LOOP(p, total) {
        // sequence containing byte p: largest s with offs[s] <= p. The lanes of a wave hold consecutive p, so the binary search
        // runs once per wave for its first position (wave-uniform: scalar loads), and a lane then walks forward from there —
        // zero or one step unless the sequences are shorter than a wave is wide (then a bounded walk, then its own search).
        // (the builtin returns int: without the cast a low half with bit 31 set sign-extends over the high half, the search lands on the last
        // sequence and the positions [2^31, 2^32) mod 2^32 of a batch come back 0 — rounds 1 and 2 did that to 43 % of config 5 at full size)
        const uint64_t p0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(p_base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)p_base);
        // the window's bytes are requested before anything is known about the sequence they belong to (a window inside the buffer can be read
        // whether or not it lies inside ONE sequence): the loads fly while the search below waits for its offsets
        const bool inb = p + k <= total;
        uint64_t w0 = 0, w1 = 0, w2 = 0;
        if (inb) { if (k == 23) load23(seqs + p, w0, w1, w2); else load13(seqs + p, w0, w1); }     // (k = 13, same box: 185 against 198 ms per 10^10 positions; k = 23 within noise)
        // The search starts from where p0 would lie if all sequences had the mean length — exact for equal lengths, the usual batch — and
        // gallops from there: two independent loads instead of log2(M) dependent ones per trip of every wave (20 round trips at 10^6 sequences,
        // several times the latency of the probe itself). Any guess gives the same s.
        uint64_t lo, hi;                                        // invariant offs[lo] <= p0 < offs[hi] (or lo == 0)
        {
            uint64_t g = (uint64_t)((double)p0 * seqs_per_byte);
            if (g >= M) g = M - 1;
            const uint64_t og = offs[g], og1 = offs[g + 1];
            if (og <= p0) {
                lo = g; hi = g + 1;
                if (og1 <= p0) {
                    uint64_t step = 1;
                    while (hi < M && offs[hi] <= p0) { lo = hi; step <<= 1; hi = (M - lo > step) ? lo + step : M; }
                }
            } else {
                hi = g; lo = g ? g - 1 : 0;
                uint64_t step = 1;
                while (lo > 0 && offs[lo] > p0) { hi = lo; step <<= 1; lo = lo > step ? lo - step : 0; }
            }
        }
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (offs[mid] <= p0) lo = mid; else hi = mid;
        }
        const uint64_t begin0 = offs[lo], end0 = offs[lo + 1];
        // one probe per lane either way; the two cases are kept apart so that the usual one carries nothing per lane across the probe but the
        // window words (its output address is a wave-uniform base plus the lane number)
        auto answer = [&](bool active) -> uint32_t {
            if (k == 23) return query23<CANON, LPP>(ix, active, w0, w1, w2, fg, [&](uint64_t& a, uint64_t& b, uint64_t& c) { load23(seqs + p, a, b, c); }).tf;
            if (!active) return 0u;
            const Enc13 e = encode13_words(w0, w1);
            return e.valid ? (uint32_t)ix.tf13_code[e.code] : 0u;
        };
        if (begin0 <= p0 && end0 > p0 + 63) {                   // all 64 positions of the wave lie in sequence lo (the usual case): nothing is looked up per lane
            uint32_t* const wdst = out + (out_offs[lo] + (p0 - begin0));      // wave-uniform
            const bool active = inb && p + k <= end0;
            const uint32_t tf = answer(active);
            if (active) wdst[threadIdx.x & 63u] = tf >= cutoff ? tf : 0u;
        } else {
            bool active = inb;
            uint64_t s = lo;
            uint32_t* dst = out;
            if (active) {
                for (int step = 0; step < 4 && s + 1 < M && offs[s + 1] <= p; ++step) ++s;
                if (s + 1 < M && offs[s + 1] <= p) {
                    uint64_t l2 = s + 1, h2 = M;                // offs[l2] <= p < offs[h2]
                    while (h2 - l2 > 1) {
                        const uint64_t mid = (l2 + h2) >> 1;
                        if (offs[mid] <= p) l2 = mid; else h2 = mid;
                    }
                    s = l2;
                }
                const uint64_t begin = offs[s], end = offs[s + 1];
                if (p < begin || p + k > end) active = false;   // p < begin: gaps between sequences are allowed
                dst = out + (out_offs[s] + (p - begin));
            }
            const uint32_t tf = answer(active);
            if (active) *dst = tf >= cutoff ? tf : 0u;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// index re-layout
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_build_keyrecs(const uint64_t* __restrict__ checker, const uint32_t* __restrict__ tf, uint64_t n, KeyRec* __restrict__ recs,
                                                         uint32_t* __restrict__ noncanon) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint32_t bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        KeyRec r;
        r.code = checker[i];
        r.tf = tf[i];
        r.pad = 0;
        recs[i] = r;
        bad |= (r.code > revcomp(r.code & ((1ULL << 46) - 1), 23)) || (r.code >> 46);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicAdd(noncanon, 1u);
}
__global__ void __launch_bounds__(kBlock) k_extract(const IndexDev ix, uint32_t* __restrict__ tf, uint64_t* __restrict__ checker) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < ix.n; i += stride) {
        const KeyRec r = key_at(ix, i);
        if (tf) tf[i] = r.tf;
        if (checker) checker[i] = r.code;
    }
}
// Fingerprints: for every stored code that sits in its own MPHF slot, write its 4-bit fingerprint into the nibble
// of its assigned node. Entries that are not where the MPHF puts them (corrupt / foreign index) are skipped, so a
// fingerprint mismatch always implies checker[rank] != query.
__global__ void __launch_bounds__(kBlock) k_init_ee(const BvRec* __restrict__ recs, uint64_t nrecs, EeRec* __restrict__ ee) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nrecs; i += stride) {
        EeRec e;
        e.pairs = recs[i].pairs;
        e.prefix = recs[i].prefix;
#pragma unroll
        for (int k = 0; k < 6; ++k) e.mask[k] = 0;
        ee[i] = e;
    }
}
__global__ void __launch_bounds__(kBlock) k_set_fp(const IndexDev ix, BvRec* __restrict__ recs, EeRec* __restrict__ ee, int do_fp) {
    const MphfDev& m = ix.m;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < ix.n; i += stride) {
        const uint64_t code = key_at(ix, i).code;
        if (code >> 46) continue;
        uint64_t w0, w1, w2, a, b, c, node;
        uint32_t fps;
        ascii23_of_rc(revcomp(code, 23), w0, w1, w2);
        jenkins23(w0, w1, w2, m.seed, a, b, c);
        if (mphf_from_hash_fp(m, a, b, c, fps, node) != i) continue;
        if (do_fp) atomicOr((unsigned long long*)&recs[node >> 4].fp, (unsigned long long)fp_of_hash(a, b, c) << (4 * (uint32_t)(node & 15)));
        if (!ee) continue;
        const uint64_t nd[3] = {fastmod(a, m.fm), m.D + fastmod(b, m.fm), 2 * m.D + fastmod(c, m.fm)};
#pragma unroll
        for (int t = 0; t < 3; ++t) {                     // the key's two presence bits at each of its three nodes
            const uint32_t bit = (uint32_t)(nd[t] & 15) * 12u, sh = bit & 31u;
            const uint32_t q = present_mask(a, b, c, t);
            uint32_t* w = ee[nd[t] >> 4].mask + (bit >> 5);
            atomicOr(w, q << sh);
            if (sh > 20u) atomicOr(w + 1, q >> (32u - sh));
        }
    }
}

// Verification table (aix_device.hpp: BkEntry). Filed: every stored code that sits in its own MPHF slot, under the bucket its
// Jenkins hash selects; the first eight arrivals of a bucket get an entry, the rest are left to the MPHF path (pass 2 raises
// the bucket's overflow bit). Which eight is decided by the order of the atomics — the answers do not depend on it.
__global__ void __launch_bounds__(kBlock) k_bk_init(BkEntry* __restrict__ bk, uint64_t nent) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nent; i += stride) {
        BkEntry e;
        e.code_lo = 0xFFFFFFFFu; e.code_hi = AIX_BK_EMPTY_HI; e.tf = 0; e.slot = 0;
        bk[i] = e;
    }
}
__global__ void __launch_bounds__(kBlock) k_bk_fill(const MphfDev m, const KeyRec* __restrict__ keys, uint64_t n, BkEntry* __restrict__ bk, uint32_t nb,
                                                   uint32_t* __restrict__ fill, uint64_t* __restrict__ bloom, uint32_t nbloom, uint32_t nbm, uint32_t* __restrict__ mfill,
                                                   uint32_t* __restrict__ side) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const KeyRec kr = keys[i];
        side[i] = 0xFFFFFFFFu;                                   // not filed until the entry has been written below
        if (kr.code >> 46) continue;
        uint64_t w0, w1, w2, a, b, c;
        ascii23_of_rc(revcomp(kr.code, 23), w0, w1, w2);
        jenkins23(w0, w1, w2, m.seed, a, b, c);
        if (mphf_from_hash(m, a, b, c) != i) continue;         // not where the MPHF puts it: the reference cannot find it, neither can a probe
        if (bloom) atomicOr((unsigned long long*)&bloom[bloom_word(b, nbloom)], (unsigned long long)bloom_mask(c));
        const uint32_t bi = bucket_of(a, nb);
        const uint32_t pos = atomicAdd(&fill[bi], 1u);
        BkEntry e;
        e.code_lo = (uint32_t)kr.code; e.code_hi = (uint32_t)(kr.code >> 32); e.tf = kr.tf; e.slot = (uint32_t)i;
        if (pos < 8u) { bk[(uint64_t)bi * 8 + pos] = e; side[i] = bi * 8u + pos; }      // (nb * 8 < 2^31: nb = n / load + 1 with n < 2^32 ... checked by the caller)
        if (nbm) atomicAdd(&mfill[mk_home(minimizer23(kr.code, revcomp(kr.code, 23)), nbm)], 1u);    // the minimizer-keyed copy: sized here, written by k_mk_fill
    }
}
// The same entries once more, grouped by the bucket of their minimizer: bucket b = mk[off[b], off[b + 1]) holds EVERY filed key whose
// minimizer hashes to b (the keys of one minimizer arrive together, ~8 at a time, so fixed-size buckets overflowed for 15-30 % of the
// windows; a bucket sized by its content never does). Same filing condition as k_bk_fill.
__global__ void __launch_bounds__(kBlock) k_mk_fill(const MphfDev m, const KeyRec* __restrict__ keys, uint64_t n, BkEntry* __restrict__ mk, const uint32_t* __restrict__ off,
                                                   uint32_t nbm, uint32_t* __restrict__ mcur) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const KeyRec kr = keys[i];
        if (kr.code >> 46) continue;
        uint64_t w0, w1, w2, a, b, c;
        ascii23_of_rc(revcomp(kr.code, 23), w0, w1, w2);
        jenkins23(w0, w1, w2, m.seed, a, b, c);
        if (mphf_from_hash(m, a, b, c) != i) continue;
        const uint32_t home = mk_home(minimizer23(kr.code, revcomp(kr.code, 23)), nbm);
        BkEntry e;
        e.code_lo = (uint32_t)kr.code; e.code_hi = (uint32_t)(kr.code >> 32); e.tf = kr.tf; e.slot = (uint32_t)i;
        mk[(uint64_t)off[home] + atomicAdd(&mcur[home], 1u)] = e;
    }
}
__global__ void __launch_bounds__(kBlock) k_side_count(const uint32_t* __restrict__ side, uint64_t n, uint32_t* __restrict__ counter) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint32_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) c += side[i] == 0xFFFFFFFFu ? 1u : 0u;
    if (c) atomicAdd(counter, c);
}
__global__ void __launch_bounds__(kBlock) k_side_unfiled(const KeyRec* __restrict__ keys, uint64_t n, uint32_t* __restrict__ side, KeyRec* __restrict__ unfiled,
                                                        uint32_t* __restrict__ counter) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        if (side[i] != 0xFFFFFFFFu) continue;
        const uint32_t idx = atomicAdd(counter, 1u);
        unfiled[idx] = keys[i];
        side[i] = AIX_SIDE_UNFILED | idx;
    }
}
__global__ void __launch_bounds__(kBlock) k_bk_flag(BkEntry* __restrict__ bk, uint32_t nb, const uint32_t* __restrict__ fill) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t bi = (uint64_t)blockIdx.x * kBlock + threadIdx.x; bi < nb; bi += stride)
        if (fill[bi] > 8u) bk[bi * 8 + 7].code_hi |= AIX_BK_OVERFLOW;
}

// I1 (hash.cpp:671-723): checker[h] = code, tf[h] = count with h = mphf(key). Keys arrive either as
// n x 23 ASCII bytes (the .dat lines) or as 2-bit codes. A slot hit twice raises `conflict`
// (the reference detects it only when the earlier tf was non-zero, then exit(12)).
template <bool ASCII>
__global__ void __launch_bounds__(kBlock) k_scatter23(const MphfDev m, uint64_t n, uint64_t nslots, const uint8_t* __restrict__ keys, const uint64_t* __restrict__ codes,
                                                     const uint32_t* __restrict__ counts, uint64_t* __restrict__ checker, uint32_t* __restrict__ tf,
                                                     uint32_t* __restrict__ occupied, uint32_t* __restrict__ conflict) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint64_t w0, w1, w2, code;
        if (ASCII) {
            load23(keys + 23 * i, w0, w1, w2);
            code = encode23_words(w0, w1, w2).code;                 // get_dna23_bitset(kmer), hash.cpp:715
        } else {
            code = codes[i] & ((1ULL << 46) - 1);
            ascii23_of_rc(revcomp(code, 23), w0, w1, w2);
        }
        uint64_t a, b, c;
        jenkins23(w0, w1, w2, m.seed, a, b, c);
        const uint64_t h = mphf_from_hash(m, a, b, c);
        if (h >= nslots) { atomicAdd(conflict, 1u); continue; }
        const uint32_t bit = 1u << (h & 31);
        if (atomicOr(&occupied[h >> 5], bit) & bit) { atomicAdd(conflict, 1u); continue; }
        checker[h] = code;
        tf[h] = counts ? counts[i] : 0u;
    }
}

// perm[code] = mphf13(ASCII(code)) for all 4^13 codes
__global__ void __launch_bounds__(kBlock) k_perm13(const MphfDev m, uint32_t* __restrict__ perm) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t code = (uint64_t)blockIdx.x * kBlock + threadIdx.x; code < 67108864ull; code += stride) {
        uint64_t w0, w1, a, b, c;
        ascii13_of_rc((uint32_t)revcomp(code, 13), w0, w1);
        jenkins13(w0, w1, m.seed, a, b, c);
        perm[code] = (uint32_t)mphf_from_hash(m, a, b, c);
    }
}
__global__ void __launch_bounds__(kBlock) k_tf13_to_code(const uint32_t* __restrict__ perm, const uint64_t* __restrict__ tf_mphf, uint64_t* __restrict__ tf_code) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t code = (uint64_t)blockIdx.x * kBlock + threadIdx.x; code < 67108864ull; code += stride) {
        const uint32_t h = perm[code];
        tf_code[code] = h < 67108864u ? tf_mphf[h] : 0ull;
    }
}
// ADD: the MPHF is not a bijection on the 13-mers (a foreign .pf): several codes may share a slot, and the reference's
// counts[mphf(window)].fetch_add(1) (count_kmers13.cpp:147-152) then sums them; slots >= 4^13 are skipped as there.
template <bool ADD>
__global__ void __launch_bounds__(kBlock) k_scatter13(const uint32_t* __restrict__ perm, const unsigned long long* __restrict__ table_code, uint64_t* __restrict__ out_mphf) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t code = (uint64_t)blockIdx.x * kBlock + threadIdx.x; code < 67108864ull; code += stride) {
        const uint32_t h = perm[code];
        if (h >= 67108864u) continue;
        if (ADD) { const unsigned long long c = table_code[code]; if (c) atomicAdd((unsigned long long*)&out_mphf[h], c); }
        else out_mphf[h] = table_code[code];
    }
}
// is code -> slot a bijection of [0, 4^13)? every slot in range and claimed once (bits: 4^13 / 32 zeroed words)
__global__ void __launch_bounds__(kBlock) k_perm13_check(const uint32_t* __restrict__ perm, uint32_t* __restrict__ bits, uint32_t* __restrict__ bad) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint32_t b = 0;
    for (uint64_t code = (uint64_t)blockIdx.x * kBlock + threadIdx.x; code < 67108864ull; code += stride) {
        const uint32_t h = perm[code];
        if (h >= 67108864u) { b = 1; continue; }
        const uint32_t bit = 1u << (h & 31);
        if (atomicOr(&bits[h >> 5], bit) & bit) b = 1;
    }
    if (__any(b) && (threadIdx.x & 63) == 0) atomicAdd(bad, 1u);
}

// ---------------------------------------------------------------------------------------------
// counting (PLAIN form: any byte that is not a base breaks the window; '\n' separates sequences)
// ---------------------------------------------------------------------------------------------
// count_kmers13.cpp:113-161: upper-case, ACGT only, forward strand; one lane per window start
__global__ void __launch_bounds__(kBlock) k_count13(const uint8_t* __restrict__ buf, uint64_t len, unsigned long long* __restrict__ table) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    if (len < 13) return;
    const uint64_t nwin = len - 12;
    for (uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x; p < nwin; p += stride) {
        uint64_t w0, w1;
        load13(buf + p, w0, w1);
        w0 &= 0xDFDFDFDFDFDFDFDFULL;                            // toupper for letters; nothing else maps onto ACGT
        w1 &= 0xDFDFDFDFDFDFDFDFULL;
        const Enc13 e = encode13_words(w0, w1);
        if (e.valid) atomicAdd(&table[e.code], 1ULL);
    }
}

// bytes equal to 'U' -> 'T' (after upper-casing), count_kmers.cpp:71-88
__device__ __forceinline__ uint64_t u_to_t(uint64_t x) {
    const uint64_t z = x ^ 0x5555555555555555ULL;
    const uint64_t nz = (((z & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL) | z) & 0x8080808080808080ULL;
    return x ^ ((~nz & 0x8080808080808080ULL) >> 7);
}
template <int LPP>
__global__ void __launch_bounds__(kBlock) k_count23_fixed(const IndexDev ix, const uint8_t* __restrict__ buf, uint64_t len, int canon_mode, uint32_t* __restrict__ tf_out) {
    if (len < 23) return;
    const uint64_t nwin = len - 22;
    IndexDev ixn = ix;
    ixn.bloom = nullptr;          // no absence filter in front of the table: it would be one more read for nearly every window
    ixn.early_exit = 0;           // windows of reads drawn from the indexed genome are mostly hits. With the verification table: one line
                                  // per window; without it (and for the overflow fall-back): the parallel three-read evaluation, whose
                                  // fingerprint (same 16 bytes as the pairs) still spares the key-record read of a window that is NOT a key
    AIX_WAVE_LOOP(p, nwin) {
        const bool in = p < nwin;
        uint64_t w0 = 0, w1 = 0, w2 = 0;
        if (in) load23(buf + p, w0, w1, w2);
        w0 = u_to_t(w0 & 0xDFDFDFDFDFDFDFDFULL);
        w1 = u_to_t(w1 & 0xDFDFDFDFDFDFDFDFULL);
        w2 = u_to_t(w2 & 0x00DFDFDFDFDFDFDFULL);
        const Enc23 e = encode23_words(w0, w1, w2);
        uint64_t key = e.code;
        if (canon_mode == 1) { const uint64_t x = revcomp_refx86(e.code, 23); key = e.code < x ? e.code : x; }
        else if (canon_mode == 2) { const uint64_t x = revcomp(e.code, 23); key = e.code < x ? e.code : x; }
        uint64_t s0, s1, s2;
        ascii23_of_rc(revcomp(key, 23), s0, s1, s2);
        const Probe pr = probe23_wave<LPP>(ixn, in && e.valid, s0, s1, s2, key);
        if (pr.found) atomicAdd(&tf_out[pr.slot], 1u);
    }
}

// count23 without global atomics, front end: the MPHF slot of every window (0xFFFFFFFF: not a key / not a clean window) as a
// coalesced 4-byte stream; aix_count13.hip's chunked-partition + LDS histogram then adds the stream into tf[] (1.3e9 scattered
// memory-side atomics per 10 M reads run at ~23 G/s and were what bounded k_count23_fixed once a probe cost one line).
template <int LPP>
__global__ void __launch_bounds__(kBlock) k_probe23_slots(const IndexDev ix, const uint8_t* __restrict__ buf, uint64_t len, int canon_mode, uint32_t* __restrict__ slots) {
    if (len < 23) return;
    const uint64_t nwin = len - 22;
    IndexDev ixn = ix;
    ixn.bloom = nullptr;
    ixn.early_exit = 0;
    AIX_WAVE_LOOP(p, nwin) {
        const bool in = p < nwin;
        uint64_t w0 = 0, w1 = 0, w2 = 0;
        if (in) load23(buf + p, w0, w1, w2);
        w0 = u_to_t(w0 & 0xDFDFDFDFDFDFDFDFULL);
        w1 = u_to_t(w1 & 0xDFDFDFDFDFDFDFDFULL);
        w2 = u_to_t(w2 & 0x00DFDFDFDFDFDFDFULL);
        const Enc23 e = encode23_words(w0, w1, w2);
        uint64_t key = e.code;
        if (canon_mode == 1) { const uint64_t x = revcomp_refx86(e.code, 23); key = e.code < x ? e.code : x; }
        else if (canon_mode == 2) { const uint64_t x = revcomp(e.code, 23); key = e.code < x ? e.code : x; }
        uint64_t s0, s1, s2;
        ascii23_of_rc(revcomp(key, 23), s0, s1, s2);
        const Probe pr = probe23_wave<LPP>(ixn, in && e.valid, s0, s1, s2, key);
        if (in) slots[p] = pr.found ? (uint32_t)pr.slot : 0xFFFFFFFFu;
    }
}

// count23 through K1: the distinct k-mers of the reads (already in the call's canonical form) with their counts; one probe per DISTINCT k-mer, and a plain
// read-modify-write of its slot — two distinct k-mers never share a slot
template <int LPP>
__global__ void __launch_bounds__(kBlock) k_add_counts23(const IndexDev ix, const uint64_t* __restrict__ keys, const uint64_t* __restrict__ counts, uint64_t n,
                                                        uint32_t* __restrict__ tf_out) {
    IndexDev ixn = ix;
    ixn.bloom = nullptr;
    ixn.early_exit = 0;
    AIX_WAVE_LOOP(i, n) {
        const bool in = i < n;
        const uint64_t key = in ? keys[i] : 0ull;
        uint64_t s0, s1, s2;
        ascii23_of_rc(revcomp(key, 23), s0, s1, s2);
        const Probe pr = probe23_wave<LPP>(ixn, in, s0, s1, s2, key);
        if (in && pr.found) tf_out[pr.slot] += (uint32_t)counts[i];          // u32 histogram: wraps like the per-window adds of the other back ends
    }
}

// K1 front end (count_kmers.cpp:93-136,297-308): the canonical 2-bit code of every window of a PLAIN
// buffer, ~0 where the window holds a non-base byte. Distinct counting = sort + run-length of this.
template <int K>
__global__ void __launch_bounds__(kBlock) k_window_codes(const uint8_t* __restrict__ buf, uint64_t len, int k, int canon_mode, uint64_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    if (len < (uint64_t)k) return;
    const uint64_t nwin = len - k + 1;
    for (uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x; p < nwin; p += stride) {
        uint64_t code = 0;
        bool valid = true;
        if (K == 23) {
            uint64_t w0, w1, w2;
            load23(buf + p, w0, w1, w2);
            w0 = u_to_t(w0 & 0xDFDFDFDFDFDFDFDFULL);
            w1 = u_to_t(w1 & 0xDFDFDFDFDFDFDFDFULL);
            w2 = u_to_t(w2 & 0x00DFDFDFDFDFDFDFULL);
            const Enc23 e = encode23_words(w0, w1, w2);
            code = e.code; valid = e.valid;
        } else if (K == 13) {
            uint64_t w0, w1;
            load13(buf + p, w0, w1);
            w0 = u_to_t(w0 & 0xDFDFDFDFDFDFDFDFULL);
            w1 = u_to_t(w1 & 0x000000DFDFDFDFDFULL);
            const Enc13 e = encode13_words(w0, w1);
            code = e.code; valid = e.valid;
        } else {
            for (int j = 0; j < k; ++j) {
                uint32_t c = buf[p + j] & 0xDFu;
                c = c == 'U' ? 'T' : c;
                const uint32_t v = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
                if (v == 4u) { valid = false; break; }
                code = (code << 2) | v;
            }
        }
        if (valid) {
            if (canon_mode == 1) { const uint64_t x = revcomp_refx86(code, k); code = code < x ? code : x; }
            else if (canon_mode == 2) { const uint64_t x = revcomp(code, k); code = code < x ? code : x; }
        }
        out[p] = valid ? code : ~0ULL;
    }
}

// ---------------------------------------------------------------------------------------------
// synthetic inputs (bit-identical to aindex_amd/synth.py)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t sm64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__global__ void __launch_bounds__(kBlock) k_synth_genome(uint64_t seed, uint64_t length, uint8_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < length; j += stride) {
        const uint64_t v = sm64(seed, j >> 5);
        out[j] = (uint8_t)(AIX_LUT_ACGT >> (8 * ((v >> (2 * (j & 31))) & 3)));
    }
}
__global__ void __launch_bounds__(kBlock) k_synth_kmers(uint64_t seed, uint64_t first, uint64_t N, int k, uint8_t* __restrict__ out) {
    const uint64_t total = N * (uint64_t)k, stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
        const uint64_t i = t / (uint64_t)k, j = t - i * (uint64_t)k;
        const uint64_t v = sm64(seed, first + i);
        out[t] = (uint8_t)(AIX_LUT_ACGT >> (8 * ((v >> (2 * (k - 1 - (int)j))) & 3)));
    }
}
__global__ void __launch_bounds__(kBlock) k_synth_reads(uint64_t seed, const uint8_t* __restrict__ genome, uint64_t glen, uint64_t first_read, uint64_t n_reads,
                                                       uint32_t read_len, int rc_half, uint32_t n_ppm, uint8_t* __restrict__ out) {
    const uint64_t rec = (uint64_t)read_len + 1, total = n_reads * rec, stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t span = glen - read_len + 1;
    for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
        const uint64_t ri = t / rec, j = t - ri * rec;
        if (j == read_len) { out[t] = '\n'; continue; }
        const uint64_t r = first_read + ri;
        const uint64_t v = sm64(seed, 2 * r);
        const uint64_t start = ((v >> 32) * span) >> 32;
        uint8_t ch;
        if (rc_half && (sm64(seed, 2 * r + 1) & 1)) {
            const uint8_t g = genome[start + (read_len - 1 - j)];
            ch = g == 'A' ? 'T' : g == 'C' ? 'G' : g == 'G' ? 'C' : g == 'T' ? 'A' : g;
        } else {
            ch = genome[start + j];
        }
        if (n_ppm) {
            const uint64_t h = sm64(seed ^ 0x5851F42D4C957F2DULL, r * read_len + j);
            if ((((h >> 32) * 1000000ULL) >> 32) < n_ppm) ch = 'N';
        }
        out[t] = ch;
    }
}

// Q_mix of SURVEY §8d: query i is, with probability 1/2, a 23-mer window of the genome at a uniform position on a
// uniform strand, else a uniform-random 23-mer.
__global__ void __launch_bounds__(kBlock) k_synth_mix23(uint64_t seed, const uint8_t* __restrict__ genome, uint64_t glen, uint64_t first, uint64_t N,
                                                       uint8_t* __restrict__ out) {
    const uint64_t total = N * 23, stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t span = glen - 22;
    for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
        const uint64_t i = t / 23, j = t - i * 23;
        const uint64_t v = sm64(seed, first + i);
        uint8_t ch;
        if (v & 1) {
            const uint64_t r = sm64(seed ^ 0xA5A5A5A5ULL, first + i);
            ch = (uint8_t)(AIX_LUT_ACGT >> (8 * ((r >> (2 * (22 - (int)j))) & 3)));
        } else {
            const uint64_t pos = ((v >> 32) * span) >> 32;
            if (v & 2) {
                const uint8_t g = genome[pos + 22 - j];
                ch = g == 'A' ? 'T' : g == 'C' ? 'G' : g == 'G' ? 'C' : g == 'T' ? 'A' : g;
            } else {
                ch = genome[pos + j];
            }
        }
        out[t] = ch;
    }
}

// ---------------------------------------------------------------------------------------------
// random-gather roofline probe: uniform-random ELEM-byte reads over a large table,
// UNROLL independent reads in flight per lane. The denominator the lookup kernels are judged against.
// ---------------------------------------------------------------------------------------------
template <int ELEM, int UNROLL>
__global__ void __launch_bounds__(kBlock) k_gather(const uint8_t* __restrict__ table, uint64_t n_elems, uint64_t n_access, uint64_t seed, uint64_t* __restrict__ sink) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock * UNROLL;
    uint64_t acc = 0;
    for (uint64_t i = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * UNROLL; i < n_access; i += stride) {
        uint64_t idx[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) idx[u] = ((sm64(seed, i + u) >> 32) * n_elems) >> 32;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (ELEM >= 32) {                                   // the whole element: ELEM / 16 loads of 16 bytes (a 64-byte half line, a 128-byte line)
                const uint4* p = (const uint4*)(table + idx[u] * (uint64_t)ELEM);
#pragma unroll
                for (int t = 0; t < ELEM / 16; ++t) { const uint4 v = p[t]; acc += (uint64_t)v.x ^ v.w; }
            } else if (ELEM == 16) {
                const BvRec r = ((const BvRec*)table)[idx[u]];
                acc += r.fp ^ r.prefix;
            } else if (ELEM == 8) {
                acc += ((const uint64_t*)table)[idx[u]];
            } else {
                acc += ((const uint32_t*)table)[idx[u]];
            }
        }
    }
    if (acc == 0x1234567887654321ULL) sink[0] = acc;   // keeps the loads alive; practically never taken
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
#define AIX_LAUNCH(kern, work, stream, ...)                                          \
    do {                                                                             \
        hipLaunchKernelGGL(kern, dim3(grid_for(work)), dim3(kBlock), 0, stream, __VA_ARGS__); \
        return hipGetLastError();                                                    \
    } while (0)

// lanes that share one bucket read: a launch-time choice (IndexDev::bk_lpp) among the instantiated widths
#define AIX_LPP_SWITCH(lpp, CALL)          \
    switch (lpp) {                          \
        case 1: return CALL(1);             \
        case 2: return CALL(2);             \
        case 4: return CALL(4);             \
        default: return CALL(8);            \
    }
template <bool CANON, int LPP>
static hipError_t lookup23_tf(const IndexDev& ix, const uint8_t* q, uint64_t N, LookupOut out, hipStream_t s) {
    AIX_LAUNCH((k_lookup23_ascii<MODE_TF, CANON, LPP>), N, s, ix, q, N, out);
}
template <bool CANON>
static hipError_t lookup23_ascii_mode(const IndexDev& ix, const uint8_t* q, uint64_t N, int mode, LookupOut out, hipStream_t s) {
    switch (mode) {
        case MODE_TF: {
#define AIX_CALL(L) lookup23_tf<CANON, L>(ix, q, N, out, s)
            AIX_LPP_SWITCH(ix.bk_lpp, AIX_CALL)
#undef AIX_CALL
        }
        case MODE_LINES: AIX_LAUNCH((k_lookup23_ascii<MODE_LINES, CANON, 8>), N, s, ix, q, N, out);
        case MODE_HASH: AIX_LAUNCH((k_lookup23_ascii<MODE_HASH, CANON, 8>), N, s, ix, q, N, out);
        case MODE_KIDSTRAND: AIX_LAUNCH((k_lookup23_ascii<MODE_KIDSTRAND, CANON, 8>), N, s, ix, q, N, out);
        case MODE_BOTH: AIX_LAUNCH((k_lookup23_ascii<MODE_BOTH, CANON, 8>), N, s, ix, q, N, out);
        case MODE_TOTAL: AIX_LAUNCH((k_lookup23_ascii<MODE_TOTAL, CANON, 8>), N, s, ix, q, N, out);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_lookup23_ascii(const IndexDev& ix, const uint8_t* q, uint64_t N, int mode, LookupOut out, hipStream_t s) {
    if (N == 0) return hipSuccess;
    return ix.canonical_only ? lookup23_ascii_mode<true>(ix, q, N, mode, out, s) : lookup23_ascii_mode<false>(ix, q, N, mode, out, s);
}
// self-test of wave_lower_bound_pair (aix_device.hpp): wave w of the grid looks up keys[w] and keys[w] + 1 in the sorted array
__global__ void __launch_bounds__(64) k_selftest_lower_bound(const uint16_t* __restrict__ a, uint32_t n, const uint32_t* __restrict__ keys, uint32_t* __restrict__ out) {
    const uint32_t r = wave_lower_bound_pair(a, n, keys[blockIdx.x]);
    if ((threadIdx.x & 31u) == 0) out[2 * blockIdx.x + (threadIdx.x >> 5)] = r;
}
hipError_t launch_selftest_lower_bound(const uint16_t* a, uint32_t n, const uint32_t* keys, uint32_t nkeys, uint32_t* out, hipStream_t s) {
    if (nkeys == 0) return hipSuccess;
    hipLaunchKernelGGL(k_selftest_lower_bound, dim3(nkeys), dim3(64), 0, s, a, n, keys, out);
    return hipGetLastError();
}
hipError_t launch_lookup23_codes(const IndexDev& ix, const uint64_t* codes, uint64_t N, uint32_t* out, hipStream_t s) {
    if (N == 0) return hipSuccess;
    if (ix.canonical_only) AIX_LAUNCH((k_lookup23_codes<true, 8>), N, s, ix, codes, N, out);
    AIX_LAUNCH((k_lookup23_codes<false, 8>), N, s, ix, codes, N, out);
}
hipError_t launch_lookup23_ragged(const IndexDev& ix, const uint8_t* bytes, const uint64_t* offs, uint64_t N, uint32_t* out, hipStream_t s) {
    if (N == 0) return hipSuccess;
    AIX_LAUNCH(k_lookup23_ragged, N, s, ix, bytes, offs, N, out);
}
hipError_t launch_lookup13_ascii(const IndexDev& ix, const uint8_t* q, uint64_t N, int mode, LookupOut out, hipStream_t s) {
    if (N == 0) return hipSuccess;
    switch (mode) {
        case MODE_TF: AIX_LAUNCH(k_lookup13_ascii<MODE_TF>, N, s, ix, q, N, out);
        case MODE_HASH: AIX_LAUNCH(k_lookup13_ascii<MODE_HASH>, N, s, ix, q, N, out);
        case MODE_BOTH: AIX_LAUNCH(k_lookup13_ascii<MODE_BOTH>, N, s, ix, q, N, out);
        case MODE_TOTAL: AIX_LAUNCH(k_lookup13_ascii<MODE_TOTAL>, N, s, ix, q, N, out);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_lookup13_ragged(const IndexDev& ix, const uint8_t* bytes, const uint64_t* offs, uint64_t N, uint32_t* out, hipStream_t s) {
    if (N == 0) return hipSuccess;
    AIX_LAUNCH(k_lookup13_ragged, N, s, ix, bytes, offs, N, out);
}
template <bool CANON, int LPP>
static hipError_t coverage_lpp(const IndexDev& ix, const uint8_t* seqs, const uint64_t* offs, uint64_t M, uint64_t total, uint32_t cutoff, uint32_t* out,
                               const uint64_t* out_offs, hipStream_t s) {
    AIX_LAUNCH((k_coverage<CANON, LPP>), total, s, ix, seqs, offs, M, total, (double)M / (double)total, cutoff, out, out_offs);
}
hipError_t launch_coverage(const IndexDev& ix, const uint8_t* seqs, const uint64_t* offs, uint64_t M, uint64_t total, uint32_t cutoff, uint32_t* out,
                           const uint64_t* out_offs, hipStream_t s) {
    if (M == 0 || total == 0) return hipSuccess;
    if (ix.k == 23 && ix.canonical_only) {
#define AIX_CALL(L) coverage_lpp<true, L>(ix, seqs, offs, M, total, cutoff, out, out_offs, s)
        AIX_LPP_SWITCH(ix.bk_lpp, AIX_CALL)
#undef AIX_CALL
    }
    return coverage_lpp<false, 8>(ix, seqs, offs, M, total, cutoff, out, out_offs, s);
}
hipError_t launch_build_keyrecs(const uint64_t* checker, const uint32_t* tf, uint64_t n, KeyRec* recs, uint32_t* noncanon, hipStream_t s) {
    if (n == 0) return hipSuccess;
    AIX_LAUNCH(k_build_keyrecs, n, s, checker, tf, n, recs, noncanon);
}
hipError_t launch_extract_tf(const IndexDev& ix, uint32_t* tf, uint64_t* checker, hipStream_t s) {
    if (ix.n == 0) return hipSuccess;
    AIX_LAUNCH(k_extract, ix.n, s, ix, tf, checker);
}
hipError_t launch_set_fingerprints(const IndexDev& ix, BvRec* recs_rw, EeRec* ee_rw, bool do_fp, hipStream_t s) {
    if (ix.n == 0) return hipSuccess;
    if (ee_rw) hipLaunchKernelGGL(k_init_ee, dim3(grid_for(ix.m.nrecs)), dim3(kBlock), 0, s, (const BvRec*)recs_rw, ix.m.nrecs, ee_rw);
    AIX_LAUNCH(k_set_fp, ix.n, s, ix, recs_rw, ee_rw, do_fp ? 1 : 0);
}
hipError_t launch_build_buckets(const MphfDev& m, const KeyRec* keys, uint64_t n, BkEntry* bk, uint32_t nb, uint32_t* fill, uint64_t* bloom, uint32_t nbloom, uint32_t nbm,
                                uint32_t* mfill, uint32_t* side, hipStream_t s) {
    if (n == 0 || nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_bk_init, dim3(grid_for((uint64_t)nb * 8)), dim3(kBlock), 0, s, bk, (uint64_t)nb * 8);
    hipLaunchKernelGGL(k_bk_fill, dim3(grid_for(n)), dim3(kBlock), 0, s, m, keys, n, bk, nb, fill, bloom, nbloom, nbm, mfill, side);
    AIX_LAUNCH(k_bk_flag, nb, s, bk, nb, (const uint32_t*)fill);
}
hipError_t launch_count_unfiled(const uint32_t* side, uint64_t n, uint32_t* counter, hipStream_t s) {
    if (n == 0) return hipSuccess;
    AIX_LAUNCH(k_side_count, n, s, side, n, counter);
}
hipError_t launch_side_unfiled(const KeyRec* keys, uint64_t n, uint32_t* side, KeyRec* unfiled, uint32_t* counter, hipStream_t s) {
    if (n == 0) return hipSuccess;
    AIX_LAUNCH(k_side_unfiled, n, s, keys, n, side, unfiled, counter);
}
hipError_t launch_fill_minimizer_table(const MphfDev& m, const KeyRec* keys, uint64_t n, BkEntry* mk, const uint32_t* off, uint32_t nbm, uint32_t* mcur, hipStream_t s) {
    if (n == 0 || nbm == 0) return hipSuccess;
    AIX_LAUNCH(k_mk_fill, n, s, m, keys, n, mk, off, nbm, mcur);
}
hipError_t launch_scatter23(const MphfDev& m, uint64_t n, uint64_t nslots, const uint8_t* keys, const uint64_t* codes, const uint32_t* counts, uint64_t* checker, uint32_t* tf,
                            uint32_t* occupied, uint32_t* conflict, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (keys) AIX_LAUNCH(k_scatter23<true>, n, s, m, n, nslots, keys, codes, counts, checker, tf, occupied, conflict);
    AIX_LAUNCH(k_scatter23<false>, n, s, m, n, nslots, keys, codes, counts, checker, tf, occupied, conflict);
}
hipError_t launch_perm13(const MphfDev& m, uint32_t* perm, hipStream_t s) { AIX_LAUNCH(k_perm13, 67108864ull, s, m, perm); }
hipError_t launch_tf13_to_code_order(const uint32_t* perm, const uint64_t* tf_mphf, uint64_t* tf_code, hipStream_t s) {
    AIX_LAUNCH(k_tf13_to_code, 67108864ull, s, perm, tf_mphf, tf_code);
}
hipError_t launch_count13_plain(const uint8_t* buf, uint64_t len, unsigned long long* table, hipStream_t s) {
    if (len < 13) return hipSuccess;
    AIX_LAUNCH(k_count13, len - 12, s, buf, len, table);
}
hipError_t launch_scatter13_to_mphf(const uint32_t* perm, const unsigned long long* table, uint64_t* out, int add, hipStream_t s) {
    if (add) AIX_LAUNCH(k_scatter13<true>, 67108864ull, s, perm, table, out);
    AIX_LAUNCH(k_scatter13<false>, 67108864ull, s, perm, table, out);
}
hipError_t launch_perm13_check(const uint32_t* perm, uint32_t* bits, uint32_t* bad, hipStream_t s) {
    AIX_LAUNCH(k_perm13_check, 67108864ull, s, perm, bits, bad);
}
template <int LPP>
static hipError_t count23_lpp(const IndexDev& ix, const uint8_t* buf, uint64_t len, int canon_mode, uint32_t* tf_out, hipStream_t s) {
    AIX_LAUNCH(k_count23_fixed<LPP>, len - 22, s, ix, buf, len, canon_mode, tf_out);
}
hipError_t launch_count23_fixed(const IndexDev& ix, const uint8_t* buf, uint64_t len, int canon_mode, uint32_t* tf_out, hipStream_t s) {
    if (len < 23 || ix.n == 0) return hipSuccess;
#define AIX_CALL(L) count23_lpp<L>(ix, buf, len, canon_mode, tf_out, s)
    AIX_LPP_SWITCH(ix.bk_lpp, AIX_CALL)
#undef AIX_CALL
}
template <int LPP>
static hipError_t probe23_slots_lpp(const IndexDev& ix, const uint8_t* buf, uint64_t len, int canon_mode, uint32_t* slots, hipStream_t s) {
    // AIX_PROBE_LDS_PAD=bytes (experiment): unused dynamic LDS per workgroup, to run the probe at a chosen number of waves per CU
    static const unsigned pad = [] { const char* e = getenv("AIX_PROBE_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }();
    if (pad) {
        hipLaunchKernelGGL(k_probe23_slots<LPP>, dim3(grid_for(len - 22)), dim3(kBlock), pad, s, ix, buf, len, canon_mode, slots);
        return hipGetLastError();
    }
    AIX_LAUNCH(k_probe23_slots<LPP>, len - 22, s, ix, buf, len, canon_mode, slots);
}
hipError_t launch_probe23_slots(const IndexDev& ix, const uint8_t* buf, uint64_t len, int canon_mode, uint32_t* slots, hipStream_t s) {
    if (len < 23 || ix.n == 0) return hipSuccess;
#define AIX_CALL(L) probe23_slots_lpp<L>(ix, buf, len, canon_mode, slots, s)
    AIX_LPP_SWITCH(ix.bk_lpp, AIX_CALL)
#undef AIX_CALL
}
hipError_t launch_add_counts23(const IndexDev& ix, const uint64_t* keys, const uint64_t* counts, uint64_t n, uint32_t* tf_out, hipStream_t s) {
    if (n == 0 || ix.n == 0) return hipSuccess;
    AIX_LAUNCH(k_add_counts23<2>, n, s, ix, keys, counts, n, tf_out);
}
hipError_t launch_window_codes(const uint8_t* buf, uint64_t len, int k, int canon_mode, uint64_t* out, hipStream_t s) {
    if (len < (uint64_t)k) return hipSuccess;
    if (k == 23) AIX_LAUNCH(k_window_codes<23>, len - k + 1, s, buf, len, k, canon_mode, out);
    if (k == 13) AIX_LAUNCH(k_window_codes<13>, len - k + 1, s, buf, len, k, canon_mode, out);
    AIX_LAUNCH(k_window_codes<0>, len - k + 1, s, buf, len, k, canon_mode, out);
}
hipError_t launch_gather(const uint8_t* table, uint64_t n_elems, int elem, int unroll, uint64_t n_access, uint64_t seed, uint64_t* sink, hipStream_t s) {
    if (n_access == 0) return hipSuccess;
    const uint64_t work = (n_access + unroll - 1) / unroll;
#define G(E, U) if (elem == E && unroll == U) AIX_LAUNCH((k_gather<E, U>), work, s, table, n_elems, n_access, seed, sink)
    G(16, 1); G(16, 4); G(8, 1); G(8, 4); G(4, 1); G(4, 4); G(32, 1); G(64, 1); G(128, 1);
#undef G
    return hipErrorInvalidValue;
}
hipError_t launch_synth_genome(uint64_t seed, uint64_t length, uint8_t* out, hipStream_t s) {
    if (length == 0) return hipSuccess;
    AIX_LAUNCH(k_synth_genome, length, s, seed, length, out);
}
hipError_t launch_synth_kmers(uint64_t seed, uint64_t first, uint64_t N, int k, uint8_t* out, hipStream_t s) {
    if (N == 0) return hipSuccess;
    AIX_LAUNCH(k_synth_kmers, N * (uint64_t)k, s, seed, first, N, k, out);
}
hipError_t launch_synth_mix23(uint64_t seed, const uint8_t* genome, uint64_t glen, uint64_t first, uint64_t N, uint8_t* out, hipStream_t s) {
    if (N == 0) return hipSuccess;
    AIX_LAUNCH(k_synth_mix23, N * 23, s, seed, genome, glen, first, N, out);
}
hipError_t launch_synth_reads(uint64_t seed, const uint8_t* genome, uint64_t glen, uint64_t first_read, uint64_t n_reads, uint32_t read_len, int rc_half,
                              uint32_t n_ppm, uint8_t* out, hipStream_t s) {
    if (n_reads == 0) return hipSuccess;
    AIX_LAUNCH(k_synth_reads, n_reads * ((uint64_t)read_len + 1), s, seed, genome, glen, first_read, n_reads, read_len, rc_half, n_ppm, out);
}

}  // namespace aix
