// aix_device.hpp — device-side primitives of the lookup / counting kernels (gfx950).
//
// Everything here is integer arithmetic and is also compiled for the host (AIX_HD) so that the CPU
// test-suite can check the exact-modulo and codec helpers without a GPU (aix_selftest_* in the ABI).
//
// Reference semantics restated here (file:line under /root/reference):
//   jenkins64 / mix          src/emphf/base_hash.hpp:38-91,127-145
//   mphf::lookup             src/emphf/mphf.hpp:79-89
//   bitpair get / rank       src/emphf/bitpair_vector.hpp:46-49, ranked_bitpair_vector.hpp:47-62
//   get_dna23_bitset         src/kmers.cpp:12-40      reverseDNA  src/kmers.cpp:355-388
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AIX_HD __host__ __device__ __forceinline__

namespace aix {

// ---------------------------------------------------------------------------------------------
// HBM layouts
// ---------------------------------------------------------------------------------------------
// One 16-byte record per 16 bit-pairs (half a word of the emphf bit-pair vector): the 16 pairs, the number of
// non-zero pairs in everything before them, and a 4-bit fingerprint per pair position.
//   rank(pos) = prefix + popc_pairs(pairs & mask(pos))            (ranked_bitpair_vector.hpp:47-62)
// so one MPHF evaluation is exactly three independent 16-byte reads and no dependent rank-directory read.
// fp nibble j = fingerprint of the key whose assigned (hidx-selected) node is pair j (0 if none): a probe whose
// fingerprint differs from the stored one cannot match checker[rank] and is answered "absent" without touching
// the key table (15/16 of all misses).
struct __attribute__((aligned(16))) BvRec {
    uint32_t pairs;      // 16 x 2 bits
    uint32_t prefix;
    uint64_t fp;         // 16 x 4 bits
};
// Early-exit table (23-mer lookups): the same pairs / prefix plus a 12-bit presence mask per pair position.
// Every key sets TWO bits (chosen by hash bits that do not depend on the node index) in the mask of EACH of its three
// nodes, so a query that finds one of its bits clear at any of its nodes is certainly not a stored key and the walk
// stops there: most absent keys cost ONE line (first node) instead of three (exactness argument: DESIGN.md §3).
struct __attribute__((aligned(32))) EeRec {
    uint32_t pairs;
    uint32_t prefix;
    uint32_t mask[6];    // 16 x 12 bits, field j at bit 12 j
};
// checker[] and tf[] of PHASH_MAP (hash.hpp:82-121) interleaved: one 16-byte read per probe.
struct __attribute__((aligned(16))) KeyRec {
    uint64_t code;
    uint32_t tf;
    uint32_t pad;
};

// Verification table of a 23-mer index ("bucket table"): ONE 128-byte line per probe instead of three MPHF records plus a
// key record. Every stored key that sits in its own MPHF slot (the only way the reference can find it: `checker[mphf(q)] ==
// code(q)`, python_wrapper.cpp:610-627) is filed under a bucket chosen by its Jenkins hash; a bucket is eight 16-byte
// entries {code, tf, slot}. A probe whose hashed bytes are the ASCII of its code reads its bucket and compares the codes
// in the line: a match is the reference's hit (slot = the MPHF value, tf = tf[slot]); no match in a bucket that never
// overflowed is the reference's miss. Keys beyond the eighth of a bucket are not filed: the bucket's last entry carries an
// overflow bit and an unmatched probe of such a bucket takes the MPHF path (exact either way).
struct __attribute__((aligned(16))) BkEntry {
    uint32_t code_lo;
    uint32_t code_hi;    // bits 0..13: code bits 32..45; bits 14..29 zero for a key, all ones for an empty entry; bit 31 (entry 7 only): overflow
    uint32_t tf;
    uint32_t slot;
};
#define AIX_BK_HI_MASK 0x3FFFFFFFu
#define AIX_BK_OVERFLOW 0x80000000u
#define AIX_BK_EMPTY_HI 0x3FFFFFFFu
#define AIX_BK_NONE 0xFFFFFFFFu

// Absence filter in front of the verification table (lookups and coverage, where most probes may be absent keys): a blocked
// Bloom filter, one 64-bit word per probe, four bits per key, filled from the same keys as the table. It is small enough to
// live in the Infinity Cache (16 bits per key by default), so an absent key is usually answered by ONE 8-byte cached read and
// never reaches HBM; a filed key passes it by construction. Word and bits come from hash words the bucket choice does not use.
__device__ __forceinline__ uint64_t bloom_mask(uint64_t c) {
    return (1ull << (c & 63)) | (1ull << ((c >> 6) & 63)) | (1ull << ((c >> 12) & 63)) | (1ull << ((c >> 18) & 63));
}
__device__ __forceinline__ uint32_t bloom_word(uint64_t b, uint32_t nwords) { return (uint32_t)__umul64hi(b, (uint64_t)nwords); }

// ---------------------------------------------------------------------------------------------
// exact h % d for a launch-invariant d (Moeller-Granlund 2-by-1 division, 32-bit limbs).
// gfx950 has no 64-bit integer divide; three of these replace the three `%` of mphf::lookup.
// ---------------------------------------------------------------------------------------------
struct FastMod {
    uint64_t d;      // divisor (m_hash_domain)
    uint32_t dn;     // d << s, top bit set
    uint32_t v;      // floor((2^64-1)/dn) - 2^32
    uint32_t s;      // clz32(d)
    uint32_t wide;   // d >= 2^32: fall back to the compiler's 64-bit remainder
};

inline FastMod make_fastmod(uint64_t d) {
    FastMod f{};
    f.d = d;
    if (d == 0) { f.wide = 1; f.d = 1; return f; }
    if (d >> 32) { f.wide = 1; return f; }
    uint32_t d32 = (uint32_t)d;
    f.s = (uint32_t)__builtin_clz(d32);
    f.dn = d32 << f.s;
    f.v = (uint32_t)((~0ull) / f.dn - (1ull << 32));
    return f;
}

AIX_HD uint32_t rem_2by1(uint32_t u1, uint32_t u0, uint32_t d, uint32_t v) {
    // precondition u1 < d, d normalised. Returns (u1*2^32 + u0) mod d.
    uint64_t q = (uint64_t)v * u1 + (((uint64_t)u1 << 32) | u0);
    uint32_t q1 = (uint32_t)(q >> 32) + 1u, q0 = (uint32_t)q;
    uint32_t r = u0 - q1 * d;
    if (r > q0) r += d;
    if (r >= d) r -= d;
    return r;
}

AIX_HD uint64_t fastmod(uint64_t h, const FastMod& f) {
    if (f.wide) return h % f.d;
    const uint32_t s = f.s;
    const uint32_t hi = (uint32_t)(h >> 32), lo = (uint32_t)h;
    // (u2,u1,u0) = h << s as 96 bits
    const uint32_t u2 = s ? (hi >> (32 - s)) : 0u;
    const uint32_t u1 = s ? ((hi << s) | (lo >> (32 - s))) : hi;
    const uint32_t u0 = lo << s;
    uint32_t r = rem_2by1(u2, u1, f.dn, f.v);
    r = rem_2by1(r, u0, f.dn, f.v);
    return (uint64_t)(r >> s);
}

// ---------------------------------------------------------------------------------------------
// Jenkins lookup8 mix
// ---------------------------------------------------------------------------------------------
AIX_HD void jmix(uint64_t& a, uint64_t& b, uint64_t& c) {
    a -= b; a -= c; a ^= (c >> 43);
    b -= c; b -= a; b ^= (a << 9);
    c -= a; c -= b; c ^= (b >> 8);
    a -= b; a -= c; a ^= (c >> 38);
    b -= c; b -= a; b ^= (a << 23);
    c -= a; c -= b; c ^= (b >> 5);
    a -= b; a -= c; a ^= (c >> 35);
    b -= c; b -= a; b ^= (a << 49);
    c -= a; c -= b; c ^= (b >> 11);
    a -= b; a -= c; a ^= (c >> 12);
    b -= c; b -= a; b ^= (a << 18);
    c -= a; c -= b; c ^= (b >> 22);
}
#define AIX_GOLDEN 0x9e3779b97f4a7c13ULL

// hash of a 23-byte key given as little-endian words: w0 = bytes 0-7, w1 = bytes 8-15,
// w2 = bytes 16-22 (56 bits). base_hash.hpp:57-88 with len = 23 (< 24: no main-loop round).
AIX_HD void jenkins23(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t seed, uint64_t& a, uint64_t& b, uint64_t& c) {
    a = seed + w0;
    b = seed + w1;
    c = AIX_GOLDEN + 23 + (w2 << 8);
    jmix(a, b, c);
}
// 13-byte key: w0 = bytes 0-7, w1 = bytes 8-12 (40 bits)
AIX_HD void jenkins13(uint64_t w0, uint64_t w1, uint64_t seed, uint64_t& a, uint64_t& b, uint64_t& c) {
    a = seed + w0;
    b = seed + w1;
    c = AIX_GOLDEN + 13;
    jmix(a, b, c);
}
// arbitrary length, bytes in (global) memory — the slow exact lane for len != k queries
AIX_HD void jenkins_bytes(const uint8_t* s, uint64_t len, uint64_t seed, uint64_t& a, uint64_t& b, uint64_t& c) {
    a = seed; b = seed; c = AIX_GOLDEN;
    uint64_t rem = len;
    while (rem >= 24) {
        uint64_t w[3] = {0, 0, 0};
        for (int i = 0; i < 24; ++i) w[i >> 3] |= (uint64_t)s[i] << (8 * (i & 7));
        a += w[0]; b += w[1]; c += w[2];
        jmix(a, b, c);
        s += 24; rem -= 24;
    }
    c += len;
    for (uint64_t i = 0; i < rem; ++i) {
        uint64_t v = (uint64_t)s[i];
        if (i < 8) a += v << (8 * i);
        else if (i < 16) b += v << (8 * (i - 8));
        else c += v << (8 * (i - 15));
    }
    jmix(a, b, c);
}

// ---------------------------------------------------------------------------------------------
// 2-bit codec
// ---------------------------------------------------------------------------------------------
AIX_HD uint32_t popc_pairs(uint64_t x) {   // nonzero_pairs(): same value as the broadword form
    x = (x | (x >> 1)) & 0x5555555555555555ULL;
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll(x);
#else
    return (uint32_t)__builtin_popcountll(x);
#endif
}

// reverse complement of a k-mer code (true rc; equals reverseDNA for k = 23 / 13)
AIX_HD uint64_t revcomp(uint64_t code, int k) {
    uint64_t y = __builtin_bitreverse64(code);                     // pairs reversed, bits in a pair swapped
    y = ((y >> 1) & 0x5555555555555555ULL) | ((y & 0x5555555555555555ULL) << 1);
    return (~y) >> (64 - 2 * k);
}
// kmer_counter's defective rc (count_kmers.cpp:116-130): the first stage swaps the two bits inside
// every base, so the net effect is reverse + (A<->T, C->C, G->G): plain bit reversal, complement.
AIX_HD uint64_t revcomp_refx86(uint64_t code, int k) {
    return (~__builtin_bitreverse64(code)) >> (64 - 2 * k);
}

// spread 4 pairs (pair i at bits 2i..2i+1 of x) to 4 bytes (byte i = pair i)
AIX_HD uint32_t spread4(uint32_t x) {
    x &= 0xffu;
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    return x;
}
// byte-wise LUT: out.byte[i] = lut.byte[sel.byte[i]] for sel bytes in 0..3
AIX_HD uint32_t lut4(uint32_t sel, uint32_t lut) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(0u, lut, sel);                    // v_perm_b32
#else
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) r |= ((lut >> (8 * ((sel >> (8 * i)) & 3))) & 0xffu) << (8 * i);
    return r;
#endif
}
#define AIX_LUT_ACGT 0x54474341u   // pair value -> 'A','C','G','T'
#define AIX_LUT_TGCA 0x41434754u   // pair value -> complement base

// ASCII words of the k-mer S whose reverse-complement code is `rc` (S[j] = comp(pair j of rc,
// counted from the least significant end)): forward string of code u = ascii_of_rc(revcomp(u));
// reverse-complement string of u = ascii_of_rc(u).
AIX_HD void ascii23_of_rc(uint64_t rc, uint64_t& w0, uint64_t& w1, uint64_t& w2) {
    uint32_t lo = (uint32_t)rc, hi = (uint32_t)(rc >> 32);
    uint32_t d0 = lut4(spread4(lo), AIX_LUT_TGCA);
    uint32_t d1 = lut4(spread4(lo >> 8), AIX_LUT_TGCA);
    uint32_t d2 = lut4(spread4(lo >> 16), AIX_LUT_TGCA);
    uint32_t d3 = lut4(spread4(lo >> 24), AIX_LUT_TGCA);
    uint32_t d4 = lut4(spread4(hi), AIX_LUT_TGCA);
    uint32_t d5 = lut4(spread4(hi >> 8), AIX_LUT_TGCA) & 0x00FFFFFFu;
    w0 = d0 | ((uint64_t)d1 << 32);
    w1 = d2 | ((uint64_t)d3 << 32);
    w2 = d4 | ((uint64_t)d5 << 32);
}
AIX_HD void ascii13_of_rc(uint32_t rc, uint64_t& w0, uint64_t& w1) {
    uint32_t d0 = lut4(spread4(rc), AIX_LUT_TGCA);
    uint32_t d1 = lut4(spread4(rc >> 8), AIX_LUT_TGCA);
    uint32_t d2 = lut4(spread4(rc >> 16), AIX_LUT_TGCA);
    uint32_t d3 = lut4(spread4(rc >> 24), AIX_LUT_TGCA) & 0x000000FFu;
    w0 = d0 | ((uint64_t)d1 << 32);
    w1 = d2 | ((uint64_t)d3 << 32);
}

// 4 ASCII bytes -> 4 pairs packed MSB-first into 8 bits, plus a per-byte "not A/C/G/T" mask.
// Invalid bytes contribute 0 bits exactly like get_dna23_bitset (kmers.cpp:12-25).
AIX_HD uint32_t encode4(uint32_t x, uint32_t& bad /* 0x01 per invalid byte */) {
    uint32_t c2 = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
    uint32_t y = lut4(c2, AIX_LUT_ACGT) ^ x;                       // zero byte <=> valid base
    uint32_t t = (((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
    bad = t >> 7;
    c2 &= ~(bad * 3u);
    return (c2 * 0x40100401u) >> 24;
}

// words of a raw 23-byte query -> sanitised code + validity
struct Enc23 {
    uint64_t code;
    bool valid;
};
AIX_HD Enc23 encode23_words(uint64_t w0, uint64_t w1, uint64_t w2) {
    uint32_t b0, b1, b2, b3, b4, b5;
    uint64_t code = encode4((uint32_t)w0, b0);
    code = (code << 8) | encode4((uint32_t)(w0 >> 32), b1);
    code = (code << 8) | encode4((uint32_t)w1, b2);
    code = (code << 8) | encode4((uint32_t)(w1 >> 32), b3);
    code = (code << 8) | encode4((uint32_t)w2, b4);
    uint32_t last = encode4((uint32_t)(w2 >> 32), b5);             // 4th byte is 0 -> "invalid", pair 0
    code = (code << 6) | (last >> 2);
    Enc23 e;
    e.code = code;
    e.valid = ((b0 | b1 | b2 | b3 | b4 | (b5 & 0x00010101u)) == 0);
    return e;
}
struct Enc13 {
    uint32_t code;
    bool valid;
};
AIX_HD Enc13 encode13_words(uint64_t w0, uint64_t w1) {
    uint32_t b0, b1, b2, b3;
    uint32_t code = encode4((uint32_t)w0, b0);
    code = (code << 8) | encode4((uint32_t)(w0 >> 32), b1);
    code = (code << 8) | encode4((uint32_t)w1, b2);
    uint32_t last = encode4((uint32_t)(w1 >> 32), b3);             // one real byte
    code = (code << 2) | (last >> 6);
    Enc13 e;
    e.code = code;
    e.valid = ((b0 | b1 | b2 | (b3 & 0x00000001u)) == 0);
    return e;
}

// ---------------------------------------------------------------------------------------------
// unaligned window loads. Every dword that is read contains at least one byte of the window, so a
// window that lies inside the caller's buffer can never fault, whatever the buffer's alignment.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void load23(const uint8_t* p, uint64_t& w0, uint64_t& w1, uint64_t& w2) {
    const uint32_t o = (uint32_t)((uintptr_t)p & 3);
    const uint32_t* q = (const uint32_t*)(p - o);     // pointer arithmetic, not integer round trip: the loads stay global_load, not flat_load
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4], d5 = q[5];
    const uint32_t d6 = (o >= 2) ? q[6] : 0u;        // bytes 20..22 reach dword 6 only when o >= 2
    const uint32_t sh = o * 8;
    const uint32_t e0 = __funnelshift_r(d0, d1, sh), e1 = __funnelshift_r(d1, d2, sh);
    const uint32_t e2 = __funnelshift_r(d2, d3, sh), e3 = __funnelshift_r(d3, d4, sh);
    const uint32_t e4 = __funnelshift_r(d4, d5, sh), e5 = __funnelshift_r(d5, d6, sh) & 0x00FFFFFFu;
    w0 = e0 | ((uint64_t)e1 << 32);
    w1 = e2 | ((uint64_t)e3 << 32);
    w2 = e4 | ((uint64_t)e5 << 32);
}
__device__ __forceinline__ void load13(const uint8_t* p, uint64_t& w0, uint64_t& w1) {
    const uint32_t o = (uint32_t)((uintptr_t)p & 3);
    const uint32_t* q = (const uint32_t*)(p - o);     // pointer arithmetic, not integer round trip: the loads stay global_load, not flat_load
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];   // byte 12 lies in dword 3 for every o
    const uint32_t sh = o * 8;
    const uint32_t e0 = __funnelshift_r(d0, d1, sh), e1 = __funnelshift_r(d1, d2, sh);
    const uint32_t e2 = __funnelshift_r(d2, d3, sh), e3 = __funnelshift_r(d3, 0u, sh) & 0x000000FFu;
    w0 = e0 | ((uint64_t)e1 << 32);
    w1 = e2 | ((uint64_t)e3 << 32);
}

// ---------------------------------------------------------------------------------------------
// MPHF evaluation against the device layout
// ---------------------------------------------------------------------------------------------
struct MphfDev {
    const EeRec* ee;     // early-exit table (nullptr when not built)
    const BvRec* recs;   // W records
    uint64_t D;          // hash domain
    uint64_t seed;
    uint64_t nrecs;      // ceil(B/16)
    FastMod fm;
};

// 4-bit fingerprint of a key, from hash bits that the node computation does not expose
__device__ __forceinline__ uint32_t fp_of_hash(uint64_t a, uint64_t b, uint64_t c) { return (uint32_t)((a ^ b ^ c) >> 60); }

// mphf::lookup (mphf.hpp:79-89) plus the stored fingerprint of the selected node and the node itself
__device__ __forceinline__ uint64_t mphf_from_hash_fp(const MphfDev& m, uint64_t a, uint64_t b, uint64_t c, uint32_t& fp_stored, uint64_t& node) {
    const uint64_t n0 = fastmod(a, m.fm);
    const uint64_t n1 = m.D + fastmod(b, m.fm);
    const uint64_t n2 = 2 * m.D + fastmod(c, m.fm);
    const BvRec r0 = m.recs[n0 >> 4];
    const BvRec r1 = m.recs[n1 >> 4];
    const BvRec r2 = m.recs[n2 >> 4];
    const uint32_t s0 = (uint32_t)(n0 & 15) * 2, s1 = (uint32_t)(n1 & 15) * 2, s2 = (uint32_t)(n2 & 15) * 2;
    const uint32_t v = ((r0.pairs >> s0) & 3) + ((r1.pairs >> s1) & 3) + ((r2.pairs >> s2) & 3);
    const uint32_t hidx = v - 3u * ((v * 11u) >> 5);               // v in 0..9 -> v % 3
    const uint32_t w = hidx == 0 ? r0.pairs : (hidx == 1 ? r1.pairs : r2.pairs);
    const uint32_t p = hidx == 0 ? r0.prefix : (hidx == 1 ? r1.prefix : r2.prefix);
    const uint32_t sh = hidx == 0 ? s0 : (hidx == 1 ? s1 : s2);
    const uint64_t f = hidx == 0 ? r0.fp : (hidx == 1 ? r1.fp : r2.fp);
    node = hidx == 0 ? n0 : (hidx == 1 ? n1 : n2);
    fp_stored = (uint32_t)(f >> (sh * 2)) & 15u;                   // nibble (node & 15)
    const uint32_t below = w & ((1u << sh) - 1u);
    return (uint64_t)p + (uint32_t)__builtin_popcount((below | (below >> 1)) & 0x55555555u);
}

// the two presence bits (of 12) a key sets at its i-th node (i = 0, 1, 2): 8 hash bits each, multiply-shift into 0..11
__device__ __forceinline__ uint32_t present_mask(uint64_t a, uint64_t b, uint64_t c, int i) {
    const uint64_t x = a ^ b ^ c;
    const uint32_t g1 = (uint32_t)(((x >> (56 - 16 * i)) & 0xFFu) * 12u) >> 8;
    const uint32_t g2 = (uint32_t)(((x >> (48 - 16 * i)) & 0xFFu) * 12u) >> 8;
    return (1u << g1) | (1u << g2);
}
// 12-bit presence field of a node, read straight from memory (same 32-byte record, i.e. same line, as pairs/prefix;
// a register copy of the whole record would be indexed at run time and spilled to scratch)
__device__ __forceinline__ uint32_t present_field(const EeRec* ee, uint64_t node) {
    const uint32_t bit = (uint32_t)(node & 15) * 12u;
    const uint32_t* w = ee[node >> 4].mask + (bit >> 5);
    const uint32_t sh = bit & 31u;
    uint32_t v = w[0] >> sh;
    if (sh > 20u) v |= w[1] << (32u - sh);                          // field straddles two dwords (never the last one)
    return v & 0xFFFu;
}

// Early-exit evaluation: the three records are read one after the other and the walk stops at the first node whose
// presence field lacks one of the key's bits. Returns false ("certainly not a stored key") or true with the slot.
__device__ __forceinline__ bool mphf_probe_early_exit(const MphfDev& m, uint64_t a, uint64_t b, uint64_t c, uint64_t& slot, uint32_t& records_read) {
    const uint64_t n0 = fastmod(a, m.fm);
    const uint2 h0 = *(const uint2*)(m.ee + (n0 >> 4));             // {pairs, prefix}
    const uint32_t q0 = present_mask(a, b, c, 0);
    records_read = 1;
    if ((present_field(m.ee, n0) & q0) != q0) return false;
    const uint64_t n1 = m.D + fastmod(b, m.fm);
    const uint2 h1 = *(const uint2*)(m.ee + (n1 >> 4));
    const uint32_t q1 = present_mask(a, b, c, 1);
    records_read = 2;
    if ((present_field(m.ee, n1) & q1) != q1) return false;
    const uint64_t n2 = 2 * m.D + fastmod(c, m.fm);
    const uint2 h2 = *(const uint2*)(m.ee + (n2 >> 4));
    const uint32_t q2 = present_mask(a, b, c, 2);
    records_read = 3;
    if ((present_field(m.ee, n2) & q2) != q2) return false;
    const uint32_t s0 = (uint32_t)(n0 & 15) * 2, s1 = (uint32_t)(n1 & 15) * 2, s2 = (uint32_t)(n2 & 15) * 2;
    const uint32_t v = ((h0.x >> s0) & 3) + ((h1.x >> s1) & 3) + ((h2.x >> s2) & 3);
    const uint32_t hidx = v - 3u * ((v * 11u) >> 5);
    const uint32_t w = hidx == 0 ? h0.x : (hidx == 1 ? h1.x : h2.x);
    const uint32_t p = hidx == 0 ? h0.y : (hidx == 1 ? h1.y : h2.y);
    const uint32_t sh = hidx == 0 ? s0 : (hidx == 1 ? s1 : s2);
    const uint32_t below = w & ((1u << sh) - 1u);
    slot = (uint64_t)p + (uint32_t)__builtin_popcount((below | (below >> 1)) & 0x55555555u);
    return true;
}

__device__ __forceinline__ uint64_t mphf_from_hash(const MphfDev& m, uint64_t a, uint64_t b, uint64_t c) {
    uint32_t fps;
    uint64_t node;
    return mphf_from_hash_fp(m, a, b, c, fps, node);
}

// ---------------------------------------------------------------------------------------------
// bucket table probe, wave-cooperative
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bucket_of(uint64_t a, uint32_t nb) { return (uint32_t)__umul64hi(a, (uint64_t)nb); }   // nb < 2^32

struct BkRes {
    uint32_t found, tf, slot, overflow, full;
};
// lower_bound of `key0` (lanes 0-31) and of `key0 + 1` (lanes 32-63) in a sorted array of `n` u16 values, by ONE wave: 32 probes per round, five
// rounds of independent loads for 10^7 entries instead of 22 dependent ones by a single lane. Returns this half-wave's answer (equal in its 32 lanes).
// All 64 lanes of the wave must call it.
__device__ __forceinline__ uint32_t wave_lower_bound_pair(const uint16_t* __restrict__ a, uint32_t n, uint32_t key0) {
    const uint32_t lane = threadIdx.x & 63u, half = lane >> 5, l5 = lane & 31u, key = key0 + half;
    uint32_t lo = 0, hi = n;                                 // the answer lies in [lo, hi]: a[i] < key for i < lo, a[i] >= key for i >= hi
    while (hi > lo) {
        const uint32_t span = hi - lo, step = (span + 32) / 33;
        const uint32_t at = lo + (l5 + 1) * step - 1;        // probes at the ends of 32 steps; those past hi count as "not less"
        const bool less = at < hi && a[at] < key;
        const uint64_t b = __ballot(less);
        const uint32_t nless = (uint32_t)__popc((uint32_t)(half ? b >> 32 : b));   // sorted input: the first nless probes say "less"
        const uint32_t nlo = min(lo + nless * step, hi);     // everything up to the last "less" probe is < key
        const uint32_t nhi = nless == 32u ? hi : min(hi, lo + (nless + 1) * step - 1);      // the first "not less" probe bounds the answer (there is none when all 32 said "less")
        lo = nlo;
        hi = max(nhi, lo);
    }
    return lo;
}

__device__ __forceinline__ uint32_t bperm(uint32_t src_lane, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }

// Read one 128-byte line of eight {code, tf, slot} entries per probe and compare the codes in it. All 64 lanes of the wave
// call this together; my_line = the line this lane's probe wants (AIX_BK_NONE: no probe). LPP consecutive lanes share one
// probe at a time: in round r the group reads the line of its r-th lane, each lane 128 / LPP bytes of it, so a wave-wide load
// instruction covers whole 128-byte lines (LPP = 8: eight lines per instruction, fully coalesced) instead of 64 lanes pulling
// 16 bytes out of 64 different lines eight times over. All loads are issued before the first compare.
// Returns per lane: found (+ tf, slot), overflow (bit 31 of the line's last entry), full (no empty entry in the line).
template <int LPP>
__device__ __forceinline__ BkRes line_probe_wave(const BkEntry* __restrict__ tab, uint32_t my_line, uint64_t code) {
    constexpr int EPL = 8 / LPP;                                   // entries per lane and round
    const uint32_t lane = __lane_id();
    const uint32_t j = lane & (LPP - 1), gbase = lane & ~(uint32_t)(LPP - 1);
    const uint32_t my_lo = (uint32_t)code, my_hi = (uint32_t)(code >> 32);
    uint4 e[LPP][EPL];
    uint32_t bsrc[LPP];
#pragma unroll
    for (int r = 0; r < LPP; ++r) {
        bsrc[r] = LPP == 1 ? my_line : bperm(gbase + r, my_line);
#pragma unroll
        for (int t = 0; t < EPL; ++t) e[r][t] = make_uint4(0xFFFFFFFFu, AIX_BK_EMPTY_HI, 0u, 0u);
        // a round in which no group of the wave has a probe (most rounds, when the absence filter has answered nearly every
        // lane) is skipped as a whole — a scalar branch; otherwise every lane loads (a group without a probe reads line 0,
        // which is always there: cheaper than masking the loads lane by lane)
        if (__ballot(bsrc[r] != AIX_BK_NONE) != 0ull) {
            const uint4* p = (const uint4*)(tab + (uint64_t)(bsrc[r] != AIX_BK_NONE ? bsrc[r] : 0u) * 8 + j * EPL);
#pragma unroll
            for (int t = 0; t < EPL; ++t) e[r][t] = p[t];
        }
    }
    BkRes res{0u, 0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < LPP; ++r) {
        const uint32_t c_lo = LPP == 1 ? my_lo : bperm(gbase + r, my_lo);
        const uint32_t c_hi = LPP == 1 ? my_hi : bperm(gbase + r, my_hi);
        const bool live = bsrc[r] != AIX_BK_NONE;
        uint32_t h_tf = 0, h_slot = 0;
        bool m = false, hole = false;
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const bool mm = live && e[r][t].x == c_lo && (e[r][t].y & AIX_BK_HI_MASK) == c_hi;
            if (mm) { m = true; h_tf = e[r][t].z; h_slot = e[r][t].w; }
            hole = hole || (e[r][t].y & AIX_BK_HI_MASK) == AIX_BK_EMPTY_HI;                      // an empty entry: the line is not full
        }
        const bool ov = live && j == (uint32_t)(LPP - 1) && (e[r][EPL - 1].y & AIX_BK_OVERFLOW);    // entry 7 of the line
        if (LPP == 1) {
            res.found = m; res.tf = h_tf; res.slot = h_slot; res.overflow = ov; res.full = live && !hole;
        } else {
            const uint64_t bal = __ballot(m), balov = __ballot(ov), balhole = __ballot(live && hole);
            const uint32_t grp = (uint32_t)(bal >> gbase) & ((1u << LPP) - 1u);
            const uint32_t ml = gbase + (grp ? (uint32_t)__builtin_ctz(grp) : 0u);
            const uint32_t v_tf = bperm(ml, h_tf), v_slot = bperm(ml, h_slot);
            if (j == (uint32_t)r) {
                res.found = grp != 0u; res.tf = v_tf; res.slot = v_slot;
                res.overflow = (uint32_t)(balov >> (gbase + LPP - 1)) & 1u;
                res.full = live && (((uint32_t)(balhole >> gbase) & ((1u << LPP) - 1u)) == 0u);
            }
        }
    }
    return res;
}

// The same probe in two halves, for kernels that keep MORE THAN ONE probe per lane in flight (k_run23_slots: the line of window j + 1 is
// requested before window j is compared, so a wave has twice as many lines outstanding): line_issue sends the loads, line_resolve compares.
template <int LPP>
struct LineRegs {
    uint4 e[LPP][8 / LPP];
    uint32_t bsrc[LPP];
};
template <int LPP>
__device__ __forceinline__ void line_issue(const BkEntry* __restrict__ tab, uint32_t my_line, LineRegs<LPP>& L) {
    constexpr int EPL = 8 / LPP;
    const uint32_t lane = __lane_id();
    const uint32_t j = lane & (LPP - 1), gbase = lane & ~(uint32_t)(LPP - 1);
#pragma unroll
    for (int r = 0; r < LPP; ++r) {
        L.bsrc[r] = LPP == 1 ? my_line : bperm(gbase + r, my_line);
#pragma unroll
        for (int t = 0; t < EPL; ++t) L.e[r][t] = make_uint4(0xFFFFFFFFu, AIX_BK_EMPTY_HI, 0u, 0u);
        if (__ballot(L.bsrc[r] != AIX_BK_NONE) != 0ull) {
            const uint4* p = (const uint4*)(tab + (uint64_t)(L.bsrc[r] != AIX_BK_NONE ? L.bsrc[r] : 0u) * 8 + j * EPL);
#pragma unroll
            for (int t = 0; t < EPL; ++t) L.e[r][t] = p[t];
        }
    }
}
template <int LPP>
__device__ __forceinline__ BkRes line_resolve(const LineRegs<LPP>& L, uint64_t code) {
    constexpr int EPL = 8 / LPP;
    const uint32_t lane = __lane_id();
    const uint32_t j = lane & (LPP - 1), gbase = lane & ~(uint32_t)(LPP - 1);
    const uint32_t my_lo = (uint32_t)code, my_hi = (uint32_t)(code >> 32);
    BkRes res{0u, 0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < LPP; ++r) {
        const uint32_t c_lo = LPP == 1 ? my_lo : bperm(gbase + r, my_lo);
        const uint32_t c_hi = LPP == 1 ? my_hi : bperm(gbase + r, my_hi);
        const bool live = L.bsrc[r] != AIX_BK_NONE;
        uint32_t h_tf = 0, h_slot = 0;
        bool m = false, hole = false;
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const bool mm = live && L.e[r][t].x == c_lo && (L.e[r][t].y & AIX_BK_HI_MASK) == c_hi;
            if (mm) { m = true; h_tf = L.e[r][t].z; h_slot = L.e[r][t].w; }
            hole = hole || (L.e[r][t].y & AIX_BK_HI_MASK) == AIX_BK_EMPTY_HI;
        }
        const bool ov = live && j == (uint32_t)(LPP - 1) && (L.e[r][EPL - 1].y & AIX_BK_OVERFLOW);
        if (LPP == 1) {
            res.found = m; res.tf = h_tf; res.slot = h_slot; res.overflow = ov; res.full = live && !hole;
        } else {
            const uint64_t bal = __ballot(m), balov = __ballot(ov), balhole = __ballot(live && hole);
            const uint32_t grp = (uint32_t)(bal >> gbase) & ((1u << LPP) - 1u);
            const uint32_t ml = gbase + (grp ? (uint32_t)__builtin_ctz(grp) : 0u);
            const uint32_t v_tf = bperm(ml, h_tf), v_slot = bperm(ml, h_slot);
            if (j == (uint32_t)r) {
                res.found = grp != 0u; res.tf = v_tf; res.slot = v_slot;
                res.overflow = (uint32_t)(balov >> (gbase + LPP - 1)) & 1u;
                res.full = live && (((uint32_t)(balhole >> gbase) & ((1u << LPP) - 1u)) == 0u);
            }
        }
    }
    return res;
}

// verification table keyed by the k-mer's own hash: bucket = mulhi64(a, nb)
template <int LPP>
__device__ __forceinline__ BkRes bucket_probe_wave(const BkEntry* __restrict__ bk, uint32_t nb, bool want, uint64_t a, uint64_t code) {
    return line_probe_wave<LPP>(bk, want ? bucket_of(a, nb) : AIX_BK_NONE, code);
}

// ---------------------------------------------------------------------------------------------
// Minimizer-keyed copy of the verification table, for the STREAMING counter (aix_stream23.hip: every window of a read).
// Consecutive windows of a sequence share 22 bases and — about seven in a row — their minimizer (the 15-mer of the window with
// the smallest hash, taken over both strands, so it is the same for a k-mer and its reverse complement). Filing a key under
// its minimizer instead of its own hash sends those windows to the SAME bucket: one read per super-k-mer instead of one per window.
// The keys of one minimizer arrive together (~8 at a time), so fixed 16-entry buckets overflowed for 15-30 % of the windows; the
// copy is therefore laid out by content: bucket b = entries [off[b], off[b + 1]) of one contiguous array, every filed key in it.
// A lane reads at most AIX_MK_ENTRIES of a bucket into registers; a window whose (longer) bucket does not show its code is
// UNDECIDED and is settled through the hash-keyed table inside the same kernel.
// ---------------------------------------------------------------------------------------------
#define AIX_MK_ENTRIES 16
__device__ __forceinline__ uint32_t mmer_mix(uint32_t x) {        // a bijection of u32: equal hashes <=> equal 15-mers
    x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13;
    return x;
}
// reverse complement of a 15-mer code (30 bits)
__device__ __forceinline__ uint32_t revcomp15(uint32_t x) {
    uint32_t y = __builtin_bitreverse32(x);
    y = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
    return (~y) >> 2;
}
// u: 46-bit code, r: its true reverse complement. The 15-mer at position i of u and the one at position 8 - i of r are
// reverse complements of each other; the smaller of the two is the canonical 15-mer of that position. Returns the HASH of the
// minimizer (mmer_mix is a bijection, so it names the 15-mer just as well).
__device__ __forceinline__ uint32_t minimizer23(uint64_t u, uint64_t r) {
    uint32_t best_h = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const uint32_t f = (uint32_t)(u >> (2 * (8 - i))) & 0x3FFFFFFFu;
        const uint32_t g = (uint32_t)(r >> (2 * i)) & 0x3FFFFFFFu;
        const uint32_t h = mmer_mix(f < g ? f : g);
        best_h = h < best_h ? h : best_h;
    }
    return best_h;
}
__device__ __forceinline__ uint32_t mk_home(uint32_t minimizer, uint32_t nbm) {      // the minimum of nine hashes is skewed towards 0: mixed again
    uint64_t z = ((uint64_t)minimizer + 1ull) * 0x9E3779B97F4A7C15ULL;
    z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 32;
    return (uint32_t)__umul64hi(z, (uint64_t)nbm);
}

}  // namespace aix
