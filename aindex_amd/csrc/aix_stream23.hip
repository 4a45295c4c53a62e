// aix_stream23.hip — streaming probe of EVERY 23-byte window of a PLAIN buffer (counting against a fixed key set: the
// config-4 path, K1 -> I1 composed; window rules of count_kmers.cpp:71-136,297-308) against the minimizer-keyed copy of the
// verification table (aix_device.hpp).
//
// Why a second probe kernel: with one window per lane (k_probe23_slots) every window pays a full encode (60 VALU operations),
// a reverse complement, a Jenkins hash of its decoded ASCII (150) and a wave-cooperative line read (200): ~500 operations and
// one 128-byte HBM line per window — 39 G windows/s is 5.0 TB/s of lines AND ~60 % of the VALU peak, both ceilings at once.
// Consecutive windows share 22 of their 23 bases. Here a lane owns 32 consecutive window starts:
//   * its 54 bytes are encoded ONCE into a 2-bit stream (4-byte SWAR, as aix_count13.hip) and a validity mask; a window's
//     code is a funnel shift of that stream;
//   * the minimizer (smallest hash among the nine canonical 15-mers of the window) is kept incrementally: one new 15-mer hash
//     per window, minimum over a ring of nine;
//   * about seven windows in a row share a minimizer, hence their bucket of the table: the bucket (its first 16 entries) stays in
//     the lane's registers and is re-read only when the minimizer changes — one offsets read and one contiguous entry read per
//     super-k-mer, no hash of the k-mer at all.
// A window the bucket cannot decide (the bucket is longer than what a lane reads and does not show the code) is settled in the
// same kernel through the hash-keyed verification table (one wave-cooperative line read for the lanes that need it), so the
// result is exactly that of k_probe23_slots; k_fix23 remains only for a handle whose hash-keyed table is switched off.
#include "aix_internal.hpp"

namespace aix {

static constexpr int S23_W = 32;                      // window starts per lane
static constexpr int S23_ND = 14;                     // aligned dwords that cover its 54 bytes
static constexpr int S23_TB = 256;
static constexpr uint32_t S23_NONE = 0xFFFFFFFFu;     // no k-mer / not a key
static constexpr uint32_t S23_UND = 0xFFFFFFFEu;      // the table could not decide

// every 'U' byte of an upper-cased word becomes 'T' (count_kmers.cpp:71-88 maps U/u to 3)
__device__ __forceinline__ uint64_t s23_u_to_t(uint64_t x) {
    const uint64_t z = x ^ 0x5555555555555555ULL;
    const uint64_t nz = (((z & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL) | z) & 0x8080808080808080ULL;
    return x ^ ((~nz & 0x8080808080808080ULL) >> 7);
}

struct Run23 {
    uint64_t hi, lo;      // 2-bit stream of bases 0..55, base 0 in the top two bits of hi
    uint64_t vmask;       // bit i: byte i is a base (A/C/G/T/U, either case)
};

// bits [SH, SH + 64) of the 128-bit stream hi:lo, SH counted from the least significant end (compile-time)
template <int SH>
__device__ __forceinline__ uint64_t stream_shr(uint64_t hi, uint64_t lo) {
    if constexpr (SH >= 64) return hi >> (SH - 64);
    else if constexpr (SH == 0) return lo;
    else return (hi << (64 - SH)) | (lo >> SH);
}

__device__ __forceinline__ Run23 encode_run23(const uint8_t* __restrict__ buf, uint64_t len, uint64_t S) {
    Run23 r{0, 0, 0};
    if (S >= len) return r;
    const uint64_t limit = len - S;                   // bytes available from S
    uint32_t e[S23_ND];
    {
        const uint32_t o = (uint32_t)((uintptr_t)(buf + S) & 3);
        const uint32_t* q = (const uint32_t*)(buf + S - o);           // pointer arithmetic keeps these global_load (not flat_load)
        const int64_t first = (int64_t)S - o, end = (int64_t)len;
        uint32_t d[S23_ND + 1];
#pragma unroll
        for (int k = 0; k <= S23_ND; ++k) d[k] = (first + 4 * k < end) ? q[k] : 0x0A0A0A0Au;   // never read a dword past the buffer
        const uint32_t sh = o * 8;
#pragma unroll
        for (int k = 0; k < S23_ND; ++k) e[k] = __funnelshift_r(d[k], d[k + 1], sh);
    }
    uint32_t w[4] = {0, 0, 0, 0};
    uint64_t vmask = 0;
#pragma unroll
    for (int k = 0; k < S23_ND; ++k) {
        uint32_t x = e[k] & 0xDFDFDFDFu;                                                       // upper case
        {   // 'U' -> 'T' (count_kmers.cpp:71-88 maps U/u to 3)
            const uint32_t z = x ^ 0x55555555u;
            const uint32_t nz = (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;
            x ^= (~nz & 0x80808080u) >> 7;
        }
        const uint32_t v = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
        const uint32_t diff = x ^ lut4(v, AIX_LUT_ACGT);
        const uint32_t z = ~(((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff | 0x7F7F7F7Fu);        // 0x80 in every byte where diff == 0
        const uint32_t f = z >> 7;
        const uint32_t nib = (f | (f >> 7) | (f >> 14) | (f >> 21)) & 0xFu;
        const uint32_t q = ((v << 6) & 0xC0u) | ((v >> 4) & 0x30u) | ((v >> 14) & 0x0Cu) | (v >> 24);
        w[k >> 2] |= q << (24 - 8 * (k & 3));
        vmask |= (uint64_t)nib << (4 * k);
    }
    if (limit < 56) vmask &= (1ull << limit) - 1;     // bytes at and past the end of the buffer are separators
    r.hi = ((uint64_t)w[0] << 32) | w[1];
    r.lo = ((uint64_t)w[2] << 32) | w[3];
    r.vmask = vmask;
    return r;
}

// hash of the canonical 15-mer at base position P of the lane's stream
template <int P>
__device__ __forceinline__ uint32_t mmer_hash_at(const Run23& r) {
    const uint32_t f = (uint32_t)stream_shr<98 - 2 * P>(r.hi, r.lo) & 0x3FFFFFFFu;
    const uint32_t g = revcomp15(f);
    return mmer_mix(f < g ? f : g);
}

struct BucketRegs {
    uint4 e[AIX_MK_ENTRIES];
};
// bucket b of the minimizer-keyed copy = entries [off[b], off[b + 1]): the first `cap` (<= 16) of them into registers, the rest of the
// registers empty. Returns true when the bucket is longer than that (its windows are left to the hash-keyed table).
__device__ __forceinline__ bool load_bucket(const BkEntry* __restrict__ tab, const uint32_t* __restrict__ off, uint32_t b, uint32_t cap, BucketRegs& L) {
    const uint32_t o0 = off[b], cnt = off[b + 1] - o0;
    const uint4* p = (const uint4*)(tab + o0);
#pragma unroll
    for (int t = 0; t < AIX_MK_ENTRIES; ++t)                     // up to sixteen independent loads in flight; the bucket is contiguous
        L.e[t] = (uint32_t)t < min(cnt, cap) ? p[t] : make_uint4(0xFFFFFFFFu, AIX_BK_EMPTY_HI, 0u, 0u);
    return cnt > cap;
}
// compare `code` with the sixteen entries: the slot (or NONE)
__device__ __forceinline__ uint32_t scan_bucket(const BucketRegs& L, uint64_t code) {
    const uint32_t lo = (uint32_t)code, hi = (uint32_t)(code >> 32);
    uint32_t slot = S23_NONE;
#pragma unroll
    for (int t = 0; t < AIX_MK_ENTRIES; ++t)
        if (L.e[t].x == lo && (L.e[t].y & AIX_BK_HI_MASK) == hi) slot = L.e[t].w;
    return slot;
}

struct StreamState {
    uint32_t mh[9];       // ring: hashes of the nine canonical 15-mers of the current window (position p lives in mh[p % 9])
    uint32_t cur;         // bucket held in `regs`
    bool long_bucket;     // ... of which only the first mk_cap entries: a code it does not show is UNDECIDED, not absent
    BucketRegs regs;
};

template <int J>
__device__ __forceinline__ void stream_steps(const IndexDev& ix, const Run23& run, uint32_t valid, int canon_mode, StreamState& st, uint32_t* __restrict__ tr) {
    if constexpr (J < S23_W) {
        if constexpr (J > 0) st.mh[(J + 8) % 9] = mmer_hash_at<J + 8>(run);          // the 15-mer that enters; the one that left is overwritten
        uint32_t slot = S23_NONE;
        const bool want = (valid >> J) & 1u;
        // code of window J: 46 bits at base J of the stream
        const uint64_t code = stream_shr<82 - 2 * J>(run.hi, run.lo) & ((1ULL << 46) - 1);
        uint64_t key = code;
        uint32_t mz;
        {
            uint32_t m = st.mh[0];
#pragma unroll
            for (int i = 1; i < 9; ++i) m = st.mh[i] < m ? st.mh[i] : m;
            mz = m;
        }
        if (canon_mode == 2) { const uint64_t x = revcomp(code, 23); key = code < x ? code : x; }
        else if (canon_mode == 1) {
            const uint64_t x = revcomp_refx86(code, 23);
            if (x < code) { key = x; mz = minimizer23(x, revcomp(x, 23)); }          // kmer_counter's pseudo-complement is no strand of the window: its own minimizer
        }
        const uint32_t home = mk_home(mz, ix.nbm);
        if (want && home != st.cur) { st.long_bucket = load_bucket(ix.mk, ix.mk_off, home, ix.mk_cap, st.regs); st.cur = home; }   // ~ once per seven windows
        if (want) {
            const uint32_t s = scan_bucket(st.regs, key);
            slot = s != S23_NONE ? s : (st.long_bucket ? S23_UND : S23_NONE);
        }
        tr[J] = slot;
        stream_steps<J + 1>(ix, run, valid, canon_mode, st, tr);
    }
}

__global__ void __launch_bounds__(S23_TB) k_stream23_slots(const IndexDev ix, const uint8_t* __restrict__ buf, uint64_t len, uint64_t nwin, int canon_mode,
                                                          uint32_t* __restrict__ slots, uint32_t* __restrict__ any_undecided) {
    __shared__ uint32_t tr[S23_TB / 64][64 * (S23_W + 1)];       // per wave: lane-major results, transposed into coalesced stores
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t wave_first = ((uint64_t)blockIdx.x * S23_TB + (threadIdx.x & ~63u)) * S23_W;     // first window start of this wave
    if (wave_first >= nwin) return;
    const uint64_t S = wave_first + (uint64_t)lane * S23_W;
    const Run23 run = encode_run23(buf, len, S);
    // window j (bytes j .. j + 22) holds a 23-mer iff 23 consecutive mask bits are set
    const uint64_t m = run.vmask, m2 = m & (m >> 1), m4 = m2 & (m2 >> 2), m8 = m4 & (m4 >> 4), m16 = m8 & (m8 >> 8);
    const uint32_t valid = (uint32_t)(m16 & (m4 >> 16) & (m2 >> 20) & (m >> 22));
    StreamState st;
    st.cur = AIX_BK_NONE;
    st.long_bucket = false;
#pragma unroll
    for (int t = 0; t < AIX_MK_ENTRIES; ++t) st.regs.e[t] = make_uint4(0xFFFFFFFFu, AIX_BK_EMPTY_HI, 0u, 0u);
    st.mh[0] = mmer_hash_at<0>(run); st.mh[1] = mmer_hash_at<1>(run); st.mh[2] = mmer_hash_at<2>(run);
    st.mh[3] = mmer_hash_at<3>(run); st.mh[4] = mmer_hash_at<4>(run); st.mh[5] = mmer_hash_at<5>(run);
    st.mh[6] = mmer_hash_at<6>(run); st.mh[7] = mmer_hash_at<7>(run); st.mh[8] = mmer_hash_at<8>(run);
    uint32_t* mine = &tr[wave][lane * (S23_W + 1)];
    stream_steps<0>(ix, run, valid, canon_mode, st, mine);
    // the wave's 2048 results leave as 32 coalesced 256-byte stores (only this wave touches its LDS region)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bool und = false;
#pragma unroll 2
    for (int i = 0; i < S23_W; ++i) {
        const uint32_t idx = (uint32_t)i * 64u + lane;                               // position inside the wave's span
        uint32_t v = tr[wave][(idx >> 5) * (S23_W + 1) + (idx & 31u)];
        const uint64_t p = wave_first + idx;
        // A window its minimizer bucket could not decide (the bucket overflowed when the table was built: a few per cent) is settled
        // right here through the hash-keyed verification table — one wave-cooperative line read for the lanes that need it, the
        // MPHF only behind an overflowed hash bucket — instead of a second pass over the whole slot stream (k_fix23)
        const bool need = p < nwin && v == S23_UND && ix.bk != nullptr;
        if (__ballot(need) != 0ull) {                                                 // uniform
            uint64_t w0 = 0, w1 = 0, w2 = 0;
            if (need) load23(buf + p, w0, w1, w2);
            w0 = s23_u_to_t(w0 & 0xDFDFDFDFDFDFDFDFULL);
            w1 = s23_u_to_t(w1 & 0xDFDFDFDFDFDFDFDFULL);
            w2 = s23_u_to_t(w2 & 0x00DFDFDFDFDFDFDFULL);
            const Enc23 e = encode23_words(w0, w1, w2);
            uint64_t key = e.code;
            if (canon_mode == 1) { const uint64_t x = revcomp_refx86(e.code, 23); key = e.code < x ? e.code : x; }
            else if (canon_mode == 2) { const uint64_t x = revcomp(e.code, 23); key = e.code < x ? e.code : x; }
            uint64_t s0, s1, s2, a, b, c;
            ascii23_of_rc(revcomp(key, 23), s0, s1, s2);
            jenkins23(s0, s1, s2, ix.m.seed, a, b, c);
            const BkRes k = bucket_probe_wave<8>(ix.bk, ix.nb, need, a, key);
            if (need) {
                uint32_t slot = S23_NONE;
                if (k.found) slot = k.slot;
                else if (k.overflow) { const uint64_t h = mphf_from_hash(ix.m, a, b, c); if (h < ix.n && key_at(ix, h).code == key) slot = (uint32_t)h; }
                v = slot;
            }
        }
        if (p < nwin) { slots[p] = v; und = und || v == S23_UND; }
    }
    if (__ballot(und) != 0ull && lane == 0) *any_undecided = 1u;                      // without the verification table: k_fix23 has something to do
}

// ---------------------------------------------------------------------------------------------
// k_run23_slots: the hash-keyed verification table (one 128-byte line per window, as k_probe23_slots) probed by lanes that own a RUN of
// W consecutive window starts. Static VALU budget of k_probe23_slots per window (ISA listing, profiles/r03/probe23_valu_budget.txt):
// unaligned 23-byte load 22, upper-case + U->T + 4-byte SWAR encode 139, reverse complement + strand pick 16, ASCII of the canonical
// strand 45, Jenkins 74, bucket choice + wave-cooperative line read + compare 87 — the encode is the largest item and consecutive windows
// share 22 of their 23 bytes. Here the lane's W + 22 bytes are encoded ONCE (encode_run23: ~25 operations per 4 bytes, i.e. ~11-16 per
// window), a window's code is a shift of the 2-bit stream (~8), and the rest is unchanged: ~260 against ~383 per window, the same single
// line per window, the same slot stream out (through LDS, so that the stores stay coalesced). Windows behind an overflowed bucket are
// marked undecided and settled in the epilogue (bucket again, then the MPHF), exactly as in k_stream23_slots.
// ---------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ Run23 encode_run_w(const uint8_t* __restrict__ buf, uint64_t len, uint64_t S) {
    static_assert(W == 16 || W == 32, "window starts per lane");
    constexpr int ND = (W + 22 + 3) / 4;                          // dwords of the lane's bytes: 10 (W = 16) / 14 (W = 32)
    Run23 r{0, 0, 0};
    if (S >= len) return r;
    const uint64_t limit = len - S;
    uint32_t e[ND];
    {
        const uint32_t o = (uint32_t)((uintptr_t)(buf + S) & 3);
        const uint32_t* q = (const uint32_t*)(buf + S - o);
        const int64_t first = (int64_t)S - o, end = (int64_t)len;
        uint32_t d[ND + 1];
#pragma unroll
        for (int k = 0; k <= ND; ++k) d[k] = (first + 4 * k < end) ? q[k] : 0x0A0A0A0Au;
        const uint32_t sh = o * 8;
#pragma unroll
        for (int k = 0; k < ND; ++k) e[k] = __funnelshift_r(d[k], d[k + 1], sh);
    }
    uint32_t w[4] = {0, 0, 0, 0};
    uint64_t vmask = 0;
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        uint32_t x = e[k] & 0xDFDFDFDFu;
        {
            const uint32_t z = x ^ 0x55555555u;
            const uint32_t nz = (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;
            x ^= (~nz & 0x80808080u) >> 7;
        }
        const uint32_t v = ((x >> 1) ^ (x >> 2)) & 0x03030303u;
        const uint32_t diff = x ^ lut4(v, AIX_LUT_ACGT);
        const uint32_t z = ~(((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff | 0x7F7F7F7Fu);
        const uint32_t f = z >> 7;
        const uint32_t nib = (f | (f >> 7) | (f >> 14) | (f >> 21)) & 0xFu;
        const uint32_t q = ((v << 6) & 0xC0u) | ((v >> 4) & 0x30u) | ((v >> 14) & 0x0Cu) | (v >> 24);
        w[k >> 2] |= q << (24 - 8 * (k & 3));
        vmask |= (uint64_t)nib << (4 * k);
    }
    if (limit < 64) vmask &= (1ull << limit) - 1;
    r.hi = ((uint64_t)w[0] << 32) | w[1];
    r.lo = ((uint64_t)w[2] << 32) | w[3];
    r.vmask = vmask;
    return r;
}

template <int W, int LPP>
__global__ void __launch_bounds__(S23_TB) k_run23_slots(const IndexDev ix, const uint8_t* __restrict__ buf, uint64_t len, uint64_t nwin, int canon_mode,
                                                       uint32_t* __restrict__ slots) {
    __shared__ uint32_t tr[S23_TB / 64][64 * (W + 1)];           // per wave: lane-major results, transposed into coalesced stores
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t wave_first = ((uint64_t)blockIdx.x * S23_TB + (threadIdx.x & ~63u)) * W;
    if (wave_first >= nwin) return;
    const uint64_t S = wave_first + (uint64_t)lane * W;
    const Run23 run = encode_run_w<W>(buf, len, S);
    const uint64_t m = run.vmask, m2 = m & (m >> 1), m4 = m2 & (m2 >> 2), m8 = m4 & (m4 >> 4), m16 = m8 & (m8 >> 8);
    const uint32_t valid = (uint32_t)(m16 & (m4 >> 16) & (m2 >> 20) & (m >> 22));   // window j holds a 23-mer iff 23 consecutive mask bits are set
    uint32_t* mine = &tr[wave][lane * (W + 1)];
    const uint64_t seed = ix.m.seed;
    // canonical key of window j and the line it wants (AIX_BK_NONE: not a 23-mer)
    auto key_of = [&](int j, uint64_t& key) -> uint32_t {
        const uint32_t sh = 82u - 2u * (uint32_t)j;                                  // 46 bits at base j of the stream: bits [82 - 2 j, 128 - 2 j) of hi:lo
        const uint64_t code = (sh >= 64u ? run.hi >> (sh - 64u) : (run.hi << (64u - sh)) | (run.lo >> sh)) & ((1ULL << 46) - 1);
        key = code;
        if (canon_mode == 2) { const uint64_t x = revcomp(code, 23); key = code < x ? code : x; }
        else if (canon_mode == 1) { const uint64_t x = revcomp_refx86(code, 23); key = code < x ? code : x; }
        uint64_t s0, s1, s2, a, b, c;
        ascii23_of_rc(revcomp(key, 23), s0, s1, s2);
        jenkins23(s0, s1, s2, seed, a, b, c);
        return ((valid >> j) & 1u) ? bucket_of(a, ix.nb) : AIX_BK_NONE;
    };
    // two probes per lane in flight: the line of window j + 1 is requested before window j is compared
    uint64_t ka, kb;
    uint32_t la, lb;
    constexpr int L_ = LPP;
    auto settle = [&](int j, const LineRegs<L_>& L, uint64_t key, uint32_t line) {
        const BkRes k = line_resolve<L_>(L, key);
        mine[j] = line == AIX_BK_NONE ? S23_NONE : (k.found ? k.slot : (k.overflow ? S23_UND : S23_NONE));
    };
    LineRegs<L_> LA, LB;
    la = key_of(0, ka);
    line_issue<L_>(ix.bk, la, LA);
#pragma unroll 1
    for (int j = 0; j < W; j += 2) {
        lb = key_of(j + 1, kb);
        line_issue<L_>(ix.bk, lb, LB);
        settle(j, LA, ka, la);
        if (j + 2 < W) { la = key_of(j + 2, ka); line_issue<L_>(ix.bk, la, LA); }
        settle(j + 1, LB, kb, lb);
    }

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll 2
    for (int i = 0; i < W; ++i) {
        const uint32_t idx = (uint32_t)i * 64u + lane;
        uint32_t v = tr[wave][(idx / W) * (W + 1) + (idx % W)];
        const uint64_t p = wave_first + idx;
        const bool need = p < nwin && v == S23_UND;                                  // an unmatched window of an overflowed bucket: the MPHF decides
        if (__ballot(need) != 0ull) {
            if (need) {
                uint64_t w0, w1, w2;
                load23(buf + p, w0, w1, w2);
                w0 = s23_u_to_t(w0 & 0xDFDFDFDFDFDFDFDFULL);
                w1 = s23_u_to_t(w1 & 0xDFDFDFDFDFDFDFDFULL);
                w2 = s23_u_to_t(w2 & 0x00DFDFDFDFDFDFDFULL);
                const Enc23 e = encode23_words(w0, w1, w2);
                uint64_t key = e.code;
                if (canon_mode == 1) { const uint64_t x = revcomp_refx86(e.code, 23); key = e.code < x ? e.code : x; }
                else if (canon_mode == 2) { const uint64_t x = revcomp(e.code, 23); key = e.code < x ? e.code : x; }
                uint64_t s0, s1, s2, a, b, c;
                ascii23_of_rc(revcomp(key, 23), s0, s1, s2);
                jenkins23(s0, s1, s2, seed, a, b, c);
                const uint64_t h = mphf_from_hash(ix.m, a, b, c);
                v = (h < ix.n && key_at(ix, h).code == key) ? (uint32_t)h : S23_NONE;
            }
        }
        if (p < nwin) slots[p] = v;
    }
}

template <int W, int LPP>
static hipError_t run23_launch(const IndexDev& ix, const uint8_t* buf, uint64_t len, int canon_mode, uint32_t* slots, hipStream_t s) {
    const uint64_t nwin = len - 22;
    const uint64_t per_block = (uint64_t)S23_TB * W;
    const uint64_t blocks = (nwin + per_block - 1) / per_block;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_run23_slots<W, LPP>), dim3((unsigned)blocks), dim3(S23_TB), 0, s, ix, buf, len, nwin, canon_mode, slots);
    return hipGetLastError();
}
// slots[0, len - 22) through the hash-keyed table with one run of `w` (16 / 32) windows per lane; needs ix.bk
hipError_t launch_run23_slots(const IndexDev& ix, const uint8_t* buf, uint64_t len, int canon_mode, uint32_t* slots, int w, hipStream_t s) {
    if (len < 23 || ix.n == 0 || !ix.bk) return hipSuccess;
    const uint32_t lpp = ix.bk_lpp;
    if (w == 32) {
        if (lpp == 8) return run23_launch<32, 8>(ix, buf, len, canon_mode, slots, s);
        if (lpp == 4) return run23_launch<32, 4>(ix, buf, len, canon_mode, slots, s);
        if (lpp == 1) return run23_launch<32, 1>(ix, buf, len, canon_mode, slots, s);
        return run23_launch<32, 2>(ix, buf, len, canon_mode, slots, s);
    }
    if (lpp == 8) return run23_launch<16, 8>(ix, buf, len, canon_mode, slots, s);
    if (lpp == 4) return run23_launch<16, 4>(ix, buf, len, canon_mode, slots, s);
    if (lpp == 1) return run23_launch<16, 1>(ix, buf, len, canon_mode, slots, s);
    return run23_launch<16, 2>(ix, buf, len, canon_mode, slots, s);
}

// windows the table left UNDECIDED: the MPHF path, lane by lane (rare)
__global__ void __launch_bounds__(256) k_fix23(const IndexDev ix_, const uint8_t* __restrict__ buf, uint64_t nwin, int canon_mode, uint32_t* __restrict__ slots,
                                              const uint32_t* __restrict__ any_undecided) {
    if (*any_undecided == 0u) return;                                                // the usual case: nothing was left undecided
    IndexDev ix = ix_;
    ix.early_exit = 0;
    // the scan: four slots per lane and load (the slot array of a piece is 16-byte aligned), then the few undecided ones one by one
    const uint64_t stride = (uint64_t)gridDim.x * 256 * 4;
    for (uint64_t q = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; q < nwin; q += stride) {
        uint32_t v[4] = {0, 0, 0, 0};
        if (q + 4 <= nwin) { const uint4 x = *(const uint4*)(slots + q); v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
        else { for (int i = 0; i < 4; ++i) if (q + i < nwin) v[i] = slots[q + i]; }
        if (v[0] != S23_UND && v[1] != S23_UND && v[2] != S23_UND && v[3] != S23_UND) continue;
      for (int i = 0; i < 4; ++i) {
        if (v[i] != S23_UND) continue;
        const uint64_t p = q + i;
        uint64_t w0, w1, w2;
        load23(buf + p, w0, w1, w2);
        const uint64_t U = 0xDFDFDFDFDFDFDFDFULL;
        auto u2t = [](uint64_t x) {
            const uint64_t z = x ^ 0x5555555555555555ULL;
            const uint64_t nz = (((z & 0x7F7F7F7F7F7F7F7FULL) + 0x7F7F7F7F7F7F7F7FULL) | z) & 0x8080808080808080ULL;
            return x ^ ((~nz & 0x8080808080808080ULL) >> 7);
        };
        w0 = u2t(w0 & U); w1 = u2t(w1 & U); w2 = u2t(w2 & 0x00DFDFDFDFDFDFDFULL);
        const Enc23 e = encode23_words(w0, w1, w2);
        uint64_t key = e.code;
        if (canon_mode == 1) { const uint64_t x = revcomp_refx86(e.code, 23); key = e.code < x ? e.code : x; }
        else if (canon_mode == 2) { const uint64_t x = revcomp(e.code, 23); key = e.code < x ? e.code : x; }
        uint64_t s0, s1, s2, a, b, c;
        ascii23_of_rc(revcomp(key, 23), s0, s1, s2);
        jenkins23(s0, s1, s2, ix.m.seed, a, b, c);
        uint32_t slot = S23_NONE;
        const uint64_t h = mphf_from_hash(ix.m, a, b, c);
        if (h < ix.n && key_at(ix, h).code == key) slot = (uint32_t)h;
        slots[p] = slot;
      }
    }
}

// slots[0, len - 22) for a PLAIN buffer; needs ix.mk. flag: one zeroed device word (set when a window was left undecided)
hipError_t launch_stream23_slots(const IndexDev& ix, const uint8_t* buf, uint64_t len, int canon_mode, uint32_t* slots, uint32_t* flag, hipStream_t s) {
    if (len < 23 || ix.n == 0 || !ix.mk) return hipSuccess;
    const uint64_t nwin = len - 22;
    const uint64_t per_block = (uint64_t)S23_TB * S23_W;
    const uint64_t blocks = (nwin + per_block - 1) / per_block;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_stream23_slots, dim3((unsigned)blocks), dim3(S23_TB), 0, s, ix, buf, len, nwin, canon_mode, slots, flag);
    const uint64_t fb = (nwin + 1023) / 1024;
    hipLaunchKernelGGL(k_fix23, dim3((unsigned)(fb > 16384 ? 16384 : fb)), dim3(256), 0, s, ix, buf, nwin, canon_mode, slots, (const uint32_t*)flag);
    return hipGetLastError();
}

}  // namespace aix
