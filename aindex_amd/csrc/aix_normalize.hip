// aix_normalize.hip — FASTA / FASTQ record normalisation to PLAIN form on the GPU.
//
// Same bytes as the host routine aix_normalize_reads (readers of count_kmers13.cpp:211-272 and
// count_kmers.cpp:250-295). Each reader is a finite-state transducer over bytes with <= 8 states that emits at
// most one byte per input byte (plus one '\n' at end of input), so the state at the start of every 128-byte chunk
// comes from an exclusive scan of per-chunk transition FUNCTIONS (3 bits x 8 states packed in a u32, composed
// associatively), and the output offset of every chunk from a second (sum) scan:
//   k_norm_summarise : every chunk, simulated from every reachable start state at once -> transition function + emit counts
//   rocPRIM scan #1  : function composition  -> true start state per chunk
//   k_norm_pick      : emit count under the true start state
//   rocPRIM scan #2  : output offsets
//   k_norm_emit      : re-simulate from the true state and write the output (dword-packed stores)
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "../../include/aindex_hip.h"
#include "aix_internal.hpp"

namespace aix {

static constexpr int NCH = 128;     // bytes per chunk (one lane)
static constexpr int NTB = 256;

enum NormKind { NORM_FASTQ = 0, NORM_FASTA13 = 1, NORM_FASTAK1 = 2 };
static constexpr uint32_t kIdentityFuncConst = 0u | (1u << 3) | (2u << 6) | (3u << 9) | (4u << 12) | (5u << 15) | (6u << 18) | (7u << 21);

// One transducer step. Returns true when a byte is emitted (in `out`). States are 0..7.
template <int KIND>
__device__ __forceinline__ bool norm_step(uint32_t& st, uint32_t b, uint32_t& out) {
    if (KIND == NORM_FASTQ) {
        // st = (line_no & 3) | (nonempty << 2); count_kmers13.cpp:240-257: lines 4i+1, non-empty
        const uint32_t ln = st & 3u;
        if (b == '\n') {
            const bool emit = (ln == 1u) && (st & 4u);
            st = (ln + 1u) & 3u;
            out = '\n';
            return emit;
        }
        if (ln == 1u) { st |= 4u; out = b; return true; }
        return false;
    } else if (KIND == NORM_FASTA13) {
        // st bit0 = at line start, bit1 = in header, bit2 = open (sequence bytes since the last separator)
        // count_kmers13.cpp:211-235: empty lines skipped, '>' lines close the open record, others are appended
        if (st & 1u) {
            if (b == '\n') return false;                                   // empty line
            if (b == '>') {
                const bool emit = (st & 4u) != 0;
                st = 2u;                                                   // in header, not open
                out = '\n';
                return emit;
            }
            st = 4u;                                                       // sequence line, open
            out = b;
            return true;
        }
        if (b == '\n') { st = (st & 4u) | 1u; return false; }
        if (st & 2u) return false;
        out = b;
        return true;
    } else {
        // kmer_counter (count_kmers.cpp:250-295): st 0 = before the first '>', 1 = in header, 2 = in sequence
        if (st == 0u) { if (b == '>') st = 1u; return false; }
        if (b == '>') { st = 1u; out = '\n'; return true; }                // a record ends wherever the next '>' is
        if (st == 1u) { if (b == '\n') st = 2u; return false; }
        if (b == '\n' || b == '\r') return false;
        out = b;
        return true;
    }
}
template <int KIND>
__device__ __host__ __forceinline__ bool norm_final_newline(uint32_t st) {
    if (KIND == NORM_FASTQ) return (st & 3u) == 1u && (st & 4u);           // last sequence line without '\n'
    if (KIND == NORM_FASTA13) return (st & 4u) != 0;                       // :229-234
    return st != 0u;                                                       // the last record ends at EOF
}
template <int KIND>
__device__ __host__ __forceinline__ uint32_t norm_initial() { return KIND == NORM_FASTA13 ? 1u : 0u; }

struct ChunkSummary {
    uint32_t func;          // end state for start state s in bits [3s, 3s+3)
    uint8_t count[8];       // bytes emitted for start state s (<= NCH = 128)
};

// the start states that can actually occur, per reader (the others keep the identity mapping / zero counts)
template <int KIND> struct NormStates;
template <> struct NormStates<NORM_FASTQ>   { static constexpr int N = 5; __device__ static constexpr uint32_t at(int i) { return i == 0 ? 0u : i == 1 ? 1u : i == 2 ? 5u : i == 3 ? 2u : 3u; } };
template <> struct NormStates<NORM_FASTA13> { static constexpr int N = 4; __device__ static constexpr uint32_t at(int i) { return i == 0 ? 1u : i == 1 ? 2u : i == 2 ? 4u : 5u; } };
template <> struct NormStates<NORM_FASTAK1> { static constexpr int N = 3; __device__ static constexpr uint32_t at(int i) { return (uint32_t)i; } };

// a chunk's bytes in registers: full, 16-byte aligned chunks come in as 8 x uint4; anything else byte by byte
struct ChunkBytes {
    uint32_t w[NCH / 4];
    __device__ __forceinline__ uint32_t byte(int i) const { return (w[i >> 2] >> (8 * (i & 3))) & 0xFFu; }
};
__device__ __forceinline__ void load_chunk(const uint8_t* __restrict__ p, uint32_t n, ChunkBytes& cb) {
    if (n == NCH && (((uintptr_t)p) & 15) == 0) {
        const uint4* q = (const uint4*)p;
#pragma unroll
        for (int k = 0; k < NCH / 16; ++k) {
            const uint4 v = q[k];
            cb.w[4 * k] = v.x; cb.w[4 * k + 1] = v.y; cb.w[4 * k + 2] = v.z; cb.w[4 * k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NCH / 4; ++k) cb.w[k] = 0;
        for (uint32_t i = 0; i < n; ++i) cb.w[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
    }
}

template <int KIND>
__global__ void __launch_bounds__(NTB) k_norm_summarise(const uint8_t* __restrict__ raw, uint64_t len, uint64_t nchunks, ChunkSummary* __restrict__ sum) {
    const uint64_t c = (uint64_t)blockIdx.x * NTB + threadIdx.x;
    if (c >= nchunks) return;
    const uint64_t lo = c * NCH;
    const uint32_t n = (uint32_t)(len - lo < NCH ? len - lo : NCH);
    constexpr int NS = NormStates<KIND>::N;
    uint32_t st[NS], cnt[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { st[s] = NormStates<KIND>::at(s); cnt[s] = 0; }
    ChunkBytes cb;
    load_chunk(raw + lo, n, cb);
#pragma unroll 16
    for (int i = 0; i < NCH; ++i) {
        if ((uint32_t)i < n) {
            const uint32_t b = cb.byte(i);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                uint32_t o;
                cnt[s] += norm_step<KIND>(st[s], b, o) ? 1u : 0u;
            }
        }
    }
    ChunkSummary r;
    r.func = kIdentityFuncConst;
#pragma unroll
    for (int s = 0; s < 8; ++s) r.count[s] = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t from = NormStates<KIND>::at(s);
        r.func = (r.func & ~(7u << (3 * from))) | ((st[s] & 7u) << (3 * from));
        r.count[from] = (uint8_t)cnt[s];
    }
    sum[c] = r;
}

struct ComposeOp {   // (a then b)(s) = b(a(s))
    __device__ __host__ uint32_t operator()(uint32_t a, uint32_t b) const {
        uint32_t r = 0;
#pragma unroll
        for (int s = 0; s < 8; ++s) r |= ((b >> (3 * ((a >> (3 * s)) & 7u))) & 7u) << (3 * s);
        return r;
    }
};
static constexpr uint32_t kIdentityFunc = 0u | (1u << 3) | (2u << 6) | (3u << 9) | (4u << 12) | (5u << 15) | (6u << 18) | (7u << 21);

__global__ void __launch_bounds__(NTB) k_norm_funcs(const ChunkSummary* __restrict__ sum, uint64_t nchunks, uint32_t* __restrict__ funcs) {
    const uint64_t c = (uint64_t)blockIdx.x * NTB + threadIdx.x;
    if (c < nchunks) funcs[c] = sum[c].func;
}
__global__ void __launch_bounds__(NTB) k_norm_pick(const ChunkSummary* __restrict__ sum, const uint32_t* __restrict__ prefix, uint64_t nchunks, uint32_t init,
                                                  uint8_t* __restrict__ start_state, uint64_t* __restrict__ counts) {
    const uint64_t c = (uint64_t)blockIdx.x * NTB + threadIdx.x;
    if (c >= nchunks) return;
    const uint32_t s = (prefix[c] >> (3 * init)) & 7u;
    start_state[c] = (uint8_t)s;
    counts[c] = sum[c].count[s];
}

template <int KIND>
__global__ void __launch_bounds__(NTB) k_norm_emit(const uint8_t* __restrict__ raw, uint64_t len, uint64_t nchunks, const uint8_t* __restrict__ start_state,
                                                  const uint64_t* __restrict__ offs, uint8_t* __restrict__ out, uint64_t* __restrict__ out_len /* [0] length, [1] end state */,
                                                  int last) {
    const uint64_t c = (uint64_t)blockIdx.x * NTB + threadIdx.x;
    if (c >= nchunks) return;
    const uint64_t lo = c * NCH;
    const uint32_t n = (uint32_t)(len - lo < NCH ? len - lo : NCH);
    uint32_t st = start_state[c];
    uint64_t w = offs[c];
    ChunkBytes cb;
    load_chunk(raw + lo, n, cb);
    uint32_t acc = 0, have = 0;                      // bytes packed towards the next aligned dword of `out`
#pragma unroll 16
    for (int i = 0; i < NCH; ++i) {
        uint32_t o;
        if ((uint32_t)i < n && norm_step<KIND>(st, cb.byte(i), o)) {
            if (have == 0 && ((uintptr_t)(out + w) & 3)) { out[w++] = (uint8_t)o; continue; }   // head up to alignment
            acc |= o << (8 * have);
            if (++have == 4) { *(uint32_t*)(out + w) = acc; w += 4; acc = 0; have = 0; }
        }
    }
    for (uint32_t k = 0; k < have; ++k) out[w++] = (uint8_t)(acc >> (8 * k));
    if (c == nchunks - 1) {
        if (last && norm_final_newline<KIND>(st)) out[w++] = '\n';     // only the end of the INPUT closes the open record; a part hands its state on
        out_len[0] = w;
        out_len[1] = st;
    }
}

// scratch of one call, from the device-block cache (a 256 MiB part needs ~60 MB; hipMalloc / hipFree per call cost more than the kernels)
struct NormScratch {
    void* p = nullptr;
    ~NormScratch() { if (p) pool_free(p); }
};

template <int KIND>
static hipError_t normalise_impl(const uint8_t* d_raw, uint64_t len, uint8_t* d_out, uint64_t* d_out_len, uint32_t init, int last, hipStream_t s) {
    const uint64_t nchunks = (len + NCH - 1) / NCH;
    const unsigned grid = (unsigned)((nchunks + NTB - 1) / NTB);
    auto up = [](uint64_t x) { return (x + 255) / 256 * 256; };
    size_t tb = 0, tb2 = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, tb, (uint32_t*)nullptr, (uint32_t*)nullptr, kIdentityFunc, (size_t)nchunks, ComposeOp(), s);
    if (e == hipSuccess) e = rocprim::exclusive_scan(nullptr, tb2, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, (size_t)nchunks, rocprim::plus<uint64_t>(), s);
    if (e != hipSuccess) return e;
    if (tb2 > tb) tb = tb2;
    NormScratch scr;
    e = pool_alloc(&scr.p, up(sizeof(ChunkSummary) * nchunks) + 2 * up(4 * nchunks) + up(nchunks) + 2 * up(8 * nchunks) + up(tb ? tb : 1));
    if (e != hipSuccess) return e;
    uint8_t* w = (uint8_t*)scr.p;
    ChunkSummary* sum = (ChunkSummary*)w;   w += up(sizeof(ChunkSummary) * nchunks);
    uint32_t* funcs = (uint32_t*)w;         w += up(4 * nchunks);
    uint32_t* prefix = (uint32_t*)w;        w += up(4 * nchunks);
    uint8_t* start = w;                     w += up(nchunks);
    uint64_t* counts = (uint64_t*)w;        w += up(8 * nchunks);
    uint64_t* offs = (uint64_t*)w;          w += up(8 * nchunks);
    void* tmp = w;
    hipLaunchKernelGGL(k_norm_summarise<KIND>, dim3(grid), dim3(NTB), 0, s, d_raw, len, nchunks, sum);
    hipLaunchKernelGGL(k_norm_funcs, dim3(grid), dim3(NTB), 0, s, sum, nchunks, funcs);
    e = hipGetLastError();
    if (e == hipSuccess) e = rocprim::exclusive_scan(tmp, tb, funcs, prefix, kIdentityFunc, (size_t)nchunks, ComposeOp(), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_norm_pick, dim3(grid), dim3(NTB), 0, s, sum, prefix, nchunks, init, start, counts);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = rocprim::exclusive_scan(tmp, tb, counts, offs, (uint64_t)0, (size_t)nchunks, rocprim::plus<uint64_t>(), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_norm_emit<KIND>, dim3(grid), dim3(NTB), 0, s, d_raw, len, nchunks, start, offs, d_out, d_out_len, last);
        e = hipGetLastError();
    }
    const hipError_t e2 = hipStreamSynchronize(s);              // the scratch goes back to the cache idle
    return e != hipSuccess ? e : e2;
}

static uint32_t norm_initial_of(int format, int fasta_mode) {
    if (format == AIX_FMT_FASTQ) return norm_initial<NORM_FASTQ>();
    return fasta_mode == 0 ? norm_initial<NORM_FASTA13>() : norm_initial<NORM_FASTAK1>();
}

// One PART of an input that is normalised piece by piece (streaming ingestion): *state_io carries the reader's state from the end of the
// previous part (AIX_NORM_START before the first one) to the start of this one, so parts may be cut at ANY byte; the concatenation of the
// parts' outputs is byte for byte the output of one call over the whole input. `last`: this part ends the input (an open record gets its
// closing '\n'). d_out needs len + 1 bytes.
hipError_t normalise_device_part(const uint8_t* d_raw, uint64_t len, int format, int fasta_mode, uint8_t* d_out, uint64_t* out_len, uint32_t* state_io, int last,
                                 hipStream_t s) {
    *out_len = 0;
    uint32_t init = (state_io && *state_io != AIX_NORM_START) ? (*state_io & 7u) : norm_initial_of(format, fasta_mode);
    if (len == 0) {
        if (state_io) *state_io = init;
        if (last) {                                              // nothing left to read: only the record still open needs its '\n'
            bool nl = false;
            if (format == AIX_FMT_FASTQ) nl = norm_final_newline<NORM_FASTQ>(init);
            else if (fasta_mode == 0) nl = norm_final_newline<NORM_FASTA13>(init);
            else nl = norm_final_newline<NORM_FASTAK1>(init);
            if (nl) {
                const uint8_t c = '\n';
                hipError_t e = hipMemcpyAsync(d_out, &c, 1, hipMemcpyHostToDevice, s);
                if (e == hipSuccess) e = hipStreamSynchronize(s);
                if (e != hipSuccess) return e;
                *out_len = 1;
            }
        }
        return hipSuccess;
    }
    NormScratch d_len;
    hipError_t e = pool_alloc(&d_len.p, 16);
    if (e != hipSuccess) return e;
    if (format == AIX_FMT_FASTQ) e = normalise_impl<NORM_FASTQ>(d_raw, len, d_out, (uint64_t*)d_len.p, init, last, s);
    else if (fasta_mode == 0) e = normalise_impl<NORM_FASTA13>(d_raw, len, d_out, (uint64_t*)d_len.p, init, last, s);
    else e = normalise_impl<NORM_FASTAK1>(d_raw, len, d_out, (uint64_t*)d_len.p, init, last, s);
    uint64_t res[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(res, d_len.p, 16, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return e;
    *out_len = res[0];
    if (state_io) *state_io = (uint32_t)res[1];
    return hipSuccess;
}

// d_out needs len + 1 bytes. *out_len (host) receives the normalised length. format: AIX_FMT_FASTA / AIX_FMT_FASTQ.
hipError_t normalise_device(const uint8_t* d_raw, uint64_t len, int format, int fasta_mode, uint8_t* d_out, uint64_t* out_len, hipStream_t s) {
    uint32_t st = AIX_NORM_START;
    return normalise_device_part(d_raw, len, format, fasta_mode, d_out, out_len, &st, 1, s);
}

}  // namespace aix
