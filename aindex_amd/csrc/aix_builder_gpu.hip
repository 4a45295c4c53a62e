// aix_builder_gpu.hip — MWHC perfect-hash construction on the GPU (N1 row, beyond the reference).
//
// Same hash function, seed stream (std::mt19937_64(37), mphf.hpp:45) and hash domain as the reference's builder and as
// aix_builder.hip, hence the same 3-hypergraph; the peeling is done in parallel rounds instead of the reference's
// sequential visit order (hypergraph_sorter_seq.hpp:81-91), so the bit-pair values — and therefore the .pf bytes and
// the key -> slot assignment — differ from compute_mphf_seq's while the file is an equally valid emphf MPHF that the
// reference loads and evaluates (mphf.hpp:79-113). Use aix_pf_build* (host) when byte-identity with the reference
// matters; use this when the keys are already in HBM and build time matters.
//
//   k_edges    : 2-bit code -> ASCII -> Jenkins -> (v0 < v1 < v2), degree and XOR-of-incident-edge-ids per vertex
//   rounds     : k_claim   every frontier vertex of degree 1 claims its only edge (atomicExch) -> peel list
//                k_remove  claimed edges leave their three vertices (degree--, xor ^= e); vertices that drop to
//                          degree 1 form the next frontier
//   k_assign   : rounds in reverse: value(hinge) = (orientation - value(other1) - value(other2)) mod 3, 0 -> 3
//                (mphf.hpp:56-64). Edges of one round never contain each other's hinge, so a round is parallel.
//   k_pack     : 2-bit values -> 64-bit words; block ranks on the host (ranked_bitpair_vector.hpp:17-31)
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/aindex_hip.h"
#include "aix_internal.hpp"

namespace aix {

static constexpr int GB = 256;
static inline unsigned ggrid(uint64_t work) {
    uint64_t b = (work + GB - 1) / GB;
    if (b > 16384) b = 16384;
    if (b == 0) b = 1;
    return (unsigned)b;
}

__global__ void __launch_bounds__(GB) k_edges(const uint64_t* __restrict__ codes, uint64_t n, int k, uint64_t seed, uint64_t D, FastMod fm,
                                             uint32_t* __restrict__ v0, uint32_t* __restrict__ v1, uint32_t* __restrict__ v2,
                                             uint32_t* __restrict__ deg, uint32_t* __restrict__ xe) {
    const uint64_t stride = (uint64_t)gridDim.x * GB;
    for (uint64_t e = (uint64_t)blockIdx.x * GB + threadIdx.x; e < n; e += stride) {
        const uint64_t code = codes[e];
        uint64_t a, b, c;
        if (k == 23) {
            uint64_t w0, w1, w2;
            ascii23_of_rc(revcomp(code, 23), w0, w1, w2);
            jenkins23(w0, w1, w2, seed, a, b, c);
        } else if (k == 13) {
            uint64_t w0, w1;
            ascii13_of_rc((uint32_t)revcomp(code, 13), w0, w1);
            jenkins13(w0, w1, seed, a, b, c);
        } else {
            uint8_t s[32];
            for (int j = 0; j < k; ++j) s[j] = (uint8_t)(AIX_LUT_ACGT >> (8 * ((code >> (2 * (k - 1 - j))) & 3)));
            jenkins_bytes(s, (uint64_t)k, seed, a, b, c);
        }
        const uint32_t x0 = (uint32_t)fastmod(a, fm), x1 = (uint32_t)(D + fastmod(b, fm)), x2 = (uint32_t)(2 * D + fastmod(c, fm));
        v0[e] = x0; v1[e] = x1; v2[e] = x2;
        atomicAdd(&deg[x0], 1u); atomicXor(&xe[x0], (uint32_t)e);
        atomicAdd(&deg[x1], 1u); atomicXor(&xe[x1], (uint32_t)e);
        atomicAdd(&deg[x2], 1u); atomicXor(&xe[x2], (uint32_t)e);
    }
}

__global__ void __launch_bounds__(GB) k_first_frontier(const uint32_t* __restrict__ deg, uint64_t m, uint32_t* __restrict__ frontier, uint32_t* __restrict__ fcount) {
    const uint64_t stride = (uint64_t)gridDim.x * GB;
    for (uint64_t v = (uint64_t)blockIdx.x * GB + threadIdx.x; v < m; v += stride)
        if (deg[v] == 1u) frontier[atomicAdd(fcount, 1u)] = (uint32_t)v;
}

__global__ void __launch_bounds__(GB) k_claim(const uint32_t* __restrict__ frontier, uint32_t fcount, const uint32_t* __restrict__ deg, const uint32_t* __restrict__ xe,
                                             uint32_t* __restrict__ claimed, uint32_t* __restrict__ peel_edge, uint32_t* __restrict__ peel_hinge,
                                             uint32_t* __restrict__ pcount) {
    const uint64_t stride = (uint64_t)gridDim.x * GB;
    for (uint64_t i = (uint64_t)blockIdx.x * GB + threadIdx.x; i < fcount; i += stride) {
        const uint32_t v = frontier[i];
        if (deg[v] != 1u) continue;                      // stale: its last edge was removed through another vertex
        const uint32_t e = xe[v];
        if (atomicExch(&claimed[e], 1u) == 0u) {
            const uint32_t idx = atomicAdd(pcount, 1u);
            peel_edge[idx] = e;
            peel_hinge[idx] = v;
        }
    }
}

__global__ void __launch_bounds__(GB) k_remove(const uint32_t* __restrict__ peel_edge, uint32_t lo, uint32_t hi, const uint32_t* __restrict__ v0,
                                              const uint32_t* __restrict__ v1, const uint32_t* __restrict__ v2, uint32_t* __restrict__ deg, uint32_t* __restrict__ xe,
                                              uint32_t* __restrict__ next_frontier, uint32_t* __restrict__ ncount) {
    const uint64_t stride = (uint64_t)gridDim.x * GB;
    for (uint64_t i = (uint64_t)lo + (uint64_t)blockIdx.x * GB + threadIdx.x; i < hi; i += stride) {
        const uint32_t e = peel_edge[i];
        const uint32_t u[3] = {v0[e], v1[e], v2[e]};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            atomicXor(&xe[u[j]], e);
            if (atomicSub(&deg[u[j]], 1u) == 2u) next_frontier[atomicAdd(ncount, 1u)] = u[j];   // just became degree 1
        }
    }
}

__global__ void __launch_bounds__(GB) k_assign(const uint32_t* __restrict__ peel_edge, const uint32_t* __restrict__ peel_hinge, uint32_t lo, uint32_t hi,
                                              const uint32_t* __restrict__ v0, const uint32_t* __restrict__ v1, const uint32_t* __restrict__ v2, uint8_t* __restrict__ bv) {
    const uint64_t stride = (uint64_t)gridDim.x * GB;
    for (uint64_t i = (uint64_t)lo + (uint64_t)blockIdx.x * GB + threadIdx.x; i < hi; i += stride) {
        const uint32_t e = peel_edge[i], h = peel_hinge[i];
        const uint32_t a = v0[e], b = v1[e], c = v2[e];                 // a < b < c
        uint32_t orient, o1, o2;
        if (h == a) { orient = 0; o1 = b; o2 = c; }
        else if (h == b) { orient = 1; o1 = a; o2 = c; }
        else { orient = 2; o1 = a; o2 = b; }
        uint32_t val = (orient + 9u - bv[o1] - bv[o2]) % 3u;            // mphf.hpp:58-63
        if (val == 0u) val = 3u;
        bv[h] = (uint8_t)val;
    }
}

__global__ void __launch_bounds__(GB) k_pack(const uint8_t* __restrict__ bv, uint64_t B, uint64_t W, uint64_t* __restrict__ words) {
    const uint64_t stride = (uint64_t)gridDim.x * GB;
    for (uint64_t w = (uint64_t)blockIdx.x * GB + threadIdx.x; w < W; w += stride) {
        uint64_t x = 0;
        for (int j = 0; j < 32; ++j) {
            const uint64_t pos = w * 32 + j;
            if (pos < B) x |= (uint64_t)(bv[pos] & 3u) << (2 * j);
        }
        words[w] = x;
    }
}

struct GpuBuf {
    void* p = nullptr;
    hipError_t alloc(uint64_t bytes, bool zero, hipStream_t s) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
        if (e == hipSuccess && zero) e = hipMemsetAsync(p, 0, bytes ? bytes : 1, s);
        return e;
    }
    ~GpuBuf() { if (p) (void)hipFree(p); }
};

// returns hipSuccess and *peelable; on success with *peelable the words (host) hold the bit-pair vector
static hipError_t try_build(const uint64_t* d_codes, uint64_t n, int k, uint64_t D, uint64_t seed, hipStream_t s, bool* peelable, std::vector<uint64_t>& words) {
    *peelable = false;
    const uint64_t m = 3 * D, B = m, W = (B + 31) / 32;
    GpuBuf v0, v1, v2, deg, xe, claimed, pedge, phinge, f0, f1, cnt, bv, dwords;
    hipError_t e = v0.alloc(4 * n, false, s);
    if (e == hipSuccess) e = v1.alloc(4 * n, false, s);
    if (e == hipSuccess) e = v2.alloc(4 * n, false, s);
    if (e == hipSuccess) e = deg.alloc(4 * m, true, s);
    if (e == hipSuccess) e = xe.alloc(4 * m, true, s);
    if (e == hipSuccess) e = claimed.alloc(4 * n, true, s);
    if (e == hipSuccess) e = pedge.alloc(4 * n, false, s);
    if (e == hipSuccess) e = phinge.alloc(4 * n, false, s);
    if (e == hipSuccess) e = f0.alloc(4 * m, false, s);
    if (e == hipSuccess) e = f1.alloc(4 * m, false, s);
    if (e == hipSuccess) e = cnt.alloc(16, true, s);                  // [0] peel count, [1] frontier A count, [2] frontier B count
    if (e != hipSuccess) return e;
    uint32_t* c = (uint32_t*)cnt.p;
    hipLaunchKernelGGL(k_edges, dim3(ggrid(n)), dim3(GB), 0, s, d_codes, n, k, seed, D, make_fastmod(D), (uint32_t*)v0.p, (uint32_t*)v1.p, (uint32_t*)v2.p,
                       (uint32_t*)deg.p, (uint32_t*)xe.p);
    hipLaunchKernelGGL(k_first_frontier, dim3(ggrid(m)), dim3(GB), 0, s, (const uint32_t*)deg.p, m, (uint32_t*)f0.p, c + 1);
    uint32_t host[4] = {0, 0, 0, 0};
    e = hipMemcpyAsync(host, c, 16, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    std::vector<uint32_t> round_start;
    uint32_t peeled = 0, fcount = host[1];
    uint32_t *fa = (uint32_t*)f0.p, *fb = (uint32_t*)f1.p;
    int cur = 1;                                                       // index of the current frontier's counter
    while (fcount) {
        round_start.push_back(peeled);
        const int nxt = cur == 1 ? 2 : 1;
        e = hipMemsetAsync(c + nxt, 0, 4, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_claim, dim3(ggrid(fcount)), dim3(GB), 0, s, fa, fcount, (const uint32_t*)deg.p, (const uint32_t*)xe.p, (uint32_t*)claimed.p,
                           (uint32_t*)pedge.p, (uint32_t*)phinge.p, c);
        e = hipMemcpyAsync(host, c, 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        const uint32_t now = host[0];
        if (now == peeled) { round_start.pop_back(); break; }
        hipLaunchKernelGGL(k_remove, dim3(ggrid(now - peeled)), dim3(GB), 0, s, (const uint32_t*)pedge.p, peeled, now, (const uint32_t*)v0.p, (const uint32_t*)v1.p,
                           (const uint32_t*)v2.p, (uint32_t*)deg.p, (uint32_t*)xe.p, fb, c + nxt);
        e = hipMemcpyAsync(host + nxt, c + nxt, 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        peeled = now;
        fcount = host[nxt];
        std::swap(fa, fb);
        cur = nxt;
    }
    if (peeled < n) return hipSuccess;                                  // not peelable with this seed
    round_start.push_back(peeled);
    e = bv.alloc(m, true, s);
    if (e == hipSuccess) e = dwords.alloc(8 * W, false, s);
    if (e != hipSuccess) return e;
    for (size_t r = round_start.size() - 1; r-- > 0;) {
        const uint32_t lo = round_start[r], hi = round_start[r + 1];
        hipLaunchKernelGGL(k_assign, dim3(ggrid(hi - lo)), dim3(GB), 0, s, (const uint32_t*)pedge.p, (const uint32_t*)phinge.p, lo, hi, (const uint32_t*)v0.p,
                           (const uint32_t*)v1.p, (const uint32_t*)v2.p, (uint8_t*)bv.p);
    }
    hipLaunchKernelGGL(k_pack, dim3(ggrid(W)), dim3(GB), 0, s, (const uint8_t*)bv.p, B, W, (uint64_t*)dwords.p);
    words.resize(W ? W : 1);
    e = hipMemcpyAsync(words.data(), dwords.p, 8 * W, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) *peelable = true;
    return e;
}

}  // namespace aix

extern "C" int aix_pf_build_codes_dev(const uint64_t* d_codes, uint64_t n, int k, int device, void* stream, void** pf_out, uint64_t* pf_len) {
    using namespace aix;
    if (!d_codes || !pf_out || !pf_len || n == 0 || k < 1 || k > 32) return AIX_ERR_ARG;
    const uint64_t D = ((uint64_t)std::ceil((double)n * 1.23) + 2) / 3;     // mphf.hpp:26
    if (D == 0) return AIX_ERR_ARG;
    if ((3 * D) >> 32 || n >> 32) return AIX_ERR_UNSUPPORTED;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return AIX_ERR_HIP;
    std::mt19937_64 rng(37);
    std::vector<uint64_t> words;
    uint64_t seed = 0;
    bool ok = false;
    int st = AIX_OK;
    for (int trial = 0; trial < 64 && !ok; ++trial) {
        seed = rng();
        hipError_t e = try_build(d_codes, n, k, D, seed, (hipStream_t)stream, &ok, words);
        if (e != hipSuccess) { st = AIX_ERR_HIP; break; }
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    if (st) return st;
    if (!ok) return AIX_ERR_CONFLICT;
    const uint64_t B = 3 * D, W = (B + 31) / 32, R = (B + 511) / 512;
    uint8_t* img = (uint8_t*)malloc(32 + 8 * (W + R));
    if (!img) return AIX_ERR_NOMEM;
    const uint64_t hdr[4] = {n, D, seed, B};
    memcpy(img, hdr, 32);
    memcpy(img + 32, words.data(), 8 * W);
    uint64_t run = 0;
    uint64_t* ranks = (uint64_t*)(img + 32 + 8 * W);
    for (uint64_t i = 0; i < W; ++i) {
        if ((i & 15) == 0) memcpy(ranks + (i >> 4), &run, 8);
        run += popc_pairs(words[i]);
    }
    *pf_out = img;
    *pf_len = 32 + 8 * (W + R);
    return AIX_OK;
}
