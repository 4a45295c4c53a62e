// aix_builder.cpp — MWHC minimal-perfect-hash builder producing emphf-format .pf images.
//
// Host code (the reference's builder is a sequential CPU program too). Bit-identical output to the
// reference's `compute_mphf_seq` for the same key list: same seed stream (std::mt19937_64(37)),
// same hash domain, same peeling visit order, same value assignment.
//   compute_mphf_main         src/emphf/compute_mphf_generic.hpp:19-61
//   mphf ctor                 src/emphf/mphf.hpp:21-67
//   try_generate_and_sort     src/emphf/hypergraph_sorter_seq.hpp:29-102
//   xored_adj_list            src/emphf/hypergraph.hpp:46-76 ; orientation :85-88
//   ranked_bitpair_vector     src/emphf/ranked_bitpair_vector.hpp:17-31 ; save :64-76
// Only the key hashing is spread over threads (it is order-independent); the graph phase follows the
// reference's sequential order because the resulting bit-pair values depend on it.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../../include/aindex_hip.h"
#include "aix_device.hpp"

namespace {

using aix::FastMod;

struct KeySource {
    const uint8_t* bytes;
    const uint64_t* offs;   // nullptr -> fixed stride
    uint64_t stride;
    uint64_t n;
    bool all13;             // keys are the 4^13 13-mers in 2-bit order, generated on the fly
    const uint64_t* codes = nullptr;   // keys are ASCII renderings of 2-bit codes of length `stride`
    inline void get(uint64_t i, const uint8_t*& p, uint64_t& len, uint8_t* tmp) const {
        if (all13 || codes) {
            static const char L[4] = {'A', 'C', 'G', 'T'};
            uint64_t x = codes ? codes[i] : i;
            const int k = codes ? (int)stride : 13;
            for (int j = k - 1; j >= 0; --j) { tmp[j] = (uint8_t)L[x & 3]; x >>= 2; }
            p = tmp; len = (uint64_t)k;
        } else if (offs) {
            p = bytes + offs[i]; len = offs[i + 1] - offs[i];
        } else {
            p = bytes + i * stride; len = stride;
        }
    }
};

template <typename node_t>
struct Adj {
    node_t degree, v1s, v2s;
};
template <typename node_t>
struct Edge {
    node_t v0, v1, v2;
};

template <typename node_t>
static void del_edge(std::vector<Adj<node_t>>& adj, node_t at, node_t lo, node_t hi) {
    Adj<node_t>& a = adj[at];
    a.degree -= 1;
    a.v1s ^= lo;
    a.v2s ^= hi;
}

// one trial: returns true when the 3-hypergraph is peelable; `order` = peeling order
template <typename node_t>
static bool try_peel(const KeySource& ks, uint64_t D, uint64_t seed, int nthreads, std::vector<Edge<node_t>>& edges,
                     std::vector<Adj<node_t>>& adj, std::vector<Edge<node_t>>& order) {
    const uint64_t n = ks.n, m = 3 * D;
    const FastMod fm = aix::make_fastmod(D);
    edges.resize(n);
    {   // hash all keys -> edges (v0 < v1 < v2 by construction: three disjoint node ranges)
        std::vector<std::thread> th;
        const uint64_t per = (n + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; ++t) {
            const uint64_t lo = std::min<uint64_t>(n, per * t), hi = std::min<uint64_t>(n, lo + per);
            th.emplace_back([&, lo, hi]() {
                uint8_t tmp[32];
                for (uint64_t i = lo; i < hi; ++i) {
                    const uint8_t* p; uint64_t len;
                    ks.get(i, p, len, tmp);
                    uint64_t a, b, c;
                    aix::jenkins_bytes(p, len, seed, a, b, c);
                    edges[i].v0 = (node_t)aix::fastmod(a, fm);
                    edges[i].v1 = (node_t)(D + aix::fastmod(b, fm));
                    edges[i].v2 = (node_t)(2 * D + aix::fastmod(c, fm));
                }
            });
        }
        for (auto& t : th) t.join();
    }
    adj.assign(m, Adj<node_t>{0, 0, 0});
    for (uint64_t i = 0; i < n; ++i) {          // hypergraph_sorter_seq.hpp:46-58, input order
        const Edge<node_t> e = edges[i];
        Adj<node_t>& a0 = adj[e.v0]; a0.degree += 1; a0.v1s ^= e.v1; a0.v2s ^= e.v2;
        Adj<node_t>& a1 = adj[e.v1]; a1.degree += 1; a1.v1s ^= e.v0; a1.v2s ^= e.v2;
        Adj<node_t>& a2 = adj[e.v2]; a2.degree += 1; a2.v1s ^= e.v0; a2.v2s ^= e.v1;
    }
    order.clear();
    order.reserve(n);
    auto visit = [&](node_t v0) {                // :64-79
        if (adj[v0].degree != 1) return;
        Edge<node_t> e{v0, adj[v0].v1s, adj[v0].v2s};
        order.push_back(e);
        node_t a = e.v0, b = e.v1, c = e.v2;     // canonicalize_edge: sort the three nodes (v1 < v2 already)
        if (a > c) std::swap(a, c);
        if (a > b) std::swap(a, b);
        del_edge(adj, a, b, c);
        del_edge(adj, b, a, c);
        del_edge(adj, c, a, b);
    };
    size_t qpos = 0;
    for (uint64_t v0 = 0; v0 < m; ++v0) {        // :81-91
        visit((node_t)v0);
        while (qpos < order.size()) {
            const Edge<node_t> cur = order[qpos];
            visit(cur.v1);
            visit(cur.v2);
            qpos += 1;
        }
    }
    return order.size() >= n;
}

template <typename node_t>
static int build_typed(const KeySource& ks, uint64_t D, int nthreads, std::vector<uint8_t>& out) {
    const uint64_t n = ks.n, B = 3 * D, W = (B + 31) / 32, R = (B + 511) / 512;
    std::mt19937_64 rng(37);                     // mphf.hpp:45
    std::vector<Edge<node_t>> edges, order;
    std::vector<Adj<node_t>> adj;
    uint64_t seed = 0;
    bool ok = false;
    for (int trial = 0; trial < 64; ++trial) {
        seed = rng();                            // jenkins64_hasher::generate
        if (try_peel<node_t>(ks, D, seed, nthreads, edges, adj, order)) { ok = true; break; }
    }
    if (!ok) return AIX_ERR_CONFLICT;            // duplicate keys: the reference would loop forever
    std::vector<Edge<node_t>>().swap(edges);
    std::vector<Adj<node_t>>().swap(adj);
    std::vector<uint64_t> words(W ? W : 1, 0);
    auto bv_get = [&](uint64_t pos) { return (words[pos >> 5] >> ((pos & 31) * 2)) & 3; };
    for (size_t i = order.size(); i-- > 0;) {    // reverse peeling order, mphf.hpp:56-64
        const Edge<node_t> e = order[i];
        const uint64_t target = (uint64_t)(e.v0 > e.v1) + (uint64_t)(e.v0 > e.v2);
        const uint64_t assigned = bv_get(e.v1) + bv_get(e.v2);
        uint64_t val = (target - assigned + 9) % 3;
        if (val == 0) val = 3;
        const uint64_t wp = (uint64_t)e.v0 >> 5, wo = ((uint64_t)e.v0 & 31) * 2;
        words[wp] &= ~(3ULL << wo);
        words[wp] |= val << wo;
    }
    out.resize(32 + 8 * (W + R));
    uint64_t hdr[4] = {n, D, seed, B};
    memcpy(out.data(), hdr, 32);
    memcpy(out.data() + 32, words.data(), 8 * W);
    uint64_t run = 0;
    uint64_t* ranks = (uint64_t*)(out.data() + 32 + 8 * W);
    for (uint64_t i = 0; i < W; ++i) {           // ranked_bitpair_vector.hpp:22-30
        if ((i & 15) == 0) { uint64_t r = run; memcpy(ranks + (i >> 4), &r, 8); }
        run += aix::popc_pairs(words[i]);
    }
    return AIX_OK;
}

static int build(const KeySource& ks, void** pf_out, uint64_t* pf_len) {
    if (!pf_out || !pf_len) return AIX_ERR_ARG;
    const uint64_t n = ks.n;
    const uint64_t D = ((uint64_t)std::ceil((double)n * 1.23) + 2) / 3;    // mphf.hpp:26
    if (D == 0) return AIX_ERR_ARG;
    int nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 32) nthreads = 32;
    std::vector<uint8_t> img;
    int st;
    try {
        if (3 * D >= (1ULL << 32)) st = build_typed<uint64_t>(ks, D, nthreads, img);   // compute_mphf_generic.hpp:42-53
        else st = build_typed<uint32_t>(ks, D, nthreads, img);
    } catch (const std::bad_alloc&) {
        return AIX_ERR_NOMEM;
    }
    if (st) return st;
    void* p = malloc(img.size());
    if (!p) return AIX_ERR_NOMEM;
    memcpy(p, img.data(), img.size());
    *pf_out = p;
    *pf_len = img.size();
    return AIX_OK;
}

}  // namespace

extern "C" int aix_pf_build(const char* keys, uint64_t n, uint32_t key_len, void** pf_out, uint64_t* pf_len) {
    if (!keys || n == 0 || key_len == 0) return AIX_ERR_ARG;
    KeySource ks{(const uint8_t*)keys, nullptr, key_len, n, false};
    return build(ks, pf_out, pf_len);
}
extern "C" int aix_pf_build_ragged(const char* bytes, const uint64_t* offsets, uint64_t n, void** pf_out, uint64_t* pf_len) {
    if (!bytes || !offsets || n == 0) return AIX_ERR_ARG;
    KeySource ks{(const uint8_t*)bytes, offsets, 0, n, false};
    return build(ks, pf_out, pf_len);
}
extern "C" int aix_pf_build_codes(const uint64_t* codes, uint64_t n, int k, void** pf_out, uint64_t* pf_len) {
    if (!codes || n == 0 || k < 1 || k > 32) return AIX_ERR_ARG;
    KeySource ks{nullptr, nullptr, (uint64_t)k, n, false, codes};
    return build(ks, pf_out, pf_len);
}
extern "C" int aix_pf_build_all_13mers(void** pf_out, uint64_t* pf_len) {
    KeySource ks{nullptr, nullptr, 13, AIX_TOTAL_13MERS, true};
    return build(ks, pf_out, pf_len);
}
extern "C" void aix_free(void* p) { free(p); }
