// aix_pool.hip — scratch memory for the calls that need multi-GB temporaries (sort buffers, staged host input).
// hipMalloc of a multi-GB block costs ~20 ms on this stack, and the K1 / A2 / I1 entry points used to make 5-15 of
// them per call (0.3 s of allocation around 0.04 s of kernels). Blocks are cached per device and handed out again
// (smallest cached block that fits, at most 2x oversize). A block is only released by code that has synchronised the
// stream it was used on, so a cached block never has work pending. AIX_SCRATCH_CACHE_GB (default 40) bounds what is
// kept; aix_scratch_trim() and aix_index_close() return everything to the driver (memory parked here is invisible to other
// allocators in the process, e.g. torch's).
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "aix_internal.hpp"

namespace aix {

namespace {
struct Block { void* p; size_t bytes; int device; };
std::mutex g_mu;
std::vector<Block> g_free;
std::unordered_map<void*, Block> g_live;
size_t g_cached = 0;
size_t cache_limit() {
    static const size_t lim = [] {
        const char* e = getenv("AIX_SCRATCH_CACHE_GB");
        const double gb = e ? atof(e) : 40.0;
        return (size_t)(gb < 0 ? 0 : gb * (double)(1ull << 30));
    }();
    return lim;
}
}  // namespace

hipError_t pool_alloc(void** out, size_t bytes) {
    *out = nullptr;
    if (bytes == 0) bytes = 1;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < g_free.size(); ++i)
            if (g_free[i].device == dev && g_free[i].bytes >= bytes && g_free[i].bytes / 2 <= bytes && (best == (size_t)-1 || g_free[i].bytes < g_free[best].bytes)) best = i;
        if (best != (size_t)-1) {
            const Block b = g_free[best];
            g_free.erase(g_free.begin() + (long)best);
            g_cached -= b.bytes;
            g_live[b.p] = b;
            *out = b.p;
            return hipSuccess;
        }
    }
    void* p = nullptr;
    e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {                          // out of memory: give the cache back and try once more
        (void)hipGetLastError();
        pool_trim();
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_live[p] = Block{p, bytes, dev};
    *out = p;
    return hipSuccess;
}

void pool_free(void* p) {
    if (!p) return;
    Block b{nullptr, 0, 0};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(p);
        if (it == g_live.end()) { b.p = p; }            // not ours (should not happen): fall through to hipFree
        else {
            b = it->second;
            g_live.erase(it);
            if (g_cached + b.bytes <= cache_limit()) {
                g_free.push_back(b);
                g_cached += b.bytes;
                return;
            }
        }
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (b.bytes && b.device != cur) (void)hipSetDevice(b.device);
    (void)hipFree(p);
    if (b.bytes && b.device != cur) (void)hipSetDevice(cur);
}

void pool_trim() {
    std::vector<Block> blocks;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        blocks.swap(g_free);
        g_cached = 0;
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (const Block& b : blocks) {
        if (b.device != cur) (void)hipSetDevice(b.device);
        (void)hipFree(b.p);
        if (b.device != cur) (void)hipSetDevice(cur);
    }
}

}  // namespace aix
