// aix_ingest.hip — streaming ingestion of sequence files for the counting paths (K13, the config-4 histogram, K1).
//
// The reference's count_kmers13 streams its input: a reader thread pushes sequences while the workers count
// (count_kmers13.cpp:166-183,211-272,277-350). The MI355X form of that pipeline:
//
//   file / caller memory --(T host threads: pread | memcpy)--> 3 pinned staging buffers --(copy stream, H2D)--> 3 device buffers
//        --(caller's stream)--> [FASTA / FASTQ: transducer normalisation, reader state handed from part to part] --> PLAIN part
//        --> counted into the resident table / distinct set
//
// Parts are cut at ANY byte: the readers are finite-state transducers whose state travels with the cut (aix_normalize.hip), and the
// PLAIN stream is counted with a carry — the last k - 1 bytes of everything counted so far stand in front of the next part, so every
// window is counted exactly once, in the part that holds its last byte. While part i is normalised and counted, part i + 1 is on the
// wire and part i + 2 is being read; HBM holds O(part) bytes however long the file is, the host holds 3 pinned parts and never a
// whole-file copy. The host-buffer twins of the ABI (aix_count13, aix_count23_fixed, aix_count_distinct, aix_positions_fill) go through
// the same pipeline with memcpy in place of pread.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "aix_handle.hpp"
#include "aix_ingest.hpp"

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---------------------------------------------------------------------------------------------
// host worker threads: slices of one large read / write / copy run side by side (one thread moves ~5-10 GB/s out of the page cache;
// PCIe 5 x16 takes ~50)
// ---------------------------------------------------------------------------------------------
class HostWorkers {
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::function<void()>> jobs;
    size_t pending = 0;
    bool stop = false;
    void run() {
        for (;;) {
            std::function<void()> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !jobs.empty(); });
                if (stop && jobs.empty()) return;
                j = std::move(jobs.back());
                jobs.pop_back();
            }
            j();
            std::lock_guard<std::mutex> lk(mu);
            if (--pending == 0) cv_done.notify_all();
        }
    }

public:
    explicit HostWorkers(unsigned n) { for (unsigned i = 0; i < n; ++i) workers.emplace_back([this] { run(); }); }
    ~HostWorkers() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_work.notify_all();
        for (auto& t : workers) t.join();
    }
    unsigned parts() const { return (unsigned)workers.size() + 1; }
    std::mutex busy;                                          // one sliced operation at a time
    // f(lo, hi) over [0, bytes) in page-aligned slices; the caller runs the first slice itself
    void sliced(uint64_t bytes, const std::function<void(uint64_t, uint64_t)>& f) {
        const uint64_t np = parts();
        if (bytes < (8u << 20) || np == 1) { f(0, bytes); return; }
        std::lock_guard<std::mutex> only(busy);
        const uint64_t slice = ((bytes + np - 1) / np + 4095) & ~(uint64_t)4095;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (uint64_t off = slice; off < bytes; off += slice) {
                const uint64_t hi = std::min(bytes, off + slice);
                jobs.emplace_back([&f, off, hi] { f(off, hi); });
                ++pending;
            }
        }
        cv_work.notify_all();
        f(0, std::min(slice, bytes));
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

HostWorkers& host_workers() {
    static HostWorkers pool([] {
        unsigned n = std::thread::hardware_concurrency();
        n = n ? std::min(12u, std::max(1u, n / 2)) : 4u;
        if (const char* e = getenv("AIX_INGEST_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) n = (unsigned)v; }
        return n - 1;                                         // + the calling thread
    }());
    return pool;
}

// ---------------------------------------------------------------------------------------------
// pinned staging blocks are kept between calls (pinning 64 MiB costs ~10 ms; a call needs three to five of them): a process that counts
// file after file pays once. AIX_PINNED_CACHE_MB (default 1024) bounds what is kept; aix_scratch_trim() frees it.
// ---------------------------------------------------------------------------------------------
struct PinnedPool {
    std::mutex mu;
    std::vector<std::pair<void*, uint64_t>> free_blocks;
    uint64_t cached = 0;
    static uint64_t limit() {
        static const uint64_t lim = [] { const char* e = getenv("AIX_PINNED_CACHE_MB"); return (uint64_t)(e ? atol(e) : 1024) << 20; }();
        return lim;
    }
    void* get(uint64_t bytes) {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < free_blocks.size(); ++i)
                if (free_blocks[i].second == bytes) {
                    void* p = free_blocks[i].first;
                    free_blocks.erase(free_blocks.begin() + (long)i);
                    cached -= bytes;
                    return p;
                }
        }
        void* p = nullptr;
        if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return p;
    }
    void put(void* p, uint64_t bytes) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (cached + bytes <= limit()) { free_blocks.emplace_back(p, bytes); cached += bytes; return; }
        }
        (void)hipHostFree(p);
    }
    void trim() {
        std::vector<std::pair<void*, uint64_t>> b;
        { std::lock_guard<std::mutex> lk(mu); b.swap(free_blocks); cached = 0; }
        for (auto& x : b) (void)hipHostFree(x.first);
    }
};
PinnedPool& pinned_pool() { static PinnedPool* p = new PinnedPool(); return *p; }     // never destroyed: the HIP runtime may be gone at exit

}  // namespace

namespace aix {

void pinned_trim() { pinned_pool().trim(); }

// pin the staging blocks of one streaming call in the background (a tool does this before it opens its index: by the time the first
// part is read the blocks are there)
void pinned_warm(int device) {
    const uint64_t part = ingest_part_bytes();
    std::thread([part, device] {
        if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return; }
        void* b[5];
        for (int i = 0; i < 5; ++i) b[i] = pinned_pool().get(i < 3 ? part : (32ull << 20));
        for (int i = 0; i < 5; ++i) pinned_pool().put(b[i], i < 3 ? part : (32ull << 20));
    }).detach();
}

// ---------------------------------------------------------------------------------------------
// ByteSource
// ---------------------------------------------------------------------------------------------
int ByteSource::open_file(const char* path) {
    fd = ::open(path, O_RDONLY);
    if (fd < 0) return AIX_ERR_IO;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); fd = -1; return AIX_ERR_IO; }
    len = (uint64_t)st.st_size;
    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    return AIX_OK;
}
void ByteSource::set_memory(const void* p, uint64_t n) {
    mem = (const uint8_t*)p;
    len = n;
    hipPointerAttribute_t a;
    if (p && hipPointerGetAttributes(&a, p) == hipSuccess) mem_pinned = (a.type == hipMemoryTypeHost);
    else (void)hipGetLastError();
}
ByteSource::~ByteSource() { if (fd >= 0) ::close(fd); }

int ByteSource::read(uint64_t off, void* dst, uint64_t n) const {
    if (off + n > len) return AIX_ERR_IO;
    if (mem) {
        host_workers().sliced(n, [&](uint64_t lo, uint64_t hi) { memcpy((char*)dst + lo, mem + off + lo, hi - lo); });
        return AIX_OK;
    }
    std::atomic<int> bad{0};
    host_workers().sliced(n, [&](uint64_t lo, uint64_t hi) {
        while (lo < hi) {
            const ssize_t r = pread(fd, (char*)dst + lo, hi - lo, (off_t)(off + lo));
            if (r <= 0) { bad = 1; return; }                  // a file that shrank under us
            lo += (uint64_t)r;
        }
    });
    return bad ? AIX_ERR_IO : AIX_OK;
}

int write_file_parallel(const char* path, const void* src, uint64_t bytes) {
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return AIX_ERR_IO;
    int st = AIX_OK;
    if (bytes && ftruncate(fd, (off_t)bytes) != 0) st = AIX_ERR_IO;
    if (!st) {
        std::atomic<int> bad{0};
        host_workers().sliced(bytes, [&](uint64_t lo, uint64_t hi) {
            while (lo < hi) {
                const ssize_t r = pwrite(fd, (const char*)src + lo, hi - lo, (off_t)lo);
                if (r <= 0) { bad = 1; return; }
                lo += (uint64_t)r;
            }
        });
        if (bad) st = AIX_ERR_IO;
    }
    if (::close(fd) != 0) st = AIX_ERR_IO;
    return st;
}

// ---------------------------------------------------------------------------------------------
// Ingest: the producer side of the pipeline
// ---------------------------------------------------------------------------------------------
Ingest::Ingest(const ByteSource& s, uint64_t part, int dev) : src(s), part_bytes(part), device(dev) {}

int Ingest::start(uint8_t* direct_dst) {
    direct = direct_dst;
    // (copies of consecutive parts on two alternating streams were measured: 41.9 / 38.9 / 32.7 GB/s with 32 / 64 / 128 MiB parts against
    // 43.8 GB/s with one stream and 64 MiB parts on the same box — one stream it is)
    if (hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return AIX_ERR_HIP; }
    for (int b = 0; b < NB; ++b) {
        if (hipEventCreate(&h2d_ev[b]) != hipSuccess || hipEventCreate(&h2d_t0[b]) != hipSuccess || hipEventCreateWithFlags(&used_ev[b], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return AIX_ERR_HIP;
        }
        if (!direct) {
            if (pool_alloc((void**)&dbuf[b], HDR + part_bytes + 64) != hipSuccess) { (void)hipGetLastError(); return AIX_ERR_NOMEM; }
            device_bytes += HDR + part_bytes + 64;
        }
    }
    started = true;
    producer = std::thread([this] { produce(); });
    return AIX_OK;
}

void Ingest::fail(int st) {
    std::lock_guard<std::mutex> lk(mu);
    if (!error) error = st;
    done = true;
    cv.notify_all();
}

void Ingest::produce() {
    if (hipSetDevice(device) != hipSuccess) { fail(AIX_ERR_HIP); return; }
    uint64_t off = 0;
    for (uint64_t c = 0; off < src.len; ++c) {
        const int b = (int)(c % NB);
        const uint64_t m = std::min(part_bytes, src.len - off);
        const void* wire = nullptr;
        if (src.mem && src.mem_pinned) {
            wire = src.mem + off;                               // the caller pinned its buffer: it goes over the wire as it is
        } else {
            if (c >= (uint64_t)NB) {                            // staging buffer b is off the wire
                if (hipEventSynchronize(h2d_ev[b]) != hipSuccess) { fail(AIX_ERR_HIP); return; }
                float ms = 0;
                if (hipEventElapsedTime(&ms, h2d_t0[b], h2d_ev[b]) == hipSuccess) seconds_h2d += ms * 1e-3; else (void)hipGetLastError();
            }
            if (!pin[b]) {                                      // allocated when first needed: the later ones while the first parts are already moving
                pin[b] = pinned_pool().get(part_bytes);
                if (!pin[b]) { fail(AIX_ERR_NOMEM); return; }
                pinned_bytes += part_bytes;
            }
            const double t0 = now_s();
            const int st = src.read(off, pin[b], m);
            seconds_read += now_s() - t0;
            if (st) { fail(st); return; }
            wire = pin[b];
        }
        uint8_t* dst = direct ? direct + off : dbuf[b] + HDR;
        if (!direct && c >= (uint64_t)NB) {                     // device buffer b: the consumer of part c - NB has let go of it
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return released + NB > c || abort_flag; });
            if (abort_flag) return;
            lk.unlock();
            if (hipStreamWaitEvent(copy_stream, used_ev[b], 0) != hipSuccess) { fail(AIX_ERR_HIP); return; }
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            if (abort_flag) return;
        }
        if (hipEventRecord(h2d_t0[b], copy_stream) != hipSuccess || hipMemcpyAsync(dst, wire, m, hipMemcpyHostToDevice, copy_stream) != hipSuccess ||
            hipEventRecord(h2d_ev[b], copy_stream) != hipSuccess) {
            fail(AIX_ERR_HIP);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            meta[b] = Part{dst, m, off, off + m == src.len};
            produced = c + 1;
            cv.notify_all();
        }
        off += m;
    }
    std::lock_guard<std::mutex> lk(mu);
    done = true;
    cv.notify_all();
}

int Ingest::next(Part* out, hipStream_t consumer) {
    const double t0 = now_s();
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return produced > handed || done; });
    if (error) return error;
    if (produced == handed) return 0;
    const int b = (int)(handed % NB);
    *out = meta[b];
    ++handed;
    lk.unlock();
    if (hipStreamWaitEvent(consumer, h2d_ev[b], 0) != hipSuccess) return AIX_ERR_HIP;
    seconds_wait += now_s() - t0;
    ++parts;
    bytes_in += out->len;
    return 1;
}

int Ingest::release(hipStream_t consumer) {
    const int b = (int)((handed - 1) % NB);
    if (hipEventRecord(used_ev[b], consumer) != hipSuccess) return AIX_ERR_HIP;
    std::lock_guard<std::mutex> lk(mu);
    released = handed;
    cv.notify_all();
    return AIX_OK;
}

// wait until every part is in HBM (direct mode: the whole source has been uploaded to the caller's buffer)
int Ingest::drain(hipStream_t consumer) {
    Part p;
    int st;
    while ((st = next(&p, consumer)) == 1) {
        st = release(consumer);
        if (st) return st;
    }
    return st;
}

Ingest::~Ingest() {
    if (started) {
        { std::lock_guard<std::mutex> lk(mu); abort_flag = true; cv.notify_all(); }
        if (producer.joinable()) producer.join();
    }
    if (copy_stream) (void)hipStreamSynchronize(copy_stream);
    for (int b = 0; b < NB; ++b) {
        if (pin[b]) pinned_pool().put(pin[b], part_bytes);
        if (dbuf[b]) pool_free(dbuf[b]);
        if (h2d_ev[b]) (void)hipEventDestroy(h2d_ev[b]);
        if (h2d_t0[b]) (void)hipEventDestroy(h2d_t0[b]);
        if (used_ev[b]) (void)hipEventDestroy(used_ev[b]);
    }
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
}

uint64_t ingest_part_bytes() {
    // 256 MiB parts, three in flight: 47.5 GB/s for an 8 GB PLAIN file with warm staging blocks (64 MiB: 43.8, 12 reader threads; 8 threads:
    // 37, 4: 22 — DESIGN.md 5). Pinning a block costs ~35 ms, so a process that will stream ONE file warms the pool first (aix_ingest_warm).
    uint64_t mb = 256;
    if (const char* e = getenv("AIX_INGEST_PART_MB")) { const long v = atol(e); if (v >= 1 && v <= 2047) mb = (uint64_t)v; }
    uint64_t bytes = mb << 20;
    if (const char* e = getenv("AIX_INGEST_TEST_PART")) { const long v = atol(e); if (v >= 1) bytes = (uint64_t)v; }   // test hook: parts of a few bytes, cuts everywhere
    return bytes;
}

// the source uploaded as it is into d_dst (len bytes), pipelined: host threads fill pinned part i + 1 while part i is on the wire
int upload_pipelined(const ByteSource& src, uint8_t* d_dst, int device, hipStream_t s) {
    if (src.len == 0) return AIX_OK;
    Ingest in(src, ingest_part_bytes(), device);
    int st = in.start(d_dst);
    if (!st) st = in.drain(s);
    return st;
}

// ---------------------------------------------------------------------------------------------
// PlainStream: parts -> PLAIN form with the carry in front
// ---------------------------------------------------------------------------------------------
// d_carry (HDR bytes) := the last k - 1 bytes of the HDR + plain_len bytes at `head`, behind '\n' filler
__global__ void k_make_carry(const uint8_t* __restrict__ head, uint64_t total, uint32_t keep, uint8_t* __restrict__ carry) {
    const uint32_t i = threadIdx.x;                            // Ingest::HDR lanes
    carry[i] = i + keep < Ingest::HDR ? (uint8_t)'\n' : head[total - Ingest::HDR + i];
}

PlainStream::PlainStream(const ByteSource& s, int fmt, int fasta_md, int kk, int dev, hipStream_t st)
    : src(s), format(fmt), fasta_mode(fasta_md), k(kk), stream(st), in(s, ingest_part_bytes(), dev) {}

int PlainStream::start() {
    if (format == AIX_FMT_AUTO) {                              // count_kmers13.cpp:194-206: the first byte decides
        uint8_t c = 0;
        format = AIX_FMT_PLAIN;
        if (src.len) {
            if (src.mem) c = src.mem[0];
            else if (pread(src.fd, &c, 1, 0) != 1) return AIX_ERR_IO;
            format = aix_detect_format((const char*)&c, 1);
        }
    }
    if (format != AIX_FMT_PLAIN && format != AIX_FMT_FASTA && format != AIX_FMT_FASTQ) return AIX_ERR_ARG;
    HIPCHK(hipMalloc((void**)&d_carry, Ingest::HDR));
    HIPCHK(hipMemsetAsync(d_carry, '\n', Ingest::HDR, stream));
    if (format != AIX_FMT_PLAIN) {
        HIPCHK(hipMalloc((void**)&d_plain, Ingest::HDR + in.part_bytes + 64));
        plain_dev_bytes = Ingest::HDR + in.part_bytes + 64;
    }
    return in.start(nullptr);
}

// the next PLAIN part: *d = HDR bytes of carry ('\n' filler + the k - 1 bytes that precede the part) followed by the part; count every
// window of [*d, *d + *len). 1 = a part, 0 = end of input, < 0 = error. The previous part must have been consumed on `stream`.
int PlainStream::next(const uint8_t** d, uint64_t* len) {
    if (holding) {                                              // the part handed out last time has been counted: its tail becomes the carry
        hipLaunchKernelGGL(k_make_carry, dim3(1), dim3(Ingest::HDR), 0, stream, cur_head, Ingest::HDR + cur_plain, (uint32_t)(k - 1), d_carry);
        HIPCHK(hipGetLastError());
        if (raw_held) { const int st = in.release(stream); if (st) return st; raw_held = false; }
        holding = false;
    }
    Ingest::Part p;
    const int st = in.next(&p, stream);
    if (st <= 0) return st;
    raw_held = true;
    uint8_t* head;
    uint64_t plain_len = p.len;
    if (format == AIX_FMT_PLAIN) {
        head = p.d - Ingest::HDR;
    } else {
        head = d_plain;
        const double tn = now_s();
        const hipError_t e = normalise_device_part(p.d, p.len, format, fasta_mode, d_plain + Ingest::HDR, &plain_len, &norm_state, p.last ? 1 : 0, stream);
        seconds_normalise += now_s() - tn;
        if (e != hipSuccess) { set_last_error(std::string("normalise part: ") + hipGetErrorString(e)); return AIX_ERR_HIP; }
        const int r = in.release(stream);                       // the raw bytes have been read (the normaliser synchronised the stream)
        if (r) return r;
        raw_held = false;
    }
    HIPCHK(hipMemcpyAsync(head, d_carry, Ingest::HDR, hipMemcpyDeviceToDevice, stream));
    cur_head = head;
    cur_plain = plain_len;
    holding = true;
    plain_total += plain_len;
    *d = head;
    *len = Ingest::HDR + plain_len;
    return 1;
}

PlainStream::~PlainStream() {
    (void)hipStreamSynchronize(stream);
    if (d_carry) (void)hipFree(d_carry);
    if (d_plain) (void)hipFree(d_plain);
}

void PlainStream::fill_stats(aix_ingest_stats_t* st) const {
    if (!st) return;
    st->bytes_in = in.bytes_in;
    st->plain_bytes = plain_total;
    st->parts = in.parts;
    st->pieces = 0;
    st->part_bytes = in.part_bytes;
    st->pinned_bytes = in.pinned_bytes;
    st->device_bytes = in.device_bytes + plain_dev_bytes;
    st->seconds_read = in.seconds_read;
    st->seconds_wait = in.seconds_wait;
    st->seconds_h2d = in.seconds_h2d;
    st->seconds_normalise = seconds_normalise;
}

}  // namespace aix

// ---------------------------------------------------------------------------------------------
// C ABI: files in, tables out
// ---------------------------------------------------------------------------------------------
static int count13_source(aix_index_t* h, const ByteSource& src, int format, uint64_t* d_tf, hipStream_t s, aix_ingest_stats_t* stats) {
    std::lock_guard<std::mutex> lk(h->count_mutex);
    int st = count13_begin_locked(h, d_tf, s);
    if (st) return st;
    {
        PlainStream ps(src, format, 0, 13, h->device, s);
        st = ps.start();
        const uint8_t* d = nullptr;
        uint64_t len = 0;
        int r = 0;
        while (!st && (r = ps.next(&d, &len)) == 1) {
            const double t0 = now_s();
            st = count13_add_locked(h, (const char*)d, len, d_tf, s);
            if (stats) stats->seconds_compute += now_s() - t0;
        }
        if (!st && r < 0) st = r;
        ps.fill_stats(stats);
        if (stats) stats->workspace_bytes = h->work13_bytes;
    }
    const int st2 = count13_end_locked(h, d_tf, s);
    return st ? st : st2;
}

static int count23_source(aix_index_t* h, const ByteSource& src, int format, int canon_mode, uint32_t* d_tf, hipStream_t s, aix_ingest_stats_t* stats) {
    PlainStream ps(src, format, 1, 23, h->device, s);
    int st = ps.start();
    const uint8_t* d = nullptr;
    uint64_t len = 0;
    int r = 0;
    while (!st && (r = ps.next(&d, &len)) == 1) {
        const double t0 = now_s();
        st = aix_count23_fixed_dev(h, (const char*)d, len, canon_mode, d_tf, s);
        if (!st && hipStreamSynchronize(s) != hipSuccess) st = AIX_ERR_HIP;        // the atomics back end of short parts returns without waiting
        if (stats) stats->seconds_compute += now_s() - t0;
    }
    if (!st && r < 0) st = r;
    ps.fill_stats(stats);
    if (stats) stats->workspace_bytes = h->work13_bytes;
    return st;
}

// The output file of a tool run is created, mapped and its pages allocated WHILE the input is being counted (tmpfs / the page cache allocate
// and zero 512 MiB at a few GB/s, and write() calls on one file are serialised by the inode lock: written afterwards with pwrite the file cost
// a third of the whole call). A background thread faults the mapping in; the result is then copied into it by the host threads.
struct OutputFile {
    int fd = -1;
    uint8_t* map = nullptr;
    uint64_t bytes = 0;
    std::thread th;
    int open_and_reserve(const char* path, uint64_t n) {
        fd = ::open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) return AIX_ERR_IO;
        bytes = n;
        if (n && ftruncate(fd, (off_t)n) != 0) { ::close(fd); fd = -1; return AIX_ERR_IO; }
        if (n) {
            void* m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (m != MAP_FAILED) map = (uint8_t*)m;               // no mapping (an odd file system): the pwrite path below
        }
        uint8_t* mp = map;
        const int f = fd;
        th = std::thread([mp, f, n] {
            if (mp) {
#ifdef MADV_POPULATE_WRITE
                if (madvise(mp, n, MADV_POPULATE_WRITE) == 0) return;
#endif
                for (uint64_t o = 0; o < n; o += 4096) ((volatile uint8_t*)mp)[o] = 0;
            } else if (n) {
                (void)posix_fallocate(f, 0, (off_t)n);
            }
        });
        return AIX_OK;
    }
    void ready() { if (th.joinable()) th.join(); }
    // file[lo, lo + m) = from[0, m), host threads side by side
    int put(uint64_t lo, const char* from, uint64_t m) {
        if (map) {
            host_workers().sliced(m, [&](uint64_t a, uint64_t b) { memcpy(map + lo + a, from + a, b - a); });
            return AIX_OK;
        }
        std::atomic<int> bad{0};
        const int f = fd;
        host_workers().sliced(m, [&](uint64_t a, uint64_t b) {
            while (a < b) {
                const ssize_t r = pwrite(f, from + a, b - a, (off_t)(lo + a));
                if (r <= 0) { bad = 1; return; }
                a += (uint64_t)r;
            }
        });
        return bad ? AIX_ERR_IO : AIX_OK;
    }
    int close_file() {
        ready();
        int st = AIX_OK;
        if (map && munmap(map, bytes) != 0) st = AIX_ERR_IO;
        map = nullptr;
        if (fd >= 0 && ::close(fd) != 0) st = AIX_ERR_IO;
        fd = -1;
        return st;
    }
    ~OutputFile() { ready(); if (map) munmap(map, bytes); if (fd >= 0) ::close(fd); }
};

// The result of a counting call leaves in 32 MiB slices through two pooled pinned blocks: slice i is written to the file (pwrite, host
// threads) and / or copied to the caller's memory (host threads) while slice i + 1 crosses the link. A destination the caller pinned takes
// one direct copy.
static int download(void* host_dst, OutputFile* of, const void* d_src, uint64_t bytes, hipStream_t s, aix_ingest_stats_t* stats) {
    const double t0 = now_s();
    int st = AIX_OK;
    bool dst_pinned = false;
    if (host_dst) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, host_dst) == hipSuccess) dst_pinned = (a.type == hipMemoryTypeHost); else (void)hipGetLastError();
    }
    if (host_dst && dst_pinned && bytes) {
        if (hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) st = AIX_ERR_HIP;
    }
    const bool staged = bytes && (of || (host_dst && !dst_pinned));
    if (!st && staged) {
        const uint64_t slice = std::min<uint64_t>(bytes, 32ull << 20);
        void* pin[2] = {pinned_pool().get(slice), pinned_pool().get(slice)};
        hipEvent_t ev[2] = {nullptr, nullptr};
        if (!pin[0] || !pin[1]) st = AIX_ERR_NOMEM;
        for (int b = 0; b < 2 && !st; ++b)
            if (hipEventCreateWithFlags(&ev[b], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); st = AIX_ERR_HIP; }
        const uint64_t nsl = (bytes + slice - 1) / slice;
        auto issue = [&](uint64_t i) -> int {
            const uint64_t lo = i * slice, m = std::min(slice, bytes - lo);
            if (hipMemcpyAsync(pin[i & 1], (const char*)d_src + lo, m, hipMemcpyDeviceToHost, s) != hipSuccess || hipEventRecord(ev[i & 1], s) != hipSuccess) return AIX_ERR_HIP;
            return AIX_OK;
        };
        if (!st) st = issue(0);
        if (of) of->ready();
        for (uint64_t i = 0; i < nsl && !st; ++i) {
            if (hipEventSynchronize(ev[i & 1]) != hipSuccess) { st = AIX_ERR_HIP; break; }
            if (i + 1 < nsl) { st = issue(i + 1); if (st) break; }
            const uint64_t lo = i * slice, m = std::min(slice, bytes - lo);
            const char* from = (const char*)pin[i & 1];
            char* mem = (host_dst && !dst_pinned) ? (char*)host_dst + lo : nullptr;
            if (mem) host_workers().sliced(m, [&](uint64_t a, uint64_t b) { memcpy(mem + a, from + a, b - a); });
            if (of) st = of->put(lo, from, m);
        }
        (void)hipStreamSynchronize(s);
        for (int b = 0; b < 2; ++b) { pinned_pool().put(pin[b], slice); if (ev[b]) (void)hipEventDestroy(ev[b]); }
    } else if (!st && of && host_dst && bytes) {                 // pinned destination and a file: the file is written from the caller's copy
        of->ready();
        st = of->put(0, (const char*)host_dst, bytes);
    }
    if (of) { const int c = of->close_file(); if (!st) st = c; }
    if (st == AIX_ERR_HIP) set_last_error("result download failed");
    if (stats) stats->seconds_output += now_s() - t0;
    return st;
}

// device -> caller memory through the pinned slices (any result of a host-buffer entry point); returns when the bytes are there
int download_to_host(void* host_dst, const void* d_src, uint64_t bytes, hipStream_t s) {
    if (bytes == 0) return AIX_OK;
    return download(host_dst, (OutputFile*)nullptr, d_src, bytes, s, nullptr);
}

// The binary images the tools write (.index.bin: 8 B per position, .tf.bin, .kmers.bin) through the mapped, multi-threaded writer of the
// 13-mer counter instead of one write() loop: a positions index of 5 M reads is 5 GB.
extern "C" int aix_file_write(const char* path, const void* data, uint64_t bytes) {
    if (!path || (bytes && !data)) return AIX_ERR_ARG;
    OutputFile of;
    int st = of.open_and_reserve(path, bytes);
    if (st) return st;
    of.ready();
    if (bytes) st = of.put(0, (const char*)data, bytes);
    const int c = of.close_file();
    return st ? st : c;
}

static int count13_any(aix_index_t* h, const ByteSource& src, int format, const char* out_path, uint64_t* tf_out, aix_ingest_stats_t* stats) {
    const double t0 = now_s();
    if (stats) memset(stats, 0, sizeof(*stats));
    DevGuard g(h->device);
    OutputFile of;
    if (out_path) { const int os_ = of.open_and_reserve(out_path, 8 * AIX_TOTAL_13MERS); if (os_) return os_; }
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int st;
    {
        DevBuf dout(s);
        st = dout.alloc(8 * AIX_TOTAL_13MERS) == hipSuccess ? AIX_OK : AIX_ERR_NOMEM;
        if (!st) st = count13_source(h, src, format, (uint64_t*)dout.p, s, stats);
        if (!st && hipStreamSynchronize(s) != hipSuccess) st = AIX_ERR_HIP;
        if (!st) st = download(tf_out, out_path ? &of : nullptr, dout.p, 8 * AIX_TOTAL_13MERS, s, stats);
        if (stats) stats->device_bytes += 8 * AIX_TOTAL_13MERS;
    }
    (void)hipStreamDestroy(s);
    if (stats) stats->seconds_total = now_s() - t0;
    return st;
}

extern "C" int aix_ingest_warm(int device) {
    int c = 0;
    const int st = aix_device_count(&c);
    if (st) return st;
    if (device < 0 || device >= c) return AIX_ERR_ARG;
    pinned_warm(device);
    return AIX_OK;
}

extern "C" int aix_count13_file(aix_index_t* h, const char* path, int format, const char* out_path, uint64_t* tf_out, aix_ingest_stats_t* stats) {
    if (!h || !path || (!out_path && !tf_out)) return AIX_ERR_ARG;
    if (h->k != 13) return AIX_ERR_MODE;
    ByteSource src;
    const int st = src.open_file(path);
    if (st) return st;
    return count13_any(h, src, format, out_path, tf_out, stats);
}

extern "C" int aix_count13(aix_index_t* h, const char* buf, uint64_t len, int format, uint64_t* tf_out) {
    if (!h || !tf_out || (len && !buf)) return AIX_ERR_ARG;
    if (h->k != 13) return AIX_ERR_MODE;
    DevGuard g(h->device);
    ByteSource src;
    src.set_memory(buf, len);
    return count13_any(h, src, format, nullptr, tf_out, nullptr);
}

static int count23_any(aix_index_t* h, const ByteSource& src, int format, int canon_mode, uint32_t* tf_out, aix_ingest_stats_t* stats) {
    const double t0 = now_s();
    if (stats) memset(stats, 0, sizeof(*stats));
    if (h->n == 0) return AIX_OK;
    DevGuard g(h->device);
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int st;
    {
        DevBuf dout(s);
        st = dout.alloc(4 * h->n) == hipSuccess ? AIX_OK : AIX_ERR_NOMEM;
        if (!st && hipMemsetAsync(dout.p, 0, 4 * h->n, s) != hipSuccess) st = AIX_ERR_HIP;
        if (!st) st = count23_source(h, src, format, canon_mode, (uint32_t*)dout.p, s, stats);
        if (!st) st = download(tf_out, (OutputFile*)nullptr, dout.p, 4 * h->n, s, stats);
        if (stats) stats->device_bytes += 4 * h->n;
    }
    (void)hipStreamDestroy(s);
    if (stats) stats->seconds_total = now_s() - t0;
    return st;
}

extern "C" int aix_count23_fixed_file(aix_index_t* h, const char* path, int format, int canon_mode, uint32_t* tf_out, aix_ingest_stats_t* stats) {
    if (!h || !path || !tf_out) return AIX_ERR_ARG;
    if (h->k != 23) return AIX_ERR_MODE;
    if (canon_mode < 0 || canon_mode > 2) return AIX_ERR_ARG;
    ByteSource src;
    const int st = src.open_file(path);
    if (st) return st;
    return count23_any(h, src, format, canon_mode, tf_out, stats);
}

extern "C" int aix_count23_fixed(aix_index_t* h, const char* buf, uint64_t len, int format, int canon_mode, uint32_t* tf_out) {
    if (!h || !tf_out || (len && !buf)) return AIX_ERR_ARG;
    if (h->k != 23) return AIX_ERR_MODE;
    if (canon_mode < 0 || canon_mode > 2) return AIX_ERR_ARG;
    DevGuard g(h->device);
    ByteSource src;
    src.set_memory(buf, len);
    return count23_any(h, src, format, canon_mode, tf_out, nullptr);
}

// ---------------------------------------------------------------------------------------------
// K1 from a stream: PLAIN parts are appended to a piece buffer of up to 2^31 windows; a full piece goes through the distinct-k-mer
// pipeline (aix_k1.hip) and its sorted set is merged into the accumulated one (aix_merge.hip); the last k - 1 bytes of a piece open the next.
// ---------------------------------------------------------------------------------------------
__global__ void k_move_tail(uint8_t* __restrict__ buf, uint64_t fill, uint32_t keep) {     // buf[0, keep) = buf[fill - keep, fill); keep < 64 <= fill - keep is not required: lanes read first
    const uint32_t i = threadIdx.x;
    uint8_t c = 0;
    if (i < keep) c = buf[fill - keep + i];
    __syncthreads();
    if (i < keep) buf[i] = c;
}

static int distinct_source(const ByteSource& src, int format, int k, int canon_mode, uint64_t min_count, int device, hipStream_t s, uint64_t** dk, uint64_t** dc,
                           uint64_t* n_out, aix_ingest_stats_t* stats) {
    *dk = nullptr; *dc = nullptr; *n_out = 0;
    uint64_t piece_win = 1ull << 30;                          // as distinct_from_plain (aix_merge.hip)
    if (const char* e = getenv("AIX_DISTINCT_PIECE")) { const uint64_t v = strtoull(e, nullptr, 10); if (v >= 1 && v <= (1ull << 31)) piece_win = v; }   // test hook: merges at small sizes
    PlainStream ps(src, format, 1, k, device, s);
    int st = ps.start();
    if (st) return st;
    // a file shorter than a piece needs no more room than itself (normalised output never exceeds the input by more than the final '\n')
    const uint64_t cap = std::max<uint64_t>(std::min<uint64_t>(piece_win + k - 1, src.len + k + 1), (uint64_t)(2 * k));
    DevBuf piece(s);
    HIPCHK(piece.alloc(cap + 64));
    uint8_t* pb = (uint8_t*)piece.p;
    uint64_t fill = 0;
    DistinctAcc acc(k, canon_mode, s);
    auto flush = [&]() -> int {
        const double t0 = now_s();
        const hipError_t e = acc.add_plain(pb, fill);
        if (e != hipSuccess) { set_last_error(std::string("count_distinct piece: ") + hipGetErrorString(e)); return AIX_ERR_HIP; }
        if (stats) stats->seconds_compute += now_s() - t0;
        const uint32_t keep = (uint32_t)std::min<uint64_t>(fill, (uint64_t)(k - 1));
        if (keep && fill > keep) { hipLaunchKernelGGL(k_move_tail, dim3(1), dim3(64), 0, s, pb, fill, keep); HIPCHK(hipGetLastError()); }
        fill = keep;
        return AIX_OK;
    };
    const uint8_t* d = nullptr;
    uint64_t len = 0;
    int r = 0;
    while ((r = ps.next(&d, &len)) == 1) {
        const uint8_t* p = d + Ingest::HDR;                    // the piece keeps its own carry: the stream's header is not needed here
        uint64_t left = len - Ingest::HDR;
        while (left) {
            if (fill == cap) { st = flush(); if (st) return st; }
            const uint64_t m = std::min(left, cap - fill);
            HIPCHK(hipMemcpyAsync(pb + fill, p, m, hipMemcpyDeviceToDevice, s));
            fill += m; p += m; left -= m;
        }
    }
    if (r < 0) return r;
    if (fill >= (uint64_t)k) { st = flush(); if (st) return st; }
    const hipError_t e = acc.finish(min_count ? min_count : 1, dk, dc, n_out);
    if (e != hipSuccess) { set_last_error(std::string("count_distinct: ") + hipGetErrorString(e)); return AIX_ERR_HIP; }
    ps.fill_stats(stats);
    if (stats) { stats->device_bytes += cap + 64; stats->pieces = acc.pieces; }
    return AIX_OK;
}

static int count_distinct_any(const ByteSource& src, int format, int k, int canon_mode, uint64_t min_count, int device, uint64_t** keys_out, uint64_t** counts_out,
                              uint64_t* n_out, aix_ingest_stats_t* stats) {
    const double t0 = now_s();
    if (stats) memset(stats, 0, sizeof(*stats));
    *keys_out = nullptr; *counts_out = nullptr; *n_out = 0;
    DevGuard g(device);
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint64_t *dk = nullptr, *dc = nullptr, m = 0;
    int st = distinct_source(src, format, k, canon_mode, min_count, device, s, &dk, &dc, &m, stats);
    uint64_t *hk = nullptr, *hc = nullptr;
    if (!st) {
        const double t1 = now_s();
        hk = (uint64_t*)malloc(8 * (m ? m : 1));
        hc = (uint64_t*)malloc(8 * (m ? m : 1));
        if (!hk || !hc) st = AIX_ERR_NOMEM;
        if (!st && m) {
            hipError_t e = hipMemcpyAsync(hk, dk, 8 * m, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipMemcpyAsync(hc, dc, 8 * m, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { set_last_error(std::string("count_distinct: ") + hipGetErrorString(e)); st = AIX_ERR_HIP; }
        }
        if (stats) stats->seconds_output += now_s() - t1;
    }
    (void)hipStreamSynchronize(s);
    if (dk) pool_free(dk);
    if (dc) pool_free(dc);
    (void)hipStreamDestroy(s);
    if (st) { free(hk); free(hc); return st; }
    *keys_out = hk; *counts_out = hc; *n_out = m;
    if (stats) stats->seconds_total = now_s() - t0;
    return AIX_OK;
}

extern "C" int aix_count_distinct(const char* buf, uint64_t len, int format, int k, int canon_mode, uint64_t min_count, int device, uint64_t** keys_out,
                                  uint64_t** counts_out, uint64_t* n_out) {
    if (!keys_out || !counts_out || !n_out || (len && !buf) || k < 1 || k > 31 || canon_mode < 0 || canon_mode > 2) return AIX_ERR_ARG;
    *keys_out = nullptr; *counts_out = nullptr; *n_out = 0;
    int c = 0;
    const int st = aix_device_count(&c);
    if (st) return st;
    if (device < 0 || device >= c) return AIX_ERR_ARG;
    DevGuard g(device);
    ByteSource src;
    src.set_memory(buf, len);
    return count_distinct_any(src, format, k, canon_mode, min_count, device, keys_out, counts_out, n_out, nullptr);
}

extern "C" int aix_count_distinct_file(const char* path, int format, int k, int canon_mode, uint64_t min_count, int device, uint64_t** keys_out,
                                       uint64_t** counts_out, uint64_t* n_out, aix_ingest_stats_t* stats) {
    if (!path || !keys_out || !counts_out || !n_out || k < 1 || k > 31 || canon_mode < 0 || canon_mode > 2) return AIX_ERR_ARG;
    *keys_out = nullptr; *counts_out = nullptr; *n_out = 0;
    int c = 0;
    int st = aix_device_count(&c);
    if (st) return st;
    if (device < 0 || device >= c) return AIX_ERR_ARG;
    ByteSource src;
    st = src.open_file(path);
    if (st) return st;
    return count_distinct_any(src, format, k, canon_mode, min_count, device, keys_out, counts_out, n_out, stats);
}
