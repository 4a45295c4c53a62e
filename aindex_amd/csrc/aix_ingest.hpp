// aix_ingest.hpp — private: the streaming-ingestion pipeline behind the file / host-buffer entry points of the counters (aix_ingest.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "../../include/aindex_hip.h"

namespace aix {

// where the bytes come from: a regular file (pread) or caller memory (memcpy; as it is when the caller pinned it)
struct ByteSource {
    int fd = -1;
    const uint8_t* mem = nullptr;
    uint64_t len = 0;
    bool mem_pinned = false;
    ByteSource() = default;
    ByteSource(const ByteSource&) = delete;
    ByteSource& operator=(const ByteSource&) = delete;
    ~ByteSource();
    int open_file(const char* path);                         // AIX_ERR_IO
    void set_memory(const void* p, uint64_t n);
    int read(uint64_t off, void* dst, uint64_t n) const;    // dst[0, n) = source[off, off + n), sliced over the host worker threads
};

int write_file_parallel(const char* path, const void* src, uint64_t bytes);

// Producer of the pipeline: a thread that reads part after part into pinned staging and puts it on the wire (its own copy stream).
// Ring mode: NB device buffers, each with HDR writable bytes in front of the part. Direct mode (start(d_dst)): parts land at d_dst + offset.
class Ingest {
public:
    static constexpr int NB = 3;
    static constexpr uint64_t HDR = 64;
    struct Part { uint8_t* d; uint64_t len; uint64_t file_off; bool last; };

    Ingest(const ByteSource& s, uint64_t part, int dev);
    ~Ingest();
    Ingest(const Ingest&) = delete;
    Ingest& operator=(const Ingest&) = delete;
    int start(uint8_t* direct_dst);
    int next(Part* out, hipStream_t consumer);               // 1: *out is valid once `consumer` reaches this point; 0: end of input; < 0: error
    int release(hipStream_t consumer);                       // the part handed out last is free once `consumer` reaches this point
    int drain(hipStream_t consumer);

    const ByteSource& src;
    const uint64_t part_bytes;
    const int device;
    // statistics (read after the last next())
    uint64_t bytes_in = 0, parts = 0, pinned_bytes = 0, device_bytes = 0;
    double seconds_read = 0, seconds_wait = 0, seconds_h2d = 0;

private:
    void produce();
    void fail(int st);
    uint8_t* direct = nullptr;
    void* pin[NB] = {};
    uint8_t* dbuf[NB] = {};
    hipEvent_t h2d_ev[NB] = {}, h2d_t0[NB] = {}, used_ev[NB] = {};
    hipStream_t copy_stream = nullptr;
    Part meta[NB] = {};
    std::thread producer;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t produced = 0, handed = 0, released = 0;
    bool done = false, abort_flag = false, started = false;
    int error = 0;
};

uint64_t ingest_part_bytes();
void pinned_trim();
void pinned_warm(int device);
int upload_pipelined(const ByteSource& src, uint8_t* d_dst, int device, hipStream_t s);
}  // namespace aix
// device -> caller memory: 32 MiB slices through two pooled pinned blocks, copied out by the host threads while the next slice is on the link
// (a pageable destination handed to hipMemcpy is staged by the runtime at a fraction of the link rate); a pinned destination takes one copy
int download_to_host(void* host_dst, const void* d_src, uint64_t bytes, hipStream_t s);
namespace aix {

// Consumer side: the source as a sequence of PLAIN parts with the carry in front (see aix_ingest.hip)
class PlainStream {
public:
    PlainStream(const ByteSource& s, int format, int fasta_mode, int k, int device, hipStream_t stream);
    ~PlainStream();
    int start();
    int next(const uint8_t** d, uint64_t* len);
    void fill_stats(aix_ingest_stats_t* st) const;
    uint64_t part_bytes() const { return in.part_bytes; }
    int resolved_format() const { return format; }

private:
    const ByteSource& src;
    int format, fasta_mode, k;
    hipStream_t stream;
    Ingest in;
    uint8_t* d_carry = nullptr;
    uint8_t* d_plain = nullptr;
    uint64_t plain_dev_bytes = 0, plain_total = 0;
    double seconds_normalise = 0;
    uint32_t norm_state = 0xFFFFFFFFu;                       // AIX_NORM_START
    uint8_t* cur_head = nullptr;
    uint64_t cur_plain = 0;
    bool holding = false, raw_held = false;
};

}  // namespace aix
