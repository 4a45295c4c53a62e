// aix_reads.hip — host-side text formats on either side of the device path (no device work, no GPU needed): compute_reads below, and at
// the end of the file the readers / writers of the tools' text files (.dat of compute_index, the keys file of compute_mphf_seq, the
// k-mer list of kmer_counter), which the Python front ends used to walk line by line.
// N4 row: the step before counting (src/compute_reads.cpp:20-216). FASTQ (paired or single), FASTA or a
// plain reads file -> <prefix>.reads (one record per line; a pair as R1~revcomp(R2)), <prefix>.ridx ("rid\tstart\tend" per record) and,
// for FASTA, <prefix>.header ("name\tstart\tlength"). Text reformatting bound by file I/O: no device work, no GPU needed; the inputs are
// memory-mapped and walked once, the outputs leave through large buffers. Line rules are std::getline's: lines end at '\n', a final
// line without one still counts, no line follows a trailing '\n'.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/aindex_hip.h"

namespace {

struct Mapped {                                   // a whole file, read-only
    const char* p = nullptr;
    size_t n = 0;
    bool open(const char* path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); return false; }
        n = (size_t)st.st_size;
        if (n) {
            void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); n = 0; return false; }
            (void)madvise(m, n, MADV_SEQUENTIAL);
            p = (const char*)m;
        }
        ::close(fd);
        return true;
    }
    ~Mapped() { if (p) munmap((void*)p, n); }
};

struct Line { const char* s = ""; size_t len = 0; };          // a std::string variable of the reference's loops, as a view into the mapped file

// std::getline on one input file, including what it does to its string argument when it FAILS, because the reference's loops go on
// using the variable: the first failing call on a stream that is still good() (everything consumed, the file ended with '\n' or is
// empty) erases the string; once eofbit is set — the last line had no '\n' — or after a failure, the string is left as it was.
struct Lines {
    const char *cur, *end;
    bool eofbit = false, failed = false;
    explicit Lines(const Mapped& m) : cur(m.p), end(m.p + m.n) {}
    bool getline(Line& v) {
        if (!eofbit && !failed && cur != nullptr && cur < end) {
            const char* nl = (const char*)memchr(cur, '\n', (size_t)(end - cur));
            v.s = cur;
            v.len = nl ? (size_t)(nl - cur) : (size_t)(end - cur);
            cur = nl ? nl + 1 : end;
            if (!nl) eofbit = true;                // the delimiter was the end of the file
            return true;
        }
        if (!eofbit && !failed) { v.s = ""; v.len = 0; }          // good() stream, nothing left: erased, then eofbit | failbit
        eofbit = failed = true;
        return false;
    }
};

struct Out {                                      // buffered writer; `bad` is sticky
    FILE* f = nullptr;
    std::string buf;
    bool bad = false;
    bool open(const std::string& path) { f = fopen(path.c_str(), "wb"); buf.reserve(1u << 23); return f != nullptr; }
    void flush() { if (f && !buf.empty()) { if (fwrite(buf.data(), 1, buf.size(), f) != buf.size()) bad = true; buf.clear(); } }
    void put(const char* s, size_t n) { buf.append(s, n); if (buf.size() >= (1u << 23) - 4096) flush(); }
    void put(char c) { buf.push_back(c); }
    void num(uint64_t v) { char t[24]; int i = 24; do { t[--i] = (char)('0' + v % 10); v /= 10; } while (v); buf.append(t + i, (size_t)(24 - i)); }
    bool close() { flush(); if (f) { if (fclose(f) != 0) bad = true; f = nullptr; } return !bad; }
    ~Out() { if (f) fclose(f); }
};

// get_revcomp(const std::string&) of the reference (kmers.cpp:310-330): reversed, A<->T, C<->G, anything else 'N'
void put_revcomp(Out& o, const char* s, size_t n) {
    static const struct Tab { char t[256]; Tab() { memset(t, 'N', sizeof t); t[(unsigned char)'A'] = 'T'; t[(unsigned char)'C'] = 'G'; t[(unsigned char)'G'] = 'C'; t[(unsigned char)'T'] = 'A'; } } tab;
    char tmp[4096];
    while (n) {
        const size_t m = n < sizeof tmp ? n : sizeof tmp;
        for (size_t i = 0; i < m; ++i) tmp[i] = tab.t[(unsigned char)s[n - 1 - i]];
        o.put(tmp, m);
        n -= m;
    }
}

void ridx_line(Out& o, uint64_t rid, uint64_t start, uint64_t end) {
    o.num(rid); o.put('\t'); o.num(start); o.put('\t'); o.num(end); o.put('\n');
    if (o.buf.size() >= (1u << 22)) o.flush();
}

}  // namespace

extern "C" int aix_compute_reads(const char* file1, const char* file2, const char* mode, const char* prefix) {
    if (!file1 || !mode || !prefix) return AIX_ERR_ARG;
    const std::string m(mode), pre(prefix);
    const bool pe = m == "fastq", se = m == "se", plain = m == "reads", fasta = m == "fasta";
    if (!pe && !se && !plain && !fasta) return AIX_ERR_ARG;
    Mapped a, b;
    if (!a.open(file1)) return AIX_ERR_IO;
    if (pe && (!file2 || !b.open(file2))) return AIX_ERR_IO;
    Out reads, ridx, header;
    if (!plain && !reads.open(pre + ".reads")) return AIX_ERR_IO;
    if (!ridx.open(pre + ".ridx")) return AIX_ERR_IO;
    if (fasta && !header.open(pre + ".header")) return AIX_ERR_IO;
    uint64_t n = 0, start = 0;
    Line line1, line2;                            // the reference's two std::string variables
    if (pe) {                                     // :77-116: four lines per record in both files, the second one is the sequence
        Lines A(a), B(b);
        while (A.getline(line1)) {
            A.getline(line1);
            B.getline(line2);
            B.getline(line2);
            const uint64_t end = start + line1.len + line2.len + 1;      // + the '~' between the mates
            reads.put(line1.s, line1.len); reads.put('~'); put_revcomp(reads, line2.s, line2.len); reads.put('\n');
            ridx_line(ridx, n, start, end);
            start = end + 1;                                              // + the newline
            ++n;
            A.getline(line1); A.getline(line1); B.getline(line2); B.getline(line2);
        }
    } else if (se) {                              // :118-147
        Lines A(a);
        while (A.getline(line1)) {
            A.getline(line1);
            const uint64_t end = start + line1.len;
            reads.put(line1.s, line1.len); reads.put('\n');
            ridx_line(ridx, n, start, end);
            start = end + 1;
            ++n;
            A.getline(line1); A.getline(line1);
        }
    } else if (plain) {                           // :149-168: the index of a reads file that exists already
        Lines A(a);
        while (A.getline(line1)) {
            const uint64_t end = start + line1.len;
            ridx_line(ridx, n, start, end);
            start = end + 1;
            ++n;
        }
    } else {                                      // :170-213: a record = the lines between two '>' lines, joined
        Lines A(a);
        std::string name;
        uint64_t cur = 0;                         // length of the sequence gathered so far (its bytes are in `reads` already)
        auto finish = [&]() {
            const uint64_t end = start + cur;
            reads.put('\n');
            ridx_line(ridx, n, start, end);
            header.put(name.data(), name.size()); header.put('\t'); header.num(start); header.put('\t'); header.num(cur); header.put('\n');
            if (header.buf.size() >= (1u << 22)) header.flush();
            start = end + 1;
            ++n;
            cur = 0;
        };
        while (A.getline(line1)) {
            if (line1.len && line1.s[0] == '>') {  // (an empty line is no header: the reference reads its terminating NUL there)
                if (cur) finish();
                name.assign(line1.s + 1, line1.len - 1);
                continue;
            }
            reads.put(line1.s, line1.len);
            cur += line1.len;
        }
        if (cur) finish();
    }
    const bool ok_reads = plain || reads.close(), ok_ridx = ridx.close(), ok_header = !fasta || header.close();      // every file is closed
    return ok_reads && ok_ridx && ok_header ? AIX_OK : AIX_ERR_IO;
}

// ---------------------------------------------------------------------------------------------
// the tools' text files
// ---------------------------------------------------------------------------------------------
// .dat of compute_index ("kmer<ws>tf" per line; worker_for_fill_index, src/hash.cpp:681-702: `is >> kmer >> tf`, a missing or unreadable
// count reads as 0, one beyond u32 as its maximum; mock: only the k-mer is read). Every k-mer must be 23 characters (AIX_ERR_FORMAT);
// empty lines are skipped. *keys_out = n * 23 bytes, *tf_out = n counts (null when mock); both malloc'd (aix_free).
extern "C" int aix_dat_load(const char* path, int mock, uint64_t* n_out, char** keys_out, uint32_t** tf_out) {
    if (!path || !n_out || !keys_out || (!mock && !tf_out)) return AIX_ERR_ARG;
    *n_out = 0; *keys_out = nullptr;
    if (tf_out) *tf_out = nullptr;
    Mapped a;
    if (!a.open(path)) return AIX_ERR_IO;
    auto is_ws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; };
    uint64_t n = 0;
    for (int pass = 0; pass < 2; ++pass) {
        Lines L(a);
        Line ln;
        uint64_t i = 0;
        while (L.getline(ln)) {
            const char *p = ln.s, *e = ln.s + ln.len;
            while (p < e && is_ws(*p)) ++p;
            if (p == e) continue;                                         // empty line
            const char* k0 = p;
            while (p < e && !is_ws(*p)) ++p;
            if (pass == 1) {
                if (p - k0 != 23) { free(*keys_out); *keys_out = nullptr; if (tf_out) { free(*tf_out); *tf_out = nullptr; } return AIX_ERR_FORMAT; }
                memcpy(*keys_out + 23 * i, k0, 23);
                if (!mock) {
                    while (p < e && is_ws(*p)) ++p;
                    uint64_t v = 0;
                    bool over = false;
                    while (p < e && *p >= '0' && *p <= '9') { v = v * 10 + (uint64_t)(*p - '0'); if (v > 0xFFFFFFFFull) over = true, v = 0xFFFFFFFFull; ++p; }
                    (*tf_out)[i] = over ? 0xFFFFFFFFu : (uint32_t)v;
                }
            }
            ++i;
        }
        if (pass == 0) {
            n = i;
            *keys_out = (char*)malloc(n ? 23 * n : 1);
            if (!*keys_out) return AIX_ERR_NOMEM;
            if (!mock) { *tf_out = (uint32_t*)malloc(n ? 4 * n : 4); if (!*tf_out) { free(*keys_out); *keys_out = nullptr; return AIX_ERR_NOMEM; } }
        }
    }
    *n_out = n;
    return AIX_OK;
}

// compute_mphf_seq <keys.txt> (compute_mphf_generic.hpp:21-30): one key per line, any lengths -> the .pf image of aix_pf_build_ragged
extern "C" int aix_pf_build_file(const char* keys_path, void** pf_out, uint64_t* pf_len) {
    if (!keys_path || !pf_out || !pf_len) return AIX_ERR_ARG;
    Mapped a;
    if (!a.open(keys_path)) return AIX_ERR_IO;
    std::string data;
    std::vector<uint64_t> offs;
    try {
        data.reserve(a.n);
        offs.push_back(0);
        Lines L(a);
        Line ln;
        while (L.getline(ln)) { data.append(ln.s, ln.len); offs.push_back(data.size()); }
    } catch (const std::bad_alloc&) { return AIX_ERR_NOMEM; }
    return aix_pf_build_ragged(data.data(), offs.data(), offs.size() - 1, pf_out, pf_len);
}

// the k-mer list kmer_counter writes (count_kmers.cpp:362-382): "KMER\tcount\n" per entry, in the order given (the caller sorts);
// keys are 2-bit codes of k bases, first base most significant
extern "C" int aix_kmers_write_text(const char* path, const uint64_t* keys, const uint64_t* counts, uint64_t n, int k) {
    if (!path || (n && (!keys || !counts)) || k < 1 || k > 32) return AIX_ERR_ARG;
    Out o;
    if (!o.open(path)) return AIX_ERR_IO;
    char km[33];
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t c = keys[i];
        for (int j = 0; j < k; ++j) km[j] = "ACGT"[(c >> (2 * (k - 1 - j))) & 3];
        o.put(km, (size_t)k); o.put('\t'); o.num(counts[i]); o.put('\n');
    }
    return o.close() ? AIX_OK : AIX_ERR_IO;
}

// the .ridx file compute_reads writes and load_reads_index reads back (python_wrapper.cpp:261-279): three unsigned decimals per line,
// blank separated. *out = 3 * n values (rid, start, end per read), malloc'd (aix_free). A line with fewer than three numbers ends the
// file like the reference's `while (fin >> rid >> start >> end)`.
extern "C" int aix_ridx_load(const char* path, uint64_t* n_out, uint64_t** out) {
    if (!path || !n_out || !out) return AIX_ERR_ARG;
    *n_out = 0; *out = nullptr;
    Mapped a;
    if (!a.open(path)) return AIX_ERR_IO;
    // `fin >> a >> b >> c` does not care about line structure: numbers separated by any white space, stop at the first thing that is no number
    uint64_t cap = a.n / 6 + 3, m = 0;                                  // a number and its separator take at least two bytes
    uint64_t* v = (uint64_t*)malloc(8 * cap);
    if (!v) return AIX_ERR_NOMEM;
    const char *p = a.p, *e = a.p + a.n;
    while (p && p < e) {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\v' || *p == '\f')) ++p;
        if (p == e || *p < '0' || *p > '9') break;
        uint64_t x = 0;
        while (p < e && *p >= '0' && *p <= '9') { x = x * 10 + (uint64_t)(*p - '0'); ++p; }
        if (m == cap) { cap *= 2; uint64_t* w = (uint64_t*)realloc(v, 8 * cap); if (!w) { free(v); return AIX_ERR_NOMEM; } v = w; }
        v[m++] = x;
    }
    *n_out = m / 3;                                                      // an incomplete last triple is not a read
    *out = v;
    return AIX_OK;
}
