// BGZF / BAM / BAI reader + writer and a SAM text parser (CPU decode stage).  See bamio.h.
#include "bamio.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <new>
#include <sstream>
#include <system_error>
#include <thread>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>

#include "../../include/bamsignals_abi.h"
#include "host_util.h"

namespace bsig {

void *column_alloc(size_t bytes)
{
    if (bytes == 0) bytes = 1;
    if (bytes >= (4u << 20)) {
        void *p = nullptr;
        const size_t rounded = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        if (posix_memalign(&p, 2u << 20, rounded) != 0) throw std::bad_alloc();
        madvise(p, rounded, MADV_HUGEPAGE);
        return p;
    }
    void *p = malloc(bytes);
    if (!p) throw std::bad_alloc();
    return p;
}

void column_free(void *p) { free(p); }

// stage timers of the last whole-file decode on this thread: block scan, waiting for inflate,
// boundary scan (serial), column extraction (parallel), total
thread_local double g_decode_timing[6] = {0, 0, 0, 0, 0, 0};
static inline double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------
// raw DEFLATE codec: libdeflate through dlopen when present, zlib otherwise
// ---------------------------------------------------------------------------------------------
namespace {

struct LibDeflate {
    void *h = nullptr;
    void *(*alloc_d)() = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*free_d)(void *) = nullptr;
    void *(*alloc_c)(int) = nullptr;
    size_t (*compress)(void *, const void *, size_t, void *, size_t) = nullptr;
    void (*free_c)(void *) = nullptr;
    uint32_t (*crc32)(uint32_t, const void *, size_t) = nullptr;
    bool ok = false;
    LibDeflate()
    {
        if (getenv("BAMSIGNALS_NO_LIBDEFLATE")) return;
        for (const char *n : {"libdeflate.so.0", "libdeflate.so"}) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return;
        alloc_d = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        free_d = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        alloc_c = (void *(*)(int))dlsym(h, "libdeflate_alloc_compressor");
        compress = (size_t (*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_deflate_compress");
        free_c = (void (*)(void *))dlsym(h, "libdeflate_free_compressor");
        crc32 = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        ok = alloc_d && decompress && free_d && alloc_c && compress && free_c && crc32;
    }
};

const LibDeflate &ld()
{
    static LibDeflate l;
    return l;
}

// per-thread inflater / deflater
struct Inflater {
    void *d = nullptr;
    Inflater() { if (ld().ok) d = ld().alloc_d(); }
    ~Inflater() { if (d) ld().free_d(d); }
    bool run(const uint8_t *in, size_t n_in, uint8_t *out, size_t n_out)
    {
        if (n_out == 0) return true;
        if (d) {
            size_t actual = 0;
            return ld().decompress(d, in, n_in, out, n_out, &actual) == 0 && actual == n_out;
        }
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return false;
        zs.next_in = const_cast<Bytef *>(in); zs.avail_in = (uInt)n_in;
        zs.next_out = out; zs.avail_out = (uInt)n_out;
        const int rc = inflate(&zs, Z_FINISH);
        const bool ok = rc == Z_STREAM_END && zs.total_out == n_out;
        inflateEnd(&zs);
        return ok;
    }
};

struct Deflater {
    void *c = nullptr;
    int level;
    explicit Deflater(int lvl) : level(lvl) { if (ld().ok) c = ld().alloc_c(lvl); }
    ~Deflater() { if (c) ld().free_c(c); }
    // returns compressed size, 0 on failure
    size_t run(const uint8_t *in, size_t n_in, uint8_t *out, size_t cap)
    {
        if (c) return ld().compress(c, in, n_in, out, cap);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return 0;
        zs.next_in = const_cast<Bytef *>(in); zs.avail_in = (uInt)n_in;
        zs.next_out = out; zs.avail_out = (uInt)cap;
        const int rc = deflate(&zs, Z_FINISH);
        const size_t n = rc == Z_STREAM_END ? zs.total_out : 0;
        deflateEnd(&zs);
        return n;
    }
};

uint32_t crc32_of(const uint8_t *p, size_t n);
}  // namespace

bool crc_check_enabled()
{
    static const bool on = getenv("BAMSIGNALS_NO_CRC") == nullptr;
    return on;
}

const uint32_t *crc32_slice8_tables()
{
    static uint32_t t[8][256];
    static const bool ready = [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 8; ++k) t[k][i] = t[0][t[k - 1][i] & 0xFFu] ^ (t[k - 1][i] >> 8);
        return true;
    }();
    (void)ready;
    return &t[0][0];
}

namespace {

// inflate one block and compare the CRC32 of what came out with the block's trailer (as htslib does)
// 0 ok, 1 inflate failed, 2 CRC mismatch
template <typename Inf>
int inflate_checked(Inf &inf, const uint8_t *file, const BgzfBlock &b, uint8_t *dst)
{
    if (!inf.run(file + b.coff + b.doff, b.dlen, dst, b.isize)) return 1;
    if (crc_check_enabled() && crc32_of(dst, b.isize) != b.crc) return 2;
    return 0;
}

uint32_t crc32_of(const uint8_t *p, size_t n)
{
    if (ld().ok) return ld().crc32(0, p, n);
    return (uint32_t)::crc32(::crc32(0L, Z_NULL, 0), p, (uInt)n);
}

inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline int32_t rdi32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

// CPUs this process may actually use: the hardware threads, cut down by the affinity mask and by a cgroup
// CPU quota (containers on a shared host: 256 hardware threads visible, 16 CPUs' worth of quota).  A pool
// sized by the hardware count alone burns such a quota in a fraction of every scheduler period and is
// then frozen for the rest of it -- the decode's copy stage was measured 0.04-0.27 s from box to box.
int eff_cpus()
{
    static const int n = [] {
        unsigned hc = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) {
            const int c = CPU_COUNT(&set);
            if (c > 0) hc = std::min(hc, (unsigned)c);
        }
        auto quota = [](const char *path_quota, const char *path_period) -> double {
            FILE *f = fopen(path_quota, "r");
            if (!f) return 0;
            char a[64] = "", b[64] = "";
            const int got = fscanf(f, "%63s %63s", a, b);
            fclose(f);
            if (got < 1 || !strcmp(a, "max")) return 0;
            double q = atof(a), per = got >= 2 ? atof(b) : 0;
            if (path_period) {
                FILE *g = fopen(path_period, "r");
                if (!g) return 0;
                if (fscanf(g, "%63s", b) == 1) per = atof(b);
                fclose(g);
            }
            return q > 0 && per > 0 ? q / per : 0;
        };
        double q = quota("/sys/fs/cgroup/cpu.max", nullptr);                                           // cgroup v2
        if (q <= 0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");   // v1
        if (q > 0) hc = std::min(hc, (unsigned)std::max(1.0, q + 0.5));
        return (int)hc;
    }();
    return n;
}

int n_threads(int t)
{
    if (t > 0) return t;
    if (const char *e = getenv("BAMSIGNALS_THREADS")) { const int v = atoi(e); if (v > 0) return v; }
    return std::max(1, std::min(eff_cpus(), 32));
}

// A small persistent worker pool: the decode runs dozens of short parallel sections per file and
// spawning threads for each of them costs more than the sections themselves.
class Pool {
public:
    static Pool &get()
    {
        static Pool p;
        return p;
    }
    // runs body(i, slot) for i in [0, n) on up to `threads` workers (the caller is one of them)
    template <typename F>
    void run(int64_t n, int threads, F &&body)
    {
        if (n <= 0) return;
        threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n));
        if (threads == 1) { for (int64_t i = 0; i < n; ++i) body(i, 0); return; }
        // two sections may run at once (inflate of the next batch beside the parse of this one):
        // keep enough workers for the tasks already queued or running plus the new ones
        ensure(outstanding_.load() + threads - 1);
        struct Job {
            std::atomic<int64_t> next{0};
            std::atomic<int> pending{0};
            int64_t n;
            std::function<void(int64_t, int)> fn;
            std::mutex m;
            std::condition_variable done;
        };
        auto job = std::make_shared<Job>();
        job->n = n;
        job->fn = body;
        job->pending = threads - 1;
        auto work = [job](int slot) {
            for (;;) {
                const int64_t i = job->next.fetch_add(1);
                if (i >= job->n) break;
                job->fn(i, slot);
            }
        };
        {
            std::lock_guard<std::mutex> lk(m_);
            outstanding_ += threads - 1;
            for (int t = 1; t < threads; ++t)
                q_.emplace_back([this, job, work, t] {
                    work(t);
                    --outstanding_;
                    if (job->pending.fetch_sub(1) == 1) { std::lock_guard<std::mutex> l2(job->m); job->done.notify_all(); }
                });
        }
        cv_.notify_all();
        work(0);
        std::unique_lock<std::mutex> lk(job->m);
        job->done.wait(lk, [&] { return job->pending.load() == 0; });
    }

private:
    Pool() = default;
    ~Pool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void ensure(int workers)
    {
        std::lock_guard<std::mutex> lk(m_);
        workers = std::min(workers, 512);
        while ((int)th_.size() < workers)
            th_.emplace_back([this] {
                for (;;) {
                    std::function<void()> task;
                    {
                        std::unique_lock<std::mutex> lk(m_);
                        cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                        if (stop_ && q_.empty()) return;
                        task = std::move(q_.front());
                        q_.pop_front();
                    }
                    task();
                }
            });
    }
    std::vector<std::thread> th_;
    std::deque<std::function<void()>> q_;
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<int> outstanding_{0};
    bool stop_ = false;
};

template <typename F>
void parallel_for(int64_t n, int threads, F &&body)
{
    Pool::get().run(n, threads, std::forward<F>(body));
}

// read-only memory map of a file
struct MappedFile {
    const uint8_t *data = nullptr;
    size_t size = 0;
    int fd = -1;
    int open(const std::string &path)
    {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return -1;
        struct stat st;
        if (fstat(fd, &st) != 0) return -1;
        size = (size_t)st.st_size;
        if (size == 0) { data = nullptr; return 0; }
        int flags = MAP_PRIVATE;
        if (getenv("BAMSIGNALS_MMAP_POPULATE")) flags |= MAP_POPULATE;
        void *p = mmap(nullptr, size, PROT_READ, flags, fd, 0);
        if (p == MAP_FAILED) return -1;
        data = (const uint8_t *)p;
        madvise(p, size, MADV_SEQUENTIAL);
        return 0;
    }
    ~MappedFile()
    {
        // Tearing down the page tables of a large populated mapping takes as long as filling them (70 ms for
        // the north star's 3-GB file, measured as the difference between the call and its decode stages):
        // nobody waits for that -- the library's unmap thread does it (Unmapper below: ONE thread, joined when
        // the library is unloaded; a detached thread could still be running this code when R's dyn.unload or a
        // dlclose takes the text away).
        void *p = (void *)data;
        const size_t n = size;
        const int f = fd;
        if (p && n >= ((size_t)64 << 20) && unmap_later(p, n, f)) return;
        if (p) munmap(p, n);
        if (f >= 0) ::close(f);
    }
    static bool unmap_later(void *p, size_t n, int f);
};

// the mappings handed over for unmapping, and the one thread that works them off; its destructor (static: run at
// exit and when the shared object is unloaded) finishes the queue and joins the thread
struct Unmapper {
    struct Job { void *p; size_t n; int fd; };
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Job> jobs;
    bool quit = false, started = false;
    std::thread th;
    bool push(void *p, size_t n, int fd)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (quit) return false;
        if (!started) {
            try { th = std::thread([this] { run(); }); } catch (const std::system_error &) { return false; }
            started = true;
        }
        jobs.push_back(Job{p, n, fd});
        cv.notify_one();
        return true;
    }
    void run()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [this] { return quit || !jobs.empty(); });
            if (jobs.empty()) return;                    // (quit: only once the queue is empty)
            const Job j = jobs.back();
            jobs.pop_back();
            lk.unlock();
            munmap(j.p, j.n);
            if (j.fd >= 0) ::close(j.fd);
            lk.lock();
        }
    }
    ~Unmapper()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            cv.notify_one();
        }
        if (th.joinable()) th.join();
    }
};
bool MappedFile::unmap_later(void *p, size_t n, int f)
{
    static Unmapper u;
    return u.push(p, n, f);
}

using Block = BgzfBlock;

// parses the block header at file offset `off`; false on a malformed block
bool parse_block(const MappedFile &f, uint64_t off, Block &b)
{
    if (off + 18 > f.size) return false;
    const uint8_t *p = f.data + off;
    if (p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) return false;
    const uint32_t xlen = rd16(p + 10);
    if (off + 12 + xlen > f.size) return false;
    uint32_t bsize = 0;
    bool found = false;
    for (uint32_t x = 0; x + 4 <= xlen;) {
        const uint8_t *s = p + 12 + x;
        const uint32_t slen = rd16(s + 2);
        if (s[0] == 'B' && s[1] == 'C' && slen == 2 && x + 6 <= xlen) { bsize = rd16(s + 4); found = true; }
        x += 4 + slen;
    }
    if (!found) return false;
    b.coff = off;
    b.csize = bsize + 1;
    if (off + b.csize > f.size || b.csize < 12 + xlen + 8) return false;
    b.doff = 12 + xlen;
    b.dlen = b.csize - b.doff - 8;
    b.isize = rd32(p + b.csize - 4);
    b.crc = rd32(p + b.csize - 8);
    return true;
}

// ---------------------------------------------------------------------------------------------
// BAM stream parser: header, then records -> columns.  Fed with consecutive pieces of the
// uncompressed stream; keeps the unparsed tail between calls.
// ---------------------------------------------------------------------------------------------
class BamParser {
public:
    BamParser(BamHeader &h, HostColumns &c, bool want_header) : hdr_(h), cols_(c), in_header_(want_header) {}

    // stop parsing at this absolute stream position (records starting at or after it are left alone)
    void set_limit(uint64_t lim) { limit_ = lim; }
    void set_position(uint64_t p) { abs_ = p; carry_.clear(); done_ = false; }
    bool reached_limit() const { return done_; }
    bool in_header() const { return in_header_; }
    size_t pending() const { return carry_.size(); }
    uint64_t position() const { return abs_; }

    // returns 0, or a negative BSIG_ERR_* (message recorded)
    int feed(const uint8_t *d, size_t n)
    {
        if (done_) return 0;
        if (!carry_.empty() || in_header_) {
            // the header and records that straddle two pieces are assembled in the carry buffer
            size_t used = 0;
            for (;;) {
                const size_t need = bytes_needed();
                if (carry_.size() < need) {
                    const size_t take = std::min(n - used, need - carry_.size());
                    if (take == 0) return 0;                       // input exhausted, unit incomplete
                    carry_.insert(carry_.end(), d + used, d + used + take);
                    used += take;
                    continue;
                }
                if (in_header_) {
                    const int rc = try_header();
                    if (rc < 0) return rc;
                    if (in_header_) continue;                      // header_need_ grew
                    break;
                }
                if (abs_ >= limit_) { done_ = true; return 0; }
                if (rdi32(carry_.data()) < 32) return fail(BSIG_ERR_FORMAT, "malformed BAM record (block_size %d)", rdi32(carry_.data()));
                const int rc = one_record(carry_.data(), carry_.size());
                if (rc < 0) return rc;
                abs_ += carry_.size();
                carry_.clear();
                break;
            }
            d += used; n -= used;
        }
        // fast path: whole records inside this piece
        size_t o = 0;
        while (o + 4 <= n) {
            if (abs_ + o >= limit_) { done_ = true; return 0; }
            const int32_t bs = rdi32(d + o);
            if (bs < 32) return fail(BSIG_ERR_FORMAT, "malformed BAM record (block_size %d)", bs);
            if (o + 4 + (size_t)bs > n) break;
            const int rc = one_record(d + o, 4 + (size_t)bs);
            if (rc < 0) return rc;
            o += 4 + (size_t)bs;
        }
        abs_ += o;
        carry_.assign(d + o, d + n);
        return 0;
    }

    void finish_refs()
    {
        const int n_ref = (int)hdr_.names.size();
        while ((int)cols_.ref_off.size() < n_ref + 1) cols_.ref_off.push_back(cols_.size());
        if (cols_.cigar_off.empty()) cols_.cigar_off.push_back(0);
    }

private:
    size_t bytes_needed() const
    {
        if (in_header_) return header_need_;
        if (carry_.size() < 4) return 4;
        const int32_t bs = rdi32(carry_.data());
        return 4 + (size_t)std::max(bs, 0);
    }

    int try_header()
    {
        const uint8_t *p = carry_.data();
        const size_t n = carry_.size();
        if (n < 12) { header_need_ = 12; return 0; }
        if (memcmp(p, "BAM\1", 4) != 0) return fail(BSIG_ERR_FORMAT, "not a BAM file (bad magic)");
        const int32_t l_text = rdi32(p + 4);
        if (l_text < 0) return fail(BSIG_ERR_FORMAT, "malformed BAM header");
        size_t o = 8 + (size_t)l_text;
        if (n < o + 4) { header_need_ = o + 4; return 0; }
        const int32_t n_ref = rdi32(p + o);
        if (n_ref < 0) return fail(BSIG_ERR_FORMAT, "malformed BAM header");
        o += 4;
        for (int r = 0; r < n_ref; ++r) {
            if (n < o + 4) { header_need_ = o + 4; return 0; }
            const int32_t l_name = rdi32(p + o);
            if (l_name < 1) return fail(BSIG_ERR_FORMAT, "malformed BAM header");
            if (n < o + 4 + (size_t)l_name + 4) { header_need_ = o + 4 + (size_t)l_name + 4; return 0; }
            o += 4 + (size_t)l_name + 4;
        }
        if (n < o) { header_need_ = o; return 0; }
        // complete: decode
        hdr_.text.assign((const char *)p + 8, (size_t)l_text);
        while (!hdr_.text.empty() && hdr_.text.back() == '\0') hdr_.text.pop_back();
        hdr_.names.clear(); hdr_.lens.clear();
        size_t q = 8 + (size_t)l_text + 4;
        for (int r = 0; r < n_ref; ++r) {
            const int32_t l_name = rdi32(p + q);
            hdr_.names.emplace_back((const char *)p + q + 4, (size_t)l_name - 1);
            hdr_.lens.push_back(rdi32(p + q + 4 + l_name));
            q += 4 + (size_t)l_name + 4;
        }
        // feed() never takes more than bytes_needed(), so the carry holds exactly the header
        abs_ += o;
        carry_.clear();
        in_header_ = false;
        return 0;
    }

    int one_record(const uint8_t *r, size_t len)
    {
        const uint8_t *c = r + 4;
        const int32_t rid = rdi32(c);
        const int32_t pos = rdi32(c + 4);
        const uint32_t l_name = c[8];
        const uint8_t mapq = c[9];
        uint32_t n_cig = rd16(c + 12);
        const uint16_t flag = rd16(c + 14);
        const int32_t l_seq = rdi32(c + 16);
        const int32_t tlen = rdi32(c + 28);
        if (36 + (size_t)l_name + 4 * (size_t)n_cig > len)
            return fail(BSIG_ERR_FORMAT, "malformed BAM record (fields exceed block_size)");
        if (rid < 0) { ++cols_.n_unplaced; return 0; }
        if (rid >= (int)hdr_.names.size()) return fail(BSIG_ERR_FORMAT, "BAM record with refID %d out of range", rid);
        if (rid < last_rid_ || (rid == last_rid_ && pos < last_pos_))
            return fail(BSIG_ERR_FORMAT, "BAM file is not sorted by coordinate");
        while ((int)cols_.ref_off.size() <= rid) cols_.ref_off.push_back(cols_.size());
        last_rid_ = rid; last_pos_ = pos;
        const uint8_t *cig = c + 32 + l_name;
        if (cols_.cigar_off.empty()) cols_.cigar_off.push_back(0);
        // long CIGARs (> 65535 ops) live in the CG:B,I tag behind a kSmN placeholder (SAM spec 4.2.2)
        bool done = false;
        if (n_cig == 2 && l_seq >= 0 && (rd32(cig) & 0xF) == 4 && (int32_t)(rd32(cig) >> 4) == l_seq && (rd32(cig + 4) & 0xF) == 3) {
            size_t a = 36 + (size_t)l_name + 8 + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
            while (a + 3 <= len) {
                const uint8_t t0 = r[a], t1 = r[a + 1], ty = r[a + 2];
                a += 3;
                size_t sz = 0;
                if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
                else if (ty == 's' || ty == 'S') sz = 2;
                else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
                else if (ty == 'Z' || ty == 'H') { while (a + sz < len && r[a + sz]) ++sz; ++sz; }
                else if (ty == 'B') {
                    if (a + 5 > len) break;
                    const uint8_t sub = r[a];
                    const uint32_t cnt = rd32(r + a + 1);
                    const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                    if (t0 == 'C' && t1 == 'G' && sub == 'I' && a + 5 + 4 * (size_t)cnt <= len) {
                        for (uint32_t k = 0; k < cnt; ++k) cols_.cigar.push_back(rd32(r + a + 5 + 4 * (size_t)k));
                        done = true;
                        break;
                    }
                    sz = 5 + es * cnt;
                } else break;
                a += sz;
            }
        }
        if (!done)
            for (uint32_t k = 0; k < n_cig; ++k) cols_.cigar.push_back(rd32(cig + 4 * (size_t)k));
        cols_.cigar_off.push_back((int64_t)cols_.cigar.size());
        cols_.pos.push_back(pos);
        cols_.flag.push_back(flag);
        cols_.mapq.push_back(mapq);
        cols_.tlen.push_back(tlen);
        return 0;
    }

    BamHeader &hdr_;
    HostColumns &cols_;
    bool in_header_;
    bool done_ = false;
    size_t header_need_ = 12;
    std::vector<uint8_t> carry_;
    uint64_t abs_ = 0;                       // stream position of the first unparsed byte
    uint64_t limit_ = ~0ull;
    int32_t last_rid_ = -1, last_pos_ = -1;
};

}  // namespace

// ---------------------------------------------------------------------------------------------
// whole-file decode: blocks are inflated by a thread pool in batches while the previous batch
// is parsed
// ---------------------------------------------------------------------------------------------
// Map the pages of the given byte spans of the file in from several threads, in slices of 4 MiB.
// Dozens of threads that each fault single pages of one mapping queue up in the kernel (an index-
// driven decode of 38,000 chunks spent 60 ms that way); populating whole slices first costs a few ms.
static void populate_spans(const MappedFile &f, std::vector<std::pair<uint64_t, uint64_t>> spans)
{
    if (spans.empty() || !f.data) return;
    std::sort(spans.begin(), spans.end());
    std::vector<std::pair<uint64_t, uint64_t>> merged;
    for (const auto &sp : spans) {
        const uint64_t a = std::min<uint64_t>(sp.first, f.size) & ~4095ull, b = std::min<uint64_t>(sp.second, f.size);
        if (a >= b) continue;
        if (!merged.empty() && a <= merged.back().second + (256u << 10)) merged.back().second = std::max(merged.back().second, b);
        else merged.emplace_back(a, b);
    }
    std::vector<std::pair<uint64_t, uint64_t>> slices;
    const uint64_t slice = 4u << 20;
    for (const auto &m : merged)
        for (uint64_t a = m.first; a < m.second; a += slice) slices.emplace_back(a, std::min(m.second, a + slice));
    parallel_for((int64_t)slices.size(), std::min(n_threads(0), 16), [&](int64_t i, int) {
        const uint64_t a = slices[(size_t)i].first, b = slices[(size_t)i].second;
#ifdef MADV_POPULATE_READ
        if (madvise((void *)(f.data + a), b - a, MADV_POPULATE_READ) == 0) return;
#endif
        volatile uint8_t sink = 0;
        for (uint64_t o = a; o < b; o += 4096) sink = sink + f.data[o];
    });
}

// ---- the same header walk through pread() -------------------------------------------------------------
// The device-side decode needs the block table, not the pages: walking the headers through the mapping
// touches nearly every page of the file, and filling the page tables of a fresh 3-GB mapping costs 0.05-0.13 s
// (more than the walk) although every byte is in the page cache.  pread() copies the few bytes the walk
// looks at -- a block's trailer and the next block's header are neighbours: one 96-byte read per block --
// and creates no mapping at all.  Same segments, same join rule as scan_blocks() below; any doubt (a segment
// without a provable start, a chain that does not meet the next segment) returns false and the mapped walk,
// which owns the error messages, runs instead.
namespace {
struct HeadReader {
    int fd;
    uint64_t size;
    uint8_t buf[192];
    uint64_t base = ~0ull;
    size_t len = 0;
    // the bytes [off, off + need) of the file, need <= 96; nullptr behind the end of the file
    const uint8_t *at(uint64_t off, size_t need)
    {
        if (off + need > size) return nullptr;
        if (base != ~0ull && off >= base && off + need <= base + len) return buf + (off - base);
        const size_t want = (size_t)std::min<uint64_t>(sizeof buf, size - off);
        size_t got = 0;
        while (got < want) {
            const ssize_t r = ::pread(fd, buf + got, want - got, (off_t)(off + got));
            if (r <= 0) return nullptr;
            got += (size_t)r;
        }
        base = off;
        len = got;
        return buf;
    }
};

// parse_block() on bytes fetched with pread(); after a success the reader holds the next block's header
bool parse_block_fd(HeadReader &r, uint64_t off, Block &b)
{
    const uint8_t *p = r.at(off, 18);
    if (!p || p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) return false;
    const uint32_t xlen = rd16(p + 10);
    if (xlen > 64) return false;                         // (odd extra fields: the mapped walk handles them)
    p = r.at(off, 12 + xlen);
    if (!p) return false;
    uint32_t bsize = 0;
    bool found = false;
    for (uint32_t x = 0; x + 4 <= xlen;) {
        const uint8_t *s = p + 12 + x;
        const uint32_t slen = rd16(s + 2);
        if (s[0] == 'B' && s[1] == 'C' && slen == 2 && x + 6 <= xlen) { bsize = rd16(s + 4); found = true; }
        x += 4 + slen;
    }
    if (!found) return false;
    b.coff = off;
    b.csize = bsize + 1;
    if (off + b.csize > r.size || b.csize < 12 + xlen + 8) return false;
    b.doff = 12 + xlen;
    b.dlen = b.csize - b.doff - 8;
    // the trailer, together with the next block's header (what the next call asks for)
    const uint64_t t = off + b.csize - 8;
    const uint8_t *q = r.at(t, (size_t)std::min<uint64_t>(8 + 32, r.size - t));
    if (!q) return false;
    b.crc = rd32(q);
    b.isize = rd32(q + 4);
    return true;
}
}  // namespace

// The walk in K parallel segments; segments [k0, k1) can be run and stitched on their own, so that the head of a
// large file is tabulated first and the rest while the GPU already works on the head (BgzfFile::open_progressive).
struct PreadScan {
    const MappedFile &f;
    size_t K = 0;
    std::vector<std::vector<Block>> part;
    std::vector<uint64_t> first, last_end;
    std::vector<char> ok;
    explicit PreadScan(const MappedFile &file) : f(file)
    {
        uint64_t seg_bytes = 8u << 20;
        if (const char *e = getenv("BAMSIGNALS_SCAN_SEGMENT_KB")) {
            const long v = atol(e);
            if (v > 0) seg_bytes = (uint64_t)v << 10;
        }
        if (f.fd < 0 || f.size < 2 * seg_bytes) return;
        K = (size_t)std::min<uint64_t>(1024, f.size / seg_bytes);
        part.resize(K);
        first.assign(K, 0);
        last_end.assign(K, 0);
        ok.assign(K, 0);
    }
    uint64_t cut(size_t k) const { return k >= K ? f.size : (uint64_t)k * (f.size / K); }
    void run(size_t k0, size_t k1, int threads)
    {
        const int kHops = 4;
        parallel_for((int64_t)(k1 - k0), threads, [&](int64_t kk, int) {
            const size_t k = k0 + (size_t)kk;
            HeadReader r{f.fd, f.size};
            const uint64_t c = cut(k), next_cut = cut(k + 1);
            uint64_t start = c;
            if (k > 0) {
                // the first offset behind the cut from which a chain of kHops valid headers follows
                std::vector<uint8_t> win((size_t)std::min<uint64_t>(next_cut - c + 2, 256u << 10));
                size_t got = 0;
                while (got < win.size()) {
                    const ssize_t n = ::pread(f.fd, win.data() + got, win.size() - got, (off_t)(c + got));
                    if (n <= 0) break;
                    got += (size_t)n;
                }
                bool found = false;
                for (size_t i = 0; i + 1 < got && !found; ++i) {
                    if (win[i] != 31 || win[i + 1] != 139) continue;
                    uint64_t q = c + i;
                    int hops = 0;
                    Block b;
                    while (hops < kHops && q < f.size && parse_block_fd(r, q, b)) { q += b.csize; ++hops; }
                    if (hops == kHops || (hops > 0 && q == f.size)) { start = c + i; found = true; }
                }
                // no block begins in this segment (the window covered all of it): fine if the chain of the
                // segments before it runs past this one -- stitch() checks that
                if (!found) { if (got >= next_cut - c) ok[k] = 2; return; }
            }
            first[k] = start;
            uint64_t o = start;
            std::vector<Block> &out = part[k];
            out.reserve((size_t)((next_cut - c) / 4096 + 16));
            while (o < next_cut) {
                Block b;
                if (!parse_block_fd(r, o, b)) return;
                out.push_back(b);
                o += b.csize;
            }
            last_end[k] = o;
            ok[k] = 1;
        });
    }
    // appends the blocks of segments [k0, k1) to `blocks` if their chains meet (`at`: where the chain stands,
    // 0 before segment 0); false on any doubt
    bool stitch(size_t k0, size_t k1, uint64_t &at, std::vector<Block> &blocks)
    {
        for (size_t k = k0; k < k1; ++k) {
            if (ok[k] == 2) {                              // a segment inside one block
                if (at < cut(k + 1)) return false;
                continue;
            }
            if (!ok[k] || first[k] != at) return false;
            at = last_end[k];
        }
        if (k1 == K && at != f.size) return false;
        size_t n = 0;
        for (size_t k = k0; k < k1; ++k) n += part[k].size();
        blocks.reserve(blocks.size() + n);
        for (size_t k = k0; k < k1; ++k) {
            blocks.insert(blocks.end(), part[k].begin(), part[k].end());
            std::vector<Block>().swap(part[k]);
        }
        return true;
    }
};

static bool scan_blocks_pread(const MappedFile &f, std::vector<Block> &blocks)
{
    PreadScan sc(f);
    if (sc.K == 0) return false;
    sc.run(0, sc.K, std::min(n_threads(0), 32));
    blocks.clear();
    uint64_t at = 0;
    if (!sc.stitch(0, sc.K, at, blocks)) { blocks.clear(); return false; }
    return true;
}

static int scan_blocks(const MappedFile &f, const std::string &path, std::vector<Block> &blocks)
{
    // The scan below touches one header per block, i.e. nearly every page of the mapping, one page
    // fault after the other: map the pages in from several threads first (8-15 ms -> a few ms for a
    // 300-MB file; the inflate would fault them in anyway).
    if (f.size >= (16u << 20)) populate_spans(f, {{0, f.size}});
    // Large files: the header chain is walked in K segments at once.  A segment begins at the first
    // offset behind its cut from which a chain of kHops valid block headers follows (the gzip magic
    // and the BC subfield make a false start all but impossible; the join below catches one
    // anyway): the segments' lists are accepted only if every one ends exactly where the next one
    // begins, otherwise the plain serial walk runs.
    const int kHops = 4;
    uint64_t seg_bytes = 32u << 20;
    if (const char *e = getenv("BAMSIGNALS_SCAN_SEGMENT_KB")) {          // testing: segments on small files
        const long v = atol(e);
        if (v > 0) seg_bytes = (uint64_t)v << 10;
    }
    if (f.size >= 4 * seg_bytes) {
        const size_t K = (size_t)std::min<uint64_t>(256, f.size / seg_bytes);
        std::vector<std::vector<Block>> part(K);
        std::vector<uint64_t> first(K, 0), last_end(K, 0);
        std::vector<char> ok(K, 0);
        parallel_for((int64_t)K, std::min(n_threads(0), 32), [&](int64_t k, int) {
            const uint64_t cut = (uint64_t)k * (f.size / K), next_cut = k + 1 == (int64_t)K ? f.size : (uint64_t)(k + 1) * (f.size / K);
            uint64_t start = cut;
            if (k > 0) {
                bool found = false;
                for (uint64_t o = cut; o < next_cut && o + 18 <= f.size && !found; ++o) {
                    if (f.data[o] != 31 || f.data[o + 1] != 139) continue;
                    uint64_t q = o;
                    int hops = 0;
                    Block b;
                    while (hops < kHops && q < f.size && parse_block(f, q, b)) { q += b.csize; ++hops; }
                    if (hops == kHops || (hops > 0 && q == f.size)) { start = o; found = true; }
                }
                if (!found) return;
            }
            first[(size_t)k] = start;
            uint64_t o = start;
            std::vector<Block> &out = part[(size_t)k];
            while (o < next_cut) {
                Block b;
                if (!parse_block(f, o, b)) return;
                out.push_back(b);
                o += b.csize;
            }
            last_end[(size_t)k] = o;
            ok[(size_t)k] = 1;
        });
        bool good = true;
        for (size_t k = 0; k < K && good; ++k) {
            good = ok[k] != 0;
            if (good && k + 1 < K) good = ok[k + 1] && last_end[k] == first[k + 1];
            if (good && k + 1 == K) good = last_end[k] == f.size;
        }
        if (good) {
            size_t n = 0;
            for (const auto &v : part) n += v.size();
            blocks.reserve(n);
            for (const auto &v : part) blocks.insert(blocks.end(), v.begin(), v.end());
            return 0;
        }
        blocks.clear();
    }
    uint64_t off = 0;
    while (off < f.size) {
        Block b;
        if (!parse_block(f, off, b)) return fail(BSIG_ERR_FORMAT, "malformed BGZF block at offset %llu of %s", (unsigned long long)off, path.c_str());
        blocks.push_back(b);
        off += b.csize;
    }
    return 0;
}

// inflates blocks [b0, b1) back to back into out[prefix ...]; out is resized to prefix + total
static int inflate_batch(const MappedFile &f, const std::vector<Block> &blocks, size_t b0, size_t b1,
                         int threads, std::vector<uint8_t> &out, size_t prefix = 0)
{
    std::vector<uint64_t> uoff(b1 - b0 + 1, 0);
    for (size_t k = b0; k < b1; ++k) uoff[k - b0 + 1] = uoff[k - b0] + blocks[k].isize;
    out.resize(prefix + uoff.back());
    std::atomic<int> bad(0);
    parallel_for((int64_t)(b1 - b0), threads, [&](int64_t i, int) {
        static thread_local Inflater inf;          // one decompressor per worker thread
        const Block &b = blocks[b0 + (size_t)i];
        if (const int r = inflate_checked(inf, f.data, b, out.data() + prefix + uoff[(size_t)i])) bad = std::max(bad.load(), r);
    });
    if (bad) return fail(BSIG_ERR_FORMAT, bad == 2 ? "BGZF block CRC mismatch (the file is damaged)" : "BGZF inflate failed");
    return 0;
}

// ---- the BGZF layer on its own (device-side record decode) ------------------------------------
struct BgzfFile::Impl {
    MappedFile f;
    std::vector<Block> blocks;
    // open_progressive: the rest of the table is being walked by `bg`
    std::unique_ptr<PreadScan> scan;
    size_t k_head = 0;
    uint64_t at = 0;
    std::thread bg;
    std::string path;
    ~Impl() { if (bg.joinable()) bg.join(); }
};
BgzfFile::BgzfFile() : p_(new Impl) {}
BgzfFile::~BgzfFile() { delete p_; }
int BgzfFile::open(const std::string &path)
{
    if (p_->f.open(path) != 0) return fail(BSIG_ERR_IO, "Fail to open BAM file %s", path.c_str());
    p_->blocks.clear();
    // The block table through pread(), without touching the mapping (env BAMSIGNALS_SCAN=mmap: through the
    // populated mapping, which is also what any doubt falls back to).  Measured on the north star's file in
    // fresh processes (scripts/cold_call_scan_ab.py): the mapped walk itself is quicker on a file that has been
    // read before (0.015 s against 0.04-0.06 s for 327,000 small reads), but the call is not -- 0.45 s against
    // 0.41 s: filling and tearing down the page tables of a 3-GB mapping holds the process's mmap lock, and the
    // 800-MB result array that is being first-touched at the other end of the call waits behind it.
    const char *how = getenv("BAMSIGNALS_SCAN");
    if (!(how && !strcmp(how, "mmap")) && scan_blocks_pread(p_->f, p_->blocks)) return 0;
    p_->blocks.clear();
    return scan_blocks(p_->f, path, p_->blocks);
}
int BgzfFile::open_progressive(const std::string &path, uint64_t head_bytes)
{
    const char *how = getenv("BAMSIGNALS_SCAN");
    if (p_->f.open(path) != 0) return fail(BSIG_ERR_IO, "Fail to open BAM file %s", path.c_str());
    p_->blocks.clear();
    p_->path = path;
    if (!(how && !strcmp(how, "mmap")) && p_->f.size > 2 * head_bytes) {
        std::unique_ptr<PreadScan> sc(new PreadScan(p_->f));
        if (sc->K > 0) {
            size_t kh = 1;
            while (kh < sc->K && sc->cut(kh) < head_bytes) ++kh;
            const int thr = std::min(n_threads(0), 32);
            sc->run(0, kh, thr);
            uint64_t at = 0;
            if (kh < sc->K && sc->stitch(0, kh, at, p_->blocks) && p_->blocks.size() > 8) {
                p_->scan = std::move(sc);
                p_->k_head = kh;
                p_->at = at;
                PreadScan *raw = p_->scan.get();
                try {
                    p_->bg = std::thread([raw, kh, thr] { raw->run(kh, raw->K, thr); });
                    return 0;
                } catch (const std::system_error &) {
                    raw->run(kh, raw->K, thr);
                    return 0;                      // (finish() stitches)
                }
            }
            p_->blocks.clear();
        }
    }
    // small file, or a head that could not be proven: the whole table now
    if (!(how && !strcmp(how, "mmap")) && scan_blocks_pread(p_->f, p_->blocks)) return 0;
    p_->blocks.clear();
    return scan_blocks(p_->f, path, p_->blocks);
}
bool BgzfFile::complete() const { return !p_->scan; }
int BgzfFile::finish()
{
    if (!p_->scan) return 0;
    if (p_->bg.joinable()) p_->bg.join();
    std::unique_ptr<PreadScan> sc = std::move(p_->scan);
    uint64_t at = p_->at;
    if (sc->stitch(p_->k_head, sc->K, at, p_->blocks)) return 0;
    // the rest could not be proven: the mapped walk over the whole file (it owns the error messages)
    p_->blocks.clear();
    return scan_blocks(p_->f, p_->path, p_->blocks);
}
int BgzfFile::map(const std::string &path)
{
    if (p_->f.open(path) != 0) return fail(BSIG_ERR_IO, "Fail to open BAM file %s", path.c_str());
    p_->blocks.clear();
    return 0;
}
void BgzfFile::append_blocks(const BgzfBlock *b, size_t n) { p_->blocks.insert(p_->blocks.end(), b, b + n); }
bool BgzfFile::block_at(uint64_t off, BgzfBlock &b) const { return parse_block(p_->f, off, b); }
void BgzfFile::populate(const std::vector<std::pair<uint64_t, uint64_t>> &spans) const { populate_spans(p_->f, spans); }
bool BgzfFile::read_span(uint64_t off, size_t len, uint8_t *dst) const
{
    if (p_->f.fd < 0 || off + len > p_->f.size) return false;
    size_t got = 0;
    while (got < len) {
        const ssize_t r = ::pread(p_->f.fd, dst + got, len - got, (off_t)(off + got));
        if (r <= 0) return false;
        got += (size_t)r;
    }
    return true;
}
int BgzfFile::inflate_list(const BgzfBlock *list, size_t n, uint8_t *dst, int threads) const
{
    if (n == 0) return 0;
    std::vector<uint64_t> uoff(n + 1, 0);
    for (size_t k = 0; k < n; ++k) uoff[k + 1] = uoff[k] + list[k].isize;
    std::atomic<int> bad(0);
    const MappedFile &f = p_->f;
    parallel_for((int64_t)n, n_threads(threads), [&](int64_t i, int) {
        static thread_local Inflater inf;
        const Block &b = list[(size_t)i];
        if (const int r = inflate_checked(inf, f.data, b, dst + uoff[(size_t)i])) bad = std::max(bad.load(), r);
    });
    if (bad) return fail(BSIG_ERR_FORMAT, bad == 2 ? "BGZF block CRC mismatch (the file is damaged)" : "BGZF inflate failed");
    return 0;
}
int decode_threads(int t) { return n_threads(t); }
int effective_cpus() { return eff_cpus(); }
void pool_for(int64_t n, int threads, const std::function<void(int64_t)> &body)
{
    parallel_for(n, n_threads(threads), [&](int64_t i, int) { body(i); });
}
const std::vector<BgzfBlock> &BgzfFile::blocks() const { return p_->blocks; }
const uint8_t *BgzfFile::data() const { return p_->f.data; }
size_t BgzfFile::size() const { return p_->f.size; }
int BgzfFile::inflate(size_t b0, size_t b1, uint8_t *dst, int threads) const
{
    if (b0 >= b1) return 0;
    std::vector<uint64_t> uoff(b1 - b0 + 1, 0);
    for (size_t k = b0; k < b1; ++k) uoff[k - b0 + 1] = uoff[k - b0] + p_->blocks[k].isize;
    std::atomic<int> bad(0);
    const MappedFile &f = p_->f;
    const std::vector<Block> &blocks = p_->blocks;
    parallel_for((int64_t)(b1 - b0), n_threads(threads), [&](int64_t i, int) {
        static thread_local Inflater inf;
        const Block &b = blocks[b0 + (size_t)i];
        if (const int r = inflate_checked(inf, f.data, b, dst + uoff[(size_t)i])) bad = std::max(bad.load(), r);
    });
    if (bad) return fail(BSIG_ERR_FORMAT, bad == 2 ? "BGZF block CRC mismatch (the file is damaged)" : "BGZF inflate failed");
    return 0;
}

int64_t bam_header_bytes(const uint8_t *p, size_t n)
{
    if (n < 12) return -1;
    if (memcmp(p, "BAM\1", 4) != 0) return -2;
    const int32_t l_text = rdi32(p + 4);
    if (l_text < 0) return -2;
    uint64_t o = 8 + (uint64_t)l_text;
    if (o + 4 > n) return -1;
    const int32_t n_ref = rdi32(p + o);
    if (n_ref < 0) return -2;
    o += 4;
    for (int32_t r = 0; r < n_ref; ++r) {
        if (o + 4 > n) return -1;
        const int32_t l_name = rdi32(p + o);
        if (l_name < 0) return -2;
        o += 4 + (uint64_t)l_name + 4;
        if (o > n) return -1;
    }
    return (int64_t)o;
}

namespace {

// real CIGAR of a record whose in-record CIGAR is the kSmN placeholder of a > 65535-op alignment
// (SAM spec 4.2.2): pointer to the CG:B,I payload and its length, or nullptr
const uint8_t *find_cg_tag(const uint8_t *r, size_t len, uint32_t *n_ops)
{
    const uint8_t *c = r + 4;
    const uint32_t l_name = c[8];
    const uint32_t n_cig = rd16(c + 12);
    const int32_t l_seq = rdi32(c + 16);
    if (n_cig != 2 || l_seq < 0) return nullptr;
    const uint8_t *cig = c + 32 + l_name;
    if (36 + (size_t)l_name + 8 > len) return nullptr;
    if ((rd32(cig) & 0xF) != 4 || (int32_t)(rd32(cig) >> 4) != l_seq || (rd32(cig + 4) & 0xF) != 3) return nullptr;
    size_t a = 36 + (size_t)l_name + 8 + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    while (a + 3 <= len) {
        const uint8_t t0 = r[a], t1 = r[a + 1], ty = r[a + 2];
        a += 3;
        size_t sz = 0;
        if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
        else if (ty == 's' || ty == 'S') sz = 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
        else if (ty == 'Z' || ty == 'H') { while (a + sz < len && r[a + sz]) ++sz; ++sz; }
        else if (ty == 'B') {
            if (a + 5 > len) return nullptr;
            const uint8_t sub = r[a];
            const uint32_t cnt = rd32(r + a + 1);
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            if (t0 == 'C' && t1 == 'G' && sub == 'I' && a + 5 + 4 * (size_t)cnt <= len) { *n_ops = cnt; return r + a + 5; }
            sz = 5 + es * cnt;
        } else return nullptr;
        a += sz;
    }
    return nullptr;
}

}  // namespace

// Whole file: batches of BGZF blocks are inflated by the thread pool while the previous batch is
// parsed; parsing a batch = a serial walk over the record lengths (boundaries, sortedness, CIGAR
// offsets) + a parallel extraction of the columns.
int bam_decode_all(const std::string &path, int threads, BamHeader &hdr, HostColumns &cols)
{
    MappedFile f;
    if (f.open(path) != 0) return fail(BSIG_ERR_IO, "Fail to open BAM file %s", path.c_str());
    threads = n_threads(threads);
    double *T = g_decode_timing;
    T[0] = T[1] = T[2] = T[3] = T[4] = T[5] = 0;
    const double t_begin = now_s();
    std::vector<Block> blocks;
    int rc = scan_blocks(f, path, blocks);
    if (rc) return rc;
    T[0] = now_s() - t_begin;
    cols = HostColumns();
    {
        // reserve from the uncompressed size (>= 38 bytes per record) so the columns never regrow
        uint64_t total_u = 0;
        for (const Block &b : blocks) total_u += b.isize;
        const size_t guess = (size_t)(total_u / 38) + 16;
        cols.pos.reserve(guess); cols.tlen.reserve(guess); cols.flag.reserve(guess); cols.mapq.reserve(guess);
        cols.cigar_off.reserve(guess + 1); cols.cigar.reserve(guess + guess / 2);
    }
    cols.cigar_off.push_back(0);

    size_t batch = 2048;                    // blocks per batch (<= 128 MiB uncompressed)
    size_t kPrefix = 8u << 20;              // room in front of a batch for the previous batch's tail
    if (const char *e = getenv("BAMSIGNALS_BATCH_BLOCKS")) {      // testing: force many small batches
        const long v = atol(e);
        if (v > 0) { batch = (size_t)v; kPrefix = 16; }
    }
    std::vector<uint8_t> cur, nxt;
    if (blocks.empty()) return fail(BSIG_ERR_FORMAT, "truncated BAM header in %s", path.c_str());
    {
        // both batch buffers get their final capacity now: a later, slightly larger batch must not
        // reallocate (and page in) 128 MiB again
        const size_t cap = kPrefix + std::min(batch, blocks.size()) * (size_t)0x10000;
        cur.reserve(cap);
        nxt.reserve(cap);
    }
    double t0 = now_s();
    rc = inflate_batch(f, blocks, 0, std::min(batch, blocks.size()), threads, cur, kPrefix);
    if (rc) return rc;
    T[1] += now_s() - t0;
    size_t begin = kPrefix;                 // first unparsed byte of `cur`
    {
        // the header (must end inside the first batch)
        HostColumns dummy;
        BamParser hp(hdr, dummy, true);
        hp.set_limit(0);
        rc = hp.feed(cur.data() + kPrefix, cur.size() - kPrefix);
        if (rc) return rc;
        if (hp.in_header()) {
            if (blocks.size() <= batch) return fail(BSIG_ERR_FORMAT, "truncated BAM header in %s", path.c_str());
            return fail(BSIG_ERR_FORMAT, "BAM header of %s is larger than 128 MiB", path.c_str());
        }
        begin = kPrefix + hp.position();
    }
    const int n_ref = (int)hdr.names.size();
    int32_t last_rid = -1, last_pos = -1;

    // One segment of a batch: the records found by walking block_size links from `start`.
    // Segments other than the first start at a BGZF block boundary, which is a record boundary in
    // files written by htslib/samtools and by BamWriter (records are kept inside a block when they
    // fit).  That is only an assumption: the serial merge below re-walks a segment whenever the
    // previous one did not end exactly where this one started.
    struct Segment {
        size_t start = 0, limit = 0, end = 0;      // walk [start, ...) until o >= limit; end = o reached
        std::vector<uint32_t> starts, ncig;
        std::vector<uint8_t> has_cg;
        std::vector<std::pair<uint32_t, int32_t>> rid_change;   // (local record index, new rid)
        int64_t unplaced = 0;
        int32_t first_rid = -1, first_pos = -1, last_rid = -1, last_pos = -1;
        bool sorted = true, any_cg = false, incomplete = false;
        int bad = 0;                               // 1 bad block_size, 2 refID range, 3 field overflow
        int bad_value = 0;
    };
    auto walk = [&](const uint8_t *d, size_t n, Segment &g) {
        g.starts.clear(); g.ncig.clear(); g.has_cg.clear(); g.rid_change.clear();
        g.unplaced = 0; g.first_rid = g.last_rid = -1; g.first_pos = g.last_pos = -1;
        g.sorted = true; g.any_cg = false; g.incomplete = false; g.bad = 0;
        size_t o = g.start;
        int32_t prid = -1, ppos = -1;
        while (o < g.limit) {
            if (o + 4 > n) { g.incomplete = true; break; }
            const int32_t bs = rdi32(d + o);
            if (bs < 32) { g.bad = 1; g.bad_value = bs; break; }
            const size_t next = o + 4 + (size_t)bs;
            if (next > n) { g.incomplete = true; break; }
            const int32_t rid = rdi32(d + o + 4), pos = rdi32(d + o + 8);
            if (rid < 0) { ++g.unplaced; o = next; continue; }
            if (rid >= n_ref) { g.bad = 2; g.bad_value = rid; break; }
            const uint32_t l_name = d[o + 12];
            uint32_t nc = rd16(d + o + 16);
            if (36 + (size_t)l_name + 4 * (size_t)nc > 4 + (size_t)bs) { g.bad = 3; break; }
            uint8_t cg = 0;
            if (nc == 2) {
                uint32_t real = 0;
                if (find_cg_tag(d + o, 4 + (size_t)bs, &real)) { nc = real; cg = 1; g.any_cg = true; }
            }
            if (g.starts.empty()) { g.first_rid = rid; g.first_pos = pos; }
            else if (rid < prid || (rid == prid && pos < ppos)) g.sorted = false;
            if (rid != prid) g.rid_change.emplace_back((uint32_t)g.starts.size(), rid);
            prid = rid; ppos = pos;
            g.starts.push_back((uint32_t)o);
            g.ncig.push_back(nc);
            g.has_cg.push_back(cg);
            o = next;
        }
        g.last_rid = prid; g.last_pos = ppos;
        g.end = o;
    };

    std::vector<Segment> segs;
    size_t b0 = 0;
    while (b0 < blocks.size()) {
        const size_t b1 = std::min(b0 + batch, blocks.size());
        const size_t b2 = std::min(b1 + batch, blocks.size());
        int rc_next = 0;
        std::thread producer;
        if (b1 < b2) producer = std::thread([&, T] {
            const double ti = now_s();
            rc_next = inflate_batch(f, blocks, b1, b2, std::max(1, threads - 1), nxt, kPrefix);
            T[5] += now_s() - ti;            // producer-side inflate time (this thread's timer array)
        });
        const int workers = producer.joinable() ? std::max(1, threads - 1) : threads;

        // ---- phase 1: record boundaries, speculatively in parallel ---------------------------------
        t0 = now_s();
        const uint8_t *d = cur.data();
        const size_t n = cur.size();
        {
            // segment starts: `begin`, then every seg_blocks-th block boundary behind it
            const size_t seg_blocks = std::max<size_t>(1, std::min<size_t>(32, (b1 - b0) / (size_t)(4 * workers) + 1));
            std::vector<size_t> cut;
            cut.push_back(begin);
            size_t boundary = n - [&] { uint64_t t = 0; for (size_t k = b0; k < b1; ++k) t += blocks[k].isize; return (size_t)t; }();
            for (size_t k = b0; k < b1; ++k) {
                if (k > b0 && (k - b0) % seg_blocks == 0 && boundary > begin) cut.push_back(boundary);
                boundary += blocks[k].isize;
            }
            segs.resize(cut.size());
            for (size_t k = 0; k < cut.size(); ++k) {
                segs[k].start = cut[k];
                segs[k].limit = k + 1 < cut.size() ? cut[k + 1] : n;
            }
        }
        parallel_for((int64_t)segs.size(), workers, [&](int64_t k, int) { walk(d, n, segs[(size_t)k]); });
        // serial merge: every segment must begin where its predecessor ended
        size_t o = begin;
        int err = 0;
        for (size_t k = 0; k < segs.size() && !err; ++k) {
            Segment &g = segs[k];
            if (o >= g.limit && k + 1 < segs.size()) {       // predecessor's last record swallowed this one
                g.starts.clear(); g.ncig.clear(); g.has_cg.clear(); g.rid_change.clear();
                g.unplaced = 0; g.sorted = true; g.any_cg = false; g.incomplete = false; g.bad = 0;
                g.start = g.end = o;
                continue;
            }
            if (g.start != o) { g.start = o; walk(d, n, g); }     // the assumption failed: walk again
            if (g.bad == 1) err = fail(BSIG_ERR_FORMAT, "malformed BAM record (block_size %d)", g.bad_value);
            else if (g.bad == 2) err = fail(BSIG_ERR_FORMAT, "BAM record with refID %d out of range", g.bad_value);
            else if (g.bad == 3) err = fail(BSIG_ERR_FORMAT, "malformed BAM record (fields exceed block_size)");
            if (err) break;
            if (!g.starts.empty()) {
                if (!g.sorted || g.first_rid < last_rid || (g.first_rid == last_rid && g.first_pos < last_pos))
                    err = fail(BSIG_ERR_FORMAT, "BAM file is not sorted by coordinate");
                last_rid = g.last_rid; last_pos = g.last_pos;
            }
            o = g.end;
            if (g.incomplete) {
                // only the tail of the batch may be incomplete; later segments (if any) hold nothing
                for (size_t q = k + 1; q < segs.size(); ++q) { segs[q].starts.clear(); segs[q].ncig.clear(); segs[q].has_cg.clear(); segs[q].rid_change.clear(); segs[q].unplaced = 0; }
                break;
            }
        }
        if (err) { if (producer.joinable()) producer.join(); return err; }
        T[2] += now_s() - t0;

        // ---- phase 2 (parallel): columns ------------------------------------------------------------
        t0 = now_s();
        const int64_t base = cols.size();
        std::vector<int64_t> rec0(segs.size() + 1, 0);
        for (size_t k = 0; k < segs.size(); ++k) rec0[k + 1] = rec0[k] + (int64_t)segs[k].starts.size();
        const size_t m = (size_t)rec0.back();
        cols.pos.resize((size_t)base + m); cols.tlen.resize((size_t)base + m);
        cols.flag.resize((size_t)base + m); cols.mapq.resize((size_t)base + m);
        cols.cigar_off.resize((size_t)base + m + 1);
        std::vector<int64_t> cig0(segs.size() + 1, cols.cigar_off[(size_t)base]);
        for (size_t k = 0; k < segs.size(); ++k) {
            int64_t t = 0;
            for (uint32_t v : segs[k].ncig) t += v;
            cig0[k + 1] = cig0[k] + t;
            cols.n_unplaced += segs[k].unplaced;
            for (const auto &rc_ : segs[k].rid_change)
                while ((int)cols.ref_off.size() <= rc_.second) cols.ref_off.push_back(base + rec0[k] + rc_.first);
        }
        cols.cigar.resize((size_t)cig0.back());
        parallel_for((int64_t)segs.size(), workers, [&](int64_t si, int) {
            const Segment &g = segs[(size_t)si];
            int64_t cacc = cig0[(size_t)si];
            for (size_t k = 0; k < g.starts.size(); ++k) {
                const uint8_t *r = d + g.starts[k];
                const uint8_t *c = r + 4;
                const size_t i = (size_t)(base + rec0[(size_t)si]) + k;
                cols.pos[i] = rdi32(c + 4);
                cols.mapq[i] = c[9];
                cols.flag[i] = rd16(c + 14);
                cols.tlen[i] = rdi32(c + 28);
                uint32_t *dst = cols.cigar.data() + cacc;
                if (g.any_cg && g.has_cg[k]) {
                    uint32_t real = 0;
                    const uint8_t *src = find_cg_tag(r, 4 + (size_t)rdi32(r), &real);
                    memcpy(dst, src, 4 * (size_t)real);
                } else if (g.ncig[k]) {
                    memcpy(dst, c + 32 + c[8], 4 * (size_t)g.ncig[k]);
                }
                cacc += g.ncig[k];
                cols.cigar_off[i + 1] = cacc;
            }
        });

        // ---- hand the unparsed tail to the next batch ------------------------------------------------
        T[3] += now_s() - t0;
        t0 = now_s();
        if (producer.joinable()) producer.join();
        T[1] += now_s() - t0;
        if (rc_next) return rc_next;
        const size_t tail = n - o;
        if (b1 < b2) {
            if (tail > kPrefix) {
                std::vector<uint8_t> big(tail + (nxt.size() - kPrefix));
                memcpy(big.data(), d + o, tail);
                memcpy(big.data() + tail, nxt.data() + kPrefix, nxt.size() - kPrefix);
                nxt.swap(big);
                begin = 0;
            } else {
                memcpy(nxt.data() + kPrefix - tail, d + o, tail);
                begin = kPrefix - tail;
            }
            cur.swap(nxt);
        } else if (tail) {
            return fail(BSIG_ERR_FORMAT, "truncated BAM record at the end of %s", path.c_str());
        }
        b0 = b1;
    }
    while ((int)cols.ref_off.size() < n_ref + 1) cols.ref_off.push_back(cols.size());
    T[4] = now_s() - t_begin;
    return 0;
}

int bam_read_header(const std::string &path, BamHeader &hdr)
{
    MappedFile f;
    if (f.open(path) != 0) return fail(BSIG_ERR_IO, "Fail to open BAM file %s", path.c_str());
    HostColumns dummy;
    BamParser parser(hdr, dummy, true);
    Inflater inf;
    uint64_t off = 0;
    std::vector<uint8_t> buf;
    while (off < f.size && parser.in_header()) {
        Block b;
        if (!parse_block(f, off, b)) return fail(BSIG_ERR_FORMAT, "malformed BGZF block in %s", path.c_str());
        buf.resize(b.isize);
        if (const int r = inflate_checked(inf, f.data, b, buf.data()))
            return fail(BSIG_ERR_FORMAT, r == 2 ? "BGZF block CRC mismatch (the file is damaged)" : "BGZF inflate failed");
        parser.set_limit(0);     // header only: never parse a record
        const int rc = parser.feed(buf.data(), buf.size());
        if (rc) return rc;
        off += b.csize;
    }
    if (parser.in_header()) return fail(BSIG_ERR_FORMAT, "truncated BAM header in %s", path.c_str());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// BAI (SAM spec 5.2)
// ---------------------------------------------------------------------------------------------
int bai_load(const std::string &bai_path, BaiIndex &idx)
{
    FILE *in = fopen(bai_path.c_str(), "rb");
    if (!in) return fail(BSIG_ERR_NOINDEX, "BAM indexing file is not available for file %s", bai_path.c_str());
    std::vector<uint8_t> d;
    {
        struct stat stt;
        if (fstat(fileno(in), &stt) == 0 && stt.st_size > 0) d.resize((size_t)stt.st_size);
        const size_t got = d.empty() ? 0 : fread(d.data(), 1, d.size(), in);
        d.resize(got);
        fclose(in);
    }
    size_t o = 0;
    auto need = [&](size_t n) { return o + n <= d.size(); };
    if (!need(8) || memcmp(d.data(), "BAI\1", 4) != 0) return fail(BSIG_ERR_FORMAT, "%s is not a BAI index", bai_path.c_str());
    const int32_t n_ref = rdi32(d.data() + 4);
    o = 8;
    // every reference takes at least 8 bytes (n_bin, n_intv): a count the file cannot hold is a
    // damaged index, not a reason to allocate gigabytes
    if (n_ref < 0 || (size_t)n_ref > (d.size() - 8) / 8) return fail(BSIG_ERR_FORMAT, "malformed BAI index");
    idx.refs.assign((size_t)n_ref, BaiRef());
    for (int r = 0; r < n_ref; ++r) {
        if (!need(4)) return fail(BSIG_ERR_FORMAT, "truncated BAI index");
        const int32_t n_bin = rdi32(d.data() + o); o += 4;
        for (int b = 0; b < n_bin; ++b) {
            if (!need(8)) return fail(BSIG_ERR_FORMAT, "truncated BAI index");
            const uint32_t bin = rd32(d.data() + o);
            const int32_t n_chunk = rdi32(d.data() + o + 4); o += 8;
            if (n_chunk < 0 || !need(16 * (size_t)n_chunk)) return fail(BSIG_ERR_FORMAT, "truncated BAI index");
            std::vector<BaiChunk> ch((size_t)n_chunk);
            for (int k = 0; k < n_chunk; ++k) { ch[(size_t)k].beg = rd64(d.data() + o); ch[(size_t)k].end = rd64(d.data() + o + 8); o += 16; }
            idx.refs[(size_t)r].bins.emplace_back(bin, std::move(ch));
        }
        std::sort(idx.refs[(size_t)r].bins.begin(), idx.refs[(size_t)r].bins.end(),
                  [](const auto &a, const auto &b) { return a.first < b.first; });
        if (!need(4)) return fail(BSIG_ERR_FORMAT, "truncated BAI index");
        const int32_t n_intv = rdi32(d.data() + o); o += 4;
        if (n_intv < 0 || !need(8 * (size_t)n_intv)) return fail(BSIG_ERR_FORMAT, "truncated BAI index");
        idx.refs[(size_t)r].linear.resize((size_t)n_intv);
        for (int k = 0; k < n_intv; ++k) { idx.refs[(size_t)r].linear[(size_t)k] = rd64(d.data() + o); o += 8; }
    }
    idx.n_no_coor = need(8) ? rd64(d.data() + o) : 0;
    return 0;
}

// CSI (CSIv1 spec): BGZF-compressed; magic "CSI\1", min_shift, depth, l_aux, aux[l_aux], n_ref, then per
// reference n_bin x { bin u32, loffset u64, n_chunk i32, chunks }, optionally n_no_coor.  There is no linear
// index: every bin carries the lowest virtual offset of a record that overlaps it (loffset).
int csi_load(const std::string &csi_path, BaiIndex &idx)
{
    gzFile g = gzopen(csi_path.c_str(), "rb");          // (zlib reads the concatenated gzip members of a BGZF file)
    if (!g) return fail(BSIG_ERR_NOINDEX, "BAM indexing file is not available for file %s", csi_path.c_str());
    std::vector<uint8_t> d;
    {
        uint8_t buf[1 << 16];
        int got;
        while ((got = gzread(g, buf, sizeof buf)) > 0) {
            d.insert(d.end(), buf, buf + got);
            if (d.size() > (1ull << 30)) { gzclose(g); return fail(BSIG_ERR_FORMAT, "%s: CSI index larger than 1 GB", csi_path.c_str()); }
        }
        gzclose(g);
        if (got < 0) return fail(BSIG_ERR_FORMAT, "%s is not a CSI index (inflate failed)", csi_path.c_str());
    }
    size_t o = 0;
    auto need = [&](size_t n) { return o + n <= d.size(); };
    if (!need(16) || memcmp(d.data(), "CSI\1", 4) != 0) return fail(BSIG_ERR_FORMAT, "%s is not a CSI index", csi_path.c_str());
    const int32_t min_shift = rdi32(d.data() + 4), depth = rdi32(d.data() + 8), l_aux = rdi32(d.data() + 12);
    o = 16;
    if (min_shift < 1 || min_shift > 30 || depth < 1 || depth > 9 || min_shift + 3 * depth > 44 || l_aux < 0 || !need((size_t)l_aux + 4))
        return fail(BSIG_ERR_FORMAT, "malformed CSI index");
    o += (size_t)l_aux;
    const int32_t n_ref = rdi32(d.data() + o); o += 4;
    if (n_ref < 0 || (size_t)n_ref > (d.size() - o) / 4) return fail(BSIG_ERR_FORMAT, "malformed CSI index");
    idx = BaiIndex();
    idx.min_shift = min_shift;
    idx.depth = depth;
    idx.refs.assign((size_t)n_ref, BaiRef());
    for (int r = 0; r < n_ref; ++r) {
        if (!need(4)) return fail(BSIG_ERR_FORMAT, "truncated CSI index");
        const int32_t n_bin = rdi32(d.data() + o); o += 4;
        if (n_bin < 0) return fail(BSIG_ERR_FORMAT, "malformed CSI index");
        BaiRef &R = idx.refs[(size_t)r];
        for (int b = 0; b < n_bin; ++b) {
            if (!need(16)) return fail(BSIG_ERR_FORMAT, "truncated CSI index");
            const uint32_t bin = rd32(d.data() + o);
            const uint64_t loffset = rd64(d.data() + o + 4);
            const int32_t n_chunk = rdi32(d.data() + o + 12); o += 16;
            if (n_chunk < 0 || !need(16 * (size_t)n_chunk)) return fail(BSIG_ERR_FORMAT, "truncated CSI index");
            std::vector<BaiChunk> ch((size_t)n_chunk);
            for (int k = 0; k < n_chunk; ++k) { ch[(size_t)k].beg = rd64(d.data() + o); ch[(size_t)k].end = rd64(d.data() + o + 8); o += 16; }
            R.bins.emplace_back(bin, std::move(ch));
            R.loff.emplace_back(bin, loffset);
        }
        std::sort(R.bins.begin(), R.bins.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        std::sort(R.loff.begin(), R.loff.end());
    }
    idx.n_no_coor = need(8) ? rd64(d.data() + o) : 0;
    return 0;
}

namespace {

// (the bins overlapping [beg, end) -- SAM spec 5.3 reg2bins, CSIv1 spec for any min_shift / depth -- are walked level
// by level in bai_region_chunks below)
uint32_t reg2bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

}  // namespace

std::vector<BaiChunk> bai_region_chunks(const BaiIndex &idx, const std::vector<Region> &regions)
{
    // candidate chunks of every region (htslib's iterator: bins + linear-index lower bound)
    std::vector<BaiChunk> chunks;
    for (const Region &rg : regions) {
        if (rg.rid < 0 || rg.rid >= (int)idx.refs.size()) continue;
        const BaiRef &R = idx.refs[(size_t)rg.rid];
        const int ms = idx.min_shift, depth = idx.depth;
        const int64_t n_leaf = 1ll << (3 * depth);                       // leaf bins (windows of 2^min_shift bp)
        const int64_t t_leaf = (n_leaf - 1) / 7;                          // number of the first leaf bin
        const int64_t beg = std::max<int64_t>(rg.beg, 0), end = std::min<int64_t>(rg.end, 1ll << (ms + 3 * depth));
        if (beg >= end) continue;
        uint64_t min_off = 0;
        if (!R.linear.empty()) {
            const size_t w = (size_t)(beg >> 14);
            min_off = w < R.linear.size() ? R.linear[w] : R.linear.back();
        } else if (!R.loff.empty()) {
            // CSI (htslib's hts_itr_query): the loffset of the leaf bin that holds `beg`, or of the nearest
            // bin to its left on its level, climbing to the parent where a level has none
            int64_t bin = t_leaf + (beg >> ms);
            uint64_t found = 0;
            bool have = false;
            while (true) {
                auto it = std::lower_bound(R.loff.begin(), R.loff.end(), std::make_pair((uint32_t)bin, (uint64_t)0));
                if (it != R.loff.end() && it->first == (uint32_t)bin) { found = it->second; have = true; break; }
                if (bin == 0) break;
                const int64_t parent = (bin - 1) >> 3, first = (parent << 3) + 1;
                bin = bin > first ? bin - 1 : parent;
            }
            min_off = have ? found : 0;
        }
        // Upper bound.  htslib's iterator stops at the first record with pos >= end; the chunk lists of
        // the coarse bins (reads that straddle a finer bin's border) run on to the end of their bin,
        // far behind the region.  The first read of a LEAF bin behind the region's last 16-kbp window
        // has pos >= end, and the file is sorted by pos: every read with pos < end lies in front of
        // that read's virtual offset, which is a record boundary.
        uint64_t max_off = ~0ull;
        {
            const int64_t w_end = (end - 1) >> ms;
            for (int64_t w = w_end + 1; w <= w_end + 64 && w < n_leaf; ++w) {
                const uint32_t leaf = (uint32_t)(t_leaf + w);
                auto it = std::lower_bound(R.bins.begin(), R.bins.end(), leaf, [](const auto &x, uint32_t v) { return x.first < v; });
                if (it == R.bins.end() || it->first != leaf || it->second.empty()) continue;
                uint64_t first = ~0ull;
                for (const BaiChunk &c : it->second) first = std::min(first, c.beg);
                max_off = first;
                break;
            }
        }
        // the bins that overlap [beg, end) (reg2bins above) are, level by level, one run of consecutive bin numbers:
        // the reference's PRESENT bins inside each run are walked, instead of every candidate being looked up -- a CSI
        // index may have a min_shift of 1, and one wide range would then list 2^30 candidate bins (the index file is
        // untrusted input on the same footing as the BAM)
        {
            int sft = ms + 3 * depth;
            int64_t t = 0;
            for (int l = 0; l <= depth; ++l, sft -= 3) {
                const int64_t b0 = t + (beg >> sft), b1 = t + ((end - 1) >> sft);
                auto it = std::lower_bound(R.bins.begin(), R.bins.end(), b0, [](const auto &x, int64_t v) { return (int64_t)x.first < v; });
                for (; it != R.bins.end() && (int64_t)it->first <= b1; ++it)
                    for (const BaiChunk &c : it->second)
                        if (c.end > min_off && c.beg < max_off) chunks.push_back(BaiChunk{c.beg, std::min(c.end, max_off)});
                t += 1ll << (3 * l);
            }
        }
    }
    std::sort(chunks.begin(), chunks.end(), [](const BaiChunk &a, const BaiChunk &b) { return a.beg < b.beg; });
    std::vector<BaiChunk> merged;
    for (const BaiChunk &c : chunks) {
        if (!merged.empty() && c.beg <= merged.back().end) merged.back().end = std::max(merged.back().end, c.end);
        else merged.push_back(c);
    }

    return merged;
}

int bam_decode_regions(const std::string &path, const BaiIndex &idx, const std::vector<Region> &regions,
                       int threads, BamHeader &hdr, HostColumns &cols)
{
    MappedFile f;
    if (f.open(path) != 0) return fail(BSIG_ERR_IO, "Fail to open BAM file %s", path.c_str());
    threads = n_threads(threads);
    int rc = bam_read_header(path, hdr);
    if (rc) return rc;
    cols = HostColumns();

    const std::vector<BaiChunk> merged = bai_region_chunks(idx, regions);
    {
        std::vector<std::pair<uint64_t, uint64_t>> spans;
        for (const BaiChunk &c : merged) spans.emplace_back(c.beg >> 16, (c.end >> 16) + 0x10000);
        populate_spans(f, spans);
    }

    // Every merged chunk is an independent job (it starts at a record boundary the index vouches
    // for): a worker inflates its blocks, parses its records into job-local columns, and the jobs
    // are concatenated in file order afterwards.
    struct Job {
        HostColumns c;
        int err = 0;
        std::string msg;
    };
    std::vector<Job> jobs(merged.size());
    const BamHeader &H = hdr;
    parallel_for((int64_t)merged.size(), threads, [&](int64_t ji, int) {
        static thread_local Inflater inf;
        Job &J = jobs[(size_t)ji];
        const BaiChunk &c = merged[(size_t)ji];
        const uint64_t cb = c.beg >> 16, ub = c.beg & 0xFFFF, ce = c.end >> 16, ue = c.end & 0xFFFF;
        BamHeader h = H;                       // the parser only reads names.size()
        BamParser parser(h, J.c, false);
        std::vector<uint8_t> buf;
        uint64_t off = cb, upos = 0;           // upos: stream position (relative to block cb) of buf's start
        bool first = true;
        auto bail = [&](int code) { J.err = code; J.msg = g_last_error; };
        // chunk end as a stream position: known once block ce has been reached
        uint64_t limit = ~0ull;
        if (ce == cb) limit = ue;
        parser.set_position(ub);
        parser.set_limit(limit);
        while (off < f.size) {
            if (limit != ~0ull && upos >= limit && !parser.pending()) break;
            Block b;
            if (!parse_block(f, off, b)) { bail(fail(BSIG_ERR_FORMAT, "BAI points at a malformed BGZF block in %s", path.c_str())); return; }
            if (b.coff == ce && limit == ~0ull) { limit = upos + ue; parser.set_limit(limit); }
            if (limit != ~0ull && upos >= limit && !parser.pending()) break;
            buf.resize(b.isize);
            if (const int r = inflate_checked(inf, f.data, b, buf.data())) {
                bail(fail(BSIG_ERR_FORMAT, r == 2 ? "BGZF block CRC mismatch (the file is damaged)" : "BGZF inflate failed"));
                return;
            }
            size_t skip = 0;
            if (first) {
                if (ub > buf.size()) { bail(fail(BSIG_ERR_FORMAT, "BAI offset beyond its BGZF block in %s", path.c_str())); return; }
                skip = (size_t)ub;
                first = false;
            }
            const int rc2 = parser.feed(buf.data() + skip, buf.size() - skip);
            if (rc2) { bail(rc2); return; }
            upos += b.isize;
            off += b.csize;
            if (parser.reached_limit()) break;
        }
    });
    for (Job &J : jobs)
        if (J.err) return fail(J.err, "%s", J.msg.c_str());

    // concatenate in file order
    std::vector<int64_t> rec0(jobs.size() + 1, 0), cig0(jobs.size() + 1, 0);
    for (size_t k = 0; k < jobs.size(); ++k) {
        rec0[k + 1] = rec0[k] + jobs[k].c.size();
        cig0[k + 1] = cig0[k] + (int64_t)jobs[k].c.cigar.size();
        cols.n_unplaced += jobs[k].c.n_unplaced;
        while (cols.ref_off.size() < jobs[k].c.ref_off.size())
            cols.ref_off.push_back(rec0[k] + jobs[k].c.ref_off[cols.ref_off.size()]);
    }
    const size_t m = (size_t)rec0.back();
    cols.pos.resize(m); cols.tlen.resize(m); cols.flag.resize(m); cols.mapq.resize(m);
    cols.cigar_off.resize(m + 1);
    cols.cigar.resize((size_t)cig0.back());
    cols.cigar_off[0] = 0;
    parallel_for((int64_t)jobs.size(), threads, [&](int64_t k, int) {
        const HostColumns &c = jobs[(size_t)k].c;
        const size_t n = (size_t)c.size(), r0 = (size_t)rec0[(size_t)k];
        if (!n) return;
        memcpy(cols.pos.data() + r0, c.pos.data(), n * sizeof(int32_t));
        memcpy(cols.tlen.data() + r0, c.tlen.data(), n * sizeof(int32_t));
        memcpy(cols.flag.data() + r0, c.flag.data(), n * sizeof(uint16_t));
        memcpy(cols.mapq.data() + r0, c.mapq.data(), n);
        memcpy(cols.cigar.data() + cig0[(size_t)k], c.cigar.data(), c.cigar.size() * sizeof(uint32_t));
        for (size_t i = 0; i < n; ++i) cols.cigar_off[r0 + i + 1] = cig0[(size_t)k] + c.cigar_off[i + 1];
    });
    const int n_ref = (int)hdr.names.size();
    while ((int)cols.ref_off.size() < n_ref + 1) cols.ref_off.push_back(cols.size());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// writer: BGZF blocks + BAI
// ---------------------------------------------------------------------------------------------
struct BamWriter::Impl {
    FILE *fp = nullptr;
    std::string path;
    int level = 6;
    std::vector<uint8_t> ubuf;               // current uncompressed block
    uint64_t coff = 0;                       // file offset of the current block
    Deflater *defl = nullptr;
    int n_ref = 0;
    // index under construction
    struct RefIdx {
        std::map<uint32_t, std::vector<BaiChunk>> bins;
        std::vector<uint64_t> linear;
        uint64_t off_beg = ~0ull, off_end = 0, n_mapped = 0, n_unmapped = 0;
        uint32_t last_bin = ~0u;
        std::vector<BaiChunk> *last_chunks = nullptr;     // bins[last_bin] (map nodes do not move)
    };
    std::vector<RefIdx> ridx;
    uint64_t n_no_coor = 0;
    int last_rid = -1, last_pos = -1;
    bool sorted = true;

    static constexpr size_t kBlockData = 0xff00;   // htslib's BGZF_BLOCK_SIZE

    uint64_t tell() const { return coff << 16 | (uint64_t)ubuf.size(); }

    int flush_block()
    {
        if (ubuf.empty()) return 0;
        return write_block(ubuf.data(), ubuf.size());
    }

    int write_block(const uint8_t *data, size_t n)
    {
        uint8_t out[0x10000 + 64];
        size_t clen = n ? defl->run(data, n, out + 18, 0x10000 - 18 - 8) : 0;
        if (n && clen == 0) {
            // incompressible: store with level 0 through zlib
            Deflater store(0);
            clen = store.run(data, n, out + 18, 0x10000 + 64 - 18 - 8);
            if (clen == 0 || clen + 26 > 0x10000) return fail(BSIG_ERR_IO, "BGZF block does not fit");
        }
        if (n == 0) { out[18] = 3; out[19] = 0; clen = 2; }      // empty final deflate block
        static const uint8_t head[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
        memcpy(out, head, 12);
        out[12] = 'B'; out[13] = 'C'; out[14] = 2; out[15] = 0;
        const uint16_t bsize = (uint16_t)(clen + 25);
        memcpy(out + 16, &bsize, 2);
        const uint32_t crc = crc32_of(data, n), isz = (uint32_t)n;
        memcpy(out + 18 + clen, &crc, 4);
        memcpy(out + 22 + clen, &isz, 4);
        const size_t total = clen + 26;
        if (fwrite(out, 1, total, fp) != total) return fail(BSIG_ERR_IO, "write to %s failed", path.c_str());
        coff += total;
        ubuf.clear();
        return 0;
    }

    int append(const uint8_t *d, size_t n, bool keep_together)
    {
        if (keep_together && ubuf.size() + n > kBlockData && !ubuf.empty()) {
            const int rc = flush_block();
            if (rc) return rc;
        }
        while (n) {
            const size_t take = std::min(n, kBlockData - ubuf.size());
            ubuf.insert(ubuf.end(), d, d + take);
            d += take; n -= take;
            if (ubuf.size() == kBlockData) { const int rc = flush_block(); if (rc) return rc; }
        }
        return 0;
    }

    void index_push(int rid, int64_t beg, int64_t end, uint64_t v0, uint64_t v1, bool mapped)
    {
        if (rid < 0) { ++n_no_coor; return; }
        // (a record placed on a reference without a position, POS 0 in SAM, is filed under position 0;
        // the callers have refused coordinates a BAI cannot address)
        if (beg < 0) beg = 0;
        if (end <= beg) end = beg + 1;
        RefIdx &R = ridx[(size_t)rid];
        const uint32_t bin = reg2bin(beg, end);
        if (bin == R.last_bin && R.last_chunks && !R.last_chunks->empty()) R.last_chunks->back().end = v1;
        else {
            R.last_chunks = &R.bins[bin];
            R.last_chunks->push_back(BaiChunk{v0, v1});
        }
        R.last_bin = bin;
        const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
        if (R.linear.size() <= w1) R.linear.resize(w1 + 1, ~0ull);
        for (size_t w = w0; w <= w1; ++w)
            if (R.linear[w] == ~0ull) R.linear[w] = v0;
        R.off_beg = std::min(R.off_beg, v0);
        R.off_end = std::max(R.off_end, v1);
        if (mapped) ++R.n_mapped; else ++R.n_unmapped;
    }

    int write_bai()
    {
        const std::string bp = path + ".bai";
        FILE *o = fopen(bp.c_str(), "wb");
        if (!o) return fail(BSIG_ERR_IO, "cannot write %s", bp.c_str());
        auto w32 = [&](uint32_t v) { fwrite(&v, 4, 1, o); };
        auto w64 = [&](uint64_t v) { fwrite(&v, 8, 1, o); };
        fwrite("BAI\1", 1, 4, o);
        w32((uint32_t)n_ref);
        for (RefIdx &R : ridx) {
            // linear index: windows without a read inherit the next window's offset (as htslib does)
            for (size_t k = R.linear.size(); k-- > 0;)
                if (R.linear[k] == ~0ull) R.linear[k] = k + 1 < R.linear.size() ? R.linear[k + 1] : R.off_end;
            const bool any = R.n_mapped + R.n_unmapped > 0;
            w32((uint32_t)(R.bins.size() + (any ? 1 : 0)));
            for (auto &kv : R.bins) {
                w32(kv.first);
                w32((uint32_t)kv.second.size());
                for (const BaiChunk &c : kv.second) { w64(c.beg); w64(c.end); }
            }
            if (any) {          // pseudo-bin 37450: file span + mapped/unmapped counts
                w32(37450); w32(2);
                w64(R.off_beg); w64(R.off_end); w64(R.n_mapped); w64(R.n_unmapped);
            }
            w32((uint32_t)R.linear.size());
            for (uint64_t v : R.linear) w64(v);
        }
        w64(n_no_coor);
        if (fclose(o) != 0) return fail(BSIG_ERR_IO, "cannot write %s", bp.c_str());
        return 0;
    }
};

BamWriter::BamWriter() : p_(new Impl) {}
BamWriter::~BamWriter()
{
    if (p_->fp) fclose(p_->fp);
    delete p_->defl;
    delete p_;
}

int BamWriter::open(const std::string &path, const BamHeader &hdr, int level)
{
    p_->fp = fopen(path.c_str(), "wb");
    if (!p_->fp) return fail(BSIG_ERR_IO, "Fail to open BAM file %s", path.c_str());
    p_->path = path;
    p_->level = level;
    p_->defl = new Deflater(level);
    p_->n_ref = (int)hdr.names.size();
    p_->ridx.assign(hdr.names.size(), Impl::RefIdx());
    std::vector<uint8_t> h;
    auto put32 = [&](int32_t v) { const uint8_t *b = (const uint8_t *)&v; h.insert(h.end(), b, b + 4); };
    h.insert(h.end(), {'B', 'A', 'M', 1});
    put32((int32_t)hdr.text.size());
    h.insert(h.end(), hdr.text.begin(), hdr.text.end());
    put32((int32_t)hdr.names.size());
    for (size_t r = 0; r < hdr.names.size(); ++r) {
        put32((int32_t)hdr.names[r].size() + 1);
        h.insert(h.end(), hdr.names[r].begin(), hdr.names[r].end());
        h.push_back(0);
        put32(hdr.lens[r]);
    }
    int rc = p_->append(h.data(), h.size(), false);
    if (rc) return rc;
    return p_->flush_block();      // records start in a fresh block, as samtools writes them
}

static int64_t cigar_rlen(const uint32_t *cig, int n, uint16_t flag)
{
    int64_t rlen = 0;
    if (!(flag & 0x4))
        for (int k = 0; k < n; ++k)
            if ((0x18Du >> (cig[k] & 0xF)) & 1u) rlen += cig[k] >> 4;
    return rlen ? rlen : 1;
}

static const char *kSeqNt16 = "=ACMGRSVTWYHKDBN";

int BamWriter::write(const BamRecord &r)
{
    std::vector<uint8_t> b;
    const size_t l_name = r.name.size() + 1;
    if (l_name > 255) return fail(BSIG_ERR_FORMAT, "read name longer than 254 characters");
    if (r.cigar.size() > 65535) return fail(BSIG_ERR_FORMAT, "more than 65535 CIGAR operations are not supported by the writer");
    const size_t l_seq = r.seq.size();
    const size_t bs = 32 + l_name + 4 * r.cigar.size() + (l_seq + 1) / 2 + l_seq + r.aux.size();
    const int64_t endpos = (int64_t)r.pos + cigar_rlen(r.cigar.data(), (int)r.cigar.size(), r.flag);
    if (r.rid >= 0 && (r.pos < -1 || endpos > (1ll << 29)))
        return fail(BSIG_ERR_FORMAT, "position %d (end %lld) cannot be addressed by a BAI index (0 .. 2^29)", r.pos, (long long)endpos);
    if (bs > (1u << 28)) return fail(BSIG_ERR_FORMAT, "record too large");
    b.resize(4 + bs);
    const int32_t bs32 = (int32_t)bs;
    const uint16_t bin = (uint16_t)reg2bin(std::max(r.pos, 0), std::max<int64_t>(endpos, 1)), ncig = (uint16_t)r.cigar.size();
    const int32_t lseq32 = (int32_t)l_seq;
    uint8_t *p = b.data();
    memcpy(p, &bs32, 4); memcpy(p + 4, &r.rid, 4); memcpy(p + 8, &r.pos, 4);
    p[12] = (uint8_t)l_name; p[13] = r.mapq;
    memcpy(p + 14, &bin, 2); memcpy(p + 16, &ncig, 2); memcpy(p + 18, &r.flag, 2);
    memcpy(p + 20, &lseq32, 4); memcpy(p + 24, &r.next_rid, 4); memcpy(p + 28, &r.next_pos, 4); memcpy(p + 32, &r.tlen, 4);
    memcpy(p + 36, r.name.c_str(), l_name);
    size_t o = 36 + l_name;
    if (!r.cigar.empty()) memcpy(p + o, r.cigar.data(), 4 * r.cigar.size());
    o += 4 * r.cigar.size();
    for (size_t i = 0; i < l_seq; ++i) {
        const char *q = strchr(kSeqNt16, toupper((unsigned char)r.seq[i]));
        const uint8_t code = q && *q ? (uint8_t)(q - kSeqNt16) : 15;
        if (i & 1) p[o + i / 2] |= code; else p[o + i / 2] = (uint8_t)(code << 4);
    }
    o += (l_seq + 1) / 2;
    for (size_t i = 0; i < l_seq; ++i) p[o + i] = r.qual.size() == l_seq ? (uint8_t)(r.qual[i] - 33) : 0xFF;
    o += l_seq;
    if (!r.aux.empty()) memcpy(p + o, r.aux.data(), r.aux.size());

    if (r.rid >= 0 && (r.rid < p_->last_rid || (r.rid == p_->last_rid && r.pos < p_->last_pos))) p_->sorted = false;
    if (r.rid >= 0) { p_->last_rid = r.rid; p_->last_pos = r.pos; }
    if (p_->ubuf.size() + b.size() > Impl::kBlockData && !p_->ubuf.empty()) { const int rc = p_->flush_block(); if (rc) return rc; }
    const uint64_t v0 = p_->tell();
    const int rc = p_->append(b.data(), b.size(), false);
    if (rc) return rc;
    if (r.rid >= p_->n_ref) return fail(BSIG_ERR_FORMAT, "record with refID %d but only %d references", r.rid, p_->n_ref);
    p_->index_push(r.rid, r.pos, endpos, v0, p_->tell(), !(r.flag & 0x4));
    return 0;
}

int BamWriter::write_core(int32_t rid, int32_t pos, uint16_t flag, uint8_t mapq, int32_t tlen,
                          const uint32_t *cigar, int n_cigar)
{
    uint8_t b[4 + 32 + 2 + 4 * 64];
    if (n_cigar > 64) {
        BamRecord r;
        r.rid = rid; r.pos = pos; r.flag = flag; r.mapq = mapq; r.tlen = tlen; r.name = "*";
        r.cigar.assign(cigar, cigar + n_cigar);
        return write(r);
    }
    const int32_t bs = 32 + 2 + 4 * n_cigar;
    const int64_t endpos = (int64_t)pos + cigar_rlen(cigar, n_cigar, flag);
    if (rid >= 0 && (pos < -1 || endpos > (1ll << 29)))
        return fail(BSIG_ERR_FORMAT, "position %d (end %lld) cannot be addressed by a BAI index (0 .. 2^29)", pos, (long long)endpos);
    const uint16_t bin = (uint16_t)reg2bin(std::max(pos, 0), std::max<int64_t>(endpos, 1)), ncig = (uint16_t)n_cigar;
    const int32_t zero = 0, m1 = -1;
    memcpy(b, &bs, 4); memcpy(b + 4, &rid, 4); memcpy(b + 8, &pos, 4);
    b[12] = 2; b[13] = mapq;
    memcpy(b + 14, &bin, 2); memcpy(b + 16, &ncig, 2); memcpy(b + 18, &flag, 2);
    memcpy(b + 20, &zero, 4); memcpy(b + 24, &m1, 4); memcpy(b + 28, &m1, 4); memcpy(b + 32, &tlen, 4);
    b[36] = '*'; b[37] = 0;
    if (n_cigar) memcpy(b + 38, cigar, 4 * (size_t)n_cigar);
    const size_t len = 4 + (size_t)bs;
    if (rid >= 0 && (rid < p_->last_rid || (rid == p_->last_rid && pos < p_->last_pos))) p_->sorted = false;
    if (rid >= 0) { p_->last_rid = rid; p_->last_pos = pos; }
    if (p_->ubuf.size() + len > Impl::kBlockData && !p_->ubuf.empty()) { const int rc = p_->flush_block(); if (rc) return rc; }
    const uint64_t v0 = p_->tell();
    const int rc = p_->append(b, len, false);
    if (rc) return rc;
    if (rid >= p_->n_ref) return fail(BSIG_ERR_FORMAT, "record with refID %d but only %d references", rid, p_->n_ref);
    p_->index_push(rid, pos, endpos, v0, p_->tell(), !(flag & 0x4));
    return 0;
}

// The columnar fast path for whole files: the same bytes write_core() produces record by record
// (same block cuts, same index), with the BGZF blocks built and deflated by the worker pool.
namespace {
inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// 64 quality values with the histogram of clip(round(40 - |N(0, 6)|), 2, 41): mostly 36-40 with a tail,
// about 3 bits per base like binned Illumina qualities
const uint8_t kQualTable[64] = {40, 40, 40, 40, 40, 39, 39, 39, 39, 39, 39, 39, 39, 38, 38, 38, 38, 38, 38, 38, 37, 37,
                                37, 37, 37, 37, 36, 36, 36, 36, 36, 36, 35, 35, 35, 35, 35, 34, 34, 34, 34, 33, 33, 33,
                                33, 32, 32, 32, 31, 31, 31, 30, 30, 29, 29, 28, 28, 27, 26, 25, 24, 22, 19, 12};
}  // namespace

int BamWriter::write_columns(int32_t n_ref, const int64_t *ref_off, const int32_t *pos, const uint16_t *flag,
                             const uint8_t *mapq, const int32_t *tlen, const int64_t *cigar_off,
                             const uint32_t *cigar, int threads, int l_seq, uint64_t seed)
{
    Impl &W = *p_;
    if (l_seq < 0 || l_seq > 4096) return fail(BSIG_ERR_ARG, "l_seq must be 0 .. 4096");
    // bytes of a record with nc CIGAR operations: bare (name "*", no sequence: 52 bytes for one
    // operation) or real-shaped (read name, l_seq bases + qualities, an NM tag: 204 bytes for 100 bp)
    const size_t fixed = l_seq ? 4 + 32 + 10 + (size_t)(l_seq + 1) / 2 + (size_t)l_seq + 4 : 38;
    if (n_ref > W.n_ref) return fail(BSIG_ERR_FORMAT, "columns with %d references but the header has %d", n_ref, W.n_ref);
    const int64_t n = n_ref > 0 ? ref_off[n_ref] : 0;
    if (n == 0) return 0;
    // anything unusual takes the record-by-record path
    bool plain = W.ubuf.empty();
    for (int64_t i = 0; i < n && plain; ++i)
        plain = cigar_off[i + 1] - cigar_off[i] <= 64 && cigar_off[i + 1] >= cigar_off[i] && pos[i] >= -1;
    if (!plain && l_seq) return fail(BSIG_ERR_ARG, "synthetic sequences need at most 64 CIGAR operations per read");
    if (!plain) {
        for (int r = 0; r < n_ref; ++r)
            for (int64_t i = ref_off[r]; i < ref_off[r + 1]; ++i) {
                const int rc = write_core(r, pos[i], flag[i], mapq[i], tlen[i], cigar + cigar_off[i], (int)(cigar_off[i + 1] - cigar_off[i]));
                if (rc) return rc;
            }
        return 0;
    }
    // block cuts: a record never straddles blocks (write_core flushes first), a block that reaches
    // kBlockData exactly is closed at once
    std::vector<int64_t> first;            // first record of every block, + n
    {
        size_t fill = 0;
        first.push_back(0);
        for (int64_t i = 0; i < n; ++i) {
            const size_t len = fixed + 4 * (size_t)(cigar_off[i + 1] - cigar_off[i]);
            if (fill + len > Impl::kBlockData && fill) { first.push_back(i); fill = 0; }
            fill += len;
            if (fill == Impl::kBlockData && i + 1 < n) { first.push_back(i + 1); fill = 0; }
        }
        first.push_back(n);
    }
    const int64_t n_blocks = (int64_t)first.size() - 1;
    const bool tim = getenv("BAMSIGNALS_WRITER_TIMING") != nullptr;
    double t_par = 0, t_ser = 0, t_mark = tim ? now_s() : 0;
    auto rid_of = [&](int64_t i, int r) { while (ref_off[r + 1] <= i) ++r; return r; };
    constexpr int64_t kBatch = 4096;
    constexpr size_t kSlot = 0x10000 + 64;
    std::vector<uint8_t> cbuf((size_t)std::min(kBatch, n_blocks) * kSlot);
    std::vector<uint32_t> csize((size_t)std::min(kBatch, n_blocks));
    const int nt = n_threads(threads);
    std::atomic<int> err(0);
    int r_cur = 0;
    for (int64_t b0 = 0; b0 < n_blocks; b0 += kBatch) {
        const int64_t nb = std::min(kBatch, n_blocks - b0);
        parallel_for(nb, nt, [&](int64_t k, int) {
            thread_local std::vector<uint8_t> u;
            thread_local Deflater *dfl = nullptr;
            thread_local int dfl_level = -1;
            if (!dfl || dfl_level != W.level) { delete dfl; dfl = new Deflater(W.level); dfl_level = W.level; }
            const int64_t i0 = first[(size_t)(b0 + k)], i1 = first[(size_t)(b0 + k + 1)];
            u.clear();
            int r = 0;
            {   // reference of the block's first record
                int lo = 0, hi = n_ref;
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ref_off[mid] <= i0) lo = mid; else hi = mid; }
                r = lo;
            }
            for (int64_t i = i0; i < i1; ++i) {
                r = rid_of(i, r);
                const int nc = (int)(cigar_off[i + 1] - cigar_off[i]);
                const uint32_t *cg = cigar + cigar_off[i];
                uint8_t b[4 + 32 + 10 + 4 * 64];
                const int32_t bs = (int32_t)(fixed - 4) + 4 * nc, rid = r, p0 = pos[i], zero = 0, m1 = -1, tl = tlen[i];
                const uint16_t fl = flag[i];
                const int64_t endpos = (int64_t)p0 + cigar_rlen(cg, nc, fl);
                if (endpos > (1ll << 29)) { err.store(2); return; }
                const uint16_t bin = (uint16_t)reg2bin(std::max(p0, 0), std::max<int64_t>(endpos, 1)), ncig = (uint16_t)nc;
                memcpy(b, &bs, 4); memcpy(b + 4, &rid, 4); memcpy(b + 8, &p0, 4);
                b[13] = mapq[i];
                memcpy(b + 14, &bin, 2); memcpy(b + 16, &ncig, 2); memcpy(b + 18, &fl, 2);
                memcpy(b + 24, &m1, 4); memcpy(b + 28, &m1, 4); memcpy(b + 32, &tl, 4);
                if (!l_seq) {
                    b[12] = 2;
                    memcpy(b + 20, &zero, 4);
                    b[36] = '*'; b[37] = 0;
                    if (nc) memcpy(b + 38, cg, 4 * (size_t)nc);
                    u.insert(u.end(), b, b + 4 + bs);
                    continue;
                }
                // real-shaped: "q" + 8 hex digits, random bases (4-bit codes of A C G T), qualities from
                // kQualTable, NM:C -- all from a counter-based generator, so the file depends on (seed, i) only
                b[12] = 10;
                memcpy(b + 20, &l_seq, 4);
                static const char hexd[] = "0123456789abcdef";
                b[36] = 'q';
                for (int d = 0; d < 8; ++d) b[37 + d] = (uint8_t)hexd[((uint64_t)i >> (4 * (7 - d))) & 15];
                b[45] = 0;
                if (nc) memcpy(b + 46, cg, 4 * (size_t)nc);
                u.insert(u.end(), b, b + 46 + 4 * nc);
                const size_t at = u.size();
                u.resize(at + (size_t)(l_seq + 1) / 2 + (size_t)l_seq + 4);
                uint8_t *sq = u.data() + at, *ql = sq + (l_seq + 1) / 2;
                uint64_t ctr = seed ^ ((uint64_t)i * 0xD1342543DE82EF95ull);
                for (int k = 0; k < (l_seq + 1) / 2; k += 16) {          // 16 bytes = 32 bases per draw
                    const uint64_t rnd = splitmix64(ctr++);
                    for (int q = 0; q < 16 && k + q < (l_seq + 1) / 2; ++q) {
                        const unsigned two = (unsigned)(rnd >> (4 * q)) & 15u;
                        sq[k + q] = (uint8_t)((1u << (two & 3)) << 4 | (1u << (two >> 2)));
                    }
                }
                if (l_seq & 1) sq[l_seq / 2] &= 0xF0;
                for (int k = 0; k < l_seq; k += 10) {                    // 6 bits per quality
                    const uint64_t rnd = splitmix64(ctr++);
                    for (int q = 0; q < 10 && k + q < l_seq; ++q) ql[k + q] = kQualTable[(rnd >> (6 * q)) & 63];
                }
                uint8_t *aux = ql + l_seq;
                aux[0] = 'N'; aux[1] = 'M'; aux[2] = 'C'; aux[3] = (uint8_t)(splitmix64(ctr) % 5);
            }
            uint8_t *out = cbuf.data() + (size_t)k * kSlot;
            size_t clen = dfl->run(u.data(), u.size(), out + 18, 0x10000 - 18 - 8);
            if (clen == 0) {
                Deflater store(0);
                clen = store.run(u.data(), u.size(), out + 18, kSlot - 18 - 8);
                if (clen == 0 || clen + 26 > 0x10000) { err.store(1); return; }
            }
            static const uint8_t head[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
            memcpy(out, head, 12);
            out[12] = 'B'; out[13] = 'C'; out[14] = 2; out[15] = 0;
            const uint16_t bsize = (uint16_t)(clen + 25);
            memcpy(out + 16, &bsize, 2);
            const uint32_t crc = crc32_of(u.data(), u.size()), isz = (uint32_t)u.size();
            memcpy(out + 18 + clen, &crc, 4);
            memcpy(out + 22 + clen, &isz, 4);
            csize[(size_t)k] = (uint32_t)(clen + 26);
        });
        if (err.load() == 2) return fail(BSIG_ERR_FORMAT, "a read ends beyond 2^29: a BAI index cannot address it");
        if (err.load()) return fail(BSIG_ERR_IO, "BGZF block does not fit");
        if (tim) { const double t = now_s(); t_par += t - t_mark; t_mark = t; }
        // in file order: the bytes, then the index entries of the batch's records
        for (int64_t k = 0; k < nb; ++k) {
            const size_t total = csize[(size_t)k];
            if (fwrite(cbuf.data() + (size_t)k * kSlot, 1, total, W.fp) != total) return fail(BSIG_ERR_IO, "write to %s failed", W.path.c_str());
            const uint64_t c0 = W.coff, c1 = W.coff + total;
            const int64_t i0 = first[(size_t)(b0 + k)], i1 = first[(size_t)(b0 + k + 1)];
            size_t uoff = 0;
            for (int64_t i = i0; i < i1; ++i) {
                r_cur = rid_of(i, r_cur);
                const int nc = (int)(cigar_off[i + 1] - cigar_off[i]);
                const size_t len = fixed + 4 * (size_t)nc;
                if (r_cur < W.last_rid || (r_cur == W.last_rid && pos[i] < W.last_pos)) W.sorted = false;
                W.last_rid = r_cur; W.last_pos = pos[i];
                const int64_t endpos = (int64_t)pos[i] + cigar_rlen(cigar + cigar_off[i], nc, flag[i]);
                const uint64_t v0 = c0 << 16 | (uint64_t)uoff;
                uoff += len;
                // (a record that fills its block to the brim ends at the start of the next one, as tell() reports)
                const uint64_t v1 = uoff == Impl::kBlockData ? c1 << 16 : c0 << 16 | (uint64_t)uoff;
                W.index_push(r_cur, pos[i], endpos, v0, v1, !(flag[i] & 0x4));
            }
            W.coff = c1;
        }
        if (tim) { const double t = now_s(); t_ser += t - t_mark; t_mark = t; }
    }
    if (tim) fprintf(stderr, "write_columns: %lld blocks, build+deflate %.3f s (%d threads), write+index %.3f s\n", (long long)n_blocks, t_par, nt, t_ser);
    // the last block stays "open" in the record-by-record writer until close(); here it is already
    // on disk, which yields the same file: ubuf is empty and close() writes the EOF marker next
    return 0;
}

int BamWriter::close()
{
    if (!p_->fp) return 0;
    int rc = p_->flush_block();
    if (rc == 0) rc = p_->write_block(nullptr, 0);         // the 28-byte EOF marker
    if (fclose(p_->fp) != 0 && rc == 0) rc = fail(BSIG_ERR_IO, "closing %s failed", p_->path.c_str());
    p_->fp = nullptr;
    if (rc) return rc;
    if (!p_->sorted) return fail(BSIG_ERR_FORMAT, "cannot index %s: records are not sorted by coordinate", p_->path.c_str());
    return p_->write_bai();
}

// ---------------------------------------------------------------------------------------------
// SAM text -> BAM + BAI (writeSamAsBamAndIndex, ref: src/bamsignals.cpp:496-534)
// ---------------------------------------------------------------------------------------------
namespace {

bool parse_cigar(const std::string &s, std::vector<uint32_t> &out)
{
    out.clear();
    if (s == "*") return true;
    uint64_t len = 0;
    bool have = false;
    for (char ch : s) {
        if (ch >= '0' && ch <= '9') { len = len * 10 + (uint64_t)(ch - '0'); have = true; continue; }
        const char *q = strchr("MIDNSHP=X", ch);
        if (!q || !have || len >= (1u << 28)) return false;
        out.push_back((uint32_t)(len << 4 | (uint32_t)(q - "MIDNSHP=X")));
        len = 0; have = false;
    }
    return !have;
}

template <typename T>
void put(std::vector<uint8_t> &v, T x)
{
    const uint8_t *b = (const uint8_t *)&x;
    v.insert(v.end(), b, b + sizeof(T));
}

bool parse_aux(const std::string &f, std::vector<uint8_t> &aux)
{
    if (f.size() < 5 || f[2] != ':' || f[4] != ':') return false;
    const char ty = f[3];
    const std::string val = f.substr(5);
    aux.push_back((uint8_t)f[0]); aux.push_back((uint8_t)f[1]);
    if (ty == 'A') { if (val.size() != 1) return false; aux.push_back('A'); aux.push_back((uint8_t)val[0]); }
    else if (ty == 'i') {
        char *e = nullptr;
        const long long x = strtoll(val.c_str(), &e, 10);
        if (!e || *e) return false;
        if (x >= 0) {
            if (x <= 0xFF) { aux.push_back('C'); put<uint8_t>(aux, (uint8_t)x); }
            else if (x <= 0xFFFF) { aux.push_back('S'); put<uint16_t>(aux, (uint16_t)x); }
            else { aux.push_back('I'); put<uint32_t>(aux, (uint32_t)x); }
        } else {
            if (x >= -128) { aux.push_back('c'); put<int8_t>(aux, (int8_t)x); }
            else if (x >= -32768) { aux.push_back('s'); put<int16_t>(aux, (int16_t)x); }
            else { aux.push_back('i'); put<int32_t>(aux, (int32_t)x); }
        }
    } else if (ty == 'f') { aux.push_back('f'); put<float>(aux, strtof(val.c_str(), nullptr)); }
    else if (ty == 'Z' || ty == 'H') { aux.push_back((uint8_t)ty); aux.insert(aux.end(), val.begin(), val.end()); aux.push_back(0); }
    else if (ty == 'B') {
        if (val.empty()) return false;
        const char sub = val[0];
        std::vector<std::string> parts;
        std::stringstream ss(val.size() > 1 ? val.substr(2) : "");
        for (std::string t; std::getline(ss, t, ',');) parts.push_back(t);
        aux.push_back('B'); aux.push_back((uint8_t)sub); put<uint32_t>(aux, (uint32_t)parts.size());
        for (const std::string &t : parts) {
            switch (sub) {
            case 'c': put<int8_t>(aux, (int8_t)atoi(t.c_str())); break;
            case 'C': put<uint8_t>(aux, (uint8_t)atoi(t.c_str())); break;
            case 's': put<int16_t>(aux, (int16_t)atoi(t.c_str())); break;
            case 'S': put<uint16_t>(aux, (uint16_t)atoi(t.c_str())); break;
            case 'i': put<int32_t>(aux, (int32_t)atoll(t.c_str())); break;
            case 'I': put<uint32_t>(aux, (uint32_t)atoll(t.c_str())); break;
            case 'f': put<float>(aux, strtof(t.c_str(), nullptr)); break;
            default: return false;
            }
        }
    } else return false;
    return true;
}

}  // namespace

int sam_to_bam_and_index(const std::string &sam_path, const std::string &bam_path)
{
    std::ifstream in(sam_path);
    if (!in) return fail(BSIG_ERR_IO, "Fail to open SAM file %s", sam_path.c_str());
    BamHeader hdr;
    std::string line;
    std::vector<std::string> body;
    bool header_done = false;
    BamWriter w;
    int64_t lineno = 0;
    auto start_writer = [&]() -> int {
        header_done = true;
        return w.open(bam_path, hdr);
    };
    while (std::getline(in, line)) {
        ++lineno;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '@' && !header_done) {
            hdr.text += line + "\n";
            if (line.compare(0, 3, "@SQ") == 0) {
                std::string sn; long long ln = -1;
                std::stringstream ss(line);
                for (std::string t; std::getline(ss, t, '\t');) {
                    if (t.compare(0, 3, "SN:") == 0) sn = t.substr(3);
                    else if (t.compare(0, 3, "LN:") == 0) ln = atoll(t.c_str() + 3);
                }
                if (sn.empty() || ln < 0) return fail(BSIG_ERR_FORMAT, "malformed @SQ line %lld in %s", (long long)lineno, sam_path.c_str());
                hdr.names.push_back(sn); hdr.lens.push_back((int32_t)ln);
            }
            continue;
        }
        if (!header_done) { const int rc = start_writer(); if (rc) return rc; }
        std::vector<std::string> f;
        {
            size_t a = 0;
            for (;;) {
                const size_t t = line.find('\t', a);
                f.push_back(line.substr(a, t == std::string::npos ? std::string::npos : t - a));
                if (t == std::string::npos) break;
                a = t + 1;
            }
        }
        if (f.size() < 11) return fail(BSIG_ERR_FORMAT, "SAM line %lld of %s has fewer than 11 fields", (long long)lineno, sam_path.c_str());
        BamRecord r;
        r.name = f[0];
        r.flag = (uint16_t)atoi(f[1].c_str());
        r.rid = f[2] == "*" ? -1 : hdr.name2id(f[2]);
        if (f[2] != "*" && r.rid < 0) return fail(BSIG_ERR_FORMAT, "SAM line %lld: unknown reference %s", (long long)lineno, f[2].c_str());
        r.pos = atoi(f[3].c_str()) - 1;
        r.mapq = (uint8_t)atoi(f[4].c_str());
        if (!parse_cigar(f[5], r.cigar)) return fail(BSIG_ERR_FORMAT, "SAM line %lld: malformed CIGAR %s", (long long)lineno, f[5].c_str());
        r.next_rid = f[6] == "*" ? -1 : f[6] == "=" ? r.rid : hdr.name2id(f[6]);
        r.next_pos = atoi(f[7].c_str()) - 1;
        r.tlen = atoi(f[8].c_str());
        if (f[9] != "*") r.seq = f[9];
        if (f[10] != "*") r.qual = f[10];
        for (size_t k = 11; k < f.size(); ++k)
            if (!f[k].empty() && !parse_aux(f[k], r.aux)) return fail(BSIG_ERR_FORMAT, "SAM line %lld: malformed tag %s", (long long)lineno, f[k].c_str());
        const int rc = w.write(r);
        if (rc) return rc;
    }
    if (!header_done) { const int rc = start_writer(); if (rc) return rc; }
    return w.close();
}

}  // namespace bsig
