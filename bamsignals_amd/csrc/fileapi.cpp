// File-level half of the C ABI: BAM handles, the drop-in bsig_pileup_core / bsig_coverage_core,
// the BAM writers.  Pure host code; the compute goes through bsig_reads_upload / bsig_plan_*.
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/bamsignals_abi.h"
#include "bamio.h"
#include "collect.h"
#include "host_util.h"
#include "runtime_internal.h"

using bsig::fail;

struct bsig_bam {
    std::string path;
    bsig::BamHeader hdr;
    bsig::BaiIndex idx;
    bool from_csi = false;      // the index was read from a .csi file (htslib's bam_index_load accepts both, ref: :207)
    bsig::HostColumns cols;
    std::string idx_err;
    // file-level handles: the index is parsed by a helper thread.  LAST member: its destructor waits for that
    // thread, which writes idx and idx_err above
    std::shared_future<int> idx_job;
};

const bsig::BaiIndex *bsig_bam_index(const bsig_bam *b);
int bam_index_wait(const bsig_bam *b);
namespace { int bam_open_impl(const char *path, bool lazy, bsig_bam **out); }

namespace {

// stage seconds of the calling thread's last file-level call: open (header + BAI), decode,
// upload + HBM layout, plan + kernels + result download, total; [5] = 1 if the BAM was already
// resident in HBM
thread_local double g_call_timing[6] = {0, 0, 0, 0, 0, 0};
// ... and where the time of its stages went (bsig_last_call_timing_ex, slots 6..15)
thread_local double g_call_timing_ex[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
struct AllocSnap {
    int64_t ns, calls;
    static AllocSnap now() { return AllocSnap{bsig::g_alloc_meter.ns.load(), bsig::g_alloc_meter.calls.load()}; }
    double seconds_since(const AllocSnap &a) const { return (double)(ns - a.ns) * 1e-9; }
};
// how the calling thread's last file-level call was carried out (bsig_last_call_route)
thread_local char g_call_route[320] = "";
inline double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void fill_columns(const bsig_bam *b, bsig_columns *c)
{
    memset(c, 0, sizeof *c);
    c->n_reads = b->cols.size();
    c->n_ref = (int32_t)b->hdr.names.size();
    c->ref_len = b->hdr.lens.data();
    c->ref_off = b->cols.ref_off.data();
    c->pos = b->cols.pos.data();
    c->flag = b->cols.flag.data();
    c->mapq = b->cols.mapq.data();
    c->tlen = b->cols.tlen.data();
    c->end = nullptr;
    c->cigar_off = b->cols.cigar_off.data();
    c->cigar = b->cols.cigar.data();
}

// ---------------------------------------------------------------------------------------------
// What the file-level calls keep between calls (the reference re-opens file and index on every
// call, src/bamsignals.cpp:449,479; here the expensive part is the decode to HBM):
//   * the GPUs in use: one context (stream) per listed device;
//   * opened BAMs: header + parsed BAI of the last few files;
//   * resident BAMs: whole files decoded to HBM (on every listed GPU), least recently used first
//     out once their bytes exceed BAMSIGNALS_CACHE_GB per GPU (default 96).
// A file is identified by path + size + mtime (ns) of the BAM and of its index: a rewritten file is
// decoded again.  The lock is held for look-ups and updates only, never across a decode or a
// launch: calls from several host threads on resident BAMs run side by side (cold decodes take
// turns, they use the whole GPU anyway); an entry in use is kept alive by its shared_ptr.
// ---------------------------------------------------------------------------------------------
// Buffers the multi-GPU result path keeps between calls (so that a call on a resident BAM does no
// hipMalloc / hipFree / hipHostMalloc: config 5 moved 8 GB of allocations per call through the driver)
struct SlotScratch {
    int32_t *d_shard = nullptr;   size_t shard_cap = 0;      // this slot's result shard (device, int32 cells)
    int32_t *h_shard = nullptr;   size_t h_cap = 0;          // ... page-locked host copy ("pcie" gather)
};
struct RootScratch {
    int32_t *d_gather = nullptr;  size_t gather_cap = 0;     // RCCL receive buffer: the shards behind each other
    int32_t *d_final = nullptr;   size_t final_cap = 0;      // the result in the caller's range order
    int64_t *d_tab[3] = {nullptr, nullptr, nullptr};         // segment offsets, segment -> range, range offsets
    size_t tab_cap[3] = {0, 0, 0};
};
struct Slots {
    std::vector<int> devices;
    std::vector<bsig_ctx *> ctx;
    std::mutex run_mu;                   // one multi-GPU run at a time per device list (they share the scratch)
    std::vector<SlotScratch> scratch;
    RootScratch root;
    std::atomic<int64_t> scratch_allocs{0};   // device / pinned allocations made for the scratch (tests: stays flat)
    ~Slots()
    {
        for (size_t k = 0; k < scratch.size(); ++k) {
            (void)hipSetDevice(devices[k]);
            if (scratch[k].d_shard) bsig::block_free(devices[k], scratch[k].d_shard, scratch[k].shard_cap * sizeof(int32_t));
            if (scratch[k].h_shard) (void)hipHostFree(scratch[k].h_shard);
        }
        if (!devices.empty()) (void)hipSetDevice(devices[0]);
        if (root.d_gather) bsig::block_free(devices[0], root.d_gather, root.gather_cap * sizeof(int32_t));
        if (root.d_final) bsig::block_free(devices[0], root.d_final, root.final_cap * sizeof(int32_t));
        for (int k = 0; k < 3; ++k)
            if (root.d_tab[k]) bsig::block_free(devices[0], root.d_tab[k], root.tab_cap[k] * sizeof(int64_t));
        for (bsig_ctx *c : ctx) if (c) bsig_ctx_destroy(c);
    }
};
struct Resident {
    std::string key;                     // file key + '@' + device list
    std::shared_ptr<Slots> slots;        // the contexts the reads live on stay alive as long as the reads
    std::vector<bsig_reads *> reads;     // one per slot
    int64_t bytes = 0;                   // per GPU
    // index-driven decodes: the (rid, beg, end) intervals the reads were decoded for, merged and sorted; a
    // later query whose regions all lie inside them can use these reads as they are
    std::vector<int32_t> cov_rid;
    std::vector<int64_t> cov_beg, cov_end;
    ~Resident() { for (bsig_reads *r : reads) if (r) bsig_reads_free(r); }
};
struct OpenBam {
    std::string key;
    bsig_bam *bam = nullptr;
    ~OpenBam() { if (bam) bsig_bam_close(bam); }
};

struct Cache {
    std::mutex mu;                                       // guards everything below
    std::mutex decode_mu;                                // cold decodes take turns
    std::vector<std::shared_ptr<Slots>> slot_sets;       // one per device list seen (an R session alternating
                                                         // between device= arguments keeps both sets resident)
    std::list<std::shared_ptr<OpenBam>> bams;            // most recently used first
    std::list<std::shared_ptr<Resident>> resident;       // whole files, most recently used first
    std::list<std::shared_ptr<Resident>> regional;       // index-driven decodes, most recently used first (few)
    void clear()
    {
        resident.clear();
        regional.clear();
        bams.clear();
        slot_sets.clear();
    }
};
Cache g_cache;

// size + mtime (ns) of a file, "" if it cannot be examined
std::string file_stamp(const std::string &path)
{
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return std::string();
    return std::to_string((long long)st.st_size) + "|" + std::to_string((long long)st.st_mtime) + "|" +
           std::to_string((long long)st.st_mtim.tv_nsec);
}

std::string file_key(const std::string &path)
{
    const std::string s = file_stamp(path);
    return s.empty() ? s : path + "|" + s;
}

int64_t cache_budget_bytes()
{
    double gb = 96.0;
    if (const char *e = getenv("BAMSIGNALS_CACHE_GB")) gb = std::max(0.0, atof(e));
    return (int64_t)(gb * (double)(1ull << 30));
}

// the reads file ("sidecar") of a BAM, or "" when sidecars are off: BAMSIGNALS_SIDECAR=1 puts it next
// to the BAM (<bam>.bsig), BAMSIGNALS_SIDECAR_DIR=<dir> into that directory
std::string sidecar_path(const std::string &bampath)
{
    if (const char *d = getenv("BAMSIGNALS_SIDECAR_DIR")) {
        if (!*d) return std::string();
        char real[4096];
        const std::string abs = realpath(bampath.c_str(), real) ? std::string(real) : bampath;
        uint64_t h = 1469598103934665603ull;                       // FNV-1a of the absolute path
        for (unsigned char ch : abs) { h ^= ch; h *= 1099511628211ull; }
        const size_t slash = abs.find_last_of('/');
        char hex[32];
        snprintf(hex, sizeof hex, "%016llx", (unsigned long long)h);
        return std::string(d) + "/" + abs.substr(slash == std::string::npos ? 0 : slash + 1) + "." + hex + ".bsig";
    }
    if (const char *e = getenv("BAMSIGNALS_SIDECAR"))
        if (*e && strcmp(e, "0") != 0) return bampath + ".bsig";
    return std::string();
}

// GPUs a file-level call uses: the `device` argument if >= 0, else BAMSIGNALS_DEVICES ("0,1,2,3":
// ranges are dealt round-robin to them), else BAMSIGNALS_DEVICE, else GPU 0
std::vector<int> pick_devices(int device)
{
    std::vector<int> d;
    if (device >= 0) return {device};
    if (const char *e = getenv("BAMSIGNALS_DEVICES")) {
        for (const char *p = e; *p;) {
            char *end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            d.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
        }
    }
    if (d.empty()) d.push_back(getenv("BAMSIGNALS_DEVICE") ? atoi(getenv("BAMSIGNALS_DEVICE")) : 0);
    return d;
}

bool env_is(const char *name, const char *value)
{
    const char *e = getenv(name);
    return e && !strcmp(e, value);
}

// body(k) for every slot on its own host thread (one thread per GPU); the first failure wins and its
// message becomes the caller's bsig_last_error()
int for_each_slot(size_t n, const std::function<int(size_t)> &body)
{
    std::vector<int> rcs(n, BSIG_OK);
    std::vector<std::string> msgs(n);
    auto one = [&](size_t k) {
        rcs[k] = body(k);
        if (rcs[k]) msgs[k] = bsig_last_error();
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < n; ++k) {
        try { th.emplace_back(one, k); } catch (const std::system_error &) { one(k); }
    }
    one(0);
    for (auto &t : th) t.join();
    for (size_t k = 0; k < n; ++k)
        if (rcs[k]) return fail(rcs[k], "%s", msgs[k].c_str());
    return BSIG_OK;
}

// grows a cached buffer of the multi-GPU result path (never shrinks; released with the Slots).  The memory comes
// from the library's cache of free device blocks (runtime_internal.h): right after a cold decode that cache
// holds the decode's scratch, and a fresh hipMalloc at that moment is exactly the one that stalls for seconds
template <typename T>
int grow_dev(Slots &sl, int device, T **p, size_t *cap, size_t want)
{
    if (*p && *cap >= want) return BSIG_OK;
    HIP_TRY(hipSetDevice(device));
    if (*p) { bsig::block_free(device, *p, *cap * sizeof(T)); *p = nullptr; *cap = 0; }
    const size_t n = std::max<size_t>(want + want / 8, 1024);          // head-room: a slightly larger call fits too
    void *q = nullptr;
    size_t got = 0;
    HIP_TRY(bsig::block_alloc(device, n * sizeof(T), 2.0, &q, &got));
    *p = (T *)q;
    *cap = got / sizeof(T);
    sl.scratch_allocs.fetch_add(1);
    return BSIG_OK;
}
int grow_pinned(Slots &sl, int device, int32_t **p, size_t *cap, size_t want)
{
    if (*p && *cap >= want) return BSIG_OK;
    HIP_TRY(hipSetDevice(device));
    if (*p) { (void)hipHostFree(*p); *p = nullptr; *cap = 0; }
    const size_t n = std::max<size_t>(want + want / 8, 1024);
    HIP_TRY(hipHostMalloc((void **)p, n * sizeof(int32_t), hipHostMallocDefault));
    *cap = n;
    sl.scratch_allocs.fetch_add(1);
    return BSIG_OK;
}

// segment k of src goes to range which[k] of the destination (the body of bsig_scatter_segments; the destination
// may be one flat buffer or one vector per range)
int scatter_segments_to(int64_t n, const int32_t *src, const int64_t *src_off, const bsig::HostDest &dst, const int64_t *which)
{
    if (n > 0 && (!src_off || !dst.off || !which)) return fail(BSIG_ERR_ARG, "NULL argument");
    for (int64_t k = 0; k < n; ++k) {
        const int64_t len = src_off[k + 1] - src_off[k];
        const int64_t i = which[k];
        if (len < 0 || i < 0) return fail(BSIG_ERR_ARG, "bad segment %lld", (long long)k);
        if (len != dst.off[i + 1] - dst.off[i]) return fail(BSIG_ERR_ARG, "segment %lld does not fit its destination", (long long)k);
    }
    if (n <= 0) return BSIG_OK;
    auto copy_range = [&](int64_t k0, int64_t k1) {
        for (int64_t k = k0; k < k1; ++k) {
            const int64_t len = src_off[k + 1] - src_off[k];
            if (len) memcpy(dst.range(which[k]), src + src_off[k], (size_t)len * sizeof(int32_t));
        }
    };
    // the destinations are disjoint (each range owns its cells): big results are moved by a few
    // threads, each taking a contiguous share of the source
    const int64_t cells = src_off[n] - src_off[0];
    int n_thr = cells * (int64_t)sizeof(int32_t) >= (16 << 20) ? 8 : 1;
    if (const char *e = getenv("BAMSIGNALS_COPY_THREADS")) n_thr = std::max(1, std::min(64, atoi(e)));
    n_thr = (int)std::min<int64_t>(n_thr, n);
    if (n_thr <= 1) { copy_range(0, n); return BSIG_OK; }
    std::vector<int64_t> cut((size_t)n_thr + 1, n);
    cut[0] = 0;
    for (int t = 1; t < n_thr; ++t) {
        const int64_t target = src_off[0] + cells * t / n_thr;
        cut[(size_t)t] = std::lower_bound(src_off, src_off + n, target) - src_off;
    }
    std::vector<std::thread> th;
    for (int t = 1; t < n_thr; ++t) th.emplace_back(copy_range, cut[(size_t)t], cut[(size_t)t + 1]);
    copy_range(cut[0], cut[1]);
    for (auto &x : th) x.join();
    return BSIG_OK;
}

// ---- several GPUs: ranges are independent (each owns its output, ref: src/bamsignals.cpp:164,181,186)
// The (rid, loc)-sorted ranges (ref: :222-226,246) are dealt round-robin to the GPUs; every GPU plans
// and runs its shard on its own stream, driven by its own host thread, into a shard buffer the slot
// keeps between calls.  The shards then reach the caller by one of
//   "xgmi/rccl"   (default where RCCL runs): RCCL grouped send/recv into a receive buffer on the first GPU
//                 (collect.h), put into the caller's range order there by k_place_segments, ONE download;
//   "xgmi/direct" (default without RCCL where the first GPU can read its peers; env BAMSIGNALS_GATHER=direct):
//                 k_place_segments on the first GPU reads every peer's shard buffer in place over xGMI --
//                 one pass, no receive buffer -- then ONE download;
//   "xgmi/peer":  as rccl with hipMemcpyPeerAsync (no RCCL, no peer mapping);
//   "pcie"        (env BAMSIGNALS_GATHER=pcie): every shard over its own GPU's PCIe link into a page-locked
//                 buffer, put in place by host threads (bsig_scatter_segments).
// All device and page-locked buffers are kept in the Slots: a call on a resident BAM allocates nothing here.
int run_on_slots(Slots &sl, const std::vector<bsig_reads *> &reads, int64_t n, const int32_t *rid, const int32_t *loc,
                 const int32_t *width, const int32_t *strand, const bsig_params &prm, const bsig::HostDest &dest,
                 std::string &gather_name)
{
    const int64_t *off = dest.off;
    std::lock_guard<std::mutex> run_lock(sl.run_mu);
    const double t0 = now_s();
    const size_t nd = sl.ctx.size();
    if (sl.scratch.size() != nd) sl.scratch.resize(nd);
    std::vector<int64_t> order;
    bsig::sort_ranges(n, rid, loc, order);
    struct Shard {
        std::vector<int64_t> which;
        std::vector<int32_t> rid, loc, len, strand;
        bsig_plan *plan = nullptr;
        int64_t cells = 0;
    };
    std::vector<Shard> sh(nd);
    for (size_t k = 0; k < nd; ++k) {
        const size_t cap = (size_t)n / nd + 1;
        sh[k].which.reserve(cap); sh[k].rid.reserve(cap); sh[k].loc.reserve(cap); sh[k].len.reserve(cap); sh[k].strand.reserve(cap);
    }
    // the sorted ranges are dealt in BLOCKS of consecutive ranges (block b to GPU b mod N; at least eight blocks per
    // GPU, at most 4,096 ranges per block): still balanced over the genome, and where the caller's order is the
    // sorted order -- tilings, sorted peaks -- a block is one contiguous slice of the caller's result, which its GPU
    // can deliver over its own PCIe link without any reassembly ("blocks" below)
    int64_t block = n / (int64_t)(nd * 8);
    block = std::max<int64_t>(1, std::min<int64_t>(block, 4096));
    if (const char *e = getenv("BAMSIGNALS_SHARD_BLOCK")) block = std::max<int64_t>(1, atoll(e));      // (1 = round-robin, range by range)
    for (int64_t k = 0; k < n; ++k) {
        Shard &S = sh[(size_t)((k / block) % (int64_t)nd)];
        const int64_t i = order[(size_t)k];
        S.which.push_back(i);
        S.rid.push_back(rid[i]); S.loc.push_back(loc[i]);
        S.len.push_back(width[i]); S.strand.push_back(strand[i]);
    }
    // runs of a shard whose ranges are consecutive in the caller's order: contiguous slices of the caller's result
    struct Run { int64_t j0, j1; };                // segments [j0, j1) of the shard
    std::vector<std::vector<Run>> runs(nd);
    int64_t n_runs = 0;
    for (size_t k = 0; k < nd; ++k) {
        const std::vector<int64_t> &w = sh[k].which;
        for (size_t j = 0; j < w.size();) {
            size_t e = j + 1;
            while (e < w.size() && w[e] == w[e - 1] + 1) ++e;
            runs[k].push_back(Run{(int64_t)j, (int64_t)e});
            j = e;
        }
        n_runs += (int64_t)runs[k].size();
    }
    // "blocks": by request, or by itself where the slices are long (64 ranges or more on average) and the result
    // is large enough for the route to matter
    const char *genv = getenv("BAMSIGNALS_GATHER");
    if (genv && !*genv) genv = nullptr;
    const bool blocks = genv ? !strcmp(genv, "blocks") : (n_runs * 64 <= n && off[n] * (int64_t)sizeof(int32_t) >= ((int64_t)32 << 20));
    const bool pcie = env_is("BAMSIGNALS_GATHER", "pcie");
    gather_name = blocks ? "pcie/blocks" : pcie ? "pcie" : "xgmi";
    // host threads that move page-locked halves on, per GPU: all GPUs download at once (slots that name the same
    // GPU take turns on its staging halves, each with the device's full share of threads)
    std::vector<int> distinct(sl.devices);
    std::sort(distinct.begin(), distinct.end());
    distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
    int copy_total = 16;
    if (const char *e = getenv("BAMSIGNALS_COPY_THREADS")) copy_total = std::max(1, std::min(64, atoi(e)));
    const int copy_thr = std::max(2, copy_total / (int)std::max<size_t>(distinct.size(), 1));
    std::vector<double> t_run_done(nd, 0.0);
    auto cleanup = [&]() {
        for (Shard &S : sh)
            if (S.plan) { bsig_plan_free(S.plan); S.plan = nullptr; }
    };
    // plan + launch, one host thread per GPU
    int rc = for_each_slot(nd, [&](size_t k) -> int {
        Shard &S = sh[k];
        SlotScratch &X = sl.scratch[k];
        int r = bsig_plan_create(sl.ctx[k], reads[k], (int64_t)S.which.size(), S.rid.data(), S.loc.data(), S.len.data(),
                                 S.strand.data(), &prm, &S.plan);
        if (r) return r;
        S.cells = bsig_plan_cells(S.plan);
        // every segment must fit its destination: the caller's offsets are only trusted after this check
        const int64_t *po = bsig_plan_offsets(S.plan);
        for (size_t j = 0; j < S.which.size(); ++j) {
            const int64_t w = S.which[j];
            if (po[j + 1] - po[j] != off[w + 1] - off[w])
                return fail(BSIG_ERR_ARG, "offsets do not match bsig_layout() for these parameters");
        }
        r = grow_dev(sl, sl.devices[k], &X.d_shard, &X.shard_cap, (size_t)std::max<int64_t>(S.cells, 4));
        if (r) return r;
        r = bsig_plan_run(S.plan, X.d_shard);
        if (r) return r;
        if (blocks) {
            // every contiguous slice of the caller's result this GPU owns leaves over this GPU's own PCIe link,
            // straight to its place: no gather on one GPU, no reassembly pass on the host
            r = bsig_ctx_sync(sl.ctx[k]);
            if (r) return r;
            t_run_done[k] = now_s();
            std::vector<int64_t> s0, d0, nc;
            for (const Run &q : runs[k]) {
                s0.push_back(po[q.j0]); d0.push_back(off[S.which[(size_t)q.j0]]); nc.push_back(po[q.j1] - po[q.j0]);
            }
            return bsig::download_slices_to_dest(sl.ctx[k], X.d_shard, (int64_t)s0.size(), s0.data(), d0.data(), nc.data(), dest, copy_thr);
        }
        if (pcie && S.cells) {
            r = grow_pinned(sl, sl.devices[k], &X.h_shard, &X.h_cap, (size_t)S.cells);
            if (r) return r;
            HIP_TRY(hipMemcpyAsync(X.h_shard, X.d_shard, (size_t)S.cells * sizeof(int32_t), hipMemcpyDeviceToHost,
                                   (hipStream_t)bsig_ctx_stream(sl.ctx[k])));
        }
        return bsig_ctx_sync(sl.ctx[k]);
    });
    if (rc) { cleanup(); return rc; }
    const double t1 = now_s();
    int64_t cells = 0;
    for (size_t k = 0; k < nd; ++k) cells += sh[k].cells;
    const int64_t total = off[n];
    if (cells != total) { cleanup(); return fail(BSIG_ERR_ARG, "offsets do not match bsig_layout() for these parameters"); }
    char times[128];
    if (blocks) {
        cleanup();
        double t_run = t0;
        for (double t : t_run_done) t_run = std::max(t_run, t);
        snprintf(times, sizeof times, " (plan+run %.3f s, download in %lld slices %.3f s)", t_run - t0, (long long)n_runs, t1 - t_run);
        gather_name += times;
        return BSIG_OK;
    }
    if (pcie) {
        for (size_t k = 0; k < nd && rc == BSIG_OK; ++k)
            rc = scatter_segments_to((int64_t)sh[k].which.size(), sl.scratch[k].h_shard, bsig_plan_offsets(sh[k].plan), dest,
                                     sh[k].which.data());
        cleanup();
        snprintf(times, sizeof times, " (plan+run+d2h %.3f s, host scatter %.3f s)", t1 - t0, now_s() - t1);
        gather_name += times;
        return rc;
    }
    if (total == 0) { cleanup(); return BSIG_OK; }
    // ---- xgmi: the shards meet on the first GPU, in the caller's range order, and leave in one download ----
    double t2 = t1, t3 = t1;
    auto body = [&]() -> int {
        RootScratch &G = sl.root;
        const int dev0 = sl.devices[0];
        hipStream_t st = (hipStream_t)bsig_ctx_stream(sl.ctx[0]);
        // tables: per shard its local segment offsets (n_k + 1 entries) and segment -> range; the ranges' offsets
        std::vector<int64_t> seg_off, seg_which;
        seg_off.reserve((size_t)n + nd);
        seg_which.reserve((size_t)n);
        std::vector<size_t> seg_base(nd), which_base(nd);
        for (size_t k = 0; k < nd; ++k) {
            seg_base[k] = seg_off.size();
            which_base[k] = seg_which.size();
            const int64_t *po = bsig_plan_offsets(sh[k].plan);
            seg_off.insert(seg_off.end(), po, po + sh[k].which.size() + 1);
            seg_which.insert(seg_which.end(), sh[k].which.begin(), sh[k].which.end());
        }
        int r = grow_dev(sl, dev0, &G.d_final, &G.final_cap, (size_t)total);
        if (!r) r = grow_dev(sl, dev0, &G.d_tab[0], &G.tab_cap[0], seg_off.size());
        if (!r) r = grow_dev(sl, dev0, &G.d_tab[1], &G.tab_cap[1], std::max<size_t>(seg_which.size(), 1));
        if (!r) r = grow_dev(sl, dev0, &G.d_tab[2], &G.tab_cap[2], (size_t)n + 1);
        if (r) return r;
        HIP_TRY(hipSetDevice(dev0));
        HIP_TRY(hipMemcpyAsync(G.d_tab[0], seg_off.data(), seg_off.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(G.d_tab[1], seg_which.data(), seg_which.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(G.d_tab[2], off, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
        bsig::ExchangeUse use;            // holds the exchange's lock until the streams are synchronised below
        r = bsig::exchange_open(sl.ctx, use);
        if (r) return r;
        // (env BAMSIGNALS_GATHER=direct: in-place reads even where RCCL runs; =copy: a receive buffer even where
        // in-place reads are possible -- both exist so that an 8-GPU node can time all three)
        const bool want_direct = env_is("BAMSIGNALS_GATHER", "direct"), want_copy = env_is("BAMSIGNALS_GATHER", "copy");
        const bool rccl = !strcmp(use.transport, "rccl") && !want_direct;
        const bool direct = !rccl && !want_copy && bsig::exchange_root_reads_peers(use);
        std::vector<const int32_t *> from(nd);          // where the place kernel finds shard k
        if (direct) {
            for (size_t k = 0; k < nd; ++k) from[k] = sl.scratch[k].d_shard;
            gather_name = "xgmi/direct";
        } else {
            r = grow_dev(sl, dev0, &G.d_gather, &G.gather_cap, (size_t)total);
            if (r) return r;
            std::vector<size_t> goff(nd), glen(nd);
            std::vector<const uint8_t *> src(nd);
            int64_t at = 0;
            for (size_t k = 0; k < nd; ++k) {
                goff[k] = (size_t)at * sizeof(int32_t);
                glen[k] = (size_t)sh[k].cells * sizeof(int32_t);
                src[k] = (const uint8_t *)sl.scratch[k].d_shard;
                from[k] = G.d_gather + at;
                at += sh[k].cells;
            }
            // (the first GPU's own shard is read in place: no copy into the receive buffer)
            from[0] = sl.scratch[0].d_shard;
            glen[0] = 0;
            r = bsig::exchange_gather(use, src, glen, (uint8_t *)G.d_gather, goff);
            if (r) return r;
            gather_name = std::string("xgmi/") + use.transport;
            for (size_t k = 1; k < nd; ++k) {              // the senders' streams (RCCL queues the sends there)
                r = bsig_ctx_sync(sl.ctx[k]);
                if (r) return r;
            }
        }
        HIP_TRY(hipSetDevice(dev0));
        for (size_t k = 0; k < nd; ++k)
            HIP_TRY(bsig::launch_place_segments((int64_t)sh[k].which.size(), from[k], G.d_tab[0] + seg_base[k], G.d_final, G.d_tab[2],
                                                G.d_tab[1] + which_base[k], st));
        HIP_TRY(hipStreamSynchronize(st));
        use.release();
        t2 = now_s();
        r = bsig::download_to_dest(sl.ctx[0], G.d_final, dest, total);
        t3 = now_s();
        return r;
    };
    rc = body();
    if (rc) for (size_t k = 0; k < nd; ++k) (void)bsig_ctx_sync(sl.ctx[k]);
    cleanup();
    snprintf(times, sizeof times, " (plan+run %.3f s, gather+place %.3f s, download %.3f s)", t1 - t0, t2 - t1, t3 - t2);
    gather_name += times;
    return rc;
}

// Out of device memory: what the cache can spare goes back to the driver -- the index-driven decodes it keeps, the
// result-path buffers of device lists nobody is running on (never those of `keep`, the caller's own), and the free
// blocks.  What a running call holds stays (shared_ptr, run_mu).
void drop_spare_device_memory(const Slots *keep)
{
    std::list<std::shared_ptr<Resident>> drop;
    std::vector<std::shared_ptr<Slots>> sets;
    {
        std::lock_guard<std::mutex> lock(g_cache.mu);
        drop.swap(g_cache.regional);
        sets = g_cache.slot_sets;
    }
    drop.clear();
    for (auto &sp : sets) {
        if (sp.get() == keep) continue;
        std::unique_lock<std::mutex> run(sp->run_mu, std::try_to_lock);
        if (!run.owns_lock()) continue;
        Slots &S = *sp;
        for (size_t k = 0; k < S.scratch.size(); ++k) {
            (void)hipSetDevice(S.devices[k]);
            if (S.scratch[k].d_shard) bsig::block_free(S.devices[k], S.scratch[k].d_shard, S.scratch[k].shard_cap * sizeof(int32_t));
            if (S.scratch[k].h_shard) (void)hipHostFree(S.scratch[k].h_shard);
            S.scratch[k] = SlotScratch{};
        }
        if (!S.devices.empty()) {
            (void)hipSetDevice(S.devices[0]);
            if (S.root.d_gather) bsig::block_free(S.devices[0], S.root.d_gather, S.root.gather_cap * sizeof(int32_t));
            if (S.root.d_final) bsig::block_free(S.devices[0], S.root.d_final, S.root.final_cap * sizeof(int32_t));
            for (int k = 0; k < 3; ++k)
                if (S.root.d_tab[k]) bsig::block_free(S.devices[0], S.root.d_tab[k], S.root.tab_cap[k] * sizeof(int64_t));
            S.root = RootScratch{};
        }
    }
    bsig::block_cache_release();
}

// size + mtime of every index file a BAM may be opened with (<bam>.csi, <stem>.csi, <bam>.bai, <stem>.bai;
// bsig_bam_open tries them in this order): whichever one is used, replacing, adding or removing it
// changes the stamp, so cached headers, resident reads and reads files are never tied to a stale index
std::string index_stamps(const std::string &bam)
{
    std::string stem = bam;
    if (stem.size() > 4 && stem.compare(stem.size() - 4, 4, ".bam") == 0) stem.erase(stem.size() - 4);
    std::string out;
    for (const std::string &cand : {bam + ".bai", stem + ".bai", bam + ".csi", stem + ".csi"}) out += file_stamp(cand) + ";";
    return out;
}

// An index-driven decode remembers the intervals it was made for (merged, sorted by reference and start).
void set_coverage(Resident &R, int64_t n, const int32_t *rid, const int64_t *beg, const int64_t *end)
{
    std::vector<int64_t> order((size_t)n);
    for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return rid[a] != rid[b] ? rid[a] < rid[b] : beg[a] < beg[b]; });
    R.cov_rid.clear(); R.cov_beg.clear(); R.cov_end.clear();
    for (int64_t k = 0; k < n; ++k) {
        const int64_t i = order[(size_t)k];
        if (end[i] <= beg[i]) continue;
        if (!R.cov_rid.empty() && R.cov_rid.back() == rid[i] && beg[i] <= R.cov_end.back()) {
            R.cov_end.back() = std::max(R.cov_end.back(), end[i]);
        } else {
            R.cov_rid.push_back(rid[i]); R.cov_beg.push_back(beg[i]); R.cov_end.push_back(end[i]);
        }
    }
}
// ... and serves a later query whose every region lies inside one of them: the BAI query of the earlier
// call returned every record overlapping its intervals, hence every record overlapping a sub-interval
bool regions_covered(const Resident &R, int64_t n, const int32_t *rid, const int64_t *beg, const int64_t *end)
{
    const size_t m = R.cov_rid.size();
    for (int64_t i = 0; i < n; ++i) {
        if (end[i] <= beg[i]) continue;
        // last interval with (rid, beg) <= (rid[i], beg[i])
        size_t lo = 0, hi = m;
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (R.cov_rid[mid] < rid[i] || (R.cov_rid[mid] == rid[i] && R.cov_beg[mid] <= beg[i])) lo = mid + 1; else hi = mid;
        }
        if (lo == 0) return false;
        const size_t k = lo - 1;
        if (R.cov_rid[k] != rid[i] || R.cov_end[k] < end[i]) return false;
    }
    return true;
}

// the common body of pileup_core / coverage_core
int file_level(const char *bampath, int64_t n, const int32_t *seq_code, int32_t n_levels,
               const char *const *levels, const int32_t *start, const int32_t *width,
               const int32_t *strand, const bsig_params &prm, int32_t device, const bsig::HostDest &dest)
{
    if (!bampath) return fail(BSIG_ERR_ARG, "bampath is NULL");
    if (n < 0 || (n > 0 && (!seq_code || !start || !width || !strand || !levels)))
        return fail(BSIG_ERR_ARG, "range arrays missing");
    const int64_t *off = dest.off;
    if (!off) return fail(BSIG_ERR_ARG, "offsets missing");
    double *T = g_call_timing, *X = g_call_timing_ex;
    for (int k = 0; k < 6; ++k) T[k] = 0;
    for (int k = 0; k < 10; ++k) X[k] = 0;
    g_call_route[0] = 0;
    const double t_begin = now_s();
    const AllocSnap a_begin = AllocSnap::now();
    // ref: Bamfile ctor :200-214 opens file + index on every call; here an unchanged file (same
    // size and mtime of the BAM and of its index) reuses the parsed header and BAI
    // (keyed by the resolved path: "a.bam" and "./a.bam" are one file, one resident copy)
    char resolved[4096];
    const std::string canon = realpath(bampath, resolved) ? std::string(resolved) : std::string(bampath);
    const std::string key = file_key(canon);
    const std::string bkey = key.empty() ? std::string() : key + "#" + index_stamps(canon);
    int rc = BSIG_OK;
    std::shared_ptr<OpenBam> ob;
    if (!bkey.empty()) {
        std::lock_guard<std::mutex> lock(g_cache.mu);
        for (auto it = g_cache.bams.begin(); it != g_cache.bams.end(); ++it)
            if ((*it)->key == bkey) {
                ob = *it;
                g_cache.bams.erase(it);
                g_cache.bams.push_front(ob);
                break;
            }
    }
    if (!ob) {
        bsig_bam *fresh = nullptr;
        rc = bam_open_impl(bampath, true, &fresh);
        if (rc) return rc;
        ob = std::make_shared<OpenBam>();
        ob->key = bkey;
        ob->bam = fresh;
        if (!bkey.empty()) {
            std::lock_guard<std::mutex> lock(g_cache.mu);
            g_cache.bams.push_front(ob);
            while (g_cache.bams.size() > 16) g_cache.bams.pop_back();
        }
    }
    bsig_bam *bam = ob->bam;
    T[0] = now_s() - t_begin;

    // seqnames -> BAM reference ids, by name (ref: parseRegions :113-120)
    std::vector<int32_t> level_rid((size_t)n_levels, -2);
    std::vector<int32_t> rid((size_t)n), loc((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t c = seq_code[i];
        if (c < 0 || c >= n_levels) return fail(BSIG_ERR_ARG, "seqnames code %d out of range", c);
        if (level_rid[(size_t)c] == -2) level_rid[(size_t)c] = bsig_bam_name2id(bam, levels[c]);
        if (level_rid[(size_t)c] < 0)
            return fail(BSIG_ERR_CHROM, "chromosome %s not present in the bam file", levels[c]);   // ref: :119
        rid[(size_t)i] = level_rid[(size_t)c];
        loc[(size_t)i] = start[i] - 1;                          // ref: :131
        if (width[i] < 0) return fail(BSIG_ERR_ARG, "range %lld has a negative width", (long long)i);
    }
    // same argument checks as bsig_plan_create, before any I/O
    const bool mid = prm.mode != BSIG_MODE_COVERAGE && prm.pe_mid;
    const bool tspan = prm.mode == BSIG_MODE_COVERAGE && prm.tspan;
    if ((mid || tspan) && prm.n_tlen_filter != 2)
        return fail(BSIG_ERR_ARG, "paired-end midpoint/extend needs a 2-element tlen_filter");
    int64_t ext = prm.mode == BSIG_MODE_COVERAGE ? (tspan ? prm.tlen_filter[1] : 0)
                                                 : std::llabs((long long)prm.shift) + (mid ? prm.tlen_filter[1] : 0);
    if (ext < 0) return fail(BSIG_ERR_EXT, "negative 'ext' values don't make sense");             // ref: :243

    // the GPUs of this call: every device list seen keeps its own contexts, scratch and resident BAMs (a
    // session that alternates between two device= arguments keeps both copies; the LRU budget bounds them)
    const std::vector<int> devs = pick_devices(device);
    std::string dev_tag = "@";
    for (int d : devs) dev_tag += std::to_string(d) + ",";
    std::shared_ptr<Slots> slots;
    {
        std::lock_guard<std::mutex> lock(g_cache.mu);
        for (auto &sp : g_cache.slot_sets)
            if (sp->devices == devs) slots = sp;
        if (!slots) {
            auto fresh = std::make_shared<Slots>();
            fresh->devices = devs;
            for (int d : devs) {
                bsig_ctx *c = nullptr;
                rc = bsig_ctx_create(d, nullptr, &c);
                if (rc) return rc;
                fresh->ctx.push_back(c);
            }
            g_cache.slot_sets.push_back(fresh);
            slots = fresh;
        }
    }
    const size_t nd = devs.size();
    const bool many = nd > 1 || env_is("BAMSIGNALS_FORCE_SHARDED", "1");     // (=1: the multi-GPU route on one GPU, testing)

    // How much of the file do the ranges need?  Small queries decode only the BGZF blocks the
    // index lists (ref: one bam_itr_queryi per chunk of ranges, :252-267); large ones decode the
    // whole file once and keep it in HBM for the next call.
    int64_t genome = 0, wanted = 0;
    for (int32_t l : bam->hdr.lens) genome += l;
    for (int64_t i = 0; i < n; ++i) wanted += (int64_t)width[i] + 2 * ext + 16384;
    const char *force = getenv("BAMSIGNALS_DECODE");   // "all" | "regions" (testing / tuning)
    // (5e7-read BAM: the whole-file decode costs 0.06-0.08 s flat and leaves the BAM resident for the
    // next call; the index-driven decode 0.02 s for 7 % of the genome, 0.09 s for 74 %)
    bool whole = wanted * 3 > genome;
    if (force && !strcmp(force, "all")) whole = true;
    if (force && !strcmp(force, "regions")) whole = false;

    // the regions an index-driven decode would ask the BAI for (ref: :252-267): range +- ext
    std::vector<int64_t> qbeg, qend;
    if (!whole) {
        qbeg.resize((size_t)n); qend.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            qbeg[(size_t)i] = (int64_t)loc[(size_t)i] - ext;
            qend[(size_t)i] = (int64_t)loc[(size_t)i] + width[i] + ext;
        }
    }

    // ---- the reads of this call on every GPU -------------------------------------------------------
    const std::string rkey = bkey + dev_tag;
    std::shared_ptr<Resident> res;
    std::string how_decoded = "resident";
    // a whole file resident on these GPUs serves every query; failing that, an earlier index-driven decode
    // whose regions contain this call's (repeated bamCount / bamProfile on the same few ranges)
    auto lookup = [&]() {
        if (key.empty()) return;
        std::lock_guard<std::mutex> lock(g_cache.mu);
        for (auto it = g_cache.resident.begin(); it != g_cache.resident.end(); ++it)
            if ((*it)->key == rkey) {
                res = *it;
                g_cache.resident.erase(it);
                g_cache.resident.push_front(res);
                how_decoded = "resident";
                return;
            }
        if (whole) return;
        for (auto it = g_cache.regional.begin(); it != g_cache.regional.end(); ++it)
            if ((*it)->key == rkey && regions_covered(**it, n, rid.data(), qbeg.data(), qend.data())) {
                res = *it;
                g_cache.regional.erase(it);
                g_cache.regional.push_front(res);
                how_decoded = "resident (index-driven decode of an earlier call)";
                return;
            }
    };
    lookup();
    std::string side_to_write, side_stamp;
    if (res) {
        T[5] = 1;
    } else {
        std::unique_lock<std::mutex> dlock(g_cache.decode_mu);
        // another thread may have decoded this very file while this one waited for its turn
        lookup();
        if (res) {
            T[5] = 1;
        } else {
        const double t_dec = now_s();
        const AllocSnap a_dec = AllocSnap::now();
        res = std::make_shared<Resident>();
        res->key = rkey;
        res->slots = slots;
        res->reads.assign(nd, nullptr);
        auto clone_rest = [&]() -> int {
            return for_each_slot(nd, [&](size_t k) -> int {
                return k == 0 ? (int)BSIG_OK : bsig_reads_clone(res->reads[0], slots->ctx[k], &res->reads[k]);
            });
        };
        double t6[6] = {0, 0, 0, 0, 0, 0};
        // (a lambda: out of device memory, the decode is tried once more after the cache has given up what it can spare)
        auto decode_once = [&]() -> int {
        for (bsig_reads *&r : res->reads) { if (r) bsig_reads_free(r); r = nullptr; }
        if (whole) {
            const std::string side = key.empty() ? std::string() : sidecar_path(bampath);
            // (the reads file is tied to the CONTENT it was made from -- size and mtime of the BAM and of its
            // index files -- not to the spelling of the path it was reached by)
            side_stamp = file_stamp(bampath) + "#" + index_stamps(bampath);
            bool loaded = false;
            if (!side.empty()) {
                // a second process (or a call after the BAM left the cache) skips inflate and parse
                struct stat sb;
                if (stat(side.c_str(), &sb) == 0) {
                    rc = for_each_slot(nd, [&](size_t k) { return bsig_reads_load(slots->ctx[k], side.c_str(), side_stamp.c_str(), &res->reads[k]); });
                    loaded = rc == BSIG_OK;
                    if (!loaded) { for (bsig_reads *&r : res->reads) { if (r) bsig_reads_free(r); r = nullptr; } rc = BSIG_OK; }
                }
            }
            if (loaded) {
                how_decoded = "sidecar";
            } else {
                rc = bsig::kNeedsCpuPath;
                if (many && !env_is("BAMSIGNALS_DEVICE_DECODE", "0") && !env_is("BAMSIGNALS_SHARDED_DECODE", "0")) {
                    // every GPU inflates and parses its share of the BGZF blocks; the column shares are
                    // all-gathered over xGMI (devdecode.hip: reads_from_bam_sharded)
                    const char *transport = "";
                    rc = bsig::reads_from_bam_sharded(slots->ctx, bampath, 0, res->reads, &transport);
                    if (rc == BSIG_OK) how_decoded = std::string("sharded decode, columns over ") + transport;
                    bsig_device_decode_timing(t6);
                }
                if (rc == bsig::kNeedsCpuPath) {
                    // the records are taken from the uncompressed stream on the first GPU itself
                    // (devdecode.hip; falls back to the CPU decode inside the call where it must); further
                    // GPUs get device-to-device copies of the resident layout
                    rc = bsig_reads_from_bam(slots->ctx[0], bam, 0, &res->reads[0]);
                    bsig_device_decode_timing(t6);
                    if (rc == BSIG_OK) rc = clone_rest();
                    how_decoded = nd > 1 ? "decode on the first GPU + clones" : "decode";
                }
                if (rc) return rc;
                side_to_write = side;          // written after the decode lock is released (best effort)
            }
        } else {
            // index-driven: only the blocks the BAI lists for the ranges (ref: one bam_itr_queryi per
            // chunk of ranges, :252-267), parsed on the GPU like the whole file.  With several GPUs the
            // islands are dealt to them and the column shares all-gathered (devdecode.hip).
            rc = bam_index_wait(bam);                 // (parsed in the background since the open)
            if (rc) return rc;
            rc = bsig::kNeedsCpuPath;
            how_decoded = "index-driven decode";
            if (many && !env_is("BAMSIGNALS_DEVICE_DECODE", "0") && !env_is("BAMSIGNALS_SHARDED_DECODE", "0")) {
                const char *transport = "";
                rc = bsig::reads_from_regions_sharded(slots->ctx, bampath, *bsig_bam_index(bam), n, rid.data(), qbeg.data(), qend.data(),
                                                      0, res->reads, &transport);
                if (rc == BSIG_OK) how_decoded = std::string("index-driven decode, sharded, columns over ") + transport;
                bsig_device_decode_timing(t6);
            }
            if (rc == bsig::kNeedsCpuPath) {
                rc = bsig_reads_from_bam_regions(slots->ctx[0], bam, n, rid.data(), qbeg.data(), qend.data(), 0, &res->reads[0]);
                bsig_device_decode_timing(t6);
                if (rc == BSIG_OK) rc = clone_rest();
            }
            if (rc) return rc;
            set_coverage(*res, n, rid.data(), qbeg.data(), qend.data());
        }
        return BSIG_OK;
        };
        rc = decode_once();
        if (rc == BSIG_ERR_NOMEM) {
            // the cache may be what fills the device: the index-driven decodes it keeps, the result buffers of idle
            // device lists and the free blocks go back to the driver, and the decode is tried once more
            drop_spare_device_memory(slots.get());
            rc = decode_once();
        }
        if (rc) return rc;
        T[2] = t6[5];
        T[1] = now_s() - t_dec - T[2];
        // driver allocation calls of the decode stage, layout included (they run on this thread and on the helper
        // that reserves the resident columns; the meter is per process: concurrent calls of other threads add to it)
        X[0] = AllocSnap::now().seconds_since(a_dec);
        bsig::decode_reservation_info(&X[8], &X[9]);
        if (!key.empty()) {
            // (what the reads hold on the device, slabs and an over-sized reservation included)
            res->bytes = res->reads[0]->pool.footprint();
            std::lock_guard<std::mutex> lock(g_cache.mu);
            bool still = false;                            // (bsig_cache_clear may have run meanwhile)
            for (auto &sp : g_cache.slot_sets) still = still || sp == slots;
            // ONE budget (env BAMSIGNALS_CACHE_GB per GPU) for whole files and index-driven decodes alike; above it
            // the index-driven decodes go first, oldest first, then the least recently used whole files.  A budget
            // of 0 keeps nothing but the entry in use.
            const int64_t budget = cache_budget_bytes();
            if (still && whole) {
                for (auto it = g_cache.resident.begin(); it != g_cache.resident.end();)
                    it = (*it)->key == rkey ? g_cache.resident.erase(it) : std::next(it);
                g_cache.resident.push_front(res);
            } else if (still) {
                // index-driven decodes: the last few (env BAMSIGNALS_REGION_CACHE, default 8; 0 = the reference's
                // behaviour, none), inside the same byte budget
                size_t keep = 8;
                if (const char *e = getenv("BAMSIGNALS_REGION_CACHE")) keep = (size_t)std::max(0, atoi(e));
                if (budget <= 0) keep = 0;
                if (keep) g_cache.regional.push_front(res);
                while (g_cache.regional.size() > keep) g_cache.regional.pop_back();
            }
            if (still) {
                auto held = [&]() {
                    int64_t t = 0;
                    for (auto &r : g_cache.resident) t += r->bytes;
                    for (auto &r : g_cache.regional) t += r->bytes;
                    return t;
                };
                while (held() > budget && !g_cache.regional.empty() && g_cache.regional.back() != res) g_cache.regional.pop_back();
                while (held() > budget && !g_cache.resident.empty() && g_cache.resident.back() != res) g_cache.resident.pop_back();
            }
        }
        }
    }
    if (!side_to_write.empty()) (void)bsig_reads_save(res->reads[0], side_to_write.c_str(), side_stamp.c_str());

    const double t_run = now_s();
    const AllocSnap a_run = AllocSnap::now();
    std::string gather;
    for (int attempt = 0; attempt < 2; ++attempt) {
    if (attempt) drop_spare_device_memory(slots.get());        // (out of device memory: once more with the cache's spare memory given back)
    if (!many) {
        // plan (ranges -> tiles in HBM), kernels, download -- timed apart
        bsig_plan *plan = nullptr;
        rc = bsig_plan_create(slots->ctx[0], res->reads[0], n, rid.data(), loc.data(), width, strand, &prm, &plan);
        X[3] = now_s() - t_run;
        if (rc == BSIG_OK && memcmp(off, bsig_plan_offsets(plan), (size_t)(n + 1) * sizeof(int64_t)) != 0)
            rc = fail(BSIG_ERR_ARG, "offsets do not match bsig_layout() for these parameters");
        if (rc == BSIG_OK) rc = bsig::plan_run_host_timed(plan, dest, &X[4], &X[5]);
        if (plan) bsig_plan_free(plan);
    } else {
        rc = run_on_slots(*slots, res->reads, n, rid.data(), loc.data(), width, strand, prm, dest, gather);
    }
    if (rc != BSIG_ERR_NOMEM) break;
    }
    T[3] = now_s() - t_run;
    X[1] = AllocSnap::now().seconds_since(a_run);
    {
        const AllocSnap a_end = AllocSnap::now();
        X[2] = a_end.seconds_since(a_begin);
        X[6] = (double)(a_end.calls - a_begin.calls);
    }
    if (rc == BSIG_OK) rc = bam_index_wait(bam);      // a damaged index fails the call, as it does in the reference's open
    T[4] = now_s() - t_begin;
    snprintf(g_call_route, sizeof g_call_route, "%zu GPU slot(s); reads: %s; result: %s", nd, how_decoded.c_str(),
             many ? gather.c_str() : "download");
    return rc;
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {
// the index files htslib's bam_index_load tries (ref: src/bamsignals.cpp:207), in its order (hts_idx_load looks for a
// CSI index first -- htslib is not vendored in the reference; this is its documented search order): <bam>.csi,
// <stem>.csi, <bam>.bai, <stem>.bai
std::vector<std::pair<std::string, bool>> index_candidates(const std::string &bam)
{
    std::string stem = bam;
    if (stem.size() > 4 && stem.compare(stem.size() - 4, 4, ".bam") == 0) stem.erase(stem.size() - 4);
    std::vector<std::pair<std::string, bool>> c = {{bam + ".csi", true}};
    if (stem != bam) c.push_back({stem + ".csi", true});
    c.push_back({bam + ".bai", false});
    if (stem != bam) c.push_back({stem + ".bai", false});
    return c;
}

// parses the index of b: the FIRST candidate that exists is the index, as in htslib's search (a file of a .csi name
// that is not a CSI index fails the open there -- bam_index_load returns NULL, ref: src/bamsignals.cpp:207-210 --
// and is not passed over for an older .bai beside it); 0, or an error with its message in b->idx_err
int load_index(bsig_bam *b)
{
    for (const auto &cand : index_candidates(b->path)) {
        struct stat sb;
        if (stat(cand.first.c_str(), &sb) != 0) continue;
        const int r = cand.second ? bsig::csi_load(cand.first, b->idx) : bsig::bai_load(cand.first, b->idx);
        if (r == 0) { b->from_csi = cand.second; return 0; }
        // a damaged BAI is reported as such; a .csi that is no CSI index gets the reference's own text
        if (!cand.second && r != BSIG_ERR_NOINDEX) { b->idx_err = bsig_last_error(); return r; }
        break;
    }
    b->idx_err = "BAM indexing file is not available for file " + b->path;
    return BSIG_ERR_NOINDEX;
}

// does some index file exist?  (cheap: the file-level calls parse the index in the background and only need to
// know now whether the reference's "not available" error is due; what the first existing candidate holds is the
// parse's business, see load_index)
bool index_present(const std::string &bam)
{
    for (const auto &cand : index_candidates(bam)) {
        struct stat sb;
        if (stat(cand.first.c_str(), &sb) == 0) return true;
    }
    return false;
}

// lazy: header now, index parsed by a helper thread (the north star's 10-MB BAI takes 35 ms; a whole-file
// decode never looks at it).  bam_index_wait() joins and reports what the parse found.
int bam_open_impl(const char *path, bool lazy, bsig_bam **out)
{
    if (!path || !out) return fail(BSIG_ERR_ARG, "NULL argument to bsig_bam_open");
    *out = nullptr;
    std::unique_ptr<bsig_bam> b(new bsig_bam);
    b->path = path;
    int rc = bsig::bam_read_header(b->path, b->hdr);
    if (rc == BSIG_ERR_IO) return fail(BSIG_ERR_IO, "Fail to open BAM file %s", path);
    if (rc) return rc;
    if (lazy && index_present(b->path)) {
        bsig_bam *raw = b.get();
        try {
            b->idx_job = std::async(std::launch::async, [raw] { return load_index(raw); }).share();
        } catch (const std::system_error &) {
            lazy = false;                                  // no thread to be had: parse here
        }
    } else {
        lazy = false;
    }
    if (!lazy) {
        rc = load_index(b.get());
        if (rc) return fail(rc, "%s", b->idx_err.c_str());
    }
    *out = b.release();
    return BSIG_OK;
}
}  // namespace

// the parsed index of an opened BAM (joins the background parse of the file-level calls' handles)
int bam_index_wait(const bsig_bam *b)
{
    if (b->idx_job.valid()) {
        const int rc = b->idx_job.get();
        if (rc) return fail(rc, "%s", b->idx_err.c_str());
    }
    return BSIG_OK;
}
const bsig::BaiIndex *bsig_bam_index(const bsig_bam *b)
{
    if (b->idx_job.valid()) b->idx_job.wait();
    return &b->idx;
}

extern "C" {

int bsig_bam_open(const char *path, bsig_bam **out) { return bam_open_impl(path, false, out); }

void bsig_bam_close(bsig_bam *b) { delete b; }

const char *bsig_bam_path(const bsig_bam *b) { return b ? b->path.c_str() : nullptr; }
}  // extern "C"
extern "C" {

int32_t bsig_bam_n_ref(const bsig_bam *b) { return b ? (int32_t)b->hdr.names.size() : 0; }

const char *bsig_bam_ref_name(const bsig_bam *b, int32_t rid)
{
    return (b && rid >= 0 && rid < (int32_t)b->hdr.names.size()) ? b->hdr.names[(size_t)rid].c_str() : nullptr;
}

int32_t bsig_bam_ref_len(const bsig_bam *b, int32_t rid)
{
    return (b && rid >= 0 && rid < (int32_t)b->hdr.lens.size()) ? b->hdr.lens[(size_t)rid] : -1;
}

int32_t bsig_bam_name2id(const bsig_bam *b, const char *name) { return (b && name) ? b->hdr.name2id(name) : -1; }

int bsig_bam_decode(bsig_bam *b, int64_t n_regions, const int32_t *rid, const int64_t *beg,
                    const int64_t *end, int32_t threads, bsig_columns *cols)
{
    if (!b || !cols) return fail(BSIG_ERR_ARG, "NULL argument to bsig_bam_decode");
    int rc;
    bsig::BamHeader h;
    if (n_regions < 0) {
        rc = bsig::bam_decode_all(b->path, threads, h, b->cols);
    } else {
        if (n_regions > 0 && (!rid || !beg || !end)) return fail(BSIG_ERR_ARG, "region arrays missing");
        std::vector<bsig::Region> rg((size_t)n_regions);
        for (int64_t i = 0; i < n_regions; ++i) rg[(size_t)i] = bsig::Region{rid[i], beg[i], end[i]};
        rc = bam_index_wait(b);
        if (rc) return rc;
        rc = bsig::bam_decode_regions(b->path, b->idx, rg, threads, h, b->cols);
    }
    if (rc) return rc;
    fill_columns(b, cols);
    return BSIG_OK;
}

int32_t bsig_effective_cpus(void) { return (int32_t)bsig::effective_cpus(); }

void bsig_last_call_timing(double *t6)
{
    for (int k = 0; k < 6; ++k) t6[k] = g_call_timing[k];
}

void bsig_last_call_timing_ex(double *t, int32_t n)
{
    for (int k = 0; k < n && k < 16; ++k) t[k] = k < 6 ? g_call_timing[k] : g_call_timing_ex[k - 6];
}

void bsig_bam_decode_timing(double *t6)
{
    for (int k = 0; k < 6; ++k) t6[k] = bsig::g_decode_timing[k];
}

static int pileup_core_impl(const char *bampath, int64_t n, const int32_t *seq_code, int32_t n_levels,
                            const char *const *levels, const int32_t *start, const int32_t *width,
                            const int32_t *strand, const int32_t *tlen_filter, int32_t n_tlen_filter,
                            int32_t mapqual, int32_t binsize, int32_t shift, int32_t ss, int32_t requiredF,
                            int32_t filteredF, int32_t pe_mid, int32_t maxgap, int32_t device, int32_t *out,
                            const int64_t *off, int32_t *const *dst)
{
    (void)maxgap;
    bsig_params p;
    memset(&p, 0, sizeof p);
    p.mode = binsize <= 0 ? BSIG_MODE_COUNT : BSIG_MODE_PROFILE;   // ref: allocateList :148
    p.mapqual = mapqual; p.binsize = binsize; p.shift = shift; p.ss = ss;
    p.requiredF = requiredF; p.filteredF = filteredF; p.pe_mid = pe_mid;
    if (n_tlen_filter != 0 && n_tlen_filter != 2) return fail(BSIG_ERR_ARG, "tlen_filter must have 0 or 2 elements");
    p.n_tlen_filter = n_tlen_filter;
    for (int k = 0; k < n_tlen_filter; ++k) p.tlen_filter[k] = tlen_filter[k];
    bsig::HostDest D;
    std::vector<int64_t> own_off;
    if (dst) {
        // in place: the layout is computed here (bsig_layout); bamCount's is one vector, dst[0]
        if (n < 0 || (n > 0 && !width)) return fail(BSIG_ERR_ARG, "range arrays missing");
        own_off.resize((size_t)n + 1);
        bsig_layout(n, width, binsize, ss, own_off.data());
        D.off = own_off.data(); D.n = n;
        if (binsize <= 0) D.flat = dst[0]; else D.ptrs = dst;
    } else {
        D.flat = out; D.off = off; D.n = n;
    }
    return file_level(bampath, n, seq_code, n_levels, levels, start, width, strand, p, device, D);
}

int bsig_pileup_core(const char *bampath, int64_t n, const int32_t *seq_code, int32_t n_levels,
                     const char *const *levels, const int32_t *start, const int32_t *width,
                     const int32_t *strand, const int32_t *tlen_filter, int32_t n_tlen_filter,
                     int32_t mapqual, int32_t binsize, int32_t shift, int32_t ss, int32_t requiredF,
                     int32_t filteredF, int32_t pe_mid, int32_t maxgap, int32_t device, int32_t *out,
                     const int64_t *off)
{
    return pileup_core_impl(bampath, n, seq_code, n_levels, levels, start, width, strand, tlen_filter, n_tlen_filter, mapqual, binsize,
                            shift, ss, requiredF, filteredF, pe_mid, maxgap, device, out, off, nullptr);
}

int bsig_pileup_core_into(const char *bampath, int64_t n, const int32_t *seq_code, int32_t n_levels,
                          const char *const *levels, const int32_t *start, const int32_t *width,
                          const int32_t *strand, const int32_t *tlen_filter, int32_t n_tlen_filter,
                          int32_t mapqual, int32_t binsize, int32_t shift, int32_t ss, int32_t requiredF,
                          int32_t filteredF, int32_t pe_mid, int32_t maxgap, int32_t device, int32_t *const *dst)
{
    if (!dst) return fail(BSIG_ERR_ARG, "destinations missing");
    return pileup_core_impl(bampath, n, seq_code, n_levels, levels, start, width, strand, tlen_filter, n_tlen_filter, mapqual, binsize,
                            shift, ss, requiredF, filteredF, pe_mid, maxgap, device, nullptr, nullptr, dst);
}

static int coverage_core_impl(const char *bampath, int64_t n, const int32_t *seq_code, int32_t n_levels,
                              const char *const *levels, const int32_t *start, const int32_t *width,
                              const int32_t *strand, const int32_t *tlen_filter, int32_t n_tlen_filter,
                              int32_t mapqual, int32_t requiredF, int32_t filteredF, int32_t tspan,
                              int32_t maxgap, int32_t device, int32_t *out, const int64_t *off, int32_t *const *dst)
{
    (void)maxgap;
    bsig_params p;
    memset(&p, 0, sizeof p);
    p.mode = BSIG_MODE_COVERAGE;
    p.mapqual = mapqual; p.binsize = 1; p.requiredF = requiredF; p.filteredF = filteredF; p.tspan = tspan;
    if (n_tlen_filter != 0 && n_tlen_filter != 2) return fail(BSIG_ERR_ARG, "tlen_filter must have 0 or 2 elements");
    p.n_tlen_filter = n_tlen_filter;
    for (int k = 0; k < n_tlen_filter; ++k) p.tlen_filter[k] = tlen_filter[k];
    bsig::HostDest D;
    std::vector<int64_t> own_off;
    if (dst) {
        if (n < 0 || (n > 0 && !width)) return fail(BSIG_ERR_ARG, "range arrays missing");
        own_off.resize((size_t)n + 1);
        bsig_layout(n, width, 1, 0, own_off.data());
        D.off = own_off.data(); D.n = n; D.ptrs = dst;
    } else {
        D.flat = out; D.off = off; D.n = n;
    }
    return file_level(bampath, n, seq_code, n_levels, levels, start, width, strand, p, device, D);
}

int bsig_coverage_core(const char *bampath, int64_t n, const int32_t *seq_code, int32_t n_levels,
                       const char *const *levels, const int32_t *start, const int32_t *width,
                       const int32_t *strand, const int32_t *tlen_filter, int32_t n_tlen_filter,
                       int32_t mapqual, int32_t requiredF, int32_t filteredF, int32_t tspan,
                       int32_t maxgap, int32_t device, int32_t *out, const int64_t *off)
{
    return coverage_core_impl(bampath, n, seq_code, n_levels, levels, start, width, strand, tlen_filter, n_tlen_filter, mapqual,
                              requiredF, filteredF, tspan, maxgap, device, out, off, nullptr);
}

int bsig_coverage_core_into(const char *bampath, int64_t n, const int32_t *seq_code, int32_t n_levels,
                            const char *const *levels, const int32_t *start, const int32_t *width,
                            const int32_t *strand, const int32_t *tlen_filter, int32_t n_tlen_filter,
                            int32_t mapqual, int32_t requiredF, int32_t filteredF, int32_t tspan,
                            int32_t maxgap, int32_t device, int32_t *const *dst)
{
    if (!dst) return fail(BSIG_ERR_ARG, "destinations missing");
    return coverage_core_impl(bampath, n, seq_code, n_levels, levels, start, width, strand, tlen_filter, n_tlen_filter, mapqual,
                              requiredF, filteredF, tspan, maxgap, device, nullptr, nullptr, dst);
}

int bsig_write_sam_as_bam_and_index(const char *sampath, const char *bampath)
{
    if (!sampath || !bampath) return fail(BSIG_ERR_ARG, "NULL path");
    return bsig::sam_to_bam_and_index(sampath, bampath);
}

static int write_columns_impl(const char *bampath, int32_t n_ref, const char *const *ref_names, const bsig_columns *c,
                              int32_t level, int32_t l_seq, uint64_t seed);

int bsig_write_columns_as_bam(const char *bampath, int32_t n_ref, const char *const *ref_names,
                              const bsig_columns *c, int32_t level)
{
    return write_columns_impl(bampath, n_ref, ref_names, c, level, 0, 0);
}

int bsig_write_columns_as_bam_with_seq(const char *bampath, int32_t n_ref, const char *const *ref_names,
                                       const bsig_columns *c, int32_t level, int32_t l_seq, uint64_t seed)
{
    if (l_seq <= 0) return fail(BSIG_ERR_ARG, "l_seq must be positive");
    return write_columns_impl(bampath, n_ref, ref_names, c, level, l_seq, seed);
}

static int write_columns_impl(const char *bampath, int32_t n_ref, const char *const *ref_names, const bsig_columns *c,
                              int32_t level, int32_t l_seq, uint64_t seed)
{
    if (!bampath || !c || (n_ref > 0 && !ref_names)) return fail(BSIG_ERR_ARG, "NULL argument");
    if (c->n_ref != n_ref) return fail(BSIG_ERR_ARG, "n_ref does not match the columns");
    if (c->n_reads > 0 && (!c->cigar_off || !c->cigar)) return fail(BSIG_ERR_ARG, "the writer needs cigar_off + cigar");
    bsig::BamHeader h;
    h.text = "@HD\tVN:1.0\tSO:coordinate\n";
    for (int r = 0; r < n_ref; ++r) {
        h.names.emplace_back(ref_names[r]);
        h.lens.push_back(c->ref_len[r]);
        h.text += "@SQ\tSN:" + h.names.back() + "\tLN:" + std::to_string(c->ref_len[r]) + "\n";
    }
    bsig::BamWriter w;
    int rc = w.open(bampath, h, level > 0 ? level : 1);
    if (rc) return rc;
    const char *how = getenv("BAMSIGNALS_WRITER");      // "serial": record by record (testing)
    if (how && !strcmp(how, "serial") && !l_seq) {
        for (int r = 0; r < n_ref; ++r)
            for (int64_t i = c->ref_off[r]; i < c->ref_off[r + 1]; ++i) {
                rc = w.write_core(r, c->pos[i], c->flag[i], c->mapq[i], c->tlen[i], c->cigar + c->cigar_off[i],
                                  (int)(c->cigar_off[i + 1] - c->cigar_off[i]));
                if (rc) return rc;
            }
    } else {
        rc = w.write_columns(n_ref, c->ref_off, c->pos, c->flag, c->mapq, c->tlen, c->cigar_off, c->cigar, 0, l_seq, seed);
        if (rc) return rc;
    }
    return w.close();
}

// checkList / fastWidth (ref: src/CountSignals.cpp:4-29)
int32_t bsig_check_list(int64_t n, const int32_t *is_int, const int32_t *n_dim, const int32_t *dim0, int32_t ss)
{
    if (n > 0 && (!is_int || (ss && (!n_dim || !dim0)))) return 0;
    for (int64_t i = 0; i < n; ++i) {
        if (!is_int[i]) return 0;                                   // ref: :8
        if (ss && (n_dim[i] != 2 || dim0[i] != 2)) return 0;        // ref: :11-12
    }
    return 1;
}

void bsig_fast_width(int64_t n, const int64_t *length, int32_t ss, int32_t *width)
{
    const int64_t div = ss ? 2 : 1;                                 // ref: :21
    for (int64_t i = 0; i < n; ++i) width[i] = (int32_t)(length[i] / div);
}

int bsig_scatter_segments(int64_t n, const int32_t *src, const int64_t *src_off, int32_t *dst,
                          const int64_t *dst_off, const int64_t *which)
{
    bsig::HostDest D;
    D.flat = dst; D.off = dst_off; D.n = 0;
    return scatter_segments_to(n, src, src_off, D, which);
}

struct bsig_segmap {
    bsig_ctx *ctx = nullptr;
    int64_t n = 0;
    int64_t *d_src_off = nullptr, *d_dst_off = nullptr, *d_which = nullptr;
    int *d_overflow = nullptr;      // set by a narrow run whose sender had more exceptions than its list holds
    int64_t src_cells = 0;          // src_off[n]: a narrow message must code at least this many cells
};

int bsig_segmap_create(bsig_ctx *ctx, int64_t n, const int64_t *src_off, int64_t n_dst, const int64_t *dst_off,
                       const int64_t *which, bsig_segmap **out)
{
    if (!ctx || !out) return fail(BSIG_ERR_ARG, "NULL argument to bsig_segmap_create");
    *out = nullptr;
    if (n < 0 || n_dst < 0 || (n > 0 && (!src_off || !dst_off || !which))) return fail(BSIG_ERR_ARG, "segment tables missing");
    for (int64_t k = 0; k < n; ++k) {
        const int64_t len = src_off[k + 1] - src_off[k];
        const int64_t i = which[k];
        if (len < 0 || src_off[k] < 0 || i < 0 || i >= n_dst) return fail(BSIG_ERR_ARG, "bad segment %lld", (long long)k);
        if (dst_off[i] < 0 || len != dst_off[i + 1] - dst_off[i]) return fail(BSIG_ERR_ARG, "segment %lld does not fit its destination", (long long)k);
    }
    std::unique_ptr<bsig_segmap> M(new bsig_segmap);
    M->ctx = ctx;
    M->n = n;
    M->src_cells = n ? src_off[n] : 0;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    hipError_t e = hipMalloc((void **)&M->d_src_off, (size_t)(n + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&M->d_dst_off, (size_t)(n_dst + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&M->d_which, (size_t)std::max<int64_t>(n, 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&M->d_overflow, sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(M->d_overflow, 0, sizeof(int), st);
    const int64_t zero = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(M->d_src_off, n ? src_off : &zero, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(M->d_dst_off, dst_off ? dst_off : &zero, (size_t)(n_dst + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n) e = hipMemcpyAsync(M->d_which, which, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        bsig_segmap_free(M.release());
        return fail(e == hipErrorOutOfMemory ? BSIG_ERR_NOMEM : BSIG_ERR_DEVICE, "segment map upload failed: %s", hipGetErrorString(e));
    }
    *out = M.release();
    return BSIG_OK;
}

int bsig_segmap_run(bsig_segmap *M, const int32_t *src_dev, int32_t *dst_dev)
{
    if (!M) return fail(BSIG_ERR_ARG, "segment map is NULL");
    if (M->n == 0) return BSIG_OK;
    if (!src_dev || !dst_dev) return fail(BSIG_ERR_ARG, "NULL device buffer");
    HIP_TRY(hipSetDevice(M->ctx->device));
    HIP_TRY(bsig::launch_place_segments(M->n, src_dev, M->d_src_off, dst_dev, M->d_dst_off, M->d_which, M->ctx->stream));
    return BSIG_OK;
}

int64_t bsig_narrow_bytes(int64_t n_cells, int64_t cap)
{
    return n_cells < 0 || cap < 0 ? 0 : bsig::narrow_message_bytes(n_cells, cap);
}

int bsig_narrow_pack(bsig_ctx *ctx, const int32_t *src_dev, int64_t n_cells, void *msg_dev, int64_t cap)
{
    if (!ctx || !msg_dev || (n_cells > 0 && !src_dev) || n_cells < 0 || cap < 0) return fail(BSIG_ERR_ARG, "bad argument to bsig_narrow_pack");
    if (((uintptr_t)src_dev & 15) != 0 || ((uintptr_t)msg_dev & 15) != 0) return fail(BSIG_ERR_ARG, "device buffers must be 16-byte aligned");
    if (n_cells > (int64_t)UINT32_MAX) return fail(BSIG_ERR_ARG, "a narrow message holds at most 2^32 - 1 cells (an exception names its cell in 32 bits)");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(bsig::launch_narrow_pack(src_dev, n_cells, msg_dev, cap, ctx->stream));
    return BSIG_OK;
}

int bsig_narrow_count(bsig_ctx *ctx, const void *msg_dev, int64_t *n_exceptions)
{
    if (!ctx || !msg_dev || !n_exceptions) return fail(BSIG_ERR_ARG, "NULL argument to bsig_narrow_count");
    HIP_TRY(hipSetDevice(ctx->device));
    uint32_t k = 0;
    HIP_TRY(hipMemcpyAsync(&k, msg_dev, sizeof k, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *n_exceptions = (int64_t)k;
    return BSIG_OK;
}

int bsig_segmap_run_narrow(bsig_segmap *M, const void *msg_dev, int64_t n_cells, int64_t cap, int32_t *dst_dev)
{
    if (!M) return fail(BSIG_ERR_ARG, "segment map is NULL");
    if (M->n == 0) return BSIG_OK;
    if (!msg_dev || !dst_dev || n_cells < 0 || cap < 0) return fail(BSIG_ERR_ARG, "bad argument to bsig_segmap_run_narrow");
    if (n_cells < M->src_cells) return fail(BSIG_ERR_ARG, "the message codes %lld cells, the map's segments span %lld", (long long)n_cells, (long long)M->src_cells);
    if (((uintptr_t)msg_dev & 15) != 0) return fail(BSIG_ERR_ARG, "device buffers must be 16-byte aligned");
    HIP_TRY(hipSetDevice(M->ctx->device));
    HIP_TRY(bsig::launch_place_narrow(M->n, msg_dev, n_cells, cap, M->d_src_off, dst_dev, M->d_dst_off, M->d_which, M->d_overflow, M->ctx->stream));
    return BSIG_OK;
}

int bsig_segmap_narrow_overflowed(bsig_segmap *M, int *overflowed)
{
    if (!M || !overflowed) return fail(BSIG_ERR_ARG, "NULL argument to bsig_segmap_narrow_overflowed");
    HIP_TRY(hipSetDevice(M->ctx->device));
    HIP_TRY(hipMemcpyAsync(overflowed, M->d_overflow, sizeof(int), hipMemcpyDeviceToHost, M->ctx->stream));
    HIP_TRY(hipStreamSynchronize(M->ctx->stream));
    return BSIG_OK;
}

void bsig_segmap_free(bsig_segmap *M)
{
    if (!M) return;
    (void)hipSetDevice(M->ctx->device);
    if (M->d_overflow) (void)hipFree(M->d_overflow);
    if (M->d_src_off) (void)hipFree(M->d_src_off);
    if (M->d_dst_off) (void)hipFree(M->d_dst_off);
    if (M->d_which) (void)hipFree(M->d_which);
    delete M;
}

void bsig_cache_clear(void)
{
    // everything is moved out under the lock and released outside it (freeing HBM takes a while); what a
    // running call holds -- its contexts, scratch, resident reads, communicators -- lives until it returns
    Cache old;
    {
        std::lock_guard<std::mutex> lock(g_cache.mu);
        old.resident.swap(g_cache.resident);
        old.regional.swap(g_cache.regional);
        old.bams.swap(g_cache.bams);
        old.slot_sets.swap(g_cache.slot_sets);
    }
    old.clear();
    bsig::exchange_close_all();
    bsig::release_decode_scratch();
}

int bsig_debug_block_table(const char *path, int64_t *n_blocks, uint64_t *checksum)
{
    // the BGZF block table as the device-side decode builds it (env BAMSIGNALS_SCAN=mmap|pread picks the walk):
    // tests compare the two walks
    if (!path || !n_blocks || !checksum) return fail(BSIG_ERR_ARG, "NULL argument");
    bsig::BgzfFile f;
    const int rc = f.open(path);
    if (rc) return rc;
    uint64_t h = 1469598103934665603ull;
    for (const bsig::BgzfBlock &b : f.blocks())
        for (uint64_t v : {(uint64_t)b.coff, (uint64_t)b.csize, (uint64_t)b.doff, (uint64_t)b.dlen, (uint64_t)b.isize, (uint64_t)b.crc}) {
            h ^= v;
            h *= 1099511628211ull;
        }
    *n_blocks = (int64_t)f.blocks().size();
    *checksum = h;
    return BSIG_OK;
}

int bsig_debug_block_table_progressive(const char *path, int64_t head_bytes, int64_t *n_head, int64_t *n_blocks, uint64_t *checksum)
{
    // the same table in two steps (BgzfFile::open_progressive): *n_head = blocks tabulated before the rest was
    // waited for (== *n_blocks when the file was tabulated whole at once)
    if (!path || !n_head || !n_blocks || !checksum || head_bytes <= 0) return fail(BSIG_ERR_ARG, "bad argument");
    bsig::BgzfFile f;
    int rc = f.open_progressive(path, (uint64_t)head_bytes);
    if (rc) return rc;
    *n_head = (int64_t)f.blocks().size();
    rc = f.finish();
    if (rc) return rc;
    uint64_t h = 1469598103934665603ull;
    for (const bsig::BgzfBlock &b : f.blocks())
        for (uint64_t v : {(uint64_t)b.coff, (uint64_t)b.csize, (uint64_t)b.doff, (uint64_t)b.dlen, (uint64_t)b.isize, (uint64_t)b.crc}) {
            h ^= v;
            h *= 1099511628211ull;
        }
    *n_blocks = (int64_t)f.blocks().size();
    *checksum = h;
    return BSIG_OK;
}

int64_t bsig_debug_scratch_allocs(void)
{
    // allocations made so far for the multi-GPU result path's cached buffers (tests: flat across resident calls)
    std::lock_guard<std::mutex> lock(g_cache.mu);
    int64_t n = 0;
    for (auto &sp : g_cache.slot_sets) n += sp->scratch_allocs.load();
    return n;
}

const char *bsig_last_call_route(void) { return g_call_route; }

}  // extern "C"
