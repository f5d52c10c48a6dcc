// File-level half of the C ABI: BAM handles, the drop-in bsig_pileup_core / bsig_coverage_core,
// the BAM writers.  Pure host code; the compute goes through bsig_reads_upload / bsig_plan_*.
#include <hip/hip_runtime.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
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
    bool csi_only = false;      // indexed by a .csi file only: accepted, never queried (whole-file decodes)
    bsig::HostColumns cols;
};

namespace {

// stage seconds of the calling thread's last file-level call: open (header + BAI), decode,
// upload + HBM layout, plan + kernels + result download, total; [5] = 1 if the BAM was already
// resident in HBM
thread_local double g_call_timing[6] = {0, 0, 0, 0, 0, 0};
// how the calling thread's last file-level call was carried out (bsig_last_call_route)
thread_local char g_call_route[160] = "";
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
struct Resident {
    std::string key;
    std::vector<bsig_reads *> reads;     // one per slot
    int64_t bytes = 0;                   // per GPU
    ~Resident() { for (bsig_reads *r : reads) if (r) bsig_reads_free(r); }
};
struct OpenBam {
    std::string key;
    bsig_bam *bam = nullptr;
    ~OpenBam() { if (bam) bsig_bam_close(bam); }
};
struct Slots {
    std::vector<int> devices;
    std::vector<bsig_ctx *> ctx;
    ~Slots() { for (bsig_ctx *c : ctx) if (c) bsig_ctx_destroy(c); }
};

struct Cache {
    std::mutex mu;                                       // guards everything below
    std::mutex decode_mu;                                // cold decodes take turns
    std::shared_ptr<Slots> slots;
    std::list<std::shared_ptr<OpenBam>> bams;            // most recently used first
    std::list<std::shared_ptr<Resident>> resident;       // most recently used first
    void clear()
    {
        resident.clear();
        bams.clear();
        slots.reset();
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

// ---- several GPUs: ranges are independent (each owns its output, ref: src/bamsignals.cpp:164,181,186)
// The (rid, loc)-sorted ranges (ref: :222-226,246) are dealt round-robin to the GPUs; every GPU plans
// and runs its shard on its own stream, driven by its own host thread.  The shards then
//   "xgmi" (default): travel to the first GPU over xGMI (RCCL grouped send/recv, or peer copies:
//           collect.h), are put into the caller's range order there by one kernel, and leave for the
//           host in ONE copy (no host-side reassembly);
//   "pcie": are pulled over each GPU's own PCIe link into page-locked buffers and put in place by host
//           threads (bsig_scatter_segments).
int run_on_slots(const Slots &sl, const std::vector<bsig_reads *> &reads, int64_t n, const int32_t *rid, const int32_t *loc,
                 const int32_t *width, const int32_t *strand, const bsig_params &prm, int32_t *out, const int64_t *off,
                 const char **gather_name)
{
    const size_t nd = sl.ctx.size();
    std::vector<int64_t> order((size_t)n);
    for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
        if (rid[a] != rid[b]) return rid[a] < rid[b];
        return loc[a] < loc[b];
    });
    struct Shard {
        std::vector<int64_t> which;
        std::vector<int32_t> rid, loc, len, strand;
        bsig_plan *plan = nullptr;
        int32_t *host = nullptr;
        int32_t *dev = nullptr;
        int64_t cells = 0;
    };
    std::vector<Shard> sh(nd);
    for (int64_t k = 0; k < n; ++k) {
        Shard &S = sh[(size_t)(k % (int64_t)nd)];
        const int64_t i = order[(size_t)k];
        S.which.push_back(i);
        S.rid.push_back(rid[i]); S.loc.push_back(loc[i]);
        S.len.push_back(width[i]); S.strand.push_back(strand[i]);
    }
    const bool pcie = env_is("BAMSIGNALS_GATHER", "pcie");
    *gather_name = pcie ? "pcie" : "xgmi";
    std::vector<void *> dev_tmp;                       // device buffers of the xgmi route (first GPU)
    auto cleanup = [&]() {
        for (size_t k = 0; k < nd; ++k) {
            Shard &S = sh[k];
            if (S.plan) bsig_plan_free(S.plan);
            if (S.host) bsig_host_free(S.host);
            if (S.dev) { (void)hipSetDevice(sl.devices[k]); (void)hipFree(S.dev); }
        }
        if (!dev_tmp.empty()) (void)hipSetDevice(sl.devices[0]);
        for (void *p : dev_tmp) (void)hipFree(p);
    };
    // plan + launch, one host thread per GPU
    int rc = for_each_slot(nd, [&](size_t k) -> int {
        Shard &S = sh[k];
        int r = bsig_plan_create(sl.ctx[k], reads[k], (int64_t)S.which.size(), S.rid.data(), S.loc.data(), S.len.data(),
                                 S.strand.data(), &prm, &S.plan);
        if (r) return r;
        S.cells = bsig_plan_cells(S.plan);
        if (pcie) {
            r = bsig_host_alloc(S.cells * (int64_t)sizeof(int32_t), (void **)&S.host);
            if (r) return r;
            r = bsig_plan_run_host_async(S.plan, S.host);
        } else {
            HIP_TRY(hipSetDevice(sl.devices[k]));
            HIP_TRY(hipMalloc((void **)&S.dev, (size_t)std::max<int64_t>(S.cells, 4) * sizeof(int32_t)));
            r = bsig_plan_run(S.plan, S.dev);
        }
        if (r) return r;
        return bsig_ctx_sync(sl.ctx[k]);
    });
    if (rc) { cleanup(); return rc; }
    if (pcie) {
        for (size_t k = 0; k < nd && rc == BSIG_OK; ++k)
            rc = bsig_scatter_segments((int64_t)sh[k].which.size(), sh[k].host, bsig_plan_offsets(sh[k].plan), out, off,
                                       sh[k].which.data());
        cleanup();
        return rc;
    }
    // ---- xgmi: gather on the first GPU, reassemble there, one download ------------------------------
    const int64_t total = off[n];
    std::vector<size_t> goff(nd), glen(nd);
    std::vector<const uint8_t *> src(nd);
    std::vector<int64_t> seg_off, seg_which;           // all shards' segments behind each other
    seg_off.reserve((size_t)n + 1);
    seg_which.reserve((size_t)n);
    int64_t cells = 0;
    for (size_t k = 0; k < nd; ++k) {
        goff[k] = (size_t)cells * sizeof(int32_t);
        glen[k] = (size_t)sh[k].cells * sizeof(int32_t);
        src[k] = (const uint8_t *)sh[k].dev;
        const int64_t *po = bsig_plan_offsets(sh[k].plan);
        for (size_t j = 0; j < sh[k].which.size(); ++j) {
            seg_off.push_back(cells + po[j]);
            seg_which.push_back(sh[k].which[j]);
        }
        cells += sh[k].cells;
    }
    seg_off.push_back(cells);
    if (cells != total) { cleanup(); return fail(BSIG_ERR_ARG, "offsets do not match bsig_layout() for these parameters"); }
    if (total == 0) { cleanup(); return BSIG_OK; }
    auto body = [&]() -> int {
        HIP_TRY(hipSetDevice(sl.devices[0]));
        hipStream_t st = (hipStream_t)bsig_ctx_stream(sl.ctx[0]);
        int32_t *d_gather = nullptr, *d_final = nullptr;
        int64_t *d_seg_off = nullptr, *d_which = nullptr, *d_dst_off = nullptr;
        auto dalloc = [&](void **p, size_t bytes) { const hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 16)); if (e == hipSuccess) dev_tmp.push_back(*p); return e; };
        HIP_TRY(dalloc((void **)&d_gather, (size_t)total * sizeof(int32_t)));
        HIP_TRY(dalloc((void **)&d_final, (size_t)total * sizeof(int32_t)));
        HIP_TRY(dalloc((void **)&d_seg_off, seg_off.size() * sizeof(int64_t)));
        HIP_TRY(dalloc((void **)&d_which, std::max<size_t>(seg_which.size(), 1) * sizeof(int64_t)));
        HIP_TRY(dalloc((void **)&d_dst_off, (size_t)(n + 1) * sizeof(int64_t)));
        HIP_TRY(hipMemcpyAsync(d_seg_off, seg_off.data(), seg_off.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_which, seg_which.data(), seg_which.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_dst_off, off, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
        bsig::Exchange *ex = nullptr;
        const char *transport = nullptr;
        int r = bsig::exchange_open(sl.ctx, &ex, &transport);
        if (r) return r;
        *gather_name = !strcmp(transport, "rccl") ? "xgmi/rccl" : "xgmi/peer";
        r = bsig::exchange_gather(ex, src, glen, (uint8_t *)d_gather, goff);
        if (r) return r;
        for (size_t k = 1; k < nd; ++k) {              // the senders' streams (RCCL queues the sends there)
            r = bsig_ctx_sync(sl.ctx[k]);
            if (r) return r;
        }
        HIP_TRY(hipSetDevice(sl.devices[0]));
        HIP_TRY(bsig::launch_place_segments((int64_t)seg_which.size(), d_gather, d_seg_off, d_final, d_dst_off, d_which, st));
        return bsig::download_to_host(sl.ctx[0], d_final, out, (size_t)total * sizeof(int32_t));
    };
    rc = body();
    if (rc) for (size_t k = 0; k < nd; ++k) (void)bsig_ctx_sync(sl.ctx[k]);
    cleanup();
    return rc;
}

// the common body of pileup_core / coverage_core
int file_level(const char *bampath, int64_t n, const int32_t *seq_code, int32_t n_levels,
               const char *const *levels, const int32_t *start, const int32_t *width,
               const int32_t *strand, const bsig_params &prm, int32_t device, int32_t *out,
               const int64_t *off)
{
    if (!bampath) return fail(BSIG_ERR_ARG, "bampath is NULL");
    if (n < 0 || (n > 0 && (!seq_code || !start || !width || !strand || !levels)))
        return fail(BSIG_ERR_ARG, "range arrays missing");
    if (!off) return fail(BSIG_ERR_ARG, "offsets missing");
    double *T = g_call_timing;
    for (int k = 0; k < 6; ++k) T[k] = 0;
    g_call_route[0] = 0;
    const double t_begin = now_s();
    // ref: Bamfile ctor :200-214 opens file + index on every call; here an unchanged file (same
    // size and mtime of the BAM and of its index) reuses the parsed header and BAI
    // (keyed by the resolved path: "a.bam" and "./a.bam" are one file, one resident copy)
    char resolved[4096];
    const std::string canon = realpath(bampath, resolved) ? std::string(resolved) : std::string(bampath);
    const std::string key = file_key(canon);
    const std::string bkey = key.empty() ? std::string() : key + "#" + file_key(canon + ".bai");
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
        rc = bsig_bam_open(bampath, &fresh);
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

    // the GPUs of this call (a changed list drops everything that lives on the old one)
    const std::vector<int> devs = pick_devices(device);
    std::shared_ptr<Slots> slots;
    {
        std::lock_guard<std::mutex> lock(g_cache.mu);
        if (!g_cache.slots || g_cache.slots->devices != devs) {
            g_cache.resident.clear();
            g_cache.slots.reset();
            auto fresh = std::make_shared<Slots>();
            fresh->devices = devs;
            for (int d : devs) {
                bsig_ctx *c = nullptr;
                rc = bsig_ctx_create(d, nullptr, &c);
                if (rc) return rc;
                fresh->ctx.push_back(c);
            }
            g_cache.slots = fresh;
        }
        slots = g_cache.slots;
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
    if (bam->csi_only) whole = true;

    // ---- the reads of this call on every GPU -------------------------------------------------------
    std::shared_ptr<Resident> res;
    std::string how_decoded = "resident";
    if (!key.empty()) {
        std::lock_guard<std::mutex> lock(g_cache.mu);
        for (auto it = g_cache.resident.begin(); it != g_cache.resident.end(); ++it)
            if ((*it)->key == bkey) {
                res = *it;
                g_cache.resident.erase(it);
                g_cache.resident.push_front(res);
                break;
            }
    }
    if (res) {
        T[5] = 1;
    } else {
        std::lock_guard<std::mutex> dlock(g_cache.decode_mu);
        const double t_dec = now_s();
        res = std::make_shared<Resident>();
        res->key = bkey;
        res->reads.assign(nd, nullptr);
        auto clone_rest = [&]() -> int {
            return for_each_slot(nd, [&](size_t k) -> int {
                return k == 0 ? (int)BSIG_OK : bsig_reads_clone(res->reads[0], slots->ctx[k], &res->reads[k]);
            });
        };
        double t6[6] = {0, 0, 0, 0, 0, 0};
        if (whole) {
            const std::string side = key.empty() ? std::string() : sidecar_path(bampath);
            // (the reads file is tied to the CONTENT it was made from -- size and mtime of the BAM and of its
            // index -- not to the spelling of the path it was reached by)
            const std::string side_stamp = file_stamp(bampath) + "#" + file_stamp(std::string(bampath) + ".bai");
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
                if (!side.empty()) (void)bsig_reads_save(res->reads[0], side.c_str(), side_stamp.c_str());   // best effort
            }
        } else {
            // index-driven: only the blocks the BAI lists for the ranges (ref: one bam_itr_queryi per
            // chunk of ranges, :252-267), parsed on the first GPU like the whole file; not cached
            std::vector<int64_t> beg((size_t)n), end((size_t)n);
            for (int64_t i = 0; i < n; ++i) {
                beg[(size_t)i] = (int64_t)loc[(size_t)i] - ext;
                end[(size_t)i] = (int64_t)loc[(size_t)i] + width[i] + ext;
            }
            rc = bsig_reads_from_bam_regions(slots->ctx[0], bam, n, rid.data(), beg.data(), end.data(), 0, &res->reads[0]);
            bsig_device_decode_timing(t6);
            if (rc == BSIG_OK) rc = clone_rest();
            if (rc) return rc;
            how_decoded = "index-driven decode";
        }
        T[2] = t6[5];
        T[1] = now_s() - t_dec - T[2];
        if (whole && !key.empty()) {
            bsig_reads_info inf;
            if (bsig_reads_get_info(res->reads[0], &inf) == BSIG_OK) res->bytes = inf.hbm_bytes;
            std::lock_guard<std::mutex> lock(g_cache.mu);
            if (g_cache.slots == slots) {                  // (the GPU list may have changed meanwhile)
                g_cache.resident.push_front(res);
                const int64_t budget = cache_budget_bytes();
                int64_t held = 0;
                for (auto it = g_cache.resident.begin(); it != g_cache.resident.end();) {
                    held += (*it)->bytes;
                    if (held > budget && it != g_cache.resident.begin()) it = g_cache.resident.erase(it);
                    else ++it;
                }
            }
        }
    }

    const double t_run = now_s();
    const char *gather = "";
    if (!many) rc = bsig_pileup_columns(slots->ctx[0], res->reads[0], n, rid.data(), loc.data(), width, strand, &prm, out, off);
    else rc = run_on_slots(*slots, res->reads, n, rid.data(), loc.data(), width, strand, prm, out, off, &gather);
    T[3] = now_s() - t_run;
    T[4] = now_s() - t_begin;
    snprintf(g_call_route, sizeof g_call_route, "%zu GPU slot(s); reads: %s; result: %s", nd, how_decoded.c_str(),
             many ? gather : "download");
    return rc;
}

}  // namespace

extern "C" {

int bsig_bam_open(const char *path, bsig_bam **out)
{
    if (!path || !out) return fail(BSIG_ERR_ARG, "NULL argument to bsig_bam_open");
    *out = nullptr;
    std::unique_ptr<bsig_bam> b(new bsig_bam);
    b->path = path;
    int rc = bsig::bam_read_header(b->path, b->hdr);
    if (rc == BSIG_ERR_IO) return fail(BSIG_ERR_IO, "Fail to open BAM file %s", path);
    if (rc) return rc;
    rc = bsig::bai_load(b->path + ".bai", b->idx);
    if (rc == BSIG_ERR_NOINDEX) {
        // samtools also accepts foo.bai next to foo.bam
        std::string alt = b->path;
        if (alt.size() > 4 && alt.compare(alt.size() - 4, 4, ".bam") == 0) {
            alt.replace(alt.size() - 4, 4, ".bai");
            if (bsig::bai_load(alt, b->idx) == 0) rc = 0;
        }
        if (rc) {
            // htslib's bam_index_load (ref: src/bamsignals.cpp:207) also accepts a CSI index (references
            // beyond 2^29 bp need one).  Its bins are not read here: a file indexed that way is always
            // decoded whole, which needs no index at all
            for (const std::string &cand : {b->path + ".csi", alt.size() > 4 ? alt.substr(0, alt.size() - 4) + ".csi" : std::string()}) {
                if (cand.empty()) continue;
                FILE *f = fopen(cand.c_str(), "rb");
                if (!f) continue;
                unsigned char magic[4] = {0, 0, 0, 0};
                const bool gz = fread(magic, 1, 4, f) == 4 && magic[0] == 31 && magic[1] == 139;     // CSI files are BGZF-compressed
                fclose(f);
                if (gz) { b->csi_only = true; rc = 0; break; }
            }
        }
        if (rc) return fail(BSIG_ERR_NOINDEX, "BAM indexing file is not available for file %s", path);
    }
    if (rc) return rc;
    *out = b.release();
    return BSIG_OK;
}

void bsig_bam_close(bsig_bam *b) { delete b; }

const char *bsig_bam_path(const bsig_bam *b) { return b ? b->path.c_str() : nullptr; }
}  // extern "C"
const bsig::BaiIndex *bsig_bam_index(const bsig_bam *b) { return &b->idx; }
bool bsig_bam_csi_only(const bsig_bam *b) { return b->csi_only; }
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
        if (b->csi_only) return fail(BSIG_ERR_NOINDEX, "region queries need a .bai index (%s has a .csi index only: decode the whole file)", b->path.c_str());
        std::vector<bsig::Region> rg((size_t)n_regions);
        for (int64_t i = 0; i < n_regions; ++i) rg[(size_t)i] = bsig::Region{rid[i], beg[i], end[i]};
        rc = bsig::bam_decode_regions(b->path, b->idx, rg, threads, h, b->cols);
    }
    if (rc) return rc;
    fill_columns(b, cols);
    return BSIG_OK;
}

void bsig_last_call_timing(double *t6)
{
    for (int k = 0; k < 6; ++k) t6[k] = g_call_timing[k];
}

void bsig_bam_decode_timing(double *t6)
{
    for (int k = 0; k < 6; ++k) t6[k] = bsig::g_decode_timing[k];
}

int bsig_pileup_core(const char *bampath, int64_t n, const int32_t *seq_code, int32_t n_levels,
                     const char *const *levels, const int32_t *start, const int32_t *width,
                     const int32_t *strand, const int32_t *tlen_filter, int32_t n_tlen_filter,
                     int32_t mapqual, int32_t binsize, int32_t shift, int32_t ss, int32_t requiredF,
                     int32_t filteredF, int32_t pe_mid, int32_t maxgap, int32_t device, int32_t *out,
                     const int64_t *off)
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
    return file_level(bampath, n, seq_code, n_levels, levels, start, width, strand, p, device, out, off);
}

int bsig_coverage_core(const char *bampath, int64_t n, const int32_t *seq_code, int32_t n_levels,
                       const char *const *levels, const int32_t *start, const int32_t *width,
                       const int32_t *strand, const int32_t *tlen_filter, int32_t n_tlen_filter,
                       int32_t mapqual, int32_t requiredF, int32_t filteredF, int32_t tspan,
                       int32_t maxgap, int32_t device, int32_t *out, const int64_t *off)
{
    (void)maxgap;
    bsig_params p;
    memset(&p, 0, sizeof p);
    p.mode = BSIG_MODE_COVERAGE;
    p.mapqual = mapqual; p.binsize = 1; p.requiredF = requiredF; p.filteredF = filteredF; p.tspan = tspan;
    if (n_tlen_filter != 0 && n_tlen_filter != 2) return fail(BSIG_ERR_ARG, "tlen_filter must have 0 or 2 elements");
    p.n_tlen_filter = n_tlen_filter;
    for (int k = 0; k < n_tlen_filter; ++k) p.tlen_filter[k] = tlen_filter[k];
    return file_level(bampath, n, seq_code, n_levels, levels, start, width, strand, p, device, out, off);
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
    if (n > 0 && (!src_off || !dst_off || !which)) return fail(BSIG_ERR_ARG, "NULL argument");
    for (int64_t k = 0; k < n; ++k) {
        const int64_t len = src_off[k + 1] - src_off[k];
        const int64_t i = which[k];
        if (len < 0 || i < 0) return fail(BSIG_ERR_ARG, "bad segment %lld", (long long)k);
        if (len != dst_off[i + 1] - dst_off[i]) return fail(BSIG_ERR_ARG, "segment %lld does not fit its destination", (long long)k);
    }
    if (n <= 0) return BSIG_OK;
    auto copy_range = [&](int64_t k0, int64_t k1) {
        for (int64_t k = k0; k < k1; ++k) {
            const int64_t len = src_off[k + 1] - src_off[k];
            if (len) memcpy(dst + dst_off[which[k]], src + src_off[k], (size_t)len * sizeof(int32_t));
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

void bsig_cache_clear(void)
{
    {
        std::lock_guard<std::mutex> lock(g_cache.mu);
        g_cache.clear();
    }
    bsig::exchange_close_all();
    bsig::release_decode_scratch();
}

const char *bsig_last_call_route(void) { return g_call_route; }

}  // extern "C"
