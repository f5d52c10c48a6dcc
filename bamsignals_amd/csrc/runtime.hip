// Host runtime behind include/bamsignals_abi.h: device context, reads resident in HBM,
// plans (ranges + parameters -> work items) and their execution on a HIP stream.
//
// There is no CPU fallback in this library: every compute entry point needs a gfx950 device
// and fails with BSIG_ERR_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>
#include <string>
#include <vector>

#include "../../include/bamsignals_abi.h"
#include "bsig_types.h"
#include "host_util.h"
#include "kernels.h"

namespace bsig {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

}  // namespace bsig

using bsig::fail;

#include "runtime_internal.h"

// ---------------------------------------------------------------------------------------------
// the per-process cache of large free device blocks (runtime_internal.h)
// ---------------------------------------------------------------------------------------------
namespace bsig {
AllocMeter g_alloc_meter;
namespace {
struct BlockCache {
    struct Blk { int dev; void *p; size_t bytes; };
    std::mutex mu;
    std::vector<Blk> free_;
    size_t cached = 0;
} g_blocks;
size_t block_cache_limit()
{
    size_t gb = 48;
    if (const char *e = getenv("BAMSIGNALS_SCRATCH_CACHE_GB")) gb = (size_t)std::max(0ll, atoll(e));
    return gb << 30;
}
}  // namespace

// ---- an arena reserved with the context (env BAMSIGNALS_ARENA_GB, default 0 = none) -------------------------
// Every hipMalloc is a trip into the driver, and on a shared host those trips are where a call's time goes
// astray (the stalls above; 0.1-0.5 s extra in the walk and the layout of one cold call in four on a busy
// box).  With an arena ONE allocation is made when the context comes up; blocks are carved out of it (first
// fit, 4-KiB aligned, neighbours merged again when they come back) and a cold call makes no allocation at all
// as long as its scratch and its resident reads fit.  What does not fit takes the paths above.
namespace {
struct Arena {
    int dev = -1;
    uint8_t *base = nullptr;
    size_t size = 0;
    std::vector<std::pair<size_t, size_t>> free_;      // (offset, length), sorted by offset, never adjacent
};
std::mutex g_arena_mu;
std::vector<Arena> g_arenas;
Arena *arena_of(int device)
{
    for (Arena &a : g_arenas)
        if (a.dev == device) return &a;
    return nullptr;
}
bool arena_take(int device, size_t bytes, void **p, size_t *got)
{
    std::lock_guard<std::mutex> lk(g_arena_mu);
    Arena *a = arena_of(device);
    if (!a) return false;
    const size_t need = (bytes + 4095) & ~(size_t)4095;
    for (size_t k = 0; k < a->free_.size(); ++k) {
        if (a->free_[k].second < need) continue;
        *p = a->base + a->free_[k].first;
        *got = need;
        a->free_[k].first += need;
        a->free_[k].second -= need;
        if (a->free_[k].second == 0) a->free_.erase(a->free_.begin() + (long)k);
        return true;
    }
    return false;
}
bool arena_give(int device, void *p, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_arena_mu);
    Arena *a = arena_of(device);
    if (!a || (uint8_t *)p < a->base || (uint8_t *)p >= a->base + a->size) return false;
    const size_t off = (size_t)((uint8_t *)p - a->base);
    auto it = std::lower_bound(a->free_.begin(), a->free_.end(), std::make_pair(off, (size_t)0));
    it = a->free_.insert(it, {off, bytes});
    if (it + 1 != a->free_.end() && it->first + it->second == (it + 1)->first) { it->second += (it + 1)->second; a->free_.erase(it + 1); }
    if (it != a->free_.begin() && (it - 1)->first + (it - 1)->second == it->first) { (it - 1)->second += it->second; a->free_.erase(it); }
    return true;
}
}  // namespace

void arena_reserve(int device)
{
    const char *e = getenv("BAMSIGNALS_ARENA_GB");
    const long long gb = e ? atoll(e) : 0;
    if (gb <= 0) return;
    std::lock_guard<std::mutex> lk(g_arena_mu);
    if (arena_of(device)) return;
    Arena a;
    a.dev = device;
    a.size = (size_t)gb << 30;
    if (hipSetDevice(device) != hipSuccess || metered_malloc((void **)&a.base, a.size) != hipSuccess) { (void)hipGetLastError(); return; }
    a.free_.push_back({0, a.size});
    g_arenas.push_back(a);
}

// (bsig_cache_clear: an arena nobody holds a block of goes back to the driver)
static void arena_release_idle()
{
    std::lock_guard<std::mutex> lk(g_arena_mu);
    for (size_t k = 0; k < g_arenas.size();) {
        Arena &a = g_arenas[k];
        if (a.free_.size() == 1 && a.free_[0].first == 0 && a.free_[0].second == a.size) {
            (void)hipSetDevice(a.dev);
            (void)metered_free(a.base);
            g_arenas.erase(g_arenas.begin() + (long)k);
        } else {
            ++k;
        }
    }
}

hipError_t block_alloc(int device, size_t bytes, double max_waste, void **p, size_t *got)
{
    if (arena_take(device, std::max<size_t>(bytes, 256), p, got)) return hipSuccess;
    bytes = (std::max<size_t>(bytes, 256) + 255) & ~(size_t)255;
    if (bytes >= kBlockCacheMin) {
        std::lock_guard<std::mutex> lk(g_blocks.mu);
        size_t best = (size_t)-1;
        const size_t cap = (size_t)((double)bytes * max_waste) + (1u << 20);
        for (size_t k = 0; k < g_blocks.free_.size(); ++k) {
            const BlockCache::Blk &b = g_blocks.free_[k];
            if (b.dev != device || b.bytes < bytes || b.bytes > cap) continue;
            if (best == (size_t)-1 || b.bytes < g_blocks.free_[best].bytes) best = k;
        }
        if (best != (size_t)-1) {
            *p = g_blocks.free_[best].p;
            *got = g_blocks.free_[best].bytes;
            g_blocks.cached -= g_blocks.free_[best].bytes;
            g_blocks.free_.erase(g_blocks.free_.begin() + (long)best);
            return hipSuccess;
        }
    }
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = metered_malloc(p, bytes);
    if (e == hipErrorOutOfMemory) {
        // the cache itself may be what fills the device: hand it back and try once more
        (void)hipGetLastError();
        block_cache_release();
        (void)hipSetDevice(device);
        e = metered_malloc(p, bytes);
    }
    if (e != hipSuccess) { *p = nullptr; return e; }
    *got = bytes;
    return hipSuccess;
}

void block_free(int device, void *p, size_t bytes)
{
    if (!p) return;
    if (arena_give(device, p, bytes)) return;
    if (bytes < kBlockCacheMin) {                     // small blocks are not worth keeping
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        (void)metered_free(p);
        if (cur != device) (void)hipSetDevice(cur);
        return;
    }
    const size_t limit = block_cache_limit();          // per device: every GPU has its own HBM
    std::vector<BlockCache::Blk> drop;
    {
        std::lock_guard<std::mutex> lk(g_blocks.mu);
        g_blocks.free_.push_back(BlockCache::Blk{device, p, bytes});
        g_blocks.cached += bytes;
        for (;;) {
            size_t on_dev = 0, big = (size_t)-1;
            for (size_t k = 0; k < g_blocks.free_.size(); ++k) {
                if (g_blocks.free_[k].dev != device) continue;
                on_dev += g_blocks.free_[k].bytes;
                if (big == (size_t)-1 || g_blocks.free_[k].bytes > g_blocks.free_[big].bytes) big = k;
            }
            if (on_dev <= limit || big == (size_t)-1) break;
            drop.push_back(g_blocks.free_[big]);
            g_blocks.cached -= g_blocks.free_[big].bytes;
            g_blocks.free_.erase(g_blocks.free_.begin() + (long)big);
        }
    }
    if (drop.empty()) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (const BlockCache::Blk &b : drop) { (void)hipSetDevice(b.dev); (void)metered_free(b.p); }
    (void)hipSetDevice(cur);
}

void block_cache_release()
{
    arena_release_idle();
    std::vector<BlockCache::Blk> all;
    {
        std::lock_guard<std::mutex> lk(g_blocks.mu);
        all.swap(g_blocks.free_);
        g_blocks.cached = 0;
    }
    if (all.empty()) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (const BlockCache::Blk &b : all) { (void)hipSetDevice(b.dev); (void)metered_free(b.p); }
    (void)hipSetDevice(cur);
}
}  // namespace bsig

struct bsig_plan {
    bsig_ctx *ctx = nullptr;
    const bsig_reads *reads = nullptr;
    int mode = 0;                  // what the caller asked for (fixes the result layout)
    int kernel_mode = 0;           // which kernel family runs: profile with very wide bins is
                                   // executed as a bamCount over the bins' sub-intervals
    BsigKParams kp{};
    int tile_cells = 0, threads = 0;
    int64_t n_ranges = 0, n_items = 0;
    std::vector<int64_t> off;
    DevPool pool;
    BsigWorkItem *items = nullptr;
    int32_t *d_out = nullptr;      // device result buffer of bsig_plan_run_host, kept between calls
    // slices of heavy tiles (tiles whose read windows hold more than kHeavyReads reads): the same
    // kernels run a second time over these items with fixed windows and accumulate = 1
    BsigWorkItem *heavy_items = nullptr;
    void *heavy_windows = nullptr;
    int64_t n_heavy_slices = 0, n_heavy_tiles = 0;
    // count family only: some cells are reached by global atomic adds (sub-intervals of a range wider than
    // one workgroup's share, wide bins) or by no work item at all (zero-width ranges), so the result must
    // be zeroed in front of the launch.  Ranges of one tile store their counters themselves.
    bool needs_zero = false;
    bool have_stats = false;
    bsig_plan_stats stats{};
    uint8_t *ptab = nullptr;            // the packed class's filter table for kp (BsigKParams::ptab)
    BsigResolved *resolved = nullptr;   // large launches: the windows of every tile, written by k_resolve_tiles
    uint64_t resolved_gen = 0;          // ... for this layout of the reads (0: not yet): a later run on the same layout reuses them
    uint64_t made_for_gen = 0;          // the layout the plan was made on: its tiles' heavy slices and the packed class's filter
                                        // table are read off that layout, so a plan does not outlive it
    int64_t runs = 0;                   // runs so far (a plan that is run AGAIN is a resident one: plan_two_launches)
};
static int64_t g_resolve_min_override = -1;     // bsig_debug_set_knob(4, n): two launches from n tiles on (sweeps)
// does a run of this plan look its windows up in a launch of its own?  (measured at the north star's read density,
// scripts/ns_variants.py: 15,000 tiles 28.3 us fused / 31.1 us in two launches, 25,000 43.3 / 46.2, 35,000 71.8 / 66.3,
// 50,000 112.0 / 107.3, 100,000 189.7 / 177.1; config 5's share 180.6 / 157.5, config 4's call 407 / 370:
// env BAMSIGNALS_RESOLVE_MIN_TILES, default 32,768)
// are a large launch's tile windows kept with the plan after its first run (default), or looked up in every run?
static bool windows_kept()
{
    const char *e = getenv("BAMSIGNALS_CACHE_WINDOWS");      // (read per run: a test flips it)
    return !(e && !strcmp(e, "0"));
}

static bool plan_two_launches(const bsig_plan *p)
{
    static const int64_t resolve_min = getenv("BAMSIGNALS_RESOLVE_MIN_TILES") ? atoll(getenv("BAMSIGNALS_RESOLVE_MIN_TILES")) : (int64_t)32768;
    // (the count family walks four tiles per wave and looks their windows up side by side: a launch of its own
    // for them measured the same or slower there -- 0.1469 fused, 0.1476 in two launches on config 3's tiling --
    // so bamCount keeps the fused form for a step that pays the lookup; with the windows kept it takes the other
    // from its second run on like everybody: 0.0949 / 0.0975 -> 0.0925 / 0.0945 ms on that tiling)
    if (g_resolve_min_override >= 0) return p->n_items > 0 && p->n_items >= g_resolve_min_override;
    if (p->n_items >= resolve_min && p->kernel_mode != BSIG_MODE_COUNT) return p->n_items > 0;
    // Those figures are for a step that pays the lookup launch.  A plan that is run a second time is a resident one, and
    // with the windows kept its later steps pay nothing for them: from its second run on a plan takes the form for
    // resolved windows whatever its size -- a workgroup's life is one dependent memory trip shorter -- (10,000 tiles: 19.56
    // -> 18.30 us a step; 4,000: 10.08 -> 9.37; 1,500: 6.76 -> 5.57; 400: 5.58 -> 4.62; 100: 5.29 -> 4.35; its second
    // run carries the lookup launch, a plan that is run once -- every file-level call -- never sees it).
    static const int64_t again_min = getenv("BAMSIGNALS_RESOLVE_AGAIN_MIN_TILES") ? atoll(getenv("BAMSIGNALS_RESOLVE_AGAIN_MIN_TILES")) : (int64_t)1;
    return windows_kept() && p->runs >= 1 && p->n_items >= again_min;
}

extern "C" {

int bsig_abi_version(void) { return BSIG_ABI_VERSION; }

const char *bsig_last_error(void) { return bsig::g_last_error.c_str(); }

// allocateList's shapes (ref: src/bamsignals.cpp:139-192)
int64_t bsig_layout(int64_t n, const int32_t *len, int32_t binsize, int32_t ss, int64_t *off)
{
    const int64_t mult = ss ? 2 : 1;
    int64_t acc = 0;
    for (int64_t i = 0; i < n; ++i) {
        off[i] = acc;
        if (binsize <= 0) acc += mult;
        else if (len[i] > 0) acc += mult * (((int64_t)len[i] + binsize - 1) / binsize);
    }
    off[n] = acc;
    return acc;
}

int bsig_device_count(int32_t *n)
{
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(BSIG_ERR_DEVICE, "no HIP device: %s", hipGetErrorString(e)); }
    *n = c;
    return BSIG_OK;
}

int bsig_ctx_create(int32_t device, void *stream, bsig_ctx **out)
{
    if (!out) return fail(BSIG_ERR_ARG, "ctx output pointer is NULL");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(BSIG_ERR_DEVICE, "no HIP device available (bamsignals_hip has no CPU fallback)");
    if (device < 0 || device >= count) return fail(BSIG_ERR_ARG, "device %d out of range [0,%d)", device, count);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(BSIG_ERR_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    {
        // the first hipMalloc of a process sets the runtime's allocator up (76 ms measured): that belongs to
        // bringing the device up, not to whichever call happens to allocate first
        void *warm = nullptr;
        if (hipMalloc(&warm, 256) == hipSuccess) (void)hipFree(warm);
        (void)hipGetLastError();
    }
    bsig::arena_reserve(device);
    bsig_ctx *c = new bsig_ctx;
    {
        static std::mutex warm_mu;
        static std::vector<int> warmed;
        std::lock_guard<std::mutex> lk(warm_mu);
        if (std::find(warmed.begin(), warmed.end(), device) == warmed.end()) { warmed.push_back(device); c->warm_pending = true; }
    }
    c->device = device;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(BSIG_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); }
        c->owns_stream = true;
    }
    if (c->warm_pending) {
        // ... and so does the first launch out of each code object of this library (walk / extract / inflate,
        // pileup, reassembly): about 10 ms apiece, once per process and device
        (void)bsig::warm_pileup_module(c->stream);
        (void)bsig::warm_decode_module(c->stream);
        (void)bsig::warm_collect_module(c->stream);
        // ... and so do the runtime's own staging buffers for copies from and to pageable memory (the record
        // walk's per-block summaries come back that way)
        {
            void *d = nullptr;
            std::vector<uint8_t> h((size_t)4 << 20, 0);
            if (hipMalloc(&d, h.size()) == hipSuccess) {
                (void)hipMemcpyAsync(d, h.data(), h.size(), hipMemcpyHostToDevice, c->stream);
                (void)hipMemcpyAsync(h.data(), d, h.size(), hipMemcpyDeviceToHost, c->stream);
                (void)hipStreamSynchronize(c->stream);
                (void)hipFree(d);
            }
        }
        (void)hipStreamSynchronize(c->stream);
        (void)hipGetLastError();
        c->warm_pending = false;
    }
    *out = c;
    return BSIG_OK;
}

void bsig_ctx_destroy(bsig_ctx *c)
{
    if (!c) return;
    if (c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int bsig_ctx_sync(bsig_ctx *c)
{
    if (!c) return fail(BSIG_ERR_ARG, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BSIG_OK;
}

void *bsig_ctx_stream(bsig_ctx *c) { return c ? (void *)c->stream : nullptr; }

int bsig_host_alloc(int64_t bytes, void **ptr)
{
    if (!ptr || bytes < 0) return fail(BSIG_ERR_ARG, "bad argument to bsig_host_alloc");
    *ptr = nullptr;
    HIP_TRY(hipHostMalloc(ptr, (size_t)std::max<int64_t>(bytes, 16), hipHostMallocDefault));
    return BSIG_OK;
}

void bsig_host_free(void *ptr)
{
    if (ptr) (void)hipHostFree(ptr);
}

// ---------------------------------------------------------------------------------------------
// reads -> HBM
// ---------------------------------------------------------------------------------------------
}  // extern "C"

// The resident layout from device columns; shared by bsig_reads_upload (host columns) and the
// device-side BAM decode (devdecode.hip).
uint64_t bsig::next_layout_gen()
{
    static std::atomic<uint64_t> g{0};
    return ++g;
}

int bsig::layout_from_device(bsig_ctx *ctx, bsig_reads *R, int64_t n, int32_t n_ref, const int32_t *ref_len,
                             const int64_t *ref_off, const int32_t *d_pos, const int32_t *d_end,
                             const uint16_t *d_flag, const uint8_t *d_mapq, const int32_t *d_tlen)
{
    hipStream_t st = ctx->stream;
    HIP_TRY(hipSetDevice(ctx->device));
    R->layout_gen = next_layout_gen();          // (whatever a plan cached for an earlier layout of R is stale from here on)

    // global coordinate: references back to back in 64-kbp units
    R->n_ref = n_ref;
    R->ref_unit0.resize(n_ref);
    R->ref_units.resize(n_ref);
    R->ref_len.assign(ref_len, ref_len + n_ref);
    uint64_t total_units = 0;
    for (int r = 0; r < n_ref; ++r) {
        if (ref_len[r] < 0) return fail(BSIG_ERR_ARG, "negative length of reference %d", r);
        const uint64_t u = ((uint64_t)ref_len[r] >> BSIG_REF_UNIT_SHIFT) + 1;
        if (total_units + u >= (1ull << 31)) return fail(BSIG_ERR_ARG, "genome too large for the bucket index");
        R->ref_unit0[r] = (uint32_t)total_units;
        R->ref_units[r] = (uint32_t)u;
        total_units += u;
    }
    const uint64_t total_bp = total_units << BSIG_REF_UNIT_SHIFT;

    R->info = bsig_reads_info{};
    R->info.n_reads = n;
    R->dev.fmtab = nullptr;
    R->dev.n_codes = 0;
    if (n == 0 || n_ref == 0) {
        for (int c = 0; c < BSIG_MAX_CLASSES; ++c) R->dev.cls[c] = BsigClassCols{};
        return BSIG_OK;
    }
    DevPool tmp(1e9);          // scratch of this call: any cached block that is large enough will do
    const bool diag = getenv("BSIG_DIAG_DECODE") != nullptr;
    const auto t_diag0 = std::chrono::steady_clock::now();
    auto diag_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_diag0).count(); };
    int64_t *d_ref_off;
    uint32_t *d_unit0, *d_units;
    HIP_TRY(tmp.alloc(&d_ref_off, n_ref + 1));
    HIP_TRY(tmp.alloc(&d_unit0, n_ref));
    HIP_TRY(tmp.alloc(&d_units, n_ref));
    HIP_TRY(hipMemcpyAsync(d_ref_off, ref_off, (n_ref + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_unit0, R->ref_unit0.data(), n_ref * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_units, R->ref_units.data(), n_ref * sizeof(uint32_t), hipMemcpyHostToDevice, st));

    // ---- the file's pair table (bsig_types.h): the BSIG_PACK_CODES most frequent (flag, mapq) pairs among a
    // sample of the short reads, most frequent first, ties by key -- the same table on every GPU that lays
    // out the same columns.  env BAMSIGNALS_PACK=0: no packed class (testing; every short read in class 0).
    uint16_t *d_codemap = nullptr;
    if (!(getenv("BAMSIGNALS_PACK") && !strcmp(getenv("BAMSIGNALS_PACK"), "0"))) {
        constexpr uint32_t kKeys = 1u << 20, kPairCap = 16384;
        // one block for the counters, the pair list, its length and the code map: 4 MB + 128 KB + 2 MB
        uint8_t *blk;
        HIP_TRY(tmp.alloc(&blk, (size_t)kKeys * 4 + kPairCap * 8 + 256 + (size_t)kKeys * 2));
        uint32_t *d_hist = (uint32_t *)blk;
        uint2 *d_pairs = (uint2 *)(blk + (size_t)kKeys * 4);
        uint32_t *d_npairs = (uint32_t *)(blk + (size_t)kKeys * 4 + kPairCap * 8);
        d_codemap = (uint16_t *)(blk + (size_t)kKeys * 4 + kPairCap * 8 + 256);
        HIP_TRY(hipMemsetAsync(d_hist, 0, (size_t)kKeys * 4 + kPairCap * 8 + 256, st));
        HIP_TRY(hipMemsetAsync(d_codemap, 0xFF, (size_t)kKeys * 2, st));
        HIP_TRY(bsig::launch_pair_sample(n, d_pos, d_end, d_flag, d_mapq, d_hist, d_pairs, kPairCap, d_npairs, st));
        uint32_t n_pairs = 0;
        HIP_TRY(hipMemcpyAsync(&n_pairs, d_npairs, sizeof n_pairs, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<uint2> pairs;
        if (n_pairs <= kPairCap) {
            pairs.resize(n_pairs);
            if (n_pairs) HIP_TRY(hipMemcpyAsync(pairs.data(), d_pairs, (size_t)n_pairs * sizeof(uint2), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        } else {
            // more distinct pairs than the list holds (which of them it caught depends on timing): read the counters
            std::vector<uint32_t> hist(kKeys);
            HIP_TRY(hipMemcpyAsync(hist.data(), d_hist, (size_t)kKeys * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            for (uint32_t k = 0; k < kKeys; ++k)
                if (hist[k]) pairs.push_back(make_uint2(k, hist[k]));
        }
        std::sort(pairs.begin(), pairs.end(), [](const uint2 &x, const uint2 &y) { return x.y != y.y ? x.y > y.y : x.x < y.x; });
        const int n_codes = (int)std::min<size_t>(pairs.size(), BSIG_PACK_CODES);
        if (n_codes > 0) {
            std::vector<uint32_t> fmtab(BSIG_PACK_CODES, 0u);
            for (int c = 0; c < n_codes; ++c) fmtab[(size_t)c] = (pairs[(size_t)c].x & 0xFFFu) | (pairs[(size_t)c].x >> 12) << 16;
            uint32_t *d_fmtab;
            HIP_TRY(R->pool.alloc(&d_fmtab, BSIG_PACK_CODES));
            HIP_TRY(hipMemcpyAsync(d_fmtab, fmtab.data(), BSIG_PACK_CODES * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            HIP_TRY(bsig::launch_codemap_fill(d_fmtab, n_codes, d_codemap, st));
            HIP_TRY(hipStreamSynchronize(st));          // (fmtab is a local)
            R->dev.fmtab = d_fmtab;
            R->dev.n_codes = n_codes;
            R->fmtab = fmtab;
        } else {
            d_codemap = nullptr;
        }
        if (diag) fprintf(stderr, "  [layout] pair table: %u pairs in the sample, %d codes, %.1f ms\n", n_pairs, n_codes, diag_ms());
    }

    // classes: per-chunk counts -> host exclusive scan
    const int64_t n_chunks = bsig::prep_chunks(n);
    uint32_t *d_counts;
    int32_t *d_maxspan;
    HIP_TRY(tmp.alloc(&d_counts, n_chunks * BSIG_MAX_CLASSES));
    HIP_TRY(tmp.alloc(&d_maxspan, BSIG_MAX_CLASSES + 1));
    HIP_TRY(hipMemsetAsync(d_maxspan, 0, (BSIG_MAX_CLASSES + 1) * sizeof(int32_t), st));
    HIP_TRY(bsig::launch_span_hist(n, n_ref, d_ref_off, d_units, d_pos, d_end, d_flag, d_mapq, d_codemap, d_counts, d_maxspan, st));
    // (the scan of the chunk counts stays on the device: only the class totals and the longest spans come back)
    uint64_t *d_base, *d_totals;
    HIP_TRY(tmp.alloc(&d_base, (size_t)n_chunks * BSIG_MAX_CLASSES));
    HIP_TRY(tmp.alloc(&d_totals, BSIG_MAX_CLASSES));
    HIP_TRY(bsig::launch_chunk_scan(n_chunks, d_counts, d_base, d_totals, st));
    int32_t maxspan[BSIG_MAX_CLASSES + 1];
    uint64_t class_n[BSIG_MAX_CLASSES] = {};
    HIP_TRY(hipMemcpyAsync(class_n, d_totals, sizeof class_n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(maxspan, d_maxspan, sizeof maxspan, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (maxspan[BSIG_MAX_CLASSES])
        return fail(BSIG_ERR_ARG, "reads must be sorted by position inside every reference (coordinate-sorted BAM order)");
    if (diag) fprintf(stderr, "  [layout] class counts %.1f ms\n", diag_ms());

    // bucket width per class: about 16 reads per bucket, 16 bp .. 64 kbp (the packed class: at most one
    // chunk of its position bits)
    bsig::ScatterPtrs S{};
    uint64_t n_buckets[BSIG_MAX_CLASSES] = {};
    int min_shift = 4;
    while ((total_bp >> min_shift) >= (1ull << 32)) ++min_shift;
    if (min_shift > BSIG_PACK_POS_BITS) return fail(BSIG_ERR_ARG, "genome too large for the bucket index");
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c) {
        BsigClassCols &C = R->dev.cls[c];
        C = BsigClassCols{};
        if (class_n[c] == 0) continue;
        if (class_n[c] >= (1ull << 32) - 8) return fail(BSIG_ERR_ARG, "more than 2^32 reads in one span class");
        double per_bucket = 16.0;
        if (const char *v = getenv("BAMSIGNALS_BUCKET_READS")) per_bucket = std::max(1.0, atof(v));
        const double bp_per_bucket = per_bucket * (double)total_bp / (double)class_n[c];
        const int max_shift = c == BSIG_CLASS_PACKED ? BSIG_PACK_POS_BITS : BSIG_REF_UNIT_SHIFT;
        int k = 4;
        while (k < max_shift && (double)(1ull << (k + 1)) <= bp_per_bucket) ++k;
        k = std::max(k, min_shift);
        if (k > max_shift) return fail(BSIG_ERR_ARG, "genome too large for the bucket index");
        const size_t cap = ((size_t)class_n[c] + 3) / 4 * 4 + 4;
        int32_t *p = nullptr, *e = nullptr, *t;
        uint32_t *f, *gb, *idx;
        if (c != BSIG_CLASS_PACKED) HIP_TRY(R->pool.alloc(&p, cap));      // a packed word carries its position
        if (c == 2 || c == 3) HIP_TRY(R->pool.alloc(&e, cap));           // classes 0, 1 and packed keep their span in fm
        HIP_TRY(R->pool.alloc(&f, cap));
        HIP_TRY(R->pool.alloc(&t, cap));
        HIP_TRY(tmp.alloc(&gb, cap));
        n_buckets[c] = total_bp >> k;
        HIP_TRY(R->pool.alloc(&idx, n_buckets[c] + 2));
        R->col_cap[c] = cap;
        R->idx_entries[c] = n_buckets[c] + 2;
        // the tail padding is read by the 16-B loads: keep it defined
        if (p) HIP_TRY(hipMemsetAsync(p + cap - 8, 0, 8 * sizeof(int32_t), st));
        if (e) HIP_TRY(hipMemsetAsync(e + cap - 8, 0, 8 * sizeof(int32_t), st));
        HIP_TRY(hipMemsetAsync(f + cap - 8, 0, 8 * sizeof(int32_t), st));
        HIP_TRY(hipMemsetAsync(t + cap - 8, 0, 8 * sizeof(int32_t), st));
        C.pos = p; C.end = e; C.fm = f; C.tlen = t; C.idx = idx;
        C.n = (int64_t)class_n[c]; C.maxspan = maxspan[c]; C.kshift = k;
        S.pos[c] = p; S.end[c] = e; S.fm[c] = f; S.tlen[c] = t; S.gb[c] = gb; S.kshift[c] = k;
        R->info.class_n[c] = C.n;
        R->info.class_maxspan[c] = C.maxspan;
        R->info.class_bucket_shift[c] = k;
        R->info.n_classes += 1;
    }
    R->info.n_codes = R->dev.n_codes;

    if (diag) fprintf(stderr, "  [layout] column + index allocations %.1f ms\n", diag_ms());
    HIP_TRY(bsig::launch_scatter(n, n_ref, d_ref_off, d_unit0, d_units, d_pos, d_end, d_flag, d_mapq, d_tlen, d_codemap,
                                 d_base, S, st));
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c)
        if (class_n[c])
            HIP_TRY(bsig::launch_build_idx((int64_t)class_n[c], S.gb[c], n_buckets[c],
                                           const_cast<uint32_t *>(R->dev.cls[c].idx), st));
    HIP_TRY(hipStreamSynchronize(st));
    if (diag) fprintf(stderr, "  [layout] scatter + indexes %.1f ms\n", diag_ms());
    R->info.hbm_bytes = R->pool.footprint();
    return BSIG_OK;
}

extern "C" {

static int upload_impl(bsig_ctx *ctx, const bsig_columns *cols, bsig_reads *R)
{
    const int64_t n = cols->n_reads;
    hipStream_t st = ctx->stream;
    HIP_TRY(hipSetDevice(ctx->device));
    for (int r = 0; r < cols->n_ref; ++r)
        if (cols->ref_len[r] < 0) return fail(BSIG_ERR_ARG, "negative length of reference %d", r);
    if (n == 0 || cols->n_ref == 0)
        return bsig::layout_from_device(ctx, R, n, cols->n_ref, cols->ref_len, cols->ref_off, nullptr, nullptr, nullptr,
                                        nullptr, nullptr);
    DevPool tmp;
    int32_t *d_pos, *d_end, *d_tlen;
    uint16_t *d_flag;
    uint8_t *d_mapq;
    HIP_TRY(tmp.alloc(&d_pos, n));
    HIP_TRY(tmp.alloc(&d_end, n));
    HIP_TRY(tmp.alloc(&d_tlen, n));
    HIP_TRY(tmp.alloc(&d_flag, n));
    HIP_TRY(tmp.alloc(&d_mapq, n));
    HIP_TRY(hipMemcpyAsync(d_pos, cols->pos, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_tlen, cols->tlen, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_flag, cols->flag, n * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_mapq, cols->mapq, n * sizeof(uint8_t), hipMemcpyHostToDevice, st));
    if (cols->end) {
        HIP_TRY(hipMemcpyAsync(d_end, cols->end, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    } else {
        const int64_t n_ops = cols->cigar_off[n];
        if (cols->cigar_off[0] != 0 || n_ops < 0) return fail(BSIG_ERR_ARG, "cigar_off must start at 0 and be non-decreasing");
        int64_t *d_coff;
        uint32_t *d_cig;
        HIP_TRY(tmp.alloc(&d_coff, n + 1));
        HIP_TRY(tmp.alloc(&d_cig, n_ops));
        HIP_TRY(hipMemcpyAsync(d_coff, cols->cigar_off, (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
        if (n_ops) HIP_TRY(hipMemcpyAsync(d_cig, cols->cigar, n_ops * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(bsig::launch_cigar_end(n, d_pos, d_flag, d_coff, d_cig, d_end, st));
    }
    // layout_from_device synchronises the stream before it returns: tmp may be released then
    return bsig::layout_from_device(ctx, R, n, cols->n_ref, cols->ref_len, cols->ref_off, d_pos, d_end, d_flag, d_mapq,
                                    d_tlen);
}

int bsig_reads_upload(bsig_ctx *ctx, const bsig_columns *cols, bsig_reads **out)
{
    if (!ctx || !cols || !out) return fail(BSIG_ERR_ARG, "NULL argument to bsig_reads_upload");
    *out = nullptr;
    const int64_t n = cols->n_reads;
    if (n < 0 || cols->n_ref < 0) return fail(BSIG_ERR_ARG, "negative read or reference count");
    if (cols->n_ref > 0 && (!cols->ref_len || !cols->ref_off)) return fail(BSIG_ERR_ARG, "ref_len/ref_off missing");
    if (n > 0) {
        if (!cols->pos || !cols->flag || !cols->mapq || !cols->tlen) return fail(BSIG_ERR_ARG, "read columns missing");
        if (!cols->end && (!cols->cigar_off || !cols->cigar)) return fail(BSIG_ERR_ARG, "need either end or cigar_off+cigar");
        if (cols->n_ref == 0) return fail(BSIG_ERR_ARG, "reads without references");
    }
    if (cols->n_ref > 0) {
        if (cols->ref_off[0] != 0 || cols->ref_off[cols->n_ref] != n)
            return fail(BSIG_ERR_ARG, "ref_off must run from 0 to n_reads");
        for (int r = 0; r < cols->n_ref; ++r)
            if (cols->ref_off[r] > cols->ref_off[r + 1]) return fail(BSIG_ERR_ARG, "ref_off must be non-decreasing");
    }
    bsig_reads *R = new bsig_reads;
    R->ctx = ctx;
    const int rc = upload_impl(ctx, cols, R);
    if (rc != BSIG_OK) { delete R; return rc; }
    *out = R;
    return BSIG_OK;
}

int bsig_reads_get_info(const bsig_reads *reads, bsig_reads_info *info)
{
    if (!reads || !info) return fail(BSIG_ERR_ARG, "NULL argument");
    *info = reads->info;
    return BSIG_OK;
}

void bsig_reads_free(bsig_reads *reads) { delete reads; }

int bsig_reads_clone(const bsig_reads *src, bsig_ctx *dst_ctx, bsig_reads **out)
{
    if (!src || !dst_ctx || !out) return fail(BSIG_ERR_ARG, "NULL argument to bsig_reads_clone");
    *out = nullptr;
    const int sdev = src->ctx->device, ddev = dst_ctx->device;
    HIP_TRY(hipSetDevice(ddev));
    if (sdev != ddev) {
        // direct copies over xGMI where the link allows it; without peer access the runtime stages
        // the copy through the host, which is still correct
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, ddev, sdev) == hipSuccess && can) {
            (void)hipDeviceEnablePeerAccess(sdev, 0);     // "already enabled" is fine
            (void)hipGetLastError();
        }
    }
    bsig_reads *R = new bsig_reads;
    R->ctx = dst_ctx;
    R->info = src->info;
    R->n_ref = src->n_ref;
    R->ref_unit0 = src->ref_unit0;
    R->ref_units = src->ref_units;
    R->ref_len = src->ref_len;
    R->fmtab = src->fmtab;
    R->dev.fmtab = nullptr;
    R->dev.n_codes = src->dev.n_codes;
    hipStream_t st = dst_ctx->stream;
    hipError_t e = hipSuccess;
    auto copy = [&](const void *from, size_t bytes, void **to) {
        if (e != hipSuccess || !from) { *to = nullptr; return; }
        uint8_t *q = nullptr;
        e = R->pool.alloc(&q, bytes);
        if (e == hipSuccess) e = hipMemcpyPeerAsync(q, ddev, from, sdev, bytes, st);
        *to = q;
    };
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c) {
        const BsigClassCols &S = src->dev.cls[c];
        BsigClassCols &D = R->dev.cls[c];
        D = S;
        if (S.n == 0) continue;
        R->col_cap[c] = src->col_cap[c];
        R->idx_entries[c] = src->idx_entries[c];
        const size_t cb = (size_t)src->col_cap[c] * sizeof(int32_t);
        void *p;
        copy(S.pos, cb, &p); D.pos = (const int32_t *)p;
        copy(S.end, cb, &p); D.end = (const int32_t *)p;
        copy(S.fm, cb, &p); D.fm = (const uint32_t *)p;
        copy(S.tlen, cb, &p); D.tlen = (const int32_t *)p;
        copy(S.idx, (size_t)src->idx_entries[c] * sizeof(uint32_t), &p); D.idx = (const uint32_t *)p;
    }
    {
        void *p;
        copy(src->dev.fmtab, BSIG_PACK_CODES * sizeof(uint32_t), &p);
        R->dev.fmtab = (const uint32_t *)p;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        delete R;
        return fail(e == hipErrorOutOfMemory ? BSIG_ERR_NOMEM : BSIG_ERR_DEVICE, "copying the reads to GPU %d failed: %s", ddev,
                    hipGetErrorString(e));
    }
    R->info.hbm_bytes = R->pool.footprint();
    *out = R;
    return BSIG_OK;
}

}  // extern "C"

namespace {

// Result download into PAGEABLE host memory (an R vector, a numpy array): the runtime's own
// pageable path stages through one buffer and is first-touch bound on the destination pages
// (80 MB in 8 ms).  Here the result crosses PCIe by DMA into two page-locked halves and a few
// threads move each half on into the destination while the next one is in flight.
struct DownloadStage {
    std::mutex mu;
    hipEvent_t ev[2] = {nullptr, nullptr};
    static constexpr size_t kHalf = 32u << 20;          // (the halves are the device's PinnedPair)
    int ensure()
    {
        for (int k = 0; k < 2; ++k)
            if (!ev[k]) HIP_TRY(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        return BSIG_OK;
    }
};
// one per GPU: the events belong to the device they were created on
DownloadStage &download_for(int device)
{
    static std::mutex mu;
    static std::vector<std::pair<int, DownloadStage *>> all;
    std::lock_guard<std::mutex> lk(mu);
    for (auto &kv : all)
        if (kv.first == device) return *kv.second;
    all.emplace_back(device, new DownloadStage);
    return *all.back().second;
}

}  // namespace
int bsig::PinnedPair::ensure(size_t bytes)
{
    if (cap >= bytes) return BSIG_OK;
    for (int k = 0; k < 2; ++k) {
        if (buf[k]) (void)bsig::metered_host_free(buf[k]);
        buf[k] = nullptr;
    }
    cap = 0;
    for (int k = 0; k < 2; ++k) HIP_TRY(bsig::metered_host_malloc((void **)&buf[k], bytes));
    cap = bytes;
    return BSIG_OK;
}
bsig::PinnedPair &bsig::pinned_pair_for(int device)
{
    static std::mutex mu;
    static std::vector<std::pair<int, PinnedPair *>> all;
    std::lock_guard<std::mutex> lk(mu);
    for (auto &kv : all)
        if (kv.first == device) return *kv.second;
    all.emplace_back(device, new PinnedPair);
    return *all.back().second;
}
namespace {
bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// One or more slices of device memory to host memory through the two page-locked halves: the slices are laid
// back to back into chunks of one half each (several DMA copies per chunk where slices are short), and while one
// half crosses PCIe a few threads move the other on -- sink(first cell of the flat result, source, cells) is
// called by several threads at once, on disjoint pieces.
struct DlSlice { const uint8_t *src; size_t bytes; int64_t dst_c0; };      // bytes: a multiple of 4
typedef std::function<void(int64_t, const int32_t *, int64_t)> DownloadSink;
int download_staged(int device, hipStream_t st, const std::vector<DlSlice> &slices, const DownloadSink &sink, int copy_threads = 0)
{
    DownloadStage &g_download = download_for(device);
    std::lock_guard<std::mutex> lock(g_download.mu);
    HIP_TRY(hipSetDevice(device));
    int rc = g_download.ensure();
    if (rc) return rc;
    bsig::PinnedPair &pair = bsig::pinned_pair_for(device);
    std::lock_guard<std::mutex> pair_lock(pair.mu);
    rc = pair.ensure(DownloadStage::kHalf);
    if (rc) return rc;
    const size_t half = DownloadStage::kHalf;
    struct Part { size_t at; const uint8_t *src; size_t len; int64_t dst_c0; };      // `at`: offset in the half
    struct Chunk { size_t p0, p1, len; };                                              // parts [p0, p1)
    std::vector<Part> parts;
    std::vector<Chunk> chunks;
    {
        size_t fill = 0, first = 0;
        for (const DlSlice &sl : slices) {
            size_t done_b = 0;
            while (done_b < sl.bytes) {
                if (fill == half) { chunks.push_back(Chunk{first, parts.size(), fill}); first = parts.size(); fill = 0; }
                const size_t take = std::min(sl.bytes - done_b, half - fill);
                parts.push_back(Part{fill, sl.src + done_b, take, sl.dst_c0 + (int64_t)(done_b / 4)});
                fill += take;
                done_b += take;
            }
        }
        if (fill) chunks.push_back(Chunk{first, parts.size(), fill});
    }
    const size_t n_chunks = chunks.size();
    if (n_chunks == 0) return BSIG_OK;
    int n_thr = 8;
    if (const char *e = getenv("BAMSIGNALS_COPY_THREADS")) n_thr = std::max(1, std::min(64, atoi(e)));
    if (copy_threads > 0) n_thr = copy_threads;
    std::atomic<int64_t> ready(-1);                 // chunks 0..ready are in their half
    std::vector<std::atomic<int>> done(n_chunks);   // workers finished with chunk c
    for (auto &d : done) d.store(0);
    std::atomic<bool> abort(false);
    // thread t's share of chunk c: bytes [len * t / n, len * (t + 1) / n) of the half, cut at 4-byte cells
    auto share = [&](size_t c, int t) {
        const Chunk &C = chunks[c];
        const size_t a = (C.len * (size_t)t / (size_t)n_thr) & ~(size_t)3;
        const size_t b = t + 1 == n_thr ? C.len : (C.len * (size_t)(t + 1) / (size_t)n_thr) & ~(size_t)3;
        const uint8_t *buf = pair.buf[c & 1];
        for (size_t p = C.p0; p < C.p1 && a < b; ++p) {
            const Part &P = parts[p];
            const size_t lo = std::max(a, P.at), hi = std::min(b, P.at + P.len);
            if (lo < hi) sink(P.dst_c0 + (int64_t)((lo - P.at) / 4), (const int32_t *)(buf + lo), (int64_t)((hi - lo) / 4));
        }
    };
    auto worker = [&](int t) {
        for (size_t c = 0; c < n_chunks; ++c) {
            while (ready.load(std::memory_order_acquire) < (int64_t)c) {
                if (abort.load()) return;
                std::this_thread::yield();
            }
            share(c, t);
            done[c].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_thr; ++t) th.emplace_back(worker, t);
    hipError_t e = hipSuccess;
    auto issue = [&](size_t c) {
        const Chunk &C = chunks[c];
        for (size_t p = C.p0; p < C.p1 && e == hipSuccess; ++p)
            e = hipMemcpyAsync(pair.buf[c & 1] + parts[p].at, parts[p].src, parts[p].len, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipEventRecord(g_download.ev[c & 1], st);
    };
    for (size_t c = 0; c < std::min<size_t>(2, n_chunks) && e == hipSuccess; ++c) issue(c);
    for (size_t c = 0; c < n_chunks && e == hipSuccess; ++c) {
        e = hipEventSynchronize(g_download.ev[c & 1]);
        if (e != hipSuccess) break;
        ready.store((int64_t)c, std::memory_order_release);
        share(c, 0);                              // the calling thread is worker 0 of this chunk
        done[c].fetch_add(1, std::memory_order_release);
        while (done[c].load(std::memory_order_acquire) < n_thr) std::this_thread::yield();
        if (c + 2 < n_chunks) issue(c + 2);       // this half is free again
    }
    if (e != hipSuccess) abort.store(true);
    ready.store((int64_t)n_chunks, std::memory_order_release);
    for (auto &t : th) t.join();
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(st);
        return fail(BSIG_ERR_DEVICE, "result download failed: %s", hipGetErrorString(e));
    }
    return BSIG_OK;
}

}  // namespace

int bsig::download_to_host(bsig_ctx *ctx, const void *src_dev, void *dst_host, size_t bytes)
{
    if (bytes == 0) return BSIG_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (bytes >= (8u << 20) && !is_pinned_host(dst_host) && bytes % 4 == 0) {
        int32_t *dst = (int32_t *)dst_host;
        return download_staged(ctx->device, ctx->stream, {DlSlice{(const uint8_t *)src_dev, bytes, 0}},
                               [dst](int64_t c0, const int32_t *src, int64_t cells) { memcpy(dst + c0, src, (size_t)cells * 4); });
    }
    hipError_t e = hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(BSIG_ERR_DEVICE, "result download failed: %s", hipGetErrorString(e));
    return BSIG_OK;
}

void bsig::HostDest::put(int64_t c0, const int32_t *src, int64_t count) const
{
    if (count <= 0) return;
    if (flat) { memcpy(flat + c0, src, (size_t)count * sizeof(int32_t)); return; }
    // the range that holds cell c0: the last i with off[i] <= c0 (empty ranges in between are skipped by the loop)
    int64_t i = (int64_t)(std::upper_bound(off, off + n + 1, c0) - off) - 1;
    while (count > 0 && i < n) {
        const int64_t end = off[i + 1];
        if (end <= c0) { ++i; continue; }
        const int64_t take = std::min(count, end - c0);
        memcpy(ptrs[i] + (c0 - off[i]), src, (size_t)take * sizeof(int32_t));
        c0 += take; src += take; count -= take;
        ++i;
    }
}

// The result straight into its final place: into one flat buffer, or range by range into the vectors the
// caller made for them (no flat staging copy of the result in host memory: the reference counts into the R
// vectors themselves, ref: src/bamsignals.cpp:172-190,361-362).  Large results cross PCIe by DMA into two
// page-locked halves, and a few threads move each half on -- range by range where the destination is one.
int bsig::download_to_dest(bsig_ctx *ctx, const int32_t *src_dev, const HostDest &dst, int64_t cells)
{
    return download_slice_to_dest(ctx, src_dev, dst, 0, cells, 0);
}

int bsig::download_slices_to_dest(bsig_ctx *ctx, const int32_t *src_dev, int64_t n_slices, const int64_t *src_c0, const int64_t *dst_c0,
                                  const int64_t *cells, const HostDest &dst, int copy_threads)
{
    std::vector<DlSlice> sl;
    int64_t total = 0;
    for (int64_t k = 0; k < n_slices; ++k)
        if (cells[k] > 0) { sl.push_back(DlSlice{(const uint8_t *)(src_dev + src_c0[k]), (size_t)cells[k] * 4, dst_c0[k]}); total += cells[k]; }
    if (sl.empty()) return BSIG_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (dst.flat && sl.size() == 1 && copy_threads == 0) return download_to_host(ctx, sl[0].src, dst.flat + sl[0].dst_c0, sl[0].bytes);
    if ((size_t)total * 4 < (1u << 20)) {
        // a small result: through a buffer of its own (the runtime stages a copy into pageable memory anyway)
        std::vector<int32_t> tmp((size_t)total);
        hipError_t e = hipSuccess;
        int64_t at = 0;
        for (const DlSlice &q : sl) {
            if (e == hipSuccess) e = hipMemcpyAsync(tmp.data() + at, q.src, q.bytes, hipMemcpyDeviceToHost, ctx->stream);
            at += (int64_t)(q.bytes / 4);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(BSIG_ERR_DEVICE, "result download failed: %s", hipGetErrorString(e));
        at = 0;
        for (const DlSlice &q : sl) { dst.put(q.dst_c0, tmp.data() + at, (int64_t)(q.bytes / 4)); at += (int64_t)(q.bytes / 4); }
        return BSIG_OK;
    }
    const HostDest *D = &dst;
    return download_staged(ctx->device, ctx->stream, sl, [D](int64_t c0, const int32_t *src, int64_t n) { D->put(c0, src, n); }, copy_threads);
}

int bsig::download_slice_to_dest(bsig_ctx *ctx, const int32_t *src_dev, const HostDest &dst, int64_t c0, int64_t cells, int copy_threads)
{
    const int64_t zero = 0;
    return download_slices_to_dest(ctx, src_dev, 1, &zero, &c0, &cells, dst, copy_threads);
}

extern "C" {

// ---------------------------------------------------------------------------------------------
// the resident layout as a file (the decoded-column sidecar of SURVEY 8f.4)
// ---------------------------------------------------------------------------------------------
namespace {
struct SidecarClass {
    int64_t n;
    int32_t maxspan, kshift;
    uint64_t col_cap, idx_entries;
};
struct SidecarHeader {
    char magic[8];                  // "BSIGRDS1"
    uint32_t version, n_ref;
    int64_t n_reads;
    uint32_t stamp_len, n_classes;
    uint64_t file_bytes;
    SidecarClass cls[BSIG_MAX_CLASSES];
    uint64_t checksum;              // of every column, index and the pair table as they lay in HBM when the file was written
    uint32_t n_codes, reserved;     // pairs in the packed class's table (the table follows the reference arrays)
};
constexpr uint32_t kSidecarVersion = 4;
// which columns a class has: pos (all but the packed class), end (classes 2 and 3), fm, tlen
inline bool class_has_pos(int c) { return c != BSIG_CLASS_PACKED; }
inline bool class_has_end(int c) { return c == 2 || c == 3; }
inline int class_columns(int c) { return 2 + (class_has_pos(c) ? 1 : 0) + (class_has_end(c) ? 1 : 0); }
inline uint64_t pad64(uint64_t v) { return (v + 63) & ~(uint64_t)63; }
}  // namespace

namespace {
// checksum of the class columns and indexes of a resident layout, computed where they lie (kernels.hip)
int layout_checksum(const bsig_reads *R, uint64_t *out)
{
    hipStream_t st = R->ctx->stream;
    HIP_TRY(hipSetDevice(R->ctx->device));
    unsigned long long *d_acc = nullptr, acc = 0;
    HIP_TRY(hipMalloc((void **)&d_acc, sizeof acc));
    hipError_t e = hipMemsetAsync(d_acc, 0, sizeof acc, st);
    uint64_t salt = 1;
    for (int c = 0; c < BSIG_MAX_CLASSES && e == hipSuccess; ++c) {
        const BsigClassCols &C = R->dev.cls[c];
        if (!C.n) continue;
        const uint64_t cap = R->col_cap[c];
        for (const void *col : {(const void *)C.pos, (const void *)C.end, (const void *)C.fm, (const void *)C.tlen}) {
            if (col && e == hipSuccess) e = bsig::launch_checksum(col, cap, salt, d_acc, st);
            salt += 0x100000001ull;
        }
        // (the index's spare last entry is never written: not part of the sum)
        if (e == hipSuccess) e = bsig::launch_checksum(C.idx, R->idx_entries[c] - 1, salt, d_acc, st);
        salt += 0x100000001ull;
    }
    if (R->dev.fmtab && e == hipSuccess) e = bsig::launch_checksum(R->dev.fmtab, BSIG_PACK_CODES, salt, d_acc, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&acc, d_acc, sizeof acc, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_acc);
    if (e != hipSuccess) return fail(BSIG_ERR_DEVICE, "layout checksum failed: %s", hipGetErrorString(e));
    *out = (uint64_t)acc;
    return BSIG_OK;
}
}  // namespace

int bsig_reads_save(const bsig_reads *reads, const char *path, const char *stamp)
{
    if (!reads || !path || !stamp) return fail(BSIG_ERR_ARG, "NULL argument to bsig_reads_save");
    HIP_TRY(hipSetDevice(reads->ctx->device));
    SidecarHeader H;
    memset(&H, 0, sizeof H);
    memcpy(H.magic, "BSIGRDS1", 8);
    H.version = kSidecarVersion;
    H.n_ref = (uint32_t)reads->n_ref;
    H.n_reads = reads->info.n_reads;
    H.stamp_len = (uint32_t)strlen(stamp);
    H.n_classes = (uint32_t)reads->info.n_classes;
    H.n_codes = (uint32_t)reads->dev.n_codes;
    uint64_t bytes = pad64(sizeof H) + pad64(H.stamp_len) + 3 * pad64((uint64_t)H.n_ref * 4) + (H.n_codes ? pad64(BSIG_PACK_CODES * 4) : 0);
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c) {
        const BsigClassCols &C = reads->dev.cls[c];
        H.cls[c] = SidecarClass{C.n, C.maxspan, C.kshift, C.n ? reads->col_cap[c] : 0, C.n ? reads->idx_entries[c] : 0};
        if (!C.n) continue;
        bytes += class_columns(c) * pad64(H.cls[c].col_cap * 4) + pad64(H.cls[c].idx_entries * 4);
    }
    H.file_bytes = bytes;
    {
        const int crc_rc = layout_checksum(reads, &H.checksum);
        if (crc_rc) return crc_rc;
    }
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((long long)getpid());
    FILE *fp = fopen(tmp.c_str(), "wb");
    if (!fp) return fail(BSIG_ERR_IO, "cannot write %s", tmp.c_str());
    bool ok = true;
    static const char zeros[64] = {0};
    auto put = [&](const void *p, uint64_t n) {
        ok = ok && (n == 0 || fwrite(p, 1, n, fp) == n);
        const uint64_t pad = pad64(n) - n;
        ok = ok && (pad == 0 || fwrite(zeros, 1, pad, fp) == pad);
    };
    put(&H, sizeof H);
    put(stamp, H.stamp_len);
    put(reads->ref_len.data(), (uint64_t)H.n_ref * 4);
    put(reads->ref_unit0.data(), (uint64_t)H.n_ref * 4);
    put(reads->ref_units.data(), (uint64_t)H.n_ref * 4);
    if (H.n_codes) put(reads->fmtab.data(), BSIG_PACK_CODES * 4);
    std::vector<uint8_t> host;
    int rc = BSIG_OK;
    auto put_dev = [&](const void *d, uint64_t n) {
        if (rc || !ok) return;
        host.resize(n);
        rc = bsig::download_to_host(reads->ctx, d, host.data(), n);
        if (!rc) put(host.data(), n);
    };
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c) {
        const BsigClassCols &C = reads->dev.cls[c];
        if (!C.n) continue;
        if (class_has_pos(c)) put_dev(C.pos, H.cls[c].col_cap * 4);
        if (class_has_end(c)) put_dev(C.end, H.cls[c].col_cap * 4);
        put_dev(C.fm, H.cls[c].col_cap * 4);
        put_dev(C.tlen, H.cls[c].col_cap * 4);
        put_dev(C.idx, H.cls[c].idx_entries * 4);
    }
    ok = (fclose(fp) == 0) && ok;
    if (rc || !ok || rename(tmp.c_str(), path) != 0) {
        remove(tmp.c_str());
        return rc ? rc : fail(BSIG_ERR_IO, "writing %s failed", path);
    }
    return BSIG_OK;
}

namespace {
// pageable host memory -> device through two page-locked halves filled by a few threads (the
// mirror image of download_staged)
int upload_staged(int device, hipStream_t st, const uint8_t *src, uint8_t *dst_dev, size_t bytes)
{
    DownloadStage &D = download_for(device);
    std::lock_guard<std::mutex> lock(D.mu);
    HIP_TRY(hipSetDevice(device));
    int rc = D.ensure();
    if (rc) return rc;
    bsig::PinnedPair &pair = bsig::pinned_pair_for(device);
    std::lock_guard<std::mutex> pair_lock(pair.mu);
    rc = pair.ensure(DownloadStage::kHalf);
    if (rc) return rc;
    const size_t half = DownloadStage::kHalf;
    int n_thr = 8;
    if (const char *e = getenv("BAMSIGNALS_COPY_THREADS")) n_thr = std::max(1, std::min(64, atoi(e)));
    bool used[2] = {false, false};
    size_t c = 0;
    for (size_t at = 0; at < bytes; at += half, ++c) {
        const size_t len = std::min(half, bytes - at);
        const int h = (int)(c & 1);
        if (used[h]) HIP_TRY(hipEventSynchronize(D.ev[h]));
        std::vector<std::thread> th;
        auto part = [&](int t) {
            const size_t a = len * (size_t)t / (size_t)n_thr, b = len * (size_t)(t + 1) / (size_t)n_thr;
            if (b > a) memcpy(pair.buf[h] + a, src + at + a, b - a);
        };
        for (int t = 1; t < n_thr; ++t) th.emplace_back(part, t);
        part(0);
        for (auto &x : th) x.join();
        HIP_TRY(hipMemcpyAsync(dst_dev + at, pair.buf[h], len, hipMemcpyHostToDevice, st));
        HIP_TRY(hipEventRecord(D.ev[h], st));
        used[h] = true;
    }
    HIP_TRY(hipStreamSynchronize(st));
    return BSIG_OK;
}
}  // namespace

int bsig_reads_load(bsig_ctx *ctx, const char *path, const char *stamp, bsig_reads **out)
{
    if (!ctx || !path || !stamp || !out) return fail(BSIG_ERR_ARG, "NULL argument to bsig_reads_load");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(BSIG_ERR_IO, "cannot open %s", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < sizeof(SidecarHeader)) { close(fd); return fail(BSIG_ERR_FORMAT, "%s is not a reads file", path); }
    const size_t size = (size_t)sb.st_size;
    void *map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return fail(BSIG_ERR_IO, "cannot map %s", path);
    const uint8_t *base = (const uint8_t *)map;
    struct Unmap { void *p; size_t n; ~Unmap() { munmap(p, n); } } unmap{map, size};
    SidecarHeader H;
    memcpy(&H, base, sizeof H);
    if (memcmp(H.magic, "BSIGRDS1", 8) != 0 || H.version != kSidecarVersion || H.file_bytes != size)
        return fail(BSIG_ERR_FORMAT, "%s is not a reads file of this version (or is truncated)", path);
    uint64_t at = pad64(sizeof H);
    auto take = [&](uint64_t n) -> const uint8_t * {
        if (at + pad64(n) > size) return nullptr;
        const uint8_t *p = base + at;
        at += pad64(n);
        return p;
    };
    const uint8_t *st_p = take(H.stamp_len);
    if (!st_p || H.stamp_len != strlen(stamp) || memcmp(st_p, stamp, H.stamp_len) != 0)
        return fail(BSIG_ERR_FORMAT, "%s was made from another version of the BAM file", path);
    if (H.n_ref > (1u << 28)) return fail(BSIG_ERR_FORMAT, "%s is damaged", path);
    const uint8_t *p_len = take((uint64_t)H.n_ref * 4), *p_u0 = take((uint64_t)H.n_ref * 4), *p_un = take((uint64_t)H.n_ref * 4);
    if (!p_len || !p_u0 || !p_un) return fail(BSIG_ERR_FORMAT, "%s is truncated", path);
    std::unique_ptr<bsig_reads> R(new bsig_reads);
    R->ctx = ctx;
    R->n_ref = (int32_t)H.n_ref;
    R->ref_len.assign((const int32_t *)p_len, (const int32_t *)p_len + H.n_ref);
    R->ref_unit0.assign((const uint32_t *)p_u0, (const uint32_t *)p_u0 + H.n_ref);
    R->ref_units.assign((const uint32_t *)p_un, (const uint32_t *)p_un + H.n_ref);
    R->info = bsig_reads_info{};
    R->info.n_reads = H.n_reads;
    R->layout_gen = bsig::next_layout_gen();
    // no table, no packed reads; the converse does not hold: the pair sample may have seen short reads of which
    // none qualified for the packed class (all beyond their reference's end), and such a layout is saved as it is
    if (H.n_codes > BSIG_PACK_CODES || (H.n_codes == 0 && H.cls[BSIG_CLASS_PACKED].n != 0))
        return fail(BSIG_ERR_FORMAT, "%s is damaged (pair table)", path);
    if (H.n_codes) {
        const uint8_t *p_tab = take(BSIG_PACK_CODES * 4);
        if (!p_tab) return fail(BSIG_ERR_FORMAT, "%s is truncated", path);
        R->fmtab.assign((const uint32_t *)p_tab, (const uint32_t *)p_tab + BSIG_PACK_CODES);
        // a code's pair: 12 flag bits, an 8-bit mapq, nothing else; unused entries zero
        for (uint32_t c = 0; c < BSIG_PACK_CODES; ++c)
            if ((R->fmtab[c] & 0xFF00F000u) || (c >= H.n_codes && R->fmtab[c]))
                return fail(BSIG_ERR_FORMAT, "%s is damaged (pair table)", path);
    }
    // The file's numbers steer device-side indexing (bucket numbers, read windows), so nothing is taken on
    // trust: the unit tables must be the ones layout_from_device derives from the reference lengths, every
    // class's shapes must follow from its read count and bucket shift, the counts must add up -- and below
    // the indexes are checked on the device and the checksum of what reached HBM must match the header's.
    uint64_t total_units = 0;
    for (uint32_t r = 0; r < H.n_ref; ++r) {
        if (R->ref_len[r] < 0) return fail(BSIG_ERR_FORMAT, "%s is damaged (reference lengths)", path);
        const uint64_t u = ((uint64_t)R->ref_len[r] >> BSIG_REF_UNIT_SHIFT) + 1;
        if (R->ref_unit0[r] != total_units || R->ref_units[r] != u || total_units + u >= (1ull << 31))
            return fail(BSIG_ERR_FORMAT, "%s is damaged (reference units)", path);
        total_units += u;
    }
    const uint64_t total_bp = total_units << BSIG_REF_UNIT_SHIFT;
    {
        int64_t sum = 0;
        uint32_t live = 0;
        for (int c = 0; c < BSIG_MAX_CLASSES; ++c) {
            if (H.cls[c].n < 0 || H.cls[c].n > H.n_reads) return fail(BSIG_ERR_FORMAT, "%s is damaged (class counts)", path);
            sum += H.cls[c].n;
            live += H.cls[c].n > 0;
        }
        if (H.n_reads < 0 || sum != H.n_reads || live != H.n_classes || (H.n_reads > 0 && H.n_ref == 0))
            return fail(BSIG_ERR_FORMAT, "%s is damaged (class counts)", path);
    }
    HIP_TRY(hipSetDevice(ctx->device));
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c) {
        BsigClassCols &C = R->dev.cls[c];
        C = BsigClassCols{};
        const SidecarClass &K = H.cls[c];
        if (K.n <= 0) continue;
        if ((uint64_t)K.n >= (1ull << 32) - 8 || K.col_cap != ((uint64_t)K.n + 3) / 4 * 4 + 4 || K.kshift < 4 ||
            K.kshift > (c == BSIG_CLASS_PACKED ? BSIG_PACK_POS_BITS : BSIG_REF_UNIT_SHIFT) || (c == BSIG_CLASS_PACKED && K.maxspan > 256) || (total_bp >> K.kshift) >= (1ull << 32) || K.idx_entries != (total_bp >> K.kshift) + 2 ||
            K.maxspan < 1)
            return fail(BSIG_ERR_FORMAT, "%s is damaged (shape of span class %d)", path, c);
        auto load = [&](uint64_t count, const void **dst) -> int {
            const uint8_t *src = take(count * 4);
            if (!src) return fail(BSIG_ERR_FORMAT, "%s is truncated", path);
            uint32_t *d = nullptr;
            HIP_TRY(R->pool.alloc(&d, (size_t)count));
            *dst = d;
            return upload_staged(ctx->device, ctx->stream, src, (uint8_t *)d, (size_t)count * 4);
        };
        int rc = BSIG_OK;
        if (class_has_pos(c)) rc = load(K.col_cap, (const void **)&C.pos);
        if (!rc && class_has_end(c)) rc = load(K.col_cap, (const void **)&C.end);
        if (!rc) rc = load(K.col_cap, (const void **)&C.fm);
        if (!rc) rc = load(K.col_cap, (const void **)&C.tlen);
        if (!rc) rc = load(K.idx_entries, (const void **)&C.idx);
        if (rc) return rc;
        C.n = K.n; C.maxspan = K.maxspan; C.kshift = K.kshift;
        R->col_cap[c] = K.col_cap;
        R->idx_entries[c] = K.idx_entries;
        R->info.class_n[c] = K.n;
        R->info.class_maxspan[c] = K.maxspan;
        R->info.class_bucket_shift[c] = K.kshift;
        R->info.n_classes += 1;
    }
    R->dev.fmtab = nullptr;
    R->dev.n_codes = (int32_t)H.n_codes;
    R->info.n_codes = (int32_t)H.n_codes;
    if (H.n_codes) {
        uint32_t *d = nullptr;
        HIP_TRY(R->pool.alloc(&d, BSIG_PACK_CODES));
        HIP_TRY(hipMemcpyAsync(d, R->fmtab.data(), BSIG_PACK_CODES * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        R->dev.fmtab = d;
    }
    R->info.hbm_bytes = R->pool.footprint();
    // what arrived in HBM: indexes the kernels can follow blindly, and the bytes the writer had
    {
        int *d_bad = nullptr, bad = 0;
        HIP_TRY(hipMalloc((void **)&d_bad, sizeof(int)));
        hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream);
        for (int c = 0; c < BSIG_MAX_CLASSES && e == hipSuccess; ++c) {
            const BsigClassCols &C = R->dev.cls[c];
            if (C.n) e = bsig::launch_check_idx(C.idx, R->idx_entries[c] - 2, (uint32_t)C.n, d_bad, ctx->stream);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_bad);
        if (e != hipSuccess) return fail(BSIG_ERR_DEVICE, "checking %s failed: %s", path, hipGetErrorString(e));
        if (bad) return fail(BSIG_ERR_FORMAT, "%s is damaged (bucket index)", path);
        uint64_t sum = 0;
        const int rc = layout_checksum(R.get(), &sum);
        if (rc) return rc;
        if (sum != H.checksum) return fail(BSIG_ERR_FORMAT, "%s is damaged (checksum)", path);
    }
    *out = R.release();
    return BSIG_OK;
}

// ---------------------------------------------------------------------------------------------
// plans
// ---------------------------------------------------------------------------------------------
int bsig_plan_create(bsig_ctx *ctx, const bsig_reads *reads, int64_t n, const int32_t *rid,
                     const int32_t *loc, const int32_t *len, const int32_t *strand,
                     const bsig_params *prm, bsig_plan **out)
{
    if (!ctx || !reads || !prm || !out) return fail(BSIG_ERR_ARG, "NULL argument to bsig_plan_create");
    *out = nullptr;
    if (n < 0 || (n > 0 && (!rid || !loc || !len || !strand))) return fail(BSIG_ERR_ARG, "range arrays missing");
    const int mode = prm->mode;
    if (mode != BSIG_MODE_PROFILE && mode != BSIG_MODE_COUNT && mode != BSIG_MODE_COVERAGE)
        return fail(BSIG_ERR_ARG, "unknown mode %d", mode);
    if (mode == BSIG_MODE_PROFILE && prm->binsize < 1)
        return fail(BSIG_ERR_ARG, "provide a binsize greater or equal to 1");       // ref: R/wrappers.R:136-137
    if (prm->n_tlen_filter != 0 && prm->n_tlen_filter != 2)
        return fail(BSIG_ERR_ARG, "tlen_filter must have 0 or 2 elements");
    const bool mid = mode != BSIG_MODE_COVERAGE && prm->pe_mid;
    const bool tspan = mode == BSIG_MODE_COVERAGE && prm->tspan;
    if ((mid || tspan) && prm->n_tlen_filter != 2)
        return fail(BSIG_ERR_ARG, "paired-end midpoint/extend needs a 2-element tlen_filter");
    // ext: ref src/bamsignals.cpp:457 (pileup) and :487 (coverage); :243 rejects negatives
    int64_t ext;
    if (mode == BSIG_MODE_COVERAGE) ext = tspan ? prm->tlen_filter[1] : 0;
    else ext = std::llabs((long long)prm->shift) + (mid ? (int64_t)prm->tlen_filter[1] : 0);
    if (ext < 0) return fail(BSIG_ERR_EXT, "negative 'ext' values don't make sense");
    if (ext > (1ll << 30)) return fail(BSIG_ERR_ARG, "shift / tlen filter too large");
    for (int64_t i = 0; i < n; ++i) {
        if (rid[i] < 0 || rid[i] >= reads->n_ref)
            return fail(BSIG_ERR_CHROM, "chromosome id %d not present in the bam file", rid[i]);
        if (len[i] < 0) return fail(BSIG_ERR_ARG, "range %lld has a negative width", (long long)i);
    }

    bsig_plan *P = new bsig_plan;
    P->ctx = ctx; P->reads = reads; P->mode = mode; P->n_ranges = n;
    P->made_for_gen = reads->layout_gen;
    // default tile: the widest range if it fits 2048 cells (less LDS per workgroup = more
    // workgroups per CU), else 2048-cell tiles
    int64_t widest = 64;
    {
        const int64_t bsz = mode == BSIG_MODE_PROFILE ? prm->binsize : 1;
        for (int64_t i = 0; i < n && mode != BSIG_MODE_COUNT; ++i)
            widest = std::max<int64_t>(widest, ((int64_t)len[i] + bsz - 1) / bsz);
    }
    // (strand-split images hold two values per cell: keep the image at 8 KiB there too, otherwise
    // only 10 workgroups fit a CU and the launch is occupancy-bound: 0.62 -> 0.55 ms on config 4)
    const int64_t cap_cells = (mode == BSIG_MODE_PROFILE && prm->ss) ? 1024 : 2048;
    P->tile_cells = prm->tile_cells > 0 ? prm->tile_cells : (int)std::min<int64_t>(widest, cap_cells);
    int min_cells = 64;
    if (mode == BSIG_MODE_PROFILE && prm->binsize > 1 && prm->tile_cells <= 0) {
        // wide bins: a tile of 2048 cells would span megabases and one wave would stream all of
        // its reads; keep a tile to about 16 kbp so that genome-wide binning still fills the chip
        const int64_t by_span = std::max<int64_t>(4, (16384 + prm->binsize - 1) / prm->binsize);
        P->tile_cells = (int)std::min<int64_t>(P->tile_cells, by_span);
        min_cells = 4;
    }
    // a tile image is at most 32 KiB of LDS
    P->tile_cells = std::min(std::max(P->tile_cells, min_cells), prm->ss && mode == BSIG_MODE_PROFILE ? 4096 : 8192);
    P->tile_cells = (P->tile_cells + 3) & ~3;
    P->threads = prm->threads > 0 ? prm->threads : 64;
    if (P->threads != 64 && P->threads != 128 && P->threads != 256) {
        delete P;
        return fail(BSIG_ERR_ARG, "threads must be 64, 128 or 256");
    }
    BsigKParams &K = P->kp;
    K.mapqual = prm->mapqual;
    K.requiredF = (uint32_t)prm->requiredF;
    K.filteredF = (uint32_t)prm->filteredF;
    K.has_tlen_filter = prm->n_tlen_filter == 2;
    K.tf0 = prm->tlen_filter[0]; K.tf1 = prm->tlen_filter[1];
    K.shift = mode == BSIG_MODE_COVERAGE ? 0 : prm->shift;
    K.midpoint = mid; K.tspan = tspan;
    K.use_tlen = K.has_tlen_filter || mid || tspan;
    K.ss = mode == BSIG_MODE_COVERAGE ? 0 : (prm->ss != 0);
    K.binsize = mode == BSIG_MODE_PROFILE ? prm->binsize : 1;
    K.ext = (int32_t)ext;
    K.tile_cells = P->tile_cells;
    bsig::magic_u31(K.binsize, &K.div_magic, &K.div_shift);
    K.div_m15 = 0; K.div_s15 = 0;
    if (K.binsize >= 2 && K.binsize <= 8192) {
        // s = 15 + ceil(log2 b), m = ceil(2^s / b): n * m / 2^s = n / b + n * e / (b * 2^s) with e < b, and the second
        // term stays below 2^-ceil(log2 b) <= 1 / b for n < 2^15, so the floor is exact; n * m < 2^32, m < 2^17
        int L = 0;
        while ((1 << L) < K.binsize) ++L;
        K.div_s15 = 15 + L;
        K.div_m15 = (uint32_t)((((uint64_t)1 << K.div_s15) + (uint64_t)K.binsize - 1) / (uint64_t)K.binsize);
    }

    const int32_t lay_binsize = mode == BSIG_MODE_COUNT ? -1 : K.binsize;
    P->off.resize(n + 1);
    bsig_layout(n, len, lay_binsize, K.ss, P->off.data());

    // tiles in genomic order (ref: std::sort by (rid, loc), src/bamsignals.cpp:222-226,246):
    // neighbouring workgroups then stream neighbouring reads
    std::vector<int64_t> order;
    bsig::sort_ranges(n, rid, loc, order);
    std::vector<BsigWorkItem> items;
    items.reserve(n);
    const int64_t mult = K.ss ? 2 : 1;
    // count mode: bases per workgroup (with the window's reach on both sides still one chunk of the packed
    // class's position bits: one index lookup per tile)
    const int count_split = 1 << (BSIG_PACK_POS_BITS - 1);
    // bins wider than a workgroup should stream on its own: every bin becomes bamCount-style
    // sub-intervals that add into the (zeroed) result with integer atomics
    const bool wide_bins = mode == BSIG_MODE_PROFILE && prm->tile_cells <= 0 && K.binsize > count_split / 2;
    P->kernel_mode = wide_bins ? BSIG_MODE_COUNT : mode;
    for (int64_t k = 0; k < n; ++k) {
        const int64_t i = order[k];
        if (len[i] <= 0) {
            if (P->off[i + 1] > P->off[i]) P->needs_zero = true;      // bamCount of a zero-width range: 0
            continue;
        }
        BsigWorkItem w{};
        w.loc = loc[i]; w.len = len[i];
        w.ref_unit0 = reads->ref_unit0[rid[i]];
        w.units_strand = reads->ref_units[rid[i]] | (strand[i] < 0 ? BSIG_ITEM_NEG : 0u);
        if (wide_bins) {
            const int64_t cells = (P->off[i + 1] - P->off[i]) / mult;
            for (int64_t c = 0; c < cells; ++c) {
                // cell c covers [c*bs, (c+1)*bs) in range orientation (ref: src/bamsignals.cpp:356-362)
                const int64_t ra = c * (int64_t)K.binsize, rb = std::min<int64_t>(len[i], ra + K.binsize);
                const int64_t g0 = strand[i] < 0 ? len[i] - rb : ra, g1 = strand[i] < 0 ? len[i] - ra : rb;
                for (int64_t a = g0; a < g1; a += count_split) {
                    w.c0 = (int32_t)a;
                    w.nc = (int32_t)std::min<int64_t>(count_split, g1 - a);
                    w.out_off = P->off[i] + c * mult;
                    w.units_strand |= BSIG_ITEM_ATOMIC;
                    P->needs_zero = true;
                    items.push_back(w);
                }
            }
        } else if (mode == BSIG_MODE_COUNT) {
            const bool split = len[i] > count_split;
            for (int64_t a = 0; a < len[i]; a += count_split) {
                w.c0 = (int32_t)a;
                w.nc = (int32_t)std::min<int64_t>(count_split, len[i] - a);
                w.out_off = P->off[i];
                if (split) { w.units_strand |= BSIG_ITEM_ATOMIC; P->needs_zero = true; }
                items.push_back(w);
            }
        } else {
            const int64_t cells = (P->off[i + 1] - P->off[i]) / mult;
            for (int64_t c0 = 0; c0 < cells; c0 += P->tile_cells) {
                w.c0 = (int32_t)c0;
                w.nc = (int32_t)std::min<int64_t>(P->tile_cells, cells - c0);
                w.out_off = P->off[i] + c0 * mult;
                items.push_back(w);
            }
        }
    }
    P->n_items = (int64_t)items.size();
    if (P->n_items >= (1ll << 31)) { delete P; return fail(BSIG_ERR_ARG, "too many tiles for one launch"); }
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = P->pool.alloc(&P->items, std::max<size_t>(items.size(), 1));
    // the packed class's filter table for these parameters (kernels.hip: k_make_ptab)
    if (e == hipSuccess) e = P->pool.alloc(&P->ptab, (size_t)BSIG_PACK_CODES);
    if (e == hipSuccess) e = bsig::launch_make_ptab(reads->dev, K, P->ptab, ctx->stream);
    K.ptab = P->ptab;
    if (e == hipSuccess && !items.empty())
        e = hipMemcpyAsync(P->items, items.data(), items.size() * sizeof(BsigWorkItem), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);

    // ---- heavy tiles --------------------------------------------------------------------------
    // One wave streams a tile's reads; a tile on a read hotspot (chrM, rDNA, an amplicon) can hold
    // millions and would keep the whole launch waiting.  Probe the window sizes once per plan and
    // cut the windows of heavy tiles into slices that separate workgroups add up with atomics.
    int64_t heavy_reads = 32768, slice_reads = 8192;
    // (k_profile's 16-bit tile image relies on the 32,768 ceiling: the environment may only lower it)
    if (const char *v = getenv("BAMSIGNALS_HEAVY_READS")) { heavy_reads = std::min<long long>(32768, std::max<long long>(4, atoll(v))); slice_reads = std::max<int64_t>(4, heavy_reads / 4); }
    // (k_coverage's cells are SIGNED 16-bit: +32,768 starts on one cell would not fit)
    if (P->kernel_mode == BSIG_MODE_COVERAGE) heavy_reads = std::min<int64_t>(heavy_reads, 32767);
    if (e == hipSuccess && !items.empty()) {
        DevPool tmp;
        uint2 *d_win = nullptr;
        e = tmp.alloc(&d_win, items.size() * BSIG_MAX_CLASSES);
        std::vector<uint2> win;
        unsigned long long *d_heavy = nullptr, n_heavy_dev = 0;
        if (e == hipSuccess) e = tmp.alloc(&d_heavy, 1);
        if (e == hipSuccess) e = hipMemsetAsync(d_heavy, 0, sizeof(unsigned long long), ctx->stream);
        if (e == hipSuccess) e = bsig::launch_resolve(reads->dev, K, P->kernel_mode, P->items, P->n_items, d_win, ctx->stream);
        if (e == hipSuccess) e = bsig::launch_count_heavy(d_win, P->n_items, heavy_reads, d_heavy, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&n_heavy_dev, d_heavy, sizeof n_heavy_dev, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        // the windows themselves are only fetched when there is something to slice
        if (e == hipSuccess && n_heavy_dev) {
            win.resize(items.size() * BSIG_MAX_CLASSES);
            e = hipMemcpyAsync(win.data(), d_win, win.size() * sizeof(uint2), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
        std::vector<BsigWorkItem> hitems;
        std::vector<uint2> hwin;
        if (e == hipSuccess && n_heavy_dev) {
            for (size_t t = 0; t < items.size(); ++t) {
                int64_t total = 0;
                for (int c = 0; c < BSIG_MAX_CLASSES; ++c) total += (int64_t)win[t * BSIG_MAX_CLASSES + c].y - win[t * BSIG_MAX_CLASSES + c].x;
                if (total <= heavy_reads) continue;
                ++P->n_heavy_tiles;
                BsigWorkItem sl = items[t];
                if (P->kernel_mode == BSIG_MODE_COUNT) sl.units_strand |= BSIG_ITEM_ATOMIC;
                for (int c = 0; c < BSIG_MAX_CLASSES; ++c) {
                    const uint2 wc = win[t * BSIG_MAX_CLASSES + c];
                    for (int64_t j0 = wc.x; j0 < (int64_t)wc.y; j0 += slice_reads) {
                        hitems.push_back(sl);
                        for (int k = 0; k < BSIG_MAX_CLASSES; ++k)
                            hwin.push_back(k == c ? make_uint2((uint32_t)j0, (uint32_t)std::min<int64_t>(j0 + slice_reads, wc.y)) : make_uint2(0u, 0u));
                    }
                }
                items[t].units_strand |= BSIG_ITEM_HEAVY;
            }
        }
        if (e == hipSuccess && !hitems.empty()) {
            P->n_heavy_slices = (int64_t)hitems.size();
            uint2 *hw = nullptr;
            e = P->pool.alloc(&P->heavy_items, hitems.size());
            if (e == hipSuccess) e = P->pool.alloc(&hw, hwin.size());
            P->heavy_windows = hw;
            if (e == hipSuccess) e = hipMemcpyAsync(P->heavy_items, hitems.data(), hitems.size() * sizeof(BsigWorkItem), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(hw, hwin.data(), hwin.size() * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream);
            // the heavy flags of the main items
            if (e == hipSuccess) e = hipMemcpyAsync(P->items, items.data(), items.size() * sizeof(BsigWorkItem), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
    }
    if (e != hipSuccess) {
        delete P;
        return fail(e == hipErrorOutOfMemory ? BSIG_ERR_NOMEM : BSIG_ERR_DEVICE, "plan upload failed: %s", hipGetErrorString(e));
    }
    *out = P;
    return BSIG_OK;
}

const int64_t *bsig_plan_offsets(const bsig_plan *p) { return p ? p->off.data() : nullptr; }

int64_t bsig_plan_cells(const bsig_plan *p) { return p ? p->off.back() : 0; }

int bsig_plan_run(bsig_plan *p, int32_t *out_dev)
{
    if (!p) return fail(BSIG_ERR_ARG, "plan is NULL");
    const int64_t cells = p->off.back();
    if (cells == 0) return BSIG_OK;
    if (!out_dev) return fail(BSIG_ERR_ARG, "output buffer is NULL");
    if (((uintptr_t)out_dev & 15) != 0) return fail(BSIG_ERR_ARG, "device output buffer must be 16-byte aligned");
    if (p->reads->layout_gen != p->made_for_gen)
        return fail(BSIG_ERR_ARG, "the reads were laid out again after this plan was made: make a new plan");
    HIP_TRY(hipSetDevice(p->ctx->device));      // the caller's thread may have another GPU current
    hipStream_t st = p->ctx->stream;
    // (heavy tiles need no fill: their main item stores 0 and only the slices add)
    if (p->kernel_mode == BSIG_MODE_COUNT && p->needs_zero)
        HIP_TRY(hipMemsetAsync(out_dev, 0, cells * sizeof(int32_t), st));
    // Large launches look their tiles' windows up in a launch of their own (k_resolve_tiles, one lane per tile), so
    // that a pileup workgroup -- which holds its LDS and registers from its first instruction on -- gets its work
    // item and its windows in ONE memory round trip instead of two dependent ones; small launches, where a second
    // launch costs more than it hides, look them up inside the pileup kernel (bam_itr_queryi's counterpart, ref: :267).
    // A plan and a layout of the reads are both immutable, so the windows are a function of the two: the lookup
    // launch runs in the plan's FIRST run on a layout and its result is kept with the plan (round 5; 9-11 us of every
    // later step of a resident plan; env BAMSIGNALS_CACHE_WINDOWS=0: looked up in every run, as in round 4).  A file-
    // level call makes its plan and runs it once: it looks its windows up once either way.
    const bool two_launches = plan_two_launches(p);
    if (two_launches) {
        const bool keep = windows_kept();
        if (!p->resolved) HIP_TRY(p->pool.alloc(&p->resolved, (size_t)p->n_items));
        const bool lookup = !keep || p->resolved_gen != p->reads->layout_gen;
        BsigKParams res = p->kp;
        res.resolved = 1;
        HIP_TRY(bsig::launch_pileup(p->kernel_mode, p->kp.ss, p->threads, p->reads->dev, res, p->items, p->n_items,
                                    p->tile_cells, p->resolved, lookup, out_dev, st));
        p->resolved_gen = p->reads->layout_gen;
    } else {
        HIP_TRY(bsig::launch_pileup(p->kernel_mode, p->kp.ss, p->threads, p->reads->dev, p->kp, p->items, p->n_items,
                                    p->tile_cells, nullptr, false, out_dev, st));
    }
    if (p->n_heavy_slices) {
        // the first launch zero-filled the heavy tiles; their slices now add their partial images
        BsigKParams acc = p->kp;
        acc.accumulate = 1;
        HIP_TRY(bsig::launch_pileup(p->kernel_mode, p->kp.ss, p->threads, p->reads->dev, acc, p->heavy_items,
                                    p->n_heavy_slices, p->tile_cells, p->heavy_windows, false, out_dev, st));
    }
    ++p->runs;
    return BSIG_OK;
}

int bsig_plan_run_host(bsig_plan *p, int32_t *out_host)
{
    if (!p) return fail(BSIG_ERR_ARG, "plan is NULL");
    const int64_t cells = p->off.back();
    if (cells == 0) return BSIG_OK;
    if (!out_host) return fail(BSIG_ERR_ARG, "output buffer is NULL");
    HIP_TRY(hipSetDevice(p->ctx->device));
    if (!p->d_out) HIP_TRY(p->pool.alloc(&p->d_out, (size_t)cells));
    int rc = bsig_plan_run(p, p->d_out);
    if (rc != BSIG_OK) return rc;
    const size_t bytes = (size_t)cells * sizeof(int32_t);
    if (bytes >= (8u << 20) && !is_pinned_host(out_host))
        return bsig::download_to_host(p->ctx, p->d_out, out_host, bytes);
    hipError_t e = hipMemcpyAsync(out_host, p->d_out, bytes, hipMemcpyDeviceToHost, p->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->ctx->stream);
    if (e != hipSuccess) rc = fail(BSIG_ERR_DEVICE, "result download failed: %s", hipGetErrorString(e));
    return rc;
}

}  // extern "C"
int bsig::plan_run_host_timed(bsig_plan *p, const HostDest &dst, double *t_kernels, double *t_download)
{
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
    if (!p) return fail(BSIG_ERR_ARG, "plan is NULL");
    const int64_t cells = p->off.back();
    if (cells == 0) return BSIG_OK;
    if (!dst.flat && !dst.ptrs) return fail(BSIG_ERR_ARG, "output buffer is NULL");
    HIP_TRY(hipSetDevice(p->ctx->device));
    if (!p->d_out) HIP_TRY(p->pool.alloc(&p->d_out, (size_t)cells));
    int rc = bsig_plan_run(p, p->d_out);
    if (rc != BSIG_OK) return rc;
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    *t_kernels = since(t0);
    const auto t1 = std::chrono::steady_clock::now();
    rc = bsig::download_to_dest(p->ctx, p->d_out, dst, cells);
    *t_download = since(t1);
    return rc;
}
extern "C" {

int bsig_plan_run_host_async(bsig_plan *p, int32_t *out_host)
{
    if (!p) return fail(BSIG_ERR_ARG, "plan is NULL");
    const int64_t cells = p->off.back();
    if (cells == 0) return BSIG_OK;
    if (!out_host) return fail(BSIG_ERR_ARG, "output buffer is NULL");
    HIP_TRY(hipSetDevice(p->ctx->device));
    if (!p->d_out) HIP_TRY(p->pool.alloc(&p->d_out, (size_t)cells));
    const int rc = bsig_plan_run(p, p->d_out);
    if (rc != BSIG_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out_host, p->d_out, cells * sizeof(int32_t), hipMemcpyDeviceToHost, p->ctx->stream));
    return BSIG_OK;
}

int bsig_plan_get_stats(bsig_plan *p, bsig_plan_stats *s)
{
    if (!p || !s) return fail(BSIG_ERR_ARG, "NULL argument");
    if (!p->have_stats) {
        hipStream_t st = p->ctx->stream;
        unsigned long long *d_acc = nullptr, acc[BSIG_MAX_CLASSES + 1] = {};
        HIP_TRY(hipSetDevice(p->ctx->device));
        HIP_TRY(hipMalloc((void **)&d_acc, sizeof acc));
        hipError_t e = hipMemsetAsync(d_acc, 0, sizeof acc, st);
        if (e == hipSuccess) e = bsig::launch_visits(p->reads->dev, p->kp, p->kernel_mode, p->items, p->n_items, d_acc, st);
        if (e == hipSuccess) e = hipMemcpyAsync(acc, d_acc, sizeof acc, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(d_acc);
        if (e != hipSuccess) return fail(BSIG_ERR_DEVICE, "visit count failed: %s", hipGetErrorString(e));
        bsig_plan_stats &t = p->stats;
        t.n_ranges = p->n_ranges;
        t.n_items = p->n_items;
        t.cells = p->off.back();
        t.visits_packed = (int64_t)acc[BSIG_CLASS_PACKED];   // one word per read
        t.visits_short = (int64_t)(acc[0] + acc[1]);         // classes 0 and 1: no end column
        t.visits = (int64_t)(acc[0] + acc[1] + acc[2] + acc[3] + acc[BSIG_CLASS_PACKED]);
        t.streamed = (int64_t)acc[BSIG_MAX_CLASSES];
        t.heavy_tiles = (int32_t)std::min<int64_t>(p->n_heavy_tiles, INT32_MAX);
        t.bytes_per_visit_packed = p->kp.use_tlen ? 8 : 4;     // the packed word [+ tlen]
        t.bytes_per_visit_short = p->kp.use_tlen ? 12 : 8;     // span <= 4096: pos + flag/mapq/span in one word [+ tlen]
        t.bytes_per_visit_long = p->kp.use_tlen ? 16 : 12;     // pos + end + flag/mapq [+ tlen]
        // reads + work items + index entries + result cells
        const int64_t per_item = (int64_t)sizeof(BsigWorkItem);
        t.algorithmic_bytes = t.bytes_per_visit_packed * t.visits_packed + t.bytes_per_visit_short * t.visits_short +
                              t.bytes_per_visit_long * (t.visits - t.visits_short - t.visits_packed) + per_item * t.n_items + 4 * t.cells;
        if (plan_two_launches(p) && windows_kept()) {
            // a resident plan's step reads the windows kept from its first run: no index entry is touched
            t.algorithmic_bytes += (int64_t)sizeof(BsigResolved) * t.n_items;
        } else {
            t.algorithmic_bytes += 8 * t.n_items * p->reads->info.n_classes;          // the index entries
            // two launches: the work item is read twice and the tile's windows are written and read once
            if (plan_two_launches(p)) t.algorithmic_bytes += (per_item + 2 * (int64_t)sizeof(BsigResolved)) * t.n_items;
        }
        p->have_stats = true;
    }
    *s = p->stats;
    return BSIG_OK;
}

void bsig_plan_free(bsig_plan *p) { delete p; }

// (tests: what a re-layout of the resident columns does to the plans made before it)
int bsig_debug_new_layout_gen(bsig_reads *reads)
{
    if (!reads) return 1;
    reads->layout_gen = bsig::next_layout_gen();
    return 0;
}

int bsig_debug_set_resolve_min(long long n_tiles)
{
    g_resolve_min_override = n_tiles;
    return 0;
}

int bsig_pileup_columns(bsig_ctx *ctx, const bsig_reads *reads, int64_t n, const int32_t *rid,
                        const int32_t *loc, const int32_t *len, const int32_t *strand,
                        const bsig_params *params, int32_t *out_host, const int64_t *off)
{
    bsig_plan *p = nullptr;
    int rc = bsig_plan_create(ctx, reads, n, rid, loc, len, strand, params, &p);
    if (rc != BSIG_OK) return rc;
    if (off && memcmp(off, p->off.data(), (n + 1) * sizeof(int64_t)) != 0) {
        bsig_plan_free(p);
        return fail(BSIG_ERR_ARG, "offsets do not match bsig_layout() for these parameters");
    }
    rc = bsig_plan_run_host(p, out_host);
    bsig_plan_free(p);
    return rc;
}

}  // extern "C"
