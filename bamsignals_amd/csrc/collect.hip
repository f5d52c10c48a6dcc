// See collect.h.  RCCL is loaded with dlopen on first use, so that single-GPU callers never pay for
// (or depend on) librccl; its types come from <rccl/rccl.h>.
#include "collect.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>

using bsig::fail;

namespace {

struct Rccl {
    void *h = nullptr;
    bool ok = false;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                 // optional
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;    // optional (self-check)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr; // optional (self-check)
    Rccl()
    {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return;
        auto sym = [&](const char *s) { return dlsym(h, s); };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        CommAbort = (decltype(CommAbort))sym("ncclCommAbort");
        CommCount = (decltype(CommCount))sym("ncclCommCount");
        CommUserRank = (decltype(CommUserRank))sym("ncclCommUserRank");
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
    }
};
Rccl &rccl()
{
    static Rccl r;
    return r;
}

bool env_wants_rccl()
{
    const char *want = getenv("BAMSIGNALS_EXCHANGE");
    return want && !strcmp(want, "rccl");
}

}  // namespace

namespace bsig {

struct Exchange {
    std::mutex mu;                      // held by the one call that is using the exchange (ExchangeUse)
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;      // empty: peer copies
    bool peer_ready = false;            // peer access between the devices has been switched on
    bool root_reads = false;            // ... and the first device can address every other one's memory
    ~Exchange()
    {
        for (ncclComm_t c : comms)
            if (c) (void)rccl().CommDestroy(c);
    }
};

namespace {
std::mutex g_ex_mu;
std::vector<std::shared_ptr<Exchange>> g_ex;

#define NCCL_TRY(expr)                                                                           \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess)                                                                   \
            return fail(BSIG_ERR_DEVICE, "RCCL error %d (%s) at %s:%d", (int)r_, rccl().GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// ncclGroupStart .. ncclGroupEnd: a failure of a Send / Recv in between must still close the group -- the depth
// is per THREAD, and a thread that leaves it open runs every later RCCL call (a communicator for another device
// list, say) inside a dangling group, where it hangs or fails.  The guard ends the group on every way out; the
// result of that ncclGroupEnd is only looked at on the good path (end()).
struct GroupGuard {
    bool open = false;
    ncclResult_t start()
    {
        const ncclResult_t r = rccl().GroupStart();
        open = r == ncclSuccess;
        return r;
    }
    ncclResult_t end()
    {
        open = false;
        return rccl().GroupEnd();
    }
    ~GroupGuard() { if (open) (void)rccl().GroupEnd(); }
};

// peer copies go direct over xGMI where the link allows it ("already enabled" is fine)
void enable_peer_access(Exchange &E)
{
    if (E.peer_ready) return;
    const std::vector<int> &devs = E.devices;
    bool root_ok = true;
    for (size_t a = 0; a < devs.size(); ++a)
        for (size_t b = 0; b < devs.size(); ++b) {
            if (devs[a] == devs[b]) continue;
            int can = 0;
            bool on = false;
            if (hipDeviceCanAccessPeer(&can, devs[a], devs[b]) == hipSuccess && can) {
                (void)hipSetDevice(devs[a]);
                const hipError_t e = hipDeviceEnablePeerAccess(devs[b], 0);
                on = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
                (void)hipGetLastError();
            }
            if (a == 0 && !on) root_ok = false;
        }
    E.root_reads = root_ok;
    E.peer_ready = true;
}

// a collective could not be queued over RCCL: give the communicators up (for the life of this exchange)
// and let the caller repeat the step with peer copies
void abandon_rccl(ExchangeUse &use)
{
    Exchange &E = *use.ex;
    for (ncclComm_t c : E.comms)
        if (c) { if (rccl().CommAbort) (void)rccl().CommAbort(c); else (void)rccl().CommDestroy(c); }
    E.comms.clear();
    enable_peer_access(E);
    use.transport = "peer (after an RCCL error)";
}
}  // namespace

int exchange_open(const std::vector<bsig_ctx *> &ctxs, ExchangeUse &use)
{
    use.release();
    std::vector<int> devs;
    for (bsig_ctx *c : ctxs) devs.push_back(c->device);
    std::shared_ptr<Exchange> E;
    {
        std::lock_guard<std::mutex> lk(g_ex_mu);
        for (auto &e : g_ex)
            if (e->devices == devs) E = e;
        if (!E) {
            E = std::make_shared<Exchange>();
            E->devices = devs;
            g_ex.push_back(E);
        }
    }
    // everything below happens under the exchange's own lock: a second thread opening the same device
    // list waits here until the first call has released its use
    std::unique_lock<std::mutex> lock(E->mu);
    if (E->comms.empty() && !E->peer_ready) {          // first use: pick the transport
        bool distinct = true;
        for (size_t a = 0; a < devs.size(); ++a)
            for (size_t b = a + 1; b < devs.size(); ++b) distinct = distinct && devs[a] != devs[b];
        const char *want = getenv("BAMSIGNALS_EXCHANGE");
        const bool no_rccl = want && !strcmp(want, "peer");
        const bool must = want && !strcmp(want, "rccl");
        auto forget = [&]() {
            std::lock_guard<std::mutex> lk(g_ex_mu);
            for (auto it = g_ex.begin(); it != g_ex.end(); ++it)
                if (*it == E) { g_ex.erase(it); break; }
        };
        if (distinct && !no_rccl && rccl().ok) {
            E->comms.assign(devs.size(), nullptr);
            ncclResult_t r = rccl().CommInitAll(E->comms.data(), (int)devs.size(), devs.data());
            std::string why = r == ncclSuccess ? std::string() : std::string(rccl().GetErrorString(r));
            // self-check: every communicator sees all slots and sits at its own slot's rank
            for (size_t k = 0; r == ncclSuccess && k < devs.size() && rccl().CommCount && rccl().CommUserRank; ++k) {
                int cnt = -1, rk = -1;
                if (rccl().CommCount(E->comms[k], &cnt) != ncclSuccess || rccl().CommUserRank(E->comms[k], &rk) != ncclSuccess ||
                    cnt != (int)devs.size() || rk != (int)k) {
                    r = ncclInternalError;
                    why = "communicator " + std::to_string(k) + " reports rank " + std::to_string(rk) + " of " + std::to_string(cnt);
                }
            }
            if (r != ncclSuccess) {
                for (ncclComm_t c : E->comms)
                    if (c) (void)rccl().CommDestroy(c);
                E->comms.clear();
                if (must) {
                    forget();
                    return fail(BSIG_ERR_DEVICE, "RCCL over %zu GPUs failed: %s", devs.size(), why.c_str());
                }
            }
        } else if (must) {
            forget();
            return fail(BSIG_ERR_DEVICE, distinct ? "BAMSIGNALS_EXCHANGE=rccl but librccl could not be loaded"
                                                  : "BAMSIGNALS_EXCHANGE=rccl needs every GPU listed once");
        }
        if (E->comms.empty()) enable_peer_access(*E);
    }
    use.ex = E;
    use.lock = std::move(lock);
    use.streams.clear();
    for (bsig_ctx *c : ctxs) use.streams.push_back(c->stream);
    use.transport = E->comms.empty() ? "peer" : "rccl";
    return BSIG_OK;
}

int exchange_allgather(ExchangeUse &use, const std::vector<uint8_t *> &bufs, const std::vector<size_t> &off,
                       const std::vector<size_t> &len)
{
    if (!use.ex || !use.lock.owns_lock()) return fail(BSIG_ERR_ARG, "exchange_allgather: the exchange is not open");
    Exchange *E = use.ex.get();
    const size_t n = E->devices.size();
    if (bufs.size() != n || off.size() != n || len.size() != n || use.streams.size() != n)
        return fail(BSIG_ERR_ARG, "exchange_allgather: bad shapes");
    if (!E->comms.empty()) {
        auto queue = [&]() -> int {
            GroupGuard grp;
            NCCL_TRY(grp.start());
            for (size_t k = 0; k < n; ++k)
                for (size_t g = 0; g < n; ++g) {
                    if (g == k) continue;
                    if (len[g]) NCCL_TRY(rccl().Recv(bufs[k] + off[g], len[g], ncclUint8, (int)g, E->comms[k], use.streams[k]));
                    if (len[k]) NCCL_TRY(rccl().Send(bufs[k] + off[k], len[k], ncclUint8, (int)g, E->comms[k], use.streams[k]));
                }
            NCCL_TRY(grp.end());
            return BSIG_OK;
        };
        if (queue() == BSIG_OK) return BSIG_OK;
        if (env_wants_rccl()) return BSIG_ERR_DEVICE;          // (the message is queue()'s)
        abandon_rccl(use);
    }
    for (size_t k = 0; k < n; ++k) {
        HIP_TRY(hipSetDevice(E->devices[k]));
        for (size_t g = 0; g < n; ++g)
            if (g != k && len[g])
                HIP_TRY(hipMemcpyPeerAsync(bufs[k] + off[g], E->devices[k], bufs[g] + off[g], E->devices[g], len[g], use.streams[k]));
    }
    return BSIG_OK;
}

int exchange_gather(ExchangeUse &use, const std::vector<const uint8_t *> &src, const std::vector<size_t> &len,
                    uint8_t *dst_root, const std::vector<size_t> &off)
{
    if (!use.ex || !use.lock.owns_lock()) return fail(BSIG_ERR_ARG, "exchange_gather: the exchange is not open");
    Exchange *E = use.ex.get();
    const size_t n = E->devices.size();
    if (src.size() != n || off.size() != n || len.size() != n || use.streams.size() != n)
        return fail(BSIG_ERR_ARG, "exchange_gather: bad shapes");
    HIP_TRY(hipSetDevice(E->devices[0]));
    if (len[0] && src[0] != dst_root + off[0])
        HIP_TRY(hipMemcpyAsync(dst_root + off[0], src[0], len[0], hipMemcpyDeviceToDevice, use.streams[0]));
    if (!E->comms.empty()) {
        // the peers send straight to the root over their own links (ingress 7 links x ~153 GB/s)
        auto queue = [&]() -> int {
            GroupGuard grp;
            NCCL_TRY(grp.start());
            for (size_t k = 1; k < n; ++k) {
                if (!len[k]) continue;
                NCCL_TRY(rccl().Recv(dst_root + off[k], len[k], ncclUint8, (int)k, E->comms[0], use.streams[0]));
                NCCL_TRY(rccl().Send(src[k], len[k], ncclUint8, 0, E->comms[k], use.streams[k]));
            }
            NCCL_TRY(grp.end());
            return BSIG_OK;
        };
        if (queue() == BSIG_OK) return BSIG_OK;
        if (env_wants_rccl()) return BSIG_ERR_DEVICE;
        abandon_rccl(use);
    }
    for (size_t k = 1; k < n; ++k)
        if (len[k]) HIP_TRY(hipMemcpyPeerAsync(dst_root + off[k], E->devices[0], src[k], E->devices[k], len[k], use.streams[0]));
    return BSIG_OK;
}

bool exchange_root_reads_peers(ExchangeUse &use)
{
    if (!use.ex || !use.lock.owns_lock()) return false;
    enable_peer_access(*use.ex);
    return use.ex->root_reads;
}

void exchange_close_all()
{
    std::vector<std::shared_ptr<Exchange>> drop;
    {
        std::lock_guard<std::mutex> lk(g_ex_mu);
        drop.swap(g_ex);
    }
    // (communicators are destroyed here unless a running call still holds its reference: then when it returns)
}

namespace {
__global__ __launch_bounds__(256) void k_place_segments(int64_t n, const int32_t *__restrict__ src,
                                                        const int64_t *__restrict__ src_off, int32_t *__restrict__ dst,
                                                        const int64_t *__restrict__ dst_off, const int64_t *__restrict__ which)
{
    // one wave per segment (4 segments per workgroup): ranges are a few hundred to a few thousand cells.
    // src may be another GPU's memory (read in place over xGMI): 16-B requests where both sides line up.
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t a = src_off[k], len = src_off[k + 1] - a, d = dst_off[which[k]];
    const int32_t *sp = src + a;
    int32_t *dp = dst + d;
    if ((((uintptr_t)sp | (uintptr_t)dp) & 15) == 0) {
        const int64_t nv = len >> 2;
        const int4 *s4 = reinterpret_cast<const int4 *>(sp);
        int4 *d4 = reinterpret_cast<int4 *>(dp);
        for (int64_t i = lane; i < nv; i += 64) d4[i] = s4[i];
        for (int64_t i = (nv << 2) + lane; i < len; i += 64) dp[i] = sp[i];
    } else {
        for (int64_t i = lane; i < len; i += 64) dp[i] = sp[i];
    }
}
}  // namespace

// ---- a narrow wire for result shards (round 5) ----------------------------------------------------------------------
// The cells of a per-base profile are almost all 0, 1 or 2 (the north star's reads: 0.2 per base), and a gather of
// int32 shards into ONE GPU is bounded by that GPU's ingress (7 links x ~55 GB/s against 5 TB/s of local writes):
// 700 MB of 800 travel for an 8-GPU north-star step.  A shard therefore travels as two bits a cell -- the value, or 3 =
// "see the exception list" -- plus a list of (cell, value) pairs for everything else (any int32: negative coverage
// differences, counts of 3 and more):  [0] number of exceptions, [1..3] 0, [4 .. 4 + ceil(n / 16)) codes,
// then `cap` pairs.  Lossless; the sizes are fixed by (n, cap), so every rank's message has the same length.
namespace {
constexpr int kNarrowHead = 4;               // dwords in front of the codes
// One wave codes 1,024 cells a turn: four coalesced 16-byte loads a lane (lane l of load j holds cells 256 j + 4 l ..+3 =
// one byte of codes; four neighbouring lanes' bytes make a word, put together with two DPP-sized shuffles).  Exceptions
// are rare and scattered (one in ~900 cells of a north-star profile), so one atomic on the message's counter per
// exception -- 230,000 on ONE address a launch -- was what the first form of this kernel spent its time on (2.5 ms for
// 800 MB): a wave keeps its exceptions in a list of its own in LDS and claims room in the message for a whole list at
// a time (when the list is half full, and when the wave is done).
constexpr int kNarrowList = 256;             // entries of a wave's list in LDS
__global__ __launch_bounds__(256) void k_narrow_pack(const int32_t *__restrict__ src, int64_t n, uint32_t *__restrict__ msg,
                                                     int64_t n_words, int64_t cap)
{
    __shared__ uint2 lst[4][kNarrowList];
    __shared__ uint32_t cnt[4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint2 *exc = reinterpret_cast<uint2 *>(msg + kNarrowHead + n_words);
    if (lane == 0) cnt[wv] = 0;
    auto flush = [&]() {                     // (wave-uniform: every lane of the wave comes here together)
        const uint32_t have = min(cnt[wv], (uint32_t)kNarrowList);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(reinterpret_cast<unsigned int *>(msg), have);
        base = __builtin_amdgcn_readfirstlane(base);
        for (uint32_t i = lane; i < have; i += 64)
            if ((int64_t)(base + i) < cap) exc[base + i] = lst[wv][i];
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) cnt[wv] = 0;
    };
    const int64_t n_turns = (n + 1023) >> 10;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wv; t < n_turns; t += (int64_t)gridDim.x * 4) {
        const int64_t c0 = t << 10;
        int4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t c = c0 + 256 * j + 4 * lane;
            if (c + 4 <= n) v[j] = *reinterpret_cast<const int4 *>(src + c);      // (src is 16-B aligned, c a multiple of 4)
            else v[j] = make_int4(c < n ? src[c] : 0, c + 1 < n ? src[c + 1] : 0, c + 2 < n ? src[c + 2] : 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t c = c0 + 256 * j + 4 * lane;
            uint32_t byte = 0;
            auto one = [&](int32_t x, int k) {
                const uint32_t code = (uint32_t)x < 3u ? (uint32_t)x : 3u;
                byte |= code << (2 * k);
                if (code == 3u) {
                    const uint32_t slot = atomicAdd(&cnt[wv], 1u);
                    if (slot < (uint32_t)kNarrowList) lst[wv][slot] = make_uint2((uint32_t)(c + k), (uint32_t)x);
                    else {                   // (a turn with more exceptions than the list holds: one by one)
                        const uint32_t g = atomicAdd(reinterpret_cast<unsigned int *>(msg), 1u);
                        if ((int64_t)g < cap) exc[g] = make_uint2((uint32_t)(c + k), (uint32_t)x);
                    }
                }
            };
            one(v[j].x, 0); one(v[j].y, 1); one(v[j].z, 2); one(v[j].w, 3);
            uint32_t two = byte | (uint32_t)__shfl_down((int)byte, 1) << 8;
            two |= (uint32_t)__shfl_down((int)two, 2) << 16;
            const int64_t w = (c0 >> 4) + 16 * j + (lane >> 2);
            if ((lane & 3) == 0 && w < n_words) msg[kNarrowHead + w] = two;
        }
        __builtin_amdgcn_wave_barrier();
        if (cnt[wv] > (uint32_t)kNarrowList / 2) flush();
    }
    if (cnt[wv] > 0) flush();
}

// segment k of the coded source -> dst at dst_off[which[k]] (one wave per segment; exceptions come as 3 and are patched below)
__global__ __launch_bounds__(256) void k_place_narrow(int64_t n, const uint32_t *__restrict__ codes, const int64_t *__restrict__ src_off,
                                                      int32_t *__restrict__ dst, const int64_t *__restrict__ dst_off,
                                                      const int64_t *__restrict__ which)
{
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t a = src_off[k], len = src_off[k + 1] - a;
    int32_t *dp = dst + dst_off[which[k]];
    const int head = (int)((4 - ((uintptr_t)dp >> 2)) & 3);             // cells in front of the first 16-B aligned one
    for (int64_t i = lane; i < head && i < len; i += 64) { const int64_t c = a + i; dp[i] = (int32_t)((codes[c >> 4] >> (2 * (c & 15))) & 3u); }
    const int64_t nv = len > head ? (len - head) >> 2 : 0;
    for (int64_t q = lane; q < nv; q += 64) {
        const int64_t c = a + head + 4 * q;                              // four cells: at most two code words
        // (cells c .. c+3 are eight bits from bit 2 * (c & 15) of the 64 bits "this word, then the word of cell c + 3" --
        // the same word twice unless the four cells straddle two)
        const uint64_t two = (uint64_t)codes[c >> 4] | (uint64_t)codes[(c + 3) >> 4] << 32;
        const uint32_t bits = (uint32_t)(two >> (2u * (uint32_t)(c & 15)));
        reinterpret_cast<int4 *>(dp + head)[q] = make_int4((int)(bits & 3u), (int)((bits >> 2) & 3u), (int)((bits >> 4) & 3u), (int)((bits >> 6) & 3u));
    }
    for (int64_t i = head + 4 * nv + lane; i < len; i += 64) { const int64_t c = a + i; dp[i] = (int32_t)((codes[c >> 4] >> (2 * (c & 15))) & 3u); }
}

// the exceptions of a coded shard, each put at its cell's place: the segment of a source cell by binary search
__global__ __launch_bounds__(256) void k_patch_narrow(const uint32_t *__restrict__ msg, int64_t n_words, int64_t cap, int64_t n_seg,
                                                      const int64_t *__restrict__ src_off, int32_t *__restrict__ dst,
                                                      const int64_t *__restrict__ dst_off, const int64_t *__restrict__ which,
                                                      int *__restrict__ overflow)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n_exc = (int64_t)msg[0];
    if (e == 0 && n_exc > cap) *overflow = 1;                            // (the sender had more than its list holds: the result is wrong)
    if (e >= n_exc || e >= cap) return;
    const uint2 x = reinterpret_cast<const uint2 *>(msg + kNarrowHead + n_words)[e];
    const int64_t c = (int64_t)x.x;
    int64_t lo = 0, hi = n_seg;                                          // the last segment with src_off <= c
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (src_off[mid] <= c) lo = mid; else hi = mid;
    }
    if (c < src_off[lo] || c >= src_off[lo + 1]) return;                // (a cell of the padding: not in any segment)
    dst[dst_off[which[lo]] + (c - src_off[lo])] = (int32_t)x.y;
}
}  // namespace

int64_t narrow_words(int64_t n_cells) { return (n_cells + 15) / 16; }
int64_t narrow_message_bytes(int64_t n_cells, int64_t cap) { return 4 * (kNarrowHead + narrow_words(n_cells) + 2 * std::max<int64_t>(cap, 0)); }

hipError_t launch_narrow_pack(const int32_t *src, int64_t n_cells, void *msg, int64_t cap, hipStream_t st)
{
    const int64_t nw = narrow_words(n_cells);
    hipError_t e = hipMemsetAsync(msg, 0, 4 * kNarrowHead, st);
    if (e != hipSuccess || nw == 0) return e;
    const int64_t turns = (n_cells + 1023) >> 10;                     // a wave's turn = 1,024 cells; 2,048 blocks of 4 waves at most
    hipLaunchKernelGGL(k_narrow_pack, dim3((unsigned)std::min<int64_t>((turns + 3) / 4, 2048)), dim3(256), 0, st, src, n_cells, (uint32_t *)msg, nw,
                       std::max<int64_t>(cap, 0));
    return hipGetLastError();
}

hipError_t launch_place_narrow(int64_t n_seg, const void *msg, int64_t n_cells, int64_t cap, const int64_t *src_off, int32_t *dst,
                               const int64_t *dst_off, const int64_t *which, int *overflow, hipStream_t st)
{
    if (n_seg <= 0) return hipSuccess;
    const int64_t nw = narrow_words(n_cells);
    const uint32_t *m = (const uint32_t *)msg;
    hipLaunchKernelGGL(k_place_narrow, dim3((unsigned)((n_seg + 3) / 4)), dim3(256), 0, st, n_seg, m + kNarrowHead, src_off, dst, dst_off, which);
    hipLaunchKernelGGL(k_patch_narrow, dim3((unsigned)((std::max<int64_t>(cap, 1) + 255) / 256)), dim3(256), 0, st, m, nw, std::max<int64_t>(cap, 0), n_seg,
                       src_off, dst, dst_off, which, overflow);
    return hipGetLastError();
}

namespace { __global__ void k_warm_collect() {} }
hipError_t warm_collect_module(hipStream_t st)
{
    hipLaunchKernelGGL(k_warm_collect, dim3(1), dim3(64), 0, st);
    return hipGetLastError();
}

hipError_t launch_place_segments(int64_t n, const int32_t *src, const int64_t *src_off, int32_t *dst,
                                 const int64_t *dst_off, const int64_t *which, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_place_segments, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, src, src_off, dst, dst_off, which);
    return hipGetLastError();
}

}  // namespace bsig
