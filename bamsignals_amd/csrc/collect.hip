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
