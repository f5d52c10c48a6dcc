// See collect.h.  RCCL is loaded with dlopen on first use, so that single-GPU callers never pay for
// (or depend on) librccl; its types come from <rccl/rccl.h>.
#include "collect.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>

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
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString;
    }
};
Rccl &rccl()
{
    static Rccl r;
    return r;
}

}  // namespace

namespace bsig {

struct Exchange {
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;      // empty: peer copies
};

namespace {
std::mutex g_ex_mu;
std::vector<std::unique_ptr<Exchange>> g_ex;

#define NCCL_TRY(expr)                                                                           \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess)                                                                   \
            return fail(BSIG_ERR_DEVICE, "RCCL error %d (%s) at %s:%d", (int)r_, rccl().GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)
}  // namespace

int exchange_open(const std::vector<bsig_ctx *> &ctxs, Exchange **out, const char **transport)
{
    std::lock_guard<std::mutex> lk(g_ex_mu);
    std::vector<int> devs;
    for (bsig_ctx *c : ctxs) devs.push_back(c->device);
    Exchange *E = nullptr;
    for (auto &e : g_ex)
        if (e->devices == devs) E = e.get();
    if (!E) {
        g_ex.emplace_back(new Exchange);
        E = g_ex.back().get();
        E->devices = devs;
        bool distinct = true;
        for (size_t a = 0; a < devs.size(); ++a)
            for (size_t b = a + 1; b < devs.size(); ++b) distinct = distinct && devs[a] != devs[b];
        const char *want = getenv("BAMSIGNALS_EXCHANGE");
        const bool no_rccl = want && !strcmp(want, "peer");
        if (distinct && !no_rccl && rccl().ok) {
            E->comms.assign(devs.size(), nullptr);
            const ncclResult_t r = rccl().CommInitAll(E->comms.data(), (int)devs.size(), devs.data());
            if (r != ncclSuccess) {
                E->comms.clear();
                if (want && !strcmp(want, "rccl")) {
                    g_ex.pop_back();
                    return fail(BSIG_ERR_DEVICE, "ncclCommInitAll over %zu GPUs failed: %s", devs.size(), rccl().GetErrorString(r));
                }
            }
        } else if (want && !strcmp(want, "rccl")) {
            g_ex.pop_back();
            return fail(BSIG_ERR_DEVICE, distinct ? "BAMSIGNALS_EXCHANGE=rccl but librccl could not be loaded"
                                                  : "BAMSIGNALS_EXCHANGE=rccl needs every GPU listed once");
        }
        if (E->comms.empty()) {
            // peer copies: direct over xGMI where the link allows it ("already enabled" is fine)
            for (size_t a = 0; a < devs.size(); ++a)
                for (size_t b = 0; b < devs.size(); ++b) {
                    if (devs[a] == devs[b]) continue;
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, devs[a], devs[b]) == hipSuccess && can) {
                        (void)hipSetDevice(devs[a]);
                        (void)hipDeviceEnablePeerAccess(devs[b], 0);
                        (void)hipGetLastError();
                    }
                }
        }
    }
    E->streams.clear();
    for (bsig_ctx *c : ctxs) E->streams.push_back(c->stream);
    *out = E;
    if (transport) *transport = E->comms.empty() ? "peer" : "rccl";
    return BSIG_OK;
}

int exchange_allgather(Exchange *E, const std::vector<uint8_t *> &bufs, const std::vector<size_t> &off,
                       const std::vector<size_t> &len)
{
    const size_t n = E->devices.size();
    if (bufs.size() != n || off.size() != n || len.size() != n) return fail(BSIG_ERR_ARG, "exchange_allgather: bad shapes");
    if (!E->comms.empty()) {
        NCCL_TRY(rccl().GroupStart());
        for (size_t k = 0; k < n; ++k)
            for (size_t g = 0; g < n; ++g) {
                if (g == k) continue;
                if (len[g]) NCCL_TRY(rccl().Recv(bufs[k] + off[g], len[g], ncclUint8, (int)g, E->comms[k], E->streams[k]));
                if (len[k]) NCCL_TRY(rccl().Send(bufs[k] + off[k], len[k], ncclUint8, (int)g, E->comms[k], E->streams[k]));
            }
        NCCL_TRY(rccl().GroupEnd());
        return BSIG_OK;
    }
    for (size_t k = 0; k < n; ++k) {
        HIP_TRY(hipSetDevice(E->devices[k]));
        for (size_t g = 0; g < n; ++g)
            if (g != k && len[g])
                HIP_TRY(hipMemcpyPeerAsync(bufs[k] + off[g], E->devices[k], bufs[g] + off[g], E->devices[g], len[g], E->streams[k]));
    }
    return BSIG_OK;
}

int exchange_gather(Exchange *E, const std::vector<const uint8_t *> &src, const std::vector<size_t> &len,
                    uint8_t *dst_root, const std::vector<size_t> &off)
{
    const size_t n = E->devices.size();
    if (src.size() != n || off.size() != n || len.size() != n) return fail(BSIG_ERR_ARG, "exchange_gather: bad shapes");
    HIP_TRY(hipSetDevice(E->devices[0]));
    if (len[0]) HIP_TRY(hipMemcpyAsync(dst_root + off[0], src[0], len[0], hipMemcpyDeviceToDevice, E->streams[0]));
    if (!E->comms.empty()) {
        // the peers send straight to the root over their own links (ingress 7 links x ~153 GB/s)
        NCCL_TRY(rccl().GroupStart());
        for (size_t k = 1; k < n; ++k) {
            if (!len[k]) continue;
            NCCL_TRY(rccl().Recv(dst_root + off[k], len[k], ncclUint8, (int)k, E->comms[0], E->streams[0]));
            NCCL_TRY(rccl().Send(src[k], len[k], ncclUint8, 0, E->comms[k], E->streams[k]));
        }
        NCCL_TRY(rccl().GroupEnd());
        return BSIG_OK;
    }
    for (size_t k = 1; k < n; ++k)
        if (len[k]) HIP_TRY(hipMemcpyPeerAsync(dst_root + off[k], E->devices[0], src[k], E->devices[k], len[k], E->streams[0]));
    return BSIG_OK;
}

void exchange_close_all()
{
    std::lock_guard<std::mutex> lk(g_ex_mu);
    for (auto &e : g_ex)
        for (ncclComm_t c : e->comms)
            if (c) (void)rccl().CommDestroy(c);
    g_ex.clear();
}

namespace {
__global__ __launch_bounds__(256) void k_place_segments(int64_t n, const int32_t *__restrict__ src,
                                                        const int64_t *__restrict__ src_off, int32_t *__restrict__ dst,
                                                        const int64_t *__restrict__ dst_off, const int64_t *__restrict__ which)
{
    // one wave per segment (4 segments per workgroup): ranges are a few hundred to a few thousand cells
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t a = src_off[k], len = src_off[k + 1] - a, d = dst_off[which[k]];
    for (int64_t i = lane; i < len; i += 64) dst[d + i] = src[a + i];
}
}  // namespace

hipError_t launch_place_segments(int64_t n, const int32_t *src, const int64_t *src_off, int32_t *dst,
                                 const int64_t *dst_off, const int64_t *which, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_place_segments, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, src, src_off, dst, dst_off, which);
    return hipGetLastError();
}

}  // namespace bsig
