// Exchanges between the GPUs one process drives (the single-process multi-GPU route of the
// file-level calls): an all-gather of the decoded column shares, and the gather of the result
// shards to the first GPU.  Transport: RCCL over xGMI (librccl loaded on first use; grouped
// ncclSend/ncclRecv, i.e. every pair of GPUs talks over its own link -- xGMI is point-to-point, so
// a ring would be bound by one link), or plain peer copies (hipMemcpyPeerAsync) where RCCL cannot
// be used (the same GPU listed twice, as the tests do on a 1-GPU box) or is not wanted
// (env BAMSIGNALS_EXCHANGE=peer).
#ifndef BSIG_COLLECT_H
#define BSIG_COLLECT_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <memory>
#include <mutex>
#include <vector>

#include "runtime_internal.h"

namespace bsig {

struct Exchange;     // one per ordered list of devices; lives while the registry or a call refers to it

// A call's hold on an exchange, from exchange_open() until the slots' streams have been synchronised:
//   * the exchange's lock -- RCCL does not allow two threads to interleave group calls on one
//     communicator, and the file-level calls may come from several host threads at once;
//   * a reference that keeps the communicators alive: bsig_cache_clear() only drops the registry's
//     reference, the communicators go when the last call using them has returned;
//   * this call's streams (one per slot), so that nothing call-specific lives in the shared object.
struct ExchangeUse {
    std::shared_ptr<Exchange> ex;
    std::unique_lock<std::mutex> lock;
    std::vector<hipStream_t> streams;
    const char *transport = "";     // "rccl", "peer", or "peer (after an RCCL error)" once a collective fell back
    ~ExchangeUse() { release(); }
    void release() { if (lock.owns_lock()) lock.unlock(); ex.reset(); }
};

// ctxs: one context per slot (its device and the stream the exchange is queued on).  Never fails for
// lack of RCCL: it then uses peer copies (unless env BAMSIGNALS_EXCHANGE=rccl demands RCCL).  A fresh
// communicator set is checked: every communicator must report ctxs.size() ranks and its own slot as rank.
int exchange_open(const std::vector<bsig_ctx *> &ctxs, ExchangeUse &use);
// Every slot k holds a buffer bufs[k] of identical layout; share g (len[g] bytes at off[g]) is valid on
// slot g.  Afterwards every slot holds every share.  Queued on the slots' streams; the shares must be
// complete (streams synchronised) before the call, and the caller synchronises the streams afterwards.
// An RCCL failure while the group is being built abandons RCCL for this exchange (for good) and the
// call is carried out with peer copies instead; use.transport says so.
int exchange_allgather(ExchangeUse &use, const std::vector<uint8_t *> &bufs, const std::vector<size_t> &off,
                       const std::vector<size_t> &len);
// src[k] (len[k] bytes on slot k) -> dst_root + off[k] on slot 0.  Same rules.
int exchange_gather(ExchangeUse &use, const std::vector<const uint8_t *> &src, const std::vector<size_t> &len,
                    uint8_t *dst_root, const std::vector<size_t> &off);
// can a kernel on slot 0's GPU read the other slots' device memory in place (same device, or peer access on)?
bool exchange_root_reads_peers(ExchangeUse &use);
// drops the registry's references (communicators in use by a running call live until it returns)
void exchange_close_all();

// a result shard as two bits a cell + a list of exceptions (collect.hip): message size, packing, placement
int64_t narrow_message_bytes(int64_t n_cells, int64_t cap);
hipError_t launch_narrow_pack(const int32_t *src, int64_t n_cells, void *msg, int64_t cap, hipStream_t st);
hipError_t launch_place_narrow(int64_t n_seg, const void *msg, int64_t n_cells, int64_t cap, const int64_t *src_off, int32_t *dst,
                               const int64_t *dst_off, const int64_t *which, int *overflow, hipStream_t st);
// Segment k of src (src_off[k] .. src_off[k+1]) goes to dst at dst_off[which[k]]: bsig_scatter_segments
// on the device (all pointers are device pointers), so that gathered shards are put into the caller's
// range order in HBM and leave for the host in ONE copy.
hipError_t launch_place_segments(int64_t n, const int32_t *src, const int64_t *src_off, int32_t *dst,
                                 const int64_t *dst_off, const int64_t *which, hipStream_t st);

}  // namespace bsig
#endif
