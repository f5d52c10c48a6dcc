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

#include <vector>

#include "runtime_internal.h"

namespace bsig {

struct Exchange;     // one per ordered list of devices, kept for the life of the process (or until exchange_close_all)

// ctxs: one context per slot (its device and the stream the exchange is queued on).  Never fails for
// lack of RCCL: it then uses peer copies.  *transport receives "rccl" or "peer".
int exchange_open(const std::vector<bsig_ctx *> &ctxs, Exchange **ex, const char **transport);
// Every slot k holds a buffer bufs[k] of identical layout; share g (len[g] bytes at off[g]) is valid on
// slot g.  Afterwards every slot holds every share.  Queued on the slots' streams; the shares must be
// complete (streams synchronised) before the call, and the caller synchronises the streams afterwards.
int exchange_allgather(Exchange *ex, const std::vector<uint8_t *> &bufs, const std::vector<size_t> &off,
                       const std::vector<size_t> &len);
// src[k] (len[k] bytes on slot k) -> dst_root + off[k] on slot 0.  Same synchronisation rules.
int exchange_gather(Exchange *ex, const std::vector<const uint8_t *> &src, const std::vector<size_t> &len,
                    uint8_t *dst_root, const std::vector<size_t> &off);
void exchange_close_all();

// Segment k of src (src_off[k] .. src_off[k+1]) goes to dst at dst_off[which[k]]: bsig_scatter_segments
// on the device (all pointers are device pointers), so that gathered shards are put into the caller's
// range order in HBM and leave for the host in ONE copy.
hipError_t launch_place_segments(int64_t n, const int32_t *src, const int64_t *src_off, int32_t *dst,
                                 const int64_t *dst_off, const int64_t *which, hipStream_t st);

}  // namespace bsig
#endif
