// Internals shared by the host runtime (runtime.hip) and the device-side BAM decode (devdecode.hip).
#ifndef BSIG_RUNTIME_INTERNAL_H
#define BSIG_RUNTIME_INTERNAL_H
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/bamsignals_abi.h"
#include "bsig_types.h"
#include "host_util.h"

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? BSIG_ERR_NOMEM : BSIG_ERR_DEVICE,          \
                        "HIP error %d (%s) at %s:%d: %s", (int)e_, hipGetErrorString(e_),     \
                        __FILE__, __LINE__, #expr);                                            \
    } while (0)

// owns a set of device allocations
struct DevPool {
    std::vector<void *> ptrs;
    int64_t bytes = 0;
    template <typename T>
    hipError_t alloc(T **p, size_t count)
    {
        void *q = nullptr;
        const size_t nbytes = std::max<size_t>(count * sizeof(T), 16);
        hipError_t e = hipMalloc(&q, nbytes);
        if (e != hipSuccess) { *p = nullptr; return e; }
        ptrs.push_back(q);
        bytes += (int64_t)nbytes;
        *p = (T *)q;
        return hipSuccess;
    }
    void release()
    {
        for (void *p : ptrs) (void)hipFree(p);
        ptrs.clear();
        bytes = 0;
    }
    ~DevPool() { release(); }
};

struct bsig_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
};

struct bsig_reads {
    bsig_ctx *ctx = nullptr;
    BsigReadsDev dev{};
    DevPool pool;
    bsig_reads_info info{};
    int32_t n_ref = 0;
    uint64_t col_cap[BSIG_MAX_CLASSES] = {0, 0, 0, 0};      // elements allocated per class column
    uint64_t idx_entries[BSIG_MAX_CLASSES] = {0, 0, 0, 0};  // entries allocated per class index
    std::vector<uint32_t> ref_unit0, ref_units;
    std::vector<int32_t> ref_len;
};

namespace bsig {
struct BaiIndex;
// Builds the resident HBM layout of R (span classes + bucket indexes, bsig_types.h) from device
// columns of n reads in BAM order; ref_off is a host array of n_ref + 1 entries.  The input
// columns are only read.
int layout_from_device(bsig_ctx *ctx, bsig_reads *R, int64_t n, int32_t n_ref, const int32_t *ref_len,
                       const int64_t *ref_off, const int32_t *d_pos, const int32_t *d_end,
                       const uint16_t *d_flag, const uint8_t *d_mapq, const int32_t *d_tlen);
// hands the device-side decode's cached scratch back to the driver (devdecode.hip)
void release_decode_scratch();
// > 0: the file (or this build's limits) needs another decode path; nothing was allocated
constexpr int kNeedsCpuPath = 1;
// Whole BAM -> resident reads on every listed GPU: each GPU decodes one share of the BGZF blocks, the
// column shares are all-gathered over xGMI (devdecode.hip).  BSIG_OK, kNeedsCpuPath or an error.
int reads_from_bam_sharded(const std::vector<bsig_ctx *> &ctxs, const std::string &path, int threads,
                           std::vector<bsig_reads *> &out, const char **transport);
// Index-driven decode with several GPUs: the merged BAI islands of the regions [beg, end) are dealt to the
// GPUs in contiguous runs, the column shares all-gathered (devdecode.hip).  BSIG_OK, kNeedsCpuPath or an error.
int reads_from_regions_sharded(const std::vector<bsig_ctx *> &ctxs, const std::string &path, const BaiIndex &idx, int64_t n_regions,
                               const int32_t *rid, const int64_t *beg, const int64_t *end, int threads,
                               std::vector<bsig_reads *> &out, const char **transport);
// a device buffer to (pageable or page-locked) host memory, staged through page-locked halves where
// that is faster (runtime.hip)
int download_to_host(bsig_ctx *ctx, const void *src_dev, void *dst_host, size_t bytes);
}  // namespace bsig
#endif
