// Test-only host wrapper around the per-lane DEFLATE decoder of the GPU inflate kernel
// (bamsignals_amd/csrc/inflate_lane.h compiles for the host as well): tests/test_inflate_lane.py
// builds it with g++ and checks it against zlib.
#include "../bamsignals_amd/csrc/inflate_lane.h"

extern "C" int inflate_lane_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len)
{
    bsig_inflate::LaneTables tables;
    alignas(8) uint8_t lens[bsig_inflate::kLensBytes];
    return bsig_inflate::inflate_block(in, in_len, out, out_len, tables, lens);
}
