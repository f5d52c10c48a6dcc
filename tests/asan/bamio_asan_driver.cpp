// Test-only driver for the CPU-side parsers of untrusted input, built with
// -fsanitize=address,undefined (make -C bamsignals_amd/csrc asan; tests/test_sanitizers.py):
// the BGZF / BAM / BAI reader and the BAM writer + SAM parser (bamsignals_amd/csrc/bamio.cpp), and
// the host build of the per-lane DEFLATE decoder of the GPU inflate kernel (csrc/inflate_lane.h).
// A malformed input must end in a clean error (exit code 3) or a successful parse (0); anything the
// sanitizers find aborts the process with another code.
//
//   driver decode <bam>            header, <bam>.bai, whole-file decode (1 and 3 threads), region decode
//   driver sam2bam <sam> <outbam>  text SAM -> BAM + BAI, then decodes what it wrote
//   driver inflate <raw> <isize>   one raw DEFLATE stream through inflate_lane.h into isize bytes
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../bamsignals_amd/csrc/bamio.h"
#include "../../bamsignals_amd/csrc/host_util.h"
#include "../../bamsignals_amd/csrc/inflate_lane.h"

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

static int report(int rc, const char *what)
{
    if (rc) { printf("err %d at %s: %s\n", rc, what, bsig::g_last_error.c_str()); return 3; }
    return 0;
}

static int decode(const std::string &path)
{
    bsig::BamHeader hdr;
    int rc = bsig::bam_read_header(path, hdr);
    if (rc) return report(rc, "header");
    bsig::BaiIndex idx;
    rc = bsig::bai_load(path + ".bai", idx);
    const bool have_idx = rc == 0;
    long long sum = 0;
    for (int threads : {1, 3}) {
        bsig::BamHeader h;
        bsig::HostColumns cols;
        rc = bsig::bam_decode_all(path, threads, h, cols);
        if (rc) return report(rc, "decode_all");
        // touch everything the decode produced
        for (size_t i = 0; i < (size_t)cols.size(); ++i) sum += cols.pos[i] + cols.flag[i] + cols.mapq[i] + cols.tlen[i];
        for (uint32_t c : cols.cigar) sum += c;
        if (cols.cigar_off.size() != (size_t)cols.size() + 1) { printf("bad cigar_off\n"); return 4; }
    }
    if (have_idx) {
        std::vector<bsig::Region> rg;
        for (int r = 0; r < (int)hdr.names.size() && r < 4; ++r) {
            rg.push_back(bsig::Region{r, 0, 1000});
            rg.push_back(bsig::Region{r, 5000, 5100});
            rg.push_back(bsig::Region{r, (int64_t)hdr.lens[(size_t)r] - 50, (int64_t)hdr.lens[(size_t)r] + 50});
            rg.push_back(bsig::Region{r, -100, 1ll << 29});
        }
        bsig::BamHeader h;
        bsig::HostColumns cols;
        rc = bsig::bam_decode_regions(path, idx, rg, 2, h, cols);
        if (rc) return report(rc, "decode_regions");
        for (size_t i = 0; i < (size_t)cols.size(); ++i) sum += cols.pos[i];
    }
    printf("ok %lld\n", sum);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && !strcmp(argv[1], "decode")) return decode(argv[2]);
    if (argc >= 4 && !strcmp(argv[1], "sam2bam")) {
        const int rc = bsig::sam_to_bam_and_index(argv[2], argv[3]);
        if (rc) return report(rc, "sam2bam");
        return decode(argv[3]);
    }
    if (argc >= 4 && !strcmp(argv[1], "inflate")) {
        FILE *f = fopen(argv[2], "rb");
        if (!f) return 2;
        std::vector<uint8_t> in;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) in.insert(in.end(), buf, buf + n);
        fclose(f);
        const uint32_t isize = (uint32_t)atol(argv[3]);
        // exactly isize bytes of output and exactly in.size() bytes of input: an over-read or an
        // over-store is a heap overflow the sanitizer sees
        std::vector<uint8_t> out(isize);
        std::vector<uint8_t> exact(in);
        bsig_inflate::LaneTables tables;
        alignas(8) uint8_t lens[bsig_inflate::kLensBytes];
        const int rc = bsig_inflate::inflate_block(exact.data(), (uint32_t)exact.size(), out.data(), isize, tables, lens);
        unsigned long sum = 0;
        if (!rc) for (uint8_t b : out) sum += b;
        printf("%s %d %lu\n", rc ? "err" : "ok", rc, sum);
        return rc ? 3 : 0;
    }
    fprintf(stderr, "usage: driver decode <bam> | sam2bam <sam> <out> | inflate <raw> <isize>\n");
    return 2;
}
