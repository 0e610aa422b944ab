// afq_host.cpp — host side around the quant hot path (include/afquant_host.h): collated-RAD front-end
// (prelude, chunk table, snappy frame format), transcript-to-gene map, and the output writers.
//
// Mirrors, around the per-cell loop that lives on the device (paths relative to /root/reference):
//   quantify / do_quantify_dispatch      src/quant.rs:359-396, 1955-2029   (collate.json, .rad vs .rad.sz)
//   prelude + file tags                  witnessed by src/convert.rs:254-398 (writer side)
//   parse_tg_map                         src/utils.rs:487-662
//   --quant-subset filter                src/quant.rs:1523-1536, 1773-1784; src/utils.rs:1074-1095
//   per-cell stats + featureDump row     src/quant.rs:1150-1262
//   cols / rows / mtx / quant.json       src/quant.rs:1596-1613, 1786-1847, 1913-1933
//   worker fan-out                       src/quant.rs:1553-1575, 1678-1765 -> one afq_ctx + host thread per device (--devices),
//                                        byte-balanced contiguous cell ranges, rows gathered in cell order
//   multi-barcode (10x Flex) records     src/quant.rs:1354-1373, 2003-2027, 1217-1262
//   -d / -b outputs                      src/quant.rs:229-355, 1850-1877
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <sstream>
#include <memory>
#include <mutex>
#include <string>
#include <chrono>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <cerrno>
#include <map>
#include <unordered_map>
#include <zlib.h>
#include <unordered_set>
#include <vector>

#include "../../include/afquant.h"
#include "../../include/afquant_host.h"
#include "afq_hooks.h"

namespace {

thread_local std::string g_herr;
int hfail(int code, const std::string& m) { g_herr = m; return code; }

// ---------------------------------------------------------------------------
// Rust `{}` for f32: shortest digits that round-trip, positional notation, "NaN" / "inf" / "-inf".
int format_f32(float v, char* buf, size_t cap) {
    if (cap < 64) return -1;
    if (std::isnan(v)) { std::memcpy(buf, "NaN", 4); return 3; }
    if (std::isinf(v)) { const char* s = v < 0 ? "-inf" : "inf"; size_t n = std::strlen(s); std::memcpy(buf, s, n + 1); return (int)n; }
    if (v >= 0.0f && v < 16777216.0f && v == (float)(uint32_t)v && !(v == 0.0f && std::signbit(v))) {  // whole counts: the common case
        char tmp[12]; int k = 0; uint32_t u = (uint32_t)v;
        do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
        for (int i = 0; i < k; ++i) buf[i] = tmp[k - 1 - i];
        buf[k] = 0;
        return k;
    }
    // shortest round-trip digits come from the scientific form; Rust then lays them out positionally
    char sci[48];
    auto r = std::to_chars(sci, sci + sizeof sci - 1, v, std::chars_format::scientific);
    *r.ptr = 0;
    std::string digits; int exp10 = 0; bool neg = false;
    const char* p = sci;
    if (*p == '-') { neg = true; ++p; }
    for (; *p && *p != 'e'; ++p) if (*p != '.') digits.push_back(*p);
    if (*p == 'e') exp10 = std::atoi(p + 1);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    std::string out;
    if (neg) out.push_back('-');
    if (digits == "0") out += "0";
    else if (exp10 >= 0) {
        const size_t int_len = (size_t)exp10 + 1;
        if (digits.size() <= int_len) { out += digits; out.append(int_len - digits.size(), '0'); }
        else { out.append(digits, 0, int_len); out.push_back('.'); out.append(digits, int_len, std::string::npos); }
    } else {
        out += "0.";
        out.append((size_t)(-exp10 - 1), '0');
        out += digits;
    }
    if (out.size() + 1 > cap) return -1;
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}

// ---------------------------------------------------------------------------
// snappy: raw block decompress + frame format (stream identifier 0xff, compressed 0x00, uncompressed 0x01,
// padding 0xfe, reserved skippable 0x80-0xfd).  CRC-32C of the uncompressed data is verified (masked).
// CRC-32C (Castagnoli), slicing-by-8
uint32_t crc32c(const uint8_t* p, size_t n) {
    static uint32_t T[8][256];
    static std::once_flag once;
    std::call_once(once, []() {
        for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1; T[0][i] = c; }
        for (uint32_t i = 0; i < 256; ++i) for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFF];
    });
    uint32_t c = ~0u;
    while (n >= 8) {
        uint64_t w; std::memcpy(&w, p, 8);
        w ^= c;
        c = T[7][w & 0xFF] ^ T[6][(w >> 8) & 0xFF] ^ T[5][(w >> 16) & 0xFF] ^ T[4][(w >> 24) & 0xFF] ^
            T[3][(w >> 32) & 0xFF] ^ T[2][(w >> 40) & 0xFF] ^ T[1][(w >> 48) & 0xFF] ^ T[0][(w >> 56) & 0xFF];
        p += 8; n -= 8;
    }
    for (size_t i = 0; i < n; ++i) c = T[0][(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return ~c;
}
uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

// uncompressed length of a raw snappy block (leading uvarint); returns false on a malformed prefix
bool snappy_raw_length(const uint8_t* in, size_t n, uint64_t& ulen, size_t& hdr) {
    ulen = 0; int shift = 0; size_t p = 0;
    for (;;) { if (p >= n || shift > 35) return false; const uint8_t b = in[p++]; ulen |= (uint64_t)(b & 0x7F) << shift; if (!(b & 0x80)) break; shift += 7; }
    hdr = p;
    return true;
}
// raw snappy block -> exactly ulen bytes at dst
bool snappy_raw_decompress_into(const uint8_t* in, size_t n, uint8_t* dst, uint64_t ulen) {
    uint64_t declared; size_t p;
    if (!snappy_raw_length(in, n, declared, p) || declared != ulen) return false;
    uint64_t w = 0;
    while (p < n) {
        const uint8_t tag = in[p++];
        const uint32_t type = tag & 3;
        if (type == 0) {  // literal
            size_t len = (tag >> 2) + 1;
            if (len > 60) { const size_t nb = len - 60; if (p + nb > n) return false; len = 0; for (size_t i = 0; i < nb; ++i) len |= (size_t)in[p + i] << (8 * i); len += 1; p += nb; }
            if (p + len > n || w + len > ulen) return false;
            std::memcpy(dst + w, in + p, len);
            p += len; w += len;
        } else {
            size_t len, off;
            if (type == 1) { if (p + 1 > n) return false; len = ((tag >> 2) & 7) + 4; off = ((size_t)(tag >> 5) << 8) | in[p]; p += 1; }
            else if (type == 2) { if (p + 2 > n) return false; len = (tag >> 2) + 1; off = in[p] | ((size_t)in[p + 1] << 8); p += 2; }
            else { if (p + 4 > n) return false; len = (tag >> 2) + 1; off = in[p] | ((size_t)in[p + 1] << 8) | ((size_t)in[p + 2] << 16) | ((size_t)in[p + 3] << 24); p += 4; }
            if (off == 0 || off > w || w + len > ulen) return false;
            if (off >= len) std::memcpy(dst + w, dst + w - off, len);
            else for (size_t i = 0; i < len; ++i) dst[w + i] = dst[w - off + i];  // overlaps its own output (run-length)
            w += len;
        }
    }
    return w == ulen;
}

// Snappy frame format.  The data chunks of a frame are independent (<= 64 KiB of output each), so the stream is
// planned in one cheap pass over the chunk headers and decoded by several threads.
struct SnappyChunk { size_t in_off, in_len; uint64_t out_off, ulen; uint32_t crc; bool compressed; };
bool snappy_frame_plan(const uint8_t* in, size_t n, std::vector<SnappyChunk>& chunks, uint64_t& total, std::string& err) {
    size_t p = 0;
    total = 0;
    bool first = true;
    while (p < n) {
        if (p + 4 > n) { err = "truncated snappy frame header"; return false; }
        if (first && in[p] != 0xff) { err = "snappy stream does not start with its stream identifier"; return false; }   // as snap::read::FrameDecoder
        first = false;
        const uint8_t type = in[p];
        const size_t len = in[p + 1] | ((size_t)in[p + 2] << 8) | ((size_t)in[p + 3] << 16);
        p += 4;
        if (p + len > n) { err = "truncated snappy frame chunk"; return false; }
        if (type == 0xff) { if (len != 6 || std::memcmp(in + p, "sNaPpY", 6) != 0) { err = "bad snappy stream identifier"; return false; } }
        else if (type == 0x00 || type == 0x01) {
            if (len < 4) { err = "snappy chunk too short"; return false; }
            SnappyChunk c;
            c.crc = in[p] | ((uint32_t)in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16) | ((uint32_t)in[p + 3] << 24);
            c.in_off = p + 4; c.in_len = len - 4; c.out_off = total; c.compressed = type == 0x00;
            if (c.compressed) { size_t h; if (!snappy_raw_length(in + c.in_off, c.in_len, c.ulen, h)) { err = "corrupt snappy block"; return false; } }
            else c.ulen = c.in_len;
            if (c.ulen > 65536) { err = "snappy chunk larger than the frame format's 65536-byte limit"; return false; }
            total += c.ulen;
            chunks.push_back(c);
        } else if (type >= 0x02 && type <= 0x7f) { err = "unskippable reserved snappy chunk"; return false; }
        // 0x80..0xfe: skippable / padding
        p += len;
    }
    return true;
}
bool snappy_frame_run(const uint8_t* in, const std::vector<SnappyChunk>& chunks, uint8_t* out, unsigned nthreads, std::string& err) {
    nthreads = std::max(1u, std::min<unsigned>(nthreads, 64u));
    if (chunks.size() < 64) nthreads = 1;
    std::vector<int> bad(nthreads, 0);   // 1 = corrupt block, 2 = CRC mismatch
    auto work = [&](unsigned t) {
        const size_t a = chunks.size() * t / nthreads, b = chunks.size() * (t + 1) / nthreads;
        for (size_t i = a; i < b && !bad[t]; ++i) {
            const SnappyChunk& c = chunks[i];
            if (!c.compressed) std::memcpy(out + c.out_off, in + c.in_off, c.ulen);
            else if (!snappy_raw_decompress_into(in + c.in_off, c.in_len, out + c.out_off, c.ulen)) { bad[t] = 1; break; }
            if (mask_crc(crc32c(out + c.out_off, c.ulen)) != c.crc) bad[t] = 2;
        }
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nthreads; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    for (int b : bad) if (b) { err = b == 1 ? "corrupt snappy block" : "snappy CRC mismatch"; return false; }
    return true;
}
bool snappy_frame_decode(const uint8_t* in, size_t n, std::vector<uint8_t>& out, std::string& err, unsigned nthreads = 1) {
    std::vector<SnappyChunk> chunks;
    uint64_t total = 0;
    if (!snappy_frame_plan(in, n, chunks, total, err)) return false;
    out.resize(total);
    return snappy_frame_run(in, chunks, out.data(), nthreads, err);
}

// ---------------------------------------------------------------------------
// RAD prelude
struct Cursor {
    const uint8_t* b; size_t n, p = 0; bool ok = true;
    template <class T> T get() { T v{}; if (p + sizeof(T) > n) { ok = false; return v; } std::memcpy(&v, b + p, sizeof(T)); p += sizeof(T); return v; }
    std::string str(size_t len) { if (p + len > n) { ok = false; return {}; } std::string s((const char*)b + p, len); p += len; return s; }
};
struct TagDesc { std::string name; uint8_t type = 0, len_type = 0, elem_type = 0; };
size_t int_type_bytes(uint8_t t) { return t == 1 ? 1 : t == 2 ? 2 : t == 3 ? 4 : t == 4 ? 8 : 0; }

struct RadPrelude {
    bool is_paired = false; uint64_t ref_count = 0, num_chunks = 0;
    std::vector<std::string> ref_names;
    std::vector<TagDesc> file_tags, read_tags, aln_tags;
    std::unordered_map<std::string, uint64_t> file_tag_vals;
    size_t first_chunk = 0;
    uint32_t bc_bytes = 0, umi_bytes = 0;
};

bool read_tag_section(Cursor& c, std::vector<TagDesc>& tags) {
    const uint16_t nt = c.get<uint16_t>();
    for (uint16_t i = 0; i < nt && c.ok; ++i) {
        TagDesc t;
        const uint16_t nl = c.get<uint16_t>();
        t.name = c.str(nl);
        t.type = c.get<uint8_t>();
        if (t.type == 7) { t.len_type = c.get<uint8_t>(); t.elem_type = c.get<uint8_t>(); }
        tags.push_back(t);
    }
    return c.ok;
}

int parse_prelude(const uint8_t* bytes, size_t n, RadPrelude& P, bool want_names) {
    Cursor c{bytes, n};
    P.is_paired = c.get<uint8_t>() != 0;
    P.ref_count = c.get<uint64_t>();
    if (!c.ok || P.ref_count > n) return hfail(AFQ_ERR_BAD_INPUT, "RAD header: bad ref_count");
    for (uint64_t i = 0; i < P.ref_count && c.ok; ++i) {
        const uint16_t nl = c.get<uint16_t>();
        if (want_names) P.ref_names.push_back(c.str(nl)); else { if (c.p + nl > c.n) c.ok = false; c.p += nl; }
    }
    P.num_chunks = c.get<uint64_t>();
    if (!c.ok) return hfail(AFQ_ERR_BAD_INPUT, "RAD header truncated");
    if (!read_tag_section(c, P.file_tags) || !read_tag_section(c, P.read_tags) || !read_tag_section(c, P.aln_tags))
        return hfail(AFQ_ERR_BAD_INPUT, "RAD tag sections truncated");
    for (auto& t : P.file_tags) {  // values of the file-level tags follow, in declaration order
        uint64_t v = 0;
        switch (t.type) {
            case 0: case 1: v = c.get<uint8_t>(); break;
            case 2: v = c.get<uint16_t>(); break;
            case 3: v = c.get<uint32_t>(); break;
            case 4: v = c.get<uint64_t>(); break;
            case 5: c.get<float>(); break;
            case 6: c.get<double>(); break;
            case 7: {
                uint64_t len = 0;
                switch (t.len_type) { case 1: len = c.get<uint8_t>(); break; case 2: len = c.get<uint16_t>(); break; case 3: len = c.get<uint32_t>(); break; case 4: len = c.get<uint64_t>(); break; default: c.ok = false; }
                const size_t eb = t.elem_type == 5 ? 4 : t.elem_type == 6 ? 8 : t.elem_type == 0 ? 1 : int_type_bytes(t.elem_type);
                if (!eb || c.p + len * eb > c.n) c.ok = false; else c.p += len * eb;
                break;
            }
            case 8: { const uint16_t sl = c.get<uint16_t>(); c.str(sl); break; }
            default: c.ok = false;
        }
        P.file_tag_vals[t.name] = v;
    }
    if (!c.ok) return hfail(AFQ_ERR_BAD_INPUT, "RAD file-tag values truncated or of unknown type");
    P.first_chunk = c.p;
    for (auto& t : P.read_tags) {
        if (t.name == "b") P.bc_bytes = (uint32_t)int_type_bytes(t.type);
        if (t.name == "u") P.umi_bytes = (uint32_t)int_type_bytes(t.type);
    }
    return 0;
}

// ---------------------------------------------------------------------------
bool file_exists(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }
bool read_file(const std::string& p, std::vector<uint8_t>& out) {
    FILE* f = std::fopen(p.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END); long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    out.resize((size_t)sz);
    bool ok = sz == 0 || std::fread(out.data(), 1, (size_t)sz, f) == (size_t)sz;
    std::fclose(f);
    return ok;
}
// read-only mapping of a (large) input file: the collated RAD is handed to afq_submit straight out of the page cache
struct MappedFile {
    const uint8_t* p = nullptr; size_t n = 0; int fd = -1;
    bool open(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (::fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) return true;
        void* m = ::mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        ::madvise(m, n, MADV_SEQUENTIAL);
        p = static_cast<const uint8_t*>(m);
        return true;
    }
    ~MappedFile() { if (p) ::munmap(const_cast<uint8_t*>(p), n); if (fd >= 0) ::close(fd); }
};
struct PhaseClock {
    bool on = std::getenv("AFQ_HOST_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[afquant] %-34s %8.3f s\n", what, std::chrono::duration<double>(n - t).count());
        t = n;
    }
};
// decimal digits of v into p, returns the end
inline char* put_u64(char* p, unsigned long long v) {
    char tmp[24]; int k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) *p++ = tmp[--k];
    return p;
}
// 0 when the directory exists afterwards
int mkdirs_checked(const std::string& p) {
    std::string cur;
    for (size_t i = 0; i <= p.size(); ++i) {
        if (i == p.size() || p[i] == '/') { if (!cur.empty()) ::mkdir(cur.c_str(), 0755); }
        if (i < p.size()) cur.push_back(p[i]);
    }
    struct stat st;
    return (::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) ? 0 : -1;
}
std::string json_escape(const std::string& s) {
    std::string o;
    for (char ch : s) { if (ch == '"' || ch == '\\') { o.push_back('\\'); o.push_back(ch); } else if (ch == '\n') o += "\\n"; else o.push_back(ch); }
    return o;
}
std::string bc_to_string(uint64_t bc, uint32_t len) {  // needletail bitmer_to_bytes: first base = most significant pair
    std::string s(len, 'A');
    for (uint32_t i = 0; i < len; ++i) s[i] = "ACGT"[(bc >> (2 * (len - 1 - i))) & 3];
    return s;
}
bool string_to_bc(const std::string& s, uint64_t& v) {
    v = 0;
    for (char ch : s) {
        uint64_t c;
        switch (ch) { case 'A': case 'a': case 'N': case 'n': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break; default: return false; }
        v = (v << 2) | c;
    }
    return true;
}

struct ResolutionInfo { const char* name; const char* debug; uint32_t id; bool pars; bool em; };
const ResolutionInfo kRes[] = {
    {"trivial", "Trivial", AFQ_RES_TRIVIAL, false, false},
    {"cr-like", "CellRangerLike", AFQ_RES_CR_LIKE, false, false},
    {"cr-like-em", "CellRangerLikeEm", AFQ_RES_CR_LIKE_EM, false, true},
    {"parsimony-em", "ParsimonyEm", AFQ_RES_PARSIMONY_EM, true, true},
    {"parsimony", "Parsimony", AFQ_RES_PARSIMONY, true, false},
    {"parsimony-gene-em", "ParsimonyGeneEm", AFQ_RES_PARSIMONY_GENE_EM, true, true},
    {"parsimony-gene", "ParsimonyGene", AFQ_RES_PARSIMONY_GENE, true, false},
};

}  // namespace

extern "C" {

const char* afq_host_last_error(void) { return g_herr.c_str(); }

int afq_format_f32(float v, char* buf, size_t cap) { return format_f32(v, buf, cap); }

int64_t afq_snappy_frame_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    std::vector<uint8_t> o; std::string err;
    if (!snappy_frame_decode(in, n, o, err, std::min(8u, std::max(1u, std::thread::hardware_concurrency())))) return hfail(AFQ_ERR_BAD_INPUT, err);
    if (out) { if (o.size() > cap) return hfail(AFQ_ERR_INVALID_ARG, "output buffer too small"); std::memcpy(out, o.data(), o.size()); }
    return (int64_t)o.size();
}

int afq_rad_parse_prelude(const uint8_t* bytes, size_t n, afq_rad_info* out) {
    if (!bytes || !out) return hfail(AFQ_ERR_INVALID_ARG, "null argument");
    RadPrelude P;
    int rc = parse_prelude(bytes, n, P, false);
    if (rc) return rc;
    out->ref_count = P.ref_count; out->num_chunks = P.num_chunks; out->first_chunk_off = P.first_chunk;
    out->is_paired = P.is_paired; out->cblen = (uint32_t)P.file_tag_vals["cblen"]; out->ulen = (uint32_t)P.file_tag_vals["ulen"];
    out->bc_bytes = P.bc_bytes; out->umi_bytes = P.umi_bytes;
    return 0;
}

// MatrixMarket as sprs::io::write_matrix_market writes a TriMatI<f32,u32> (coordinate real general, 1-based), from CSR
static bool write_mtx_file(const std::string& path, uint64_t n_rows, uint64_t n_cols, const std::vector<uint64_t>& rp,
                       const uint32_t* cols, const float* vals, size_t nz, uint32_t num_threads) {
    const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return false;
    char head[160];
    const int hl = std::snprintf(head, sizeof head, "%%%%MatrixMarket matrix coordinate real general\n%% written by sprs\n%llu %llu %zu\n",
                                 (unsigned long long)n_rows, (unsigned long long)n_cols, nz);
    bool ok = ::pwrite(fd, head, (size_t)hl, 0) == hl;
    // the entries are formatted by -t threads into per-slice buffers (same text as one fprintf per entry) and written by the
    // same threads at their final file offsets; a slice is a run of consecutive entries, its first row found by binary
    // search in the row pointers
    // -t defaults to every core in the reference (main.rs:303); 0 = not given.  Slices of 2^18 entries: a 40 M-entry matrix
    // keeps 160 threads busy in one round (at 2^20 it was 41 threads formatting a million entries each)
    const unsigned nth = std::max(1u, std::min(num_threads ? num_threads : std::thread::hardware_concurrency(), 256u));
    const size_t slice = 1u << 18;
    uint64_t file_off = (uint64_t)hl;
    for (size_t base = 0; base < nz && ok; base += slice * nth) {
        // (plain arrays, not strings: a string would zero-fill its 112 bytes per entry before a tenth of them is written)
        std::vector<std::unique_ptr<char[]>> bufs(nth);
        std::vector<size_t> blen(nth, 0);
        auto each = [&](const std::function<void(unsigned, size_t, size_t)>& f) {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nth; ++t) {
                const size_t a = base + t * slice, b = std::min(nz, a + slice);
                if (a >= nz) break;
                th.emplace_back(f, t, a, b);
            }
            for (auto& x : th) x.join();
        };
        each([&](unsigned t, size_t a, size_t b) {
            bufs[t].reset(new char[(b - a) * 112]);  // 2 x <= 20 digits + an f32 in positional notation (<= 48 chars) + separators
            char* const p0 = bufs[t].get();
            char* p = p0;
            size_t row = (size_t)(std::upper_bound(rp.begin(), rp.end(), (uint64_t)a) - rp.begin()) - 1;
            for (size_t k = a; k < b; ++k) {
                while (rp[row + 1] <= k) ++row;   // skips empty rows too
                p = put_u64(p, (unsigned long long)row + 1); *p++ = ' ';
                p = put_u64(p, (unsigned long long)cols[k] + 1); *p++ = ' ';
                p += format_f32(vals[k], p, 64); *p++ = '\n';
            }
            blen[t] = (size_t)(p - p0);
        });
        std::vector<uint64_t> off(nth + 1, file_off);
        for (unsigned t = 0; t < nth; ++t) off[t + 1] = off[t] + blen[t];
        std::vector<int> bad(nth, 0);
        each([&](unsigned t, size_t, size_t) {
            for (size_t w = 0; w < blen[t];) {
                const ssize_t g = ::pwrite(fd, bufs[t].get() + w, blen[t] - w, (off_t)(off[t] + w));
                if (g <= 0) { bad[t] = 1; return; }
                w += (size_t)g;
            }
        });
        for (int x : bad) if (x) ok = false;
        file_off = off[nth];
        std::fill(blen.begin(), blen.end(), 0);
    }
    if (::close(fd) != 0) ok = false;
    return ok;
}

// `alevin-fry infer` (src/infer.rs:31-426): files in, EM per row on the device (afq_infer), files out.
int afq_infer_files(const afq_infer_opts* o) {
    if (!o || !o->count_mat || !o->eq_labels || !o->output_dir) return hfail(AFQ_ERR_INVALID_ARG, "null option");
    const std::string cm = o->count_mat, outd = o->output_dir;
    const size_t slash = cm.find_last_of('/');
    const std::string parent = slash == std::string::npos ? std::string(".") : cm.substr(0, slash);
    // count matrix: MatrixMarket coordinate, `real` (what quant writes) or `integer` (infer.rs:61-83); rows are cells
    uint64_t n_rows = 0, n_cols = 0, nnz = 0;
    std::vector<std::pair<uint64_t, std::pair<uint32_t, uint32_t>>> trip;   // (row, (class, rounded count))
    {
        std::ifstream f(cm);
        if (!f) return hfail(AFQ_ERR_BAD_INPUT, "cannot open " + cm);
        std::string line;
        if (!std::getline(f, line) || line.compare(0, 14, "%%MatrixMarket") != 0) return hfail(AFQ_ERR_BAD_INPUT, "error reading mtx format matrix : no MatrixMarket banner");
        std::string low = line;
        for (auto& ch : low) ch = (char)std::tolower((unsigned char)ch);
        if (low.find("coordinate") == std::string::npos || (low.find("real") == std::string::npos && low.find("integer") == std::string::npos) ||
            low.find("general") == std::string::npos)
            return hfail(AFQ_ERR_BAD_INPUT, "error reading mtx format matrix : only `coordinate real|integer general` is read");
        while (std::getline(f, line)) if (!line.empty() && line[0] != '%') break;
        if (std::sscanf(line.c_str(), "%llu %llu %llu", (unsigned long long*)&n_rows, (unsigned long long*)&n_cols, (unsigned long long*)&nnz) != 3)
            return hfail(AFQ_ERR_BAD_INPUT, "error reading mtx format matrix : bad size line");
        trip.reserve(nnz);
        unsigned long long r, c2; double v;
        for (uint64_t k = 0; k < nnz; ++k) {
            if (!std::getline(f, line) || std::sscanf(line.c_str(), "%llu %llu %lf", &r, &c2, &v) != 3 || r < 1 || r > n_rows || c2 < 1 || c2 > n_cols)
                return hfail(AFQ_ERR_BAD_INPUT, "error reading mtx format matrix : bad entry");
            trip.push_back({r - 1, {(uint32_t)(c2 - 1), (uint32_t)std::llround((float)v)}});   // e.1.round() as u32, infer.rs:389
        }
    }
    std::stable_sort(trip.begin(), trip.end(), [](const auto& a, const auto& b) { return a.first != b.first ? a.first < b.first : a.second.first < b.second.first; });
    // global classes (IndexedEqList::init_from_eqc_file, eq_class.rs:249-298)
    uint64_t num_genes = 0, num_eqc = 0;
    std::vector<std::vector<uint32_t>> eq;
    {
        gzFile gz = gzopen(o->eq_labels, "rb");
        if (!gz) return hfail(AFQ_ERR_BAD_INPUT, std::string("cannot open ") + o->eq_labels);
        std::string txt;
        char buf[1 << 16];
        int got;
        while ((got = gzread(gz, buf, sizeof buf)) > 0) txt.append(buf, (size_t)got);
        gzclose(gz);
        std::istringstream is(txt);
        std::string line;
        if (!std::getline(is, line)) return hfail(AFQ_ERR_BAD_INPUT, "empty equivalence class file");
        num_genes = std::strtoull(line.c_str(), nullptr, 10);
        if (!std::getline(is, line)) return hfail(AFQ_ERR_BAD_INPUT, "truncated equivalence class file");
        num_eqc = std::strtoull(line.c_str(), nullptr, 10);
        eq.assign((size_t)num_eqc, {});
        while (std::getline(is, line)) {
            std::istringstream ls(line);
            std::vector<uint32_t> v;
            unsigned long long x;
            while (ls >> x) v.push_back((uint32_t)x);
            if (v.empty()) continue;
            const uint32_t id = v.back();
            v.pop_back();
            if (id >= num_eqc) return hfail(AFQ_ERR_BAD_INPUT, "equivalence class id out of range");
            eq[id] = std::move(v);
        }
    }
    if (n_cols > num_eqc) return hfail(AFQ_ERR_BAD_INPUT, "the count matrix has more columns than there are equivalence classes");
    // barcodes of the rows, optional subset (infer.rs:113-147, 360-373)
    std::vector<std::string> bcs;
    {
        std::ifstream f(parent + "/quants_mat_rows.txt");
        if (!f) return hfail(AFQ_ERR_BAD_INPUT, "Unable to read first barcode from " + parent + "/quants_mat_rows.txt");
        std::string line;
        while (std::getline(f, line)) { while (!line.empty() && std::isspace((unsigned char)line.back())) line.pop_back(); if (!line.empty()) bcs.push_back(line); }
    }
    std::unordered_set<std::string> keep;
    const bool filter = o->filter_list != nullptr;
    if (filter) {
        std::ifstream f(o->filter_list);
        if (!f) return hfail(AFQ_ERR_BAD_INPUT, std::string("cannot open ") + o->filter_list);
        std::string line;
        while (std::getline(f, line)) { while (!line.empty() && std::isspace((unsigned char)line.back())) line.pop_back(); if (!line.empty()) keep.insert(line); }
    }
    std::vector<uint32_t> lab, ce, cc;
    std::vector<uint64_t> lp(1, 0), cp(1, 0);
    for (auto& v : eq) { lab.insert(lab.end(), v.begin(), v.end()); lp.push_back(lab.size()); }
    if (mkdir(outd.c_str(), 0777) != 0 && errno != EEXIST) return hfail(AFQ_ERR_BAD_INPUT, "cannot create " + outd);
    {
        std::ifstream in(parent + "/quants_mat_cols.txt", std::ios::binary);
        if (!in) return hfail(AFQ_ERR_BAD_INPUT, "could not copy column (gene) names to output");
        std::ofstream out(outd + "/quants_mat_cols.txt", std::ios::binary);
        out << in.rdbuf();
    }
    FILE* rows_f = std::fopen((outd + "/quants_mat_rows.txt").c_str(), "w");
    if (!rows_f) return hfail(AFQ_ERR_BAD_INPUT, "couldn't create output barcode file");
    size_t t = 0;
    uint64_t n_out = 0;
    for (uint64_t r = 0; r < n_rows && r < bcs.size(); ++r) {   // rows zip barcodes (infer.rs:364)
        const size_t t0 = t;
        while (t < trip.size() && trip[t].first == r) ++t;
        if (filter && !keep.count(bcs[r])) continue;
        std::fprintf(rows_f, "%s\n", bcs[r].c_str());
        for (size_t k = t0; k < t; ++k) {
            if (!ce.empty() && cp.back() < ce.size() && ce.back() == trip[k].second.first) { cc.back() += trip[k].second.second; continue; }   // to_csr sums duplicates
            ce.push_back(trip[k].second.first); cc.push_back(trip[k].second.second);
        }
        cp.push_back(ce.size());
        ++n_out;
    }
    std::fclose(rows_f);
    afq_config cfg{};
    cfg.abi_version = AFQ_ABI_VERSION; cfg.resolution = AFQ_RES_CR_LIKE; cfg.num_genes = 1; cfg.num_rows = 1; cfg.small_thresh = 100;
    cfg.pug_exact_umi = 1; cfg.bc_bytes = 4; cfg.umi_bytes = 4;
    const uint32_t t2g0 = 0;
    afq_ctx* ctx = nullptr;
    int rc = afq_create(&cfg, &t2g0, 1, (int)o->device, &ctx);
    if (rc) return hfail(rc, afq_last_error(nullptr));
    afq_result res{};
    rc = afq_infer(ctx, lab.data(), lp.data(), (uint32_t)num_eqc, cp.data(), ce.data(), cc.data(), (uint32_t)n_out, (uint32_t)num_genes, o->usa_mode, &res);
    if (rc) { const std::string m = afq_last_error(ctx); afq_destroy(ctx); return hfail(rc, m); }
    std::vector<uint64_t> rp(res.cell_ptr, res.cell_ptr + res.n_cells + 1);
    std::vector<uint32_t> cols(res.gene, res.gene + res.nnz);
    std::vector<float> vals(res.val, res.val + res.nnz);
    afq_result_release(&res);
    afq_destroy(ctx);
    // (num_cells, num_genes) with num_cells = the subset's size when one is given (infer.rs:141, 175-178)
    if (!write_mtx_file(outd + "/quants_mat.mtx", filter ? keep.size() : n_rows, num_genes, rp, cols.data(), vals.data(), vals.size(), o->num_threads))
        return hfail(AFQ_ERR_BAD_INPUT, "could not write quants_mat.mtx");
    return 0;
}

// ---- the device side of afq_quantify: one context (+ the -d sibling) per device over a contiguous range of cells ----
struct FileCloser { void operator()(FILE* f) const { if (f) std::fclose(f); } };
using FilePtr = std::unique_ptr<FILE, FileCloser>;
struct CtxCloser { void operator()(afq_ctx* c) const { if (c) afq_destroy(c); } };
using CtxPtr = std::unique_ptr<afq_ctx, CtxCloser>;

struct DevOut {   // what one device produced for its range of cells, in cell order
    int rc = 0; std::string err;
    std::vector<uint32_t> gene; std::vector<float> val; std::vector<uint64_t> row_end;   // CSR (row_end cumulative within this part)
    std::vector<uint64_t> bc; std::vector<uint32_t> nrec; std::vector<uint8_t> flags;
    std::vector<uint64_t> eq_cell_end, eq_label_end; std::vector<uint32_t> eq_labels, eq_count;   // -d
    bool have_eq = false;
    std::vector<uint64_t> bm_end, bv_end; std::vector<uint32_t> bm_col, bv_col; std::vector<float> bm_val, bv_val;   // -b
    double t_submit = 0, t_collect = 0;
    // a range that went through in ONE batch keeps the library's result (pinned, already paged in) instead of copying its
    // rows into the vectors above - 330 MB of first-touch page faults on a PBMC-sized run
    afq_result held{};
    bool is_held = false;
    DevOut() = default;
    DevOut(const DevOut&) = delete;
    DevOut& operator=(const DevOut&) = delete;
    ~DevOut() { if (is_held) afq_result_release(&held); }
};

// the collated file as afq_submit_reader sees it: pread into the library's pinned staging - the page cache is copied once, by
// several threads, without the page-fault storm that reading the same bytes through a fresh mapping sets off
struct FileSource { int fd; uint64_t base; };
static int file_read_cb(void* user, uint64_t offset, void* dst, size_t len) {
    const FileSource* f = static_cast<const FileSource*>(user);
    uint8_t* d = static_cast<uint8_t*>(dst);
    while (len) {
        const ssize_t g = ::pread(f->fd, d, len, (off_t)(f->base + offset));
        if (g <= 0) return -1;
        d += g; offset += (uint64_t)g; len -= (size_t)g;
    }
    return 0;
}

// Best effort: run the calling thread (and the staging threads it starts) on the CPUs of the NUMA node the device hangs off,
// so that the context's pinned staging is first touched there and the copies into it do not cross sockets
// (/sys/bus/pci/devices/<bus id>/numa_node, /sys/devices/system/node/node<N>/cpulist).  Silent when the topology is not exposed.
static void bind_thread_to_device_node(int device) {
    char bus[64] = {0};
    if (afq_device_pci_bus_id(device, bus, sizeof(bus)) != 0) return;
    for (char* q = bus; *q; ++q) *q = (char)std::tolower((unsigned char)*q);
    std::vector<uint8_t> f;
    if (!read_file(std::string("/sys/bus/pci/devices/") + bus + "/numa_node", f) || f.empty()) return;
    const int node = std::atoi(std::string(f.begin(), f.end()).c_str());
    if (node < 0) return;
    if (!read_file("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist", f) || f.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    const std::string list(f.begin(), f.end());
    size_t i = 0;
    int n_set = 0;
    while (i < list.size()) {   // "0-63,128-191"
        char* e = nullptr;
        const long a = std::strtol(list.c_str() + i, &e, 10);
        if (e == list.c_str() + i) break;
        long b = a;
        i = (size_t)(e - list.c_str());
        if (i < list.size() && list[i] == '-') { b = std::strtol(list.c_str() + i + 1, &e, 10); i = (size_t)(e - list.c_str()); }
        for (long cpu = a; cpu <= b && cpu < CPU_SETSIZE; ++cpu) { CPU_SET((int)cpu, &set); ++n_set; }
        if (i < list.size() && list[i] == ',') ++i; else break;
    }
    if (n_set) (void)sched_setaffinity(0, sizeof(set), &set);
}

struct DevStat { double busy_s = 0; uint64_t bytes = 0, cells = 0; uint32_t batches = 0; int rc = 0; std::string err; };

// One device's host thread: its contexts, then batches of cells popped off the shared queue until it is empty - what the
// reference's workers do with chunks (quant.rs:1553-1575, 1678-1765); the rows of batch b go to parts[b], so the gather in
// batch order is the gather in cell order whatever device took which batch.
static void run_device_worker(const afq_config& cfg, const afq_config* cfg_eq, const std::vector<uint32_t>& t2g, int device, bool bind_numa,
                             const uint8_t* rad, int rad_fd, const std::vector<uint64_t>& chunk_off, const std::vector<uint64_t>& chunk_nb,
                             const std::vector<uint32_t>& chunk_nr, const std::vector<std::pair<size_t, size_t>>& batches, std::atomic<size_t>& next,
                             std::atomic<int>& stop, bool want_eq, bool res_is_em, std::vector<DevOut>& parts, DevStat& stat) {
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    if (bind_numa) bind_thread_to_device_node(device);
    afq_ctx* raw = nullptr;
    int rc = afq_create(&cfg, t2g.data(), (uint32_t)t2g.size(), device, &raw);
    if (rc) { stat.rc = rc; stat.err = std::string("afq_create: ") + afq_last_error(nullptr); stop = 1; return; }
    CtxPtr ctx(raw), ctx_eq;
    if (cfg_eq) {
        rc = afq_create(cfg_eq, t2g.data(), (uint32_t)t2g.size(), device, &raw);
        if (rc) { stat.rc = rc; stat.err = std::string("afq_create: ") + afq_last_error(nullptr); stop = 1; return; }
        ctx_eq.reset(raw);
    }
    const bool single = batches.size() == 1;
    for (;;) {
        if (stop.load()) return;
        const size_t bi = next.fetch_add(1);
        if (bi >= batches.size()) return;
        DevOut& out = parts[bi];
        const size_t c0 = batches[bi].first, c1 = batches[bi].second;
        const auto t_batch = now();
        {
        // hand over just this batch's byte span (offsets relative to it), not the whole file
        const uint64_t span0 = chunk_off[c0];
        std::vector<uint64_t> rel(c1 - c0);
        for (size_t k = c0; k < c1; ++k) rel[k - c0] = chunk_off[k] - span0;
        const uint64_t span1 = chunk_off[c1 - 1] + chunk_nb[c1 - 1];   // with --quant-subset the span also covers chunks that were filtered out
        auto ta = now();
        if (rad_fd >= 0) {   // an uncompressed file: the library pulls the bytes itself (pread into its staging)
            FileSource fs{rad_fd, span0};
            std::vector<uint32_t> hdr(2 * (c1 - c0));
            for (size_t k = c0; k < c1; ++k) { hdr[2 * (k - c0)] = (uint32_t)chunk_nb[k]; hdr[2 * (k - c0) + 1] = chunk_nr[k]; }
            rc = afq_submit_reader(ctx.get(), file_read_cb, &fs, (size_t)(span1 - span0), rel.data(), hdr.data(), (uint32_t)(c1 - c0), c0);
        } else
            rc = afq_submit(ctx.get(), rad + span0, (size_t)(span1 - span0), rel.data(), (uint32_t)(c1 - c0), c0);
        auto tb = now();
        afq_result res{};
        if (!rc) rc = afq_collect(ctx.get(), &res);
        out.t_submit += secs(ta, tb); out.t_collect += secs(tb, now());
        if (rc) { out.rc = rc; out.err = afq_last_error(ctx.get()); stat.rc = rc; stat.err = out.err; stop = 1; return; }
        if (cfg.num_bootstraps) {   // quant.rs:1270-1277
            afq_bootstraps bs{};
            if (afq_result_bootstraps(&res, &bs) == 0) {
                const uint64_t m0 = out.bm_col.size(), v0 = out.bv_col.size();
                out.bm_col.insert(out.bm_col.end(), bs.mean_col, bs.mean_col + bs.mean_ptr[bs.n_cells]);
                out.bm_val.insert(out.bm_val.end(), bs.mean_val, bs.mean_val + bs.mean_ptr[bs.n_cells]);
                out.bv_col.insert(out.bv_col.end(), bs.var_col, bs.var_col + bs.var_ptr[bs.n_cells]);
                out.bv_val.insert(out.bv_val.end(), bs.var_val, bs.var_val + bs.var_ptr[bs.n_cells]);
                for (uint64_t i = 0; i < bs.n_cells; ++i) { out.bm_end.push_back(m0 + bs.mean_ptr[i + 1]); out.bv_end.push_back(v0 + bs.var_ptr[i + 1]); }
            }
        }
        if (want_eq) {
            afq_result res_eq{};
            afq_eqclasses ec{};
            if (ctx_eq) {
                rc = afq_submit(ctx_eq.get(), rad + span0, (size_t)(span1 - span0), rel.data(), (uint32_t)(c1 - c0), c0);
                if (!rc) rc = afq_collect(ctx_eq.get(), &res_eq);
                if (rc) { out.rc = rc; out.err = afq_last_error(ctx_eq.get()); stat.rc = rc; stat.err = out.err; stop = 1; afq_result_release(&res); return; }
            }
            const bool have = (ctx_eq || res_is_em) && afq_result_eqclasses(ctx_eq ? &res_eq : &res, &ec) == 0;
            const uint64_t k0 = out.eq_count.size(), w0 = out.eq_labels.size();
            if (have) {
                out.have_eq = true;
                out.eq_count.insert(out.eq_count.end(), ec.count, ec.count + ec.n_classes);
                out.eq_labels.insert(out.eq_labels.end(), ec.labels, ec.labels + ec.n_words);
                for (uint64_t k = 0; k < ec.n_classes; ++k) out.eq_label_end.push_back(w0 + ec.label_ptr[k + 1]);
            }
            for (uint64_t i = 0; i < res.n_cells; ++i) out.eq_cell_end.push_back(k0 + (have ? ec.cell_ptr[i + 1] : 0));
            if (ctx_eq) afq_result_release(&res_eq);
        }
        const uint64_t g0 = out.gene.size();
        const bool whole = single;   // (the one batch of a one-device run: its rows stay in the library's pinned result, no copy)
        if (!whole) {
            out.gene.insert(out.gene.end(), res.gene, res.gene + res.nnz);
            out.val.insert(out.val.end(), res.val, res.val + res.nnz);
        }
        for (uint64_t i = 0; i < res.n_cells; ++i) out.row_end.push_back(g0 + res.cell_ptr[i + 1]);
        out.bc.insert(out.bc.end(), res.bc, res.bc + res.n_cells);
        out.nrec.insert(out.nrec.end(), res.nrec, res.nrec + res.n_cells);
        out.flags.insert(out.flags.end(), res.flags, res.flags + res.n_cells);
        if (whole) { out.held = res; out.is_held = true; } else afq_result_release(&res);
        }
        stat.busy_s += secs(t_batch, now());
        stat.batches += 1; stat.cells += c1 - c0;
        for (size_t k = c0; k < c1; ++k) stat.bytes += chunk_nb[k];
    }
}

// `alevin-fry atac deduplicate` (src/atac/deduplicate.rs:68-309): collated scATAC RAD in, <input_dir>/map.bed out.  The
// record walk, the na==1 && type==4 filter, the per-cell sort and the run-length count run on the device
// (afq_atac_dedup_rad); here: the directory protocol, the prelude, and write_bed (deduplicate.rs:37-66).
int afq_atac_deduplicate(const afq_atac_dedup_opts* o) {
    if (!o || !o->input_dir) return hfail(AFQ_ERR_INVALID_ARG, "null option");
    const std::string in = o->input_dir;
    if (!file_exists(in + "/generate_permit_list.json") || !file_exists(in + "/collate.json"))   // atac/run.rs:149-167
        return hfail(AFQ_ERR_BAD_INPUT, "The provided input directory lacks a generate_permit_list.json or collate.json file; this should not happen.");
    std::vector<uint8_t> cj;
    if (!read_file(in + "/collate.json", cj)) return hfail(AFQ_ERR_BAD_INPUT, "could not open the collate.json file.");
    std::string cjs(cj.begin(), cj.end());
    bool compressed = false, have_flag = false;
    { size_t k = cjs.find("\"compressed_output\""); if (k != std::string::npos) { size_t v = cjs.find_first_not_of(" \t\r\n:", k + 19); have_flag = v != std::string::npos; compressed = have_flag && cjs.compare(v, 4, "true") == 0; } }
    if (!have_flag) return hfail(AFQ_ERR_BAD_INPUT, "could not read compressed_output field from collate metadata.");
    PhaseClock pc;
    MappedFile mf;
    if (!mf.open(in + (compressed ? "/map.collated.rad.sz" : "/map.collated.rad"))) return hfail(AFQ_ERR_BAD_INPUT, "could not read the collated RAD file (run collate before deduplicate)");
    std::unique_ptr<uint8_t[]> rad_buf;
    uint64_t rad_buf_n = 0;
    const unsigned nthreads = o->num_threads ? o->num_threads : std::max(1u, std::thread::hardware_concurrency());
    if (compressed) {
        std::string err;
        std::vector<SnappyChunk> chunks;
        if (!snappy_frame_plan(mf.p, mf.n, chunks, rad_buf_n, err)) return hfail(AFQ_ERR_BAD_INPUT, "map.collated.rad.sz: " + err);
        rad_buf.reset(new (std::nothrow) uint8_t[rad_buf_n ? rad_buf_n : 1]);
        if (!rad_buf) return hfail(AFQ_ERR_OOM, "map.collated.rad.sz: not enough host memory for the decompressed file");
        if (!snappy_frame_run(mf.p, chunks, rad_buf.get(), nthreads, err)) return hfail(AFQ_ERR_BAD_INPUT, "map.collated.rad.sz: " + err);
    }
    const uint8_t* rad = compressed ? rad_buf.get() : mf.p;
    const size_t rad_n = compressed ? (size_t)rad_buf_n : mf.n;
    RadPrelude P;
    int rc = parse_prelude(rad, rad_n, P, true);
    if (rc) return rc;
    // AtacSeqReadRecord: read tag b; alignment tags ref:u32, type:u8, start_pos:u32, frag_len:u16 (tests/atac_integration.rs:110-121)
    if (P.read_tags.size() != 1 || P.read_tags[0].name != "b" || !P.bc_bytes) return hfail(AFQ_ERR_UNSUPPORTED, "scATAC RAD: the read-level tags must be exactly one integer 'b'");
    static const struct { const char* name; uint8_t type; } kAln[4] = {{"ref", 3}, {"type", 1}, {"start_pos", 3}, {"frag_len", 2}};
    bool aln_ok = P.aln_tags.size() == 4;
    for (size_t i = 0; aln_ok && i < 4; ++i) aln_ok = P.aln_tags[i].name == kAln[i].name && P.aln_tags[i].type == kAln[i].type;
    if (!aln_ok) return hfail(AFQ_ERR_UNSUPPORTED, "scATAC RAD: the alignment-level tags must be ref:u32, type:u8, start_pos:u32, frag_len:u16");
    if (!P.file_tag_vals.count("cblen")) return hfail(AFQ_ERR_BAD_INPUT, "tag map must contain cblen");
    const uint32_t cblen = (uint32_t)P.file_tag_vals["cblen"];
    std::vector<uint64_t> chunk_off;
    for (size_t p = P.first_chunk; p < rad_n;) {
        if (p + 8 > rad_n) return hfail(AFQ_ERR_BAD_INPUT, "trailing bytes after the last chunk");
        uint32_t nb; std::memcpy(&nb, rad + p, 4);
        if (nb < 8 || p + nb > rad_n) return hfail(AFQ_ERR_BAD_INPUT, "corrupt chunk header");
        chunk_off.push_back(p); p += nb;
    }
    pc.lap("map + prelude + chunk table");
    afq_config cfg{};
    cfg.abi_version = AFQ_ABI_VERSION; cfg.resolution = AFQ_RES_CR_LIKE; cfg.num_genes = 1; cfg.num_rows = 1; cfg.small_thresh = 100;
    cfg.pug_exact_umi = 1; cfg.bc_bytes = 4; cfg.umi_bytes = 4;
    const uint32_t t2g0 = 0;
    afq_ctx* raw = nullptr;
    rc = afq_create(&cfg, &t2g0, 1, (int)o->device, &raw);
    if (rc) return hfail(rc, afq_last_error(nullptr));
    CtxPtr ctx(raw);
    uint64_t *optr = nullptr, *obc = nullptr; uint32_t *oref = nullptr, *ostart = nullptr; uint16_t *oflen = nullptr, *ocnt = nullptr;
    afq_atac_stats st{};
    const uint64_t span0 = chunk_off.empty() ? 0 : chunk_off[0];
    std::vector<uint64_t> rel(chunk_off.size());
    for (size_t i = 0; i < chunk_off.size(); ++i) rel[i] = chunk_off[i] - span0;
    rc = afq_atac_dedup_rad(ctx.get(), rad + span0, rad_n - span0, rel.data(), (uint32_t)chunk_off.size(), P.bc_bytes, 0, &optr, &obc, &oref, &ostart, &oflen, &ocnt, &st);
    if (rc) return hfail(rc, afq_last_error(ctx.get()));
    struct Freer { uint64_t* a; uint64_t* b; uint32_t* c2; uint32_t* d; uint16_t* e; uint16_t* f; ~Freer() { afq_free(a); afq_free(b); afq_free(c2); afq_free(d); afq_free(e); afq_free(f); } } fr{optr, obc, oref, ostart, oflen, ocnt};
    pc.lap("device: parse + dedup");
    // write_bed (deduplicate.rs:37-66): chr name, start, start + frag_len, barcode (reverse-complemented with -d rc), count;
    // fragments of 2000 bases and more are counted, not written.  Cells in file order (the reference: worker completion order).
    const size_t n_cells = chunk_off.size();
    const unsigned nth = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)nthreads, 64, n_cells / 64 + 1}));
    std::vector<std::string> txt(nth);
    std::vector<int> bad(nth, 0);
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nth; ++t)
            th.emplace_back([&, t]() {
                std::string& out = txt[t];
                char num[32];
                for (size_t i = n_cells * t / nth; i < n_cells * (t + 1) / nth; ++i) {
                    uint64_t bc = obc[i];
                    if (o->rev) {   // needletail::bitkmer::reverse_complement
                        uint64_t r = 0;
                        for (uint32_t k = 0; k < cblen; ++k) { r = (r << 2) | (3 - (bc & 3)); bc >>= 2; }
                        bc = r;
                    }
                    const std::string bcs = bc_to_string(bc, cblen);
                    for (uint64_t k = optr[i]; k < optr[i + 1]; ++k) {
                        if (oflen[k] >= 2000) continue;
                        if (oref[k] >= P.ref_names.size()) { bad[t] = 1; return; }
                        out += P.ref_names[oref[k]]; out += '\t';
                        *put_u64(num, ostart[k]) = 0; out += num; out += '\t';
                        *put_u64(num, (unsigned long long)(uint32_t)(ostart[k] + (uint32_t)oflen[k])) = 0; out += num; out += '\t';
                        out += bcs; out += '\t';
                        *put_u64(num, ocnt[k]) = 0; out += num; out += '\n';
                    }
                }
            });
        for (auto& x : th) x.join();
    }
    for (int b2 : bad) if (b2) return hfail(AFQ_ERR_BAD_INPUT, "a fragment's reference id is beyond the RAD header's reference names");
    FilePtr bed(std::fopen((in + "/map.bed").c_str(), "w"));
    if (!bed) return hfail(AFQ_ERR_BAD_INPUT, "could not create map.bed");
    for (auto& tx : txt) if (!tx.empty() && std::fwrite(tx.data(), 1, tx.size(), bed.get()) != tx.size()) return hfail(AFQ_ERR_BAD_INPUT, "could not write map.bed");
    bed.reset();
    pc.lap("map.bed");
    std::fprintf(stderr, "finished parsing RAD file; processed %llu total records\n", (unsigned long long)st.n_records);
    std::fprintf(stderr, "Number of records with greater than 1 mapping %llu\n", (unsigned long long)st.n_multimapped);
    std::fprintf(stderr, "Number of records that are deduplicated %llu\n", (unsigned long long)st.n_deduplicated);
    std::fprintf(stderr, "Number of records that are not mapped pairs %llu\n", (unsigned long long)st.n_not_mapped_pair);
    std::fprintf(stderr, "Number of records that have frag length > 2000 %llu\n", (unsigned long long)st.n_long_fragments);
    if (o->stats_out) *o->stats_out = st;
    return 0;
}

// collation_manifest.bin (libradicl::collation::CollationManifest, written by src/collate.rs:1861-1890): the names of the
// samples, indexed by the integer the scatter phase left in barcodes[0].  libradicl's source is not under /root/reference, so
// the layout read here is a RESTATEMENT of its serde derive under bincode's default options (parity unpinned):
//   level_names: u64 n, n x (u64 len, bytes);  sample_groups: u64 n, n x { key u64, name: u8 tag (0 None / 1 Some) [u64 len,
//   bytes], chunk_start u64, num_chunks u64, num_records u64 }
// The file must tile exactly; anything else is refused.  A sample without a name is called by its key in hex (quant.rs:1364-1368).
static bool parse_collation_manifest(const std::vector<uint8_t>& b, std::vector<std::string>& names) {
    Cursor c{b.data(), b.size()};
    const uint64_t nl = c.get<uint64_t>();
    if (!c.ok || nl > 64) return false;
    for (uint64_t i = 0; i < nl; ++i) { const uint64_t l = c.get<uint64_t>(); if (!c.ok || l > 4096) return false; c.str((size_t)l); }
    const uint64_t ng = c.get<uint64_t>();
    if (!c.ok || ng > (1u << 24)) return false;
    for (uint64_t i = 0; i < ng; ++i) {
        const uint64_t key = c.get<uint64_t>();
        const uint8_t tag = c.get<uint8_t>();
        std::string nm;
        if (tag == 1) { const uint64_t l = c.get<uint64_t>(); if (!c.ok || l > 65536) return false; nm = c.str((size_t)l); }
        else if (tag == 0) { char hx[32]; std::snprintf(hx, sizeof hx, "%llx", (unsigned long long)key); nm = hx; }
        else return false;
        c.get<uint64_t>(); c.get<uint64_t>(); c.get<uint64_t>();
        if (!c.ok) return false;
        names.push_back(nm);
    }
    return c.p == c.n;
}

int afq_quantify(const afq_quant_opts* o) {
    if (!o || !o->input_dir || !o->tg_map || !o->output_dir || !o->resolution) return hfail(AFQ_ERR_INVALID_ARG, "null option");
    const ResolutionInfo* R = nullptr;
    for (auto& r : kRes) { std::string a = o->resolution; for (auto& ch : a) ch = (char)std::tolower(ch); if (a == r.name) R = &r; }
    if (!R) return hfail(AFQ_ERR_INVALID_ARG, std::string("unknown resolution ") + o->resolution);
    // flag compatibility, src/main.rs:652-728
    const int edist = o->umi_edit_dist < 0 ? (R->pars ? 1 : 0) : o->umi_edit_dist;
    if (edist > 1 || (edist == 1 && !R->pars)) return hfail(AFQ_ERR_INVALID_ARG, "resolution does not support this --umi-edit-dist");
    const uint32_t large_thresh = o->large_graph_thresh < 0 ? (R->pars ? 1000u : 0u) : (uint32_t)o->large_graph_thresh;
    if (o->dump_eq && R->id == AFQ_RES_TRIVIAL)   // src/main.rs:705-711
        return hfail(AFQ_ERR_INVALID_ARG, "Gene equivalence classes are not meaningful in case of Trivial resolution.");
    if (o->num_bootstraps && !(R->id == AFQ_RES_CR_LIKE_EM || R->id == AFQ_RES_PARSIMONY_EM || R->id == AFQ_RES_PARSIMONY_GENE_EM))   // src/main.rs:713-728
        return hfail(AFQ_ERR_INVALID_ARG, "The num_bootstraps argument was set to " + std::to_string(o->num_bootstraps) +
                     ", but bootstrapping can only be used with the cr-like-em, parsimony-em, or parsimony-gene-em resolution strategies");
    if (o->num_bootstraps && !o->summary_stat)   // BootstrapHelper::new, src/quant.rs:142-148
        std::fprintf(stderr, "NOTE: Full per-replicate bootstrap output is not yet supported in MTX format. Summary statistics (mean, variance) will be written instead. "
                             "Full replicate output will be available in a future release via AnnData/h5ad.\n");
    const std::string in = o->input_dir, outd = o->output_dir;
    // the permit-list json must exist (src/main.rs:733-734)
    if (!file_exists(in + "/generate_permit_list.json")) return hfail(AFQ_ERR_BAD_INPUT, "the input directory has no generate_permit_list.json");
    std::vector<uint8_t> cj;
    if (!read_file(in + "/collate.json", cj)) return hfail(AFQ_ERR_BAD_INPUT, "could not read collate.json");
    std::string cjs(cj.begin(), cj.end());
    bool compressed = false;
    { size_t k = cjs.find("\"compressed_output\""); if (k != std::string::npos) { size_t v = cjs.find_first_not_of(" \t\r\n:", k + 19); compressed = v != std::string::npos && cjs.compare(v, 4, "true") == 0; } }
    PhaseClock pc;
    // the HIP runtime takes a few hundred ms to come up: let it do so while the prelude and the tg-map are parsed
    std::thread warm([dev = (int)(o->devices && o->n_devices ? o->devices[0] : (int)o->device)]() { afq_device_warmup(dev); });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } warm_join{warm};
    MappedFile mf;
    if (!mf.open(in + (compressed ? "/map.collated.rad.sz" : "/map.collated.rad"))) return hfail(AFQ_ERR_BAD_INPUT, "could not read the collated RAD file");
    std::unique_ptr<uint8_t[]> rad_buf;   // (not a vector: no point zero-filling gigabytes that are about to be overwritten)
    uint64_t rad_buf_n = 0;
    if (compressed) {
        std::string err;
        std::vector<SnappyChunk> chunks;
        if (!snappy_frame_plan(mf.p, mf.n, chunks, rad_buf_n, err)) return hfail(AFQ_ERR_BAD_INPUT, "map.collated.rad.sz: " + err);
        rad_buf.reset(new (std::nothrow) uint8_t[rad_buf_n ? rad_buf_n : 1]);
        if (!rad_buf) return hfail(AFQ_ERR_OOM, "map.collated.rad.sz: not enough host memory for the decompressed file");
        if (!snappy_frame_run(mf.p, chunks, rad_buf.get(), o->num_threads ? o->num_threads : std::max(1u, std::thread::hardware_concurrency()), err)) return hfail(AFQ_ERR_BAD_INPUT, "map.collated.rad.sz: " + err);
    }
    struct { const uint8_t* p; size_t n; const uint8_t* data() const { return p; } size_t size() const { return n; } } rad{compressed ? rad_buf.get() : mf.p, compressed ? (size_t)rad_buf_n : mf.n};
    pc.lap(compressed ? "map + snappy decode" : "map the RAD file");
    RadPrelude P;
    int rc = parse_prelude(rad.data(), rad.size(), P, true);
    if (rc) return rc;
    // Record layout.  Single-barcode scRNA: read tags (b, u).  Multi-barcode (10x Flex; KnownRecordType::RnaShortMultiBC,
    // src/utils.rs:313-340): file tag num_barcodes = 2, read tags (b0, b1, u) - after collation b0 is the integer sample
    // index and b1 the cell barcode (src/quant.rs:2003-2027).  The device decoders see (b0, b1) as ONE barcode field of
    // w0 + w1 bytes: every record of a collated chunk carries the same pair, so candidate detection, the proof and the
    // reported collate key work unchanged, and the host splits the 64-bit key back into sample index and cell barcode.
    // (Field order b0, b1, u inside a record = the order of the read-tag section; libradicl's MultiBarcodeReadRecord
    // writer is not under /root/reference: parity unpinned.)
    const bool multi_bc = P.file_tag_vals.count("num_barcodes") && P.file_tag_vals["num_barcodes"] > 1;
    uint32_t w_sample = 0, cblen = 0, bc_split = 0;
    if (multi_bc) {
        if (P.file_tag_vals["num_barcodes"] != 2 || P.read_tags.size() != 3 || P.read_tags[0].name != "b0" || P.read_tags[1].name != "b1" || P.read_tags[2].name != "u")
            return hfail(AFQ_ERR_UNSUPPORTED, "multi-barcode RAD: only two barcode levels with read tags (b0, b1, u) are supported");
        const uint32_t w0 = (uint32_t)int_type_bytes(P.read_tags[0].type), w1 = (uint32_t)int_type_bytes(P.read_tags[1].type);
        P.umi_bytes = (uint32_t)int_type_bytes(P.read_tags[2].type);
        if (!w0 || !w1 || w0 > 4 || w1 > 4 || !P.umi_bytes) return hfail(AFQ_ERR_UNSUPPORTED, "multi-barcode RAD: b0 and b1 must be integers of 1, 2 or 4 bytes (u8/u16/u32)");
        // A RAD writer picks the narrowest integer per barcode length (an 8-nt sample barcode is a u16, a 16-nt cell barcode a
        // u32): unequal widths go to the library as a split barcode field, which it rewrites with two dwords (afq_config.bc_split)
        P.bc_bytes = w0 + w1;
        if (w0 != w1) { bc_split = w0; w_sample = 4; } else w_sample = w0;
        if (!P.file_tag_vals.count("b1len")) return hfail(AFQ_ERR_BAD_INPUT, "multi-barcode RAD file should have a \"b1len\" file-level tag");
        cblen = (uint32_t)P.file_tag_vals["b1len"];
    } else {
        if (!P.bc_bytes || !P.umi_bytes) return hfail(AFQ_ERR_UNSUPPORTED, "RAD read tags must hold integer 'b' and 'u' (single-barcode records)");
        // the device decoders walk `na, b, u, na x u32` records (src/convert.rs:124-144): any other tag layout would be misread, not skipped
        if (P.read_tags.size() != 2 || P.read_tags[0].name != "b" || P.read_tags[1].name != "u")
            return hfail(AFQ_ERR_UNSUPPORTED, "RAD read-level tags other than (b, u) are not supported");
        if (P.file_tag_vals.count("cblen")) cblen = (uint32_t)P.file_tag_vals["cblen"];
        else return hfail(AFQ_ERR_UNSUPPORTED, "no cblen file tag");
    }
    if (P.aln_tags.size() != 1 || P.aln_tags[0].type != 3)
        return hfail(AFQ_ERR_UNSUPPORTED, "RAD alignment-level tags other than one u32 (compressed_ori_refid) are not supported");
    std::vector<std::string> sample_names;   // multi-barcode: name of sample i (src/quant.rs:1354-1373)
    bool have_samples = false;
    if (file_exists(in + "/collation_manifest.bin")) {
        std::vector<uint8_t> mb;
        if (!read_file(in + "/collation_manifest.bin", mb) || !parse_collation_manifest(mb, sample_names))
            return hfail(AFQ_ERR_UNSUPPORTED, "collation_manifest.bin is not in the layout this build reads (bincode of CollationManifest; see afq_host.cpp)");
        have_samples = true;
    }
    // chunk table: hop the nbytes headers (what the producer thread does)
    const uint32_t rec_hdr = 4 + P.bc_bytes + P.umi_bytes;
    std::vector<uint64_t> chunk_off, chunk_nb;
    std::vector<uint32_t> chunk_nr;
    for (size_t p = P.first_chunk; p < rad.size();) {
        if (p + 8 > rad.size()) return hfail(AFQ_ERR_BAD_INPUT, "trailing bytes after the last chunk");
        uint32_t nb, nr; std::memcpy(&nb, rad.data() + p, 4); std::memcpy(&nr, rad.data() + p + 4, 4);
        if (nb < 8 || p + nb > rad.size()) return hfail(AFQ_ERR_BAD_INPUT, "corrupt chunk header");
        if (nr == 0 || nb < 8 + rec_hdr) return hfail(AFQ_ERR_BAD_INPUT, "chunk " + std::to_string(chunk_off.size()) + " holds no record");   // (quant.rs:756 panics)
        chunk_off.push_back(p); chunk_nb.push_back(nb); chunk_nr.push_back(nr); p += nb;
    }
    auto cell_key_of = [&](uint64_t raw_bc) -> uint64_t { return multi_bc ? raw_bc >> (8 * w_sample) : raw_bc; };
    // --quant-subset (src/quant.rs:1523-1536, 1776): the first record's collate key decides
    size_t subset_size = 0;
    if (o->filter_list) {
        std::ifstream f(o->filter_list);
        if (!f) return hfail(AFQ_ERR_BAD_INPUT, "could not read the --quant-subset file");
        std::unordered_set<uint64_t> keep; std::string line;
        while (std::getline(f, line)) { while (!line.empty() && std::isspace((unsigned char)line.back())) line.pop_back(); uint64_t v; if (!line.empty() && string_to_bc(line, v)) keep.insert(v); }
        subset_size = keep.size();
        std::vector<uint64_t> kept, kept_nb;
        std::vector<uint32_t> kept_nr;
        for (size_t i = 0; i < chunk_off.size(); ++i) { uint64_t bc = 0; std::memcpy(&bc, rad.data() + chunk_off[i] + 8 + 4, P.bc_bytes); if (keep.count(cell_key_of(bc))) { kept.push_back(chunk_off[i]); kept_nb.push_back(chunk_nb[i]); kept_nr.push_back(chunk_nr[i]); } }
        chunk_off.swap(kept); chunk_nb.swap(kept_nb); chunk_nr.swap(kept_nr);
    }
    // transcript-to-gene map (src/utils.rs:487-662)
    std::unordered_map<std::string, uint32_t> rname_to_id;
    for (uint32_t i = 0; i < P.ref_names.size(); ++i) rname_to_id[P.ref_names[i]] = i;
    std::vector<uint32_t> t2g(P.ref_count, UINT32_MAX);
    std::vector<std::string> gene_names; std::unordered_map<std::string, uint32_t> gene_id;
    bool usa = false; size_t found = 0;
    {
        std::ifstream f(o->tg_map);
        if (!f) return hfail(AFQ_ERR_BAD_INPUT, "couldn't open the transcript-to-gene map");
        std::string line; int ncol = -1; uint32_t next_gid = 0;
        while (std::getline(f, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) continue;
            std::vector<std::string> col; { std::stringstream ss(line); std::string c2; while (std::getline(ss, c2, '\t')) col.push_back(c2); }
            if (ncol < 0) { ncol = (int)col.size(); if (ncol != 2 && ncol != 3) return hfail(AFQ_ERR_BAD_INPUT, "Transcript-gene mapping must have either 2 or 3 columns."); usa = ncol == 3; }
            if ((int)col.size() != ncol) return hfail(AFQ_ERR_BAD_INPUT, "failed to parse the transcript-to-gene map : ragged row");
            uint32_t gid;
            auto it = gene_id.find(col[1]);
            if (it == gene_id.end()) { gid = usa ? next_gid : (uint32_t)gene_names.size(); next_gid += 2; gene_id.emplace(col[1], gid); gene_names.push_back(col[1]); }
            else gid = it->second;
            auto rt = rname_to_id.find(col[0]);
            if (rt == rname_to_id.end()) continue;
            ++found;
            if (!usa) t2g[rt->second] = gid;
            else if (col[2] == "U" || col[2] == "u") t2g[rt->second] = gid + 1;
            else if (col[2] == "S" || col[2] == "s") t2g[rt->second] = gid;
            else return hfail(AFQ_ERR_BAD_INPUT, "Third column in 3 column txp-to-gene file must be S or U");
        }
    }
    if (found != P.ref_count) return hfail(AFQ_ERR_BAD_INPUT, "The tg-map must contain a gene mapping for all transcripts in the header");
    const uint32_t G = (uint32_t)gene_names.size();
    afq_config cfg{};
    cfg.abi_version = AFQ_ABI_VERSION; cfg.resolution = R->id; cfg.usa_mode = usa;
    if (o->sa_model > AFQ_SA_PREFER_AMBIG) return hfail(AFQ_ERR_INVALID_ARG, "unknown sa_model");
    cfg.sa_model = o->sa_model;
    if (!usa && cfg.sa_model != AFQ_SA_WINNER_TAKE_ALL) {   // src/quant.rs:1456-1469
        std::fprintf(stderr, "When not operating in USA-mode (all-in-one unspliced/spliced/ambiguous), the SplicedAmbiguityModel will be ignored.\n");
        cfg.sa_model = AFQ_SA_WINNER_TAKE_ALL;
    }
    cfg.num_genes = usa ? 2 * G : G; cfg.num_rows = usa ? 3 * G : G;  // src/quant.rs:1627-1645
    cfg.small_thresh = o->small_thresh; cfg.large_graph_thresh = large_thresh; cfg.pug_exact_umi = (R->pars && edist == 0) ? 1 : 0;
    cfg.em_init_uniform = o->init_uniform; cfg.bc_bytes = P.bc_bytes; cfg.bc_split = bc_split; cfg.umi_bytes = P.umi_bytes; { const uint64_t ul = P.file_tag_vals.count("ulen") ? P.file_tag_vals["ulen"] : 0; cfg.umi_len = ul <= 4ull * P.umi_bytes ? (uint32_t)ul : 0u; }
    // -d: the device keeps the gene-level classes only in the -em resolutions (there they are the EM's input); a plain
    // resolution leaves the same classes behind as its -em sibling (same resolution step, different count extraction:
    // quant.rs:882-924, 966-1019), so for it a second context runs the sibling for the classes alone.  `trivial` never
    // fills gene_eqc: every cell reports no classes.
    const bool res_is_em = cfg.resolution == AFQ_RES_CR_LIKE_EM || cfg.resolution == AFQ_RES_PARSIMONY_EM || cfg.resolution == AFQ_RES_PARSIMONY_GENE_EM;
    cfg.num_bootstraps = o->num_bootstraps; cfg.summary_stat = o->summary_stat; cfg.boot_seed = o->boot_seed;
    afq_config cfg_eq = cfg;
    if (o->dump_eq) {
        if (res_is_em) cfg.dump_eq = 1;
        else if (cfg.resolution != AFQ_RES_TRIVIAL) {
            cfg_eq.dump_eq = 1;
            cfg_eq.resolution = cfg.resolution == AFQ_RES_CR_LIKE ? AFQ_RES_CR_LIKE_EM : cfg.resolution == AFQ_RES_PARSIMONY ? AFQ_RES_PARSIMONY_EM : AFQ_RES_PARSIMONY_GENE_EM;
        }
    }

    // Unmapped reads per corrected barcode (quant.rs:1484-1494; only CorrectedReads / MappingRate of featureDump use
    // them).  Any failure to read the file means "no unmapped reads", as in the reference.  The layout read here is
    // the bincode HashMap<u64, u32> one (count:u64, then key:u64 value:u32 pairs - atac/collate.rs:258-282 writes it);
    // the self-describing layout of libradicl 0.18's CollatedUnmappedCounts is not witnessed anywhere under the
    // reference tree, so a file that does not tile as the former is reported and ignored.
    std::unordered_map<uint64_t, uint32_t> unmapped;
    {
        std::vector<uint8_t> ub;
        if (read_file(in + "/unmapped_bc_count_collated.bin", ub)) {
            uint64_t n = 0;
            if (ub.size() >= 8) std::memcpy(&n, ub.data(), 8);
            if (ub.size() >= 8 && n <= (ub.size() - 8) / 12 && ub.size() == 8 + 12 * n) {
                unmapped.reserve((size_t)n);
                for (uint64_t i = 0; i < n; ++i) {
                    uint64_t k; uint32_t v;
                    std::memcpy(&k, ub.data() + 8 + 12 * i, 8); std::memcpy(&v, ub.data() + 16 + 12 * i, 4);
                    unmapped[k] += v;
                }
            } else if (!ub.empty())
                std::fprintf(stderr, "unmapped_bc_count_collated.bin is not in the (count, key/value pairs) layout; unmapped reads are taken as 0\n");
        }
    }

    // ---- the device work: the worker fan-out of do_quantify (quant.rs:1553-1575, 1678-1765) becomes one context and one
    // host thread per device over a contiguous, byte-balanced range of cells; no device talks to another ----
    pc.lap("prelude, chunk table, tg-map");
    if (warm.joinable()) warm.join();
    pc.lap("(wait for the HIP runtime)");
    std::vector<int> devices;
    if (o->devices && o->n_devices) devices.assign(o->devices, o->devices + o->n_devices); else devices.push_back((int)o->device);
    if (devices.size() > chunk_off.size() && !chunk_off.empty()) devices.resize(chunk_off.size());
    // The queue: contiguous batches of cells in file order (largest cells first, collate.rs:272-274).  One device: as large
    // as memory allows (batch_bytes).  Several: about eight batches per device, so that the devices finish together whatever
    // the cells cost (a parsimony cell's time grows faster than its bytes; static byte-balanced cuts gave the first device all
    // the giant cells) - every device thread pops the next batch when it has finished its last.
    const uint64_t batch_cap = o->batch_bytes ? o->batch_bytes : (16ull << 30);
    uint64_t total_bytes = 0;
    for (uint64_t nb : chunk_nb) total_bytes += nb;
    uint64_t batch_bytes = batch_cap;
    if (devices.size() > 1) batch_bytes = std::min<uint64_t>(batch_cap, std::max<uint64_t>(total_bytes / (8 * devices.size()) + 1, 32ull << 20));
    if (const char* e = afq::test_hook("QUEUE_BATCH_BYTES")) batch_bytes = std::max<uint64_t>(1, (uint64_t)std::atof(e));   // tests: many small batches
    std::vector<std::pair<size_t, size_t>> batches;
    for (size_t c0 = 0; c0 < chunk_off.size();) {
        size_t c1 = c0; uint64_t bytes = 0;
        while (c1 < chunk_off.size()) { const uint64_t nb = chunk_nb[c1]; if (c1 > c0 && bytes + nb > batch_bytes) break; bytes += nb; ++c1; }
        batches.emplace_back(c0, c1);
        c0 = c1;
    }
    if (devices.size() > batches.size() && !batches.empty()) devices.resize(batches.size());
    std::vector<DevOut> parts(batches.size());
    std::vector<DevStat> dstat(devices.size());
    {
        std::atomic<size_t> next{0};
        std::atomic<int> stop{0};
        auto work = [&](size_t d) {
            run_device_worker(cfg, cfg_eq.dump_eq ? &cfg_eq : nullptr, t2g, devices[d], devices.size() > 1, rad.data(), compressed ? -1 : mf.fd, chunk_off, chunk_nb, chunk_nr,
                              batches, next, stop, o->dump_eq != 0, res_is_em, parts, dstat[d]);
        };
        if (devices.size() == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (size_t d = 0; d < devices.size(); ++d) th.emplace_back(work, d);
            for (auto& x : th) x.join();
        }
    }
    for (size_t d = 0; d < dstat.size(); ++d)
        if (dstat[d].rc) return hfail(dstat[d].rc, (devices.size() > 1 ? "device " + std::to_string(devices[d]) + ": " : std::string()) + dstat[d].err);
    if (pc.on) for (size_t d = 0; d < dstat.size(); ++d)
        std::fprintf(stderr, "[afquant]   device %d: %u batches, %llu cells, %.3f GB, busy %.3f s\n", devices[d], dstat[d].batches, (unsigned long long)dstat[d].cells, (double)dstat[d].bytes / 1e9, dstat[d].busy_s);
    if (devices.size() > 1) {   // how the queue spread the work: a small extra file next to quant.json (not one of the reference's outputs)
        (void)mkdirs_checked(outd);   // (the writers create it further down; this file comes first)
        FilePtr df(std::fopen((outd + "/afquant_devices.json").c_str(), "w"));
        if (df) {
            std::fprintf(df.get(), "{\n  \"batches\": %zu,\n  \"batch_bytes\": %llu,\n  \"devices\": [\n", batches.size(), (unsigned long long)batch_bytes);
            for (size_t d = 0; d < dstat.size(); ++d)
                std::fprintf(df.get(), "    {\"device\": %d, \"batches\": %u, \"cells\": %llu, \"bytes\": %llu, \"busy_s\": %.6f}%s\n", devices[d], dstat[d].batches,
                             (unsigned long long)dstat[d].cells, (unsigned long long)dstat[d].bytes, dstat[d].busy_s, d + 1 < dstat.size() ? "," : "");
            std::fprintf(df.get(), "  ]\n}\n");
        }
    }
    pc.lap("device batches");

    // ---- host-side gather in cell order (the only communication the path has) ----
    const uint64_t row_index = chunk_off.size();
    std::vector<uint32_t> all_gene; std::vector<float> all_val;
    std::vector<uint64_t> row_ptr(1, 0), bc_all; std::vector<uint32_t> nrec_all; std::vector<uint8_t> flags_all;
    std::map<std::vector<uint32_t>, uint32_t> eq_ids;          // global_eqc: gene set -> class id, ids in order of first appearance
    std::vector<uint32_t> eq_col, eq_cnt;                       // cell_level_count
    std::vector<uint64_t> eq_row_ptr(1, 0);                     // cell_offset
    std::vector<uint32_t> bm_col, bv_col; std::vector<float> bm_val, bv_val;   // BootstrapHelper's mean / variance triplets, as CSR
    std::vector<uint64_t> bm_ptr(1, 0), bv_ptr(1, 0);
    auto part_nnz = [](const DevOut& d) -> size_t { return d.is_held ? (size_t)d.held.nnz : d.gene.size(); };
    const bool one_held = parts.size() == 1 && parts[0].is_held;
    if (parts.size() == 1) { all_gene.swap(parts[0].gene); all_val.swap(parts[0].val); }
    else {
        size_t tot = 0;
        for (auto& p2 : parts) tot += part_nnz(p2);
        all_gene.reserve(tot); all_val.reserve(tot);
    }
    row_ptr.reserve(row_index + 1); bc_all.reserve(row_index); nrec_all.reserve(row_index); flags_all.reserve(row_index);
    for (auto& p2 : parts) {
        const uint64_t g0 = row_ptr.back();
        if (parts.size() > 1) {
            if (p2.is_held) { all_gene.insert(all_gene.end(), p2.held.gene, p2.held.gene + p2.held.nnz); all_val.insert(all_val.end(), p2.held.val, p2.held.val + p2.held.nnz); }
            else { all_gene.insert(all_gene.end(), p2.gene.begin(), p2.gene.end()); all_val.insert(all_val.end(), p2.val.begin(), p2.val.end()); std::vector<uint32_t>().swap(p2.gene); std::vector<float>().swap(p2.val); }
        }
        for (uint64_t e : p2.row_end) row_ptr.push_back(g0 + e);
        bc_all.insert(bc_all.end(), p2.bc.begin(), p2.bc.end());
        nrec_all.insert(nrec_all.end(), p2.nrec.begin(), p2.nrec.end());
        flags_all.insert(flags_all.end(), p2.flags.begin(), p2.flags.end());
        if (o->num_bootstraps && !p2.bm_end.empty()) {
            const uint64_t m0 = bm_col.size(), v0 = bv_col.size();
            bm_col.insert(bm_col.end(), p2.bm_col.begin(), p2.bm_col.end()); bm_val.insert(bm_val.end(), p2.bm_val.begin(), p2.bm_val.end());
            bv_col.insert(bv_col.end(), p2.bv_col.begin(), p2.bv_col.end()); bv_val.insert(bv_val.end(), p2.bv_val.begin(), p2.bv_val.end());
            for (size_t i = 0; i < p2.bm_end.size(); ++i) { bm_ptr.push_back(m0 + p2.bm_end[i]); bv_ptr.push_back(v0 + p2.bv_end[i]); }
        }
        if (o->dump_eq) {   // the global dictionary is filled in cell order (quant.rs:1282-1307), so that ids do not depend on the device count
            std::vector<uint32_t> key;
            uint64_t k = 0;
            for (size_t i = 0; i < p2.eq_cell_end.size(); ++i) {
                for (; k < p2.eq_cell_end[i]; ++k) {
                    const uint64_t w0 = k ? p2.eq_label_end[k - 1] : 0, w1 = p2.eq_label_end[k];
                    key.assign(p2.eq_labels.begin() + (ptrdiff_t)w0, p2.eq_labels.begin() + (ptrdiff_t)w1);
                    auto it = eq_ids.find(key);
                    if (it == eq_ids.end()) it = eq_ids.emplace(key, (uint32_t)eq_ids.size()).first;
                    eq_col.push_back(it->second); eq_cnt.push_back(p2.eq_count[k]);
                }
                eq_row_ptr.push_back(eq_col.size());
            }
        }
    }
    if (bc_all.size() != row_index) return hfail(AFQ_ERR_STATE, "internal: the devices returned a different number of cells than were submitted");
    struct HeldResult { afq_result r{}; ~HeldResult() { afq_result_release(&r); } } keep;   // (a result outlives its context)
    if (one_held) { keep.r = parts[0].held; parts[0].is_held = false; }
    const uint32_t* const GG = one_held ? keep.r.gene : all_gene.data();
    const float* const V = one_held ? keep.r.val : all_val.data();
    parts.clear();

    // ---- per-cell rows: quants_mat_rows.txt + featureDump.txt (src/quant.rs:1150-1262), formatted by -t threads over
    // contiguous runs of cells and written in order ----
    if (mkdirs_checked(outd + "/alevin")) return hfail(AFQ_ERR_BAD_INPUT, "could not create the output directory " + outd + "/alevin");
    FilePtr rows_f(std::fopen((outd + "/alevin/quants_mat_rows.txt").c_str(), "w"));
    FilePtr feat_f(std::fopen((outd + "/featureDump.txt").c_str(), "w"));
    FilePtr cols_f(std::fopen((outd + "/alevin/quants_mat_cols.txt").c_str(), "w"));
    if (!rows_f || !feat_f || !cols_f) return hfail(AFQ_ERR_BAD_INPUT, "could not create the output files");
    const bool sample_cols = multi_bc && have_samples;   // sample-prefixed row labels + the sample_name column (quant.rs:1217-1262)
    std::fputs(have_samples ?   // the header goes by the manifest alone (quant.rs:1603-1613)
 "CB\tsample_name\tCorrectedReads\tMappedReads\tDeduplicatedReads\tMappingRate\tDedupRate\tMeanByMax\tNumGenesExpressed\tNumGenesOverMean\n"
                           : "CB\tCorrectedReads\tMappedReads\tDeduplicatedReads\tMappingRate\tDedupRate\tMeanByMax\tNumGenesExpressed\tNumGenesOverMean\n", feat_f.get());
    for (auto& g : gene_names) std::fprintf(cols_f.get(), "%s\n", g.c_str());
    if (usa) { for (auto& g : gene_names) std::fprintf(cols_f.get(), "%s-U\n", g.c_str()); for (auto& g : gene_names) std::fprintf(cols_f.get(), "%s-A\n", g.c_str()); }
    cols_f.reset();
    std::vector<uint64_t> alt_cells, empty_cells, tiny_cells;
    uint64_t total_records = 0;
    {
        const unsigned nth = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)(o->num_threads ? o->num_threads : std::thread::hardware_concurrency()), 64, row_index / 256 + 1}));
        std::vector<std::string> rows_txt(nth), feat_txt(nth);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nth; ++t)
            th.emplace_back([&, t]() {
                const uint64_t i0 = row_index * t / nth, i1 = row_index * (t + 1) / nth;
                std::string &rt = rows_txt[t], &ft = feat_txt[t];
                char num[64];
                for (uint64_t i = i0; i < i1; ++i) {
                    const uint64_t a = row_ptr[i], b = row_ptr[i + 1];
                    float sum = 0.0f, mx = 0.0f;  // src/quant.rs:1150-1171 (f32 sum in column order)
                    for (uint64_t k = a; k < b; ++k) { sum += V[k]; if (V[k] > mx) mx = V[k]; }
                    const uint32_t num_expr = (uint32_t)(b - a);
                    const uint32_t nrec = nrec_all[i];
                    const float dedup_rate = sum / (float)nrec;
                    const uint64_t cell_bc = cell_key_of(bc_all[i]);
                    uint64_t num_unmapped = 0;
                    if (!unmapped.empty()) { auto it = unmapped.find(cell_bc); if (it != unmapped.end()) num_unmapped = it->second; }
                    const float mapping_rate = (float)nrec / (float)(nrec + num_unmapped);
                    const float mean_expr = sum / (float)num_expr;
                    uint32_t over = 0;
                    for (uint64_t k = a; k < b; ++k) if (V[k] > mean_expr) ++over;
                    const float mean_by_max = mean_expr / mx;
                    const std::string bcs = bc_to_string(cell_bc, cblen);
                    const std::string* sn = nullptr;
                    if (sample_cols) { const uint64_t si = bc_all[i] & ((1ull << (8 * w_sample)) - 1); if (si < sample_names.size()) sn = &sample_names[(size_t)si]; }
                    if (sn) { rt += *sn; rt += '_'; }
                    rt += bcs; rt += '\n';
                    ft += bcs; ft += '\t';
                    if (sn) { ft += *sn; ft += '\t'; }
                    *put_u64(num, (unsigned long long)(nrec + num_unmapped)) = 0; ft += num; ft += '\t';
                    *put_u64(num, nrec) = 0; ft += num; ft += '\t';
                    format_f32(sum, num, sizeof num); ft += num; ft += '\t';
                    format_f32(mapping_rate, num, sizeof num); ft += num; ft += '\t';
                    format_f32(dedup_rate, num, sizeof num); ft += num; ft += '\t';
                    format_f32(mean_by_max, num, sizeof num); ft += num; ft += '\t';
                    *put_u64(num, num_expr) = 0; ft += num; ft += '\t';
                    *put_u64(num, over) = 0; ft += num; ft += '\n';
                }
            });
        for (auto& x : th) x.join();
        for (unsigned t = 0; t < nth; ++t) {
            std::fwrite(rows_txt[t].data(), 1, rows_txt[t].size(), rows_f.get());
            std::fwrite(feat_txt[t].data(), 1, feat_txt[t].size(), feat_f.get());
        }
        for (uint64_t i = 0; i < row_index; ++i) {
            if (flags_all[i] & AFQ_CELL_ALT_RES) alt_cells.push_back(i);
            if (flags_all[i] & AFQ_CELL_EMPTY) empty_cells.push_back(i);
            if (flags_all[i] & AFQ_CELL_TINY_PATH) tiny_cells.push_back(i);
            total_records += nrec_all[i];
        }
    }
    rows_f.reset(); feat_f.reset();
    pc.lap("gather + per-cell rows");
    // With --quant-subset the reference sizes the matrix (and reports num_quantified_cells) by the SUBSET's size, whether or
    // not every listed barcode occurs in the file (quant.rs:1529, 1836, 1918); rows exist only for the cells found.
    const uint64_t num_cells = o->filter_list ? (uint64_t)subset_size : row_index;
    auto write_mtx = [&](const std::string& path, uint64_t n_rows, uint64_t n_cols, const std::vector<uint64_t>& rp,
                         const std::vector<uint32_t>& cols, const std::vector<float>& vals) -> bool {
        return write_mtx_file(path, n_rows, n_cols, rp, cols.data(), vals.data(), vals.size(), o->num_threads);
    };
    if (!write_mtx_file(outd + "/alevin/quants_mat.mtx", num_cells, cfg.num_rows, row_ptr, GG, V, (size_t)row_ptr.back(), o->num_threads)) return hfail(AFQ_ERR_BAD_INPUT, "could not create quants_mat.mtx");
    pc.lap("quants_mat.mtx");
    // -b: bootstrap summary matrices, cells x num_rows (src/quant.rs:1850-1877; nothing is written when no cell had a bootstrap)
    if (o->num_bootstraps && !bm_val.empty()) {
        if (!write_mtx(outd + "/alevin/bootstraps_mean.mtx", num_cells, cfg.num_rows, bm_ptr, bm_col, bm_val) ||
            !write_mtx(outd + "/alevin/bootstraps_var.mtx", num_cells, cfg.num_rows, bv_ptr, bv_col, bv_val))
            return hfail(AFQ_ERR_BAD_INPUT, "could not write the bootstrap matrices");
        pc.lap("bootstraps_mean.mtx + bootstraps_var.mtx");
    }
    // -d: cells x gene-level equivalence classes + the classes' gene sets (write_eqc_counts, src/quant.rs:229-355)
    if (o->dump_eq) {
        std::vector<float> ev(eq_cnt.begin(), eq_cnt.end());
        if (!write_mtx(outd + "/alevin/geqc_counts.mtx", eq_row_ptr.size() - 1 /* the cells processed: geqmap.cell_offset.len(), quant.rs:248-251 */, eq_ids.size(), eq_row_ptr, eq_col, ev)) return hfail(AFQ_ERR_BAD_INPUT, "could not write geqc_counts.mtx");
        std::vector<const std::vector<uint32_t>*> by_id(eq_ids.size());
        for (auto& kv : eq_ids) by_id[kv.second] = &kv.first;
        std::string txt = std::to_string(cfg.num_rows) + "\n" + std::to_string(eq_ids.size()) + "\n";
        const uint32_t uo = cfg.num_rows / 3, ao = 2 * uo;
        for (size_t id = 0; id < by_id.size(); ++id) {
            const std::vector<uint32_t>& gl = *by_id[id];
            for (size_t k = 0; k < gl.size(); ++k) {
                uint32_t g = gl[k];
                if (usa) {   // S -> g/2, U -> g/2 + unspliced offset, S followed by its own U -> ambiguous (quant.rs:284-335)
                    if (k + 1 < gl.size() && (gl[k + 1] >> 1) == (g >> 1)) { g = (g >> 1) + ao; ++k; }
                    else g = (g & 1u) ? (g >> 1) + uo : (g >> 1);
                }
                txt += std::to_string(g); txt += '\t';
            }
            txt += std::to_string(id); txt += '\n';
        }
        gzFile gz = gzopen((outd + "/alevin/gene_eqclass.txt.gz").c_str(), "wb");
        if (!gz) return hfail(AFQ_ERR_BAD_INPUT, "could not write gene_eqclass.txt.gz");
        for (size_t off = 0; off < txt.size();) { const unsigned len = (unsigned)std::min<size_t>(txt.size() - off, 1u << 30); if (gzwrite(gz, txt.data() + off, len) <= 0) { gzclose(gz); return hfail(AFQ_ERR_BAD_INPUT, "could not write gene_eqclass.txt.gz"); } off += len; }
        gzclose(gz);
        pc.lap("geqc_counts.mtx + gene_eqclass.txt.gz");
    }
    // quant.json (src/quant.rs:1913-1933)
    {
        FILE* j = std::fopen((outd + "/quant.json").c_str(), "w");
        if (!j) return hfail(AFQ_ERR_BAD_INPUT, "could not create quant.json");
        auto list = [&](const std::vector<uint64_t>& v) { std::string s = "["; for (size_t i = 0; i < v.size(); ++i) { if (i) s += ", "; s += std::to_string(v[i]); } return s + "]"; };
        std::fprintf(j, "{\n  \"cmd\": \"%s\",\n  \"version_str\": \"afquant-hip 0.1 (alevin-fry 0.18.0 quant semantics)\",\n  \"resolution_strategy\": \"%s\",\n",
                     json_escape(o->cmdline ? o->cmdline : "").c_str(), R->debug);
        std::fprintf(j, "  \"num_quantified_cells\": %llu,\n  \"num_genes\": %u,\n  \"dump_eq\": %s,\n  \"usa_mode\": %s,\n", (unsigned long long)num_cells, cfg.num_rows, o->dump_eq ? "true" : "false", usa ? "true" : "false");
        std::fprintf(j, "  \"alt_resolved_cell_numbers\": %s,\n  \"empty_resolved_cell_numbers\": %s,\n  \"num_tiny_cell_resolved\": %zu,\n  \"tiny_cell_resolved_cell_numbers\": %s,\n",
                     list(alt_cells).c_str(), list(empty_cells).c_str(), tiny_cells.size(), list(tiny_cells).c_str());
        std::fprintf(j, "  \"total_records\": %llu,\n  \"quant_options\": {\n    \"input_dir\": \"%s\",\n    \"tg_map\": \"%s\",\n    \"output_dir\": \"%s\",\n    \"num_threads\": %u,\n    \"num_bootstraps\": %u,\n    \"init_uniform\": %s,\n    \"summary_stat\": %s,\n    \"dump_eq\": %s,\n    \"resolution\": \"%s\",\n    \"pug_exact_umi\": %s,\n    \"sa_model\": \"%s\",\n    \"small_thresh\": %u,\n    \"large_graph_thresh\": %u,\n    \"filter_list\": %s%s%s,\n    \"cmdline\": \"%s\"\n  }\n}\n",
                     (unsigned long long)total_records, json_escape(in).c_str(), json_escape(o->tg_map).c_str(), json_escape(outd).c_str(), o->num_threads, o->num_bootstraps, o->init_uniform ? "true" : "false", o->summary_stat ? "true" : "false", o->dump_eq ? "true" : "false",
                     R->debug, cfg.pug_exact_umi ? "true" : "false", cfg.sa_model == AFQ_SA_PREFER_AMBIG ? "PreferAmbiguity" : "WinnerTakeAll", o->small_thresh, large_thresh, o->filter_list ? "\"" : "", o->filter_list ? json_escape(o->filter_list).c_str() : "null", o->filter_list ? "\"" : "",
                     json_escape(o->cmdline ? o->cmdline : "").c_str());
        std::fclose(j);
    }
    return 0;
}

}  // extern "C"
