// afq_synth.hip — synthetic collated-RAD generator (include/afquant_synth.h): a multi-threaded host
// implementation and a gfx950 one over the same integer record model (afq_synth_model.h), so that a
// rank can produce its shard of a data set directly in HBM (configs[3] is ~350 GB of RAD as a whole)
// and tests can check the device bytes against the host bytes.  Bench / test tooling.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/afquant_synth.h"
#include "afq_synth_model.h"

using afq_synth::Model;

namespace {

inline uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline double u01(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }
inline uint32_t thr32(double p) { return p <= 0 ? 0u : p >= 1 ? 0xFFFFFFFFu : (uint32_t)std::floor(p * 4294967296.0); }

struct HostModel {
    Model m{};
    std::vector<uint32_t> thr, idx;
};

bool params_ok(const afq_synth_params* p) {
    return p && p->umi_len > 0 && p->umi_len <= 16 && p->num_genes > 0 && (p->ref_count ? p->ref_count >= p->num_genes : p->txp_per_gene > 0);
}

// thresholds + Walker/Vose alias table of the gene popularity
void build_model(const afq_synth_params& p, HostModel& H) {
    Model& m = H.m;
    m.k0 = (uint32_t)p.seed; m.k1 = (uint32_t)(p.seed >> 32);
    m.num_genes = p.num_genes;
    const uint64_t rc = p.ref_count ? p.ref_count : (uint64_t)p.num_genes * p.txp_per_gene;
    m.tpg = (uint32_t)(rc / p.num_genes); m.tx_big = (uint32_t)(rc % p.num_genes); m.n_spliced = (uint32_t)rc;
    m.usa = p.usa ? 1u : 0u;
    m.umi_len = p.umi_len; m.umi_mask = p.umi_len >= 16 ? 0xFFFFFFFFu : ((1u << (2 * p.umi_len)) - 1u);
    m.mol_q = thr32(1.0 - p.dup);
    m.thr_na3 = thr32(p.p_na3); m.thr_na23 = thr32(p.p_na3 + p.p_na2);
    m.thr_cross = thr32(p.cross); m.thr_umi_err = thr32(p.umi_err);
    m.thr_unspl = thr32(p.p_unspliced); m.thr_unspl_both = thr32(p.p_unspliced + p.p_both);
    m.bc_salt = (uint32_t)mix(p.seed * 1000003ull + 0xBCull);
    m.thr_tail = thr32(p.tail);
    m.tail_max = p.tail_max ? std::min<uint32_t>(p.tail_max, afq_synth::kMaxRefs) : afq_synth::kMaxRefs;
    m.family = std::max<uint32_t>(1, std::min<uint32_t>(p.family ? p.family : 8, p.num_genes));
    const uint32_t G = p.num_genes;
    std::vector<double> w(G);
    double tot = 0;
    for (uint32_t g = 0; g < G; ++g) {
        if (p.pow_skew > 0) w[g] = std::pow((double)(g + 1) / G, 1.0 / p.pow_skew) - std::pow((double)g / G, 1.0 / p.pow_skew);
        else if (p.zipf > 0) w[g] = std::pow((double)(g + 1), -p.zipf);
        else w[g] = 1.0;
        tot += w[g];
    }
    H.thr.assign(G, 0xFFFFFFFFu);
    H.idx.resize(G);
    std::vector<uint32_t> small, large;
    for (uint32_t g = 0; g < G; ++g) { w[g] = w[g] / tot * G; H.idx[g] = g; (w[g] < 1.0 ? small : large).push_back(g); }
    while (!small.empty() && !large.empty()) {
        const uint32_t s = small.back(), l = large.back();
        small.pop_back();
        H.thr[s] = thr32(w[s]); H.idx[s] = l;
        w[l] = (w[l] + w[s]) - 1.0;
        if (w[l] < 1.0) { large.pop_back(); small.push_back(l); }
    }
    m.alias_thr = H.thr.data(); m.alias_idx = H.idx.data();
}

void parallel_for(uint32_t n, uint32_t n_threads, const std::function<void(uint32_t)>& f) {
    n_threads = std::max<uint32_t>(1, std::min<uint32_t>(n_threads ? n_threads : std::thread::hardware_concurrency(), 256));
    n_threads = std::min(n_threads, std::max(n, 1u));
    std::vector<std::thread> th;
    // interleaved so the large leading cells spread over threads
    for (uint32_t t = 0; t < n_threads; ++t)
        th.emplace_back([=, &f]() { for (uint32_t i = t; i < n; i += n_threads) f(i); });
    for (auto& x : th) x.join();
}

// ---- device side -----------------------------------------------------------------------------------------------
constexpr uint32_t kTile = 1024;  // records per workgroup
constexpr uint32_t kNT = 256;

__device__ __forceinline__ uint32_t cell_of_tile(const uint32_t* __restrict__ tile_prefix, uint32_t n_cells, uint32_t tile) {
    uint32_t lo = 0, hi = n_cells;   // largest c with tile_prefix[c] <= tile
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tile_prefix[mid] <= tile) lo = mid; else hi = mid; }
    return lo;
}

// pass 1: dwords of every tile of 1024 records
__global__ __launch_bounds__(kNT) void k_synth_sizes(Model m, uint64_t first_cell, const uint32_t* __restrict__ cell_nrec,
                                                     const uint32_t* __restrict__ tile_prefix, uint32_t n_cells,
                                                     uint32_t* __restrict__ tile_words) {
    __shared__ uint32_t s_sum[kNT / 64];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    const uint32_t c = cell_of_tile(tile_prefix, n_cells, tile);
    const uint32_t nrec = cell_nrec[c], n_mol = afq_synth::num_molecules(m, nrec);
    const uint32_t r0 = (tile - tile_prefix[c]) * kTile;
    uint32_t sum = 0, refs[afq_synth::kMaxRefs], umi;
    for (uint32_t k = 0; k < kTile / kNT; ++k) {
        const uint32_t r = r0 + k * kNT + tid;
        if (r < nrec) sum += 3u + afq_synth::record<false>(m, first_cell + c, r, n_mol, refs, umi);
    }
    for (int d = 32; d; d >>= 1) sum += __shfl_xor(sum, d);
    if ((tid & 63) == 0) s_sum[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) tile_words[tile] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// pass 2: the records, staged per 256 in LDS and written out as one contiguous run
__global__ __launch_bounds__(kNT) void k_synth_fill(Model m, uint64_t first_cell, const uint32_t* __restrict__ cell_nrec,
                                                    const uint32_t* __restrict__ tile_prefix, uint32_t n_cells,
                                                    const uint64_t* __restrict__ tile_off, const uint32_t* __restrict__ cell_words,
                                                    uint32_t* __restrict__ out) {
    __shared__ uint32_t s_stage[kNT * 6];
    __shared__ uint32_t s_wave[kNT / 64];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t c = cell_of_tile(tile_prefix, n_cells, tile);
    const uint32_t nrec = cell_nrec[c], n_mol = afq_synth::num_molecules(m, nrec);
    const uint32_t r0 = (tile - tile_prefix[c]) * kTile;
    const uint32_t bc = afq_synth::barcode(m, first_cell + c);
    uint64_t base = tile_off[tile];   // dword index in `out`
    if (r0 == 0 && tid == 0) { out[base - 2] = cell_words[c] * 4u; out[base - 1] = nrec; }
    for (uint32_t k = 0; k < kTile / kNT; ++k) {
        const uint32_t r = r0 + k * kNT + tid;
        uint32_t refs[afq_synth::kMaxRefs], umi = 0, na = 0, sz = 0;
        if (r < nrec) { na = afq_synth::record<true>(m, first_cell + c, r, n_mol, refs, umi); sz = 3u + na; }
        uint32_t inc = sz;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d); if ((int)lane >= d) inc += t; }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
        for (uint32_t w = 0; w < kNT / 64; ++w) { const uint32_t v = s_wave[w]; if (w < wave) wbase += v; total += v; }
        if (m.thr_tail) {   // (records of up to 67 dwords: no staging, every lane writes its own)
            if (sz) {
                uint64_t o = base + wbase + inc - sz;
                out[o++] = na; out[o++] = bc; out[o++] = umi;
                for (uint32_t j = 0; j < na; ++j) out[o++] = refs[j] | 0x80000000u;
            }
            base += total;
            __syncthreads();
            continue;
        }
        if (sz) {
            uint32_t o = wbase + inc - sz;
            s_stage[o++] = na; s_stage[o++] = bc; s_stage[o++] = umi;
            for (uint32_t j = 0; j < na; ++j) s_stage[o++] = refs[j] | 0x80000000u;
        }
        __syncthreads();
        for (uint32_t i = tid; i < total; i += kNT) out[base + i] = s_stage[i];
        base += total;
        __syncthreads();
    }
}

}  // namespace

extern "C" {

void afq_synth_dims(const afq_synth_params* p, uint32_t* ref_count, uint32_t* num_genes, uint32_t* num_rows) {
    const uint32_t ns = p->ref_count ? p->ref_count : p->num_genes * p->txp_per_gene;
    if (ref_count) *ref_count = p->usa ? ns + p->num_genes : ns;
    if (num_genes) *num_genes = p->usa ? 2 * p->num_genes : p->num_genes;
    if (num_rows) *num_rows = p->usa ? 3 * p->num_genes : p->num_genes;
}

void afq_synth_t2g(const afq_synth_params* p, uint32_t* t2g) {
    HostModel H;
    build_model(*p, H);
    const Model& m = H.m;
    for (uint32_t g = 0; g < m.num_genes; ++g) {
        const uint32_t t0 = afq_synth::first_txp(m, g), nt = afq_synth::num_txp(m, g);
        for (uint32_t t = t0; t < t0 + nt; ++t) t2g[t] = p->usa ? 2 * g : g;
    }
    if (p->usa) for (uint32_t g = 0; g < m.num_genes; ++g) t2g[m.n_spliced + g] = 2 * g + 1;
}

int afq_synth_cell_sizes(const afq_synth_params* p, uint32_t* cell_nrec) {
    if (!params_ok(p) || !cell_nrec) return -1;
    for (uint32_t c = 0; c < p->n_cells; ++c) {
        double v = p->median_reads;
        if (p->sigma > 0) {
            const double u1 = std::max(1e-12, u01(mix(mix(mix(p->seed) ^ 0xCE11) ^ c))), u2 = u01(mix(mix(mix(p->seed) ^ 0xCE12) ^ c));
            const double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
            v = std::exp(std::log(p->median_reads) + p->sigma * z);
        }
        cell_nrec[c] = (uint32_t)std::max<double>(p->min_reads ? p->min_reads : 1, std::min(4.0e9, std::round(v)));
    }
    std::sort(cell_nrec, cell_nrec + p->n_cells, std::greater<uint32_t>());
    return 0;
}

int afq_synth_host_plan(const afq_synth_params* p, uint64_t first_cell, uint32_t n, const uint32_t* cell_nrec,
                        uint64_t* chunk_off, uint64_t* total_bytes) {
    if (!params_ok(p) || (n && (!cell_nrec || !chunk_off))) return -1;
    HostModel H;
    build_model(*p, H);
    std::vector<uint64_t> sizes(n);
    parallel_for(n, p->n_threads, [&](uint32_t c) {
        uint32_t refs[afq_synth::kMaxRefs], umi;
        const uint32_t nm = afq_synth::num_molecules(H.m, cell_nrec[c]);
        uint64_t words = 2;
        for (uint32_t r = 0; r < cell_nrec[c]; ++r) words += 3 + afq_synth::record<false>(H.m, first_cell + c, r, nm, refs, umi);
        sizes[c] = words * 4;
    });
    uint64_t off = 0;
    for (uint32_t c = 0; c < n; ++c) {
        if (sizes[c] > 0xFFFFFFFFull) return -2;  // nbytes is u32 on the wire
        chunk_off[c] = off; off += sizes[c];
    }
    if (total_bytes) *total_bytes = off;
    return 0;
}

int afq_synth_host_fill(const afq_synth_params* p, uint64_t first_cell, uint32_t n, const uint32_t* cell_nrec,
                        const uint64_t* chunk_off, uint8_t* out, uint64_t total_bytes) {
    if (!params_ok(p) || (n && (!cell_nrec || !chunk_off || !out))) return -1;
    HostModel H;
    build_model(*p, H);
    int bad = 0;
    parallel_for(n, p->n_threads, [&](uint32_t c) {
        uint32_t* w = reinterpret_cast<uint32_t*>(out + chunk_off[c]);
        const uint64_t lim = (c + 1 < n ? chunk_off[c + 1] : total_bytes) - chunk_off[c];
        const uint32_t nm = afq_synth::num_molecules(H.m, cell_nrec[c]);
        const uint32_t bc = afq_synth::barcode(H.m, first_cell + c);
        uint64_t k = 2;
        uint32_t refs[afq_synth::kMaxRefs], umi;
        for (uint32_t r = 0; r < cell_nrec[c]; ++r) {
            const uint32_t na = afq_synth::record<true>(H.m, first_cell + c, r, nm, refs, umi);
            if ((k + 3 + na) * 4 > lim) { bad = 1; return; }
            w[k++] = na; w[k++] = bc; w[k++] = umi;
            for (uint32_t j = 0; j < na; ++j) w[k++] = refs[j] | 0x80000000u;
        }
        w[0] = (uint32_t)(k * 4);
        w[1] = cell_nrec[c];
        if (k * 4 != lim) bad = 1;
    });
    return bad ? -3 : 0;
}

int afq_synth_plan(const afq_synth_params* p, uint32_t* cell_nrec, uint64_t* chunk_off, uint64_t* total_bytes, uint64_t* total_reads) {
    int rc = afq_synth_cell_sizes(p, cell_nrec);
    if (rc) return rc;
    rc = afq_synth_host_plan(p, 0, p->n_cells, cell_nrec, chunk_off, total_bytes);
    if (rc) return rc;
    if (total_reads) { uint64_t r = 0; for (uint32_t c = 0; c < p->n_cells; ++c) r += cell_nrec[c]; *total_reads = r; }
    return 0;
}

int afq_synth_fill(const afq_synth_params* p, const uint32_t* cell_nrec, const uint64_t* chunk_off, uint8_t* out, uint64_t total_bytes) {
    return afq_synth_host_fill(p, 0, p->n_cells, cell_nrec, chunk_off, out, total_bytes);
}

}  // extern "C" (reopened below)

// ---- scATAC (BASELINE configs[4]): host generator of a collated scATAC RAD body ----
// Record i of cell c: with p_unmapped no alignment, with p_multi two alignments, else one properly mapped pair
// (type 4) - a copy of record i-1's fragment with p_dup, else chr uniform, start uniform below ref_len, frag_len
// log-normal clipped to [30, 2500].  Records: na:u32, bc:u32, na x {ref:u32, type:u8, start_pos:u32, frag_len:u16}.
namespace {
struct AtacRec { uint32_t na; uint32_t ref[2], start[2]; uint16_t flen[2]; };
inline void atac_fragment(const afq_synth_atac_params& p, uint64_t cell, uint32_t i, uint32_t k0, uint32_t k1, uint32_t& ref, uint32_t& start, uint16_t& flen) {
    // walk back over the duplicate chain (every link is decided by its own record's draw)
    for (;;) {
        uint32_t w[4];
        afq_synth::philox(i, (uint32_t)cell, (uint32_t)(cell >> 32), 0x41544143u, k0, k1, w);
        if (i > 0 && w[0] < thr32(p.p_dup)) { --i; continue; }
        uint32_t q[4];
        afq_synth::philox(i, (uint32_t)cell, (uint32_t)(cell >> 32), 0x41544144u, k0, k1, q);
        ref = afq_synth::below(q[0], p.n_refs);
        start = afq_synth::below(q[1], p.ref_len);
        const double u1 = std::max(1e-12, (q[2] + 0.5) / 4294967296.0), u2 = (q[3] + 0.5) / 4294967296.0;
        const double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
        flen = (uint16_t)std::min(2500.0, std::max(30.0, std::exp(p.flen_mu + p.flen_sigma * z)));
        return;
    }
}
inline void atac_record(const afq_synth_atac_params& p, uint64_t cell, uint32_t i, uint32_t k0, uint32_t k1, AtacRec& r) {
    uint32_t w[4];
    afq_synth::philox(i, (uint32_t)cell, (uint32_t)(cell >> 32), 0x41544142u, k0, k1, w);
    const uint32_t tu = thr32(p.p_unmapped), tm = thr32(p.p_unmapped + p.p_multi);
    if (w[0] < tu) { r.na = 0; return; }
    atac_fragment(p, cell, i, k0, k1, r.ref[0], r.start[0], r.flen[0]);
    if (w[0] < tm) { r.na = 2; r.ref[1] = (r.ref[0] + 1) % p.n_refs; r.start[1] = r.start[0] + 250; r.flen[1] = r.flen[0]; }
    else r.na = 1;
}
}  // namespace

extern "C" int afq_synth_atac(const afq_synth_atac_params* p, uint64_t* chunk_off, uint64_t* total_bytes, uint8_t* out) {
    if (!p || !chunk_off || !total_bytes || p->n_refs == 0 || p->ref_len == 0) return -1;
    const uint32_t k0 = (uint32_t)p->seed, k1 = (uint32_t)(p->seed >> 32);
    Model bm{}; bm.bc_salt = (uint32_t)mix(p->seed * 1000003ull + 0xBCull);
    if (!out) {   // plan: sizes only
        std::vector<uint64_t> sz(p->n_cells);
        parallel_for(p->n_cells, p->n_threads, [&](uint32_t c) {
            uint64_t b = 8;
            AtacRec r;
            for (uint32_t i = 0; i < p->frags_per_cell; ++i) { atac_record(*p, c, i, k0, k1, r); b += 8 + 11ull * r.na; }
            sz[c] = b;
        });
        uint64_t off = 0;
        for (uint32_t c = 0; c < p->n_cells; ++c) { if (sz[c] > 0xFFFFFFFFull) return -2; chunk_off[c] = off; off += sz[c]; }
        *total_bytes = off;
        return 0;
    }
    parallel_for(p->n_cells, p->n_threads, [&](uint32_t c) {
        uint8_t* w = out + chunk_off[c];
        const uint32_t bc = afq_synth::barcode(bm, c);
        uint64_t k = 8;
        AtacRec r;
        for (uint32_t i = 0; i < p->frags_per_cell; ++i) {
            atac_record(*p, c, i, k0, k1, r);
            std::memcpy(w + k, &r.na, 4); std::memcpy(w + k + 4, &bc, 4); k += 8;
            for (uint32_t j = 0; j < r.na; ++j) {
                const uint8_t ty = 4;
                std::memcpy(w + k, &r.ref[j], 4); w[k + 4] = ty; std::memcpy(w + k + 5, &r.start[j], 4); std::memcpy(w + k + 9, &r.flen[j], 2);
                k += 11;
            }
        }
        const uint32_t nb = (uint32_t)k, nr = p->frags_per_cell;
        std::memcpy(w, &nb, 4); std::memcpy(w + 4, &nr, 4);
    });
    return 0;
}

extern "C" {

int afq_synth_device_generate(const afq_synth_params* p, int device, uint64_t first_cell, uint32_t n, const uint32_t* cell_nrec,
                              uint64_t* chunk_off, uint64_t* total_bytes, void** d_bytes) {
    if (!params_ok(p) || !d_bytes || !total_bytes || (n && (!cell_nrec || !chunk_off))) return -1;
    *d_bytes = nullptr;
    if (hipSetDevice(device) != hipSuccess) return -4;
    HostModel H;
    build_model(*p, H);
    std::vector<uint32_t> tile_prefix(n + 1);
    uint64_t n_tiles = 0;
    for (uint32_t c = 0; c < n; ++c) { tile_prefix[c] = (uint32_t)n_tiles; n_tiles += (cell_nrec[c] + kTile - 1) / kTile; }
    if (n_tiles >= 0x7FFFFFFFull) return -2;
    tile_prefix[n] = (uint32_t)n_tiles;
    const uint32_t G = p->num_genes;
    uint32_t *d_thr = nullptr, *d_idx = nullptr, *d_nrec = nullptr, *d_tp = nullptr, *d_tw = nullptr, *d_cw = nullptr;
    uint64_t* d_toff = nullptr;
    void* d_out = nullptr;
    hipError_t e = hipSuccess;
    auto T = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    auto cleanup = [&]() {
        for (void* q : {(void*)d_thr, (void*)d_idx, (void*)d_nrec, (void*)d_tp, (void*)d_tw, (void*)d_cw, (void*)d_toff}) if (q) (void)hipFree(q);
    };
    const size_t n1 = std::max<size_t>(n, 1), t1 = std::max<size_t>(n_tiles, 1);
    T(hipMalloc((void**)&d_thr, 4ull * G)); T(hipMalloc((void**)&d_idx, 4ull * G));
    T(hipMalloc((void**)&d_nrec, 4 * n1)); T(hipMalloc((void**)&d_tp, 4 * (n1 + 1)));
    T(hipMalloc((void**)&d_tw, 4 * t1)); T(hipMalloc((void**)&d_cw, 4 * n1)); T(hipMalloc((void**)&d_toff, 8 * t1));
    if (e == hipSuccess) {
        T(hipMemcpy(d_thr, H.thr.data(), 4ull * G, hipMemcpyHostToDevice));
        T(hipMemcpy(d_idx, H.idx.data(), 4ull * G, hipMemcpyHostToDevice));
        if (n) T(hipMemcpy(d_nrec, cell_nrec, 4ull * n, hipMemcpyHostToDevice));
        T(hipMemcpy(d_tp, tile_prefix.data(), 4ull * (n + 1), hipMemcpyHostToDevice));
    }
    Model m = H.m;
    m.alias_thr = d_thr; m.alias_idx = d_idx;
    std::vector<uint32_t> tw(n_tiles), cw(n);
    std::vector<uint64_t> toff(n_tiles);
    if (e == hipSuccess && n_tiles) {
        k_synth_sizes<<<(uint32_t)n_tiles, kNT>>>(m, first_cell, d_nrec, d_tp, n, d_tw);
        T(hipGetLastError());
        T(hipMemcpy(tw.data(), d_tw, 4 * n_tiles, hipMemcpyDeviceToHost));
    }
    if (e != hipSuccess) { cleanup(); return e == hipErrorOutOfMemory ? -5 : -4; }
    uint64_t off = 0;
    for (uint32_t c = 0; c < n; ++c) {
        uint64_t words = 2;
        for (uint32_t t = tile_prefix[c]; t < tile_prefix[c + 1]; ++t) { toff[t] = off / 4 + words; words += tw[t]; }
        if (words * 4 > 0xFFFFFFFFull) { cleanup(); return -2; }
        chunk_off[c] = off; cw[c] = (uint32_t)words; off += words * 4;
    }
    *total_bytes = off;
    T(hipMalloc(&d_out, off + 16));
    if (e == hipSuccess) {
        T(hipMemset((uint8_t*)d_out + off, 0, 16));
        if (n_tiles) {
            T(hipMemcpy(d_toff, toff.data(), 8 * n_tiles, hipMemcpyHostToDevice));
            T(hipMemcpy(d_cw, cw.data(), 4ull * n, hipMemcpyHostToDevice));
            k_synth_fill<<<(uint32_t)n_tiles, kNT>>>(m, first_cell, d_nrec, d_tp, n, d_toff, d_cw, (uint32_t*)d_out);
            T(hipGetLastError());
        }
        T(hipDeviceSynchronize());
    }
    cleanup();
    if (e != hipSuccess) { if (d_out) (void)hipFree(d_out); return e == hipErrorOutOfMemory ? -5 : -4; }
    *d_bytes = d_out;
    return 0;
}

void afq_synth_device_free(int device, void* d_bytes) {
    if (!d_bytes) return;
    (void)hipSetDevice(device);
    (void)hipFree(d_bytes);
}

int afq_synth_device_read(int device, const void* d_src, uint64_t n, void* host_dst) {
    if (hipSetDevice(device) != hipSuccess) return -4;
    return hipMemcpy(host_dst, d_src, n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -4;
}

}  // extern "C"
