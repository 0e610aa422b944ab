// afq_synth.cpp — host-side synthetic collated-RAD generator (see include/afquant_synth.h).
// Counter-based: every draw is a hash of (seed, cell, read, stream), so cells are
// generated independently and in parallel, and the plan pass (sizes only) and the
// fill pass produce the same records.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/afquant_synth.h"

namespace {

inline uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline uint64_t h3(uint64_t a, uint64_t b, uint64_t c) { return mix(mix(mix(a) ^ b) ^ c); }
inline double u01(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }

struct Gen {
    const afq_synth_params& p;
    uint32_t n_spliced;
    explicit Gen(const afq_synth_params& p_) : p(p_), n_spliced(p_.num_genes * p_.txp_per_gene) {}

    // one record -> refs[0..na) ascending distinct, umi; returns na
    inline uint32_t record(uint64_t cell, uint32_t read, uint32_t n_mol, uint32_t* refs, uint32_t& umi) const {
        const uint64_t base = h3(p.seed, cell, read);
        const uint32_t mol = (uint32_t)(u01(mix(base ^ 1)) * n_mol);
        const uint64_t mk = h3(p.seed ^ 0x5555, cell, mol);
        const uint64_t umask = p.umi_len >= 16 ? 0xFFFFFFFFull : ((1ull << (2 * p.umi_len)) - 1);
        uint64_t u = mk & umask;
        uint32_t gene;
        if (p.zipf > 0) {
            const double x = u01(mix(mk ^ 2)), e = 1.0 + 3.0 * p.zipf;
            double xe;
            const int ei = (int)e;
            if ((double)ei == e && ei >= 1 && ei <= 64 && (ei & (ei - 1)) == 0) {  // power-of-two exponent: squarings
                xe = x;
                for (int q = ei; q > 1; q >>= 1) xe *= xe;
            } else xe = std::pow(x, e);
            gene = (uint32_t)std::min<double>(p.num_genes - 1, std::floor(p.num_genes * xe));
        } else gene = (uint32_t)(mix(mk ^ 2) % p.num_genes);
        const double rn = u01(mix(base ^ 3));
        uint32_t na = rn < p.p_na3 ? 3 : (rn < p.p_na3 + p.p_na2 ? 2 : 1);
        uint32_t n = 0;
        bool both = false;
        if (p.usa) {
            const double st = u01(mix(base ^ 4));
            if (st < p.p_unspliced) refs[n++] = n_spliced + gene;
            else {
                refs[n++] = gene * p.txp_per_gene + (uint32_t)(mix(base ^ 5) % p.txp_per_gene);
                if (st < p.p_unspliced + p.p_both) { both = true; refs[n++] = n_spliced + gene; if (na < 2) na = 2; }
            }
        } else refs[n++] = gene * p.txp_per_gene + (uint32_t)(mix(base ^ 5) % p.txp_per_gene);
        for (uint32_t k = both ? 2 : 1; k < na; ++k) {
            const uint64_t hk = mix(base ^ (16 + k));
            uint32_t g = u01(hk) < p.cross ? (uint32_t)(mix(hk ^ 7) % p.num_genes) : gene;
            uint32_t t = g * p.txp_per_gene + (uint32_t)(mix(hk ^ 8) % p.txp_per_gene);
            if (p.usa && u01(mix(hk ^ 9)) < p.p_unspliced) t = n_spliced + g;
            refs[n++] = t;
        }
        // sort + unique (n <= 3)
        std::sort(refs, refs + n);
        n = (uint32_t)(std::unique(refs, refs + n) - refs);
        if (u01(mix(base ^ 10)) < p.umi_err) {
            const uint32_t pos = (uint32_t)(mix(base ^ 11) % p.umi_len);
            const uint64_t delta = 1 + mix(base ^ 12) % 3;
            const uint64_t b = (u >> (2 * pos)) & 3;
            u = (u & ~(3ull << (2 * pos))) | (((b + delta) & 3) << (2 * pos));
        }
        umi = (uint32_t)u;
        return n;
    }
    inline uint32_t n_mol(uint32_t nrec) const {
        return std::max<uint32_t>(1, (uint32_t)std::llround((double)nrec * (1.0 - p.dup)));
    }
    inline uint32_t barcode(uint64_t cell) const { return (uint32_t)(mix(p.seed * 1000003ull + cell) ^ (cell * 2654435761ull)); }
};

void parallel_for(uint32_t n, uint32_t n_threads, const std::function<void(uint32_t, uint32_t)>& f) {
    n_threads = std::max<uint32_t>(1, std::min<uint32_t>(n_threads ? n_threads : std::thread::hardware_concurrency(), 256));
    std::vector<std::thread> th;
    // interleaved blocks so the large leading cells spread over threads
    for (uint32_t t = 0; t < n_threads; ++t)
        th.emplace_back([=, &f]() { for (uint32_t i = t; i < n; i += n_threads) f(i, t); });
    for (auto& x : th) x.join();
}

}  // namespace

extern "C" {

void afq_synth_dims(const afq_synth_params* p, uint32_t* ref_count, uint32_t* num_genes, uint32_t* num_rows) {
    const uint32_t ns = p->num_genes * p->txp_per_gene;
    if (ref_count) *ref_count = p->usa ? ns + p->num_genes : ns;
    if (num_genes) *num_genes = p->usa ? 2 * p->num_genes : p->num_genes;
    if (num_rows) *num_rows = p->usa ? 3 * p->num_genes : p->num_genes;
}

void afq_synth_t2g(const afq_synth_params* p, uint32_t* t2g) {
    const uint32_t ns = p->num_genes * p->txp_per_gene;
    for (uint32_t t = 0; t < ns; ++t) t2g[t] = p->usa ? 2 * (t / p->txp_per_gene) : t / p->txp_per_gene;
    if (p->usa) for (uint32_t g = 0; g < p->num_genes; ++g) t2g[ns + g] = 2 * g + 1;
}

int afq_synth_plan(const afq_synth_params* p, uint32_t* cell_nrec, uint64_t* chunk_off, uint64_t* total_bytes,
                   uint64_t* total_reads) {
    if (!p || !cell_nrec || !chunk_off || p->umi_len == 0 || p->umi_len > 16 || p->num_genes == 0 || p->txp_per_gene == 0)
        return -1;
    for (uint32_t c = 0; c < p->n_cells; ++c) {
        double v = p->median_reads;
        if (p->sigma > 0) {
            const double u1 = std::max(1e-12, u01(h3(p->seed, 0xCE11, c))), u2 = u01(h3(p->seed, 0xCE12, c));
            const double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
            v = std::exp(std::log(p->median_reads) + p->sigma * z);
        }
        cell_nrec[c] = (uint32_t)std::max<double>(p->min_reads ? p->min_reads : 1, std::min(4.0e9, std::round(v)));
    }
    std::sort(cell_nrec, cell_nrec + p->n_cells, std::greater<uint32_t>());
    Gen g(*p);
    std::vector<uint64_t> sizes(p->n_cells);
    parallel_for(p->n_cells, p->n_threads, [&](uint32_t c, uint32_t) {
        uint32_t refs[4], umi;
        const uint32_t nm = g.n_mol(cell_nrec[c]);
        uint64_t words = 2;
        for (uint32_t r = 0; r < cell_nrec[c]; ++r) words += 3 + g.record(c, r, nm, refs, umi);
        sizes[c] = words * 4;
    });
    uint64_t off = 0, reads = 0;
    for (uint32_t c = 0; c < p->n_cells; ++c) {
        if (sizes[c] > 0xFFFFFFFFull) return -2;  // nbytes is u32 on the wire
        chunk_off[c] = off; off += sizes[c]; reads += cell_nrec[c];
    }
    if (total_bytes) *total_bytes = off;
    if (total_reads) *total_reads = reads;
    return 0;
}

int afq_synth_fill(const afq_synth_params* p, const uint32_t* cell_nrec, const uint64_t* chunk_off, uint8_t* out,
                   uint64_t total_bytes) {
    if (!p || !cell_nrec || !chunk_off || !out) return -1;
    Gen g(*p);
    std::vector<int> bad(1, 0);
    parallel_for(p->n_cells, p->n_threads, [&](uint32_t c, uint32_t) {
        uint32_t* w = reinterpret_cast<uint32_t*>(out + chunk_off[c]);
        const uint64_t lim = (c + 1 < p->n_cells ? chunk_off[c + 1] : total_bytes) - chunk_off[c];
        const uint32_t nm = g.n_mol(cell_nrec[c]);
        const uint32_t bc = g.barcode(c);
        uint64_t k = 2;
        uint32_t refs[4], umi;
        for (uint32_t r = 0; r < cell_nrec[c]; ++r) {
            const uint32_t na = g.record(c, r, nm, refs, umi);
            if ((k + 3 + na) * 4 > lim) { bad[0] = 1; return; }
            w[k++] = na; w[k++] = bc; w[k++] = umi;
            for (uint32_t j = 0; j < na; ++j) w[k++] = refs[j] | 0x80000000u;
        }
        w[0] = (uint32_t)(k * 4);
        w[1] = cell_nrec[c];
        if (k * 4 != lim) bad[0] = 1;
    });
    return bad[0] ? -3 : 0;
}

}  // extern "C"
