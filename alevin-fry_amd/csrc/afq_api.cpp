// afq_api.cpp — C ABI of include/afquant.h: host planner + stream orchestration.
//
// Replaces (reference paths relative to /root/reference):
//   worker set-up            src/quant.rs:1678-1765   -> afq_create
//   per-cell loop body       src/quant.rs:733-1322    -> afq_submit / afq_collect
// No CPU compute path exists here: every count is produced by the gfx950
// kernels in afq_kernels.hip; a missing device is an error, never a fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/afquant.h"
#include "afq_common.h"
#include "afq_hooks.h"
#include "afq_kernels.h"

using namespace afq;

namespace {

thread_local std::string g_create_err;

// AFQ_HOST_TIMING=1 prints where the host side of a batch spends its time (stderr)
struct HostClock {
    bool on = std::getenv("AFQ_HOST_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[afq host] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = n + n / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { e = hipMalloc(&p, n); want = n; }
        if (e == hipSuccess) cap = want; else p = nullptr;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

enum KernelId { K_GATHER = 0, K_DECODE_PAR, K_DECODE, K_HIST, K_BSCAN, K_SCATTER, K_RESOLVE, K_RESOLVE_BIG, K_PUG, K_CELL_HIST, K_EM, K_BOOT, K_COMPACT, K_ATAC, K_ATAC_PARSE, K_FIX_SLABS, K_P2_SPLIT, K_P2_PART, K_P2_SEARCH, K_P2_LONE, K_P2_GRAPH, K_COUNT };
const char* const kKernelNames[K_COUNT] = {"k_gather_headers", "k_decode_par", "k_decode", "k_hist", "k_bucket_scan", "k_scatter",
                                           "k_resolve", "k_resolve_big", "k_pug_cell", "k_cell_hist", "k_em", "k_boot", "k_compact", "k_atac_dedup", "k_atac_parse", "k_fix_slabs",
                                           "k_p2_split", "k_p2_part", "k_p2_search", "k_p2_lone", "k_p2_graph"};

struct TimedLaunch { int id; hipEvent_t a, b; bool own_a = true; };   // own_a false: a is the b of the bracket in front (TimerChain) - it goes back to the event pool once

// Pinned, grow-only host array (D2H lands here directly; handed to the caller zero-copy).
template <class T>
struct PinnedVec {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    hipError_t reserve(size_t want) {
        if (want <= cap) return hipSuccess;
        size_t nc = std::max(want + want / 4, (size_t)4096);
        T* q = nullptr;
        hipError_t e = hipHostMalloc((void**)&q, nc * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) return e;
        if (n) std::memcpy(q, p, n * sizeof(T));
        if (p) (void)hipHostFree(p);
        p = q; cap = nc;
        return hipSuccess;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = cap = 0; }
};

struct ResultPool;
struct HostResult {
    std::vector<uint64_t> cell_ptr, bc;
    std::vector<uint32_t> nrec;
    PinnedVec<uint32_t> gene;
    PinnedVec<float> val;
    std::vector<uint8_t> flags;
    // -d: gene-level equivalence classes per cell (cfg.dump_eq): CSR cell -> classes -> label words
    std::vector<uint64_t> eq_cell_ptr, eq_label_ptr;
    std::vector<uint32_t> eq_labels, eq_count;
    // -b: bootstrap mean / variance per cell, non-zero entries (cfg.num_bootstraps)
    std::vector<uint64_t> bm_ptr, bv_ptr;
    std::vector<uint32_t> bm_col, bv_col;
    std::vector<float> bm_val, bv_val;
    ResultPool* pool = nullptr;
    void clear() {
        cell_ptr.clear(); bc.clear(); nrec.clear(); flags.clear(); gene.n = 0; val.n = 0;
        eq_cell_ptr.clear(); eq_label_ptr.clear(); eq_labels.clear(); eq_count.clear();
        bm_ptr.clear(); bv_ptr.clear(); bm_col.clear(); bv_col.clear(); bm_val.clear(); bv_val.clear();
    }
};

// Results outlive a collect call (library-owned until afq_result_release), and pinning
// host memory is slow, so result storage is recycled through a small per-context pool.
struct ResultPool {
    std::mutex mu;
    std::vector<HostResult*> free_list;
    bool ctx_alive = true;
    int outstanding = 0;
};

HostResult* pool_get(ResultPool* P) {
    std::lock_guard<std::mutex> g(P->mu);
    HostResult* r;
    if (!P->free_list.empty()) { r = P->free_list.back(); P->free_list.pop_back(); }
    else { r = new HostResult(); r->pool = P; }
    r->clear();
    ++P->outstanding;
    return r;
}

void pool_put(HostResult* r) {
    ResultPool* P = r->pool;
    bool destroy_pool = false;
    {
        std::lock_guard<std::mutex> g(P->mu);
        --P->outstanding;
        if (P->ctx_alive && P->free_list.size() < 2) { P->free_list.push_back(r); r = nullptr; }
        else destroy_pool = !P->ctx_alive && P->outstanding == 0;
    }
    if (r) { r->gene.release(); r->val.release(); delete r; }
    if (destroy_pool) delete P;
}

struct Range { uint32_t c0, c1; };

struct RangeState {
    hipStream_t stream = nullptr;
    DevBuf d_meta, d_keys0, d_keys1, d_cell_nkeys, d_bucket_cnt, d_bucket_cell, d_multi_cells, d_tile_desc, d_src_off, d_slab_ovf, d_ncols,
        d_nnz, d_ovf, d_status, d_bc, d_cell_ptr, d_gene, d_val, d_chk, d_slab_prefix, d_slab_cell, d_cell_bc, d_bdesc, d_lab,
        d_lab_cnt, d_em_off, d_em_scratch, d_em_nnz, d_pug_cells, d_rd_off, d_rd_h, d_rd_u, d_rd_o, d_pug_scr_off,
        d_pug_scratch, d_epool, d_epool_cur, d_alt, d_hist_cells, d_fix, d_em_hdr, d_em_order, d_eq_ncls, d_eq_nw, d_eq_cptr,
        d_p2_small, d_eq_wptr, d_eq_len, d_eq_cnt, d_eq_lab, d_bt_off, d_bt_scratch, d_bt_ns, d_bt_col, d_bt_mean, d_bt_var, d_bt_sptr, d_bt_ccol,
        d_bt_cmean, d_bt_cvar, d_em2_off, d_em2_scratch, d_em2_tiers, d_arena;
    PinnedVec<uint8_t> h_arena;   // the range's small uploads, gathered (RangeInit)
    PinnedVec<uint32_t> h_pack;   // what the host reads when the range is done: k_pack_small writes it from the device
    ResolveArgs last_ra{};
    std::vector<CellMeta> meta;
    Range cur{};
    const PfDev* pf_stats_src = nullptr; uint64_t pf_stats_reads = 0, pf_stats_parts = 0;   // AFQ_TEST_PF_STATS
    uint32_t hash_try = 0;   // which salt the range's label hashes were made with (a collision re-runs the range under the next)
    uint32_t pool_try = 0;   // how often the range was run again with four times the parsimony pool (a cell's graph outgrew it)
    uint64_t att_records = 0, att_ref_words = 0, att_buckets = 0;   // what the current attempt added to the batch statistics
    bool chained = false;    // the rows' compaction was enqueued behind the range's kernels (row offsets made on the device, k_row_ptr) ...
    uint64_t chain_cap = 0;  // ... against d_gene / d_val of this many entries: finish_range compacts again, after growing them, if the range has more
    bool em_inline = false;  // the EM was enqueued behind the range's kernels (offsets made on the device); finish_range only checks that its scratch sufficed
    bool in_flight = false;
    bool pug_cell_launched = true;   // the range's k_pug_cell launch was made (else a handed-back cell means: run the range again)
    hipEvent_t kernels_done = nullptr;
    std::vector<TimedLaunch> launches;  // HIP-event brackets of this range's kernels (cfg.profile)
    std::vector<DevBuf*> all() {
        return {&d_meta, &d_keys0, &d_keys1, &d_cell_nkeys, &d_bucket_cnt, &d_bucket_cell, &d_multi_cells, &d_tile_desc, &d_src_off, &d_slab_ovf,
                &d_ncols, &d_nnz, &d_ovf, &d_status, &d_bc, &d_cell_ptr, &d_gene, &d_val, &d_chk, &d_slab_prefix, &d_slab_cell,
                &d_cell_bc, &d_bdesc, &d_lab, &d_lab_cnt, &d_em_off, &d_em_scratch, &d_em_nnz, &d_pug_cells, &d_rd_off, &d_rd_h,
                &d_rd_u, &d_rd_o, &d_pug_scr_off, &d_pug_scratch, &d_epool, &d_epool_cur, &d_p2_small, &d_alt, &d_hist_cells, &d_fix, &d_em_hdr, &d_em_order,
                &d_eq_ncls, &d_eq_nw, &d_eq_cptr, &d_eq_wptr, &d_eq_len, &d_eq_cnt, &d_eq_lab, &d_bt_off, &d_bt_scratch, &d_bt_ns, &d_bt_col,
                &d_bt_mean, &d_bt_var, &d_bt_sptr, &d_bt_ccol, &d_bt_cmean, &d_bt_cvar, &d_em2_off, &d_em2_scratch, &d_em2_tiers, &d_arena};
    }
};

}  // namespace

struct afq_ctx {
    afq_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t ref_count = 0;
    std::string err;
    std::mutex err_mu;   // (the upload thread of afq_submit reports through fail() too)
    DevBuf d_t2g;
    // input
    DevBuf d_bytes_own;
    const uint8_t* d_bytes = nullptr;
    size_t n_bytes = 0;
    DevBuf d_chunk_off, d_hdr;
    DevBuf atac[27];  // afq_atac_dedup[_rad]'s device buffers, kept between calls
    void* stage[3] = {nullptr, nullptr, nullptr};          // pinned staging for large host->device input copies
    hipEvent_t stage_ev[3] = {nullptr, nullptr, nullptr};
    // afq_submit: the input crosses PCIe range by range while earlier ranges already run (h2d_ev[i] = range i's bytes landed)
    std::vector<hipEvent_t> h2d_ev;
    bool h2d_piped = false;
    std::mutex up_mu;
    std::condition_variable up_cv;
    size_t up_enqueued = 0;
    int up_rc = 0;
    std::string up_err;
    // Two sets of per-range device state: while the rows of range i cross PCIe, the kernels of range i+1 run.
    RangeState rs[2];
    bool all_aligned = true;  // every chunk offset is a multiple of 4
    // 1/2-byte barcode or UMI fields: the batch is rewritten on the device with 4-byte fields (k_widen) and everything
    // downstream works on that copy - w_off / w_nbytes are the chunks of the copy, chunk_off / hdr stay the caller's
    bool widen = false;
    uint32_t eff_bc = 0, eff_umi = 0;
    std::vector<uint64_t> w_off;
    std::vector<uint32_t> w_nbytes;
    uint64_t wide_bytes = 0;
    DevBuf d_wide;
    ResultPool* pool = nullptr;
    // host planning state
    std::vector<uint64_t> chunk_off;
    std::vector<uint32_t> hdr;  // nbytes, nrec per cell
    std::vector<Range> ranges;
    size_t next_range = 0;
    uint64_t first_cell_index = 0;
    uint32_t n_cells = 0;
    bool pending = false;
    HostResult* res = nullptr;
    // stats / timers
    afq_batch_stats stats{};
    uint64_t n_label_rehash = 0;   // ranges decoded again under another label-hash salt (life of the context)
    uint64_t n_pool_regrow = 0;    // ranges run again with a larger parsimony pool
    uint64_t n_mono_cells = 0;     // parsimony cells resolved by the one-workgroup kernel (sent there directly, or handed back by the phase kernels)
    uint64_t n_em_resized = 0;     // ranges whose EM scratch was sized on the host after the device-side plan did not fit
    bool handback_seen = false;    // the phase kernels have handed a cell back to the one-workgroup kernel in some range of this context
    uint32_t retry_cuts = 0;       // how many times the range being finished has been cut around a failing cell (finish_range)
    std::vector<TimedLaunch> launches;
    std::vector<hipEvent_t> event_pool;
    double k_ms[K_COUNT] = {0};
    uint32_t k_launches[K_COUNT] = {0};
};

namespace {

int fail(afq_ctx* c, int code, const std::string& msg) {
    if (c) { std::lock_guard<std::mutex> g(c->err_mu); c->err = msg; } else g_create_err = msg;
    return code;
}
#define HIP_TRY(c, expr)                                                                          \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return fail((c), e__ == hipErrorOutOfMemory ? AFQ_ERR_OOM : AFQ_ERR_HIP,              \
                        std::string(#expr) + ": " + hipGetErrorString(e__));                      \
    } while (0)

hipEvent_t get_event(afq_ctx* c) {
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct ScopedTimer {
    afq_ctx* c; int id; hipStream_t s; std::vector<TimedLaunch>* sink; hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(afq_ctx* c_, int id_, hipStream_t s_ = nullptr, std::vector<TimedLaunch>* sink_ = nullptr)
        : c(c_), id(id_), s(s_ ? s_ : c_->stream), sink(sink_ ? sink_ : &c_->launches) {
        if (c->cfg.profile) { a = get_event(c); b = get_event(c); (void)hipEventRecord(a, s); }
    }
    ~ScopedTimer() {
        if (c->cfg.profile) { (void)hipEventRecord(b, s); sink->push_back({id, a, b}); }
    }
};

// The brackets of a range's kernels.  An event between two kernels of a stream is not free: the timeline of a configs[1] step
// (profiles/r04_timeline_configs1.txt) shows 10-14 us between two kernels wherever one bracket ended and the next began - two
// event packets - and nothing between kernels of one bracket.  At seven brackets per cr-like range, five ranges per step, that
// was ~0.35 ms of a 13.6 ms step spent on being timed.  A range's brackets therefore share their events - the end of one is the
// start of the next: ONE packet between two timed kernels - and the 5 us kernels around the large ones are timed with their
// neighbour (the proof's fix-up decode with the decoder, k_fix_slabs with the scatter, k_resolve_mid / k_resolve_big with
// k_resolve).
struct TimerChain {
    afq_ctx* c; hipStream_t s; std::vector<TimedLaunch>* sink; bool par;
    hipEvent_t last = nullptr; int id = -1; bool last_shared = false;   // last: the event the open bracket started on; shared: it also ended the bracket in front
    TimerChain(afq_ctx* c_, hipStream_t s_, std::vector<TimedLaunch>* sink_, bool par_) : c(c_), s(s_), sink(sink_), par(par_) {}
    int fold(int k) const {
        if (k == K_DECODE && par) return K_DECODE_PAR;   // (without the walk-free decoders k_decode IS the decode and keeps its name)
        if (k == K_FIX_SLABS) return K_SCATTER;
        if (k == K_RESOLVE_BIG) return K_RESOLVE;
        return k;
    }
    // the kernels enqueued from here on belong to bracket `next` (-1: to none)
    void seg(int next) {
        if (!c->cfg.profile) return;
        if (next >= 0) next = fold(next);
        if (next == id) return;
        hipEvent_t e = get_event(c);   // (next != id: a bracket ends here, or one begins, or both)
        (void)hipEventRecord(e, s);
        if (id >= 0) sink->push_back({id, last, e, !last_shared});
        last_shared = id >= 0;
        last = next >= 0 ? e : nullptr;
        id = next;
    }
    void end() { seg(-1); }
};
void recycle_events(afq_ctx* c, const TimedLaunch& t) {
    if (t.own_a) c->event_pool.push_back(t.a);
    c->event_pool.push_back(t.b);
}

void harvest_timers(afq_ctx* c, std::vector<TimedLaunch>* list = nullptr) {
    if (list) { c->launches.insert(c->launches.end(), list->begin(), list->end()); list->clear(); }
    for (auto& t : c->launches) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { c->k_ms[t.id] += ms; c->k_launches[t.id] += 1; }
        recycle_events(c, t);
    }
    c->launches.clear();
}

uint32_t hdr_bytes(const afq_config& cfg) { return 4 + cfg.bc_bytes + cfg.umi_bytes; }

// Multi-bucket cells are placed into fixed-capacity bucket slabs (no counting pass) unless AFQ_TEST_FIXED_SLABS=0;
// AFQ_TEST_SLAB_CAP shrinks the slabs (tests: forces the overflow path).
bool fixed_slabs() { return !test_hook_is("FIXED_SLABS", "0"); }
// Default 384 slots for buckets planned at <= 256 keys (kBucketTarget): measured on the bench input, 512 costs the scatter
// 10 % (a sparser target), 320 already sends a tenth of the cells through the exact placement (profiles/history/run_r02s.sh).
constexpr uint32_t kSlabCap = 384;
uint32_t slab_capacity() { const long v = test_hook_long("SLAB_CAP", 0); return v > 0 ? (uint32_t)v : kSlabCap; }

uint32_t bucket_target() { return kBucketTarget; }   // planned mean keys per bucket

bool valid_width(uint32_t w) { return w == 1 || w == 2 || w == 4 || w == 8; }

// What the device path implements today.  Anything else is refused loudly.
int check_supported(afq_ctx* c) {
    const afq_config& g = c->cfg;
    if (g.sa_model > AFQ_SA_PREFER_AMBIG) return fail(c, AFQ_ERR_INVALID_ARG, "unknown sa_model");
    if (g.num_bootstraps && !(g.resolution == AFQ_RES_CR_LIKE_EM || g.resolution == AFQ_RES_PARSIMONY_EM || g.resolution == AFQ_RES_PARSIMONY_GENE_EM))
        return fail(c, AFQ_ERR_INVALID_ARG, "bootstrapping can only be used with the cr-like-em, parsimony-em, or parsimony-gene-em resolution strategies");   // main.rs:713-724
    if (g.dump_eq && !(g.resolution == AFQ_RES_CR_LIKE_EM || g.resolution == AFQ_RES_PARSIMONY_EM || g.resolution == AFQ_RES_PARSIMONY_GENE_EM))
        return fail(c, AFQ_ERR_UNSUPPORTED, "dump_eq: the gene-level classes are kept only by the -em resolutions (they resolve to the same "
                                            "classes as their plain siblings; afq_quantify runs the sibling for -d)");
    if (g.umi_len > 4 * g.umi_bytes) return fail(c, AFQ_ERR_INVALID_ARG, "umi_len does not fit the UMI field");
    return 0;
}

// Layout of RangeState::d_p2_small (u32 words), the per-cell / per-partition / per-tile arrays of the phase-kernel parsimony
// path: a region that starts zeroed, the arrays the kernels fill, and a region uploaded from the host in one copy.
struct P2Small {
    uint64_t pcnt, pnp, pncls, pn3, gcnt, fb, ctr, gdesc, pfd, route, zero_words;      // zeroed: per partition reads / pairs / staged classes, per-cell counters / flags, work counter, the range-wide graph build's block and per-cell routing flags
    uint64_t poff, pcur, pnv, pcell, tq, bq, old_list, nta, nba, pcpre, pbq, npa, ptile;  // filled on the device (nta / nba: entries per tile quantity / per block quantity)
    uint64_t up, fb_count, fb_list, order, cells, tiles, up_words, words;   // uploaded
};
P2Small p2_small_layout(uint64_t n, uint64_t parts, uint64_t tiles, uint64_t n_pug) {
    P2Small L{};
    uint64_t o = 0;
    L.pcnt = o; o += parts; L.pnp = o; o += parts; L.pncls = o; o += parts; L.pn3 = o; o += parts; L.gcnt = o; o += 4 * n; L.fb = o; o += n; L.ctr = o; o += 12; L.gdesc = o; o += kGDescWords * n;
    o = (o + 1) & ~1ull;   // (8-byte fields from here)
    L.pfd = o; o += (sizeof(PfDev) + 3) / 4; L.route = o; o += n;
    L.zero_words = o;
    L.poff = o; o += parts; L.pcur = o; o += parts; L.pnv = o; o += parts; L.pcell = o; o += parts;
    L.nba = (tiles + 1 + 1023) / 1024; L.nta = L.nba * 1024;   // (six quantities per tile, scanned in blocks of 1024 tiles; entry `tiles` is the end)
    L.tq = o; o += 6 * L.nta; L.bq = o; o += 6 * L.nba; L.old_list = o; o += n;
    L.npa = (parts + 1 + 1023) / 1024 * 1024; L.pcpre = o; o += L.npa; L.pbq = o; o += L.npa / 1024;
    o = (o + 15) & ~15ull; L.ptile = o; o += tiles * (sizeof(PfTile) / 4);
    o = (o + 3) & ~3ull;
    L.up = o; L.fb_count = o; o += 4; L.fb_list = o; o += n_pug; L.order = o; o += n; o = (o + 3) & ~3ull;
    L.cells = o; o += n * (sizeof(P2Cell) / 4); L.tiles = o; o += 2 * tiles;
    L.up_words = o - L.up; L.words = o;
    return L;
}
uint64_t p2_small_bytes(uint64_t n, uint64_t parts, uint64_t tiles, uint64_t n_pug) { return 4 * p2_small_layout(n, parts, tiles, n_pug).words + 64; }

// Split the batch into ranges of cells that fit the memory budget, build nothing yet.
int plan_ranges(afq_ctx* c) {
    const uint32_t H = hdr_bytes(c->cfg);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(c, hipMemGetInfo(&free_b, &total_b));
    // buffers already held by this context are reusable
    size_t held = 0;
    for (auto& rs : c->rs) for (DevBuf* b : rs.all()) held += b->cap;
    const double mem_budget = 0.40 * (double)(free_b + held);  // two range buffer sets are alive at a time
    const uint32_t res = c->cfg.resolution;
    const bool em_res = res == AFQ_RES_CR_LIKE_EM || res == AFQ_RES_PARSIMONY_EM || res == AFQ_RES_PARSIMONY_GENE_EM;
    const bool pug_res = res >= AFQ_RES_PARSIMONY_EM && res <= AFQ_RES_PARSIMONY_GENE;
    // pass 1: validate the chunk headers, device bytes each cell needs
    std::vector<double> need(c->n_cells);
    double total_need = 0, pug_fixed = 0, wide_new = 0;
    const bool use_slabs = fixed_slabs();            // (environment switches: read once per batch, not per cell)
    const double slab_slots = (double)std::max<uint32_t>(slab_capacity(), 512u);   // (ranges of many-gene reads take 512-slot slabs: run_range)
    c->all_aligned = true;
    for (uint32_t i = 0; i < c->n_cells; ++i) {
        const uint64_t off = c->chunk_off[i];
        if (off & 3) c->all_aligned = false;
        const uint32_t nbytes = c->hdr[2 * i], nrec = c->hdr[2 * i + 1];
        if (off + 8 > c->n_bytes || nbytes < 8 || off + nbytes > c->n_bytes)
            return fail(c, AFQ_ERR_BAD_INPUT, "cell " + std::to_string(i) + ": chunk header/size out of range");
        if (nrec == 0) return fail(c, AFQ_ERR_BAD_INPUT, "cell " + std::to_string(i) + ": chunk with no reads");
        const uint64_t fixed = 8ull + (uint64_t)nrec * H;
        if (fixed > nbytes || ((nbytes - fixed) & 3))
            return fail(c, AFQ_ERR_BAD_INPUT, "cell " + std::to_string(i) + ": chunk nbytes does not match its records");
        const uint64_t n_ref = (nbytes - fixed) / 4;
        // (EM resolutions: the canonical kernels' scratch is 40 B per ref and output space; the fixed-point EM sets aside ~25-30 B per ref
        //  behind the range's kernels and, when that plan falls short, the host sizes it exactly - up to three times as much: the larger of
        //  the two is what a range sized to fill the device must have room for)
        double nd = (em_res ? 24.0 + std::max(40.0 * (c->cfg.usa_mode ? 3 : 1), 90.0) : 16.0) * (double)n_ref + 128.0;
        if (pug_res) {  // per read: decode outputs + edge pool; the PUG scratch is per workgroup (sized for the largest cell)
            nd += 20.0 * nrec + 96.0 * nrec;   // rd_h/rd_u/rd_o + the edge pool (24 words per read), as run_range allocates them
            pug_fixed = std::max(pug_fixed, 4.0 * (double)pug_scratch_words(nrec, (uint32_t)n_ref, true) * pug_max_blocks() + 4.0 * (double)(1ull << 22));
        }
        if (n_ref > bucket_target()) nd += 16.0 * (double)(n_ref / bucket_target() + 1) + (use_slabs && !pug_res ? 8.0 * 2.0 * slab_slots / bucket_target() * (double)n_ref : 0.0);   // (+ the slabs of keys1: up to 2 x slab capacity slots per kBucketTarget refs)
        if (nd > mem_budget) return fail(c, AFQ_ERR_OOM, "cell " + std::to_string(i) + " alone exceeds device memory");
        need[i] = nd;
        total_need += nd;
    }
    // (also: parsimony over 4/8-byte fields whose chunks the caller placed at offsets that are not dword aligned - same copy, nothing widened)
    c->widen = c->cfg.bc_split != 0 || !decode_par_supported(c->cfg.bc_bytes, c->cfg.umi_bytes) || (pug_res && !c->all_aligned);
    c->eff_bc = c->cfg.bc_split ? 8 : (c->cfg.bc_bytes < 4 ? 4 : c->cfg.bc_bytes);
    c->eff_umi = c->cfg.umi_bytes < 4 ? 4 : c->cfg.umi_bytes;
    if (c->widen) {
        const uint32_t delta = c->eff_bc + c->eff_umi - c->cfg.bc_bytes - c->cfg.umi_bytes;
        c->w_off.resize(c->n_cells);
        c->w_nbytes.resize(c->n_cells);
        uint64_t o = 0;
        for (uint32_t i = 0; i < c->n_cells; ++i) {
            const uint64_t nb = (uint64_t)c->hdr[2 * i] + (uint64_t)c->hdr[2 * i + 1] * delta;
            if (nb >= (1ull << 32)) return fail(c, AFQ_ERR_UNSUPPORTED, "cell " + std::to_string(i) + ": chunk over 4 GiB once its fields are widened");
            c->w_off[i] = o; c->w_nbytes[i] = (uint32_t)nb;
            o += nb;   // (a multiple of 4: 8 + nrec * (4 + 4|8 + 4|8) + 4 * refs)
        }
        c->wide_bytes = o;
        const size_t had = c->d_wide.cap;
        HIP_TRY(c, c->d_wide.ensure(o + 16));
        wide_new = (double)(c->d_wide.cap - had);
    }
    // pass 2: cut into ranges.  Big batches are cut into a handful of ranges even when memory
    // would allow one, so that the D2H of one range's rows hides under the kernels of the next.  The ranges taper:
    // the last one's compaction + D2H is the only part nothing hides, so it is the smallest.
    // (parsimony: every range ends in the tail of its persistent workgroups and nothing of the next range can share a CU with
    // them, while its rows are few - three ranges; cr-like: five, tapering, so that the last D2H is small)
    // (late round 4, cr-like: a range's rows take about half as long to cross PCIe as its kernels run - more for ranges of small
    //  cells, whose rows are longer per read - so every range is 0.6 of the one before it: six ranges, the last 3.3 % of the work;
    //  28/28/22/14/8 % left 0.65 ms of the last range's rows in the open.  12.96-13.26 -> 12.48-12.68 ms, profiles/history/run_r04ad.sh)
    static const double kTaperCr[] = {0.419, 0.671, 0.822, 0.913, 0.967, 1.0, 1.0, 1.0}, kTaperPug[] = {0.40, 0.76, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
    const double* kTaper = pug_res ? kTaperPug : kTaperCr;
    const size_t kTaperN = 8;
    if (pug_fixed > 0.5 * mem_budget) return fail(c, AFQ_ERR_OOM, "the largest parsimony cell's scratch does not fit device memory");
    double budget = mem_budget - pug_fixed - 0.40 * wide_new;   // (the widened copy was allocated after the free-memory query)
    if (budget <= 0) return fail(c, AFQ_ERR_OOM, "the widened copy of the batch leaves no room for the ranges");
    const bool pipe = c->n_bytes >= (256u << 20);
    if (const char* e = test_hook("RANGE_BYTES")) budget = std::min(budget, std::atof(e));  // tests: force many ranges
    c->ranges.clear();
    double used = 0, done = 0;
    uint32_t c0 = 0;
    size_t step = 0;
    for (uint32_t i = 0; i < c->n_cells; ++i) {
        const bool taper_cut = pipe && step + 1 < kTaperN && kTaper[step] < 1.0 && done + used >= kTaper[step] * total_need;
        if ((used + need[i] > budget || taper_cut) && i > c0) {
            c->ranges.push_back({c0, i}); c0 = i; done += used; used = 0;
            while (step + 1 < kTaperN && done >= kTaper[step] * total_need) ++step;
        }
        used += need[i];
    }
    if (c->n_cells > c0) c->ranges.push_back({c0, c->n_cells});
    return 0;
}

// Which walk-free decoder suits the batch: lane-per-record when records are short (few alignment words each),
// lane-per-dword otherwise.  AFQ_TEST_DECODE=recs|keys overrides (tests run both).
static uint32_t decode_short_records(uint64_t n_ref_words, uint64_t n_records) {
    if (const char* e = test_hook("DECODE")) {
        if (!strcmp(e, "recs")) return 1;
        if (!strcmp(e, "keys")) return 0;
    }
    return n_ref_words < 2 * n_records ? 1u : 0u;
}

// Label hashes of a range: salt 0 first; after a collision (two different labels under one 62-bit key) the range is decoded
// again under another salt.  AFQ_TEST_LABEL_HASH_BITS=n keeps only n bits of the first try's hashes (tests: collisions for certain).
static uint64_t label_salt(uint32_t hash_try) {
    uint64_t x = 0x9E3779B97F4A7C15ull * hash_try;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return hash_try ? (x ^ (x >> 31)) : 0ull;
}
static uint64_t label_mask(uint32_t hash_try) {
    if (hash_try == 0) { const long n = test_hook_long("LABEL_HASH_BITS", 0); if (n > 0 && n < 64) return (1ull << n) - 1; }
    return ~0ull;
}
constexpr uint32_t kMaxHashTries = 4;
constexpr uint32_t kMaxPoolTries = 3;   // 32 words per read x 4^3

// Reads that carry many alignments carry many genes: most UMIs then outgrow the three gene counters of a slot of k_resolve's
// UMI table and their buckets end up sorted after the table has been tried.  Such ranges (two or more alignment words per
// record on average - the decoders switch on the same figure) sort every bucket at once.
static uint32_t resolve_sort_only(uint64_t n_ref_words, uint64_t n_records) {
    return n_ref_words >= 2 * n_records ? 1u : 0u;
}

// The clears and small uploads of one range, collected and issued as ONE pinned-arena H2D copy + ONE kernel (k_range_init).
struct RangeInit {
    struct Pending { void* dst; const void* src; size_t bytes; };
    std::vector<Pending> ops;
    void zero(void* dst, size_t bytes) { if (bytes) ops.push_back({dst, nullptr, bytes}); }
    void upload(void* dst, const void* src, size_t bytes) { if (bytes) ops.push_back({dst, src, bytes}); }
    int flush(afq_ctx* c, RangeState& B, hipStream_t s);
};
int RangeInit::flush(afq_ctx* c, RangeState& B, hipStream_t s) {
    size_t total = 0;
    for (const Pending& o : ops) if (o.src) total += (o.bytes + 15) & ~size_t(15);
    if (total) {
        B.h_arena.n = 0;
        HIP_TRY(c, B.h_arena.reserve(total));
        HIP_TRY(c, B.d_arena.ensure(total));
    }
    size_t at = 0;
    RangeInitOps k{};
    auto launch = [&]() { if (k.n) launch_range_init(s, k, B.d_arena.as<uint8_t>()); k.n = 0; };
    std::vector<RangeInitOp> all;
    for (const Pending& o : ops) {
        if (o.src) {
            std::memcpy(B.h_arena.p + at, o.src, o.bytes);
            all.push_back({o.dst, (uint64_t)at, (uint64_t)o.bytes});
            at += (o.bytes + 15) & ~size_t(15);
        } else all.push_back({o.dst, ~0ull, (uint64_t)o.bytes});
    }
    if (total) HIP_TRY(c, hipMemcpyAsync(B.d_arena.p, B.h_arena.p, total, hipMemcpyHostToDevice, s));   // (pinned source: the stream sync behind the uploads covers its reuse)
    for (const RangeInitOp& o : all) {
        k.op[k.n++] = o;
        if (k.n == kRangeInitOps) launch();
    }
    launch();
    HIP_TRY(c, hipGetLastError());
    ops.clear();
    return 0;
}

// Plan + enqueue one range of cells on the context's stream.
int run_range(afq_ctx* c, Range r, int slot, hipEvent_t h2d_done = nullptr, uint32_t hash_try = 0, uint32_t pool_try = 0) {
    HostClock hc;
    RangeState& B = c->rs[slot];
    B.hash_try = hash_try;
    B.pool_try = pool_try;
    afq_config g = c->cfg;
    if (c->widen) { g.bc_bytes = c->eff_bc; g.umi_bytes = c->eff_umi; }   // what the kernels see: the widened copy
    const uint32_t H = hdr_bytes(g);
    const uint32_t n = r.c1 - r.c0;
    B.meta.resize(n);
    std::vector<uint32_t> multi, slab_prefix, pug_cells, hist_cells;
    std::vector<uint64_t> rd_off(n, 0);
    uint64_t n_pug_reads = 0, pug_words = 0;  // pug_words: scratch of the largest parsimony cell
    const bool par = (c->widen || c->all_aligned) && decode_par_supported(g.bc_bytes, g.umi_bytes);
    uint64_t key_off = 0, n_buckets = 0, n_tiles = 0, n_slabs = 0, k1_slots = 0;
    uint32_t max_lg_nb = 0;
    const bool slabs = fixed_slabs();
    uint32_t slab_cap = slab_capacity();
    {   // reads of many genes each (the range averages two or more alignment words per record): all keys of a UMI share a bucket, so
        // the buckets' sizes spread and 384-slot slabs overflow in a tenth of the cells (k_fix_slabs: 3.5 of 41 ms on the tail
        // model); such ranges get 512-slot slabs (AFQ_TEST_SLAB_CAP still overrides)
        uint64_t words = 0, recs = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t ci = r.c0 + i;
            const uint64_t nb = c->widen ? c->w_nbytes[ci] : c->hdr[2 * ci], nr = c->hdr[2 * ci + 1];
            words += (nb - 8ull - nr * H) / 4; recs += nr;
        }
        if (!test_hook("SLAB_CAP") && resolve_sort_only(words, recs)) slab_cap = std::max<uint32_t>(slab_cap, 512u);
    }
    if (par) slab_prefix.reserve(n + 1);
    // bucket -> cell and scatter tile -> (cell, tile) are written on the device from the cells' plans (k_fill_tables)
    uint64_t nrec_total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t ci = r.c0 + i;
        CellMeta& m = B.meta[i];
        m.chunk_off = c->widen ? c->w_off[ci] : c->chunk_off[ci];
        m.nbytes = c->widen ? c->w_nbytes[ci] : c->hdr[2 * ci];
        m.nrec = c->hdr[2 * ci + 1];
        m.n_ref = (uint32_t)((m.nbytes - 8ull - (uint64_t)m.nrec * H) / 4);
        m.key_off = key_off;
        key_off += (uint64_t)m.n_ref + 1;
        // strategy dispatch of src/quant.rs:794-938: tiny cells take the cr-like fast path whatever -r says
        const bool tiny = g.sa_model == AFQ_SA_WINNER_TAKE_ALL && m.nrec < g.small_thresh;
        m.mode = tiny ? kModeCrLike
                 : g.resolution == AFQ_RES_TRIVIAL ? kModeTrivial
                 : g.resolution == AFQ_RES_CR_LIKE_EM ? kModeCrLikeEm
                 : g.resolution == AFQ_RES_PARSIMONY ? kModePug
                 : g.resolution == AFQ_RES_PARSIMONY_EM ? kModePugEm
                 : g.resolution == AFQ_RES_PARSIMONY_GENE ? kModePugGene
                 : g.resolution == AFQ_RES_PARSIMONY_GENE_EM ? kModePugGeneEm : kModeCrLike;
        // parsimony cells emit reads, not keys: they take no part in the bucket pipeline (one empty bucket, no tiles)
        uint32_t lg = 0;
        if (!mode_is_pug(m.mode)) while (((uint64_t)bucket_target() << lg) < m.n_ref && lg < kMaxLgNb) ++lg;
        m.lg_nb = lg;
        max_lg_nb = std::max(max_lg_nb, lg);
        m.bucket_base = (uint32_t)n_buckets;
        n_buckets += 1ull << lg;
        m.slab_cap = 0; m.k1_off = 0; m.tile_base = 0;
        if (lg && slabs) {
            m.slab_cap = slab_cap;
            m.k1_off = k1_slots;
            k1_slots += std::max<uint64_t>((uint64_t)slab_cap << lg, (uint64_t)m.n_ref + 1);
        }
        if (lg) {
            multi.push_back(i);
            const uint32_t nt = (m.n_ref + kScatterTileHost - 1) / kScatterTileHost;
            m.tile_base = (uint32_t)n_tiles;   // (n_tiles is checked against 32 bits below)
            n_tiles += nt;
        }
        if (mode_is_pug(m.mode)) {
            pug_cells.push_back(i); rd_off[i] = n_pug_reads; n_pug_reads += m.nrec;
            pug_words = std::max<uint64_t>(pug_words, pug_scratch_words(m.nrec, m.n_ref, mode_pug_gene(m.mode)));
        }
        nrec_total += m.nrec;
        if (par) {
            slab_prefix.push_back((uint32_t)n_slabs);
            n_slabs += ((uint64_t)(m.nbytes >> 2) + kSlabWords - 1) / kSlabWords;
        }
    }
    if (n_slabs >= 0xFFFFFFF0ull) return fail(c, AFQ_ERR_UNSUPPORTED, "batch too large for 32-bit slab ids");
    if (par) {
        slab_prefix.push_back((uint32_t)n_slabs);
    }
    if (n_buckets >= 0xFFFFFFF0ull || n_tiles >= 0xFFFFFFF0ull || key_off >= (1ull << 40))
        return fail(c, AFQ_ERR_UNSUPPORTED, "batch too large for 32-bit bucket/tile ids");
    const uint32_t n_multi = (uint32_t)multi.size();

    HIP_TRY(c, B.d_meta.ensure(sizeof(CellMeta) * n));
    HIP_TRY(c, B.d_keys0.ensure(8 * key_off));
    HIP_TRY(c, B.d_keys1.ensure((n_multi || !pug_cells.empty()) ? 8 * std::max(key_off, k1_slots) : 8));  // bucket slabs, then pair staging of multi-bucket and parsimony cells
    HIP_TRY(c, B.d_slab_ovf.ensure(4ull * n));
    HIP_TRY(c, B.d_cell_nkeys.ensure(4ull * n));
    HIP_TRY(c, B.d_bucket_cnt.ensure(4 * n_buckets));
    HIP_TRY(c, B.d_bucket_cell.ensure(4 * n_buckets));
    HIP_TRY(c, B.d_multi_cells.ensure(4ull * std::max<uint32_t>(n_multi, 1)));
    HIP_TRY(c, B.d_tile_desc.ensure(8ull * std::max<uint64_t>(n_tiles, 1)));
    HIP_TRY(c, B.d_ncols.ensure(4ull * n));
    HIP_TRY(c, B.d_nnz.ensure(4ull * n));
    HIP_TRY(c, B.d_ovf.ensure(sizeof(OverflowEnt) * std::max<uint64_t>(n_buckets, 1)));
    HIP_TRY(c, B.d_bdesc.ensure(bucket_desc_bytes() * std::max<uint64_t>(n_buckets, 1)));
    const bool em = g.resolution == AFQ_RES_CR_LIKE_EM || g.resolution == AFQ_RES_PARSIMONY_EM || g.resolution == AFQ_RES_PARSIMONY_GENE_EM;
    const uint32_t n_pug = (uint32_t)pug_cells.size();
    const uint32_t n_pug_blocks = std::min<uint32_t>(n_pug, pug_max_blocks());
    // largest cells first: the persistent workgroups take them in list order, so the long ones do not end up as the tail
    std::stable_sort(pug_cells.begin(), pug_cells.end(), [&](uint32_t a, uint32_t b) { return B.meta[a].nrec > B.meta[b].nrec; });
    if (n_pug && !par) return fail(c, AFQ_ERR_UNSUPPORTED, "device parsimony needs dword-aligned chunk offsets");
    // Parsimony cells go through the phase kernels of afq_pug2.hip (partition-parallel) unless their labels are
    // gene-level, their UMI field is wider than 4 bytes or they hold 2^22 reads or more: those - and the cells the phase kernels
    // hand back - are resolved by the one-workgroup kernel of afq_pug.hip.  AFQ_TEST_PUG_ROUTE=mono sends every cell there (tests).
    std::vector<P2Cell> p2cells;
    std::vector<uint2> p2tiles;
    std::vector<uint32_t> p2_up;
    std::vector<uint32_t> mono_cells;
    uint64_t p2_parts = 0;
    const uint32_t p2_tile = kP2TileHost;
    {
        const char* route = test_hook("PUG_ROUTE");
        const bool p2_ok = n_pug && g.umi_bytes == 4 && !(g.resolution == AFQ_RES_PARSIMONY_GENE || g.resolution == AFQ_RES_PARSIMONY_GENE_EM) &&
                           !(route && !std::strcmp(route, "mono"));
        for (uint32_t ci : pug_cells) {   // (largest first)
            const CellMeta& m = B.meta[ci];
            if (!p2_ok || m.nrec >= (1u << 22) || m.n_ref < m.nrec) { mono_cells.push_back(ci); continue; }   // (n_ref < nrec: records without alignments - the column list is sized by n_ref)
            P2Cell pc{};
            pc.rd_base = rd_off[ci]; pc.chunk_off = m.chunk_off; pc.cell = ci; pc.R = m.nrec; pc.n_ref = m.n_ref; pc.key_off = m.key_off;
            uint32_t lg = 0;
            while (((m.nrec + (1u << lg) - 1) >> lg) > kP2PartTarget) ++lg;
            pc.lgP = lg; pc.part_base = (uint32_t)p2_parts;
            p2_parts += 1ull << lg;
            const uint32_t j = (uint32_t)p2cells.size();
            pc.tile0 = (uint32_t)p2tiles.size();
            for (uint32_t t = 0; t * p2_tile < m.nrec; ++t) p2tiles.push_back(make_uint2(j, t));
            p2cells.push_back(pc);
        }
        if (p2_parts >= 0xFFFFFFF0ull) return fail(c, AFQ_ERR_UNSUPPORTED, "batch too large for 32-bit partition ids");
    }
    const uint32_t n_p2 = (uint32_t)p2cells.size();
    hist_cells = multi;
    hist_cells.insert(hist_cells.end(), pug_cells.begin(), pug_cells.end());
    // (the phase kernels' per-read arrays take ten of the words per read, the rest is the pool; a range whose densest cell outgrows
    //  it - short UMIs, hundreds of reads per UMI: the pairs of a vertex are no longer a handful - is run again with four times as much)
    // (24 words per read since round 6, 32 before: the range-wide graph build takes 3.6 words per read out of the pool on the bench sample
    //  and 8.8 under the label-tail model, next to the 10.25 of the per-read arrays - profiles/round6_01_10_flat_graph.txt, call 26 -
    //  and a first parsimony range of a PBMC-10k sample is 5 GB of hipMalloc less)
    const uint64_t pool_words_per_read = [] { const long v = test_hook_long("POOL_WORDS", 0); return v >= 12 && v <= 32 ? (uint64_t)v : 24ull; }();   // (tests: a small first pool; read per range - a test sets it for itself)
    const uint64_t epool_words = ((pool_words_per_read * n_pug_reads + (pool_words_per_read >= 24 ? (1ull << 22) : (1ull << 20))) << (2 * pool_try));
    if (n_pug) {
        HIP_TRY(c, B.d_pug_cells.ensure(4ull * n_pug));
        HIP_TRY(c, B.d_rd_off.ensure(8ull * n));
        HIP_TRY(c, B.d_rd_h.ensure(8 * n_pug_reads + 8));
        HIP_TRY(c, B.d_rd_u.ensure(8 * n_pug_reads + 8));
        HIP_TRY(c, B.d_rd_o.ensure(4 * n_pug_reads + 8));
        HIP_TRY(c, B.d_pug_scr_off.ensure(8));  // work counter
        HIP_TRY(c, B.d_pug_scratch.ensure(4 * pug_words * n_pug_blocks + 64));
        HIP_TRY(c, B.d_epool.ensure(4 * epool_words));
        HIP_TRY(c, B.d_epool_cur.ensure(8));
        HIP_TRY(c, B.d_p2_small.ensure(p2_small_bytes(n_p2, p2_parts, p2tiles.size(), n_pug)));
    }
    HIP_TRY(c, B.d_alt.ensure(4ull * n));
    HIP_TRY(c, B.d_hist_cells.ensure(4ull * std::max<size_t>(hist_cells.size(), 1)));
    if (em) {
        HIP_TRY(c, B.d_lab.ensure(8 * key_off));
        HIP_TRY(c, B.d_lab_cnt.ensure(8ull * n));
    }
    HIP_TRY(c, B.d_status.ensure(sizeof(DevStatus)));
    HIP_TRY(c, B.d_bc.ensure(8ull * n));
    if (par) {
        HIP_TRY(c, B.d_chk.ensure(sizeof(CellChk) * n));
        HIP_TRY(c, B.d_fix.ensure(4ull * n));
        HIP_TRY(c, B.d_slab_prefix.ensure(4ull * (n + 1)));
        HIP_TRY(c, B.d_slab_cell.ensure(4ull * std::max<uint64_t>(n_slabs, 1)));
        HIP_TRY(c, B.d_cell_bc.ensure(8ull * n));
    }

    hc.lap("run: plan + ensure buffers");
    hipStream_t s = B.stream;
    RangeInit init;   // every clear and every small upload of the range: one H2D copy + one kernel (flushed below)
    if (par) {
        init.upload(B.d_slab_prefix.p, slab_prefix.data(), 4ull * (n + 1));
        init.zero(B.d_chk.p, sizeof(CellChk) * n);
        init.zero(B.d_cell_nkeys.p, 4ull * n);
    }
    init.upload(B.d_meta.p, B.meta.data(), sizeof(CellMeta) * n);
    if (n_multi) {
        init.upload(B.d_multi_cells.p, multi.data(), 4ull * n_multi);
    }
    init.zero(B.d_bucket_cnt.p, 4 * n_buckets);
    init.zero(B.d_slab_ovf.p, 4ull * n);
    init.zero(B.d_nnz.p, 4ull * n);
    init.zero(B.d_ncols.p, 4ull * n);
    if (em) init.zero(B.d_lab_cnt.p, 8ull * n);
    init.zero(B.d_alt.p, 4ull * n);
    if (!hist_cells.empty())
        init.upload(B.d_hist_cells.p, hist_cells.data(), 4ull * hist_cells.size());
    if (n_pug) {
        init.upload(B.d_pug_cells.p, pug_cells.data(), 4ull * n_pug);
        init.upload(B.d_rd_off.p, rd_off.data(), 8ull * n);
        init.zero(B.d_pug_scr_off.p, 8);
        init.zero(B.d_epool_cur.p, 8);
        const P2Small L = p2_small_layout(n_p2, p2_parts, p2tiles.size(), n_pug);
        init.zero(B.d_p2_small.p, 4 * L.zero_words);
        p2_up.assign(L.up_words, 0);
        uint32_t* up = p2_up.data() - L.up;
        up[L.fb_count] = (uint32_t)mono_cells.size();   // the one-workgroup kernel's list starts with the cells that go there directly
        std::copy(mono_cells.begin(), mono_cells.end(), up + L.fb_list);
        for (uint32_t j = 0; j < n_p2; ++j) up[L.order + j] = j;   // (p2cells is largest first already)
        if (n_p2) std::memcpy(up + L.cells, p2cells.data(), sizeof(P2Cell) * n_p2);
        if (!p2tiles.empty()) std::memcpy(up + L.tiles, p2tiles.data(), sizeof(uint2) * p2tiles.size());
        init.upload(B.d_p2_small.as<uint32_t>() + L.up, p2_up.data(), 4 * L.up_words);
    }
    init.zero(B.d_status.p, sizeof(DevStatus));
    init.zero(B.d_bc.p, 8ull * n);
    // EM resolutions: the EM follows the range's kernels on the device (afq_em2.hip; k_em2_plan packs the cells' scratch slices from
    // the counts the kernels leave) - no trip to the host between resolution and EM.  Its scratch is set aside from an upper
    // bound of the cells' label areas (a fraction of it: real cells use a tenth); if that ever falls short the kernels return at
    // once and finish_range sizes the EM itself.  -d / -b read the classes off the canonical set-up and take that route too.
    uint64_t em2_cap = 0;
    std::vector<uint32_t> em_order;
    const uint32_t na_em = g.usa_mode ? g.num_rows : g.num_genes;
    {
        B.em_inline = em && !em_order_canonical() && em2_supported(na_em) && !(g.dump_eq || g.num_bootstraps);
    }
    if (B.em_inline) {
        double worst = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t cap1 = B.meta[i].n_ref + 1;
            worst += (double)em2_scratch_words(std::min(cap1, na_em), cap1, cap1 / 2, g.usa_mode != 0);
        }
        const double frac = [] { const char* e = test_hook("EM2_SCRATCH_FRAC"); const double v = e ? std::atof(e) : 0.0; return v > 0 && v <= 1 ? v : 0.35; }();   // (tests: a sliver, so that the fallback runs; read per range like the other hooks)
        em2_cap = (uint64_t)std::max(worst * frac, 4096.0);
        em_order.resize(n);
        for (uint32_t i = 0; i < n; ++i) em_order[i] = i;
        std::stable_sort(em_order.begin(), em_order.end(), [&](uint32_t a, uint32_t b) { return B.meta[a].nrec > B.meta[b].nrec; });
        HIP_TRY(c, B.d_em_order.ensure(4ull * n));
        init.upload(B.d_em_order.p, em_order.data(), 4ull * n);
        HIP_TRY(c, B.d_em2_tiers.ensure(4ull * (8 + 5ull * n)));
        init.zero(B.d_em2_tiers.p, 32);
        HIP_TRY(c, B.d_em_nnz.ensure(4ull * n));
        HIP_TRY(c, B.d_em2_off.ensure(8ull * (n + 1)));
        HIP_TRY(c, B.d_em2_scratch.ensure(4 * em2_cap + 16));
        HIP_TRY(c, B.d_em2_tiers.ensure(4ull * (8 + 5ull * n)));
    }
    if (const int rc = init.flush(c, B, s)) return rc;
    launch_fill_tables(s, B.d_meta.as<CellMeta>(), n, B.d_bucket_cell.as<uint32_t>(), B.d_tile_desc.as<uint2>());
    // (the uploads come out of the slot's pinned arena, which the next range of this slot fills only after finish_range has
    //  waited for this one; until the arena existed they came out of the vectors above, hence the wait here)
    HIP_TRY(c, hipStreamSynchronize(s));
    {   // this range's kernels start after the previous range's kernels (clean per-kernel timings, no cache
        // thrash between ranges); what overlaps them is the previous range's D2H and this range's enqueue
        RangeState& O = c->rs[slot ^ 1];
        if (O.in_flight && O.kernels_done) HIP_TRY(c, hipStreamWaitEvent(s, O.kernels_done, 0));
    }
    if (h2d_done) HIP_TRY(c, hipStreamWaitEvent(s, h2d_done, 0));   // afq_submit: this range's input bytes have landed
    hc.lap("run: uploads + memsets");

    const uint8_t* const in_bytes = c->widen ? c->d_wide.as<uint8_t>() : c->d_bytes;
    const size_t in_n = c->widen ? (size_t)c->wide_bytes : c->n_bytes;
    TimerChain tc(c, s, &B.launches, par);   // the brackets of the range's kernels (HIP events; cfg.profile)
    if (c->widen) {
        HIP_TRY(c, B.d_src_off.ensure(8ull * n));
        HIP_TRY(c, hipMemcpyAsync(B.d_src_off.p, c->chunk_off.data() + r.c0, 8ull * n, hipMemcpyHostToDevice, s));   // (c->chunk_off outlives the batch)
        tc.seg(K_DECODE);
        launch_widen(s, c->d_bytes, c->n_bytes, B.d_src_off.as<uint64_t>(), B.d_meta.as<CellMeta>(), n, c->cfg.bc_bytes, c->cfg.umi_bytes,
                     c->eff_bc, c->eff_umi, c->d_wide.as<uint8_t>(), B.d_status.as<DevStatus>(), c->cfg.bc_split);
    }
    DecodeArgs da{in_bytes, in_n, B.d_meta.as<CellMeta>(), n, c->d_t2g.as<uint32_t>(), c->ref_count,
                  g.num_genes, B.d_keys0.as<uint64_t>(), B.d_cell_nkeys.as<uint32_t>(),
                  B.d_bc.as<uint64_t>(), B.d_status.as<DevStatus>(),
                  par ? B.d_chk.as<CellChk>() : nullptr, B.d_slab_prefix.as<uint32_t>(), B.d_slab_cell.as<uint32_t>(),
                  B.d_cell_bc.as<uint64_t>(),
                  (uint32_t)n_slabs,
                  n_pug ? PugOut{B.d_rd_h.as<uint64_t>(), B.d_rd_u.as<uint64_t>(), B.d_rd_o.as<uint32_t>(), B.d_rd_off.as<uint64_t>(), label_salt(hash_try), label_mask(hash_try)}
                        : PugOut{nullptr, nullptr, nullptr, nullptr, 0, ~0ull},
                  g.resolution == AFQ_RES_TRIVIAL ? 1u : 0u, decode_short_records(key_off - n, nrec_total), par ? B.d_fix.as<uint32_t>() : nullptr};
    if (par) {
        tc.seg(K_DECODE_PAR);
        if (launch_decode_par(s, da, g.bc_bytes, g.umi_bytes)) { tc.end(); return fail(c, AFQ_ERR_INVALID_ARG, "bad field widths"); }
    }
    {   // sequential walk: the whole decode for unaligned layouts, the verified fix-up otherwise
        tc.seg(K_DECODE);
        if (launch_decode(s, da, g.bc_bytes, g.umi_bytes)) { tc.end(); return fail(c, AFQ_ERR_INVALID_ARG, "bad field widths"); }
    }
    ResolveArgs ra{B.d_meta.as<CellMeta>(), B.d_bucket_cell.as<uint32_t>(), B.d_multi_cells.as<uint32_t>(),
                   B.d_tile_desc.as<uint2>(), B.d_cell_nkeys.as<uint32_t>(), B.d_bucket_cnt.as<uint32_t>(),
                   B.d_keys0.as<uint64_t>(), B.d_keys1.as<uint64_t>(), B.d_ncols.as<uint32_t>(),
                   B.d_nnz.as<uint32_t>(), B.d_ovf.as<OverflowEnt>(), B.d_bdesc.p, em ? B.d_lab.as<uint32_t>() : nullptr,
                   em ? B.d_lab_cnt.as<uint32_t>() : nullptr, B.d_status.as<DevStatus>(),
                   (uint32_t)n_buckets, n_multi, (uint32_t)n_tiles, B.d_hist_cells.as<uint32_t>(),
                   (uint32_t)hist_cells.size(), g.usa_mode, g.num_rows,
                   (g.usa_mode && g.sa_model == AFQ_SA_PREFER_AMBIG) ? 1u : 0u, max_lg_nb, B.d_slab_ovf.as<uint32_t>(), slabs ? 1u : 0u,
                   resolve_sort_only(key_off - n, nrec_total)};
    if (n_multi) {
        if (!slabs) {
            tc.seg(K_HIST); launch_hist(s, ra);
            tc.seg(K_BSCAN); launch_bucket_scan(s, ra);
        }
        tc.seg(K_SCATTER); launch_scatter(s, ra);
        if (slabs) { tc.seg(K_FIX_SLABS); launch_fix_slabs(s, ra); }
    }
    tc.seg(K_RESOLVE); launch_resolve(s, ra);
    if (n_multi) { tc.seg(K_RESOLVE_BIG); launch_resolve_big(s, ra); }
    if (n_pug) {
        const P2Small L = p2_small_layout(n_p2, p2_parts, p2tiles.size(), n_pug);
        uint32_t* sm = B.d_p2_small.as<uint32_t>();
        // the big per-read arrays of the phase kernels come out of the edge pool's allocation (24 words per read): key and
        // UMI words, pair lists, offsets, local ids, flags; what is left is the pool both paths draw their per-cell scratch from
        uint32_t* ep = B.d_epool.as<uint32_t>();
        uint64_t eo = 0;
        P2Args p2{};
        if (n_p2) {
            p2.s_h = reinterpret_cast<uint64_t*>(ep + eo); eo += 2 * n_pug_reads;
            p2.s_u = reinterpret_cast<uint64_t*>(ep + eo); eo += 2 * n_pug_reads;
            p2.pairs = reinterpret_cast<uint64_t*>(ep + eo); eo += 2 * n_pug_reads;
            p2.cstage = reinterpret_cast<uint64_t*>(ep + eo); eo += em ? 2 * n_pug_reads : 0;
            p2.v_off = ep + eo; eo += n_pug_reads;
            p2.lidx = ep + eo; eo += n_pug_reads;
            p2.v_flag = reinterpret_cast<uint8_t*>(ep + eo); eo += n_pug_reads / 4 + 1;
            eo = (eo + 3) & ~3ull;
        }
        if (eo + (1ull << 20) > epool_words) { tc.end(); return fail(c, AFQ_ERR_OOM, "parsimony work arrays do not fit the edge pool"); }
        uint32_t* const pool = ep + eo;
        const unsigned long long pool_cap = epool_words - eo;
        if (n_p2) {
            p2.bytes = in_bytes; p2.meta = ra.meta; p2.cells = reinterpret_cast<const P2Cell*>(sm + L.cells);
            p2.tiles = reinterpret_cast<const uint2*>(sm + L.tiles); p2.order = sm + L.order;
            p2.rd_h = da.pug.h; p2.rd_u = da.pug.u;
            p2.pcnt = sm + L.pcnt; p2.poff = sm + L.poff; p2.pcur = sm + L.pcur; p2.pnv = sm + L.pnv; p2.pcell = sm + L.pcell;
            p2.pnp = sm + L.pnp; p2.pncls = sm + L.pncls; p2.pn3 = sm + L.pn3; p2.gcnt = sm + L.gcnt; p2.fb = sm + L.fb; p2.fb_list = sm + L.fb_list; p2.fb_count = sm + L.fb_count;
            p2.pool = pool; p2.pool_cur = B.d_epool_cur.as<unsigned long long>(); p2.pool_cap = pool_cap;
            p2.work_counter = sm + L.ctr; p2.work_counter2 = sm + L.ctr + 1; p2.gdesc = sm + L.gdesc;
            p2.tq = sm + L.tq; p2.bq = sm + L.bq; p2.nta = (uint32_t)L.nta; p2.nba = (uint32_t)L.nba; p2.old_list = sm + L.old_list;
            p2.pfd = reinterpret_cast<PfDev*>(sm + L.pfd); p2.route = sm + L.route;
            p2.pcpre = sm + L.pcpre; p2.pbq = sm + L.pbq; p2.npa = (uint32_t)L.npa; p2.ptile = reinterpret_cast<PfTile*>(sm + L.ptile);
            // The graph phase: range-wide flat kernels (afq_pugflat.hip) unless the range's reads outgrow their 32-bit slot numbers
            // (>= 2^31 parsimony reads in ONE range: the per-cell kernel then takes every cell) or a test asks for the per-cell kernel.
            p2.graph_flat = (n_pug_reads < (1ull << 31) && !test_hook_is("P2_GRAPH", "cell")) ? 1u : 0u;
            p2.cell_nkeys = ra.cell_nkeys; p2.t2g = c->d_t2g.as<uint32_t>(); p2.keys0 = ra.keys0; p2.cell_ncols = ra.cell_ncols;
            p2.lab = ra.lab; p2.lab_cnt = ra.lab_cnt; p2.st = ra.st; p2.alt = B.d_alt.as<uint32_t>();
            p2.n_cells = n_p2; p2.n_tiles = (uint32_t)p2tiles.size(); p2.n_parts = (uint32_t)p2_parts;
            {
                const uint32_t big_reads = [] { const char* e = test_hook("P2_BIG_READS"); const long v = e ? std::atol(e) : 0; return v > 0 ? (uint32_t)v : 15000u; }();   // (tests: read per range; configs[2] graph + cover + tie kernels per step, round 5: 60 000: 24.6 ms, 40 000: 23.9, 25 000: 22.2-23.1, 15 000: 20.3-20.5)
                p2.max_comp = [] { const long v = test_hook_long("P2_MAX_COMP", 0); return v >= 64 && v <= (long)kP2MaxComp ? (uint32_t)v : kP2MaxComp; }();   // (tests: 64 = larger components are handed back, as before round 4)
                p2.tile = p2_tile;
                p2.n_big = 0;
                while (p2.n_big < n_p2 && p2cells[p2.n_big].R >= big_reads) ++p2.n_big;   // (p2cells is largest first)
                p2.defer_min = [] { const char* e = test_hook("P2_DEFER_MIN"); return e ? (uint32_t)std::max(0L, std::atol(e)) : 0xFFFFFFFFu; }();   // (tests: 0 = every cell takes the set-aside route)
            }
            // k_p2_lone, labels over four refs: 0: by the vertex's lane alone, in scratch memory (rounds 3-4); 1: labels of 5..64 refs by the
            // wave; 2: 5..8 by the lane in eight registers, 9..64 by the wave - an instance of 86 instead of 69 VGPRs, five waves per SIMD
            // instead of seven: on the tail model k_p2_lone 25.1 -> 16.5 ms per step, on the plain one 5.6 -> 7.3 (profiles/history/run_r04ao.sh).
            // The range's own figure decides, the one that picks its decoder: two or more alignment words per record.
            p2.lone_coop = [&] { const char* e = test_hook("P2_LONE_COOP"); return e && e[0] >= '0' && e[0] <= '2' ? (uint32_t)(e[0] - '0') : (key_off - n >= 2 * nrec_total ? 2u : 1u); }();
            p2.part_cap = kP2PartCap;
            if (const char* e = test_hook("P2_PART_CAP")) p2.part_cap = (uint32_t)std::max(1, std::atoi(e));   // tests: force cells back to the one-workgroup kernel
            p2.ref_count = c->ref_count; p2.num_genes = g.num_genes; p2.usa = g.usa_mode; p2.num_rows = g.num_rows; p2.em = em ? 1u : 0u;
            p2.exact_umi = g.pug_exact_umi; p2.large_thresh = g.large_graph_thresh; p2.hw = 1 + g.bc_bytes / 4 + g.umi_bytes / 4;
            p2.umi_pairs = std::min<uint32_t>(g.umi_len ? g.umi_len : g.umi_bytes * 4, 16);
            tc.seg(K_P2_SPLIT); launch_p2_split(s, p2);
            tc.seg(K_P2_PART); launch_p2_part(s, p2);
            tc.seg(K_P2_SEARCH); launch_p2_search(s, p2);
            tc.seg(K_P2_LONE); launch_p2_lone(s, p2);
            tc.seg(K_P2_GRAPH); launch_p2_graph(s, p2, n_pug_reads);
            B.pf_stats_src = p2.graph_flat && test_hook("PF_STATS") ? p2.pfd : nullptr; B.pf_stats_reads = n_pug_reads; B.pf_stats_parts = p2_parts;
        }
        PugCellArgs pa{};
        pa.bytes = in_bytes; pa.meta = ra.meta; pa.pug_cells = sm + L.fb_list; pa.cell_nkeys = ra.cell_nkeys;
        pa.n_pug_dev = sm + L.fb_count;
        pa.rd = da.pug; pa.scr_stride = pug_words; pa.scratch = B.d_pug_scratch.as<uint32_t>(); pa.work_counter = B.d_pug_scr_off.as<uint32_t>(); pa.n_pug = n_pug;
        pa.epool = pool; pa.epool_cursor = B.d_epool_cur.as<unsigned long long>(); pa.epool_cap = pool_cap;
        pa.t2g = c->d_t2g.as<uint32_t>(); pa.keys0 = ra.keys0; pa.cell_ncols = ra.cell_ncols; pa.lab = ra.lab; pa.lab_cnt = ra.lab_cnt;
        pa.alt = B.d_alt.as<uint32_t>(); pa.st = ra.st; pa.ref_count = c->ref_count; pa.num_genes = g.num_genes; pa.usa = g.usa_mode;
        pa.num_rows = g.num_rows; pa.em = em ? 1u : 0u; pa.exact_umi = g.pug_exact_umi; pa.large_thresh = g.large_graph_thresh; pa.umi32 = g.umi_bytes == 4 ? 1u : 0u;
        pa.hw = 1 + g.bc_bytes / 4 + g.umi_bytes / 4; pa.umi_pairs = std::min<uint32_t>(g.umi_len ? g.umi_len : g.umi_bytes * 4, 22);
        pa.gene_level = (g.resolution == AFQ_RES_PARSIMONY_GENE || g.resolution == AFQ_RES_PARSIMONY_GENE_EM) ? 1u : 0u;
        pa.force_global_route = test_hook("PUG_GLOBAL_ROUTE") ? 1u : 0u;
        // The one-workgroup kernel is 0.36 ms of a range even when its list is empty and whatever its grid (its private segment -
        // 138 spilled registers per lane - is set up per dispatch: 1.1 ms of a configs[2] step that hands no cell back).  It is
        // therefore launched only when the host sent it cells itself or this context has seen the phase kernels hand one back;
        // the first range that does (finish_range reads the count) is run again, with the kernel, once in a context's life.
        B.pug_cell_launched = !mono_cells.empty() || c->handback_seen;
        if (B.pug_cell_launched) { tc.seg(K_PUG); launch_pug(s, pa, n_pug_blocks); }
    } else B.pug_cell_launched = true;
    // What the next range's kernels wait for: all of this range's.  (Letting them start beside the per-cell histograms a range
    // without an EM ends in was measured on the headline in round 4 and not kept: the decoder fills every SIMD at eight
    // waves, the histogram workgroups - 73 KiB of LDS each - get a CU only as decoder workgroups drain, the bracket of
    // k_cell_hist grows from 0.56 to 3.3 ms per step and the range's rows start across PCIe that much later: 12.97 -> 15.38 ms
    // per step, profiles/history/run_r04aa.sh.  The same with the histograms on a stream of the device's highest priority: 3.4 ms,
    // 13.1 -> 15.0 ms per step, profiles/history/run_r04ad.sh - queue priority does not put a 73 KiB workgroup in front of the
    // decoder's 9 KiB ones.)
    if (!B.kernels_done) HIP_TRY(c, hipEventCreateWithFlags(&B.kernels_done, hipEventDisableTiming));
    if (!hist_cells.empty()) { tc.seg(K_CELL_HIST); launch_cell_hist(s, ra); }
    if (B.em_inline) {
        tc.seg(K_EM);
        launch_em2(s, ra, n, B.d_em2_off.as<uint64_t>(), B.d_em2_scratch.as<uint32_t>(), B.d_em_nnz.as<uint32_t>(), B.d_em_order.as<uint32_t>(),
                   B.d_em2_tiers.as<uint32_t>(), na_em, g.em_init_uniform, em2_cap);
    }
    tc.end();
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(B.kernels_done, s));
    PackSmallArgs pack{};
    {   // what finish_range reads first: written by the last kernel of the range STRAIGHT into pinned host memory (20 bytes per cell over
        // PCIe).  An async D2H copy here instead would sit in the copy queue until the range's kernels are done - with the NEXT range's
        // upload queued behind it: the next range then started 150 us after this one ended instead of right behind it (seen in the
        // timeline: 0.45 ms per step).
        const size_t words = kPackHdrWords + 5ull * n;
        HIP_TRY(c, B.h_pack.reserve(words));
        void* d_view = nullptr;
        HIP_TRY(c, hipHostGetDevicePointer(&d_view, B.h_pack.p, 0));
        pack = PackSmallArgs{B.d_status.as<DevStatus>(), B.em_inline ? B.d_em2_tiers.as<uint32_t>() + 7 : nullptr, B.d_alt.as<uint32_t>(),
                             B.em_inline ? B.d_em_nnz.as<uint32_t>() : nullptr, B.d_bc.as<uint64_t>(),
                             n_pug ? B.d_p2_small.as<uint32_t>() + p2_small_layout(n_p2, p2_parts, p2tiles.size(), n_pug).fb_count : nullptr, reinterpret_cast<uint32_t*>(d_view)};
        // (resolutions without an EM: k_row_ptr below packs it in its own launch - one 5 us kernel and one boundary less per range)
        if (em) launch_pack_small(s, pack.st, pack.em_flag, pack.alt, B.d_nnz.as<uint32_t>(), pack.em_nnz, pack.bc, pack.n_mono, n, pack.out);
    }
    // The rows' compaction follows at once, with row offsets made on the device - it used to wait for the host to read the row
    // lengths, sum them and send the offsets back: 0.27 ms between the last kernel of a batch's last range and its compaction,
    // with nothing else for the device to do (profiles/r04_timeline_configs1.txt).  The buffers are the slot's as they stand (they
    // grow to the largest range seen); a range with more entries is compacted by finish_range as before.  EM resolutions take their
    // rows out of the EM's scratch (finish_range).
    B.chained = false;
    if (!em) {
        HIP_TRY(c, B.d_cell_ptr.ensure(8ull * (n + 1)));
        B.chain_cap = std::min(B.d_gene.cap, B.d_val.cap) / 4;
        tc.seg(K_COMPACT);
        launch_row_ptr(s, B.d_nnz.as<uint32_t>(), n, B.d_cell_ptr.as<uint64_t>(), pack);
        launch_compact(s, B.d_meta.as<CellMeta>(), n, B.d_keys0.as<uint64_t>(), B.d_keys1.as<uint64_t>(), B.d_nnz.as<uint32_t>(),
                       B.d_cell_ptr.as<uint64_t>(), B.d_gene.as<uint32_t>(), B.d_val.as<float>(), B.chain_cap);
        tc.end();
        HIP_TRY(c, hipGetLastError());
        B.chained = true;
    }
    hc.lap("run: enqueue kernels");
    B.last_ra = ra;
    B.cur = r;
    B.in_flight = true;
    // (a range that is run again - another label hash, a larger pool - counts once: finish_range takes the failed attempt back)
    B.att_records = nrec_total; B.att_ref_words = key_off; B.att_buckets = n_buckets;
    c->stats.n_records += nrec_total;
    c->stats.n_ref_words += key_off;
    c->stats.n_buckets += n_buckets;
    return 0;
}

// Wait for the range in flight, compact its rows and append them to the host result.
int finish_range(afq_ctx* c, int slot) {
    RangeState& B = c->rs[slot];
    if (!B.in_flight) return 0;
    B.in_flight = false;
    HostClock hc;
    const uint32_t n = B.cur.c1 - B.cur.c0;
    hipStream_t s = B.stream;
    HIP_TRY(c, hipStreamSynchronize(s));
    hc.lap("finish: wait for kernels");
    DevStatus st{};
    std::memcpy(&st, B.h_pack.p, sizeof(st));   // (k_pack_small wrote it there, behind the range's kernels)
    const uint32_t* const pk = B.h_pack.p + kPackHdrWords;
    auto take_back_attempt = [&]() {   // the failed attempt's share of the statistics and its kernel timings
        c->stats.n_records -= B.att_records; c->stats.n_ref_words -= B.att_ref_words; c->stats.n_buckets -= B.att_buckets;
        for (TimedLaunch& t : B.launches) recycle_events(c, t);   // (back to the pool, not into the kernel times)
        B.launches.clear();
    };
    // A label-hash collision or a graph that outgrew the pool names its cell.  The range is cut around that cell: the cells before
    // and behind it run again as they were, the cell itself ALONE under the next hash function / with four times the pool - a pool
    // for one cell, not for the range (a range sized to fill the device cannot have its pool quadrupled), and the other cells keep
    // their hashes.  Every cut costs the range a re-run, so a range that keeps failing (every cell of it dense, or - in the tests -
    // every label colliding) goes back to the round-3 answer after three cuts: the whole range under the next setting.
    const bool rehash = st.err_code == kErrLabelHash && B.hash_try + 1 < kMaxHashTries;
    const bool regrow = st.err_code == kErrPugPool && B.pool_try < kMaxPoolTries;
    if (rehash || regrow) {
        (rehash ? c->n_label_rehash : c->n_pool_regrow) += 1;
        take_back_attempt();
        const Range whole = B.cur;
        const uint32_t ht = B.hash_try, pt = B.pool_try, bad = whole.c0 + std::min(st.err_cell, whole.c1 - whole.c0 - 1);
        int rc = 0;
        // A label-hash collision names its cell.  A pool that ran out names the cell that asked LAST, not the one whose graph outgrew
        // it (the pool is one bump allocator for the range): the whole range runs again with four times the pool, and only when the
        // device has no room for that is the range cut - around the named cell for want of a better one; a hog that fails again in its
        // part is cut again (three cuts at most, then the round-3 answer: the error).
        bool cut = rehash;
        if (regrow) {
            // (a range sized to fill the device has no room for four times its pool: ask before trying - a failed hipMalloc frees the
            //  pool, costs a full set-up and leaves hipErrorOutOfMemory as the runtime's last error, which the next range would report)
            size_t free_b = 0, total_b = 0;
            const bool room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)free_b + (double)B.d_epool.cap > 4.2 * (double)B.d_epool.cap;
            if (room) {
                rc = run_range(c, whole, slot, nullptr, ht, pt + 1);
                if (!rc) rc = finish_range(c, slot);
            } else rc = AFQ_ERR_OOM;
            if (rc == AFQ_ERR_OOM) { cut = true; rc = 0; (void)hipGetLastError(); { std::lock_guard<std::mutex> g(c->err_mu); c->err.clear(); } }
        }
        if (cut && whole.c1 - whole.c0 > 1 && c->retry_cuts < 3) {
            c->retry_cuts += 1;
            const Range parts[3] = {{whole.c0, bad}, {bad, bad + 1}, {bad + 1, whole.c1}};
            for (int k = 0; k < 3 && !rc; ++k) {
                if (parts[k].c1 == parts[k].c0) continue;
                const bool the_cell = k == 1;
                rc = run_range(c, parts[k], slot, nullptr, ht + (the_cell && rehash ? 1 : 0), pt + (the_cell && regrow ? 1 : 0));
                if (!rc) rc = finish_range(c, slot);
            }
            c->retry_cuts -= 1;
        } else if (cut) {
            rc = run_range(c, whole, slot, nullptr, ht + (rehash ? 1 : 0), pt + (regrow ? 1 : 0));
            if (!rc) rc = finish_range(c, slot);
        }
        if (regrow) B.d_epool.release();   // the enlarged pool is that attempt's alone: the next range plans its own
        return rc;
    }
    if (B.pf_stats_src) {   // AFQ_TEST_PF_STATS: the sizes of the range's flat graph build, on stderr (measurement scripts)
        PfDev d{};
        unsigned long long pool_used = 0;
        if (hipMemcpy(&d, B.pf_stats_src, sizeof(d), hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(&pool_used, B.d_epool_cur.p, 8, hipMemcpyDeviceToHost) == hipSuccess)
            std::fprintf(stderr, "[afq] flat graph build: reads %llu partitions %llu vertices_with_an_edge %u two_vertex_components %u listed_components %u record_slots %u cells_routed_to_the_per_cell_kernel %u pool_words_used %llu (%.2f per read; the per-read arrays take 10.25 more)\n",
                         (unsigned long long)B.pf_stats_reads, (unsigned long long)B.pf_stats_parts, d.T, d.NP, d.NC, d.S, d.n_old, pool_used, (double)pool_used / (double)std::max<uint64_t>(1, B.pf_stats_reads));
        B.pf_stats_src = nullptr;
    }
    if (!st.err_code && !B.pug_cell_launched && B.h_pack.p[9]) {   // cells were handed back and the kernel that takes them was not launched
        c->handback_seen = true;
        take_back_attempt();
        int rc = run_range(c, B.cur, slot, nullptr, B.hash_try, B.pool_try);
        if (!rc) rc = finish_range(c, slot);
        return rc;
    }
    if (st.err_code) {
        const std::string cell = "cell " + std::to_string(B.cur.c0 + st.err_cell) + ": ";
        switch (st.err_code) {
            case kErrRecordWalk: return fail(c, AFQ_ERR_BAD_INPUT, cell + "chunk nbytes does not match its records");
            case kErrRefRange: return fail(c, AFQ_ERR_BAD_INPUT, cell + "ref id out of range");
            case kErrGeneRange: return fail(c, AFQ_ERR_BAD_INPUT, cell + "gene id out of range of num_genes");
            case kErrUmiWide: return fail(c, AFQ_ERR_UNSUPPORTED, cell + "UMI wider than 22 nt is not supported");
            case kErrSlotRange: return fail(c, AFQ_ERR_BAD_INPUT, cell + "resolved column >= num_rows");
            case kErrLabelHash: return fail(c, AFQ_ERR_UNSUPPORTED, cell + "two ref lists share a 62-bit label hash under four different hash functions");
            case kErrPugLimit: return fail(c, AFQ_ERR_UNSUPPORTED, cell + "a device-side PUG limit was exceeded: 2^22 reads in the cell; 2^20 reads when the cell needs the one-workgroup kernel (gene-level labels, a UMI field over 4 bytes, or a cell of 2^20..2^22 reads the partition kernels handed back: a UMI partition over 256 reads or a component over 64 vertices / over --large-graph-thresh); a component of more than 4096 vertices under a raised --large-graph-thresh; or a vertex with an empty label");
            case kErrPugPool: return fail(c, AFQ_ERR_OOM, cell + "PUG edge pool exhausted");
            case kErrInternal: return fail(c, AFQ_ERR_HIP, cell + "internal consistency check failed in the parsimony kernels");
            default: return fail(c, AFQ_ERR_HIP, cell + "device error code " + std::to_string(st.err_code) + (st.err_code >= 20 ? " (internal consistency check of the parsimony kernels)" : ""));
        }
    }
    c->stats.n_keys += st.n_keys;
    c->stats.n_overflow_buckets += st.n_overflow;
    c->stats.n_fallback_cells += st.n_fallback;
    c->n_mono_cells += B.h_pack.p[9];
    std::vector<uint32_t> nnz(n);
    std::vector<uint64_t> bc(n), ptr(n + 1);
    const bool em = c->cfg.resolution == AFQ_RES_CR_LIKE_EM || c->cfg.resolution == AFQ_RES_PARSIMONY_EM ||
                    c->cfg.resolution == AFQ_RES_PARSIMONY_GENE_EM;
    std::vector<uint32_t> alt(n);
    bool em2 = false;   // the EM ran in afq_em2.hip: the rows sit in its scratch
    std::memcpy(alt.data(), pk, 4ull * n);
    std::memcpy(nnz.data(), pk + n, 4ull * n);
    if (em && B.em_inline) {   // the EM ran behind the range's kernels: did its scratch suffice?
        const uint32_t short_of_scratch = B.h_pack.p[8];
        if (!short_of_scratch) {
            em2 = true;
            std::memcpy(nnz.data(), pk + 2ull * n, 4ull * n);
        } else c->n_em_resized += 1;
    }
    if (em && !em2) {
        // per-cell EM (src/em.rs) over the single-label counts + the ambiguous molecules' labels, sized on the host
        std::vector<uint32_t> lc(2ull * n);
        std::vector<uint64_t> eoff(n + 1);
        HIP_TRY(c, hipMemcpy(lc.data(), B.d_lab_cnt.p, 8ull * n, hipMemcpyDeviceToHost));
        // The EM runs in order-free fixed-point arithmetic (afq_em2.hip) unless AFQ_EM_ORDER=canonical asks for the sequential f32
        // sums in canonical class order (afq_em.hip, rounds 1-3) or the output space does not fit the set-up kernel's bitmap.
        // -d / -b read the cell's classes off the canonical set-up, which then runs as well (set-up only).
        const uint32_t na_em = c->cfg.usa_mode ? c->cfg.num_rows : c->cfg.num_genes;
        em2 = !em_order_canonical() && em2_supported(na_em);
        const bool need_classes = c->cfg.dump_eq || c->cfg.num_bootstraps;
        std::vector<uint32_t> em_order(n);
        for (uint32_t i = 0; i < n; ++i) em_order[i] = i;
        std::stable_sort(em_order.begin(), em_order.end(), [&](uint32_t a, uint32_t b) { return B.meta[a].nrec > B.meta[b].nrec; });
        HIP_TRY(c, B.d_em_order.ensure(4ull * n));
        HIP_TRY(c, hipMemcpyAsync(B.d_em_order.p, em_order.data(), 4ull * n, hipMemcpyHostToDevice, s));
        HIP_TRY(c, B.d_em_nnz.ensure(4ull * n));
        if (!em2 || need_classes) {
            eoff[0] = 0;
            for (uint32_t i = 0; i < n; ++i) eoff[i + 1] = eoff[i] + em_scratch_words(nnz[i], lc[2 * i], lc[2 * i + 1], c->cfg.usa_mode != 0);
            HIP_TRY(c, B.d_em_off.ensure(8ull * (n + 1)));
            HIP_TRY(c, B.d_em_scratch.ensure(4 * eoff[n] + 16));
            HIP_TRY(c, B.d_em_hdr.ensure(16ull * n));
            HIP_TRY(c, hipMemcpyAsync(B.d_em_off.p, eoff.data(), 8ull * (n + 1), hipMemcpyHostToDevice, s));
            ScopedTimer t(c, K_EM, s, &B.launches);
            launch_em(s, B.last_ra, n, B.d_em_off.as<uint64_t>(), B.d_em_scratch.as<uint32_t>(), B.d_em_nnz.as<uint32_t>(), B.d_em_hdr.p, B.d_em_order.as<uint32_t>(),
                      na_em, c->cfg.em_init_uniform, !em2);
        }
        std::vector<uint64_t> eoff2(n + 1);
        if (em2) {
            eoff2[0] = 0;
            for (uint32_t i = 0; i < n; ++i) eoff2[i + 1] = eoff2[i] + em2_scratch_words(nnz[i], lc[2 * i], lc[2 * i + 1], c->cfg.usa_mode != 0);
            HIP_TRY(c, B.d_em2_off.ensure(8ull * (n + 1)));
            HIP_TRY(c, B.d_em2_scratch.ensure(4 * eoff2[n] + 16));
            HIP_TRY(c, B.d_em2_tiers.ensure(4ull * (8 + 5ull * n)));
            HIP_TRY(c, hipMemcpyAsync(B.d_em2_off.p, eoff2.data(), 8ull * (n + 1), hipMemcpyHostToDevice, s));
            ScopedTimer t(c, K_EM, s, &B.launches);
            launch_em2(s, B.last_ra, n, B.d_em2_off.as<uint64_t>(), B.d_em2_scratch.as<uint32_t>(), B.d_em_nnz.as<uint32_t>(), B.d_em_order.as<uint32_t>(),
                       B.d_em2_tiers.as<uint32_t>(), na_em, c->cfg.em_init_uniform);
        }
        HIP_TRY(c, hipStreamSynchronize(s));
        HIP_TRY(c, hipMemcpy(nnz.data(), B.d_em_nnz.p, 4ull * n, hipMemcpyDeviceToHost));
        if (c->cfg.dump_eq || c->cfg.num_bootstraps) {
            // the cells' gene-level classes, read back off the EM set-up (k_eqc_dump): size, prefix on the host, fill
            HostResult& R = *c->res;
            const uint32_t na = c->cfg.usa_mode ? c->cfg.num_rows : c->cfg.num_genes;
            HIP_TRY(c, B.d_eq_ncls.ensure(4ull * n)); HIP_TRY(c, B.d_eq_nw.ensure(4ull * n));
            launch_eqc_dump(s, B.last_ra, n, B.d_em_off.as<uint64_t>(), B.d_em_scratch.as<uint32_t>(), B.d_em_hdr.p, na,
                            B.d_eq_ncls.as<uint32_t>(), B.d_eq_nw.as<uint32_t>(), nullptr, nullptr, nullptr, nullptr, nullptr);
            std::vector<uint32_t> ncls(n), nw(n);
            HIP_TRY(c, hipMemcpyAsync(ncls.data(), B.d_eq_ncls.p, 4ull * n, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipMemcpyAsync(nw.data(), B.d_eq_nw.p, 4ull * n, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            std::vector<uint64_t> cp(n + 1), wp(n + 1);
            cp[0] = wp[0] = 0;
            for (uint32_t i = 0; i < n; ++i) { cp[i + 1] = cp[i] + ncls[i]; wp[i + 1] = wp[i] + nw[i]; }
            HIP_TRY(c, B.d_eq_cptr.ensure(8ull * (n + 1))); HIP_TRY(c, B.d_eq_wptr.ensure(8ull * (n + 1)));
            HIP_TRY(c, B.d_eq_len.ensure(std::max<uint64_t>(4 * cp[n], 16))); HIP_TRY(c, B.d_eq_cnt.ensure(std::max<uint64_t>(4 * cp[n], 16)));
            HIP_TRY(c, B.d_eq_lab.ensure(std::max<uint64_t>(4 * wp[n], 16)));
            HIP_TRY(c, hipMemcpyAsync(B.d_eq_cptr.p, cp.data(), 8ull * (n + 1), hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipMemcpyAsync(B.d_eq_wptr.p, wp.data(), 8ull * (n + 1), hipMemcpyHostToDevice, s));
            launch_eqc_dump(s, B.last_ra, n, B.d_em_off.as<uint64_t>(), B.d_em_scratch.as<uint32_t>(), B.d_em_hdr.p, na, nullptr, nullptr,
                            B.d_eq_cptr.as<uint64_t>(), B.d_eq_wptr.as<uint64_t>(), B.d_eq_len.as<uint32_t>(), B.d_eq_cnt.as<uint32_t>(),
                            B.d_eq_lab.as<uint32_t>());
            if (c->cfg.dump_eq) {
                std::vector<uint32_t> len(cp[n]);
                const size_t k0 = R.eq_count.size(), w0 = R.eq_labels.size();
                R.eq_count.resize(k0 + cp[n]); R.eq_labels.resize(w0 + wp[n]);
                if (cp[n]) {
                    HIP_TRY(c, hipMemcpyAsync(len.data(), B.d_eq_len.p, 4 * cp[n], hipMemcpyDeviceToHost, s));
                    HIP_TRY(c, hipMemcpyAsync(R.eq_count.data() + k0, B.d_eq_cnt.p, 4 * cp[n], hipMemcpyDeviceToHost, s));
                }
                if (wp[n]) HIP_TRY(c, hipMemcpyAsync(R.eq_labels.data() + w0, B.d_eq_lab.p, 4 * wp[n], hipMemcpyDeviceToHost, s));
                HIP_TRY(c, hipStreamSynchronize(s));
                for (uint32_t i = 0; i < n; ++i) R.eq_cell_ptr.push_back(k0 + cp[i + 1]);
                for (uint64_t k = 0; k < cp[n]; ++k) R.eq_label_ptr.push_back(R.eq_label_ptr.back() + len[k]);
            }
            if (c->cfg.num_bootstraps) {
                // bootstrap replicates over those classes (k_boot), then the per-entry summaries compacted and filtered to non-zeros
                const uint32_t NB = c->cfg.num_bootstraps;
                const bool ss = c->cfg.summary_stat != 0;
                std::vector<uint64_t> so(n + 1);
                so[0] = 0;
                for (uint32_t i = 0; i < n; ++i) so[i + 1] = so[i] + ((boot_scratch_words(ncls[i], nw[i], NB, ss) + 1) & ~1ull);
                HIP_TRY(c, B.d_bt_off.ensure(8ull * (n + 1))); HIP_TRY(c, B.d_bt_scratch.ensure(4 * so[n] + 16));
                HIP_TRY(c, B.d_bt_ns.ensure(4ull * n));
                const uint64_t wtot = std::max<uint64_t>(wp[n], 4);
                HIP_TRY(c, B.d_bt_col.ensure(4 * wtot)); HIP_TRY(c, B.d_bt_mean.ensure(4 * wtot)); HIP_TRY(c, B.d_bt_var.ensure(4 * wtot));
                HIP_TRY(c, hipMemcpyAsync(B.d_bt_off.p, so.data(), 8ull * (n + 1), hipMemcpyHostToDevice, s));
                {
                    ScopedTimer t(c, K_BOOT, s, &B.launches);
                    launch_boot(s, n, B.d_eq_cptr.as<uint64_t>(), B.d_eq_wptr.as<uint64_t>(), B.d_eq_len.as<uint32_t>(), B.d_eq_cnt.as<uint32_t>(),
                                B.d_eq_lab.as<uint32_t>(), B.d_bt_off.as<uint64_t>(), B.d_bt_scratch.as<uint32_t>(), NB, ss ? 1u : 0u, c->cfg.boot_seed,
                                c->first_cell_index + B.cur.c0, B.d_bt_ns.as<uint32_t>(), B.d_bt_col.as<uint32_t>(), B.d_bt_mean.as<float>(),
                                B.d_bt_var.as<float>());
                }
                std::vector<uint32_t> ns(n);
                HIP_TRY(c, hipMemcpyAsync(ns.data(), B.d_bt_ns.p, 4ull * n, hipMemcpyDeviceToHost, s));
                HIP_TRY(c, hipStreamSynchronize(s));
                std::vector<uint64_t> sp(n + 1);
                sp[0] = 0;
                for (uint32_t i = 0; i < n; ++i) sp[i + 1] = sp[i] + ns[i];
                const uint64_t stot = std::max<uint64_t>(sp[n], 4);
                HIP_TRY(c, B.d_bt_sptr.ensure(8ull * (n + 1)));
                HIP_TRY(c, B.d_bt_ccol.ensure(4 * stot)); HIP_TRY(c, B.d_bt_cmean.ensure(4 * stot)); HIP_TRY(c, B.d_bt_cvar.ensure(4 * stot));
                HIP_TRY(c, hipMemcpyAsync(B.d_bt_sptr.p, sp.data(), 8ull * (n + 1), hipMemcpyHostToDevice, s));
                launch_boot_compact(s, n, B.d_eq_wptr.as<uint64_t>(), B.d_bt_sptr.as<uint64_t>(), B.d_bt_col.as<uint32_t>(), B.d_bt_mean.as<float>(),
                                    B.d_bt_var.as<float>(), B.d_bt_ccol.as<uint32_t>(), B.d_bt_cmean.as<float>(), B.d_bt_cvar.as<float>());
                std::vector<uint32_t> col(sp[n]);
                std::vector<float> mean(sp[n]), var(sp[n]);
                if (sp[n]) {
                    HIP_TRY(c, hipMemcpyAsync(col.data(), B.d_bt_ccol.p, 4 * sp[n], hipMemcpyDeviceToHost, s));
                    HIP_TRY(c, hipMemcpyAsync(mean.data(), B.d_bt_cmean.p, 4 * sp[n], hipMemcpyDeviceToHost, s));
                    HIP_TRY(c, hipMemcpyAsync(var.data(), B.d_bt_cvar.p, 4 * sp[n], hipMemcpyDeviceToHost, s));
                }
                HIP_TRY(c, hipStreamSynchronize(s));
                for (uint32_t i = 0; i < n; ++i) {
                    for (uint64_t k = sp[i]; k < sp[i + 1]; ++k) {
                        // record_cell keeps the non-zero entries of each vector (quant.rs:171-180); from replicates a variance
                        // is only looked at where the mean is non-zero (quant.rs:192-206) - k_boot leaves it 0 there
                        if (mean[k] != 0.0f) { R.bm_col.push_back(col[k]); R.bm_val.push_back(mean[k]); }
                        if (var[k] != 0.0f) { R.bv_col.push_back(col[k]); R.bv_val.push_back(var[k]); }
                    }
                    R.bm_ptr.push_back(R.bm_col.size()); R.bv_ptr.push_back(R.bv_col.size());
                }
            }
        }
    }
    std::memcpy(bc.data(), pk + 3ull * n, 8ull * n);
    ptr[0] = 0;
    for (uint32_t i = 0; i < n; ++i) ptr[i + 1] = ptr[i] + nnz[i];
    const uint64_t tot = ptr[n];
    hc.lap("finish: small D2H + prefix");
    const bool compacted = B.chained && !em && tot <= B.chain_cap;   // the compaction behind the range's kernels had room for every row
    if (!compacted) {
        HIP_TRY(c, B.d_cell_ptr.ensure(8ull * (n + 1)));
        HIP_TRY(c, B.d_gene.ensure(std::max<uint64_t>(4 * tot, 16)));
        HIP_TRY(c, B.d_val.ensure(std::max<uint64_t>(4 * tot, 16)));
        HIP_TRY(c, hipMemcpyAsync(B.d_cell_ptr.p, ptr.data(), 8ull * (n + 1), hipMemcpyHostToDevice, s));
        ScopedTimer t(c, K_COMPACT, s, &B.launches);
        if (em)
            launch_compact_em(s, n, (em2 ? B.d_em2_off : B.d_em_off).as<uint64_t>(), (em2 ? B.d_em2_scratch : B.d_em_scratch).as<uint32_t>(), B.d_em_nnz.as<uint32_t>(),
                              B.d_cell_ptr.as<uint64_t>(), B.d_gene.as<uint32_t>(), B.d_val.as<float>());
        else
            launch_compact(s, B.d_meta.as<CellMeta>(), n, B.d_keys0.as<uint64_t>(), B.d_keys1.as<uint64_t>(), B.d_nnz.as<uint32_t>(),
                           B.d_cell_ptr.as<uint64_t>(), B.d_gene.as<uint32_t>(), B.d_val.as<float>());
    }
    // (the row offsets still go up, although the device has made its own: under rocprofv3 the runtime moved the rows' two copies
    //  with __amd_rocclr_copyBuffer kernels - 27 % of the GPU time of a profiled step, next to the following range's decoder -
    //  whenever the command in front of them on the stream was a kernel; behind a small upload they stay on the DMA engines, as
    //  they did while the compaction was enqueued from here.  profiles/history/run_r04ak.sh, run_r04al.sh, run_r04am.sh)
    if (compacted) HIP_TRY(c, hipMemcpyAsync(B.d_cell_ptr.p, ptr.data(), 8ull * (n + 1), hipMemcpyHostToDevice, s));
    HostResult& R = *c->res;
    const size_t g0 = R.gene.n;
    HIP_TRY(c, R.gene.reserve(g0 + tot));
    HIP_TRY(c, R.val.reserve(g0 + tot));
    R.gene.n = R.val.n = g0 + tot;
    if (tot) {
        HIP_TRY(c, hipMemcpyAsync(R.gene.p + g0, B.d_gene.p, 4 * tot, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(R.val.p + g0, B.d_val.p, 4 * tot, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    hc.lap("finish: compact + D2H of CSR");
    const afq_config& g = c->cfg;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t nrec = c->hdr[2 * (B.cur.c0 + i) + 1];
        uint8_t f = 0;
        // used_fast_path, src/quant.rs:794-797 (same counts as the general cr-like route)
        if (g.sa_model == AFQ_SA_WINNER_TAKE_ALL && nrec < g.small_thresh) f |= AFQ_CELL_TINY_PATH;
        if (nnz[i] == 0) f |= AFQ_CELL_EMPTY;
        if (alt[i]) f |= AFQ_CELL_ALT_RES;
        R.cell_ptr.push_back(g0 + ptr[i + 1]);
        R.bc.push_back(bc[i]);
        R.nrec.push_back(nrec);
        R.flags.push_back(f);
    }
    harvest_timers(c, &B.launches);
    hc.lap("finish: host result");
    return 0;
}

// Large pageable (or file-mapped) input -> device: the runtime's own pageable path is one staging thread (~8 GB/s,
// slower still when every page of a mapped file faults on first touch); here several threads fill pinned pieces
// while the previous piece is on the wire.  Memory the caller has pinned (hipHostMalloc / hipHostRegister) goes
// straight to the DMA engine.
// (Piece size and thread count, late round 6, `afquant quant` on the 6.9 GB sample in /dev/shm, best of three per box: 32 threads x 64 MiB
//  0.96 s, 64 x 64 1.10, 128 x 64 1.05, 32 x 256 1.27, 8 x 64 0.92, 16 x 16 0.84, 16 x 32 0.82, 8 x 16 0.87 - the reads of a tmpfs file
//  do not scale past some sixteen threads, 17-19 GB/s whatever the scheme.  Every filler a pipeline of its own - its own two pinned
//  pieces, its own stream, no meeting per piece - was built and measured too: 1.05-1.38 s, worse with more threads.  Not kept.)
static const size_t kStagePiece = (size_t)std::max(1L, test_hook_long("STAGE_PIECE_MB", 32)) << 20;   // (the hook: the staged path on small inputs, and the measurement above)
bool host_ptr_is_pinned(const void* p) {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}
unsigned stage_threads() {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const long forced = test_hook_long("STAGE_THREADS", 0);
    if (forced > 0) return (unsigned)std::min(256L, forced);
    return std::min(16u, std::max(4u, hw / 4));
}
struct ByteSource {   // where afq_submit's input comes from: the caller's buffer, or a reader callback (afq_submit_reader)
    const uint8_t* bytes = nullptr;
    afq_read_fn read = nullptr;
    void* user = nullptr;
};
int staged_h2d(afq_ctx* c, uint8_t* dst, const uint8_t* src, size_t n, hipStream_t s, bool pinned, const ByteSource* rd = nullptr, uint64_t rd_off = 0) {
    const bool use_reader = rd && rd->read;
    if (!use_reader && (pinned || n < 2 * kStagePiece)) { HIP_TRY(c, hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, s)); return 0; }
    for (int i = 0; i < 3; ++i) {
        if (!c->stage[i]) HIP_TRY(c, hipHostMalloc(&c->stage[i], kStagePiece, hipHostMallocDefault));
        if (!c->stage_ev[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming));
    }
    // The filler threads live for the whole copy (round 6; rounds 1-5 spawned and joined sixteen per 64 MiB piece: 100 pieces of a
    // PBMC-10k RAD, half a millisecond of thread start-up in front of every one - `afquant quant` spent 0.4 of its second here at
    // 17 GB/s): the submitting thread hands out a piece number, every filler fills its slice of that piece, the last one says so.
    // (sixteen of them since late round 6, and pieces of 32 MiB: see kStagePiece)
    const unsigned nth = stage_threads();
    const size_t n_pieces = (n + kStagePiece - 1) / kStagePiece;
    // (they SLEEP between pieces - a condition variable, not a spin: a parsimony range's kernels run for tens of milliseconds, the next
    //  range's pieces wait for them, and thirty-two fillers yielding in a loop on the CPUs of the device's NUMA node starved the
    //  submitting thread and the runtime's own: `afquant quant -r parsimony-em` 1.35 -> 2.14 s when they spun)
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    long go = -1;                 // the piece the fillers may fill
    unsigned done = 0;            // slices of that piece filled
    bool quit = false;
    std::atomic<int> bad{0};
    auto filler = [&](unsigned t) {
        for (size_t i = 0; i < n_pieces; ++i) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return go >= (long)i || quit; });
                if (quit) return;
            }
            const size_t off = i * kStagePiece, len = std::min(kStagePiece, n - off), slice = (len + nth - 1) / nth;
            const size_t a = t * slice, e = std::min(len, a + slice);
            uint8_t* dstp = (uint8_t*)c->stage[i % 3];
            if (a < e) {
                if (use_reader) { if (rd->read(rd->user, rd_off + off + a, dstp + a, e - a) != 0) bad.store(1); }
                else std::memcpy(dstp + a, src + off + a, e - a);
            }
            std::lock_guard<std::mutex> lk(mu);
            if (++done == nth) cv_done.notify_one();
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nth; ++t) th.emplace_back(filler, t);
    int rc = 0;
    for (size_t i = 0; i < n_pieces && !rc; ++i) {
        const int bsel = (int)(i % 3);
        const size_t off = i * kStagePiece, len = std::min(kStagePiece, n - off);
        if (hipEventSynchronize(c->stage_ev[bsel]) != hipSuccess) { rc = fail(c, AFQ_ERR_HIP, "hipEventSynchronize (staging) failed"); break; }   // the copy that last used this piece (in this call or an earlier one) is done
        {
            std::unique_lock<std::mutex> lk(mu);
            done = 0; go = (long)i;
            cv_go.notify_all();
            cv_done.wait(lk, [&] { return done == nth; });
        }
        if (bad.load()) { rc = fail(c, AFQ_ERR_BAD_INPUT, "the input reader reported an error"); break; }
        if (hipMemcpyAsync(dst + off, c->stage[bsel], len, hipMemcpyHostToDevice, s) != hipSuccess || hipEventRecord(c->stage_ev[bsel], s) != hipSuccess)
            rc = fail(c, AFQ_ERR_HIP, "hipMemcpyAsync (staging) failed");
    }
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv_go.notify_all();
    for (auto& x : th) x.join();
    return rc;
}

// Reset the per-batch state and cut the batch into ranges.
int begin_batch(afq_ctx* c, uint32_t n_cells, uint64_t first_cell_index) {
    c->n_cells = n_cells;
    c->first_cell_index = first_cell_index;
    if (c->res) pool_put(c->res);
    c->res = pool_get(c->pool);
    c->res->cell_ptr.push_back(0);
    if (c->cfg.dump_eq) { c->res->eq_cell_ptr.push_back(0); c->res->eq_label_ptr.push_back(0); }
    if (c->cfg.num_bootstraps) { c->res->bm_ptr.push_back(0); c->res->bv_ptr.push_back(0); }
    c->stats = afq_batch_stats{};
    c->stats.input_bytes = c->n_bytes;
    for (int i = 0; i < K_COUNT; ++i) { c->k_ms[i] = 0; c->k_launches[i] = 0; }
    c->h2d_piped = false;
    return plan_ranges(c);
}

// Software pipeline over the ranges with two buffer sets: range i is enqueued before range i-1 is
// finished (sync + compaction + D2H of its rows), so that copy overlaps range i's kernels.  The last range
// stays in flight until afq_collect.  With h2d_piped, range i's kernels also wait for its input bytes, which an
// upload thread (or the DMA engine alone, for pinned sources) is still bringing over while earlier ranges run.
int run_batch(afq_ctx* c) {
    c->next_range = 0;
    c->pending = true;
    const size_t k = c->ranges.size();
    int rc = 0;
    for (size_t i = 0; i < k && !rc; ++i) {
        hipEvent_t ev = nullptr;
        if (c->h2d_piped) {
            std::unique_lock<std::mutex> lk(c->up_mu);
            c->up_cv.wait(lk, [&]() { return c->up_enqueued > i || c->up_rc != 0; });
            if (c->up_rc) { rc = c->up_rc; std::lock_guard<std::mutex> g(c->err_mu); c->err = c->up_err; break; }
            ev = c->h2d_ev[i];
        }
        rc = run_range(c, c->ranges[i], (int)(i & 1), ev);
        if (!rc && i > 0) rc = finish_range(c, (int)((i - 1) & 1));
    }
    if (rc) {
        c->pending = false;
        for (auto& rs : c->rs) { if (rs.stream) (void)hipStreamSynchronize(rs.stream); rs.in_flight = false; }
        return rc;
    }
    c->next_range = k;
    return 0;
}

// the byte span [first chunk's offset, end of the last chunk) of a range (chunk offsets ascending)
inline void range_span(const afq_ctx* c, Range r, uint64_t& a, uint64_t& b) {
    a = c->chunk_off[r.c0];
    b = c->chunk_off[r.c1 - 1] + c->hdr[2 * (size_t)(r.c1 - 1)];
}

}  // namespace

extern "C" {

int afq_abi_version(void) { return AFQ_ABI_VERSION; }

int afq_device_warmup(int device) {
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return AFQ_ERR_NO_DEVICE; }
    if (hipFree(nullptr) != hipSuccess) return AFQ_ERR_NO_DEVICE;
    warm_code_object();
    return 0;
}

uint64_t afq_label_rehash_count(const afq_ctx* ctx) { return ctx ? ctx->n_label_rehash : 0; }
uint64_t afq_pool_regrow_count(const afq_ctx* ctx) { return ctx ? ctx->n_pool_regrow : 0; }
uint64_t afq_em_resize_count(const afq_ctx* ctx) { return ctx ? ctx->n_em_resized : 0; }
uint64_t afq_mono_cell_count(const afq_ctx* ctx) { return ctx ? ctx->n_mono_cells : 0; }

int afq_device_pci_bus_id(int device, char* out, size_t out_len) {
    if (!out || out_len < 13) return AFQ_ERR_INVALID_ARG;
    if (hipDeviceGetPCIBusId(out, (int)out_len, device) != hipSuccess) { (void)hipGetLastError(); return AFQ_ERR_NO_DEVICE; }
    return 0;
}

const char* afq_last_error(const afq_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int afq_create(const afq_config* cfg, const uint32_t* tid_to_gid, uint32_t ref_count, int device, afq_ctx** out) {
    if (!cfg || !tid_to_gid || !out) return fail(nullptr, AFQ_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (cfg->abi_version != AFQ_ABI_VERSION) return fail(nullptr, AFQ_ERR_INVALID_ARG, "afq_config.abi_version mismatch");
    {   // one line, once per process, when the host has not asked for copies on the DMA engines (the library cannot: the runtime
        // reads the setting when it comes up, which is before this call; INTEGRATION.md "runtime settings")
        static std::once_flag noted;
        std::call_once(noted, [] {
            const char* e = std::getenv("GPU_FORCE_BLIT_COPY_SIZE");
            if (!(e && e[0] == '0' && e[1] == 0))
                std::fprintf(stderr, "[afquant] GPU_FORCE_BLIT_COPY_SIZE=0 is not set: the runtime may move a range's rows with blit kernels on the compute queue, beside the next range's kernels (set it before the HIP runtime starts)\n");
        });
    }
    if (cfg->resolution > AFQ_RES_PARSIMONY_GENE) return fail(nullptr, AFQ_ERR_INVALID_ARG, "bad resolution");
    const bool split_ok = cfg->bc_split >= 1 && cfg->bc_split <= 4 && cfg->bc_bytes > cfg->bc_split && cfg->bc_bytes - cfg->bc_split <= 4;
    if ((cfg->bc_split ? !split_ok : !valid_width(cfg->bc_bytes)) || !valid_width(cfg->umi_bytes))
        return fail(nullptr, AFQ_ERR_INVALID_ARG, "bc_bytes/umi_bytes must be 1, 2, 4 or 8 (bc_split: two barcode integers of 1..4 bytes each)");
    if (cfg->umi_len > 4 * cfg->umi_bytes) return fail(nullptr, AFQ_ERR_INVALID_ARG, "umi_len does not fit the UMI field");
    if (cfg->num_genes == 0 || cfg->num_rows == 0 || ref_count == 0)
        return fail(nullptr, AFQ_ERR_INVALID_ARG, "num_genes, num_rows and ref_count must be non-zero");
    if (cfg->num_genes > (1u << kGeneBits))
        return fail(nullptr, AFQ_ERR_UNSUPPORTED, "gene-id space above 2^20 is not supported");
    if (cfg->usa_mode && (cfg->num_rows % 3 != 0 || cfg->num_genes != 2 * (cfg->num_rows / 3)))
        return fail(nullptr, AFQ_ERR_INVALID_ARG, "USA mode needs num_rows = 3*G and num_genes = 2*G");
    if (!cfg->usa_mode && cfg->num_rows != cfg->num_genes)
        return fail(nullptr, AFQ_ERR_INVALID_ARG, "num_rows must equal num_genes outside USA mode");
    for (uint32_t i = 0; i < ref_count; ++i)
        if (tid_to_gid[i] >= cfg->num_genes) return fail(nullptr, AFQ_ERR_INVALID_ARG, "tid_to_gid entry >= num_genes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, AFQ_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return fail(nullptr, AFQ_ERR_NO_DEVICE, "device ordinal out of range");
    afq_ctx* c = new afq_ctx();
    c->cfg = *cfg;
    if (!c->cfg.usa_mode) c->cfg.sa_model = AFQ_SA_WINNER_TAKE_ALL;  // src/quant.rs:1456-1469
    c->device = device;
    c->ref_count = ref_count;
    c->pool = new ResultPool();
    auto bail = [&](int code, const std::string& m) { g_create_err = m; afq_destroy(c); return code; };
    if (hipSetDevice(device) != hipSuccess) return bail(AFQ_ERR_NO_DEVICE, "hipSetDevice failed");
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(AFQ_ERR_HIP, "hipStreamCreate failed");
    for (auto& rs : c->rs)
        if (hipStreamCreateWithFlags(&rs.stream, hipStreamNonBlocking) != hipSuccess) return bail(AFQ_ERR_HIP, "hipStreamCreate failed");
    if (c->d_t2g.ensure(4ull * ref_count) != hipSuccess) return bail(AFQ_ERR_OOM, "tid_to_gid allocation failed");
    if (hipMemcpy(c->d_t2g.p, tid_to_gid, 4ull * ref_count, hipMemcpyHostToDevice) != hipSuccess)
        return bail(AFQ_ERR_HIP, "tid_to_gid upload failed");
    *out = c;
    return 0;
}

void afq_destroy(afq_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    harvest_timers(c);
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (auto& rs : c->rs) {
        if (rs.stream) (void)hipStreamSynchronize(rs.stream);
        for (DevBuf* b : rs.all()) b->release();
        if (rs.kernels_done) (void)hipEventDestroy(rs.kernels_done);
        if (rs.stream) (void)hipStreamDestroy(rs.stream);
    }
    DevBuf* bufs[] = {&c->d_t2g, &c->d_bytes_own, &c->d_chunk_off, &c->d_hdr, &c->d_wide};
    for (auto b : bufs) b->release();
    for (auto& b : c->atac) b.release();
    for (auto& p : c->stage) if (p) (void)hipHostFree(p);
    for (auto& ev : c->stage_ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : c->h2d_ev) if (ev) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->res) pool_put(c->res);
    if (c->pool) {
        bool del = false;
        {
            std::lock_guard<std::mutex> g(c->pool->mu);
            c->pool->ctx_alive = false;
            for (auto r : c->pool->free_list) { r->gene.release(); r->val.release(); delete r; }
            c->pool->free_list.clear();
            del = c->pool->outstanding == 0;
        }
        if (del) delete c->pool;
    }
    delete c;
}

static int submit_host(afq_ctx* c, const ByteSource& src, size_t n_bytes, const uint64_t* chunk_off, const uint32_t* chunk_hdr,
                       uint32_t n_cells, uint64_t first_cell_index) {
    const uint8_t* bytes = src.bytes;
    if (c->pending) return fail(c, AFQ_ERR_STATE, "previous batch not collected");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = check_supported(c);
    if (rc) return rc;
    c->chunk_off.assign(chunk_off, chunk_off + n_cells);
    c->hdr.assign(2ull * n_cells, 0);
    for (uint32_t i = 0; i < n_cells; ++i) {
        if (chunk_off[i] + 8 > n_bytes) return fail(c, AFQ_ERR_BAD_INPUT, "cell " + std::to_string(i) + ": chunk offset out of range");
        uint32_t h[2];
        if (chunk_hdr) { h[0] = chunk_hdr[2 * i]; h[1] = chunk_hdr[2 * i + 1]; }
        else std::memcpy(h, bytes + chunk_off[i], 8);
        c->hdr[2 * i] = h[0];
        c->hdr[2 * i + 1] = h[1];
    }
    // A RAD file's prelude has an arbitrary length, so chunk offsets inside the caller's buffer are usually not
    // dword-aligned although every chunk size is a multiple of 4.  The bytes are copied anyway: land them shifted so
    // that the chunks start on dword boundaries on the device (that is what the walk-free decode and the PUG path need).
    uint32_t shift = 0;
    bool ascending = true;
    if (n_cells) {
        const uint32_t r = (uint32_t)(chunk_off[0] & 3);
        bool same = true;
        for (uint32_t i = 1; i < n_cells; ++i) { same = same && (chunk_off[i] & 3) == r; ascending = ascending && chunk_off[i] >= chunk_off[i - 1]; }
        if (same) shift = (4 - r) & 3;
    }
    HIP_TRY(c, c->d_bytes_own.ensure(n_bytes + shift + 16));
    uint8_t* const dst = (uint8_t*)c->d_bytes_own.p + shift;
    if (shift) for (auto& o : c->chunk_off) o += shift;
    c->d_bytes = c->d_bytes_own.as<uint8_t>();
    c->n_bytes = n_bytes + shift;
    rc = begin_batch(c, n_cells, first_cell_index);   // validates the headers, cuts the ranges
    if (rc) return rc;
    if (shift) HIP_TRY(c, hipMemsetAsync(c->d_bytes_own.p, 0, 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(dst + n_bytes, 0, 16, c->stream));
    const bool pinned = !src.read && n_bytes && host_ptr_is_pinned(bytes) && host_ptr_is_pinned(bytes + n_bytes - 1);
    const size_t k = c->ranges.size();
    if (k < 2 || !ascending || test_hook("NO_H2D_PIPELINE")) {
        if (n_bytes) { int rc2 = staged_h2d(c, dst, bytes, n_bytes, c->stream, pinned, &src, 0); if (rc2) return rc2; }
        HIP_TRY(c, hipStreamSynchronize(c->stream));  // caller keeps ownership of `bytes`
        return run_batch(c);
    }
    // Range by range: the kernels of range i start when ITS bytes have landed, the later ranges are still crossing.
    while (c->h2d_ev.size() < k) { hipEvent_t e = nullptr; HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->h2d_ev.push_back(e); }
    c->h2d_piped = true;
    c->up_enqueued = 0; c->up_rc = 0; c->up_err.clear();
    auto upload = [&, bytes, pinned, k, shift]() {
        (void)hipSetDevice(c->device);
        for (size_t i = 0; i < k; ++i) {
            uint64_t a, b;
            range_span(c, c->ranges[i], a, b);   // device-buffer coordinates (shift included)
            int rc2 = staged_h2d(c, (uint8_t*)c->d_bytes_own.p + a, bytes ? bytes + (a - shift) : nullptr, (size_t)(b - a), c->stream, pinned, &src, a - shift);
            if (!rc2 && hipEventRecord(c->h2d_ev[i], c->stream) != hipSuccess) rc2 = AFQ_ERR_HIP;
            std::lock_guard<std::mutex> lk(c->up_mu);
            if (rc2) { std::lock_guard<std::mutex> g(c->err_mu); c->up_rc = rc2; c->up_err = c->err.empty() ? "input upload failed" : c->err; }
            c->up_enqueued = i + 1;
            c->up_cv.notify_all();
            if (rc2) return;
        }
    };
    if (pinned) {   // nothing for the host to do but enqueue: all copies go out now, in range order
        upload();
        rc = run_batch(c);
    } else {
        std::thread th(upload);
        rc = run_batch(c);
        th.join();
    }
    {   // the caller keeps ownership of `bytes`: every copy out of them has finished before this returns
        hipError_t e = hipStreamSynchronize(c->stream);
        if (!rc && e != hipSuccess) rc = fail(c, AFQ_ERR_HIP, std::string("input upload: ") + hipGetErrorString(e));
    }
    c->h2d_piped = false;
    return rc;
}

int afq_submit(afq_ctx* c, const uint8_t* bytes, size_t n_bytes, const uint64_t* chunk_off, uint32_t n_cells,
               uint64_t first_cell_index) {
    if (!c) return AFQ_ERR_INVALID_ARG;
    if ((!bytes && n_bytes) || (!chunk_off && n_cells)) return fail(c, AFQ_ERR_INVALID_ARG, "null argument");
    ByteSource src;
    src.bytes = bytes;
    return submit_host(c, src, n_bytes, chunk_off, nullptr, n_cells, first_cell_index);
}

int afq_submit_reader(afq_ctx* c, afq_read_fn read, void* user, size_t n_bytes, const uint64_t* chunk_off, const uint32_t* chunk_hdr,
                      uint32_t n_cells, uint64_t first_cell_index) {
    if (!c) return AFQ_ERR_INVALID_ARG;
    if (!read || (n_cells && (!chunk_off || !chunk_hdr))) return fail(c, AFQ_ERR_INVALID_ARG, "null argument");
    ByteSource src;
    src.read = read; src.user = user;
    return submit_host(c, src, n_bytes, chunk_off, chunk_hdr, n_cells, first_cell_index);
}

int afq_submit_device(afq_ctx* c, const void* d_bytes, size_t n_bytes, const uint64_t* chunk_off, uint32_t n_cells,
                      uint64_t first_cell_index) {
    if (!c) return AFQ_ERR_INVALID_ARG;
    if ((!d_bytes && n_bytes) || (!chunk_off && n_cells)) return fail(c, AFQ_ERR_INVALID_ARG, "null argument");
    if (((uintptr_t)d_bytes) & 3) return fail(c, AFQ_ERR_INVALID_ARG, "device buffer must be 4-byte aligned");
    if (c->pending) return fail(c, AFQ_ERR_STATE, "previous batch not collected");
    HostClock hc;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = check_supported(c);
    if (rc) return rc;
    c->chunk_off.assign(chunk_off, chunk_off + n_cells);
    c->hdr.assign(2ull * n_cells, 0);
    c->d_bytes = (const uint8_t*)d_bytes;
    c->n_bytes = n_bytes;
    if (n_cells) {
        HIP_TRY(c, c->d_chunk_off.ensure(8ull * n_cells));
        HIP_TRY(c, c->d_hdr.ensure(8ull * n_cells));
        HIP_TRY(c, hipMemcpyAsync(c->d_chunk_off.p, chunk_off, 8ull * n_cells, hipMemcpyHostToDevice, c->stream));
        {
            ScopedTimer t(c, K_GATHER);
            launch_gather_headers(c->stream, c->d_bytes, n_bytes, c->d_chunk_off.as<uint64_t>(), n_cells, c->d_hdr.as<uint32_t>());
        }
        HIP_TRY(c, hipMemcpyAsync(c->hdr.data(), c->d_hdr.p, 8ull * n_cells, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    hc.lap("submit: chunk headers");
    int rc2 = begin_batch(c, n_cells, first_cell_index);
    hc.lap("submit: plan ranges");
    return rc2 ? rc2 : run_batch(c);
}

int afq_collect(afq_ctx* c, afq_result* out) {
    if (!c || !out) return AFQ_ERR_INVALID_ARG;
    if (!c->pending) return fail(c, AFQ_ERR_STATE, "afq_collect without a submitted batch");
    HIP_TRY(c, hipSetDevice(c->device));
    c->pending = false;
    int rc = c->ranges.empty() ? 0 : finish_range(c, (int)((c->ranges.size() - 1) & 1));
    if (rc) return rc;
    HostResult* R = c->res;
    c->res = nullptr;
    std::memset(out, 0, sizeof(*out));
    out->n_cells = c->n_cells;
    out->first_cell_index = c->first_cell_index;
    out->nnz = R->gene.n;
    out->cell_ptr = R->cell_ptr.data();
    out->gene = R->gene.p;
    out->val = R->val.p;
    out->bc = R->bc.data();
    out->nrec = R->nrec.data();
    out->flags = R->flags.data();
    out->opaque = R;
    return 0;
}

int afq_result_eqclasses(const afq_result* res, afq_eqclasses* out) {
    if (!res || !out || !res->opaque) return AFQ_ERR_INVALID_ARG;
    const HostResult* R = reinterpret_cast<const HostResult*>(res->opaque);
    std::memset(out, 0, sizeof(*out));
    if (R->eq_cell_ptr.empty()) return AFQ_ERR_STATE;   // the context was not created with dump_eq
    out->n_cells = R->eq_cell_ptr.size() - 1;
    out->n_classes = R->eq_count.size();
    out->n_words = R->eq_labels.size();
    out->cell_ptr = R->eq_cell_ptr.data();
    out->label_ptr = R->eq_label_ptr.data();
    out->labels = R->eq_labels.data();
    out->count = R->eq_count.data();
    return 0;
}

int afq_result_bootstraps(const afq_result* res, afq_bootstraps* out) {
    if (!res || !out || !res->opaque) return AFQ_ERR_INVALID_ARG;
    const HostResult* R = reinterpret_cast<const HostResult*>(res->opaque);
    std::memset(out, 0, sizeof(*out));
    if (R->bm_ptr.empty()) return AFQ_ERR_STATE;   // the context was created with num_bootstraps = 0
    out->n_cells = R->bm_ptr.size() - 1;
    out->mean_ptr = R->bm_ptr.data(); out->mean_col = R->bm_col.data(); out->mean_val = R->bm_val.data();
    out->var_ptr = R->bv_ptr.data(); out->var_col = R->bv_col.data(); out->var_val = R->bv_val.data();
    return 0;
}

void afq_result_release(afq_result* res) {
    if (res && res->opaque) {
        pool_put(reinterpret_cast<HostResult*>(res->opaque));
        std::memset(res, 0, sizeof(*res));
    }
}

int afq_infer(afq_ctx* c, const uint32_t* eq_labels, const uint64_t* eq_label_ptr, uint32_t n_eq, const uint64_t* cell_ptr,
              const uint32_t* cell_eq, const uint32_t* cell_count, uint32_t n_cells, uint32_t num_alphas, uint32_t usa_mode,
              afq_result* out) {
    if (!c) return AFQ_ERR_INVALID_ARG;
    if (!eq_labels || !eq_label_ptr || !cell_ptr || !out || (cell_ptr[n_cells] && (!cell_eq || !cell_count))) return fail(c, AFQ_ERR_INVALID_ARG, "null argument");
    if (c->pending) return fail(c, AFQ_ERR_STATE, "a quant batch is pending on this context");
    HIP_TRY(c, hipSetDevice(c->device));
    // per cell: its classes' label lengths, counts and label words, gathered from the global list (row order = class id order)
    std::vector<uint64_t> cp(n_cells + 1), wp(n_cells + 1), so(n_cells + 1);
    cp[0] = wp[0] = so[0] = 0;
    const bool usa = usa_mode != 0;
    for (uint32_t i = 0; i < n_cells; ++i) {
        if (cell_ptr[i + 1] < cell_ptr[i]) return fail(c, AFQ_ERR_INVALID_ARG, "cell_ptr must be non-decreasing");
        uint64_t w = 0;
        for (uint64_t k = cell_ptr[i]; k < cell_ptr[i + 1]; ++k) {
            if (cell_eq[k] >= n_eq) return fail(c, AFQ_ERR_BAD_INPUT, "equivalence class id out of range");
            w += eq_label_ptr[cell_eq[k] + 1] - eq_label_ptr[cell_eq[k]];
        }
        cp[i + 1] = cell_ptr[i + 1];
        wp[i + 1] = wp[i] + w;
        so[i + 1] = so[i] + ((infer_scratch_words(cell_ptr[i + 1] - cell_ptr[i], w, usa) + 1) & ~1ull);
    }
    const uint64_t nk = cp[n_cells], nw = wp[n_cells], wmul = usa ? 3 : 1;
    std::vector<uint32_t> len(nk), lab(nw);
    {
        uint64_t w = 0;
        for (uint64_t k = 0; k < nk; ++k) {
            const uint64_t a = eq_label_ptr[cell_eq[k]], b = eq_label_ptr[cell_eq[k] + 1];
            len[k] = (uint32_t)(b - a);
            for (uint64_t j = a; j < b; ++j) {
                if (eq_labels[j] >= num_alphas) return fail(c, AFQ_ERR_BAD_INPUT, "equivalence-class label is outside the alphas");   // em.rs:72-75
                lab[w++] = eq_labels[j];
            }
        }
    }
    RangeState& B = c->rs[0];
    hipStream_t s = B.stream;
    HIP_TRY(c, B.d_eq_cptr.ensure(8ull * (n_cells + 1))); HIP_TRY(c, B.d_eq_wptr.ensure(8ull * (n_cells + 1)));
    HIP_TRY(c, B.d_eq_len.ensure(std::max<uint64_t>(4 * nk, 16))); HIP_TRY(c, B.d_eq_cnt.ensure(std::max<uint64_t>(4 * nk, 16)));
    HIP_TRY(c, B.d_eq_lab.ensure(std::max<uint64_t>(4 * nw, 16)));
    HIP_TRY(c, B.d_bt_off.ensure(8ull * (n_cells + 1))); HIP_TRY(c, B.d_bt_scratch.ensure(4 * so[n_cells] + 16));
    HIP_TRY(c, B.d_bt_ns.ensure(4ull * std::max<uint32_t>(n_cells, 1)));
    HIP_TRY(c, B.d_bt_col.ensure(std::max<uint64_t>(4 * nw * wmul, 16))); HIP_TRY(c, B.d_bt_mean.ensure(std::max<uint64_t>(4 * nw * wmul, 16)));
    HIP_TRY(c, hipMemcpyAsync(B.d_eq_cptr.p, cp.data(), 8ull * (n_cells + 1), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(B.d_eq_wptr.p, wp.data(), 8ull * (n_cells + 1), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(B.d_bt_off.p, so.data(), 8ull * (n_cells + 1), hipMemcpyHostToDevice, s));
    if (nk) {
        HIP_TRY(c, hipMemcpyAsync(B.d_eq_len.p, len.data(), 4 * nk, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(B.d_eq_cnt.p, cell_count, 4 * nk, hipMemcpyHostToDevice, s));
    }
    if (nw) HIP_TRY(c, hipMemcpyAsync(B.d_eq_lab.p, lab.data(), 4 * nw, hipMemcpyHostToDevice, s));
    launch_infer(s, n_cells, B.d_eq_cptr.as<uint64_t>(), B.d_eq_wptr.as<uint64_t>(), B.d_eq_len.as<uint32_t>(), B.d_eq_cnt.as<uint32_t>(),
                 B.d_eq_lab.as<uint32_t>(), B.d_bt_off.as<uint64_t>(), B.d_bt_scratch.as<uint32_t>(), usa ? 1u : 0u, num_alphas,
                 B.d_bt_ns.as<uint32_t>(), B.d_bt_col.as<uint32_t>(), B.d_bt_mean.as<float>());
    std::vector<uint32_t> ns(n_cells), col(nw * wmul);
    std::vector<float> al(nw * wmul);
    if (n_cells) HIP_TRY(c, hipMemcpyAsync(ns.data(), B.d_bt_ns.p, 4ull * n_cells, hipMemcpyDeviceToHost, s));
    if (nw) {
        HIP_TRY(c, hipMemcpyAsync(col.data(), B.d_bt_col.p, 4 * nw * wmul, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(al.data(), B.d_bt_mean.p, 4 * nw * wmul, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    HostResult* R = pool_get(c->pool);
    R->cell_ptr.push_back(0);
    uint64_t tot = 0;
    for (uint32_t i = 0; i < n_cells; ++i) for (uint32_t k = 0; k < ns[i]; ++k) tot += al[wp[i] * wmul + k] > 0.0f;
    if (R->gene.reserve(tot) || R->val.reserve(tot)) { pool_put(R); return fail(c, AFQ_ERR_HIP, "host allocation failed"); }
    R->gene.n = R->val.n = tot;
    uint64_t o = 0;
    for (uint32_t i = 0; i < n_cells; ++i) {
        for (uint32_t k = 0; k < ns[i]; ++k) {   // expressed_ind / expressed_vec, infer.rs:233-241
            const float a = al[wp[i] * wmul + k];
            if (a > 0.0f) { R->gene.p[o] = col[wp[i] * wmul + k]; R->val.p[o] = a; ++o; }
        }
        R->cell_ptr.push_back(o);
        R->bc.push_back(0); R->nrec.push_back(0); R->flags.push_back(0);
    }
    std::memset(out, 0, sizeof(*out));
    out->n_cells = n_cells; out->nnz = tot;
    out->cell_ptr = R->cell_ptr.data(); out->gene = R->gene.p; out->val = R->val.p;
    out->bc = R->bc.data(); out->nrec = R->nrec.data(); out->flags = R->flags.data();
    out->opaque = R;
    return 0;
}

// The ATAC entry points hand out gigabytes of results: into pageable malloc memory the D2H runs at a fraction of the link.
// Their output arrays are pinned instead and recycled through a process-wide pool (afq_free returns them to it).
namespace {
struct PinnedPool {
    std::mutex mu;
    struct Ent { void* p; size_t cap; bool busy; };
    std::vector<Ent> ents;
    void* get(size_t n) {
        std::lock_guard<std::mutex> g(mu);
        Ent* best = nullptr;
        for (auto& e : ents) if (!e.busy && e.cap >= n && (!best || e.cap < best->cap)) best = &e;
        if (best) { best->busy = true; return best->p; }
        void* q = nullptr;
        const size_t cap = n + n / 8 + 4096;
        if (hipHostMalloc(&q, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        ents.push_back({q, cap, true});
        return q;
    }
    bool put(void* p) {
        std::lock_guard<std::mutex> g(mu);
        for (auto& e : ents) if (e.p == p) { e.busy = false; return true; }
        return false;
    }
};
PinnedPool* pinned_pool() { static PinnedPool* P = new PinnedPool(); return P; }
}  // namespace

// Shared back half of the two ATAC entry points: de-duplicate the fragments sitting in d_ref/d_start/d_flen (cell i at
// d_ptr[i], cell_cnt[i] of them when d_cnt is given, else up to d_ptr[i+1]) and hand the distinct ones out as malloc'd arrays.
static int atac_dedup_device(afq_ctx* c, uint64_t n, uint32_t n_cells, const uint32_t* d_cnt, uint64_t** out_cell_ptr, uint32_t** out_ref,
                             uint32_t** out_start, uint16_t** out_frag_len, uint16_t** out_count, HostClock& hc, unsigned long long* tally_out = nullptr) {
    DevBuf &d_ref = c->atac[0], &d_start = c->atac[1], &d_flen = c->atac[2], &d_ptr = c->atac[3], &d_scr = c->atac[4],
           &d_oref = c->atac[5], &d_ostart = c->atac[6], &d_oflen = c->atac[7], &d_ocnt = c->atac[8], &d_on = c->atac[9],
           &d_optr = c->atac[10], &d_cref = c->atac[11], &d_cstart = c->atac[12], &d_cflen = c->atac[13], &d_ccnt = c->atac[14],
           &d_flag = c->atac[15];
    hipStream_t s = c->stream;
    const uint64_t n1 = std::max<uint64_t>(n, 1);
    hipError_t e = hipSuccess;
    auto T = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    T(d_scr.ensure(16 * n1)); T(d_oref.ensure(4 * n1)); T(d_ostart.ensure(4 * n1)); T(d_oflen.ensure(2 * n1));
    T(d_ocnt.ensure(2 * n1)); T(d_on.ensure(4ull * std::max<uint32_t>(n_cells, 1))); T(d_optr.ensure(8ull * (n_cells + 1)));
    T(d_flag.ensure(4));
    if (e == hipSuccess) T(hipMemsetAsync(d_flag.p, 0, 4, s));
    std::vector<uint32_t> on(n_cells);
    uint32_t wide = 0;
    if (e == hipSuccess) {
        ScopedTimer t(c, K_ATAC, s);
        launch_atac_dedup64(s, n_cells, d_ref.as<uint32_t>(), d_start.as<uint32_t>(), d_flen.as<uint16_t>(), d_ptr.as<uint64_t>(),
                            d_scr.p, d_oref.as<uint32_t>(), d_ostart.as<uint32_t>(), d_oflen.as<uint16_t>(),
                            d_ocnt.as<uint16_t>(), d_on.as<uint32_t>(), d_flag.as<uint32_t>(), d_cnt);
        T(hipGetLastError());
    }
    if (e == hipSuccess) {
        T(hipMemcpyAsync(&wide, d_flag.p, 4, hipMemcpyDeviceToHost, s));
        T(hipStreamSynchronize(s));
    }
    if (e == hipSuccess && wide) {  // a reference id >= 65536: the 16-byte-record kernel
        ScopedTimer t(c, K_ATAC, s);
        launch_atac_dedup(s, n_cells, d_ref.as<uint32_t>(), d_start.as<uint32_t>(), d_flen.as<uint16_t>(), d_ptr.as<uint64_t>(),
                          d_scr.p, d_oref.as<uint32_t>(), d_ostart.as<uint32_t>(), d_oflen.as<uint16_t>(),
                          d_ocnt.as<uint16_t>(), d_on.as<uint32_t>(), d_cnt);
        T(hipGetLastError());
    }
    if (e == hipSuccess && n_cells) T(hipMemcpyAsync(on.data(), d_on.p, 4ull * n_cells, hipMemcpyDeviceToHost, s));
    if (e == hipSuccess) T(hipStreamSynchronize(s));
    hc.lap("atac: kernel");
    if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? AFQ_ERR_OOM : AFQ_ERR_HIP, std::string("afq_atac_dedup: ") + hipGetErrorString(e));
    uint64_t* optr = (uint64_t*)std::malloc(8ull * (n_cells + 1));
    if (!optr) return fail(c, AFQ_ERR_OOM, "afq_atac_dedup: host allocation failed");
    optr[0] = 0;
    for (uint32_t i = 0; i < n_cells; ++i) optr[i + 1] = optr[i] + on[i];
    const uint64_t tot = optr[n_cells], tot1 = std::max<uint64_t>(tot, 1);
    uint32_t* oref = (uint32_t*)pinned_pool()->get(4 * tot1);
    uint32_t* ostart = (uint32_t*)pinned_pool()->get(4 * tot1);
    uint16_t* oflen = (uint16_t*)pinned_pool()->get(2 * tot1);
    uint16_t* ocnt = (uint16_t*)pinned_pool()->get(2 * tot1);
    if (!oref || !ostart || !oflen || !ocnt) {
        std::free(optr); afq_free(oref); afq_free(ostart); afq_free(oflen); afq_free(ocnt);
        return fail(c, AFQ_ERR_OOM, "afq_atac_dedup: host allocation failed");
    }
    // dense runs on the device, then straight into the caller's arrays
    T(d_cref.ensure(4 * tot1)); T(d_cstart.ensure(4 * tot1)); T(d_cflen.ensure(2 * tot1)); T(d_ccnt.ensure(2 * tot1));
    if (e == hipSuccess) T(hipMemcpyAsync(d_optr.p, optr, 8ull * (n_cells + 1), hipMemcpyHostToDevice, s));
    DevBuf& d_tally = c->atac[24];
    T(d_tally.ensure(16));
    if (e == hipSuccess) T(hipMemsetAsync(d_tally.p, 0, 16, s));
    if (e == hipSuccess && tot) {
        launch_atac_compact(s, n_cells, d_ptr.as<uint64_t>(), d_optr.as<uint64_t>(), d_oref.as<uint32_t>(), d_ostart.as<uint32_t>(),
                            d_oflen.as<uint16_t>(), d_ocnt.as<uint16_t>(), d_cref.as<uint32_t>(), d_cstart.as<uint32_t>(),
                            d_cflen.as<uint16_t>(), d_ccnt.as<uint16_t>(), tally_out ? d_tally.as<unsigned long long>() : nullptr);
        T(hipGetLastError());
        T(hipMemcpyAsync(oref, d_cref.p, 4 * tot, hipMemcpyDeviceToHost, s));
        T(hipMemcpyAsync(ostart, d_cstart.p, 4 * tot, hipMemcpyDeviceToHost, s));
        T(hipMemcpyAsync(oflen, d_cflen.p, 2 * tot, hipMemcpyDeviceToHost, s));
        T(hipMemcpyAsync(ocnt, d_ccnt.p, 2 * tot, hipMemcpyDeviceToHost, s));
    }
    if (e == hipSuccess && tally_out) T(hipMemcpyAsync(tally_out, d_tally.p, 16, hipMemcpyDeviceToHost, s));
    if (e == hipSuccess) T(hipStreamSynchronize(s));
    hc.lap("atac: compact + D2H");
    harvest_timers(c);
    if (e != hipSuccess) {
        std::free(optr); afq_free(oref); afq_free(ostart); afq_free(oflen); afq_free(ocnt);
        return fail(c, e == hipErrorOutOfMemory ? AFQ_ERR_OOM : AFQ_ERR_HIP, std::string("afq_atac_dedup: ") + hipGetErrorString(e));
    }
    *out_cell_ptr = optr; *out_ref = oref; *out_start = ostart; *out_frag_len = oflen; *out_count = ocnt;
    return 0;
}

int afq_atac_dedup(afq_ctx* c, const uint32_t* ref, const uint32_t* start, const uint16_t* frag_len,
                   const uint64_t* cell_ptr, uint32_t n_cells, uint64_t** out_cell_ptr, uint32_t** out_ref,
                   uint32_t** out_start, uint16_t** out_frag_len, uint16_t** out_count) {
    if (!c) return AFQ_ERR_INVALID_ARG;
    if (!cell_ptr || !out_cell_ptr || !out_ref || !out_start || !out_frag_len || !out_count)
        return fail(c, AFQ_ERR_INVALID_ARG, "null argument");
    if (c->pending) return fail(c, AFQ_ERR_STATE, "a quant batch is pending on this context");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t n = cell_ptr[n_cells];
    if (n && (!ref || !start || !frag_len)) return fail(c, AFQ_ERR_INVALID_ARG, "null argument");
    for (uint32_t i = 0; i < n_cells; ++i)
        if (cell_ptr[i + 1] < cell_ptr[i] || cell_ptr[i + 1] - cell_ptr[i] > 0x7FFFFFFFull)
            return fail(c, AFQ_ERR_INVALID_ARG, "cell_ptr must be non-decreasing");
    HostClock hc;
    // device buffers live in the context and are reused by the next call (hipMalloc/hipFree of GBs is not free)
    DevBuf &d_ref = c->atac[0], &d_start = c->atac[1], &d_flen = c->atac[2], &d_ptr = c->atac[3];
    hipStream_t s = c->stream;
    const uint64_t n1 = std::max<uint64_t>(n, 1);
    hipError_t e = hipSuccess;
    auto T = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    T(d_ref.ensure(4 * n1)); T(d_start.ensure(4 * n1)); T(d_flen.ensure(2 * n1)); T(d_ptr.ensure(8ull * (n_cells + 1)));
    if (e == hipSuccess && n) {   // (the pinned staging path of afq_submit was measured here too: no faster for these arrays)
        T(hipMemcpyAsync(d_ref.p, ref, 4 * n, hipMemcpyHostToDevice, s));
        T(hipMemcpyAsync(d_start.p, start, 4 * n, hipMemcpyHostToDevice, s));
        T(hipMemcpyAsync(d_flen.p, frag_len, 2 * n, hipMemcpyHostToDevice, s));
    }
    if (e == hipSuccess) T(hipMemcpyAsync(d_ptr.p, cell_ptr, 8ull * (n_cells + 1), hipMemcpyHostToDevice, s));
    if (hc.on) { T(hipStreamSynchronize(s)); hc.lap("atac: alloc + H2D"); }
    if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? AFQ_ERR_OOM : AFQ_ERR_HIP, std::string("afq_atac_dedup: ") + hipGetErrorString(e));
    for (int i = 0; i < K_COUNT; ++i) { c->k_ms[i] = 0; c->k_launches[i] = 0; }
    return atac_dedup_device(c, n, n_cells, nullptr, out_cell_ptr, out_ref, out_start, out_frag_len, out_count, hc);
}

int afq_atac_dedup_rad(afq_ctx* c, const uint8_t* bytes, size_t n_bytes, const uint64_t* chunk_off, uint32_t n_cells, uint32_t bc_bytes,
                       int bytes_on_device, uint64_t** out_cell_ptr, uint64_t** out_bc, uint32_t** out_ref, uint32_t** out_start,
                       uint16_t** out_frag_len, uint16_t** out_count, afq_atac_stats* stats) {
    if (!c) return AFQ_ERR_INVALID_ARG;
    if ((!bytes && n_bytes) || (!chunk_off && n_cells) || !out_cell_ptr || !out_bc || !out_ref || !out_start || !out_frag_len || !out_count)
        return fail(c, AFQ_ERR_INVALID_ARG, "null argument");
    if (!valid_width(bc_bytes)) return fail(c, AFQ_ERR_INVALID_ARG, "bc_bytes must be 1, 2, 4 or 8");
    if (c->pending) return fail(c, AFQ_ERR_STATE, "a quant batch is pending on this context");
    HIP_TRY(c, hipSetDevice(c->device));
    HostClock hc;
    hipStream_t s = c->stream;
    // chunk headers (from the host copy, or gathered off the device), capacity offsets = prefix of nrec
    std::vector<uint32_t> hdr(2ull * n_cells);
    if (!bytes_on_device) {
        for (uint32_t i = 0; i < n_cells; ++i) {
            if (chunk_off[i] + 8 > n_bytes) return fail(c, AFQ_ERR_BAD_INPUT, "cell " + std::to_string(i) + ": chunk offset out of range");
            std::memcpy(&hdr[2 * i], bytes + chunk_off[i], 8);
        }
    } else if (n_cells) {
        HIP_TRY(c, c->d_chunk_off.ensure(8ull * n_cells));
        HIP_TRY(c, c->d_hdr.ensure(8ull * n_cells));
        HIP_TRY(c, hipMemcpyAsync(c->d_chunk_off.p, chunk_off, 8ull * n_cells, hipMemcpyHostToDevice, s));
        launch_gather_headers(s, bytes, n_bytes, c->d_chunk_off.as<uint64_t>(), n_cells, c->d_hdr.as<uint32_t>());
        HIP_TRY(c, hipMemcpyAsync(hdr.data(), c->d_hdr.p, 8ull * n_cells, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    std::vector<AtacCell> cells(n_cells);
    std::vector<uint64_t> cap_ptr(n_cells + 1, 0);
    uint64_t bm_words = 0, n_rec = 0;
    for (uint32_t i = 0; i < n_cells; ++i) {
        const uint32_t nb = hdr[2 * i], nr = hdr[2 * i + 1];
        if (chunk_off[i] + 8 > n_bytes || nb < 8 || chunk_off[i] + nb > n_bytes) return fail(c, AFQ_ERR_BAD_INPUT, "cell " + std::to_string(i) + ": chunk header/size out of range");
        if ((uint64_t)nr * (4 + bc_bytes) + 8 > nb) return fail(c, AFQ_ERR_BAD_INPUT, "cell " + std::to_string(i) + ": chunk nbytes does not match its records");
        cells[i] = AtacCell{chunk_off[i], n_rec, bm_words, nb, nr};
        cap_ptr[i] = n_rec;
        n_rec += nr;
        bm_words += 4ull * (((uint64_t)nb + 3 + 255) / 256);   // four ballots per group of 256 positions (counted from the dword boundary below the chunk)
    }
    cap_ptr[n_cells] = n_rec;
    DevBuf &d_ref = c->atac[0], &d_start = c->atac[1], &d_flen = c->atac[2], &d_ptr = c->atac[3];
    DevBuf &d_cells = c->atac[16], &d_bm = c->atac[17], &d_cnt = c->atac[18], &d_bc = c->atac[19], &d_stat = c->atac[20], &d_walk = c->atac[21],
           &d_nwalk = c->atac[22], &d_status = c->atac[23];
    const uint64_t n1 = std::max<uint64_t>(n_rec, 1), nc1 = std::max<uint32_t>(n_cells, 1);
    HIP_TRY(c, d_ref.ensure(4 * n1)); HIP_TRY(c, d_start.ensure(4 * n1)); HIP_TRY(c, d_flen.ensure(2 * n1)); HIP_TRY(c, d_ptr.ensure(8ull * (n_cells + 1)));
    HIP_TRY(c, d_cells.ensure(sizeof(AtacCell) * nc1)); HIP_TRY(c, d_bm.ensure(8 * std::max<uint64_t>(bm_words, 1))); HIP_TRY(c, d_cnt.ensure(4ull * nc1));
    HIP_TRY(c, d_bc.ensure(8ull * nc1)); HIP_TRY(c, d_stat.ensure(8ull * nc1)); HIP_TRY(c, d_walk.ensure(4ull * nc1)); HIP_TRY(c, d_nwalk.ensure(4));
    HIP_TRY(c, d_status.ensure(sizeof(DevStatus)));
    const uint8_t* d_bytes = bytes;
    if (!bytes_on_device) {
        HIP_TRY(c, c->d_bytes_own.ensure(n_bytes + 16));
        if (n_bytes) { int rc2 = staged_h2d(c, (uint8_t*)c->d_bytes_own.p, bytes, n_bytes, s, host_ptr_is_pinned(bytes) && host_ptr_is_pinned(bytes + n_bytes - 1)); if (rc2) return rc2; }
        d_bytes = c->d_bytes_own.as<uint8_t>();
    }
    if (n_cells) HIP_TRY(c, hipMemcpyAsync(d_cells.p, cells.data(), sizeof(AtacCell) * n_cells, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_ptr.p, cap_ptr.data(), 8ull * (n_cells + 1), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemsetAsync(d_nwalk.p, 0, 4, s));
    HIP_TRY(c, hipMemsetAsync(d_status.p, 0, sizeof(DevStatus), s));
    if (hc.on) { HIP_TRY(c, hipStreamSynchronize(s)); hc.lap("atac: alloc + H2D"); }
    for (int i = 0; i < K_COUNT; ++i) { c->k_ms[i] = 0; c->k_launches[i] = 0; }
    DevStatus st{};
    std::vector<uint32_t> stat(2ull * n_cells);
    uint64_t* obc = (uint64_t*)std::malloc(8ull * nc1);
    if (!obc) return fail(c, AFQ_ERR_OOM, "afq_atac_dedup_rad: host allocation failed");
    unsigned long long tally[2] = {0, 0};
    auto parse_args = [&](uint32_t c0, uint32_t c1, uint32_t* nwalk, DevStatus* dst) {
        return AtacParseArgs{d_bytes, d_cells.as<AtacCell>() + c0, c1 - c0, bc_bytes, d_bm.as<uint64_t>(), d_ref.as<uint32_t>(), d_start.as<uint32_t>(),
                             d_flen.as<uint16_t>(), d_cnt.as<uint32_t>() + c0, d_bc.as<uint64_t>() + c0, d_stat.as<uint32_t>() + 2ull * c0,
                             d_walk.as<uint32_t>() + c0, nwalk, dst, (uint64_t)n_bytes};
    };
    // Big batches go through in eight ranges of cells: the distinct fragments of range r cross PCIe on a second stream while the
    // later ranges are still parsed and sorted.  The rows are the long pole - 12 bytes a row, 1.8 GB for 2*10^8 records, 34 ms at
    // the 52 GB/s the link gives (the e2e leg), against 20 ms for all the kernels - so (round 5) the ref column stays on the device
    // (8 bytes a row cross, 23 ms; the host writes the column from a list of runs, below), and what matters then is how soon the
    // FIRST rows can leave and how few are left when the last kernel ends: the ranges GROW (4, 8, 12, 14, 15, 16, 16, 15 % of the
    // records; four equal ranges kept the link idle for the first quarter of the kernels), the opposite of the cr-like taper,
    // whose rows are short.  Measured and not kept: the parse of range r+1 on a stream of its own next to the sort of range r
    // (the sort's workgroups wait for CUs behind the parse's: 12.7 -> 18.3 ms of sort, 28.6 -> 30.9 ms a step).
    const char* pipe_env = test_hook("ATAC_PIPE_BYTES");   // (tests: pipeline small inputs too)
    const size_t pipe_min = pipe_env ? (size_t)std::atoll(pipe_env) : ((size_t)128 << 20);
    const bool piped = n_cells >= 8 && n_bytes >= pipe_min;
    bool piped_done = false;
    if (piped) {
        constexpr uint32_t kR = 8;
        static const double kGrow[kR] = {0.04, 0.12, 0.24, 0.38, 0.53, 0.69, 0.85, 1.0};
        uint32_t cut[kR + 1];
        cut[0] = 0;
        for (uint32_t r = 1; r < kR; ++r) {
            const uint64_t target = (uint64_t)((double)n_rec * kGrow[r - 1]);
            cut[r] = (uint32_t)(std::lower_bound(cap_ptr.begin(), cap_ptr.begin() + n_cells, target) - cap_ptr.begin());
            if (cut[r] < cut[r - 1]) cut[r] = cut[r - 1];
        }
        cut[kR] = n_cells;
        DevBuf &d_scr = c->atac[4], &d_oref = c->atac[5], &d_ostart = c->atac[6], &d_oflen = c->atac[7], &d_ocnt = c->atac[8], &d_on = c->atac[9],
               &d_optr = c->atac[10], &d_cref = c->atac[11], &d_cstart = c->atac[12], &d_cflen = c->atac[13], &d_ccnt = c->atac[14],
               &d_flag = c->atac[15], &d_tally = c->atac[24], &d_rstat = c->atac[25], &d_runctr = c->atac[26];
        hipError_t e = hipSuccess;
        auto T = [&](hipError_t x) { if (e == hipSuccess) e = x; };
        T(d_scr.ensure(16 * n1)); T(d_oref.ensure(4 * n1)); T(d_ostart.ensure(4 * n1)); T(d_oflen.ensure(2 * n1)); T(d_ocnt.ensure(2 * n1));
        T(d_on.ensure(4ull * nc1)); T(d_optr.ensure(8ull * (n_cells + 1))); T(d_flag.ensure(4)); T(d_tally.ensure(16));
        T(d_cref.ensure(4 * n1)); T(d_cstart.ensure(4 * n1)); T(d_cflen.ensure(2 * n1)); T(d_ccnt.ensure(2 * n1));   // (sized for "nothing is a duplicate")
        T(d_rstat.ensure(kR * (sizeof(DevStatus) + 16))); T(d_runctr.ensure(4));
        // The ref column does not cross PCIe: a cell's rows are sorted by ref first, so the column is a few runs per cell - the
        // compaction kernel lists them (first row, length, ref: 16 bytes a run, written straight into pinned host memory) and
        // host threads write the column from the list while the other three columns (8 of the 12 bytes of a row) are on the link.
        // A list that overflows (more than 64 runs per cell on average: thousands of contigs) sends the column itself, as before.
        const long cap_hook = test_hook_long("ATAC_RUN_CAP", -1);   // (tests: force the overflow)
        // (at most 2^22 runs = 64 MiB of pinned memory whatever the number of barcodes - an unfiltered sample has a million of them; a
        //  list that does not fit, or that the host has no pinned memory for, is the overflow case: the column crosses as a copy)
        uint32_t run_cap = cap_hook >= 0 ? (uint32_t)cap_hook : (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1u << 20, 64ull * n_cells), 1u << 22);
        uint4* runs = (uint4*)pinned_pool()->get(16ull * std::max<uint32_t>(run_cap, 1));
        if (!runs) { run_cap = 0; runs = (uint4*)pinned_pool()->get(16); }
        uint32_t* oref = (uint32_t*)pinned_pool()->get(4 * n1);
        uint32_t* ostart = (uint32_t*)pinned_pool()->get(4 * n1);
        uint16_t* oflen = (uint16_t*)pinned_pool()->get(2 * n1);
        uint16_t* ocnt = (uint16_t*)pinned_pool()->get(2 * n1);
        uint64_t* optr = (uint64_t*)std::malloc(8ull * (n_cells + 1));
        auto drop = [&]() { std::free(optr); afq_free(oref); afq_free(ostart); afq_free(oflen); afq_free(ocnt); afq_free(runs); };
        if (!oref || !ostart || !oflen || !ocnt || !optr || !runs) { drop(); std::free(obc); return fail(c, AFQ_ERR_OOM, "afq_atac_dedup_rad: host allocation failed"); }
        hipStream_t s2 = c->rs[0].stream;
        DevStatus* d_rst = reinterpret_cast<DevStatus*>(d_rstat.p);
        uint32_t* d_rnw = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(d_rstat.p) + kR * sizeof(DevStatus));
        // (the per-range counts and flags land in PINNED memory: an async copy into pageable memory holds the host until the
        // stream gets there, and the ranges would be enqueued one at a time)
        uint8_t* pin = (uint8_t*)pinned_pool()->get(4ull * nc1 + kR * (sizeof(DevStatus) + 16));
        if (!pin) { drop(); std::free(obc); return fail(c, AFQ_ERR_OOM, "afq_atac_dedup_rad: host allocation failed"); }
        uint32_t* on = reinterpret_cast<uint32_t*>(pin);
        DevStatus* rst = reinterpret_cast<DevStatus*>(pin + 4ull * nc1);
        uint32_t* wide = reinterpret_cast<uint32_t*>(pin + 4ull * nc1 + kR * sizeof(DevStatus));
        volatile uint32_t* snap = wide + kR;   // the run counter as it stood after each range's compaction (the block has four words per range)
        uint32_t* pin_dev = nullptr;   // the same block as the device sees it
        T(hipHostGetDevicePointer(reinterpret_cast<void**>(&pin_dev), pin, 0));
        uint4* runs_dev = nullptr;
        T(hipHostGetDevicePointer(reinterpret_cast<void**>(&runs_dev), runs, 0));
        hipEvent_t ev[kR], ev2[kR];
        for (auto& x : ev) x = get_event(c);
        for (auto& x : ev2) x = get_event(c);
        T(hipMemsetAsync(d_runctr.p, 0, 4, s));
        // the threads that write the ref column: thread 0 waits for a range's list (the event after its compaction), all fill
        std::atomic<int> recorded{0}, listed{0}, stop{0};
        const unsigned nfill = stage_threads();
        std::vector<std::thread> fillers;
        if (e == hipSuccess) {
            for (unsigned t = 0; t < nfill; ++t)
                fillers.emplace_back([&, t]() {
                    if (t == 0) (void)hipSetDevice(c->device);
                    uint32_t lo = 0;
                    for (uint32_t r = 0; r < kR; ++r) {
                        if (t == 0) {
                            while (recorded.load(std::memory_order_acquire) <= (int)r) { if (stop.load(std::memory_order_relaxed)) return; std::this_thread::sleep_for(std::chrono::microseconds(20)); }
                            if (hipEventSynchronize(ev2[r]) != hipSuccess) { (void)hipGetLastError(); stop.store(1); return; }
                            listed.store((int)r + 1, std::memory_order_release);
                        } else {
                            while (listed.load(std::memory_order_acquire) <= (int)r) { if (stop.load(std::memory_order_relaxed)) return; std::this_thread::sleep_for(std::chrono::microseconds(20)); }
                        }
                        const uint32_t hi = snap[r];
                        if (hi > run_cap) return;   // this range's column and every later one come as copies
                        for (uint32_t i = lo + t; i < hi; i += nfill) {
                            const uint4 q = runs[i];
                            std::fill_n(oref + (((uint64_t)q.y << 32) | q.x), (size_t)q.z, q.w);
                        }
                        lo = hi;
                    }
                });
        }
        T(hipMemsetAsync(d_flag.p, 0, 4, s));
        T(hipMemsetAsync(d_tally.p, 0, 16, s));
        T(hipMemsetAsync(d_rstat.p, 0, kR * (sizeof(DevStatus) + 16), s));
        for (uint32_t r = 0; r < kR && e == hipSuccess; ++r) {
            const uint32_t c0 = cut[r], nr = cut[r + 1] - c0;
            { ScopedTimer t(c, K_ATAC_PARSE, s); launch_atac_parse(s, parse_args(c0, cut[r + 1], d_rnw + r, d_rst + r)); }
            { ScopedTimer t(c, K_ATAC, s);
              launch_atac_dedup64(s, nr, d_ref.as<uint32_t>(), d_start.as<uint32_t>(), d_flen.as<uint16_t>(), d_ptr.as<uint64_t>() + c0, d_scr.p,
                                  d_oref.as<uint32_t>(), d_ostart.as<uint32_t>(), d_oflen.as<uint16_t>(), d_ocnt.as<uint16_t>(), d_on.as<uint32_t>() + c0,
                                  d_flag.as<uint32_t>(), d_cnt.as<uint32_t>() + c0); }
            T(hipGetLastError());
            // (the range's counts, flag and status go to the pinned block by a kernel, not as copies: see k_copy_words3)
            launch_copy_words3(s, d_on.as<uint32_t>() + c0, nr, pin_dev + c0, d_flag.as<uint32_t>(), 1, pin_dev + (wide - on) + r,
                               reinterpret_cast<const uint32_t*>(d_rst + r), (uint32_t)(sizeof(DevStatus) / 4), pin_dev + (reinterpret_cast<uint32_t*>(rst + r) - on));
            T(hipGetLastError());
            T(hipEventRecord(ev[r], s));
        }
        optr[0] = 0;
        bool redo = false;
        int bad_rc = 0;
        for (uint32_t r = 0; r < kR && e == hipSuccess && !redo && !bad_rc; ++r) {
            const uint32_t c0 = cut[r], c1 = cut[r + 1];
            T(hipEventSynchronize(ev[r]));
            if (e != hipSuccess) break;
            if (wide[r]) { redo = true; break; }   // a reference id >= 65536: the plain route below runs the 16-byte-record kernel
            if (rst[r].err_code) { bad_rc = fail(c, AFQ_ERR_BAD_INPUT, "cell " + std::to_string(c0 + rst[r].err_cell) + ": chunk nbytes does not match its records"); break; }
            st.n_fallback += rst[r].n_fallback;
            for (uint32_t i = c0; i < c1; ++i) optr[i + 1] = optr[i] + on[i];
            const uint64_t o0 = optr[c0], tot_r = optr[c1] - o0;
            T(hipMemcpyAsync(d_optr.as<uint64_t>() + c0, optr + c0, 8ull * (c1 - c0 + 1), hipMemcpyHostToDevice, s2));
            if (tot_r) {
                launch_atac_compact(s2, c1 - c0, d_ptr.as<uint64_t>() + c0, d_optr.as<uint64_t>() + c0, d_oref.as<uint32_t>(), d_ostart.as<uint32_t>(),
                                    d_oflen.as<uint16_t>(), d_ocnt.as<uint16_t>(), d_cref.as<uint32_t>(), d_cstart.as<uint32_t>(),
                                    d_cflen.as<uint16_t>(), d_ccnt.as<uint16_t>(), d_tally.as<unsigned long long>(), runs_dev, d_runctr.as<uint32_t>(), run_cap);
                T(hipGetLastError());
            }
            launch_copy_words3(s2, d_runctr.as<uint32_t>(), 1, pin_dev + (const_cast<uint32_t*>(snap) - on) + r, nullptr, 0, nullptr, nullptr, 0, nullptr);
            T(hipGetLastError());
            T(hipEventRecord(ev2[r], s2));
            if (e == hipSuccess) recorded.store((int)r + 1, std::memory_order_release);
            if (tot_r) {
                T(hipMemcpyAsync(ostart + o0, d_cstart.as<uint32_t>() + o0, 4 * tot_r, hipMemcpyDeviceToHost, s2));
                T(hipMemcpyAsync(oflen + o0, d_cflen.as<uint16_t>() + o0, 2 * tot_r, hipMemcpyDeviceToHost, s2));
                T(hipMemcpyAsync(ocnt + o0, d_ccnt.as<uint16_t>() + o0, 2 * tot_r, hipMemcpyDeviceToHost, s2));
            }
        }
        if (e != hipSuccess || redo || bad_rc) stop.store(1);
        for (auto& th : fillers) th.join();
        if (e == hipSuccess && !redo && !bad_rc) {
            T(hipEventSynchronize(ev2[kR - 1]));   // (every range's count is on the host; the fillers stop at the first list that overflowed)
            for (uint32_t r = 0; r < kR && e == hipSuccess; ++r)
                if (snap[r] > run_cap) {
                    const uint64_t o0 = optr[cut[r]], tot_r = optr[cut[r + 1]] - o0;
                    if (tot_r) T(hipMemcpyAsync(oref + o0, d_cref.as<uint32_t>() + o0, 4 * tot_r, hipMemcpyDeviceToHost, s2));
                }
            T(hipMemcpyAsync(tally, d_tally.p, 16, hipMemcpyDeviceToHost, s2));
            if (n_cells) T(hipMemcpyAsync(stat.data(), d_stat.p, 8ull * n_cells, hipMemcpyDeviceToHost, s2));
            if (n_cells) T(hipMemcpyAsync(obc, d_bc.p, 8ull * n_cells, hipMemcpyDeviceToHost, s2));
        }
        (void)hipStreamSynchronize(s);
        (void)hipStreamSynchronize(s2);
        for (auto x : ev) c->event_pool.push_back(x);
        for (auto x : ev2) c->event_pool.push_back(x);
        afq_free(pin); afq_free(runs); runs = nullptr;
        harvest_timers(c);
        hc.lap("atac: ranges (parse, sort, compact, D2H)");
        if (e != hipSuccess || bad_rc) {
            drop(); std::free(obc);
            return bad_rc ? bad_rc : fail(c, e == hipErrorOutOfMemory ? AFQ_ERR_OOM : AFQ_ERR_HIP, std::string("afq_atac_dedup_rad: ") + hipGetErrorString(e));
        }
        if (redo) { drop(); st = DevStatus{}; for (int i = 0; i < K_COUNT; ++i) { c->k_ms[i] = 0; c->k_launches[i] = 0; } }
        else { *out_cell_ptr = optr; *out_ref = oref; *out_start = ostart; *out_frag_len = oflen; *out_count = ocnt; piped_done = true; }
    }
    if (!piped_done) {
        HIP_TRY(c, hipMemsetAsync(d_nwalk.p, 0, 4, s));
        HIP_TRY(c, hipMemsetAsync(d_status.p, 0, sizeof(DevStatus), s));
        {
            ScopedTimer t(c, K_ATAC_PARSE, s);
            launch_atac_parse(s, parse_args(0, n_cells, d_nwalk.as<uint32_t>(), d_status.as<DevStatus>()));
            HIP_TRY(c, hipGetLastError());
        }
        hipError_t e = hipMemcpyAsync(&st, d_status.p, sizeof(st), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && n_cells) e = hipMemcpyAsync(stat.data(), d_stat.p, 8ull * n_cells, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && n_cells) e = hipMemcpyAsync(obc, d_bc.p, 8ull * n_cells, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);   // the caller keeps ownership of `bytes`: the copy out of them is done by now, too
        if (e != hipSuccess) { std::free(obc); return fail(c, AFQ_ERR_HIP, std::string("afq_atac_dedup_rad: ") + hipGetErrorString(e)); }
        if (st.err_code) { std::free(obc); return fail(c, AFQ_ERR_BAD_INPUT, "cell " + std::to_string(st.err_cell) + ": chunk nbytes does not match its records"); }
        int rc = atac_dedup_device(c, n_rec, n_cells, d_cnt.as<uint32_t>(), out_cell_ptr, out_ref, out_start, out_frag_len, out_count, hc, tally);
        if (rc) { std::free(obc); return rc; }
    }
    *out_bc = obc;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->n_records = n_rec;
        for (uint32_t i = 0; i < n_cells; ++i) { stats->n_multimapped += stat[2 * i]; stats->n_not_mapped_pair += stat[2 * i + 1]; }
        const uint64_t tot = (*out_cell_ptr)[n_cells];
        stats->n_distinct = tot;
        stats->n_deduplicated = tally[0]; stats->n_long_fragments = tally[1];   // (tallied by the compaction kernel)
        stats->n_fallback_cells = st.n_fallback;
    }
    return 0;
}

void afq_free(void* p) { if (p && !pinned_pool()->put(p)) std::free(p); }

int afq_get_kernel_times(afq_ctx* c, afq_kernel_time* out, uint32_t cap) {
    if (!c || (!out && cap)) return AFQ_ERR_INVALID_ARG;
    uint32_t n = 0;
    for (int i = 0; i < K_COUNT; ++i) {
        if (!c->k_launches[i]) continue;
        if (n < cap) { out[n].name = kKernelNames[i]; out[n].ms = c->k_ms[i]; out[n].launches = c->k_launches[i]; out[n].pad = 0; }
        ++n;
    }
    return (int)std::min<uint32_t>(n, cap);
}

int afq_get_batch_stats(afq_ctx* c, afq_batch_stats* out) {
    if (!c || !out) return AFQ_ERR_INVALID_ARG;
    *out = c->stats;
    return 0;
}

}  // extern "C"
