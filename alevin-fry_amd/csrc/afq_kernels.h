// afq_kernels.h — host-callable launchers of the gfx950 kernels in afq_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "afq_common.h"

namespace afq {

struct DecodeArgs {
    const uint8_t* bytes;
    size_t n_bytes;
    const CellMeta* meta;
    uint32_t n_cells;
    const uint32_t* t2g;
    uint32_t ref_count;
    uint32_t num_genes;
    uint64_t* keys0;
    uint32_t* cell_nkeys;
    uint64_t* bc_out;
    DevStatus* st;
    // walk-free decode (k_decode_par) + fix-up mode of k_decode; null chk = plain sequential decode
    const CellChk* chk;
    const uint32_t* slab_prefix;  // [n_cells+1] 1 KiB slabs per cell, prefix
    uint32_t* slab_cell;          // [n_slabs] device-filled: cell of each slab
    uint64_t* cell_bc;            // [n_cells] device-filled: barcode words of each cell's first record
    uint32_t n_slabs;
    PugOut pug;                   // PUG cells: per-read outputs (null pointers when the batch has none)
    uint32_t trivial;             // the batch has cells in `trivial` mode
    uint32_t short_records;       // the batch averages < 2 alignment words per record: lane-per-record decode
    uint32_t* fix_list;           // [n_cells] cells whose walk-free proof failed (filled by k_verify_cells)
};

struct ResolveArgs {
    const CellMeta* meta;
    const uint32_t* bucket_cell;
    const uint32_t* multi_cells;
    const uint2* tile_desc;   // per scatter tile: (cell, tile index inside the cell)
    const uint32_t* cell_nkeys;
    uint32_t* cursor;      // per-bucket: count -> exclusive offset -> end offset
    uint64_t* keys0;
    uint64_t* keys1;
    uint32_t* cell_ncols;  // per-cell length of the resolved-column list (multi-bucket cells)
    uint32_t* nnz;
    OverflowEnt* ovf_list;
    void* bucket_desc;     // [n_buckets] x bucket_desc_bytes()
    uint32_t* lab;         // EM modes: label area, 2 words per key slot (null otherwise)
    uint32_t* lab_cnt;     // EM modes: per cell (label words, ambiguous molecules)
    DevStatus* st;
    uint32_t n_buckets;
    uint32_t n_multi;
    uint32_t n_tiles;
    const uint32_t* hist_cells;  // cells counted by k_cell_hist: multi-bucket cells + PUG cells
    uint32_t n_hist;
    uint32_t usa;
    uint32_t num_rows;
    uint32_t prefer_ambig;       // --sa-model prefer-ambig in USA mode (cr-like, cr-like-em)
    uint32_t max_lg_nb;          // largest lg_nb of the batch's multi-bucket cells (picks the scatter instance)
    uint32_t* slab_ovf;          // fixed-slab placement: per cell, set when one of its buckets outgrew its slab (the cell is then placed exactly)
    uint32_t slabs;              // the range's multi-bucket cells use fixed slabs (no k_hist / k_bucket_scan)
    uint32_t sort_only;          // reads of the range average two or more alignments: buckets are resolved by sorting, not through the UMI table
};

// one launch that clears / fills a range's small buffers (k_range_init): dst <- zeros (src_off == ~0) or arena[src_off ...)
struct RangeInitOp { void* dst; uint64_t src_off; uint64_t bytes; };
constexpr uint32_t kRangeInitOps = 32;
struct RangeInitOps { RangeInitOp op[kRangeInitOps]; uint32_t n; };
void launch_range_init(hipStream_t s, const RangeInitOps& ops, const uint8_t* arena);
constexpr uint32_t kPackHdrWords = 16;   // k_pack_small: words in front of the per-cell arrays
void launch_copy_words3(hipStream_t s, const uint32_t* a, uint32_t na, uint32_t* da, const uint32_t* b, uint32_t nb, uint32_t* db,
                        const uint32_t* c, uint32_t nc, uint32_t* dc);
void launch_pack_small(hipStream_t s, const DevStatus* st, const uint32_t* em_flag, const uint32_t* alt, const uint32_t* nnz, const uint32_t* em_nnz,
                       const uint64_t* bc, const uint32_t* n_mono, uint32_t n, uint32_t* out);
void launch_gather_headers(hipStream_t s, const uint8_t* bytes, size_t n_bytes, const uint64_t* chunk_off,
                           uint32_t n_cells, uint32_t* hdr);
int launch_decode(hipStream_t s, const DecodeArgs& a, uint32_t bw, uint32_t uw);
// chunks with 1/2-byte barcode or UMI fields rewritten with 4-byte ones (meta: the widened chunks; src_off: where each cell's chunk sits in src)

void launch_widen(hipStream_t s, const uint8_t* src, size_t n_src, const uint64_t* src_off, const CellMeta* meta, uint32_t n_cells,
                  uint32_t bw, uint32_t uw, uint32_t ebw, uint32_t euw, uint8_t* dst, DevStatus* st, uint32_t bsplit = 0);
bool decode_par_supported(uint32_t bw, uint32_t uw);
int launch_decode_par(hipStream_t s, const DecodeArgs& a, uint32_t bw, uint32_t uw);
void launch_hist(hipStream_t s, const ResolveArgs& a);
void launch_bucket_scan(hipStream_t s, const ResolveArgs& a);
void launch_fix_slabs(hipStream_t s, const ResolveArgs& a);
void launch_scatter(hipStream_t s, const ResolveArgs& a);
size_t bucket_desc_bytes();
uint64_t em_scratch_words(uint32_t nU, uint32_t W, uint32_t M, bool usa);
void launch_em(hipStream_t s, const ResolveArgs& a, uint32_t n_cells, const uint64_t* em_off, uint32_t* scratch,
               uint32_t* out_nnz, void* em_hdr /* 16 B per cell */, const uint32_t* em_order /* cells, largest first */,
               uint32_t num_alphas, uint32_t init_uniform, bool rounds = true /* false: the set-up only (the classes, for -d / -b) */);
// the EM in order-free fixed-point arithmetic (afq_em2.hip; the default): per-cell scratch words, launcher, and whether the
// output space fits the set-up kernel's bitmap (otherwise the canonical kernels above run)
uint64_t em2_scratch_words(uint32_t nU, uint32_t W, uint32_t M, bool usa);
bool em2_supported(uint32_t num_alphas);
// plan_cap_words != 0: the per-cell offsets are made on the device (k_em2_plan) from the counts the range's kernels left, packed
// into a scratch buffer of that many words; tiers[7] != 0 afterwards = it did not fit and nothing ran
void launch_em2(hipStream_t s, const ResolveArgs& a, uint32_t n_cells, uint64_t* em_off, uint32_t* scratch, uint32_t* out_nnz,
                const uint32_t* em_order, uint32_t* tiers /* 8 + 5 * n_cells words */, uint32_t num_alphas, uint32_t init_uniform,
                uint64_t plan_cap_words = 0);
// -d: sizes (cls_ptr == null) or fills the per-cell gene-level classes; see k_eqc_dump
void launch_eqc_dump(hipStream_t s, const ResolveArgs& a, uint32_t n_cells, const uint64_t* em_off, const uint32_t* scratch,
                     const void* em_hdr, uint32_t num_alphas, uint32_t* n_cls, uint32_t* n_words, const uint64_t* cls_ptr,
                     const uint64_t* word_ptr, uint32_t* o_len, uint32_t* o_count, uint32_t* o_labels);
// -b: bootstrap mean / variance per support entry of every cell, from the dumped classes; see k_boot
uint64_t boot_scratch_words(uint64_t K, uint64_t W, uint32_t B, bool summary_stat);
void launch_boot(hipStream_t s, uint32_t n_cells, const uint64_t* cls_ptr, const uint64_t* word_ptr, const uint32_t* len,
                 const uint32_t* cnt, const uint32_t* lab, const uint64_t* scr_off, uint32_t* scratch, uint32_t B, uint32_t summary_stat,
                 uint64_t seed, uint64_t first_cell_index, uint32_t* n_support, uint32_t* o_col, float* o_mean, float* o_var);
// `alevin-fry infer`: one EM per cell over its row of the equivalence-class count matrix (k_boot<true>); outputs per cell at
// word_ptr[cell] * (usa ? 3 : 1): the support's columns and abundances, n_support of them
uint64_t infer_scratch_words(uint64_t K, uint64_t W, bool usa);
void launch_infer(hipStream_t s, uint32_t n_cells, const uint64_t* cls_ptr, const uint64_t* word_ptr, const uint32_t* len,
                  const uint32_t* cnt, const uint32_t* lab, const uint64_t* scr_off, uint32_t* scratch, uint32_t usa, uint32_t num_alphas,
                  uint32_t* n_support, uint32_t* o_col, float* o_alpha);
void launch_boot_compact(hipStream_t s, uint32_t n_cells, const uint64_t* word_ptr, const uint64_t* sup_ptr, const uint32_t* i_col,
                         const float* i_mean, const float* i_var, uint32_t* o_col, float* o_mean, float* o_var);
void launch_compact_em(hipStream_t s, uint32_t n_cells, const uint64_t* em_off, const uint32_t* scratch, const uint32_t* nnz,
                       const uint64_t* cell_ptr, uint32_t* gene, float* val);
void launch_atac_dedup(hipStream_t s, uint32_t n_cells, const uint32_t* ref, const uint32_t* start, const uint16_t* flen,
                       const uint64_t* cell_ptr, void* scratch /* 16 B per fragment */, uint32_t* o_ref, uint32_t* o_start,
                       uint16_t* o_flen, uint16_t* o_cnt, uint32_t* o_n, const uint32_t* cell_cnt = nullptr);
void launch_atac_dedup64(hipStream_t s, uint32_t n_cells, const uint32_t* ref, const uint32_t* start, const uint16_t* flen,
                         const uint64_t* cell_ptr, void* scratch, uint32_t* o_ref, uint32_t* o_start, uint16_t* o_flen,
                         uint16_t* o_cnt, uint32_t* o_n, uint32_t* flag, const uint32_t* cell_cnt = nullptr);
// ATAC records straight from collated-RAD chunks (afq_atac.hip)
struct AtacCell { uint64_t chunk_off; uint64_t out_off; uint64_t bm_off; uint32_t nbytes; uint32_t nrec; };
struct AtacParseArgs {
    const uint8_t* bytes; const AtacCell* cells; uint32_t n_cells; uint32_t bc_bytes;
    uint64_t* bitmap; uint32_t* o_ref; uint32_t* o_start; uint16_t* o_flen; uint32_t* cell_cnt; uint64_t* cell_bc;
    uint32_t* cell_stat;   // [n_cells][2]: records with > 1 alignment, records that are not one properly mapped pair
    uint32_t* walk_list; uint32_t* n_walk; DevStatus* st;
    uint64_t n_bytes;      // size of the input buffer (aligned dword reads stop there)
};
void launch_atac_parse(hipStream_t s, const AtacParseArgs& a);
void launch_atac_compact(hipStream_t s, uint32_t n_cells, const uint64_t* cell_ptr, const uint64_t* out_ptr, const uint32_t* i_ref,
                         const uint32_t* i_start, const uint16_t* i_flen, const uint16_t* i_cnt, uint32_t* o_ref, uint32_t* o_start,
                         uint16_t* o_flen, uint16_t* o_cnt, unsigned long long* tally = nullptr, uint4* runs = nullptr, uint32_t* run_ctr = nullptr,
                         uint32_t run_cap = 0);
void warm_code_object();
void launch_resolve(hipStream_t s, const ResolveArgs& a);
void launch_resolve_big(hipStream_t s, const ResolveArgs& a);
void launch_cell_hist(hipStream_t s, const ResolveArgs& a);
void launch_compact(hipStream_t s, const CellMeta* meta, uint32_t n_cells, const uint64_t* keys0, const uint64_t* keys1,
                    const uint32_t* nnz, const uint64_t* cell_ptr, uint32_t* gene, float* val, uint64_t cap = ~0ull);
void launch_fill_tables(hipStream_t s, const CellMeta* meta, uint32_t n_cells, uint32_t* bucket_cell, uint2* tile_desc);   // bucket -> cell and tile -> (cell, tile) from the cells' plans
struct PackSmallArgs { const DevStatus* st; const uint32_t* em_flag; const uint32_t* alt; const uint32_t* em_nnz; const uint64_t* bc; const uint32_t* n_mono; uint32_t* out; };   // what k_pack_small packs (out: pinned host memory as the device sees it)
void launch_row_ptr(hipStream_t s, const uint32_t* nnz, uint32_t n, uint64_t* cell_ptr, const PackSmallArgs& pk);   // row lengths -> row offsets (+ total at [n]); pk.out != nullptr: and k_pack_small's work in the same launch

// ---- parsimony (afq_pug.hip) ----
struct PugCellArgs {
    const uint8_t* bytes;
    const CellMeta* meta;
    const uint32_t* pug_cells;
    const uint32_t* cell_nkeys;   // reads the decode emitted per cell
    PugOut rd;
    uint64_t scr_stride;          // words of scratch per WORKGROUP (sized for the largest cell of the batch, reused cell after cell)
    uint32_t* scratch;
    uint32_t* work_counter;       // next entry of pug_cells to take (persistent workgroups)
    uint32_t n_pug;
    uint32_t* epool;              // edge pool (u32 words) + multi-word adjacency rows
    unsigned long long* epool_cursor;
    unsigned long long epool_cap;
    const uint32_t* t2g;
    uint64_t* keys0;
    uint32_t* cell_ncols;
    uint32_t* lab;
    uint32_t* lab_cnt;
    uint32_t* alt;                // [n_cells] set to 1 when a component took the cr-like fallback
    DevStatus* st;
    uint32_t ref_count, num_genes, usa, num_rows, em, exact_umi, large_thresh, hw, umi_pairs, gene_level;
    uint32_t umi32;               // the record's UMI field is 4 bytes: UMI and record offset share one sort word
    uint32_t force_global_route;  // tests: neighbour search through the global-memory hash table for every cell
    const uint32_t* n_pug_dev;    // when set: the length of pug_cells lives on the device (cells the phase kernels of afq_pug2.hip handed over)
};

void launch_pug(hipStream_t s, const PugCellArgs& a, uint32_t n_blocks);

// ---- parsimony as phase kernels over UMI partitions (afq_pug2.hip) ----
struct P2Cell {
    uint64_t rd_base;     // the cell's first read slot: rd_h / rd_u / s_h / s_u / v_off / v_flag / lidx share the indexing
    uint64_t chunk_off;   // the cell's chunk in the input bytes (= meta[cell].chunk_off)
    uint32_t cell;        // index into meta[]
    uint32_t R;           // reads
    uint32_t lgP;         // log2 of the partition count (partition = low lgP bits of the UMI)
    uint32_t part_base;   // first partition of the cell in the per-partition arrays
    uint32_t n_ref;       // alignment words of the cell (= meta[cell].n_ref: sizes the column list and the label area)
    uint32_t tile0;       // the cell's first tile (its tiles are consecutive)
    uint64_t key_off;     // = meta[cell].key_off
};
constexpr uint32_t kP2MaxComp = 4096;   // 64 mask words, one per lane: the workgroup cover of k_p2_cover (afq_pug_common.h)
constexpr uint32_t kGDescWords = 20;    // per cell: what the graph phase hands the cover kernels (P2Args.gdesc)
// The range-wide graph build (afq_pugflat.hip): one block per range, filled on the device.  Offsets are u32 words into the pool.
struct PfDev {
    unsigned long long par, cnt, rk, umi, rc, tl, loff, tcell, lh;   // per vertex that has an edge [T]: root, component size -> first slot, position | size class, UMI, reads, slot inside the cell, label offset, its cell, label key
    unsigned long long prv, midoff, mrec, tied, slow, cmv;    // two-vertex components (two slots each), first record slot per listed component (+ the end), 32-byte records, the covers' set-aside lists
    uint32_t T, NP, NC, S;
    uint32_t n_old, pad[3];                        // cells routed to the per-cell graph kernel (old_list)
};
struct PfTile {   // per tile (k_pf_tiles): what the workgroups that take a tile's roots and components need of it, in one 64-byte load
    uint32_t j, live;               // its cell; 0: the cell is not the flat kernels' (handed back, or routed to the per-cell kernel)
    uint32_t tb, te;                // its vertices' dense numbers
    uint32_t comp_base, n_tiny;     // the cell's run of the component list, its 3..8 components
    uint32_t p0, n_pr;              // the tile's slice of the pair list (range-wide entries)
    uint32_t a0, na, b0, nb;        // ... of the cell's 3..8 list and of its 9..64 list (entries counted from the cell's first)
    uint32_t s_ti, s_mi;            // first record slots of the two slices
    uint32_t pad[2];
};
static_assert(sizeof(PfTile) == 64, "PfTile");
struct P2Args {
    const uint8_t* bytes; const CellMeta* meta; const P2Cell* cells; const uint2* tiles; const uint32_t* order;
    const uint64_t* rd_h; const uint64_t* rd_u;        // the decode's reads: label key, umi << 32 | record offset
    uint64_t* s_h; uint64_t* s_u;                      // reads grouped by partition, then the vertices (key; umi << 32 | word)
    uint32_t* v_off; uint8_t* v_flag; uint32_t* lidx;  // vertex: smallest record offset, has-an-edge flag, id among the touched
    uint32_t* pcnt; uint32_t* poff; uint32_t* pcur; uint32_t* pnv; uint32_t* pcell; uint32_t* pn3;   // per partition
    uint64_t* pairs; uint32_t* pnp;                    // vertex pairs with an edge: a partition's pairs sit at its own slots; per partition: how many
    uint64_t* cstage; uint32_t* pncls;                 // two-gene classes of lone vertices (em), staged the same way
    uint32_t* gcnt;                                    // per cell [4]: columns, label words, classes, error
    uint32_t* fb; uint32_t* fb_list; uint32_t* fb_count;   // cells handed to the one-workgroup kernel
    uint32_t* pool; unsigned long long* pool_cur; unsigned long long pool_cap;
    uint32_t* work_counter;
    uint32_t* work_counter2; uint32_t* gdesc;          // the cover kernel's cell counter; per cell [16]: what the graph kernel hands it (k_p2_graph -> k_p2_cover)
    const uint32_t* cell_nkeys; const uint32_t* t2g; uint64_t* keys0; uint32_t* cell_ncols; uint32_t* lab; uint32_t* lab_cnt;
    DevStatus* st;
    uint32_t* alt;                                     // [cells of the range] set to 1 when a component took the winner-take-all fallback
    uint32_t n_cells, n_tiles, n_parts, part_cap;
    uint32_t ref_count, num_genes, usa, num_rows, em, exact_umi, large_thresh, hw, umi_pairs;
    uint32_t tile;       // reads per tile of k_p2_hist / k_p2_scatter: 2048, 4096 or 8192
    uint32_t max_comp;   // vertices of the largest component the phase kernels cover themselves (kP2MaxComp; tests: AFQ_TEST_P2_MAX_COMP)
    uint32_t n_big;   // the first n_big cells of `order` (largest first: 15 000 reads or more) get a 1024-thread workgroup each in k_p2_graph / k_p2_cover / k_p2_tied
    uint32_t defer_min;   // a cell sets the tied components of its covers aside (k_p2_tied) when its listed components hold more vertices than this; 0xFFFFFFFF: than the graph kernel's LDS class table takes (tests: 0 = every cell)
    // the range-wide graph build: six per-tile quantities and their scans (tq: nta entries each; bq: per block of 1024 tiles, nba each),
    // per cell: routed to the per-cell graph kernel; the range's block; the list of routed cells
    uint32_t* tq; uint32_t* bq; uint32_t nta, nba; uint32_t* route; PfDev* pfd; uint32_t* old_list;
    PfTile* ptile;
    uint32_t* pcpre; uint32_t* pbq; uint32_t npa;      // the scan of the lone vertices' staged class counts over the partitions (npa entries, blocks of 1024)
    uint32_t graph_flat;  // the graph phase as range-wide kernels (afq_pugflat.hip); 0: the per-cell kernel for every cell (tests: AFQ_TEST_P2_GRAPH=cell)
    uint32_t lone_coop;   // k_p2_lone: a lone vertex whose label has 5..64 refs is resolved by its whole wave (0: by its lane alone, as until late in round 4 - tests, measurements)
};
#ifndef AFQ_P2_PART_TARGET
#define AFQ_P2_PART_TARGET 144
#endif
constexpr uint32_t kP2PartTarget = AFQ_P2_PART_TARGET;   // planned mean reads per partition (the partition count is a power of two: 72..144).  It was 160 until round 5: of a
                                          // sample's 11 000 cells the one or two whose mean sat just under 160 had a partition over the capacity below (245-260 reads in
                                          // the largest of 1024 partitions) and were handed back - and k_pug_cell costs 0.4-3 ms per launch whatever it is given
constexpr uint32_t kP2PartCap = 256;      // reads one partition may hold (one wave sorts it in registers)
constexpr uint32_t kP2TileHost = 4096;   // reads per histogram / scatter tile (P2Args.tile; k_p2_scatter<512>: 8 reads per thread)
void launch_p2_split(hipStream_t s, const P2Args& a);
void launch_p2_part(hipStream_t s, const P2Args& a);
void launch_p2_search(hipStream_t s, const P2Args& a);
void launch_p2_check(hipStream_t s, const P2Args& a);   // (afq_pugflat.hip; launch_p2_search ends with it)
void launch_p2_lone(hipStream_t s, const P2Args& a);
void launch_p2_graph(hipStream_t s, const P2Args& a, uint64_t n_reads);
void launch_pf_build(hipStream_t s, const P2Args& a, uint64_t n_reads);
void launch_pf_cover(hipStream_t s, const P2Args& a);
void launch_pf_resume(hipStream_t s, const P2Args& a);
uint32_t pug_max_blocks();
uint64_t pug_scratch_words(uint32_t nrec, uint32_t n_ref, bool gene_level);

#ifndef AFQ_SCATTER_TILE
#define AFQ_SCATTER_TILE 2048
#endif
constexpr uint32_t kScatterTileHost = AFQ_SCATTER_TILE;  // keys per histogram/scatter tile (measurement builds: 4096)

}  // namespace afq
