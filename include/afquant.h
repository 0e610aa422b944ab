/*
 * afquant.h — C ABI of the MI355X-native `alevin-fry quant` hot path.
 *
 * The reference (COMBINE-lab/alevin-fry v0.18.0, pure Rust) has no FFI layer:
 * the hot path is the body of the per-cell loop in `run_worker_thread`
 * (src/quant.rs:733-1322), configured by `WorkerConfig` (src/quant.rs:398-416).
 * This header is the seam a Rust host would bind instead of that loop body:
 * "collated chunk bytes in -> sparse row + flags out".  Every entry point
 * cites the reference item it replaces.  Plain pointers and sizes only.
 *
 * All functions return 0 on success or a negative AFQ_ERR_* code; the text of
 * the last error on a context is available from afq_last_error().  Nothing in
 * this library aborts the process (the reference panics / exit(1)s instead:
 * src/pugutils.rs:1148-1151, Cargo.toml:100-108).
 */
#ifndef AFQUANT_H
#define AFQUANT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFQ_ABI_VERSION 3   /* 3: afq_result.mmrate and afq_snappy_decode_device are gone, afq_em_resize_count and afq_mono_cell_count are new */

/* error codes */
#define AFQ_OK 0
#define AFQ_ERR_INVALID_ARG (-1)  /* null pointer, bad enum, inconsistent sizes  */
#define AFQ_ERR_BAD_INPUT (-2)    /* malformed chunk bytes / ref id out of range */
#define AFQ_ERR_UNSUPPORTED (-3)  /* valid request the device path cannot take   */
/* What AFQ_ERR_UNSUPPORTED stands for today (the reference has none of these limits, quant.rs:733-757; never approximated, the
 * whole batch is refused and afq_last_error names the cell):
 *   - UMIs over 22 nt, gene-id spaces over 2^20, more than two barcode levels;
 *   - parsimony: a cell of 2^22 reads or more; a cell of 2^20 .. 2^22 reads goes through the partition-parallel kernels and is
 *     fine as long as they can finish it - whatever its graph looks like under the default --large-graph-thresh: components of
 *     any size, pair lists of any length.  Only when they have to hand it to the one-workgroup kernel (a UMI partition over 256
 *     reads: UMIs that share their low bases, e.g. a constant primer tail; gene-level labels or an 8-byte UMI field send every
 *     cell there) that kernel's 2^20-read limit applies;
 *   - parsimony: a component over 4096 vertices under a --large-graph-thresh raised beyond that. */
#define AFQ_ERR_HIP (-4)          /* a HIP runtime call failed                   */
#define AFQ_ERR_NO_DEVICE (-5)    /* no usable gfx950 device                     */
#define AFQ_ERR_STATE (-6)        /* call sequence error (collect before submit) */
#define AFQ_ERR_OOM (-7)

/* ResolutionStrategy, src/quant.rs:82-114 (CLI spellings in the comments). */
typedef enum afq_resolution {
    AFQ_RES_TRIVIAL = 0,           /* trivial            */
    AFQ_RES_CR_LIKE = 1,           /* cr-like            */
    AFQ_RES_CR_LIKE_EM = 2,        /* cr-like-em         */
    AFQ_RES_PARSIMONY_EM = 3,      /* parsimony-em       */
    AFQ_RES_PARSIMONY = 4,         /* parsimony          */
    AFQ_RES_PARSIMONY_GENE_EM = 5, /* parsimony-gene-em  */
    AFQ_RES_PARSIMONY_GENE = 6     /* parsimony-gene     */
} afq_resolution;

/* SplicedAmbiguityModel (src/quant.rs, `--sa-model`). */
typedef enum afq_sa_model { AFQ_SA_WINNER_TAKE_ALL = 0, AFQ_SA_PREFER_AMBIG = 1 } afq_sa_model;

/* per-cell flag bits in afq_result.flags */
#define AFQ_CELL_TINY_PATH 0x1u /* used_fast_path,   src/quant.rs:794-797      */
#define AFQ_CELL_ALT_RES 0x2u   /* alt_resolution,   src/quant.rs:966, 1131     */
#define AFQ_CELL_EMPTY 0x4u     /* num_expr == 0,    src/quant.rs:1173          */

/*
 * Mirror of WorkerConfig (src/quant.rs:398-416) — the part of QuantOpts
 * (src/prog_opts.rs:24-44) that reaches the per-cell code — plus the record
 * field widths the reference gets from the RAD prelude (src/convert.rs:323-344).
 * POD; copied by afq_create.
 */
typedef struct afq_config {
    uint32_t abi_version;        /* AFQ_ABI_VERSION                                              */
    uint32_t resolution;         /* afq_resolution                                               */
    uint32_t sa_model;           /* afq_sa_model (forced to WTA when !usa_mode, quant.rs:1456)   */
    uint32_t usa_mode;           /* 3-column tg-map: spliced gid even, unspliced odd             */
    uint32_t num_genes;          /* gene-id space of tid_to_gid (USA: 2*G, spliced/unspliced)    */
    uint32_t num_rows;           /* output columns: num_genes, or 3*G in USA (quant.rs:1627-1645)*/
    uint32_t small_thresh;       /* tiny_cell_thresh; cells with nrec < this take the tiny path  */
    uint32_t large_graph_thresh; /* PUG component size above which cr-like is used (pugutils.rs:1055) */
    uint32_t pug_exact_umi;      /* 1: only identical UMIs induce PUG edges (--umi-edit-dist 0)  */
    uint32_t em_init_uniform;    /* EmInitType::Uniform instead of Informative (em.rs:37-41)     */
    uint32_t bc_bytes;           /* width of the barcode field in a record: 1,2,4,8              */
    uint32_t umi_bytes;          /* width of the UMI field in a record: 1,2,4,8                  */
    uint32_t profile;            /* 1: bracket every kernel with HIP events (afq_get_kernel_times)*/
    uint32_t umi_len;            /* UMI length in bases when the caller knows it (the RAD file tag `ulen`); 0 = unknown.
                                    Only bounds the 1-mismatch neighbour probes of the PUG (positions past the UMI
                                    cannot differ), so a value that is too SMALL would lose edges: pass 0 when unsure. */
    uint32_t dump_eq;            /* -d / dump_eq (quant.rs:1282-1307): also keep each cell's gene-level equivalence classes,
                                    handed out by afq_result_eqclasses(); -em resolutions only                  */
    uint32_t bc_split;           /* multi-barcode records (10x Flex: b0 = sample index, b1 = cell barcode) whose two barcode
                                    integers differ in width: bc_bytes = w0 + w1 (2..8) and bc_split = w0 (1..4, w1 = bc_bytes - w0
                                    in 1..4).  The batch is rewritten on the device with both as 4-byte fields and the reported
                                    barcode is b0 | b1 << 32.  0 = the barcode is one field of 1, 2, 4 or 8 bytes (equal-width
                                    pairs travel that way too: b0 | b1 << 8 w0).                                               */
    /* -b / --num-bootstraps with --summary-stat (src/quant.rs:1028-1038, em.rs:585-757; -em resolutions only, main.rs:713-724):
       per non-tiny cell, num_bootstraps times: the counts of its gene-level classes are redrawn from a multinomial over the
       observed counts and re-estimated by the EM from a random start; afq_result_bootstraps() hands out the per-column mean
       and variance.  The reference draws from an unseeded thread RNG - its numbers are not reproducible; here the draws
       are Philox4x32-10 keyed by boot_seed and the cell's index (first_cell_index + i), so a run is reproducible and does
       not depend on batching. */
    uint32_t num_bootstraps;
    uint32_t summary_stat;       /* 1: mean and E[x^2]-mean^2 (em.rs:673-683); 0: mean and the n-1 sample variance over the
                                    replicates (quant.rs:185-210).  Either way only the two summaries are produced, as in the reference */
    uint64_t boot_seed;
} afq_config;

typedef struct afq_ctx afq_ctx;

/*
 * One result batch = what run_worker_thread accumulates for a run of cells:
 * `expressed_ind/expressed_vec` per cell (src/quant.rs:1156-1168, 808-845) as
 * CSR, plus the per-cell scalars it logs.  Library-owned until
 * afq_result_release(); all pointers are host memory.
 */
typedef struct afq_result {
    uint64_t n_cells;
    uint64_t first_cell_index; /* cell_num of cell 0 (MetaChunk.first_chunk_index, quant.rs:734) */
    uint64_t nnz;
    const uint64_t* cell_ptr; /* [n_cells+1] offsets into gene/val                               */
    const uint32_t* gene;     /* [nnz] output column, ascending within a cell                     */
    const float* val;         /* [nnz] count > 0                                                  */
    const uint64_t* bc;       /* [n_cells] collate_key of the cell's first record (quant.rs:757)  */
    const uint32_t* nrec;     /* [n_cells] records in the chunk                                   */
    const uint8_t* flags;     /* [n_cells] AFQ_CELL_*                                             */
    /* (`trivial` also computes a multi-mapping rate in the reference, pugutils.rs:909 - stored at quant.rs:936 and never read
       or written anywhere: it is not part of the result) */
    void* opaque;
} afq_result;

/*
 * Replaces the worker set-up of do_quantify (src/quant.rs:1678-1765): builds a
 * per-device context holding the config and the transcript->gene table
 * (`tid_to_gid`, src/utils.rs:487-662; len = ref_count).  `device` is the HIP
 * device ordinal.  One context per device; contexts are independent.
 */
int afq_create(const afq_config* cfg, const uint32_t* tid_to_gid, uint32_t ref_count, int device,
               afq_ctx** out);
void afq_destroy(afq_ctx* ctx);

/*
 * Replaces handing a MetaChunk of cells to a worker (src/quant.rs:733-757).
 * `bytes` are collated-RAD chunks exactly as on disk, back to back or not:
 * chunk i starts at bytes[chunk_off[i]] with its 8-byte header
 * (nbytes:u32 incl. header, nrec:u32; src/convert.rs:473-481) followed by nrec
 * records `na:u32, bc, umi, na x u32(ori<<31|ref)` (src/convert.rs:124-144).
 * One chunk = one cell.  The caller keeps ownership; the bytes are copied to
 * the device before return.  Results are produced asynchronously on the
 * context's stream; afq_collect waits for them.
 */
int afq_submit(afq_ctx* ctx, const uint8_t* bytes, size_t n_bytes, const uint64_t* chunk_off,
               uint32_t n_cells, uint64_t first_cell_index);

/*
 * Same, for a host that would rather be asked for the bytes than hold them all in memory (the reference's producer reads
 * the collated file chunk by chunk, src/quant.rs:1773-1784): `read(user, offset, dst, len)` fills dst with bytes
 * [offset, offset + len) of the chunk stream and returns 0; it is called from several threads at once, with dst in pinned
 * staging memory, while earlier parts are already on their way to the device (a pread on a file descriptor is all it has
 * to be).  chunk_hdr = the 8-byte chunk headers (nbytes, nrec per cell), which the host has from walking the chunk table.
 */
typedef int (*afq_read_fn)(void* user, uint64_t offset, void* dst, size_t len);
int afq_submit_reader(afq_ctx* ctx, afq_read_fn read, void* user, size_t n_bytes, const uint64_t* chunk_off,
                      const uint32_t* chunk_hdr, uint32_t n_cells, uint64_t first_cell_index);

/*
 * Same, for chunk bytes already resident in this context's device memory
 * (what a host that overlaps H2D itself would use; bench.py times this form).
 * `d_bytes` must stay valid until afq_collect returns.  `chunk_off` is host
 * memory.
 */
int afq_submit_device(afq_ctx* ctx, const void* d_bytes, size_t n_bytes, const uint64_t* chunk_off,
                      uint32_t n_cells, uint64_t first_cell_index);

/* Waits for the submitted batch and returns its rows (src/quant.rs:1131-1179, 1266-1268). */
int afq_collect(afq_ctx* ctx, afq_result* out);
void afq_result_release(afq_result* res);

/*
 * cfg.dump_eq: what the reference copies out of `gene_eqc` per cell for -d (src/quant.rs:1282-1307) - the gene-level
 * equivalence classes left by the resolution, label = ascending gene ids of tid_to_gid's id space (USA: spliced 2k,
 * unspliced 2k+1), count = molecules.  Cells that took the tiny-cell path have none (they never touch gene_eqc).
 * The reference walks a hash map, so the order of a cell's classes is not defined there; here: single-gene (and USA
 * S+U of one gene) labels by output column, then the others in lexicographic order.  Owned by `res`.
 */
typedef struct afq_eqclasses {
    uint64_t n_cells, n_classes, n_words;
    const uint64_t* cell_ptr;  /* [n_cells+1]   classes of cell i: cell_ptr[i] .. cell_ptr[i+1]     */
    const uint64_t* label_ptr; /* [n_classes+1] label of class k: labels[label_ptr[k] .. label_ptr[k+1]) */
    const uint32_t* labels;    /* [n_words]                                                       */
    const uint32_t* count;     /* [n_classes]                                                     */
} afq_eqclasses;
int afq_result_eqclasses(const afq_result* res, afq_eqclasses* out);

/*
 * cfg.num_bootstraps > 0: what BootstrapHelper::record_cell[_from_replicates] collects (src/quant.rs:157-210): per cell
 * the non-zero bootstrap means and variances, as two CSR matrices over the same cells.  Columns are the alpha indices the
 * reference's run_bootstrap uses - the gene ids of gene_eqc's labels (in USA mode that is the spliced/unspliced id 2k /
 * 2k+1, NOT the S/U/A output column: run_bootstrap_with_scratch is handed gene_eqc as is, quant.rs:1028-1038).
 * Tiny-path cells have no bootstraps.  Owned by `res`.
 */
typedef struct afq_bootstraps {
    uint64_t n_cells;
    const uint64_t* mean_ptr; const uint32_t* mean_col; const float* mean_val;   /* [n_cells+1], [nnz], [nnz] */
    const uint64_t* var_ptr;  const uint32_t* var_col;  const float* var_val;
} afq_bootstraps;
int afq_result_bootstraps(const afq_result* res, afq_bootstraps* out);

/*
 * The per-cell work of `alevin-fry infer` (src/infer.rs:31-426): one EM (em_optimize_subset, EmInitType::Informative,
 * USA offsets (num_alphas/3, 2*num_alphas/3) when usa_mode) per row of an equivalence-class count matrix.
 *   eq_labels / eq_label_ptr : the global classes (IndexedEqList::init_from_eqc_file): class e = eq_labels[eq_label_ptr[e] ..
 *                              eq_label_ptr[e+1]), output columns (a gene_eqclass.txt.gz as `quant -d` writes it)
 *   cell_ptr / cell_eq / cell_count : the count matrix in CSR, class ids ascending within a row (what sprs' to_csr() gives),
 *                              counts already rounded to integers (infer.rs:389)
 * `out` gets the non-zero abundances per cell (gene/val ascending by column; bc, nrec, flags are 0).  The context's own
 * resolution settings play no part.  Release with afq_result_release().
 */
int afq_infer(afq_ctx* ctx, const uint32_t* eq_labels, const uint64_t* eq_label_ptr, uint32_t n_eq, const uint64_t* cell_ptr,
              const uint32_t* cell_eq, const uint32_t* cell_count, uint32_t n_cells, uint32_t num_alphas, uint32_t usa_mode,
              afq_result* out);

/*
 * Per-cell fragment de-duplication of `alevin-fry atac deduplicate`
 * (src/atac/deduplicate.rs:199-237, HitInfo order src/atac/sort.rs:37-64).
 * Input per cell: the (ref, start, frag_len) of every record that has exactly
 * one alignment of map_type 4; output: distinct fragments in
 * (ref,start,frag_len) order with their multiplicity (u16, as in the reference).
 * `cell_ptr` has n_cells+1 entries.  Output arrays are malloc'd; free with
 * afq_free().
 */
int afq_atac_dedup(afq_ctx* ctx, const uint32_t* ref, const uint32_t* start,
                   const uint16_t* frag_len, const uint64_t* cell_ptr, uint32_t n_cells,
                   uint64_t** out_cell_ptr, uint32_t** out_ref, uint32_t** out_start,
                   uint16_t** out_frag_len, uint16_t** out_count);
void afq_free(void* p);

/*
 * The same, from the collated-RAD chunks themselves: the record loop of deduplicate.rs:199-218 (walk the
 * AtacSeqReadRecords `na:u32, bc, na x {ref:u32, type:u8, start_pos:u32, frag_len:u16}` - tests/atac_integration.rs:110-121 -
 * keep those with exactly one alignment of map_type 4, count the rest) runs on the device in front of the sort.
 * `bytes` / `chunk_off` as for afq_submit (bytes_on_device != 0: `bytes` is device memory of this context's device).
 * Outputs as afq_atac_dedup, plus the barcode of every cell; free each with afq_free().
 */
typedef struct afq_atac_stats {
    uint64_t n_records;          /* records read                                                                  */
    uint64_t n_multimapped;      /* "records with greater than 1 mapping", deduplicate.rs:210-212                 */
    uint64_t n_not_mapped_pair;  /* neither kept nor multi-mapped (no alignment, or one that is not type 4), :213-215 */
    uint64_t n_distinct;         /* distinct fragments over all cells                                             */
    uint64_t n_deduplicated;     /* distinct fragments seen more than once, :222-224                              */
    uint64_t n_long_fragments;   /* distinct fragments with frag_len >= 2000 (left out of the BED, :47-63)        */
    uint64_t n_fallback_cells;   /* cells the walk-free parse could not prove and walked record by record         */
} afq_atac_stats;
int afq_atac_dedup_rad(afq_ctx* ctx, const uint8_t* bytes, size_t n_bytes, const uint64_t* chunk_off, uint32_t n_cells,
                       uint32_t bc_bytes, int bytes_on_device, uint64_t** out_cell_ptr, uint64_t** out_bc, uint32_t** out_ref,
                       uint32_t** out_start, uint16_t** out_frag_len, uint16_t** out_count, afq_atac_stats* stats);

/*
 * Kernel timing of the last collected batch (HIP events on the context's own
 * stream; only when cfg.profile != 0).  Fills up to `cap` entries; returns the
 * number of kernels, or a negative error.  `name[i]` points to static storage.
 */
typedef struct afq_kernel_time {
    const char* name;
    double ms;       /* summed over launches   */
    uint32_t launches;
    uint32_t pad;
} afq_kernel_time;
int afq_get_kernel_times(afq_ctx* ctx, afq_kernel_time* out, uint32_t cap);

/* Number of records / keys the last collected batch processed (for the bench's byte model). */
typedef struct afq_batch_stats {
    uint64_t n_records;
    uint64_t n_ref_words;  /* sum of na                                            */
    uint64_t n_keys;       /* (umi,gene) pairs after per-read gene dedup           */
    uint64_t n_buckets;
    uint64_t n_overflow_buckets;
    uint64_t input_bytes;
    uint64_t n_fallback_cells; /* cells the walk-free decode could not prove and re-decoded sequentially */
} afq_batch_stats;
int afq_get_batch_stats(afq_ctx* ctx, afq_batch_stats* out);
/* Parsimony: labels of three or more ids are keyed by a 62-bit hash; two different labels of one cell under one key are
   detected on the device, and the range of cells is then decoded again with another hash function instead of being refused
   (the reference keys its map by the list itself, eq_class.rs:859-903).  How often that happened since afq_create. */
uint64_t afq_label_rehash_count(const afq_ctx* ctx);
/* Parsimony: the per-cell graphs are built in a pool sized by the range's reads (32 words per read).  A cell whose graph
   outgrows it (short UMIs: hundreds of reads per UMI, so a vertex has many neighbours) makes the library run the range
   again with four times the pool, up to three times, before AFQ_ERR_OOM (the reference allocates per graph,
   pugutils.rs:65-267).  How often that happened since afq_create. */
uint64_t afq_pool_regrow_count(const afq_ctx* ctx);
/* EM resolutions: ranges whose EM did not fit the device scratch set aside for it ahead of time and was sized on the host
 * instead (one extra trip to the host for that range; results identical).  Diagnostics only. */
uint64_t afq_em_resize_count(const afq_ctx* ctx);
/* Parsimony: cells resolved by the one-workgroup kernel instead of the partition-parallel phase kernels - sent there directly
 * (gene-level parsimony, 8-byte UMIs) or handed back (a UMI partition of more than 256 reads; a component of more than 4096
 * vertices under a --large-graph-thresh raised beyond that).  Results identical; diagnostics only. */
uint64_t afq_mono_cell_count(const afq_ctx* ctx);

/* Brings the HIP runtime up on `device` (first-call initialisation) - a host can call it from a side thread while it parses its
   inputs.  Returns 0 or AFQ_ERR_NO_DEVICE. */
int afq_device_warmup(int device);
/* PCI bus id of `device` ("0000:c1:00.0"), for a host that wants to place its threads and buffers on the device's NUMA node
   (/sys/bus/pci/devices/<id>/numa_node).  Returns 0, AFQ_ERR_NO_DEVICE or AFQ_ERR_INVALID_ARG (out too small). */
int afq_device_pci_bus_id(int device, char* out, size_t out_len);

const char* afq_last_error(const afq_ctx* ctx); /* ctx may be NULL: last create error */
int afq_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* AFQUANT_H */
