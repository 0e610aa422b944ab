/*
 * afquant_host.h — the host side around the quant hot path, in C++ because the
 * reference's host is compiled code and this image has no Rust toolchain.  It mirrors what
 * `alevin_fry::quant::quantify(QuantOpts)` does around the per-cell loop (src/quant.rs:359-396,
 * 1327-1951 of the reference): read `collate.json`, open `map.collated.rad[.sz]`, parse the RAD
 * prelude and the transcript-to-gene map, feed the collated chunks to the device through afquant.h,
 * and write `alevin/quants_mat.{mtx,_rows.txt,_cols.txt}`, `featureDump.txt` and `quant.json`.
 */
#ifndef AFQUANT_HOST_H
#define AFQUANT_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

struct afq_atac_stats;

/* QuantOpts (src/prog_opts.rs:24-44) — the CLI-visible options of `alevin-fry quant` (src/main.rs:294-348). */
typedef struct afq_quant_opts {
    const char* input_dir;      /* -i : directory holding map.collated.rad[.sz], collate.json, generate_permit_list.json */
    const char* tg_map;         /* -m : 2- or 3-column transcript-to-gene TSV                                             */
    const char* output_dir;     /* -o                                                                                      */
    const char* resolution;     /* -r : trivial | cr-like | cr-like-em | parsimony | parsimony-em | parsimony-gene | parsimony-gene-em */
    const char* filter_list;    /* --quant-subset : file of barcodes to quantify, or NULL                                  */
    const char* cmdline;        /* recorded in quant.json                                                                  */
    uint32_t num_threads;       /* -t : accepted for compatibility; the device does the per-cell work                      */
    uint32_t small_thresh;      /* --small-thresh (default 100)                                                            */
    int32_t umi_edit_dist;      /* --umi-edit-dist : -1 = default by resolution (main.rs:652-703)                          */
    int32_t large_graph_thresh; /* --large-graph-thresh : -1 = default by resolution (main.rs:332-341)                     */
    uint32_t init_uniform;      /* --init-uniform                                                                          */
    uint32_t dump_eq;           /* -d : also write geqc_counts.mtx + gene_eqclass.txt.gz (quant.rs:229-355)                */
    uint32_t num_bootstraps;    /* -b : bootstrap replicates; writes bootstraps_mean.mtx + bootstraps_var.mtx (quant.rs:1850-1877) */
    uint32_t device;            /* HIP device ordinal                                                                      */
    uint64_t batch_bytes;       /* chunk bytes handed to a device per afq_submit (0 = 16 GiB; a batch is pipelined internally) */
    uint32_t sa_model;          /* --sa-model (hidden): afq_sa_model; ignored with a log line outside USA mode (quant.rs:1456) */
    uint32_t summary_stat;      /* --summary-stat (requires -b)                                                            */
    uint64_t boot_seed;         /* seed of the bootstrap draws (the reference's are unseeded); --boot-seed, default 0     */
    const int32_t* devices;     /* --devices 0,1,... : HIP device ordinals to spread the cells over (NULL / 0 = just `device`).  The
                                   worker fan-out of do_quantify (quant.rs:1553-1575, 1678-1765): one context and one host thread
                                   per device over a contiguous, byte-balanced range of cells; rows are gathered in cell order, so
                                   the output does not depend on the device count                                          */
    uint32_t n_devices;
    uint32_t reserved;
} afq_quant_opts;

/* The CLI-visible options of `alevin-fry infer` (src/main.rs:350-365). */
typedef struct afq_infer_opts {
    const char* count_mat;      /* -c : cells x equivalence classes, MatrixMarket (`geqc_counts.mtx` of `quant -d`); the barcode and
                                        gene-name files are looked up next to it (quants_mat_rows.txt, quants_mat_cols.txt)        */
    const char* eq_labels;      /* -e : gene_eqclass.txt.gz                                                                */
    const char* output_dir;     /* -o                                                                                      */
    const char* filter_list;    /* --quant-subset, or NULL                                                                 */
    uint32_t usa_mode;          /* --usa                                                                                   */
    uint32_t num_threads;       /* -t : threads of the MTX writer                                                          */
    uint32_t device;
    uint32_t reserved;
} afq_infer_opts;
/* Runs the whole `infer` sub-command (src/infer.rs:31-426): quants_mat.mtx, quants_mat_rows.txt, quants_mat_cols.txt in output_dir. */
int afq_infer_files(const afq_infer_opts* opts);

/* The options of `alevin-fry atac deduplicate` (src/atac/prog_opts.rs:47-55, src/main.rs:942-956). */
typedef struct afq_atac_dedup_opts {
    const char* input_dir;      /* -i : directory holding collate.json, generate_permit_list.json, map.collated.rad[.sz]; map.bed is written here */
    uint32_t num_threads;       /* -t : BED formatting / snappy threads (0 = all cores)                                   */
    uint32_t rev;               /* -d/--permit-bc-ori rc (the CLI default) : barcodes are written reverse-complemented    */
    uint32_t device;
    uint32_t reserved;
    struct afq_atac_stats* stats_out; /* optional: the counters the reference logs (deduplicate.rs:285-308)               */
} afq_atac_dedup_opts;
/* Runs the whole `atac deduplicate` sub-command (src/atac/deduplicate.rs:68-309) on the device: <input_dir>/map.bed. */
int afq_atac_deduplicate(const afq_atac_dedup_opts* opts);

/* Runs the whole `quant` sub-command.  Returns 0 or a negative AFQ_ERR_* code; message via afq_host_last_error(). */
int afq_quantify(const afq_quant_opts* opts);
const char* afq_host_last_error(void);

/* ---- pieces exposed for the CPU tests (no GPU needed) ---- */
/* Rust `{}` formatting of an f32 (shortest round-trip digits, never an exponent; "NaN", "inf"). Returns length. */
int afq_format_f32(float v, char* buf, size_t cap);
/* Snappy *frame format* decode (what `snap::read::FrameDecoder` undoes, src/quant.rs:376). out may be NULL to size. */
int64_t afq_snappy_frame_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap);
/* Parse a RAD prelude; fills the scalars, returns the byte offset of the first chunk or a negative error. */
typedef struct afq_rad_info {
    uint64_t ref_count, num_chunks, first_chunk_off;
    uint32_t is_paired, cblen, ulen, bc_bytes, umi_bytes;
} afq_rad_info;
int afq_rad_parse_prelude(const uint8_t* bytes, size_t n, afq_rad_info* out);

#ifdef __cplusplus
}
#endif
#endif
