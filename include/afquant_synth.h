/*
 * afquant_synth.h — seeded synthetic collated-RAD generator (host, multi-threaded).
 * Bench/test tooling shipped in libafquant.so; not part of the quant boundary.
 * Model = SURVEY.md §8(d): log-normal cell sizes sorted descending (collate order,
 * src/collate.rs:272-274 of the reference), u32 barcode / u32 UMI records
 * (10x v3), reads drawn from a per-cell molecule pool (PCR duplicates), na in
 * {1,2,3}, half of the multi-ref reads crossing genes, 1-mismatch UMI errors,
 * refs ascending and duplicate-free.
 */
#ifndef AFQUANT_SYNTH_H
#define AFQUANT_SYNTH_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct afq_synth_params {
    uint64_t seed;
    uint32_t n_cells;
    uint32_t min_reads;
    double median_reads; /* log-normal median of reads per cell       */
    double sigma;        /* log-normal sigma (0: every cell = median) */
    uint32_t num_genes;  /* G                                          */
    uint32_t txp_per_gene;
    uint32_t usa;        /* splici-like: G*tpg spliced txps then G unspliced; gids 2g / 2g+1 */
    uint32_t umi_len;    /* nt, <= 16                                  */
    double dup;          /* fraction of reads that are PCR duplicates  */
    double p_na2, p_na3; /* P(na=2), P(na=3); P(na=1) = rest           */
    double cross;        /* P(an extra ref is on another gene)         */
    double umi_err;      /* P(read carries a 1-base UMI error)         */
    double zipf;         /* gene popularity skew (0 = uniform)         */
    double p_unspliced, p_both; /* USA only                            */
    uint32_t n_threads;
    uint32_t reserved;
} afq_synth_params;

/* ref_count / gene-id space / output columns implied by the params */
void afq_synth_dims(const afq_synth_params* p, uint32_t* ref_count, uint32_t* num_genes, uint32_t* num_rows);
/* fills tid_to_gid[ref_count] */
void afq_synth_t2g(const afq_synth_params* p, uint32_t* tid_to_gid);
/* pass 1: cell_nrec[n_cells] (descending), chunk_off[n_cells], totals */
int afq_synth_plan(const afq_synth_params* p, uint32_t* cell_nrec, uint64_t* chunk_off, uint64_t* total_bytes,
                   uint64_t* total_reads);
/* pass 2: writes the chunks at out + chunk_off[i]; out must hold total_bytes */
int afq_synth_fill(const afq_synth_params* p, const uint32_t* cell_nrec, const uint64_t* chunk_off, uint8_t* out,
                   uint64_t total_bytes);

#ifdef __cplusplus
}
#endif
#endif
