/*
 * afquant_synth.h — seeded synthetic collated-RAD generator, on the host and on the device.
 * Bench/test tooling shipped in libafquant.so; not part of the quant boundary.
 *
 * Model = SURVEY.md §8(d): log-normal cell sizes sorted descending (collate order,
 * src/collate.rs:272-274 of the reference), u32 barcode / u32 UMI records
 * (10x v3), reads drawn from a per-cell molecule pool (PCR duplicates), gene
 * popularity Zipf(s) over num_genes genes, ref_count transcripts, na in {1,2,3},
 * half of the multi-ref reads crossing genes, 1-mismatch UMI errors, refs
 * ascending and duplicate-free.
 *
 * Every draw is a word of a Philox4x32-10 block keyed by the seed and counted by
 * (read or molecule, GLOBAL cell index, stream); probabilities are 32-bit integer
 * thresholds.  The host functions and the gfx950 kernels therefore write the same
 * bytes, and a rank can generate any range [first_cell, first_cell+n) of the data
 * set by itself - configs[3] (10^6 cells x 2*10^4 reads, ~350 GB of RAD) is made
 * shard by shard in HBM and never exists as a file.
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
    uint32_t n_cells;    /* cells of the WHOLE data set (the size list is drawn and sorted over all of them) */
    uint32_t min_reads;
    double median_reads; /* log-normal median of reads per cell       */
    double sigma;        /* log-normal sigma (0: every cell = median) */
    uint32_t num_genes;  /* G                                          */
    uint32_t txp_per_gene; /* spliced transcripts per gene when ref_count == 0 */
    uint32_t usa;        /* splici-like: the spliced txps then G unspliced; gids 2g / 2g+1 */
    uint32_t umi_len;    /* nt, <= 16                                  */
    double dup;          /* fraction of reads that are PCR duplicates  */
    double p_na2, p_na3; /* P(na=2), P(na=3); P(na=1) = rest           */
    double cross;        /* P(an extra ref is on another gene)         */
    double umi_err;      /* P(read carries a 1-base UMI error)         */
    double zipf;         /* gene popularity P(rank k) ~ k^-zipf (SURVEY: 1.1); 0 = uniform */
    double pow_skew;     /* > 0: popularity of gene floor(G * x^pow_skew), x uniform, instead (16 = the round-1 stress
                            variant: half of all molecules on gene 0)                                               */
    double p_unspliced, p_both; /* USA only                            */
    uint32_t n_threads;  /* host generator                             */
    uint32_t ref_count;  /* spliced transcripts in total (SURVEY config 2: 199 138); 0 = num_genes * txp_per_gene:
                            gene g owns ref_count / G transcripts, the first ref_count % G genes one more        */
    double tail;         /* label-length tail: P(one more ref) of a geometric run of further refs on the gene's family
                            (0 = off: at most three refs per record, the plain model); 0.6 gives E[na] ~ 3        */
    uint32_t tail_max;   /* refs per record at most under the tail model (<= 64; 0 = 64)                          */
    uint32_t family;     /* genes per family: the tail's refs sit on genes of the read's block of `family` gene ids (0 = 8) */
} afq_synth_params;

/* ref_count (USA: + G unspliced) / gene-id space / output columns implied by the params */
void afq_synth_dims(const afq_synth_params* p, uint32_t* ref_count, uint32_t* num_genes, uint32_t* num_rows);
/* fills tid_to_gid[ref_count] */
void afq_synth_t2g(const afq_synth_params* p, uint32_t* tid_to_gid);
/* reads per cell of the whole data set, descending: cell_nrec[p->n_cells] */
int afq_synth_cell_sizes(const afq_synth_params* p, uint32_t* cell_nrec);

/*
 * Host generator for the cells [first_cell, first_cell + n) (cell_nrec = THEIR sizes):
 * pass 1 gives the chunk offsets, pass 2 writes the chunks at out + chunk_off[i].
 */
int afq_synth_host_plan(const afq_synth_params* p, uint64_t first_cell, uint32_t n, const uint32_t* cell_nrec,
                        uint64_t* chunk_off, uint64_t* total_bytes);
int afq_synth_host_fill(const afq_synth_params* p, uint64_t first_cell, uint32_t n, const uint32_t* cell_nrec,
                        const uint64_t* chunk_off, uint8_t* out, uint64_t total_bytes);

/*
 * Device generator: the same bytes, produced in this device's memory (a size pass, a prefix on the host,
 * a fill pass).  *d_bytes is allocated by the library (total_bytes + 16, zero-padded) and released with
 * afq_synth_device_free; it can be handed to afq_submit_device as it is.
 */
int afq_synth_device_generate(const afq_synth_params* p, int device, uint64_t first_cell, uint32_t n,
                              const uint32_t* cell_nrec, uint64_t* chunk_off, uint64_t* total_bytes, void** d_bytes);
void afq_synth_device_free(int device, void* d_bytes);
/* copy n bytes of a device buffer to the host (tests, the CPU leg of the bench) */
int afq_synth_device_read(int device, const void* d_src, uint64_t n, void* host_dst);

/*
 * scATAC (configs[4]): the chunks of a collated scATAC RAD file (records `na:u32, bc:u32,
 * na x {ref:u32, type:u8, start_pos:u32, frag_len:u16}`), host generator.  Call with out = NULL for the chunk offsets
 * and the total size, then again with a buffer of that size.
 */
typedef struct afq_synth_atac_params {
    uint64_t seed;
    uint32_t n_cells, frags_per_cell;
    uint32_t n_refs, ref_len;          /* chromosomes, and the length positions are drawn below        */
    double p_dup, p_multi, p_unmapped; /* exact duplicate of the previous fragment / two alignments / none */
    double flen_mu, flen_sigma;        /* log-normal fragment length, clipped to [30, 2500]             */
    uint32_t n_threads, reserved;
} afq_synth_atac_params;
int afq_synth_atac(const afq_synth_atac_params* p, uint64_t* chunk_off, uint64_t* total_bytes, uint8_t* out);

/* Whole data set on the host, as in round 1: sizes + offsets, then the bytes. */
int afq_synth_plan(const afq_synth_params* p, uint32_t* cell_nrec, uint64_t* chunk_off, uint64_t* total_bytes,
                   uint64_t* total_reads);
int afq_synth_fill(const afq_synth_params* p, const uint32_t* cell_nrec, const uint64_t* chunk_off, uint8_t* out,
                   uint64_t total_bytes);

#ifdef __cplusplus
}
#endif
#endif
