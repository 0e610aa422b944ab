// afq_common.h — layouts shared by the host planner (afq_api.cpp) and the gfx950
// kernels (afq_kernels.hip).  Product code; nothing here touches oracle/.
#pragma once
#include <stdint.h>

namespace afq {

// ---- key layout -----------------------------------------------------------
// One key per (read, distinct gene of the read): umi << 20 | gene.  Sorting keys
// ascending is the (umi, gene) order the reference sorts its triplets in
// (src/pugutils.rs:652).  20 gene bits cover USA gene-id spaces up to 2^20;
// 44 UMI bits cover UMIs up to 22 nt.  Wider inputs are refused, not truncated.
constexpr int kGeneBits = 20;
constexpr uint32_t kGeneMask = (1u << kGeneBits) - 1u;
constexpr int kUmiBits = 64 - kGeneBits;
constexpr uint64_t kKeySentinel = ~0ull;

// ---- bucket geometry ------------------------------------------------------
// A cell's keys are split into 2^lg_nb buckets by a multiplicative hash of the
// UMI (all keys of one UMI share a bucket, which is all cr-like needs); one
// workgroup sorts and resolves one bucket in LDS.
constexpr uint32_t kBucketCap = 512;     // keys the one-wave resolve workgroup holds in LDS
constexpr uint32_t kBucketTarget = 256;  // planned mean keys per bucket
constexpr uint32_t kMaxLgNb = 20;
constexpr uint64_t kHashMul = 0x9E3779B97F4A7C15ull;

// Bucket of a UMI inside its cell: a 32-bit multiplicative hash of the UMI folded to one word (UMIs are <= 44 bits; a
// 64 x 64-bit product is three quarter-rate multiplies on the vector unit, this is one).  Only the device kernels that
// place and resolve keys use it, so it can be any function of the UMI alone.
__host__ __device__ inline uint32_t bucket_of(uint64_t umi, uint32_t lg_nb) {
    const uint32_t lo = (uint32_t)umi, hi = (uint32_t)(umi >> 32);
    const uint32_t h = (lo ^ (hi << 19) ^ hi) * 0x9E3779B1u;
    return lg_nb == 0 ? 0u : h >> (32 - lg_nb);
}

// Per-cell plan computed on the host from the chunk header.
struct CellMeta {
    uint64_t chunk_off;    // byte offset of the chunk header in the input
    uint64_t key_off;      // first slot of this cell in keys0/keys1 (capacity n_ref)
    uint32_t nbytes;       // chunk size incl. 8-byte header
    uint32_t nrec;
    uint32_t n_ref;        // sum of na = key capacity
    uint32_t bucket_base;  // first global bucket id of the cell
    uint32_t lg_nb;        // log2(#buckets)
    uint32_t mode;         // kModeCrLike / kModeTrivial: how this cell is resolved (tiny cells are always cr-like)
    uint32_t slab_cap;     // multi-bucket cells placed without a counting pass: every bucket owns slab_cap slots of keys1 (0: exact layout)
    uint32_t tile_base;    // multi-bucket cells: the cell's first scatter tile (k_fill_tables writes the range's tile table from it)
    uint64_t k1_off;       // ... starting here (the cell's region holds max(nb * slab_cap, n_ref + 1) slots)
};

// device-side error / statistics block
struct DevStatus {
    uint32_t err_code;       // first error (0 = none)
    uint32_t err_cell;
    uint32_t n_overflow;     // buckets larger than kBucketCap
    uint32_t n_fallback;     // cells re-decoded by the sequential walk
    unsigned long long n_keys;
};

// Per-cell verification block of the walk-free decode (k_decode_par): the candidate
// record starts it used are exactly the sequential parse iff ok==0, count==nrec and
// words==nbytes/4-2 (proof in DESIGN.md "walk-free decode").
struct CellChk {
    uint32_t count;  // candidate records decoded
    uint32_t words;  // sum of their sizes in dwords
    uint32_t fail;   // a local check failed
    uint32_t pad;
};

constexpr uint32_t kModeCrLike = 0;   // winner-take-all (cr-like; every tiny cell, src/quant.rs:794-845)
constexpr uint32_t kModeCrLikeEm = 2; // cr-like-em: ties are kept as gene-level classes and resolved by the EM (quant.rs:882-924)
constexpr uint32_t kModePug = 3;       // parsimony / parsimony-em: PUG + monochromatic cover (pugutils.rs:65-391, 989-1331)
constexpr uint32_t kModePugEm = 4;
constexpr uint32_t kModePugGene = 5;   // parsimony-gene / parsimony-gene-em: gene-level EqMap (eq_class.rs:723-821)
constexpr uint32_t kModePugGeneEm = 6;
constexpr uint32_t kModeTrivial = 1;  // `trivial`: single-gene reads only, distinct UMIs per gene (src/pugutils.rs:852-911)

constexpr uint32_t kSlabWords = 256;  // dwords one wave of k_decode_par covers (1 KiB)

constexpr uint32_t kErrRecordWalk = 1;   // records do not tile the chunk
constexpr uint32_t kErrRefRange = 2;     // ref id >= ref_count
constexpr uint32_t kErrUmiWide = 3;      // UMI needs more than kUmiBits bits
constexpr uint32_t kErrGeneRange = 4;    // gene id >= 2^kGeneBits or >= num_genes
constexpr uint32_t kErrSlotRange = 5;    // resolved slot >= num_rows

struct OverflowEnt { uint32_t bucket; uint32_t n; };

constexpr uint32_t kErrLabelHash = 6;    // two different ref lists with the same 64-bit label hash
constexpr uint32_t kErrPugLimit = 7;     // a PUG size limit of the device path was exceeded
constexpr uint32_t kErrPugPool = 8;      // edge pool exhausted
constexpr uint32_t kErrInternal = 9;     // a consistency check of the device code failed (a bug, never the input)

__host__ __device__ inline bool mode_is_pug(uint32_t m) { return m >= kModePug && m <= kModePugGeneEm; }
__host__ __device__ inline bool mode_pug_gene(uint32_t m) { return m == kModePugGene || m == kModePugGeneEm; }
// order-independent hash of a set of gene ids (gene-level class labels)
__host__ __device__ inline uint64_t gene_set_hash_term(uint32_t g) {
    uint64_t x = (uint64_t)g + 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Per-read output of the decode for PUG cells: 64-bit hash of the read's ref-list label, its UMI, and the
// dword offset of the record inside its chunk (= appearance order, and where the label lives).
struct PugOut {
    uint64_t* h;
    uint64_t* u;
    uint32_t* o;
    const uint64_t* rd_off;  // [n_cells] first read slot of a PUG cell
    // A label of three or more ids is keyed by a 62-bit hash and the kernels check that equal keys are equal labels; when that
    // check fails (kErrLabelHash) the range is decoded again under another salt - another hash function - instead of being
    // refused.  mask (all ones; tests: a few bits, so that the first salt collides for certain) is applied to the hash.
    uint64_t salt, mask;
};
__host__ __device__ inline uint64_t label_hash_init(uint32_t na) { return 0x9E3779B97F4A7C15ull ^ na; }
// The 64-bit class key of a label.  Labels of one or two ids (transcripts at txp level, genes at gene level; ids < 2^31)
// are carried IN the key - tag 1: the id; tag 2: (smaller id, larger id) - so equal keys mean equal labels by
// construction and nothing needs to be re-read to verify them; an empty label is key 0.  Longer labels keep a
// 62-bit hash under tag 3 and are verified against the record (kErrLabelHash on a collision).
__host__ __device__ inline uint64_t label_key(uint64_t hash, uint32_t n, uint32_t a, uint32_t b) {
    if (n == 0) return 0;
    if (n == 1) return (1ull << 62) | a;
    if (n == 2) return (2ull << 62) | ((uint64_t)(a < b ? a : b) << 31) | (a < b ? b : a);
    return (3ull << 62) | (hash >> 2);
}
__host__ __device__ inline bool label_key_is_exact(uint64_t key) { return (key >> 62) != 3; }
__host__ __device__ inline uint64_t label_hash_step(uint64_t h, uint32_t t) {
    h = (h ^ t) * 0xBF58476D1CE4E5B9ull;
    return h ^ (h >> 29);
}

}  // namespace afq
