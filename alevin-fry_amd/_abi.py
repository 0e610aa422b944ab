"""ctypes view of include/afquant.h (struct layouts and prototypes).

Kept separate from the loader so the CPU oracle binding (oracle/oracle.py, test
infrastructure) can share the struct definitions without loading the HIP library.
"""
import ctypes as C

AFQ_ABI_VERSION = 3

AFQ_OK = 0
AFQ_ERR_INVALID_ARG = -1
AFQ_ERR_BAD_INPUT = -2
AFQ_ERR_UNSUPPORTED = -3
AFQ_ERR_HIP = -4
AFQ_ERR_NO_DEVICE = -5
AFQ_ERR_STATE = -6
AFQ_ERR_OOM = -7

# ResolutionStrategy spellings of the CLI (src/quant.rs:98-111 of the reference)
RESOLUTIONS = {
    "trivial": 0,
    "cr-like": 1,
    "cr-like-em": 2,
    "parsimony-em": 3,
    "parsimony": 4,
    "parsimony-gene-em": 5,
    "parsimony-gene": 6,
}

CELL_TINY_PATH = 0x1
CELL_ALT_RES = 0x2
CELL_EMPTY = 0x4


class AfqConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("resolution", C.c_uint32),
        ("sa_model", C.c_uint32),
        ("usa_mode", C.c_uint32),
        ("num_genes", C.c_uint32),
        ("num_rows", C.c_uint32),
        ("small_thresh", C.c_uint32),
        ("large_graph_thresh", C.c_uint32),
        ("pug_exact_umi", C.c_uint32),
        ("em_init_uniform", C.c_uint32),
        ("bc_bytes", C.c_uint32),
        ("umi_bytes", C.c_uint32),
        ("profile", C.c_uint32),
        ("umi_len", C.c_uint32),
        ("dump_eq", C.c_uint32),
        ("bc_split", C.c_uint32),
        ("num_bootstraps", C.c_uint32),
        ("summary_stat", C.c_uint32),
        ("boot_seed", C.c_uint64),
    ]


class AfqResult(C.Structure):
    _fields_ = [
        ("n_cells", C.c_uint64),
        ("first_cell_index", C.c_uint64),
        ("nnz", C.c_uint64),
        ("cell_ptr", C.POINTER(C.c_uint64)),
        ("gene", C.POINTER(C.c_uint32)),
        ("val", C.POINTER(C.c_float)),
        ("bc", C.POINTER(C.c_uint64)),
        ("nrec", C.POINTER(C.c_uint32)),
        ("flags", C.POINTER(C.c_uint8)),
        ("opaque", C.c_void_p),
    ]


class AfqEqclasses(C.Structure):
    _fields_ = [
        ("n_cells", C.c_uint64),
        ("n_classes", C.c_uint64),
        ("n_words", C.c_uint64),
        ("cell_ptr", C.POINTER(C.c_uint64)),
        ("label_ptr", C.POINTER(C.c_uint64)),
        ("labels", C.POINTER(C.c_uint32)),
        ("count", C.POINTER(C.c_uint32)),
    ]


class AfqBootstraps(C.Structure):
    _fields_ = [
        ("n_cells", C.c_uint64),
        ("mean_ptr", C.POINTER(C.c_uint64)), ("mean_col", C.POINTER(C.c_uint32)), ("mean_val", C.POINTER(C.c_float)),
        ("var_ptr", C.POINTER(C.c_uint64)), ("var_col", C.POINTER(C.c_uint32)), ("var_val", C.POINTER(C.c_float)),
    ]


class AfqKernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_double), ("launches", C.c_uint32), ("pad", C.c_uint32)]


class AfqBatchStats(C.Structure):
    _fields_ = [
        ("n_records", C.c_uint64),
        ("n_ref_words", C.c_uint64),
        ("n_keys", C.c_uint64),
        ("n_buckets", C.c_uint64),
        ("n_overflow_buckets", C.c_uint64),
        ("input_bytes", C.c_uint64),
        ("n_fallback_cells", C.c_uint64),
    ]


# every symbol include/afquant.h declares (tests check the .so exports them all)
EXPORTS = [
    "afq_create",
    "afq_destroy",
    "afq_submit",
    "afq_submit_device",
    "afq_submit_reader",
    "afq_collect",
    "afq_result_release",
    "afq_result_eqclasses",
    "afq_result_bootstraps",
    "afq_infer",
    "afq_atac_dedup",
    "afq_atac_dedup_rad",
    "afq_device_warmup", "afq_device_pci_bus_id", "afq_label_rehash_count", "afq_pool_regrow_count", "afq_em_resize_count", "afq_mono_cell_count",
    "afq_free",
    "afq_get_kernel_times",
    "afq_get_batch_stats",
    "afq_last_error",
    "afq_abi_version",
]


class AfqAtacStats(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_multimapped", C.c_uint64), ("n_not_mapped_pair", C.c_uint64), ("n_distinct", C.c_uint64),
                ("n_deduplicated", C.c_uint64), ("n_long_fragments", C.c_uint64), ("n_fallback_cells", C.c_uint64)]
