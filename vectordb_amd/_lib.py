"""ctypes binding of libvdb.so (include/vdb.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VDB_LIBRARY names another build of the same library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("VDB_LIBRARY") or os.path.join(_HERE, "libvdb.so")

OK = 0
EINVAL, ENOTFOUND, EEXISTS, ENOGPU, EHIP, ENOMEM, ESTATE = -1, -2, -3, -4, -5, -6, -7
METRIC_L2, METRIC_IP, METRIC_COSINE = 1, 2, 3
F32, BF16, F16 = 0, 1, 2
INDEX_FLAT, INDEX_IVFLAT = 1, 2
MAX_TOPK = 1024

_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)

# name -> (restype, argtypes): every symbol include/vdb.h declares
SIGNATURES = {
    "vdb_init": (C.c_int, [C.c_int]),
    "vdb_shutdown": (C.c_int, []),
    "vdb_set_option": (C.c_int, [C.c_char_p, C.c_double]),
    "vdb_last_error": (C.c_char_p, []),
    "vdb_create_table": (C.c_int, [C.c_char_p, C.c_int64, C.c_int, C.c_int]),
    "vdb_has_table": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "vdb_drop_table": (C.c_int, [C.c_char_p]),
    "vdb_create_index": (C.c_int, [C.c_char_p, C.c_int, C.c_int]),
    "vdb_count": (C.c_int, [C.c_char_p, _i64p]),
    "vdb_reserve": (C.c_int, [C.c_char_p, C.c_int64]),
    "vdb_table_info": (C.c_int, [C.c_char_p, _i64p, C.POINTER(C.c_int), C.POINTER(C.c_int), _i64p, _i64p, _i64p]),
    "vdb_table_mirror_bytes": (C.c_int, [C.c_char_p, _i64p]),
    "vdb_tables_created": (C.c_int64, []),
    "vdb_insert": (C.c_int, [C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "vdb_insert_dev": (C.c_int, [C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdb_insert_synthetic": (C.c_int, [C.c_char_p, C.c_int64, C.c_uint64, C.c_int64]),
    "vdb_synth_fill_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64, C.c_int64, C.c_void_p]),
    "vdb_fetch_rows": (C.c_int, [C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "vdb_save_table": (C.c_int, [C.c_char_p, C.c_char_p]),
    "vdb_save_table_rows": (C.c_int, [C.c_char_p, C.c_char_p, _i64p]),
    "vdb_load_table": (C.c_int, [C.c_char_p, C.c_char_p]),
    "vdb_search": (C.c_int, [C.c_char_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdb_search_dev": (C.c_int, [C.c_char_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdb_search_dev_noalloc": (C.c_int, [C.c_char_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdb_merge_topk_dev": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdb_record_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "vdb_merge_records_dev": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdb_bench_read_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vdb_prof_enable": (C.c_int, [C.c_int]),
    "vdb_prof_read": (C.c_int, [C.POINTER(C.c_double), _i64p]),
    "vdb_stats": (C.c_int, [_i64p, _i64p, _i64p]),
}

_lib = None


class VdbError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("vdb error %d: %s" % (code, message))
        self.code = code
        self.message = message


def load() -> C.CDLL:
    """Load libvdb.so and declare every entry point.  Raises if the library has not been built
    (run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C vectordb_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libvdb.so is missing at %s: build it with make -C vectordb_amd/csrc" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)         # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    msg = load().vdb_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(code: int) -> None:
    if code != OK:
        raise VdbError(code, last_error())
