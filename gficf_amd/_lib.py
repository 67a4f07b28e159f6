"""ctypes binding of libgficf_hip.so (the C ABI declared in include/gficf_hip.h).

The library is built in-tree by ``gficf_amd.build.build_extension()`` (hipcc, gfx950) and
lives next to this file.  There is no CPU fallback: if the library is missing or no GPU
is visible, the product path raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GFICF_HIP_LIB") or os.path.join(_HERE, "libgficf_hip.so")

GFICF_OK = 0
STATUS_NAMES = {
    0: "GFICF_OK", 1: "GFICF_ERR_INVALID_ARG", 2: "GFICF_ERR_BAD_ID", 3: "GFICF_ERR_BAD_CSC",
    4: "GFICF_ERR_NO_DEVICE", 5: "GFICF_ERR_HIP", 6: "GFICF_ERR_UNSUPPORTED", 7: "GFICF_ERR_CAPACITY",
    8: "GFICF_ERR_BAD_VALUE", 9: "GFICF_ERR_EXPLICIT_ZEROS", 10: "GFICF_ERR_DUPLICATE_IDS", 11: "GFICF_ERR_SET_OVERFLOW",
}
JACCARD_MAX_K = 256          # the fast kernels; beyond it the exact sorted-row path, up to JACCARD_MAX_K_EXACT
JACCARD_MAX_K_EXACT = 65535
KNN_MAX_K = 128
KNN_METRICS = {"manhattan": 0, "euclidean": 1, "cosine": 2, "correlation": 3}


class GficfError(RuntimeError):
    """A libgficf_hip call returned a non-zero gficf_status."""

    def __init__(self, code: int, message: str):
        self.code = code
        self.status = STATUS_NAMES.get(code, str(code))
        super().__init__(f"{self.status}: {message}")


_i64, _int, _dbl, _vp = ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/gficf_hip.h declares
SIGNATURES = {
    "gficf_hip_abi_version": (_int, []),
    "gficf_device_count": (_int, [ctypes.POINTER(_int)]),
    "gficf_ctx_create": (_int, [_int, _vp, ctypes.POINTER(_vp)]),
    "gficf_ctx_destroy": (None, [_vp]),
    "gficf_ctx_set_stream": (_int, [_vp, _vp]),
    "gficf_ctx_sync": (_int, [_vp]),
    "gficf_ctx_set_gficf_options": (_int, [_vp, _int, _int]),
    "gficf_ctx_set_louvain_options": (_int, [_vp, _int]),
    "gficf_ctx_set_jaccard_options": (_int, [_vp, _int]),
    "gficf_ctx_set_jaccard_distinct": (_int, [_vp, _int]),
    "gficf_ctx_set_jaccard_direct_max_edges": (_int, [_vp, _i64]),
    "gficf_jaccard_one_launch": (_int, [_vp, _i64, _int]),
    "gficf_last_error": (ctypes.c_char_p, []),
    "gficf_ctx_set_print": (_int, [_vp, _vp]),
    "gficf_ctx_trim": (_int, [_vp]),
    "gficf_jaccard_counts_host": (_int, [_vp, _vp, _int, _i64, _int, _i64, _vp]),
    "gficf_jaccard_expand_host": (_int, [_vp, _int, _i64, _int, _i64, _vp, _vp, _int]),
    "gficf_multi_create": (_int, [_vp, _int, ctypes.POINTER(_vp)]),
    "gficf_multi_destroy": (None, [_vp]),
    "gficf_multi_set_print": (_int, [_vp, _vp]),
    "gficf_multi_cell_blocks": (_int, [_i64, _int, _vp]),
    "gficf_multi_cell_blocks_by_nnz": (_int, [_i64, _vp, _int, _int, _vp]),
    "gficf_jaccard_host_multi": (_int, [_vp, _vp, _int, _i64, _int, _i64, _vp, _int]),
    "gficf_multi_jaccard_device": (_int, [_vp, _vp, _int, _vp, _i64, _int, _vp, _vp]),
    "gficf_multi_jaccard_halo_device": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "gficf_multi_sync": (_int, [_vp]),
    "gficf_multi_set_jaccard_distinct": (_int, [_vp, _int]),
    "gficf_normalize_csc_host_multi_plan": (_int, [_vp, _i64, _i64, _vp, _int, _vp, _vp, _dbl, _dbl, _vp,
                                                   ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "gficf_normalize_csc_host_multi_finish": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gficf_jaccard_host": (_int, [_vp, _vp, _int, _i64, _int, _i64, _vp, _int]),
    "gficf_jaccard_coeff_host": (_int, [_vp, _vp, _int, _i64, _int, _i64, _vp, _int]),
    "gficf_jaccard_kpad": (_int, [_int]),
    "gficf_jaccard_row_words": (_int, [_i64, _int]),
    "gficf_jaccard_ingest_device": (_int, [_vp, _vp, _int, _i64, _int, _i64, _i64, _vp]),
    "gficf_jaccard_packed_words": (_int, [_i64, _int]),
    "gficf_jaccard_pack_rows_device": (_int, [_vp, _vp, _i64, _int, _i64, _vp]),
    "gficf_jaccard_unpack_rows_device": (_int, [_vp, _vp, _i64, _int, _i64, _vp]),
    "gficf_jaccard_edges_device": (_int, [_vp, _vp, _i64, _int, _i64, _i64, _vp, _vp, _vp, _vp]),
    "gficf_jaccard_halo_workspace_bytes": (ctypes.c_size_t, [_i64, _int]),
    "gficf_jaccard_halo_plan_device": (_int, [_vp, _vp, _i64, _int, _i64, _i64, _i64, _int, _i64, _int, _vp, _vp]),
    "gficf_jaccard_halo_serve_device": (_int, [_vp, _vp, _i64, _int, _i64, _i64, _vp, _i64, _vp]),
    "gficf_jaccard_halo_relabel_device": (_int, [_vp, _vp, _i64, _int, _i64, _i64, _i64, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp]),
    "gficf_jaccard_ingest_local_device": (_int, [_vp, _vp, _i64, _int, _i64, _vp]),
    "gficf_jaccard_halo_serve_ingest_device": (_int, [_vp, _vp, _i64, _int, _i64, _i64, _i64, _int, _i64, _int, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "gficf_jaccard_halo_ingest_slots_device": (_int, [_vp, _vp, _i64, _int, _i64, _i64, _i64, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp]),
    "gficf_jaccard_halo_ingest_peer_device": (_int, [_vp, _vp, _i64, _int, _i64, _i64, _i64, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gficf_jaccard_edges_mapped_device": (_int, [_vp, _vp, _i64, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gficf_jaccard_device": (_int, [_vp, _vp, _int, _i64, _int, _i64, _vp, _vp, _vp]),
    "gficf_jaccard_edges_filtered_device": (_int, [_vp, _vp, _i64, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gficf_jaccard_filtered_host_plan": (_int, [_vp, _vp, _int, _i64, _int, _i64, ctypes.POINTER(_i64)]),
    "gficf_jaccard_filtered_host_finish": (_int, [_vp, _vp, _vp, _vp]),
    "gficf_adjacency_workspace_bytes": (ctypes.c_size_t, [_i64, _i64]),
    "gficf_adjacency_device": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _int, _vp, ctypes.c_size_t, _vp, _vp, _vp]),
    "gficf_adjacency_host_plan": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, ctypes.POINTER(_i64)]),
    "gficf_adjacency_host_finish": (_int, [_vp, _vp, _int, _vp, _vp]),
    "gficf_normalize_csc_host_plan": (_int, [_vp, _i64, _i64, _vp, _int, _vp, _vp, _dbl, _dbl, _vp,
                                             ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "gficf_normalize_csc_host_finish": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gficf_normalize_csc_host_finish_raw": (_int, [_vp] * 11),
    "gficf_csc_kept_values_host": (_int, [_i64, _i64, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gficf_csc_genes_bytes": (ctypes.c_size_t, [_i64]),
    "gficf_csc_count_device": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp]),
    "gficf_csc_genes_device": (_int, [_vp, _i64, _i64, _vp, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp]),
    "gficf_csc_colptr_device": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gficf_csc_scale_device": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gficf_csc_scale_be_device": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gficf_csc_be_device": (_int, [_vp, _int, _i64, _i64, _vp, _vp, _vp, _i64, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gficf_cluster_signatures_be_device": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _int, _vp]),
    "gficf_csc_transpose_be_device": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, ctypes.c_size_t]),
    "gficf_cluster_signatures_device": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _int, _vp]),
    "gficf_cluster_signatures_host": (_int, [_vp, _i64, _i64, _vp, _int, _vp, _vp, _vp, _int, _vp]),
    "gficf_csc_transpose_workspace_bytes": (ctypes.c_size_t, [_i64, _i64]),
    "gficf_csc_transpose_device": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, ctypes.c_size_t]),
    "gficf_csc_transpose_host": (_int, [_vp, _i64, _i64, _vp, _int, _vp, _vp, _vp, _vp, _vp]),
    "gficf_louvain_workspace_bytes": (ctypes.c_size_t, [_i64, _i64, _int]),
    "gficf_louvain_device": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _dbl, _int, _int, _int, _int, _vp, ctypes.POINTER(_i64),
                                    ctypes.POINTER(_dbl), _vp, ctypes.c_size_t]),
    "gficf_louvain_host": (_int, [_vp, _i64, _vp, _int, _vp, _vp, _dbl, _int, _int, _int, _int, _vp, ctypes.POINTER(_i64),
                                  ctypes.POINTER(_dbl)]),
    "gficf_phenograph_host": (_int, [_vp, _vp, _i64, _int, _i64, _int, _int, _dbl, _int, _int, _int, _int, _vp, ctypes.POINTER(_i64),
                                     ctypes.POINTER(_dbl), ctypes.POINTER(_i64)]),
    "gficf_knn_dpad": (_int, [_int]),
    "gficf_knn_prepare_device": (_int, [_vp, _vp, _int, _i64, _int, _i64, _int, _vp]),
    "gficf_knn_workspace_bytes": (ctypes.c_size_t, [_vp, _i64, _i64, _int]),
    "gficf_knn_search_device": (_int, [_vp, _vp, _i64, _int, _int, _int, _i64, _i64, _vp, ctypes.c_size_t, _vp, _vp, _i64]),
    "gficf_knn_pivot_order_device": (_int, [_vp, _vp, _i64, _int, _int, _vp, ctypes.c_size_t, _vp]),
    "gficf_knn_host": (_int, [_vp, _vp, _i64, _int, _i64, _int, _int, _vp, _vp]),
    "gficf_csc_device": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp]),
    "gficf_csc_exact_device": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _vp, _vp]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libgficf_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C gficf_amd/csrc` (hipcc, --offload-arch=gfx950). gficf_amd has no CPU fallback.")
        # Load order matters in a Python process that also uses torch: torch ships its own copy of
        # the HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  If /opt/rocm's copy
        # is mapped first, torch later maps a second runtime and then finds no GPU; if torch's is
        # mapped first, our NEEDED libamdhip64.so.7 resolves to it and both share one runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI and this table diverge
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return (load().gficf_last_error() or b"").decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != GFICF_OK:
        raise GficfError(rc, last_error())
