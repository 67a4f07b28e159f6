"""Python side of the runnable R stand-in (tests/r_mock/r_runtime.c): builds integration/gficf_hip_glue.c against it into
tests/r_mock/libgficf_glue_mock.so and calls the glue's `.Call` entry points through the routine table they register.
NOT R and not an oracle: test infrastructure for OUR glue file only."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MOCK = os.path.join(ROOT, "tests", "r_mock")
SO = os.path.join(MOCK, "libgficf_glue_mock.so")
SOURCES = [os.path.join(ROOT, "integration", "gficf_hip_glue.c"), os.path.join(MOCK, "r_runtime.c"), os.path.join(MOCK, "package_rows.c")]
HEADERS = [os.path.join(MOCK, "R.h"), os.path.join(MOCK, "Rinternals.h"), os.path.join(MOCK, "R_ext", "Rdynload.h"),
           os.path.join(ROOT, "include", "gficf_hip.h")]

INTSXP, REALSXP, LGLSXP, STRSXP, VECSXP, RAWSXP, S4SXP = 13, 14, 10, 16, 19, 24, 25


class RError(RuntimeError):
    """An Rf_error raised inside a `.Call` entry (what R would turn into an R-level error)."""


def build(force: bool = False) -> str:
    lib = os.path.join(ROOT, "gficf_amd", "libgficf_hip.so")
    if not os.path.exists(lib):
        raise RuntimeError("gficf_amd/libgficf_hip.so is missing: run __graft_entry__.build() first")
    newest = max(os.path.getmtime(f) for f in SOURCES + HEADERS)
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < newest:
        cmd = ["gcc", "-std=c99", "-O1", "-g", "-shared", "-fPIC", "-Wall", "-Wextra", "-Werror", "-Wno-cast-function-type",
               "-I", MOCK, "-I", os.path.join(ROOT, "include")] + SOURCES + \
              ["-L", os.path.join(ROOT, "gficf_amd"), "-lgficf_hip", "-Wl,-rpath," + os.path.join(ROOT, "gficf_amd"),
               "-Wl,-rpath,$ORIGIN/../../gficf_amd", "-o", SO]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            raise RuntimeError("building the glue against the R stand-in failed:\n" + r.stderr[-3000:])
    return SO


class Sexp:
    def __init__(self, rt, ptr):
        self.rt, self.ptr = rt, ptr

    @property
    def type(self):
        return self.rt.L.rmock_type(self.ptr)

    def __len__(self):
        return int(self.rt.L.rmock_length(self.ptr))

    @property
    def dim(self):
        d0 = self.rt.L.rmock_dim(self.ptr, 0)
        return None if d0 < 0 else (d0, self.rt.L.rmock_dim(self.ptr, 1))

    def attr(self, name):
        p = self.rt.L.rmock_get_attr(self.ptr, name.encode())
        return None if p == self.rt.nil else Sexp(self.rt, p)

    def elt(self, i):
        return Sexp(self.rt, self.rt.L.rmock_elt(self.ptr, i))

    def numpy(self):
        """A copy of the payload (matrices come back in R's column-major layout as (nrow, ncol) arrays)."""
        t, n = self.type, len(self)
        dt = {INTSXP: np.int32, LGLSXP: np.int32, REALSXP: np.float64, RAWSXP: np.uint8}[t]
        if n == 0:
            a = np.zeros(0, dtype=dt)
        else:
            buf = (ctypes.c_char * (n * np.dtype(dt).itemsize)).from_address(self.rt.L.rmock_data(self.ptr))
            a = np.frombuffer(buf, dtype=dt).copy()
        d = self.dim
        return a.reshape(d, order="F") if d is not None else a

    def strings(self):
        return [self.rt.L.rmock_chars(self.rt.L.rmock_elt(self.ptr, i)).decode() for i in range(len(self))]


class RMock:
    def __init__(self):
        import torch  # noqa: F401  (one HIP runtime in the process: gficf_amd/_lib.py does the same before dlopen)

        from gficf_amd import _lib

        _lib.load()
        self.L = L = ctypes.CDLL(build())
        vp, ci, cl, cs = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_char_p
        for name, res, args in [
            ("rmock_init", None, []), ("rmock_unload", None, []), ("rmock_reset", None, []), ("rmock_set_torture", None, [ci]),
            ("rmock_n_routines", ci, []), ("rmock_routine_name", cs, [ci]), ("rmock_routine_nargs", ci, [ci]),
            ("rmock_new_vector", vp, [ci, cl]), ("rmock_new_matrix", vp, [ci, ci, ci]), ("rmock_nil", vp, []),
            ("rmock_new_string", vp, [cs]), ("rmock_set_slot", None, [vp, cs, vp]), ("rmock_get_attr", vp, [vp, cs]),
            ("rmock_data", vp, [vp]), ("rmock_length", cl, [vp]), ("rmock_type", ci, [vp]), ("rmock_is_dead", ci, [vp]),
            ("rmock_elt", vp, [vp, cl]), ("rmock_chars", cs, [vp]), ("rmock_dim", ci, [vp, ci]),
            ("rmock_error_message", cs, []), ("rmock_printed", cs, []), ("rmock_driver_message", cs, []),
            ("rmock_protect_depth", ci, []), ("rmock_protect_max", ci, []), ("rmock_collected", cl, []),
            ("rmock_dead_touched", cl, []), ("rmock_allocations", cl, []),
            ("rmock_use_dynamic_symbols", ci, []), ("rmock_n_registrations", ci, []), ("rmock_fail_allocation_after", None, [cl]),
            ("rmock_selftest_second_registration_replaces", ci, []),
            ("rmock_call", ci, [cs, ci, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ci)]),
        ]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        L.rmock_init()
        self.nil = L.rmock_nil()
        self.last_protect_exit = 0

    # ---- argument builders
    def routines(self) -> dict:
        return {self.L.rmock_routine_name(i).decode(): self.L.rmock_routine_nargs(i) for i in range(self.L.rmock_n_routines())}

    def _fill(self, ptr, a):
        if a.size:
            ctypes.memmove(self.L.rmock_data(ptr), a.ctypes.data, a.nbytes)

    def vector(self, a) -> Sexp:
        a = np.asarray(a)
        if a.dtype == np.bool_:
            t, a = LGLSXP, a.astype(np.int32)
        elif a.dtype.kind in "iu":
            t, a = INTSXP, a.astype(np.int32)
        else:
            t, a = REALSXP, a.astype(np.float64)
        p = self.L.rmock_new_vector(t, a.size)
        self._fill(p, np.ascontiguousarray(a.reshape(-1)))
        return Sexp(self, p)

    def matrix(self, a) -> Sexp:
        a = np.asarray(a)
        assert a.ndim == 2
        t, a = (INTSXP, a.astype(np.int32)) if a.dtype.kind in "iu" else (REALSXP, a.astype(np.float64))
        p = self.L.rmock_new_matrix(t, a.shape[0], a.shape[1])
        self._fill(p, np.asfortranarray(a).reshape(-1, order="F").copy())
        return Sexp(self, p)

    def null(self) -> Sexp:
        return Sexp(self, self.nil)

    def string(self, s: str) -> Sexp:
        return Sexp(self, self.L.rmock_new_string(s.encode()))

    def dgcmatrix(self, M) -> Sexp:
        """An S4-style object with the slots of a Matrix::dgCMatrix (i, p, x, Dim)."""
        import scipy.sparse as sp

        M = sp.csc_matrix(M)
        M.sort_indices()
        o = Sexp(self, self.L.rmock_new_vector(S4SXP, 0))
        for name, v in (("i", M.indices.astype(np.int32)), ("p", M.indptr.astype(np.int32)), ("x", M.data.astype(np.float64)),
                        ("Dim", np.array(M.shape, dtype=np.int32))):
            self.L.rmock_set_slot(o.ptr, name.encode(), self.vector(v).ptr)
        return o

    # ---- .Call
    def call(self, name: str, *args: Sexp) -> Sexp:
        arr = (ctypes.c_void_p * max(len(args), 1))(*[a.ptr for a in args])
        out, depth = ctypes.c_void_p(), ctypes.c_int()
        st = self.L.rmock_call(name.encode(), len(args), arr, ctypes.byref(out), ctypes.byref(depth))
        self.last_protect_exit = depth.value
        if st == 1:
            raise RError(self.L.rmock_error_message().decode())
        if st == 2:
            raise KeyError(f"no registered routine {name}")
        if st == 3:
            raise TypeError(f"{name}: wrong number of arguments ({len(args)})")
        if st == 4:
            raise AssertionError(f"{name} returned with the protect stack unbalanced by {depth.value}")
        if st == 5:
            raise AssertionError(f"{name} returned an object the collector had taken (missing PROTECT)")
        msg = self.L.rmock_driver_message().decode()
        if msg:
            raise AssertionError(f"{name}: {msg}")
        return Sexp(self, out.value)

    def printed(self) -> str:
        return self.L.rmock_printed().decode()

    def unload(self):
        """R_unload_gficf: contexts destroyed, GFICF_HIP_DEVICES read again at the next call."""
        self.L.rmock_unload()
