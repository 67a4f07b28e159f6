"""Boundary hygiene of the R glue (integration/gficf_hip_glue.c): R is absent from this image, so the file is syntax-
and type-checked against minimal, clearly-labelled mock declarations of the R C API (tests/r_mock/).  This pins nothing
about R's behaviour and is not an oracle; it catches signature drift between the glue and include/gficf_hip.h, and
registration-table / arity mistakes (reference convention: src/RcppExports.cpp:85-97)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "integration", "gficf_hip_glue.c")


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_glue_compiles_against_the_mock_r_headers():
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-cast-function-type",
                        "-I", os.path.join(ROOT, "tests", "r_mock"), "-I", os.path.join(ROOT, "include"), GLUE],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]


def test_registration_table_matches_the_entry_points():
    src = open(GLUE).read()
    defs = {m.group(1): len([a for a in m.group(2).split(",") if a.strip()])
            for m in re.finditer(r"^SEXP (_gficf_\w+)\(([^)]*)\)\s*\{", src, re.M)}
    table = {m.group(1): int(m.group(2)) for m in re.finditer(r'\{"(_gficf_\w+)", \(DL_FUNC\)&\1, (\d+)\}', src)}
    # the three Rcpp wrappers that stay in the package are rows of the ONE table too, declared extern (not defined) in the glue with
    # the arity of the reference's own table (src/RcppExports.cpp:86,88,90)
    package = {"_gficf_RunModularityClusteringCpp": 9, "_gficf_rcpp_WMU_test": 3, "_gficf_rcpp_parallel_WMU_test": 3}
    for name, n in package.items():
        assert table.pop(name) == n and name not in defs
        ext = re.search(r"^extern SEXP " + name + r"\(([^)]*)\);", src, re.M)
        assert ext and len(ext.group(1).split(",")) == n
    assert table == defs and len(table) == 10
    assert int(re.search(r"#define GFICF_HIP_N_ROWS (\d+)", src).group(1)) == len(table)
    # exactly one R_registerRoutines call in the file, inside gficf_hip_register, followed by R_useDynamicSymbols(dll, FALSE) (:95-96)
    assert len(re.findall(r"^\s*R_registerRoutines\(", src, re.M)) == 1 and "R_useDynamicSymbols(dll, FALSE);" in src
    # the entry that replaces the reference's keeps its name and arity (src/RcppExports.cpp:61,89)
    assert table["_gficf_rcpp_parallel_jaccard_coef"] == 2 and table["_gficf_jaccard_coeff"] == 2
    # the (N*k) x 3 allocation is range-checked, never a bare (int)(N * k)
    assert "(int)(N * k), 3" not in src and src.count("edge_rows(N, k), 3") == 2
