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
    assert table == defs and len(table) >= 9
    # the entry that replaces the reference's keeps its name and arity (src/RcppExports.cpp:61,89)
    assert table["_gficf_rcpp_parallel_jaccard_coef"] == 2 and table["_gficf_jaccard_coeff"] == 2
    # the (N*k) x 3 allocation is range-checked, never a bare (int)(N * k)
    assert "(int)(N * k), 3" not in src and src.count("edge_rows(N, k), 3") == 2
