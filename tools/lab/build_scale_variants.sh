#!/bin/bash
# Lab builds of libgficf_hip.so for the GF-ICF scaling pass (tools/lab/scale_probe.py times them): tools/lab/abl_scale/libgficf_hip_<name>.so
#   base    the product source
#   strip   the same loads and stores, no gene lookups in LDS and no division (3 of 4 entries kept by a bit test of the row id):
#           what the pass's memory pattern alone costs
#   plain   no non-temporal hints on the loads and stores
#   t<T>c<C> GFICF_SL_THREADS = T, GFICF_SL_CH = C
#   rev     with GFICF_SCALE_STATIC_CELLS=1 (cells dealt round-robin to all waves of the grid: the grid sweeps the matrix as one front)
#           the sweep runs from the LAST cell backwards — towards what the count pass left in the memory-side cache
set -e
cd "$(dirname "$0")/../../gficf_amd/csrc"
make -s
OUT=../../tools/lab/abl_scale
mkdir -p $OUT
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -I../../include -I."
link() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libgficf_hip_$1.so ctx.o jaccard.o halo.o $2 knn.o adjacency.o transpose.o louvain.o phenograph.o multi.o; }
for V in "$@"; do
  case $V in
    base) link base gficf_csc.o ;;
    strip)
      python3 - <<'PY'
s = open("gficf_csc.hip").read()
a = "          const uint32_t r = g < (uint32_t)G ? (uint32_t)s_remap[g] : 0xFFFFu;\n          rv[m] = r == 0xFFFFu ? -1 : (int32_t)r;"
assert s.count(a) == 1
s = s.replace(a, "          const uint32_t r = (g & 3u) ? (g & 0x1FFFu) : 0xFFFFu;\n          rv[m] = r == 0xFFFFu ? -1 : (int32_t)r;")
b = "          if (rv[m] >= 0 && Sc != 0.0) v = (xv[m] / Sc) * weight(rv[m]);"
assert s.count(b) == 1
s = s.replace(b, "          if (rv[m] >= 0 && Sc != 0.0) v = xv[m] * Sc;")
open("/tmp/gficf_csc_strip.hip", "w").write(s)
PY
      /opt/rocm/bin/hipcc $FLAGS -c /tmp/gficf_csc_strip.hip -o /tmp/gficf_csc_strip.o; link strip /tmp/gficf_csc_strip.o ;;
    plain)
      sed -e 's/__builtin_nontemporal_load(\([^)]*\))/(*(\1))/g' -e 's/__builtin_nontemporal_store(\([^,]*\), \([^)]*\))/(*(\2) = (\1))/g' gficf_csc.hip > /tmp/gficf_csc_plain.hip
      /opt/rocm/bin/hipcc $FLAGS -c /tmp/gficf_csc_plain.hip -o /tmp/gficf_csc_plain.o; link plain /tmp/gficf_csc_plain.o ;;
    rev)
      python3 - <<'PY'
s = open("gficf_csc.hip").read()
a = "    c_next = grab();                               // asked for early: the round trip hides behind this cell's loads\n"
assert s.count(a) == 1
s = s.replace(a, a + "    if (static_cells) c = n_cells - 1 - c;\n")
open("/tmp/gficf_csc_rev.hip", "w").write(s)
PY
      /opt/rocm/bin/hipcc $FLAGS -c /tmp/gficf_csc_rev.hip -o /tmp/gficf_csc_rev.o; link rev /tmp/gficf_csc_rev.o ;;
    t*c*)
      T=${V#t}; T=${T%c*}; C=${V#*c}
      /opt/rocm/bin/hipcc $FLAGS -DGFICF_SL_THREADS=$T -DGFICF_SL_CH=$C -c gficf_csc.hip -o /tmp/gficf_csc_$V.o; link $V /tmp/gficf_csc_$V.o ;;
  esac
done
