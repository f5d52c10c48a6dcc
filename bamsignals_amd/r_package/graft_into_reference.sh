#!/bin/sh
# Turns a checkout of lamortenera/bamsignals into the MI355X-backed package: the reference's own
# R/ (generics, CountSignals class, argument normalisation, messages), man/, vignettes/, inst/ and
# tests/ stay exactly as they are; ONLY the native half changes:
#   src/bamsignals.cpp, src/CountSignals.cpp   (Rcpp + htslib)   -> removed
#   src/RcppExports.cpp, src/bamsignals_init.c (generated glue)  -> src/shim.c (plain C .Call shim)
#   src/Makevars(.win)                         (Rhtslib)         -> src/Makevars (links libbamsignals_hip.so)
# R/RcppExports.R keeps working unchanged: it reaches the five routines by their string names
# (.Call('bamsignals_pileup_core', PACKAGE = 'bamsignals', ...)), which shim.c registers with the same
# names and arities (ref: src/bamsignals_init.c:12-19).
#
# usage: graft_into_reference.sh <bamsignals checkout> [<dir holding include/ and bamsignals_amd/>]
set -eu
REF=${1:?usage: graft_into_reference.sh <bamsignals checkout> [<this repository>]}
HERE=$(cd "$(dirname "$0")" && pwd)
REPO=${2:-$(cd "$HERE/../.." && pwd)}
test -f "$REF/DESCRIPTION" && grep -q '^Package: bamsignals' "$REF/DESCRIPTION" || { echo "$REF is not a bamsignals checkout" >&2; exit 1; }
test -f "$REPO/bamsignals_amd/libbamsignals_hip.so" || { echo "build the library first: make -C $REPO/bamsignals_amd/csrc" >&2; exit 1; }
rm -f "$REF/src/bamsignals.cpp" "$REF/src/CountSignals.cpp" "$REF/src/RcppExports.cpp" "$REF/src/bamsignals_init.c" \
      "$REF/src/Makevars" "$REF/src/Makevars.win" "$REF"/src/*.o "$REF"/src/*.so
cp "$HERE/src/shim.c" "$REF/src/shim.c"
sed "s|^BAMSIGNALS_HIP_DIR ?=.*|BAMSIGNALS_HIP_DIR ?= $REPO|" "$HERE/src/Makevars" > "$REF/src/Makevars"
# Rcpp and Rhtslib are no longer linked to or imported
sed -i -e '/^LinkingTo:/d' -e '/^    Rcpp (>= 0.10.6),$/d' -e 's/^SystemRequirements: GNU make$/SystemRequirements: GNU make, ROCm >= 7.0, an AMD Instinct MI355X (gfx950), libbamsignals_hip.so/' "$REF/DESCRIPTION"
sed -i -e '/^import(Rcpp)$/d' "$REF/NAMESPACE"
# acceptance tests of this port, next to the reference's own
cp "$HERE"/tests/testthat/*.R "$REF/tests/testthat/"
echo "grafted.  Next:  R CMD INSTALL $REF  &&  (cd $REF/tests && Rscript testthat.R)"
