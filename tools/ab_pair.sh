#!/bin/bash
# A/B timing of the pair launch at C3 (lattice at rest): the libraries named on the command line
# (default: the built one), each with the groups-per-wave settings in $GPWS.
# usage (GPU box): GPWS="0 1 8" tools/ab_pair.sh [lib.so ...]
cd "$GRAFT_REPO_ROOT" || exit 1
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(sph_amd/lib/libsph_amd.so)
for lib in "${libs[@]}"; do
  for g in ${GPWS:-0}; do
    echo "== $lib gpw=$g"
    SPH_AMD_LIB=$lib SPH_PAIR_GPW=$g timeout -k 10 200 python tools/ablate_pair.py 2>&1 | tail -1 || exit 1
  done
done
