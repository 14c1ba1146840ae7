#!/bin/bash
# profiles/exp/build_variant.sh NAME "-DCC_EXP_..."  -- a copy of the library whose headline geometry (g255_24) and
# host-side dealer are compiled with extra flags; everything else is linked from the product build (csrc/build).
# The result profiles/exp/lib_NAME.so is selected with CHANNELCODING_AMD_LIB (see channelcoding_amd/_capi.py).
set -e
name=$1; shift
here=$(cd $(dirname $0) && pwd); src=$here/../../channelcoding_amd/csrc; obj=$here/obj_$name; mkdir -p $obj
flags="-std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-parameter --offload-arch=gfx950 -DCC_AMD_EXPERIMENTS $*"
geo=$(awk -F'[(,) ]+' '$1 == "GEO" && $2 == "g255_24" {print "-DGEO_NAME=" $2 " -DGEO_K=" $4 " -DGEO_D=" $6 " -DGEO_LPF=" $7 " -DGEO_CPL=" $8 " -DGEO_OCC=" $9 " -DGEO_SCMS=" $10 " -DGEO_PARTIAL=" $11 " -DGEO_LINKS=" $12 " -DGEO_PARTS=" $13}' $src/minsum_diag_geos.inc)
parts=$(awk -F'[(,) ]+' '$1 == "GEO" && $2 == "g255_24" {print $13}' $src/minsum_diag_geos.inc)
mine=""
for ((i = 0; i < parts; ++i)); do  # the variants of the geometry are spread over `parts` objects (minsum_diag_geos.inc)
  hipcc $flags -fno-slp-vectorize $geo -DGEO_PART=$i -c $src/minsum_diag_geo.hip -o $obj/geo_g255_24_p$i.o &
  mine="$mine $obj/geo_g255_24_p$i.o"
done
hipcc $flags -c $src/minsum_diag.hip -o $obj/minsum_diag.o &
wait
others=$(ls $src/build/*.o | grep -v -e geo_g255_24_p -e /minsum_diag.o)
hipcc -shared -fPIC --offload-arch=gfx950 -o $here/lib_$name.so $others $mine $obj/minsum_diag.o
echo built $here/lib_$name.so
