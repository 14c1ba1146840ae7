#!/bin/bash
# profiles/exp/build_rs_variant.sh NAME "-DFLAGS": a copy of the library whose algebraic / algebraic_chunk / bitslice units are
# compiled with -DCC_AMD_EXPERIMENTS and extra flags; everything else is linked from the product build.
set -e
name=$1; shift
here=$(cd $(dirname $0) && pwd); src=$here/../../channelcoding_amd/csrc; obj=$here/obj_$name; mkdir -p $obj
flags="-std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-parameter --offload-arch=gfx950 -DCC_AMD_EXPERIMENTS $*"
hipcc $flags -c $src/algebraic_chunk.hip -o $obj/algebraic_chunk.o &
hipcc $flags -c $src/bitslice.hip -o $obj/bitslice.o &
hipcc $flags -c $src/algebraic.hip -o $obj/algebraic.o &
wait
others=$(ls $src/build/*.o | grep -v -e /algebraic_chunk.o -e /bitslice.o -e /algebraic.o)
hipcc -shared -fPIC --offload-arch=gfx950 -o $here/lib_$name.so $others $obj/algebraic_chunk.o $obj/bitslice.o $obj/algebraic.o
echo built $here/lib_$name.so
