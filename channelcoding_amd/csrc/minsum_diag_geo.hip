// minsum_diag_geo.hip -- ONE geometry of the diagonal-parallel min-sum kernel (minsum_diag_impl.hpp): compiled once
// per line of minsum_diag_geos.inc with -DGEO_NAME=... -DGEO_K=... (the Makefile derives the flags from that file),
// so that the instantiations build in parallel.
#include "minsum_diag_impl.hpp"

#if !defined(GEO_NAME) || !defined(GEO_K) || !defined(GEO_D) || !defined(GEO_LPF) || !defined(GEO_CPL) || \
    !defined(GEO_OCC) || !defined(GEO_SCMS) || !defined(GEO_PARTIAL)
#error "compile through the Makefile: one geometry of minsum_diag_geos.inc per object"
#endif
#ifndef GEO_LINKS
#define GEO_LINKS 0
#endif
#ifndef GEO_PARTS  // the variants of this geometry are spread over GEO_PARTS objects; this one is number GEO_PART
#define GEO_PARTS 1
#define GEO_PART 0
#endif
#define CC_GEO_CAT2(a, b) a##b
#define CC_GEO_CAT(a, b) CC_GEO_CAT2(a, b)
#define CC_GEO_FN CC_GEO_CAT(CC_GEO_CAT(CC_GEO_CAT(launch_minsum_diag_, GEO_NAME), _p), GEO_PART)

namespace ccamd {

int CC_GEO_FN(const cc_code *code, const MinSumParams &p, const float *d_llr, const uint16_t *d_er,
              const uint32_t *d_er_off, uint8_t *d_hard, float *d_L, uint16_t *d_iters, int32_t *d_status, size_t B,
              hipStream_t stream) {
#if GEO_LINKS == 2
  return launch_diag_geometry<GEO_K, GEO_D, GEO_LPF, GEO_CPL, GEO_OCC, GEO_SCMS, GEO_PARTIAL, PairGaps<1, 1>, true, GEO_PARTS,
                              GEO_PART>(
      code, p, d_llr, d_er, d_er_off, d_hard, d_L, d_iters, d_status, B, stream);
#elif GEO_LINKS == 1
  return launch_diag_geometry<GEO_K, GEO_D, GEO_LPF, GEO_CPL, GEO_OCC, GEO_SCMS, GEO_PARTIAL, PairGaps<1>, true, GEO_PARTS,
                              GEO_PART>(
      code, p, d_llr, d_er, d_er_off, d_hard, d_L, d_iters, d_status, B, stream);
#else
  return launch_diag_geometry<GEO_K, GEO_D, GEO_LPF, GEO_CPL, GEO_OCC, GEO_SCMS, GEO_PARTIAL, PairGaps<>, false, GEO_PARTS,
                              GEO_PART>(
      code, p, d_llr, d_er, d_er_off, d_hard, d_L, d_iters, d_status, B, stream);
#endif
}

}  // namespace ccamd
