// minsum_diag_small.hip -- instantiations of the diagonal-parallel min-sum kernel (minsum_diag_impl.hpp) for the
// short codes: eight lanes per frame, eight frames per wavefront.
#include "minsum_diag_impl.hpp"

namespace ccamd {

int launch_minsum_diag_small(const cc_code *code, const DiagGeometry &g, const MinSumParams &p, const float *d_llr,
                             const uint16_t *d_er, const uint32_t *d_er_off, uint8_t *d_hard, float *d_L,
                             uint16_t *d_iters, int32_t *d_status, size_t B, hipStream_t stream) {
#define CC_GEO(KK, DD, LL, CC, OO, SC, PA)                                                                      \
  if (g.k == KK && g.D == DD && g.LPF == LL && g.CPL == CC && (g.w != static_cast<unsigned>(LL * DD)) == PA)    \
  return launch_diag_geometry<KK, DD, LL, CC, OO, SC, PA>(code, p, d_llr, d_er, d_er_off, d_hard, d_L, d_iters,   \
                                                          d_status, B, stream)
  //      K  D LPF CPL OCC SCMS  partial
  CC_GEO(6, 4, 8, 8, 4, true, false);    // BCH(63,57)
  CC_GEO(12, 4, 8, 8, 4, true, true);    // BCH(63,51)
  CC_GEO(18, 3, 8, 8, 4, true, false);   // BCH(63,45): 54 message registers (108 with the self-correcting q)
  CC_GEO(24, 4, 8, 8, 3, true, true);    // BCH(63,39): 96 message registers
  CC_GEO(5, 2, 8, 4, 4, true, false);    // BCH(31,26)
  CC_GEO(10, 2, 8, 4, 4, true, true);    // BCH(31,21)
  CC_GEO(15, 1, 8, 4, 4, true, false);   // BCH(31,16)
  CC_GEO(20, 1, 8, 4, 4, true, true);    // BCH(31,11)
  CC_GEO(4, 1, 8, 2, 4, true, false);    // BCH(15,11)
  CC_GEO(8, 1, 8, 2, 4, true, true);     // BCH(15,7)
  CC_GEO(10, 1, 8, 2, 4, true, true);    // BCH(15,5)
#undef CC_GEO
  return CC_ERR_UNSUPPORTED;
}

}  // namespace ccamd
