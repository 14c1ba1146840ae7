// minsum_diag_127.hip -- instantiations of the diagonal-parallel min-sum kernel (minsum_diag_impl.hpp) for n = 127:
// eight lanes per frame, sixteen columns per lane.
#include "minsum_diag_impl.hpp"

namespace ccamd {

int launch_minsum_diag_127(const cc_code *code, const DiagGeometry &g, const MinSumParams &p, const float *d_llr,
                           const uint16_t *d_er, const uint32_t *d_er_off, uint8_t *d_hard, float *d_L,
                           uint16_t *d_iters, int32_t *d_status, size_t B, hipStream_t stream) {
#define CC_GEO(KK, DD, OO, SC)                                                                                   \
  if (g.k == KK && g.D == DD && g.LPF == 8 && g.CPL == 16)                                                       \
  return launch_diag_geometry<KK, DD, 8, 16, OO, SC>(code, p, d_llr, d_er, d_er_off, d_hard, d_L, d_iters, d_status, \
                                                     B, stream)
  CC_GEO(7, 8, 3, true);    // BCH(127,120)
  CC_GEO(14, 7, 2, true);   // BCH(127,113): 98 message registers
  CC_GEO(21, 6, 2, true);   // BCH(127,106): 126 message registers
  CC_GEO(28, 7, 1, false);  // BCH(127,99): 196 message registers, one wave per SIMD
#undef CC_GEO
  return CC_ERR_UNSUPPORTED;
}

}  // namespace ccamd
