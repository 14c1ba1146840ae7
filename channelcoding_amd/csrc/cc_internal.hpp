// cc_internal.hpp -- shared between the C-ABI translation unit and the kernel
// launchers.  Not installed; the public contract is include/channelcoding_amd.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/channelcoding_amd.h"
#include "galois.hpp"

namespace ccamd {

// Geometry of the min-sum kernels: a frame occupies W consecutive lanes of a
// wave64 (W = 16 / 32 / 64), lane `li` of the group owns the columns
// j = li + W*c, c < C.  Parity-check row i touches column j iff row0[j - i] != 0
// (H is banded Toeplitz: cyclic.h:346-359), so connectivity is a per-lane bit
// mask over rows: colmask[(w*C + c)*64 + lane] bit (i & 31), w = i >> 5.
struct MinSumGeometry {
  int W = 64, C = 4, frames_per_wave = 1, KW = 1;
};

struct MinSumParams {
  int n, K, KW;
  int variant;      // cc_algorithm
  int stop_rule;    // cc_stop_rule
  unsigned iterations;
  float alpha_f;    // NMS / 2D-NMS horizontal factor (soft_decision.h:211-213: const R& alpha)
  float beta_f;     // 2D-NMS vertical factor        (soft_decision.h:215-218: const R& beta)
  double beta_d;    // OMS offset, evaluated in double (soft_decision.h:245-251)
  const uint32_t *colmask;  // device
  float *gstate;            // generic kernel, state too large for LDS: per-workgroup slabs in HBM (else nullptr)
  unsigned long long gslab; // floats per workgroup slab
  // Two-pass decoding (minsum_diag.hip: launch_two_pass), all zero otherwise.  ctl: device words [1] frames the first
  // pass handed on, [2] frames of the sample that did not stop, [3] list overflow.  The device's choice: two passes iff
  // ctl[2] * 8 < sample (first pass), and its result is used iff the list did not overflow either (everything after it).
  uint32_t *ctl = nullptr;
  unsigned sample = 0;
  uint32_t *list = nullptr;   // frames handed on
  unsigned list_cap = 0;
  int gate = -1;              // 1: run only if the sample says "two passes"; -1: always
  int first_pass = 0;         // message-free kernel: 1 = a frame that does not stop is appended to list instead of
                              // written; 2 = the same on a sample, counting into ctl[2] only
  // general kernel, launched ONCE per call: if two passes are in effect it decodes the compacted batch (these buffers,
  // ctl[1] frames) instead of the caller's -- one launch per call, whichever branch the device took
  int dual = 0;
  const float *llr2 = nullptr;
  uint8_t *hard2 = nullptr;
  uint16_t *iters2 = nullptr;
  int32_t *status2 = nullptr;
};

#if defined(__HIPCC__)
__device__ __forceinline__ bool two_pass_sampled(const MinSumParams &p) { return p.ctl[2] * 8u < p.sample; }
__device__ __forceinline__ bool two_pass_in_effect(const uint32_t *ctl, unsigned sample, unsigned cap) {
  return ctl[2] * 8u < sample && ctl[3] == 0u && ctl[1] <= cap;
}
#endif

// device-resident tables of one code for the algebraic chain and the encoder
struct AlgebraicTables {
  uint8_t exp[512];
  uint8_t log[512];
  uint8_t g[256];
  uint8_t roots_log[64];
  int n, k, l, t, nroots, family, q;
};

// q = 9..15 (wide.hip): tables in global memory, passed to the kernels by value
struct WideTables {
  const uint16_t *exp = nullptr, *log = nullptr;  // 2 * 2^q entries each
  const uint16_t *g = nullptr;                    // k + 1 coefficients
  uint32_t root_log[64] = {0};
  uint32_t n = 0, k = 0, l = 0, t = 0, nroots = 0, q = 0;
  int family = 0;
};

}  // namespace ccamd

namespace ccamd {
struct McWorkspace;
void mc_workspace_free(McWorkspace *w);
struct HostStage;  // capi.hip: staging of the host-pointer entry points (private streams, grow-only device buffers)
void host_stage_free(HostStage *s);
}  // namespace ccamd

struct cc_code {
  cc_desc desc;
  int device = 0;
  std::unique_ptr<ccamd::Field> field;
  ccamd::CodeTables tab;
  // q = 9..15: symbols are uint16_t; `tab` then only carries the scalars (n, k, l, t, dmin, ...)
  bool wide = false;
  std::unique_ptr<ccamd::FieldT<uint16_t>> field16;
  ccamd::CodeTablesT<uint16_t> tab16;
  ccamd::WideTables wide_dev;
  uint16_t *d_wide = nullptr;  // one allocation behind wide_dev's pointers
  bool soft = false;
  std::vector<uint8_t> custom_H;  // rows x n, empty = the code's own H()
  bool matrix_only = false;  // cc_minsum_create: no field / code tables
  unsigned ms_rows = 0;           // check nodes of the min-sum graph (tab.k unless custom_H)
  ccamd::MinSumGeometry geo;
  uint32_t *d_colmask = nullptr;
  uint16_t *d_diag = nullptr;    // [D][LPF] diagonals (row-0 support) dealt to LPF lanes x D slots (minsum_diag)
  uint32_t *d_colbits = nullptr;  // [256] per column: bit i = H[i][col]
  uint8_t *d_parity = nullptr;  // k x l table of x^(k+j) mod g (division_tag encoder)
  ccamd::AlgebraicTables *d_alg = nullptr;
  ccamd::AlgebraicTables h_alg;
  mutable ccamd::McWorkspace *mc = nullptr;  // lazily allocated Monte-Carlo chunk buffers
  mutable ccamd::HostStage *stage = nullptr;  // lazily allocated staging of the host-pointer entry points
  mutable std::mutex lazy_lock;               // guards the creation of the two above
  // stream-ordered workspace of the multi-pass kernels (bit-plane RS path, large min-sum state, PGZ trials): a pool of
  // the handle's own whose release threshold is "never", so that after the first call of a given size nothing is
  // requested from or returned to the driver (the default pool trims at every synchronisation).  nullptr: default pool.
  hipMemPool_t pool = nullptr;
  int num_cus = 256;
  bool force_generic = false;  // CC_AMD_FORCE_GENERIC=1: A/B the generic kernel against the fast one
  std::string name;
};

namespace ccamd {

inline hipError_t workspace_alloc(const cc_code *code, void **p, size_t bytes, hipStream_t stream) {
  return code->pool ? hipMallocFromPoolAsync(p, bytes, code->pool, stream) : hipMallocAsync(p, bytes, stream);
}

void set_last_error(const std::string &s);
int hip_fail(hipError_t e, const char *what);

#define CC_HIP_TRY(expr)                                   \
  do {                                                     \
    hipError_t _e = (expr);                                \
    if (_e != hipSuccess) return ccamd::hip_fail(_e, #expr); \
  } while (0)

// minsum.hip
int launch_minsum(const cc_code *code, const float *d_llr, const uint16_t *d_er, const uint32_t *d_er_off,
                  uint8_t *d_hard, float *d_L, uint16_t *d_iters, int32_t *d_status, size_t B, hipStream_t stream);
// minsum_diag.hip
struct DiagGeometry {
  unsigned n, k, w;  // code length, rows of H, row weight
  int D, LPF, CPL;   // diagonals per lane, lanes per frame, columns per lane
  bool scms;         // 2 K D registers fit: the self-correcting variants are instantiated too
  int np = 0;        // paired slots: slots 2p and 2p+1 of every lane hold diagonals s and s + gap[p] (one
  int gap[4] = {0, 0, 0, 0};  // two-address LDS instruction serves both); the remaining D - 2 np slots are singles
};
const DiagGeometry *diag_geometry(const CodeTables &t);  // nullptr: no diagonal kernel for this code
std::vector<uint16_t> build_diag_table(const CodeTables &t, int D, int W, int np = 0, const int *gap = nullptr);
size_t minsum_diag_lds_bytes(const DiagGeometry &g);
MinSumParams minsum_params(const cc_code *code);
bool minsum_shortcuts_enabled();
bool minsum_diag_supported(const cc_code *code);
// the general diagonal kernel over a compacted batch whose size only the device knows (ctl[1], at most cap frames);
// ctl: four zeroed device words (MinSumParams::ctl), the producer counts into ctl[1]
int launch_minsum_diag_compact(const cc_code *code, uint32_t *d_ctl, unsigned cap, const float *d_llr, uint8_t *d_hard,
                               uint16_t *d_iters, int32_t *d_status, hipStream_t stream);
std::string minsum_diag_name(const cc_code *code);
int launch_minsum_diag(const cc_code *code, const MinSumParams &p, const float *d_llr, const uint16_t *d_er,
                       const uint32_t *d_er_off, uint8_t *d_hard, float *d_L, uint16_t *d_iters, int32_t *d_status,
                       size_t B, hipStream_t stream);
// algebraic.hip
// algebraic_long.hip
bool algebraic_long_needed(const cc_code *code, bool erasures);
int launch_algebraic_long(const cc_code *code, bool float_in, const void *d_in, const uint16_t *d_er,
                          const uint32_t *d_er_off, uint8_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B,
                          hipStream_t stream);
int launch_algebraic(const cc_code *code, bool float_in, const void *d_in, const uint16_t *d_er,
                     const uint32_t *d_er_off, uint8_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B,
                     hipStream_t stream);
// algebraic_chunk.hip: BM / PGZ without erasures, Berlekamp-Massey with one lane per frame
bool algebraic_chunk_supported(const cc_code *code, bool erasures);
int launch_algebraic_chunk(const cc_code *code, bool float_in, const void *d_in, const uint16_t *d_er,
                           const uint32_t *d_er_off, uint8_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B,
                           hipStream_t stream);
// bitslice.hip: syndromes of GF(2^8) codes on bit planes (32 frames per register)
bool bitslice_supported(const cc_code *code);
int launch_bitslice_syndromes(const cc_code *code, bool float_in, const void *d_in, uint8_t *d_out, uint8_t *d_synd, size_t B,
                              hipStream_t stream);
bool bitslice_encode_supported(const cc_code *code);
int launch_bitslice_encode(const cc_code *code, const uint8_t *d_msg, uint8_t *d_cw, size_t B, hipStream_t stream);
int launch_pgz_erasures(const cc_code *code, const uint8_t *d_in, const uint16_t *d_er, const uint32_t *d_er_off,
                        uint8_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B, hipStream_t stream);
// the object (part) of a geometry that carries a min-sum variant when the geometry is split over `parts` objects
// (minsum_diag_geos.inc, last column): the two self-correcting variants, the slowest to compile, never share a part
constexpr int diag_variant_part(int variant, int parts) {
  const int order = variant == CC_ALG_MS ? 0 : variant == CC_ALG_NMS ? 1 : variant == CC_ALG_OMS ? 2
                  : variant == CC_ALG_SCMS2 ? 3 : variant == CC_ALG_SCMS1 ? 4 : 5;
  return order % parts;
}

// wide.hip
int launch_wide_correct(const cc_code *code, const uint16_t *d_in, const uint16_t *d_er, const uint32_t *d_off,
                        uint16_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B, hipStream_t stream);
int launch_wide_pgz_erasures(const cc_code *code, const uint16_t *d_in, const uint16_t *d_er, const uint32_t *d_off,
                        uint16_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B, hipStream_t stream);
int launch_wide_encode(const cc_code *code, const uint16_t *d_msg, uint16_t *d_cw, size_t B, hipStream_t stream);
int launch_wide_extract(const cc_code *code, const uint16_t *d_cw, uint16_t *d_msg, size_t B, hipStream_t stream);
// encode.hip
std::vector<uint8_t> build_parity_table(const Field &f, const CodeTables &t);
int launch_encode(const cc_code *code, const uint8_t *d_msg, uint8_t *d_cw, size_t B, hipStream_t stream);
int launch_encode_bits(const cc_code *code, const uint8_t *d_msg, uint8_t *d_cw, size_t B, hipStream_t stream);
int launch_extract(const cc_code *code, const uint8_t *d_cw, uint8_t *d_msg, size_t B, hipStream_t stream);
// mc.hip (Monte-Carlo calls on one handle must be issued on one stream at a time: they share a workspace)
int mc_run(cc_code *code, double ebno_db, uint64_t seed, uint64_t first_frame, size_t frames, int random_codewords,
           uint64_t *d_counters, hipStream_t stream);
int mc_awgn(cc_code *code, double ebno_db, uint64_t seed, uint64_t first_frame, size_t frames, int random_codewords,
            float *d_llr, uint8_t *d_sent, hipStream_t stream);
int minsum_kernel_info(const cc_code *code, std::string &name, uint32_t &frames_per_wg, uint32_t &threads,
                       uint32_t &lds);

}  // namespace ccamd
