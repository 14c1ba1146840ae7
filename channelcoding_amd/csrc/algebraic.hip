// algebraic.hip -- hard-decision decoding chain over GF(2^q), one codeword per
// wavefront, GF log/antilog tables staged in LDS.
//
// Replaces, per frame: cyclic::correct_(hard_decision_tag) src/codes/cyclic.h:207-252
//   syndromes        calculate_syndromes cyclic.h:53-63 (Horner, polynomial.h:273-284)
//   error locator    Berlekamp-Massey with erasure pre-load, hard_decision.h:116-155
//   root search      cyclic::zeroes cyclic.h:126-150 (brute force over the field, polynomial.h:16-28)
//   error values     primitive_bch::error_values bch.h:80-83 (all ones) /
//                    rs::error_values rs.h:41-78 (the reference solves the v x v system by Gauss
//                    elimination; here Forney's formula, which yields the same unique solution)
//   apply + re-check cyclic.h:237-248
//
// Equivalences used (all exact field arithmetic, so results are identical, not approximate):
//   * S_j = sum_p b_p alpha^(r_j p) instead of Horner;
//   * the reference returns lambda reversed, whose roots are the locators X = alpha^pos;
//     X is a root of reverse(lambda) iff lambda(X^-1) = 0, so lambda is evaluated at alpha^(-p);
//   * the re-check "syndromes of the corrected word are all zero" is evaluated as
//     "syndromes of the error pattern equal the received syndromes" (linearity);
//   * the reference's BM reads lambda out of bounds when deg(lambda) < l; lanes
//     beyond the degree hold zero here, which is the textbook algorithm.
//   * the Euklid tag runs Sugiyama's algorithm itself (hard_decision.h:157-196), including erasures;
//   * the PGZ tag decodes to the same word as BM whenever at most t errors occurred; it is run as
//     "BM + degree bound" (bounded-distance decoding), see DESIGN.md for the reference defect this
//     sidesteps (Q9).
//
// Lane roles: in the syndrome / root-search / verify phases lane l owns the positions
// p = l + 64c (c < 4); in the Berlekamp-Massey phase lane j owns coefficient j of lambda and b.
#include <cstdlib>

#include "cc_internal.hpp"
#include "wave_ops.hpp"

namespace ccamd {
namespace {

#ifndef CC_ALG_BM_SHORTCUT
#define CC_ALG_BM_SHORTCUT 1
#endif
constexpr bool kBmShortcut = CC_ALG_BM_SHORTCUT != 0;

struct WaveScratch {
  uint8_t S[64];    // syndromes
  uint8_t lam[72];  // lambda coefficients
  uint8_t om[72];   // omega coefficients
  uint8_t rp[64];   // positions of the located errors, in ascending position order
  uint8_t val[64];  // their values
};

__device__ __forceinline__ uint32_t bcast63(uint32_t v) { return __builtin_amdgcn_readlane(v, 63); }
// lane j <- lane j-1, lane 0 <- 0   (multiplication of a polynomial by x)
__device__ __forceinline__ uint32_t shift_up(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x138 /* wave_shr:1 */, 0xF, 0xF, true));
}

template <bool FLOAT_IN>
__global__ void __launch_bounds__(256)
algebraic_kernel(const AlgebraicTables *__restrict__ T, int alg, const void *__restrict__ in_raw,
                 const uint16_t *__restrict__ er, const uint32_t *__restrict__ er_off, uint8_t *__restrict__ out,
                 int32_t *__restrict__ nerr_out, int32_t *__restrict__ status_out, unsigned long long B) {
  __shared__ uint8_t ex[512];
  __shared__ uint8_t lg[256];
  __shared__ WaveScratch scratch[4];
  for (int i = threadIdx.x; i < 512; i += 256) ex[i] = T->exp[i];
  lg[threadIdx.x] = T->log[threadIdx.x];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  WaveScratch &W = scratch[wid];
  const int dbg_stop = (alg >> 8) & 0xFF;  // timing experiments only (CC_AMD_ALG_STOP): 1 after syndromes, 2 after BM, 3 after roots
  const bool redo = (alg >> 16) & 1;       // only the frames another path left with a non-zero status (launch_algebraic)
  alg &= 0xFF;
  const int n = T->n, nroots = T->nroots, nn = n;  // full-length codes: n = 2^q - 1
  const int t2 = nroots;
  const bool is_rs = T->family == CC_FAMILY_RS;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + wid;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;

  auto gmul = [&](uint32_t a, uint32_t b) -> uint32_t { return (a && b) ? ex[lg[a] + lg[b]] : 0u; };
  // a * alpha^e, 0 <= e < nn
  auto gmul_pow = [&](uint32_t a, uint32_t e) -> uint32_t { return a ? ex[lg[a] + e] : 0u; };

  // per-lane exponent bookkeeping: e0[c] = r_0 * p mod nn, d[c] = step * p mod nn (roots are alpha^(r_0 + j*step))
  const int r0 = T->roots_log[0];
  const int step = nroots > 1 ? (T->roots_log[1] + nn - r0) % nn : 0;
  uint32_t e0[4], dstep[4], xinv[4];
  bool valid[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int p = lane + 64 * c;
    valid[c] = p < n;
    e0[c] = static_cast<uint32_t>((r0 * p) % nn);
    dstep[c] = static_cast<uint32_t>((step * p) % nn);
    xinv[c] = static_cast<uint32_t>((nn - (p % nn)) % nn);  // log of X^-1 for X = alpha^p
  }

  for (unsigned long long frame = wave; frame < B; frame += nwaves) {
    if (redo && status_out[frame] == CC_FRAME_OK) continue;  // wave-uniform
    // ---- load (hard decision of a signed sequence: cyclic.h:163-173, codes.h:43-52) ----
    uint32_t sym[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int p = lane + 64 * c;
      if (FLOAT_IN)
        sym[c] = valid[c] ? (static_cast<const float *>(in_raw)[frame * n + p] < 0.0f ? 1u : 0u) : 0u;
      else
        sym[c] = valid[c] ? (static_cast<const uint8_t *>(in_raw)[frame * n + p] & static_cast<uint32_t>(n)) : 0u;
    }
    uint32_t nerase = 0, ebase = 0;
    if (er_off != nullptr) {
      ebase = er_off[frame];
      nerase = er_off[frame + 1] - ebase;
    }

    // ---- syndromes, four per DPP reduction ----
    uint32_t lsym[4], ecur[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      lsym[c] = lg[sym[c]];
      ecur[c] = e0[c];
    }
    uint32_t any_syndrome = 0;
    for (int j0 = 0; j0 < t2; j0 += 4) {
      uint32_t packed = 0;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        uint32_t term = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          term ^= sym[c] ? ex[lsym[c] + ecur[c]] : 0u;
          ecur[c] += dstep[c];
          ecur[c] = ecur[c] >= static_cast<uint32_t>(nn) ? ecur[c] - nn : ecur[c];
        }
        packed |= (j0 + jj < t2 ? term : 0u) << (8 * jj);
      }
      packed = bcast63(wave_xor(packed));
      any_syndrome |= packed;
      if (lane < 4 && j0 + lane < t2) W.S[j0 + lane] = static_cast<uint8_t>(packed >> (8 * lane));
    }

    int status = CC_FRAME_OK;
    int nerr = 0;
    uint32_t corr[4] = {0, 0, 0, 0};
    if (any_syndrome != 0 && nerase > static_cast<uint32_t>(t2)) {
      status = CC_FRAME_ERASURES;  // more erasures than 2t cannot be located (bch.h:105-107)
    } else if (any_syndrome != 0 && dbg_stop != 1) {  // wave-uniform
      uint32_t lam;
      int bm_len = -1;  // LFSR length L of Berlekamp-Massey (errors only), -1 otherwise
      const int rho = static_cast<int>(nerase);
      if (alg == CC_ALG_EUKLID) {
        // ---- Euklid / Sugiyama with erasures, hard_decision.h:157-196 (lane j <-> coefficient j) ----
        // r_prev = S(x) u(x), r_cur = x^2t, w_prev = u, w_cur = 0; divide until deg r_cur < (2t + rho) / 2;
        // lambda = w_cur / w_cur(0).  One long-division step per loop trip, at most 2t + rho trips.
        uint32_t u = (lane == 0) ? 1u : 0u;
        for (uint32_t e = 0; e < nerase; ++e) u ^= gmul(ex[er[ebase + e] % nn], shift_up(u));  // :171-172
        uint32_t rp = 0;  // S(x) * u(x): coefficient j = sum_m S_{j-m} u_m
        for (int m = 0; m <= rho; ++m) {
          const uint32_t um = __builtin_amdgcn_readlane(u, m);
          const uint32_t sj = (lane >= m && lane - m < t2) ? W.S[lane - m] : 0u;
          rp ^= gmul(um, sj);
        }
        uint32_t rc = (lane == t2) ? 1u : 0u, wp = u, wc = 0u;
        const int max_deg = (t2 + rho) / 2;
        auto degree_of = [&](uint32_t v) { return 63 - __builtin_clzll(__ballot(v != 0) | 1ull) - ((__ballot(v != 0) == 0) ? 1 : 0); };
        int guard = 0;
        while (degree_of(rc) >= max_deg && guard++ < 130) {
          // one Euclid step: (q, next) = divmod(rp, rc); w_next = wp + q * wc
          const int dr = degree_of(rc);
          const uint32_t lead = __builtin_amdgcn_readlane(rc, dr);
          uint32_t rem = rp, wn = wp;
          for (int pos = degree_of(rem); pos >= dr; --pos) {
            const uint32_t top = __builtin_amdgcn_readlane(rem, pos);
            if (top == 0) continue;
            const uint32_t coef = ex[lg[top] + nn - lg[lead]];
            const int sh = pos - dr;
            const uint32_t rc_sh = __shfl(rc, lane - sh, 64), wc_sh = __shfl(wc, lane - sh, 64);
            rem ^= (lane >= sh) ? gmul(coef, rc_sh) : 0u;
            wn ^= (lane >= sh) ? gmul(coef, wc_sh) : 0u;
          }
          rp = rc;
          rc = rem;
          wp = wc;
          wc = wn;
        }
        const uint32_t w0 = __builtin_amdgcn_readlane(wc, 0);
        if (w0 == 0) status = CC_FRAME_LOCATOR;  // "Cannot invert last element", :191-192
        lam = (w0 && wc) ? ex[lg[wc] + nn - lg[w0]] : 0u;
      } else {
      // ---- Berlekamp-Massey, hard_decision.h:116-155 (lane j <-> coefficient j) ----
      lam = (lane == 0) ? 1u : 0u;
      for (uint32_t e = 0; e < nerase; ++e) {  // lambda *= (1 + alpha^erasure x), :128-131
        const uint32_t X = ex[er[ebase + e] % nn];
        lam ^= gmul(X, shift_up(lam));
      }
      uint32_t bpoly = lam;
      int l = static_cast<int>(nerase);
      bm_len = -2;  // set below
      for (int i = rho; i < t2; ++i) {
        bpoly = shift_up(bpoly);  // b = b * x
        const bool in_sum = lane >= 1 && lane <= l && lane <= i;
        const uint32_t sij = in_sum ? W.S[i - lane] : 0u;
        const uint32_t delta = (bcast63(wave_xor(gmul(lam, sij))) ^ W.S[i]) & 0xFFu;
        if (delta != 0) {  // wave-uniform
          const uint32_t tnew = lam ^ gmul(delta, bpoly);
          if (2 * l <= i + rho) {
            bpoly = lam ? ex[lg[lam] + nn - lg[delta]] : 0u;  // lambda * delta^-1
            l = i + rho - l + 1;
          }
          lam = tnew;
        }
      }
      bm_len = l;
      }
      const unsigned long long nz = __ballot(lam != 0);
      const int deg = 63 - __builtin_clzll(nz | 1ull);
      W.lam[lane] = static_cast<uint8_t>(lam);  // 64 coefficients
      // the PGZ tag runs as bounded-distance decoding: locator degree within capability
      if (alg == CC_ALG_PGZ && 2 * deg - rho > t2) status = CC_FRAME_LOCATOR;
      if (deg < 1) status = CC_FRAME_LOCATOR;  // cyclic.h:145-147
      if (dbg_stop == 2) status = CC_FRAME_LOCATOR;

      // ---- root search: position p is in error iff lambda(alpha^-p) = 0 ----
      uint32_t isroot[4] = {0, 0, 0, 0}, rank[4] = {0, 0, 0, 0};
      if (status == CC_FRAME_OK) {
        uint32_t acc[4];
        const uint32_t lead = W.lam[deg];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = lead;
        for (int j = deg - 1; j >= 0; --j) {
          const uint32_t lj = W.lam[j];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = gmul_pow(acc[c], xinv[c]) ^ lj;
        }
        int count = 0;
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          isroot[c] = (valid[c] && acc[c] == 0) ? 1u : 0u;
          const unsigned long long mk = __ballot(isroot[c] != 0);
          rank[c] = static_cast<uint32_t>(count + __builtin_popcountll(mk & below));
          count += __builtin_popcountll(mk);
        }
        nerr = count;
        if (count != deg) status = CC_FRAME_LOCATOR;  // cyclic.h:134-143
      }

      if (dbg_stop == 3) status = CC_FRAME_LOCATOR;
      // ---- error values ----
      uint32_t yv[4] = {1, 1, 1, 1};  // bch.h:80-83
      if (status == CC_FRAME_OK && is_rs) {
        // omega_j = sum_{m<=j} S_{j-m} lambda_m, j < deg  (S(x) lambda(x) mod x^deg)
        uint32_t om = 0;
        for (int m = 0; m <= deg; ++m) {
          const uint32_t lm = W.lam[m];
          const uint32_t s = (lane >= m && lane < deg && lane - m < t2) ? W.S[lane - m] : 0u;
          om ^= gmul(lm, s);
        }
        W.om[lane] = static_cast<uint8_t>(om);
        // one lane per located error (ranks from the ballots of the root search) instead of a Horner chain over
        // every position: numerator omega(X^-1), denominator lambda'(X^-1) = sum_{m odd} lambda_m X^-(m-1)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (isroot[c]) W.rp[rank[c]] = static_cast<uint8_t>(lane + 64 * c);
        uint32_t y = 0;
        if (lane < deg) {
          const uint32_t p = W.rp[lane];
          const uint32_t xi = p ? static_cast<uint32_t>(nn) - p : 0u;
          const uint32_t x2 = (2 * xi) % static_cast<uint32_t>(nn);
          uint32_t num = 0, den = 0;
          for (int j = deg - 1; j >= 0; --j) num = gmul_pow(num, xi) ^ W.om[j];
          const int mtop = (deg & 1) ? deg : deg - 1;
          for (int m = mtop; m >= 1; m -= 2) den = gmul_pow(den, x2) ^ W.lam[m];
          y = (num && den) ? ex[lg[num] + nn - lg[den]] : 0u;
        }
        W.val[lane] = static_cast<uint8_t>(y);
#pragma unroll
        for (int c = 0; c < 4; ++c) yv[c] = isroot[c] ? W.val[rank[c]] : 0u;
      }

      // ---- verify: syndromes of the error pattern must equal the received syndromes (cyclic.h:243-248) ----
      // Without erasures the test is decided by BM's own bookkeeping: lambda generates S_1..S_2t as an LFSR of
      // length L.  If L = deg lambda and lambda has L distinct roots X_i^-1 (checked above), the values Y_i that
      // solve the first L syndrome equations (Forney; all ones for a binary word, since S_2j = S_j^2 forces
      // Y_i^2 = Y_i and Y_i = 0 would contradict the minimality of L) reproduce all 2t syndromes, because both
      // sequences obey the same recurrence from the same L initial values: the re-check cannot fail.  If
      // L != deg lambda nothing is known and the syndromes of the error pattern are evaluated.
      const bool verified_by_bm = kBmShortcut && rho == 0 && bm_len == deg;
      if (status == CC_FRAME_OK)
#pragma unroll
        for (int c = 0; c < 4; ++c) corr[c] = isroot[c] ? yv[c] : 0u;
      if (status == CC_FRAME_OK && !verified_by_bm) {
        uint32_t ly[4], ev[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          ly[c] = lg[corr[c]];
          ev[c] = e0[c];
        }
        uint32_t mismatch = 0;
        for (int j0 = 0; j0 < t2; j0 += 4) {
          uint32_t packed = 0, want = 0;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            uint32_t term = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              term ^= corr[c] ? ex[ly[c] + ev[c]] : 0u;
              ev[c] += dstep[c];
              ev[c] = ev[c] >= static_cast<uint32_t>(nn) ? ev[c] - nn : ev[c];
            }
            if (j0 + jj < t2) {
              packed |= term << (8 * jj);
              want |= static_cast<uint32_t>(W.S[j0 + jj]) << (8 * jj);
            }
          }
          mismatch |= bcast63(wave_xor(packed)) ^ want;
        }
        if (mismatch != 0) status = CC_FRAME_RECHECK;
      }
    }

    // ---- store ----
    const bool ok = status == CC_FRAME_OK;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (valid[c]) out[frame * n + lane + 64 * c] = static_cast<uint8_t>(sym[c] ^ (ok ? corr[c] : 0u));
    if (lane == 0) {
      if (nerr_out) nerr_out[frame] = ok ? nerr : -1;
      if (status_out) status_out[frame] = status;
    }
  }
}

// ---- primitive_bch::correct with PGZ and erasures, bch.h:97-149: decode twice with the erased positions
//      forced to 0 and to 1, keep the result with fewer corrected errors (the first wins ties) ----
__global__ void __launch_bounds__(256)
force_erasures_kernel(const uint8_t *__restrict__ in, const uint16_t *__restrict__ er, const uint32_t *__restrict__ er_off,
                      uint8_t *__restrict__ in0, uint8_t *__restrict__ in1, int n, unsigned long long B) {
  const unsigned long long total = B * static_cast<unsigned long long>(n);
  const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
  for (unsigned long long idx = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += stride) {
    const unsigned long long f = idx / n;
    const int p = static_cast<int>(idx - f * n);
    bool erased = false;
    for (uint32_t e = er_off[f]; e < er_off[f + 1]; ++e) erased |= (er[e] == p);
    const uint8_t v = in[idx];
    in0[idx] = erased ? 0 : v;
    in1[idx] = erased ? 1 : v;
  }
}

__global__ void __launch_bounds__(256)
select_trial_kernel(const uint8_t *__restrict__ in, const uint32_t *__restrict__ er_off, uint8_t *__restrict__ out0,
                    int32_t *__restrict__ nerr0, int32_t *__restrict__ st0, const uint8_t *__restrict__ out1,
                    const int32_t *__restrict__ nerr1, const int32_t *__restrict__ st1, int n, int t2,
                    unsigned long long B) {
  const int lane = threadIdx.x & 63;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long f = wave; f < B; f += nwaves) {
    const uint32_t ne = er_off[f + 1] - er_off[f];
    if (ne == 0) continue;  // trial 0 decoded the untouched word: the plain path (bch.h:100-101)
    const int s0 = st0[f], s1 = st1[f], e0 = nerr0[f], e1 = nerr1[f];
    int pick, status;
    if (ne > static_cast<uint32_t>(t2)) {
      pick = -1;
      status = CC_FRAME_ERASURES;  // bch.h:105-107
    } else if (s0 != CC_FRAME_OK && s1 != CC_FRAME_OK) {
      pick = -1;
      status = CC_FRAME_LOCATOR;  // "Erasure decoding failed."
    } else {
      pick = (s0 != CC_FRAME_OK || (s1 == CC_FRAME_OK && e1 < e0)) ? 1 : 0;
      status = CC_FRAME_OK;
    }
    for (int p = lane; p < n; p += 64) {
      const uint8_t v = pick < 0 ? in[f * n + p] : (pick == 1 ? out1[f * n + p] : out0[f * n + p]);
      out0[f * n + p] = v;
    }
    if (lane == 0) {
      if (nerr0) nerr0[f] = pick < 0 ? -1 : (pick == 1 ? e1 : e0);
      if (st0) st0[f] = status;
    }
  }
}

}  // namespace

int launch_algebraic(const cc_code *code, bool float_in, const void *d_in, const uint16_t *d_er,
                     const uint32_t *d_er_off, uint8_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B,
                     hipStream_t stream);

int launch_pgz_erasures(const cc_code *code, const uint8_t *d_in, const uint16_t *d_er, const uint32_t *d_er_off,
                        uint8_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B, hipStream_t stream) {
  if (B == 0) return CC_OK;
  const size_t n = code->tab.n;
  uint8_t *in0 = nullptr, *in1 = nullptr, *out1 = nullptr;
  int32_t *aux = nullptr;  // nerr0, st0 (when the caller passed none), nerr1, st1
  CC_HIP_TRY(workspace_alloc(code, reinterpret_cast<void **>(&in0), 3 * B * n, stream));
  in1 = in0 + B * n;
  out1 = in1 + B * n;
  CC_HIP_TRY(workspace_alloc(code, reinterpret_cast<void **>(&aux), 4 * B * sizeof(int32_t), stream));
  int32_t *nerr0 = d_nerr ? d_nerr : aux, *st0 = d_status ? d_status : aux + B, *nerr1 = aux + 2 * B, *st1 = aux + 3 * B;
  const unsigned long long Bq = B;
  const int grid = code->num_cus * 8;
  hipLaunchKernelGGL(force_erasures_kernel, dim3(grid), dim3(256), 0, stream, d_in, d_er, d_er_off, in0, in1,
                     static_cast<int>(n), Bq);
  int rc = launch_algebraic(code, false, in0, nullptr, nullptr, d_out, nerr0, st0, B, stream);
  if (rc == CC_OK) rc = launch_algebraic(code, false, in1, nullptr, nullptr, out1, nerr1, st1, B, stream);
  if (rc == CC_OK) {
    hipLaunchKernelGGL(select_trial_kernel, dim3(grid), dim3(256), 0, stream, d_in, d_er_off, d_out, nerr0, st0, out1,
                       nerr1, st1, static_cast<int>(n), static_cast<int>(code->tab.roots.size()), Bq);
    if (hipGetLastError() != hipSuccess) rc = CC_ERR_HIP;
  }
  (void)hipFreeAsync(in0, stream);
  (void)hipFreeAsync(aux, stream);
  return rc;
}

int launch_algebraic(const cc_code *code, bool float_in, const void *d_in, const uint16_t *d_er,
                     const uint32_t *d_er_off, uint8_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B,
                     hipStream_t stream) {
  if (B == 0) return CC_OK;
  if (algebraic_long_needed(code, d_er_off != nullptr))
    return launch_algebraic_long(code, float_in, d_in, d_er, d_er_off, d_out, d_nerr, d_status, B, stream);
  // The bit-plane chain is seven launches with a floor of 60 .. 100 us per call; below ~4e5 frame-syndromes one
  // wavefront per frame is faster (profiles/tools/hard_size_sweep.py: RS(255,223) 2^12 frames 32 vs 102 us,
  // BCH(255,231) 2^14 frames 24 vs 65 us; equal at 2^14 / 2^16 frames)
  static const size_t planes_min_work = [] {
    const char *e = std::getenv("CC_AMD_PLANES_MIN_WORK");
    return e ? static_cast<size_t>(std::strtoull(e, nullptr, 10)) : static_cast<size_t>(3) << 17;
  }();
  const bool small_call = bitslice_supported(code) && B * code->tab.roots.size() < planes_min_work;
  if (algebraic_chunk_supported(code, d_er_off != nullptr) && !small_call)
    return launch_algebraic_chunk(code, float_in, d_in, d_er, d_er_off, d_out, d_nerr, d_status, B, stream);
  // The Euklid tag WITH erasures on a bit-plane code: the chain first, as bounded-distance Berlekamp-Massey -- a frame
  // it corrects lies within the capability (2e + rho <= 2t), where the key equation has one solution and Sugiyama's
  // remainder sequence finds the same one -- then Sugiyama itself (the kernel below, in place on the output) over the
  // frames the chain left with a non-zero status: the hopeless ones, and the ones hard_decision.h:176's integer stop
  // rule lets the reference decode beyond the capability when rho is odd (E39).
  const bool chain_first = d_er_off != nullptr && code->desc.algorithm == CC_ALG_EUKLID && bitslice_supported(code) && !small_call;
  int32_t *st_tmp = nullptr;
  if (chain_first) {
    if (d_status == nullptr) {
      CC_HIP_TRY(workspace_alloc(code, reinterpret_cast<void **>(&st_tmp), B * sizeof(int32_t), stream));
      d_status = st_tmp;
    }
    const int rc = launch_algebraic_chunk(code, float_in, d_in, d_er, d_er_off, d_out, d_nerr, d_status, B, stream);
    if (rc != CC_OK) {
      if (st_tmp) (void)hipFreeAsync(st_tmp, stream);
      return rc;
    }
    d_in = d_out;  // (failed frames hold the received word, hard-decided)
    float_in = false;
  }
  const unsigned long long blocks_needed = (B + 3) / 4;
  const unsigned long long max_grid = static_cast<unsigned long long>(code->num_cus) * 16;
  const int grid = static_cast<int>(blocks_needed < max_grid ? blocks_needed : max_grid);
  const unsigned long long Bq = B;
#ifdef CC_AMD_EXPERIMENTS  // phase timing (profiles/tools/rs_bench.py); the product library always runs the whole chain
  static const int dbg_stop = [] {
    const char *e = std::getenv("CC_AMD_ALG_STOP");
    return e ? std::atoi(e) : 0;
  }();
#else
  const int dbg_stop = 0;
#endif
  // The Euklid tag without erasures runs as bounded-distance Berlekamp-Massey (the kernel's PGZ branch), as on the
  // chunk / plane / long paths: the same corrected words and the same failing frames as the remainder sequence of
  // hard_decision.h:157-196 (argument in algebraic_chunk_supported, algebraic_chunk.hip); Sugiyama itself runs where
  // the erasure locator enters the start polynomials.
  const int alg_eff = (code->desc.algorithm == CC_ALG_EUKLID && d_er_off == nullptr) ? CC_ALG_PGZ : code->desc.algorithm;
  const int alg_arg = alg_eff | (dbg_stop << 8) | (chain_first ? 1 << 16 : 0);
  if (float_in)
    hipLaunchKernelGGL(algebraic_kernel<true>, dim3(grid), dim3(256), 0, stream, code->d_alg, alg_arg, d_in,
                       d_er, d_er_off, d_out, d_nerr, d_status, Bq);
  else
    hipLaunchKernelGGL(algebraic_kernel<false>, dim3(grid), dim3(256), 0, stream, code->d_alg, alg_arg,
                       d_in, d_er, d_er_off, d_out, d_nerr, d_status, Bq);
  hipError_t e = hipGetLastError();
  if (st_tmp) (void)hipFreeAsync(st_tmp, stream);
  if (e != hipSuccess) return hip_fail(e, "algebraic kernel launch");
  return CC_OK;
}

}  // namespace ccamd
