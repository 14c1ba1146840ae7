// algebraic_long.hip -- the hard-decision chain of algebraic.hip for codes with more than 64 syndromes
// (errors<t> with t > 32: bch.h:28-46, rs.h:18-28 instantiate any t; hard_decision.h:116-155) and for the calls the
// lane-per-coefficient kernels refuse: the Euklid tag with 2t > 63, erasure decoding with 2t > 32.
//
// One codeword per wavefront as in algebraic.hip; what changes is the Berlekamp-Massey phase, where a lane owns FOUR
// coefficients of lambda and b (indices lane + 64 c), so that polynomials of degree up to 255 fit -- every degree a
// GF(2^8) code can ask for (2t <= 254, plus 2t erasures' worth of locator).  The multiplication by x crosses the lane
// blocks (coefficient 64 c comes from lane 63 of block c - 1).  Syndromes, root search at alpha^-p, Forney with one
// lane per located error (four errors per lane now), the re-check and every status class are those of algebraic.hip;
// the coefficient / syndrome / error arrays in LDS hold 256 entries instead of 64.
//
// The Euklid tag is served here as bounded-distance decoding on the Berlekamp-Massey locator (2 deg - rho <= 2t),
// for the reason given at algebraic_chunk_supported: the remainder sequence of hard_decision.h:157-196 ends with a
// locator of degree <= (2t + rho) / 2, a frame decodes exactly when the errors-and-erasures key equation has its
// (unique) solution within that bound, and that solution is the one Berlekamp-Massey finds.
//
// A correctness path, not a throughput path: tests/test_gpu_algebraic.py (RS(255,191), RS(255,127), BCH(255, t = 40),
// Euklid at t = 32, erasures at t = 20 .. 40, all against the oracle).
#include <cstdlib>

#include "cc_internal.hpp"
#include "wave_ops.hpp"

namespace ccamd {
namespace {

struct LongScratch {
  uint8_t S[256];    // syndromes
  uint8_t lam[256];  // lambda coefficients
  uint8_t om[256];   // omega coefficients
  uint8_t rp[256];   // positions of the located errors, in ascending position order
  uint8_t val[256];  // their values
};

__device__ __forceinline__ uint32_t bcast63(uint32_t v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ uint32_t shift_up(uint32_t v) {  // lane j <- lane j-1, lane 0 <- 0
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x138 /* wave_shr:1 */, 0xF, 0xF, true));
}
// p(x) * x on four coefficients per lane (index lane + 64 c)
__device__ __forceinline__ void poly_shift_up(uint32_t (&v)[4], int lane) {
  uint32_t carry = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t top = bcast63(v[c]);
    const uint32_t s = shift_up(v[c]);
    v[c] = lane == 0 ? carry : s;
    carry = top;
  }
}

template <bool FLOAT_IN>
__global__ void __launch_bounds__(256)
algebraic_long_kernel(const AlgebraicTables *__restrict__ T, int alg, const void *__restrict__ in_raw,
                      const uint16_t *__restrict__ er, const uint32_t *__restrict__ er_off, uint8_t *__restrict__ out,
                      int32_t *__restrict__ nerr_out, int32_t *__restrict__ status_out, unsigned long long B) {
  __shared__ uint8_t ex[512];
  __shared__ uint8_t lg[256];
  __shared__ LongScratch scratch[4];
  for (int i = threadIdx.x; i < 512; i += 256) ex[i] = T->exp[i];
  lg[threadIdx.x] = T->log[threadIdx.x];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  LongScratch &W = scratch[wid];
  const int n = T->n, nn = n, t2 = T->nroots;
  const bool is_rs = T->family == CC_FAMILY_RS;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + wid;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;

  auto gmul = [&](uint32_t a, uint32_t b) -> uint32_t { return (a && b) ? ex[lg[a] + lg[b]] : 0u; };
  auto gmul_pow = [&](uint32_t a, uint32_t e) -> uint32_t { return a ? ex[lg[a] + e] : 0u; };

  const int r0 = T->roots_log[0];
  const int step = t2 > 1 ? (T->roots_log[1] + nn - r0) % nn : 0;
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
    uint32_t sym[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int p = lane + 64 * c;
      if (FLOAT_IN)  // hard decision of a signed sequence: cyclic.h:163-173, codes.h:43-52
        sym[c] = valid[c] ? (static_cast<const float *>(in_raw)[frame * n + p] < 0.0f ? 1u : 0u) : 0u;
      else
        sym[c] = valid[c] ? (static_cast<const uint8_t *>(in_raw)[frame * n + p] & static_cast<uint32_t>(n)) : 0u;
    }
    uint32_t nerase = 0, ebase = 0;
    if (er_off != nullptr) {
      ebase = er_off[frame];
      nerase = er_off[frame + 1] - ebase;
    }

    // ---- syndromes, four per DPP reduction (cyclic.h:53-63) ----
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
    } else if (any_syndrome != 0) {  // wave-uniform
      const int rho = static_cast<int>(nerase);
      // ---- Berlekamp-Massey, hard_decision.h:116-155; coefficient lane + 64 c of lambda / b in register c ----
      uint32_t lam[4] = {lane == 0 ? 1u : 0u, 0u, 0u, 0u};
      for (uint32_t e = 0; e < nerase; ++e) {  // lambda *= (1 + alpha^erasure x), :128-131
        const uint32_t X = ex[er[ebase + e] % nn];
        uint32_t sh[4] = {lam[0], lam[1], lam[2], lam[3]};
        poly_shift_up(sh, lane);
#pragma unroll
        for (int c = 0; c < 4; ++c) lam[c] ^= gmul(X, sh[c]);
      }
      uint32_t bpoly[4] = {lam[0], lam[1], lam[2], lam[3]};
      int l = rho;
      for (int i = rho; i < t2; ++i) {
        poly_shift_up(bpoly, lane);  // b = b * x
        uint32_t part = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int m = lane + 64 * c;
          const bool in_sum = m >= 1 && m <= l && m <= i;
          part ^= gmul(lam[c], in_sum ? W.S[i - m] : 0u);
        }
        const uint32_t delta = (bcast63(wave_xor(part)) ^ W.S[i]) & 0xFFu;
        if (delta != 0) {  // wave-uniform
          const bool grow = 2 * l <= i + rho;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const uint32_t tnew = lam[c] ^ gmul(delta, bpoly[c]);
            if (grow) bpoly[c] = lam[c] ? ex[lg[lam[c]] + nn - lg[delta]] : 0u;  // lambda * delta^-1
            lam[c] = tnew;
          }
          if (grow) l = i + rho - l + 1;
        }
      }
      const int bm_len = l;
      int deg = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned long long nz = __ballot(lam[c] != 0);
        if (nz) deg = 64 * c + 63 - __builtin_clzll(nz);
        W.lam[lane + 64 * c] = static_cast<uint8_t>(lam[c]);
      }
      // PGZ and Euklid run as bounded-distance decoding: locator degree within the capability
      if (alg != CC_ALG_BM && 2 * deg - rho > t2) status = CC_FRAME_LOCATOR;
      if (deg < 1) status = CC_FRAME_LOCATOR;  // cyclic.h:145-147

      // ---- root search: position p is in error iff lambda(alpha^-p) = 0 (cyclic.h:126-150) ----
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

      // ---- error values (bch.h:80-83: all ones; rs.h:41-78: the unique solution, here by Forney's formula) ----
      uint32_t yv[4] = {1, 1, 1, 1};
      if (status == CC_FRAME_OK && is_rs) {
        // omega_j = sum_{m<=j} S_{j-m} lambda_m, j < deg  (S(x) lambda(x) mod x^deg); coefficient lane + 64 c
        uint32_t om[4] = {0, 0, 0, 0};
        for (int m = 0; m <= deg; ++m) {
          const uint32_t lm = W.lam[m];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int j = lane + 64 * c;
            const uint32_t s = (j >= m && j < deg && j - m < t2) ? W.S[j - m] : 0u;
            om[c] ^= gmul(lm, s);
          }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          W.om[lane + 64 * c] = static_cast<uint8_t>(om[c]);
          if (isroot[c]) W.rp[rank[c]] = static_cast<uint8_t>(lane + 64 * c);
        }
        // located error number lane + 64 c: numerator omega(X^-1), denominator lambda'(X^-1)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int eidx = lane + 64 * c;
          uint32_t y = 0;
          if (eidx < deg) {
            const uint32_t p = W.rp[eidx];
            const uint32_t xi = p ? static_cast<uint32_t>(nn) - p : 0u;
            const uint32_t x2 = (2 * xi) % static_cast<uint32_t>(nn);
            uint32_t num = 0, den = 0;
            for (int j = deg - 1; j >= 0; --j) num = gmul_pow(num, xi) ^ W.om[j];
            const int mtop = (deg & 1) ? deg : deg - 1;
            for (int m = mtop; m >= 1; m -= 2) den = gmul_pow(den, x2) ^ W.lam[m];
            y = (num && den) ? ex[lg[num] + nn - lg[den]] : 0u;
          }
          W.val[eidx] = static_cast<uint8_t>(y);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) yv[c] = isroot[c] ? W.val[rank[c]] : 0u;
      }

      // ---- verify (cyclic.h:243-248), decided by Berlekamp-Massey's length where it can be: see algebraic.hip ----
      const bool verified_by_bm = rho == 0 && bm_len == deg;
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

    const bool ok = status == CC_FRAME_OK;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (valid[c]) out[frame * n + lane + 64 * c] = static_cast<uint8_t>(sym[c] ^ (ok ? corr[c] : 0u));
    if (lane == 0) {
      if (nerr_out) nerr_out[frame] = ok ? nerr : -1;
      if (status_out) status_out[frame] = status;
    }
    __builtin_amdgcn_wave_barrier();  // the scratch arrays are reused by the next frame
  }
}

}  // namespace

// codes / calls the lane-per-coefficient kernels cannot serve (capi.hip: hard_supported)
bool algebraic_long_needed(const cc_code *code, bool erasures) {
  const size_t t2 = code->tab.roots.size();
  if (t2 > 64) return true;
  if (code->desc.algorithm == CC_ALG_EUKLID && (t2 > 63 || (erasures && t2 > 32))) return true;
  return false;
}

int launch_algebraic_long(const cc_code *code, bool float_in, const void *d_in, const uint16_t *d_er,
                          const uint32_t *d_er_off, uint8_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B,
                          hipStream_t stream) {
  if (B == 0) return CC_OK;
  const unsigned long long blocks_needed = (B + 3) / 4;
  const unsigned long long max_grid = static_cast<unsigned long long>(code->num_cus) * 16;
  const int grid = static_cast<int>(blocks_needed < max_grid ? blocks_needed : max_grid);
  const unsigned long long Bq = B;
  if (float_in)
    hipLaunchKernelGGL(algebraic_long_kernel<true>, dim3(grid), dim3(256), 0, stream, code->d_alg, code->desc.algorithm,
                       d_in, d_er, d_er_off, d_out, d_nerr, d_status, Bq);
  else
    hipLaunchKernelGGL(algebraic_long_kernel<false>, dim3(grid), dim3(256), 0, stream, code->d_alg, code->desc.algorithm,
                       d_in, d_er, d_er_off, d_out, d_nerr, d_status, Bq);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "algebraic long kernel launch");
  return CC_OK;
}

}  // namespace ccamd
