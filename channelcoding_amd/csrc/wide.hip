// wide.hip -- codes over GF(2^q), q = 9 .. 15: symbols are 16 bits wide (the reference's storage_type for those
// fields, src/math/galois.h:44-53), n = 2^q - 1 up to 32767, and the field is built from a modular polynomial the
// caller names (modular_polynomial<>, galois.h:23-25; the reference has defaults for q <= 8 only, :57-67).
//
// Same per-frame chain as algebraic.hip (cyclic::correct_(hard_decision_tag), src/codes/cyclic.h:207-252), one
// codeword per wavefront, but laid out for long words: a frame does not fit the registers of a wave, so every phase
// walks the positions p = lane + 64 c from HBM / L2, and the log / antilog tables (up to 256 KB) stay in global
// memory behind the vector cache instead of LDS.
//   syndromes   S_j = sum_p b_p alpha^(r_j p), four syndromes per pass over the frame       (cyclic.h:53-63)
//   locator     Berlekamp-Massey with erasure pre-load / Euklid (Sugiyama), lane j = coefficient j
//               (hard_decision.h:116-196); the PGZ tag runs as BM + degree bound, as in algebraic.hip
//   roots       lambda(alpha^-p) = 0 by Horner, ranks of the roots from ballots            (cyclic.h:126-159)
//   values      all ones (bch.h:80-83) / Forney for RS (the reference's Gauss elimination, rs.h:41-78, has the
//               same unique solution)
//   re-check    syndromes of the error pattern = received syndromes, one lane per syndrome  (cyclic.h:243-248)
// Encoding (division_tag, cyclic.h:35-40): the remainder of a(x) x^k by g(x) in a k-stage feedback register kept in
// LDS, one message symbol per step, lanes = register stages.  Throughput is not the point of this path (the
// reference itself cannot instantiate a code with q > 8 without an edit); results are pinned like the byte path.
#include "cc_internal.hpp"
#include "wave_ops.hpp"

namespace ccamd {
namespace {

struct WideScratch {
  uint16_t S[64];
  uint16_t lam[72];
  uint16_t om[72];
  uint16_t rp[64];
  uint16_t val[64];
};

__device__ __forceinline__ uint32_t bcast63w(uint32_t v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ uint32_t shift_up_w(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x138 /* wave_shr:1 */, 0xF, 0xF, true));
}

__global__ void __launch_bounds__(256)
wide_correct_kernel(WideTables T, int alg, const uint16_t *__restrict__ in, const uint16_t *__restrict__ er,
                    const uint32_t *__restrict__ er_off, uint16_t *__restrict__ out, int32_t *__restrict__ nerr_out,
                    int32_t *__restrict__ status_out, unsigned long long B) {
  __shared__ WideScratch scratch[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  WideScratch &W = scratch[wid];
  const uint16_t *__restrict__ ex = T.exp;
  const uint16_t *__restrict__ lg = T.log;
  const uint32_t n = T.n, nn = T.n, t2 = T.nroots;
  const bool is_rs = T.family == CC_FAMILY_RS;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + wid;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  auto gmul = [&](uint32_t a, uint32_t b) -> uint32_t { return (a && b) ? ex[lg[a] + lg[b]] : 0u; };
  auto gmul_pow = [&](uint32_t a, uint32_t e) -> uint32_t { return a ? ex[lg[a] + e] : 0u; };  // a alpha^e, e < nn
  const uint32_t r0 = T.root_log[0];
  const uint32_t step = t2 > 1 ? (T.root_log[1] + nn - r0) % nn : 0u;

  for (unsigned long long frame = wave; frame < B; frame += nwaves) {
    const uint16_t *src = in + frame * n;
    uint16_t *dst = out + frame * n;
    uint32_t nerase = 0, ebase = 0;
    if (er_off != nullptr) {
      ebase = er_off[frame];
      nerase = er_off[frame + 1] - ebase;
    }
    // ---- syndromes, four per pass over the frame; the first pass also copies the word out ----
    uint32_t any_syndrome = 0;
    for (uint32_t j0 = 0; j0 < t2; j0 += 4) {
      uint32_t acc[4] = {0, 0, 0, 0};
      for (uint32_t p = lane; p < n; p += 64) {
        const uint32_t b = src[p] & n;
        if (j0 == 0) dst[p] = static_cast<uint16_t>(b);
        if (b) {
          const uint32_t lb = lg[b];
          uint32_t e = static_cast<uint32_t>((static_cast<unsigned long long>(r0 + j0 * step) * p) % nn);
          const uint32_t d = static_cast<uint32_t>((static_cast<unsigned long long>(step) * p) % nn);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            acc[jj] ^= ex[lb + e];
            e += d;
            e = e >= nn ? e - nn : e;
          }
        }
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const uint32_t s = bcast63w(wave_xor(acc[jj])) & 0xFFFFu;
        if (j0 + jj < t2) {
          any_syndrome |= s;
          if (lane == 0) W.S[j0 + jj] = static_cast<uint16_t>(s);
        }
      }
    }

    int status = CC_FRAME_OK, nerr = 0, deg = 0;
    if (any_syndrome != 0 && nerase > t2) {
      status = CC_FRAME_ERASURES;  // bch.h:105-107
    } else if (any_syndrome != 0) {  // wave-uniform
      uint32_t lam;
      const int rho = static_cast<int>(nerase);
      if (alg == CC_ALG_EUKLID) {
        // ---- Euklid / Sugiyama with erasures, hard_decision.h:157-196 (lane j <-> coefficient j) ----
        uint32_t u = (lane == 0) ? 1u : 0u;
        for (uint32_t e = 0; e < nerase; ++e) u ^= gmul(ex[er[ebase + e] % nn], shift_up_w(u));
        uint32_t rp = 0;
        for (int m = 0; m <= rho; ++m) {
          const uint32_t um = __builtin_amdgcn_readlane(u, m);
          const uint32_t sj = (lane >= m && static_cast<uint32_t>(lane - m) < t2) ? W.S[lane - m] : 0u;
          rp ^= gmul(um, sj);
        }
        uint32_t rc = (static_cast<uint32_t>(lane) == t2) ? 1u : 0u, wp = u, wc = 0u;
        const int max_deg = (static_cast<int>(t2) + rho) / 2;
        auto degree_of = [&](uint32_t v) { return 63 - __builtin_clzll(__ballot(v != 0) | 1ull) - ((__ballot(v != 0) == 0) ? 1 : 0); };
        int guard = 0;
        while (degree_of(rc) >= max_deg && guard++ < 130) {
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
        for (uint32_t e = 0; e < nerase; ++e) lam ^= gmul(ex[er[ebase + e] % nn], shift_up_w(lam));  // :128-131
        uint32_t bpoly = lam;
        int l = rho;
        for (int i = rho; i < static_cast<int>(t2); ++i) {
          bpoly = shift_up_w(bpoly);
          const bool in_sum = lane >= 1 && lane <= l && lane <= i;
          const uint32_t sij = in_sum ? W.S[i - lane] : 0u;
          const uint32_t delta = (bcast63w(wave_xor(gmul(lam, sij))) ^ W.S[i]) & 0xFFFFu;
          if (delta != 0) {  // wave-uniform
            const uint32_t tnew = lam ^ gmul(delta, bpoly);
            if (2 * l <= i + rho) {
              bpoly = lam ? ex[lg[lam] + nn - lg[delta]] : 0u;
              l = i + rho - l + 1;
            }
            lam = tnew;
          }
        }
      }
      const unsigned long long nz = __ballot(lam != 0);
      deg = 63 - __builtin_clzll(nz | 1ull);
      W.lam[lane] = static_cast<uint16_t>(lam);
      if (alg == CC_ALG_PGZ && 2 * deg - rho > static_cast<int>(t2)) status = CC_FRAME_LOCATOR;  // bounded distance
      if (deg < 1) status = CC_FRAME_LOCATOR;  // cyclic.h:145-147

      // ---- root search: position p is in error iff lambda(alpha^-p) = 0 ----
      if (status == CC_FRAME_OK) {
        int count = 0;
        const unsigned long long below = (1ull << lane) - 1ull;
        const uint32_t lead = W.lam[deg];
        for (uint32_t base = 0; base < n; base += 64) {  // wave-uniform trip count
          const uint32_t p = base + lane;
          uint32_t acc = 0;
          if (p < n) {
            const uint32_t xi = p ? nn - p : 0u;  // log of X^-1 for X = alpha^p
            acc = lead;
            for (int j = deg - 1; j >= 0; --j) acc = gmul_pow(acc, xi) ^ W.lam[j];
          }
          const bool root = p < n && acc == 0;
          const unsigned long long mk = __ballot(root);
          if (root) {
            const int rank = count + __builtin_popcountll(mk & below);
            if (rank < 64) W.rp[rank] = static_cast<uint16_t>(p);
          }
          count += __builtin_popcountll(mk);
        }
        nerr = count;
        if (count != deg) status = CC_FRAME_LOCATOR;  // cyclic.h:134-143
      }

      // ---- error values: one lane per located error ----
      uint32_t y = 1;  // bch.h:80-83
      if (status == CC_FRAME_OK && is_rs) {
        uint32_t om = 0;  // omega_j = sum_{m<=j} S_{j-m} lambda_m, j < deg
        for (int m = 0; m <= deg; ++m) {
          const uint32_t lm = W.lam[m];
          const uint32_t s = (lane >= m && lane < deg && static_cast<uint32_t>(lane - m) < t2) ? W.S[lane - m] : 0u;
          om ^= gmul(lm, s);
        }
        W.om[lane] = static_cast<uint16_t>(om);
        y = 0;
        if (lane < deg) {
          const uint32_t p = W.rp[lane];
          const uint32_t xi = p ? nn - p : 0u;
          const uint32_t x2 = (2 * xi) % nn;
          uint32_t num = 0, den = 0;
          for (int j = deg - 1; j >= 0; --j) num = gmul_pow(num, xi) ^ W.om[j];
          const int mtop = (deg & 1) ? deg : deg - 1;
          for (int m = mtop; m >= 1; m -= 2) den = gmul_pow(den, x2) ^ W.lam[m];
          y = (num && den) ? ex[lg[num] + nn - lg[den]] : 0u;
        }
      }
      if (status == CC_FRAME_OK) W.val[lane] = static_cast<uint16_t>(lane < deg ? y : 0u);

      // ---- re-check (cyclic.h:243-248): lane j evaluates syndrome j of the error pattern ----
      if (status == CC_FRAME_OK) {
        uint32_t sj = 0;
        if (static_cast<uint32_t>(lane) < t2) {
          const uint32_t rj = (r0 + static_cast<uint32_t>(lane) * step) % nn;
          for (int i = 0; i < deg; ++i) {
            const uint32_t e = static_cast<uint32_t>((static_cast<unsigned long long>(rj) * W.rp[i]) % nn);
            sj ^= gmul_pow(W.val[i], e);
          }
          sj ^= W.S[lane];
        }
        if (__ballot(sj != 0) != 0) status = CC_FRAME_RECHECK;
      }
      // ---- apply (cyclic.h:237-241); the word itself went out with the first syndrome pass ----
      if (status == CC_FRAME_OK && lane < deg) dst[W.rp[lane]] ^= W.val[lane];
    }
    if (lane == 0) {
      if (nerr_out) nerr_out[frame] = status == CC_FRAME_OK ? nerr : -1;
      if (status_out) status_out[frame] = status;
    }
  }
}

// systematic encoder: c = a x^k + (a x^k mod g), parity in coefficients 0..k-1, message in k..n-1
__global__ void __launch_bounds__(256)
wide_encode_kernel(WideTables T, const uint16_t *__restrict__ msg, uint16_t *__restrict__ cw, unsigned long long B) {
  extern __shared__ uint16_t rem_all[];  // 4 waves x (k + 1)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint32_t n = T.n, k = T.k, l = T.l, nn = T.n;
  uint16_t *rem = rem_all + wid * (k + 1);
  const uint16_t *__restrict__ ex = T.exp;
  const uint16_t *__restrict__ lg = T.log;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + wid;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long frame = wave; frame < B; frame += nwaves) {
    const uint16_t *a = msg + frame * l;
    uint16_t *c = cw + frame * n;
    for (uint32_t i = lane; i < k; i += 64) rem[i] = 0;
    for (uint32_t p = lane; p < l; p += 64) c[k + p] = static_cast<uint16_t>(a[p] & nn);
    for (int j = static_cast<int>(l) - 1; j >= 0; --j) {  // g is monic: feedback = a_j + rem[k-1]
      const uint32_t fb = (static_cast<uint32_t>(a[j]) & nn) ^ rem[k - 1];
      const uint32_t lfb = fb ? lg[fb] : 0u;
      // every stage takes its left neighbour's OLD value: within a chunk of 64 stages the reads of the wave
      // complete before its writes (data dependence + in-order LDS), and the chunks go
      // top-down so that rem[i - 1] of the next lower chunk is still the old value
      for (int base = static_cast<int>((k - 1) / 64) * 64; base >= 0; base -= 64) {
        const uint32_t i = static_cast<uint32_t>(base) + lane;
        uint32_t prev = 0, gi = 0;
        if (i < k) {
          prev = i ? rem[i - 1] : 0u;
          gi = T.g[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (i < k) rem[i] = static_cast<uint16_t>(prev ^ ((fb && gi) ? ex[lfb + lg[gi]] : 0u));
        __builtin_amdgcn_wave_barrier();
      }
    }
    for (uint32_t i = lane; i < k; i += 64) c[i] = rem[i];
  }
}

// message extraction for division_tag (cyclic.h:47-51): the top l coefficients
__global__ void __launch_bounds__(256)
wide_extract_kernel(const uint16_t *__restrict__ cw, uint16_t *__restrict__ msg, uint32_t n, uint32_t k,
                    unsigned long long B) {
  const uint32_t l = n - k;
  const unsigned long long total = B * l, stride = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
  for (unsigned long long idx = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += stride) {
    const unsigned long long f = idx / l;
    msg[idx] = cw[f * n + k + (idx - f * l)];
  }
}

}  // namespace

int launch_wide_correct(const cc_code *code, const uint16_t *d_in, const uint16_t *d_er, const uint32_t *d_off,
                        uint16_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B, hipStream_t stream) {
  if (B == 0) return CC_OK;
  if (d_off && code->desc.algorithm == CC_ALG_PGZ)  // BCH only (capi.hip refuses RS): the two-trial rule below
    return launch_wide_pgz_erasures(code, d_in, d_er, d_off, d_out, d_nerr, d_status, B, stream);
  const unsigned long long blocks = (B + 3) / 4, max_grid = static_cast<unsigned long long>(code->num_cus) * 8;
  hipLaunchKernelGGL(wide_correct_kernel, dim3(static_cast<int>(blocks < max_grid ? blocks : max_grid)), dim3(256), 0,
                     stream, code->wide_dev, code->desc.algorithm, d_in, d_er, d_off, d_out, d_nerr, d_status,
                     static_cast<unsigned long long>(B));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CC_OK : hip_fail(e, "wide_correct_kernel launch");
}

// ---- primitive_bch::correct with PGZ and erasures on 16-bit symbols, bch.h:97-149 (width-agnostic there): decode twice
//      with the erased positions forced to 0 and to 1, keep the result with fewer corrected errors (the first wins
//      ties).  Same rule as launch_pgz_erasures of the byte path (algebraic.hip); the trials run without erasures. ----
namespace {
__global__ void __launch_bounds__(256)
wide_force_erasures_kernel(const uint16_t *__restrict__ in, const uint16_t *__restrict__ er,
                           const uint32_t *__restrict__ er_off, uint16_t *__restrict__ in0, uint16_t *__restrict__ in1,
                           uint32_t n, unsigned long long B) {
  const int lane = threadIdx.x & 63;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long f = wave; f < B; f += nwaves) {
    for (uint32_t p = lane; p < n; p += 64) {
      const uint16_t v = in[f * n + p];
      in0[f * n + p] = v;
      in1[f * n + p] = v;
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t e = er_off[f] + lane; e < er_off[f + 1]; e += 64) {  // (the copies above are this wave's own stores)
      in0[f * n + er[e]] = 0;
      in1[f * n + er[e]] = 1;
    }
  }
}
__global__ void __launch_bounds__(256)
wide_select_trial_kernel(const uint16_t *__restrict__ in, const uint32_t *__restrict__ er_off, uint16_t *__restrict__ out0,
                         int32_t *__restrict__ nerr0, int32_t *__restrict__ st0, const uint16_t *__restrict__ out1,
                         const int32_t *__restrict__ nerr1, const int32_t *__restrict__ st1, uint32_t n, uint32_t t2,
                         unsigned long long B) {
  const int lane = threadIdx.x & 63;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long f = wave; f < B; f += nwaves) {
    const uint32_t ne = er_off[f + 1] - er_off[f];
    if (ne == 0) continue;  // trial 0 decoded the untouched word: the plain path (bch.h:100-101)
    const int s0 = st0[f], s1 = st1[f], e0 = nerr0[f], e1 = nerr1[f];
    int pick, status;
    if (ne > t2) {
      pick = -1;
      status = CC_FRAME_ERASURES;  // bch.h:105-107
    } else if (s0 != CC_FRAME_OK && s1 != CC_FRAME_OK) {
      pick = -1;
      status = CC_FRAME_LOCATOR;  // "Erasure decoding failed."
    } else {
      pick = (s0 != CC_FRAME_OK || (s1 == CC_FRAME_OK && e1 < e0)) ? 1 : 0;
      status = CC_FRAME_OK;
    }
    for (uint32_t p = lane; p < n; p += 64) {
      const uint16_t v = pick < 0 ? in[f * n + p] : (pick == 1 ? out1[f * n + p] : out0[f * n + p]);
      out0[f * n + p] = v;
    }
    if (lane == 0) {
      nerr0[f] = pick < 0 ? -1 : (pick == 1 ? e1 : e0);
      st0[f] = status;
    }
  }
}
}  // namespace

int launch_wide_pgz_erasures(const cc_code *code, const uint16_t *d_in, const uint16_t *d_er, const uint32_t *d_off,
                             uint16_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B, hipStream_t stream) {
  if (B == 0) return CC_OK;
  const size_t n = code->tab.n;
  uint16_t *in0 = nullptr;
  int32_t *aux = nullptr;  // nerr0, st0 (when the caller passed none), nerr1, st1
  CC_HIP_TRY(workspace_alloc(code, reinterpret_cast<void **>(&in0), 3 * B * n * sizeof(uint16_t), stream));
  uint16_t *in1 = in0 + B * n, *out1 = in1 + B * n;
  CC_HIP_TRY(workspace_alloc(code, reinterpret_cast<void **>(&aux), 4 * B * sizeof(int32_t), stream));
  int32_t *nerr0 = d_nerr ? d_nerr : aux, *st0 = d_status ? d_status : aux + B, *nerr1 = aux + 2 * B, *st1 = aux + 3 * B;
  const unsigned long long Bq = B;
  const int grid = code->num_cus * 8;
  hipLaunchKernelGGL(wide_force_erasures_kernel, dim3(grid), dim3(256), 0, stream, d_in, d_er, d_off, in0, in1,
                     static_cast<uint32_t>(n), Bq);
  int rc = launch_wide_correct(code, in0, nullptr, nullptr, d_out, nerr0, st0, B, stream);
  if (rc == CC_OK) rc = launch_wide_correct(code, in1, nullptr, nullptr, out1, nerr1, st1, B, stream);
  if (rc == CC_OK) {
    hipLaunchKernelGGL(wide_select_trial_kernel, dim3(grid), dim3(256), 0, stream, d_in, d_off, d_out, nerr0, st0, out1,
                       nerr1, st1, static_cast<uint32_t>(n), code->wide_dev.nroots, Bq);
    if (hipGetLastError() != hipSuccess) rc = CC_ERR_HIP;
  }
  (void)hipFreeAsync(in0, stream);
  (void)hipFreeAsync(aux, stream);
  return rc;
}

int launch_wide_encode(const cc_code *code, const uint16_t *d_msg, uint16_t *d_cw, size_t B, hipStream_t stream) {
  if (B == 0) return CC_OK;
  const unsigned long long blocks = (B + 3) / 4, max_grid = static_cast<unsigned long long>(code->num_cus) * 8;
  const size_t lds = 4 * (static_cast<size_t>(code->wide_dev.k) + 1) * sizeof(uint16_t);
  hipLaunchKernelGGL(wide_encode_kernel, dim3(static_cast<int>(blocks < max_grid ? blocks : max_grid)), dim3(256), lds,
                     stream, code->wide_dev, d_msg, d_cw, static_cast<unsigned long long>(B));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CC_OK : hip_fail(e, "wide_encode_kernel launch");
}

int launch_wide_extract(const cc_code *code, const uint16_t *d_cw, uint16_t *d_msg, size_t B, hipStream_t stream) {
  if (B == 0) return CC_OK;
  hipLaunchKernelGGL(wide_extract_kernel, dim3(code->num_cus * 8), dim3(256), 0, stream, d_cw, d_msg, code->wide_dev.n,
                     code->wide_dev.k, static_cast<unsigned long long>(B));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? CC_OK : hip_fail(e, "wide_extract_kernel launch");
}

}  // namespace ccamd
