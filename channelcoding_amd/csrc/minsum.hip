// minsum.hip -- min-sum belief propagation over the banded-Toeplitz BCH
// parity-check matrix, hand-written for gfx950 (wave64).
//
// Replaces, per frame: cyclic::correct_(soft_decision_tag) src/codes/cyclic.h:254-267
//   -> min_sum__ src/codes/soft_decision.h:161-202 with vertical__ :125-140,
//   column_sum :86-98, horizontal__ :101-122 and the variant functors :204-295.
//
// Numerics contract (bit-exact hard decisions, L within 1e-5 -- in practice
// bit-exact -- against the reference):
//   * column sums are accumulated in ascending row order starting from +0.0f
//     (soft_decision.h:88-95): each lane owns whole columns and walks the rows
//     sequentially, so no tree / shuffle reduction ever touches a float sum;
//   * extrinsic = cs - r, then + y (two roundings, :135-136,:207-209);
//     L = cs + y (:180-182); compiled with -ffp-contract=off;
//   * the O(w^2) "minimum over all other edges" of horizontal__ equals
//     (|q| == min1 ? min2 : min1) with min2 the second smallest counting
//     multiplicity; the exclusive sign is 0 if any OTHER message is zero
//     (signum(0) = 0, :75-77), else the parity of the other negatives;
//   * h(.) (alpha*min, max(min-beta,0) in double) is applied to the two
//     row-uniform candidates, exactly as `sign * fn(min)` does per edge.
#include "cc_internal.hpp"

namespace ccamd {
namespace {

constexpr float kFltMax = 3.402823466e+38f;  // std::numeric_limits<float>::max(), soft_decision.h:110

__device__ __forceinline__ int signum(float v) { return (0.0f < v) - (v < 0.0f); }

// vertical functor: q = fn(cs - r, y, q_old)
template <int VARIANT>
__device__ __forceinline__ float vertical(float e, float y, float q_old, float beta_f) {
  if constexpr (VARIANT == CC_ALG_SCMS1) {  // soft_decision.h:261-266
    const float tmp = e + y;
    const int so = signum(q_old);
    return (so == 0 || so == signum(tmp)) ? tmp : 0.0f;
  } else if constexpr (VARIANT == CC_ALG_SCMS2) {  // :275-280
    const float tmp = e + y;
    return (tmp * q_old > 0.0f) ? tmp : 0.5f * (tmp + q_old);
  } else if constexpr (VARIANT == CC_ALG_2DNMS) {  // :215-218  beta * arg + y, two roundings
    const float scaled = __fmul_rn(beta_f, e);
    return __fadd_rn(scaled, y);
  } else {  // :205-209
    return e + y;
  }
}

// horizontal functor applied to an exclusive minimum (row-uniform value)
template <int VARIANT>
__device__ __forceinline__ float horizontal(float m, float alpha_f, double beta_d) {
  if constexpr (VARIANT == CC_ALG_NMS || VARIANT == CC_ALG_2DNMS) {  // :211-213
    return __fmul_rn(alpha_f, m);
  } else if constexpr (VARIANT == CC_ALG_OMS) {  // :245-251 std::max(min - beta, 0.0) in double
    const double a = static_cast<double>(m) - beta_d;
    return static_cast<float>((a < 0.0) ? 0.0 : a);
  } else {
    return m;
  }
}

// ---------------------------------------------------------------------------
// Generic kernel: any k (rows walked by a run-time loop), per-edge state r
// (and q for the self-correcting variants) in LDS, one wave per workgroup,
// 64/W frames per wave.  Correct for every supported code; the specialised
// register-resident kernels below take over for the benchmark geometries.
// GSTATE: the k x n message state exceeds the 160 KiB of LDS (long low-rate
// codes, large caller-supplied matrices) and lives in a per-workgroup HBM slab
// instead -- same addressing (64 consecutive floats per access), L2-resident.
// ---------------------------------------------------------------------------
template <int C, int W, int VARIANT, bool GSTATE>
__global__ void __launch_bounds__(64)
minsum_generic_kernel(MinSumParams p, const float *__restrict__ llr, const uint16_t *__restrict__ er,
                      const uint32_t *__restrict__ er_off, uint8_t *__restrict__ hard, float *__restrict__ Lout,
                      uint16_t *__restrict__ iters_out, int32_t *__restrict__ status_out, unsigned long long B) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int FPW = 64 / W;
  constexpr bool NEEDQ = (VARIANT == CC_ALG_SCMS1 || VARIANT == CC_ALG_SCMS2);
  const int lane = threadIdx.x;
  const int li = lane & (W - 1);
  const int sub = lane / W;
  const int n = p.n, K = p.K;
  float *R = GSTATE ? p.gstate + static_cast<size_t>(blockIdx.x) * p.gslab : lds;
  float *Q = R + static_cast<size_t>(K) * C * 64;
  const unsigned long long group_mask = (W == 64) ? ~0ull : ((1ull << W) - 1ull);
  const int group_shift = sub * W;

  const unsigned long long ngroups = (B + FPW - 1) / FPW;
  for (unsigned long long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const unsigned long long frame = g * FPW + sub;
    const bool active = frame < B;
    float y[C], cs[C], Lv[C];
    bool colv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int j = li + W * c;
      colv[c] = active && j < n;
      // + 0.0f maps an input of -0.0f to +0.0f; every later use is value-identical
      y[c] = colv[c] ? (llr[frame * n + j] + 0.0f) : 0.0f;
      cs[c] = 0.0f;
      Lv[c] = 0.0f;
    }
    if (er_off != nullptr && active) {  // cyclic.h:259-262: LLR at an erasure := 0
      for (uint32_t e = er_off[frame]; e < er_off[frame + 1]; ++e) {
        const int pos = er[e];
#pragma unroll
        for (int c = 0; c < C; ++c)
          if (pos == li + W * c) y[c] = 0.0f;
      }
    }
    for (int s = 0; s < K * C; ++s) {
      R[s * 64 + lane] = 0.0f;
      if (NEEDQ) Q[s * 64 + lane] = 0.0f;
    }

    bool done = !active;
    unsigned my_iter = p.iterations;
    for (unsigned it = 0; it < p.iterations; ++it) {
      float csn[C];
#pragma unroll
      for (int c = 0; c < C; ++c) csn[c] = 0.0f;
      uint32_t cm[C];
      for (int i = 0; i < K; ++i) {
        if ((i & 31) == 0) {
#pragma unroll
          for (int c = 0; c < C; ++c) cm[c] = p.colmask[((i >> 5) * C + c) * 64 + lane];
        }
        float q[C];
        bool edge[C];
        float m1 = kFltMax, m2 = kFltMax;
        unsigned nneg = 0, nzero = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          edge[c] = colv[c] && ((cm[c] >> (i & 31)) & 1u);
          const float r_old = R[(i * C + c) * 64 + lane];
          const float q_old = NEEDQ ? Q[(i * C + c) * 64 + lane] : 0.0f;
          const float e = cs[c] - r_old;
          q[c] = vertical<VARIANT>(e, y[c], q_old, p.beta_f);
          const float a = edge[c] ? fabsf(q[c]) : kFltMax;
          // insert a into the sorted pair (m1 <= m2)
          m2 = fminf(m2, fmaxf(m1, a));
          m1 = fminf(m1, a);
          const unsigned long long bneg = __ballot(edge[c] && q[c] < 0.0f);
          const unsigned long long bzero = __ballot(edge[c] && signum(q[c]) == 0);
          nneg += __popcll((bneg >> group_shift) & group_mask);
          nzero += __popcll((bzero >> group_shift) & group_mask);
        }
#pragma unroll
        for (int m = 1; m < W; m <<= 1) {
          const float o1 = __shfl_xor(m1, m, 64), o2 = __shfl_xor(m2, m, 64);
          const float lo = fminf(m1, o1);
          const float hi = fminf(fmaxf(m1, o1), fminf(m2, o2));
          m1 = lo;
          m2 = hi;
        }
        const float h1 = horizontal<VARIANT>(m1, p.alpha_f, p.beta_d);
        const float h2 = horizontal<VARIANT>(m2, p.alpha_f, p.beta_d);
#pragma unroll
        for (int c = 0; c < C; ++c) {
          if (edge[c]) {
            const bool neg = q[c] < 0.0f;
            const bool zero = signum(q[c]) == 0;
            const unsigned others_zero = nzero - (zero ? 1u : 0u);
            const float sign = others_zero ? 0.0f : (((nneg - (neg ? 1u : 0u)) & 1u) ? -1.0f : 1.0f);
            const float mag = (fabsf(q[c]) == m1) ? h2 : h1;
            const float r_new = sign * mag;  // static_cast<R>(sign * fn(min)), soft_decision.h:118
            R[(i * C + c) * 64 + lane] = r_new;
            if (NEEDQ) Q[(i * C + c) * 64 + lane] = q[c];
            csn[c] += r_new;  // column_sum: ascending rows, soft_decision.h:88-95
          }
        }
      }
      bool bit[C];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        cs[c] = csn[c];
        const float Lc = cs[c] + y[c];  // :180-182
        if (!done) Lv[c] = Lc;
        bit[c] = colv[c] && (Lc < 0.0f);  // codes.h:51
      }
      // stop test, soft_decision.h:185-186 (see cc_stop_rule)
      bool ok;
      if (p.stop_rule == CC_STOP_AS_SHIPPED) {
        ok = true;
      } else {
        unsigned bad = 0;
        for (int w = 0; w < p.KW; ++w) {
          uint32_t pv = 0;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const uint32_t mask = p.colmask[(w * C + c) * 64 + lane];
            if (p.stop_rule == CC_STOP_PARITY)
              pv ^= bit[c] ? mask : 0u;  // GF(2) syndrome bits of rows 32w..32w+31
            else
              pv |= bit[c] ? mask : 0u;  // integer dot products: any covered 1 makes a row non-zero
          }
#pragma unroll
          for (int m = 1; m < W; m <<= 1) {
            const uint32_t o = __shfl_xor(pv, m, 64);
            pv = (p.stop_rule == CC_STOP_PARITY) ? (pv ^ o) : (pv | o);
          }
          bad |= pv;
        }
        ok = (bad == 0);
      }
      if (ok && !done) {
        done = true;
        my_iter = it;
      }
      if (__all(done)) break;
    }
    if (active) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int j = li + W * c;
        if (j < n) {
          hard[frame * n + j] = (Lv[c] < 0.0f) ? 1 : 0;
          if (Lout) Lout[frame * n + j] = Lv[c];
        }
      }
      if (li == 0) {
        if (iters_out) iters_out[frame] = static_cast<uint16_t>(my_iter);
        if (status_out) status_out[frame] = (my_iter < p.iterations) ? CC_FRAME_OK : CC_FRAME_NOT_CONVERGED;
      }
    }
  }
}

template <int C, int W, bool GSTATE>
hipError_t launch_generic_variant(const MinSumParams &p, int grid, size_t lds, hipStream_t st, const float *llr,
                                  const uint16_t *er, const uint32_t *er_off, uint8_t *hard, float *L,
                                  uint16_t *iters, int32_t *status, unsigned long long B) {
#define CC_LAUNCH(V)                                                                                              \
  if (lds > 48 * 1024) {                                                                                          \
    hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&minsum_generic_kernel<C, W, V, GSTATE>),          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));       \
    if (ea != hipSuccess) return ea;                                                                              \
  }                                                                                                               \
  hipLaunchKernelGGL((minsum_generic_kernel<C, W, V, GSTATE>), dim3(grid), dim3(64), lds, st, p, llr, er, er_off, hard, L, \
                     iters, status, B);                                                                           \
  break
  switch (p.variant) {
    case CC_ALG_MS: CC_LAUNCH(CC_ALG_MS);
    case CC_ALG_NMS: CC_LAUNCH(CC_ALG_NMS);
    case CC_ALG_OMS: CC_LAUNCH(CC_ALG_OMS);
    case CC_ALG_SCMS1: CC_LAUNCH(CC_ALG_SCMS1);
    case CC_ALG_SCMS2: CC_LAUNCH(CC_ALG_SCMS2);
    case CC_ALG_2DNMS: CC_LAUNCH(CC_ALG_2DNMS);
    default: return hipErrorInvalidValue;
  }
#undef CC_LAUNCH
  return hipGetLastError();
}

size_t generic_lds_bytes(const cc_code *code) {
  const bool needq = code->desc.algorithm == CC_ALG_SCMS1 || code->desc.algorithm == CC_ALG_SCMS2;
  return static_cast<size_t>(code->ms_rows) * code->geo.C * 64 * sizeof(float) * (needq ? 2 : 1);
}

}  // namespace

int minsum_kernel_info(const cc_code *code, std::string &name, uint32_t &frames_per_wg, uint32_t &threads,
                       uint32_t &lds) {
  if (minsum_diag_supported(code) && !code->force_generic) {
    const DiagGeometry *dg = diag_geometry(code->tab);
    name = minsum_diag_name(code);
    frames_per_wg = static_cast<uint32_t>(4 * 64 / dg->LPF);
    threads = 256;
    lds = static_cast<uint32_t>(minsum_diag_lds_bytes(*dg));
    return CC_OK;
  }
  name = "minsum_generic_kernel<C=" + std::to_string(code->geo.C) + ",W=" + std::to_string(code->geo.W) + ">";
  frames_per_wg = static_cast<uint32_t>(code->geo.frames_per_wave);
  threads = 64;
  lds = static_cast<uint32_t>(generic_lds_bytes(code));
  if (lds > 160 * 1024) {
    name += "[state in HBM]";
    lds = 0;
  }
  return CC_OK;
}

MinSumParams minsum_params(const cc_code *code) {
  MinSumParams p;
  p.n = static_cast<int>(code->tab.n);
  p.K = static_cast<int>(code->ms_rows);
  p.KW = code->geo.KW;
  p.variant = code->desc.algorithm;
  p.stop_rule = code->desc.stop_rule;
  p.iterations = code->desc.iterations;
  p.alpha_f = static_cast<float>(code->desc.alpha);
  p.beta_f = static_cast<float>(code->desc.beta);
  p.beta_d = code->desc.beta;
  p.colmask = code->d_colmask;
  return p;
}

int launch_minsum(const cc_code *code, const float *d_llr, const uint16_t *d_er, const uint32_t *d_er_off,
                  uint8_t *d_hard, float *d_L, uint16_t *d_iters, int32_t *d_status, size_t B, hipStream_t stream) {
  if (B == 0) return CC_OK;
  MinSumParams p = minsum_params(code);

  if (minsum_diag_supported(code) && !code->force_generic)
    return launch_minsum_diag(code, p, d_llr, d_er, d_er_off, d_hard, d_L, d_iters, d_status, B, stream);

  size_t lds = generic_lds_bytes(code);
  const bool gstate = lds > 160 * 1024;
  const int fpw = code->geo.frames_per_wave;
  const unsigned long long groups = (B + fpw - 1) / fpw;
  unsigned long long waves_per_cu = gstate ? 16 : (lds ? (160 * 1024) / lds : 32);
  if (waves_per_cu > 32) waves_per_cu = 32;
  const unsigned long long max_grid = static_cast<unsigned long long>(code->num_cus) * waves_per_cu;
  const int grid = static_cast<int>(groups < max_grid ? groups : max_grid);
  p.gstate = nullptr;
  p.gslab = lds / sizeof(float);
  if (gstate) {  // stream-ordered scratch: one slab per resident workgroup, freed behind the kernel
    hipError_t ea = workspace_alloc(code, reinterpret_cast<void **>(&p.gstate), lds * static_cast<size_t>(grid), stream);
    if (ea != hipSuccess) return hip_fail(ea, "hipMallocAsync(min-sum state)");
    lds = 0;
  }
  hipError_t e = hipErrorInvalidValue;
  const unsigned long long Bq = B;
#define CC_GEO(CC, WW, GS)                                                                                       \
  if (code->geo.C == CC && code->geo.W == WW && gstate == GS)                                                    \
  e = launch_generic_variant<CC, WW, GS>(p, grid, lds, stream, d_llr, d_er, d_er_off, d_hard, d_L, d_iters,      \
                                         d_status, Bq)
  CC_GEO(1, 16, false);
  CC_GEO(1, 32, false);
  CC_GEO(1, 64, false);
  CC_GEO(2, 64, false);
  CC_GEO(4, 64, false);
  CC_GEO(8, 64, false);   // up to 512 columns (BCH(511, .) soft decoding, larger caller-supplied matrices)
  CC_GEO(16, 64, false);  // up to 1024
  CC_GEO(32, 64, false);  // up to 2048
  CC_GEO(1, 64, true);
  CC_GEO(2, 64, true);
  CC_GEO(4, 64, true);
  CC_GEO(8, 64, true);
  CC_GEO(16, 64, true);
  CC_GEO(32, 64, true);
#undef CC_GEO
  if (p.gstate) {
    const hipError_t ef = hipFreeAsync(p.gstate, stream);
    if (e == hipSuccess) e = ef;
  }
  if (e != hipSuccess) return hip_fail(e, "minsum kernel launch");
  return CC_OK;
}

}  // namespace ccamd
