// encode.hip -- batched systematic encoder and message extraction.
//
// Replaces cyclic::encode src/codes/cyclic.h:289-311 with the free encode() of cyclic.h:29-40:
//   division_tag        c(x) = a(x) x^k + (a(x) x^k mod g(x))   -> parity in 0..k-1, message in k..n-1
//   multiplication_tag  c(x) = a(x) g(x)
// and the message-extraction half of cyclic::decode cyclic.h:313-327 / :42-51 (division_tag: the
// top l coefficients).
//
// The remainder is linear in the message: (a x^k mod g) = sum_j a_j (x^(k+j) mod g).  The k x l
// table PT[i][j] = coefficient i of x^(k+j) mod g is built once per code on the host and staged in
// LDS; one codeword per wavefront, lane l owns message symbols j = l + 64c, the k parity symbols are
// XOR-reduced across the wave four at a time (bytes packed into one dword per DPP chain).
#include "cc_internal.hpp"
#include "wave_ops.hpp"

namespace ccamd {
namespace {

__global__ void __launch_bounds__(256)
encode_division_kernel(const AlgebraicTables *__restrict__ T, const uint8_t *__restrict__ PT,
                       const uint8_t *__restrict__ msg, uint8_t *__restrict__ cw, unsigned long long B) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t *ex = smem;             // 512
  uint8_t *lg = smem + 512;       // 256
  uint8_t *par = smem + 768;       // 4 waves x 256 parity bytes (k <= 254)
  uint8_t *pt = smem + 768 + 1024;  // k * l
  const int n = T->n, k = T->k, l = T->l;
  for (int i = threadIdx.x; i < 512; i += 256) ex[i] = T->exp[i];
  lg[threadIdx.x] = T->log[threadIdx.x];
  for (int i = threadIdx.x; i < k * l; i += 256) pt[i] = PT[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint8_t *mypar = par + wid * 256;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + wid;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long f = wave; f < B; f += nwaves) {
    uint32_t m[4], lm[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = lane + 64 * c;
      m[c] = j < l ? (msg[f * l + j] & static_cast<uint32_t>(n)) : 0u;
      lm[c] = lg[m[c]];
    }
    for (int i0 = 0; i0 < k; i0 += 4) {
      uint32_t packed = 0;
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = i0 + ii;
        uint32_t term = 0;
        if (i < k) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int j = lane + 64 * c;
            const uint32_t pv = j < l ? pt[i * l + j] : 0u;
            term ^= (m[c] && pv) ? ex[lm[c] + lg[pv]] : 0u;
          }
        }
        packed |= term << (8 * ii);
      }
      packed = __builtin_amdgcn_readlane(wave_xor(packed), 63);
      if (lane < 4 && i0 + lane < k) mypar[i0 + lane] = static_cast<uint8_t>(packed >> (8 * lane));
    }
    // coalesced stores: parity (positions 0..k-1), then the message (positions k..n-1)
    for (int i = lane; i < k; i += 64) cw[f * n + i] = mypar[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = lane + 64 * c;
      if (j < l) cw[f * n + k + j] = static_cast<uint8_t>(m[c]);
    }
  }
}

// Binary BCH, message symbols known to be 0 / 1 (the Monte-Carlo source), at most 32 parity bits: the k parity
// bits of a frame are one 32-bit word, the XOR of the words P[j] = bits of x^(k+j) mod g over the set message
// bits.  Lane l keeps P for its four message positions in registers: four masked XORs and one wave reduction per
// frame instead of k GF multiply-accumulate passes.
__global__ void __launch_bounds__(256)
encode_bch_bits_kernel(const uint8_t *__restrict__ PT, const uint8_t *__restrict__ msg, uint8_t *__restrict__ cw, int n,
                       int k, int l, unsigned long long B) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t P[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = lane + 64 * c;
    P[c] = 0;
    if (j < l)
      for (int i = 0; i < k; ++i) P[c] |= static_cast<uint32_t>(PT[i * l + j] & 1u) << i;
  }
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + wid;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long f = wave; f < B; f += nwaves) {
    uint32_t b[4], acc = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = lane + 64 * c;
      b[c] = j < l ? (msg[f * l + j] & 1u) : 0u;
      acc ^= P[c] & (0u - b[c]);
    }
    acc = __builtin_amdgcn_readlane(wave_xor(acc), 63);
    if (lane < k) cw[f * n + lane] = static_cast<uint8_t>((acc >> lane) & 1u);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = lane + 64 * c;
      if (j < l) cw[f * n + k + j] = static_cast<uint8_t>(b[c]);
    }
  }
}

// Binary generator (BCH), any symbol values of GF(2^q): parity symbol i is the XOR of the message symbols m_j
// with bit i of P[j] set (x^(k+j) mod g has 0/1 coefficients), i.e. the bit-parallel kernel above applied to each
// of the q bit planes of the symbols: q masked XORs per message position instead of k multiply-accumulates.
__global__ void __launch_bounds__(256)
encode_bch_planes_kernel(const uint8_t *__restrict__ PT, const uint8_t *__restrict__ msg, uint8_t *__restrict__ cw, int n,
                         int k, int l, int q, unsigned long long B) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t P[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = lane + 64 * c;
    P[c] = 0;
    if (j < l)
      for (int i = 0; i < k; ++i) P[c] |= static_cast<uint32_t>(PT[i * l + j] & 1u) << i;
  }
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + wid;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long f = wave; f < B; f += nwaves) {
    uint32_t m[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = lane + 64 * c;
      m[c] = j < l ? (msg[f * l + j] & static_cast<uint32_t>(n)) : 0u;
    }
    uint32_t par = 0;  // parity symbol of position `lane` (lane < k), assembled plane by plane
    for (int b = 0; b < q; ++b) {
      uint32_t acc = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) acc ^= P[c] & (0u - ((m[c] >> b) & 1u));
      acc = __builtin_amdgcn_readlane(wave_xor(acc), 63);
      par |= ((acc >> lane) & 1u) << b;
    }
    if (lane < k) cw[f * n + lane] = static_cast<uint8_t>(par);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = lane + 64 * c;
      if (j < l) cw[f * n + k + j] = static_cast<uint8_t>(m[c]);
    }
  }
}

__global__ void __launch_bounds__(256)
encode_multiplication_kernel(const AlgebraicTables *__restrict__ T, const uint8_t *__restrict__ msg,
                             uint8_t *__restrict__ cw, unsigned long long B) {
  __shared__ uint8_t ex[512];
  __shared__ uint8_t lg[256];
  __shared__ uint8_t a[4][256];
  const int n = T->n, k = T->k, l = T->l;
  for (int i = threadIdx.x; i < 512; i += 256) ex[i] = T->exp[i];
  lg[threadIdx.x] = T->log[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + wid;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long f = wave; f < B; f += nwaves) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = lane + 64 * c;
      a[wid][j] = j < l ? static_cast<uint8_t>(msg[f * l + j] & n) : 0;
    }
    uint32_t acc[4] = {0, 0, 0, 0};
    for (int d = 0; d <= k; ++d) {  // c_p = sum_d g_d a_{p-d}
      const uint32_t gd = T->g[d];
      const uint32_t lgd = lg[gd];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int p = lane + 64 * c;
        const uint32_t av = (p >= d && p - d < l) ? a[wid][p - d] : 0u;
        acc[c] ^= (gd && av) ? ex[lg[av] + lgd] : 0u;
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int p = lane + 64 * c;
      if (p < n) cw[f * n + p] = static_cast<uint8_t>(acc[c]);
    }
  }
}

__global__ void __launch_bounds__(256)
extract_division_kernel(const uint8_t *__restrict__ cw, uint8_t *__restrict__ msg, int n, int k, int l,
                        unsigned long long total) {
  const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * blockDim.x;
  for (unsigned long long idx = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += stride) {
    const unsigned long long f = idx / l;
    const int j = static_cast<int>(idx - f * l);
    msg[idx] = cw[f * n + k + j];
  }
}

// multiplication_tag: a = b / g (cyclic.h:42-46), the quotient of the long division by the monic generator.
// One wavefront per frame; the running remainder lives in LDS.  Step i reads the uniform coefficient c of
// x^i and subtracts c * g * x^(i-k) on lanes j < k; position i itself is left untouched, so after the last
// step rem[k + j] holds quotient coefficient j.  (LDS operations of one wave execute in order.)
__global__ void __launch_bounds__(256)
extract_multiplication_kernel(const AlgebraicTables *__restrict__ T, const uint8_t *__restrict__ cw,
                              uint8_t *__restrict__ msg, unsigned long long B) {
  __shared__ uint8_t ex[512];
  __shared__ uint8_t lg[256];
  __shared__ uint8_t rem[4][256];
  const int n = T->n, k = T->k, l = T->l;
  for (int i = threadIdx.x; i < 512; i += 256) ex[i] = T->exp[i];
  lg[threadIdx.x] = T->log[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t lgg[4];
  bool gnz[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = lane + 64 * c;
    const uint32_t gj = j < k ? T->g[j] : 0u;
    gnz[c] = gj != 0;
    lgg[c] = lg[gj];
  }
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + wid;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long f = wave; f < B; f += nwaves) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int p = lane + 64 * c;
      rem[wid][p] = p < n ? static_cast<uint8_t>(cw[f * n + p] & n) : 0;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = n - 1; i >= k; --i) {
      const uint32_t c0 = __builtin_amdgcn_readfirstlane(rem[wid][i]);
      if (c0 != 0) {
        const uint32_t lc = lg[c0];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int j = lane + 64 * c;
          if (j < k && gnz[c]) rem[wid][i - k + j] ^= ex[lc + lgg[c]];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = lane + 64 * c;
      if (j < l) msg[f * l + j] = rem[wid][k + j];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

// PT[i][j] = coefficient i of x^(k+j) mod g  (i < k, j < l), row-major k x l
std::vector<uint8_t> build_parity_table(const Field &f, const CodeTables &t) {
  std::vector<uint8_t> pt(static_cast<size_t>(t.k) * t.l, 0);
  std::vector<uint8_t> r(t.k, 0);
  for (unsigned i = 0; i < t.k; ++i) r[i] = t.g[i];  // x^k mod g = g - x^k (char 2), g monic
  for (unsigned j = 0; j < t.l; ++j) {
    for (unsigned i = 0; i < t.k; ++i) pt[static_cast<size_t>(i) * t.l + j] = r[i];
    const uint8_t top = r[t.k - 1];  // multiply by x, reduce with g
    for (unsigned i = t.k - 1; i > 0; --i) r[i] = r[i - 1] ^ f.mul(top, t.g[i]);
    r[0] = f.mul(top, t.g[0]);
  }
  return pt;
}

int launch_encode(const cc_code *code, const uint8_t *d_msg, uint8_t *d_cw, size_t B, hipStream_t stream) {
  if (B == 0) return CC_OK;
  if (bitslice_encode_supported(code)) return launch_bitslice_encode(code, d_msg, d_cw, B, stream);
  const unsigned long long blocks_needed = (B + 3) / 4;
  const unsigned long long max_grid = static_cast<unsigned long long>(code->num_cus) * 8;
  const int grid = static_cast<int>(blocks_needed < max_grid ? blocks_needed : max_grid);
  const unsigned long long Bq = B;
  if (code->desc.coding == CC_CODING_MULTIPLICATION) {
    hipLaunchKernelGGL(encode_multiplication_kernel, dim3(grid), dim3(256), 0, stream, code->d_alg, d_msg, d_cw, Bq);
  } else if (code->tab.family == CC_FAMILY_BCH && code->tab.k <= 32 && code->d_parity != nullptr) {
    hipLaunchKernelGGL(encode_bch_planes_kernel, dim3(grid), dim3(256), 0, stream, code->d_parity, d_msg, d_cw,
                       static_cast<int>(code->tab.n), static_cast<int>(code->tab.k), static_cast<int>(code->tab.l),
                       static_cast<int>(code->tab.q), Bq);
  } else {
    const size_t lds = 768 + 1024 + static_cast<size_t>(code->tab.k) * code->tab.l;
    if (lds > 64 * 1024) {
      set_last_error("encoder parity table does not fit 64 KiB of LDS for this code");
      return CC_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(encode_division_kernel, dim3(grid), dim3(256), lds, stream, code->d_alg, code->d_parity, d_msg,
                       d_cw, Bq);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "encode kernel launch");
  return CC_OK;
}

// systematic encode of 0 / 1 message symbols (callers guarantee the symbol range: the Monte-Carlo source)
int launch_encode_bits(const cc_code *code, const uint8_t *d_msg, uint8_t *d_cw, size_t B, hipStream_t stream) {
  if (B == 0) return CC_OK;
  if (code->tab.family != CC_FAMILY_BCH || code->desc.coding != CC_CODING_DIVISION || code->tab.k > 32 ||
      code->d_parity == nullptr)
    return launch_encode(code, d_msg, d_cw, B, stream);
  const unsigned long long blocks_needed = (B + 3) / 4;
  const unsigned long long max_grid = static_cast<unsigned long long>(code->num_cus) * 8;
  const int grid = static_cast<int>(blocks_needed < max_grid ? blocks_needed : max_grid);
  hipLaunchKernelGGL(encode_bch_bits_kernel, dim3(grid), dim3(256), 0, stream, code->d_parity, d_msg, d_cw,
                     static_cast<int>(code->tab.n), static_cast<int>(code->tab.k), static_cast<int>(code->tab.l),
                     static_cast<unsigned long long>(B));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "encode kernel launch");
  return CC_OK;
}

int launch_extract(const cc_code *code, const uint8_t *d_cw, uint8_t *d_msg, size_t B, hipStream_t stream) {
  if (B == 0) return CC_OK;
  if (code->desc.coding == CC_CODING_MULTIPLICATION) {
    const unsigned long long blocks_needed = (B + 3) / 4;
    const unsigned long long cap = static_cast<unsigned long long>(code->num_cus) * 8;
    hipLaunchKernelGGL(extract_multiplication_kernel, dim3(static_cast<int>(blocks_needed < cap ? blocks_needed : cap)),
                       dim3(256), 0, stream, code->d_alg, d_cw, d_msg, static_cast<unsigned long long>(B));
    hipError_t em = hipGetLastError();
    if (em != hipSuccess) return hip_fail(em, "extract kernel launch");
    return CC_OK;
  }
  const unsigned long long total = static_cast<unsigned long long>(B) * code->tab.l;
  const unsigned long long want = (total + 255) / 256;
  const unsigned long long max_grid = static_cast<unsigned long long>(code->num_cus) * 8;
  const int grid = static_cast<int>(want < max_grid ? want : max_grid);
  hipLaunchKernelGGL(extract_division_kernel, dim3(grid), dim3(256), 0, stream, d_cw, d_msg,
                     static_cast<int>(code->tab.n), static_cast<int>(code->tab.k), static_cast<int>(code->tab.l), total);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "extract kernel launch");
  return CC_OK;
}

}  // namespace ccamd
