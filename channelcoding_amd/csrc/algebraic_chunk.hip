// algebraic_chunk.hip -- the algebraic chain for Berlekamp-Massey / PGZ without erasures, restructured so
// that each stage runs in the lane mapping that suits it.  A wavefront owns a chunk of FPW frames:
//
//   A  syndromes            one frame at a time, lane l owns positions l + 64c (parallel over positions);
//                           S_j and log S_j go to LDS as [j][frame]
//   B  Berlekamp-Massey     ONE LANE PER FRAME (hard_decision.h:116-155).  BM is a serial recurrence of 2t
//                           steps; with a wavefront per frame every step pays a 6-stage cross-lane reduction for
//                           the discrepancy plus dependent table look-ups for 17 useful lanes.  Here lambda, b and
//                           the syndromes of frame f live in LDS column f ([coefficient][frame], conflict-free for
//                           any row), polynomials are kept in the log domain with log 0 := 512 and an antilog
//                           table that is zero above 510, so a GF multiply-accumulate is two linear LDS reads, one
//                           table gather and an XOR, with no zero tests
//   C  root search, error values, re-check, store: one frame at a time again (parallel over positions)
//
// (Round 3: the Euklid tag and, on the bit-plane chain, erasures are served here too -- algebraic_chunk_supported.)  Same results as that kernel bit for
// bit (tests/test_gpu_algebraic.py runs both through CC_AMD_NO_CHUNK=1).
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "bitplane.hpp"
#include "cc_internal.hpp"
#include "wave_ops.hpp"

namespace ccamd {
int launch_bitslice_chien(const void *d_lamp, void *d_masks, size_t B, bool long_locators, hipStream_t stream);  // bitslice.hip
int launch_bitslice_roots_transpose(const void *d_masks, void *d_rootsT, size_t B, hipStream_t stream);
namespace {

constexpr uint32_t kLogZero = 512;  // log of 0: ex[kLogZero + anything < 512] = 0

__device__ __forceinline__ uint32_t lane63(uint32_t v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) { return ~lane63(wave_umin(~v)); }

struct ChunkLayout {  // byte offsets inside one wavefront's LDS region
  int SL, LL, BL, SV, LV, DEG, LEN, CSL, CLL, OML, CS, RP, VAL, bytes;
};
__host__ __device__ inline ChunkLayout chunk_layout(int t2, int fpw) {
  ChunkLayout c;
  const int nc = t2 + 1;
  c.SL = 0;                      // u16 [t2][fpw]   log S_j
  c.LL = c.SL + 2 * t2 * fpw;    // u16 [nc][fpw]   log lambda_m
  c.BL = c.LL + 2 * nc * fpw;    // u16 [nc][fpw]   log b_m
  c.SV = c.BL + 2 * nc * fpw;    // u8  [t2][fpw]   S_j
  c.LV = c.SV + t2 * fpw;        // u8  [nc][fpw]   lambda_m
  c.DEG = c.LV + nc * fpw;       // u8  [fpw]       deg lambda
  c.LEN = c.DEG + fpw;           // u8  [fpw]       LFSR length L
  // stage C scratch for the frame being corrected (contiguous copies of its column)
  c.CSL = (c.LEN + fpw + 1) & ~1;  // u16 [64]   log S_j
  c.CLL = c.CSL + 128;             // u16 [72]   log lambda_m
  c.OML = c.CLL + 144;             // u16 [64]   log omega_j
  c.CS = c.OML + 128;              // u8  [64]   S_j
  c.RP = c.CS + 64;                // u8  [64]   positions of the located errors
  c.VAL = c.RP + 64;               // u8  [64]   their values
  c.bytes = (c.VAL + 64 + 15) & ~15;
  return c;
}

template <bool FLOAT_IN, int FPW>
__global__ void __launch_bounds__(256, 4)
algebraic_chunk_kernel(const AlgebraicTables *__restrict__ T, int alg, const void *__restrict__ in_raw,
                       uint8_t *__restrict__ out, int32_t *__restrict__ nerr_out, int32_t *__restrict__ status_out,
                       unsigned long long B) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t *ex = smem;                                            // [1024]
  uint16_t *lg2 = reinterpret_cast<uint16_t *>(smem + 1024);     // [256]
  uint8_t *lg = smem + 1536;                                     // [256] plain log table (log 0 = 0), stages A / C
  for (int i = threadIdx.x; i < 1024; i += 256) ex[i] = i < 512 ? T->exp[i] : 0;
  lg2[threadIdx.x] = threadIdx.x ? T->log[threadIdx.x] : kLogZero;
  lg[threadIdx.x] = T->log[threadIdx.x];
  __syncthreads();

  const int dbg_stop = alg >> 8;  // timing experiments only (CC_AMD_ALG_STOP): 1 after syndromes, 2 after BM, 3 after roots
  alg &= 0xFF;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n = T->n, nn = n, t2 = T->nroots, nc = t2 + 1;
  const bool is_rs = T->family == CC_FAMILY_RS;
  const ChunkLayout lay = chunk_layout(t2, FPW);
  uint8_t *base = smem + 1792 + wid * lay.bytes;
  uint16_t *SL = reinterpret_cast<uint16_t *>(base + lay.SL);
  uint16_t *LL = reinterpret_cast<uint16_t *>(base + lay.LL);
  uint16_t *BL = reinterpret_cast<uint16_t *>(base + lay.BL);
  uint8_t *SV = base + lay.SV, *LV = base + lay.LV, *DEG = base + lay.DEG, *LEN = base + lay.LEN;
  uint16_t *CSL = reinterpret_cast<uint16_t *>(base + lay.CSL);
  uint16_t *CLL = reinterpret_cast<uint16_t *>(base + lay.CLL);
  uint8_t *CS = base + lay.CS, *RP = base + lay.RP, *VAL = base + lay.VAL;

  // per-lane exponent bookkeeping (roots alpha^(r0 + j step)): position p = lane + 64 c
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
    xinv[c] = static_cast<uint32_t>((nn - (p % nn)) % nn);
  }
  uint32_t dk[4][4];  // (k + 1) * dstep mod nn, k = 0..3
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) dk[c][k] = (static_cast<uint32_t>(k + 1) * dstep[c]) % static_cast<uint32_t>(nn);
  auto load_symbols = [&](unsigned long long frame, uint32_t (&sym)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int p = lane + 64 * c;
      if (FLOAT_IN)  // hard decision of a signed sequence: cyclic.h:163-173, codes.h:43-52
        sym[c] = valid[c] ? (static_cast<const float *>(in_raw)[frame * n + p] < 0.0f ? 1u : 0u) : 0u;
      else
        sym[c] = valid[c] ? (static_cast<const uint8_t *>(in_raw)[frame * n + p] & static_cast<uint32_t>(n)) : 0u;
    }
  };

  const unsigned long long nchunks = (B + FPW - 1) / FPW;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + wid;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long chunk = wave; chunk < nchunks; chunk += nwaves) {
    const unsigned long long first = chunk * FPW;
    const int frames = static_cast<int>((B - first) < static_cast<unsigned long long>(FPW) ? (B - first) : FPW);

    // ---------------- A: syndromes (cyclic.h:53-63), four per DPP reduction ----------------
    // Term of S_j at position p: b_p alpha^(root_j p) = ex[lt + k d] with lt = log b_p + r0 p (mod nn) advanced
    // by 4 d per group of four syndromes, d = step * p (mod nn); the multiples k d (mod nn) are per-lane
    // constants, so a term costs one add, one table read and one XOR.  A zero symbol parks lt on the zero
    // part of the table (log 0 = 512) and advances by nn, which the wrap undoes.
    unsigned long long smask = 0;  // frames with a non-zero syndrome
    constexpr int PF = 3;          // frames in flight: the stage is latency-bound otherwise (255 B per frame)
    uint32_t symq[PF][4];
#pragma unroll
    for (int k = 0; k < PF; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) symq[k][c] = 0;
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (k < frames) load_symbols(first + k, symq[k]);
    for (int s = 0; s < frames; ++s) {
      {
        uint32_t sym[4], lt[4], adv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint32_t v = sym[c] = symq[0][c];
          uint32_t l0 = lg[v] + e0[c];
          l0 = umin32(l0, l0 - static_cast<uint32_t>(nn));
          lt[c] = v ? l0 : kLogZero;
          adv[c] = v ? dk[c][3] : static_cast<uint32_t>(nn);
        }
#pragma unroll
        for (int k = 0; k + 1 < PF; ++k)  // rotate the queue, refill its tail
#pragma unroll
          for (int c = 0; c < 4; ++c) symq[k][c] = symq[k + 1][c];
        if (s + PF < frames) load_symbols(first + s + PF, symq[PF - 1]);
        uint32_t any = 0;
        for (int j0 = 0; j0 < t2; j0 += 4) {
          uint32_t t0 = 0, t1 = 0, t2v = 0, t3 = 0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            t0 ^= ex[lt[c]];
            t1 ^= ex[lt[c] + dk[c][0]];
            t2v ^= ex[lt[c] + dk[c][1]];
            t3 ^= ex[lt[c] + dk[c][2]];
            lt[c] += adv[c];
            lt[c] = umin32(lt[c], lt[c] - static_cast<uint32_t>(nn));
          }
          uint32_t packed = t0 | (t1 << 8) | (t2v << 16) | (t3 << 24);
          packed = lane63(wave_xor(packed));
          if (j0 + 4 > t2) packed &= 0xFFFFFFFFu >> (8 * (j0 + 4 - t2));  // t2 is not a multiple of four
          any |= packed;
          if (lane < 4 && j0 + lane < t2) {
            const uint32_t v = (packed >> (8 * lane)) & 0xFFu;
            SV[(j0 + lane) * FPW + s] = static_cast<uint8_t>(v);
            SL[(j0 + lane) * FPW + s] = lg2[v];
          }
        }
        if (any != 0 && dbg_stop != 1) {
          smask |= 1ull << s;
        } else {  // a codeword: done (cyclic.h:225-231)
          const unsigned long long frame = first + s;
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (valid[c]) out[frame * n + lane + 64 * c] = static_cast<uint8_t>(sym[c]);
          if (lane == 0) {
            if (nerr_out) nerr_out[frame] = 0;
            if (status_out) status_out[frame] = CC_FRAME_OK;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();

    // ---------------- B: Berlekamp-Massey, one lane per frame (hard_decision.h:116-155) ----------------
    if (smask != 0) {  // wave-uniform
      const int f = lane & (FPW - 1);
      const bool mine = lane < FPW && ((smask >> lane) & 1ull);
      if (lane < FPW) {
        for (int m = 0; m < nc; ++m) {  // lambda = b = 1
          LV[m * FPW + f] = m == 0;
          LL[m * FPW + f] = static_cast<uint16_t>(m == 0 ? 0 : kLogZero);
          BL[m * FPW + f] = static_cast<uint16_t>(m == 0 ? 0 : kLogZero);
        }
      }
      int l = 0, shift = 0;  // b is stored unshifted; b(x) x^shift is the polynomial of the recurrence
      for (int i = 0; i < t2; ++i) {
        shift += 1;  // b = b * x, :134
        const int lw = static_cast<int>(wave_umax(mine ? static_cast<uint32_t>(l) : 0u));
        // discrepancy :139-141; lambda_m = 0 (log 512) beyond its degree, so no per-lane bound is needed
        uint32_t d = SV[i * FPW + f];
        const int mm = i < lw ? i : lw;
        // lambda_m = 0 (log 512) for m > L, and L <= i: running past mm in blocks of four adds zeros
        for (int m0 = 1; m0 <= mm; m0 += 4) {
          uint32_t la[4], sa[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int m = m0 + u < nc ? m0 + u : nc - 1;
            la[u] = LL[m * FPW + f];
            sa[u] = SL[(i - m > 0 ? i - m : 0) * FPW + f];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) d ^= ex[la[u] + sa[u]];
        }
        const bool upd = mine && d != 0;
        const bool grow = upd && 2 * l <= i;  // :145 (rho = 0)
        const uint32_t ld = lg2[d];
        const uint32_t linv = static_cast<uint32_t>(nn) - ld;  // log of d^-1 (or nn for d = 1: wrapped below)
        const int lnew = grow ? i + 1 - l : l;
        const int cap = static_cast<int>(wave_umax(upd ? static_cast<uint32_t>(lnew) : 0u));
        if (__any(upd)) {
          // lambda += d * b * x^shift, and where the register grows b := lambda_old / d; descending m so that the
          // shifted reads of the old b (index m - shift < m) happen before that index is overwritten
          for (int m = cap; m >= 0; --m) {
            const uint32_t lold = LL[m * FPW + f];
            const uint32_t lv = LV[m * FPW + f];
            const int bi = m - shift;
            const uint32_t bt = bi >= 0 ? BL[(bi >= 0 ? bi : 0) * FPW + f] : kLogZero;
            const uint32_t nv = lv ^ ex[ld + bt];
            if (upd) {
              LV[m * FPW + f] = static_cast<uint8_t>(nv);
              LL[m * FPW + f] = lg2[nv];
            }
            if (grow) {
              uint32_t q = lold + linv;
              q = q >= static_cast<uint32_t>(nn) ? q - nn : q;
              BL[m * FPW + f] = static_cast<uint16_t>(lold >= kLogZero ? kLogZero : q);
            }
          }
        }
        if (grow) {
          l = lnew;
          shift = 0;
        }
      }
      if (mine) {
        int deg = 0;
        for (int m = t2; m >= 1; --m)
          if (deg == 0 && LV[m * FPW + f] != 0) deg = m;
        DEG[f] = static_cast<uint8_t>(deg);
        LEN[f] = static_cast<uint8_t>(l);
      }
    }
    __builtin_amdgcn_wave_barrier();

    // ---------------- C: roots, error values, re-check, store ----------------
    // All polynomial arithmetic on logs with log 0 = 512 (no zero tests): a term lambda_m X^-m is
    // ex[log lambda_m + (m * log X^-1 mod nn)], the exponent advancing by one add + one wrap per coefficient.
    // only frames with a non-zero syndrome are visited (the others were stored in stage A); two in flight
    auto pop = [](unsigned long long &m) {
      const int i = m ? __builtin_ctzll(m) : -1;
      m &= m - 1;
      return i;
    };
    unsigned long long todo = smask;
    int s0 = pop(todo), s1 = pop(todo);
    uint32_t q0[4] = {0, 0, 0, 0}, q1[4] = {0, 0, 0, 0};
    if (s0 >= 0) load_symbols(first + s0, q0);
    if (s1 >= 0) load_symbols(first + s1, q1);
    while (s0 >= 0) {
      const int s = s0;
      const unsigned long long frame = first + s;
      uint32_t sym[4], corr[4] = {0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sym[c] = q0[c];
        q0[c] = q1[c];
      }
      s0 = s1;
      s1 = pop(todo);
      if (s1 >= 0) load_symbols(first + s1, q1);
      int status = CC_FRAME_OK, nerr = 0;
      {
        const int deg = DEG[s], len = LEN[s];
        // column s of the chunk arrays -> contiguous scratch (the strided reads conflict, do them once)
        if (lane < t2) {
          CS[lane] = SV[lane * FPW + s];
          CSL[lane] = SL[lane * FPW + s];
        }
        for (int m = lane; m < nc; m += 64) CLL[m] = LL[m * FPW + s];
        const uint32_t cll = LL[(lane < nc ? lane : 0) * FPW + s];  // log lambda_lane, read by readlane (nc <= 64)
        // the PGZ tag runs as bounded-distance decoding: locator degree within capability
        if (alg != CC_ALG_BM && 2 * deg > t2) status = CC_FRAME_LOCATOR;
        if (deg < 1) status = CC_FRAME_LOCATOR;  // cyclic.h:145-147
        if (dbg_stop == 2) status = CC_FRAME_LOCATOR;

        // root search: position p is in error iff lambda(alpha^-p) = 0  (cyclic.h:126-150)
        uint32_t isroot[4] = {0, 0, 0, 0}, rank[4] = {0, 0, 0, 0};
        if (status == CC_FRAME_OK) {
          uint32_t acc[4] = {0, 0, 0, 0}, e[4] = {0, 0, 0, 0};
          for (int m = 0; m <= deg; ++m) {
            const uint32_t lm = __builtin_amdgcn_readlane(cll, m);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              acc[c] ^= ex[lm + e[c]];
              e[c] += xinv[c];
              e[c] = umin32(e[c], e[c] - static_cast<uint32_t>(nn));
            }
          }
          uint32_t count = 0;
          const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            isroot[c] = (valid[c] && acc[c] == 0) ? 1u : 0u;
            const unsigned long long mk = __ballot(isroot[c] != 0);
            rank[c] = count + static_cast<uint32_t>(__builtin_popcountll(mk & below));
            count += static_cast<uint32_t>(__builtin_popcountll(mk));
          }
          nerr = static_cast<int>(count);
          if (nerr != deg) status = CC_FRAME_LOCATOR;  // cyclic.h:134-143
        }
        if (dbg_stop == 3) status = CC_FRAME_LOCATOR;

        // error values: bch.h:80-83 (all ones) / Forney for rs.h:41-78, one lane per located error
        if (status == CC_FRAME_OK && is_rs) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (isroot[c]) RP[rank[c]] = static_cast<uint8_t>(lane + 64 * c);
          uint32_t om = 0;  // omega_j = sum_{m<=j} S_{j-m} lambda_m, j < deg
          for (int m = 0; m <= deg; ++m) {
            const uint32_t lm = __builtin_amdgcn_readlane(cll, m);
            const bool in = lane >= m && lane < deg && lane - m < t2;
            om ^= in ? ex[lm + CSL[in ? lane - m : 0]] : 0u;
          }
          const uint32_t oml = lg2[om];
          uint32_t y = 0;
          if (lane < deg) {
            const uint32_t p = RP[lane];
            const uint32_t xi = p ? static_cast<uint32_t>(nn) - p : 0u;  // log X^-1
            uint32_t x2 = 2 * xi;
            x2 = umin32(x2, x2 - static_cast<uint32_t>(nn));
            uint32_t num = 0, den = 0, e = 0;
            for (int j = 0; j < deg; ++j) {  // omega(X^-1)
              num ^= ex[__builtin_amdgcn_readlane(oml, j) + e];
              e += xi;
              e = umin32(e, e - static_cast<uint32_t>(nn));
            }
            e = 0;
            for (int m = 1; m <= deg; m += 2) {  // lambda'(X^-1) = sum_{m odd} lambda_m X^-(m-1)
              den ^= ex[__builtin_amdgcn_readlane(cll, m) + e];
              e += x2;
              e = umin32(e, e - static_cast<uint32_t>(nn));
            }
            y = (num && den) ? ex[lg[num] + nn - lg[den]] : 0u;
          }
          VAL[lane] = static_cast<uint8_t>(y);
#pragma unroll
          for (int c = 0; c < 4; ++c) corr[c] = isroot[c] ? VAL[rank[c]] : 0u;
        } else if (status == CC_FRAME_OK) {
#pragma unroll
          for (int c = 0; c < 4; ++c) corr[c] = isroot[c];
        }
        // re-check (cyclic.h:243-248): decided by L = deg lambda (proof in algebraic.hip), evaluated otherwise
        if (status == CC_FRAME_OK && len != deg) {
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
                want |= static_cast<uint32_t>(CS[j0 + jj]) << (8 * jj);
              }
            }
            mismatch |= lane63(wave_xor(packed)) ^ want;
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
}


// ================= split form behind the bit-plane syndromes (bitslice.hip) =================
// Stage B is bound by the LDS pipe (eight LDS operations per coefficient step) and wants every lane busy: 64
// frames per wavefront, 17 KB of LDS per wavefront, two wavefronts per SIMD.  Stage C walks the dirty frames one at
// a time through chains of dependent table look-ups and wants many wavefronts: its own kernel with 0.3 KB of LDS
// per wavefront.  lambda (logs), deg / L and the dirty masks travel through HBM (70 B per frame).
struct BmLayout {
  int SL, LL, BL, bytes;
};
__host__ __device__ inline BmLayout bm_layout(int t2) {
  BmLayout c;
  const int nc = t2 + 1;
  c.SL = 0;                     // u16 [t2][64]  log S_j
  c.LL = c.SL + 2 * t2 * 64;    // u16 [nc][64]  log lambda_m
  c.BL = c.LL + 2 * nc * 64;    // u16 [nc][64]  log b_m (after the recurrence: lambda_0 .. lambda_16 as bytes)
  c.bytes = (c.BL + 2 * nc * 64 + 15) & ~15;
  return c;
}

// Everything in the log domain (log 0 = 512, antilog table zero above 510): 12.3 KB of LDS per wavefront for 2t = 32,
// three wavefronts per SIMD -- the stage is bound by the latency of its dependent LDS operations, so occupancy is
// what it is sized for.
__global__ void __launch_bounds__(256, 3)
chunk_bm_kernel(const AlgebraicTables *__restrict__ T, int dbg_stop, const uint8_t *__restrict__ synd,
                const uint16_t *__restrict__ er, const uint32_t *__restrict__ er_off,
                uint16_t *__restrict__ llg, uint16_t *__restrict__ meta, unsigned long long *__restrict__ mask,
                uint4 *__restrict__ lamp, int ncoef, uint32_t *__restrict__ nleft,
                int32_t *__restrict__ nerr_out, int32_t *__restrict__ status_out, unsigned long long B) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (blockIdx.x == 0 && threadIdx.x == 0) *nleft = 0;  // chunks chunk_fixl_kernel will hand on (it runs after this kernel)
  uint8_t *ex = smem;                                         // [1024]
  uint16_t *lg2 = reinterpret_cast<uint16_t *>(smem + 1024);  // [256]
  for (int i = threadIdx.x; i < 1024; i += 256) ex[i] = i < 512 ? T->exp[i] : 0;
  lg2[threadIdx.x] = threadIdx.x ? T->log[threadIdx.x] : kLogZero;
  __syncthreads();
  constexpr int FPW = 64, U = 4;  // U coefficients per trip of the two inner loops (8: measured slower)
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), f = lane;
  const int nn = T->n, t2 = T->nroots, nc = t2 + 1;
  const BmLayout lay = bm_layout(t2);
  uint8_t *base = smem + 1536 + wid * lay.bytes;
  uint16_t *SL = reinterpret_cast<uint16_t *>(base + lay.SL);
  uint16_t *LL = reinterpret_cast<uint16_t *>(base + lay.LL);
  uint16_t *BL = reinterpret_cast<uint16_t *>(base + lay.BL);

  const unsigned long long nchunks = (B + FPW - 1) / FPW;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + wid;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long chunk = wave; chunk < nchunks; chunk += nwaves) {
    const unsigned long long first = chunk * FPW;
    const int frames = static_cast<int>((B - first) < static_cast<unsigned long long>(FPW) ? (B - first) : FPW);
    // syndromes of frame 8i + k of group g: byte 4k + i of [block of 64 groups][j][group in block][32]
    const unsigned long long group = 2 * chunk + (f >> 5);
    const int fi = f & 31;
    const uint8_t *src = synd + ((group >> 6) * t2 * 64 + (group & 63)) * 32 + 4 * (fi & 7) + (fi >> 3);
    uint32_t any = 0;
#pragma unroll 8
    for (int j = 0; j < t2; ++j) {
      const uint32_t v = src[j * 2048];
      SL[j * FPW + f] = lg2[v];
      any |= v;
    }
    // erasures of the lane's frame (CSR; none without the arrays): more than 2t cannot be located (bch.h:105-107)
    uint32_t rho = 0, ebase = 0;
    if (er_off != nullptr && f < frames) {
      ebase = er_off[first + f];
      rho = er_off[first + f + 1] - ebase;
    }
    const bool too_many = rho > static_cast<uint32_t>(t2);
    const unsigned long long smask = dbg_stop == 1 ? 0ull : __ballot(any != 0 && f < frames && !too_many);
    const bool mine = (smask >> lane) & 1ull;
    if (f < frames && !mine) {  // a codeword: done (cyclic.h:225-231) -- or settled as "too many erasures"
      const bool refused = any != 0 && too_many;
      if (nerr_out) nerr_out[first + f] = refused ? -1 : 0;
      if (status_out) status_out[first + f] = refused ? CC_FRAME_ERASURES : CC_FRAME_OK;
    }
    if (lane == 0) mask[chunk] = smask;
    if (smask == 0) continue;  // wave-uniform

    // Berlekamp-Massey, one lane per frame (hard_decision.h:116-155)
    for (int m = 0; m < nc; ++m) LL[m * FPW + f] = static_cast<uint16_t>(m == 0 ? 0 : kLogZero);  // lambda = 1
    // lambda *= (1 + alpha^p x) for every erased position p, :128-131; the recurrence then starts at i = rho with
    // b = lambda and L = rho
    const int rmax = static_cast<int>(wave_umax(mine ? rho : 0u));
    for (int e = 0; e < rmax; ++e) {
      const bool act = mine && static_cast<uint32_t>(e) < rho;
      const uint32_t px = act ? static_cast<uint32_t>(er[ebase + e]) % static_cast<uint32_t>(nn) : 0u;
      for (int m = e + 1; m >= 1; --m) {
        const uint32_t nv = ex[LL[m * FPW + f]] ^ ex[LL[(m - 1) * FPW + f] + px];
        if (act) LL[m * FPW + f] = lg2[nv];
      }
    }
    for (int m = 0; m < nc; ++m) BL[m * FPW + f] = LL[m * FPW + f];
    const int irho = static_cast<int>(rho);
    int l = irho, shift = 0;  // b is stored unshifted; b(x) x^shift is the polynomial of the recurrence
    int lw = rmax;            // longest register in the wavefront: max(lw, cap) after every step (cap covers all that grew)
    for (int i = 0; i < t2; ++i) {
      const bool started = i >= irho;  // (a lane with erasures joins at step rho)
      shift += started ? 1 : 0;        // b = b * x, :134
      uint32_t d = ex[SL[i * FPW + f]];
      const int mm = i < lw ? i : lw;
      // discrepancy :139-141; lambda_m = 0 (log 512) for m > L, and L <= i: running past mm in blocks of U adds zeros
      for (int m0 = 1; m0 <= mm; m0 += U) {
        uint32_t la[U], sa[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int m = m0 + u < nc ? m0 + u : nc - 1;
          la[u] = LL[m * FPW + f];
          sa[u] = SL[(i - m > 0 ? i - m : 0) * FPW + f];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) d ^= ex[la[u] + sa[u]];
      }
      const bool upd = mine && started && d != 0;
      const bool grow = upd && 2 * l <= i + irho;  // :145
      const uint32_t ld = lg2[d];
      const uint32_t linv = static_cast<uint32_t>(nn) - ld;  // log of d^-1 (or nn for d = 1: wrapped below)
      const int lnew = grow ? i + irho + 1 - l : l;
      const int cap = static_cast<int>(wave_umax(upd ? static_cast<uint32_t>(lnew) : 0u));
      if (__any(upd)) {
        // lambda += d * b * x^shift, and where the register grows b := lambda_old / d; descending m so that the
        // shifted reads of the old b (index m - shift < m) happen before that index is overwritten.
        // U coefficients per trip, all reads before the look-ups before the writes: a read of b at m - shift
        // always precedes the write of that index in the sequential order too.
        for (int m1 = cap; m1 >= 0; m1 -= U) {
          uint32_t lold[U], bt[U], nv[U], ln[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int m = m1 - u > 0 ? m1 - u : 0, bi = m1 - u - shift;
            lold[u] = LL[m * FPW + f];
            bt[u] = bi >= 0 ? BL[(bi >= 0 ? bi : 0) * FPW + f] : kLogZero;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) nv[u] = ex[lold[u]] ^ ex[ld + bt[u]];
#pragma unroll
          for (int u = 0; u < U; ++u) ln[u] = lg2[nv[u]];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int m = m1 - u;
            if (m < 0) break;  // wave-uniform
            if (upd) LL[m * FPW + f] = static_cast<uint16_t>(ln[u]);
            if (grow) {
              uint32_t q = lold[u] + linv;
              q = q >= static_cast<uint32_t>(nn) ? q - nn : q;
              BL[m * FPW + f] = static_cast<uint16_t>(lold[u] >= kLogZero ? kLogZero : q);
            }
          }
        }
      }
      if (grow) {
        l = lnew;
        shift = 0;
      }
      lw = cap > lw ? cap : lw;
    }
    int deg = 0;
    for (int m = t2; m >= 1; --m)
      if (deg == 0 && LL[m * FPW + f] != kLogZero) deg = m;
    if (f < frames) meta[first + f] = static_cast<uint16_t>(deg | (l << 8));
    for (int m = 0; m < nc; ++m) llg[(chunk * nc + m) * FPW + f] = LL[m * FPW + f];
    // lambda_0 .. lambda_16 as planes for the Chien kernel ([block of 64 groups][m][group][8]): the values go to the
    // (now free) b area as bytes [m][64]; lane (m, half) takes the 32 bytes of coefficient m of one group as eight
    // dwords (word j = frames 4j .. 4j+3) through the butterfly, so bit 8 (f & 3) + (f >> 2) of a plane belongs to
    // frame f of the group
    uint8_t *LV = reinterpret_cast<uint8_t *>(BL);
    // (ncoef = 17 coefficients for the root search on planes, 25 for calls with erasures: bitslice.hip)
    for (int m = 0; m < ncoef && m < nc; ++m) LV[m * FPW + f] = ex[LL[m * FPW + f]];
    if (lane < 2 * ncoef) {
      const int m = lane >> 1, half = lane & 1;
      uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (m < nc) {
        const uint4 *row = reinterpret_cast<const uint4 *>(LV + m * FPW + 32 * half);
        const uint4 a = row[0], b = row[1];
        w[0] = a.x, w[1] = a.y, w[2] = a.z, w[3] = a.w, w[4] = b.x, w[5] = b.y, w[6] = b.z, w[7] = b.w;
        bitplane::butterfly(w);
      }
      const unsigned long long g = 2 * chunk + half;
      uint4 *dst = lamp + (((g >> 6) * ncoef + m) * 64 + (g & 63)) * 2;
      dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
      dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
}

// f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}): a loop whose index is a constant
// expression in the body (register arrays need that)
template <class F, int... I> __device__ __forceinline__ void for_each_index_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void for_each_index(F &&f) {
  for_each_index_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>());
}

// The same recurrence with lambda, b and the syndromes in REGISTERS (logs), for 2t = T2 known at compile time: the 2t
// steps and both inner loops are unrolled, so every coefficient has a fixed register and the only LDS traffic left
// is the table look-ups -- four dependent LDS latencies per step instead of one per coefficient group.  b is kept as
// B = b x^shift: the multiplication by x that every frame performs at every step is a renaming of registers
// (coefficient k of B lives in P[(k - i) mod (T2 + 1)] at step i), a frame whose register grows overwrites B with
// lambda_old / d in place.  Blocks of four coefficients are skipped under wave-uniform bounds (longest register /
// longest update in the wavefront), as in chunk_bm_kernel.
template <int T2>
__global__ void __launch_bounds__(256, 3)
chunk_bm_reg_kernel(const AlgebraicTables *__restrict__ T, int dbg_stop, const uint8_t *__restrict__ synd,
                    uint16_t *__restrict__ llg, uint16_t *__restrict__ meta, unsigned long long *__restrict__ mask,
                    uint4 *__restrict__ lamp, uint32_t *__restrict__ nleft, int32_t *__restrict__ nerr_out,
                    int32_t *__restrict__ status_out, unsigned long long B) {
  constexpr int NC = T2 + 1, FPW = 64;
  __shared__ __attribute__((aligned(16))) uint8_t smem[1536 + 4 * 17 * 64];
  if (blockIdx.x == 0 && threadIdx.x == 0) *nleft = 0;
  uint8_t *ex = smem;                                         // [1024]
  uint16_t *lg2 = reinterpret_cast<uint16_t *>(smem + 1024);  // [256]
  for (int i = threadIdx.x; i < 1024; i += 256) ex[i] = i < 512 ? T->exp[i] : 0;
  lg2[threadIdx.x] = threadIdx.x ? T->log[threadIdx.x] : kLogZero;
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), f = lane;
  const uint32_t nn = static_cast<uint32_t>(T->n);
  uint8_t *LV = smem + 1536 + wid * (17 * 64);  // lambda_0 .. lambda_16 as bytes [m][64] for the transposition

  const unsigned long long nchunks = (B + FPW - 1) / FPW;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + wid;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long chunk = wave; chunk < nchunks; chunk += nwaves) {
    const unsigned long long first = chunk * FPW;
    const int frames = static_cast<int>((B - first) < static_cast<unsigned long long>(FPW) ? (B - first) : FPW);
    const unsigned long long group = 2 * chunk + (f >> 5);
    const int fi = f & 31;
    const uint8_t *src = synd + ((group >> 6) * T2 * 64 + (group & 63)) * 32 + 4 * (fi & 7) + (fi >> 3);
    uint32_t sl[T2];  // log S_j
    uint32_t any = 0;
#pragma unroll
    for (int j = 0; j < T2; ++j) {
      const uint32_t v = src[j * 2048];
      any |= v;
      sl[j] = v;
    }
#pragma unroll
    for (int j = 0; j < T2; ++j) sl[j] = lg2[sl[j]];
    const unsigned long long smask = dbg_stop == 1 ? 0ull : __ballot(any != 0 && f < frames);
    const bool mine = (smask >> lane) & 1ull;
    if (f < frames && !mine) {  // a codeword: done (cyclic.h:225-231)
      if (nerr_out) nerr_out[first + f] = 0;
      if (status_out) status_out[first + f] = CC_FRAME_OK;
    }
    if (lane == 0) mask[chunk] = smask;
    if (smask == 0) continue;  // wave-uniform

    uint32_t ll[NC], P[NC];  // log lambda_m; log of B's coefficients, renamed every step
#pragma unroll
    for (int m = 0; m < NC; ++m) ll[m] = P[m] = m == 0 ? 0u : kLogZero;
    int l = 0, lw = 0;
    for_each_index<T2>([&](auto step) {
      constexpr int i = decltype(step)::value;
      // B <- B x: coefficient k of B is P[(k - (i + 1)) mod NC] from here on (k = 0 takes the slot of k = T2, zero)
      auto Bk = [&](int k) -> uint32_t & { return P[((k - (i + 1)) % NC + NC) % NC]; };
      uint32_t d = ex[sl[i]];
      // discrepancy :139-141; lambda_m = 0 beyond L <= i
#pragma unroll
      for (int m0 = 1; m0 <= i; m0 += 4) {
        if (m0 <= lw) {  // wave-uniform
#pragma unroll
          for (int m = m0; m < m0 + 4 && m <= i; ++m) d ^= ex[ll[m] + sl[i - m]];
        }
      }
      const bool upd = mine && d != 0;
      const bool grow = upd && 2 * l <= i;  // :145 (rho = 0)
      const uint32_t ld = lg2[d];
      const uint32_t linv = nn - ld;  // log of d^-1 (or nn for d = 1: wrapped below)
      const int lnew = grow ? i + 1 - l : l;
      const int cap = static_cast<int>(wave_umax(upd ? static_cast<uint32_t>(lnew) : 0u));
      if (__any(upd)) {
        // lambda += d B, and where the register grows B := lambda_old / d
#pragma unroll
        for (int m0 = 0; m0 <= i + 1 && m0 < NC; m0 += 4) {
          if (m0 <= cap) {  // wave-uniform
            uint32_t nv[4] = {0, 0, 0, 0}, ln[4] = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int m = m0 + u;
              if (m <= i + 1 && m < NC) nv[u] = ex[ll[m]] ^ ex[ld + Bk(m)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) ln[u] = lg2[nv[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // selects, not branches: upd / grow differ from lane to lane
              const int m = m0 + u;
              if (m <= i + 1 && m < NC) {
                const uint32_t lold = ll[m];
                uint32_t qv = lold + linv;
                qv = qv >= nn ? qv - nn : qv;
                qv = lold >= kLogZero ? kLogZero : qv;
                Bk(m) = grow ? qv : Bk(m);
                ll[m] = upd ? ln[u] : lold;
              }
            }
          }
        }
      }
      if (grow) l = lnew;
      lw = cap > lw ? cap : lw;
    });
    int deg = 0;
#pragma unroll
    for (int m = T2; m >= 1; --m)
      if (deg == 0 && ll[m] != kLogZero) deg = m;
    if (f < frames) meta[first + f] = static_cast<uint16_t>(deg | (l << 8));
#pragma unroll
    for (int m = 0; m < NC; ++m) llg[(chunk * NC + m) * FPW + f] = static_cast<uint16_t>(ll[m]);
    // lambda_0 .. lambda_16 as planes for the Chien kernel (see chunk_bm_kernel)
#pragma unroll
    for (int m = 0; m < 17 && m < NC; ++m) LV[m * FPW + f] = ex[ll[m]];
    if (lane < 2 * 17) {
      const int m = lane >> 1, half = lane & 1;
      uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (m < NC) {
        const uint4 *row = reinterpret_cast<const uint4 *>(LV + m * FPW + 32 * half);
        const uint4 a = row[0], b = row[1];
        w[0] = a.x, w[1] = a.y, w[2] = a.z, w[3] = a.w, w[4] = b.x, w[5] = b.y, w[6] = b.z, w[7] = b.w;
        bitplane::butterfly(w);
      }
      const unsigned long long g = 2 * chunk + half;
      uint4 *dst = lamp + (((g >> 6) * 17 + m) * 64 + (g & 63)) * 2;
      dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
      dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// roots, error values, re-check and the patch of `out` (which already holds the received words), one dirty frame
// of a 64-frame chunk at a time.  All polynomial arithmetic on logs with log 0 = 512 (no zero tests): a term
// lambda_m X^-m is ex[log lambda_m + (m * log X^-1 mod nn)], the exponent advancing by one add + one wrap per
// coefficient; wave-uniform coefficients come from a register by v_readlane, not from LDS.
__global__ void __launch_bounds__(256)
chunk_fix_kernel(const AlgebraicTables *__restrict__ T, int alg, const uint8_t *__restrict__ synd,
                 const uint16_t *__restrict__ llg, const uint16_t *__restrict__ meta,
                 const unsigned long long *__restrict__ mask, const uint32_t *__restrict__ roots,
                 const uint32_t *__restrict__ nleft, const uint32_t *__restrict__ er_off, int plane_deg,
                 uint8_t *__restrict__ out, int32_t *__restrict__ nerr_out, int32_t *__restrict__ status_out,
                 unsigned long long B) {
  if (nleft && *nleft == 0) return;  // nothing was handed on by chunk_fixl_kernel (the usual case)
  // exl: antilog table long enough for a Horner / Chien exponent that is never wrapped -- index = log of the
  // coefficient (<= 254, or kLongZero for a zero coefficient) + up to 32 steps of <= 254; zero above kLongZero
  constexpr uint32_t kLongZero = 8448, kLongSize = 16640;
  __shared__ __attribute__((aligned(16))) uint8_t smem[1792 + 4 * 320 + kLongSize];
  uint8_t *ex = smem;                                         // [1024]
  uint16_t *lg2 = reinterpret_cast<uint16_t *>(smem + 1024);  // [256]
  uint8_t *lg = smem + 1536;                                  // [256] plain log table (log 0 = 0)
  uint8_t *exl = smem + 1792 + 4 * 320;
  for (uint32_t i = threadIdx.x; i < kLongSize; i += 256) exl[i] = i < kLongZero ? T->exp[i % 255u] : 0;
  for (int i = threadIdx.x; i < 1024; i += 256) ex[i] = i < 512 ? T->exp[i] : 0;
  lg2[threadIdx.x] = threadIdx.x ? T->log[threadIdx.x] : kLogZero;
  lg[threadIdx.x] = T->log[threadIdx.x];
  __syncthreads();
  const int dbg_stop = alg >> 8;
  alg &= 0xFF;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = T->n, nn = n, t2 = T->nroots, nc = t2 + 1;
  const bool is_rs = T->family == CC_FAMILY_RS;
  uint8_t *base = smem + 1792 + wid * 320;
  uint16_t *CSL = reinterpret_cast<uint16_t *>(base);  // u16 [64] log S_j
  uint8_t *CS = base + 128, *RP = base + 192, *VAL = base + 256;

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
    xinv[c] = static_cast<uint32_t>((nn - (p % nn)) % nn);
  }

  const unsigned long long nchunks = (B + 63) / 64;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + wid;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  const int jl = lane < t2 ? lane : t2 - 1, ml = lane < nc ? lane : nc - 1;
  for (unsigned long long chunk = wave; chunk < nchunks; chunk += nwaves) {
    const unsigned long long first = chunk * 64;
    unsigned long long todo = mask[chunk];
    if (todo == 0) continue;
    // root masks of the chunk's two groups (bitslice_chien_kernel): word p of a group, bit 8 (f & 3) + (f >> 2) = frame f
    uint32_t rw[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < 4; ++c) rw[h][c] = roots[(2 * chunk + h) * 256 + lane + 64 * c];
    auto pop = [](unsigned long long &m) {
      const int i = m ? __builtin_ctzll(m) : -1;
      m &= m - 1;
      return i;
    };
    auto fetch = [&](int f, uint32_t &sv, uint32_t &cll, uint32_t &md) {
      const unsigned long long group = 2 * chunk + (f >> 5);
      const int fi = f & 31;
      sv = synd[((group >> 6) * t2 * 64 + (group & 63)) * 32 + 4 * (fi & 7) + (fi >> 3) + jl * 2048];
      cll = llg[(chunk * nc + ml) * 64 + f];
      md = meta[first + f];
    };
    int s0 = pop(todo);
    uint32_t sv0 = 0, cll0 = 0, md0 = 0;
    if (s0 >= 0) fetch(s0, sv0, cll0, md0);
    while (s0 >= 0) {
      const int s = s0;
      const unsigned long long frame = first + s;
      const uint32_t sv = sv0, cll = cll0;
      const int deg = __builtin_amdgcn_readfirstlane(md0) & 0xFF, len = __builtin_amdgcn_readfirstlane(md0) >> 8;
      s0 = pop(todo);
      if (s0 >= 0) fetch(s0, sv0, cll0, md0);  // the next frame's operands travel while this one is worked on
      uint32_t sym[4] = {0, 0, 0, 0}, corr[4] = {0, 0, 0, 0};
      int status = CC_FRAME_OK, nerr = 0;
      if (lane < t2) {
        CS[lane] = static_cast<uint8_t>(sv);
        CSL[lane] = lg2[sv];
      }
      // the PGZ / Euklid tags run as bounded-distance decoding: locator degree within capability, (2t + rho) / 2
      const int rho = er_off ? static_cast<int>(er_off[frame + 1] - er_off[frame]) : 0;  // wave-uniform
      // (erasures reach this chain with the BM tag only: Euklid's integer stop rule, hard_decision.h:176, lets its locator
      //  be one longer than the capability when rho is odd, and there its answer is not Berlekamp-Massey's)
      if (alg != CC_ALG_BM && 2 * deg - rho > t2) status = CC_FRAME_LOCATOR;
      if (deg < 1) status = CC_FRAME_LOCATOR;  // cyclic.h:145-147
      if (dbg_stop == 2) status = CC_FRAME_LOCATOR;

      // root search: position p is in error iff lambda(alpha^-p) = 0  (cyclic.h:126-150)
      uint32_t isroot[4] = {0, 0, 0, 0}, rank[4] = {0, 0, 0, 0};
      if (status == CC_FRAME_OK) {
        uint32_t acc[4] = {0, 0, 0, 0};
        if (deg <= plane_deg) {  // searched on planes already (16, or 24 for calls with erasures)
          const int fi = s & 31, bit = 8 * (fi & 3) + (fi >> 2);
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = (((s >> 5) ? rw[1][c] : rw[0][c]) >> bit) & 1u ? 0u : 1u;
        } else {
          uint32_t e[4] = {0, 0, 0, 0};
          for (int m = 0; m <= deg; ++m) {
            const uint32_t l0 = __builtin_amdgcn_readlane(cll, m), lm = l0 >= kLogZero ? kLongZero : l0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              acc[c] ^= exl[lm + e[c]];
              e[c] += xinv[c];
            }
          }
        }
        uint32_t count = 0;
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          isroot[c] = (valid[c] && acc[c] == 0) ? 1u : 0u;
          const unsigned long long mk = __ballot(isroot[c] != 0);
          rank[c] = count + static_cast<uint32_t>(__builtin_popcountll(mk & below));
          count += static_cast<uint32_t>(__builtin_popcountll(mk));
        }
        nerr = static_cast<int>(count);
        if (nerr != deg) status = CC_FRAME_LOCATOR;  // cyclic.h:134-143
        if (status == CC_FRAME_OK) {  // the symbols to patch: fetched now, needed after the error values
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (isroot[c]) sym[c] = out[frame * n + lane + 64 * c];
        }
      }
      if (dbg_stop == 3) status = CC_FRAME_LOCATOR;

      // error values: bch.h:80-83 (all ones) / Forney for rs.h:41-78, one lane per located error
      if (status == CC_FRAME_OK && is_rs) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (isroot[c]) RP[rank[c]] = static_cast<uint8_t>(lane + 64 * c);
        uint32_t om = 0;  // omega_j = sum_{m<=j} S_{j-m} lambda_m, j < deg
        for (int m = 0; m <= deg; ++m) {
          const uint32_t lm = __builtin_amdgcn_readlane(cll, m);
          const bool in = lane >= m && lane < deg && lane - m < t2;
          om ^= in ? ex[lm + CSL[in ? lane - m : 0]] : 0u;
        }
        const uint32_t oml = lg2[om];
        uint32_t y = 0;
        if (lane < deg) {
          const uint32_t p = RP[lane];
          const uint32_t xi = p ? static_cast<uint32_t>(nn) - p : 0u;  // log X^-1
          uint32_t x2 = 2 * xi;
          x2 = umin32(x2, x2 - static_cast<uint32_t>(nn));
          uint32_t num = 0, den = 0, e = 0;
          for (int j = 0; j < deg; ++j) {  // omega(X^-1)
            const uint32_t l0 = __builtin_amdgcn_readlane(oml, j);
            num ^= exl[(l0 >= kLogZero ? kLongZero : l0) + e];
            e += xi;
          }
          e = 0;
          for (int m = 1; m <= deg; m += 2) {  // lambda'(X^-1) = sum_{m odd} lambda_m X^-(m-1)
            const uint32_t l0 = __builtin_amdgcn_readlane(cll, m);
            den ^= exl[(l0 >= kLogZero ? kLongZero : l0) + e];
            e += x2;
          }
          y = (num && den) ? ex[lg[num] + nn - lg[den]] : 0u;
        }
        VAL[lane] = static_cast<uint8_t>(y);
#pragma unroll
        for (int c = 0; c < 4; ++c) corr[c] = isroot[c] ? VAL[rank[c]] : 0u;
      } else if (status == CC_FRAME_OK) {
#pragma unroll
        for (int c = 0; c < 4; ++c) corr[c] = isroot[c];
      }
      // re-check (cyclic.h:243-248): decided by L = deg lambda (proof in algebraic.hip; with erasures it needs the error
      // VALUES to be the ones the syndromes determine, which a binary code's all-ones are not), evaluated otherwise
      if (status == CC_FRAME_OK && (len != deg || (rho > 0 && !is_rs))) {
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
              want |= static_cast<uint32_t>(CS[j0 + jj]) << (8 * jj);
            }
          }
          mismatch |= lane63(wave_xor(packed)) ^ want;
        }
        if (mismatch != 0) status = CC_FRAME_RECHECK;
      }
      const bool ok = status == CC_FRAME_OK;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (ok && corr[c]) out[frame * n + lane + 64 * c] = static_cast<uint8_t>(sym[c] ^ corr[c]);
      if (lane == 0) {
        if (nerr_out) nerr_out[frame] = ok ? nerr : -1;
        if (status_out) status_out[frame] = status;
      }
      __builtin_amdgcn_wave_barrier();  // the scratch arrays are reused by the next frame
    }
  }
}


// The correction stage with ONE LANE PER FRAME (64 frames of a chunk per wavefront), everything of a frame in that
// lane's registers: 16 syndrome logs, 17 locator logs, the 16 omega logs it computes, and its 255-bit root vector
// (eight words of bitslice_roots_transpose_kernel's output).  The errors of a frame are taken off the root words
// (lowest set bit) into a list, then served one per trip; each costs the two Forney sums with compile-time
// coefficient indices and unwrapped exponents on the long antilog table.  No cross-lane traffic at all; a trip of
// the error loop serves up to 64 frames.  Frames that need the general treatment -- locator longer than 16, or L != deg (the re-check has to be
// evaluated) -- go to chunk_fix_kernel through `left`.
// MD = longest locator served here: 16 (= t of the largest code; calls without erasures), 24 for calls with erasures
template <int MD>
__global__ void __launch_bounds__(256)
chunk_fixl_kernel(const AlgebraicTables *__restrict__ T, int alg, const uint8_t *__restrict__ synd,
                  const uint16_t *__restrict__ llg, const uint16_t *__restrict__ meta,
                  const unsigned long long *__restrict__ mask, const uint32_t *__restrict__ rootsT,
                  unsigned long long *__restrict__ left, uint32_t *__restrict__ nleft, const uint32_t *__restrict__ er_off,
                  uint8_t *__restrict__ out, int32_t *__restrict__ nerr_out, int32_t *__restrict__ status_out,
                  unsigned long long B) {
  // exl: alpha^i for i < kZ, zero from kZ on; kZ marks a zero operand (log of 0), kZ + kZ still inside the table
  constexpr uint32_t kZ = 8448, kLongSize = 2 * kZ + 64, kN = 255;
  __shared__ __attribute__((aligned(16))) uint8_t smem[kLongSize + 512 + 256 + 4 * 2 * MD * 64];
  uint8_t *exl = smem;
  uint16_t *lgz = reinterpret_cast<uint16_t *>(smem + kLongSize);  // [256] log, kZ for 0
  uint8_t *lg = smem + kLongSize + 512;                            // [256] plain log table (log 0 = 0)
  uint8_t *plist = smem + kLongSize + 512 + 256;                   // [wavefront][2 MD][64] error positions, symbols
  for (uint32_t i = threadIdx.x; i < kLongSize; i += 256) exl[i] = i < kZ ? T->exp[i % 255u] : 0;
  lgz[threadIdx.x] = static_cast<uint16_t>(threadIdx.x ? T->log[threadIdx.x] : kZ);
  lg[threadIdx.x] = T->log[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), f = lane;
  const int n = T->n, t2 = T->nroots, nc = t2 + 1;
  const bool is_rs = T->family == CC_FAMILY_RS;
#ifdef CC_AMD_EXPERIMENTS  // CC_EXP_FIXL bits: 1 no load/store of the symbols (16 no load, 32 no store), 2 no Forney sums, 4 no error loop, 8 no omega
  const int xf = alg >> 8;
  alg &= 0xFF;
#else
  constexpr int xf = 0;
#endif

  const unsigned long long nchunks = (B + 63) / 64;
  const unsigned long long wave = static_cast<unsigned long long>(blockIdx.x) * 4 + wid;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long chunk = wave; chunk < nchunks; chunk += nwaves) {
    const unsigned long long first = chunk * 64, frame = first + f;
    const unsigned long long smask = mask[chunk];
    if (smask == 0) {
      if (lane == 0) left[chunk] = 0;
      continue;
    }
    const bool dirty = (smask >> lane) & 1ull;
    const uint32_t md = dirty ? meta[frame] : 0u;
    const int deg = md & 0xFF, len = md >> 8;
    const int rho = (er_off && dirty) ? static_cast<int>(er_off[frame + 1] - er_off[frame]) : 0;
    // chunk_fix_kernel's business: long locators, L != deg lambda, and binary codes with erasures (re-check to be evaluated)
    const bool general = dirty && (deg > MD || len != deg || (rho > 0 && !is_rs));
    const unsigned long long lmask = __ballot(general);
    if (lane == 0) {
      left[chunk] = lmask;
      if (lmask) atomicAdd(nleft, 1u);
    }
    int status = CC_FRAME_OK;
    if (alg != CC_ALG_BM && 2 * deg - rho > t2) status = CC_FRAME_LOCATOR;  // bounded-distance decoding (rho = 0 here: see chunk_fix_kernel)
    if (deg < 1) status = CC_FRAME_LOCATOR;                            // cyclic.h:145-147
    const unsigned long long group = 2 * chunk + (f >> 5);
    const int fi = f & 31, bit = 8 * (fi & 3) + (fi >> 2);
    uint32_t R[8];  // bit j of R[k]: position 32 k + j is a root
#pragma unroll
    for (int k = 0; k < 8; ++k) R[k] = rootsT[(group * 8 + k) * 32 + bit];
    R[7] &= 0x7FFFFFFFu;  // there is no position 255
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) cnt += __builtin_popcount(R[k]);
    if (static_cast<int>(cnt) != deg) status = CC_FRAME_LOCATOR;  // cyclic.h:134-143
    const bool live = dirty && !general;
    const bool fixing = live && status == CC_FRAME_OK;
    if (live) {
      if (nerr_out) nerr_out[frame] = fixing ? static_cast<int>(cnt) : -1;
      if (status_out) status_out[frame] = status;
    }
    if (__ballot(fixing) == 0) continue;  // wave-uniform

    // the error positions of every lane as a list in LDS ([error][lane], bytes): taking them off the root words inside
    // the Forney loop would cost a trip per (word, error-in-word) of the WORST lane -- about 36 trips for 8 errors
    // per frame -- instead of one per error of the worst lane
    uint8_t *PL = plist + wid * (2 * MD * 64);  // [MD][64] positions, [MD][64] symbols
    uint32_t have = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      uint32_t w = fixing ? R[k] : 0u;
      while (__any(w != 0)) {
        if (w != 0) {
          PL[have * 64 + lane] = static_cast<uint8_t>(32 * k + __builtin_ctz(w));
          ++have;
        }
        w &= w - 1;
      }
    }
    const int emax = (xf & 4) ? 0 : static_cast<int>(wave_umax(have));  // <= MD: a fixing lane has cnt = deg <= MD roots
    // All symbols to patch are requested here (four at a time, into LDS), long before the first one is stored: the
    // memory counter retires in order, so a load issued after a store would wait for that store's acknowledgement on
    // every trip (measured: 272 us for the kernel with load and store alternating, 150 / 165 us with only one of them).
    const uint8_t *obase = out + first * n;  // wave-uniform base + 32-bit lane offset
    uint32_t ll[MD + 1], ol[MD], sl[MD];  // log lambda_m, log omega_j, log S_j (kZ for zero); j < MD: omega_j, j < deg <= MD, needs no more
    {
      for (int e0 = 0; e0 < MD; e0 += 4) {
        uint32_t sy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool has = static_cast<uint32_t>(e0 + u) < have;
          sy[u] = has ? obase[static_cast<uint32_t>(f * n) + PL[(e0 + u) * 64 + lane]] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) PL[(MD + e0 + u) * 64 + lane] = static_cast<uint8_t>(sy[u]);
      }
      if (is_rs) {
        const uint8_t *sb = synd + ((group >> 6) * t2 * 64 + (group & 63)) * 32 + 4 * (fi & 7) + (fi >> 3);
#pragma unroll
        for (int j = 0; j < MD; ++j) sl[j] = j < t2 ? sb[j * 2048] : 0u;
#pragma unroll
        for (int m = 0; m < MD + 1; ++m) ll[m] = m < nc ? llg[(chunk * nc + m) * 64 + f] : kLogZero;
      }
    }
    if (is_rs) {
#pragma unroll
      for (int m = 0; m < MD + 1; ++m) ll[m] = ll[m] >= kLogZero ? kZ : ll[m];
#pragma unroll
      for (int j = 0; j < MD; ++j) sl[j] = lgz[sl[j]];
      const uint32_t dmax = static_cast<uint32_t>(wave_umax(fixing ? static_cast<uint32_t>(deg) : 0u));
      // omega_j = sum_{m <= j} lambda_m S_{j-m} for j < deg (S lambda mod x^deg)
#pragma unroll
      for (int j = 0; j < MD; ++j) {
        uint32_t om = 0;
        if (static_cast<uint32_t>(j) < dmax && !(xf & 8)) {  // wave-uniform
#pragma unroll
          for (int m = 0; m <= j; ++m) om ^= exl[ll[m] + sl[j - m]];
        }
        ol[j] = j < deg ? static_cast<uint32_t>(lgz[om]) : kZ;
      }
    }
    for (int e = 0; e < emax; ++e) {
      const bool has = static_cast<uint32_t>(e) < have;
      const uint32_t p = has ? PL[e * 64 + lane] : 0u;
      const uint32_t sym = PL[(MD + e) * 64 + lane];
      uint32_t y = 1;  // bch.h:80-83
      if (is_rs && !(xf & 2)) {  // Forney, rs.h:41-78
        const uint32_t xi = p ? kN - p : 0u;  // log X^-1
        uint32_t x2 = 2 * xi;
        x2 = umin32(x2, x2 - kN);
        uint32_t num = 0, den = 0, ee = 0;
#pragma unroll
        for (int j = 0; j < MD; ++j) {  // omega(X^-1)
          num ^= exl[ol[j] + ee];
          ee += xi;
        }
        ee = 0;
#pragma unroll
        for (int m = 1; m < MD + 1; m += 2) {  // lambda'(X^-1) = sum_{m odd} lambda_m X^-(m-1)
          den ^= exl[ll[m] + ee];
          ee += x2;
        }
        y = (num && den) ? exl[lg[num] + kN - lg[den]] : 0u;
      }
      // (an atomic XOR on the surrounding dword instead of the load / store pair was measured slower: 942 vs 1024 M)
      if (has && y && !(xf & 33)) out[frame * n + p] = static_cast<uint8_t>(sym ^ y);
    }
    __builtin_amdgcn_wave_barrier();  // the list is reused by the next chunk
  }
}

}  // namespace

bool algebraic_chunk_supported(const cc_code *code, bool erasures) {
  static const bool disabled = [] {
    const char *e = std::getenv("CC_AMD_NO_CHUNK");
    return e && e[0] == '1';
  }();
  if (disabled) return false;
  // with erasures: the bit-plane chain only (its LDS-form Berlekamp-Massey kernel starts the recurrence per lane), and the
  // BM tag only.  Euklid: with an odd number of erasures the remainder sequence stops one step later than the capability
  // (integer (2t + rho) / 2, hard_decision.h:176) and the reference decodes frames with 2e + rho = 2t + 1 -- to ITS
  // answer, which is not Berlekamp-Massey's (measured: RS(255,223), rho = 31, one error); launch_algebraic runs this
  // chain FIRST for the Euklid tag too and Sugiyama itself over the frames it leaves undecoded.  PGZ: two trials without
  // erasures for BCH, refused for RS.
  if (erasures) return bitslice_supported(code) && code->desc.algorithm == CC_ALG_BM;
  // measured (profiles/tools/rs_bench.py): with few syndromes the per-frame work is dominated by the frame's
  // load/store latency and the one-wavefront-per-frame kernel with its higher occupancy wins
  // (BCH(255,231), 6 syndromes: 1128 vs 899 M frames/s); with 32 syndromes this kernel wins (309 vs 220)
  if (code->tab.roots.size() < 8 && !bitslice_supported(code)) return false;
  // The Euklid tag without erasures is served as bounded-distance decoding on the Berlekamp-Massey locator, like PGZ:
  // the remainder sequence of hard_decision.h:157-196 stops at deg r < t, so its sigma has degree <= t, and a frame
  // comes back corrected exactly when a codeword lies within t symbols of it -- then the key equation has ONE solution
  // of degree <= t up to a scalar, the one Berlekamp-Massey finds (L <= t), and the corrected word is the same; a
  // longer register (L > t) is refused here as the reference's root-count / re-check refuses its sigma.  Which of the
  // failure texts a hopeless frame gets is the only thing that can differ.  With erasures the Sugiyama kernel of
  // algebraic.hip runs (one wavefront per frame).  tests: hard_golden / seeded (oracle = the reference's Euklid).
  return code->desc.algorithm == CC_ALG_BM || code->desc.algorithm == CC_ALG_PGZ || code->desc.algorithm == CC_ALG_EUKLID;
}

// CC_AMD_ALG_STOP=1|2|3 (builds with -DCC_AMD_EXPERIMENTS only): stop the chain after syndromes / Berlekamp-Massey /
// root search -- phase timing for profiles/tools/rs_bench.py; the product library always runs the whole chain
static int alg_stop_stage() {
#ifdef CC_AMD_EXPERIMENTS
  static const int v = [] {
    const char *e = std::getenv("CC_AMD_ALG_STOP");
    return e ? std::atoi(e) : 0;
  }();
  return v;
#else
  return 0;
#endif
}

static int fixl_exp() {  // experiment bits of chunk_fixl_kernel (builds with -DCC_AMD_EXPERIMENTS only)
#ifdef CC_AMD_EXPERIMENTS
  static const int v = [] {
    const char *e = std::getenv("CC_EXP_FIXL");
    return e ? std::atoi(e) << 8 : 0;
  }();
  return v;
#else
  return 0;
#endif
}

template <int FPW>
static int launch_chunk_fpw(const cc_code *code, bool float_in, const void *d_in, uint8_t *d_out, int32_t *d_nerr,
                            int32_t *d_status, size_t B, hipStream_t stream) {
  const int t2 = static_cast<int>(code->tab.roots.size());
  const size_t lds = 1792 + 4 * static_cast<size_t>(chunk_layout(t2, FPW).bytes);
  const unsigned long long chunks = (B + FPW - 1) / FPW;
  const unsigned long long blocks_needed = (chunks + 3) / 4;
  unsigned long long per_cu = (160 * 1024) / lds;
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  const unsigned long long max_grid = static_cast<unsigned long long>(code->num_cus) * per_cu;
  const int grid = static_cast<int>(blocks_needed < max_grid ? blocks_needed : max_grid);
  const unsigned long long Bq = B;
  const int alg_arg = code->desc.algorithm | (alg_stop_stage() << 8);
  hipError_t e = hipSuccess;
  if (float_in) {
    if (lds > 48 * 1024)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&algebraic_chunk_kernel<true, FPW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e == hipSuccess)
      hipLaunchKernelGGL((algebraic_chunk_kernel<true, FPW>), dim3(grid), dim3(256), lds, stream, code->d_alg,
                         alg_arg, d_in, d_out, d_nerr, d_status, Bq);
  } else {
    if (lds > 48 * 1024)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&algebraic_chunk_kernel<false, FPW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e == hipSuccess)
      hipLaunchKernelGGL((algebraic_chunk_kernel<false, FPW>), dim3(grid), dim3(256), lds, stream, code->d_alg,
                         alg_arg, d_in, d_out, d_nerr, d_status, Bq);
  }
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "algebraic chunk kernel launch");
  return CC_OK;
}

// syndromes on bit planes (bitslice.hip), Berlekamp-Massey over chunks of 64 frames, root search on planes, corrections
static int launch_chunk_bitsliced(const cc_code *code, bool float_in, const void *d_in, const uint16_t *d_er,
                                  const uint32_t *d_er_off, uint8_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B,
                                  hipStream_t stream) {
  const int t2 = static_cast<int>(code->tab.roots.size()), nc = t2 + 1;
  const unsigned long long G = (B + 31) / 32, chunks = (B + 63) / 64;
  const size_t G64 = static_cast<size_t>((G + 63) / 64) * 64;  // syndromes, locators and root masks are laid out in blocks of 64 groups
  auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  const size_t synd_bytes = G64 * t2 * 32;
  const size_t llg_bytes = up(static_cast<size_t>(chunks) * nc * 64 * 2), meta_bytes = up(static_cast<size_t>(chunks) * 64 * 2);
  const size_t mask_bytes = up(static_cast<size_t>(chunks) * 8);
  // calls with erasures: combined (erasure x error) locators up to degree 24 on the lane-per-frame path -- 25 coefficient
  // planes, the long instantiations of the root search and of the corrector; without erasures a correctable locator
  // has degree <= t <= 16
  const bool long_loc = d_er_off != nullptr && t2 > 16;
  const int ncoef = long_loc ? 25 : 17;
  const size_t lamp_bytes = G64 * ncoef * 32, roots_bytes = G64 * 256 * 4, left_bytes = mask_bytes;
  uint8_t *ws = nullptr;  // stream-ordered and pool-cached: no device-wide synchronisation, no allocation after the first call
  CC_HIP_TRY(workspace_alloc(code, reinterpret_cast<void **>(&ws),
                            synd_bytes + llg_bytes + meta_bytes + mask_bytes + lamp_bytes + 2 * roots_bytes + left_bytes + 256, stream));
  uint8_t *d_synd = ws;
  uint16_t *d_llg = reinterpret_cast<uint16_t *>(d_synd + synd_bytes);
  uint16_t *d_meta = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(d_llg) + llg_bytes);
  unsigned long long *d_mask = reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(d_meta) + meta_bytes);
  uint8_t *d_lamp = reinterpret_cast<uint8_t *>(d_mask) + mask_bytes;
  uint8_t *d_roots = d_lamp + lamp_bytes;
  unsigned long long *d_left = reinterpret_cast<unsigned long long *>(d_roots + roots_bytes);
  uint32_t *d_nleft = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(d_left) + left_bytes);
  uint8_t *d_rootsT = reinterpret_cast<uint8_t *>(d_nleft) + 256;
  int rc = launch_bitslice_syndromes(code, float_in, d_in, d_out, d_synd, B, stream);
  if (rc == CC_OK) {
    const int dbg_stop = alg_stop_stage();
    const unsigned long long Bq = B, blocks_needed = (chunks + 3) / 4;
    const size_t lds = 1536 + 4 * static_cast<size_t>(bm_layout(t2).bytes);
    unsigned long long per_cu = (160 * 1024) / lds;
    if (per_cu < 1) per_cu = 1;
    unsigned long long max_grid = static_cast<unsigned long long>(code->num_cus) * per_cu;
    int grid = static_cast<int>(blocks_needed < max_grid ? blocks_needed : max_grid);
    hipError_t e = hipSuccess;
    if (lds > 48 * 1024)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&chunk_bm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(lds));
    if (e == hipSuccess) {
      const unsigned long long reg_cap = static_cast<unsigned long long>(code->num_cus) * 3;
      const int reg_grid = static_cast<int>(blocks_needed < reg_cap ? blocks_needed : reg_cap);
      if (t2 == 32 && !d_er_off)  // (with erasures the recurrence starts per lane at i = rho: the LDS form below)
        hipLaunchKernelGGL((chunk_bm_reg_kernel<32>), dim3(reg_grid), dim3(256), 0, stream, code->d_alg, dbg_stop, d_synd,
                           d_llg, d_meta, d_mask, reinterpret_cast<uint4 *>(d_lamp), d_nleft, d_nerr, d_status, Bq);
      else if (t2 == 16 && !d_er_off)
        hipLaunchKernelGGL((chunk_bm_reg_kernel<16>), dim3(reg_grid), dim3(256), 0, stream, code->d_alg, dbg_stop, d_synd,
                           d_llg, d_meta, d_mask, reinterpret_cast<uint4 *>(d_lamp), d_nleft, d_nerr, d_status, Bq);
      else
        hipLaunchKernelGGL(chunk_bm_kernel, dim3(grid), dim3(256), lds, stream, code->d_alg, dbg_stop, d_synd, d_er, d_er_off,
                           d_llg, d_meta, d_mask, reinterpret_cast<uint4 *>(d_lamp), ncoef, d_nleft, d_nerr, d_status, Bq);
      e = hipGetLastError();
    }
    if (e == hipSuccess && launch_bitslice_chien(d_lamp, d_roots, B, long_loc, stream) != CC_OK) e = hipErrorLaunchFailure;
    if (e == hipSuccess) {
      max_grid = static_cast<unsigned long long>(code->num_cus) * 8;
      grid = static_cast<int>(blocks_needed < max_grid ? blocks_needed : max_grid);
      const bool four = dbg_stop == 0;
      if (four) {  // one lane per frame; what it cannot settle goes on through d_left
        if (launch_bitslice_roots_transpose(d_roots, d_rootsT, B, stream) != CC_OK) e = hipErrorLaunchFailure;
        if (e == hipSuccess) {
          static const int fixl_per_cu[2] = {[] {  // resident workgroups per CU: registers and LDS of the built kernels
                                               int v = 0;
                                               if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, chunk_fixl_kernel<16>, 256, 0) != hipSuccess || v < 1) v = 3;
                                               return v;
                                             }(),
                                             [] {
                                               int v = 0;
                                               if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, chunk_fixl_kernel<24>, 256, 0) != hipSuccess || v < 1) v = 2;
                                               return v;
                                             }()};
          const unsigned long long lcap = static_cast<unsigned long long>(code->num_cus) * fixl_per_cu[long_loc];
          const int lgrid = static_cast<int>(blocks_needed < lcap ? blocks_needed : lcap);
          if (long_loc)
            hipLaunchKernelGGL(chunk_fixl_kernel<24>, dim3(lgrid), dim3(256), 0, stream, code->d_alg,
                               code->desc.algorithm | fixl_exp(), d_synd, d_llg, d_meta, d_mask,
                               reinterpret_cast<const uint32_t *>(d_rootsT), d_left, d_nleft, d_er_off, d_out, d_nerr, d_status, Bq);
          else
            hipLaunchKernelGGL(chunk_fixl_kernel<16>, dim3(lgrid), dim3(256), 0, stream, code->d_alg,
                               code->desc.algorithm | fixl_exp(), d_synd, d_llg, d_meta, d_mask,
                               reinterpret_cast<const uint32_t *>(d_rootsT), d_left, d_nleft, d_er_off, d_out, d_nerr, d_status, Bq);
          e = hipGetLastError();
        }
      }
      if (e == hipSuccess) {
        hipLaunchKernelGGL(chunk_fix_kernel, dim3(grid), dim3(256), 0, stream, code->d_alg,
                           code->desc.algorithm | (dbg_stop << 8), d_synd, d_llg, d_meta, four ? d_left : d_mask,
                           reinterpret_cast<const uint32_t *>(d_roots), four ? d_nleft : nullptr, d_er_off, long_loc ? 24 : 16,
                           d_out, d_nerr, d_status, Bq);
        e = hipGetLastError();
      }
    }
    if (e != hipSuccess) rc = hip_fail(e, "algebraic chunk kernels launch");
  }
  (void)hipFreeAsync(ws, stream);
  return rc;
}

int launch_algebraic_chunk(const cc_code *code, bool float_in, const void *d_in, const uint16_t *d_er,
                           const uint32_t *d_er_off, uint8_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B,
                           hipStream_t stream) {
  if (B == 0) return CC_OK;
  if (bitslice_supported(code))
    return launch_chunk_bitsliced(code, float_in, d_in, d_er, d_er_off, d_out, d_nerr, d_status, B, stream);
  return launch_chunk_fpw<32>(code, float_in, d_in, d_out, d_nerr, d_status, B, stream);
}

}  // namespace ccamd
