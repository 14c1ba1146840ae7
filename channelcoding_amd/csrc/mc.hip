// mc.hip -- batched AWGN Monte-Carlo on the device (replaces the per-frame loop of
// awgn_simulation::operator(), src/simulation/simulation.c++:95-150).
//
//   channel   y = (1 - 2c) + sigma * N(0,1),  sigma = 1/sqrt(2 R 10^(EbN0/10))  (simulation.c++:83-85,
//             :113-125; the reference transmits the all-zero word, i.e. N(1, sigma))
//   noise     Philox4x32-10, key = (seed_lo, seed_hi), counter = (frame_lo, frame_hi, quad, domain);
//             quad q yields the four normals of symbols 4q..4q+3 via two Box-Muller pairs.  A frame's
//             noise depends only on (seed, global frame index): results are independent of how frames
//             are sharded over GPUs or chunked inside a call.
//   messages  (random_codewords) l bits per frame from the same generator, domain 1, then the device
//             encoder.
//   counters  word / bit / failure / undetected / channel-bit errors and the iteration histogram,
//             accumulated in LDS per workgroup and flushed with one 64-bit atomic per counter.
#include <cmath>
#include <mutex>

#include "cc_internal.hpp"
#include "wave_ops.hpp"

namespace ccamd {
namespace {

struct Philox {
  uint32_t c[4];
};

__device__ __forceinline__ Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one v_mad_u64_u32 per product instead of a mul_lo / mul_hi pair (both quarter rate)
    const uint64_t p0 = static_cast<uint64_t>(M0) * c0, p1 = static_cast<uint64_t>(M1) * c2;
    const uint32_t hi0 = static_cast<uint32_t>(p0 >> 32), lo0 = static_cast<uint32_t>(p0);
    const uint32_t hi1 = static_cast<uint32_t>(p1 >> 32), lo1 = static_cast<uint32_t>(p1);
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += W0;
    k1 += W1;
  }
  return Philox{{c0, c1, c2, c3}};
}

// Box-Muller on the hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 take their
// argument in turns, so no range reduction is needed for u2 in [0, 1).
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float &z0, float &z1) {
  const float u1 = (static_cast<float>(a >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
  const float u2 = static_cast<float>(b >> 8) * (1.0f / 16777216.0f);           // [0, 1)
  const float r = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1));  // sqrt(-2 ln u1)
  z0 = r * __builtin_amdgcn_cosf(u2);
  z1 = r * __builtin_amdgcn_sinf(u2);
}

// Message bit j of frame f is bit (j & 127) of Philox counter (f, j >> 7, 1).  G = lanes per frame (power of two
// >= ceil(l / 16)); a lane expands 16 bits to bytes (eight lanes share one Philox block and each evaluates it:
// cheaper than the byte-by-byte loop of one lane per block, which serialised 128 stores).
__global__ void __launch_bounds__(256)
random_bits_kernel(uint8_t *__restrict__ msg, int l, int group_log2, unsigned long long first_frame,
                   unsigned long long frames, uint32_t k0, uint32_t k1) {
  const int G = 1 << group_log2;
  const unsigned long long tid = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const unsigned long long stride = (static_cast<unsigned long long>(gridDim.x) * blockDim.x) >> group_log2;
  const int q = static_cast<int>(tid & static_cast<unsigned long long>(G - 1));
  if (16 * q >= l) return;
  for (unsigned long long f = tid >> group_log2; f < frames; f += stride) {
    const unsigned long long gf = first_frame + f;
    const Philox p = philox4x32_10(static_cast<uint32_t>(gf), static_cast<uint32_t>(gf >> 32), q >> 3, 1u, k0, k1);
    const int w = (q & 7) >> 1;
    const uint32_t word = w == 0 ? p.c[0] : w == 1 ? p.c[1] : w == 2 ? p.c[2] : p.c[3];
    const uint32_t bits = (word >> (16 * (q & 1))) & 0xFFFFu;
    uint8_t *dst = msg + f * l + 16 * q;
    const int count = l - 16 * q < 16 ? l - 16 * q : 16;
    for (int b = 0; b < count; ++b) dst[b] = (bits >> b) & 1u;
  }
}

// G = lanes per frame (power of two >= ceil(n / 4)), lane q of a group draws the four values 4q .. 4q + 3 of its
// frame from Philox counter (frame, q): no division by n anywhere, 16-byte stores except for a ragged tail.
// HARD: the hard decision of y (bit = y < 0, cyclic.h:163-173) as bytes instead of y itself -- what a hard-decision decoder
// takes from the channel: a quarter of the bytes written here and read by the syndrome kernel
template <bool HARD>
__global__ void __launch_bounds__(256)
awgn_kernel(float *__restrict__ llr, const uint8_t *__restrict__ sent, int n, int group_log2,
            unsigned long long first_frame, unsigned long long frames, float sigma, uint32_t k0, uint32_t k1,
            unsigned long long *__restrict__ counters) {
  const int G = 1 << group_log2;
  const unsigned long long tid = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const unsigned long long stride = (static_cast<unsigned long long>(gridDim.x) * blockDim.x) >> group_log2;
  const int qd = static_cast<int>(tid & static_cast<unsigned long long>(G - 1));
  // channel bit errors (hard decision of y differs from the bit sent) are counted where y is made: the counting
  // kernel then never reads the channel values again.  One 64-bit atomic per wavefront at the end.
  unsigned cherr = 0;
  for (unsigned long long f = tid >> group_log2; 4 * qd < n && f < frames; f += stride) {
    const unsigned long long gf = first_frame + f;
    const Philox p = philox4x32_10(static_cast<uint32_t>(gf), static_cast<uint32_t>(gf >> 32), qd, 0u, k0, k1);
    float z[4];
    box_muller(p.c[0], p.c[1], z[0], z[1]);
    box_muller(p.c[2], p.c[3], z[2], z[3]);
    float *dst = llr + f * n + 4 * qd;
    const uint8_t *src = sent ? sent + f * n + 4 * qd : nullptr;
    if (4 * qd + 4 <= n) {
      float x[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool one = src && src[s];
        x[s] = (one ? -1.0f : 1.0f) + sigma * z[s];  // BPSK 0 -> +1
        cherr += (x[s] < 0.0f) != one;
      }
      if (HARD) {
        uint8_t *hd = reinterpret_cast<uint8_t *>(llr) + f * n + 4 * qd;
#pragma unroll
        for (int s = 0; s < 4; ++s) hd[s] = x[s] < 0.0f ? 1 : 0;
      } else {
        // frames are n floats apart, so dst is 4-byte aligned only: four dword stores the compiler may merge
        dst[0] = x[0];
        dst[1] = x[1];
        dst[2] = x[2];
        dst[3] = x[3];
      }
    } else {
      for (int s = 0; 4 * qd + s < n; ++s) {
        const bool one = src && src[s];
        const float x = (one ? -1.0f : 1.0f) + sigma * z[s];
        cherr += (x < 0.0f) != one;
        if (HARD) reinterpret_cast<uint8_t *>(llr)[f * n + 4 * qd + s] = x < 0.0f ? 1 : 0;
        else dst[s] = x;
      }
    }
  }
  if (counters) {
    for (int m = 32; m >= 1; m >>= 1) cherr += __shfl_xor(cherr, m, 64);
    if ((threadIdx.x & 63) == 0 && cherr)
      atomicAdd(&counters[CC_MC_CHANNEL_BIT_ERRORS], static_cast<unsigned long long>(cherr));
  }
}

// Channel + pre-check for high signal-to-noise ratios (one wavefront per frame: 128 < n <= 256, 4 values per lane).
// A frame whose channel hard decision is already a codeword stops in iteration 0 with that word whatever the min-sum
// variant: every check's sign product over the OTHER edges then equals the sign of the edge's own value, so every
// first-iteration message has the sign of y_j, L_j = y_j + (terms of the same sign) keeps it, and hard(L) = hard(y)
// passes the stop test (rule O1: only if that word is all-zero).  Such a frame is counted HERE -- iteration 0, bit
// errors = its channel errors -- and its 4 n bytes of channel values are never written; every other frame (and any
// frame with an exact 0.0 among its values, where the sign argument does not hold) goes to a compact batch for the
// decoder.  At 8 dB 91 % of the BCH(255,231) frames are clean.  Same Philox counters as awgn_kernel: the noise of a
// frame does not depend on the route.
// ctl[1] = frames appended; list[k] = frame index (relative to the chunk) of compact frame k
__global__ void __launch_bounds__(256)
awgn_precheck_kernel(float *__restrict__ llr2, uint32_t *__restrict__ list, uint32_t *__restrict__ ctl,
                     const uint8_t *__restrict__ sent, const uint32_t *__restrict__ colbits, int n, int stop_rule,
                     unsigned long long first_frame, unsigned long long frames, float sigma, uint32_t k0, uint32_t k1,
                     unsigned long long *__restrict__ counters) {
  // A workgroup takes 32 frames per round, eight per wavefront, and reserves the compact-batch slots of all its dirty
  // frames with ONE atomic (a single counter word takes ~10 ns per atomic: one per dirty frame cost 6 ms per 2^20
  // frames at 6 dB, more than the decoder).  Every wavefront runs the same number of rounds (barriers inside).
  constexpr int R = 8;
  __shared__ uint32_t wave_dirty[4], wg_base;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t cb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) cb[s] = colbits[(4 * lane + s) & 255];  // zero beyond the frame
  unsigned long long c_frames = 0, c_bit = 0, c_word = 0, c_cherr = 0;  // lane 0's tallies of the clean frames
  const unsigned long long per_round = static_cast<unsigned long long>(gridDim.x) * 4 * R;
  const unsigned long long rounds = (frames + per_round - 1) / per_round;
  for (unsigned long long rd = 0; rd < rounds; ++rd) {
    const unsigned long long f0 = rd * per_round + (static_cast<unsigned long long>(blockIdx.x) * 4 + wid) * R;
    float x[R][4];
    uint32_t dirty = 0;  // wave-uniform bit mask
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const unsigned long long f = f0 + r;
      if (f >= frames) continue;  // wave-uniform
      const unsigned long long gf = first_frame + f;
      const Philox p = philox4x32_10(static_cast<uint32_t>(gf), static_cast<uint32_t>(gf >> 32), lane, 0u, k0, k1);
      float z[4];
      box_muller(p.c[0], p.c[1], z[0], z[1]);
      box_muller(p.c[2], p.c[3], z[2], z[3]);
      uint32_t synd = 0;
      unsigned nerr = 0, nones = 0;
      bool zero = false;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int j = 4 * lane + s;
        const bool in = j < n;
        const bool one = in && sent && sent[f * n + j];
        x[r][s] = (one ? -1.0f : 1.0f) + sigma * z[s];  // BPSK 0 -> +1
        const bool hb = in && x[r][s] < 0.0f;
        synd ^= hb ? cb[s] : 0u;
        nerr += static_cast<unsigned>(__popcll(__ballot(in && hb != one)));   // wave-uniform counts
        nones += static_cast<unsigned>(__popcll(__ballot(hb)));
        zero |= in && x[r][s] == 0.0f;
      }
      synd = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wave_xor(synd)), 63));
      const bool clean = !__any(zero) && (stop_rule == CC_STOP_PUBLISHED ? nones == 0 : synd == 0);
      c_cherr += nerr;
      if (clean) {
        c_frames += 1;
        c_bit += nerr;
        c_word += nerr ? 1u : 0u;
      } else {
        dirty |= 1u << r;
      }
    }
    const uint32_t mine = static_cast<uint32_t>(__builtin_popcount(dirty));
    if (lane == 0) wave_dirty[wid] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t total = wave_dirty[0] + wave_dirty[1] + wave_dirty[2] + wave_dirty[3];
      wg_base = total ? atomicAdd(&ctl[1], total) : 0u;
    }
    __syncthreads();
    uint32_t at = wg_base;
    for (int w = 0; w < wid; ++w) at += wave_dirty[w];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!((dirty >> r) & 1u)) continue;  // wave-uniform
      if (lane == 0) list[at] = static_cast<uint32_t>(f0 + r);
      float *dst = llr2 + static_cast<unsigned long long>(at) * n + 4 * lane;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (4 * lane + s < n) dst[s] = x[r][s];
      ++at;
    }
    __syncthreads();  // wave_dirty / wg_base are rewritten by the next round
  }
  if (lane == 0) {
    if (c_cherr) atomicAdd(&counters[CC_MC_CHANNEL_BIT_ERRORS], c_cherr);
    if (c_frames) {
      atomicAdd(&counters[CC_MC_FRAMES], c_frames);
      atomicAdd(&counters[CC_MC_ITER_SUM], c_frames);        // one iteration executed each
      atomicAdd(&counters[CC_MC_ITER_HIST + 0], c_frames);   // stopped in iteration 0
      if (c_bit) atomicAdd(&counters[CC_MC_BIT_ERRORS], c_bit);
      if (c_word) {  // the channel's hard decision was ANOTHER codeword: an undetected word error
        atomicAdd(&counters[CC_MC_WORD_ERRORS], c_word);
        atomicAdd(&counters[CC_MC_UNDETECTED], c_word);
      }
    }
  }
}

// Compares the decoder output with the transmitted word (nullptr: the all-zero word), 16 lanes per frame with one
// 16-byte load per lane and buffer (the last lane of a frame takes the 16 bytes that END at n and masks the overlap, so
// nothing is read past a frame), mismatching symbols counted on packed bytes, the sums of a frame combined by a DPP
// row reduction.  The group leaders keep their counters in registers; LDS / global atomics once per wavefront.
__device__ __forceinline__ unsigned nonzero_bytes(uint32_t x) {
  return static_cast<unsigned>(__builtin_popcount((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u));
}
template <int CTRL> __device__ __forceinline__ unsigned dpp_add(unsigned v) {
  return v + static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, false));
}
__global__ void __launch_bounds__(256)
count_kernel(const uint8_t *__restrict__ hard, const uint8_t *__restrict__ sent, const uint16_t *__restrict__ iters,
             const int32_t *__restrict__ status, int n, unsigned iterations, unsigned long long frames,
             unsigned long long *__restrict__ counters, const uint32_t *__restrict__ list = nullptr,
             const uint32_t *__restrict__ list_count = nullptr) {
  // compact batch (awgn_precheck_kernel): frame k of hard / iters / status is frame list[k] of `sent`, and only the
  // device knows how many there are
  if (list_count) frames = *list_count;
  __shared__ unsigned int acc[CC_MC_NCOUNTERS];
  if (threadIdx.x < CC_MC_NCOUNTERS) acc[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, sl = lane & 15;
  const unsigned long long group = (static_cast<unsigned long long>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const unsigned long long ngroups = static_cast<unsigned long long>(gridDim.x) * 16;
  unsigned c_frames = 0, c_bit = 0, c_word = 0, c_fail = 0, c_und = 0, c_iter = 0;  // of the frames this group saw
  const unsigned long long trips = (frames + ngroups - 1) / ngroups;  // same trip count in every lane (DPP below)
  for (unsigned long long tr = 0; tr < trips; ++tr) {
    const unsigned long long f = group + tr * ngroups;
    const bool live = f < frames;
    unsigned cnt = 0;
    if (live) {
      const uint8_t *h = hard + f * n, *s = sent ? sent + (list ? static_cast<unsigned long long>(list[f]) : f) * n : nullptr;
      if (n >= 16) {
        for (int o = 16 * sl; o < n; o += 256) {
          const int o2 = o + 16 > n ? n - 16 : o, skip = o - o2;  // bytes [o2, o) belong to the neighbour
          uint32_t x[4];
          __builtin_memcpy(x, h + o2, 16);
          if (s) {
            uint32_t y[4];
            __builtin_memcpy(y, s + o2, 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] ^= y[j];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int sk = skip - 4 * j;
            const uint32_t m = sk >= 4 ? 0u : sk <= 0 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * sk);
            cnt += nonzero_bytes(x[j] & m);
          }
        }
      } else if (sl < n) {
        cnt = h[sl] != (s ? s[sl] : 0);
      }
    }
    cnt = dpp_add<0xB1>(cnt);   // quad_perm [1,0,3,2]
    cnt = dpp_add<0x4E>(cnt);   // quad_perm [2,3,0,1]
    cnt = dpp_add<0x141>(cnt);  // row_half_mirror
    cnt = dpp_add<0x140>(cnt);  // row_mirror: every lane of the row holds the frame's sum
    if (live && sl == 0) {
      const bool failed = status[f] != CC_FRAME_OK;
      c_frames += 1;
      c_bit += cnt;
      c_word += (failed || cnt) ? 1u : 0u;  // simulation.c++:128-135
      c_fail += failed ? 1u : 0u;
      c_und += (!failed && cnt) ? 1u : 0u;
      if (iters) {
        const unsigned it = iters[f];
        c_iter += failed ? iterations : it + 1;  // iterations executed
        if (!failed && it <= 55) atomicAdd(&acc[CC_MC_ITER_HIST + it], 1u);
      }
    }
  }
  if (sl == 0 && c_frames) {
    atomicAdd(&acc[CC_MC_FRAMES], c_frames);
    if (c_bit) atomicAdd(&acc[CC_MC_BIT_ERRORS], c_bit);
    if (c_word) atomicAdd(&acc[CC_MC_WORD_ERRORS], c_word);
    if (c_fail) atomicAdd(&acc[CC_MC_FAILURES], c_fail);
    if (c_und) atomicAdd(&acc[CC_MC_UNDETECTED], c_und);
    if (c_iter) atomicAdd(&acc[CC_MC_ITER_SUM], c_iter);
  }
  __syncthreads();
  if (threadIdx.x < CC_MC_NCOUNTERS && acc[threadIdx.x])
    atomicAdd(&counters[threadIdx.x], static_cast<unsigned long long>(acc[threadIdx.x]));
}

int grid_for(const cc_code *code, unsigned long long items_per_thread_total) {
  const unsigned long long want = (items_per_thread_total + 255) / 256;
  const unsigned long long max_grid = static_cast<unsigned long long>(code->num_cus) * 8;
  return static_cast<int>(want < max_grid ? (want ? want : 1) : max_grid);
}

}  // namespace

struct McWorkspace {
  std::mutex lock;
  size_t chunk = 0;
  float *llr = nullptr;
  uint8_t *sent = nullptr, *msg = nullptr, *hard = nullptr;
  uint16_t *iters = nullptr;
  int32_t *status = nullptr, *nerr = nullptr;
  uint32_t *list = nullptr;  // [chunk + 64]: 64 control words (MinSumParams::ctl in the first four), then the frame list
  // recorded behind the last work enqueued on the buffers: the lock only covers the ENQUEUE, so a later call on
  // another stream first waits (on the device) for this event before it overwrites them
  hipEvent_t done = nullptr;
  int fence_in(hipStream_t stream) {
    if (done) CC_HIP_TRY(hipStreamWaitEvent(stream, done, 0));
    return CC_OK;
  }
  int fence_out(hipStream_t stream) {
    if (!done) CC_HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    CC_HIP_TRY(hipEventRecord(done, stream));
    return CC_OK;
  }
  ~McWorkspace() {
    if (done) (void)hipEventDestroy(done);
    for (void *p : {static_cast<void *>(llr), static_cast<void *>(sent), static_cast<void *>(msg),
                    static_cast<void *>(hard), static_cast<void *>(iters), static_cast<void *>(status),
                    static_cast<void *>(nerr), static_cast<void *>(list)})
      if (p) (void)hipFree(p);
  }
};

void mc_workspace_free(McWorkspace *w) { delete w; }

static int ensure_workspace(cc_code *code, size_t chunk) {
  if (!code->mc) code->mc = new McWorkspace();
  McWorkspace &w = *code->mc;
  if (w.chunk >= chunk) return CC_OK;
  const size_t n = code->tab.n, l = code->tab.l;
  for (void **p : {reinterpret_cast<void **>(&w.llr), reinterpret_cast<void **>(&w.sent),
                   reinterpret_cast<void **>(&w.msg), reinterpret_cast<void **>(&w.hard),
                   reinterpret_cast<void **>(&w.iters), reinterpret_cast<void **>(&w.status),
                   reinterpret_cast<void **>(&w.nerr), reinterpret_cast<void **>(&w.list)})
    if (*p) {
      (void)hipFree(*p);
      *p = nullptr;
    }
  w.chunk = 0;
  CC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&w.llr), chunk * n * sizeof(float)));
  CC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&w.sent), chunk * n));
  CC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&w.msg), chunk * l));
  CC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&w.hard), chunk * n));
  CC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&w.iters), chunk * sizeof(uint16_t)));
  CC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&w.status), chunk * sizeof(int32_t)));
  CC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&w.nerr), chunk * sizeof(int32_t)));
  CC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&w.list), (chunk + 64) * sizeof(uint32_t)));
  w.chunk = chunk;
  return CC_OK;
}

// the transmitted words of frames [first, first + frames): random messages through the device encoder
static int launch_sent_words(const cc_code *code, uint64_t seed, uint64_t first_frame, size_t frames, uint8_t *d_sent,
                             uint8_t *d_msg_scratch, hipStream_t stream) {
  if (!d_sent || !d_msg_scratch) return CC_ERR_INVALID_ARGUMENT;
  const int l = static_cast<int>(code->tab.l);
  const uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
  int bits_log2 = 0;
  while ((16 << bits_log2) < l) ++bits_log2;
  const unsigned long long items = static_cast<unsigned long long>(frames) << bits_log2;
  hipLaunchKernelGGL(random_bits_kernel, dim3(grid_for(code, items)), dim3(256), 0, stream, d_msg_scratch, l, bits_log2,
                     static_cast<unsigned long long>(first_frame), static_cast<unsigned long long>(frames), k0, k1);
  return launch_encode_bits(code, d_msg_scratch, d_sent, frames, stream);
}

// The pre-check route (awgn_precheck_kernel) serves the diagonal min-sum kernels of the n = 129..256 codes and pays
// once a fair share of the frames is clean: P(no channel error in n bits) = (1 - Q(1 / sigma))^n >= 1/4 -- from
// ~5.7 dB on for BCH(255,231) (6 dB: 40 %, 8 dB: 91 %); below that the plain route is used, so 4 dB costs nothing.
static bool mc_precheck_pays(const cc_code *code, double ebno_db) {
  if (!code->soft || code->d_colbits == nullptr || !minsum_diag_supported(code) || code->force_generic) return false;
  if (!minsum_shortcuts_enabled()) return false;
  const unsigned n = code->tab.n;
  if (n <= 128 || n > 256) return false;
  const int alg = code->desc.algorithm;
  const bool scaled = alg == CC_ALG_NMS || alg == CC_ALG_2DNMS;
  if (scaled && !(code->desc.alpha > 0.0)) return false;  // the sign argument needs h(m) >= 0
  const double sigma = cc_sigma(code, ebno_db);
  const double pbit = 0.5 * std::erfc(1.0 / (sigma * std::sqrt(2.0)));
  return std::pow(1.0 - pbit, static_cast<double>(n)) >= 0.25;
}

// writes y (and the transmitted words when d_sent != nullptr) for frames [first, first + frames)
int launch_awgn(const cc_code *code, double ebno_db, uint64_t seed, uint64_t first_frame, size_t frames,
                int random_codewords, float *d_llr, uint8_t *d_sent, uint8_t *d_msg_scratch, hipStream_t stream,
                unsigned long long *d_counters = nullptr, bool hard_bytes = false) {
  if (frames == 0) return CC_OK;
  const int n = static_cast<int>(code->tab.n);
  const float sigma = static_cast<float>(cc_sigma(code, ebno_db));  // normal_distribution<float>(1.0, float(sigma))
  const uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
  const uint8_t *sent = nullptr;
  if (random_codewords) {
    const int rc = launch_sent_words(code, seed, first_frame, frames, d_sent, d_msg_scratch, stream);
    if (rc != CC_OK) return rc;
    sent = d_sent;
  } else if (d_sent) {
    CC_HIP_TRY(hipMemsetAsync(d_sent, 0, frames * static_cast<size_t>(n), stream));
  }
  int group_log2 = 0;
  while ((4 << group_log2) < n) ++group_log2;
  const unsigned long long items = static_cast<unsigned long long>(frames) << group_log2;
  if (hard_bytes)
    hipLaunchKernelGGL(awgn_kernel<true>, dim3(grid_for(code, items)), dim3(256), 0, stream, d_llr, sent, n, group_log2,
                       static_cast<unsigned long long>(first_frame), static_cast<unsigned long long>(frames), sigma, k0,
                       k1, d_counters);
  else
    hipLaunchKernelGGL(awgn_kernel<false>, dim3(grid_for(code, items)), dim3(256), 0, stream, d_llr, sent, n, group_log2,
                       static_cast<unsigned long long>(first_frame), static_cast<unsigned long long>(frames), sigma, k0,
                       k1, d_counters);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "awgn kernel launch");
  return CC_OK;
}

int mc_run(cc_code *code, double ebno_db, uint64_t seed, uint64_t first_frame, size_t frames, int random_codewords,
           uint64_t *d_counters, hipStream_t stream) {
  if (frames == 0) return CC_OK;
  const size_t chunk_max = size_t(1) << 20;  // ~1.6 GB of workspace for n = 255: long launches, short tails
  const size_t chunk = frames < chunk_max ? frames : chunk_max;
  if (!code->mc) code->mc = new McWorkspace();
  std::lock_guard<std::mutex> guard(code->mc->lock);
  int rc = ensure_workspace(code, chunk);
  if (rc != CC_OK) return rc;
  McWorkspace &w = *code->mc;
  rc = w.fence_in(stream);
  if (rc != CC_OK) return rc;
  const int n = static_cast<int>(code->tab.n);
  const bool precheck = mc_precheck_pays(code, ebno_db);
  for (size_t done = 0; done < frames; done += chunk) {
    const size_t m = frames - done < chunk ? frames - done : chunk;
    // all-zero transmission (simulation.c++:113-125): no word to keep, nothing to clear or to read back
    uint8_t *sent = random_codewords ? w.sent : nullptr;
    if (precheck) {
      // transmitted words first (random codewords), then channel + pre-check, the decoder on what is left, the counts
      if (random_codewords) {
        rc = launch_sent_words(code, seed, first_frame + done, m, w.sent, w.msg, stream);
        if (rc != CC_OK) return rc;
      }
      uint32_t *ctl = w.list, *list = w.list + 64;
      CC_HIP_TRY(hipMemsetAsync(ctl, 0, 64 * sizeof(uint32_t), stream));
      const float sigma = static_cast<float>(cc_sigma(code, ebno_db));
      const unsigned long long wg = (m + 31) / 32, cap = static_cast<unsigned long long>(code->num_cus) * 8;
      hipLaunchKernelGGL(awgn_precheck_kernel, dim3(static_cast<int>(wg < cap ? wg : cap)), dim3(256), 0, stream, w.llr, list,
                         ctl, sent, code->d_colbits, n, code->desc.stop_rule,
                         static_cast<unsigned long long>(first_frame + done), static_cast<unsigned long long>(m), sigma,
                         static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32),
                         reinterpret_cast<unsigned long long *>(d_counters));
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return hip_fail(e, "awgn pre-check kernel launch");
      rc = launch_minsum_diag_compact(code, ctl, static_cast<unsigned>(m), w.llr, w.hard, w.iters, w.status, stream);
      if (rc != CC_OK) return rc;
      const unsigned long long blocks = (m + 15) / 16, max_grid = static_cast<unsigned long long>(code->num_cus) * 8;
      hipLaunchKernelGGL(count_kernel, dim3(static_cast<int>(blocks < max_grid ? blocks : max_grid)), dim3(256), 0, stream,
                         w.hard, sent, w.iters, w.status, n, code->desc.iterations, static_cast<unsigned long long>(m),
                         reinterpret_cast<unsigned long long *>(d_counters), list, ctl + 1);
      e = hipGetLastError();
      if (e != hipSuccess) return hip_fail(e, "count kernel launch");
      continue;
    }
    // (a hard-decision decoder takes bits from the channel: the hard decisions as bytes in the same buffer)
    rc = launch_awgn(code, ebno_db, seed, first_frame + done, m, random_codewords, w.llr, sent, w.msg, stream,
                     reinterpret_cast<unsigned long long *>(d_counters), !code->soft);
    if (rc != CC_OK) return rc;
    if (code->soft)
      rc = launch_minsum(code, w.llr, nullptr, nullptr, w.hard, nullptr, w.iters, w.status, m, stream);
    else
      rc = launch_algebraic(code, false, w.llr, nullptr, nullptr, w.hard, w.nerr, w.status, m, stream);
    if (rc != CC_OK) return rc;
    const unsigned long long blocks = (m + 15) / 16;
    const unsigned long long max_grid = static_cast<unsigned long long>(code->num_cus) * 8;
    hipLaunchKernelGGL(count_kernel, dim3(static_cast<int>(blocks < max_grid ? blocks : max_grid)), dim3(256), 0, stream,
                       w.hard, sent, code->soft ? w.iters : nullptr, w.status, n, code->desc.iterations,
                       static_cast<unsigned long long>(m), reinterpret_cast<unsigned long long *>(d_counters));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "count kernel launch");
  }
  return w.fence_out(stream);
}


// cc_awgn_llr_dev: channel only, chunked so that the message scratch stays bounded
int mc_awgn(cc_code *code, double ebno_db, uint64_t seed, uint64_t first_frame, size_t frames, int random_codewords,
            float *d_llr, uint8_t *d_sent, hipStream_t stream) {
  if (frames == 0) return CC_OK;
  if (!random_codewords) return launch_awgn(code, ebno_db, seed, first_frame, frames, 0, d_llr, d_sent, nullptr, stream);
  const size_t chunk_max = size_t(1) << 20;  // ~1.6 GB of workspace for n = 255: long launches, short tails
  const size_t chunk = frames < chunk_max ? frames : chunk_max;
  if (!code->mc) code->mc = new McWorkspace();
  std::lock_guard<std::mutex> guard(code->mc->lock);
  int rc = ensure_workspace(code, chunk);
  if (rc != CC_OK) return rc;
  McWorkspace &w = *code->mc;
  rc = w.fence_in(stream);
  if (rc != CC_OK) return rc;
  const size_t n = code->tab.n;
  for (size_t done = 0; done < frames; done += chunk) {
    const size_t m = frames - done < chunk ? frames - done : chunk;
    uint8_t *sent = d_sent ? d_sent + done * n : w.sent;
    rc = launch_awgn(code, ebno_db, seed, first_frame + done, m, 1, d_llr + done * n, sent, w.msg, stream);
    if (rc != CC_OK) return rc;
  }
  return w.fence_out(stream);
}

}  // namespace ccamd
