// minsum_diag.hip -- host side of the diagonal-parallel min-sum kernel (geometry table, dealing of the diagonals
// to lanes, dispatch) and the instantiations for n = 255; the kernel itself is minsum_diag_impl.hpp, the
// instantiations for n = 127 and n <= 63 are compiled in minsum_diag_127.hip / minsum_diag_small.hip (three
// translation units build in parallel).
#include "minsum_diag_impl.hpp"

#include <algorithm>
#include <cstdlib>

namespace ccamd {

// Geometries with a diagonal instantiation: minsum_diag_geos.inc, one object file each.
// (the variants of a geometry are spread over PP objects, part I defines launch_minsum_diag_NAME_pI: diag_variant_part)
using DiagLaunch = int (*)(const cc_code *, const MinSumParams &, const float *, const uint16_t *, const uint32_t *,
                           uint8_t *, float *, uint16_t *, int32_t *, size_t, hipStream_t);
#define CC_PART_DECL(NAME, I)                                                                                        \
  int launch_minsum_diag_##NAME##_p##I(const cc_code *, const MinSumParams &, const float *, const uint16_t *,       \
                                       const uint32_t *, uint8_t *, float *, uint16_t *, int32_t *, size_t, hipStream_t);
#define CC_PARTS_1(M, NAME) M(NAME, 0)
#define CC_PARTS_2(M, NAME) CC_PARTS_1(M, NAME) M(NAME, 1)
#define CC_PARTS_3(M, NAME) CC_PARTS_2(M, NAME) M(NAME, 2)
#define CC_PARTS_6(M, NAME) CC_PARTS_3(M, NAME) M(NAME, 3) M(NAME, 4) M(NAME, 5)
#define GEO(NAME, N, KK, W, DD, LL, CC, OO, SC, PA, LK, PP) CC_PARTS_##PP(CC_PART_DECL, NAME)
#include "minsum_diag_geos.inc"
#undef GEO
namespace {
struct DiagEntry {
  DiagGeometry geo;
  int parts;
  DiagLaunch launch[6];
};
#define CC_PART_REF(NAME, I) &launch_minsum_diag_##NAME##_p##I,
const DiagEntry kDiagGeometries[] = {
#define GEO(NAME, N, KK, W, DD, LL, CC, OO, SC, PA, LK, PP)                                                          \
  {{N, KK, W, DD, LL, CC, SC, LK, {1, 1, 1, 1}}, PP, {CC_PARTS_##PP(CC_PART_REF, NAME)}},
#include "minsum_diag_geos.inc"
#undef GEO
};
// the object of the geometry that carries the variant of `p`
DiagLaunch launcher(const DiagEntry *e, const MinSumParams &p) { return e->launch[diag_variant_part(p.variant, e->parts)]; }
const DiagEntry *diag_entry(const CodeTables &t) {
  for (const DiagEntry &e : kDiagGeometries)
    if (t.n == e.geo.n && t.k == e.geo.k && t.row0_support.size() == e.geo.w) return &e;
  return nullptr;
}
}  // namespace
const DiagGeometry *diag_geometry(const CodeTables &t) {
  const DiagEntry *e = diag_entry(t);
  return e ? &e->geo : nullptr;
}

// Paired deal: np slot pairs (2p, 2p+1) whose diagonals differ by gap[p] in EVERY lane, so that the two columns a
// lane touches in such a pair sit a compile-time constant apart and one two-address LDS instruction
// (ds_read2_b64 / ds_read2_b32 / ds_write2_b32) serves both edges.  Needs W disjoint pairs (s, s + gap[p]) of the
// support per p, all np W pairs disjoint; the rest of the support fills the single slots.  Randomised greedy
// with a fixed seed (deterministic), keeping the deal with the fewest LDS passes (lanes of a frame whose
// diagonals agree modulo 16 share a bank).  Empty result: no such deal exists for these gaps.
static std::vector<uint16_t> build_paired_table(const CodeTables &t, int D, int W, int np, const int *gap) {
  const std::vector<unsigned> &sup = t.row0_support;
  std::vector<char> in_sup(t.n + 512, 0);
  for (unsigned s : sup) in_sup[s] = 1;
  const int singles_per_lane = D - 2 * np;
  uint64_t rng = 0x9E3779B97F4A7C15ull;
  auto next = [&]() {
    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
    return static_cast<uint32_t>(rng >> 33);
  };
  auto passes = [&](const std::vector<unsigned> &v) {
    int cnt[16] = {0}, mx = 0;
    for (unsigned s : v) mx = std::max(mx, ++cnt[s % 16]);
    return mx;
  };
  int best_cost = 1 << 30;
  std::vector<std::vector<unsigned>> best_low;
  std::vector<unsigned> best_single;
  for (int attempt = 0; attempt < 4000 && best_cost > np + singles_per_lane; ++attempt) {
    std::vector<char> used(t.n + 512, 0);
    std::vector<std::vector<unsigned>> low(np);
    bool ok = true;
    for (int p = 0; p < np && ok; ++p) {
      std::vector<unsigned> c;
      for (unsigned s : sup)
        if (in_sup[s + gap[p]]) c.push_back(s);
      for (size_t i = c.size(); i > 1; --i) std::swap(c[i - 1], c[next() % i]);
      bool res[16] = {false};
      for (int pass = 0; pass < 2 && low[p].size() < static_cast<size_t>(W); ++pass)  // distinct residues first
        for (unsigned s : c) {
          if (used[s] || used[s + gap[p]] || (pass == 0 && res[s % 16])) continue;
          used[s] = used[s + gap[p]] = 1;
          res[s % 16] = true;
          low[p].push_back(s);
          if (low[p].size() == static_cast<size_t>(W)) break;
        }
      ok = low[p].size() == static_cast<size_t>(W);
    }
    if (!ok) continue;
    std::vector<unsigned> single;
    for (unsigned s : sup)
      if (!used[s]) single.push_back(s);
    int cost = 0;
    for (int p = 0; p < np; ++p) cost += passes(low[p]);
    // singles: dealt to their slots below by residue class, cost = ceil(max class size / slots) per slot at best
    {
      int cnt[16] = {0}, mx = 0;
      for (unsigned s : single) mx = std::max(mx, ++cnt[s % 16]);
      cost += std::max(singles_per_lane, mx);
    }
    if (cost < best_cost) {
      best_cost = cost;
      best_low = low;
      best_single = single;
    }
  }
  if (best_cost == 1 << 30) return {};
  std::vector<uint16_t> out(static_cast<size_t>(D) * W, 0xFFFFu);
  for (int p = 0; p < np; ++p) {
    std::sort(best_low[p].begin(), best_low[p].end());
    for (int l = 0; l < W; ++l) {
      out[(2 * p) * W + l] = static_cast<uint16_t>(best_low[p][l]);
      out[(2 * p + 1) * W + l] = static_cast<uint16_t>(best_low[p][l] + gap[p]);
    }
  }
  // single slots: round-robin over the residue classes so that equal residues land in different slots
  std::vector<std::vector<unsigned>> cls(16);
  for (unsigned s : best_single) cls[s % 16].push_back(s);
  std::vector<std::vector<unsigned>> slot(singles_per_lane);
  int turn = 0;
  for (int r = 0; r < 16; ++r)
    for (unsigned s : cls[r]) {
      int tries = 0;
      while (slot[turn % singles_per_lane].size() >= static_cast<size_t>(W) && tries++ < singles_per_lane) ++turn;
      slot[turn++ % singles_per_lane].push_back(s);
    }
  for (int g = 0; g < singles_per_lane; ++g)
    for (size_t l = 0; l < slot[g].size(); ++l) out[(2 * np + g) * W + l] = static_cast<uint16_t>(slot[g][l]);
  return out;
}

std::vector<uint16_t> build_diag_table(const CodeTables &t, int D, int W, int np, const int *gap) {
#ifdef CC_AMD_EXPERIMENTS
  // timing experiment (results are WRONG): a bank-conflict-free pseudo deal, lane + 16 slot (links: tail, tail + 1)
  if (const char *e = std::getenv("CC_AMD_EXP_FAKE_DEAL"); e && e[0] == '1') {
    std::vector<uint16_t> out(static_cast<size_t>(D) * W);
    for (int d = 0; d < D; ++d)
      for (int l = 0; l < W; ++l)
        out[d * W + l] = static_cast<uint16_t>(d < 2 * np ? l + W * (d & ~1) + (d & 1) : l + W * d);
    return out;
  }
#endif
  if (np > 0) return build_paired_table(t, D, W, np, gap);
  std::vector<std::vector<unsigned>> cls(W);
  for (unsigned s : t.row0_support) cls[s % W].push_back(s);
  std::vector<std::vector<unsigned>> grp(D);
  std::vector<unsigned> left;
  for (int r = 0; r < W; ++r)
    for (size_t e = 0; e < cls[r].size(); ++e) {
      if (e < static_cast<size_t>(D))
        grp[e].push_back(cls[r][e]);
      else
        left.push_back(cls[r][e]);
    }
  for (int g = 0; g < D; ++g) {  // fill holes, preferring residues not yet doubled in the group
    while (grp[g].size() < static_cast<size_t>(W) && !left.empty()) {
      size_t best = 0;
      int best_mult = 1 << 30;
      for (size_t c = 0; c < left.size(); ++c) {
        int mult = 1;
        for (unsigned s : grp[g]) mult += (s % W == left[c] % W);
        if (mult < best_mult) {
          best_mult = mult;
          best = c;
        }
      }
      grp[g].push_back(left[best]);
      left.erase(left.begin() + static_cast<long>(best));
    }
  }
  // w < W * D: only the last slot may stay short -- top the others up from it, pad it with the marker 0xFFFF
  for (int g = 0; g + 1 < D; ++g)
    while (grp[g].size() < static_cast<size_t>(W) && !grp[D - 1].empty()) {
      grp[g].push_back(grp[D - 1].back());
      grp[D - 1].pop_back();
    }
  const size_t real_last = grp[D - 1].size();
  // local search: swap elements between groups while the total bank multiplicity (sum over slots of the
  // largest number of lanes on one residue class -- the LDS passes a slot instruction needs) decreases
  auto group_cost = [&](const std::vector<unsigned> &g) {
    int cnt[16] = {0}, mx = 0, sq = 0;
    for (unsigned s : g) ++cnt[s % 16];
    for (int c : cnt) {
      mx = c > mx ? c : mx;
      sq += c * c;
    }
    return mx * 1000 + sq;  // primary: max multiplicity, secondary: fewer doubled banks
  };
  bool improved = true;
  while (improved) {
    improved = false;
    for (int a = 0; a < D && !improved; ++a)
      for (int b = a + 1; b < D && !improved; ++b)
        for (size_t x = 0; x < grp[a].size() && !improved; ++x)
          for (size_t y = 0; y < grp[b].size() && !improved; ++y) {
            const int before = group_cost(grp[a]) + group_cost(grp[b]);
            std::swap(grp[a][x], grp[b][y]);
            if (group_cost(grp[a]) + group_cost(grp[b]) < before)
              improved = true;
            else
              std::swap(grp[a][x], grp[b][y]);
          }
  }
  std::vector<uint16_t> out(static_cast<size_t>(D) * W, 0xFFFFu);
  for (int g = 0; g < D; ++g)
    for (size_t l = 0; l < grp[g].size(); ++l) out[g * W + l] = static_cast<uint16_t>(grp[g][l]);
  (void)real_last;
  return out;
}

bool minsum_diag_supported(const cc_code *code) {
  if (code->d_diag == nullptr || code->desc.iterations == 0) return false;
  const int alg = code->desc.algorithm;
  const DiagGeometry *geo = diag_geometry(code->tab);
  if (!geo) return false;
  if (alg == CC_ALG_SCMS1) return true;        // two bits per edge: every geometry
  if (alg == CC_ALG_SCMS2) return geo->scms;  // (every geometry of minsum_diag_geos.inc since the q-only form, E38)
  if (alg != CC_ALG_MS && alg != CC_ALG_NMS && alg != CC_ALG_OMS && alg != CC_ALG_2DNMS) return false;
  if (alg == CC_ALG_OMS && !(code->desc.beta >= 0.0)) return false;
  const float a = static_cast<float>(code->desc.alpha);
  if ((alg == CC_ALG_NMS || alg == CC_ALG_2DNMS) && !(a == a && a - a == 0.0f)) return false;
  return diag_geometry(code->tab) != nullptr;
}


std::string minsum_diag_name(const cc_code *code) {
  const DiagGeometry *g = diag_geometry(code->tab);
  if (!g) return "minsum_diag_kernel";
  return "minsum_diag_kernel<K=" + std::to_string(g->k) + ",D=" + std::to_string(g->D) + ",LPF=" +
         std::to_string(g->LPF) + ",row-pipelined>";
}

size_t minsum_diag_lds_bytes(const DiagGeometry &g) {
  const bool partial = g.w != static_cast<unsigned>(g.LPF * g.D);
  const size_t fpw = 64 / g.LPF, rc = static_cast<size_t>(g.LPF) * g.CPL + (partial ? 48 : (g.LPF == 8 && g.CPL == 8 ? 8 : 16));
  return 4 * (fpw * rc * 8 + (fpw / 2) * rc * 8) + 256 * 4 +  // + column masks
         4 * static_cast<size_t>(g.CPL) * 256;                 // + staging area of the next frames
}


// ---------------- two-pass decoding ----------------
// At high SNR almost every frame stops after its first iteration, which the message-free kernel runs at about half the
// cost of the general one (no messages, one column-sum array) -- but only the device knows the operating point.  So:
//   sample      message-free first pass over 4096 frames (four runs spread over the batch), counting the frames that did not stop;
//               two passes if fewer than 1 in 8 of them did not
//   first pass  (if two passes) message-free kernel over all frames: a frame that stops is written, the others are
//               appended to a list
//   second pass (if two passes) their channel values gathered into a compact batch, the general kernel over it from
//               the first iteration on (same arithmetic, same results), results scattered back
//   otherwise   the general kernel over all frames -- also if the list overflowed (more than B / 4 frames)
// Everything is enqueued unconditionally; the helper kernels of the branch not taken return at once
// (MinSumParams::gate), and the general kernel is launched once and picks its batch on the device (MinSumParams::dual).
namespace {

__global__ void __launch_bounds__(256)
twopass_gather_kernel(const uint32_t *__restrict__ ctl, unsigned sample, unsigned cap, const uint32_t *__restrict__ list,
                      const float *__restrict__ llr, float *__restrict__ llr2, int n) {
  if (!two_pass_in_effect(ctl, sample, cap)) return;
  const unsigned long long cnt = ctl[1];
  const int lane = threadIdx.x & 63;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long k = static_cast<unsigned long long>(blockIdx.x) * 4 + (threadIdx.x >> 6); k < cnt; k += nwaves) {
    const float *src = llr + static_cast<unsigned long long>(list[k]) * n;
    float *dst = llr2 + k * n;
    for (int j = lane; j < n; j += 64) dst[j] = src[j];
  }
}
__global__ void __launch_bounds__(256)
twopass_scatter_kernel(const uint32_t *__restrict__ ctl, unsigned sample, unsigned cap, const uint32_t *__restrict__ list,
                       const uint8_t *__restrict__ hard2, const uint16_t *__restrict__ iters2,
                       const int32_t *__restrict__ status2, uint8_t *__restrict__ hard, uint16_t *__restrict__ iters,
                       int32_t *__restrict__ status, int n) {
  if (!two_pass_in_effect(ctl, sample, cap)) return;
  const unsigned long long cnt = ctl[1];
  const int lane = threadIdx.x & 63;
  const unsigned long long nwaves = static_cast<unsigned long long>(gridDim.x) * 4;
  for (unsigned long long k = static_cast<unsigned long long>(blockIdx.x) * 4 + (threadIdx.x >> 6); k < cnt; k += nwaves) {
    const unsigned long long f = list[k];
    for (int j = lane; j < n; j += 64) hard[f * n + j] = hard2[k * n + j];
    if (lane == 0) {
      if (iters) iters[f] = iters2[k];
      if (status) status[f] = status2[k];
    }
  }
}

bool two_pass_enabled() { return minsum_shortcuts_enabled(); }

int launch_two_pass(const cc_code *code, const DiagEntry *e, const MinSumParams &p, const float *d_llr, uint8_t *d_hard,
                    uint16_t *d_iters, int32_t *d_status, size_t B, hipStream_t stream) {
  const size_t n = code->tab.n, cap = B / 4, sample = 4096;
  auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  const size_t o_list = 256, o_llr = o_list + up(cap * 4), o_hard = o_llr + up(cap * n * 4), o_it = o_hard + up(cap * n),
               o_st = o_it + up(cap * 2), total = o_st + up(cap * 4);
  uint8_t *ws = nullptr;
  CC_HIP_TRY(workspace_alloc(code, reinterpret_cast<void **>(&ws), total, stream));
  uint32_t *ctl = reinterpret_cast<uint32_t *>(ws), *list = reinterpret_cast<uint32_t *>(ws + o_list);
  float *llr2 = reinterpret_cast<float *>(ws + o_llr);
  uint8_t *hard2 = ws + o_hard;
  uint16_t *it2 = reinterpret_cast<uint16_t *>(ws + o_it);
  int32_t *st2 = reinterpret_cast<int32_t *>(ws + o_st);
  int rc = CC_OK;
  hipError_t he = hipMemsetAsync(ctl, 0, 256, stream);
  if (he != hipSuccess) rc = hip_fail(he, "two-pass control words");
  MinSumParams q = p;
  q.ctl = ctl;
  q.list = list;
  q.list_cap = static_cast<unsigned>(cap);
  // two passes iff fewer than 1 in `den` frames of the sample did not stop (the device tests ctl[2] * 8 < q.sample)
  constexpr unsigned den = 8;  // threshold sweep: profiles/r02_experiments.md, E13
  q.sample = static_cast<unsigned>(sample * 8 / den);
  // the sample: four runs of 1024 frames spread over the batch (an ordered batch -- SNR sweep, clean frames first --
  // would fool a leading sample into a first pass that overflows its list)
  q.first_pass = 2;
  const size_t runs = B >= 16 * sample ? 4 : 1, per = sample / runs, hop = (B / runs) & ~static_cast<size_t>(63);
  for (size_t r = 0; r < runs && rc == CC_OK; ++r) {
    const size_t at = r * hop;
    rc = launcher(e, q)(code, q, d_llr + at * n, nullptr, nullptr, d_hard + at * n, nullptr, d_iters ? d_iters + at : nullptr,
                        d_status ? d_status + at : nullptr, per, stream);
  }
  if (rc == CC_OK) {
    q.first_pass = 1;  // first pass over everything (returns at once unless the sample says so)
    q.gate = 1;
    rc = launcher(e, q)(code, q, d_llr, nullptr, nullptr, d_hard, nullptr, d_iters, d_status, B, stream);
  }
  if (rc == CC_OK) {
    const int grid = code->num_cus * 4;
    const unsigned su = q.sample, cu = q.list_cap;  // the same test in every kernel
    hipLaunchKernelGGL(twopass_gather_kernel, dim3(grid), dim3(256), 0, stream, ctl, su, cu, list, d_llr, llr2,
                       static_cast<int>(n));
    // the general kernel, once: over the compacted frames if the device chose two passes, over everything otherwise
    q.first_pass = 0;
    q.gate = -1;
    q.dual = 1;
    q.llr2 = llr2;
    q.hard2 = hard2;
    q.iters2 = it2;
    q.status2 = st2;
    rc = launcher(e, q)(code, q, d_llr, nullptr, nullptr, d_hard, nullptr, d_iters, d_status, B, stream);
    if (rc == CC_OK)
      hipLaunchKernelGGL(twopass_scatter_kernel, dim3(grid), dim3(256), 0, stream, ctl, su, cu, list, hard2, it2, st2,
                         d_hard, d_iters, d_status, static_cast<int>(n));
  }
  if (rc == CC_OK) {
    he = hipGetLastError();
    if (he != hipSuccess) rc = hip_fail(he, "two-pass kernels launch");
  }
  (void)hipFreeAsync(ws, stream);
  return rc;
}

}  // namespace

// CC_AMD_TWO_PASS=0: no route that depends on the operating point -- neither two-pass decoding nor the Monte-Carlo
// pre-check (mc.hip); the tests compare both against the plain route through this switch
bool minsum_shortcuts_enabled() {
  static const bool v = [] {
    const char *e = std::getenv("CC_AMD_TWO_PASS");
    return !(e && e[0] == '0');
  }();
  return v;
}

int launch_minsum_diag_compact(const cc_code *code, uint32_t *d_ctl, unsigned cap, const float *d_llr, uint8_t *d_hard,
                               uint16_t *d_iters, int32_t *d_status, hipStream_t stream) {
  const DiagEntry *e = diag_entry(code->tab);
  if (!e) return CC_ERR_UNSUPPORTED;
  MinSumParams q = minsum_params(code);
  q.ctl = d_ctl;  // [2] = [3] = 0: "two passes in effect", [1] = frames in the compact batch
  q.sample = 1;
  q.list_cap = cap;
  q.gate = -1;
  q.dual = 1;
  q.llr2 = d_llr;
  q.hard2 = d_hard;
  q.iters2 = d_iters;
  q.status2 = d_status;
  // (the caller's-batch arguments are never touched: the kernel switches to the compact buffers on the device)
  return launcher(e, q)(code, q, d_llr, nullptr, nullptr, d_hard, nullptr, d_iters, d_status, cap, stream);
}

int launch_minsum_diag(const cc_code *code, const MinSumParams &p, const float *d_llr, const uint16_t *d_er,
                       const uint32_t *d_er_off, uint8_t *d_hard, float *d_L, uint16_t *d_iters, int32_t *d_status,
                       size_t B, hipStream_t stream) {
  const DiagEntry *e = diag_entry(code->tab);
  if (!e) return CC_ERR_UNSUPPORTED;
  const bool plain = p.variant == CC_ALG_MS || p.variant == CC_ALG_NMS || p.variant == CC_ALG_OMS || p.variant == CC_ALG_2DNMS;
  // the helper launches cost ~0.06 ms per call: only where a call runs for a millisecond or more (frames x edges per
  // iteration; 1.5e9 = 2^18 frames of BCH(255,231), never BCH(63,45) at 2^20)
  const double work = static_cast<double>(B) * code->tab.n * code->tab.k;
  if (two_pass_enabled() && plain && p.stop_rule != CC_STOP_AS_SHIPPED && p.iterations >= 2 && d_er_off == nullptr &&
      d_L == nullptr && work >= 1.5e9)
    return launch_two_pass(code, e, p, d_llr, d_hard, d_iters, d_status, B, stream);
  return launcher(e, p)(code, p, d_llr, d_er, d_er_off, d_hard, d_L, d_iters, d_status, B, stream);
}

}  // namespace ccamd
