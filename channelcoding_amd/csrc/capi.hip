// capi.hip -- the extern "C" boundary of libchannelcoding_amd.so.
// Every function mirrors a member of the reference's cyclic<>/primitive_bch/rs
// API (see include/channelcoding_amd.h for the file:line map).  Host-pointer
// entry points stage through device memory and call the _dev ones; there is no
// CPU decode path in this library.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <thread>
#include <vector>

#include "cc_internal.hpp"

namespace ccamd {

static thread_local std::string g_last_error;
void set_last_error(const std::string &s) { g_last_error = s; }
int hip_fail(hipError_t e, const char *what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return CC_ERR_HIP;
}

namespace {

// binds the calling thread to the code's device for the duration of a call
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
      switched = hipSetDevice(dev) == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

}  // namespace (anonymous)
// Staging of the host-pointer entry points ("no hidden allocation per call; thread-safe per
// stream").  Owned by the handle, created on first use: two private streams and, per stream, grow-only device
// buffers.  A host-pointer call cuts the batch into chunks that alternate between the two streams -- the upload of
// chunk k + 1 overlaps the kernel and the download of chunk k -- and waits for ITS streams only
// (hipStreamSynchronize, never hipDeviceSynchronize: other streams of the caller keep running).  Calls on one
// handle are serialised by `lock`; different handles are independent.
// Copies between pageable caller memory and the page-locked staging ring are plain memcpy calls spread over a few
// worker threads (one thread moves ~10 GB/s, the DMA engine 55 GB/s; round 2 handed pageable pointers to
// hipMemcpyAsync, which then blocks the calling thread until the data has moved and with it the pipeline:
// profiles/r02_host_path.txt, 49 ms for what upload and kernel together should do in 31).  Process-wide, created on
// first use, never joined (the workers touch no HIP state and sleep on a condition variable).
class CopyPool {
 public:
  static CopyPool &get() {
    static CopyPool *pool = new CopyPool();
    return *pool;
  }
  void copy(void *dst, const void *src, size_t bytes) {
    const size_t piece = 4u << 20;
    if (bytes <= piece || workers_ == 0) {
      std::memcpy(dst, src, bytes);
      return;
    }
    const size_t parts = std::min<size_t>((bytes + piece - 1) / piece, static_cast<size_t>(workers_) + 1);
    const size_t each = ((bytes + parts - 1) / parts + 4095) & ~static_cast<size_t>(4095);
    std::atomic<int> left{0};
    std::mutex dm;
    std::condition_variable dcv;
    size_t off = each;  // the caller's own share is [0, each)
    {
      std::lock_guard<std::mutex> g(m_);
      for (; off < bytes; off += each) {
        const size_t len = std::min(each, bytes - off);
        ++left;
        jobs_.push_back(Job{static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, len, &left, &dm, &dcv});
      }
    }
    cv_.notify_all();
    std::memcpy(dst, src, std::min(each, bytes));
    std::unique_lock<std::mutex> lk(dm);
    dcv.wait(lk, [&] { return left.load() == 0; });
  }

 private:
  struct Job {
    char *dst;
    const char *src;
    size_t len;
    std::atomic<int> *left;
    std::mutex *dm;
    std::condition_variable *dcv;
  };
  CopyPool() {
    const unsigned hc = std::thread::hardware_concurrency();
    workers_ = hc >= 16 ? 7 : hc >= 8 ? 5 : hc >= 4 ? 2 : 0;  // 3 / 7 / 15 workers: 35 / 33 / 32 ms per 2^20 frames at 4 dB
    for (int i = 0; i < workers_; ++i) std::thread([this] { run(); }).detach();
  }
  void run() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !jobs_.empty(); });
        j = jobs_.back();
        jobs_.pop_back();
      }
      std::memcpy(j.dst, j.src, j.len);
      {
        std::lock_guard<std::mutex> g(*j.dm);  // the waiter cannot leave (and destroy dm / dcv) between the two lines
        if (j.left->fetch_sub(1) == 1) j.dcv->notify_one();
      }
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::vector<Job> jobs_;
  int workers_ = 0;
};

struct HostStage {
  std::mutex lock;
  hipStream_t stream[2] = {nullptr, nullptr};
  struct Buf {
    void *p = nullptr;
    size_t cap = 0;
  };
  Buf buf[2][8];
  Buf pin[2][8];  // page-locked twins of buf for pageable caller memory (same slot / index)
  struct Pending {
    void *dst;
    const void *src;
    size_t bytes;
  };
  std::vector<Pending> pending[2];  // results waiting in pin[slot][*] for their stream to finish
  int init() {
    for (hipStream_t &s : stream)
      if (!s) CC_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return CC_OK;
  }
  // A call of several chunks starts on a fresh pair of streams.  Measured (profiles/r03_host_path.txt,
  // profiles/tools/host_path_trace.py): once a stream pair has been through a call whose chunks were fed from the host
  // side with gaps (pageable caller memory), every later call on that pair runs its copies and kernels one after the
  // other -- 46 ms for 2^20 frames at 4 dB where the same call on new streams takes 25 -- for the rest of the process;
  // rocprofv3's copy trace shows the transfers of the fast case on the DMA engines next to the kernels.  Creating two
  // streams costs ~40 us, so calls of one or two chunks (below ~64 MiB) keep the pair they have.
  int fresh_streams() {
    for (hipStream_t &s : stream)
      if (s) {
        CC_HIP_TRY(hipStreamSynchronize(s));
        CC_HIP_TRY(hipStreamDestroy(s));
        s = nullptr;
      }
    return init();
  }
  template <typename T>
  int get(int slot, int idx, size_t count, T **out) {  // grow-only; contents are not preserved
    Buf &b = buf[slot][idx];
    const size_t bytes = count * sizeof(T) + 16;
    if (bytes > b.cap) {
      CC_HIP_TRY(hipStreamSynchronize(stream[slot]));
      if (b.p) (void)hipFree(b.p);
      b.p = nullptr;
      b.cap = 0;
      const size_t want = bytes + bytes / 4;  // grow by 25 % so that slowly growing batches do not reallocate each time
      CC_HIP_TRY(hipMalloc(&b.p, want));
      b.cap = want;
    }
    *out = static_cast<T *>(b.p);
    return CC_OK;
  }
  int get_pinned(int slot, int idx, size_t bytes, void **out) {
    Buf &b = pin[slot][idx];
    if (bytes > b.cap) {
      if (b.p) (void)hipHostFree(b.p);
      b.p = nullptr;
      b.cap = 0;
      const size_t want = bytes + bytes / 4;
      CC_HIP_TRY(hipHostMalloc(&b.p, want, hipHostMallocDefault));
      b.cap = want;
    }
    *out = b.p;
    return CC_OK;
  }
  // is the caller's buffer something the DMA engines reach directly (page-locked / registered / device memory)?
  static bool dma_ready(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
      (void)hipGetLastError();  // plain malloc'ed memory: "invalid value", not an error of ours
      return false;
    }
    return a.type == hipMemoryTypeHost || a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
  }
  // host -> device on the slot's stream; `direct` = dma_ready(h_src base), decided once per call
  int upload(int slot, int idx, void *d_dst, const void *h_src, size_t bytes, bool direct) {
    if (bytes == 0) return CC_OK;
    if (!direct) {
      void *ring = nullptr;
      if (int rc = get_pinned(slot, idx, bytes, &ring)) return rc;
      CopyPool::get().copy(ring, h_src, bytes);
      h_src = ring;
    }
    CC_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, stream[slot]));
    return CC_OK;
  }
  // device -> host behind the slot's kernels; a pageable destination receives its bytes in retire()
  int download(int slot, int idx, void *h_dst, const void *d_src, size_t bytes, bool direct) {
    if (bytes == 0) return CC_OK;
    if (direct) {
      CC_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, stream[slot]));
      return CC_OK;
    }
    void *ring = nullptr;
    if (int rc = get_pinned(slot, idx, bytes, &ring)) return rc;
    CC_HIP_TRY(hipMemcpyAsync(ring, d_src, bytes, hipMemcpyDeviceToHost, stream[slot]));
    pending[slot].push_back(Pending{h_dst, ring, bytes});
    return CC_OK;
  }
  // wait for everything enqueued on the slot and hand its staged results to the caller's buffers
  int retire(int slot) {
    if (!stream[slot]) return CC_OK;
    const hipError_t e = hipStreamSynchronize(stream[slot]);
    if (e == hipSuccess)
      for (const Pending &q : pending[slot]) CopyPool::get().copy(q.dst, q.src, q.bytes);
    pending[slot].clear();
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize (host staging)");
    return CC_OK;
  }
  int drain() {
    const int a = retire(0), b = retire(1);
    return a != CC_OK ? a : b;
  }
};
void host_stage_free(HostStage *st) {
  if (!st) return;
  for (int slot = 0; slot < 2; ++slot) {
    if (st->stream[slot]) (void)hipStreamSynchronize(st->stream[slot]);
    for (HostStage::Buf &b : st->buf[slot])
      if (b.p) (void)hipFree(b.p);
    for (HostStage::Buf &b : st->pin[slot])
      if (b.p) (void)hipHostFree(b.p);
    if (st->stream[slot]) (void)hipStreamDestroy(st->stream[slot]);
  }
  delete st;
}
namespace {

// the handle's staging object (created on first use) with its lock held for the duration of one host-pointer call
struct StageLock {
  HostStage *st = nullptr;
  std::unique_lock<std::mutex> held;
  int rc = CC_OK;
  explicit StageLock(const cc_code *code) {
    {
      std::lock_guard<std::mutex> g(code->lazy_lock);
      if (!code->stage) code->stage = new HostStage();
      st = code->stage;
    }
    held = std::unique_lock<std::mutex>(st->lock);
    rc = st->init();
  }
  // every way out of a host-pointer call, error paths included, leaves nothing in flight that still writes to the
  // caller's buffers or reads the staging buffers (a second wait on idle streams costs microseconds)
  ~StageLock() {
    if (st && held.owns_lock()) (void)st->drain();
  }
};
// frames per chunk: about 32 MiB of the widest per-frame stream, at least 16 frames
size_t chunk_frames(size_t bytes_per_frame, size_t B) {
  static const size_t chunk_bytes = [] {  // CC_AMD_HOST_CHUNK_BYTES: tests force many small chunks
    const char *e = std::getenv("CC_AMD_HOST_CHUNK_BYTES");
    const long long v = e ? std::atoll(e) : 0;
    return v > 0 ? static_cast<size_t>(v) : static_cast<size_t>(32u << 20);
  }();
  size_t ch = chunk_bytes / (bytes_per_frame ? bytes_per_frame : 1);
  if (ch < 16) ch = 16;
  return ch < B ? ch : B;
}
// uploads the erasure lists of frames [c0, c0 + m) and returns device pointers with which the kernels index them
// by the GLOBAL offsets: d_er is shifted back by off[c0] elements (only [off[c0], off[c0 + m]) is ever read)
int upload_erasures(HostStage &st, int slot, int idx, const uint16_t *erasures, const uint32_t *offsets, size_t c0,
                    size_t m, const uint16_t **d_er, const uint32_t **d_off) {
  *d_er = nullptr;
  *d_off = nullptr;
  if (!erasures) return CC_OK;
  const size_t e0 = offsets[c0], ne = offsets[c0 + m] - e0;
  uint16_t *er = nullptr;
  uint32_t *off = nullptr;
  if (int rc = st.get(slot, idx, ne + 1, &er)) return rc;
  if (int rc = st.get(slot, idx + 1, m + 1, &off)) return rc;
  if (ne) CC_HIP_TRY(hipMemcpyAsync(er, erasures + e0, ne * sizeof(uint16_t), hipMemcpyHostToDevice, st.stream[slot]));
  CC_HIP_TRY(hipMemcpyAsync(off, offsets + c0, (m + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st.stream[slot]));
  *d_er = er - e0;
  *d_off = off;
  return CC_OK;
}

bool is_soft(int alg) { return alg >= CC_ALG_MS && alg <= CC_ALG_2DNMS; }
bool is_hard(int alg) { return alg >= CC_ALG_PGZ && alg <= CC_ALG_EUKLID; }

const char *alg_name(int alg) {
  switch (alg) {  // Algorithm::to_string(): hard_decision.h:15-24, soft_decision.h:20-73
    case CC_ALG_PGZ: return "PGZ";
    case CC_ALG_BM: return "BM";
    case CC_ALG_EUKLID: return "EUKLID";
    case CC_ALG_MS: return "MS";
    case CC_ALG_NMS: return "NMS";
    case CC_ALG_OMS: return "OMS";
    case CC_ALG_SCMS1: return "SCMS1";
    case CC_ALG_SCMS2: return "SCMS2";
    case CC_ALG_2DNMS: return "2DNMS";
    default: return "?";
  }
}

}  // namespace
}  // namespace ccamd

using namespace ccamd;

#pragma GCC visibility push(default)
extern "C" {

const char *cc_version(void) { return "channelcoding_amd 0.1 (gfx950, ABI 1)"; }

const char *cc_status_string(int s) {
  switch (s) {
    case CC_OK: return "ok";
    case CC_ERR_INVALID_ARGUMENT: return "invalid argument";
    case CC_ERR_UNSUPPORTED: return "not supported on the device path";
    case CC_ERR_NO_DEVICE: return "no usable HIP device";
    case CC_ERR_HIP: return "HIP runtime error";
    case CC_ERR_OUT_OF_MEMORY: return "out of memory";
    case CC_ERR_LENGTH: return "sequence has the wrong length";
    case CC_ERR_NOT_IN_FIELD: return "value is not an element of the field";
    default: return "unknown status";
  }
}

const char *cc_last_error(void) { return g_last_error.c_str(); }

void cc_desc_init(cc_desc *d) {
  if (!d) return;
  std::memset(d, 0, sizeof *d);
  d->struct_size = sizeof *d;
  d->family = CC_FAMILY_BCH;
  d->mu = 1;
  d->step = 1;
  d->coding = CC_CODING_DIVISION;
  d->algorithm = CC_ALG_PGZ;  // the reference's default Sigma, bch.h:17
  d->iterations = 50;         // min_sum_tag<Iterations = 50>, soft_decision.h:20
  d->alpha = 1.0;
  d->beta = 0.0;
  d->stop_rule = CC_STOP_PARITY;
  d->device = -1;
}

static int code_create_impl(const cc_desc *desc, const uint8_t *customH, uint32_t custom_rows, uint32_t custom_cols,
                            cc_code **out) {
  if (!desc || !out || desc->struct_size != sizeof(cc_desc)) return CC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!is_soft(desc->algorithm) && !is_hard(desc->algorithm)) return CC_ERR_INVALID_ARGUMENT;
  const bool matrix_only = custom_cols != 0;  // cc_minsum_create: the free min_sum(H, y, tag), no code behind it
  if (!matrix_only && desc->family != CC_FAMILY_BCH && desc->family != CC_FAMILY_RS) return CC_ERR_INVALID_ARGUMENT;
  if (desc->stop_rule < CC_STOP_AS_SHIPPED || desc->stop_rule > CC_STOP_PARITY) return CC_ERR_INVALID_ARGUMENT;
  if (desc->coding != CC_CODING_DIVISION && desc->coding != CC_CODING_MULTIPLICATION) return CC_ERR_INVALID_ARGUMENT;
  if (!matrix_only && (desc->q < 2 || desc->q > 15)) return CC_ERR_INVALID_ARGUMENT;
  if (desc->reserved != 0) return CC_ERR_INVALID_ARGUMENT;
  const bool wide = !matrix_only && desc->q > 8;
  if (wide) {
    // min-sum on BCH(2^q - 1, .) for q = 9..11 (cyclic.h:254-267 is width-agnostic): the generic kernel over H's
    // banded rows, up to 2048 columns; RS has no binary H, a caller-supplied H goes through cc_minsum_create
    if (customH || (is_soft(desc->algorithm) && (desc->family != CC_FAMILY_BCH || desc->q > 11))) {
      set_last_error("q > 8: min-sum for BCH codes of up to 2047 columns (q <= 11); custom matrices through cc_minsum_create");
      return CC_ERR_UNSUPPORTED;
    }
    if (desc->coding != CC_CODING_DIVISION) {
      set_last_error("q > 8: division_tag coding only");
      return CC_ERR_UNSUPPORTED;
    }
    if (is_hard(desc->algorithm) && desc->t > 32) {
      set_last_error("hard algorithms: one lane per syndrome, t <= 32");
      return CC_ERR_UNSUPPORTED;
    }
    if (desc->family == CC_FAMILY_RS && (desc->mu != 1 || desc->step != 1)) {
      set_last_error("RS hard decoding with mu / step != 1 is not supported");
      return CC_ERR_UNSUPPORTED;
    }
  }
  if (!matrix_only && desc->n != 0 && desc->n != (1u << desc->q) - 1) {
    set_last_error("shortened codes (N != 2^q-1) are not supported");
    return CC_ERR_UNSUPPORTED;
  }
  int dev = desc->device;
  if (dev != CC_DEVICE_NONE) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      set_last_error("no HIP device: this library has no CPU fallback");
      return CC_ERR_NO_DEVICE;
    }
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return CC_ERR_NO_DEVICE;
    if (dev < 0 || dev >= ndev) return CC_ERR_INVALID_ARGUMENT;
  }

  // every early return below releases what has been allocated on the device so far (cc_code_destroy frees the
  // device members, not only the struct)
  std::unique_ptr<cc_code, void (*)(cc_code *)> code(new (std::nothrow) cc_code(), &cc_code_destroy);
  if (!code) return CC_ERR_OUT_OF_MEMORY;
  code->desc = *desc;
  code->device = dev;
  code->matrix_only = matrix_only;
  try {
    if (matrix_only) {
      code->tab.family = CC_FAMILY_BCH;
      code->tab.n = custom_cols;
      code->tab.k = custom_rows;
      code->tab.l = custom_cols > custom_rows ? custom_cols - custom_rows : 0;
    } else if (wide) {
      code->wide = true;
      code->field16.reset(new FieldT<uint16_t>(desc->q, desc->modular_polynomial));
      code->tab16 = build_code(*code->field16, desc->family, desc->t, desc->mu, desc->step);
      const CodeTablesT<uint16_t> &w = code->tab16;  // the scalars every getter reads
      code->tab.family = w.family;
      code->tab.q = w.q;
      code->tab.t = w.t;
      code->tab.n = w.n;
      code->tab.k = w.k;
      code->tab.l = w.l;
      code->tab.dmin = w.dmin;
      code->tab.mu = w.mu;
      code->tab.step = w.step;
      // H's first row (h reversed), for min-sum and cc_get_H: 0 / 1 for BCH, anything else makes binary_h false
      code->tab.binary_h = w.binary_h;
      code->tab.row0_support = w.row0_support;
      code->tab.row0.assign(w.row0.size(), 0);
      for (size_t j = 0; j < w.row0.size(); ++j) code->tab.row0[j] = w.row0[j] ? (w.row0[j] == 1 ? 1 : 2) : 0;
    } else {
      code->field.reset(new Field(desc->q, desc->modular_polynomial));
      code->tab = build_code(*code->field, desc->family, desc->t, desc->mu, desc->step);
    }
  } catch (const std::invalid_argument &e) {
    set_last_error(e.what());
    return CC_ERR_INVALID_ARGUMENT;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return CC_ERR_INVALID_ARGUMENT;
  }
  code->soft = is_soft(desc->algorithm);
  code->ms_rows = code->tab.k;
  if (customH) {
    if (!code->soft || custom_rows == 0) {
      set_last_error("a custom parity-check matrix needs a min-sum algorithm and at least one row");
      return CC_ERR_INVALID_ARGUMENT;
    }
    code->custom_H.assign(customH, customH + static_cast<size_t>(custom_rows) * code->tab.n);
    code->ms_rows = custom_rows;
    for (uint8_t v : code->custom_H)
      if (v > 1) code->tab.binary_h = false;
  }
  {
    const char *fg = std::getenv("CC_AMD_FORCE_GENERIC");
    code->force_generic = fg && fg[0] == '1';
  }
  const CodeTables &t = code->tab;
  {
    char buf[96];
    if (matrix_only)
      std::snprintf(buf, sizeof buf, "%ux%u-%s", t.k, t.n, alg_name(desc->algorithm));
    else
      std::snprintf(buf, sizeof buf, "(%u, %u, %u)-%s", t.n, t.l, t.dmin, alg_name(desc->algorithm));
    code->name = buf;
  }
  if (code->soft && !t.binary_h) {
    set_last_error("min-sum over a non-binary parity-check matrix (RS) is not supported");
    return CC_ERR_UNSUPPORTED;
  }
  if (code->soft) {  // lanes per frame (W) and columns per lane (C) of the min-sum kernels
    MinSumGeometry &g = code->geo;
    if (t.n <= 16) {
      g.W = 16;
      g.C = 1;
    } else if (t.n <= 32) {
      g.W = 32;
      g.C = 1;
    } else if (t.n <= 64) {
      g.W = 64;
      g.C = 1;
    } else if (t.n <= 128) {
      g.W = 64;
      g.C = 2;
    } else if (t.n <= 256) {
      g.W = 64;
      g.C = 4;
    } else if (t.n <= 2048) {  // the generic kernel's C = 8 / 16 / 32 instantiations: a lane owns columns l + 64 c
      g.W = 64;
      g.C = t.n <= 512 ? 8 : t.n <= 1024 ? 16 : 32;
    } else {
      set_last_error("min-sum kernels hold one frame per wavefront: at most 2048 columns");
      return CC_ERR_UNSUPPORTED;
    }
    g.frames_per_wave = 64 / g.W;
    g.KW = static_cast<int>((code->ms_rows + 31) / 32);
  }
  if (dev == CC_DEVICE_NONE) {
    *out = code.release();
    return CC_OK;
  }
  DeviceGuard guard(dev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess) code->num_cus = prop.multiProcessorCount;
  {  // workspace pool of the handle (see cc_internal.hpp); without one the default pool serves
    hipMemPoolProps pp = {};
    pp.allocType = hipMemAllocationTypePinned;
    pp.handleTypes = hipMemHandleTypeNone;
    pp.location.type = hipMemLocationTypeDevice;
    pp.location.id = dev;
    hipMemPool_t pool = nullptr;
    if (hipMemPoolCreate(&pool, &pp) == hipSuccess) {
      // freed workspace stays in the pool for the next call (no allocation after the first call of a size);
      // CC_AMD_POOL_KEEP_MB caps what a handle keeps -- for processes that hold dozens of decoders
      uint64_t keep = ~0ull;
      if (const char *mb = std::getenv("CC_AMD_POOL_KEEP_MB")) keep = static_cast<uint64_t>(std::strtoull(mb, nullptr, 10)) << 20;
      if (hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) == hipSuccess)
        code->pool = pool;
      else
        (void)hipMemPoolDestroy(pool);
    }
    (void)hipGetLastError();
  }

  if (code->soft) {
    MinSumGeometry &g = code->geo;
    std::vector<uint32_t> cm(static_cast<size_t>(g.KW) * g.C * 64, 0u);
    for (int lane = 0; lane < 64; ++lane) {
      const int li = lane % g.W;
      for (int c = 0; c < g.C; ++c) {
        const unsigned j = static_cast<unsigned>(li + g.W * c);
        if (j >= t.n) continue;
        for (unsigned i = 0; i < code->ms_rows; ++i) {
          const bool edge = code->custom_H.empty() ? (i <= j && t.row0[j - i] != 0)  // H[i][j] = row0[j - i], cyclic.h:346-359
                                                   : (code->custom_H[static_cast<size_t>(i) * t.n + j] != 0);
          if (edge) cm[(static_cast<size_t>(i >> 5) * g.C + c) * 64 + lane] |= 1u << (i & 31);
        }
      }
    }
    CC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&code->d_colmask), cm.size() * sizeof(uint32_t)));
    CC_HIP_TRY(hipMemcpy(code->d_colmask, cm.data(), cm.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (const DiagGeometry *dgeo = code->custom_H.empty() ? diag_geometry(t) : nullptr) {
      const std::vector<uint16_t> dg = build_diag_table(t, dgeo->D, dgeo->LPF, dgeo->np, dgeo->gap);
      if (dg.empty()) {
        set_last_error("no paired deal of the row-0 support for this geometry's gaps");
        return CC_ERR_UNSUPPORTED;
      }
      std::vector<uint32_t> cb(256, 0u);
      for (unsigned j = 0; j < t.n; ++j)
        for (unsigned i = 0; i < t.k && i <= j; ++i)
          if (t.row0[j - i]) cb[j] |= 1u << i;
      CC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&code->d_diag), dg.size() * sizeof(uint16_t)));
      CC_HIP_TRY(hipMemcpy(code->d_diag, dg.data(), dg.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      CC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&code->d_colbits), cb.size() * sizeof(uint32_t)));
      CC_HIP_TRY(hipMemcpy(code->d_colbits, cb.data(), cb.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
  }
  if (matrix_only) {
    *out = code.release();
    return CC_OK;
  }
  if (code->wide) {  // tables of the 16-bit path: one allocation {exp, log, g}
    const FieldT<uint16_t> &f = *code->field16;
    const CodeTablesT<uint16_t> &w = code->tab16;
    const size_t ne = f.exp.size(), ng = w.g.size();
    std::vector<uint16_t> blob(2 * ne + ng + 8, 0);
    std::copy(f.exp.begin(), f.exp.end(), blob.begin());
    std::copy(f.log.begin(), f.log.end(), blob.begin() + ne);
    std::copy(w.g.begin(), w.g.end(), blob.begin() + 2 * ne);
    CC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&code->d_wide), blob.size() * sizeof(uint16_t)));
    CC_HIP_TRY(hipMemcpy(code->d_wide, blob.data(), blob.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    WideTables &wt = code->wide_dev;
    wt.exp = code->d_wide;
    wt.log = code->d_wide + ne;
    wt.g = code->d_wide + 2 * ne;
    for (size_t i = 0; i < w.root_powers.size() && i < 64; ++i) wt.root_log[i] = w.root_powers[i];
    wt.n = w.n;
    wt.k = w.k;
    wt.l = w.l;
    wt.t = w.t;
    wt.nroots = static_cast<uint32_t>(w.roots.size());
    wt.q = w.q;
    wt.family = w.family;
    *out = code.release();
    return CC_OK;
  }
  AlgebraicTables &a = code->h_alg;
  std::memset(&a, 0, sizeof a);
  std::memcpy(a.exp, code->field->exp.data(), code->field->exp.size());
  std::memcpy(a.log, code->field->log.data(), code->field->log.size());
  std::memcpy(a.g, t.g.data(), t.g.size());
  for (size_t i = 0; i < t.root_powers.size() && i < 64; ++i) a.roots_log[i] = static_cast<uint8_t>(t.root_powers[i]);
  a.n = static_cast<int>(t.n);
  a.k = static_cast<int>(t.k);
  a.l = static_cast<int>(t.l);
  a.t = static_cast<int>(t.t);
  a.nroots = static_cast<int>(t.roots.size());
  a.family = t.family;
  a.q = static_cast<int>(t.q);
  CC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&code->d_alg), sizeof a));
  CC_HIP_TRY(hipMemcpy(code->d_alg, &a, sizeof a, hipMemcpyHostToDevice));
  if (desc->coding == CC_CODING_DIVISION) {
    const std::vector<uint8_t> pt = build_parity_table(*code->field, t);
    CC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&code->d_parity), pt.size()));
    CC_HIP_TRY(hipMemcpy(code->d_parity, pt.data(), pt.size(), hipMemcpyHostToDevice));
  }

  *out = code.release();
  return CC_OK;
}

int cc_code_create(const cc_desc *desc, cc_code **out) { return code_create_impl(desc, nullptr, 0, 0, out); }

int cc_code_create_with_H(const cc_desc *desc, const uint8_t *H, uint32_t rows, cc_code **out) {
  if (!H) return CC_ERR_INVALID_ARGUMENT;
  return code_create_impl(desc, H, rows, 0, out);
}

int cc_minsum_create(const cc_desc *desc, const uint8_t *H, uint32_t rows, uint32_t cols, cc_code **out) {
  if (!H || rows == 0 || cols == 0) return CC_ERR_INVALID_ARGUMENT;
  if (cols > 2048) {
    set_last_error("min-sum kernels hold one frame per wavefront: at most 2048 columns");
    return CC_ERR_UNSUPPORTED;
  }
  return code_create_impl(desc, H, rows, cols, out);
}

static int not_wide(const cc_code *code) {
  if (code->wide) {
    set_last_error("GF(2^q), q > 8: symbols are 16 bits wide, use the _u16 entry points");
    return CC_ERR_UNSUPPORTED;
  }
  return CC_OK;
}

static int needs_code(const cc_code *c) {
  if (c->matrix_only) {
    set_last_error("handle was made by cc_minsum_create: it has a parity-check matrix but no code behind it");
    return CC_ERR_INVALID_ARGUMENT;
  }
  return CC_OK;
}

int cc_get_H_alt(const cc_code *c, uint8_t *H, uint32_t *rows) {
  if (!c || !H) return CC_ERR_INVALID_ARGUMENT;
  if (needs_code(c) != CC_OK) return CC_ERR_INVALID_ARGUMENT;
  if (int rc = not_wide(c)) return rc;
  const unsigned n = c->tab.n, q = c->tab.q, t = c->tab.t;
  for (unsigned r = 0; r < t; ++r)
    for (unsigned bit = 0; bit < q; ++bit)
      for (unsigned col = 0; col < n; ++col) {
        const uint8_t v = c->field->alpha_pow(col * (2 * r + 1));  // from_power: exponent mod 2^q (sic)
        H[(static_cast<size_t>(r) * q + bit) * n + col] = (v >> bit) & 1u;
      }
  if (rows) *rows = t * q;
  return CC_OK;
}

void cc_code_destroy(cc_code *code) {
  if (!code) return;
  if (code->device != CC_DEVICE_NONE) {
    DeviceGuard guard(code->device);
    if (code->d_colmask) (void)hipFree(code->d_colmask);
    if (code->d_diag) (void)hipFree(code->d_diag);
    if (code->d_colbits) (void)hipFree(code->d_colbits);
    if (code->d_parity) (void)hipFree(code->d_parity);
    if (code->mc) mc_workspace_free(code->mc);
    if (code->stage) host_stage_free(code->stage);
    if (code->d_wide) (void)hipFree(code->d_wide);
    if (code->d_alg) (void)hipFree(code->d_alg);
    if (code->pool) (void)hipMemPoolDestroy(code->pool);  // (blocks of calls still in flight go back when they complete)
  }
  delete code;
}

uint32_t cc_n(const cc_code *c) { return c ? c->tab.n : 0; }
uint32_t cc_k(const cc_code *c) { return c ? c->tab.k : 0; }
uint32_t cc_l(const cc_code *c) { return c ? c->tab.l : 0; }
uint32_t cc_t(const cc_code *c) { return c ? c->tab.t : 0; }
uint32_t cc_dmin(const cc_code *c) { return c ? c->tab.dmin : 0; }
double cc_rate(const cc_code *c) { return c ? static_cast<double>(c->tab.l) / c->tab.n : 0.0; }

int cc_to_string(const cc_code *c, char *out, size_t cap) {
  if (!c || !out || cap == 0) return CC_ERR_INVALID_ARGUMENT;
  std::snprintf(out, cap, "%s", c->name.c_str());
  return CC_OK;
}

int cc_get_poly(const cc_code *c, int which, uint8_t *out, size_t cap) {
  if (!c || !out || c->wide) return -1;
  const std::vector<uint8_t> *v = which == 0 ? &c->tab.g : which == 1 ? &c->tab.h : which == 2 ? &c->tab.roots : nullptr;
  if (!v || v->size() > cap) return -1;
  std::memcpy(out, v->data(), v->size());
  return static_cast<int>(v->size());
}

int cc_get_poly_u16(const cc_code *c, int which, uint16_t *out, size_t cap) {
  if (!c || !out || which < 0 || which > 2) return -1;
  if (c->wide) {
    const std::vector<uint16_t> &v = which == 0 ? c->tab16.g : which == 1 ? c->tab16.h : c->tab16.roots;
    if (v.size() > cap) return -1;
    std::copy(v.begin(), v.end(), out);
    return static_cast<int>(v.size());
  }
  const std::vector<uint8_t> &v = which == 0 ? c->tab.g : which == 1 ? c->tab.h : c->tab.roots;
  if (v.size() > cap) return -1;
  std::copy(v.begin(), v.end(), out);
  return static_cast<int>(v.size());
}

uint32_t cc_q(const cc_code *c) { return c ? c->tab.q : 0; }

int cc_get_H(const cc_code *c, uint8_t *H) {
  if (!c || !H) return CC_ERR_INVALID_ARGUMENT;
  if (c->wide && !c->tab.binary_h) return CC_ERR_UNSUPPORTED;  // 16-bit entries: not through the byte getter
  const unsigned n = c->tab.n;
  if (c->matrix_only) {
    std::memcpy(H, c->custom_H.data(), c->custom_H.size());
    return CC_OK;
  }
  for (unsigned i = 0; i < c->tab.k; ++i)
    for (unsigned j = 0; j < n; ++j) H[i * n + j] = c->tab.row0[(j + n - i) % n];
  return CC_OK;
}

double cc_sigma(const cc_code *c, double ebno_db) {
  if (!c) return 0.0;
  return 1.0 / std::sqrt(2.0 * cc_rate(c) * std::pow(10.0, ebno_db / 10.0));  // simulation.c++:83-85
}

/* ------------------------------ soft decode ------------------------------ */

int cc_correct_soft_batch_dev(const cc_code *code, const float *d_llr, const uint16_t *d_erasures,
                              const uint32_t *d_erasure_offsets, uint8_t *d_hard, float *d_L, uint16_t *d_iters,
                              int32_t *d_status, size_t B, void *stream) {
  if (!code || (B && (!d_llr || !d_hard))) return CC_ERR_INVALID_ARGUMENT;
  if ((d_erasures == nullptr) != (d_erasure_offsets == nullptr)) return CC_ERR_INVALID_ARGUMENT;
  if (!code->soft) {
    set_last_error("code was created with a hard-decision algorithm; use cc_correct_hard_f32_batch");
    return CC_ERR_INVALID_ARGUMENT;
  }
  if (code->device == CC_DEVICE_NONE) return CC_ERR_NO_DEVICE;
  DeviceGuard guard(code->device);
  return launch_minsum(code, d_llr, d_erasures, d_erasure_offsets, d_hard, d_L, d_iters, d_status, B,
                       static_cast<hipStream_t>(stream));
}

int cc_correct_soft_batch(const cc_code *code, const float *llr, const uint16_t *erasures,
                          const uint32_t *erasure_offsets, uint8_t *hard, float *L, uint16_t *iters, int32_t *status,
                          size_t B) {
  if (!code || (B && (!llr || !hard))) return CC_ERR_INVALID_ARGUMENT;
  if ((erasures == nullptr) != (erasure_offsets == nullptr)) return CC_ERR_INVALID_ARGUMENT;
  if (!code->soft) {
    set_last_error("code was created with a hard-decision algorithm; use cc_correct_hard_f32_batch");
    return CC_ERR_INVALID_ARGUMENT;
  }
  if (code->device == CC_DEVICE_NONE) return CC_ERR_NO_DEVICE;
  if (B == 0) return CC_OK;
  const size_t n = code->tab.n;
  if (erasures) {
    const size_t ne = erasure_offsets[B];
    for (size_t e = 0; e < ne; ++e)
      if (erasures[e] >= n) return CC_ERR_INVALID_ARGUMENT;  // copy.at(erasure) would throw, cyclic.h:261
  }
  DeviceGuard guard(code->device);
  StageLock sl(code);
  if (sl.rc != CC_OK) return sl.rc;
  HostStage &st = *sl.st;
  const size_t CH = chunk_frames(n * sizeof(float), B);
  if (B > 2 * CH)
    if (int rc_fs = st.fresh_streams()) return rc_fs;
  // pageable caller memory goes through the page-locked ring (HostStage::upload / download); buffers the DMA engines
  // reach themselves (hipHostMalloc, hipHostRegister) are used in place
  const bool dma_in = HostStage::dma_ready(llr), dma_out = HostStage::dma_ready(hard),
             dma_it = iters && HostStage::dma_ready(iters), dma_st = status && HostStage::dma_ready(status);
  size_t k = 0;
  for (size_t c0 = 0; c0 < B; c0 += CH, ++k) {
    const int slot = static_cast<int>(k & 1);
    const size_t m = B - c0 < CH ? B - c0 : CH;
    hipStream_t s = st.stream[slot];
    if (int rc = st.retire(slot)) return rc;  // the chunk that used this slot's buffers two turns ago is home
    float *d_llr = nullptr, *d_L = nullptr;
    uint8_t *d_hard = nullptr;
    uint16_t *d_iters = nullptr;
    int32_t *d_status = nullptr;
    const uint16_t *d_er = nullptr;
    const uint32_t *d_off = nullptr;
    if (int rc = st.get(slot, 0, m * n, &d_llr)) return rc;
    if (int rc = st.get(slot, 1, m * n, &d_hard)) return rc;
    if (int rc = st.get(slot, 2, m, &d_iters)) return rc;
    if (int rc = st.get(slot, 3, m, &d_status)) return rc;
    if (L)
      if (int rc = st.get(slot, 4, m * n, &d_L)) return rc;
    if (int rc = st.upload(slot, 0, d_llr, llr + c0 * n, m * n * sizeof(float), dma_in)) return rc;
    if (int rc = upload_erasures(st, slot, 5, erasures, erasure_offsets, c0, m, &d_er, &d_off)) return rc;
    if (int rc = launch_minsum(code, d_llr, d_er, d_off, d_hard, d_L, d_iters, d_status, m, s)) return rc;
    if (int rc = st.download(slot, 1, hard + c0 * n, d_hard, m * n, dma_out)) return rc;
    if (L)
      if (int rc = st.download(slot, 4, L + c0 * n, d_L, m * n * sizeof(float), HostStage::dma_ready(L))) return rc;
    if (iters)
      if (int rc = st.download(slot, 2, iters + c0, d_iters, m * sizeof(uint16_t), dma_it)) return rc;
    if (status)
      if (int rc = st.download(slot, 3, status + c0, d_status, m * sizeof(int32_t), dma_st)) return rc;
  }
  return st.drain();
}


/* ------------------------------ hard decode ------------------------------ */

static int hard_supported(const cc_code *code, bool erasures) {
  if (int rc = not_wide(code)) return rc;
  if (code->soft) {
    set_last_error("code was created with a min-sum algorithm; use cc_correct_soft_batch");
    return CC_ERR_INVALID_ARGUMENT;
  }
  if (code->device == CC_DEVICE_NONE) return CC_ERR_NO_DEVICE;
  // more than 64 syndromes, the Euklid tag with 2t > 63 and erasure decoding with 2t > 32 (Euklid) run
  // algebraic_long.hip (four locator coefficients per lane) -- launch_algebraic routes them
  if (code->tab.family == CC_FAMILY_RS && (code->desc.mu != 1 || code->desc.step != 1)) {
    set_last_error("RS error values on the device assume roots alpha^1..alpha^2t (mu = step = 1), as rs.h:55-69 does");
    return CC_ERR_UNSUPPORTED;
  }
  if (erasures && code->desc.algorithm == CC_ALG_PGZ && code->tab.family == CC_FAMILY_RS) {
    // std::runtime_error "The PGZ-Algorithm does not support erasure decoding" (hard_decision.h:66-68);
    // for BCH the two-trial rule of bch.h:97-149 applies (launch_pgz_erasures)
    set_last_error("The PGZ-Algorithm does not support erasure decoding");
    return CC_ERR_UNSUPPORTED;
  }
  return CC_OK;
}

int cc_correct_hard_batch_dev(const cc_code *code, const uint8_t *d_in, const uint16_t *d_erasures,
                              const uint32_t *d_erasure_offsets, uint8_t *d_out, int32_t *d_nerr, int32_t *d_status,
                              size_t B, void *stream) {
  if (!code || (B && (!d_in || !d_out))) return CC_ERR_INVALID_ARGUMENT;
  if ((d_erasures == nullptr) != (d_erasure_offsets == nullptr)) return CC_ERR_INVALID_ARGUMENT;
  const int rc = hard_supported(code, d_erasures != nullptr);
  if (rc != CC_OK) return rc;
  DeviceGuard guard(code->device);
  if (d_erasures && code->desc.algorithm == CC_ALG_PGZ)
    return launch_pgz_erasures(code, d_in, d_erasures, d_erasure_offsets, d_out, d_nerr, d_status, B,
                               static_cast<hipStream_t>(stream));
  return launch_algebraic(code, false, d_in, d_erasures, d_erasure_offsets, d_out, d_nerr, d_status, B,
                          static_cast<hipStream_t>(stream));
}

// bit = (x < 0) of a soft value (cyclic.h:163-184) as a byte: the Peterson-Gorenstein-Zierler erasure rule of
// bch.h:97-149 re-decodes the word with the erased positions forced to 0 and to 1, on symbols
static __global__ void sign_bytes_kernel(const float *__restrict__ in, uint8_t *__restrict__ out, unsigned long long count) {
  for (unsigned long long i = blockIdx.x * static_cast<unsigned long long>(blockDim.x) + threadIdx.x; i < count;
       i += static_cast<unsigned long long>(gridDim.x) * blockDim.x)
    out[i] = in[i] < 0.0f ? 1 : 0;
}

// hard decoding of device-resident words; float input + PGZ + erasures goes through a byte image of the signs
static int hard_dev(const cc_code *code, bool float_in, const void *d_in, const uint16_t *d_er, const uint32_t *d_off,
                    uint8_t *d_out, int32_t *d_nerr, int32_t *d_status, size_t B, hipStream_t stream) {
  if (!(d_er && code->desc.algorithm == CC_ALG_PGZ))
    return launch_algebraic(code, float_in, d_in, d_er, d_off, d_out, d_nerr, d_status, B, stream);
  if (!float_in)
    return launch_pgz_erasures(code, static_cast<const uint8_t *>(d_in), d_er, d_off, d_out, d_nerr, d_status, B, stream);
  uint8_t *bytes = nullptr;
  const size_t count = B * code->tab.n;
  CC_HIP_TRY(workspace_alloc(code, reinterpret_cast<void **>(&bytes), count + 16, stream));
  hipLaunchKernelGGL(sign_bytes_kernel, dim3(code->num_cus * 8), dim3(256), 0, stream, static_cast<const float *>(d_in),
                     bytes, static_cast<unsigned long long>(count));
  const int rc = launch_pgz_erasures(code, bytes, d_er, d_off, d_out, d_nerr, d_status, B, stream);
  (void)hipFreeAsync(bytes, stream);
  return rc;
}

int cc_correct_hard_f32_batch_dev(const cc_code *code, const float *d_in, const uint16_t *d_erasures,
                                  const uint32_t *d_erasure_offsets, uint8_t *d_out, int32_t *d_nerr,
                                  int32_t *d_status, size_t B, void *stream) {
  if (!code || (B && (!d_in || !d_out))) return CC_ERR_INVALID_ARGUMENT;
  if ((d_erasures == nullptr) != (d_erasure_offsets == nullptr)) return CC_ERR_INVALID_ARGUMENT;
  const int rc = hard_supported(code, d_erasures != nullptr);
  if (rc != CC_OK) return rc;
  DeviceGuard guard(code->device);
  return hard_dev(code, true, d_in, d_erasures, d_erasure_offsets, d_out, d_nerr, d_status, B,
                  static_cast<hipStream_t>(stream));
}

static int hard_host(const cc_code *code, bool float_in, const void *in, const uint16_t *erasures,
                     const uint32_t *erasure_offsets, uint8_t *out, int32_t *nerr, int32_t *status, size_t B) {
  if (!code || (B && (!in || !out))) return CC_ERR_INVALID_ARGUMENT;
  if ((erasures == nullptr) != (erasure_offsets == nullptr)) return CC_ERR_INVALID_ARGUMENT;
  const int rc = hard_supported(code, erasures != nullptr);
  if (rc != CC_OK) return rc;
  if (B == 0) return CC_OK;
  const size_t n = code->tab.n;
  if (!float_in) {  // Element(v) throws for v outside the field, galois.h:149-152
    const uint8_t *b = static_cast<const uint8_t *>(in);
    const uint8_t mask = static_cast<uint8_t>(~code->tab.n);
    for (size_t i = 0; i < B * n; ++i)
      if (b[i] & mask) return CC_ERR_NOT_IN_FIELD;
  }
  if (erasures) {
    const size_t ne = erasure_offsets[B];
    for (size_t e = 0; e < ne; ++e)
      if (erasures[e] >= n) return CC_ERR_INVALID_ARGUMENT;
  }
  DeviceGuard guard(code->device);
  StageLock sl(code);
  if (sl.rc != CC_OK) return sl.rc;
  HostStage &st = *sl.st;
  const size_t esz = float_in ? sizeof(float) : 1;
  const size_t CH = chunk_frames(n * esz, B);
  if (B > 2 * CH)
    if (int rc_fs = st.fresh_streams()) return rc_fs;
  const bool dma_in = HostStage::dma_ready(in), dma_out = HostStage::dma_ready(out),
             dma_ne = nerr && HostStage::dma_ready(nerr), dma_st = status && HostStage::dma_ready(status);
  size_t k = 0;
  for (size_t c0 = 0; c0 < B; c0 += CH, ++k) {
    const int slot = static_cast<int>(k & 1);
    const size_t m = B - c0 < CH ? B - c0 : CH;
    hipStream_t s = st.stream[slot];
    if (int r = st.retire(slot)) return r;
    uint8_t *d_in = nullptr, *d_out = nullptr;
    int32_t *d_nerr = nullptr, *d_status = nullptr;
    const uint16_t *d_er = nullptr;
    const uint32_t *d_off = nullptr;
    if (int r = st.get(slot, 0, m * n * esz, &d_in)) return r;
    if (int r = st.get(slot, 1, m * n, &d_out)) return r;
    if (int r = st.get(slot, 2, m, &d_nerr)) return r;
    if (int r = st.get(slot, 3, m, &d_status)) return r;
    if (int r = st.upload(slot, 0, d_in, static_cast<const uint8_t *>(in) + c0 * n * esz, m * n * esz, dma_in)) return r;
    if (int r = upload_erasures(st, slot, 5, erasures, erasure_offsets, c0, m, &d_er, &d_off)) return r;
    if (int r = hard_dev(code, float_in, d_in, d_er, d_off, d_out, d_nerr, d_status, m, s)) return r;
    if (int r = st.download(slot, 1, out + c0 * n, d_out, m * n, dma_out)) return r;
    if (nerr)
      if (int r = st.download(slot, 2, nerr + c0, d_nerr, m * sizeof(int32_t), dma_ne)) return r;
    if (status)
      if (int r = st.download(slot, 3, status + c0, d_status, m * sizeof(int32_t), dma_st)) return r;
  }
  return st.drain();
}

int cc_correct_hard_batch(const cc_code *code, const uint8_t *in, const uint16_t *erasures,
                          const uint32_t *erasure_offsets, uint8_t *out, int32_t *nerr, int32_t *status, size_t B) {
  return hard_host(code, false, in, erasures, erasure_offsets, out, nerr, status, B);
}

int cc_correct_hard_f32_batch(const cc_code *code, const float *in, const uint16_t *erasures,
                              const uint32_t *erasure_offsets, uint8_t *out, int32_t *nerr, int32_t *status, size_t B) {
  return hard_host(code, true, in, erasures, erasure_offsets, out, nerr, status, B);
}

/* ------------------------------ encode / extract ------------------------------ */

int cc_encode_batch_dev(const cc_code *code, const uint8_t *d_msg, uint8_t *d_cw, size_t B, void *stream) {
  if (!code || (B && (!d_msg || !d_cw))) return CC_ERR_INVALID_ARGUMENT;
  if (needs_code(code) != CC_OK) return CC_ERR_INVALID_ARGUMENT;
  if (int rc = not_wide(code)) return rc;
  if (code->device == CC_DEVICE_NONE) return CC_ERR_NO_DEVICE;
  DeviceGuard guard(code->device);
  return launch_encode(code, d_msg, d_cw, B, static_cast<hipStream_t>(stream));
}

int cc_extract_batch_dev(const cc_code *code, const uint8_t *d_cw, uint8_t *d_msg, size_t B, void *stream) {
  if (!code || (B && (!d_cw || !d_msg))) return CC_ERR_INVALID_ARGUMENT;
  if (needs_code(code) != CC_OK) return CC_ERR_INVALID_ARGUMENT;
  if (int rc = not_wide(code)) return rc;
  if (code->device == CC_DEVICE_NONE) return CC_ERR_NO_DEVICE;
  DeviceGuard guard(code->device);
  return launch_extract(code, d_cw, d_msg, B, static_cast<hipStream_t>(stream));
}

static int byte_map_host(const cc_code *code, bool encode, const uint8_t *src, uint8_t *dst, size_t B) {
  if (!code || (B && (!src || !dst))) return CC_ERR_INVALID_ARGUMENT;
  if (needs_code(code) != CC_OK) return CC_ERR_INVALID_ARGUMENT;
  if (int rc = not_wide(code)) return rc;
  if (code->device == CC_DEVICE_NONE) return CC_ERR_NO_DEVICE;
  if (B == 0) return CC_OK;
  const size_t n = code->tab.n, l = code->tab.l;
  const size_t in_w = encode ? l : n, out_w = encode ? n : l;
  if (encode) {  // Element(e) throws for values outside the field, galois.h:149-152 via cyclic.h:300-301
    const uint8_t mask = static_cast<uint8_t>(~code->tab.n);
    for (size_t i = 0; i < B * in_w; ++i)
      if (src[i] & mask) return CC_ERR_NOT_IN_FIELD;
  }
  DeviceGuard guard(code->device);
  StageLock sl(code);
  if (sl.rc != CC_OK) return sl.rc;
  HostStage &st = *sl.st;
  const size_t CH = chunk_frames(n, B);
  if (B > 2 * CH)
    if (int rc_fs = st.fresh_streams()) return rc_fs;
  const bool dma_in = HostStage::dma_ready(src), dma_out = HostStage::dma_ready(dst);
  size_t k = 0;
  for (size_t c0 = 0; c0 < B; c0 += CH, ++k) {
    const int slot = static_cast<int>(k & 1);
    const size_t m = B - c0 < CH ? B - c0 : CH;
    hipStream_t s = st.stream[slot];
    if (int r = st.retire(slot)) return r;
    uint8_t *d_src = nullptr, *d_dst = nullptr;
    if (int r = st.get(slot, 0, m * in_w, &d_src)) return r;
    if (int r = st.get(slot, 1, m * out_w, &d_dst)) return r;
    if (int r = st.upload(slot, 0, d_src, src + c0 * in_w, m * in_w, dma_in)) return r;
    const int rc = encode ? launch_encode(code, d_src, d_dst, m, s) : launch_extract(code, d_src, d_dst, m, s);
    if (rc != CC_OK) return rc;
    if (int r = st.download(slot, 1, dst + c0 * out_w, d_dst, m * out_w, dma_out)) return r;
  }
  return st.drain();
}

int cc_encode_batch(const cc_code *code, const uint8_t *msg, uint8_t *cw, size_t B) {
  return byte_map_host(code, true, msg, cw, B);
}
int cc_extract_batch(const cc_code *code, const uint8_t *cw, uint8_t *msg, size_t B) {
  return byte_map_host(code, false, cw, msg, B);
}

/* ------------------------------ decode = correct + extract ------------------------------ */

static int decode_host(const cc_code *code, bool float_in, const void *in, const uint16_t *erasures,
                       const uint32_t *erasure_offsets, uint8_t *msg, uint8_t *words, uint16_t *iters, int32_t *nerr,
                       int32_t *status, size_t B) {
  if (!code || (B && (!in || !msg))) return CC_ERR_INVALID_ARGUMENT;
  if (needs_code(code) != CC_OK) return CC_ERR_INVALID_ARGUMENT;
  if (int rc = not_wide(code)) return rc;
  if (B == 0) return CC_OK;
  std::vector<uint8_t> tmp;
  if (!words) {
    tmp.resize(B * code->tab.n);
    words = tmp.data();
  }
  int rc;
  if (code->soft) {
    if (!float_in) {
      set_last_error("min-sum needs a signed (soft) input sequence");
      return CC_ERR_INVALID_ARGUMENT;
    }
    rc = cc_correct_soft_batch(code, static_cast<const float *>(in), erasures, erasure_offsets, words, nullptr, iters,
                               status, B);
  } else if (float_in) {
    rc = cc_correct_hard_f32_batch(code, static_cast<const float *>(in), erasures, erasure_offsets, words, nerr, status, B);
  } else {
    rc = cc_correct_hard_batch(code, static_cast<const uint8_t *>(in), erasures, erasure_offsets, words, nerr, status, B);
  }
  if (rc != CC_OK) return rc;
  return cc_extract_batch(code, words, msg, B);
}

int cc_decode_hard_batch(const cc_code *code, const uint8_t *in, const uint16_t *erasures,
                         const uint32_t *erasure_offsets, uint8_t *msg, uint8_t *words, int32_t *nerr, int32_t *status,
                         size_t B) {
  return decode_host(code, false, in, erasures, erasure_offsets, msg, words, nullptr, nerr, status, B);
}

int cc_decode_soft_batch(const cc_code *code, const float *y, const uint16_t *erasures, const uint32_t *erasure_offsets,
                         uint8_t *msg, uint8_t *words, uint16_t *iters, int32_t *status, size_t B) {
  return decode_host(code, true, y, erasures, erasure_offsets, msg, words, iters, nullptr, status, B);
}

/* ------------------------------ q = 9 .. 15: 16-bit symbols (wide.hip) ------------------------------ */

// lane-count limits of wide_correct_kernel (wide.hip), the same as the byte path's hard_supported: one lane per
// coefficient of x^2t (Euklid), of S(x) u(x) (Euklid with erasures, degree < 2t + erasures <= 4t) and of the erasure
// locator (Berlekamp-Massey pre-load, degree <= 2t)
static int wide_hard_supported(const cc_code *code, bool erasures) {
  const size_t t2 = code->tab16.roots.size();  // (a 16-bit handle keeps its vectors in tab16; tab has the scalars only)
  if (code->desc.algorithm == CC_ALG_EUKLID && (t2 > 63 || (erasures && t2 > 32))) {
    set_last_error("the Euklid tag on the device handles t <= 31 (t <= 16 with erasures)");
    return CC_ERR_UNSUPPORTED;
  }
  if (erasures && t2 > 63 && code->desc.algorithm != CC_ALG_PGZ) {  // (the PGZ trials run without erasures)
    set_last_error("erasure decoding on the device handles t <= 31 (one lane per coefficient of the erasure locator)");
    return CC_ERR_UNSUPPORTED;
  }
  if (erasures && code->desc.algorithm == CC_ALG_PGZ && code->tab.family == CC_FAMILY_RS) {
    set_last_error("The PGZ-Algorithm does not support erasure decoding");  // hard_decision.h:66-68
    return CC_ERR_UNSUPPORTED;
  }
  return CC_OK;  // (BCH with the PGZ tag and erasures: the two-trial rule of bch.h:97-149, launch_wide_pgz_erasures)
}
static int wide_ready(const cc_code *code) {
  if (!code->wide) {
    set_last_error("the _u16 entry points serve GF(2^q) with q > 8; this handle has byte symbols");
    return CC_ERR_UNSUPPORTED;
  }
  if (code->device == CC_DEVICE_NONE) return CC_ERR_NO_DEVICE;
  return CC_OK;
}

int cc_encode_batch_u16_dev(const cc_code *code, const uint16_t *d_msg, uint16_t *d_cw, size_t B, void *stream) {
  if (!code || (B && (!d_msg || !d_cw))) return CC_ERR_INVALID_ARGUMENT;
  if (int rc = wide_ready(code)) return rc;
  DeviceGuard guard(code->device);
  return launch_wide_encode(code, d_msg, d_cw, B, static_cast<hipStream_t>(stream));
}

int cc_extract_batch_u16_dev(const cc_code *code, const uint16_t *d_cw, uint16_t *d_msg, size_t B, void *stream) {
  if (!code || (B && (!d_cw || !d_msg))) return CC_ERR_INVALID_ARGUMENT;
  if (int rc = wide_ready(code)) return rc;
  DeviceGuard guard(code->device);
  return launch_wide_extract(code, d_cw, d_msg, B, static_cast<hipStream_t>(stream));
}

int cc_correct_hard_batch_u16_dev(const cc_code *code, const uint16_t *d_in, const uint16_t *d_erasures,
                                  const uint32_t *d_erasure_offsets, uint16_t *d_out, int32_t *d_nerr,
                                  int32_t *d_status, size_t B, void *stream) {
  if (!code || (B && (!d_in || !d_out))) return CC_ERR_INVALID_ARGUMENT;
  if ((d_erasures == nullptr) != (d_erasure_offsets == nullptr)) return CC_ERR_INVALID_ARGUMENT;
  if (int rc = wide_ready(code)) return rc;
  if (int rc = wide_hard_supported(code, d_erasures != nullptr)) return rc;
  DeviceGuard guard(code->device);
  return launch_wide_correct(code, d_in, d_erasures, d_erasure_offsets, d_out, d_nerr, d_status, B,
                             static_cast<hipStream_t>(stream));
}

// host pointers: same chunked staging as the byte entry points; kind 0 = encode, 1 = extract
static int wide_map_host(const cc_code *code, int kind, const uint16_t *src, uint16_t *dst, size_t B) {
  if (!code || (B && (!src || !dst))) return CC_ERR_INVALID_ARGUMENT;
  if (int rc = wide_ready(code)) return rc;
  if (B == 0) return CC_OK;
  const size_t n = code->tab.n, l = code->tab.l;
  const size_t in_w = kind == 0 ? l : n, out_w = kind == 0 ? n : l;
  if (kind == 0)
    for (size_t i = 0; i < B * in_w; ++i)
      if (src[i] > n) return CC_ERR_NOT_IN_FIELD;  // Element(e) throws, galois.h:149-152
  DeviceGuard guard(code->device);
  StageLock sl(code);
  if (sl.rc != CC_OK) return sl.rc;
  HostStage &st = *sl.st;
  const size_t CH = chunk_frames(n * 2, B);
  if (B > 2 * CH)
    if (int rc_fs = st.fresh_streams()) return rc_fs;
  const bool dma_in = HostStage::dma_ready(src), dma_out = HostStage::dma_ready(dst);
  size_t k = 0;
  for (size_t c0 = 0; c0 < B; c0 += CH, ++k) {
    const int slot = static_cast<int>(k & 1);
    const size_t m = B - c0 < CH ? B - c0 : CH;
    hipStream_t s = st.stream[slot];
    if (int r = st.retire(slot)) return r;
    uint16_t *d_src = nullptr, *d_dst = nullptr;
    if (int r = st.get(slot, 0, m * in_w, &d_src)) return r;
    if (int r = st.get(slot, 1, m * out_w, &d_dst)) return r;
    if (int r = st.upload(slot, 0, d_src, src + c0 * in_w, m * in_w * 2, dma_in)) return r;
    const int rc = kind == 0 ? launch_wide_encode(code, d_src, d_dst, m, s) : launch_wide_extract(code, d_src, d_dst, m, s);
    if (rc != CC_OK) return rc;
    if (int r = st.download(slot, 1, dst + c0 * out_w, d_dst, m * out_w * 2, dma_out)) return r;
  }
  return st.drain();
}

int cc_encode_batch_u16(const cc_code *code, const uint16_t *msg, uint16_t *cw, size_t B) {
  return wide_map_host(code, 0, msg, cw, B);
}
int cc_extract_batch_u16(const cc_code *code, const uint16_t *cw, uint16_t *msg, size_t B) {
  return wide_map_host(code, 1, cw, msg, B);
}

int cc_correct_hard_batch_u16(const cc_code *code, const uint16_t *in, const uint16_t *erasures,
                              const uint32_t *erasure_offsets, uint16_t *out, int32_t *nerr, int32_t *status,
                              size_t B) {
  if (!code || (B && (!in || !out))) return CC_ERR_INVALID_ARGUMENT;
  if ((erasures == nullptr) != (erasure_offsets == nullptr)) return CC_ERR_INVALID_ARGUMENT;
  if (int rc = wide_ready(code)) return rc;
  if (int rc = wide_hard_supported(code, erasures != nullptr)) return rc;
  if (B == 0) return CC_OK;
  const size_t n = code->tab.n;
  for (size_t i = 0; i < B * n; ++i)
    if (in[i] > n) return CC_ERR_NOT_IN_FIELD;
  if (erasures) {
    const size_t ne = erasure_offsets[B];
    for (size_t e = 0; e < ne; ++e)
      if (erasures[e] >= n) return CC_ERR_INVALID_ARGUMENT;
  }
  DeviceGuard guard(code->device);
  StageLock sl(code);
  if (sl.rc != CC_OK) return sl.rc;
  HostStage &st = *sl.st;
  const size_t CH = chunk_frames(n * 2, B);
  if (B > 2 * CH)
    if (int rc_fs = st.fresh_streams()) return rc_fs;
  const bool dma_in = HostStage::dma_ready(in), dma_out = HostStage::dma_ready(out),
             dma_ne = nerr && HostStage::dma_ready(nerr), dma_st = status && HostStage::dma_ready(status);
  size_t k = 0;
  for (size_t c0 = 0; c0 < B; c0 += CH, ++k) {
    const int slot = static_cast<int>(k & 1);
    const size_t m = B - c0 < CH ? B - c0 : CH;
    hipStream_t s = st.stream[slot];
    if (int r = st.retire(slot)) return r;
    uint16_t *d_in = nullptr, *d_out = nullptr;
    int32_t *d_nerr = nullptr, *d_status = nullptr;
    const uint16_t *d_er = nullptr;
    const uint32_t *d_off = nullptr;
    if (int r = st.get(slot, 0, m * n, &d_in)) return r;
    if (int r = st.get(slot, 1, m * n, &d_out)) return r;
    if (int r = st.get(slot, 2, m, &d_nerr)) return r;
    if (int r = st.get(slot, 3, m, &d_status)) return r;
    if (int r = st.upload(slot, 0, d_in, in + c0 * n, m * n * 2, dma_in)) return r;
    if (int r = upload_erasures(st, slot, 5, erasures, erasure_offsets, c0, m, &d_er, &d_off)) return r;
    if (int r = launch_wide_correct(code, d_in, d_er, d_off, d_out, d_nerr, d_status, m, s)) return r;
    if (int r = st.download(slot, 1, out + c0 * n, d_out, m * n * 2, dma_out)) return r;
    if (nerr)
      if (int r = st.download(slot, 2, nerr + c0, d_nerr, m * sizeof(int32_t), dma_ne)) return r;
    if (status)
      if (int r = st.download(slot, 3, status + c0, d_status, m * sizeof(int32_t), dma_st)) return r;
  }
  return st.drain();
}

/* ------------------------------ Monte-Carlo ------------------------------ */

static int mc_supported(const cc_code *code) {
  if (needs_code(code) != CC_OK) return CC_ERR_INVALID_ARGUMENT;
  if (int rc = not_wide(code)) return rc;
  if (code->device == CC_DEVICE_NONE) return CC_ERR_NO_DEVICE;
  if (code->tab.family != CC_FAMILY_BCH) {
    set_last_error("the BPSK/AWGN Monte-Carlo channel is defined for binary (BCH) codes");
    return CC_ERR_UNSUPPORTED;
  }
  return CC_OK;
}

int cc_mc_run_dev(const cc_code *code, double ebno_db, uint64_t seed, uint64_t first_frame, size_t frames,
                  int random_codewords, uint64_t *d_counters, void *stream) {
  if (!code || !d_counters) return CC_ERR_INVALID_ARGUMENT;
  const int rc = mc_supported(code);
  if (rc != CC_OK) return rc;
  if (random_codewords && code->desc.coding != CC_CODING_DIVISION && code->desc.coding != CC_CODING_MULTIPLICATION)
    return CC_ERR_INVALID_ARGUMENT;
  DeviceGuard guard(code->device);
  return mc_run(const_cast<cc_code *>(code), ebno_db, seed, first_frame, frames, random_codewords, d_counters,
                static_cast<hipStream_t>(stream));
}

int cc_awgn_llr_dev(const cc_code *code, double ebno_db, uint64_t seed, uint64_t first_frame, size_t frames,
                    int random_codewords, float *d_llr, uint8_t *d_sent, void *stream) {
  if (!code || (frames && !d_llr)) return CC_ERR_INVALID_ARGUMENT;
  const int rc = mc_supported(code);
  if (rc != CC_OK) return rc;
  DeviceGuard guard(code->device);
  return mc_awgn(const_cast<cc_code *>(code), ebno_db, seed, first_frame, frames, random_codewords, d_llr, d_sent,
                 static_cast<hipStream_t>(stream));
}

int cc_diag_table(const cc_code *code, uint16_t *out, size_t cap, uint32_t *D, uint32_t *LPF, uint32_t *links) {
  if (!code || !out) return -1;
  if (code->matrix_only || code->wide || !code->custom_H.empty()) return 0;
  const DiagGeometry *g = diag_geometry(code->tab);
  if (!g) return 0;
  const std::vector<uint16_t> t = build_diag_table(code->tab, g->D, g->LPF, g->np, g->gap);
  if (t.empty() || t.size() > cap) return -1;
  std::copy(t.begin(), t.end(), out);
  if (D) *D = static_cast<uint32_t>(g->D);
  if (LPF) *LPF = static_cast<uint32_t>(g->LPF);
  if (links) *links = static_cast<uint32_t>(g->np);
  return static_cast<int>(t.size());
}

int cc_kernel_info(const cc_code *code, char *name, size_t cap, uint32_t *frames_per_workgroup,
                   uint32_t *threads_per_workgroup, uint32_t *lds_bytes) {
  if (!code) return CC_ERR_INVALID_ARGUMENT;
  std::string nm = "algebraic_kernel";
  uint32_t f = 4, t = 256, l = 1024;
  if (code->soft) {
    minsum_kernel_info(code, nm, f, t, l);
  } else if (algebraic_chunk_supported(code, false)) {
    if (bitslice_supported(code)) {
      nm = "algebraic_chunk_kernel on bit planes: bitslice_fused_syndrome / chunk_bm_reg / bitslice_chien / chunk_fixl kernels (erasures: chunk_bm with the BM tag, algebraic_kernel with Euklid; small calls: algebraic_kernel)";
      f = 256;
    } else {
      nm = "algebraic_chunk_kernel<FPW=32> (algebraic_kernel with erasures)";
      f = 128;
    }
  }
  if (name && cap) std::snprintf(name, cap, "%s", nm.c_str());
  if (frames_per_workgroup) *frames_per_workgroup = f;
  if (threads_per_workgroup) *threads_per_workgroup = t;
  if (lds_bytes) *lds_bytes = l;
  return CC_OK;
}



}  // extern "C"
#pragma GCC visibility pop
