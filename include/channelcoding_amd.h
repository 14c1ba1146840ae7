/* channelcoding_amd.h -- C ABI of the MI355X-native BCH / Reed-Solomon decoder.
 *
 * This is the drop-in boundary for the hot path of hannesweisbach/channelcoding:
 * the reference has no FFI of its own (it is a header-only C++14 template
 * library), so every entry point below names the reference member it replaces.
 * Citations are file:line relative to the reference's repository root.
 * The header-only C++ facade include/channelcoding_amd/cyclic.hpp re-creates the
 * reference's template API (cyclic::primitive_bch<>, cyclic::rs<>, tags,
 * decoding_failure) on top of these functions; INTEGRATION.md shows the binding.
 *
 * Conventions (identical to the reference, src/codes/cyclic.h:163-184, :289-344):
 *   - index i of a word is the coefficient of x^i;
 *   - a codeword has n symbols, the k = deg g parity symbols in positions
 *     0..k-1 and the l = n - k information symbols in positions k..n-1
 *     (the reference calls the parity count `k` and the information count `l`,
 *     cyclic.h:104-105 -- so do we);
 *   - symbols are one byte each (q <= 8), frames are contiguous: frame f of a
 *     batch starts at element f*n (or f*l for messages);
 *   - soft values are float32 LLR-like channel values, positive <=> bit 0
 *     (BPSK 0 -> +1); a hard decision is bit = (x < 0) (codes.h:43-52);
 *   - all `_dev` entry points take DEVICE pointers and a hipStream_t passed as
 *     void* (NULL = the null stream); they enqueue work and return without
 *     synchronising.  The plain entry points take HOST pointers, copy, run and
 *     synchronise.
 *   - nothing here falls back to the CPU: without a usable HIP device the
 *     create call fails with CC_ERR_NO_DEVICE.
 */
#ifndef CHANNELCODING_AMD_H
#define CHANNELCODING_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CC_ABI_VERSION 1

typedef struct cc_code cc_code; /* opaque handle: immutable after creation, thread-safe */

typedef enum cc_status {
  CC_OK = 0,
  CC_ERR_INVALID_ARGUMENT = 1, /* NULL pointer, bad enum, q/t out of range                       */
  CC_ERR_UNSUPPORTED = 2,      /* valid in the reference, not (yet) available on the device path */
  CC_ERR_NO_DEVICE = 3,        /* no HIP device / HIP runtime failure at creation                */
  CC_ERR_HIP = 4,              /* a HIP call failed; see cc_last_error()                         */
  CC_ERR_OUT_OF_MEMORY = 5,
  CC_ERR_LENGTH = 6,           /* std::runtime_error "wrong size" of cyclic.h:213-218, :291-296  */
  CC_ERR_NOT_IN_FIELD = 7      /* "Value is not an element of the field." galois.h:149-152       */
} cc_status;

/* cyclic::primitive_bch (src/codes/bch.h:16-19) / cyclic::rs (src/codes/rs.h:6-10) */
typedef enum cc_family { CC_FAMILY_BCH = 0, CC_FAMILY_RS = 1 } cc_family;

/* Algorithm tags: src/codes/hard_decision.h:15-24 and src/codes/soft_decision.h:20-73 */
typedef enum cc_algorithm {
  CC_ALG_PGZ = 0,    /* peterson_gorenstein_zierler_tag (the reference's default).  NOT run as Peterson-Gorenstein-
                      * Zierler on the device: bounded-distance decoding = Berlekamp-Massey + "locator degree <= t"
                      * (+ the two-trial erasure rule of bch.h:97-149 for BCH).  Same corrected words and failures
                      * as the reference wherever its Gauss elimination is sound; it is not on 1-4 % of decodable
                      * RS frames (linear_equation_system.h:24-35), which this decodes */
  CC_ALG_BM = 1,     /* berlekamp_massey_tag                                        */
  CC_ALG_EUKLID = 2, /* euklid_tag.  With erasures and 2t <= 32: Sugiyama's remainder sequence itself (hard_decision.h:
                      * 157-196).  Otherwise bounded-distance decoding on the Berlekamp-Massey locator, like PGZ: the
                      * remainder sequence ends with a locator of degree <= (2t + erasures) / 2, a frame decodes exactly
                      * when the key equation has its unique solution within that bound, and that is the solution
                      * Berlekamp-Massey finds -- same corrected words, same frames failing; only WHICH failure text a
                      * hopeless frame gets may differ */
  CC_ALG_MS = 16,    /* min_sum_tag<It>                                             */
  CC_ALG_NMS = 17,   /* normalized_min_sum_tag<It, ratio>          alpha            */
  CC_ALG_OMS = 18,   /* offset_min_sum_tag<It, ratio>              beta             */
  CC_ALG_SCMS1 = 19, /* self_correcting_1_min_sum_tag<It>                           */
  CC_ALG_SCMS2 = 20, /* self_correcting_2_min_sum_tag<It>                           */
  CC_ALG_2DNMS = 21  /* normalized_2d_min_sum_tag<It, Alpha, Beta> alpha, beta      */
} cc_algorithm;

/* cyclic.h:19-23 */
typedef enum cc_coding { CC_CODING_DIVISION = 0, CC_CODING_MULTIPLICATION = 1 } cc_coding;

/* Stop rule of the min-sum driver (soft_decision.h:185-186).  The reference as
 * shipped never iterates (src/math/matrix.h:50 makes H*b empty); see DESIGN.md. */
typedef enum cc_stop_rule {
  CC_STOP_AS_SHIPPED = 0, /* O0: return after the first iteration, never fail (bit-exact with the shipped code) */
  CC_STOP_PUBLISHED = 1,  /* O1: matrix.h:50 repaired, integer dot product: accepts only the all-zero word     */
  CC_STOP_PARITY = 2      /* O2: true GF(2) parity check H b^T = 0 (the intended behaviour; default)           */
} cc_stop_rule;

/* Runtime descriptor replacing the reference's template arguments. */
typedef struct cc_desc {
  uint32_t struct_size; /* = sizeof(cc_desc)                                                   */
  int32_t family;       /* cc_family                                                           */
  uint32_t q;           /* GF(2^q), 2..15; q > 8: name modular_polynomial below, use the _u16 calls */
  uint32_t t;           /* errors<t>; for dmin<d> pass (d-1)/2 (codes.h:14-26)                 */
  uint32_t n;           /* 0 or 2^q-1 (the reference's N is a TODO there too, cyclic.h:66)     */
  uint32_t mu, step;    /* RS only: roots alpha^(mu + i*step), rs.h:18-39; use 1, 1            */
  int32_t coding;       /* cc_coding                                                           */
  int32_t algorithm;    /* cc_algorithm                                                        */
  uint32_t iterations;  /* min-sum only: the tag's Iterations                                  */
  double alpha;         /* NMS / 2D-NMS: the tag's ::alpha (a double; rounded to float on use) */
  double beta;          /* OMS: ::beta (used in double); 2D-NMS: ::beta = Beta::num/Alpha::den */
  int32_t stop_rule;    /* cc_stop_rule                                                        */
  int32_t device;       /* HIP device ordinal, CC_DEVICE_CURRENT, or CC_DEVICE_NONE            */
  uint32_t modular_polynomial; /* math::modular_polynomial<> (galois.h:23-25): bit i = coefficient of x^i, degree q,
                                * primitive.  0 = the default of galois.h:18-20, which exists for q <= 8 only
                                * (default_modular_polynomial, galois.h:57-67) */
  uint32_t reserved;    /* 0 */
} cc_desc;

#define CC_DEVICE_CURRENT (-1)
/* Introspection-only handle: builds g, h, roots, H, dmin, to_string on the host and owns no device
 * resources; every encode/decode/Monte-Carlo call on it fails with CC_ERR_NO_DEVICE (there is no CPU
 * decode path).  Used by host-side tests and by tools that only need the code's constants. */
#define CC_DEVICE_NONE (-2)

/* Per-frame status written by the batch decoders (what the reference signals
 * with decoding_failure, src/codes/codes.h:28-36). */
enum {
  CC_FRAME_OK = 0,
  CC_FRAME_NOT_CONVERGED = 1, /* soft_decision.h:201  "Decoding failure"                                 */
  CC_FRAME_LOCATOR = 2,       /* cyclic.h:134-147 root count != degree; hard_decision.h:103,109,191-192 */
  CC_FRAME_RECHECK = 3,       /* cyclic.h:243-248 "Corrected word is not a codeword"                    */
  CC_FRAME_ERASURES = 4       /* bch.h:105-107 too many erasures                                        */
};

const char *cc_version(void);
const char *cc_status_string(int status);
/* thread-local text of the last failing HIP call */
const char *cc_last_error(void);

/* ---- construction: primitive_bch() bch.h:152, rs() rs.h:87, cyclic ctor cyclic.h:270-280 ---- */
int cc_code_create(const cc_desc *desc, cc_code **out);
void cc_code_destroy(cc_code *code);
void cc_desc_init(cc_desc *desc); /* zero + defaults: BCH, PGZ, division, alpha 1, beta 0, O2, device -1 */

/* ---- introspection: public members cyclic.h:94-95,:111 and protected g/h/roots/k/l/dmin :97-106 ---- */
uint32_t cc_n(const cc_code *code);
uint32_t cc_k(const cc_code *code); /* parity symbols = deg g */
uint32_t cc_l(const cc_code *code); /* information symbols    */
uint32_t cc_t(const cc_code *code);
uint32_t cc_dmin(const cc_code *code); /* consecutive_zeroes(g)+1, cyclic.h:186-204 (incl. its over-count for RS) */
double cc_rate(const cc_code *code);   /* l / n */
int cc_to_string(const cc_code *code, char *out, size_t cap); /* "(n, l, dmin)-ALG", cyclic.h:282-287 */
/* which: 0 = g, 1 = h, 2 = syndrome roots; returns the number of symbols written or -1 */
int cc_get_poly(const cc_code *code, int which, uint8_t *out, size_t cap);
/* cyclic::H<uint8_t>() cyclic.h:346-359, k*n bytes row-major */
int cc_get_H(const cc_code *code, uint8_t *H);

/* cyclic::H_alt<uint8_t>() cyclic.h:361-385: binary image of the t x n matrix alpha^(col*(2 row+1)), t*q rows
 * (row r of the power matrix expands to q rows, least significant bit first).  Reproduces the reference's
 * exponent reduction modulo 2^q (galois.h:182-184).  Writes t*q*n bytes; *rows = t*q. */
int cc_get_H_alt(const cc_code *code, uint8_t *H, uint32_t *rows);
/* Min-sum over a caller-supplied parity-check matrix (rows x n bytes, entries 0/1) instead of H(): what the
 * reference spells min_sum<float, U>(code.H_alt<U>(), y, tag) (soft_decision.h:220-295).  The descriptor must
 * name a min-sum algorithm; all other members of the new handle (encode, Monte-Carlo, ...) behave as usual. */
int cc_code_create_with_H(const cc_desc *desc, const uint8_t *H, uint32_t rows, cc_code **out);
/* The free functions min_sum<R, U>(H, y, tag) of soft_decision.h:220-295 on any rows x cols 0/1 matrix
 * (cols <= 2048), no code behind it: only algorithm, iterations, alpha, beta, stop_rule and device of the
 * descriptor are read.  The handle serves cc_correct_soft_batch(_dev), cc_n (= cols), cc_k (= rows), cc_get_H,
 * cc_to_string and cc_kernel_info; every entry point that needs a code returns CC_ERR_INVALID_ARGUMENT. */
int cc_minsum_create(const cc_desc *desc, const uint8_t *H, uint32_t rows, uint32_t cols, cc_code **out);

/* ---- encode: cyclic::encode cyclic.h:289-311 (+ free encode :29-40) ---- */
int cc_encode_batch(const cc_code *code, const uint8_t *msg /* B*l */, uint8_t *cw /* B*n */, size_t B);
int cc_encode_batch_dev(const cc_code *code, const uint8_t *d_msg, uint8_t *d_cw, size_t B, void *stream);

/* ---- hard-decision correct: cyclic::correct / correct_(hard_decision_tag) cyclic.h:207-252,:331-344,
 *      bch.h:85-160.  Erasures in CSR form: frame f owns erasures[erasure_offsets[f] .. erasure_offsets[f+1]);
 *      both pointers NULL = no erasures.  out = corrected word (= hard-decided input when the frame fails),
 *      nerr = number of corrected symbols or -1, status = CC_FRAME_*.  nerr/status may be NULL.  out may be the
 *      same buffer as in (decoding in place; the _dev form then skips its copy of the words). ---- */
int cc_correct_hard_batch(const cc_code *code, const uint8_t *in /* B*n symbols */, const uint16_t *erasures,
                          const uint32_t *erasure_offsets, uint8_t *out /* B*n */, int32_t *nerr, int32_t *status,
                          size_t B);
int cc_correct_hard_batch_dev(const cc_code *code, const uint8_t *d_in, const uint16_t *d_erasures,
                              const uint32_t *d_erasure_offsets, uint8_t *d_out, int32_t *d_nerr, int32_t *d_status,
                              size_t B, void *stream);
/* signed input sequence (cyclic.h:163-173): bit = (x < 0), then as above -- erasures included: the reference's
 * correct_ takes them for any InputSequence (cyclic.h:207-252, bch.h:97-149) */
int cc_correct_hard_f32_batch(const cc_code *code, const float *in /* B*n */, const uint16_t *erasures,
                              const uint32_t *erasure_offsets, uint8_t *out, int32_t *nerr, int32_t *status, size_t B);
int cc_correct_hard_f32_batch_dev(const cc_code *code, const float *d_in, const uint16_t *d_erasures,
                                  const uint32_t *d_erasure_offsets, uint8_t *d_out, int32_t *d_nerr,
                                  int32_t *d_status, size_t B, void *stream);

/* ---- soft-decision correct: cyclic::correct_(soft_decision_tag) cyclic.h:254-267 -> min_sum
 *      soft_decision.h:161-295.  hard = b (B*n bytes, 0/1), L = a-posteriori values (B*n floats, may be
 *      NULL), iters = 0-based index of the returning iteration (= iterations when not converged; may be
 *      NULL), status = CC_FRAME_OK / CC_FRAME_NOT_CONVERGED (may be NULL).  Erasures zero the LLR. ---- */
int cc_correct_soft_batch(const cc_code *code, const float *llr /* B*n */, const uint16_t *erasures,
                          const uint32_t *erasure_offsets, uint8_t *hard /* B*n */, float *L, uint16_t *iters,
                          int32_t *status, size_t B);
int cc_correct_soft_batch_dev(const cc_code *code, const float *d_llr, const uint16_t *d_erasures,
                              const uint32_t *d_erasure_offsets, uint8_t *d_hard, float *d_L, uint16_t *d_iters,
                              int32_t *d_status, size_t B, void *stream);

/* ---- decode = correct + message extraction: cyclic::decode cyclic.h:313-327 (+ free decode :42-51) ---- */
int cc_extract_batch(const cc_code *code, const uint8_t *cw /* B*n */, uint8_t *msg /* B*l */, size_t B);
int cc_extract_batch_dev(const cc_code *code, const uint8_t *d_cw, uint8_t *d_msg, size_t B, void *stream);
/* decode<InputSequence, Return_type>(b, erasures) in one call (cyclic.h:313-327): correct, then take the message
 * of the corrected word (failed frames: of the hard-decided input, as cc_correct_* leaves it).  The input is
 * symbols for cc_decode_hard_batch, signed channel values for cc_decode_soft_batch -- with a hard algorithm the
 * latter takes bit = (x < 0) first (cyclic.h:163-173) and erasures must be NULL; words may be NULL. */
int cc_decode_hard_batch(const cc_code *code, const uint8_t *in /* B*n */, const uint16_t *erasures,
                         const uint32_t *erasure_offsets, uint8_t *msg /* B*l */, uint8_t *words /* B*n or NULL */,
                         int32_t *nerr, int32_t *status, size_t B);
int cc_decode_soft_batch(const cc_code *code, const float *y /* B*n */, const uint16_t *erasures,
                         const uint32_t *erasure_offsets, uint8_t *msg /* B*l */, uint8_t *words /* B*n or NULL */,
                         uint16_t *iters, int32_t *status, size_t B);

/* ---- batched AWGN Monte-Carlo (replaces awgn_simulation::operator(), src/simulation/simulation.c++:95-150).
 *      Frames [first_frame, first_frame + frames) of one Eb/N0 point are generated ON DEVICE (Philox4x32-10
 *      keyed by (seed, global frame index) + Box-Muller, y = (1 - 2c) + sigma*N(0,1),
 *      sigma = 1/sqrt(2*rate*10^(ebno/10)), simulation.c++:83-85), decoded with the code's algorithm and
 *      counted.  random_codewords = 0 transmits the all-zero word as the reference does (:113-125).
 *      d_counters accumulates (atomically) CC_MC_NCOUNTERS uint64 values; reduce them across ranks with
 *      one all-reduce.  Results depend only on (seed, ebno, global frame index), not on the sharding. ---- */
enum {
  CC_MC_FRAMES = 0,
  CC_MC_WORD_ERRORS = 1, /* decoded word != transmitted word, or decoder failure (simulation.c++:128-135) */
  CC_MC_BIT_ERRORS = 2,  /* wrong bits among the n code bits (failed frames count their hard output)      */
  CC_MC_FAILURES = 3,    /* decoder reported failure                                                      */
  CC_MC_UNDETECTED = 4,  /* decoder reported success with a wrong word                                    */
  CC_MC_ITER_SUM = 5,    /* sum of iterations run (min-sum), 0 for algebraic                              */
  CC_MC_CHANNEL_BIT_ERRORS = 6, /* raw hard-decision errors before decoding                               */
  CC_MC_RESERVED = 7,
  CC_MC_ITER_HIST = 8,   /* [8 + i] = frames that returned at iteration index i, i <= 55                  */
  CC_MC_NCOUNTERS = 64
};
int cc_mc_run_dev(const cc_code *code, double ebno_db, uint64_t seed, uint64_t first_frame, size_t frames,
                  int random_codewords, uint64_t *d_counters, void *stream);
/* just the channel: writes y (frames*n floats) and, if not NULL, the transmitted words (frames*n bytes) */
int cc_awgn_llr_dev(const cc_code *code, double ebno_db, uint64_t seed, uint64_t first_frame, size_t frames,
                    int random_codewords, float *d_llr, uint8_t *d_sent, void *stream);
double cc_sigma(const cc_code *code, double ebno_db); /* simulation.c++:83-85 */

/* ---- fields GF(2^q) with q = 9 .. 15 (galois.h:44-53: "uint16_t allows galois fields up to 2^15"): symbols are
 *      16 bits wide, n = 2^q - 1 <= 32767.  Hard-decision algorithms (PGZ as bounded-distance BM, BM, Euklid), with
 *      erasures; division_tag coding.  The byte entry points above return CC_ERR_UNSUPPORTED on such a handle and
 *      these return it on a q <= 8 handle.  Min-sum serves BCH codes up to q = 11 (n <= 2047) through the byte entry points
 *      (bits and LLRs have no symbol width); the Monte-Carlo calls do not apply. ---- */
int cc_encode_batch_u16(const cc_code *code, const uint16_t *msg /* B*l */, uint16_t *cw /* B*n */, size_t B);
int cc_encode_batch_u16_dev(const cc_code *code, const uint16_t *d_msg, uint16_t *d_cw, size_t B, void *stream);
int cc_correct_hard_batch_u16(const cc_code *code, const uint16_t *in /* B*n symbols */, const uint16_t *erasures,
                              const uint32_t *erasure_offsets, uint16_t *out /* B*n */, int32_t *nerr, int32_t *status,
                              size_t B);
int cc_correct_hard_batch_u16_dev(const cc_code *code, const uint16_t *d_in, const uint16_t *d_erasures,
                                  const uint32_t *d_erasure_offsets, uint16_t *d_out, int32_t *d_nerr,
                                  int32_t *d_status, size_t B, void *stream);
int cc_extract_batch_u16(const cc_code *code, const uint16_t *cw /* B*n */, uint16_t *msg /* B*l */, size_t B);
int cc_extract_batch_u16_dev(const cc_code *code, const uint16_t *d_cw, uint16_t *d_msg, size_t B, void *stream);
/* cc_get_poly for 16-bit coefficients (works on every handle) */
int cc_get_poly_u16(const cc_code *code, int which, uint16_t *out, size_t cap);
uint32_t cc_q(const cc_code *code);

/* ---- introspection for the benchmark: name and launch geometry of the kernel a call would use ---- */
int cc_kernel_info(const cc_code *code, char *name, size_t cap, uint32_t *frames_per_workgroup,
                   uint32_t *threads_per_workgroup, uint32_t *lds_bytes);

/* The deal of the diagonal min-sum kernel for this code (host logic, works on CC_DEVICE_NONE handles): D slots x
 * LPF lanes of row-0 support positions, slot-major (0xFFFF = empty slot of a "partial" geometry); *links = number of
 * chained slot pairs (slots 2p and 2p + 1 hold diagonals s and s + 1 in every lane).  Returns the number of entries
 * written, 0 if the code has no diagonal geometry, < 0 on error. */
int cc_diag_table(const cc_code *code, uint16_t *out, size_t cap, uint32_t *D, uint32_t *LPF, uint32_t *links);

#ifdef __cplusplus
}
#endif
#endif /* CHANNELCODING_AMD_H */
