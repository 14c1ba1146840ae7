/* oracle/cc_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the hot path of hannesweisbach/channelcoding
 * (GF(2^q) tables, cyclic-code construction, systematic encode, the algebraic
 * chain syndromes -> locator (BM / PGZ / Euklid) -> roots -> error values ->
 * re-check, and the six min-sum variants with the three stop rules O0/O1/O2 of
 * SURVEY.md section 8c).  Every function cites the reference file:line it
 * follows (paths relative to /root/reference/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library -- as the checker, never as the thing measured or shipped.
 * Parity status: PINNED -- validated against the real reference built by
 * oracle/Makefile (oracle/_ref/) and against the committed vectors under
 * tests/golden/ (tests/test_oracle_vs_ref.py, tests/test_oracle_golden.py).
 */
#ifndef CC_ORACLE_H
#define CC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_BCH = 0, ORC_RS = 1 };
enum { ORC_ALG_PGZ = 0, ORC_ALG_BM = 1, ORC_ALG_EUKLID = 2 };
enum { ORC_CODING_DIVISION = 0, ORC_CODING_MULTIPLICATION = 1 };
/* min-sum variants: soft_decision.h:220-295 */
enum { ORC_MS = 0, ORC_NMS = 1, ORC_OMS = 2, ORC_SCMS1 = 3, ORC_SCMS2 = 4, ORC_2DNMS = 5 };
/* stop rules: 0 as shipped, 1 published (integer dot product),
 * 2 intended (GF(2) parity) */
enum { ORC_STOP_O0 = 0, ORC_STOP_O1 = 1, ORC_STOP_O2 = 2 };
/* per-frame status, same numbering as include/channelcoding_amd.h */
enum {
  ORC_FRAME_OK = 0,
  ORC_FRAME_NOT_CONVERGED = 1, /* soft_decision.h:201 */
  ORC_FRAME_LOCATOR = 2,       /* cyclic.h:134-147, hard_decision.h:103,109,191 */
  ORC_FRAME_RECHECK = 3,       /* cyclic.h:243-248 */
  ORC_FRAME_ERASURES = 4       /* bch.h:105-107 */
};

typedef struct orc_code {
  int family, q, t, n, k /* parity symbols = deg g */, l /* information symbols */;
  int dmin, mu, step, coding, size /* 2^q */;
  uint8_t exp_[512]; /* galois.h:269-301 (doubled antilog table) */
  uint8_t log_[512];
  uint8_t g[256];
  int glen; /* deg g + 1 */
  uint8_t h[256];
  int hlen;
  uint8_t roots[256]; /* syndrome evaluation points, 2t of them */
  int nroots;
} orc_code;

/* returns 0 on success, <0 for unsupported parameters */
int orc_code_init(orc_code *c, int family, int q, int t, int mu, int step, int coding);
size_t orc_code_sizeof(void);

void orc_get_H(const orc_code *c, uint8_t *H /* k*n row-major */);
/* "(n, l, dmin)-ALG" exactly as cyclic::to_string (cyclic.h:282-287) */
int orc_to_string(const orc_code *c, const char *alg_name, char *out, size_t cap);

int orc_encode(const orc_code *c, const uint8_t *msg /* l */, uint8_t *cw /* n */);
void orc_syndromes(const orc_code *c, const uint8_t *b /* n */, uint8_t *S /* 2t */);

/* Error-locator polynomial as error_locator_polynomial() returns it (already
 * reversed for BM/Euklid).  Returns a frame status; *ref_ub is set when the
 * reference's BM would read lambda out of bounds here, in which
 * case the reference's own result is undefined. */
int orc_locator(const orc_code *c, int alg, const uint8_t *S, const uint16_t *erasures, int nerasures,
                uint8_t *sigma, int *nsigma, int *ref_ub);

/* cyclic::correct_ (hard_decision_tag), cyclic.h:207-252.  out = corrected
 * word on success, = hard-decided input on failure.  *nerr = number of
 * corrected positions (or -1 on failure). */
int orc_correct_hard(const orc_code *c, int alg, const uint8_t *in, const uint16_t *erasures, int nerasures,
                     uint8_t *out, int *nerr, int *ref_ub);
/* signed (soft) input to a hard algorithm: bit = (x < 0), codes.h:43-52 */
int orc_correct_hard_f32(const orc_code *c, int alg, const float *in, const uint16_t *erasures, int nerasures,
                         uint8_t *out, int *nerr, int *ref_ub);
/* message extraction after correction: cyclic.h:313-327 */
void orc_extract(const orc_code *c, const uint8_t *cw /* n */, uint8_t *msg /* l */);

/* min_sum__ (soft_decision.h:161-202) with the variant functors of
 * soft_decision.h:204-295 and cyclic::correct_(soft) erasure zeroing
 * (cyclic.h:259-262).  alpha/beta are the tag's double constants.
 * Outputs are written on failure too (last iteration's b/L); *iter is the
 * 0-based index of the returning iteration (= iterations on failure). */
int orc_minsum(const orc_code *c, int variant, unsigned iterations, double alpha, double beta, int stop_rule,
               const float *y, const uint16_t *erasures, int nerasures, uint8_t *b, float *L, unsigned *iter);

/* cyclic::H_alt<T>() cyclic.h:361-385 (t*q rows x n), including from_power's exponent reduction mod 2^q */
void orc_get_H_alt(const orc_code *c, uint8_t *H, int *rows);
/* min_sum__ over a caller-supplied rows x n matrix (what min_sum<float,U>(matrix, y, tag) does) */
int orc_minsum_H(const uint8_t *H, int rows, int cols, int variant, unsigned iterations, double alpha, double beta,
                 int stop_rule, const float *y, uint8_t *b, float *L, unsigned *iter);

/* Same algorithm, O(w) check-node update and no per-frame allocation; must be
 * bit-identical to orc_minsum (used to cross-check and as an optimised CPU
 * timing point). */
int orc_minsum_fast(const orc_code *c, int variant, unsigned iterations, double alpha, double beta, int stop_rule,
                    const float *y, uint8_t *b, float *L, unsigned *iter);

#ifdef __cplusplus
}
#endif
#endif
