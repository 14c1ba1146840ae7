/* oracle/cc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * See cc_oracle.h.  Plain C11; compiled with -ffp-contract=off so that every
 * float operation rounds exactly as in the reference's (un-fused) x86-64 build.
 * All file:line citations are relative to /root/reference/.
 */
#include "cc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* GF(2^q)                                                                   */
/* ------------------------------------------------------------------------ */

/* src/math/galois.h:18-20 */
static const unsigned modular_polynomials[9] = {0, 0x3, 0x7, 0xb, 0x13, 0x25, 0x43, 0x83, 0x11d};

/* src/math/galois.h:269-301 (init_tables) */
static void init_tables(orc_code *c) {
  const int size = c->size;
  unsigned polynomial = 1;
  memset(c->exp_, 0, sizeof c->exp_);
  memset(c->log_, 0, sizeof c->log_);
  for (int power = 0; power < size - 1; power++) {
    c->log_[polynomial] = (uint8_t)power;
    c->log_[polynomial + (unsigned)size] = (uint8_t)power;
    c->exp_[power] = (uint8_t)polynomial;
    c->exp_[power + size - 1] = (uint8_t)polynomial;
    int carry = (polynomial & (1u << (c->q - 1))) != 0;
    polynomial = (polynomial << 1) & (unsigned)(size - 1); /* storage_type truncation */
    if (carry)
      polynomial ^= modular_polynomials[c->q] & (unsigned)(size - 1);
  }
  c->log_[0] = 0;
  c->log_[size] = 0;
  c->exp_[size - 1] = 1;
  c->exp_[2 * size - 2] = 1;
}

/* galois.h:194-198 */
static inline uint8_t gmul(const orc_code *c, uint8_t a, uint8_t b) {
  if (a == 0 || b == 0)
    return 0;
  return c->exp_[c->log_[a] + c->log_[b]];
}
/* galois.h:200-209 (b != 0) */
static inline uint8_t gdiv(const orc_code *c, uint8_t a, uint8_t b) {
  if (a == 0)
    return 0;
  return c->exp_[c->log_[a] - c->log_[b] + c->size - 1];
}
static inline uint8_t ginv(const orc_code *c, uint8_t a) { return gdiv(c, 1, a); }
/* galois.h:182-184: reduces the exponent mod 2^q, not 2^q-1 */
static inline uint8_t from_power(const orc_code *c, unsigned p) { return c->exp_[p % (unsigned)c->size]; }

/* ------------------------------------------------------------------------ */
/* dense polynomials (index = power of x), src/math/polynomial.h              */
/* ------------------------------------------------------------------------ */
#define PMAX 1100
typedef struct {
  uint8_t c[PMAX];
  int len;
} poly;

static void p_set(poly *p, const uint8_t *v, int len) {
  memset(p->c, 0, sizeof p->c);
  memcpy(p->c, v, (size_t)len);
  p->len = len;
}
static void p_const(poly *p, uint8_t v) { p_set(p, &v, 1); }
/* polynomial.h:131-135 */
static int p_degree(const poly *p) {
  for (int i = p->len - 1; i >= 0; i--)
    if (p->c[i])
      return i;
  return -1;
}
/* polynomial.h:273-284: Horner over all stored coefficients; 0 at x == 0 */
static uint8_t p_eval(const orc_code *c, const uint8_t *coef, int len, uint8_t x) {
  if (len == 0 || x == 0)
    return 0;
  uint8_t result = coef[len - 1];
  for (int i = len - 2; i >= 0; i--)
    result = gmul(c, result, x) ^ coef[i];
  return result;
}
/* polynomial.h:207-231; result length as the reference computes it */
static void p_mul(const orc_code *c, poly *out, const poly *a, const poly *b) {
  poly r;
  int da = p_degree(a), db = p_degree(b);
  if (da < 0 || db < 0) {
    p_const(out, 0);
    return;
  }
  memset(r.c, 0, sizeof r.c);
  r.len = da + db + 1;
  for (int i = 0; i <= da; i++)
    if (a->c[i])
      for (int j = 0; j <= db; j++)
        r.c[i + j] ^= gmul(c, a->c[i], b->c[j]);
  *out = r;
}
static void p_scale(const orc_code *c, poly *p, uint8_t s) {
  for (int i = 0; i < p->len; i++)
    p->c[i] = gmul(c, p->c[i], s);
}
/* polynomial.h:77-92 (element_wise with plus): grows to the longer operand */
static void p_add(poly *a, const poly *b) {
  if (a->len < b->len)
    a->len = b->len;
  for (int i = 0; i < b->len; i++)
    a->c[i] ^= b->c[i];
}
/* polynomial.h:41-75 (division): quotient and remainder */
static void p_divmod(const orc_code *c, const poly *lhs, const poly *rhs, poly *quot, poly *rem) {
  int dl = p_degree(lhs), dr = p_degree(rhs);
  poly q, r;
  if (dl < dr) {
    p_const(&q, 0);
    r = *lhs;
  } else {
    memset(q.c, 0, sizeof q.c);
    q.len = dl - dr + 1;
    r = *lhs;
    uint8_t lead = rhs->c[dr];
    for (int pos = dl; pos >= dr; pos--) {
      uint8_t coef = gdiv(c, r.c[pos], lead);
      q.c[pos - dr] ^= coef;
      if (coef)
        for (int j = 0; j <= dr; j++)
          r.c[pos - dr + j] ^= gmul(c, rhs->c[j], coef);
    }
  }
  if (quot)
    *quot = q;
  if (rem)
    *rem = r;
}

/* ------------------------------------------------------------------------ */
/* code construction                                                         */
/* ------------------------------------------------------------------------ */

/* cyclic.h:186-204 (consecutive_zeroes), including the over-count when all
 * root exponents are consecutive. */
static int consecutive_zeroes(const orc_code *c) {
  int powers[256], np = 0;
  for (int v = 1; v < c->size; v++)
    if (p_eval(c, c->g, c->glen, (uint8_t)v) == 0)
      powers[np++] = c->log_[v];
  /* sort ascending */
  for (int i = 1; i < np; i++) {
    int x = powers[i], j = i - 1;
    while (j >= 0 && powers[j] > x) {
      powers[j + 1] = powers[j];
      j--;
    }
    powers[j + 1] = x;
  }
  int first = 0;
  while (first < np && powers[first] != 1)
    first++;
  int last = np; /* adjacent_find(first, end, lhs + 1 != rhs) */
  for (int i = first; i + 1 < np; i++)
    if (powers[i] + 1 != powers[i + 1]) {
      last = i;
      break;
    }
  return (last - first) + 1;
}

int orc_code_init(orc_code *c, int family, int q, int t, int mu, int step, int coding) {
  if (q < 2 || q > 8 || t < 1)
    return -1;
  memset(c, 0, sizeof *c);
  c->family = family;
  c->q = q;
  c->t = t;
  c->size = 1 << q;
  c->n = c->size - 1;
  c->mu = mu;
  c->step = step;
  c->coding = coding;
  if (2 * t > 254) /* roots[256], S[256]: every t a GF(2^8) code can have */
    return -4;
  if (2 * t >= c->n)
    return -1;
  init_tables(c);

  poly g;
  p_const(&g, 1);
  if (family == ORC_BCH) {
    /* bch.h:28-46 g = lcm of the minimal polynomials of alpha^1, alpha^3, ...;
     * bch.h:62-78 cyclotomic cosets.  Minimal polynomials are either equal
     * or coprime, so the lcm is the product over the union of the cosets. */
    uint8_t seen[256];
    memset(seen, 0, sizeof seen);
    for (unsigned p = 1; p < 2u * (unsigned)t; p += 2) {
      unsigned r = p % (unsigned)c->n;
      if (seen[r])
        continue;
      unsigned cur = r;
      do {
        seen[cur] = 1;
        uint8_t f[2] = {from_power(c, cur), 1};
        poly fac;
        p_set(&fac, f, 2);
        p_mul(c, &g, &g, &fac);
        cur = (cur * 2) % (unsigned)c->n;
      } while (cur != r);
    }
    /* bch.h:48-55 */
    c->nroots = 2 * t;
    for (int i = 0; i < c->nroots; i++)
      c->roots[i] = from_power(c, (unsigned)(i + 1));
  } else if (family == ORC_RS) {
    /* rs.h:18-28, rs.h:30-39 */
    c->nroots = 2 * t;
    for (int i = 0; i < 2 * t; i++) {
      uint8_t root = from_power(c, (unsigned)(mu + i * step));
      uint8_t f[2] = {root, 1};
      poly fac;
      p_set(&fac, f, 2);
      p_mul(c, &g, &g, &fac);
      c->roots[i] = root;
    }
  } else
    return -1;
  c->glen = p_degree(&g) + 1;
  if (c->glen > 256)
    return -1;
  memcpy(c->g, g.c, (size_t)c->glen);
  c->k = c->glen - 1;
  c->l = c->n - c->k;
  if (c->l < 1)
    return -1;

  /* cyclic.h:120-123, :272: h = (x^n + 1) / g */
  poly f, h;
  memset(f.c, 0, sizeof f.c);
  f.len = c->n + 1;
  f.c[0] = 1;
  f.c[c->n] = 1;
  p_divmod(c, &f, &g, &h, NULL);
  c->hlen = h.len;
  memcpy(c->h, h.c, (size_t)h.len);
  c->dmin = consecutive_zeroes(c) + 1;
  if (c->dmin > c->n)
    return -1; /* cyclic.h:276-279 */
  return 0;
}

size_t orc_code_sizeof(void) { return sizeof(orc_code); }

/* cyclic.h:346-359: row 0 = h reversed, zero padded; row i = row 0 rotated
 * right by i. */
void orc_get_H(const orc_code *c, uint8_t *H) {
  uint8_t row[256];
  memset(row, 0, sizeof row);
  for (int j = 0; j < c->hlen; j++)
    row[j] = c->h[c->hlen - 1 - j];
  for (int i = 0; i < c->k; i++)
    for (int j = 0; j < c->n; j++)
      H[i * c->n + j] = row[((j - i) % c->n + c->n) % c->n];
}

int orc_to_string(const orc_code *c, const char *alg_name, char *out, size_t cap) {
  return snprintf(out, cap, "(%d, %d, %d)-%s", c->n, c->l, c->dmin, alg_name);
}

/* cyclic.h:289-311 + free encode cyclic.h:29-40 */
int orc_encode(const orc_code *c, const uint8_t *msg, uint8_t *cw) {
  poly a, g, enc;
  for (int i = 0; i < c->l; i++)
    if (msg[i] & ~(c->size - 1))
      return -2; /* Element ctor: "Value is not an element of the field." galois.h:149-152 */
  p_set(&a, msg, c->l);
  p_set(&g, c->g, c->glen);
  if (c->coding == ORC_CODING_MULTIPLICATION) {
    p_mul(c, &enc, &a, &g);
  } else {
    /* a * x^k ; x_k + (x_k % g) */
    poly xk, shift, rem;
    memset(shift.c, 0, sizeof shift.c);
    shift.len = c->k + 1;
    shift.c[c->k] = 1;
    p_mul(c, &xk, &a, &shift);
    p_divmod(c, &xk, &g, NULL, &rem);
    enc = xk;
    p_add(&enc, &rem);
  }
  memset(cw, 0, (size_t)c->n);
  memcpy(cw, enc.c, (size_t)(enc.len < c->n ? enc.len : c->n));
  return 0;
}

/* cyclic.h:42-51 + :313-327 */
void orc_extract(const orc_code *c, const uint8_t *cw, uint8_t *msg) {
  memset(msg, 0, (size_t)c->l);
  if (c->coding == ORC_CODING_MULTIPLICATION) {
    poly b, g, qo;
    p_set(&b, cw, c->n);
    p_set(&g, c->g, c->glen);
    p_divmod(c, &b, &g, &qo, NULL);
    memcpy(msg, qo.c, (size_t)(qo.len < c->l ? qo.len : c->l));
  } else {
    memcpy(msg, cw + c->k, (size_t)c->l); /* b / x^k */
  }
}

/* cyclic.h:53-63 */
void orc_syndromes(const orc_code *c, const uint8_t *b, uint8_t *S) {
  for (int j = 0; j < c->nroots; j++)
    S[j] = p_eval(c, b, c->n, c->roots[j]);
}

/* ------------------------------------------------------------------------ */
/* error-locator polynomials, src/codes/hard_decision.h                      */
/* ------------------------------------------------------------------------ */

/* polynomial.h:176-179: reverse the first degree()+1 coefficients */
static void p_reverse(poly *p) {
  int d = p_degree(p);
  for (int i = 0, j = d; i < j; i++, j--) {
    uint8_t tmp = p->c[i];
    p->c[i] = p->c[j];
    p->c[j] = tmp;
  }
}

/* hard_decision.h:116-155.  lambda entries beyond its stored size are taken as
 * zero (the reference reads out of bounds there: SURVEY F3) and *ref_ub is
 * raised so that callers know the reference's own answer is undefined. */
static int locator_bm(const orc_code *c, const uint8_t *S, int nS, const uint16_t *er, int ne, poly *out,
                      int *ref_ub) {
  const int fk = nS / 2;
  const int rho = ne;
  poly lambda, b, xfac;
  p_const(&lambda, 1);
  int l = ne;
  for (int e = 0; e < ne; e++) {
    uint8_t f[2] = {1, from_power(c, er[e])};
    poly fac;
    p_set(&fac, f, 2);
    p_mul(c, &lambda, &lambda, &fac);
  }
  b = lambda;
  {
    uint8_t f[2] = {0, 1};
    p_set(&xfac, f, 2);
  }
  for (int i = rho; i < 2 * fk; i++) {
    p_mul(c, &b, &b, &xfac);
    if (l + 1 > lambda.len)
      *ref_ub = 1;
    uint8_t delta = S[i];
    for (int j = 1; j <= l; j++) {
      uint8_t lj = j < lambda.len ? lambda.c[j] : 0;
      if (i - j >= 0)
        delta ^= gmul(c, lj, S[i - j]);
      else
        *ref_ub = 1;
    }
    if (delta) {
      poly t = b;
      p_scale(c, &t, delta);
      poly sum = lambda;
      p_add(&sum, &t);
      if (2 * l <= i + rho) {
        b = lambda;
        p_scale(c, &b, ginv(c, delta));
        l = i + rho - l + 1;
      }
      lambda = sum;
    }
  }
  p_reverse(&lambda);
  *out = lambda;
  return ORC_FRAME_OK;
}

/* hard_decision.h:157-196 */
static int locator_euklid(const orc_code *c, const uint8_t *S, int nS, const uint16_t *er, int ne, poly *out) {
  const int fk = nS / 2;
  const int max = (2 * fk + ne) / 2;
  poly u, s, r_prev, r_cur, w_prev, w_cur;
  p_const(&u, 1);
  for (int e = 0; e < ne; e++) {
    uint8_t f[2] = {1, from_power(c, er[e])};
    poly fac;
    p_set(&fac, f, 2);
    p_mul(c, &u, &u, &fac);
  }
  p_set(&s, S, nS);
  p_mul(c, &r_prev, &s, &u);
  memset(r_cur.c, 0, sizeof r_cur.c);
  r_cur.len = 2 * fk + 1;
  r_cur.c[2 * fk] = 1;
  w_prev = u;
  p_const(&w_cur, 0);
  while (p_degree(&r_cur) >= max) {
    poly q, next, qw, w_next;
    p_divmod(c, &r_prev, &r_cur, &q, &next);
    p_mul(c, &qw, &q, &w_cur);
    w_next = w_prev;
    p_add(&w_next, &qw);
    r_prev = r_cur;
    r_cur = next;
    w_prev = w_cur;
    w_cur = w_next;
  }
  if (w_cur.c[0] == 0)
    return ORC_FRAME_LOCATOR; /* "Cannot invert last element" */
  p_scale(c, &w_cur, ginv(c, w_cur.c[0]));
  p_reverse(&w_cur);
  *out = w_cur;
  return ORC_FRAME_OK;
}

/* Gauss-Jordan over GF(2^q) on an m x (m+1) augmented matrix; returns 0 when
 * the system is regular.  (Mathematical restatement of
 * linear_equation_system.h:12-49,67-88; the reference's pivoting quirk Q9 is
 * NOT reproduced -- see DESIGN.md "reference defects".) */
static int gauss_solve(const orc_code *c, uint8_t A[][257], int m, uint8_t *x) {
  for (int col = 0; col < m; col++) {
    int piv = -1;
    for (int r = col; r < m; r++)
      if (A[r][col]) {
        piv = r;
        break;
      }
    if (piv < 0)
      return -1;
    if (piv != col)
      for (int j = 0; j <= m; j++) {
        uint8_t tmp = A[piv][j];
        A[piv][j] = A[col][j];
        A[col][j] = tmp;
      }
    uint8_t inv = ginv(c, A[col][col]);
    for (int j = 0; j <= m; j++)
      A[col][j] = gmul(c, A[col][j], inv);
    for (int r = 0; r < m; r++)
      if (r != col && A[r][col]) {
        uint8_t f = A[r][col];
        for (int j = 0; j <= m; j++)
          A[r][j] ^= gmul(c, A[col][j], f);
      }
  }
  for (int i = 0; i < m; i++)
    x[i] = A[i][m];
  return 0;
}

/* hard_decision.h:61-114: try v = t..1, solve the v x v Hankel system
 *   S_{i+v} = sum_j sigma_j S_{i+j},  i = 0..v-1
 * and return sigma = [sigma_0..sigma_{v-1}, 1] (monic, roots = locators). */
static int locator_pgz(const orc_code *c, const uint8_t *S, int nS, poly *out) {
  static uint8_t A[256][257];
  for (int v = nS / 2; v >= 1; v--) {
    for (int i = 0; i < v; i++) {
      for (int j = 0; j < v; j++)
        A[i][j] = S[i + j];
      A[i][v] = S[i + v];
    }
    uint8_t sol[256];
    if (gauss_solve(c, A, v, sol) == 0) {
      memset(out->c, 0, sizeof out->c);
      memcpy(out->c, sol, (size_t)v);
      out->c[v] = 1;
      out->len = v + 1;
      return ORC_FRAME_OK;
    }
  }
  /* hard_decision.h:99-113 fallback: sigma = S_0 if S_{i+1}/S_i is constant */
  uint8_t sigma = S[0];
  for (int i = 0; i + 1 < nS; i++) {
    if (S[i] == 0)
      return ORC_FRAME_LOCATOR;
    if (gmul(c, S[i + 1], ginv(c, S[i])) != sigma)
      return ORC_FRAME_LOCATOR;
  }
  out->len = 2;
  memset(out->c, 0, sizeof out->c);
  out->c[0] = sigma;
  out->c[1] = 1;
  return ORC_FRAME_OK;
}

int orc_locator(const orc_code *c, int alg, const uint8_t *S, const uint16_t *erasures, int nerasures,
                uint8_t *sigma, int *nsigma, int *ref_ub) {
  poly out;
  int ub = 0, st;
  p_const(&out, 0);
  switch (alg) {
  case ORC_ALG_BM:
    st = locator_bm(c, S, c->nroots, erasures, nerasures, &out, &ub);
    break;
  case ORC_ALG_EUKLID:
    st = locator_euklid(c, S, c->nroots, erasures, nerasures, &out);
    break;
  case ORC_ALG_PGZ:
    if (nerasures > 0)
      return -3; /* runtime_error, hard_decision.h:66-68 */
    st = locator_pgz(c, S, c->nroots, &out);
    break;
  default:
    return -1;
  }
  if (ref_ub)
    *ref_ub = ub;
  if (st == ORC_FRAME_OK) {
    *nsigma = out.len;
    memcpy(sigma, out.c, (size_t)out.len);
  } else
    *nsigma = 0;
  return st;
}

/* ------------------------------------------------------------------------ */
/* cyclic::correct_ (hard decision), cyclic.h:207-252                        */
/* ------------------------------------------------------------------------ */
static int correct_core(const orc_code *c, int alg, const uint8_t *in, const uint16_t *er, int ne, uint8_t *out,
                        int *nerr, int *ref_ub) {
  uint8_t S[256], sigma[PMAX];
  int nsigma = 0;
  memcpy(out, in, (size_t)c->n);
  if (nerr)
    *nerr = -1;
  orc_syndromes(c, in, S);
  int any = 0;
  for (int j = 0; j < c->nroots; j++)
    any |= S[j];
  if (!any) {
    if (nerr)
      *nerr = 0;
    return ORC_FRAME_OK;
  }
  int st = orc_locator(c, alg, S, er, ne, sigma, &nsigma, ref_ub);
  if (st != ORC_FRAME_OK)
    return st;
  /* cyclic.h:126-150 zeroes(): brute force over the non-zero elements
   * (polynomial.h:16-28), sorted by power (galois.h:240-249) */
  int deg = -1;
  for (int i = nsigma - 1; i >= 0; i--)
    if (sigma[i]) {
      deg = i;
      break;
    }
  int positions[256], nz = 0;
  for (int p = 0; p < c->n; p++) { /* ascending log == sorted order */
    uint8_t x = c->exp_[p];
    if (p_eval(c, sigma, nsigma, x) == 0)
      positions[nz++] = p; /* cyclic.h:152-159: position = log(zero) */
  }
  if (nz != deg)
    return ORC_FRAME_LOCATOR; /* cyclic.h:134-143 */
  if (nz == 0)
    return ORC_FRAME_LOCATOR; /* cyclic.h:145-147 */
  uint8_t values[256];
  if (c->family == ORC_BCH) {
    for (int i = 0; i < nz; i++)
      values[i] = 1; /* bch.h:80-83 */
  } else {
    /* rs.h:41-78: S_i = sum_k y_k X_k^(i+1), i = 0..v-1 */
    static uint8_t A[256][257];
    if (nz > 256 || nz > c->nroots)
      return ORC_FRAME_LOCATOR; /* syndromes.at(i) would throw */
    for (int i = 0; i < nz; i++) {
      for (int kx = 0; kx < nz; kx++) {
        uint8_t X = c->exp_[positions[kx]], pw = X;
        for (int e = 0; e < i; e++)
          pw = gmul(c, pw, X);
        A[i][kx] = pw;
      }
      A[i][nz] = S[i];
    }
    if (gauss_solve(c, A, nz, values) != 0)
      return ORC_FRAME_LOCATOR;
  }
  uint8_t tmp[256];
  memcpy(tmp, in, (size_t)c->n);
  for (int i = 0; i < nz; i++)
    tmp[positions[i]] ^= values[i]; /* cyclic.h:237-241 */
  orc_syndromes(c, tmp, S);
  any = 0;
  for (int j = 0; j < c->nroots; j++)
    any |= S[j];
  if (any)
    return ORC_FRAME_RECHECK; /* cyclic.h:243-248 */
  memcpy(out, tmp, (size_t)c->n);
  if (nerr)
    *nerr = nz;
  return ORC_FRAME_OK;
}

int orc_correct_hard(const orc_code *c, int alg, const uint8_t *in, const uint16_t *erasures, int nerasures,
                     uint8_t *out, int *nerr, int *ref_ub) {
  if (ref_ub)
    *ref_ub = 0;
  for (int i = 0; i < c->n; i++)
    if (in[i] & ~(c->size - 1))
      return -2; /* galois.h:149-152 */
  for (int e = 0; e < nerasures; e++)
    if (erasures[e] >= c->n)
      return -2;
  if (c->family == ORC_BCH && alg == ORC_ALG_PGZ && nerasures > 0) {
    /* bch.h:97-149: decode twice with erasures forced to 0 and to 1, keep
     * the result with fewer corrected errors (first wins ties). */
    if (nerasures > 2 * c->t) {
      memcpy(out, in, (size_t)c->n);
      if (nerr)
        *nerr = -1;
      return ORC_FRAME_ERASURES;
    }
    uint8_t tmp[256], o0[256], o1[256];
    int e0 = -1, e1 = -1;
    memcpy(tmp, in, (size_t)c->n);
    for (int e = 0; e < nerasures; e++)
      tmp[erasures[e]] = 0;
    int s0 = correct_core(c, alg, tmp, NULL, 0, o0, &e0, ref_ub);
    for (int e = 0; e < nerasures; e++)
      tmp[erasures[e]] = 1;
    int s1 = correct_core(c, alg, tmp, NULL, 0, o1, &e1, ref_ub);
    if (s0 != ORC_FRAME_OK && s1 != ORC_FRAME_OK) {
      memcpy(out, in, (size_t)c->n);
      if (nerr)
        *nerr = -1;
      return ORC_FRAME_LOCATOR;
    }
    int pick1 = (s0 != ORC_FRAME_OK) || (s1 == ORC_FRAME_OK && e1 < e0);
    memcpy(out, pick1 ? o1 : o0, (size_t)c->n);
    if (nerr)
      *nerr = pick1 ? e1 : e0;
    return ORC_FRAME_OK;
  }
  return correct_core(c, alg, in, erasures, nerasures, out, nerr, ref_ub);
}

int orc_correct_hard_f32(const orc_code *c, int alg, const float *in, const uint16_t *erasures, int nerasures,
                         uint8_t *out, int *nerr, int *ref_ub) {
  uint8_t bits[256];
  for (int i = 0; i < c->n; i++)
    bits[i] = (uint8_t)(in[i] < 0); /* codes.h:43-52 via cyclic.h:163-173 */
  return orc_correct_hard(c, alg, bits, erasures, nerasures, out, nerr, ref_ub);
}

/* ------------------------------------------------------------------------ */
/* min-sum, src/codes/soft_decision.h                                        */
/* ------------------------------------------------------------------------ */

/* soft_decision.h:75-77 */
static inline int signum_f(float v) { return (0.0f < v) - (v < 0.0f); }
/* std::min / std::max semantics (NaN behaviour included) */
static inline float std_minf(float a, float b) { return (b < a) ? b : a; }
static inline double std_maxd(double a, double b) { return (a < b) ? b : a; }

typedef struct {
  int variant;
  float alpha_f, beta_f;
  double beta_d;
} ms_params;

/* horizontal functor applied to the exclusive minimum, then `sign * fn(min)`
 * converted to R=float (soft_decision.h:118 with :204,:211-213,:245-251) */
static inline float hor_apply(const ms_params *p, int sign, float min) {
  switch (p->variant) {
  case ORC_NMS:
  case ORC_2DNMS:
    return (float)sign * (p->alpha_f * min);
  case ORC_OMS:
    return (float)((double)sign * std_maxd((double)min - p->beta_d, 0.0));
  default:
    return (float)sign * min;
  }
}
/* vertical functor (soft_decision.h:205-209,:215-218,:261-266,:275-280) */
static inline float vert_apply(const ms_params *p, float e, float y, float q_old) {
  switch (p->variant) {
  case ORC_SCMS1: {
    float tmp = e + y;
    if (signum_f(q_old) == 0 || signum_f(q_old) == signum_f(tmp))
      return tmp;
    return 0.0f;
  }
  case ORC_SCMS2: {
    float tmp = e + y;
    if (tmp * q_old > 0)
      return tmp;
    return 0.5f * (tmp + q_old);
  }
  case ORC_2DNMS: {
    float scaled = p->beta_f * e;
    return scaled + y;
  }
  default:
    return e + y;
  }
}

static void ms_params_init(ms_params *p, int variant, double alpha, double beta) {
  p->variant = variant;
  p->alpha_f = (float)alpha; /* const R& alpha: soft_decision.h:211,:233-236 */
  p->beta_f = (float)beta;   /* const R& beta: soft_decision.h:215-218 */
  p->beta_d = beta;          /* OMS evaluates in double: soft_decision.h:245-251 */
}

/* stop test: soft_decision.h:79-84 through matrix::operator* matrix.h:57-67 */
static int stop_test(int rule, const uint8_t *H, int k, int n, const uint8_t *b) {
  if (rule == ORC_STOP_O0)
    return 1; /* H*b is empty (matrix.h:50), none_of(empty) == true */
  for (int i = 0; i < k; i++) {
    if (rule == ORC_STOP_O1) {
      uint8_t acc = 0; /* inner_product in uint8_t */
      for (int j = 0; j < n; j++)
        acc = (uint8_t)(acc + H[i * n + j] * b[j]);
      if (acc)
        return 0;
    } else {
      unsigned acc = 0; /* GF(2) */
      for (int j = 0; j < n; j++)
        acc ^= (unsigned)((H[i * n + j] != 0) & (b[j] != 0));
      if (acc)
        return 0;
    }
  }
  return 1;
}

/* cyclic.h:361-385 */
void orc_get_H_alt(const orc_code *c, uint8_t *H, int *rows) {
  for (int r = 0; r < c->t; r++)
    for (int bit = 0; bit < c->q; bit++)
      for (int col = 0; col < c->n; col++) {
        uint8_t v = from_power(c, (unsigned)(col * (2 * r + 1)));
        H[((size_t)r * (size_t)c->q + (size_t)bit) * (size_t)c->n + (size_t)col] = (uint8_t)((v >> bit) & 1);
      }
  if (rows)
    *rows = c->t * c->q;
}

static int minsum_core(const orc_code *c, const uint8_t *Hin, int k, int n, int variant, unsigned iterations,
                       double alpha, double beta, int stop_rule, const float *yin, const uint16_t *erasures,
                       int nerasures, uint8_t *b, float *L, unsigned *iter);

int orc_minsum(const orc_code *c, int variant, unsigned iterations, double alpha, double beta, int stop_rule,
               const float *yin, const uint16_t *erasures, int nerasures, uint8_t *b, float *L, unsigned *iter) {
  return minsum_core(c, NULL, c->k, c->n, variant, iterations, alpha, beta, stop_rule, yin, erasures, nerasures, b, L, iter);
}

int orc_minsum_H(const uint8_t *H, int rows, int cols, int variant, unsigned iterations, double alpha, double beta,
                 int stop_rule, const float *y, uint8_t *b, float *L, unsigned *iter) {
  return minsum_core(NULL, H, rows, cols, variant, iterations, alpha, beta, stop_rule, y, NULL, 0, b, L, iter);
}

static int minsum_core(const orc_code *c, const uint8_t *Hin, int k, int n, int variant, unsigned iterations,
                       double alpha, double beta, int stop_rule, const float *yin, const uint16_t *erasures,
                       int nerasures, uint8_t *b, float *L, unsigned *iter) {
  ms_params P;
  ms_params_init(&P, variant, alpha, beta);
  uint8_t *H = (uint8_t *)malloc((size_t)k * (size_t)n);
  float *q = (float *)calloc((size_t)k * (size_t)n, sizeof(float));
  float *r = (float *)calloc((size_t)k * (size_t)n, sizeof(float));
  float *cs = (float *)malloc((size_t)n * sizeof(float));
  float *y = (float *)malloc((size_t)n * sizeof(float));
  int status = ORC_FRAME_NOT_CONVERGED;
  if (Hin)
    memcpy(H, Hin, (size_t)k * (size_t)n);
  else
    orc_get_H(c, H); /* rebuilt per call: cyclic.h:265 */
  memcpy(y, yin, (size_t)n * sizeof(float));
  for (int e = 0; e < nerasures; e++)
    y[erasures[e]] = 0.0f; /* cyclic.h:259-262 */
  for (int j = 0; j < n; j++) {
    L[j] = 0.0f;
    b[j] = 0;
  }
  *iter = iterations;
  for (unsigned it = 0; it < iterations; it++) {
    /* vertical__ :125-140 with column_sum :86-98 */
    for (int j = 0; j < n; j++)
      cs[j] = 0.0f;
    for (int i = 0; i < k; i++)
      for (int j = 0; j < n; j++)
        if (H[i * n + j])
          cs[j] += r[i * n + j];
    for (int i = 0; i < k; i++)
      for (int j = 0; j < n; j++)
        if (H[i * n + j]) {
          float e = cs[j] - r[i * n + j];
          q[i * n + j] = vert_apply(&P, e, y[j], q[i * n + j]);
        }
    /* horizontal__ :101-122 */
    for (int i = 0; i < k; i++)
      for (int j = 0; j < n; j++)
        if (H[i * n + j]) {
          int sign = 1;
          float min = FLT_MAX;
          for (int x = 0; x < n; x++)
            if (x != j && H[i * n + x]) {
              sign *= signum_f(q[i * n + x]);
              min = std_minf(min, fabsf(q[i * n + x])); /* std::abs(float) */
            }
          r[i * n + j] = hor_apply(&P, sign, min);
        }
    /* :178-183 */
    for (int j = 0; j < n; j++)
      cs[j] = 0.0f;
    for (int i = 0; i < k; i++)
      for (int j = 0; j < n; j++)
        if (H[i * n + j])
          cs[j] += r[i * n + j];
    for (int j = 0; j < n; j++) {
      L[j] = cs[j] + y[j];
      b[j] = (uint8_t)(L[j] < 0);
    }
    if (stop_test(stop_rule, H, k, n, b)) {
      *iter = it;
      status = ORC_FRAME_OK;
      break;
    }
  }
  free(H);
  free(q);
  free(r);
  free(cs);
  free(y);
  return status;
}

/* O(w) restatement: per check node keep min1, min2 (second smallest counting
 * multiplicity), the number of negative and of zero messages; the exclusive
 * sign / minimum of edge j follow from those and q_j itself.  Bit-identical to
 * orc_minsum for finite inputs (asserted by tests/test_oracle_vs_ref.py::test_minsum_matches_reference and test_oracle_golden.py). */
int orc_minsum_fast(const orc_code *c, int variant, unsigned iterations, double alpha, double beta, int stop_rule,
                    const float *y, uint8_t *b, float *L, unsigned *iter) {
  const int n = c->n, k = c->k;
  ms_params P;
  ms_params_init(&P, variant, alpha, beta);
  int support[256], w = 0;
  for (int j = 0; j < c->hlen; j++)
    if (c->h[c->hlen - 1 - j])
      support[w++] = j;
  float *q = (float *)calloc((size_t)k * (size_t)w, sizeof(float));
  float *r = (float *)calloc((size_t)k * (size_t)w, sizeof(float));
  float cs[256];
  int status = ORC_FRAME_NOT_CONVERGED;
  *iter = iterations;
  for (int j = 0; j < n; j++) {
    L[j] = 0.0f;
    b[j] = 0;
    cs[j] = 0.0f;
  }
  for (unsigned it = 0; it < iterations; it++) {
    float ncs[256];
    for (int j = 0; j < n; j++)
      ncs[j] = 0.0f;
    for (int i = 0; i < k; i++) {
      float min1 = FLT_MAX, min2 = FLT_MAX;
      int neg = 0, zeros = 0;
      float *qi = q + (size_t)i * (size_t)w, *ri = r + (size_t)i * (size_t)w;
      for (int s = 0; s < w; s++) {
        int j = support[s] + i;
        float e = cs[j] - ri[s];
        float v = vert_apply(&P, e, y[j], qi[s]);
        qi[s] = v;
        float a = v < 0 ? -v : v;
        if (a < min1) {
          min2 = min1;
          min1 = a;
        } else if (a < min2)
          min2 = a;
        neg += (v < 0);
        zeros += (signum_f(v) == 0);
      }
      for (int s = 0; s < w; s++) {
        float v = qi[s];
        float a = v < 0 ? -v : v;
        int self_zero = (signum_f(v) == 0);
        int others_zero = zeros - self_zero;
        int sign = others_zero ? 0 : (((neg - (v < 0)) & 1) ? -1 : 1);
        float m = (a == min1) ? min2 : min1;
        ri[s] = hor_apply(&P, sign, m);
      }
    }
    for (int i = 0; i < k; i++)
      for (int s = 0; s < w; s++)
        ncs[support[s] + i] += r[(size_t)i * (size_t)w + (size_t)s];
    int allzero = 1;
    for (int j = 0; j < n; j++) {
      cs[j] = ncs[j];
      L[j] = cs[j] + y[j];
      b[j] = (uint8_t)(L[j] < 0);
      allzero &= !b[j];
    }
    int ok;
    if (stop_rule == ORC_STOP_O0)
      ok = 1;
    else if (stop_rule == ORC_STOP_O1)
      ok = allzero; /* binary H, every column covered, weights < 256 */
    else {
      ok = 1;
      for (int i = 0; i < k && ok; i++) {
        unsigned acc = 0;
        for (int s = 0; s < w; s++)
          acc ^= b[support[s] + i];
        ok = !acc;
      }
    }
    if (ok) {
      *iter = it;
      status = ORC_FRAME_OK;
      break;
    }
  }
  free(q);
  free(r);
  return status;
}
