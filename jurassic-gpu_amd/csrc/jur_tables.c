/* jur_tables.c -- host side of the emissivity/source-function tables.
 *
 * Builds, per (gas, channel), a ragged p -> T -> (u, eps) hierarchy from the
 * reference's ASCII table rows and flattens it to the CSR-like arrays the
 * kernels read.  Row acceptance reproduces the behaviour of the reference
 * loader (src/jurassic.c:346-395): a new pressure / temperature block starts
 * when the value changes; a row extends a curve only if both eps and u grow
 * (or it opens the curve), otherwise it overwrites the curve's last entry;
 * rows beyond TBLNU entries are dropped.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <unistd.h>
#include "jur_internal.h"

/* ---- error text ----------------------------------------------------------- */
static __thread char g_err[512];

void jur_set_error(char const *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

char const *jur_last_error(void) { return g_err; }

void jur_abi_sizes(size_t out[5]) {
  out[0] = sizeof(ctl_t); out[1] = sizeof(atm_t); out[2] = sizeof(obs_t);
  out[3] = JUR_ND; out[4] = JUR_NG;
}

/* the physical constants this build computes with (include/jurassic_abi.h), in the order
 * C1, C2, P0, RE, N_A, k_B, R: tests hold them against literals of their own */
void jur_abi_constants(double out[7]) {
  out[0] = JUR_C1; out[1] = JUR_C2; out[2] = JUR_P0; out[3] = JUR_RE;
  out[4] = JUR_AVOGADRO; out[5] = JUR_BOLTZMANN; out[6] = JUR_MOLAR_GAS;
}

/* ---- continuum coefficient data ------------------------------------------- */
#ifndef CTM_BLOB_PATH
#error "compile with -DCTM_BLOB_PATH"
#endif
__asm__(".section .rodata\n"
        ".balign 16\n"
        ".global jur_ctm_blob\n"
        ".hidden jur_ctm_blob\n"
        "jur_ctm_blob:\n"
        ".incbin \"" CTM_BLOB_PATH "\"\n"
        ".previous\n");

enum { B_CO2_296 = 0, B_CO2_260 = 2001, B_CO2_230 = 4002, B_H2O_296 = 6003, B_H2O_260 = 8004,
       B_H2O_FRN = 10005, B_N2_B = 12006, B_N2_BETA = 12104, B_O2_B = 12202, B_O2_BETA = 12292 };

static double grid_lerp(int base, int iw, double ew, double dw) {
  return ew * jur_ctm_blob[base + iw - 1] + dw * jur_ctm_blob[base + iw];
}

/* Channel-only part of the four continua.  The expressions keep the operand
 * order of jr_common.h:315-390 so the products formed here are the same
 * doubles the reference forms per call. */
int jur_chan_setup(jur_chan_t *ch, double nu, int window) {
  memset(ch, 0, sizeof *ch);
  ch->nu = nu;
  ch->window = window;
  if (!(nu < 0 || nu >= 4000)) {                        /* CO2, 2 cm^-1 grid */
    double const xw = nu * 0.5 + 1;
    int const iw = (int)xw;
    double const dw = xw - iw, ew = 1 - dw;
    ch->co2_on = 1;
    ch->co2_cw296 = grid_lerp(B_CO2_296, iw, ew, dw);
    ch->co2_cw260 = grid_lerp(B_CO2_260, iw, ew, dw);
    ch->co2_cw230 = grid_lerp(B_CO2_230, iw, ew, dw);
  }
  if (!(nu < 0 || nu >= 20000)) {                       /* H2O, 10 cm^-1 grid */
    double const xw = nu / 10 + 1;
    int const iw = (int)xw;
    double const dw = xw - iw, ew = 1 - dw;
    double const cw296 = grid_lerp(B_H2O_296, iw, ew, dw);
    double const cw260 = grid_lerp(B_H2O_260, iw, ew, dw);
    double const cwfrn = grid_lerp(B_H2O_FRN, iw, ew, dw);
    double sfac = 1.;
    if ((nu > 820.) && (nu < 960.)) {                   /* single-precision correction, jr_common.h:345-351 */
      static char const xfcrev[16] = {3, 9, 15, 23, 29, 33, 37, 39, 40, 46, 36, 27, 10, 2, 0, 0};
      float const xx = nu * 0.1 - 82;
      int const ix = (int)xx;
      float const dx = xx - ix;
      sfac += .001 * ((1 - dx) * xfcrev[ix] + dx * xfcrev[ix + 1]);
    }
    double const vf1 = nu - 370.;
    double const vf2 = vf1 * vf1;
    double const vf6 = vf2 * vf2 * vf2;
    double const fscal = 36100. / (vf2 + vf6 * 1e-8 + 36100.) * -.25 + 1.;
    ch->h2o_on = 1;
    ch->h2o_sc = sfac * cw296;
    ch->h2o_ratio = cw260 / cw296;
    /* the kernel forms ratio^y as exp(y ln ratio): ln(ratio) to 64 bits, split into two doubles.  The
     * shipped coefficients are positive everywhere (ratio 1.21 ... 3.67), so the check never fires. */
    if (!(isnormal(ch->h2o_ratio) && ch->h2o_ratio > 0)) {
      jur_set_error("H2O continuum: self-broadening ratio %g at %g cm^-1 is not positive", ch->h2o_ratio, nu);
      return JUR_EINVAL;
    }
    long double const l = logl((long double)ch->h2o_ratio);
    ch->h2o_lnr_hi = (double)l;
    ch->h2o_lnr_lo = (double)(l - (long double)ch->h2o_lnr_hi);
    ch->h2o_ctwfrn = cwfrn * fscal;
  }
  if (!(nu < 2120 || nu > 2605)) {                      /* N2, 5 cm^-1 grid, 98 entries */
    double const xnu = nu * 0.2 - 424;
    int const idx = (int)xnu;
    int const idx1 = (idx + 1 < 98) ? idx + 1 : 97;     /* upstream reads one past the end at weight 0 */
    double const a1 = xnu - idx, a0 = 1 - a1;
    ch->n2_on = 1;
    ch->n2_b = a0 * jur_ctm_blob[B_N2_B + idx] + a1 * jur_ctm_blob[B_N2_B + idx1];
    ch->n2_beta = a0 * jur_ctm_blob[B_N2_BETA + idx] + a1 * jur_ctm_blob[B_N2_BETA + idx1];
  }
  if (!(nu < 1360 || nu > 1805)) {                      /* O2, 5 cm^-1 grid, 90 entries */
    double const xnu = nu * 0.2 - 272;
    int const idx = (int)xnu;
    int const idx1 = (idx + 1 < 90) ? idx + 1 : 89;
    double const a1 = xnu - idx, a0 = 1 - a1;
    ch->o2_on = 1;
    ch->o2_b = a0 * jur_ctm_blob[B_O2_B + idx] + a1 * jur_ctm_blob[B_O2_B + idx1];
    ch->o2_beta = a0 * jur_ctm_blob[B_O2_BETA + idx] + a1 * jur_ctm_blob[B_O2_BETA + idx1];
  }
  return JUR_OK;
}

/* ---- table builder --------------------------------------------------------- */
jur_tables_t *jur_tables_new(int ng, int nd) {
  if (ng < 0 || nd < 1) { jur_set_error("jur_tables_new: bad dimensions ng=%d nd=%d", ng, nd); return NULL; }
  jur_tables_t *tb = (jur_tables_t *)calloc(1, sizeof *tb);
  if (!tb) return NULL;
  tb->ng = ng;
  tb->nd = nd;
  tb->pair = (jur_pair_t *)calloc((size_t)(ng > 0 ? ng : 1) * nd, sizeof(jur_pair_t));
  tb->sr = (double *)calloc((size_t)nd * JUR_TBLNS, sizeof(double));
  tb->have_sr = (char *)calloc(nd, 1);
  if (!tb->pair || !tb->sr || !tb->have_sr) { jur_tables_free(tb); return NULL; }
  return tb;
}

static void pair_clear(jur_pair_t *pr) {
  for (int ip = 0; ip < pr->np; ip++)
    for (int it = 0; it < JUR_TBLNT; it++) { free(pr->lv[ip].cv[it].u); free(pr->lv[ip].cv[it].eps); }
  free(pr->lv);
  pr->lv = NULL;
  pr->np = 0;
}

void jur_tables_free(jur_tables_t *tb) {
  if (!tb) return;
  if (tb->pair)
    for (long i = 0; i < (long)tb->ng * tb->nd; i++) pair_clear(&tb->pair[i]);
  free(tb->pair); free(tb->sr); free(tb->have_sr); free(tb);
}

typedef struct {
  double press_old, temp_old, eps_old, u_old;
  int ip, it, iu, cap_lv;
} feeder_t;

static int feed_one(long *ignored_rows, jur_pair_t *pr, feeder_t *f, double press, double temp, double u, double eps) {
  if (press != f->press_old) {
    f->press_old = press;
    if (++f->ip >= JUR_TBLNP) { jur_set_error("too many pressure levels (max %d)", JUR_TBLNP); return JUR_EINVAL; }
    if (f->ip >= f->cap_lv) {
      int const cap = f->cap_lv ? 2 * f->cap_lv : 8;
      jur_level_t *lv = (jur_level_t *)realloc(pr->lv, sizeof(jur_level_t) * cap);
      if (!lv) return JUR_ENOMEM;
      memset(lv + f->cap_lv, 0, sizeof(jur_level_t) * (cap - f->cap_lv));
      pr->lv = lv;
      f->cap_lv = cap;
    }
    pr->np = f->ip + 1;
    f->it = -1;
  }
  jur_level_t *lv = &pr->lv[f->ip];
  if (temp != f->temp_old) {
    f->temp_old = temp;
    if (++f->it >= JUR_TBLNT) { jur_set_error("too many temperatures (max %d)", JUR_TBLNT); return JUR_EINVAL; }
    lv->nt = f->it + 1;
    f->iu = -1;
  }
  if (f->it < 0) {
    /* upstream would index [-1] here: a pressure block whose first temperature repeats the
     * previous block's last one */
    jur_set_error("table block at p=%g repeats the previous temperature %g", press, temp);
    return JUR_EINVAL;
  }
  jur_curve_t *cv = &lv->cv[f->it];
  if ((eps > f->eps_old && u > f->u_old) || f->iu < 0) {
    f->eps_old = eps;
    f->u_old = u;
    if (++f->iu >= JUR_TBLNU) {
      (*ignored_rows)++;
      f->iu--;
      return JUR_OK;
    }
  }
  if (f->iu >= cv->cap) {
    int const cap = cv->cap ? 2 * cv->cap : 64;
    float *nu_ = (float *)realloc(cv->u, sizeof(float) * cap);
    float *ne = (float *)realloc(cv->eps, sizeof(float) * cap);
    if (nu_) cv->u = nu_;
    if (ne) cv->eps = ne;
    if (!nu_ || !ne) return JUR_ENOMEM;
    cv->cap = cap;
  }
  lv->p = press;
  cv->t = temp;
  cv->u[f->iu] = (float)u;
  cv->eps[f->iu] = (float)eps;
  cv->nu = f->iu + 1;
  return JUR_OK;
}

static void feeder_init(feeder_t *f) {
  f->press_old = f->temp_old = f->eps_old = f->u_old = -999;
  f->ip = f->it = f->iu = -1;
  f->cap_lv = 0;
}

int jur_tables_feed_rows(jur_tables_t *tb, int ig, int id, long nrows, double const *p, double const *t,
                         double const *u, double const *eps) {
  if (!tb || ig < 0 || ig >= tb->ng || id < 0 || id >= tb->nd) { jur_set_error("feed_rows: index out of range"); return JUR_EINVAL; }
  jur_pair_t *pr = &tb->pair[(size_t)ig * tb->nd + id];
  pair_clear(pr);
  feeder_t f;
  feeder_init(&f);
  for (long i = 0; i < nrows; i++) {
    int const rc = feed_one(&tb->ignored_rows, pr, &f, p[i], t[i], u[i], eps[i]);
    if (rc) return rc;
  }
  return JUR_OK;
}

/* One decimal number from *pp, as strtod reads it.  Numbers of up to 15 significant digits whose decimal
 * exponent stays within +-22 -- everything a table written with %g or %.9g holds -- are formed as
 * (integer < 2^53) * or / (exact power of ten): one correctly rounded operation, the same double strtod
 * returns (Clinger's fast path).  Anything else (more digits, large exponents, inf/nan, hex floats, a token that
 * does not end at white space) goes to strtod.  glibc's strtod does not scale over threads here (8 threads:
 * the throughput of one), and it is the whole cost of reading a table file.  Returns 1 and advances *pp on
 * success, 0 if no number starts there. */
int jur_parse_number(char const **pp, double *out) {
  static double const p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  char const *p = *pp;
  while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\v' || *p == '\f') p++;
  char const *const start = p;
  int neg = 0;
  if (*p == '-') { neg = 1; p++; } else if (*p == '+') p++;
  unsigned long long m = 0;
  int sig = 0, dexp = 0, any = 0, simple = 1;
  for (; *p >= '0' && *p <= '9'; p++) {
    any = 1;
    if (sig < 15) { m = m * 10 + (unsigned)(*p - '0'); if (m) sig++; }
    else { simple = 0; dexp++; }
  }
  if (*p == '.') {
    p++;
    for (; *p >= '0' && *p <= '9'; p++) {
      any = 1;
      if (sig < 15) { m = m * 10 + (unsigned)(*p - '0'); if (m) sig++; dexp--; }
      else simple = 0;
    }
  }
  if (any && (*p == 'e' || *p == 'E')) {
    char const *q = p + 1;
    int eneg = 0, e = 0, edig = 0;
    if (*q == '-') { eneg = 1; q++; } else if (*q == '+') q++;
    for (; *q >= '0' && *q <= '9'; q++) { if (e < 10000) e = e * 10 + (*q - '0'); edig = 1; }
    if (edig) { dexp += eneg ? -e : e; p = q; }
  }
  int const ends = (*p == '\0' || *p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\v' || *p == '\f');
  if (any && simple && ends && dexp >= -22 && dexp <= 22) {
    double v = (double)m;                                /* exact: m < 10^15 < 2^53 */
    v = dexp < 0 ? v / p10[-dexp] : v * p10[dexp];
    *out = neg ? -v : v;
    *pp = p;
    return 1;
  }
  char *end;
  double const v = strtod(start, &end);
  if (end == start) return 0;
  *out = v;
  *pp = end;
  return 1;
}

/* Host threads worth starting for file work: online CPUs capped by the cgroup CPU quota and by 32
 * (JUR_IO_THREADS overrides). */
static int io_threads(void) {
  char const *e = getenv("JUR_IO_THREADS");
  if (e && atoi(e) > 0) return atoi(e);
  long n = sysconf(_SC_NPROCESSORS_ONLN);
  FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f) {
    char q[64];
    long period = 0;
    if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      long const c = (atol(q) + period - 1) / period;
      if (c >= 1 && c < n) n = c;
    }
    fclose(f);
  }
  if (n < 1) n = 1;
  return n > 32 ? 32 : (int)n;
}

/* One table file per (emitter, channel) pair, jurassic.c:329-400; the files are independent, so they are
 * parsed by a few threads (upstream reads them one after the other: 1.2 s for the 10 files of the example,
 * minutes for thousands of channels). */
typedef struct {
  jur_tables_t *tb;
  ctl_t const *ctl;
  int next, found, rc;
  long ignored;
  char err[512];
  pthread_mutex_t mu;
} ascii_job_t;

static void *ascii_worker(void *arg) {
  ascii_job_t *job = (ascii_job_t *)arg;
  ctl_t const *ctl = job->ctl;
  jur_tables_t *tb = job->tb;
  char *filename = (char *)malloc(2 * JUR_LEN + 64);
  long ignored = 0;
  int found = 0;
  for (;;) {
    pthread_mutex_lock(&job->mu);
    int const k = (job->rc == JUR_OK) ? job->next++ : ctl->ng * ctl->nd;
    pthread_mutex_unlock(&job->mu);
    if (k >= ctl->ng * ctl->nd || !filename) break;
    int const ig = k / ctl->nd, id = k % ctl->nd;
    snprintf(filename, 2 * JUR_LEN + 64, "%s_%.4f_%s.tab", ctl->tblbase, ctl->nu[id], ctl->emitter[ig]);
    FILE *in = fopen(filename, "r");
    if (!in) continue;                                  /* transparent gas for this channel */
    found++;
    /* the whole file in one buffer (few large reads), lines cut in place */
    size_t cap = 1 << 22, len = 0;
    char *buf = (char *)malloc(cap + 1);
    while (buf) {
      size_t const got = fread(buf + len, 1, cap - len, in);
      len += got;
      if (len < cap) break;
      cap *= 2;
      char *nb = (char *)realloc(buf, cap + 1);
      if (!nb) { free(buf); buf = NULL; }
      else buf = nb;
    }
    fclose(in);
    int rc = JUR_OK;
    if (!buf) { jur_set_error("out of memory reading table file"); rc = JUR_ENOMEM; }
    else {
      buf[len] = '\0';
      jur_pair_t *pr = &tb->pair[(size_t)ig * tb->nd + id];
      pair_clear(pr);
      feeder_t f;
      feeder_init(&f);
      for (char *l = buf; !rc && l < buf + len;) {
        char *const nl = (char *)memchr(l, '\n', (size_t)(buf + len - l));
        char *const next = nl ? nl + 1 : buf + len;
        /* upstream reads with fgets(line, LEN): a longer line continues as a new line */
        char *const stop = (next - l > JUR_LEN - 1) ? l + (JUR_LEN - 1) : next;
        char const saved = *stop;
        *stop = '\0';
        double eps = 0, press = 0, temp = 0, u = 0;
        char const *q = l;                               /* four numbers, as sscanf("%lg %lg %lg %lg") accepts them */
        if (jur_parse_number(&q, &press) && jur_parse_number(&q, &temp) && jur_parse_number(&q, &u) &&
            jur_parse_number(&q, &eps))
          rc = feed_one(&ignored, pr, &f, press, temp, u, eps);
        *stop = saved;
        l = stop;
      }
      free(buf);
    }
    if (rc) {
      pthread_mutex_lock(&job->mu);
      if (job->rc == JUR_OK) { job->rc = rc; snprintf(job->err, sizeof job->err, "%s: %s", filename, jur_last_error()); }
      pthread_mutex_unlock(&job->mu);
    }
  }
  free(filename);
  pthread_mutex_lock(&job->mu);
  job->found += found;
  job->ignored += ignored;
  pthread_mutex_unlock(&job->mu);
  return NULL;
}

int jur_tables_read_ascii(jur_tables_t *tb, ctl_t const *ctl) {
  if (!tb || tb->ng < ctl->ng || tb->nd < ctl->nd) { jur_set_error("read_ascii: tables smaller than ctl"); return JUR_EINVAL; }
  ascii_job_t job;
  memset(&job, 0, sizeof job);
  job.tb = tb; job.ctl = ctl;
  pthread_mutex_init(&job.mu, NULL);
  int nt = io_threads();
  if (nt > ctl->ng * ctl->nd) nt = ctl->ng * ctl->nd;
  pthread_t th[32];
  int started = 0;
  for (int i = 1; i < nt; i++)
    if (pthread_create(&th[started], NULL, ascii_worker, &job) == 0) started++;
  ascii_worker(&job);                                    /* the calling thread works too */
  for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
  pthread_mutex_destroy(&job.mu);
  tb->ignored_rows += job.ignored;
  if (job.rc) { jur_set_error("%s", job.err); return job.rc; }
  if (tb->ignored_rows > 0)
    fprintf(stderr, "Warning! %ld table entries ignored (more than %d column densities per curve)\n",
            tb->ignored_rows, JUR_TBLNU);
  return job.found;
}

/* Planck function as the reference evaluates it: C1 nu^3 / (exp(C2 nu / T) - 1)
 * (jurassic.c:860; GSL's expm1 is exp(x)-1 for |x| >= ln 2). */
static double planck(double t, double nu) {
  double const x = JUR_C2 * nu / t;
  double const em1 = (fabs(x) < M_LN2) ? expm1(x) : exp(x) - 1;
  return JUR_C1 * (nu * nu * nu) / em1;
}

int jur_tables_set_filter(jur_tables_t *tb, int id, int n, double const *nu, double const *f) {
  if (!tb || id < 0 || id >= tb->nd || n < 1) { jur_set_error("set_filter: bad arguments"); return JUR_EINVAL; }
  double *sr = tb->sr + (size_t)id * JUR_TBLNS;
  for (int it = 0; it < JUR_TBLNS; it++) {
    double const st = 100 + ((double)it - 0.0) * (400 - 100) / ((JUR_TBLNS - 1.0) - 0.0);
    double fsum = 0, fpsum = 0;
    for (int i = 0; i < n; i++) {
      fsum += f[i];
      fpsum += f[i] * planck(st, nu[i]);
    }
    sr[it] = fpsum / fsum;
  }
  tb->have_sr[id] = 1;
  return JUR_OK;
}

int jur_tables_read_filters(jur_tables_t *tb, ctl_t const *ctl) {
  double *nu = (double *)malloc(sizeof(double) * JUR_NSHAPE), *f = (double *)malloc(sizeof(double) * JUR_NSHAPE);
  char *line = (char *)malloc(JUR_LEN);
  int rc = JUR_OK;
  for (int id = 0; id < ctl->nd && rc == JUR_OK; id++) {
    char filename[JUR_LEN + 64];
    snprintf(filename, sizeof filename, "%s_%.4f.filt", ctl->tblbase, ctl->nu[id]);
    FILE *in = fopen(filename, "r");
    if (!in) { jur_set_error("cannot open filter function %s", filename); rc = JUR_EIO; break; }
    int n = 0;
    while (fgets(line, JUR_LEN, in))
      if (n < JUR_NSHAPE && sscanf(line, "%lg %lg", &nu[n], &f[n]) == 2) n++;
    fclose(in);
    if (n < 1) { jur_set_error("no data in filter function %s", filename); rc = JUR_EIO; break; }
    rc = jur_tables_set_filter(tb, id, n, nu, f);
  }
  free(nu); free(f); free(line);
  return rc;
}

long jur_tables_entries(jur_tables_t const *tb) {
  long n = 0;
  for (long i = 0; i < (long)tb->ng * tb->nd; i++)
    for (int ip = 0; ip < tb->pair[i].np; ip++)
      for (int it = 0; it < tb->pair[i].lv[ip].nt; it++) n += tb->pair[i].lv[ip].cv[it].nu;
  return n;
}

/* ---- flatten --------------------------------------------------------------- */
int jur_tables_flatten(jur_tables_t const *tb, jur_flat_t *out) {
  memset(out, 0, sizeof *out);
  long const npair = (long)tb->ng * tb->nd;
  long nlevel = 0, ncurve = 0, nentry = 0;
  for (long i = 0; i < npair; i++) {
    nlevel += tb->pair[i].np;
    for (int ip = 0; ip < tb->pair[i].np; ip++) {
      ncurve += tb->pair[i].lv[ip].nt;
      for (int it = 0; it < tb->pair[i].lv[ip].nt; it++) nentry += tb->pair[i].lv[ip].cv[it].nu;
    }
  }
  /* The kernels address descriptors (16 B) with 32-bit byte offsets from the array's base, and table entries (8 B)
   * with 32-bit byte offsets from the first entry of the (gas, channel) PAIR, whose 64-bit base is wave-uniform: any
   * number of entries, pairs of at most TBLNP x TBLNT x TBLNU = 364 800. */
  if (ncurve >= (1L << 27) || nlevel >= (1L << 27)) {
    jur_set_error("tables too large for 32-bit descriptor offsets (%ld curves, %ld levels)", ncurve, nlevel);
    return JUR_EINVAL;
  }
  out->nlevel = nlevel; out->ncurve = ncurve; out->nentry = nentry;
  out->pair = (jur_int2 *)calloc(npair > 0 ? npair : 1, sizeof(jur_int2));
  out->pair_e0 = (long long *)calloc(npair > 0 ? npair : 1, sizeof(long long));
  out->lvl = (jur_lvl_t *)calloc(nlevel + 2, sizeof(jur_lvl_t));
  out->crv = (jur_crv_t *)calloc(ncurve + 2, sizeof(jur_crv_t));
  out->ue = (jur_ue_t *)calloc(nentry + 2, sizeof(jur_ue_t));
  if (!out->pair || !out->pair_e0 || !out->lvl || !out->crv || !out->ue) { jur_flat_free(out); return JUR_ENOMEM; }
  long L = 0, K = 0, E = 0;
  int sorted = 1, strict = 1;
  for (long i = 0; i < npair; i++) {
    jur_pair_t const *pr = &tb->pair[i];
    out->pair[i].a = pr->np;
    out->pair[i].b = (int)L;
    out->pair_e0[i] = E;
    long const E0 = E;
    for (int ip = 0; ip < pr->np; ip++, L++) {
      out->lvl[L].p = pr->lv[ip].p;
      out->lvl[L].nt = pr->lv[ip].nt;
      out->lvl[L].c0 = (int)K;
      if (ip > 0 && !(pr->lv[ip - 1].p <= pr->lv[ip].p)) sorted = 0;
      if (ip > 0 && !(pr->lv[ip - 1].p < pr->lv[ip].p)) strict = 0;
      for (int it = 0; it < pr->lv[ip].nt; it++, K++) {
        jur_curve_t const *cv = &pr->lv[ip].cv[it];
        out->crv[K].t = cv->t;
        out->crv[K].nu = cv->nu;
        out->crv[K].e0 = (int)(E - E0);
        if (it > 0 && !(pr->lv[ip].cv[it - 1].t <= cv->t)) sorted = 0;
        if (it > 0 && !(pr->lv[ip].cv[it - 1].t < cv->t)) strict = 0;
        for (int iu = 0; iu < cv->nu; iu++, E++) {
          out->ue[E].u = cv->u[iu];
          out->ue[E].eps = cv->eps[iu];
          if (iu > 0 && !(cv->u[iu - 1] <= cv->u[iu] && cv->eps[iu - 1] <= cv->eps[iu])) sorted = 0;
          if (iu > 0 && !(cv->u[iu - 1] < cv->u[iu] && cv->eps[iu - 1] < cv->eps[iu])) strict = 0;   /* as stored: fp32 */
        }
      }
    }
  }
  out->sorted = sorted;
  out->strict = sorted && strict;
  out->max_pair_curves = 0;
  for (long i = 0; i < npair; i++) {
    int nc = 0;
    for (int ip = 0; ip < tb->pair[i].np; ip++) nc += tb->pair[i].lv[ip].nt;
    if (nc > out->max_pair_curves) out->max_pair_curves = nc;
  }
  return JUR_OK;
}

/* Do pairs a and b stand on the same (p, T) grid: same levels, same number of curves per level, same temperatures?
 * (Compared as doubles, exactly: a shared bracket must be THE bracket of every channel that shares it.) */
static int same_grid(jur_flat_t const *f, long a, long b) {
  if (f->pair[a].a != f->pair[b].a) return 0;
  jur_lvl_t const *la = f->lvl + f->pair[a].b, *lb = f->lvl + f->pair[b].b;
  for (int i = 0; i < f->pair[a].a; i++) {
    if (la[i].p != lb[i].p || la[i].nt != lb[i].nt) return 0;
    jur_crv_t const *ca = f->crv + la[i].c0, *cb = f->crv + lb[i].c0;
    for (int k = 0; k < la[i].nt; k++)
      if (ca[k].t != cb[k].t) return 0;
  }
  return 1;
}

/* Grid class of every pair (pairs of one gas with equal class stand on the same (p, T) grid; -1: no table) and
 * whether all its curves have at least two entries. */
int jur_flat_grid_classes(jur_flat_t const *f, int ng, int nd, int *cls, unsigned char *all_curves) {
  long *rep = (long *)malloc(sizeof(long) * (nd > 0 ? nd : 1));
  if (!rep) return JUR_ENOMEM;
  for (int g = 0; g < ng; g++) {
    int nclass = 0;
    for (int d = 0; d < nd; d++) {
      long const i = (long)g * nd + d;
      cls[i] = -1; all_curves[i] = 1;
      if (f->pair[i].a < 2) continue;            /* no table: the look-up answers 1, nothing to do */
      int c = 0;
      while (c < nclass && !same_grid(f, rep[c], i)) c++;
      if (c == nclass) rep[nclass++] = i;
      cls[i] = c;
      jur_lvl_t const *lv = f->lvl + f->pair[i].b;
      for (int ip = 0; ip < f->pair[i].a; ip++)
        for (int k = 0; k < lv[ip].nt; k++)
          if (f->crv[lv[ip].c0 + k].nu < 2) all_curves[i] = 0;
    }
  }
  free(rep);
  return JUR_OK;
}

/* Items of at most nch channels each from the classes: per gas, the channels of a class in ascending order. */
int jur_group_items(int ng, int nd, int nch, int const *cls, unsigned char const *all_curves, long long const *pair_e0,
                    jur_item_t **items, int *nitems, int *max_nch) {
  *items = NULL; *nitems = 0; *max_nch = 0;
  if (nch < 1) nch = 1;
  if (nch > JUR_EGA_NCH) nch = JUR_EGA_NCH;
  long const npair = (long)ng * nd;
  jur_item_t *out = (jur_item_t *)calloc(npair > 0 ? npair : 1, sizeof(jur_item_t));
  int *open = (int *)malloc(sizeof(int) * (nd > 0 ? nd : 1));      /* per class of the gas: its unfinished item */
  if (!out || !open) { free(out); free(open); return JUR_ENOMEM; }
  int n = 0;
  for (int g = 0; g < ng; g++) {
    for (int c = 0; c < nd; c++) open[c] = -1;
    for (int d = 0; d < nd; d++) {
      long const i = (long)g * nd + d;
      int const c = cls[i];
      if (c < 0) continue;
      if (open[c] < 0) { open[c] = n++; out[open[c]].g = g; out[open[c]].nch = 0; out[open[c]].flags = 1; }
      jur_item_t *it = &out[open[c]];
      if (!all_curves[i]) it->flags &= ~1;       /* the kernel tests for short curves only where an item has any */
      it->e0[it->nch] = pair_e0[i];
      it->d[it->nch++] = d;
      if (it->nch > *max_nch) *max_nch = it->nch;
      if (it->nch == nch) open[c] = -1;
    }
  }
  free(open);
  *items = out; *nitems = n;
  return JUR_OK;
}

void jur_flat_free(jur_flat_t *f) {
  free(f->pair); free(f->pair_e0); free(f->lvl); free(f->crv); free(f->ue);
  memset(f, 0, sizeof *f);
}

/* ---- compact binary cache (own format; upstream's cache is a raw dump of its dense 8.8 GB
 *      tbl_t, jr_binary_tables_io.h:213-233) ------------------------------------------------ */
#define JUR_CACHE_MAGIC "JURASSIC-HIP compact emissivity tables\n"
#define JUR_CACHE_VERSION 2      /* 2: curve offsets count from the first entry of their pair */

void jur_tables_cache_filename(char *out, size_t len, ctl_t const *ctl) {
  snprintf(out, len, "bin.jurassic-hip-tables-g%d-d%d", ctl->ng, ctl->nd);   /* cf. jr_binary_tables_io.h:12-16 */
}

unsigned long long jur_tables_checksum(jur_tables_t const *tb) {
  jur_flat_t fl;
  if (jur_tables_flatten(tb, &fl)) return 0;
  unsigned long long h = 1469598103934665603ULL;                      /* FNV-1a over the flattened image */
  struct { void const *p; size_t n; } part[] = {
    {fl.pair, sizeof(jur_int2) * (size_t)tb->ng * tb->nd}, {fl.lvl, sizeof(jur_lvl_t) * fl.nlevel},
    {fl.crv, sizeof(jur_crv_t) * fl.ncurve}, {fl.ue, sizeof(jur_ue_t) * fl.nentry},
    {tb->sr, sizeof(double) * JUR_TBLNS * (size_t)tb->nd}};
  for (size_t k = 0; k < sizeof part / sizeof part[0]; k++) {
    unsigned char const *b = (unsigned char const *)part[k].p;
    for (size_t i = 0; i < part[k].n; i++) { h ^= b[i]; h *= 1099511628211ULL; }
  }
  jur_flat_free(&fl);
  return h;
}

static int cache_header(char *buf, size_t len, ctl_t const *ctl, jur_flat_t const *fl) {
  int n = snprintf(buf, len, JUR_CACHE_MAGIC "version %d\nng %d\nnd %d\nnlevel %ld\nncurve %ld\nnentry %ld\ntblns %d\n",
                   JUR_CACHE_VERSION, ctl->ng, ctl->nd, fl ? fl->nlevel : 0L, fl ? fl->ncurve : 0L, fl ? fl->nentry : 0L,
                   JUR_TBLNS);
  for (int ig = 0; ig < ctl->ng && n < (int)len; ig++) n += snprintf(buf + n, len - n, "emitter %d %s\n", ig, ctl->emitter[ig]);
  for (int id = 0; id < ctl->nd && n < (int)len; id++) n += snprintf(buf + n, len - n, "channel %d %.4f\n", id, ctl->nu[id]);
  if (n < (int)len) n += snprintf(buf + n, len - n, "header_end\n");
  return (n < (int)len) ? n : -1;
}

int jur_tables_save(jur_tables_t const *tb, ctl_t const *ctl, char const *path) {
  if (!tb || tb->ng != ctl->ng || tb->nd != ctl->nd) { jur_set_error("tables_save: tables do not match ctl"); return JUR_EINVAL; }
  jur_flat_t fl;
  int rc = jur_tables_flatten(tb, &fl);
  if (rc) return rc;
  size_t const hlen = 256 + 64 * ((size_t)ctl->ng + ctl->nd) + (size_t)ctl->ng * JUR_LEN;
  char *hdr = (char *)malloc(hlen);
  int const n = hdr ? cache_header(hdr, hlen, ctl, &fl) : -1;
  FILE *out = (n > 0) ? fopen(path, "wb") : NULL;
  if (!out) { free(hdr); jur_flat_free(&fl); jur_set_error("cannot write table cache %s", path); return JUR_EIO; }
  size_t ok = fwrite(hdr, 1, (size_t)n, out) == (size_t)n;
  ok &= fwrite(fl.pair, sizeof(jur_int2), (size_t)tb->ng * tb->nd, out) == (size_t)tb->ng * tb->nd;
  ok &= fwrite(fl.lvl, sizeof(jur_lvl_t), fl.nlevel, out) == (size_t)fl.nlevel;
  ok &= fwrite(fl.crv, sizeof(jur_crv_t), fl.ncurve, out) == (size_t)fl.ncurve;
  ok &= fwrite(fl.ue, sizeof(jur_ue_t), fl.nentry, out) == (size_t)fl.nentry;
  ok &= fwrite(tb->sr, sizeof(double), (size_t)JUR_TBLNS * tb->nd, out) == (size_t)JUR_TBLNS * tb->nd;
  unsigned long long const sum = jur_tables_checksum(tb);
  ok &= fwrite(&sum, sizeof sum, 1, out) == 1;
  ok &= fclose(out) == 0;
  free(hdr);
  jur_flat_free(&fl);
  if (!ok) { jur_set_error("short write on table cache %s", path); return JUR_EIO; }
  return JUR_OK;
}

/* JUR_OK: *out holds the tables.  JUR_EIO: no such file / unreadable.  JUR_EINVAL: the file
 * belongs to other emitters, channels or dimensions, or is damaged (checksum). */
int jur_tables_load(jur_tables_t **out, ctl_t const *ctl, char const *path) {
  *out = NULL;
  FILE *in = fopen(path, "rb");
  if (!in) { jur_set_error("no table cache %s", path); return JUR_EIO; }
  size_t const hlen = 256 + 64 * ((size_t)ctl->ng + ctl->nd) + (size_t)ctl->ng * JUR_LEN;
  char *want = (char *)malloc(hlen), *have = (char *)malloc(hlen);
  jur_flat_t fl;
  memset(&fl, 0, sizeof fl);
  jur_tables_t *tb = NULL;
  int rc = JUR_EINVAL;
  /* the header is plain text; everything but the three counts must be what this ctl would write */
  int const n0 = cache_header(want, hlen, ctl, NULL);
  if (n0 < 0 || !have) goto fail;
  {
    size_t got = 0;
    int lines = 0, need = 9 + ctl->ng + ctl->nd;   /* magic, version, ng, nd, 3 counts, tblns, emitters, channels, header_end */
    while (got + 1 < hlen && lines < need) {
      int const ch = fgetc(in);
      if (ch == EOF) goto fail;
      have[got++] = (char)ch;
      if (ch == '\n') lines++;
    }
    have[got] = 0;
    long nl = -1, nc = -1, ne = -1;
    char const *p1 = strstr(have, "nlevel "), *p2 = strstr(have, "ncurve "), *p3 = strstr(have, "nentry ");
    if (!p1 || !p2 || !p3 || sscanf(p1, "nlevel %ld", &nl) != 1 || sscanf(p2, "ncurve %ld", &nc) != 1 ||
        sscanf(p3, "nentry %ld", &ne) != 1 || nl < 0 || nc < 0 || ne < 0) goto mismatch;
    fl.nlevel = nl; fl.ncurve = nc; fl.nentry = ne;
    if (cache_header(want, hlen, ctl, &fl) != (int)got || memcmp(want, have, got) != 0) goto mismatch;
  }
  {
    size_t const npair = (size_t)ctl->ng * ctl->nd;
    fl.pair = (jur_int2 *)malloc(sizeof(jur_int2) * (npair + 1));
    fl.lvl = (jur_lvl_t *)malloc(sizeof(jur_lvl_t) * (fl.nlevel + 1));
    fl.crv = (jur_crv_t *)malloc(sizeof(jur_crv_t) * (fl.ncurve + 1));
    fl.ue = (jur_ue_t *)malloc(sizeof(jur_ue_t) * (fl.nentry + 1));
    tb = jur_tables_new(ctl->ng, ctl->nd);
    if (!fl.pair || !fl.lvl || !fl.crv || !fl.ue || !tb) { rc = JUR_ENOMEM; goto fail; }
    unsigned long long sum = 0;
    if (fread(fl.pair, sizeof(jur_int2), npair, in) != npair || fread(fl.lvl, sizeof(jur_lvl_t), fl.nlevel, in) != (size_t)fl.nlevel ||
        fread(fl.crv, sizeof(jur_crv_t), fl.ncurve, in) != (size_t)fl.ncurve || fread(fl.ue, sizeof(jur_ue_t), fl.nentry, in) != (size_t)fl.nentry ||
        fread(tb->sr, sizeof(double), (size_t)JUR_TBLNS * ctl->nd, in) != (size_t)JUR_TBLNS * ctl->nd ||
        fread(&sum, sizeof sum, 1, in) != 1) goto mismatch;
    /* rebuild the hierarchy, validating every extent and offset on the way */
    long E0 = 0;                                   /* first entry of the pair: pairs lie one after the other */
    for (size_t i = 0; i < npair; i++) {
      int const np = fl.pair[i].a;
      long const L0 = fl.pair[i].b;
      long pair_entries = 0;
      if (np < 0 || np > JUR_TBLNP || L0 < 0 || L0 + np > fl.nlevel) goto mismatch;
      jur_pair_t *pr = &tb->pair[i];
      if (np == 0) continue;
      pr->lv = (jur_level_t *)calloc((size_t)np, sizeof(jur_level_t));
      if (!pr->lv) { rc = JUR_ENOMEM; goto fail; }
      pr->np = np;
      for (int ip = 0; ip < np; ip++) {
        jur_lvl_t const *l = &fl.lvl[L0 + ip];
        if (l->nt < 0 || l->nt > JUR_TBLNT || l->c0 < 0 || (long)l->c0 + l->nt > fl.ncurve) goto mismatch;
        pr->lv[ip].p = l->p;
        pr->lv[ip].nt = l->nt;
        for (int it = 0; it < l->nt; it++) {
          jur_crv_t const *c = &fl.crv[l->c0 + it];
          if (c->nu < 0 || c->nu > JUR_TBLNU || c->e0 != pair_entries || E0 + c->e0 + c->nu > fl.nentry) goto mismatch;
          pair_entries += c->nu;
          jur_curve_t *cv = &pr->lv[ip].cv[it];
          cv->t = c->t;
          cv->nu = cv->cap = c->nu;
          cv->u = (float *)malloc(sizeof(float) * (c->nu > 0 ? c->nu : 1));
          cv->eps = (float *)malloc(sizeof(float) * (c->nu > 0 ? c->nu : 1));
          if (!cv->u || !cv->eps) { rc = JUR_ENOMEM; goto fail; }
          for (int iu = 0; iu < c->nu; iu++) { cv->u[iu] = fl.ue[E0 + c->e0 + iu].u; cv->eps[iu] = fl.ue[E0 + c->e0 + iu].eps; }
        }
      }
      E0 += pair_entries;
    }
    for (int id = 0; id < ctl->nd; id++) tb->have_sr[id] = 1;
    if (jur_tables_checksum(tb) != sum) goto mismatch;
  }
  fclose(in);
  free(want); free(have);
  jur_flat_free(&fl);
  *out = tb;
  return JUR_OK;
mismatch:
  jur_set_error("table cache %s does not match this control block (emitters, channels, dimensions) or is damaged", path);
  rc = JUR_EINVAL;
fail:
  fclose(in);
  free(want); free(have);
  jur_flat_free(&fl);
  jur_tables_free(tb);
  return rc;
}
