/* jurassic_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference CPU forward model.  Every function
 * names the reference lines it follows (paths relative to the reference's
 * src/).  Floating-point expressions keep the reference's association order so
 * that a same-libm build of the reference would agree to the last bits.
 * See jurassic_oracle.h for the parity-pin status.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <assert.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "jurassic_oracle.h"
/* The struct schema comes from the product's ABI header (it is what both sides must agree on, and it is held
 * against the reference's offsets by tests/test_abi_cpu.py); the physical constants do NOT: */
#include "oracle_constants.h"
#undef JUR_C1
#undef JUR_C2
#undef JUR_P0
#undef JUR_RE
#undef JUR_AVOGADRO
#undef JUR_BOLTZMANN
#undef JUR_MOLAR_GAS

#define NLOS JUR_NLOS
#define NGX  JUR_NG
#define NWX  JUR_NW

/* ------------------------------------------------------------------------ */
/* continuum coefficient data (tools/extract_ctm.py)                         */
/* ------------------------------------------------------------------------ */
#ifndef CTM_BLOB_PATH
#error "compile with -DCTM_BLOB_PATH=\"...ctm.bin\""
#endif
__asm__(".section .rodata\n"
        ".balign 16\n"
        ".global orc_ctm_blob\n"
        ".hidden orc_ctm_blob\n"
        "orc_ctm_blob:\n"
        ".incbin \"" CTM_BLOB_PATH "\"\n"
        ".previous\n");
extern const double orc_ctm_blob[] __attribute__((visibility("hidden")));
#define CO2_296 (orc_ctm_blob + 0)
#define CO2_260 (orc_ctm_blob + 2001)
#define CO2_230 (orc_ctm_blob + 4002)
#define H2O_296 (orc_ctm_blob + 6003)
#define H2O_260 (orc_ctm_blob + 8004)
#define H2O_FRN (orc_ctm_blob + 10005)
#define N2_B    (orc_ctm_blob + 12006)
#define N2_BETA (orc_ctm_blob + 12104)
#define O2_B    (orc_ctm_blob + 12202)
#define O2_BETA (orc_ctm_blob + 12292)

/* line-of-sight point, reference jurassic.h:351-369 */
typedef struct {
  double z, lon, lat, p, t;
  double q[NGX];
  double k[NWX];
  double ds;
  double u[NGX];
} pos_t;

/* ------------------------------------------------------------------------ */
/* table storage                                                              */
/* ------------------------------------------------------------------------ */
#define T_NP(tb, ig, id)          (tb)->np[(size_t)(ig) * (tb)->nd + (id)]
#define T_NT(tb, ig, ip, id)      (tb)->nt[((size_t)(ig) * (tb)->mp + (ip)) * (tb)->nd + (id)]
#define T_NU(tb, ig, ip, it, id)  (tb)->nu[(((size_t)(ig) * (tb)->mp + (ip)) * (tb)->mt + (it)) * (tb)->nd + (id)]
#define T_P(tb, ig, ip, id)       (tb)->p[((size_t)(ig) * (tb)->mp + (ip)) * (tb)->nd + (id)]
#define T_T(tb, ig, ip, it, id)   (tb)->t[(((size_t)(ig) * (tb)->mp + (ip)) * (tb)->mt + (it)) * (tb)->nd + (id)]
#define T_IDX(tb, ig, ip, it, iu, id) \
  (((((size_t)(ig) * (tb)->mp + (ip)) * (tb)->mt + (it)) * (tb)->mu + (iu)) * (tb)->nd + (id))
#define T_U(tb, ig, ip, it, iu, id)   (tb)->u[T_IDX(tb, ig, ip, it, iu, id)]
#define T_EPS(tb, ig, ip, it, iu, id) (tb)->eps[T_IDX(tb, ig, ip, it, iu, id)]
#define T_SR(tb, it, id)          (tb)->sr[(size_t)(it) * (tb)->nd + (id)]

orc_tbl_t *orc_tbl_new(int ng, int nd, int mp, int mt, int mu) {
  orc_tbl_t *tb = (orc_tbl_t *)calloc(1, sizeof(orc_tbl_t));
  if (mp <= 0) mp = JUR_TBLNP;
  if (mt <= 0) mt = JUR_TBLNT;
  if (mu <= 0) mu = JUR_TBLNU;
  tb->ng = ng; tb->nd = nd; tb->mp = mp; tb->mt = mt; tb->mu = mu;
  size_t gd = (size_t)ng * nd;
  tb->np  = (int32_t *)calloc(gd, sizeof(int32_t));
  tb->nt  = (int32_t *)calloc(gd * mp, sizeof(int32_t));
  tb->nu  = (int32_t *)calloc(gd * mp * mt, sizeof(int32_t));
  tb->p   = (double *)calloc(gd * mp, sizeof(double));
  tb->t   = (double *)calloc(gd * mp * mt, sizeof(double));
  tb->u   = (float *)calloc(gd * mp * mt * mu, sizeof(float));
  tb->eps = (float *)calloc(gd * mp * mt * mu, sizeof(float));
  tb->sr  = (double *)calloc((size_t)JUR_TBLNS * nd, sizeof(double));
  /* jurassic.c:613-615: st = LIN(0, 100, TBLNS-1, 400, it) */
  for (int it = 0; it < JUR_TBLNS; it++)
    tb->st[it] = 100 + ((double)it - 0.0) * (400 - 100) / ((JUR_TBLNS - 1.0) - 0.0);
  return tb;
}

void orc_tbl_free(orc_tbl_t *tb) {
  if (!tb) return;
  free(tb->np); free(tb->nt); free(tb->nu); free(tb->p); free(tb->t);
  free(tb->u); free(tb->eps); free(tb->sr); free(tb);
}

/* Row-acceptance state machine of jurassic.c:346-395.  The three running
 * counters live in the table arrays themselves, as in the reference (they hold
 * "last index" while parsing and are turned into counts afterwards). */
typedef struct {
  double eps_old, press_old, temp_old, u_old;
  long ignored;
} feed_t;

static void feed_begin(orc_tbl_t *tb, int ig, int id, feed_t *f) {
  T_NP(tb, ig, id) = -1;
  f->eps_old = f->press_old = f->temp_old = f->u_old = -999;
  f->ignored = 0;
}

static void feed_row(orc_tbl_t *tb, int ig, int id, feed_t *f,
                     double press, double temp, double u, double eps) {
  if (press != f->press_old) {                       /* jurassic.c:353-357 */
    f->press_old = press;
    if (++T_NP(tb, ig, id) >= tb->mp) { fprintf(stderr, "oracle: too many pressure levels\n"); exit(1); }
    T_NT(tb, ig, T_NP(tb, ig, id), id) = -1;
  }
  int ip = T_NP(tb, ig, id);
  if (temp != f->temp_old) {                         /* jurassic.c:358-363 */
    f->temp_old = temp;
    if (++T_NT(tb, ig, ip, id) >= tb->mt) { fprintf(stderr, "oracle: too many temperatures\n"); exit(1); }
    T_NU(tb, ig, ip, T_NT(tb, ig, ip, id), id) = -1;
  }
  int it = T_NT(tb, ig, ip, id);
  if (it < 0) { fprintf(stderr, "oracle: table block repeats previous temperature\n"); exit(1); }
  if ((eps > f->eps_old && u > f->u_old) || T_NU(tb, ig, ip, it, id) < 0) { /* :364-375 */
    f->eps_old = eps;
    f->u_old = u;
    if (++T_NU(tb, ig, ip, it, id) >= tb->mu) {
      f->ignored++;
      T_NU(tb, ig, ip, it, id)--;
      return;
    }
  }
  int iu = T_NU(tb, ig, ip, it, id);
  T_P(tb, ig, ip, id) = press;                       /* jurassic.c:377-380 */
  T_T(tb, ig, ip, it, id) = temp;
  T_U(tb, ig, ip, it, iu, id) = (float)u;
  T_EPS(tb, ig, ip, it, iu, id) = (float)eps;
}

static void feed_end(orc_tbl_t *tb, int ig, int id) { /* jurassic.c:386-391 */
  T_NP(tb, ig, id)++;
  for (int ip = 0; ip < T_NP(tb, ig, id); ip++) {
    T_NT(tb, ig, ip, id)++;
    for (int it = 0; it < T_NT(tb, ig, ip, id); it++) T_NU(tb, ig, ip, it, id)++;
  }
}

void orc_tbl_feed_rows(orc_tbl_t *tb, int ig, int id, long nrows, double const *press,
                       double const *temp, double const *u, double const *eps) {
  feed_t f;
  feed_begin(tb, ig, id, &f);
  for (long i = 0; i < nrows; i++) feed_row(tb, ig, id, &f, press[i], temp[i], u[i], eps[i]);
  feed_end(tb, ig, id);
}

int orc_tbl_read_ascii(orc_tbl_t *tb, ctl_t const *ctl) {
  int missing = 0;
  for (int ig = 0; ig < ctl->ng; ig++)
    for (int id = 0; id < ctl->nd; id++) {
      char filename[2 * JUR_LEN + 64], line[JUR_LEN];
      snprintf(filename, sizeof filename, "%s_%.4f_%s.tab", ctl->tblbase, ctl->nu[id], ctl->emitter[ig]);
      FILE *in = fopen(filename, "r");                /* jurassic.c:337-345 */
      if (!in) { missing++; continue; }
      feed_t f;
      feed_begin(tb, ig, id, &f);
      while (fgets(line, JUR_LEN, in)) {
        double eps = 0, press = 0, temp = 0, u = 0;
        if (sscanf(line, "%lg %lg %lg %lg", &press, &temp, &u, &eps) != 4) continue;
        feed_row(tb, ig, id, &f, press, temp, u, eps);
      }
      feed_end(tb, ig, id);
      fclose(in);
    }
  return missing;
}

/* jurassic.c:860 planck(); gsl_pow_3 = x*x*x.  gsl_expm1 (GSL 2.5 sys/expm1.c)
 * is exp(x)-1 for |x| >= ln 2 and a Taylor sum below; libm expm1 stands in for
 * the latter branch (never taken for IR channels: C2*nu/T >= 2.3 at 650 cm^-1,
 * 400 K). */
double orc_planck(double t, double nu) {
  double const x = ORC_C2 * nu / t;
  double const em1 = (fabs(x) < M_LN2) ? expm1(x) : exp(x) - 1;
  return ORC_C1 * (nu * nu * nu) / em1;
}

void orc_tbl_planck_shape(orc_tbl_t *tb, int id, int n, double const *nu, double const *f) {
  for (int it = 0; it < JUR_TBLNS; it++) {             /* jurassic.c:654-664 */
    double fsum = 0, fpsum = 0;
    for (int i = 0; i < n; i++) {
      fsum += f[i];
      fpsum += f[i] * orc_planck(tb->st[it], nu[i]);
    }
    T_SR(tb, it, id) = fpsum / fsum;
  }
}

int orc_tbl_planck_filt(orc_tbl_t *tb, ctl_t const *ctl) {
  for (int id = 0; id < ctl->nd; id++) {
    char filename[JUR_LEN + 64], line[JUR_LEN];
    static double f[JUR_NSHAPE], nu[JUR_NSHAPE];
    snprintf(filename, sizeof filename, "%s_%.4f.filt", ctl->tblbase, ctl->nu[id]);
    FILE *in = fopen(filename, "r");
    if (!in) return -1;
    int n = 0;                                          /* read_shape, jurassic.c:1134-1150 */
    while (fgets(line, JUR_LEN, in))
      if (sscanf(line, "%lg %lg", &nu[n], &f[n]) == 2)
        if ((++n) > JUR_NSHAPE) { fprintf(stderr, "oracle: too many shape points\n"); exit(1); }
    fclose(in);
    if (n < 1) return -1;
    orc_tbl_planck_shape(tb, id, n, nu, f);
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* small helpers, jr_common.h:43-57                                           */
/* ------------------------------------------------------------------------ */
static inline double c01(double x) { return (x > 1.) ? 1. : ((x < 0.) ? 0. : x); }

static inline double lip(double x0, double y0, double x1, double y1, double x) {
  return y0 + (x - x0) * (y1 - y0) / (x1 - x0);
}

static inline double eip(double x0, double y0, double x1, double y1, double x) {
  if ((y0 > 0) && (y1 > 0)) return y0 * exp(log(y1 / y0) / (x1 - x0) * (x - x0));
  return lip(x0, y0, x1, y1, x);
}

/* jr_common.h:87-104 */
static inline int locate(double const *xx, int n, double x) {
  int ilo = 0, ihi = n - 1, i = (n - 1) >> 1;
  if (xx[i] < xx[i + 1]) {
    while (ihi > ilo + 1) {
      i = (ihi + ilo) >> 1;
      if (xx[i] > x) ihi = i; else ilo = i;
    }
  } else {
    while (ihi > ilo + 1) {
      i = (ihi + ilo) >> 1;
      if (xx[i] <= x) ihi = i; else ilo = i;
    }
  }
  return ilo;
}

/* ------------------------------------------------------------------------ */
/* EGA, jr_common.h:106-125,156-185,237-280                                   */
/* ------------------------------------------------------------------------ */
static inline int locate_p(orc_tbl_t const *tb, int ig, int n, double x, int id) {
  int ilo = 0, ihi = n - 1;                            /* locate_id on tbl->p[ig] */
  while (ihi > ilo + 1) {
    int i = (ihi + ilo) >> 1;
    if (T_P(tb, ig, i, id) > x) ihi = i; else ilo = i;
  }
  return ilo;
}

static inline int locate_t(orc_tbl_t const *tb, int ig, int ip, int n, double x, int id) {
  int ilo = 0, ihi = n - 1;                            /* locate_id on tbl->t[ig][ip] */
  while (ihi > ilo + 1) {
    int i = (ihi + ilo) >> 1;
    if (T_T(tb, ig, ip, i, id) > x) ihi = i; else ilo = i;
  }
  return ilo;
}

/* locate_tbl_id (jr_common.h:116-125) on a float curve with stride nd */
static inline int locate_curve(float const *xx, size_t stride, int n, double x) {
  int ilo = 0, ihi = n - 1;
  while (ihi > ilo + 1) {
    int i = (ihi + ilo) >> 1;
    if (xx[(size_t)i * stride] > x) ihi = i; else ilo = i;
  }
  return ilo;
}

static inline double get_eps(orc_tbl_t const *tb, int ig, int id, int ip, int it, double u) {
  size_t const s = (size_t)tb->nd;
  float const *uu = &T_U(tb, ig, ip, it, 0, id), *ee = &T_EPS(tb, ig, ip, it, 0, id);
  int const idx = locate_curve(uu, s, T_NU(tb, ig, ip, it, id), u);
  return lip(uu[idx * s], ee[idx * s], uu[(idx + 1) * s], ee[(idx + 1) * s], u);
}

static inline double get_u(orc_tbl_t const *tb, int ig, int id, int ip, int it, double eps) {
  size_t const s = (size_t)tb->nd;
  float const *uu = &T_U(tb, ig, ip, it, 0, id), *ee = &T_EPS(tb, ig, ip, it, 0, id);
  int const idx = locate_curve(ee, s, T_NU(tb, ig, ip, it, id), eps);
  return lip(ee[idx * s], uu[idx * s], ee[(idx + 1) * s], uu[(idx + 1) * s], eps);
}

double orc_ega_eps(orc_tbl_t const *tb, double tau, double t, double u, double p, int ig, int id) {
  if (tau < 1e-9) return 0.;                           /* jr_common.h:239 */
  if (T_NP(tb, ig, id) < 2) return 1.;
  int const ipr = locate_p(tb, ig, T_NP(tb, ig, id), p, id);
  if (T_NT(tb, ig, ipr, id) < 2 || T_NT(tb, ig, ipr + 1, id) < 2) return 1.;
  int const it0 = locate_t(tb, ig, ipr, T_NT(tb, ig, ipr, id), t, id);
  if (T_NU(tb, ig, ipr, it0, id) < 2 || T_NU(tb, ig, ipr, it0 + 1, id) < 2) return 1.;
  int const it1 = locate_t(tb, ig, ipr + 1, T_NT(tb, ig, ipr + 1, id), t, id);
  if (T_NU(tb, ig, ipr + 1, it1, id) < 2 || T_NU(tb, ig, ipr + 1, it1 + 1, id) < 2) return 1.;

  double const eps = 1 - tau;
  double const u00 = get_u(tb, ig, id, ipr, it0, eps);
  double const u01 = get_u(tb, ig, id, ipr, it0 + 1, eps);
  double const u10 = get_u(tb, ig, id, ipr + 1, it1, eps);
  double const u11 = get_u(tb, ig, id, ipr + 1, it1 + 1, eps);

  double const eps00 = c01(get_eps(tb, ig, id, ipr, it0, u00 + u));
  double const eps01 = c01(get_eps(tb, ig, id, ipr, it0 + 1, u01 + u));
  double const eps10 = c01(get_eps(tb, ig, id, ipr + 1, it1, u10 + u));
  double const eps11 = c01(get_eps(tb, ig, id, ipr + 1, it1 + 1, u11 + u));

  double const eps_p0 = c01(lip(T_T(tb, ig, ipr, it0, id), eps00, T_T(tb, ig, ipr, it0 + 1, id), eps01, t));
  double const eps_p1 = c01(lip(T_T(tb, ig, ipr + 1, it1, id), eps10, T_T(tb, ig, ipr + 1, it1 + 1, id), eps11, t));
  double const eps_t = c01(lip(T_P(tb, ig, ipr, id), eps_p0, T_P(tb, ig, ipr + 1, id), eps_p1, p));
  return (1. - eps_t) / tau;
}

/* jr_common.h:270-280: the loop runs to NG with eps=1 beyond ng */
static inline double apply_ega_core(orc_tbl_t const *tb, pos_t const *los, double *tau_path, int ng, int id) {
  double tau_gas = 1.0;
  for (int ig = 0; ig < NGX; ig++) {
    double eps = 1.0;
    if (ig < ng) eps = orc_ega_eps(tb, tau_path[ig], los->t, los->u[ig], los->p, ig, id);
    tau_path[ig] *= eps;
    tau_gas *= eps;
  }
  return tau_gas;
}

/* ------------------------------------------------------------------------ */
/* continua, jr_common.h:315-390                                              */
/* ------------------------------------------------------------------------ */
double orc_ctmco2(double nu, double p, double t, double u) {
  if (nu < 0 || nu >= 4000) return 0;
  double const xw = nu * 0.5 + 1;
  int const iw = (int)xw;
  double const dw = xw - iw;
  double const ew = 1 - dw;
  double const cw296 = ew * CO2_296[iw - 1] + dw * CO2_296[iw];
  double const cw260 = ew * CO2_260[iw - 1] + dw * CO2_260[iw];
  double const cw230 = ew * CO2_230[iw - 1] + dw * CO2_230[iw];
  double const dt230 = t - 230;
  double const dt260 = t - 260;
  double const dt296 = t - 296;
  double const ctw = dt260 * 5.050505e-4 * dt296 * cw230 - dt230 * 9.259259e-4 * dt296 * cw260
                   + dt230 * 4.208754e-4 * dt260 * cw296;
  return u * p * ctw / (ORC_AVOGADRO * 1000 * ORC_P0);
}

double orc_ctmh2o(double nu, double p, double t, double q, double u) {
  if (nu < 0 || nu >= 20000) return 0;
  double const xw = nu / 10 + 1;
  int const iw = (int)xw;
  double const dw = xw - iw;
  double const ew = 1 - dw;
  double const cw296 = ew * H2O_296[iw - 1] + dw * H2O_296[iw];
  double const cw260 = ew * H2O_260[iw - 1] + dw * H2O_260[iw];
  double const cwfrn = ew * H2O_FRN[iw - 1] + dw * H2O_FRN[iw];
  double sfac = 1.;
  if ((nu > 820.) && (nu < 960.)) {                    /* float island, jr_common.h:345-351 */
    char const xfcrev_char[16] = {3, 9, 15, 23, 29, 33, 37, 39, 40, 46, 36, 27, 10, 2, 0, 0};
    float const xx = nu * 0.1 - 82;
    int const ix = (int)xx;
    float const dx = xx - ix;
    sfac += .001 * ((1 - dx) * xfcrev_char[ix] + dx * xfcrev_char[ix + 1]);
  }
  double const ctwslf = sfac * cw296 * pow(cw260 / cw296, (296. - t) / (296. - 260.));
  double const vf1 = nu - 370.;
  double const vf2 = vf1 * vf1;
  double const vf6 = vf2 * vf2 * vf2;
  double const fscal = 36100. / (vf2 + vf6 * 1e-8 + 36100.) * -.25 + 1.;
  double const ctwfrn = cwfrn * fscal;
  double const a1 = nu * u * tanh(.7193876 / t * nu);
  double const a2 = 296. / t;
  double const a3 = p / ORC_P0 * (q * ctwslf + (1 - q) * ctwfrn) * 1e-20;
  return a1 * a2 * a3;
}

/* The reference reads ba[idx+1] one past the array when nu sits exactly on the
 * upper window edge (weight a1 = 0 there); the oracle clamps that read. */
double orc_ctmn2(double nu, double p, double t) {
  if (nu < 2120 || nu > 2605) return 0;
  double const xnu = nu * 0.2 - 424;
  int const idx = (int)xnu;
  int const idx1 = (idx + 1 < 98) ? idx + 1 : 97;
  double const a1 = xnu - idx, a0 = 1 - a1;
  double const b = a0 * N2_B[idx] + a1 * N2_B[idx1];
  double const beta = a0 * N2_BETA[idx] + a1 * N2_BETA[idx1];
  double const q_n2 = 0.79, t0 = 273, tr = 296;
  return 0.1 * (p / ORC_P0) * (p / ORC_P0) * (t0 / t) * (t0 / t) * exp(beta * (1 / tr - 1 / t)) * q_n2 * b
         * (q_n2 + (1 - q_n2) * (1.294 - 0.4545 * t / tr));
}

double orc_ctmo2(double nu, double p, double t) {
  if (nu < 1360 || nu > 1805) return 0;
  double const xnu = nu * 0.2 - 272;
  int const idx = (int)xnu;
  int const idx1 = (idx + 1 < 90) ? idx + 1 : 89;
  double const a1 = xnu - idx, a0 = 1 - a1;
  double const b = a0 * O2_B[idx] + a1 * O2_B[idx1];
  double const beta = a0 * O2_BETA[idx] + a1 * O2_BETA[idx1];
  double const q_o2 = 0.21, t0 = 273, tr = 296;
  return 0.1 * (p / ORC_P0) * (p / ORC_P0) * (t0 / t) * (t0 / t) * exp(beta * (1 / tr - 1 / t)) * q_o2 * b;
}

/* jr_continua_core.mv4g.h:1-14 with the four switches as run-time bits
 * (CO2=8, H2O=4, N2=2, O2=1; CPUdrivers.c:130-134) */
static inline double continua_core(int fourbit, ctl_t const *ctl, pos_t const *los, int ig_co2, int ig_h2o, int id) {
  double const p = los->p, t = los->t, ds = los->ds;
  double beta_ds = los->k[ctl->window[id]] * ds;
  if (fourbit & 8) beta_ds += orc_ctmco2(ctl->nu[id], p, t, los->u[ig_co2]);
  if (fourbit & 4) beta_ds += orc_ctmh2o(ctl->nu[id], p, t, los->q[ig_h2o], los->u[ig_h2o]);
  if (fourbit & 2) beta_ds += orc_ctmn2(ctl->nu[id], p, t) * ds;
  if (fourbit & 1) beta_ds += orc_ctmo2(ctl->nu[id], p, t) * ds;
  return beta_ds;
}

/* ------------------------------------------------------------------------ */
/* source function and radiance update, jr_common.h:187-234,293-300           */
/* ------------------------------------------------------------------------ */
static inline double src_planck_core(orc_tbl_t const *tb, double t, int id) {
  /* locate_st (jr_common.h:82-84) has no range check: for T outside [100, 400) K the reference reads outside
   * st[]/sr[] (undefined behaviour; whatever lies next to the table).  Oracle and device agree on ONE defined
   * behaviour for those temperatures instead: the index is clamped, i.e. the end intervals extrapolate linearly
   * (jur_kernels.hip planck_src; tests/test_kat_gpu.py holds the two against each other just outside the range).
   * Inside [100, 400) nothing changes. */
  int it = (int)(4 * t) - 400;                         /* locate_st */
  if (it < 0) it = 0;
  if (it > JUR_TBLNS - 2) it = JUR_TBLNS - 2;
  return lip(tb->st[it], T_SR(tb, it, id), tb->st[it + 1], T_SR(tb, it + 1, id), t);
}

/* function-level entries for known-answer tests */
double orc_src_planck(orc_tbl_t const *tb, double t, int id) { return src_planck_core(tb, t, id); }

void orc_new_obs(double tau_gas, double beta_ds, double src, double *rad, double *tau) {   /* new_obs_core, jr_common.h:293-300 */
  if (tau_gas > 1e-50) {
    double const eps = 1. - tau_gas * exp(-beta_ds);
    *rad += src * eps * (*tau);
    *tau *= (1. - eps);
  }
}

void orc_add_surface(orc_tbl_t const *tb, double tsurf, int id, double *rad, double tau) {  /* add_surface_core, :227-234 */
  if (tsurf > 0.) *rad += src_planck_core(tb, tsurf, id) * tau;
}

double orc_brightness(double rad, double nu) {        /* brightness_core */
  return ORC_C2 * nu / log1p((ORC_C1 * nu * nu * nu) / rad);
}

/* ------------------------------------------------------------------------ */
/* geometry, jr_common.h:475-500                                              */
/* ------------------------------------------------------------------------ */
#define DOTP(a, b) (a[0] * b[0] + a[1] * b[1] + a[2] * b[2])
#define NORM(a) sqrt(DOTP(a, a))
#define RAD2GRD (180 / M_PI)
#define GRD2RAD (M_PI / 180)

static inline double refractivity(double p, double t) { return 7.753e-05 * p / t; }

static inline void cart2geo(double const x[], double *alt, double *lon, double *lat) {
  double const radius = NORM(x);
  *lat = asin(x[2] / radius) * RAD2GRD;
  *lon = atan2(x[1], x[0]) * RAD2GRD;
  *alt = radius - ORC_RE;
}

static inline double cart2alt(double const x[]) { return NORM(x) - ORC_RE; }

static inline void geo2cart(double alt, double lon, double lat, double x[]) {
  double const radius = alt + ORC_RE, clat = cos(lat * GRD2RAD);
  x[0] = radius * clat * cos(lon * GRD2RAD);
  x[1] = radius * clat * sin(lon * GRD2RAD);
  x[2] = radius * sin(lat * GRD2RAD);
}

/* jr_common.h:127-154 */
static void locate_atm(atm_t const *atm, double time, size_t *atmIdx, int *atmNp) {
  int lo = 0, hi = atm->np - 1, i;
  while (hi > lo + 1) {
    i = (lo + hi) / 2;
    if (atm->time[i] < time) lo = i; else hi = i;
  }
  int const lower = (0 == lo) ? lo : hi;
  *atmIdx = (unsigned)lower;
  lo = lower;
  hi = atm->np - 1;
  while (hi > lo + 1) {
    i = (lo + hi) / 2;
    if (atm->time[i] > time) hi = i; else lo = i;
  }
  int const upper = (hi == atm->np - 1) ? atm->np : hi;
  *atmNp = upper - lower;
}

/* jr_common.h:411-420 */
static void altitude_range_nn(atm_t const *atm, size_t atmIdx, int atmNp, double *zmin, double *zmax) {
  *zmax = *zmin = atm->z[atmIdx];
  for (size_t ipp = atmIdx;
       (ipp < atmIdx + atmNp) && (atm->lon[ipp] == atm->lon[atmIdx]) && (atm->lat[ipp] == atm->lat[atmIdx]); ++ipp) {
    *zmax = fmax(*zmax, atm->z[ipp]);
    *zmin = fmin(*zmin, atm->z[ipp]);
  }
}

/* jr_common.h:549-567 (ctl->ip == 1 only, as the reference asserts) */
static inline void intpol_pt(atm_t const *atm, int idx0, int n, double z0, double *p, double *t) {
  int const ip = idx0 + locate(&atm->z[idx0], n, z0);
  *p = eip(atm->z[ip], atm->p[ip], atm->z[ip + 1], atm->p[ip + 1], z0);
  *t = lip(atm->z[ip], atm->t[ip], atm->z[ip + 1], atm->t[ip + 1], z0);
}

static inline void intpol_qk(ctl_t const *ctl, atm_t const *atm, int idx0, int n, double z0, double q[], double k[]) {
  int const ip = idx0 + locate(&atm->z[idx0], n, z0);
  for (int ig = 0; ig < ctl->ng; ig++)
    q[ig] = lip(atm->z[ip], atm->q[ig][ip], atm->z[ip + 1], atm->q[ig][ip + 1], z0);
  for (int iw = 0; iw < ctl->nw; iw++)
    k[iw] = lip(atm->z[ip], atm->k[iw][ip], atm->z[ip + 1], atm->k[iw][ip + 1], z0);
}

/* jr_common.h:502-539 */
static void tangent_point(pos_t const los[], int np, int ip, double *tpz, double *tplon, double *tplat) {
  if (ip <= 0 || ip >= np - 1) {
    *tpz = los[np - 1].z;
    *tplon = los[np - 1].lon;
    *tplat = los[np - 1].lat;
  } else {
    double const yy0 = los[ip - 1].z, yy1 = los[ip].z, yy2 = los[ip + 1].z,
                 ds0 = los[ip].ds, ds1 = los[ip + 1].ds,
                 dyy10 = yy1 - yy0, dyy21 = yy2 - yy1,
                 x1 = sqrt(ds0 * ds0 - dyy10 * dyy10),
                 x2 = x1 + sqrt(ds1 * ds1 - dyy21 * dyy21),
                 dx12 = x1 - x2,
                 a = (dyy10 * x2 + (yy0 - yy2) * x1) / (x1 * x2 * dx12),
                 b = dyy10 / x1 - a * x1,
                 c = yy0,
                 x = -b / (2 * a);
    *tpz = (a * x + b) * x + c;
    double v[3], v0[3], v2[3], dummy;
    geo2cart(los[ip - 1].z, los[ip - 1].lon, los[ip - 1].lat, v0);
    geo2cart(los[ip + 1].z, los[ip + 1].lon, los[ip + 1].lat, v2);
    for (int i = 0; i < 3; i++) v[i] = lip(0.0, v0[i], x2, v2[i], x);
    cart2geo(v, &dummy, tplon, tplat);
  }
}

/* jr_common.h:585-711.  geom = {time, obsz, obslon, obslat, vpz, vplon, vplat} */
static int traceray(ctl_t const *ctl, atm_t const *atm, double const geom[7], pos_t los[], double *tsurf, double tp[3]) {
  double ex0[3], ex1[3], q[NGX], k[NWX], lat, lon, p, t, x[3], xobs[3], xvp[3], z = 1e99, z_low = z, zmax, zmin,
         zrefrac = 60;
  double const obsz = geom[1], obslon = geom[2], obslat = geom[3], vpz = geom[4], vplon = geom[5], vplat = geom[6];
  *tsurf = -999;
  for (int ig = 0; ig < NGX; ig++) q[ig] = 0;
  for (int iw = 0; iw < NWX; iw++) k[iw] = 0;
  tp[0] = vpz; tp[1] = vplon; tp[2] = vplat;
  size_t atmIdx = 0;
  int atmNp = 0;
  locate_atm(atm, geom[0], &atmIdx, &atmNp);
  altitude_range_nn(atm, atmIdx, atmNp, &zmin, &zmax);
  if (obsz < zmin) return 0;
  if (vpz > zmax - 0.001) return 0;
  geo2cart(obsz, obslon, obslat, xobs);
  geo2cart(vpz, vplon, vplat, xvp);
  for (int i = 0; i < 3; i++) ex0[i] = xvp[i] - xobs[i];
  double const norm = NORM(ex0);
  for (int i = 0; i < 3; i++) {
    ex0[i] /= norm;
    x[i] = xobs[i];
  }
  if (obsz > zmax) {                                    /* entry-point bisection :610-621 */
    double dmax = norm, dmin = 0.;
    while (fabs(dmin - dmax) > 0.001) {
      double const d = 0.5 * (dmax + dmin);
      for (int i = 0; i < 3; i++) x[i] = xobs[i] + d * ex0[i];
      z = cart2alt(x);
      if ((z <= zmax) && (z > zmax - 0.001)) break;
      if (z < zmax - 0.0005) dmax = d; else dmin = d;
    }
  }

  int np = 0, z_low_idx = -1;
  for (int stop = 0; np < NLOS; ++np) {
    double ds = ctl->rayds, dz = ctl->raydz;
    if (dz > 0.) {
      double const norm_x = 1.0 / NORM(x);
      double dot = 0.;
      for (int i = 0; i < 3; i++) dot += ex0[i] * x[i] * norm_x;
      double const cosa = fabs(dot);
      if (cosa != 0.) ds = fmin(ds, dz / cosa);
    }
    cart2geo(x, &z, &lon, &lat);
    if ((z < zmin) || (z > zmax)) {                     /* LOS escaped :637-648 */
      double xh[3];
      stop = (z < zmin) ? 2 : 1;
      geo2cart(los[np - 1].z, los[np - 1].lon, los[np - 1].lat, xh);
      double const zfrac = (z < zmin) ? zmin : zmax;
      double const frac = (zfrac - los[np - 1].z) / (z - los[np - 1].z);
      for (int i = 0; i < 3; i++) x[i] = xh[i] + frac * (x[i] - xh[i]);
      cart2geo(x, &z, &lon, &lat);
      los[np - 1].ds = ds * frac;
      ds = 0.;
    }
    intpol_pt(atm, (int)atmIdx, atmNp, z, &p, &t);
    intpol_qk(ctl, atm, (int)atmIdx, atmNp, z, q, k);
    pos_t *pt = los + np;                               /* write_pos_point :422-434 */
    pt->lon = lon; pt->lat = lat; pt->z = z; pt->p = p; pt->t = t;
    for (int ig = 0; ig < NGX; ig++) pt->q[ig] = q[ig];
    for (int iw = 0; iw < NWX; iw++) pt->k[iw] = k[iw];
    pt->ds = ds;
    if (z < z_low) { z_low = z; z_low_idx = np; }
    if (stop) { *tsurf = (stop == 2 ? t : -999); break; }

    double n = 1., ngr[] = {0., 0., 0.};
    if (ctl->refrac && z <= zrefrac) {                  /* :665-681 */
      n += refractivity(p, t);
      double xh[3];
      for (int i = 0; i < 3; i++) xh[i] = x[i] + 0.5 * ds * ex0[i];
      cart2geo(xh, &z, &lon, &lat);
      intpol_pt(atm, (int)atmIdx, atmNp, z, &p, &t);
      double const n2 = refractivity(p, t);
      for (int i = 0; i < 3; i++) {
        double const h = 0.02;
        xh[i] += h;
        cart2geo(xh, &z, &lon, &lat);
        intpol_pt(atm, (int)atmIdx, atmNp, z, &p, &t);
        ngr[i] = (refractivity(p, t) - n2) / h;
        xh[i] -= h;
      }
    }
    for (int i = 0; i < 3; i++) ex1[i] = ex0[i] * n + ds * ngr[i];
    double const norm_ex1 = NORM(ex1);
    for (int i = 0; i < 3; i++) {
      ex1[i] /= norm_ex1;
      x[i] += 0.5 * ds * (ex0[i] + ex1[i]);
      ex0[i] = ex1[i];
    }
  }
  ++np;
  if (NLOS <= np) { printf("\nError (oracle traceray): Too many LOS points!\n\n"); exit(EXIT_FAILURE); }

  tangent_point(los, np, z_low_idx, &tp[0], &tp[1], &tp[2]);   /* before the trapezoid rule :698 */
  for (int ip = np - 1; ip >= 1; ip--) los[ip].ds = 0.5 * (los[ip - 1].ds + los[ip].ds); /* :437-443 */
  los[0].ds *= 0.5;
  for (int ip = 0; ip < np; ip++)                        /* column_density :446-453 */
    for (int ig = 0; ig < ctl->ng; ig++)
      los[ip].u[ig] = 10. * los[ip].q[ig] * los[ip].p / (ORC_BOLTZMANN * los[ip].t) * los[ip].ds;
  assert(1 != ctl->formod);
  return np;
}

int orc_traceray(ctl_t const *ctl, atm_t const *atm, double const geom[7], double *z, double *lon, double *lat,
                 double *p, double *t, double *ds, double *k, double *q, double *u, double *tsurf, double tp[3]) {
  pos_t *los = (pos_t *)malloc(sizeof(pos_t) * NLOS);
  int const np = traceray(ctl, atm, geom, los, tsurf, tp);
  for (int i = 0; i < np; i++) {
    z[i] = los[i].z; lon[i] = los[i].lon; lat[i] = los[i].lat; p[i] = los[i].p; t[i] = los[i].t;
    ds[i] = los[i].ds; k[i] = los[i].k[0];
    for (int ig = 0; ig < ctl->ng; ig++) { q[ig * NLOS + i] = los[i].q[ig]; u[ig * NLOS + i] = los[i].u[ig]; }
  }
  free(los);
  return np;
}

/* ------------------------------------------------------------------------ */
/* hydrostatic equilibrium, jr_common.h:212-217,713-761                       */
/* ------------------------------------------------------------------------ */
static double gravity(double z, double lat) {
  double const deg2rad = M_PI / 180., x = sin(lat * deg2rad), y = sin(2 * lat * deg2rad);
  return 9.780318 * (1. + 0.0053024 * x * x - 5.8e-6 * y * y) - 3.086e-3 * z;
}

static void hydrostatic_1d_h2o(ctl_t const *ctl, atm_t *atm, int ip0, int ip1, int ig_h2o) {
  int const npts = 20;
  double dzmin = 1e99;
  int ipref = 0;
  for (int ip = ip0; ip < ip1; ip++) {                  /* find_reference_parcel */
    double const dz = fabs(atm->z[ip] - ctl->hydz);
    if (dz < dzmin) { dzmin = dz; ipref = ip; }
  }
  double const lat = atm->lat[ipref];
  double const mmair = 28.96456e-3, mmh2o = 18.0153e-3;
  double e = 0.;
  for (int ip = ipref + 1; ip < ip1; ip++) {
    double mean = 0.;
    for (int i = 0; i < npts; i++) {
      double const z = lip(0.0, atm->z[ip - 1], npts - 1.0, atm->z[ip], (double)i);
      double const grav = gravity(z, lat);
      if (ig_h2o >= 0) e = lip(0.0, atm->q[ig_h2o][ip - 1], npts - 1.0, atm->q[ig_h2o][ip], (double)i);
      double const temp = lip(0.0, atm->t[ip - 1], npts - 1.0, atm->t[ip], (double)i);
      mean += (e * mmh2o + (1 - e) * mmair) * grav / (ORC_MOLAR_GAS * temp * npts);
    }
    atm->p[ip] = atm->p[ip - 1] * exp(-1000 * mean * (atm->z[ip] - atm->z[ip - 1]));
  }
  for (int ip = ipref - 1; ip >= ip0; ip--) {
    double mean = 0.;
    for (int i = 0; i < npts; i++) {
      double const z = lip(0.0, atm->z[ip + 1], npts - 1.0, atm->z[ip], (double)i);
      double const grav = gravity(z, lat);
      if (ig_h2o >= 0) e = lip(0.0, atm->q[ig_h2o][ip + 1], npts - 1.0, atm->q[ig_h2o][ip], (double)i);
      double const temp = lip(0.0, atm->t[ip + 1], npts - 1.0, atm->t[ip], (double)i);
      mean += (e * mmh2o + (1 - e) * mmair) * grav / (ORC_MOLAR_GAS * temp * npts);
    }
    atm->p[ip] = atm->p[ip + 1] * exp(-1000 * mean * (atm->z[ip] - atm->z[ip + 1]));
  }
}

int orc_find_emitter(ctl_t const *ctl, char const *name) { /* jurassic.c:199-209 */
  for (int ig = 0; ig < ctl->ng; ig++)
    if (0 == strcasecmp(ctl->emitter[ig], name)) return ig;
  return -1;
}

static void continua_config(ctl_t const *ctl, int *ig_co2, int *ig_h2o, int *fourbit) {
  *ig_co2 = -999; *ig_h2o = -999;                       /* CPUdrivers.c:126-134 */
  if (ctl->ctm_h2o) *ig_h2o = orc_find_emitter(ctl, "H2O");
  if (ctl->ctm_co2) *ig_co2 = orc_find_emitter(ctl, "CO2");
  *fourbit = ((1 == ctl->ctm_co2) && (*ig_co2 >= 0)) * 8 + ((1 == ctl->ctm_h2o) && (*ig_h2o >= 0)) * 4
           + (1 == ctl->ctm_n2) * 2 + (1 == ctl->ctm_o2) * 1;
}

void orc_hydrostatic(ctl_t const *ctl, atm_t *atm) {    /* CPUdrivers.c:98-103 (idempotent, done once) */
  int ig_co2, ig_h2o, fourbit;
  if (ctl->hydz < 0) return;
  continua_config(ctl, &ig_co2, &ig_h2o, &fourbit);
  hydrostatic_1d_h2o(ctl, atm, 0, atm->np, ig_h2o);
}

/* ------------------------------------------------------------------------ */
/* drivers, CPUdrivers.c:5-151                                                */
/* ------------------------------------------------------------------------ */
static void integrate_ray(ctl_t const *ctl, orc_tbl_t const *tb, pos_t const *los, int np, double tsurf,
                          int ig_co2, int ig_h2o, int fourbit, double *rad, double *tau, int nd_stride) {
  double tau_path[JUR_ND][NGX];                         /* apply_kernels_CPU :56-84 */
  for (int id = 0; id < nd_stride && id < JUR_ND; id++) {
    rad[id] = 0.0;
    tau[id] = 1.0;
  }
  for (int id = 0; id < ctl->nd; id++)
    for (int ig = 0; ig < NGX; ig++) tau_path[id][ig] = 1.0;
  for (int ip = 0; ip < np; ++ip)
    for (int id = 0; id < ctl->nd; id++) {
      double const beta_ds = continua_core(fourbit, ctl, &los[ip], ig_co2, ig_h2o, id);
      double const tau_gas = apply_ega_core(tb, &los[ip], tau_path[id], ctl->ng, id);
      double const planck = src_planck_core(tb, los[ip].t, id);
      if (tau_gas > 1e-50) {                            /* new_obs_core, jr_common.h:293-300 */
        double const eps = 1. - tau_gas * exp(-beta_ds);
        rad[id] += planck * eps * tau[id];
        tau[id] *= (1. - eps);
      }
    }
  if (tsurf > 0.)                                       /* add_surface_core, jr_common.h:227-234 */
    for (int id = 0; id < ctl->nd; id++) {
      int const it = (int)(4 * tsurf) - 400;
      double const src = lip(tb->st[it], T_SR(tb, it, id), tb->st[it + 1], T_SR(tb, it + 1, id), tsurf);
      rad[id] += src * tau[id];
    }
}

void orc_formod_rays(ctl_t const *ctl, atm_t *atm, orc_tbl_t const *tb, long nr, int nd_stride,
                     double const *time, double const *obsz, double const *obslon, double const *obslat,
                     double const *vpz, double const *vplon, double const *vplat,
                     double *tpz, double *tplon, double *tplat, double *rad, double *tau,
                     int *np_out, double *tsurf_out, int serial_trace) {
  if (ctl->checkmode) return;
  int ig_co2, ig_h2o, fourbit;
  continua_config(ctl, &ig_co2, &ig_h2o, &fourbit);
  if (!(ctl->hydz < 0)) hydrostatic_1d_h2o(ctl, atm, 0, atm->np, ig_h2o);

  if (serial_trace == 2) {  /* "rays-parallel": every thread traces and integrates its own rays, one LOS buffer per
                             * thread -- the fair many-core arrangement, same results per ray */
#pragma omp parallel
    {
      pos_t *los1 = (pos_t *)malloc(sizeof(pos_t) * NLOS);
#pragma omp for schedule(dynamic, 8)
      for (long ir = 0; ir < nr; ir++) {
        char mask1[JUR_ND];
        for (int id = 0; id < ctl->nd; id++) mask1[id] = !isfinite(rad[ir * nd_stride + id]);
        double const geom[7] = {time[ir], obsz[ir], obslon[ir], obslat[ir], vpz[ir], vplon[ir], vplat[ir]};
        double tp[3], ts;
        int const n1 = traceray(ctl, atm, geom, los1, &ts, tp);
        tpz[ir] = tp[0]; tplon[ir] = tp[1]; tplat[ir] = tp[2];
        integrate_ray(ctl, tb, los1, n1, ts, ig_co2, ig_h2o, fourbit, rad + ir * nd_stride, tau + ir * nd_stride, nd_stride);
        if (ctl->write_bbt)
          for (int id = 0; id < ctl->nd; id++) rad[ir * nd_stride + id] = orc_brightness(rad[ir * nd_stride + id], ctl->nu[id]);
        for (int id = 0; id < ctl->nd; id++)
          if (mask1[id]) rad[ir * nd_stride + id] = NAN;
        if (np_out) np_out[ir] = n1;
        if (tsurf_out) tsurf_out[ir] = ts;
      }
      free(los1);
    }
    return;
  }
  long const chunk = JUR_NR;                            /* the reference works in packages of NR rays */
  pos_t *los = (pos_t *)malloc(sizeof(pos_t) * NLOS * (size_t)chunk);
  int *np = (int *)malloc(sizeof(int) * chunk);
  double *tsurf = (double *)malloc(sizeof(double) * chunk);
  char *mask = (char *)malloc((size_t)chunk * (ctl->nd > 0 ? ctl->nd : 1));
  for (long r0 = 0; r0 < nr; r0 += chunk) {
    long const n = (nr - r0 < chunk) ? nr - r0 : chunk;
    for (long i = 0; i < n; i++)                        /* save_mask, jr_common.h:193-200 */
      for (int id = 0; id < ctl->nd; id++) mask[i * ctl->nd + id] = !isfinite(rad[(r0 + i) * nd_stride + id]);
#pragma omp parallel for schedule(dynamic, 4) if (serial_trace != 1)
    for (long i = 0; i < n; i++) {                      /* raytrace_rays_CPU */
      long const ir = r0 + i;
      double const geom[7] = {time[ir], obsz[ir], obslon[ir], obslat[ir], vpz[ir], vplon[ir], vplat[ir]};
      double tp[3];
      np[i] = traceray(ctl, atm, geom, los + (size_t)i * NLOS, &tsurf[i], tp);
      tpz[ir] = tp[0]; tplon[ir] = tp[1]; tplat[ir] = tp[2];
    }
#pragma omp parallel for schedule(dynamic, 4)
    for (long i = 0; i < n; i++) {                      /* apply_kernels_CPU + surface_terms_CPU */
      long const ir = r0 + i;
      integrate_ray(ctl, tb, los + (size_t)i * NLOS, np[i], tsurf[i], ig_co2, ig_h2o, fourbit,
                    rad + ir * nd_stride, tau + ir * nd_stride, nd_stride);
      if (ctl->write_bbt)                               /* radiance_to_brightness_CPU */
        for (int id = 0; id < ctl->nd; id++) rad[ir * nd_stride + id] = orc_brightness(rad[ir * nd_stride + id], ctl->nu[id]);
      for (int id = 0; id < ctl->nd; id++)              /* apply_mask */
        if (mask[i * ctl->nd + id]) rad[ir * nd_stride + id] = NAN;
      if (np_out) np_out[ir] = np[i];
      if (tsurf_out) tsurf_out[ir] = tsurf[i];
    }
  }
  free(los); free(np); free(tsurf); free(mask);
}

void orc_formod(ctl_t const *ctl, atm_t *atm, obs_t *obs, orc_tbl_t const *tb) {
  orc_formod_rays(ctl, atm, tb, obs->nr, JUR_ND, obs->time, obs->obsz, obs->obslon, obs->obslat, obs->vpz,
                  obs->vplon, obs->vplat, obs->tpz, obs->tplon, obs->tplat, &obs->rad[0][0], &obs->tau[0][0],
                  NULL, NULL, 1);
}

/* ------------------------------------------------------------------------ */
/* algorithmic byte count, SURVEY.md section 8(d)                             */
/* ------------------------------------------------------------------------ */
static inline int Lprobe(int n) {                       /* ceil(log2(n-1)) probes of the bisection */
  int l = 0;
  while ((1 << l) < n - 1) l++;
  return l;
}

double orc_algorithmic_bytes(ctl_t const *ctl, atm_t *atm, orc_tbl_t const *tb, long nr, double const *time,
                             double const *obsz, double const *obslon, double const *obslat, double const *vpz,
                             double const *vplon, double const *vplat, long *nseg_out, double *trace_part,
                             double *ega_part) {
  double total = 0, ttrace = 0, tega = 0;
  long nseg = 0;
#pragma omp parallel reduction(+ : total, ttrace, tega, nseg)
  {
    pos_t *los = (pos_t *)malloc(sizeof(pos_t) * NLOS);
#pragma omp for schedule(dynamic, 16)
    for (long ir = 0; ir < nr; ir++) {
      double const geom[7] = {time[ir], obsz[ir], obslon[ir], obslat[ir], vpz[ir], vplon[ir], vplat[ir]};
      double tp[3], tsurf;
      int const np = traceray(ctl, atm, geom, los, &tsurf, tp);
      size_t atmIdx; int atmNp;
      locate_atm(atm, geom[0], &atmIdx, &atmNp);
      int const La = Lprobe(atmNp);
      double bytes = 80 + 24 * ctl->nd;                 /* B_io */
      double tr = bytes, eg = 0;
      for (int ip = 0; ip < np; ip++) {
        double b_atm = 16 * La + 16 * (3 + ctl->ng + ctl->nw);
        if (ctl->refrac && los[ip].z <= 60 && ip < np - 1) b_atm += 4 * (8 * La + 48);
        tr += b_atm;
        bytes += b_atm + 8 * (5 + ctl->ng) + ctl->nd * 32.0;
        for (int id = 0; id < ctl->nd; id++)
          for (int ig = 0; ig < ctl->ng; ig++) {
            int const n_p = T_NP(tb, ig, id);
            if (n_p < 2) continue;
            int const ipr = locate_p(tb, ig, n_p, los[ip].p, id);
            int const n_t = T_NT(tb, ig, ipr, id);
            if (n_t < 2) continue;
            int const it0 = locate_t(tb, ig, ipr, n_t, los[ip].t, id);
            int const n_u = T_NU(tb, ig, ipr, it0, id);
            if (n_u < 2) continue;
            eg += 236 + 8 * Lprobe(n_p) + 16 * Lprobe(n_t) + 32 * Lprobe(n_u);
          }
      }
      total += bytes + eg;
      ttrace += tr;
      tega += eg;
      nseg += np;
    }
    free(los);
  }
  if (nseg_out) *nseg_out = nseg;
  if (trace_part) *trace_part = ttrace;
  if (ega_part) *ega_part = tega;
  return total;
}

/* ------------------------------------------------------------------------ */
/* retrieval interface: state/measurement vectors and the finite-difference   */
/* Jacobian, jurassic.c:812-857 (kernel), :1473-1541 (x2atm, atm2x, obs2y)    */
/* ------------------------------------------------------------------------ */
#define IDXP 0
#define IDXT 1
#define IDXQ(ig) (2 + (ig))
#define IDXK(iw) (2 + ctl->ng + (iw))

static void atm2x_help(atm_t const *atm, double zmin, double zmax, double const *value, int val_iqa, double *x,
                       int *iqa, int *ipa, size_t *n) {
  for (int ip = 0; ip < atm->np; ip++)
    if (atm->z[ip] >= zmin && atm->z[ip] <= zmax) {
      if (x) x[*n] = value[ip];
      if (iqa) iqa[*n] = val_iqa;
      if (ipa) ipa[*n] = ip;
      (*n)++;
    }
}

size_t orc_atm2x(ctl_t const *ctl, atm_t const *atm, double *x, int *iqa, int *ipa) {
  size_t n = 0;
  atm2x_help(atm, ctl->retp_zmin, ctl->retp_zmax, atm->p, IDXP, x, iqa, ipa, &n);
  atm2x_help(atm, ctl->rett_zmin, ctl->rett_zmax, atm->t, IDXT, x, iqa, ipa, &n);
  for (int ig = 0; ig < ctl->ng; ig++)
    atm2x_help(atm, ctl->retq_zmin[ig], ctl->retq_zmax[ig], atm->q[ig], IDXQ(ig), x, iqa, ipa, &n);
  for (int iw = 0; iw < ctl->nw; iw++)
    atm2x_help(atm, ctl->retk_zmin[iw], ctl->retk_zmax[iw], atm->k[iw], IDXK(iw), x, iqa, ipa, &n);
  return n;
}

static void x2atm_help(atm_t *atm, double zmin, double zmax, double *value, double const *x, size_t *n) {
  for (int ip = 0; ip < atm->np; ip++)
    if ((atm->z[ip] >= zmin) && (atm->z[ip] <= zmax)) {
      value[ip] = x[*n];
      (*n)++;
    }
}

static void x2atm(ctl_t const *ctl, double const *x, atm_t *atm) {
  size_t n = 0;
  x2atm_help(atm, ctl->retp_zmin, ctl->retp_zmax, atm->p, x, &n);
  x2atm_help(atm, ctl->rett_zmin, ctl->rett_zmax, atm->t, x, &n);
  for (int ig = 0; ig < ctl->ng; ig++) x2atm_help(atm, ctl->retq_zmin[ig], ctl->retq_zmax[ig], atm->q[ig], x, &n);
  for (int iw = 0; iw < ctl->nw; iw++) x2atm_help(atm, ctl->retk_zmin[iw], ctl->retk_zmax[iw], atm->k[iw], x, &n);
}

size_t orc_obs2y(ctl_t const *ctl, obs_t const *obs, double *y) {
  size_t m = 0;
  for (int ir = 0; ir < obs->nr; ir++)
    for (int id = 0; id < ctl->nd; id++)
      if (isfinite(obs->rad[ir][id])) {
        if (y) y[m] = obs->rad[ir][id];
        ++m;
      }
  return m;
}

/* k is row-major m x n; obs holds the unperturbed result on return. */
void orc_kernel(ctl_t const *ctl, atm_t *atm, obs_t *obs, orc_tbl_t const *tb, double *k, size_t m, size_t n) {
  int *iqa = (int *)malloc(sizeof(int) * (n + 1));
  double *x0 = (double *)malloc(sizeof(double) * (n + 1)), *x1 = (double *)malloc(sizeof(double) * (n + 1));
  double *yy0 = (double *)malloc(sizeof(double) * (m + 1)), *yy1 = (double *)malloc(sizeof(double) * (m + 1));
  atm_t *atm1 = (atm_t *)malloc(sizeof(atm_t));
  obs_t *obs1 = (obs_t *)malloc(sizeof(obs_t));
  orc_formod(ctl, atm, obs, tb);
  orc_atm2x(ctl, atm, x0, iqa, NULL);
  orc_obs2y(ctl, obs, yy0);
  memset(k, 0, sizeof(double) * m * n);
  for (size_t j = 0; j < n; j++) {
    double h;
    if (iqa[j] == IDXP) h = fmax(fabs(0.01 * x0[j]), 1e-7);
    else if (iqa[j] == IDXT) h = 1;
    else if (iqa[j] >= IDXQ(0) && iqa[j] < IDXQ(ctl->ng)) h = fmax(fabs(0.01 * x0[j]), 1e-15);
    else if (iqa[j] >= IDXK(0) && iqa[j] < IDXK(ctl->nw)) h = 1e-4;
    else { fprintf(stderr, "oracle: cannot set perturbation size\n"); exit(1); }
    memcpy(x1, x0, sizeof(double) * n);
    x1[j] = x1[j] + h;
    memcpy(atm1, atm, sizeof(atm_t));               /* copy_atm(..., 0) */
    memcpy(obs1, obs, sizeof(obs_t));               /* copy_obs(..., 0) */
    x2atm(ctl, x1, atm1);
    orc_formod(ctl, atm1, obs1, tb);
    orc_obs2y(ctl, obs1, yy1);
    for (size_t i = 0; i < m; i++) k[i * n + j] = (yy1[i] - yy0[i]) / h;
  }
  free(iqa); free(x0); free(x1); free(yy0); free(yy1); free(atm1); free(obs1);
}

/* ------------------------------------------------------------------------ */
/* Curtis-Godson means, jr_common.h:455-473 (compiled upstream only with        */
/* -DCURTIS_GODSON): sequential prefix sums per gas                             */
/* ------------------------------------------------------------------------ */
int orc_curtis_godson(ctl_t const *ctl, atm_t const *atm, double const geom[7], double *cgp, double *cgt, double *cgu) {
  pos_t *los = (pos_t *)malloc(sizeof(pos_t) * NLOS);
  double tsurf, tp[3];
  int const np = traceray(ctl, atm, geom, los, &tsurf, tp);
  for (int ig = 0; ig < ctl->ng; ig++) {
    double *p = cgp + (size_t)ig * NLOS, *t = cgt + (size_t)ig * NLOS, *u = cgu + (size_t)ig * NLOS;
    if (np > 0) {
      p[0] = los[0].u[ig] * los[0].p;
      t[0] = los[0].u[ig] * los[0].t;
      u[0] = los[0].u[ig];
    }
    for (int ip = 1; ip < np; ip++) {
      p[ip] = p[ip - 1] + los[ip].u[ig] * los[ip].p;
      t[ip] = t[ip - 1] + los[ip].u[ig] * los[ip].t;
      u[ip] = u[ip - 1] + los[ip].u[ig];
    }
    for (int ip = 0; ip < np; ip++) {
      p[ip] /= u[ip];
      t[ip] /= u[ip];
    }
  }
  free(los);
  return np;
}

/* ---- field-of-view convolution: formod_fov, src/jurassic.c:214-258 (shape already read: read_shape
 *      :1134-1150 is two columns, altitude offset and weight) -------------------------------------- */
#define LIN(x0, y0, x1, y1, x) ((y0) + ((x) - (x0)) * ((y1) - (y0)) / ((x1) - (x0)))   /* jurassic.h:81 */
int orc_formod_fov(ctl_t const *ctl, obs_t *obs, int n, double const *dz, double const *w) {
  enum { NFOV = 5 };                                    /* jurassic.h:175 */
  obs_t *obs2 = (obs_t *)malloc(sizeof(obs_t));
  if (!obs2) return -3;
  memcpy(obs2, obs, sizeof(obs_t));                     /* copy_obs(ctl, &obs2, obs, 0) */
  double (*rad)[JUR_ND] = malloc(sizeof(double) * (2 * NFOV + 1) * JUR_ND), (*tau)[JUR_ND] = malloc(sizeof(double) * (2 * NFOV + 1) * JUR_ND);
  if (!rad || !tau) { free(rad); free(tau); free(obs2); return -3; }
  double z[2 * NFOV + 1];
  for (int ir = 0; ir < obs->nr; ir++) {
    int nz = 0;
    int const first = ir - NFOV > 0 ? ir - NFOV : 0, last = ir + 1 + NFOV < obs->nr ? ir + 1 + NFOV : obs->nr;
    for (int ir2 = first; ir2 < last; ir2++)
      if (obs->time[ir2] == obs->time[ir]) {
        z[nz] = obs2->vpz[ir2];
        for (int id = 0; id < ctl->nd; id++) {
          rad[nz][id] = obs2->rad[ir2][id];
          tau[nz][id] = obs2->tau[ir2][id];
        }
        nz++;
      }
    if (nz < 2) { free(obs2); free(rad); free(tau); return -1; }   /* ERRMSG("Cannot apply FOV convolution!") */
    double wsum = 0;
    for (int id = 0; id < ctl->nd; id++) {
      obs->rad[ir][id] = 0;
      obs->tau[ir][id] = 0;
    }
    for (int i = 0; i < n; i++) {
      double const zfov = obs->vpz[ir] + dz[i];
      int const idx = locate(z, nz, zfov);
      for (int id = 0; id < ctl->nd; id++) {
        obs->rad[ir][id] += w[i] * LIN(z[idx], rad[idx][id], z[idx + 1], rad[idx + 1][id], zfov);
        obs->tau[ir][id] += w[i] * LIN(z[idx], tau[idx][id], z[idx + 1], tau[idx + 1][id], zfov);
      }
      wsum += w[i];
    }
    for (int id = 0; id < ctl->nd; id++) {
      obs->rad[ir][id] /= wsum;
      obs->tau[ir][id] /= wsum;
    }
  }
  free(obs2); free(rad); free(tau);
  return 0;
}

/* number of OpenMP threads the next calls use (0: leave as is); returns the current maximum */

/* ------------------------------------------------------------------------ */
/* atmosphere regridding, jurassic.c:675-804 (intpol_atm, intpol_atm_geo,     */
/* intpol_atm_1d / 2d / 3d): the step in front of the path for atmospheres    */
/* that are not one profile.  Returns 0, or the negative number of the         */
/* upstream ERRMSG that would have ended the process.                          */
/* ------------------------------------------------------------------------ */
#define DIST2(a, b) ((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]))

static void intpol_atm_1d(ctl_t const *ctl, atm_t const *atm, int idx0, int n, double z0, double *p, double *t, double *q,
                          double *k) {
  int const ip = idx0 + locate(&atm->z[idx0], n, z0);
  *p = eip(atm->z[ip], atm->p[ip], atm->z[ip + 1], atm->p[ip + 1], z0);
  *t = lip(atm->z[ip], atm->t[ip], atm->z[ip + 1], atm->t[ip + 1], z0);
  for (int ig = 0; ig < ctl->ng; ig++) q[ig] = lip(atm->z[ip], atm->q[ig][ip], atm->z[ip + 1], atm->q[ig][ip + 1], z0);
  for (int iw = 0; iw < ctl->nw; iw++) k[iw] = lip(atm->z[ip], atm->k[iw][ip], atm->z[ip + 1], atm->k[iw][ip + 1], z0);
}

int orc_intpol_atm(ctl_t const *ctl, atm_t *dest, atm_t const *src) {
  static double x1[JUR_NP][3];
  static int idx[JUR_NP], nz[JUR_NP];
  int nx = 0;
  if (ctl->ip == 2) {                                   /* profiles of a satellite track, :716-735 */
    double lat1 = -999, lon1 = -999;
    for (int ip = 0; ip < src->np; ip++) {
      if ((src->lon[ip] != lon1) || (src->lat[ip] != lat1)) {
        if ((++nx) > JUR_NP) return -1;
        nz[nx - 1] = 0;
        lon1 = src->lon[ip];
        lat1 = src->lat[ip];
        geo2cart(0, lon1, lat1, x1[nx - 1]);
        idx[nx - 1] = ip;
      }
      ++nz[nx - 1];
    }
    for (int ix = 0; ix < nx; ix++) {
      if (nz[ix] <= 1) return -2;
      if ((ix > 0) && (fabs(src->lat[idx[ix - 1]] - src->lat[idx[ix]]) > 10)) return -3;
    }
  } else if (ctl->ip == 3) {
    for (int ip = 0; ip < src->np; ip++) geo2cart(0, src->lon[ip], src->lat[ip], x1[ip]);
  } else if (ctl->ip != 1) return -4;
  for (int id = 0; id < dest->np; id++) {
    double const z0 = dest->z[id], lon0 = dest->lon[id], lat0 = dest->lat[id];
    double q[JUR_NG], k[JUR_NW], *p = &dest->p[id], *t = &dest->t[id];
    if (ctl->ip == 1) intpol_atm_1d(ctl, src, 0, src->np, z0, p, t, q, k);
    else if (ctl->ip == 2) {                            /* :737-766 */
      double dhmin0 = 1e99, dhmin1 = 1e99, dlat = 10, k0[JUR_NW], k1[JUR_NW], p0, p1, q0[JUR_NG], q1[JUR_NG], r, t0, t1, x0[3];
      int ix0 = 0, ix1 = 0;
      geo2cart(0, lon0, lat0, x0);
      for (int ix = 0; ix < nx; ix++)
        if (fabs(lat0 - src->lat[idx[ix]]) <= dlat) {
          double const dh = DIST2(x0, x1[ix]);
          if (dh <= dhmin0) { dhmin1 = dhmin0; ix1 = ix0; dhmin0 = dh; ix0 = ix; }
          else if (dh <= dhmin1) { dhmin1 = dh; ix1 = ix; }
        }
      intpol_atm_1d(ctl, src, idx[ix0], nz[ix0], z0, &p0, &t0, q0, k0);
      intpol_atm_1d(ctl, src, idx[ix1], nz[ix1], z0, &p1, &t1, q1, k1);
      double const x2 = DIST2(x1[ix0], x1[ix1]);
      double const x = sqrt(x2);
      double const r0 = (dhmin0 - dhmin1 + x2) / (2 * x);
      double const r1 = x - r0;
      if (r0 <= 0) r = 0;
      else r = (r1 <= 0) ? 1 : r0 / (r0 + r1);
      *p = (1 - r) * p0 + r * p1;
      *t = (1 - r) * t0 + r * t1;
      for (int ig = 0; ig < ctl->ng; ig++) q[ig] = (1 - r) * q0[ig] + r * q1[ig];
      for (int iw = 0; iw < ctl->nw; iw++) k[iw] = (1 - r) * k0[iw] + r * k1[iw];
    } else {                                            /* :768-804 */
      double const rm2 = ctl->cx * ctl->cx;
      double wsum = 0, x0[3];
      *p = *t = 0.;
      for (int ig = 0; ig < ctl->ng; ig++) q[ig] = 0;
      for (int iw = 0; iw < ctl->nw; iw++) k[iw] = 0;
      for (int ip = 0; ip < src->np; ip++) {
        double const dz = fabs(src->z[ip] - z0);
        if (dz >= ctl->cz) continue;
        if (fabs(src->lat[ip] - lat0) * 111.13 >= ctl->cx) continue;
        geo2cart(0, lon0, lat0, x0);
        double const dx2 = DIST2(x0, x1[ip]);
        if (dx2 >= rm2) continue;
        double const w = (1 - dz / ctl->cz) * (rm2 - dx2) / (rm2 + dx2);
        wsum += w;
        *p += w * src->p[ip];
        *t += w * src->t[ip];
        for (int ig = 0; ig < ctl->ng; ig++) q[ig] += w * src->q[ig][ip];
        for (int iw = 0; iw < ctl->nw; iw++) k[iw] += w * src->k[iw][ip];
      }
      if (wsum >= 1e-6) {
        *p /= wsum;
        *t /= wsum;
        for (int ig = 0; ig < ctl->ng; ig++) q[ig] /= wsum;
        for (int iw = 0; iw < ctl->nw; iw++) k[iw] /= wsum;
      } else {
        *p = *t = NAN;
        for (int ig = 0; ig < ctl->ng; ig++) q[ig] = NAN;
        for (int iw = 0; iw < ctl->nw; iw++) k[iw] = NAN;
      }
    }
    for (int ig = 0; ig < ctl->ng; ig++) dest->q[ig][id] = q[ig];
    for (int iw = 0; iw < ctl->nw; iw++) dest->k[iw][id] = k[iw];
  }
  return 0;
}

int orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

/* the oracle's own constants (oracle_constants.h) as compiled: C1, C2, P0, RE, N_A, k_B, R */
void orc_constants(double out[7]) {
  out[0] = ORC_C1; out[1] = ORC_C2; out[2] = ORC_P0; out[3] = ORC_RE;
  out[4] = ORC_AVOGADRO; out[5] = ORC_BOLTZMANN; out[6] = ORC_MOLAR_GAS;
}
