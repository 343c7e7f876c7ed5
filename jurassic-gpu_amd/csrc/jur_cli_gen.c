/* jur_cli_gen.c -- the input generators of the reference's example scripts, one executable each
 * (built three times with -DJUR_TOOL_LIMB / _NADIR / _CLIMATOLOGY):
 *
 *   limb        <ctl> <obs> [OBSZ T0 T1 DT Z0 Z1 DZ]        tangent-height scan     src/limb.c:27-69
 *   nadir       <ctl> <obs> [OBSZ T0 T1 DT LAT0 LAT1 DLAT]  sub-satellite sweep     src/nadir.c:27-63
 *   climatology <ctl> <atm> [T0 T1 DT Z0 Z1 DZ RAND]        mid-latitude profiles   src/climatology.c:28-83,
 *                                                                                    src/jurassic.c:79-140
 * Host-only; they write the text files `formod` reads (jur_textio.c).  The loops accumulate
 * `t += dt`, `z += dz` in floating point exactly as upstream, so the row counts agree.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <strings.h>
#include "jur_textio.h"

#if defined(JUR_TOOL_LIMB) || defined(JUR_TOOL_NADIR)

/* Both observation generators are one sweep -- time outside, a scan coordinate inside -- over keys read from the
 * control file / command line; they differ in the keys' names and defaults, in what a row's scan coordinate means,
 * and in when the row count is refused.  The sweeps accumulate `t += dt`, `x += dx` as upstream does. */
enum { K_OBSZ, K_T0, K_T1, K_DT, K_X0, K_X1, K_DX, K_COUNT };
typedef struct { char const *key, *dflt; } genkey_t;

#if defined(JUR_TOOL_LIMB)
/* scan coordinate: tangent altitude; the observer stands on the view point's meridian (limb.c:49-59) */
static genkey_t const g_keys[K_COUNT] = {{"OBSZ", "780"}, {"T0", "0"}, {"T1", "0"}, {"DT", "1"}, {"Z0", "3"}, {"Z1", "68"}, {"DZ", "1"}};
static void fill_row(obs_t *obs, int i, double obsz, double z) {
  obs->vpz[i] = z;
  obs->vplat[i] = 180 / M_PI * acos((JUR_RE + z) / (JUR_RE + obsz));
}
/* limb.c counts every row and refuses afterwards, naming the count */
static void check_count(int nr, int final) { if (final && nr > JUR_NR) DIE("Too many rays! found %d, max is %d", nr, JUR_NR); }
#else
/* scan coordinate: latitude of the view point on the ground below the observer's meridian (nadir.c:51-58) */
static genkey_t const g_keys[K_COUNT] = {{"OBSZ", "700"}, {"T0", "0"}, {"T1", "0"}, {"DT", "1"}, {"LAT0", "-8.01"}, {"LAT1", "8.01"}, {"DLAT", "0.18"}};
static void fill_row(obs_t *obs, int i, double obsz, double lat) { (void)obsz; obs->vplat[i] = lat; }
/* nadir.c refuses as soon as the last slot is taken */
static void check_count(int nr, int final) { if (!final && nr >= JUR_NR) DIE("Too many rays!"); }
#endif

int main(int argc, char *argv[]) {
  if (argc < 3) DIE("Give parameters: <ctl> <obs>");
  ctl_t *ctl = (ctl_t *)calloc(1, sizeof(ctl_t));
  obs_t *obs = (obs_t *)calloc(1, sizeof(obs_t));
  if (!ctl || !obs) DIE("Out of memory!");
  read_ctl(argc, argv, ctl);
  double val[K_COUNT];
  for (int k = 0; k < K_COUNT; k++) val[k] = scan_ctl(argc, argv, g_keys[k].key, -1, g_keys[k].dflt, NULL);
  int nr = 0;
  for (double t = val[K_T0]; t <= val[K_T1]; t += val[K_DT])
    for (double x = val[K_X0]; x <= val[K_X1]; x += val[K_DX]) {
      if (nr < JUR_NR) {
        obs->time[nr] = t;
        obs->obsz[nr] = val[K_OBSZ];
        fill_row(obs, nr, val[K_OBSZ], x);
      }
      check_count(++nr, 0);
    }
  check_count(nr, 1);
  obs->nr = nr;
  write_obs(argv[2], ctl, obs);
  free(ctl); free(obs);
  return EXIT_SUCCESS;
}

#elif defined(JUR_TOOL_CLIMATOLOGY)

/* clim.bin: repeated { char name[8]; double v[121] }, packed by tools/extract_clim.py */
#define CLIM_NZ 121
typedef struct { char name[8]; double v[CLIM_NZ]; } clim_profile_t;
extern const clim_profile_t jur_clim_blob[];
extern const char jur_clim_blob_end[];
__asm__(".section .rodata\n"
        ".balign 8\n"
        ".global jur_clim_blob\n"
        "jur_clim_blob:\n"
        ".incbin \"" CLIM_BLOB_PATH "\"\n"
        ".global jur_clim_blob_end\n"
        "jur_clim_blob_end:\n"
        ".previous\n");

static double const *clim_profile(char const *name) {
  size_t const n = (size_t)(jur_clim_blob_end - (char const *)jur_clim_blob) / sizeof(clim_profile_t);
  for (size_t i = 0; i < n; i++)
    if (0 == strcasecmp(jur_clim_blob[i].name, name)) return jur_clim_blob[i].v;
  return NULL;
}

/* bracket of x in an ascending or descending axis (jr_common.h:87-104) */
static int bracket(double const *xx, int n, double x) {
  int lo = 0, hi = n - 1;
  if (xx[0] < xx[n - 1]) {
    while (hi > lo + 1) { int const m = (lo + hi) / 2; if (xx[m] > x) hi = m; else lo = m; }
  } else {
    while (hi > lo + 1) { int const m = (lo + hi) / 2; if (xx[m] <= x) hi = m; else lo = m; }
  }
  return lo;
}

static double lin(double x0, double y0, double x1, double y1, double x) { return y0 + (x - x0) * (y1 - y0) / (x1 - x0); }
static double expi(double x0, double y0, double x1, double y1, double x) {
  return (y0 > 0 && y1 > 0) ? y0 * exp(log(y1 / y0) / (x1 - x0) * (x - x0)) : lin(x0, y0, x1, y1, x);
}

/* MT19937 (Matsumoto & Nishimura 1998) with the 2002 initialisation -- the generator behind GSL's
 * gsl_rng_default; seed 0 means 4357 there, GSL_RNG_SEED overrides it.  Only RAND=1 uses it; GSL is
 * not available in this build environment, so the stream is not checked against GSL's. */
static uint32_t mt[624];
static int mti = 625;
static void mt_seed(uint32_t s) {
  if (s == 0) s = 4357;
  mt[0] = s;
  for (mti = 1; mti < 624; mti++) mt[mti] = 1812433253u * (mt[mti - 1] ^ (mt[mti - 1] >> 30)) + (uint32_t)mti;
}
static uint32_t mt_next(void) {
  if (mti >= 624) {
    for (int k = 0; k < 624; k++) {
      uint32_t const y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
      mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    mti = 0;
  }
  uint32_t y = mt[mti++];
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}
static double uniform_pos(void) { double x; do x = mt_next() / 4294967296.0; while (x == 0); return x; }

int main(int argc, char *argv[]) {
  if (argc < 3) DIE("Give parameters: <ctl> <atm>");
  ctl_t *ctl = (ctl_t *)calloc(1, sizeof(ctl_t));
  atm_t *atm = (atm_t *)calloc(1, sizeof(atm_t));
  if (!ctl || !atm) DIE("Out of memory!");
  read_ctl(argc, argv, ctl);
  double const t0 = scan_ctl(argc, argv, "T0", -1, "0", NULL);
  double const t1 = scan_ctl(argc, argv, "T1", -1, "0", NULL);
  double const dt = scan_ctl(argc, argv, "DT", -1, "1", NULL);
  double const z0 = scan_ctl(argc, argv, "Z0", -1, "0", NULL);
  double const z1 = scan_ctl(argc, argv, "Z1", -1, "90", NULL);
  double const dz = scan_ctl(argc, argv, "DZ", -1, "1", NULL);
  int const randomise = (int)scan_ctl(argc, argv, "RAND", -1, "0", NULL);

  for (double t = t0; t <= t1; t += dt)
    for (double z = z0; z <= z1; z += dz) {
      atm->time[atm->np] = t;
      atm->z[atm->np] = z;
      if (++atm->np >= JUR_NP) DIE("Too many atmospheric grid points!");
    }

  double const *zc = clim_profile("z"), *pre = clim_profile("pre"), *tem = clim_profile("tem");
  double const *q[JUR_NG] = {NULL};
  int ig_co2 = -1;
  for (int ig = 0; ig < ctl->ng; ig++) {
    if (0 == strcasecmp(ctl->emitter[ig], "CO2")) { if (ig_co2 < 0) ig_co2 = ig; continue; }
    if (0 == strcasecmp(ctl->emitter[ig], "z") || 0 == strcasecmp(ctl->emitter[ig], "pre") ||
        0 == strcasecmp(ctl->emitter[ig], "tem")) continue;
    q[ig] = clim_profile(ctl->emitter[ig]);
    if (!q[ig]) printf("# Warning! no climatology table for found emitter %s\n", ctl->emitter[ig]);
  }
  if (!ctl->checkmode)
    for (int ip = 0; ip < atm->np; ip++) {
      double const z = atm->z[ip];
      int const iz = bracket(zc, CLIM_NZ, z);
      atm->p[ip] = expi(zc[iz], pre[iz], zc[iz + 1], pre[iz + 1], z);
      atm->t[ip] = lin(zc[iz], tem[iz], zc[iz + 1], tem[iz + 1], z);
      for (int ig = 0; ig < ctl->ng; ig++) atm->q[ig][ip] = q[ig] ? lin(zc[iz], q[ig][iz], zc[iz + 1], q[ig][iz + 1], z) : 0;
      /* CO2: linear trend in time, 371.79 ppm at t = 63158400 s, +2.026 ppm per year */
      if (ig_co2 >= 0) atm->q[ig_co2][ip] = 371.789948e-6 + 2.026214e-6 * (atm->time[ip] - 63158400.) / 31557600.;
      for (int iw = 0; iw < ctl->nw; iw++) atm->k[iw][ip] = 0;
    }

  if (randomise) {       /* one (dp, dT) per profile: p*(1+dp), dp in [-5,5) %; T+dT, dT in [-30,30) K */
    char const *seed = getenv("GSL_RNG_SEED");
    mt_seed(seed ? (uint32_t)strtoul(seed, NULL, 0) : 0);
    double dpress = 0, dtemp = 0;
    for (int ip = 0; ip < atm->np; ip++) {
      if (ip == 0 || atm->time[ip - 1] != atm->time[ip]) {
        dpress = 0.05 - 0.1 * uniform_pos();
        dtemp = 30. - 60. * uniform_pos();
      }
      atm->p[ip] *= (1.0 + dpress);
      atm->t[ip] += dtemp;
    }
  }
  write_atm(argv[2], ctl, atm);
  free(ctl); free(atm);
  return EXIT_SUCCESS;
}

#else
#error "build with -DJUR_TOOL_LIMB, -DJUR_TOOL_NADIR or -DJUR_TOOL_CLIMATOLOGY"
#endif
