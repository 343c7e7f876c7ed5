/* jur_multi.c -- one process, several GPUs: the rays of a call are dealt to several models (one per device), each
 * works on a contiguous range of the caller's arrays, nothing is gathered.
 *
 * The reference's counterpart is the device loop inside formod_GPU (GPUdrivers.cu:344-358, `omp parallel for
 * num_threads(numDevices)`), which hands the SAME package to every device; here the rays are partitioned (the path
 * has no exchange step between rays: SURVEY.md section 8e).  Ranges are contiguous -- results land in place, pinned
 * caller arrays move at PCIe speed without staging -- and their boundaries are chosen so that the estimated number of
 * line-of-sight points, not the number of rays, is equal: a tangent-height scan in its natural order has 122 ... 393
 * points per ray, and equal ray counts would leave the device with the high tangent altitudes idle.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "jur_internal.h"

#define HIPCHK(call)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      jur_set_error("HIP error %d (%s) at %s:%d", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); \
      return JUR_EHIP;                                                                     \
    }                                                                                      \
  } while (0)

/* ---- cost of a ray: how many points the ray tracer will put on it ---------------------------------------------
 * Straight-line geometry (refraction moves the count by a few points): with s the path coordinate from the point of
 * closest approach and b the distance of that point from the Earth's centre, r^2 = b^2 + s^2 and the tracer steps
 * ds = min(RAYDS, RAYDZ / |cos a|), cos a = s / r (jr_common.h:625-634).  Points between s1 and s2:
 *   N = integral of max(1 / RAYDS, |s| / (RAYDZ r)) ds,   and   integral of s / r ds = r. */
static double steps_from_tangent(double a, double b, double rayds, double raydz) {   /* a >= 0 */
  if (!(raydz > 0)) return a / rayds;
  double const c = raydz / rayds;
  if (c >= 1) return a / rayds;
  double const sc = b * c / sqrt(1 - c * c);               /* where the altitude step takes over */
  if (a <= sc) return a / rayds;
  return sc / rayds + (sqrt(b * b + a * a) - sqrt(b * b + sc * sc)) / raydz;
}

static void geo2cart_host(double z, double lon, double lat, double x[3]) {
  double const r = z + JUR_RE, d2r = M_PI / 180., cl = cos(lat * d2r);
  x[0] = r * cl * cos(lon * d2r); x[1] = r * cl * sin(lon * d2r); x[2] = r * sin(lat * d2r);
}

static double ray_cost(double const *const geom[7], long i, double rayds, double raydz, double zmin, double zmax) {
  double const obsz = geom[1][i], vpz = geom[4][i];
  if (obsz < zmin || vpz > zmax - 0.001) return 0;         /* never enters (jr_common.h:599-603) */
  double xo[3], xv[3], e[3];
  geo2cart_host(obsz, geom[2][i], geom[3][i], xo);
  geo2cart_host(vpz, geom[5][i], geom[6][i], xv);
  double n = 0, so = 0, ro2 = 0;
  for (int k = 0; k < 3; k++) { e[k] = xv[k] - xo[k]; n += e[k] * e[k]; }
  n = sqrt(n);
  if (!(n > 0)) return 0;
  for (int k = 0; k < 3; k++) { so += xo[k] * e[k] / n; ro2 += xo[k] * xo[k]; }
  double const b2 = fmax(ro2 - so * so, 0.), b = sqrt(b2);
  double const rt = JUR_RE + zmax, rb = JUR_RE + zmin;
  if (b >= rt) return 0;
  double const st = sqrt(rt * rt - b2);
  double s1 = so > -st ? so : -st, s2 = st;                /* from the observer or the entry point to the exit */
  if (b < rb) {                                            /* the line dips below the lowest level: the ray ends there */
    double const sb = sqrt(rb * rb - b2);
    if (s1 < -sb) s2 = -sb;
  }
  if (s2 <= s1) return 0;
  double const f1 = (s1 < 0 ? -1 : 1) * steps_from_tangent(fabs(s1), b, rayds, raydz);
  double const f2 = (s2 < 0 ? -1 : 1) * steps_from_tangent(fabs(s2), b, rayds, raydz);
  double const np = f2 - f1 + 2;
  return np < JUR_NLOS ? np : JUR_NLOS;
}

/* the estimate for nr rays, without a model or a GPU: the tracer's step sizes and the altitude range of the atmosphere */
int jur_estimate_los_points(double rayds, double raydz, double zmin, double zmax, long nr, double const *const geom[7], double *points) {
  if (!(rayds > 0) || !(zmax > zmin) || nr < 0 || (nr > 0 && (!geom || !points))) { jur_set_error("estimate_los_points: bad arguments"); return JUR_EINVAL; }
  for (long i = 0; i < nr; i++) points[i] = ray_cost(geom, i, rayds, raydz, zmin, zmax);
  return JUR_OK;
}

/* bounds[0 .. nparts]: rays [bounds[k], bounds[k+1]) go to part k; equal estimated points + a per-ray constant */
int jur_balance_rays(double rayds, double raydz, double zmin, double zmax, long nr, double const *const geom[7], int nparts, long *bounds) {
  if (!(rayds > 0) || !(zmax > zmin) || nr < 0 || nparts < 1 || !bounds) { jur_set_error("balance_rays: bad arguments"); return JUR_EINVAL; }
  bounds[0] = 0; bounds[nparts] = nr;
  if (nparts == 1 || nr == 0) { for (int k = 1; k < nparts; k++) bounds[k] = nr; return JUR_OK; }
  double const per_ray = 8;                                /* set-up, epilogue: a ray costs something even if it is short */
  double total = 0;
  for (long i = 0; i < nr; i++) total += ray_cost(geom, i, rayds, raydz, zmin, zmax) + per_ray;
  double acc = 0;
  int k = 1;
  for (long i = 0; i < nr && k < nparts; i++) {
    acc += ray_cost(geom, i, rayds, raydz, zmin, zmax) + per_ray;
    while (k < nparts && acc >= total * k / nparts) bounds[k++] = i + 1;
  }
  while (k < nparts) bounds[k++] = nr;
  return JUR_OK;
}

int jur_multi_balance(jur_model_t const *m, long nr, double const *const geom[7], int nparts, long *bounds) {
  if (!m) { jur_set_error("multi_balance: bad arguments"); return JUR_EINVAL; }
  double rayds, raydz, zmin, zmax;
  jur_model_cost_params(m, &rayds, &raydz, &zmin, &zmax);
  if (!(zmax > zmin)) { jur_set_error("multi_balance: the model has no atmosphere yet (jur_model_set_atm)"); return JUR_EINVAL; }
  return jur_balance_rays(rayds, raydz, zmin, zmax, nr, geom, nparts, bounds);
}

static int check_models(jur_model_t *const models[], int nmodel, char const *who) {
  if (!models || nmodel < 1 || nmodel > 64) { jur_set_error("%s: 1 .. 64 models", who); return JUR_EINVAL; }
  for (int k = 0; k < nmodel; k++) {
    if (!models[k]) { jur_set_error("%s: model %d is NULL", who, k); return JUR_EINVAL; }
    if (jur_model_nd(models[k]) != jur_model_nd(models[0])) { jur_set_error("%s: model %d has another channel count", who, k); return JUR_EINVAL; }
    for (int j = 0; j < k; j++)
      if (models[j] == models[k]) { jur_set_error("%s: model %d is listed twice (one model per share: create one per device, or several on one)", who, k); return JUR_EINVAL; }
  }
  return JUR_OK;
}

/* ---- host arrays: one host thread and one stream set per device ---------------------------------------------- */
typedef struct {
  jur_model_t *m;
  long lo, hi;
  double const *const *geom;
  double *rad, *tau;
  double *const *tp;
  int *np_out;
  int nd, rc;
  char err[256];
} job_t;

static void *host_worker(void *arg) {
  job_t *j = (job_t *)arg;
  long const n = j->hi - j->lo;
  j->rc = JUR_OK;
  if (n <= 0) return NULL;
  double const *g[7];
  double *tp[3];
  for (int k = 0; k < 7; k++) g[k] = j->geom[k] + j->lo;
  for (int k = 0; k < 3; k++) tp[k] = j->tp[k] + j->lo;
  j->rc = jur_formod_host(j->m, n, g, j->rad + (size_t)j->lo * j->nd, j->tau + (size_t)j->lo * j->nd, tp,
                          j->np_out ? j->np_out + j->lo : NULL);
  if (j->rc) { strncpy(j->err, jur_last_error(), sizeof j->err - 1); j->err[sizeof j->err - 1] = 0; }   /* the text is per thread */
  return NULL;
}

int jur_formod_host_multi(jur_model_t *const models[], int nmodel, long nr, double const *const geom[7], double *rad,
                          double *tau, double *const tp[3], int *np_out) {
  int rc = check_models(models, nmodel, "formod_host_multi");
  if (rc) return rc;
  if (nr < 0) { jur_set_error("formod_host_multi: bad ray count"); return JUR_EINVAL; }
  if (nr == 0) return JUR_OK;
  if (nmodel == 1) return jur_formod_host(models[0], nr, geom, rad, tau, tp, np_out);
  long bounds[65];
  if ((rc = jur_multi_balance(models[0], nr, geom, nmodel, bounds))) return rc;
  job_t job[64];
  pthread_t th[64];
  int started[64];
  for (int k = 0; k < nmodel; k++) {
    job[k] = (job_t){models[k], bounds[k], bounds[k + 1], geom, rad, tau, tp, np_out, jur_model_nd(models[0]), JUR_OK, {0}};
    started[k] = 0;
  }
  for (int k = 1; k < nmodel; k++)                         /* shares 1 .. on threads of their own, share 0 here */
    started[k] = (job[k].hi > job[k].lo) && pthread_create(&th[k], NULL, host_worker, &job[k]) == 0;
  host_worker(&job[0]);
  for (int k = 1; k < nmodel; k++) {
    if (started[k]) pthread_join(th[k], NULL);
    else host_worker(&job[k]);                             /* no thread to be had (or an empty share): here, afterwards */
  }
  for (int k = 0; k < nmodel; k++) {
    /* a line of sight beyond NLOS points is reported after every share has been computed, as the single-model call does */
    if (job[k].rc && (rc == JUR_OK || rc == JUR_ENLOS)) { rc = job[k].rc; jur_set_error("model %d (rays %ld .. %ld): %s", k, job[k].lo, job[k].hi, job[k].err); }
  }
  return rc;
}

/* ---- device arrays on models[0]'s GPU --------------------------------------------------------------------------
 * Share 0 is computed in place on the caller's stream.  Every other share travels to its model's device (peer copies
 * of the 7 geometry rows and of the input radiances on that model's own stream), is computed there, and its radiances,
 * transmittances, tangent points and point counts travel back into the caller's arrays; the caller's stream then waits
 * for all of them.  Nothing is synchronised with the host. */
int jur_formod_device_multi(jur_model_t *const models[], int nmodel, long nr, long const *bounds_in, double const *d_geom,
                            double *d_rad, double *d_tau, double *d_tp, int *d_np, int *d_status, void *stream) {
  int rc = check_models(models, nmodel, "formod_device_multi");
  if (rc) return rc;
  if (nr < 0) { jur_set_error("formod_device_multi: bad ray count"); return JUR_EINVAL; }
  if (nr == 0) return JUR_OK;
  long bounds[65];
  for (int k = 0; k <= nmodel; k++) bounds[k] = bounds_in ? bounds_in[k] : nr * k / nmodel;
  if (bounds[0] != 0 || bounds[nmodel] != nr) { jur_set_error("formod_device_multi: bounds must run from 0 to nr"); return JUR_EINVAL; }
  for (int k = 0; k < nmodel; k++)
    if (bounds[k + 1] < bounds[k]) { jur_set_error("formod_device_multi: bounds must not decrease"); return JUR_EINVAL; }
  int const d0 = jur_model_device(models[0]), nd = jur_model_nd(models[0]);
  hipStream_t const s0 = (hipStream_t)stream;
  hipEvent_t ev_in = NULL, ev_k[64];
  int nev = 0;
  if (nmodel > 1) {
    HIPCHK(hipSetDevice(d0));
    HIPCHK(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev_in, s0));                     /* the inputs are ready when the caller's stream gets here */
  }
  for (int k = 1; k < nmodel && rc == JUR_OK; k++) {
    long const lo = bounds[k], n = bounds[k + 1] - lo;
    if (n <= 0) continue;
    jur_model_t *const m = models[k];
    int const dk = jur_model_device(m);
    hipStream_t const sk = (hipStream_t)jur_model_stream(m);
    double *io = NULL;
    int *io_np = NULL;
    if ((rc = jur_model_io(m, n, &io, &io_np))) break;
    size_t const N = (size_t)n, nrd = N * nd;
    double *const g = io, *const r = g + 7 * N, *const t = r + nrd, *const p = t + nrd;
    int *const st = jur_model_status_word(m);
    hipError_t e = hipSetDevice(dk);
    if (e == hipSuccess) e = hipStreamWaitEvent(sk, ev_in, 0);
    for (int f = 0; f < 7 && e == hipSuccess; f++)
      e = hipMemcpyPeerAsync(g + f * N, dk, d_geom + (size_t)f * nr + lo, d0, sizeof(double) * N, sk);
    if (e == hipSuccess) e = hipMemcpyPeerAsync(r, dk, d_rad + (size_t)lo * nd, d0, sizeof(double) * nrd, sk);
    if (e == hipSuccess) e = hipMemsetAsync(st, 0, sizeof(int), sk);
    if (e != hipSuccess) { jur_set_error("formod_device_multi: model %d: %s", k, hipGetErrorString(e)); rc = JUR_EHIP; break; }
    if ((rc = jur_formod_device(m, n, g, r, t, p, io_np, st, sk))) break;
    e = hipMemcpyPeerAsync(d_rad + (size_t)lo * nd, d0, r, dk, sizeof(double) * nrd, sk);
    if (e == hipSuccess) e = hipMemcpyPeerAsync(d_tau + (size_t)lo * nd, d0, t, dk, sizeof(double) * nrd, sk);
    for (int f = 0; f < 3 && e == hipSuccess; f++)
      e = hipMemcpyPeerAsync(d_tp + (size_t)f * nr + lo, d0, p + f * N, dk, sizeof(double) * N, sk);
    if (e == hipSuccess && d_np) e = hipMemcpyPeerAsync(d_np + lo, d0, io_np, dk, sizeof(int) * N, sk);
    if (e == hipSuccess && d_status) e = hipMemcpyPeerAsync(d_status + k, d0, st, dk, sizeof(int), sk);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_k[nev], hipEventDisableTiming);
    if (e == hipSuccess) { nev++; e = hipEventRecord(ev_k[nev - 1], sk); }
    if (e != hipSuccess) { jur_set_error("formod_device_multi: model %d: %s", k, hipGetErrorString(e)); rc = JUR_EHIP; break; }
  }
  if (rc == JUR_OK && bounds[1] > 0) {
    /* share 0 in place; its status word is d_status[0] */
    rc = jur_formod_device_ld(models[0], bounds[1], d_geom, nr, d_rad, d_tau, d_tp, nr, d_np, d_status, s0);
  }
  if (nmodel > 1) {
    if (hipSetDevice(d0) != hipSuccess) rc = rc ? rc : JUR_EHIP;
    for (int i = 0; i < nev; i++) {
      if (hipStreamWaitEvent(s0, ev_k[i], 0) != hipSuccess && rc == JUR_OK) { jur_set_error("formod_device_multi: cannot wait for a share"); rc = JUR_EHIP; }
      (void)hipEventDestroy(ev_k[i]);                      /* (released once the recorded work has completed) */
    }
    (void)hipEventDestroy(ev_in);
  }
  return rc;
}

/* the same atmosphere on every model */
int jur_models_set_atm(jur_model_t *const models[], int nmodel, atm_t const *atm) {
  int rc = check_models(models, nmodel, "models_set_atm");
  for (int k = 0; k < nmodel && rc == JUR_OK; k++) rc = jur_model_set_atm(models[k], atm);
  return rc;
}
