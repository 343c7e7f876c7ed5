/* jur_intpol.c -- regridding of an atmosphere, the step in front of the forward model for atmospheres that are
 * not one profile.
 *
 * Reference: intpol_atm / intpol_atm_geo / intpol_atm_1d / _2d / _3d (src/jurassic.c:675-804; jurassic.h:581-633).
 * The hot path itself only takes profiles (upstream asserts IP == 1 there, jr_common.h:573); these functions bring
 * a satellite-track (IP = 2) or point-cloud (IP = 3) atmosphere onto the points of another one first.
 *
 * Host part here: what upstream caches in function statics on the first call (profile starts and lengths, Cartesian
 * positions of source points, jurassic.c:716-735, 770-772) and the positions of the destination points; the
 * searches and the interpolation of every destination point run in jur_intpol_kernel (one lane per point).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "jur_internal.h"

static void geo2cart0(double lon, double lat, double x[3]) {   /* geo2cart(0, lon, lat, x), jr_common.h:494-500 */
  double const radius = 0 + JUR_RE, deg = M_PI / 180, clat = cos(lat * deg);
  x[0] = radius * clat * cos(lon * deg);
  x[1] = radius * clat * sin(lon * deg);
  x[2] = radius * sin(lat * deg);
}

int jur_intpol_atm(ctl_t const *ctl, atm_t *dest, atm_t const *src, int device) {
  if (!ctl || !dest || !src) { jur_set_error("jur_intpol_atm: null argument"); return JUR_EINVAL; }
  int const ng = ctl->ng, nw = ctl->nw, ns = src->np, nd_ = dest->np, ip = ctl->ip;
  if (ip < 1 || ip > 3) { jur_set_error("Unknown interpolation method, check IP!"); return JUR_EINVAL; }
  if (ns < 2 || ns > JUR_NP || nd_ < 0 || nd_ > JUR_NP || ng < 0 || ng > JUR_NG || nw < 0 || nw > JUR_NW) {
    jur_set_error("jur_intpol_atm: sizes out of range");
    return JUR_EINVAL;
  }
  if (nd_ == 0) return JUR_OK;
  size_t const nrow_s = 5 + (size_t)ng + nw, nrow_o = 2 + (size_t)ng + nw;
  int nx = 0;
  int *idx = (int *)malloc(sizeof(int) * (size_t)ns), *nz = (int *)malloc(sizeof(int) * (size_t)ns);
  double *x1 = (double *)malloc(sizeof(double) * 3 * (size_t)ns), *hs = (double *)malloc(sizeof(double) * nrow_s * ns);
  double *hd = (double *)malloc(sizeof(double) * 6 * (size_t)nd_), *ho = (double *)malloc(sizeof(double) * nrow_o * nd_);
  int rc = JUR_OK;
  double *d = NULL;
  int *di = NULL;
  if (!idx || !nz || !x1 || !hs || !hd || !ho) { rc = JUR_ENOMEM; goto done; }
  if (ip == 2) {   /* grid dimensions of the track, jurassic.c:716-735 */
    double lat1 = -999, lon1 = -999;
    for (int i = 0; i < ns; i++) {
      if ((src->lon[i] != lon1) || (src->lat[i] != lat1)) {
        nz[nx] = 0;
        lon1 = src->lon[i];
        lat1 = src->lat[i];
        geo2cart0(lon1, lat1, x1 + 3 * (size_t)nx);
        idx[nx++] = i;
      }
      ++nz[nx - 1];
    }
    for (int ix = 0; ix < nx && !rc; ix++) {
      if (nz[ix] <= 1) { jur_set_error("Cannot identify profiles. Check ordering of data points!"); rc = JUR_EINVAL; }
      else if ((ix > 0) && (fabs(src->lat[idx[ix - 1]] - src->lat[idx[ix]]) > 10)) { jur_set_error("Distance of profiles is too large!"); rc = JUR_EINVAL; }
    }
    if (rc) goto done;
  } else if (ip == 3)
    for (int i = 0; i < ns; i++) geo2cart0(src->lon[i], src->lat[i], x1 + 3 * (size_t)i);
  {
    double const *rows[5] = {src->z, src->lon, src->lat, src->p, src->t};
    for (int r = 0; r < 5; r++) memcpy(hs + (size_t)r * ns, rows[r], sizeof(double) * ns);
    for (int g = 0; g < ng; g++) memcpy(hs + (5 + (size_t)g) * ns, src->q[g], sizeof(double) * ns);
    for (int w = 0; w < nw; w++) memcpy(hs + (5 + (size_t)ng + w) * ns, src->k[w], sizeof(double) * ns);
    memcpy(hd, dest->z, sizeof(double) * nd_);
    memcpy(hd + nd_, dest->lon, sizeof(double) * nd_);
    memcpy(hd + 2 * (size_t)nd_, dest->lat, sizeof(double) * nd_);
    for (int i = 0; i < nd_; i++) geo2cart0(dest->lon[i], dest->lat[i], hd + 3 * (size_t)nd_ + 3 * (size_t)i);
  }
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { jur_set_error("no HIP device available"); rc = JUR_ENODEV; goto done; }
    if (device < 0 || device >= ndev) { jur_set_error("device %d not in 0..%d", device, ndev - 1); rc = JUR_ENODEV; goto done; }
    size_t const nsrc = nrow_s * ns, nx1 = 3 * (size_t)ns, ndst = 6 * (size_t)nd_, nout = nrow_o * nd_;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void **)&d, sizeof(double) * (nsrc + nx1 + ndst + nout));
    if (e == hipSuccess) e = hipMalloc((void **)&di, sizeof(int) * 2 * (size_t)ns);
    if (e == hipSuccess) e = hipMemcpy(d, hs, sizeof(double) * nsrc, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + nsrc, x1, sizeof(double) * nx1, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + nsrc + nx1, hd, sizeof(double) * ndst, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(di, idx, sizeof(int) * ns, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(di + ns, nz, sizeof(int) * ns, hipMemcpyHostToDevice);
    if (e == hipSuccess)
      e = (hipError_t)jurk_launch_intpol(ip, ng, nw, ns, nd_, nx, ctl->cx, ctl->cz, d, d + nsrc, di, di + ns, d + nsrc + nx1,
                                         d + nsrc + nx1 + 3 * (size_t)nd_, d + nsrc + nx1 + ndst, NULL);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(ho, d + nsrc + nx1 + ndst, sizeof(double) * nout, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { jur_set_error("jur_intpol_atm: %s", hipGetErrorString(e)); rc = JUR_EHIP; goto done; }
  }
  memcpy(dest->p, ho, sizeof(double) * nd_);
  memcpy(dest->t, ho + nd_, sizeof(double) * nd_);
  for (int g = 0; g < ng; g++) memcpy(dest->q[g], ho + (2 + (size_t)g) * nd_, sizeof(double) * nd_);
  for (int w = 0; w < nw; w++) memcpy(dest->k[w], ho + (2 + (size_t)ng + w) * nd_, sizeof(double) * nd_);
done:
  if (d) (void)hipFree(d);
  if (di) (void)hipFree(di);
  free(idx); free(nz); free(x1); free(hs); free(hd); free(ho);
  return rc;
}

/* drop-in: void return, errors print and exit, as upstream's ERRMSG (jurassic.h:581-585) */
void intpol_atm(ctl_t *ctl, atm_t *atm_dest, atm_t *atm_src) {
  if (jur_intpol_atm(ctl, atm_dest, atm_src, (ctl && ctl->MPIlocalrank > 0) ? ctl->MPIlocalrank : 0) != JUR_OK) {
    printf("\nError (%s, %s, l%d): %s\n\n", __FILE__, __func__, __LINE__, jur_last_error());
    fflush(stdout);
    exit(EXIT_FAILURE);
  }
}
