/* jur_fov.c -- field-of-view convolution, the post-processing step after the forward model.
 *
 * Reference: formod_fov (src/jurassic.c:214-258) with read_shape (:1134-1150) and copy_obs (:168-195).
 * Upstream declares formod_fov in jurassic.h:521 but no program of this code base calls it; it is rebuilt
 * for the drop-in symbol set.  Host code: the work is nr x n_fov x nd interpolations on results that are
 * already back in obs_t.
 *
 * For every ray: the (view-point altitude, radiance, transmittance) profiles of the rays within
 * +-JUR_NFOV positions that carry the same time stamp are interpolated to vpz + dz[i] and summed with
 * weights w[i]; the sum is divided by the sum of the weights.  All rays read the values the forward model
 * wrote (a copy), not values an earlier ray of the loop has already convolved.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "jur_internal.h"

int jur_fov_read_shape(char const *filename, int *n, double *dz, double *w) {
  if (!filename || !n || !dz || !w) { jur_set_error("jur_fov_read_shape: null argument"); return JUR_EINVAL; }
  FILE *in = fopen(filename, "r");
  if (!in) { jur_set_error("cannot open field-of-view file %s", filename); return JUR_EIO; }
  static char line[JUR_LEN];
  int k = 0;
  while (fgets(line, JUR_LEN, in))
    if (sscanf(line, "%lg %lg", &dz[k < JUR_NSHAPE ? k : JUR_NSHAPE - 1], &w[k < JUR_NSHAPE ? k : JUR_NSHAPE - 1]) == 2)
      if (++k > JUR_NSHAPE) { fclose(in); jur_set_error("%s: more than %d field-of-view points", filename, JUR_NSHAPE); return JUR_EINVAL; }
  fclose(in);
  if (k < 1) { jur_set_error("%s: no field-of-view data", filename); return JUR_EIO; }
  *n = k;
  return JUR_OK;
}

/* bracket of x in an ascending or descending axis (jr_common.h:87-104) */
static int locate(double const *xx, int n, double x) {
  int ilo = 0, ihi = n - 1, i = (n - 1) >> 1;
  if (xx[i] < xx[i + 1]) {
    while (ihi > ilo + 1) { i = (ihi + ilo) >> 1; if (xx[i] > x) ihi = i; else ilo = i; }
  } else {
    while (ihi > ilo + 1) { i = (ihi + ilo) >> 1; if (xx[i] <= x) ihi = i; else ilo = i; }
  }
  return ilo;
}

int jur_fov_apply(int nd, long nr, double const *time, double const *vpz, double *rad, double *tau, long ld,
                  int n, double const *dz, double const *w) {
  if (nd < 1 || nr < 0 || ld < nd || n < 1 || !time || !vpz || !rad || !tau || !dz || !w) {
    jur_set_error("jur_fov_apply: bad arguments");
    return JUR_EINVAL;
  }
  size_t const bytes = sizeof(double) * (size_t)nr * (size_t)nd;
  double *rad0 = (double *)malloc(bytes ? bytes : 8), *tau0 = (double *)malloc(bytes ? bytes : 8);
  if (!rad0 || !tau0) { free(rad0); free(tau0); jur_set_error("jur_fov_apply: out of memory"); return JUR_ENOMEM; }
  for (long ir = 0; ir < nr; ir++) {
    memcpy(rad0 + ir * nd, rad + ir * ld, sizeof(double) * nd);
    memcpy(tau0 + ir * nd, tau + ir * ld, sizeof(double) * nd);
  }
  int rc = JUR_OK;
  for (long ir = 0; ir < nr && !rc; ir++) {
    double z[2 * JUR_NFOV + 1];
    long src[2 * JUR_NFOV + 1];
    int nz = 0;
    long const lo = ir - JUR_NFOV > 0 ? ir - JUR_NFOV : 0, hi = ir + 1 + JUR_NFOV < nr ? ir + 1 + JUR_NFOV : nr;
    for (long ir2 = lo; ir2 < hi; ir2++)
      if (time[ir2] == time[ir]) { z[nz] = vpz[ir2]; src[nz] = ir2; nz++; }
    if (nz < 2) { jur_set_error("Cannot apply FOV convolution!"); rc = JUR_EINVAL; break; }
    double wsum = 0;
    double *r = rad + ir * ld, *t = tau + ir * ld;
    for (int id = 0; id < nd; id++) { r[id] = 0; t[id] = 0; }
    for (int i = 0; i < n; i++) {
      double const zfov = vpz[ir] + dz[i];
      int const idx = locate(z, nz, zfov);
      double const *r0 = rad0 + src[idx] * nd, *r1 = rad0 + src[idx + 1] * nd;
      double const *t0 = tau0 + src[idx] * nd, *t1 = tau0 + src[idx + 1] * nd;
      for (int id = 0; id < nd; id++) {
        r[id] += w[i] * (r0[id] + (zfov - z[idx]) * (r1[id] - r0[id]) / (z[idx + 1] - z[idx]));
        t[id] += w[i] * (t0[id] + (zfov - z[idx]) * (t1[id] - t0[id]) / (z[idx + 1] - z[idx]));
      }
      wsum += w[i];
    }
    for (int id = 0; id < nd; id++) { r[id] /= wsum; t[id] /= wsum; }
  }
  free(rad0); free(tau0);
  return rc;
}

/* drop-in: void return, errors print and exit; the shape file is read on the first call and kept for the
 * life of the process, as upstream's function-statics do */
void formod_fov(ctl_t const *ctl, obs_t *obs) {
  static double dz[JUR_NSHAPE], w[JUR_NSHAPE];
  static int init = 0, n = 0;
  if (ctl->fov[0] == '-') return;
  int rc = JUR_OK;
  if (!init) {
    printf("Read shape function: %s\n", ctl->fov);
    if (ctl->checkmode) { printf("# read_shape found %s\n", ctl->fov); init = 1; n = 0; }
    else if (!(rc = jur_fov_read_shape(ctl->fov, &n, dz, w))) init = 1;
  }
  if (!rc && n > 0)
    rc = jur_fov_apply(ctl->nd, obs->nr, obs->time, obs->vpz, &obs->rad[0][0], &obs->tau[0][0], JUR_ND, n, dz, w);
  if (rc) {
    printf("\nError (%s, %s, l%d): %s\n\n", __FILE__, __func__, __LINE__, jur_last_error());
    fflush(stdout);
    exit(EXIT_FAILURE);
  }
}
