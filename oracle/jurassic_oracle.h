/* jurassic_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the JURASSIC EGA forward model (slcs-jsc/jurassic-gpu,
 * src/jr_common.h + src/CPUdrivers.c + the table/Planck part of
 * src/jurassic.c), used only as the checker in tests/, in
 * __graft_entry__.smoke() and as bench.py's cpu_baseline leg.  The shipped
 * library (jurassic-gpu_amd/csrc) never includes or links this.
 *
 * PARITY PIN STATUS: the ray-tracing half (traceray, tangent_point) is pinned
 * against the tangent-point columns of the reference's own golden files
 * example/limb/rad.org and example/nadir/rad.org (tests/golden/).  The
 * radiance/transmittance half is "parity unpinned": the emissivity tables the
 * reference's goldens were made with are not in the reference tree
 * (.MISSING_LARGE_BLOBS) and the reference cannot be built here (needs GSL,
 * absent).  See DESIGN.md section "Oracle".
 */
#ifndef JURASSIC_ORACLE_H
#define JURASSIC_ORACLE_H

#include <stdint.h>
#include "jurassic_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Emissivity look-up tables: same index order as the reference's tbl_t
 * (jurassic.h:390-425; channel index fastest) but heap-allocated at the
 * dimensions actually requested instead of 8.8 GB. */
typedef struct {
  int ng, nd;            /* gases, channels allocated           */
  int mp, mt, mu;        /* allocated extents: pressure, T, u   */
  int32_t *np;           /* [ng][nd]                            */
  int32_t *nt;           /* [ng][mp][nd]                        */
  int32_t *nu;           /* [ng][mp][mt][nd]                    */
  double  *p;            /* [ng][mp][nd]                        */
  double  *t;            /* [ng][mp][mt][nd]                    */
  float   *u;            /* [ng][mp][mt][mu][nd]                */
  float   *eps;          /* [ng][mp][mt][mu][nd]                */
  double  *sr;           /* [JUR_TBLNS][nd]                     */
  double   st[JUR_TBLNS];
} orc_tbl_t;

orc_tbl_t *orc_tbl_new(int ng, int nd, int mp, int mt, int mu);
void       orc_tbl_free(orc_tbl_t *tbl);

/* Parse `${tblbase}_${nu:%.4f}_${emitter}.tab` for every (gas, channel) with
 * the reference's row-acceptance rules (jurassic.c:329-400).  Returns the
 * number of missing files (those pairs stay "no table"). */
int  orc_tbl_read_ascii(orc_tbl_t *tbl, ctl_t const *ctl);

/* In-memory equivalent of feeding one table file: rows (p,T,u,eps) in file
 * order for pair (ig,id). */
void orc_tbl_feed_rows(orc_tbl_t *tbl, int ig, int id, long nrows,
                       double const *press, double const *temp,
                       double const *u, double const *eps);

/* Source function: filter-weighted Planck radiance on the 0.25 K grid
 * (jurassic.c:612-667).  `_filt` reads `${tblbase}_${nu:%.4f}.filt`;
 * `_shape` takes the filter in memory. */
int  orc_tbl_planck_filt(orc_tbl_t *tbl, ctl_t const *ctl);
void orc_tbl_planck_shape(orc_tbl_t *tbl, int id, int n,
                          double const *nu, double const *f);

/* formod_CPU on a reference-layout package (CPUdrivers.c:109-151). */
void orc_formod(ctl_t const *ctl, atm_t *atm, obs_t *obs, orc_tbl_t const *tbl);

/* Same computation on flat arrays of any length (rad/tau are [nr][nd_stride],
 * rad is read first for the NaN mask).  np_out/tsurf_out may be NULL.
 * serial_trace!=0 mirrors the reference's serial ray tracing
 * (CPUdrivers.c:91 is an orphaned omp-for). */
void orc_formod_rays(ctl_t const *ctl, atm_t *atm, orc_tbl_t const *tbl,
                     long nr, int nd_stride,
                     double const *time, double const *obsz, double const *obslon,
                     double const *obslat, double const *vpz, double const *vplon,
                     double const *vplat,
                     double *tpz, double *tplon, double *tplat,
                     double *rad, double *tau,
                     int *np_out, double *tsurf_out, int serial_trace);

/* Algorithmic bytes of the reference algorithm for these rays
 * (A_ray summed over rays).  Also returns the number of
 * LOS segments in *nseg and, in *trace_part, the share that belongs to ray
 * tracing (B_io + sum of B_atm) and, in *ega_part, the emissivity-growth look-ups
 * (sum of B_ega); the rest (B_seg, B_src) belongs to the per-segment combine step. */
double orc_algorithmic_bytes(ctl_t const *ctl, atm_t *atm, orc_tbl_t const *tbl,
                             long nr,
                             double const *time, double const *obsz, double const *obslon,
                             double const *obslat, double const *vpz, double const *vplon,
                             double const *vplat, long *nseg, double *trace_part,
                             double *ega_part);

/* Retrieval interface (jurassic.c:812-857, 1473-1541): state vector of the
 * atmosphere inside the ctl->ret*_zmin/zmax windows, measurement vector of the
 * finite radiances, and the forward-difference Jacobian k[m][n] (row-major).
 * orc_kernel leaves the unperturbed forward model result in obs. */
size_t orc_atm2x(ctl_t const *ctl, atm_t const *atm, double *x, int *iqa, int *ipa);
size_t orc_obs2y(ctl_t const *ctl, obs_t const *obs, double *y);
void   orc_kernel(ctl_t const *ctl, atm_t *atm, obs_t *obs, orc_tbl_t const *tbl,
                  double *k, size_t m, size_t n);

/* Function-level entry points for known-answer tests. */
double orc_ega_eps(orc_tbl_t const *tbl, double tau, double t, double u, double p, int ig, int id);
double orc_ctmco2(double nu, double p, double t, double u);
double orc_ctmh2o(double nu, double p, double t, double q, double u);
double orc_ctmn2(double nu, double p, double t);
double orc_ctmo2(double nu, double p, double t);
double orc_planck(double t, double nu);
double orc_src_planck(orc_tbl_t const *tbl, double t, int id);                         /* src_planck_core, jr_common.h:220-224 */
void   orc_new_obs(double tau_gas, double beta_ds, double src, double *rad, double *tau);   /* new_obs_core, :293-300 */
void   orc_add_surface(orc_tbl_t const *tbl, double tsurf, int id, double *rad, double tau);   /* add_surface_core, :227-234 */
double orc_brightness(double rad, double nu);
/* One line of sight: returns np, fills SoA outputs of length JUR_NLOS
 * (q/u are [JUR_NG][JUR_NLOS]); tp[3] = tpz,tplon,tplat. */
int orc_traceray(ctl_t const *ctl, atm_t const *atm, double const geom[7],
                 double *z, double *lon, double *lat, double *p, double *t,
                 double *ds, double *k, double *q, double *u,
                 double *tsurf, double tp[3]);
void orc_hydrostatic(ctl_t const *ctl, atm_t *atm);

/* the oracle's own physical constants as compiled (oracle_constants.h): C1, C2, P0, RE, N_A, k_B, R */
void orc_constants(double out[7]);
/* Curtis-Godson means of one ray (jr_common.h:455-473): cgp/cgt/cgu are [JUR_NG... ng][JUR_NLOS]; returns np */
int orc_set_threads(int n);
/* intpol_atm (jurassic.c:675-804): dest->p, t, q, k at dest's z / lon / lat from src by ctl->ip = 1 (one profile),
 * 2 (nearest two profiles of a track) or 3 (distance-weighted mean of a point cloud, ctl->cx, ctl->cz).
 * 0, or -(number of the upstream error: 1 too many profiles, 2 ordering, 3 profile distance, 4 unknown IP). */
int orc_intpol_atm(ctl_t const *ctl, atm_t *dest, atm_t const *src);
int orc_formod_fov(ctl_t const *ctl, obs_t *obs, int n, double const *dz, double const *w);
int orc_curtis_godson(ctl_t const *ctl, atm_t const *atm, double const geom[7], double *cgp, double *cgt, double *cgu);
int  orc_find_emitter(ctl_t const *ctl, char const *name);

#ifdef __cplusplus
}
#endif
#endif
