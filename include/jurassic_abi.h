/* jurassic_abi.h -- caller-visible data model of the JURASSIC forward model.
 *
 * Layout-compatible restatement of the three structs that cross the
 * formod() boundary in slcs-jsc/jurassic-gpu:
 *   ctl_t  reference src/jurassic.h:229-347
 *   atm_t  reference src/jurassic.h:215-226
 *   obs_t  reference src/jurassic.h:371-385
 * and of the compile-time dimensions they depend on (jurassic.h:137-193).
 * A program compiled against the reference's jurassic.h can pass its structs
 * to this library unchanged as long as both sides agree on JUR_ND / JUR_NG
 * (the reference's -D ND / -D NG); the library exports jur_abi_sizes() so the
 * caller can verify that at start-up.
 *
 * Only the members the forward model reads or writes are documented; the
 * others exist to keep offsets identical.
 */
#ifndef JURASSIC_ABI_H
#define JURASSIC_ABI_H

#include <stddef.h>

/* ---- dimensions (reference jurassic.h:137-193) -------------------------- */
#ifndef JUR_ND
#  ifdef ND
#    define JUR_ND ND
#  else
#    define JUR_ND 100      /* max radiance channels               */
#  endif
#endif
#ifndef JUR_NG
#  ifdef NG
#    define JUR_NG NG
#  else
#    define JUR_NG 30       /* max emitters                         */
#  endif
#endif
#define JUR_NP    9600      /* max atmospheric data points          */
#define JUR_NR    1088      /* max rays per obs_t package           */
#define JUR_NW    1         /* max spectral windows                 */
#define JUR_LEN   5000      /* max ASCII line / string length       */
#define JUR_NLOS  400       /* max points on one line of sight      */
#define JUR_NSHAPE 2048     /* max filter-function grid points      */
#define JUR_NFOV  5         /* rays on either side used by the FOV convolution (jurassic.h:175) */
#define JUR_TBLNP 40        /* max pressure levels per table        */
#define JUR_TBLNT 30        /* max temperatures per pressure level  */
#define JUR_TBLNU 304       /* max column densities per curve       */
#define JUR_TBLNS 1201      /* source-function temperatures         */

/* ---- physical constants used on the path -------------------------------- */
#define JUR_C1 1.19104259e-8     /* jurassic.h:111  2hc^2   */
#define JUR_C2 1.43877506        /* jurassic.h:114  hc/k    */
#define JUR_P0 1013.25           /* jurassic.h:120  hPa     */
#define JUR_RE 6367.421          /* jurassic.h:126  km      */
/* GSL 2.5 values the reference picks up through gsl_const_*.h
 * (jr_common.h:330,450,744). */
#define JUR_AVOGADRO  6.02214199e23
#define JUR_BOLTZMANN 1.3806504e-23
#define JUR_MOLAR_GAS 8.314472

/* ---- atmosphere: SoA profile(s); several profiles may be stacked and are
 *      told apart by their time stamp (jr_common.h:127-154) ---------------- */
typedef struct {
  double time[JUR_NP];
  double z[JUR_NP];            /* km   */
  double lon[JUR_NP];          /* deg  */
  double lat[JUR_NP];          /* deg  */
  double p[JUR_NP];            /* hPa  */
  double t[JUR_NP];            /* K    */
  double q[JUR_NG][JUR_NP];    /* volume mixing ratio per emitter */
  double k[JUR_NW][JUR_NP];    /* extinction 1/km per window      */
  int np;
  int init;
} atm_t;

/* ---- control block ------------------------------------------------------- */
typedef struct {
  int ng;
  char emitter[JUR_NG][JUR_LEN];
  int nd;
  int nw;
  double nu[JUR_ND];           /* channel centroid wavenumber cm^-1 */
  int window[JUR_ND];
  char tblbase[JUR_LEN];
  double hydz;
  int ctm_co2, ctm_h2o, ctm_n2, ctm_o2;
  int ip;
  double cz, cx;
  int refrac;
  double rayds, raydz;
  char fov[JUR_LEN];
  double retp_zmin, retp_zmax, rett_zmin, rett_zmax;
  double retq_zmin[JUR_NG], retq_zmax[JUR_NG];
  double retk_zmin[JUR_NW], retk_zmax[JUR_NW];
  int write_bbt;
  int write_matrix;
  int formod;
  char rfmbin[JUR_LEN];
  char rfmhit[JUR_LEN];
  char rfmxsc[JUR_NG][JUR_LEN];
  int useGPU;
  int checkmode;
  int MPIglobrank, MPIlocalrank;
  int read_binary, write_binary;
  int gpu_nbytes_shared_memory;
} ctl_t;

/* ---- observation package: geometry in, radiance/transmittance out -------- */
typedef struct {
  double time[JUR_NR];
  double obsz[JUR_NR], obslon[JUR_NR], obslat[JUR_NR];
  double vpz[JUR_NR],  vplon[JUR_NR],  vplat[JUR_NR];
  double tpz[JUR_NR],  tplon[JUR_NR],  tplat[JUR_NR];
  double tau[JUR_NR][JUR_ND];
  double rad[JUR_NR][JUR_ND];
  int nr;
} obs_t;

/* Offsets/sizes measured on the reference's own headers with gcc x86-64 at the
 * default dimensions. */
#if JUR_ND == 100 && JUR_NG == 30
#  if defined(__cplusplus)
#    define JUR_SA(c, m) static_assert(c, m)
#  else
#    define JUR_SA(c, m) _Static_assert(c, m)
#  endif
JUR_SA(sizeof(ctl_t) == 321856, "ctl_t size");
JUR_SA(offsetof(ctl_t, emitter) == 4, "ctl.emitter");
JUR_SA(offsetof(ctl_t, nd) == 150004, "ctl.nd");
JUR_SA(offsetof(ctl_t, nu) == 150016, "ctl.nu");
JUR_SA(offsetof(ctl_t, window) == 150816, "ctl.window");
JUR_SA(offsetof(ctl_t, tblbase) == 151216, "ctl.tblbase");
JUR_SA(offsetof(ctl_t, hydz) == 156216, "ctl.hydz");
JUR_SA(offsetof(ctl_t, ctm_co2) == 156224, "ctl.ctm_co2");
JUR_SA(offsetof(ctl_t, ip) == 156240, "ctl.ip");
JUR_SA(offsetof(ctl_t, refrac) == 156264, "ctl.refrac");
JUR_SA(offsetof(ctl_t, rayds) == 156272, "ctl.rayds");
JUR_SA(offsetof(ctl_t, raydz) == 156280, "ctl.raydz");
JUR_SA(offsetof(ctl_t, fov) == 156288, "ctl.fov");
JUR_SA(offsetof(ctl_t, write_bbt) == 161816, "ctl.write_bbt");
JUR_SA(offsetof(ctl_t, formod) == 161824, "ctl.formod");
JUR_SA(offsetof(ctl_t, useGPU) == 321828, "ctl.useGPU");
JUR_SA(offsetof(ctl_t, checkmode) == 321832, "ctl.checkmode");
JUR_SA(offsetof(ctl_t, MPIglobrank) == 321836, "ctl.MPIglobrank");
JUR_SA(offsetof(ctl_t, read_binary) == 321844, "ctl.read_binary");
JUR_SA(offsetof(ctl_t, gpu_nbytes_shared_memory) == 321852, "ctl.gpu_nbytes");
JUR_SA(sizeof(atm_t) == 2841608, "atm_t size");
JUR_SA(offsetof(atm_t, z) == 76800, "atm.z");
JUR_SA(offsetof(atm_t, p) == 307200, "atm.p");
JUR_SA(offsetof(atm_t, t) == 384000, "atm.t");
JUR_SA(offsetof(atm_t, q) == 460800, "atm.q");
JUR_SA(offsetof(atm_t, k) == 2764800, "atm.k");
JUR_SA(offsetof(atm_t, np) == 2841600, "atm.np");
JUR_SA(sizeof(obs_t) == 1827848, "obs_t size");
JUR_SA(offsetof(obs_t, obsz) == 8704, "obs.obsz");
JUR_SA(offsetof(obs_t, vpz) == 34816, "obs.vpz");
JUR_SA(offsetof(obs_t, tpz) == 60928, "obs.tpz");
JUR_SA(offsetof(obs_t, tau) == 87040, "obs.tau");
JUR_SA(offsetof(obs_t, rad) == 957440, "obs.rad");
JUR_SA(offsetof(obs_t, nr) == 1827840, "obs.nr");
#endif

#endif /* JURASSIC_ABI_H */
