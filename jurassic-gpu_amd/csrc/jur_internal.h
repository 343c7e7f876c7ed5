/* jur_internal.h -- structures shared by the C host code and the HIP kernels.
 * Plain C so that gcc (host) and hipcc (device) agree on the layout. */
#ifndef JUR_INTERNAL_H
#define JUR_INTERNAL_H

#include <stdint.h>
#include "jurassic_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int a, b; } jur_int2;
typedef struct { float u, eps; } jur_ue_t;
/* slopes of the bracket [entry i, entry i+1] of a curve, formed on the device once per model from the fp32 entries
 * (strictly increasing tables only): what get_u and get_eps multiply with instead of dividing by the bracket width */
typedef struct { double du_de, de_du; } __attribute__((aligned(16))) jur_sl_t;
/* bracket [entry i, entry i+1] of a curve in one record: the two entries as stored and both slopes (strict tables) */
typedef struct { float u0, e0, u1, e1; double du_de, de_du; } __attribute__((aligned(32))) jur_rec_t;
/* 16-byte descriptors: one load brings the axis value together with the extent
 * and the offset of the next level of the hierarchy. */
typedef struct { double p; int nt; int c0; } jur_lvl_t;   /* pressure level: nt curves from curve c0 */
typedef struct { double t; int nu; int e0; } jur_crv_t;   /* curve: nu (u,eps) entries from entry e0 OF ITS PAIR
                                                             (pair_e0[pair] + e0 in the ue array): 32 bits hold any
                                                             pair (<= 40 x 30 x 304 entries), the set may hold > 2^31 */

/* Work item of jur_ega_group_kernel: up to JUR_EGA_NCH channels of ONE gas whose tables stand on the same (p, T)
 * grid (same levels, same temperatures per level) -- what tables produced on one grid do.  A lane then owns a ray and
 * the gas and walks the item's channels inside the segment loop: the LOS row, the p bracket, both T brackets are
 * found once per segment and item instead of once per (channel, gas) pair (jr_common.h:241-246: the brackets do not
 * depend on the curve). */
#define JUR_EGA_NCH 4
typedef struct {
  int g, nch;
  int flags;                    /* bit 0: every curve of the item's tables has >= 2 entries */
  int pad;
  int d[JUR_EGA_NCH];
  long long e0[JUR_EGA_NCH];    /* first entry of pair (g, d[k]) in ue */
} jur_item_t;

/* Everything about the continua that depends on the channel only, reduced on
 * the host once per model with the reference's own expression order
 * (jr_common.h:318-325, 336-357, 367-372, 381-386). */
typedef struct {
  double nu;
  double co2_cw296, co2_cw260, co2_cw230;   /* 2 cm^-1 grid interpolates            */
  double h2o_sc;                            /* sfac * cw296                         */
  double h2o_ratio;                         /* cw260 / cw296                        */
  double h2o_lnr_hi, h2o_lnr_lo;            /* ln(h2o_ratio) = hi + lo (from logl), for ratio^y = exp(y ln ratio) */
  double h2o_ctwfrn;                        /* cwfrn * fscal                        */
  double n2_b, n2_beta;
  double o2_b, o2_beta;
  int co2_on, h2o_on, n2_on, o2_on;         /* channel inside the continuum's range */
  int window;
  int pad;
} jur_chan_t;

/* LOS workspace: tiles of 64 ray slots, fields stored as [tile][point][field][64]. */
enum { JUR_F_P = 0, JUR_F_T = 1, JUR_F_DS = 2, JUR_F_QH2O = 3, JUR_F_K = 4 /* + nw, then u[ng] */ };

/* Read-only device view of a model; passed to kernels by value. */
typedef struct {
  int ng, nd, nw;
  int fourbit, ig_co2, ig_h2o;
  int refrac, write_bbt;
  double rayds, raydz;
  jur_chan_t const *chan;       /* [nd]                                         */
  double const *sr;             /* [nd][JUR_TBLNS] source function              */
  /* emissivity tables, CSR-like:
   * pair[g*nd+d] = {np, first level}; lvl per level; crv per curve;
   * ue = interleaved (u, eps) float pairs, unit stride along u. */
  jur_int2 const *pair;
  jur_lvl_t const *lvl;
  jur_crv_t const *crv;
  jur_ue_t const *ue;
  jur_sl_t const *sl;           /* [entries] bracket slopes, indexed like ue; NULL unless strict_tables            */
  jur_rec_t const *rec;         /* [entries] bracket records (entries i, i+1 and the slopes of [i, i+1]), or NULL      */
  long long const *pair_e0;     /* [ng*nd] first entry of every pair in ue (64 bits: a full-extent many-channel set --
                                   2378 channels x 3 gases x 40 x 30 x 304 = 2.6e9 entries -- is addressed as
                                   wave-uniform pair base + 32-bit offset inside the pair)                        */
  int sorted_tables;            /* every axis and curve is non-decreasing: any bracket search
                                   finds what the reference's bisection finds                  */
  int atm_maxslice;             /* longest run of equal time stamps in the atmosphere          */
  int max_pair_curves;          /* most curves any (gas, channel) pair has (LDS staging size)  */
  int strict_tables;            /* sorted, and p, T axes and (as stored, fp32) all curves strictly increasing:
                                   no bracket of the look-up has zero width                                     */
  int fast_arith;               /* the look-up runs the strict-table arithmetic (JUR_ARITH_FAST on strict tables whose
                                   descriptors fit the LDS staging); 0: the reference's divisions operand for operand */
  jur_item_t const *ega_items;  /* [ega_nitems] channel groups on a shared (p, T) grid, or NULL: one pair per workgroup */
  int ega_nitems;
  int ega_nch;                  /* channels of the largest item (1: nothing is shared)                           */
  /* atmosphere, compact SoA of atm_np points */
  int atm_np;
  int atm_sorted;               /* time stamps non-decreasing and z strictly monotone inside every slice */
  double const *atm_time, *atm_z, *atm_lon, *atm_lat, *atm_p, *atm_t;
  double const *atm_q;          /* [ng][atm_np] */
  double const *atm_k;          /* [nw][atm_np] */
  double const *atm_pslope;     /* [atm_np] ln-pressure slope to the next level, NaN if not defined */
} jur_view_t;

/* One chunk of rays handed to the kernels; all pointers are device memory.
 * Slot i of the chunk works on ray order[i] of the caller's arrays (order ==
 * NULL: ray first+i), so that the caller-visible arrays are indexed by ray id
 * while the workspace arrays (np, tsurf, los) are indexed by slot. */
typedef struct {
  int n;                        /* rays in this chunk                           */
  int stride;                   /* Rt: slots the LOS workspace holds (multiple of 64)           */
  int stride_eps;               /* R: slots the transmittance workspace holds (multiple of 64)  */
  long first;                   /* first ray id when order == NULL              */
  int const *order;             /* [n] ray ids of this chunk, or NULL           */
  double const *geom[7];        /* time, obsz, obslon, obslat, vpz, vplon, vplat; [nr] */
  double *tp[3];                /* tpz, tplon, tplat; [nr]                      */
  double *rad, *tau;            /* [nr][nd]                                     */
  int *np_out;                  /* [nr] LOS points per ray, or NULL             */
  int *np;                      /* [n] LOS points per slot                      */
  double *tsurf;                /* [n]                                          */
  double *los;                  /* tiles of 64 slots: [slot / 64][JUR_NLOS][nfield][64]  */
  double *eps;                  /* path transmittances, tiles of 64 slots: [slot / 64][JUR_NLOS][nd*ng][64] */
  int const *eps_off;           /* [tiles of the chunk] first point slot of every tile in eps when the tiles are laid out
                                   by their longest path instead of JUR_NLOS points each; NULL: tile * JUR_NLOS       */
  int *status;                  /* device flag: bit0 = NLOS overflow            */
} jur_chunk_t;

/* kernel launchers (jur_kernels.hip); return hipError_t as int */
int jurk_prepare_atm(jur_view_t const *v, double *d_pslope, void *stream);   /* fills atm_pslope */
int jurk_launch_trace(jur_view_t const *v, jur_chunk_t const *c, void *stream);
int jurk_launch_ega(jur_view_t const *v, jur_chunk_t const *c, void *stream);
int jurk_launch_combine(jur_view_t const *v, jur_chunk_t const *c, void *stream);
int jurk_tile_max(int n, int const *d_np, int *d_tile_np, void *stream);
/* dense difference quotients kq[nq][n] from rad[(n + 1) * nq] (nq = rays x channels of the unperturbed block) and steps h[n] */
int jurk_launch_kquot(long nq, long n, double const *d_rad, double const *d_h, double *d_kq, void *stream);   /* longest path per tile of 64 slots */
void jurk_tune_combine(int group, int sync, long min_lanes);
void jurk_tune_trace(int lanes);
/* Curtis-Godson columns of the traced chunk: outputs [ray][gas][JUR_NLOS], indexed by ray id */
int jurk_launch_cg(jur_view_t const *v, jur_chunk_t const *c, double *cgp, double *cgt, double *cgu, void *stream);
/* bracket slopes of all n table entries (the last entry of a curve gets a value nobody reads) */
int jurk_fill_slopes(jur_ue_t const *ue, jur_sl_t *sl, long long n, void *stream);
int jurk_fill_records(jur_ue_t const *ue, jur_sl_t const *sl, jur_rec_t *rec, long long n, void *stream);   /* ue, sl: n + 1 entries */
/* order rays by their geometric tangent altitude: fills order[nr]; `tmp` is a
 * device scratch of jurk_sort_tmp_bytes(nr) bytes */
long jurk_sort_tmp_bytes(long nr);
int jurk_sort_rays(jur_view_t const *v, int by_profile, long nr, double const *d_geom, long ld, int *d_order, void *tmp,
                   long tmp_bytes, void *stream);   /* field k of ray r at d_geom[k * ld + r] */

/* the whole path of small calls in one kernel, RB rays per workgroup (c->order is ignored: nothing is sorted) */
long jurk_pencil_lds_bytes(jur_view_t const *v, int RB);
int jurk_launch_pencil(jur_view_t const *v, jur_chunk_t const *c, int RB, void *stream);

/* field-of-view convolution of device arrays: rad0/tau0 [nr][nd] -> rad/tau [nr][ld]; status bit 1: a ray alone in its scan */
int jurk_launch_fov(long nr, int nd, double const *time, double const *vpz, double const *rad0, double const *tau0, double *rad,
                    double *tau, long ld, int n, double const *dz, double const *w, int *status, void *stream);

/* atmosphere regridding (intpol_atm): device rows as documented at jur_intpol_kernel */
int jurk_launch_intpol(int ip, int ng, int nw, int ns, int nd_, int nx, double cx, double cz, double const *src, double const *x1,
                       int const *idx, int const *nz, double const *dst, double const *x0, double *out, void *stream);

/* known-answer hooks (tests): device functions on arrays, see jurassic_hip.h */
int jurk_kat_ega(jur_view_t const *v, int g, int d, long n, double const *tau, double const *t, double const *u, double const *p,
                 int mode, int chain, double *out, void *stream);
int jurk_kat_continua(jur_view_t const *v, int d, long n, double const *p, double const *t, double const *q, double const *u_co2,
                      double const *u_h2o, double *out, void *stream);
int jurk_kat_update(jur_view_t const *v, int d, long n, int what, double const *a, double const *b, double const *c, double *rad,
                    double *tau, double *src, void *stream);

/* internals of a model that jur_multi.c needs (jur_model.c) */
/* jur_formod_device on rays that are part of larger arrays: geometry field k at d_geom + k * ldg, tangent-point field k
 * at d_tp + k * ldtp (jur_formod_device: ldg = ldtp = nr) */
int jur_formod_device_ld(jur_model_t *m, long nr, double const *d_geom, long ldg, double *d_rad, double *d_tau, double *d_tp,
                         long ldtp, int *d_np, int *d_status, void *stream);
int jur_model_device(jur_model_t const *m);
int jur_model_nd(jur_model_t const *m);
void *jur_model_stream(jur_model_t const *m);
int *jur_model_status_word(jur_model_t const *m);
/* step sizes of the ray tracer and the altitude range of the atmosphere on the device (for cost estimates) */
void jur_model_cost_params(jur_model_t const *m, double *rayds, double *raydz, double *zmin, double *zmax);
/* device scratch of the model for nr rays laid out as jur_formod_host's image: geom[7][nr] | rad[nr][nd] | tau[nr][nd] |
 * tp[3][nr] doubles, np[nr] ints */
int jur_model_io(jur_model_t *m, long nr, double **d_io, int **d_io_np);

/* host tables (jur_tables.c) */
typedef struct {
  double t;
  int nu, cap;
  float *u, *eps;
} jur_curve_t;

typedef struct {
  double p;
  int nt;
  jur_curve_t cv[JUR_TBLNT];
} jur_level_t;

typedef struct {
  int np;                       /* 0 = no table                                 */
  jur_level_t *lv;              /* [np]                                         */
} jur_pair_t;

struct jur_tables {
  int ng, nd;
  jur_pair_t *pair;             /* [ng*nd]                                      */
  double *sr;                   /* [nd][JUR_TBLNS]                              */
  char *have_sr;                /* [nd]                                         */
  long ignored_rows;            /* rows dropped because a curve was full        */
};

/* flatten host tables into the CSR arrays of jur_view_t (malloc'ed) */
typedef struct {
  long nlevel, ncurve, nentry;
  int sorted;                   /* all axes and curves non-decreasing           */
  int strict;                   /* ... strictly increasing, curves as stored     */
  int max_pair_curves;
  jur_int2 *pair;
  long long *pair_e0;           /* [npair] first entry of the pair in ue */
  jur_lvl_t *lvl;
  jur_crv_t *crv;
  jur_ue_t *ue;
} jur_flat_t;
int  jur_tables_flatten(jur_tables_t const *tb, jur_flat_t *out);
/* channel groups on a shared (p, T) grid: the class of every pair, then items of at most nch channels (malloc'ed;
 * pairs without a table are in no item) */
int  jur_flat_grid_classes(jur_flat_t const *f, int ng, int nd, int *cls, unsigned char *all_curves);
int  jur_group_items(int ng, int nd, int nch, int const *cls, unsigned char const *all_curves, long long const *pair_e0,
                     jur_item_t **items, int *nitems, int *max_nch);
void jur_flat_free(jur_flat_t *f);

void jur_tables_cache_filename(char *out, size_t len, ctl_t const *ctl);
int jur_chan_setup(jur_chan_t *ch, double nu, int window);
void jur_set_error(char const *fmt, ...);
int  jur_parse_number(char const **pp, double *out);   /* strtod-equivalent reader of the table files */

extern const double jur_ctm_blob[] __attribute__((visibility("hidden")));

#ifdef __cplusplus
}
#endif
#endif
