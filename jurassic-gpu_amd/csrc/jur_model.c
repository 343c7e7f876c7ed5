/* jur_model.c -- host orchestration: device residency of a model, the batched
 * formod entry points and the drop-in formod()/formod_GPU()/formod_pencil().
 *
 * Plain C over the HIP runtime C API.  There is deliberately no CPU fallback:
 * without a GPU every entry point fails loudly.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <pthread.h>
#include <unistd.h>
#include <sched.h>
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

struct jur_model {
  int device;
  int shared_tables;            /* 1: a lane of the drop-in entry; the table arrays belong to another model */
  ctl_t *ctl;                   /* private copy of the control block             */
  jur_view_t view;              /* device pointers                               */
  long table_bytes;
  /* device allocations owned by the model */
  void *d_chan, *d_sr, *d_pair, *d_pair_e0, *d_lvl, *d_crv, *d_ue, *d_sl, *d_items, *d_rec;
  int arith;                    /* JUR_ARITH_*                                   */
  int *grid_cls;                /* [ng*nd] grid class of every pair (strict tables), see jur_model_set_ega_group */
  unsigned char *grid_all;
  long long *h_pair_e0;
  void *d_atm;                  /* one slab for the compact atmosphere           */
  int atm_cap;
  int atm_slices;               /* distinct time stamps in the atmosphere        */
  double atm_zmin, atm_zmax;    /* altitude range of the atmosphere on the device */
  /* per-call workspace */
  int chunk_rays;               /* R                                             */
  int nfield;
  double *d_los;
  double *d_eps;                /* segment transmittances per (channel, gas)     */
  long ws_budget;               /* bytes of workspace the model may hold         */
  long ws_rays;                 /* R: rays per ega/combine launch the workspace is laid out for */
  long ws_trace_rays;           /* Rt >= R: rays per trace launch (LOS workspace stride)        */
  int trace_mult;               /* wanted Rt / R                                                */
  long use_rays, use_trace_rays;/* R, Rt of the current call (<= the allocated capacities)     */
  int *d_np;
  double *d_tsurf;
  int compact_ws;               /* 1 (default): calls that need several integration launches lay the transmittance tiles
                                   out by the path lengths that occur instead of JUR_NLOS points per ray           */
  int use_compact;              /* ... and the current call does                                                 */
  int *d_tile_np, *d_eps_off;   /* [ws_trace_rays / 64] longest path per tile; first point slot of the tile in d_eps */
  int *h_tile;                  /* pinned: [2][ws_trace_rays / 64] host images of the two                        */
  long n_launch_ega;            /* integration launches of the last batched call (reporting)                     */
  int *d_status;
  long los_bytes;
  /* ray ordering */
  int sort_rays;                /* 1: process rays in order of tangent altitude  */
  int *d_order;
  void *d_sort_tmp;
  long order_cap, sort_tmp_bytes;
  /* staging for the host entry */
  double *d_io;                 /* geom[7][cap] tp[3][cap] rad/tau[cap][nd]      */
  int *d_io_np;
  long io_cap;
  int *h_status;                /* pinned: the device status word comes back here (a pageable target would make the
                                   copy a blocking one inside the runtime and serialise concurrent callers)      */
  double *h_io;                 /* pinned host image of d_io (+ np behind it) for callers with pageable arrays */
  long h_io_cap;
  double *h_pkg;                /* pinned scratch of the drop-in entry: rad/tau of one package, [2][NR][nd] */
  double *h_atm;                /* host image of the atmosphere rows last uploaded (after the hydrostatic step) */
  long h_atm_n, h_atm_cap;
  double h_atm_hydz;
  hipStream_t stream;
  hipStream_t stream2;          /* copies that run beside the kernels of `stream` */
  hipEvent_t ev_mask, ev_trace;
  int host_call;                /* 1 while jur_formod_host drives jur_formod_device: record / wait for the events above */
  hipEvent_t ev_done;           /* recorded on the caller's stream at the end of every jur_formod_device call: what a
                                   later upload of the atmosphere waits for (no handle of the caller's is kept) */
  int have_done;
  /* field-of-view convolution of device arrays: grow-only scratch and its own status word */
  double *d_fov;
  long fov_cap;
  double *d_kq, *h_kq;          /* jur_kernel: perturbation steps and dense difference quotients (device, pinned host) */
  long kq_cap;
  /* small calls: the fused kernel (jur_pencil_kernel) instead of sort + three batched kernels */
  long pencil_rays;             /* calls of up to this many rays take it (0: never)             */
  int pencil_rb;                /* rays per workgroup (0: chosen from the call size)            */
  /* timing */
  int timing;
  hipEvent_t *evpool;           /* 2 events per timed launch                     */
  unsigned char *evkind;        /* 0 trace, 1 ega, 2 combine, 3 fused (pencil)   */
  int ntimed;
  double pencil_ms;
  long pencil_launches;
};

#define JUR_MAX_TIMED 4096

static int pencil_rays_per_group(jur_model_t const *m, long nr);

static int find_emitter(ctl_t const *ctl, char const *name) {
  for (int ig = 0; ig < ctl->ng; ig++)
    if (0 == strcasecmp(ctl->emitter[ig], name)) return ig;
  return -1;
}

static int upload(void **dst, void const *src, size_t bytes) {
  if (bytes == 0) bytes = 8;
  HIPCHK(hipMalloc(dst, bytes));
  if (src) HIPCHK(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return JUR_OK;
}

static int check_ctl(ctl_t const *ctl) {
  if (ctl->ng < 0 || ctl->ng > JUR_NG) { jur_set_error("ctl->ng=%d outside 0..%d", ctl->ng, JUR_NG); return JUR_EINVAL; }
  if (ctl->nd < 1 || ctl->nd > JUR_ND) { jur_set_error("ctl->nd=%d outside 1..%d", ctl->nd, JUR_ND); return JUR_EINVAL; }
  if (ctl->nw < 0 || ctl->nw > JUR_NW) { jur_set_error("ctl->nw=%d outside 0..%d", ctl->nw, JUR_NW); return JUR_EINVAL; }
  if (ctl->ip != 1) { jur_set_error("only 1-D profile interpolation is supported (ctl->ip == 1, as upstream asserts)"); return JUR_EINVAL; }
  if (ctl->formod == 1) { jur_set_error("FORMOD=1 (CGA) has no integrator upstream either"); return JUR_EINVAL; }
  for (int id = 0; id < ctl->nd; id++)
    if (ctl->window[id] < 0 || ctl->window[id] >= (ctl->nw > 0 ? ctl->nw : 1)) { jur_set_error("ctl->window[%d] out of range", id); return JUR_EINVAL; }
  return JUR_OK;
}

static int create_streams(jur_model_t *m) {
  if (hipHostMalloc((void **)&m->h_status, 64, hipHostMallocDefault) != hipSuccess ||
      hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&m->stream2, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&m->ev_mask, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&m->ev_trace, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&m->ev_done, hipEventDisableTiming) != hipSuccess) {
    jur_set_error("cannot create the model's streams and events");
    return JUR_EHIP;
  }
  return JUR_OK;
}

int jur_model_create(jur_model_t **out, ctl_t const *ctl, jur_tables_t const *tb, int device) {
  *out = NULL;
  int rc = check_ctl(ctl);
  if (rc) return rc;
  if (!tb || tb->ng < ctl->ng || tb->nd < ctl->nd) { jur_set_error("tables do not cover ctl (ng, nd)"); return JUR_EINVAL; }
  for (int id = 0; id < ctl->nd; id++)
    if (!tb->have_sr[id]) { jur_set_error("no filter function / source table for channel %d", id); return JUR_EINVAL; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { jur_set_error("no HIP device available"); return JUR_ENODEV; }
  if (device < 0 || device >= ndev) { jur_set_error("device %d not in 0..%d", device, ndev - 1); return JUR_ENODEV; }
  HIPCHK(hipSetDevice(device));

  jur_model_t *m = (jur_model_t *)calloc(1, sizeof *m);
  if (!m) return JUR_ENOMEM;
  m->device = device;
  m->ctl = (ctl_t *)malloc(sizeof(ctl_t));
  if (!m->ctl) { free(m); return JUR_ENOMEM; }
  memcpy(m->ctl, ctl, sizeof(ctl_t));
  jur_view_t *v = &m->view;
  v->ng = ctl->ng; v->nd = ctl->nd; v->nw = ctl->nw > 0 ? ctl->nw : 1;
  v->refrac = ctl->refrac; v->write_bbt = ctl->write_bbt;
  v->rayds = ctl->rayds; v->raydz = ctl->raydz;
  /* continuum switches, CPUdrivers.c:126-134 */
  v->ig_co2 = -999; v->ig_h2o = -999;
  if (ctl->ctm_h2o) v->ig_h2o = find_emitter(ctl, "H2O");
  if (ctl->ctm_co2) v->ig_co2 = find_emitter(ctl, "CO2");
  v->fourbit = ((1 == ctl->ctm_co2) && (v->ig_co2 >= 0)) * 8 + ((1 == ctl->ctm_h2o) && (v->ig_h2o >= 0)) * 4
             + (1 == ctl->ctm_n2) * 2 + (1 == ctl->ctm_o2) * 1;

  jur_chan_t *chan = (jur_chan_t *)calloc(ctl->nd, sizeof(jur_chan_t));
  if (!chan) { jur_model_destroy(m); return JUR_ENOMEM; }
  for (int id = 0; id < ctl->nd; id++)
    if ((rc = jur_chan_setup(&chan[id], ctl->nu[id], ctl->window[id]))) { free(chan); jur_model_destroy(m); return rc; }
  rc = upload(&m->d_chan, chan, sizeof(jur_chan_t) * ctl->nd);
  free(chan);
  if (rc) { jur_model_destroy(m); return rc; }
  if ((rc = upload(&m->d_sr, tb->sr, sizeof(double) * JUR_TBLNS * ctl->nd))) { jur_model_destroy(m); return rc; }

  /* tables restricted to the (ng, nd) the control block uses */
  jur_tables_t sub = *tb;
  jur_pair_t *subpair = NULL;
  if (tb->nd != ctl->nd || tb->ng != ctl->ng) {
    subpair = (jur_pair_t *)calloc((size_t)ctl->ng * ctl->nd + 1, sizeof(jur_pair_t));
    if (!subpair) { jur_model_destroy(m); return JUR_ENOMEM; }
    for (int g = 0; g < ctl->ng; g++)
      for (int d = 0; d < ctl->nd; d++) subpair[(size_t)g * ctl->nd + d] = tb->pair[(size_t)g * tb->nd + d];
    sub.pair = subpair; sub.ng = ctl->ng; sub.nd = ctl->nd;
  }
  jur_flat_t fl;
  rc = jur_tables_flatten(&sub, &fl);
  free(subpair);
  if (rc) { jur_model_destroy(m); return rc; }
  long const npair = (long)ctl->ng * ctl->nd;
  rc = upload(&m->d_pair, fl.pair, sizeof(jur_int2) * (npair > 0 ? npair : 1));
  if (!rc) rc = upload(&m->d_pair_e0, fl.pair_e0, sizeof(long long) * (npair > 0 ? npair : 1));
  if (!rc) rc = upload(&m->d_lvl, fl.lvl, sizeof(jur_lvl_t) * (fl.nlevel + 2));
  if (!rc) rc = upload(&m->d_crv, fl.crv, sizeof(jur_crv_t) * (fl.ncurve + 2));
  if (!rc) rc = upload(&m->d_ue, fl.ue, sizeof(jur_ue_t) * (fl.nentry + 2));
  if (!rc && fl.strict) {       /* bracket slopes for the strict-table look-up: 16 B per entry next to the 8 B of the entry */
    rc = upload(&m->d_sl, NULL, sizeof(jur_sl_t) * (fl.nentry + 2));
    if (!rc && (jurk_fill_slopes((jur_ue_t const *)m->d_ue, (jur_sl_t *)m->d_sl, fl.nentry + 2, NULL) || hipStreamSynchronize(NULL) != hipSuccess)) {
      jur_set_error("cannot form the bracket slopes of the tables");
      rc = JUR_EHIP;
    }
  }
  if (!rc && fl.strict && !getenv("JUR_EGA_NO_REC")) {   /* bracket records: entries i, i+1 and both slopes in 32 bytes (what the
                                                            batched look-up kernel reads; JUR_EGA_NO_REC: the two arrays, A/B) */
    rc = upload(&m->d_rec, NULL, sizeof(jur_rec_t) * (fl.nentry + 2));
    if (!rc && (jurk_fill_records((jur_ue_t const *)m->d_ue, (jur_sl_t const *)m->d_sl, (jur_rec_t *)m->d_rec, fl.nentry + 1, NULL) ||
                hipStreamSynchronize(NULL) != hipSuccess)) {
      jur_set_error("cannot form the bracket records of the tables");
      rc = JUR_EHIP;
    }
    v->rec = (jur_rec_t const *)m->d_rec;
    if (!rc) {   /* every kernel of the strict-table arithmetic reads the records; the slope array was their source only */
      (void)hipFree(m->d_sl);
      m->d_sl = NULL;
    }
  }
  if (!rc && fl.strict && npair > 0) {   /* which channels of a gas stand on one (p, T) grid: jur_model_set_ega_group */
    m->grid_cls = (int *)malloc(sizeof(int) * npair);
    m->grid_all = (unsigned char *)malloc(npair);
    m->h_pair_e0 = (long long *)malloc(sizeof(long long) * npair);
    if (!m->grid_cls || !m->grid_all || !m->h_pair_e0) rc = JUR_ENOMEM;
    else {
      memcpy(m->h_pair_e0, fl.pair_e0, sizeof(long long) * npair);
      rc = jur_flat_grid_classes(&fl, ctl->ng, ctl->nd, m->grid_cls, m->grid_all);
    }
  }
  v->sorted_tables = fl.sorted;
  v->strict_tables = fl.strict;
  v->max_pair_curves = fl.max_pair_curves;
  m->table_bytes = (long)((sizeof(jur_ue_t) + (m->d_sl ? sizeof(jur_sl_t) : 0) + (m->d_rec ? sizeof(jur_rec_t) : 0)) * fl.nentry + 16 * fl.ncurve + 16 * fl.nlevel + 8 * npair);
  jur_flat_free(&fl);
  if (rc) { jur_model_destroy(m); return rc; }
  v->chan = (jur_chan_t const *)m->d_chan;
  v->sr = (double const *)m->d_sr;
  v->pair = (jur_int2 const *)m->d_pair;
  v->pair_e0 = (long long const *)m->d_pair_e0;
  v->lvl = (jur_lvl_t const *)m->d_lvl;
  v->crv = (jur_crv_t const *)m->d_crv;
  v->ue = (jur_ue_t const *)m->d_ue;
  v->sl = (jur_sl_t const *)m->d_sl;
  if ((rc = jur_model_set_arithmetic(m, getenv("JUR_EGA_NO_RCP") ? JUR_ARITH_EXACT : JUR_ARITH_FAST))) { jur_model_destroy(m); return rc; }
  /* measured slower than one pair per workgroup: only on request */
  if (getenv("JUR_EGA_GROUP") && atoi(getenv("JUR_EGA_GROUP")) >= 2 && (rc = jur_model_set_ega_group(m, atoi(getenv("JUR_EGA_GROUP"))))) {
    jur_model_destroy(m);
    return rc;
  }

  m->nfield = JUR_F_K + v->nw + v->ng;
  m->chunk_rays = 1 << 21;      /* upper bound; the workspace budget sets the real size (1.4 M rays for 96 KB per ray).
                                   Measured: 7.7 M rays/s at 1 M rays per launch vs 6.8 M at 131072 (tails, launch fill) */
  m->sort_rays = 1;
  m->ws_budget = 128L << 30;    /* of 288 GB HBM; C3 needs 96 KB per ray */
  m->compact_ws = getenv("JUR_NO_COMPACT_WS") ? 0 : 1;
  m->trace_mult = 1;
  m->pencil_rays = 10000;       /* measured crossover with the batched kernels (4 channels x 5 emitters) */
  m->pencil_rb = 0;
  if (getenv("JUR_PENCIL_RAYS")) m->pencil_rays = atol(getenv("JUR_PENCIL_RAYS"));
  if (getenv("JUR_PENCIL_RB")) m->pencil_rb = atoi(getenv("JUR_PENCIL_RB"));
  /* tuning overrides for experiments; the setters of the API do the same */
  if (getenv("JUR_CHUNK_RAYS") && atoi(getenv("JUR_CHUNK_RAYS")) >= 64) m->chunk_rays = (atoi(getenv("JUR_CHUNK_RAYS")) + 63) / 64 * 64;
  if (getenv("JUR_TRACE_MULT") && atoi(getenv("JUR_TRACE_MULT")) >= 1) m->trace_mult = atoi(getenv("JUR_TRACE_MULT"));
  if (getenv("JUR_WS_GIB") && atoi(getenv("JUR_WS_GIB")) >= 1) m->ws_budget = (long)atoi(getenv("JUR_WS_GIB")) << 30;
  if ((rc = create_streams(m))) { jur_model_destroy(m); return rc; }
  if ((rc = upload((void **)&m->d_status, NULL, sizeof(int)))) { jur_model_destroy(m); return rc; }
  if (hipMemset(m->d_status, 0, sizeof(int)) != hipSuccess) { jur_set_error("cannot clear the status word"); jur_model_destroy(m); return JUR_EHIP; }
  *out = m;
  return JUR_OK;
}

/* JUR_ARITH_FAST (default): on strictly increasing tables whose descriptors fit the LDS staging the look-up uses bracket
 * slopes, reciprocal bracket widths and carries the path transmittance as 1 - eps (~1e-13 from the reference's
 * divisions); JUR_ARITH_EXACT: the reference's divisions operand for operand (what every other table gets anyway).
 * Takes effect with the next call; calls in flight are the caller's to wait for. */
int jur_model_set_arithmetic(jur_model_t *m, int mode) {
  if (mode != JUR_ARITH_FAST && mode != JUR_ARITH_EXACT) { jur_set_error("set_arithmetic: JUR_ARITH_FAST or JUR_ARITH_EXACT"); return JUR_EINVAL; }
  jur_view_t *v = &m->view;
  /* (the staging size of jurk_launch_ega: 24 B per level and curve of the largest pair, at most 48 KB) */
  int const lds_ok = v->max_pair_curves > 0 && 24L * JUR_TBLNP + 24L * v->max_pair_curves <= 48 * 1024 && !getenv("JUR_EGA_NO_LDS");
  m->arith = mode;
  v->fast_arith = (mode == JUR_ARITH_FAST) && v->strict_tables && (v->sl || v->rec) && lds_ok;
  return JUR_OK;
}
int jur_model_arithmetic(jur_model_t const *m) { return m->arith; }

/* Channels of a gas whose tables stand on one (p, T) grid walked together by one lane (jur_ega_group_kernel), at most
 * nch (2 .. JUR_EGA_NCH) per lane; nch < 2: one (channel, gas) pair per workgroup (jur_ega_kernel, the default).
 * Strict tables only; results are the same doubles either way. */
int jur_model_set_ega_group(jur_model_t *m, int nch) {
  if (m->shared_tables) { jur_set_error("set_ega_group: a lane shares its tables with another model"); return JUR_EINVAL; }
  HIPCHK(hipSetDevice(m->device));
  jur_view_t *v = &m->view;
  if (m->have_done) HIPCHK(hipEventSynchronize(m->ev_done));      /* launches in flight still read the old items */
  if (m->d_items) { (void)hipFree(m->d_items); m->d_items = NULL; }
  v->ega_items = NULL; v->ega_nitems = 0; v->ega_nch = 0;
  if (nch < 2 || !v->strict_tables || !m->grid_cls) return JUR_OK;      /* (runs only while fast_arith is on) */
  jur_item_t *items = NULL;
  int nitems = 0, max_nch = 0;
  int rc = jur_group_items(v->ng, v->nd, nch, m->grid_cls, m->grid_all, m->h_pair_e0, &items, &nitems, &max_nch);
  if (!rc && max_nch >= 2) {
    rc = upload(&m->d_items, items, sizeof(jur_item_t) * nitems);
    if (!rc) { v->ega_items = (jur_item_t const *)m->d_items; v->ega_nitems = nitems; v->ega_nch = max_nch; }
  }
  free(items);
  return rc;
}

/* channels per lane the next call's look-up kernel walks (0: one pair per workgroup) */
int jur_model_ega_group(jur_model_t const *m) { return (m->view.ega_items && m->view.fast_arith && m->view.rec) ? m->view.ega_nch : 0; }

int jur_model_create_from_files(jur_model_t **out, ctl_t const *ctl, int device) {
  *out = NULL;
  int rc = check_ctl(ctl);
  if (rc) return rc;
  /* READ_BINARY / WRITE_BINARY as upstream's init_tbl (jurassic.c:312-320, 669-671): != 0 try the
   * cache first, > 0 insist on it; write it after parsing the ASCII files when WRITE_BINARY != 0. */
  char cache[256];
  jur_tables_cache_filename(cache, sizeof cache, ctl);
  jur_tables_t *tb = NULL;
  if (ctl->read_binary) {
    rc = jur_tables_load(&tb, ctl, cache);
    if (rc == JUR_OK) printf("matching binary tables file found\n");
    else if (ctl->read_binary > 0) return rc;
  }
  if (!tb) {
    tb = jur_tables_new(ctl->ng, ctl->nd);
    if (!tb) return JUR_ENOMEM;
    rc = jur_tables_read_ascii(tb, ctl);
    if (rc >= 0) {
      int const found = rc;
      if (found < ctl->ng * ctl->nd) printf("Warning! %d files were not found!\n", ctl->ng * ctl->nd - found);
      rc = jur_tables_read_filters(tb, ctl);
    }
    if (rc == JUR_OK && ctl->write_binary && jur_tables_save(tb, ctl, cache) != JUR_OK)
      printf("Warning! could not write %s: %s\n", cache, jur_last_error());
  }
  if (rc == JUR_OK) rc = jur_model_create(out, ctl, tb, device);
  jur_tables_free(tb);
  return rc;
}

void jur_model_destroy(jur_model_t *m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  if (m->shared_tables) m->d_chan = m->d_sr = m->d_pair = m->d_pair_e0 = m->d_lvl = m->d_crv = m->d_ue = m->d_sl = m->d_items = m->d_rec = NULL;
  void *ptrs[] = {m->d_chan, m->d_sr, m->d_pair, m->d_pair_e0, m->d_lvl, m->d_crv, m->d_ue, m->d_sl, m->d_items, m->d_rec, m->d_atm, m->d_order, m->d_sort_tmp,
                  m->d_los, m->d_eps, m->d_np, m->d_tsurf, m->d_status, m->d_io, m->d_io_np, m->d_fov, m->d_kq};
  for (size_t i = 0; i < sizeof ptrs / sizeof ptrs[0]; i++)
    if (ptrs[i]) (void)hipFree(ptrs[i]);
  if (m->h_io) (void)hipHostFree(m->h_io);
  if (m->h_pkg) (void)hipHostFree(m->h_pkg);
  if (m->h_kq) (void)hipHostFree(m->h_kq);
  if (m->h_status) (void)hipHostFree(m->h_status);
  free(m->h_atm);
  if (!m->shared_tables) { free(m->grid_cls); free(m->grid_all); free(m->h_pair_e0); }
  if (m->stream) (void)hipStreamDestroy(m->stream);
  if (m->stream2) (void)hipStreamDestroy(m->stream2);
  if (m->ev_mask) (void)hipEventDestroy(m->ev_mask);
  if (m->ev_trace) (void)hipEventDestroy(m->ev_trace);
  if (m->ev_done) (void)hipEventDestroy(m->ev_done);
  if (m->evpool) {
    for (int i = 0; i < 2 * JUR_MAX_TIMED; i++) (void)hipEventDestroy(m->evpool[i]);
    free(m->evpool);
    free(m->evkind);
  }
  free(m->ctl);
  free(m);
}

/* ---- hydrostatic adjustment on the host (jr_common.h:212-217, 713-761) ------ */
static double gravity(double z, double lat) {
  double const deg2rad = M_PI / 180., x = sin(lat * deg2rad), y = sin(2 * lat * deg2rad);
  return 9.780318 * (1. + 0.0053024 * x * x - 5.8e-6 * y * y) - 3.086e-3 * z;
}

static double lin(double x0, double y0, double x1, double y1, double x) { return y0 + (x - x0) * (y1 - y0) / (x1 - x0); }

static void layer_pressure(double const *z, double const *t, double const *qh2o, double *p, double lat, int from, int to) {
  /* integrates the hydrostatic equation across one layer with 20 sub-points */
  int const npts = 20;
  double const mmair = 28.96456e-3, mmh2o = 18.0153e-3;
  double mean = 0., e = 0.;
  for (int i = 0; i < npts; i++) {
    double const zz = lin(0.0, z[from], npts - 1.0, z[to], (double)i);
    double const grav = gravity(zz, lat);
    if (qh2o) e = lin(0.0, qh2o[from], npts - 1.0, qh2o[to], (double)i);
    double const temp = lin(0.0, t[from], npts - 1.0, t[to], (double)i);
    mean += (e * mmh2o + (1 - e) * mmair) * grav / (JUR_MOLAR_GAS * temp * npts);
  }
  p[to] = p[from] * exp(-1000 * mean * (z[to] - z[from]));
}

static void hydrostatic(ctl_t const *ctl, int ig_h2o, int n, double const *z, double const *lat, double const *t,
                        double const *qh2o, double *p) {
  double dzmin = 1e99;
  int ipref = 0;
  for (int ip = 0; ip < n; ip++) {
    double const dz = fabs(z[ip] - ctl->hydz);
    if (dz < dzmin) { dzmin = dz; ipref = ip; }
  }
  double const *q = (ig_h2o >= 0) ? qh2o : NULL;
  for (int ip = ipref + 1; ip < n; ip++) layer_pressure(z, t, q, p, lat[ipref], ip - 1, ip);
  for (int ip = ipref - 1; ip >= 0; ip--) layer_pressure(z, t, q, p, lat[ipref], ip + 1, ip);
}

/* Host image of an atmosphere: rows time, z, lon, lat, p, T, q[ng], k[nw], each n long. */
static void pack_atm_rows(jur_model_t const *m, atm_t const *atm, double *h, size_t stride, size_t at) {
  int const n = atm->np, ng = m->view.ng, nw = m->view.nw;
  memcpy(h + 0 * stride + at, atm->time, sizeof(double) * n);
  memcpy(h + 1 * stride + at, atm->z, sizeof(double) * n);
  memcpy(h + 2 * stride + at, atm->lon, sizeof(double) * n);
  memcpy(h + 3 * stride + at, atm->lat, sizeof(double) * n);
  memcpy(h + 4 * stride + at, atm->p, sizeof(double) * n);
  memcpy(h + 5 * stride + at, atm->t, sizeof(double) * n);
  for (int g = 0; g < ng; g++) memcpy(h + (6 + (size_t)g) * stride + at, atm->q[g], sizeof(double) * n);
  for (int w = 0; w < nw; w++) memcpy(h + (6 + (size_t)ng + w) * stride + at, atm->k[w], sizeof(double) * n);
}

/* hydrostatic adjustment of one packed atmosphere of n points starting at `at` (CPUdrivers.c:98-103) */
static void hydrostatic_rows(jur_model_t const *m, double *h, size_t stride, size_t at, int n) {
  if (m->ctl->hydz < 0) return;
  int const ig = m->view.ig_h2o;
  hydrostatic(m->ctl, ig, n, h + 1 * stride + at, h + 3 * stride + at, h + 5 * stride + at,
              ig >= 0 ? h + (6 + (size_t)ig) * stride + at : NULL, h + 4 * stride + at);
}

/* upload packed rows [6+ng+nw][n] and derive what the kernels want to know about them */
static int upload_atm_rows(jur_model_t *m, double const *h, long n) {
  HIPCHK(hipSetDevice(m->device));
  jur_view_t *v = &m->view;
  int const ng = v->ng, nw = v->nw;
  size_t const nrow = 6 + (size_t)ng + nw;
  if (n > m->atm_cap) {
    if (m->d_atm) (void)hipFree(m->d_atm);
    m->d_atm = NULL;
    m->atm_cap = 0;
    HIPCHK(hipMalloc(&m->d_atm, sizeof(double) * (nrow + 1) * (size_t)n));   /* + one row for atm_pslope */
    m->atm_cap = n;
  }
  /* Kernels of an earlier jur_formod_device call may still be reading the atmosphere on the CALLER's stream: wait
   * for the event that call left behind on it.  The event is the model's own; the caller's stream may have been
   * destroyed since (destruction completes its work, and the event with it). */
  if (m->have_done) {
    HIPCHK(hipEventSynchronize(m->ev_done));
    m->have_done = 0;
  }
  HIPCHK(hipMemcpyAsync(m->d_atm, h, sizeof(double) * nrow * (size_t)n, hipMemcpyHostToDevice, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  double const *time = h, *z = h + (size_t)n;
  m->atm_slices = 1;
  m->atm_zmin = m->atm_zmax = z[0];
  for (long i = 1; i < n; i++) { if (z[i] < m->atm_zmin) m->atm_zmin = z[i]; if (z[i] > m->atm_zmax) m->atm_zmax = z[i]; }
  v->atm_sorted = 1;
  v->atm_maxslice = 1;
  for (long i = 1, dir = 0, run = 1; i < n; i++) {
    run = (time[i] != time[i - 1]) ? 1 : run + 1;
    if (run > v->atm_maxslice) v->atm_maxslice = (int)run;
    if (time[i] != time[i - 1]) {
      m->atm_slices++;
      if (time[i] < time[i - 1]) v->atm_sorted = 0;
      dir = 0;
      continue;
    }
    long const d = (z[i] > z[i - 1]) - (z[i] < z[i - 1]);
    if (d == 0 || (dir != 0 && d != dir)) v->atm_sorted = 0;
    dir = d;
  }
  double const *d = (double const *)m->d_atm;
  v->atm_np = (int)n;
  v->atm_time = d; v->atm_z = d + (size_t)n; v->atm_lon = d + 2 * (size_t)n; v->atm_lat = d + 3 * (size_t)n;
  v->atm_p = d + 4 * (size_t)n; v->atm_t = d + 5 * (size_t)n;
  v->atm_q = d + 6 * (size_t)n; v->atm_k = d + (6 + (size_t)ng) * n;
  v->atm_pslope = d + nrow * (size_t)n;
  int const ek = jurk_prepare_atm(v, (double *)m->d_atm + nrow * (size_t)n, m->stream);
  if (ek || hipStreamSynchronize(m->stream) != hipSuccess) { jur_set_error("atm preparation kernel failed"); return JUR_EHIP; }
  return JUR_OK;
}

int jur_model_set_atm(jur_model_t *m, atm_t const *atm) {
  if (!m || !atm || atm->np < 2 || atm->np > JUR_NP) { jur_set_error("set_atm: need 2..%d atmospheric points", JUR_NP); return JUR_EINVAL; }
  int const n = atm->np;
  size_t const nrow = 6 + (size_t)m->view.ng + m->view.nw;
  double *h = (double *)malloc(sizeof(double) * nrow * n);
  if (!h) return JUR_ENOMEM;
  pack_atm_rows(m, atm, h, (size_t)n, 0);
  /* The same atmosphere again (a caller looping over observation packages, formod.c:100, or the lanes of the
   * drop-in entry): what is on the device already is what this upload would put there.  Compared before the
   * hydrostatic step, which is a function of these rows and ctl->hydz. */
  if (m->h_atm && m->h_atm_n == n && m->h_atm_hydz == m->ctl->hydz && m->view.atm_np == n &&
      0 == memcmp(m->h_atm, h, sizeof(double) * nrow * n)) {
    free(h);
    return JUR_OK;
  }
  if ((long)(nrow * n) > m->h_atm_cap) {
    free(m->h_atm);
    m->h_atm = (double *)malloc(sizeof(double) * nrow * n);
    m->h_atm_cap = m->h_atm ? (long)(nrow * n) : 0;
  }
  m->h_atm_n = 0;
  if (m->h_atm) { memcpy(m->h_atm, h, sizeof(double) * nrow * n); m->h_atm_n = n; m->h_atm_hydz = m->ctl->hydz; }
  hydrostatic_rows(m, h, (size_t)n, 0, n);   /* on the private copy; the caller's atm is not modified */
  int const rc = upload_atm_rows(m, h, n);
  if (rc) m->h_atm_n = 0;
  free(h);
  return rc;
}

/* ---- workspace --------------------------------------------------------------- */
static void free_workspace(jur_model_t *m) {
  if (m->d_los) (void)hipFree(m->d_los);
  if (m->d_eps) (void)hipFree(m->d_eps);
  if (m->d_np) (void)hipFree(m->d_np);
  if (m->d_tsurf) (void)hipFree(m->d_tsurf);
  if (m->d_tile_np) (void)hipFree(m->d_tile_np);
  if (m->d_eps_off) (void)hipFree(m->d_eps_off);
  if (m->h_tile) (void)hipHostFree(m->h_tile);
  m->d_los = NULL; m->d_eps = NULL; m->d_np = NULL; m->d_tsurf = NULL; m->d_tile_np = NULL; m->d_eps_off = NULL; m->h_tile = NULL;
  m->los_bytes = 0; m->ws_rays = 0; m->ws_trace_rays = 0;
}

static int ensure_workspace(jur_model_t *m, long nr) {
  /* bytes per ray: LOS fields (per traced ray), one double per (channel, gas, point) (per integrated ray).
   * Tracing is latency-bound and wants many rays per launch, so its launches cover Rt = trace_mult * R rays
   * while the ega/combine launches cover R. */
  long const per_ray_los = (long)sizeof(double) * m->nfield * JUR_NLOS;
  long const per_ray_eps = (long)sizeof(double) * m->view.nd * (m->view.ng > 0 ? m->view.ng : 1) * JUR_NLOS;
  long budget = m->ws_budget;
  int asked = 0;
  for (int attempt = 0;; attempt++) {
    long R = m->chunk_rays;
    long const fit = budget / (per_ray_los + per_ray_eps);
    if (R > fit) R = fit / 64 * 64;
    if (R < 64) R = 64;
    if (nr <= R) R = (nr + 63) / 64 * 64;
    else {  /* several launches: equal shares instead of full chunks and a short tail */
      long const nchunk = (nr + R - 1) / R;
      R = ((nr + nchunk - 1) / nchunk + 63) / 64 * 64;
    }
    long mult = m->trace_mult > 0 ? m->trace_mult : 1;
    while (mult > 1 && (per_ray_los * mult + per_ray_eps) * R > budget) mult--;
    long Rt = R * mult;
    if (nr < Rt) Rt = (nr + R - 1) / R * R;
    int compact = 0;
    if (nr > R && m->compact_ws) {
      /* Several integration launches with JUR_NLOS points set aside per ray.  Rays rarely have that many (nadir
       * 181 .. 182, limb 122 .. 393 of 400: SURVEY.md section 6): trace as many rays as the budget allows first -- the
       * LOS rows are the small part -- and give the rest of it to the transmittances, whose tiles jur_formod_device
       * then lays out by the longest path of each tile.  R stays the capacity in rays of JUR_NLOS points. */
      long Rt_c = (nr + 63) / 64 * 64;
      if (per_ray_los * Rt_c > budget / 2) Rt_c = budget / 2 / per_ray_los / 64 * 64;
      long R_c = (budget - per_ray_los * Rt_c) / per_ray_eps / 64 * 64;
      if (R_c > m->chunk_rays) R_c = m->chunk_rays;            /* (the tuning knob: rays of JUR_NLOS points per launch) */
      if (R_c >= 64 && Rt_c >= R_c) { R = R_c; Rt = Rt_c; compact = 1; }
    }
    if (R <= m->ws_rays && Rt <= m->ws_trace_rays) {   /* the held workspace serves (smaller strides fit inside it) */
      m->use_rays = R;
      m->use_trace_rays = Rt;
      m->use_compact = compact;
      return JUR_OK;
    }
    if (!asked) {  /* about to allocate: never plan beyond what the device can give right now (another allocator
                      in the process may hold part of the HBM) */
      size_t free_b = 0, total_b = 0;
      asked = 1;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        long const avail = (long)free_b + m->los_bytes - (2L << 30);
        if (avail > (64L << 20) && avail < budget) { budget = avail; continue; }
      } else (void)hipGetLastError();
    }
    free_workspace(m);
    hipError_t e = hipMalloc((void **)&m->d_los, (size_t)per_ray_los * Rt);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_eps, (size_t)per_ray_eps * R);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_np, sizeof(int) * Rt);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_tsurf, sizeof(double) * Rt);
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_tile_np, sizeof(int) * (Rt / 64 + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_eps_off, sizeof(int) * (Rt / 64 + 1));
    if (e == hipSuccess) e = hipHostMalloc((void **)&m->h_tile, sizeof(int) * 2 * (Rt / 64 + 1), hipHostMallocDefault);
    if (e == hipSuccess) {
      m->los_bytes = per_ray_los * Rt + per_ray_eps * R;
      m->ws_rays = R;
      m->ws_trace_rays = Rt;
      m->use_rays = R;
      m->use_trace_rays = Rt;
      m->use_compact = compact;
      return JUR_OK;
    }
    (void)hipGetLastError();
    free_workspace(m);
    if (R <= 64 || attempt >= 24) {
      jur_set_error("cannot allocate the workspace (%ld B per ray, even for %ld rays): %s", per_ray_los + per_ray_eps, R,
                    hipGetErrorString(e));
      return JUR_EHIP;
    }
    budget = (per_ray_los + per_ray_eps) * R / 2;   /* allocation refused: try again with half the rays per launch */
  }
}

long jur_model_workspace_bytes(jur_model_t const *m) { return m->los_bytes; }
long jur_model_table_bytes(jur_model_t const *m) { return m->table_bytes; }

/* What a caller that plans its memory wants to know about a device before it allocates: PCI bus id (text), free and
 * total bytes right now. */
int jur_device_info(int device, char *pci_bus_id, int len, size_t *free_bytes, size_t *total_bytes) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { jur_set_error("device %d not available", device); return JUR_ENODEV; }
  HIPCHK(hipSetDevice(device));
  if (pci_bus_id && len > 0) HIPCHK(hipDeviceGetPCIBusId(pci_bus_id, len, device));
  size_t f = 0, t = 0;
  HIPCHK(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return JUR_OK;
}
int jur_model_device(jur_model_t const *m) { return m->device; }
int jur_model_nd(jur_model_t const *m) { return m->view.nd; }
void *jur_model_stream(jur_model_t const *m) { return (void *)m->stream; }
int *jur_model_status_word(jur_model_t const *m) { return m->d_status; }
void jur_model_cost_params(jur_model_t const *m, double *rayds, double *raydz, double *zmin, double *zmax) {
  *rayds = m->view.rayds; *raydz = m->view.raydz; *zmin = m->atm_zmin; *zmax = m->atm_zmax;
}
int jur_model_chunk_rays(jur_model_t const *m) { return m->chunk_rays; }
long jur_model_last_launches(jur_model_t const *m) { return m->n_launch_ega; }
int jur_model_set_compact_workspace(jur_model_t *m, int on) { m->compact_ws = on ? 1 : 0; return JUR_OK; }

int jur_model_set_sort_rays(jur_model_t *m, int on) { m->sort_rays = on ? 1 : 0; return JUR_OK; }

int jur_model_set_workspace_budget(jur_model_t *m, long bytes) {
  if (bytes < (64L << 20)) { jur_set_error("workspace budget below 64 MiB"); return JUR_EINVAL; }
  m->ws_budget = bytes;
  return JUR_OK;
}

int jur_model_set_trace_multiple(jur_model_t *m, int mult) {
  if (mult < 1 || mult > 64) { jur_set_error("trace multiple must be in 1..64"); return JUR_EINVAL; }
  m->trace_mult = mult;
  return JUR_OK;
}

int jur_model_set_chunk_rays(jur_model_t *m, int rays) {
  if (rays < 64 || rays > (1 << 22)) { jur_set_error("chunk_rays must be in 64..4194304"); return JUR_EINVAL; }
  m->chunk_rays = (rays + 63) / 64 * 64;
  return JUR_OK;
}

int jur_model_set_pencil(jur_model_t *m, long max_rays, int rays_per_group) {
  if (max_rays < 0 || rays_per_group < 0 || rays_per_group > 64) { jur_set_error("set_pencil: max_rays >= 0, rays_per_group in 0..64"); return JUR_EINVAL; }
  m->pencil_rays = max_rays;
  m->pencil_rb = rays_per_group;
  return JUR_OK;
}

void jur_tune_trace(int lanes_per_ray) { jurk_tune_trace(lanes_per_ray); }

void jur_tune_combine(int channels_per_group, int sync_segments, long min_lanes) {
  jurk_tune_combine(channels_per_group, sync_segments, min_lanes);
}

int jur_model_enable_timing(jur_model_t *m, int on) {
  HIPCHK(hipSetDevice(m->device));
  if (on && !m->evpool) {
    m->evpool = (hipEvent_t *)calloc(2 * JUR_MAX_TIMED, sizeof(hipEvent_t));
    m->evkind = (unsigned char *)calloc(JUR_MAX_TIMED, 1);
    if (!m->evpool || !m->evkind) return JUR_ENOMEM;
    for (int i = 0; i < 2 * JUR_MAX_TIMED; i++) HIPCHK(hipEventCreate(&m->evpool[i]));
  }
  m->timing = on;
  m->ntimed = 0;
  return JUR_OK;
}

/* Sums the event-bracketed durations of the launches recorded since the last
 * call (at most JUR_MAX_TIMED chunks), then starts over.
 * [0] trace, [1] ega, [2] combine. */
int jur_model_last_kernel_ms(jur_model_t *m, double out_ms[3], long out_launches[3]) {
  for (int k = 0; k < 3; k++) { out_ms[k] = 0; out_launches[k] = 0; }
  if (!m->evpool) return JUR_OK;
  HIPCHK(hipSetDevice(m->device));
  for (int i = 0; i < m->ntimed; i++) {
    float ms = 0;
    HIPCHK(hipEventSynchronize(m->evpool[2 * i + 1]));
    HIPCHK(hipEventElapsedTime(&ms, m->evpool[2 * i], m->evpool[2 * i + 1]));
    if (m->evkind[i] < 3) { out_ms[m->evkind[i]] += ms; out_launches[m->evkind[i]]++; }
    else { m->pencil_ms += ms; m->pencil_launches++; }
  }
  m->ntimed = 0;
  return JUR_OK;
}

/* the fused kernel's share of the launches timed since the last call of this function (call
 * jur_model_last_kernel_ms first: it collects the events) */
int jur_model_last_pencil_ms(jur_model_t *m, double *out_ms, long *out_launches) {
  *out_ms = m->pencil_ms;
  *out_launches = m->pencil_launches;
  m->pencil_ms = 0;
  m->pencil_launches = 0;
  return JUR_OK;
}

static int ensure_sort_buffers(jur_model_t *m, long nr) {
  long const need = jurk_sort_tmp_bytes(nr);
  if (nr > m->order_cap || need > m->sort_tmp_bytes) {
    if (m->d_order) (void)hipFree(m->d_order);
    if (m->d_sort_tmp) (void)hipFree(m->d_sort_tmp);
    m->d_order = NULL; m->d_sort_tmp = NULL; m->order_cap = 0; m->sort_tmp_bytes = 0;
    HIPCHK(hipMalloc((void **)&m->d_order, sizeof(int) * (size_t)nr));
    HIPCHK(hipMalloc(&m->d_sort_tmp, (size_t)need));
    m->order_cap = nr; m->sort_tmp_bytes = need;
  }
  return JUR_OK;
}

/* Allocate everything a later jur_formod_device(m, nr, ...) needs, so that the call itself only
 * enqueues kernels (it can then be captured into a HIP graph). */
int jur_model_reserve(jur_model_t *m, long nr) {
  if (!m || nr < 1 || nr > 0x7fffffffL) { jur_set_error("reserve: bad ray count"); return JUR_EINVAL; }
  HIPCHK(hipSetDevice(m->device));
  if (pencil_rays_per_group(m, nr) > 0) return JUR_OK;      /* the fused kernel keeps its state in LDS */
  int rc = ensure_workspace(m, nr);
  if (rc == JUR_OK && m->sort_rays && nr > 64) rc = ensure_sort_buffers(m, nr);
  return rc;
}

/* ---- forward model ------------------------------------------------------------ */
/* Rays per workgroup of the fused kernel for a call of nr rays, or 0 when the call goes to the batched kernels
 * (too many rays, or more (channel, gas) chains per ray than the LDS rings hold). */
static int pencil_rays_per_group(jur_model_t const *m, long nr) {
  if (m->pencil_rays <= 0 || nr > m->pencil_rays) return 0;
  long const npair = (long)m->view.nd * (m->view.ng > 0 ? m->view.ng : 1);
  int rb = m->pencil_rb;
  if (rb <= 0) {
    /* Measured (tools/bench_small.py, profiles/r02_small_call_latency.json): with 10 chains per ray the fused kernel
     * beats the batched ones up to ~10 000 rays, with 20 up to ~4 500: the chains' look-ups (four lanes each) must
     * keep up with the tracer.  Rays per workgroup: about 136 workgroups per call and at most 8 rays for up to 12
     * chains per ray, 1 or 4 rays for up to 25, fewer beyond, so that a workgroup's chain lanes stay within its
     * eight look-up wavefronts. */
    if (nr * npair > 100000) return 0;
    if (npair <= 12) {
      rb = 1;
      while (rb < 8 && nr > 136L * rb) rb *= 2;
    } else if (npair <= 25) rb = (nr <= 600) ? 1 : 4;      /* (2 rays per workgroup measured slower than 1 and 4) */
    else rb = (npair <= 50 && nr > 600) ? 2 : 1;
  }
  if (rb > 64) rb = 64;
  while (rb > 1 && jurk_pencil_lds_bytes(&m->view, rb) <= 0) rb /= 2;
  return jurk_pencil_lds_bytes(&m->view, rb) > 0 ? rb : 0;
}

/* Leaves the model's event behind the work just enqueued on the caller's stream (see upload_atm_rows).  Not while
 * the stream is being captured into a graph: the graph's launches are the caller's to order against later uploads. */
static int record_done(jur_model_t *m, hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); st = hipStreamCaptureStatusNone; }
  if (st != hipStreamCaptureStatusNone) return JUR_OK;
  HIPCHK(hipEventRecord(m->ev_done, s));
  m->have_done = 1;
  return JUR_OK;
}

int jur_formod_device(jur_model_t *m, long nr, double const *d_geom, double *d_rad, double *d_tau, double *d_tp,
                      int *d_np, int *d_status, void *stream) {
  return jur_formod_device_ld(m, nr, d_geom, nr, d_rad, d_tau, d_tp, nr, d_np, d_status, stream);
}

int jur_formod_device_ld(jur_model_t *m, long nr, double const *d_geom, long ldg, double *d_rad, double *d_tau, double *d_tp,
                         long ldtp, int *d_np, int *d_status, void *stream) {
  if (!m || nr < 0) { jur_set_error("formod_device: bad arguments"); return JUR_EINVAL; }
  if (nr == 0) return JUR_OK;
  if (nr > 0x7fffffffL) { jur_set_error("formod_device: at most 2^31-1 rays per call"); return JUR_EINVAL; }
  if (m->view.atm_np < 2) { jur_set_error("formod_device: no atmosphere set"); return JUR_EINVAL; }
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = (hipStream_t)stream;
  int const rb = pencil_rays_per_group(m, nr);
  if (rb > 0) {
    /* a package-sized call: the whole path in one launch, a workgroup per rb rays, LOS state in LDS */
    jur_chunk_t c;
    memset(&c, 0, sizeof c);
    c.n = (int)nr;
    for (int k = 0; k < 7; k++) c.geom[k] = d_geom + (size_t)k * ldg;
    for (int k = 0; k < 3; k++) c.tp[k] = d_tp + (size_t)k * ldtp;
    c.rad = d_rad;
    c.tau = d_tau;
    c.np_out = d_np;
    c.status = d_status ? d_status : m->d_status;
    if (m->host_call) HIPCHK(hipStreamWaitEvent(s, m->ev_mask, 0));
    int const ti = (m->timing && m->ntimed < JUR_MAX_TIMED) ? m->ntimed++ : -1;
    if (ti >= 0) { m->evkind[ti] = 3; HIPCHK(hipEventRecord(m->evpool[2 * ti], s)); }
    int const e = jurk_launch_pencil(&m->view, &c, rb, s);
    if (e) { jur_set_error("fused kernel launch failed: %s", hipGetErrorString((hipError_t)e)); return JUR_EHIP; }
    if (ti >= 0) HIPCHK(hipEventRecord(m->evpool[2 * ti + 1], s));
    if (m->host_call) HIPCHK(hipEventRecord(m->ev_trace, s));
    return record_done(m, s);
  }
  int rc = ensure_workspace(m, nr);
  if (rc) return rc;
  long const R = m->use_rays;
  int const *order = NULL;
  if (m->sort_rays && nr > 64) {
    /* similar rays side by side: equal trip counts inside a wavefront and neighbouring
     * table/profile addresses across its lanes */
    rc = ensure_sort_buffers(m, nr);
    if (rc) return rc;
    /* group by atmosphere slice when every slice is used by many rays */
    int const by_profile = m->atm_slices > 1 && nr >= 1024L * m->atm_slices;
    int const e = jurk_sort_rays(&m->view, by_profile, nr, d_geom, ldg, m->d_order, m->d_sort_tmp, m->sort_tmp_bytes, s);
    if (e) { jur_set_error("ray sort failed: %s", hipGetErrorString((hipError_t)e)); return JUR_EHIP; }
    order = m->d_order;
  }
  long const Rt = m->use_trace_rays;
  m->n_launch_ega = 0;
#define TIMED(kind, launch, what)                                                              \
  do {                                                                                         \
    int const ti_ = (m->timing && m->ntimed < JUR_MAX_TIMED) ? m->ntimed++ : -1;               \
    if (ti_ >= 0) { m->evkind[ti_] = (kind); HIPCHK(hipEventRecord(m->evpool[2 * ti_], s)); } \
    int const e_ = (launch);                                                                   \
    if (e_) { jur_set_error(what " kernel launch failed: %s", hipGetErrorString((hipError_t)e_)); return JUR_EHIP; } \
    if (ti_ >= 0) HIPCHK(hipEventRecord(m->evpool[2 * ti_ + 1], s));                          \
  } while (0)
  for (long t0 = 0; t0 < nr; t0 += Rt) {
    jur_chunk_t c;
    long const nt = (nr - t0 < Rt) ? nr - t0 : Rt;
    c.stride = (int)Rt;
    c.stride_eps = (int)R;
    for (int k = 0; k < 7; k++) c.geom[k] = d_geom + (size_t)k * ldg;
    for (int k = 0; k < 3; k++) c.tp[k] = d_tp + (size_t)k * ldtp;
    c.rad = d_rad;
    c.tau = d_tau;
    c.np_out = d_np;
    c.eps = m->d_eps;
    c.status = d_status ? d_status : m->d_status;
    /* trace the whole super-chunk */
    c.n = (int)nt;
    c.first = t0;
    c.order = order ? order + t0 : NULL;
    c.np = m->d_np;
    c.tsurf = m->d_tsurf;
    c.los = m->d_los;
    TIMED(0, jurk_launch_trace(&m->view, &c, s), "trace");
    if (m->host_call) {
      /* host entry: the input radiances, which only the epilogue reads (NaN mask), are uploaded beside the first
       * ray-tracing launch; tangent points and point counts are final after the last one and are copied out
       * beside the integration */
      if (t0 == 0) HIPCHK(hipStreamWaitEvent(s, m->ev_mask, 0));
      if (t0 + Rt >= nr) HIPCHK(hipEventRecord(m->ev_trace, s));
    }
    c.eps_off = NULL;
    int compact = m->use_compact && nt > R;
    if (compact) {   /* not while the stream is being captured: the layout needs the path lengths on the host */
      hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); st = hipStreamCaptureStatusNone; }
      if (st != hipStreamCaptureStatusNone) compact = 0;
    }
    if (compact) {
      /* Tiles laid out by the longest path of each (rays are sorted by tangent altitude: a tile's paths are nearly
       * equal): the longest path per tile comes back to the host -- the one wait of such a call --, consecutive tiles are
       * packed into launches of about equal size that fit the transmittance workspace, and every tile's first point
       * slot goes back up. */
      long const ntile = (nt + 63) / 64, cap = R / 64 * JUR_NLOS;      /* tile-points the workspace holds */
      int *const h_np = m->h_tile, *const h_off = m->h_tile + (m->ws_trace_rays / 64 + 1);
      int const ek = jurk_tile_max((int)nt, m->d_np, m->d_tile_np, s);
      if (ek) { jur_set_error("tile kernel launch failed: %s", hipGetErrorString((hipError_t)ek)); return JUR_EHIP; }
      HIPCHK(hipMemcpyAsync(h_np, m->d_tile_np, sizeof(int) * ntile, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      long total = 0;
      for (long t = 0; t < ntile; t++) total += h_np[t];
      long const nchunk = (total + cap - 1) / cap > 0 ? (total + cap - 1) / cap : 1;
      long const target = (total + nchunk - 1) / nchunk;               /* equal shares instead of full launches and a tail */
      /* first point slot of every tile, restarting per launch; a launch of many tiles ends on a multiple of 32 tiles
       * (8 ray blocks of 256 rays: one per XCD, see xcd_block_item in jur_kernels.hip) */
      for (long t = 0; t < ntile;) {
        long acc = 0, e = t;
        while (e < ntile && (e == t || (acc + h_np[e] <= cap && acc < target))) acc += h_np[e++];
        if (e < ntile && e - t >= 64) e = t + (e - t) / 32 * 32;
        acc = 0;
        for (long q = t; q < e; q++) { h_off[q] = (int)acc; acc += h_np[q]; h_np[q] = (q == t); }   /* (lengths not needed again: launch starts) */
        t = e;
      }
      HIPCHK(hipMemcpyAsync(m->d_eps_off, h_off, sizeof(int) * ntile, hipMemcpyHostToDevice, s));
      for (long t0c = 0; t0c < ntile;) {
        long t1c = t0c + 1;
        while (t1c < ntile && !h_np[t1c]) t1c++;
        long const s0 = t0c * 64, s1 = (t1c * 64 < nt) ? t1c * 64 : nt;
        c.n = (int)(s1 - s0);
        c.first = t0 + s0;
        c.order = order ? order + t0 + s0 : NULL;
        c.np = m->d_np + s0;
        c.tsurf = m->d_tsurf + s0;
        c.los = m->d_los + (size_t)s0 * JUR_NLOS * m->nfield;
        c.eps_off = m->d_eps_off + t0c;
        TIMED(1, jurk_launch_ega(&m->view, &c, s), "ega");
        TIMED(2, jurk_launch_combine(&m->view, &c, s), "combine");
        m->n_launch_ega++;
        t0c = t1c;
      }
      continue;
    }
    /* integrate it in chunks of R rays; slot s0 of the super-chunk is slot 0 of the chunk */
    for (long s0 = 0; s0 < nt; s0 += R) {
      c.n = (int)((nt - s0 < R) ? nt - s0 : R);
      c.first = t0 + s0;
      c.order = order ? order + t0 + s0 : NULL;
      c.np = m->d_np + s0;
      c.tsurf = m->d_tsurf + s0;
      c.los = m->d_los + (size_t)s0 * JUR_NLOS * m->nfield;   /* tiles of 64 slots, [tile][point][field][64]: s0 is a multiple of 64 */
      TIMED(1, jurk_launch_ega(&m->view, &c, s), "ega");
      TIMED(2, jurk_launch_combine(&m->view, &c, s), "combine");
      m->n_launch_ega++;
    }
  }
#undef TIMED
  return record_done(m, s);
}

/* ---- host entry ----------------------------------------------------------------- */
/* Pinned allocations for callers that want their arrays to travel at PCIe speed without staging. */
void *jur_host_alloc(size_t bytes) {
  void *p = NULL;
  if (hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    jur_set_error("hipHostMalloc of %zu bytes failed", bytes);
    return NULL;
  }
  return p;
}
void jur_host_free(void *p) { if (p) (void)hipHostFree(p); }

static int is_pinned(void const *p) {
  hipPointerAttribute_t a;
  if (!p) return 0;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return a.type == hipMemoryTypeHost;
}

/* memcpy spread over a few host threads: a pageable array of a million rays moves at one core's ~10 GB/s otherwise */
typedef struct { char *dst; char const *src; size_t n; } cpjob_t;
static void *cp_worker(void *a) { cpjob_t const *j = (cpjob_t const *)a; memcpy(j->dst, j->src, j->n); return NULL; }
static void par_memcpy(void *dst, void const *src, size_t n) {
  enum { MAXT = 8 };
  int nt = 1;
  if (n >= ((size_t)4 << 20)) {
    long const ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    nt = (int)(n >> 21);
    if (nt > MAXT) nt = MAXT;
    if (ncpu > 0 && nt > ncpu) nt = (int)ncpu;
  }
  if (nt <= 1) { memcpy(dst, src, n); return; }
  pthread_t th[MAXT];
  cpjob_t job[MAXT];
  size_t const per = ((n / nt) + 4095) & ~(size_t)4095;
  int started = 0;
  for (int i = 0; i < nt; i++) {
    size_t const o = per * i;
    if (o >= n) break;
    job[i].dst = (char *)dst + o; job[i].src = (char const *)src + o; job[i].n = (n - o < per) ? n - o : per;
    if (i == nt - 1 || o + per >= n || pthread_create(&th[started], NULL, cp_worker, &job[i]) != 0) {
      job[i].n = n - o;                          /* last share (or no thread to be had): the rest, here */
      memcpy(job[i].dst, job[i].src, job[i].n);
      break;
    }
    started++;
  }
  for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
}

/* device / pinned-host images of one call's arrays:
 *   doubles: geom[7][nr] | rad[nr][nd] | tau[nr][nd] | tp[3][nr]   (inputs are a prefix, outputs a suffix)
 *   ints:    np[nr] */
static int ensure_io(jur_model_t *m, long nr, int want_host) {
  int const nd = m->view.nd;
  size_t const nval = (size_t)nr * (10 + 2 * (size_t)nd);
  if (nr > m->io_cap) {
    if (m->d_io) (void)hipFree(m->d_io);
    if (m->d_io_np) (void)hipFree(m->d_io_np);
    m->d_io = NULL; m->d_io_np = NULL; m->io_cap = 0;
    HIPCHK(hipMalloc((void **)&m->d_io, sizeof(double) * nval));
    HIPCHK(hipMalloc((void **)&m->d_io_np, sizeof(int) * (size_t)nr));
    m->io_cap = nr;
  }
  if (want_host && nr > m->h_io_cap) {
    if (m->h_io) (void)hipHostFree(m->h_io);
    m->h_io = NULL; m->h_io_cap = 0;
    HIPCHK(hipHostMalloc((void **)&m->h_io, sizeof(double) * nval + sizeof(int) * (size_t)nr, hipHostMallocDefault));
    m->h_io_cap = nr;
  }
  return JUR_OK;
}

int jur_model_io(jur_model_t *m, long nr, double **d_io, int **d_io_np) {
  HIPCHK(hipSetDevice(m->device));
  int const rc = ensure_io(m, nr, 0);
  if (rc) return rc;
  *d_io = m->d_io; *d_io_np = m->d_io_np;
  return JUR_OK;
}

#define JUR_SMALL_CALL 65536   /* rays: below this one staged transfer each way beats overlapping several */

int jur_formod_host(jur_model_t *m, long nr, double const *const geom[7], double *rad, double *tau, double *const tp[3],
                    int *np_out) {
  if (!m || nr < 0) { jur_set_error("formod_host: bad arguments"); return JUR_EINVAL; }
  if (nr == 0) return JUR_OK;
  HIPCHK(hipSetDevice(m->device));
  int const nd = m->view.nd;
  size_t const N = (size_t)nr, nrd = N * nd;
  hipStream_t const s = m->stream, s2 = m->stream2;
  int status = 0, rc;

  if (nr <= JUR_SMALL_CALL) {
    /* a package: everything through the pinned image, one transfer in, one out */
    if ((rc = ensure_io(m, nr, 1))) return rc;
    double *const d_geom = m->d_io, *const d_rad = d_geom + 7 * N, *const d_tau = d_rad + nrd, *const d_tp = d_tau + nrd;
    double *const h_geom = m->h_io, *const h_rad = h_geom + 7 * N, *const h_tau = h_rad + nrd, *const h_tp = h_tau + nrd;
    int *const h_np = (int *)(h_tp + 3 * N);
    for (int k = 0; k < 7; k++) memcpy(h_geom + k * N, geom[k], sizeof(double) * N);
    memcpy(h_rad, rad, sizeof(double) * nrd);
    static int no_zero_copy = -1;                /* A/B switch, read once */
    if (no_zero_copy < 0) no_zero_copy = getenv("JUR_NO_ZERO_COPY") ? 1 : 0;
    if (pencil_rays_per_group(m, nr) > 0 && !no_zero_copy) {
      /* The fused kernel reads the geometry from the pinned image and writes its results there itself (a few
       * hundred KB over PCIe at the two ends of the kernel): ONE launch and one wait per call.  With copy
       * commands around the kernel, concurrent callers (the lanes of the drop-in entry) were serialised by the
       * runtime's copy path: 0.63 M rays/s with 16 threads against 1.9 M this way (tools/run_lanes_bench.sh). */
      *m->h_status = 0;
      if ((rc = jur_formod_device(m, nr, h_geom, h_rad, h_tau, h_tp, np_out ? h_np : NULL, m->h_status, s))) return rc;
    } else {
      HIPCHK(hipMemcpyAsync(d_geom, h_geom, sizeof(double) * (7 * N + nrd), hipMemcpyHostToDevice, s));
      HIPCHK(hipMemsetAsync(m->d_status, 0, sizeof(int), s));
      if ((rc = jur_formod_device(m, nr, d_geom, d_rad, d_tau, d_tp, m->d_io_np, m->d_status, s))) return rc;
      HIPCHK(hipMemcpyAsync(h_rad, d_rad, sizeof(double) * (2 * nrd + 3 * N), hipMemcpyDeviceToHost, s));
      if (np_out) HIPCHK(hipMemcpyAsync(h_np, m->d_io_np, sizeof(int) * N, hipMemcpyDeviceToHost, s));
      HIPCHK(hipMemcpyAsync(m->h_status, m->d_status, sizeof(int), hipMemcpyDeviceToHost, s));
    }
    {  /* a package takes about a millisecond: poll instead of a wait that may sleep (in a process whose device is
        * set to blocking synchronisation -- PyTorch's default -- hipStreamSynchronize adds about 1 ms per call) */
      hipError_t q;
      int spins = 0;
      while ((q = hipStreamQuery(s)) == hipErrorNotReady)
        if (++spins > 2000) sched_yield();
      if (q != hipSuccess) { jur_set_error("HIP error %d (%s) while waiting for the package", (int)q, hipGetErrorString(q)); return JUR_EHIP; }
    }
    status = *m->h_status;
    memcpy(rad, h_rad, sizeof(double) * nrd);
    memcpy(tau, h_tau, sizeof(double) * nrd);
    for (int k = 0; k < 3; k++) memcpy(tp[k], h_tp + k * N, sizeof(double) * N);
    if (np_out) memcpy(np_out, h_np, sizeof(int) * N);
  } else {
    /* a batch: copies run beside the kernels on a second stream.  Arrays the caller keeps in pinned memory
     * (jur_host_alloc, hipHostMalloc, hipHostRegister) are transferred in place; pageable ones go through the
     * model's pinned image with a threaded memcpy.
     *   stream:  H2D geometry -> sort, trace | wait(mask) -> ega -> combine -> D2H rad, tau
     *   stream2: H2D input rad (NaN mask, read by the epilogue only) ........ wait(trace) -> D2H tp, np    */
    int pin_g[7], pin_tp[3], stage = 0;
    for (int k = 0; k < 7; k++) stage |= !(pin_g[k] = is_pinned(geom[k]));
    for (int k = 0; k < 3; k++) stage |= !(pin_tp[k] = is_pinned(tp[k]));
    int const pin_rad = is_pinned(rad), pin_tau = is_pinned(tau), pin_np = np_out ? is_pinned(np_out) : 1;
    stage |= !pin_rad | !pin_tau | !pin_np;
    if ((rc = ensure_io(m, nr, stage))) return rc;
    double *const d_geom = m->d_io, *const d_rad = d_geom + 7 * N, *const d_tau = d_rad + nrd, *const d_tp = d_tau + nrd;
    double *const h_geom = m->h_io, *const h_rad = h_geom ? h_geom + 7 * N : NULL, *const h_tau = h_geom ? h_rad + nrd : NULL,
           *const h_tp = h_geom ? h_tau + nrd : NULL;
    int *const h_np = h_geom ? (int *)(h_tp + 3 * N) : NULL;
    for (int k = 0; k < 7; k++) {
      double const *src = geom[k];
      if (!pin_g[k]) { par_memcpy(h_geom + k * N, geom[k], sizeof(double) * N); src = h_geom + k * N; }
      HIPCHK(hipMemcpyAsync(d_geom + k * N, src, sizeof(double) * N, hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipMemsetAsync(m->d_status, 0, sizeof(int), s));
    {
      double const *src = rad;
      if (!pin_rad) { par_memcpy(h_rad, rad, sizeof(double) * nrd); src = h_rad; }
      HIPCHK(hipMemcpyAsync(d_rad, src, sizeof(double) * nrd, hipMemcpyHostToDevice, s2));
      HIPCHK(hipEventRecord(m->ev_mask, s2));
    }
    m->host_call = 1;
    rc = jur_formod_device(m, nr, d_geom, d_rad, d_tau, d_tp, m->d_io_np, m->d_status, s);
    m->host_call = 0;
    if (rc) { (void)hipStreamSynchronize(s2); return rc; }
    HIPCHK(hipStreamWaitEvent(s2, m->ev_trace, 0));
    for (int k = 0; k < 3; k++)
      HIPCHK(hipMemcpyAsync(pin_tp[k] ? tp[k] : h_tp + k * N, d_tp + k * N, sizeof(double) * N, hipMemcpyDeviceToHost, s2));
    if (np_out) HIPCHK(hipMemcpyAsync(pin_np ? np_out : h_np, m->d_io_np, sizeof(int) * N, hipMemcpyDeviceToHost, s2));
    HIPCHK(hipMemcpyAsync(pin_rad ? rad : h_rad, d_rad, sizeof(double) * nrd, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(pin_tau ? tau : h_tau, d_tau, sizeof(double) * nrd, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(m->h_status, m->d_status, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s2));             /* tangent points are here while the integration still runs */
    for (int k = 0; k < 3; k++)
      if (!pin_tp[k]) par_memcpy(tp[k], h_tp + k * N, sizeof(double) * N);
    if (np_out && !pin_np) par_memcpy(np_out, h_np, sizeof(int) * N);
    HIPCHK(hipStreamSynchronize(s));
    status = *m->h_status;
    if (!pin_rad) par_memcpy(rad, h_rad, sizeof(double) * nrd);
    if (!pin_tau) par_memcpy(tau, h_tau, sizeof(double) * nrd);
  }
  if (status & 1) { jur_set_error("Too many LOS points! (a ray needs %d or more)", JUR_NLOS); return JUR_ENLOS; }
  return JUR_OK;
}

/* ---- Curtis-Godson columns ------------------------------------------------------ */
/* Traces the rays and returns, per ray, gas and LOS point, the Curtis-Godson pressure, temperature and
 * cumulative column (curtis_godson, jr_common.h:455-473).  cgp/cgt/cgu are host arrays
 * [nr][ng][JUR_NLOS] (points at and beyond np[ray] are 0); tp/np_out as in jur_formod_host. */
int jur_curtis_godson_host(jur_model_t *m, long nr, double const *const geom[7], double *cgp, double *cgt, double *cgu,
                           double *const tp[3], int *np_out) {
  if (!m || nr < 1 || !cgp || !cgt || !cgu) { jur_set_error("curtis_godson: bad arguments"); return JUR_EINVAL; }
  if (m->view.atm_np < 2) { jur_set_error("curtis_godson: no atmosphere set"); return JUR_EINVAL; }
  HIPCHK(hipSetDevice(m->device));
  int const ng = m->view.ng;
  int rc = ensure_workspace(m, nr);
  if (rc) return rc;
  long const Rt = m->use_trace_rays;
  hipStream_t s = m->stream;
  size_t const per_ray = (size_t)(ng > 0 ? ng : 1) * JUR_NLOS;
  double *d_geom = NULL, *d_out = NULL, *d_tp = NULL;
  int *d_np = NULL;
  hipError_t e = hipMalloc((void **)&d_geom, sizeof(double) * 7 * (size_t)nr);
  if (e == hipSuccess) e = hipMalloc((void **)&d_tp, sizeof(double) * 3 * (size_t)nr);
  if (e == hipSuccess) e = hipMalloc((void **)&d_np, sizeof(int) * (size_t)nr);
  if (e == hipSuccess) e = hipMalloc((void **)&d_out, sizeof(double) * 3 * per_ray * (size_t)nr);
  if (e != hipSuccess) { rc = JUR_EHIP; jur_set_error("curtis_godson: hipMalloc failed"); goto done; }
  for (int k = 0; k < 7; k++)
    if (hipMemcpyAsync(d_geom + (size_t)k * nr, geom[k], sizeof(double) * nr, hipMemcpyHostToDevice, s) != hipSuccess) { rc = JUR_EHIP; goto done; }
  if (hipMemsetAsync(m->d_status, 0, sizeof(int), s) != hipSuccess) { rc = JUR_EHIP; goto done; }
  {
    double *d_cgp = d_out, *d_cgt = d_out + per_ray * (size_t)nr, *d_cgu = d_out + 2 * per_ray * (size_t)nr;
    for (long t0 = 0; t0 < nr; t0 += Rt) {
      jur_chunk_t c;
      memset(&c, 0, sizeof c);
      c.n = (int)((nr - t0 < Rt) ? nr - t0 : Rt);
      c.stride = (int)Rt;
      c.stride_eps = (int)m->use_rays;
      c.first = t0;
      c.order = NULL;
      for (int k = 0; k < 7; k++) c.geom[k] = d_geom + (size_t)k * nr;
      for (int k = 0; k < 3; k++) c.tp[k] = d_tp + (size_t)k * nr;
      c.np_out = d_np;
      c.np = m->d_np;
      c.tsurf = m->d_tsurf;
      c.los = m->d_los;
      c.eps = m->d_eps;
      c.status = m->d_status;
      int ek = jurk_launch_trace(&m->view, &c, s);
      if (!ek) ek = jurk_launch_cg(&m->view, &c, d_cgp, d_cgt, d_cgu, s);
      if (ek) { jur_set_error("curtis_godson: kernel launch failed: %s", hipGetErrorString((hipError_t)ek)); rc = JUR_EHIP; goto done; }
    }
    int status = 0;
    hipError_t ec = hipMemcpyAsync(cgp, d_cgp, sizeof(double) * per_ray * (size_t)nr, hipMemcpyDeviceToHost, s);
    if (ec == hipSuccess) ec = hipMemcpyAsync(cgt, d_cgt, sizeof(double) * per_ray * (size_t)nr, hipMemcpyDeviceToHost, s);
    if (ec == hipSuccess) ec = hipMemcpyAsync(cgu, d_cgu, sizeof(double) * per_ray * (size_t)nr, hipMemcpyDeviceToHost, s);
    for (int k = 0; k < 3 && tp && ec == hipSuccess; k++)
      ec = hipMemcpyAsync(tp[k], d_tp + (size_t)k * nr, sizeof(double) * nr, hipMemcpyDeviceToHost, s);
    if (np_out && ec == hipSuccess) ec = hipMemcpyAsync(np_out, d_np, sizeof(int) * nr, hipMemcpyDeviceToHost, s);
    if (ec == hipSuccess) ec = hipMemcpyAsync(&status, m->d_status, sizeof(int), hipMemcpyDeviceToHost, s);
    if (ec == hipSuccess) ec = hipStreamSynchronize(s);
    if (ec != hipSuccess) { jur_set_error("curtis_godson: copy back failed: %s", hipGetErrorString(ec)); rc = JUR_EHIP; goto done; }
    if (status & 1) { jur_set_error("Too many LOS points! (a ray needs %d or more)", JUR_NLOS); rc = JUR_ENLOS; }
  }
done:
  if (d_geom) (void)hipFree(d_geom);
  if (d_tp) (void)hipFree(d_tp);
  if (d_np) (void)hipFree(d_np);
  if (d_out) (void)hipFree(d_out);
  return rc;
}

/* ---- field-of-view convolution of results that stay on the device ------------------- */
int jur_fov_apply_device(jur_model_t *m, long nr, double const *d_time, double const *d_vpz, double *d_rad, double *d_tau,
                         int n, double const *dz, double const *w, void *stream) {
  if (!m || nr < 1 || n < 1 || n > JUR_NSHAPE || !d_time || !d_vpz || !d_rad || !d_tau || !dz || !w) {
    jur_set_error("jur_fov_apply_device: bad arguments");
    return JUR_EINVAL;
  }
  HIPCHK(hipSetDevice(m->device));
  hipStream_t s = (hipStream_t)stream;
  int const nd = m->view.nd;
  size_t const vals = (size_t)nr * nd;
  /* scratch kept by the model and only ever grown: rad0 | tau0 | dz | w | status word.  Every ray reads the
   * UNconvolved neighbours; the status word is this path's own (a forward-model call in flight on another stream
   * keeps m->d_status to itself), and nothing here allocates or frees per call (hipFree would wait for the whole
   * device). */
  long const need = (long)(2 * vals + 2 * (size_t)JUR_NSHAPE + 1);
  if (need > m->fov_cap) {
    if (m->d_fov) (void)hipFree(m->d_fov);
    m->d_fov = NULL; m->fov_cap = 0;
    HIPCHK(hipMalloc((void **)&m->d_fov, sizeof(double) * (size_t)need));
    m->fov_cap = need;
  }
  double *const tmp = m->d_fov;
  int *const d_st = (int *)(tmp + 2 * vals + 2 * (size_t)JUR_NSHAPE);
  int *const h_st = m->h_status + 1;             /* second word of the pinned status block */
  int rc = JUR_OK, status = 0;
  hipError_t e = hipMemcpyAsync(tmp, d_rad, sizeof(double) * vals, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(tmp + vals, d_tau, sizeof(double) * vals, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(tmp + 2 * vals, dz, sizeof(double) * n, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(tmp + 2 * vals + JUR_NSHAPE, w, sizeof(double) * n, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemsetAsync(d_st, 0, sizeof(int), s);
  if (e == hipSuccess)
    e = (hipError_t)jurk_launch_fov(nr, nd, d_time, d_vpz, tmp, tmp + vals, d_rad, d_tau, nd, n, tmp + 2 * vals,
                                    tmp + 2 * vals + JUR_NSHAPE, d_st, s);
  if (e == hipSuccess) e = hipMemcpyAsync(h_st, d_st, sizeof(int), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) { jur_set_error("jur_fov_apply_device: %s", hipGetErrorString(e)); rc = JUR_EHIP; }
  else status = *h_st;
  if (!rc && (status & 2)) { jur_set_error("Cannot apply FOV convolution!"); rc = JUR_EINVAL; }
  return rc;
}

/* ---- known-answer hooks ----------------------------------------------------------- */
/* nin input arrays and nout in/out arrays of n doubles each go to the device as one slab [nin + nout][n] */
static int kat_slab(jur_model_t *m, long n, int nin, double const *const in[], int nout, double *const out[], double **d) {
  *d = NULL;
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipMalloc((void **)d, sizeof(double) * (size_t)n * (size_t)(nin + nout)));
  for (int k = 0; k < nin; k++) HIPCHK(hipMemcpy(*d + (size_t)k * n, in[k], sizeof(double) * n, hipMemcpyHostToDevice));
  for (int k = 0; k < nout; k++) HIPCHK(hipMemcpy(*d + (size_t)(nin + k) * n, out[k], sizeof(double) * n, hipMemcpyHostToDevice));
  return JUR_OK;
}

static int kat_finish(jur_model_t *m, int ek, long n, int nin, int nout, double *const out[], double *d) {
  int rc = JUR_OK;
  if (ek) { jur_set_error("known-answer kernel not launched: %s", hipGetErrorString((hipError_t)ek)); rc = (ek == (int)hipErrorInvalidValue) ? JUR_EINVAL : JUR_EHIP; }
  if (!rc && hipStreamSynchronize(m->stream) != hipSuccess) { jur_set_error("known-answer kernel failed"); rc = JUR_EHIP; }
  for (int k = 0; k < nout && !rc; k++)
    if (hipMemcpy(out[k], d + (size_t)(nin + k) * n, sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) rc = JUR_EHIP;
  if (d) (void)hipFree(d);
  return rc;
}

int jur_kat_ega_eps(jur_model_t *m, int ig, int id, long n, double const *tau, double const *t, double const *u,
                    double const *p, int mode, int chain, double *out) {
  if (!m || n < 1 || ig < 0 || ig >= m->view.ng || id < 0 || id >= m->view.nd) { jur_set_error("kat_ega_eps: bad arguments"); return JUR_EINVAL; }
  double const *in[4] = {tau, t, u, p};
  double *io[1] = {out}, *d;
  int rc = kat_slab(m, n, 4, in, 1, io, &d);
  if (rc) { if (d) (void)hipFree(d); return rc; }
  int const ek = jurk_kat_ega(&m->view, ig, id, n, d, d + n, d + 2 * n, d + 3 * n, mode, chain, d + 4 * n, m->stream);
  return kat_finish(m, ek, n, 4, 1, io, d);
}

int jur_kat_continua(jur_model_t *m, int id, long n, double const *p, double const *t, double const *q, double const *u_co2,
                     double const *u_h2o, double *out) {
  if (!m || n < 1 || id < 0 || id >= m->view.nd) { jur_set_error("kat_continua: bad arguments"); return JUR_EINVAL; }
  double const *in[5] = {p, t, q, u_co2, u_h2o};
  double *io[4] = {out, out + n, out + 2 * n, out + 3 * n}, *d;
  int rc = kat_slab(m, n, 5, in, 4, io, &d);
  if (rc) { if (d) (void)hipFree(d); return rc; }
  int const ek = jurk_kat_continua(&m->view, id, n, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, d + 5 * n, m->stream);
  return kat_finish(m, ek, n, 5, 4, io, d);
}

int jur_kat_update(jur_model_t *m, int id, long n, int what, double const *a, double const *b, double const *c, double *rad,
                   double *tau, double *src) {
  if (!m || n < 1 || id < 0 || id >= m->view.nd || (what != 0 && what != 1)) { jur_set_error("kat_update: bad arguments"); return JUR_EINVAL; }
  double const *in[3] = {a, b, c};
  double *io[3] = {rad, tau, src}, *d;
  int rc = kat_slab(m, n, 3, in, 3, io, &d);
  if (rc) { if (d) (void)hipFree(d); return rc; }
  int const ek = jurk_kat_update(&m->view, id, n, what, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, d + 5 * n, m->stream);
  return kat_finish(m, ek, n, 3, 3, io, d);
}

/* ---- retrieval Jacobian -------------------------------------------------------- */
/* State vector of the atmosphere inside the retrieval windows (atm2x, jurassic.c:1491-1513):
 * quantity index iqa (0 p, 1 T, 2+g q, 2+ng+w k) and atmosphere point ipa per element. */
static size_t state_vector(ctl_t const *ctl, atm_t const *atm, double *x, int *iqa, int *ipa) {
  size_t n = 0;
  int const nquant = 2 + ctl->ng + ctl->nw;
  for (int iq = 0; iq < nquant; iq++) {
    double zmin, zmax;
    double const *value;
    if (iq == 0) { zmin = ctl->retp_zmin; zmax = ctl->retp_zmax; value = atm->p; }
    else if (iq == 1) { zmin = ctl->rett_zmin; zmax = ctl->rett_zmax; value = atm->t; }
    else if (iq < 2 + ctl->ng) { zmin = ctl->retq_zmin[iq - 2]; zmax = ctl->retq_zmax[iq - 2]; value = atm->q[iq - 2]; }
    else { int const w = iq - 2 - ctl->ng; zmin = ctl->retk_zmin[w]; zmax = ctl->retk_zmax[w]; value = atm->k[w]; }
    for (int ip = 0; ip < atm->np; ip++)
      if (atm->z[ip] >= zmin && atm->z[ip] <= zmax) {
        if (x) x[n] = value[ip];
        if (iqa) iqa[n] = iq;
        if (ipa) ipa[n] = ip;
        n++;
      }
  }
  return n;
}

size_t jur_state_size(jur_model_t const *m, atm_t const *atm) { return state_vector(m->ctl, atm, NULL, NULL, NULL); }

size_t jur_measurement_size(jur_model_t const *m, obs_t const *obs) {
  size_t n = 0;
  for (int ir = 0; ir < obs->nr; ir++)
    for (int id = 0; id < m->view.nd; id++) n += isfinite(obs->rad[ir][id]) ? 1 : 0;
  return n;
}

/* Forward-difference Jacobian (kernel(), jurassic.c:812-857) as ONE batched forward-model call:
 * the n perturbed atmospheres are stacked behind the unperturbed one as further profile slices
 * (time stamps shifted by j * span) and every ray is replicated once per slice. */
static int kernel_ld(jur_model_t *m, atm_t const *atm, obs_t *obs, double *k, size_t mrows, size_t ncols, size_t ldk);

int jur_kernel(jur_model_t *m, atm_t const *atm, obs_t *obs, double *k, size_t mrows, size_t ncols) {
  return kernel_ld(m, atm, obs, k, mrows, ncols, ncols);
}

static int kernel_ld(jur_model_t *m, atm_t const *atm, obs_t *obs, double *k, size_t mrows, size_t ncols, size_t ldk) {
  if (!m || !atm || !obs || !k) { jur_set_error("jur_kernel: null argument"); return JUR_EINVAL; }
  ctl_t const *ctl = m->ctl;
  int const np0 = atm->np, nr = obs->nr, nd = ctl->nd, ng = m->view.ng, nw = m->view.nw;
  if (np0 < 2 || np0 > JUR_NP || nr < 1 || nr > JUR_NR) { jur_set_error("jur_kernel: bad atm->np / obs->nr"); return JUR_EINVAL; }
  size_t const n = state_vector(ctl, atm, NULL, NULL, NULL);
  if (n != ncols) { jur_set_error("jur_kernel: state vector has %zu elements, matrix has %zu columns", n, ncols); return JUR_EINVAL; }
  if (jur_measurement_size(m, obs) != mrows) { jur_set_error("jur_kernel: measurement vector size does not match the matrix rows"); return JUR_EINVAL; }
  double *x0 = (double *)malloc(sizeof(double) * (n + 1)), *hstep = (double *)malloc(sizeof(double) * (n + 1));
  int *iqa = (int *)malloc(sizeof(int) * (n + 1)), *ipa = (int *)malloc(sizeof(int) * (n + 1));
  state_vector(ctl, atm, x0, iqa, ipa);

  size_t const ncopy = n + 1, nrow = 6 + (size_t)ng + nw;
  size_t const NT = ncopy * (size_t)np0, NRT = ncopy * (size_t)nr;
  double *h = (double *)malloc(sizeof(double) * nrow * NT);
  double *g = NULL;                                /* the model's pinned image of the call's arrays (ensure_io) */
  int rc = JUR_OK;
  if (!x0 || !hstep || !iqa || !ipa || !h) { rc = JUR_ENOMEM; goto done; }
  if (hipSetDevice(m->device) != hipSuccess) { jur_set_error("jur_kernel: cannot select the device"); rc = JUR_EHIP; goto done; }
  if ((rc = ensure_io(m, (long)NRT, 1))) goto done;
  g = m->h_io;
  {
    double tmin = atm->time[0], tmax = atm->time[0];
    for (int i = 0; i < np0; i++) { tmin = fmin(tmin, atm->time[i]); tmax = fmax(tmax, atm->time[i]); }
    for (int i = 0; i < nr; i++) { tmin = fmin(tmin, obs->time[i]); tmax = fmax(tmax, obs->time[i]); }
    double const span = (tmax - tmin) + 1.0;
    for (size_t j = 0; j < ncopy; j++) {
      size_t const at = j * (size_t)np0;
      pack_atm_rows(m, atm, h, NT, at);
      if (j > 0) {  /* perturbation sizes, jurassic.c:831-837 */
        size_t const e = j - 1;
        int const iq = iqa[e];
        double hh;
        if (iq == 0) hh = fmax(fabs(0.01 * x0[e]), 1e-7);
        else if (iq == 1) hh = 1;
        else if (iq < 2 + ng) hh = fmax(fabs(0.01 * x0[e]), 1e-15);
        else hh = 1e-4;
        hstep[e] = hh;
        size_t const row = (iq == 0) ? 4 : (iq == 1) ? 5 : (size_t)(4 + iq);   /* q rows start at 6, k rows follow */
        h[row * NT + at + ipa[e]] = x0[e] + hh;
      }
      hydrostatic_rows(m, h, NT, at, np0);
      for (int i = 0; i < np0; i++) h[at + i] = atm->time[i] + (double)j * span;
    }
    m->h_atm_n = 0;                            /* the device no longer holds the caller's atmosphere */
    rc = upload_atm_rows(m, h, (long)NT);
    if (rc) goto done;
    double *geom[7], *tp[3], *rad, *tau;
    for (int q = 0; q < 7; q++) geom[q] = g + (size_t)q * NRT;
    rad = g + 7 * NRT;                             /* the image's layout: geom[7][N] | rad[N][nd] | tau[N][nd] | tp[3][N] */
    tau = rad + (size_t)nd * NRT;
    for (int q = 0; q < 3; q++) tp[q] = tau + (size_t)nd * NRT + (size_t)q * NRT;
    double const *src[7] = {obs->time, obs->obsz, obs->obslon, obs->obslat, obs->vpz, obs->vplon, obs->vplat};
    for (size_t j = 0; j < ncopy; j++)
      for (int i = 0; i < nr; i++) {
        size_t const r = j * (size_t)nr + i;
        geom[0][r] = src[0][i] + (double)j * span;
        for (int q = 1; q < 7; q++) geom[q][r] = src[q][i];
        for (int id = 0; id < nd; id++) rad[r * nd + id] = obs->rad[i][id];   /* carries the NaN mask */
      }
    /* One batched call on device arrays; the difference quotients are formed there too (jur_kquot_kernel) and only
     * the unperturbed block and the quotients come back. */
    size_t const nq = (size_t)nr * nd, nrd = NRT * (size_t)nd;
    if ((long)(nq * n + n) > m->kq_cap) {
      if (m->d_kq) (void)hipFree(m->d_kq);
      if (m->h_kq) (void)hipHostFree(m->h_kq);
      m->d_kq = NULL; m->h_kq = NULL; m->kq_cap = 0;
      if (hipMalloc((void **)&m->d_kq, sizeof(double) * (nq * n + n)) != hipSuccess ||
          hipHostMalloc((void **)&m->h_kq, sizeof(double) * (nq * n + n), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError(); jur_set_error("jur_kernel: no memory for the quotients"); rc = JUR_ENOMEM; goto done;
      }
      m->kq_cap = (long)(nq * n + n);
    }
    hipStream_t const s = m->stream;
    double *const d_geom = m->d_io, *const d_rad = d_geom + 7 * NRT, *const d_tau = d_rad + nrd, *const d_tp = d_tau + nrd;
    double *const d_h = m->d_kq, *const d_kq = m->d_kq + n;
    int status = 0;
    memcpy(m->h_kq, hstep, sizeof(double) * n);
    hipError_t e = hipMemcpyAsync(d_geom, g, sizeof(double) * (7 * NRT + nrd), hipMemcpyHostToDevice, s);   /* geometry | input radiances */
    if (e == hipSuccess) e = hipMemcpyAsync(d_h, m->h_kq, sizeof(double) * n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(m->d_status, 0, sizeof(int), s);
    if (e != hipSuccess) { jur_set_error("jur_kernel: %s", hipGetErrorString(e)); rc = JUR_EHIP; goto done; }
    if ((rc = jur_formod_device(m, (long)NRT, d_geom, d_rad, d_tau, d_tp, NULL, m->d_status, s))) goto done;
    if (jurk_launch_kquot((long)nq, (long)n, d_rad, d_h, d_kq, s)) { jur_set_error("jur_kernel: quotient kernel launch failed"); rc = JUR_EHIP; goto done; }
    double *const kq = m->h_kq + n;
    e = hipMemcpyAsync(kq, d_kq, sizeof(double) * nq * n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(rad, d_rad, sizeof(double) * nq, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(tau, d_tau, sizeof(double) * nq, hipMemcpyDeviceToHost, s);
    for (int q = 0; q < 3 && e == hipSuccess; q++) e = hipMemcpyAsync(tp[q], d_tp + (size_t)q * NRT, sizeof(double) * nr, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&status, m->d_status, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { jur_set_error("jur_kernel: %s", hipGetErrorString(e)); rc = JUR_EHIP; goto done; }
    if (status) { jur_set_error("Too many LOS points!"); rc = JUR_ENLOS; goto done; }
    for (int i = 0; i < nr; i++) {   /* unperturbed result back to the caller, as kernel() leaves it */
      for (int id = 0; id < nd; id++) { obs->rad[i][id] = rad[(size_t)i * nd + id]; obs->tau[i][id] = tau[(size_t)i * nd + id]; }
      for (int id = nd; id < JUR_ND; id++) { obs->rad[i][id] = 0.0; obs->tau[i][id] = 1.0; }
      obs->tpz[i] = tp[0][i]; obs->tplon[i] = tp[1][i]; obs->tplat[i] = tp[2][i];
    }
    size_t row = 0;
    for (size_t q = 0; q < nq; q++) {              /* rows of the finite measurements (obs2y, jurassic.c:1527-1541) */
      if (!isfinite(rad[q])) continue;
      if (row < mrows) memcpy(k + row * ldk, kq + q * n, sizeof(double) * n);
      row++;
    }
    if (row != mrows) { jur_set_error("jur_kernel: %zu finite measurements after the forward model, the matrix has %zu rows", row, mrows); rc = JUR_EINVAL; }
  }
done:
  free(x0); free(hstep); free(iqa); free(ipa); free(h);
  if (rc == JUR_OK) rc = jur_model_set_atm(m, atm);   /* leave the model with the caller's atmosphere */
  return rc;
}

/* ---- drop-in entry points ------------------------------------------------------ */
#define DIE(...)                                                                 \
  do {                                                                           \
    printf("\nError (%s, %s, l%d): ", __FILE__, __func__, __LINE__);             \
    printf(__VA_ARGS__);                                                         \
    printf("\n\n");                                                              \
    fflush(stdout);                                                              \
    exit(EXIT_FAILURE);                                                          \
  } while (0)

/* The reference serves concurrent callers (OpenMP threads) through up to 4 "lanes", each with its own
 * stream and buffers (GPUdrivers.cu:262-342).  Same idea here: lane 0 is the process-lifetime model that
 * owns the tables (like upstream's static tbl), further lanes are created on demand as shallow copies that
 * share the table arrays and own their atmosphere, workspace and stream.  A small package (<= 1088 rays)
 * leaves most of the GPU idle, so concurrent packages overlap almost perfectly. */
#define JUR_MAX_LANES 16
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_cond = PTHREAD_COND_INITIALIZER;
static jur_model_t *g_lane[JUR_MAX_LANES];   /* g_lane[0]: owner of the tables */
static int g_busy[JUR_MAX_LANES];
static int g_nlane = 0, g_maxlane = 0;

static jur_model_t *clone_lane(jur_model_t const *m) {
  jur_model_t *c = (jur_model_t *)malloc(sizeof *c);
  if (!c) return NULL;
  *c = *m;
  c->shared_tables = 1;
  c->ctl = (ctl_t *)malloc(sizeof(ctl_t));
  if (!c->ctl) { free(c); return NULL; }
  memcpy(c->ctl, m->ctl, sizeof(ctl_t));
  c->d_atm = NULL; c->atm_cap = 0; c->atm_slices = 0;
  c->view.atm_np = 0;
  c->d_los = NULL; c->d_eps = NULL; c->d_np = NULL; c->d_tsurf = NULL; c->d_status = NULL;
  c->los_bytes = 0; c->ws_rays = 0; c->ws_trace_rays = 0;
  c->d_order = NULL; c->d_sort_tmp = NULL; c->order_cap = 0; c->sort_tmp_bytes = 0;
  c->d_io = NULL; c->d_io_np = NULL; c->io_cap = 0;
  c->h_io = NULL; c->h_io_cap = 0; c->h_pkg = NULL; c->h_atm = NULL; c->h_atm_n = 0; c->h_atm_cap = 0; c->h_status = NULL;
  c->stream = NULL; c->stream2 = NULL; c->ev_mask = NULL; c->ev_trace = NULL;
  c->host_call = 0; c->have_done = 0; c->ev_done = NULL; c->d_fov = NULL; c->fov_cap = 0; c->d_kq = NULL; c->h_kq = NULL; c->kq_cap = 0;
  c->timing = 0; c->evpool = NULL; c->evkind = NULL; c->ntimed = 0;
  if (hipSetDevice(c->device) != hipSuccess || create_streams(c) != JUR_OK ||
      hipMalloc((void **)&c->d_status, sizeof(int)) != hipSuccess || hipMemset(c->d_status, 0, sizeof(int)) != hipSuccess) {
    jur_model_destroy(c);
    return NULL;
  }
  return c;
}

/* run-time switches of the control block are honoured on every call (upstream re-uploads ctl per call,
 * GPUdrivers.cu:355); nu/emitter/tables are latched by the first call (jr_common.h:62-63) */
static void refresh_switches(jur_model_t *m, ctl_t const *ctl) {
  jur_view_t *v = &m->view;
  if (ctl->ng != v->ng || ctl->nd != v->nd) DIE("ng/nd changed after the tables were initialised");
  memcpy(m->ctl, ctl, sizeof(ctl_t));
  v->refrac = ctl->refrac; v->write_bbt = ctl->write_bbt; v->rayds = ctl->rayds; v->raydz = ctl->raydz;
  /* upstream looks the emitters up on the first call that has the switch on, whichever call that is
   * (CPUdrivers.c:126-128: `if (ctl->ctm_h2o && -999 == ig_h2o) ig_h2o = find_emitter(...)`) */
  if (ctl->ctm_h2o && v->ig_h2o == -999) { v->ig_h2o = find_emitter(ctl, "H2O"); m->h_atm_n = 0; }   /* hydrostatic uses it */
  if (ctl->ctm_co2 && v->ig_co2 == -999) v->ig_co2 = find_emitter(ctl, "CO2");
  v->fourbit = ((1 == ctl->ctm_co2) && (v->ig_co2 >= 0)) * 8 + ((1 == ctl->ctm_h2o) && (v->ig_h2o >= 0)) * 4
             + (1 == ctl->ctm_n2) * 2 + (1 == ctl->ctm_o2) * 1;
}

/* returns a lane index marked busy; blocks while all lanes are in use */
static int acquire_lane(ctl_t const *ctl) {
  pthread_mutex_lock(&g_lock);
  if (g_nlane == 0) {   /* first call of the process: load the tables (upstream: get_tbl under omp critical) */
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
      DIE("no GPU found: this library has no CPU path (USEGPU = %d)", ctl->useGPU);
    int device = ctl->MPIlocalrank;               /* GPUdrivers.cu:344 */
    if (device < 0 || device >= ndev) device = 0;
    printf("Initialize emissivity tables and source function (MI355X path, device %d)...\n", device);
    if (jur_model_create_from_files(&g_lane[0], ctl, device) != JUR_OK) DIE("%s", jur_last_error());
    g_nlane = 1;
    char const *e = getenv("JUR_LANES");
    g_maxlane = e ? atoi(e) : 4;
    if (g_maxlane < 1) g_maxlane = 1;
    if (g_maxlane > JUR_MAX_LANES) g_maxlane = JUR_MAX_LANES;
  }
  int lane = -1;
  for (;;) {
    for (int i = 0; i < g_nlane && lane < 0; i++)
      if (!g_busy[i]) lane = i;
    if (lane < 0 && g_nlane < g_maxlane) {
      g_lane[g_nlane] = clone_lane(g_lane[0]);
      if (!g_lane[g_nlane]) DIE("cannot create another lane: %s", jur_last_error());
      lane = g_nlane++;
    }
    if (lane >= 0) break;
    pthread_cond_wait(&g_cond, &g_lock);
  }
  g_busy[lane] = 1;
  pthread_mutex_unlock(&g_lock);
  refresh_switches(g_lane[lane], ctl);
  return lane;
}

static void release_lane(int lane) {
  pthread_mutex_lock(&g_lock);
  g_busy[lane] = 0;
  pthread_cond_signal(&g_cond);
  pthread_mutex_unlock(&g_lock);
}

static void formod_range(ctl_t const *ctl, atm_t *atm, obs_t *obs, int r0, int nr) {
  if (ctl->checkmode) { printf("# %s: no operation in checkmode\n", __func__); return; }
  if (!obs || !atm) DIE("null argument");
  if (r0 < 0 || nr < 0 || r0 + nr > JUR_NR) DIE("ray range outside the package (max %d rays)", JUR_NR);
  if (nr == 0) return;
  int const lane = acquire_lane(ctl);
  jur_model_t *m = g_lane[lane];
  if (jur_model_set_atm(m, atm) != JUR_OK) DIE("%s", jur_last_error());
  int const nd = ctl->nd;
  if (!m->h_pkg && hipHostMalloc((void **)&m->h_pkg, sizeof(double) * 2 * JUR_NR * (size_t)nd, hipHostMallocDefault) != hipSuccess)
    DIE("Out of memory!");
  double *const buf = m->h_pkg;                  /* the lane's package scratch, kept for the life of the process */
  double *rad = buf, *tau = buf + (size_t)nr * nd;
  for (int i = 0; i < nr; i++)
    for (int id = 0; id < nd; id++) rad[(size_t)i * nd + id] = obs->rad[r0 + i][id];
  double const *geom[7] = {obs->time + r0, obs->obsz + r0, obs->obslon + r0, obs->obslat + r0,
                           obs->vpz + r0, obs->vplon + r0, obs->vplat + r0};
  double *tp[3] = {obs->tpz + r0, obs->tplon + r0, obs->tplat + r0};
  int const rc = jur_formod_host(m, nr, geom, rad, tau, tp, NULL);
  if (rc == JUR_ENLOS) DIE("Too many LOS points!");
  if (rc != JUR_OK) DIE("%s", jur_last_error());
  for (int i = 0; i < nr; i++) {
    for (int id = 0; id < nd; id++) {
      obs->rad[r0 + i][id] = rad[(size_t)i * nd + id];
      obs->tau[r0 + i][id] = tau[(size_t)i * nd + id];
    }
    for (int id = nd; id < JUR_ND; id++) {          /* upstream clears all ND channels, CPUdrivers.c:58-60 */
      obs->rad[r0 + i][id] = 0.0;
      obs->tau[r0 + i][id] = 1.0;
    }
  }
  release_lane(lane);
}

/* Explicit finalize for the drop-in entry's process-global state (upstream never frees its own,
 * GPUdrivers.cu:263-273, 309): every lane with its stream, atmosphere, workspace and pinned images, and the tables
 * of lane 0.  Waits for lanes that are still in a call.  A later formod() initialises again from the files named
 * by its ctl.  Safe to call twice, before any formod(), and from an atexit handler. */
int jur_dropin_finalize(void) {
  pthread_mutex_lock(&g_lock);
  for (int i = 0; i < g_nlane; i++)
    while (g_busy[i]) pthread_cond_wait(&g_cond, &g_lock);
  int const n = g_nlane;
  for (int i = n - 1; i >= 0; i--) {              /* lane 0 owns the tables the others share: last */
    if (g_lane[i]) {
      (void)hipSetDevice(g_lane[i]->device);
      if (g_lane[i]->stream) (void)hipStreamSynchronize(g_lane[i]->stream);
      jur_model_destroy(g_lane[i]);
    }
    g_lane[i] = NULL;
  }
  g_nlane = 0;
  g_maxlane = 0;
  pthread_mutex_unlock(&g_lock);
  return n;
}

void formod_GPU(ctl_t const *ctl, atm_t *atm, obs_t *obs) { formod_range(ctl, atm, obs, 0, obs ? obs->nr : 0); }

void formod(ctl_t const *ctl, atm_t *atm, obs_t *obs) {
  if (ctl->checkmode)
    printf("# %s: %d max %d rays , %d of max %d gases, %d of max %d channels\n", __func__, obs->nr, JUR_NR, ctl->ng,
           JUR_NG, ctl->nd, JUR_ND);
  /* USEGPU = 0 ("never") has no meaning here: every path is the GPU path. */
  formod_GPU(ctl, atm, obs);
}

void formod_pencil(ctl_t const *ctl, atm_t *atm, obs_t *obs, int const ir) { formod_range(ctl, atm, obs, ir, 1); }

/* kernel() under its own name (jurassic.h:664, jurassic.c:812-857): the Jacobian into a gsl_matrix of
 * (finite measurements) x (state elements).  GSL is not a dependency of this library: jur_gsl_matrix_t restates the
 * layout of gsl_matrix (include/jurassic_hip.h).  obs returns the unperturbed forward model, as upstream. */
void kernel(ctl_t const *ctl, atm_t *atm, obs_t *obs, jur_gsl_matrix_t *k) {
  if (!ctl || !atm || !obs || !k || !k->data) DIE("null argument");
  if (ctl->checkmode) { printf("# %s: no operation in checkmode\n", __func__); return; }
  if (k->tda < k->size2) DIE("matrix with tda < size2");
  int const lane = acquire_lane(ctl);
  jur_model_t *m = g_lane[lane];
  /* (upstream sizes the matrix from obs2y / atm2x of the caller's obs and atm before the call, retrieval.c; so does the caller here) */
  int const rc = kernel_ld(m, atm, obs, k->data, k->size1, k->size2, k->tda);
  if (rc == JUR_ENLOS) DIE("Too many LOS points!");
  if (rc != JUR_OK) DIE("%s", jur_last_error());
  release_lane(lane);
}
