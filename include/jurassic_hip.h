/* jurassic_hip.h -- C-ABI of the MI355X forward-model library (libjurassic_hip.so).
 *
 * Two groups of entry points, all plain C, no C++/torch types:
 *
 * (1) DROP-IN symbols, exactly what the reference's host code binds:
 *       formod()         reference src/jurassic.h:515-518, body CPUdrivers.c:179-194
 *       formod_GPU()     reference src/GPUdrivers.cu:253-262 (declared CPUdrivers.c:153-155);
 *                        this is the object the reference links instead of GPUdrivers.o
 *       formod_pencil()  reference src/jurassic.h:526-530 (prototype only upstream)
 *     Same argument meaning, ownership and error behaviour as upstream: void
 *     return, a failure prints a message and terminates the process; tables are
 *     loaded from ${TBLBASE}_${nu}_${gas}.tab / .filt on first use and cached for
 *     the life of the process (jr_common.h:60-78).
 *
 * (2) ADDITIVE batched API (jur_*), for more than NR rays per call and for
 *     callers that already hold their arrays in device memory.  Returns 0 on
 *     success, a negative JUR_E* code otherwise; jur_last_error() has the text.
 */
#ifndef JURASSIC_HIP_H
#define JURASSIC_HIP_H

#include "jurassic_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- (1) drop-in ---------------------------------------------------------- */
void formod(ctl_t const *ctl, atm_t *atm, obs_t *obs);
void formod_GPU(ctl_t const *ctl, atm_t *atm, obs_t *obs);
void formod_pencil(ctl_t const *ctl, atm_t *atm, obs_t *obs, int const ir);
/* field-of-view convolution of the radiances / transmittances in obs (jurassic.h:521, jurassic.c:214-258);
 * no-op when ctl->fov is "-"; host code */
void formod_fov(ctl_t const *ctl, obs_t *obs);

/* regridding of an atmosphere onto the points of another (jurassic.h:581-585, jurassic.c:675-683): dest->p, t, q, k
 * at dest's z / lon / lat from src by ctl->ip = 1 (one profile), 2 (the nearest two profiles of a satellite
 * track) or 3 (distance-weighted mean of a point cloud, ctl->cx / ctl->cz); GPU, device ctl->MPIlocalrank */
void intpol_atm(ctl_t *ctl, atm_t *atm_dest, atm_t *atm_src);

/* kernel() (reference jurassic.h:664, jurassic.c:812-857): forward-difference Jacobian of the finite radiances with
 * respect to the state elements inside the ctl->ret*_zmin/zmax windows, as ONE batched forward-model call on the GPU
 * (the difference quotients are formed on the device too); obs returns the unperturbed forward model.  The reference's
 * last argument is a gsl_matrix *: this library does not depend on GSL, jur_gsl_matrix_t restates that struct's layout
 * (gsl/gsl_matrix_double.h, unchanged since GSL 1.0: size1, size2, tda, data, block, owner) -- a caller that has GSL
 * passes its gsl_matrix * (same object), one that has not fills the first four fields.  size1 must be the number of
 * finite radiances of obs, size2 the state size (what obs2y / atm2x return upstream; jur_measurement_size /
 * jur_state_size here). */
typedef struct { size_t size1, size2, tda; double *data; void *block; int owner; } jur_gsl_matrix_t;
void kernel(ctl_t const *ctl, atm_t *atm, obs_t *obs, jur_gsl_matrix_t *k);

/* ---- (2) additive --------------------------------------------------------- */
enum {
  JUR_OK = 0,
  JUR_EINVAL = -1,    /* bad argument / unsupported control setting          */
  JUR_EIO = -2,       /* table or filter file problem                        */
  JUR_ENOMEM = -3,
  JUR_EHIP = -4,      /* HIP runtime error                                   */
  JUR_ENLOS = -5,     /* a line of sight needs >= NLOS points (jr_common.h:693) */
  JUR_ENODEV = -6     /* no usable GPU                                       */
};

char const *jur_last_error(void);

/* sizeof(ctl_t), sizeof(atm_t), sizeof(obs_t), ND, NG the library was built with */
void jur_abi_sizes(size_t out[5]);
/* C1, C2, P0, RE (reference jurassic.h:109-126) and the GSL 2.5 values of N_A, k_B, R (used at jr_common.h:330,
 * 450, 744) the library was built with */
void jur_abi_constants(double out[7]);

/* Host-side emissivity tables under construction. */
typedef struct jur_tables jur_tables_t;
jur_tables_t *jur_tables_new(int ng, int nd);
void jur_tables_free(jur_tables_t *tb);
/* Feed the rows (p [hPa], T [K], u [molec/cm^2], eps) of one (gas, channel)
 * table in file order; row acceptance follows reference jurassic.c:346-395. */
int jur_tables_feed_rows(jur_tables_t *tb, int ig, int id, long nrows,
                         double const *p, double const *t, double const *u, double const *eps);
/* Parse every ${tblbase}_${nu:%.4f}_${emitter}.tab named by ctl; missing files
 * leave that pair without table (transparent), as upstream.  Returns the number
 * of files found or a negative error. */
int jur_tables_read_ascii(jur_tables_t *tb, ctl_t const *ctl);
/* Filter function of one channel -> source-function table (jurassic.c:612-667). */
int jur_tables_set_filter(jur_tables_t *tb, int id, int n, double const *nu, double const *f);
int jur_tables_read_filters(jur_tables_t *tb, ctl_t const *ctl);
/* Compact binary cache of parsed tables + source functions (own format: text header naming
 * emitters and channels, then the flattened arrays and a checksum; ~8 bytes per table entry).
 * jur_tables_load: JUR_OK, JUR_EIO (no file) or JUR_EINVAL (other emitters/channels, damaged).
 * jur_model_create_from_files uses it under `bin.jurassic-hip-tables-g<ng>-d<nd>` in the
 * working directory as ctl->read_binary / ctl->write_binary ask (reference jurassic.c:312-320). */
int  jur_tables_save(jur_tables_t const *tb, ctl_t const *ctl, char const *path);
int  jur_tables_load(jur_tables_t **out, ctl_t const *ctl, char const *path);
void jur_tables_cache_filename(char *out, size_t len, ctl_t const *ctl);
/* FNV-1a over the flattened tables and source functions */
unsigned long long jur_tables_checksum(jur_tables_t const *tb);
/* number of stored (u,eps) entries, for reporting */
long jur_tables_entries(jur_tables_t const *tb);

/* A model = control settings + tables + per-channel continuum constants,
 * resident on one GPU (`device` = HIP device ordinal). */
typedef struct jur_model jur_model_t;
int  jur_model_create(jur_model_t **out, ctl_t const *ctl, jur_tables_t const *tb, int device);
/* convenience: tables + filters from the files ctl names */
int  jur_model_create_from_files(jur_model_t **out, ctl_t const *ctl, int device);
void jur_model_destroy(jur_model_t *m);

/* Upload the atmosphere (only atm->np points and ctl->ng gases travel).
 * Applies the hydrostatic adjustment first if ctl->hydz >= 0 -- on a private
 * copy; the caller's atm is not modified (GPU-path behaviour upstream,
 * GPUdrivers.cu:243).  An atmosphere identical to the one on the device is not
 * uploaded again.
 * Ordering: the upload waits for the model's last jur_formod_device call
 * (whose kernels may still read the old atmosphere) through an event of the
 * model's own that the call left on the caller's stream -- no handle of the
 * caller's is kept, the stream may have been destroyed since.  Calls enqueued
 * on OTHER streams before that one, and launches of a captured graph, are the
 * caller's to synchronise. */
int  jur_model_set_atm(jur_model_t *m, atm_t const *atm);

/* Forward model for nr rays, host arrays.  geom[7] = {time, obsz, obslon,
 * obslat, vpz, vplon, vplat}, each [nr].  rad/tau are [nr][nd] (nd = ctl->nd,
 * channel fastest); rad is read first: channels that hold a non-finite value
 * on input come back NaN (jr_common.h:193-210).  tp[3] = {tpz, tplon, tplat},
 * each [nr].  np_out (optional) receives the number of LOS points per ray. */
int  jur_formod_host(jur_model_t *m, long nr, double const *const geom[7],
                     double *rad, double *tau, double *const tp[3], int *np_out);

/* jur_formod_host moves arrays that lie in pinned host memory (these allocators, hipHostMalloc,
 * hipHostRegister) in place at PCIe speed and overlaps the transfers with the kernels; pageable arrays
 * are staged through a pinned image inside the model (threaded memcpy).  NULL on failure. */
void *jur_host_alloc(size_t bytes);
void  jur_host_free(void *p);

/* Same, all pointers in device memory of the model's GPU; work is enqueued on
 * `stream` (a hipStream_t, may be NULL) and the call returns without waiting.
 * d_geom is [7][nr], d_tp is [3][nr], d_np (optional) [nr].
 * d_status (optional, int[1]) is set non-zero on device if a ray overflowed NLOS. */
int  jur_formod_device(jur_model_t *m, long nr, double const *d_geom,
                       double *d_rad, double *d_tau, double *d_tp, int *d_np,
                       int *d_status, void *stream);

/* ---- several GPUs in one process ------------------------------------------------------------------------------
 * The reference's device loop lives inside formod_GPU (GPUdrivers.cu:344-358: one OpenMP thread per device, every
 * device given the same package); SURVEY.md section 8e asks for "one host process, one stream set per device" with the
 * rays partitioned.  models[0 .. nmodel) are models of the same control block and tables, normally one per device
 * (several on one device are allowed: that is how a one-GPU box rehearses the path); each needs the atmosphere
 * (jur_models_set_atm).  Rays are dealt in CONTIGUOUS ranges whose boundaries equalise the estimated number of
 * line-of-sight points (jur_multi_balance: straight-line path through the atmosphere's altitude range in steps of
 * min(RAYDS, RAYDZ / |cos a|)), not the number of rays -- a tangent-height scan in its natural order has 122 .. 393
 * points per ray.  Results are the single-model results bit for bit.
 *
 * jur_formod_host_multi: host arrays as jur_formod_host; one host thread per model, every share written straight
 *   into the caller's arrays (pinned arrays move in place), no gather.
 * jur_formod_device_multi: all arrays in device memory of models[0]'s GPU (layout as jur_formod_device).  Share 0 is
 *   computed in place on `stream`; the others travel to their model's device and back with hipMemcpyPeerAsync on that
 *   model's own stream (xGMI between the GPUs of a node), and `stream` waits for them: no host synchronisation.
 *   bounds[0 .. nmodel] (optional, NULL: equal ray counts): share k = rays [bounds[k], bounds[k+1]), e.g. from
 *   jur_multi_balance when the geometry is known on the host.  d_status (optional) is int[nmodel], one word per share. */
int  jur_multi_balance(jur_model_t const *m, long nr, double const *const geom[7], int nparts, long *bounds);
/* The same without a model or a GPU (host arithmetic only): the estimated number of LOS points of every ray from the
 * tracer's step sizes (ctl->rayds, ctl->raydz) and the altitude range [zmin, zmax] of the atmosphere, and the balanced
 * boundaries from it -- what a launcher that deals a SORTED observation set to one process per GPU needs
 * (jurassic_hip/shard.py: balanced_ranges). */
int  jur_estimate_los_points(double rayds, double raydz, double zmin, double zmax, long nr, double const *const geom[7], double *points);
int  jur_balance_rays(double rayds, double raydz, double zmin, double zmax, long nr, double const *const geom[7], int nparts, long *bounds);
int  jur_models_set_atm(jur_model_t *const models[], int nmodel, atm_t const *atm);
int  jur_formod_host_multi(jur_model_t *const models[], int nmodel, long nr, double const *const geom[7],
                           double *rad, double *tau, double *const tp[3], int *np_out);
int  jur_formod_device_multi(jur_model_t *const models[], int nmodel, long nr, long const *bounds, double const *d_geom,
                             double *d_rad, double *d_tau, double *d_tp, int *d_np, int *d_status, void *stream);

/* Retrieval support (reference kernel(), jurassic.c:812-857; state/measurement
 * vectors as atm2x/obs2y, :1491-1541): forward-difference Jacobian dy/dx of the
 * finite radiances with respect to the atmosphere values inside the
 * ctl->ret{p,t,q,k}_zmin/zmax windows, evaluated as one batched forward-model
 * call (n+1 stacked atmospheres) whose difference quotients are formed on the device.  k is row-major
 * [jur_measurement_size][jur_state_size];
 * obs returns the unperturbed result.  The reference's signature takes a gsl_matrix;
 * bind with k = matrix->data when matrix->tda == matrix->size2. */
size_t jur_state_size(jur_model_t const *m, atm_t const *atm);
size_t jur_measurement_size(jur_model_t const *m, obs_t const *obs);
int    jur_kernel(jur_model_t *m, atm_t const *atm, obs_t *obs, double *k, size_t mrows, size_t ncols);

/* Curtis-Godson means along each line of sight (reference curtis_godson(), jr_common.h:455-473, which
 * upstream compiles only with -DCURTIS_GODSON for FORMOD=1): per ray, emitter and LOS point the
 * column-weighted pressure cgp, temperature cgt and the cumulative column cgu.  Host arrays
 * [nr][ng][JUR_NLOS]; entries from np[ray] on are 0.  tp (optional) and np_out (optional) as above. */
int  jur_curtis_godson_host(jur_model_t *m, long nr, double const *const geom[7],
                            double *cgp, double *cgt, double *cgu, double *const tp[3], int *np_out);

/* Field-of-view convolution on flat arrays (the arithmetic of formod_fov): time[nr], vpz[nr], rad/tau rows of
 * nd values with row stride ld, n weights w at altitude offsets dz (jur_fov_read_shape reads the
 * two-column file ctl->fov names, at most JUR_NSHAPE rows).  In place; host code. */
int  jur_fov_read_shape(char const *filename, int *n, double *dz, double *w);
int  jur_fov_apply(int nd, long nr, double const *time, double const *vpz, double *rad, double *tau, long ld,
                   int n, double const *dz, double const *w);

/* The same convolution for results that stay in HBM (after jur_formod_device): d_time / d_vpz [nr] and
 * d_rad / d_tau [nr][nd] in device memory of the model's GPU, convolved in place by a HIP kernel (one lane per
 * ray and channel, the arithmetic of formod_fov in its order: bit-identical to jur_fov_apply); dz / w are host
 * arrays.  Returns after the work on `stream` has finished; JUR_EINVAL if a ray is alone in its scan. */
int  jur_fov_apply_device(jur_model_t *m, long nr, double const *d_time, double const *d_vpz, double *d_rad,
                          double *d_tau, int n, double const *dz, double const *w, void *stream);

/* intpol_atm with an error code instead of exit(): JUR_EINVAL for what upstream aborts on (profiles of one point,
 * profiles more than 10 degrees apart, unknown IP).  One lane per destination point on `device`. */
int  jur_intpol_atm(ctl_t const *ctl, atm_t *dest, atm_t const *src, int device);

/* Allocate the workspace for calls of up to nr rays now.  jur_formod_device allocates lazily on
 * first use; after jur_model_reserve (or one call of the same size) it only enqueues kernels on
 * the stream -- no allocation, no host synchronisation -- and can be captured into a HIP graph. */
int  jur_model_reserve(jur_model_t *m, long nr);

/* Bytes of device workspace the model holds for `nr` rays per call, and the
 * chunk size (rays per kernel launch) it uses. */
long jur_model_workspace_bytes(jur_model_t const *m);
int  jur_model_chunk_rays(jur_model_t const *m);
/* Bytes of device memory the model's tables hold (entries, bracket slopes, descriptors). */
long jur_model_table_bytes(jur_model_t const *m);
/* PCI bus id (text, e.g. "0000:0c:00.0"; len >= 16) and the free / total device memory of `device` right now: what a
 * multi-GPU caller records and plans its workspace budget with. */
int  jur_device_info(int device, char *pci_bus_id, int len, size_t *free_bytes, size_t *total_bytes);
/* Tuning knobs: rays per chunk; whether rays are processed in order of their
 * geometric tangent altitude (default on; results do not depend on it). */
int  jur_model_set_chunk_rays(jur_model_t *m, int rays);
int  jur_model_set_sort_rays(jur_model_t *m, int on);
/* Rays per ray-tracing launch as a multiple of the rays per integration launch (default 1). */
int  jur_model_set_trace_multiple(jur_model_t *m, int mult);
/* Calls whose rays do not fit the workspace at JUR_NLOS points per ray (several integration launches) lay the
 * transmittance tiles out by the path lengths that occur (default on): all rays are traced first, the longest path of
 * every tile of 64 sorted rays comes back to the host -- ONE wait inside such a call; calls that fit in one launch, and
 * calls being captured into a graph, never wait -- and consecutive tiles are packed into launches of equal size.
 * Nadir rays use 182 of the 400 points: 2.2 x the rays per launch.  Same results bit for bit.
 * jur_model_last_launches: integration launches (jur_ega_kernel / jur_combine_kernel pairs) of the last batched call. */
int  jur_model_set_compact_workspace(jur_model_t *m, int on);
long jur_model_last_launches(jur_model_t const *m);
/* Upper bound of the per-call device workspace (LOS state + per-segment gas
 * transmittances); the rays-per-chunk shrink to fit.  Default 128 GiB of the 288 GB. */
int  jur_model_set_workspace_budget(jur_model_t *m, long bytes);

/* Calls of up to max_rays rays (default 10000; 0: never) run as ONE fused kernel -- ray tracing, emissivity growth
 * and radiance update of a ray as producer/consumer wavefronts of one workgroup, the line of sight handed on through
 * LDS -- instead of the three batched kernels: the sizes the reference's callers use (packages of <= NR rays).
 * rays_per_group: rays per workgroup, 0 = chosen from the call size.  Configurations with more (channel, gas)
 * chains per ray than the LDS rings hold use the batched kernels whatever the size.  Same results bit for bit. */
int  jur_model_set_pencil(jur_model_t *m, long max_rays, int rays_per_group);
/* Process-wide: lanes per ray of the batched ray tracer (1 or 4; 0 = chosen per launch: a quad of lanes per ray for
 * launches of up to 65 536 rays with at least three emitters, one lane otherwise).  With four lanes the refraction
 * probes of a step (jr_common.h:665-681) and the emitters' columns are taken side by side; same results bit for bit
 * (tests/test_multi_gpu.py). */
void jur_tune_trace(int lanes_per_ray);
/* Process-wide tuning of the radiance-update kernel of the batched path: up to `channels_per_group` (0 .. 6, default
 * 4; 0 = one channel per workgroup always) channels of a ray block share a workgroup, with a barrier every
 * `sync_segments` segments (default 8; <= 0 none), for launches of at least `min_lanes` rays x channels (default
 * 1 000 000).  Without a call (or after one with channels_per_group < 0) the library groups by four, and only when
 * the channel count is a multiple of four: other group shapes were measured slower than one channel per workgroup.
 * Results do not depend on it (tests/test_parity_gpu.py compares the arrangements bit for bit). */
void jur_tune_combine(int channels_per_group, int sync_segments, long min_lanes);

/* Arithmetic of the emissivity-growth look-up, per model (process default: JUR_ARITH_FAST, or JUR_ARITH_EXACT when
 * the environment has JUR_EGA_NO_RCP set).
 *   JUR_ARITH_FAST   strictly increasing tables (every table that passes the reference's row rule) are interpolated
 *                    through bracket slopes formed once per model, blended through reciprocal bracket widths, and the
 *                    path transmittance is carried as 1 - eps: within ~1e-13 of the reference's divisions on
 *                    transmittances, ~6e-12 relative on radiances (the contract is 1e-6), a seventh fewer instructions;
 *   JUR_ARITH_EXACT  the reference's divisions operand for operand (jr_common.h:156-185, 237-268) -- what tables that
 *                    are not strictly increasing get in either mode.  For difference quotients with steps so small
 *                    that 1e-12 of a radiance matters (jur_kernel on mixing ratios that are zero: see there).
 * The fused kernel and the batched kernels follow the same switch, so they stay bit-identical to each other. */
enum { JUR_ARITH_FAST = 0, JUR_ARITH_EXACT = 1 };
int  jur_model_set_arithmetic(jur_model_t *m, int mode);
int  jur_model_arithmetic(jur_model_t const *m);

/* Look-up kernel arrangement (experiment of round 4): nch in 2 .. 4 lets one lane walk up to nch
 * channels of a gas whose tables stand on the same (p, T) grid (brackets and LOS row once per segment and group);
 * nch < 2 (the default) keeps one (channel, gas) pair per workgroup, which is faster on every shape measured.
 * Strictly increasing tables only; bit-identical results. */
int  jur_model_set_ega_group(jur_model_t *m, int nch);
int  jur_model_ega_group(jur_model_t const *m);   /* channels per lane of the next call (0: one pair per workgroup) */

/* Frees the process-global state behind formod() / formod_GPU() / formod_pencil(): the lanes (streams, atmosphere,
 * workspaces, pinned images) and the emissivity tables loaded by the first call.  Upstream keeps its counterparts for
 * the life of the process (GPUdrivers.cu:263-273, 309; jr_common.h:60-78).  Waits for calls in flight; returns the
 * number of lanes freed (0: nothing was initialised).  The next formod() loads the tables again. */
int  jur_dropin_finalize(void);

/* Summed duration in ms and launch count of each kernel since the last query,
 * measured with HIP events on the launch stream while timing is enabled:
 * [0] jur_trace_kernel, [1] jur_ega_kernel, [2] jur_combine_kernel. */
int  jur_model_enable_timing(jur_model_t *m, int on);
int  jur_model_last_kernel_ms(jur_model_t *m, double out_ms[3], long out_launches[3]);
/* ... and of the fused kernel (call jur_model_last_kernel_ms first) */
int  jur_model_last_pencil_ms(jur_model_t *m, double *out_ms, long *out_launches);

/* ---- known-answer hooks (for tests; not on the product path) -------------------------------------------
 * The device functions of the path evaluated on host arrays of n inputs, one element per lane, so that each can be
 * checked against its reference counterpart at thresholds and range edges.
 * jur_kat_ega_eps: ega_eps (jr_common.h:237-268) of pair (ig, id).  mode 0 = the reference's bisections,
 *   1 = warm-started searches, 2 = + descriptors in LDS, 3 = + reciprocal bracket widths (what the bench runs);
 *   JUR_EINVAL when the tables do not admit the mode.  chain != 0: one lane evaluates the n inputs in order,
 *   carrying the search state from one to the next as the kernel does along a ray.
 * jur_kat_continua: out[4][n] = continua_ctmco2/h2o/n2/o2 (jr_common.h:315-390) of channel id (0 outside a window).
 * jur_kat_update: what 0: src[i] = src_planck_core(a[i]); (rad, tau)[i] updated by new_obs_core with
 *   tau_gas = b[i], beta_ds = c[i] (jr_common.h:220-224, 293-300); what 1: add_surface_core with surface
 *   temperature a[i] and, if b[i] != 0, brightness_core (jr_common.h:187-190, 227-234). */
int jur_kat_ega_eps(jur_model_t *m, int ig, int id, long n, double const *tau, double const *t, double const *u,
                    double const *p, int mode, int chain, double *out);
int jur_kat_continua(jur_model_t *m, int id, long n, double const *p, double const *t, double const *q,
                     double const *u_co2, double const *u_h2o, double *out);
int jur_kat_update(jur_model_t *m, int id, long n, int what, double const *a, double const *b, double const *c,
                   double *rad, double *tau, double *src);

#ifdef __cplusplus
}
#endif
#endif
