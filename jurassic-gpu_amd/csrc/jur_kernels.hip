// jur_kernels.hip -- CDNA4 (gfx950) kernels of the JURASSIC EGA forward model.
//
//   jur_trace_kernel      one lane per ray: line-of-sight ray tracing through the 1-D
//                         atmosphere (reference algorithm: jr_common.h:585-711), writes the
//                         per-segment state p, T, ds, k, q_H2O, u[g] to the LOS workspace
//                         (tiles of 64 ray slots, [tile][point][field][64]), the tangent
//                         point, the point count and the surface temperature.
//   jur_ega_kernel        one lane per ray, one (channel, gas) pair per workgroup: the sequential
//                         emissivity-growth recurrence along the line of sight with its band
//                         table look-ups (jr_common.h:237-280); writes the gas's transmittance of the
//                         path up to and including every segment.
//   jur_combine_kernel    one lane per ray, one channel per workgroup: continua
//                         (jr_common.h:315-390), product over gases, Planck source (:220-224),
//                         radiance update (:293-300); surface term, brightness temperature and
//                         the NaN mask in the epilogue (CPUdrivers.c:5-24, jr_common.h:193-210).
//   jur_combine_group_kernel  the same with up to four channels of a ray block as the wavefronts of one
//                         workgroup (what runs when there is more than one channel).
//   Two table-search strategies in jur_ega_kernel: WARM (tables whose axes and curves are sorted: every
//   bracket is unique, so the search resumes from the bracket the previous segment ended in -- the
//   accumulated transmittance only falls, the column only grows) and EXACT (the reference's bisections
//   probe for probe, for unsorted tables).
//   jur_cg_kernel         optional: Curtis-Godson means along the path, one wavefront per (ray, gas)
//                         pencil, along-path prefix sums as wavefront scans (jr_common.h:455-473).
//   jur_raykey_kernel     geometric tangent altitude per ray; rays are then processed in
//                         that order (hipCUB radix sort) so that the lanes of a wavefront
//                         walk similar paths.
//   jur_pencil_kernel     package-sized calls: the whole path of a few rays in ONE workgroup -- a tracer wavefront,
//                         emissivity-growth wavefronts (lane per (ray, channel, gas) chain) and radiance-update
//                         wavefronts hand the line of sight on through rings in LDS while it is being traced; the
//                         same device functions as the three batched kernels, bit-identical results.
//   jur_fov_kernel        field-of-view convolution of device arrays (formod_fov, jurassic.c:214-258).
//   jur_intpol_kernel     regridding of a track / point-cloud atmosphere (intpol_atm, jurassic.c:675-804).
//   jur_kat_*_kernel      known-answer hooks for tests: the device functions on arrays of inputs.
//
// All arithmetic is IEEE fp64; tables are fp32 in memory.  Compiled with -ffp-contract=off so that no fused
// multiply-adds are formed that the reference's x86-64 build does not form.
//   * Ray tracing, and the look-ups on tables that are not strictly increasing, keep the reference's operand order
//     throughout; where a division is replaced there (div_rcp, div_finite, div_const) the replacement returns the same
//     double as the division (tools/compare_builds.py).
//   * Since round 3 the look-up on strictly increasing tables (every table that passes the reference's row rule) and the
//     radiance update spend part of the contract's tolerance -- 1e-6 relative on radiances, 1e-9 in the test suite -- on
//     cheaper arithmetic: curve interpolations through bracket slopes formed once per model, blends through reciprocal
//     widths, the path transmittance carried as 1 - eps, one shared 1/T, exp through a table, tanh through exp
//     (lip_slope, lip_mulr, rcp_t, exp_tab, tanh_pos, segment_tau_gas below say what each costs in accuracy): 6e-12
//     relative on radiances against the bit-exact build.  The bit-exact look-ups remain as known-answer modes (jur_kat_ega_eps, modes 0 .. 2).

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdlib>
#include <mutex>
#include "jur_internal.h"

#define NLOS JUR_NLOS
#define TBLNS JUR_TBLNS

namespace {

// ---------------------------------------------------------------------------------------
// small helpers (jr_common.h:43-57)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double c01(double x) { return (x > 1.) ? 1. : ((x < 0.) ? 0. : x); }
// the same clamp for an x that is known not to be NaN (v_min_f64 / v_max_f64 return the other operand for a
// NaN, the comparisons above return the NaN): 2 instructions instead of 6
__device__ __forceinline__ double c01_num(double x) { return __builtin_fmax(__builtin_fmin(x, 1.), 0.); }

__device__ __forceinline__ double lip(double x0, double y0, double x1, double y1, double x) {
  return y0 + (x - x0) * (y1 - y0) / (x1 - x0);
}

// The same interpolation with the bracket width's reciprocal r = RN(1/(x1 - x0)) at hand.  The quotient
// a/b is formed as q = RN(a r), q' = RN(q + (a - b q) r) with fused multiply-adds: q' is the correctly
// rounded a/b (Markstein 1990: r correctly rounded and q faithful suffice), i.e. the very double the
// division returns, for 3 instructions instead of the ~14 of an fp64 division.  Needs b != 0 and finite.
__device__ __forceinline__ double div_rcp(double a, double b, double r) {
  double const q = a * r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}
__device__ __forceinline__ double lip_rcp(double x0, double y0, double x1, double y1, double x, double r) {
  return y0 + div_rcp((x - x0) * (y1 - y0), x1 - x0, r);
}

// a / b for finite a and finite b != 0 whose exponents are nowhere near the ends of the fp64 range
// (bracket widths of validated tables, emissivities, column densities, 1e-9 <= tau <= 1): the compiler's
// fp64 division is rcp + two Newton steps + quotient + residual correction, wrapped in v_div_scale /
// v_div_fmas (exponent rescaling) and v_div_fixup (zero, infinity, NaN, denormal results).  For such
// operands the wrapping is the identity, so the bare sequence returns the same double in 8 instead of
// 11 instructions.
__device__ __forceinline__ double div_finite(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
  double const q = a * r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}
__device__ __forceinline__ double lip_finite(double x0, double y0, double x1, double y1, double x) {
  return y0 + div_finite((x - x0) * (y1 - y0), x1 - x0);
}

// ---- arithmetic of the strict-table path (round 3) -------------------------------------------------------------
// The contract is 1e-6 relative on the radiances and the suite holds 1e-9; bit-identity with the reference's
// divisions is not needed to stay 1e3 .. 1e6 times inside that, and it is what 40 % of the look-up's fp64 work went
// into.  On tables whose axes and curves are strictly increasing (no zero-width bracket) the look-up uses:
//   lip_slope  the eight curve interpolations as y0 + (x - x0) s, one fused multiply-add, with the bracket's slope s
//              read from jur_sl_t (jur_slopes_kernel: the correctly rounded quotient of the bracket's fp64 differences,
//              formed once per model) where rounds 1 - 2 divided (8 instructions) and the first round-3 version
//              multiplied with a Newton-refined v_rcp_f64 (4, and the difference y1 - y0): good to an ulp or two of
//              the interpolated value;
//   lip_mulr   the p and T blends as y0 + ((x - x0) (y1 - y0)) r with r = RN(1 / (x1 - x0)): two roundings instead
//              of the division's one, 1 instruction instead of 3;
//   the path transmittance is carried as tau <- 1 - eps_t instead of tau <- tau * ((1 - eps_t) / tau): the same
//   number up to the two roundings the reference spends on dividing by tau and multiplying with it again (the
//   radiance update divides the products over the gases of consecutive segments instead: one division per
//   (channel, segment) in place of one per (channel, gas, segment)).
// tests/test_kat_gpu.py holds this path against the bit-exact ones (modes 0 .. 2), which stay the known-answer
// reference on the device.
__device__ __forceinline__ double lip_slope(double x0, double y0, double s, double x) { return __builtin_fma(x - x0, s, y0); }
__device__ __forceinline__ double lip_mulr(double x0, double y0, double y1, double x, double r) {
  return y0 + ((x - x0) * (y1 - y0)) * r;
}

// exp(x) of the radiance update and the continua (round 3).  The device library's exp is a degree-11 polynomial whose
// eleven coefficients the compiler keeps in 22 VGPRs across the segment loop and copies in front of every Horner
// step (v_mov_b64 + v_fmac_f64): ~35 vector instructions per call, three to four calls per (segment, channel).  This
// one reduces x = (64 m + j) ln2/64 + r, |r| <= ln2/128, takes 2^(j/64) from a 64-entry table (correctly rounded
// doubles; in LDS in the batched kernels) and needs a degree-5 polynomial in r (remainder r^6/720 < 4e-17):
// 17 vector instructions and one table read, within ~1 ulp of the library's result -- 1e-16 relative where the
// contract is 1e-6.  |x| is clamped to 1000 (beyond +-745 the result is 0 or inf either way), NaN goes through.
__device__ const double JUR_EXP2_64[64] = {
  1.0, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284,
  1.0442737824274138, 1.0556451783605572, 1.0671404006768237, 1.0787607977571199,
  1.0905077326652577, 1.102382583307841, 1.1143867425958924, 1.1265216186082418,
  1.1387886347566916, 1.1511892299529827, 1.1637248587775775, 1.1763969916502812,
  1.189207115002721, 1.202156731452703, 1.215247359980469, 1.22848053610687,
  1.241857812073484, 1.255380757024691, 1.2690509571917332, 1.2828700160787783,
  1.2968395546510096, 1.3109612115247644, 1.3252366431597413, 1.339667524053303,
  1.3542555469368927, 1.3690024229745905, 1.383909881963832, 1.3989796725383112,
  1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647,
  1.4768261459394993, 1.4929077282912648, 1.5091644275934228, 1.5255981507445384,
  1.5422108254079407, 1.559004400237837, 1.5759808451078865, 1.593142151342267,
  1.6104903319492543, 1.6280274218573478, 1.645755478153965, 1.6636765803267364,
  1.681792830507429, 1.7001063537185235, 1.718619298122478, 1.7373338352737062,
  1.7562521603732995, 1.7753764925265212, 1.7947090750031072, 1.8142521755003989,
  1.8340080864093424, 1.8539791250833855, 1.8741676341103, 1.8945759815869656,
  1.9152065613971474, 1.9360617934922943, 1.9571441241754002, 1.978456026387951};

template <class Tab>
__device__ __forceinline__ double exp_tab(Tab const &tab, double x) {
  double const xc = __builtin_fmin(__builtin_fmax(x, -1000.), 1000.);
  double const k = __builtin_rint(xc * 92.33248261689366);              // 64 / ln 2
  int const n = (int)k;
  double r = __builtin_fma(k, -0.01083042468962958, xc);                // ln2/64, upper bits (trailing zeros: exact)
  r = __builtin_fma(k, -6.619564634077006e-12, r);                      // ... and the rest
  double q = __builtin_fma(r, 1. / 120, 1. / 24);
  q = __builtin_fma(q, r, 1. / 6);
  q = __builtin_fma(q, r, 0.5);
  q = __builtin_fma(q * r, r, r);                                       // e^r - 1
  double const t = tab[n & 63];
  double const res = __builtin_ldexp(__builtin_fma(t, q, t), n >> 6);
  return (x != x) ? x : res;
}
// the table where a kernel has no copy in LDS
struct Exp2Global { __device__ __forceinline__ double operator[](int j) const { return JUR_EXP2_64[j]; } };
struct Exp2Lds { double const *t; __device__ __forceinline__ double operator[](int j) const { return t[j]; } };

// bracket search on an ascending or descending axis (jr_common.h:87-104)
__device__ __forceinline__ int locate_axis(double const *__restrict__ xx, int n, double x) {
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

// ---------------------------------------------------------------------------------------
// geometry (jr_common.h:475-500)
// ---------------------------------------------------------------------------------------
#define JUR_PI 3.14159265358979323846
#define RAD2GRD (180 / JUR_PI)
#define GRD2RAD (JUR_PI / 180)

__device__ __forceinline__ double norm3(double const x[3]) { return sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]); }

__device__ __forceinline__ void cart2geo(double const x[3], double &alt, double &lon, double &lat) {
  double const radius = norm3(x);
  lat = asin(x[2] / radius) * RAD2GRD;
  lon = atan2(x[1], x[0]) * RAD2GRD;
  alt = radius - JUR_RE;
}

__device__ __forceinline__ void geo2cart(double alt, double lon, double lat, double x[3]) {
  double const radius = alt + JUR_RE, clat = cos(lat * GRD2RAD);
  x[0] = radius * clat * cos(lon * GRD2RAD);
  x[1] = radius * clat * sin(lon * GRD2RAD);
  x[2] = radius * sin(lat * GRD2RAD);
}

__device__ __forceinline__ double refractivity(double p, double t) { return 7.753e-05 * p / t; }

// Same bracket as locate_axis for a sorted axis, found by walking from a guess (the previous
// point's bracket: altitude changes by <= RAYDZ per step).  dir > 0 ascending, < 0 descending.
__device__ __forceinline__ int locate_axis_from(double const *__restrict__ xx, int n, double x, int dir, int g) {
  int i = min(max(g, 0), n - 2);
  if (dir > 0) {
    while (i > 0 && xx[i] > x) --i;
    while (i < n - 2 && !(xx[i + 1] > x)) ++i;
  } else {
    while (i > 0 && xx[i] <= x) --i;
    while (i < n - 2 && !(xx[i + 1] <= x)) ++i;
  }
  return i;
}

// pressure and temperature of the profile slice [i0, i0+n) at altitude z0 (jr_common.h:549-555);
// `hint` carries the bracket from call to call (dir == 0: exact bisection, axis not sorted).
// WANT_R: the caller interpolates more quantities on the same bracket; for a sorted axis (dir != 0: z strictly
// monotone, bracket width non-zero) it gets rdz = RN(1 / (zb - za)) and every such interpolation, the
// temperature's included, divides through it (lip_rcp: the same doubles, 3 instructions per quotient).
template <bool WANT_R = false>
__device__ __forceinline__ int intpol_pt(jur_view_t const &v, int i0, int n, double z0, double &p, double &t, int dir,
                                         int &hint, double *rdz = nullptr) {
  int const loc = dir ? locate_axis_from(v.atm_z + i0, n, z0, dir, hint) : locate_axis(v.atm_z + i0, n, z0);
  hint = loc;
  int const ip = i0 + loc;
  double const za = v.atm_z[ip], zb = v.atm_z[ip + 1];
  // eip (jr_common.h:53-57) with log(p1/p0)/(z1-z0) taken from the per-level array that
  // jur_pslope_kernel filled with exactly that expression; NaN marks a non-positive pressure
  double const sl = v.atm_pslope[ip];
  if (WANT_R && dir) {
    double const r = 1. / (zb - za);
    *rdz = r;
    p = (sl == sl) ? v.atm_p[ip] * exp(sl * (z0 - za)) : lip_rcp(za, v.atm_p[ip], zb, v.atm_p[ip + 1], z0, r);
    t = lip_rcp(za, v.atm_t[ip], zb, v.atm_t[ip + 1], z0, r);
  } else {
    p = (sl == sl) ? v.atm_p[ip] * exp(sl * (z0 - za)) : lip(za, v.atm_p[ip], zb, v.atm_p[ip + 1], z0);
    t = lip(za, v.atm_t[ip], zb, v.atm_t[ip + 1], z0);
  }
  return ip;
}

// once per atmosphere upload: slope of ln p between neighbouring levels, as eip forms it
__global__ void jur_pslope_kernel(int n, double const *__restrict__ z, double const *__restrict__ p,
                                  double *__restrict__ slope) {
  int const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = __builtin_nan("");
  if (i + 1 < n) {
    double const y0 = p[i], y1 = p[i + 1];
    if ((y0 > 0) && (y1 > 0)) s = log(y1 / y0) / (z[i + 1] - z[i]);
  }
  slope[i] = s;
}

// ---------------------------------------------------------------------------------------
// ray tracing, one lane per ray
// ---------------------------------------------------------------------------------------
// Exit clipping (jr_common.h:637-648), once per ray: the previous point is rebuilt from its geodetic
// coordinates and the last segment is cut at the atmosphere's boundary.  Kept out of line: inlined, the
// asin/atan2/sin/cos polynomials have their ~40 constants hoisted in front of the stepping loop, where
// they cost the loop its registers (spills whose reloads queue behind the LOS stores).
// The point before the exit gets its segment length only when the exit is found (jr_common.h:645-646):
// its trapezoid weight and column densities, written provisionally one step earlier, are redone here from
// the same profile bracket (same inputs, same doubles).  Out of line, once per ray.
__device__ __attribute__((noinline)) void redo_columns(double const *__restrict__ atm_z, double const *__restrict__ atm_q,
                                                       int atm_np, int i0, int n, int ng, double z, double p, double t,
                                                       double dsn, double *__restrict__ los_u, size_t fs) {
  int const ip = i0 + locate_axis(atm_z + i0, n, z);
  double const za = atm_z[ip], zb = atm_z[ip + 1];
  for (int ig = 0; ig < ng; ig++) {
    double const *q = atm_q + (size_t)ig * atm_np;
    los_u[(size_t)ig * fs] = 10. * lip(za, q[ip], zb, q[ip + 1], z) * p / (JUR_BOLTZMANN * t) * dsn;
  }
}

struct ClipOut { double x0, x1, x2, frac; };
__device__ __attribute__((noinline)) ClipOut clip_exit(double px0, double px1, double px2, double pz, double x0, double x1,
                                                       double x2, double z, double zfrac) {
  double const px[3] = {px0, px1, px2};
  double xh[3], pzz, plon, plat;
  cart2geo(px, pzz, plon, plat);   // == the previous point's stored geolocation upstream
  geo2cart(pz, plon, plat, xh);
  double const frac = (zfrac - pz) / (z - pz);
  return {xh[0] + frac * (x0 - xh[0]), xh[1] + frac * (x1 - xh[1]), xh[2] + frac * (x2 - xh[2]), frac};
}

#define TR_PZ tr_sh[0][threadIdx.x]
#define TR_PX(i) tr_sh[1 + (i)][threadIdx.x]
#define TR_LZ0 tr_sh[4][threadIdx.x]
#define TR_LX0(i) tr_sh[5 + (i)][threadIdx.x]
#define TR_LZ1 tr_sh[8][threadIdx.x]
#define TR_LDS1 tr_sh[9][threadIdx.x]
#define TR_LZ2 tr_sh[10][threadIdx.x]
#define TR_LX2(i) tr_sh[11 + (i)][threadIdx.x]
#define TR_LDS2 tr_sh[14][threadIdx.x]
// value of lane K of the caller's quad (lanes 4q .. 4q+3), for all four lanes: two DPP moves
template <int K>
__device__ __forceinline__ double quad_bcast(double x) {
  constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);               // quad_perm:[K,K,K,K]
  int const lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, 0xf, 0xf, false);
  int const hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// One line of sight (traceray, jr_common.h:585-711, with tangent_point :502-539, trapezoid_rule_pos :437-443 and
// column_density :446-453 folded in), one lane per ray.  L says where the LOS fields of a point live -- the HBM
// workspace of the batched kernels, or a ring in LDS in the fused kernel -- through
//   L.at(field, point)      reference to the field's slot;  L.put(field, point, x) writes it
//   L.field_stride()        distance between the slots of consecutive fields (the emitters' columns are consecutive)
//   L.begin_point(i)        called before point i is first written
//   L.points_final(i)       called when points 0 .. i-1 will not change any more (point i still may: the exit
//                           clipping rewrites the point before the exit, jr_common.h:645-646)
//   L.never_enter(mask)     called once by all lanes together: the lanes whose rays never enter the atmosphere
//   L.ray_final(n, tsurf)   called when the ray has left the atmosphere with n points, all final (not for a ray
//                           that runs into the NLOS limit: see the returned np)
// tr_sh: the per-lane tangent-point bookkeeping in LDS, column threadIdx.x.
// QUAD: the ray is traced by the four lanes of a quad together (fused kernel, where the tracer wavefront has lanes
// to spare and the call's latency is the tracer's dependent instruction chain).  All four run the step
// redundantly -- same inputs, same doubles -- except for the refractivity gradient (jr_common.h:665-681), 58 % of
// the step's instructions: lane 0 takes the centre probe, lanes 1..3 one displaced probe each, and the four
// refractivities are exchanged with DPP moves.  Each probe is evaluated exactly as the sequential loop evaluates
// it (including the coordinates the loop has displaced and restored before it), so the results are the same doubles.
struct TraceResult { int np; double tsurf, tpz, tplon, tplat; };

// LANES = 4 is the QUAD described above, 1 a lane per ray.  (A pair of lanes per ray -- two probes each -- was built and
// measured in round 4: bit-identical and no faster than one lane at the launch sizes it was meant for, 100 000 rays:
// profiles/r04_size_sweep.json.)
template <class Los, int LANES = 1>
__device__ __forceinline__ TraceResult trace_ray(jur_view_t const &v, double const time, double const obsz, double const obslon,
                                                 double const obslat, double const vpz, double const vplon, double const vplat,
                                                 Los &L, double (&tr_sh)[15][64], int *status) {
  int const f_k = JUR_F_K, f_u = JUR_F_K + v.nw;
  double tsurf = -999;
  double tpz = vpz, tplon = vplon, tplat = vplat;
  int np = 0;

  // profile slice that carries this ray's time stamp (jr_common.h:127-154)
  int atm0, atmn;
  {
    int lo = 0, hi = v.atm_np - 1;
    while (hi > lo + 1) {
      int const i = (lo + hi) / 2;
      if (v.atm_time[i] < time) lo = i; else hi = i;
    }
    int const lower = (0 == lo) ? lo : hi;
    lo = lower;
    hi = v.atm_np - 1;
    while (hi > lo + 1) {
      int const i = (lo + hi) / 2;
      if (v.atm_time[i] > time) hi = i; else lo = i;
    }
    int const upper = (hi == v.atm_np - 1) ? v.atm_np : hi;
    atm0 = lower;
    atmn = upper - lower;
  }
  // altitude range of the first column of that slice (jr_common.h:411-420)
  double zmin = v.atm_z[atm0], zmax = zmin;
  {
    double const lon0 = v.atm_lon[atm0], lat0 = v.atm_lat[atm0];
    for (int ipp = atm0; (ipp < atm0 + atmn) && (v.atm_lon[ipp] == lon0) && (v.atm_lat[ipp] == lat0); ++ipp) {
      zmax = fmax(zmax, v.atm_z[ipp]);
      zmin = fmin(zmin, v.atm_z[ipp]);
    }
  }

  bool const outside = (obsz < zmin) || (vpz > zmax - 0.001);
  // (fused kernel: which lanes' rays never enter the atmosphere is said by ONE lane, in its own program order
  // before it starts stepping -- a branch of the lanes concerned could be scheduled behind the others' loop)
  L.never_enter(__ballot(outside));
  if (!outside) {
    double x[3], ex0[3], xobs[3], xvp[3];
    geo2cart(obsz, obslon, obslat, xobs);
    geo2cart(vpz, vplon, vplat, xvp);
    for (int i = 0; i < 3; i++) ex0[i] = xvp[i] - xobs[i];
    double const norm = norm3(ex0);
    for (int i = 0; i < 3; i++) {
      ex0[i] /= norm;
      x[i] = xobs[i];
    }
    if (obsz > zmax) {  // observer above the atmosphere: bisect for the entry point (:610-621)
      double dmax = norm, dmin = 0.;
      while (fabs(dmin - dmax) > 0.001) {
        double const d = 0.5 * (dmax + dmin);
        for (int i = 0; i < 3; i++) x[i] = xobs[i] + d * ex0[i];
        double const z = norm3(x) - JUR_RE;
        if ((z <= zmax) && (z > zmax - 0.001)) break;
        if (z < zmax - 0.0005) dmax = d; else dmin = d;
      }
    }

    // Bookkeeping for the tangent point: the three points around the lowest one.  Longitude and
    // latitude of a LOS point are only ever needed for the point before the exit (clipping), the two
    // neighbours of the lowest point and the last point, so the per-step asin/atan2 of cart2geo is
    // deferred: the Cartesian position is kept and converted on demand (same input, same result).
    // This bookkeeping is written often and read a few times per ray; it lives in LDS (one column per
    // lane) so that its 30 registers stay free for the loop: the kernel is latency-bound and was
    // spilling to scratch, whose stores queue in front of the profile loads (one vmcnt for both).
    double z_low = 1e99;
    int low_idx = -1;
    TR_PZ = 0; TR_PX(0) = 0; TR_PX(1) = 0; TR_PX(2) = 0;    // previous point (np-1)
    TR_LZ0 = 0; TR_LX0(0) = 0; TR_LX0(1) = 0; TR_LX0(2) = 0;   // point low_idx-1
    TR_LZ1 = 0; TR_LDS1 = 0;                                 // point low_idx
    TR_LZ2 = 0; TR_LX2(0) = 0; TR_LX2(1) = 0; TR_LX2(2) = 0; TR_LDS2 = 0;   // point low_idx+1

    // altitude brackets are resumed from the previous point when the slice's axis is strictly monotone
    int const zdir = (v.atm_sorted && atmn >= 2) ? ((v.atm_z[atm0] < v.atm_z[atm0 + 1]) ? 1 : -1) : 0;
    int zhint = (zdir > 0) ? atmn - 2 : 0, rhint = zhint;   // rays usually enter at the top

    // raw segment lengths of the two previous points: the trapezoid rule (jr_common.h:437-443) and the
    // column densities (:446-453) of a point are written in its own step, ds[i] <- (ds[i-1] + ds[i]) / 2
    double ds_p = 0, ds_pp = 0;

    int stop = 0;
    for (; np < NLOS; ++np) {
      // the bookkeeping written to LDS is to be read back from there, not kept in registers as well
      asm volatile("" ::: "memory");
      L.begin_point(np);                  // (fused kernel: the ring slot of this point must be free)
      double ds = v.rayds;
      double const dz = v.raydz;
      if (dz > 0.) {
        double const norm_x = 1.0 / norm3(x);
        double dot = 0.;
        for (int i = 0; i < 3; i++) dot += ex0[i] * x[i] * norm_x;
        double const cosa = fabs(dot);
        if (cosa != 0.) ds = fmin(ds, dz / cosa);
      }
      double z = norm3(x) - JUR_RE;
      if ((z < zmin) || (z > zmax)) {  // LOS left the atmosphere: clip the last segment (:637-648)
        stop = (z < zmin) ? 2 : 1;
        if (np > 0) {
          ClipOut const co = clip_exit(TR_PX(0), TR_PX(1), TR_PX(2), TR_PZ, x[0], x[1], x[2], z, (z < zmin) ? zmin : zmax);
          x[0] = co.x0; x[1] = co.x1; x[2] = co.x2;
          double const frac = co.frac;
          z = norm3(x) - JUR_RE;
          double const dsp = ds * frac;            // the previous point's segment, cut at the boundary
          if (low_idx == np - 1) TR_LDS1 = dsp;
          if (low_idx + 1 == np - 1) TR_LDS2 = dsp;
          double const dsn = (np >= 2) ? 0.5 * (ds_pp + dsp) : dsp * 0.5;
          L.put(JUR_F_DS, np - 1, dsn);
          redo_columns(v.atm_z, v.atm_q, v.atm_np, atm0, atmn, v.ng, TR_PZ, L.at(JUR_F_P, np - 1), L.at(JUR_F_T, np - 1), dsn,
                       &L.at(f_u, np - 1), L.field_stride());
          ds_p = dsp;
        }
        ds = 0.;
      }

      double p, t, rdz = 0;
      int const ia = intpol_pt<true>(v, atm0, atmn, z, p, t, zdir, zhint, &rdz);
      double const dsn = (np >= 1) ? 0.5 * (ds_p + ds) : ds * 0.5;   // redone for the point before the exit
      L.put(JUR_F_DS, np, dsn);
      {  // remaining quantities on the same bracket (jr_common.h:557-567)
        double const za = v.atm_z[ia], zb = v.atm_z[ia + 1];
        double const kt = JUR_BOLTZMANN * t, rkt = 1. / kt;   // one division for all emitters' columns (div_rcp)
        // (LANES > 1: the lanes of the ray share the emitters' columns between them; nobody in the tracer reads them back)
        for (int ig = LANES > 1 ? (int)(threadIdx.x & (LANES - 1)) : 0; ig < v.ng; ig += LANES) {
          double const *q = v.atm_q + (size_t)ig * v.atm_np;
          double qv;
          if (zdir) qv = lip_rcp(za, q[ia], zb, q[ia + 1], z, rdz); else qv = lip(za, q[ia], zb, q[ia + 1], z);
          L.put(f_u + ig, np, div_rcp(10. * qv * p, kt, rkt) * dsn);
          if (ig == v.ig_h2o) L.put(JUR_F_QH2O, np, qv);
        }
        for (int iw = 0; iw < v.nw; iw++) {
          double const *k = v.atm_k + (size_t)iw * v.atm_np;
          if (zdir) L.put(f_k + iw, np, lip_rcp(za, k[ia], zb, k[ia + 1], z, rdz));
          else L.put(f_k + iw, np, lip(za, k[ia], zb, k[ia + 1], z));
        }
      }
      L.put(JUR_F_P, np, p);
      L.put(JUR_F_T, np, t);
      ds_pp = ds_p; ds_p = ds;

      if (low_idx >= 0 && low_idx == np - 1) { TR_LZ2 = z; TR_LDS2 = ds; for (int i = 0; i < 3; i++) TR_LX2(i) = x[i]; }
      if (z < z_low) {
        z_low = z;
        low_idx = np;
        TR_LZ0 = TR_PZ;
        for (int i = 0; i < 3; i++) TR_LX0(i) = TR_PX(i);
        TR_LZ1 = z; TR_LDS1 = ds;
      }
      TR_PZ = z;                          // also the last point's altitude after the loop
      for (int i = 0; i < 3; i++) TR_PX(i) = x[i];

      if (stop) {
        tsurf = (stop == 2 ? t : -999);
        L.ray_final(np + 1, tsurf);       // (fused kernel: this ray has np + 1 points, all of them final)
        break;
      }
      L.points_final(np);                 // (fused kernel: points 0 .. np-1 of this ray will not change any more)

      double n = 1., ngr[3] = {0., 0., 0.};
      if (v.refrac && z <= 60.) {  // refractivity gradient by finite differences (:665-681)
        n += refractivity(p, t);
        double xh[3], zz, llon, llat, pp, tt;
        for (int i = 0; i < 3; i++) xh[i] = x[i] + 0.5 * ds * ex0[i];
        if constexpr (LANES == 4) {
          // one probe per lane of the quad.  The sequential loop displaces a coordinate by h, probes, and takes h
          // off again before it goes on: probe i sees (x_k + h) - h in the coordinates k < i
          double const h = 0.02;
          int const j = threadIdx.x & 3;
          double const r0 = (xh[0] + h) - h, r1 = (xh[1] + h) - h;
          double const xq[3] = {(j == 1) ? xh[0] + h : (j >= 2) ? r0 : xh[0], (j == 2) ? xh[1] + h : (j == 3) ? r1 : xh[1],
                                (j == 3) ? xh[2] + h : xh[2]};
          cart2geo(xq, zz, llon, llat);
          double rdzb = 0;
          intpol_pt<true>(v, atm0, atmn, zz, pp, tt, zdir, rhint, &rdzb);   // (lip_rcp == lip: same doubles as the plain search)
          double const nj = refractivity(pp, tt), n2 = quad_bcast<0>(nj);
          ngr[0] = div_rcp(quad_bcast<1>(nj) - n2, h, 1. / h);
          ngr[1] = div_rcp(quad_bcast<2>(nj) - n2, h, 1. / h);
          ngr[2] = div_rcp(quad_bcast<3>(nj) - n2, h, 1. / h);
        } else {
        cart2geo(xh, zz, llon, llat);
        double rdzb = 0;
        int const ib = intpol_pt<true>(v, atm0, atmn, zz, pp, tt, zdir, rhint, &rdzb);
        double const n2 = refractivity(pp, tt);
        // the three displaced probes lie 0.02 km away: almost always in the bracket just found, whose six
        // values are then reused instead of being looked up and loaded again
        double const za = v.atm_z[ib], zb = v.atm_z[ib + 1], pa = v.atm_p[ib], pb = v.atm_p[ib + 1],
                     ta = v.atm_t[ib], tb = v.atm_t[ib + 1], sl = v.atm_pslope[ib];
        bool const first = (ib == atm0), lastb = (ib == atm0 + atmn - 2);
        for (int i = 0; i < 3; i++) {
          double const h = 0.02;
          xh[i] += h;
          cart2geo(xh, zz, llon, llat);
          bool const inside = (zdir > 0) ? ((zz >= za || first) && (zz < zb || lastb))
                            : (zdir < 0) ? ((zz < za || first) && (zz >= zb || lastb)) : false;
          if (inside) {
            pp = (sl == sl) ? pa * exp(sl * (zz - za)) : lip_rcp(za, pa, zb, pb, zz, rdzb);   // inside => sorted axis
            tt = lip_rcp(za, ta, zb, tb, zz, rdzb);
          } else {
            intpol_pt(v, atm0, atmn, zz, pp, tt, zdir, rhint);
          }
          ngr[i] = div_rcp(refractivity(pp, tt) - n2, h, 1. / h);
          xh[i] -= h;
        }
        }
      }
      double ex1[3];
      for (int i = 0; i < 3; i++) ex1[i] = ex0[i] * n + ds * ngr[i];
      double const norm_ex1 = norm3(ex1), rnorm = 1. / norm_ex1;
      for (int i = 0; i < 3; i++) {
        ex1[i] = div_rcp(ex1[i], norm_ex1, rnorm);
        x[i] += 0.5 * ds * (ex0[i] + ex1[i]);
        ex0[i] = ex1[i];
      }
    }
    ++np;
    if (NLOS <= np) {  // the reference aborts here (jr_common.h:693-695); flag and clamp
      // a plain store of the one value this path ever reports: the word may live in pinned host memory (the fused
      // kernel of a package-sized host call writes it across PCIe), where a device atomic needs PCIe atomics
      // from the platform and is silently dropped without them
      *reinterpret_cast<volatile int *>(status) = 1;
      np = NLOS - 1;
    }

    asm volatile("" ::: "memory");
    // tangent point from the raw segment lengths (jr_common.h:502-539)
    if (low_idx <= 0 || low_idx >= np - 1) {
      double zz;
      double const px[3] = {TR_PX(0), TR_PX(1), TR_PX(2)};   // the last point here
      cart2geo(px, zz, tplon, tplat);
      tpz = TR_PZ;
    } else {
      double const lz0 = TR_LZ0, lz2 = TR_LZ2;
      double const lx0[3] = {TR_LX0(0), TR_LX0(1), TR_LX0(2)}, lx2[3] = {TR_LX2(0), TR_LX2(1), TR_LX2(2)};
      double const yy0 = lz0, yy1 = TR_LZ1, yy2 = lz2, ds0 = TR_LDS1, ds1 = TR_LDS2,
                   dyy10 = yy1 - yy0, dyy21 = yy2 - yy1,
                   x1 = sqrt(ds0 * ds0 - dyy10 * dyy10),
                   x2 = x1 + sqrt(ds1 * ds1 - dyy21 * dyy21),
                   dx12 = x1 - x2,
                   a = (dyy10 * x2 + (yy0 - yy2) * x1) / (x1 * x2 * dx12),
                   b = dyy10 / x1 - a * x1,
                   cc = yy0,
                   xt = -b / (2 * a);
      tpz = (a * xt + b) * xt + cc;
      double w[3], v0[3], v2[3], dummy, llon0, llat0, llon2, llat2;
      cart2geo(lx0, dummy, llon0, llat0);
      cart2geo(lx2, dummy, llon2, llat2);
      geo2cart(lz0, llon0, llat0, v0);
      geo2cart(lz2, llon2, llat2, v2);
      for (int i = 0; i < 3; i++) w[i] = lip(0.0, v0[i], x2, v2[i], xt);
      cart2geo(w, dummy, tplon, tplat);
    }

  }

  return {np, tsurf, tpz, tplon, tplat};
}

// LOS fields in the HBM workspace: tiles of 64 ray slots (one wavefront), [tile][point][field][64].  Everything a
// wavefront reads or writes for one LOS point is one contiguous run of nfield x 512 B, and everything it touches in its
// life lies within NLOS x nfield x 512 B (2 MB for ten fields): a handful of pages per wavefront, where the round-1/2
// layout [field][point][ray] put every field of every point on a page of its own (rows 8 MB apart for 1e6 rays).
__device__ __forceinline__ size_t los_tile_doubles(int nfield) { return (size_t)NLOS * nfield * 64; }
// First point slot of a tile in the transmittance workspace: tiles of NLOS points each, or -- when the call has been
// laid out by the path lengths that occur (jur_model.c: compact chunks) -- the prefix sum of the tiles' longest paths.
__device__ __forceinline__ size_t eps_tile_point0(jur_chunk_t const &c, int tile) {
  return c.eps_off ? (size_t)c.eps_off[tile] : (size_t)tile * NLOS;
}
struct LosWorkspace {
  double *tile;                 // first double of this wavefront's tile, + lane
  int nfield;
  __device__ __forceinline__ double &at(int field, int ip) const { return tile[((size_t)ip * nfield + field) * 64]; }
  // the LOS rows leave the tracer for good (the look-up kernel reads them from HBM, 20 GB per 1e6 limb rays later):
  // written past the L2, which then keeps the atmosphere's profiles -- jur_trace_kernel 8.23 -> 7.70 ms per 1e6 limb rays
  __device__ __forceinline__ void put(int field, int ip, double x) const {
    __builtin_nontemporal_store(x, &tile[((size_t)ip * nfield + field) * 64]);
  }
  __device__ __forceinline__ size_t field_stride() const { return 64; }
  __device__ __forceinline__ void begin_point(int) const {}
  __device__ __forceinline__ void points_final(int) const {}
  __device__ __forceinline__ void ray_final(int, double) const {}
  __device__ __forceinline__ void never_enter(unsigned long long) const {}
};

// 4 waves per SIMD (128 VGPRs, some scratch): measured 20 % faster than 2 waves without spills once a
// launch carries enough rays (>= 4 x 131072) to fill them
__global__ __launch_bounds__(64, 4) void jur_trace_kernel(jur_view_t v, jur_chunk_t c) {
  __shared__ double tr_sh[15][64];
  int const r = blockIdx.x * blockDim.x + threadIdx.x;   // slot in the chunk
  if (r >= c.n) return;
  long const ray = c.order ? (long)c.order[r] : c.first + r;
  LosWorkspace L{c.los + (size_t)(r >> 6) * los_tile_doubles(JUR_F_K + v.nw + v.ng) + (r & 63), JUR_F_K + v.nw + v.ng};
  TraceResult const t = trace_ray(v, c.geom[0][ray], c.geom[1][ray], c.geom[2][ray], c.geom[3][ray], c.geom[4][ray],
                                  c.geom[5][ray], c.geom[6][ray], L, tr_sh, c.status);
  c.np[r] = t.np;
  c.tsurf[r] = t.tsurf;
  if (c.np_out) c.np_out[ray] = t.np;
  c.tp[0][ray] = t.tpz;
  c.tp[1][ray] = t.tplon;
  c.tp[2][ray] = t.tplat;
}

// The same with a quad of lanes per ray, for launches that leave most of the chip's tracer wavefront slots empty with one
// lane per ray (up to 65 536 rays: a quad per ray still fits the 4096 slots): the step's dependent chain -- what such a
// launch waits for -- is shorter by the refraction probes and the emitters' columns the other lanes take.  Same doubles;
// the lanes of a ray write the same values to the same slots.  Measured (profiles/r04_size_sweep.json): five emitters,
// 10 000 .. 50 000 limb rays: 1.9 .. 2.0 -> 1.2 .. 1.4 ms; ONE emitter (nadir shape): no gain at 20 000, +15 % at
// 50 000 -- with a single column there is little to share beside the probes, whose profile look-ups each lane then
// does for itself.  Hence the rule in jurk_launch_trace.
template <int LANES>
__global__ __launch_bounds__(64, 4) void jur_trace_lanes_kernel(jur_view_t v, jur_chunk_t c) {
  __shared__ double tr_sh[15][64];
  int const r = blockIdx.x * (64 / LANES) + (int)(threadIdx.x / LANES);   // slot in the chunk
  if (r >= c.n) return;
  long const ray = c.order ? (long)c.order[r] : c.first + r;
  LosWorkspace L{c.los + (size_t)(r >> 6) * los_tile_doubles(JUR_F_K + v.nw + v.ng) + (r & 63), JUR_F_K + v.nw + v.ng};
  TraceResult const t = trace_ray<LosWorkspace, LANES>(v, c.geom[0][ray], c.geom[1][ray], c.geom[2][ray], c.geom[3][ray],
                                                       c.geom[4][ray], c.geom[5][ray], c.geom[6][ray], L, tr_sh, c.status);
  if (threadIdx.x & (LANES - 1)) return;
  c.np[r] = t.np;
  c.tsurf[r] = t.tsurf;
  if (c.np_out) c.np_out[ray] = t.np;
  c.tp[0][ray] = t.tpz;
  c.tp[1][ray] = t.tplon;
  c.tp[2][ray] = t.tplat;
}

#undef TR_PZ
#undef TR_PX
#undef TR_LZ0
#undef TR_LX0
#undef TR_LZ1
#undef TR_LDS1
#undef TR_LZ2
#undef TR_LX2
#undef TR_LDS2

// ---------------------------------------------------------------------------------------
// emissivity-growth look-up (ega_eps, jr_common.h:237-268)
// ---------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) Lvl { double p; int nt; int c0; };
struct __attribute__((aligned(16))) Crv { double t; int nu; int e0; };
struct __attribute__((aligned(8))) Ue { float u; float eps; };
struct __attribute__((aligned(8))) Ue2 { Ue a, b; };
static_assert(sizeof(Lvl) == sizeof(jur_lvl_t) && sizeof(Crv) == sizeof(jur_crv_t) && sizeof(Ue) == sizeof(jur_ue_t), "layout");

// Table element access as (wave-uniform base pointer) + (32-bit byte offset): the address is
// formed by the memory instruction itself (SGPR base + VGPR offset) instead of 64-bit vector
// arithmetic per load.  Descriptors count from their array's base (the host refuses sets beyond 2^27 levels or
// curves); table entries count from the first entry of their (gas, channel) pair (PairDesc::ueb, a 64-bit base that
// is uniform wherever a workgroup works on one pair), so a set may hold any number of entries.
template <class T>
__device__ __forceinline__ T ldg(void const *__restrict__ base, unsigned index) {
  return *reinterpret_cast<T const *>(static_cast<char const *>(base) + (size_t)(index * (unsigned)sizeof(T)));
}
// The path transmittances -- 40 GB per 1e6 limb rays, written once by the look-up kernel and read once by the radiance
// update -- go past the L2 as non-temporal accesses: the lines that ARE re-used there (the LOS rows the workgroups of a
// ray block share, the tables) stay longer.  Measured (round 4, profiles/r04_cache_policy_experiment.json): look-up
// 35.1 -> 33.7 ms with the stores, radiance update 10.69 -> 10.53 ms with the loads.  The LOS loads of the look-up as
// non-temporal: 38.5 ms -- that re-use is real.
__device__ __forceinline__ void st_stream(double *base, unsigned lane, double x) {
  __builtin_nontemporal_store(x, reinterpret_cast<double *>(reinterpret_cast<char *>(base) + (size_t)(lane * 8u)));
}
__device__ __forceinline__ double ld_stream(double const *__restrict__ base, unsigned lane) {
  return __builtin_nontemporal_load(reinterpret_cast<double const *>(reinterpret_cast<char const *>(base) + (size_t)(lane * 8u)));
}
__device__ __forceinline__ Ue ld_ue(void const *__restrict__ ue, unsigned idx) { return ldg<Ue>(ue, idx); }
// entries idx and idx+1 of a curve are adjacent: one 16-byte load
__device__ __forceinline__ void ld_pair(void const *__restrict__ ue, unsigned idx, Ue &a, Ue &b) {
  Ue2 const ab = *reinterpret_cast<Ue2 const *>(static_cast<char const *>(ue) + (size_t)(idx * 8u));
  a = ab.a; b = ab.b;
}

// slope of the bracket [idx, idx+1]: WHICH = 0 du/deps (get_u), 1 deps/du (get_eps)
template <int WHICH>
__device__ __forceinline__ double ld_slope(void const *__restrict__ sl, unsigned idx) {
  return *reinterpret_cast<double const *>(static_cast<char const *>(sl) + (size_t)(idx * 16u + WHICH * 8u));
}

// The same loads where the lanes of a wavefront may all want the same element -- rays sorted by tangent altitude walk
// through the same brackets of the same curves nearly in step.  Then the element is fetched ONCE through the scalar
// data cache (s_load into SGPRs) instead of 64 times through the vector L1, whose address pipeline -- 16 accesses per
// gather -- is what the kernel runs out of next to its vector ALUs (profiles/pmc_current.json: TCP stalled or busy
// > 90 % of the time).  Any lane that differs sends the wavefront through the gather; the values are the same.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <class T>
__device__ __forceinline__ T ld_scalar(void const *__restrict__ base, unsigned byte0) {   // byte0: wave-uniform offset
  typedef T const __attribute__((address_space(4))) *cptr;
  return *(cptr)(unsigned long long)(static_cast<char const *>(base) + (size_t)byte0);
}
__device__ __forceinline__ bool all_lanes_at(unsigned idx, unsigned &idx0) {
  idx0 = __builtin_amdgcn_readfirstlane(idx);
  return __builtin_amdgcn_ballot_w64(idx != idx0) == 0;
}
// (Serving the lanes that disagree in rounds of scalar loads -- a "waterfall", one round per distinct element -- was
// measured too: 59.6 ms with two rounds, 47.8 with one round and the gather for the rest, against 40.0 for the plain
// all-or-nothing test below; each round waits for its own scalar load inside divergent control flow.)
__device__ __forceinline__ Ue ld_ue_u(void const *__restrict__ ue, unsigned idx) {
  unsigned i0;
  if (all_lanes_at(idx, i0)) { f32x2 const v = ld_scalar<f32x2>(ue, i0 * 8u); return Ue{v.x, v.y}; }
  return ld_ue(ue, idx);
}
// the brackets of the two curves of a pressure level, requested together
__device__ __forceinline__ void ld_pair2(void const *__restrict__ ue, unsigned ia, unsigned ib, Ue &a0, Ue &b0, Ue &a1, Ue &b1) {
  unsigned fa, fb;
  bool const ua = all_lanes_at(ia, fa), ub = all_lanes_at(ib, fb);
  if (ua & ub) {
    f32x4 const x = ld_scalar<f32x4>(ue, fa * 8u), y = ld_scalar<f32x4>(ue, fb * 8u);
    a0 = Ue{x.x, x.y}; b0 = Ue{x.z, x.w}; a1 = Ue{y.x, y.y}; b1 = Ue{y.z, y.w};
    return;
  }
  ld_pair(ue, ia, a0, b0);
  ld_pair(ue, ib, a1, b1);
}
template <int WHICH>
__device__ __forceinline__ void ld_slope2(void const *__restrict__ sl, unsigned ia, unsigned ib, double &s0, double &s1) {
  unsigned fa, fb;
  bool const ua = all_lanes_at(ia, fa), ub = all_lanes_at(ib, fb);
  if (ua & ub) {
    s0 = ld_scalar<double>(sl, fa * 16u + WHICH * 8u);
    s1 = ld_scalar<double>(sl, fb * 16u + WHICH * 8u);
    return;
  }
  s0 = ld_slope<WHICH>(sl, ia);
  s1 = ld_slope<WHICH>(sl, ib);
}

template <bool ON_EPS>
__device__ __forceinline__ double ukey(Ue const &e) { return ON_EPS ? (double)e.eps : (double)e.u; }

// EXACT: the reference's bisection (locate_tbl_id, jr_common.h:116-125) on curve [e0, e0+n)
template <bool ON_EPS>
__device__ __forceinline__ int bisect_curve(void const *__restrict__ ue, unsigned e0, int n, double x) {
  int ilo = 0, ihi = n - 1;
  while (ihi > ilo + 1) {
    int const i = (ihi + ilo) >> 1;
    if (ukey<ON_EPS>(ld_ue(ue, e0 + i)) > x) ihi = i; else ilo = i;
  }
  return ilo;
}

// WARM: move bracket i (entries a = e[i], b = e[i+1] already loaded) to the one that holds x:
// key(e[i]) <= x < key(e[i+1]), clamped to [0, n-2].  One step is the common case; otherwise
// gallop, then bisect inside the gap.
template <bool ON_EPS>
__device__ __forceinline__ void seek_curve(void const *__restrict__ ue, unsigned e0, int n, double x, int &i, Ue &a, Ue &b) {
  bool const up = x >= ukey<ON_EPS>(b), down = x < ukey<ON_EPS>(a);
  if (!(up | down)) return;                      // one test for the common case: still in the bracket
  if (up) {
    if (i >= n - 2) return;
    Ue const c = ld_ue(ue, e0 + i + 2);
    if (i + 2 >= n - 1 || ukey<ON_EPS>(c) > x) { ++i; a = b; b = c; return; }
    int lo = i + 2, hi, step = 2;
    for (;;) {
      hi = lo + step;
      if (hi >= n - 1) { hi = n - 1; break; }
      if (ukey<ON_EPS>(ld_ue(ue, e0 + hi)) > x) break;
      lo = hi;
      step <<= 1;
    }
    while (hi > lo + 1) {
      int const mid = (lo + hi) >> 1;
      if (ukey<ON_EPS>(ld_ue(ue, e0 + mid)) > x) hi = mid; else lo = mid;
    }
    i = lo;
    ld_pair(ue, e0 + i, a, b);
   
  } else {
    if (i <= 0) return;
    Ue const c = ld_ue(ue, e0 + i - 1);
    if (i - 1 <= 0 || ukey<ON_EPS>(c) <= x) { --i; b = a; a = c; return; }
    int hi = i - 1, lo, step = 2;
    for (;;) {
      lo = hi - step;
      if (lo <= 0) { lo = 0; break; }
      if (ukey<ON_EPS>(ld_ue(ue, e0 + lo)) <= x) break;
      hi = lo;
      step <<= 1;
    }
    while (hi > lo + 1) {
      int const mid = (lo + hi) >> 1;
      if (ukey<ON_EPS>(ld_ue(ue, e0 + mid)) > x) hi = mid; else lo = mid;
    }
    i = lo;
    ld_pair(ue, e0 + i, a, b);
   
  }
}

// fp32 -> fp64 as an instruction the compiler may neither repeat nor re-create at a later use: left to itself it
// keeps the four floats of a bracket and converts them again wherever a double is wanted (48 conversions per
// look-up, 12 % of its vector instructions); held as doubles from the test that needs them first, it is 28.
__device__ __forceinline__ double cvt_keep(float f) {
  double d;
  asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(f));
  return d;
}
template <bool ON_EPS>
__device__ __forceinline__ double kkey(Ue const &e) { return cvt_keep(ON_EPS ? e.eps : e.u); }

// seek_curve that works on, and hands back, the keys of its bracket as doubles: ka = key(e[i]), kb = key(e[i+1]) --
// formed here unless the caller HAS them already (get_eps's search starts on the bracket whose column keys get_u's
// interpolation has just converted).  Same probes, same bracket as seek_curve.  (The two far-move branches each end in
// their own reload: with a shared tail the register allocator needs 91 VGPRs for jur_ega_kernel instead of 71 --
// tests/test_abi_cpu.py watches that number.)
template <bool ON_EPS, bool HAVE>
__device__ __forceinline__ void seek_curve_keys(void const *__restrict__ ue, unsigned e0, int n, double x, int &i, Ue &a, Ue &b,
                                                double &ka, double &kb) {
  if (!HAVE) { ka = kkey<ON_EPS>(a); kb = kkey<ON_EPS>(b); }
  bool const up = x >= kb, down = x < ka;
  if (!(up | down)) return;
  if (up) {
    if (i >= n - 2) return;
    Ue const c = ld_ue_u(ue, e0 + i + 2);
    double const kc = kkey<ON_EPS>(c);
    if (i + 2 >= n - 1 || kc > x) { ++i; a = b; b = c; ka = kb; kb = kc; return; }
    int lo = i + 2, hi, step = 2;
    for (;;) {
      hi = lo + step;
      if (hi >= n - 1) { hi = n - 1; break; }
      if (ukey<ON_EPS>(ld_ue(ue, e0 + hi)) > x) break;
      lo = hi;
      step <<= 1;
    }
    while (hi > lo + 1) {
      int const mid = (lo + hi) >> 1;
      if (ukey<ON_EPS>(ld_ue(ue, e0 + mid)) > x) hi = mid; else lo = mid;
    }
    i = lo;
    ld_pair(ue, e0 + i, a, b);
    ka = kkey<ON_EPS>(a); kb = kkey<ON_EPS>(b);
  } else {
    if (i <= 0) return;
    Ue const c = ld_ue_u(ue, e0 + i - 1);
    double const kc = kkey<ON_EPS>(c);
    if (i - 1 <= 0 || kc <= x) { --i; b = a; a = c; kb = ka; ka = kc; return; }
    int hi = i - 1, lo, step = 2;
    for (;;) {
      lo = hi - step;
      if (lo <= 0) { lo = 0; break; }
      if (ukey<ON_EPS>(ld_ue(ue, e0 + lo)) <= x) break;
      hi = lo;
      step <<= 1;
    }
    while (hi > lo + 1) {
      int const mid = (lo + hi) >> 1;
      if (ukey<ON_EPS>(ld_ue(ue, e0 + mid)) > x) hi = mid; else lo = mid;
    }
    i = lo;
    ld_pair(ue, e0 + i, a, b);
    ka = kkey<ON_EPS>(a); kb = kkey<ON_EPS>(b);
  }
}

// ---- bracket records (round 4) ---------------------------------------------------------------------------------
// Everything the strict-table look-up needs of bracket [i, i+1] of a curve in ONE 32-byte record, indexed like the
// entries: the four fp32 values of the two entries and the two slopes.  Rounds 1 - 3 fetched them from two arrays in
// three accesses (16-byte entry pair, 8-byte slope for get_u, 8-byte slope for get_eps): two to three cache lines and
// three uniformity tests per curve where this is one line (records never straddle one) and one test -- and what the
// round-4 experiments say holds the kernel is the number of table lines a wavefront touches per segment
// (profiles/r04_ega_*_experiment.json).  A search that leaves its bracket by one step fetches the neighbouring record,
// which IS the new bracket (keys and slopes); get_eps finds itself in get_u's bracket two times out of three and
// fetches nothing.  Same operations on the same operands as the separate arrays: the same doubles.
struct __attribute__((aligned(32))) Rec { float u0, e0, u1, e1; double du_de, de_du; };
static_assert(sizeof(Rec) == sizeof(jur_rec_t) && sizeof(Rec) == 32, "layout");
typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ Rec rec_from(f32x8 const v) {
  Rec r;
  r.u0 = v.s0; r.e0 = v.s1; r.u1 = v.s2; r.e1 = v.s3;
  r.du_de = __hiloint2double(__float_as_int(v.s5), __float_as_int(v.s4));
  r.de_du = __hiloint2double(__float_as_int(v.s7), __float_as_int(v.s6));
  return r;
}
__device__ __forceinline__ Rec ld_rec(void const *__restrict__ rec, unsigned idx) { return ldg<Rec>(rec, idx); }
__device__ __forceinline__ Rec ld_rec_u(void const *__restrict__ rec, unsigned idx) {     // ... once per wavefront where all lanes agree
  unsigned i0;
  if (all_lanes_at(idx, i0)) return rec_from(ld_scalar<f32x8>(rec, i0 * 32u));
  return ld_rec(rec, idx);
}
__device__ __forceinline__ void ld_rec2(void const *__restrict__ rec, unsigned ia, unsigned ib, Rec &a, Rec &b) {
  unsigned fa, fb;
  bool const ua = all_lanes_at(ia, fa), ub = all_lanes_at(ib, fb);
  if (ua & ub) {
    f32x8 const x = ld_scalar<f32x8>(rec, fa * 32u), y = ld_scalar<f32x8>(rec, fb * 32u);
    a = rec_from(x); b = rec_from(y);
    return;
  }
  a = ld_rec(rec, ia);
  b = ld_rec(rec, ib);
}
// key of ENTRY idx (the lower end of record idx), as a double, for the far moves of a search
template <bool ON_EPS>
__device__ __forceinline__ double rec_key(void const *__restrict__ rec, unsigned idx) {
  return (double)*reinterpret_cast<float const *>(static_cast<char const *>(rec) + (size_t)(idx * 32u + (ON_EPS ? 4u : 0u)));
}
// seek_curve_keys on records: bracket i of curve [e0, e0 + n) with its record r and its keys ka, kb as doubles (formed
// here unless the caller HAS them).  Same probes, same bracket.
template <bool ON_EPS, bool HAVE, bool SCALAR = true>      // SCALAR: try the scalar cache for the neighbouring record (lanes = neighbouring rays)
__device__ __forceinline__ void seek_rec(void const *__restrict__ rec, unsigned e0, int n, double x, int &i, Rec &r, double &ka, double &kb) {
  if (!HAVE) { ka = cvt_keep(ON_EPS ? r.e0 : r.u0); kb = cvt_keep(ON_EPS ? r.e1 : r.u1); }
  // (two comparisons on the way that nearly every call takes; written as `up = x >= kb, down = x < ka` the compiler adds
  // the NaN-safe complement of `up` for the branch below to them)
  if ((x >= ka) & (x < kb)) return;
  if (x >= kb) {
    if (i >= n - 2) return;
    Rec const c = SCALAR ? ld_rec_u(rec, e0 + i + 1) : ld_rec(rec, e0 + i + 1);           // bracket [i+1, i+2]
    double const kc = cvt_keep(ON_EPS ? c.e1 : c.u1);
    if (i + 2 >= n - 1 || kc > x) { ++i; r = c; ka = kb; kb = kc; return; }
    int lo = i + 2, hi, step = 2;
    for (;;) {
      hi = lo + step;
      if (hi >= n - 1) { hi = n - 1; break; }
      if (rec_key<ON_EPS>(rec, e0 + hi) > x) break;
      lo = hi;
      step <<= 1;
    }
    while (hi > lo + 1) {
      int const mid = (lo + hi) >> 1;
      if (rec_key<ON_EPS>(rec, e0 + mid) > x) hi = mid; else lo = mid;
    }
    i = lo;
    r = ld_rec(rec, e0 + i);                              // (each far-move branch ends in its own reload)
    ka = cvt_keep(ON_EPS ? r.e0 : r.u0); kb = cvt_keep(ON_EPS ? r.e1 : r.u1);
  } else {
    if (i <= 0) return;
    Rec const c = SCALAR ? ld_rec_u(rec, e0 + i - 1) : ld_rec(rec, e0 + i - 1);           // bracket [i-1, i]
    double const kc = cvt_keep(ON_EPS ? c.e0 : c.u0);
    if (i - 1 <= 0 || kc <= x) { --i; r = c; kb = ka; ka = kc; return; }
    int hi = i - 1, lo, step = 2;
    for (;;) {
      lo = hi - step;
      if (lo <= 0) { lo = 0; break; }
      if (rec_key<ON_EPS>(rec, e0 + lo) <= x) break;
      hi = lo;
      step <<= 1;
    }
    while (hi > lo + 1) {
      int const mid = (lo + hi) >> 1;
      if (rec_key<ON_EPS>(rec, e0 + mid) > x) hi = mid; else lo = mid;
    }
    i = lo;
    r = ld_rec(rec, e0 + i);
    ka = cvt_keep(ON_EPS ? r.e0 : r.u0); kb = cvt_keep(ON_EPS ? r.e1 : r.u1);
  }
}

// Table descriptors of ONE (gas, channel) pair as the look-up sees them: read from global memory, or
// from a copy the workgroup has staged in LDS (every lane of the workgroup works on the same pair; LDS
// reads are counted by lgkmcnt and stay out of the queue of the curve gathers).
extern __shared__ __attribute__((aligned(16))) unsigned char jur_lds[];

template <bool LDS>
struct PairDesc {
  void const *lvb, *cvb;       // global arrays
  void const *ueb;             // first (u, eps) entry of the pair
  void const *slb;             // ... and its bracket slopes (strict tables)
  unsigned l0;                 // first level of the pair
  unsigned kbase;              // first curve of the pair (LDS copy starts there)
  __device__ __forceinline__ Lvl lvl(int i) const {
    if constexpr (LDS) return reinterpret_cast<Lvl const *>(jur_lds)[i];
    else return ldg<Lvl>(lvb, l0 + (unsigned)i);
  }
  __device__ __forceinline__ Crv crv(unsigned k) const {
    if constexpr (LDS) return reinterpret_cast<Crv const *>(jur_lds + JUR_TBLNP * sizeof(Lvl))[k - kbase];
    else return ldg<Crv>(cvb, k);
  }
  // 1/(p[i+1] - p[i]) and 1/(T[k+1] - T[k]): formed by the staging loop (LDS), or here with the same division --
  // the same doubles either way (the fused kernel reads its descriptors through L1 and has no staged copy)
  unsigned rp_off, rt_off;     // byte offsets of the two reciprocal arrays in the LDS block
  void const *recb = nullptr;  // bracket records of the pair (set by the kernels that use them)
  __device__ __forceinline__ double rp(int i) const {
    if constexpr (LDS) return reinterpret_cast<double const *>(jur_lds + rp_off)[i];
    else return 1. / (lvl(i + 1).p - lvl(i).p);
  }
  __device__ __forceinline__ double rt(unsigned k) const {
    if constexpr (LDS) return reinterpret_cast<double const *>(jur_lds + rt_off)[k - kbase];
    else return 1. / (crv(k + 1).t - crv(k).t);
  }
};

// EXACT look-up: the reference's bisections probe for probe (locate_id jr_common.h:106-114, locate_tbl_id
// :116-125), for tables whose axes or curves are not sorted.
template <bool LDS>
__device__ __forceinline__ double ega_eps_exact(jur_view_t const &v, jur_int2 const pr, PairDesc<LDS> const &D, double tau, double t,
                                                double u, double p) {
  if (tau < 1e-9) return 0.;
  if (pr.a < 2) return 1.;
  void const *const ueb = D.ueb;
  int ilo = 0, ihi = pr.a - 1;
  while (ihi > ilo + 1) {  // ascending-only bisection, whatever the axis looks like
    int const i = (ihi + ilo) >> 1;
    if (D.lvl(i).p > p) ihi = i; else ilo = i;
  }
  Lvl const l0 = D.lvl(ilo), l1 = D.lvl(ilo + 1);
  if (l0.nt < 2 || l1.nt < 2) return 1.;
  unsigned const k0 = (unsigned)l0.c0, k1 = (unsigned)l1.c0;
  ilo = 0; ihi = l0.nt - 1;
  while (ihi > ilo + 1) {
    int const i = (ihi + ilo) >> 1;
    if (D.crv(k0 + i).t > t) ihi = i; else ilo = i;
  }
  Crv const c00 = D.crv(k0 + ilo), c01_ = D.crv(k0 + ilo + 1);
  if (c00.nu < 2 || c01_.nu < 2) return 1.;
  ilo = 0; ihi = l1.nt - 1;
  while (ihi > ilo + 1) {
    int const i = (ihi + ilo) >> 1;
    if (D.crv(k1 + i).t > t) ihi = i; else ilo = i;
  }
  Crv const c10 = D.crv(k1 + ilo), c11 = D.crv(k1 + ilo + 1);
  if (c10.nu < 2 || c11.nu < 2) return 1.;
  // the four (p,T) corners: u at which the curve reaches eps (get_u, jr_common.h:179-185), then the
  // curve's emissivity at that u plus the segment's column (get_eps, :156-177)
  double const eps = 1 - tau;
  unsigned const e0[4] = {(unsigned)c00.e0, (unsigned)c01_.e0, (unsigned)c10.e0, (unsigned)c11.e0};
  int const n[4] = {c00.nu, c01_.nu, c10.nu, c11.nu};
  double ec[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    Ue a, b;
    ld_pair(ueb, e0[k] + bisect_curve<true>(ueb, e0[k], n[k], eps), a, b);
    double const x = lip((double)a.eps, (double)a.u, (double)b.eps, (double)b.u, eps) + u;
    ld_pair(ueb, e0[k] + bisect_curve<false>(ueb, e0[k], n[k], x), a, b);
    ec[k] = c01(lip((double)a.u, (double)a.eps, (double)b.u, (double)b.eps, x));
  }
  double const eps_p0 = c01(lip(c00.t, ec[0], c01_.t, ec[1], t));
  double const eps_p1 = c01(lip(c10.t, ec[2], c11.t, ec[3], t));
  double const eps_t = c01(lip(l0.p, eps_p0, l1.p, eps_p1, p));
  return (1. - eps_t) / tau;
}

// WARM look-up (sorted tables: every bracket is unique, so any search finds the reference's bracket).
// Search state carried from segment to segment in three packed registers:
//   br = ipr | it0 << 8 | it1 << 16,  ia = idx00 | idx01 << 16,  ib = idx10 | idx11 << 16
// The curve positions kept are those where get_eps ended: the next segment's path emissivity is this
// segment's result, i.e. close to eps(x) on every curve, so its get_u search usually starts inside its
// bracket (-10 % kernel time against resuming from get_u's position).
// The two pressure levels are handled one after the other by a rolled loop, two curves (the temperature
// bracket of the level) at a time: half the curve state is live, 70 VGPRs (62 with RCPB: 8 waves per SIMD), 7 waves per SIMD (-5 %
// against all four curves side by side at 88 VGPRs).
// RCPB (LDS copy present; p and T axes and, as stored in fp32, every curve strictly increasing: no bracket
// has zero width): the three blends divide by multiplying with the reciprocal bracket widths staged in
// LDS (div_rcp), the other nine divisions use div_finite.
// RCPB is the strict-table arithmetic described at lip_slope; with PATH the function returns the NEW path
// transmittance (1 - eps_t, or tau where the reference's look-up answers 1) instead of the segment's
// transmittance (1 - eps_t) / tau -- what the kernels carry and write; the known-answer hook asks for the quotient.
template <bool LDS, bool RCPB, bool PATH = false, bool REC = false>
__device__ __forceinline__ double ega_eps_warm(jur_view_t const &v, jur_int2 const pr, PairDesc<LDS> const &D, double tau, double t,
                                               double u, double p, unsigned &br, unsigned &ia, unsigned &ib) {
  static_assert(RCPB || !PATH, "only the strict-table arithmetic carries the path transmittance itself");
  static_assert(RCPB || !REC, "bracket records hold the slopes of the strict-table arithmetic");
  double const one = PATH ? tau : 1.;          // the look-up's "no change" answer
  if (tau < 1e-9) return 0.;
  if (pr.a < 2) return one;
  void const *const ueb = D.ueb;
  int ipr = min((int)(br & 0xffu), pr.a - 2);
  Lvl l0 = D.lvl(ipr), l1 = D.lvl(ipr + 1);
  if ((p < l0.p) | (p >= l1.p)) {
    while (p < l0.p && ipr > 0) { --ipr; l1 = l0; l0 = D.lvl(ipr); }
    while (p >= l1.p && ipr < pr.a - 2) { ++ipr; l0 = l1; l1 = D.lvl(ipr + 1); }
  }
  br = (br & ~0xffu) | (unsigned)ipr;
  if (l0.nt < 2 || l1.nt < 2) return one;
  unsigned const k0 = (unsigned)l0.c0, k1 = (unsigned)l1.c0;
  int it0 = min((int)((br >> 8) & 0xffu), l0.nt - 2), it1 = min((int)((br >> 16) & 0xffu), l1.nt - 2);
  {
    Crv c00 = D.crv(k0 + it0), c01_ = D.crv(k0 + it0 + 1), c10 = D.crv(k1 + it1), c11 = D.crv(k1 + it1 + 1);
    if ((t < c00.t) | (t >= c01_.t) | (t < c10.t) | (t >= c11.t)) {
      while (t < c00.t && it0 > 0) { --it0; c01_ = c00; c00 = D.crv(k0 + it0); }
      while (t >= c01_.t && it0 < l0.nt - 2) { ++it0; c00 = c01_; c01_ = D.crv(k0 + it0 + 1); }
      while (t < c10.t && it1 > 0) { --it1; c11 = c10; c10 = D.crv(k1 + it1); }
      while (t >= c11.t && it1 < l1.nt - 2) { ++it1; c10 = c11; c11 = D.crv(k1 + it1 + 1); }
    }
    br = (unsigned)ipr | ((unsigned)it0 << 8) | ((unsigned)it1 << 16);
    if (c00.nu < 2 || c01_.nu < 2 || c10.nu < 2 || c11.nu < 2) return one;
  }
  // RCPB clamps with min/max, which would turn a NaN into 0: a NaN among the inputs (the tables hold none) is
  // answered here with what the comparisons of c01 would have handed through
  if (RCPB && (tau != tau || t != t || u != u || p != p)) return __builtin_nan("");
  double const eps = 1 - tau;
  double eps_p0 = 0, eps_p1 = 0;
  // (the records of BOTH levels requested before any is used: 92 VGPRs -- 40.9 ms at 5 wavefronts per SIMD, 101 ms with the
  // excess in scratch at 6, against 35.0: profiles/r04_ega_bracket_records_experiment.json)
#pragma unroll 1
  for (int h = 0; h < 2; h++) {  // pressure level l0, then l1
    unsigned const kc = h ? k1 + (unsigned)it1 : k0 + (unsigned)it0;
    Crv const ca = D.crv(kc), cb = D.crv(kc + 1);
    unsigned const packed = h ? ib : ia;
    unsigned const e0[2] = {(unsigned)ca.e0, (unsigned)cb.e0};
    int const n[2] = {ca.nu, cb.nu};
    int i[2] = {(int)(packed & 0xffffu), (int)(packed >> 16)};
    Ue a[2], b[2];
#pragma unroll
    for (int k = 0; k < 2; k++) i[k] = min(i[k], n[k] - 2);
    double x[2], ec[2];
    if constexpr (REC) {   // one record per bracket: keys and both slopes in one fetch
      Rec r[2];
      ld_rec2(D.recb, e0[0] + i[0], e0[1] + i[1], r[0], r[1]);      // both fetches in flight, then one curve after the other
      // (carrying the four table values of a record as doubles, converted where the record is fetched, costs four more
      // registers than the floats with the two keys the searches hold: 65 VGPRs, the eighth wavefront gone, 36.5 against
      // 35.0 ms -- with the shorter chain of fetches the eighth wavefront matters again)
#pragma unroll
      for (int k = 0; k < 2; k++) {
        double ka, kb;
        seek_rec<true, false>(D.recb, e0[k], n[k], eps, i[k], r[k], ka, kb);
        double const ya = cvt_keep(r[k].u0);
        x[k] = lip_slope(ka, ya, r[k].du_de, eps) + u;
        ka = ya; kb = cvt_keep(r[k].u1);
        seek_rec<false, true>(D.recb, e0[k], n[k], x[k], i[k], r[k], ka, kb);
        ec[k] = c01_num(lip_slope(ka, (double)r[k].e0, r[k].de_du, x[k]));
      }
    } else
    if constexpr (RCPB) ld_pair2(ueb, e0[0] + i[0], e0[1] + i[1], a[0], b[0], a[1], b[1]);
    else {
#pragma unroll
      for (int k = 0; k < 2; k++) ld_pair(ueb, e0[k] + i[k], a[k], b[k]);
    }
    // get_u (jr_common.h:179-185): u at which the curve reaches eps; get_eps (:156-177): the curve's
    // emissivity at that u plus the segment's column -- the column only grows, so the second search
    // starts where the first ended
    if constexpr (REC) {
    } else if constexpr (RCPB) {  // the keys of each bracket as doubles from the search that tests them to the interpolation
      double ka[2], kb[2];
#pragma unroll
      for (int k = 0; k < 2; k++) seek_curve_keys<true, false>(ueb, e0[k], n[k], eps, i[k], a[k], b[k], ka[k], kb[k]);
      double s[2];
      ld_slope2<0>(D.slb, e0[0] + i[0], e0[1] + i[1], s[0], s[1]);
#pragma unroll
      for (int k = 0; k < 2; k++) {
        double const ya = kkey<false>(a[k]), yb = kkey<false>(b[k]);    // ... which serve get_eps's search as its keys
        x[k] = lip_slope(ka[k], ya, s[k], eps) + u;
        ka[k] = ya; kb[k] = yb;
      }
#pragma unroll
      for (int k = 0; k < 2; k++) seek_curve_keys<false, true>(ueb, e0[k], n[k], x[k], i[k], a[k], b[k], ka[k], kb[k]);
      ld_slope2<1>(D.slb, e0[0] + i[0], e0[1] + i[1], s[0], s[1]);
#pragma unroll
      for (int k = 0; k < 2; k++) ec[k] = c01_num(lip_slope(ka[k], (double)a[k].eps, s[k], x[k]));
    } else {
#pragma unroll
      for (int k = 0; k < 2; k++) seek_curve<true>(ueb, e0[k], n[k], eps, i[k], a[k], b[k]);
#pragma unroll
      for (int k = 0; k < 2; k++) x[k] = lip((double)a[k].eps, (double)a[k].u, (double)b[k].eps, (double)b[k].u, eps) + u;
#pragma unroll
      for (int k = 0; k < 2; k++) seek_curve<false>(ueb, e0[k], n[k], x[k], i[k], a[k], b[k]);
#pragma unroll
      for (int k = 0; k < 2; k++) ec[k] = c01(lip((double)a[k].u, (double)a[k].eps, (double)b[k].u, (double)b[k].eps, x[k]));
    }
    unsigned const last = (unsigned)i[0] | ((unsigned)i[1] << 16);
    if (h) ib = last; else ia = last;
    double e;
    if constexpr (RCPB) e = c01_num(lip_mulr(ca.t, ec[0], ec[1], t, D.rt(kc)));
    else e = c01(lip(ca.t, ec[0], cb.t, ec[1], t));
    if (h) eps_p1 = e; else eps_p0 = e;
  }
  if constexpr (RCPB) {
    // the two level pressures are read from LDS again rather than kept through both levels: 4 VGPRs, which is what
    // kept the kernel at 7 waves per SIMD with the keys held as doubles (round 2; 8 waves at 62 VGPRs since the slopes)
    int q = ipr;
    asm volatile("" : "+v"(q));
    double const tau_new = 1. - c01_num(lip_mulr(D.lvl(q).p, eps_p0, eps_p1, p, D.rp(q)));
    if constexpr (PATH) return tau_new;
    else return div_finite(tau_new, tau);
  }
  return (1. - c01(lip(l0.p, eps_p0, l1.p, eps_p1, p))) / tau;
}

// The same look-up by the four lanes of a quad, one (p, T) corner curve each (fused kernel: the chain's latency is
// what counts there, and the four curve searches + interpolations are its longest part).  All four lanes hold the
// chain's inputs and run the bracket searches redundantly; lane q takes curve q = 2 * level + temperature side,
// the four curve emissivities are exchanged with DPP moves and every lane forms the blends -- the operations of
// ega_eps_warm<false, false> in the same order, hence the same doubles.  State: br as there, ix = this lane's
// position in its own curve.
// FAST: the strict-table arithmetic of ega_eps_warm<.., true, true> (lip_slope, lip_mulr with the reciprocal widths
// formed here by division -- the same doubles as the staged ones), returning the new path transmittance: the same
// doubles as the batched kernel.  Otherwise the reference's arithmetic, returning the segment's transmittance.
template <bool FAST>
__device__ __forceinline__ double ega_eps_warm_quad(jur_view_t const &v, jur_int2 const pr, PairDesc<false> const &D, double tau,
                                                    double t, double u, double p, unsigned &br, unsigned &ix) {
  double const one = FAST ? tau : 1.;
  if (tau < 1e-9) return 0.;
  if (pr.a < 2) return one;
  void const *const ueb = D.ueb;
  int ipr = min((int)(br & 0xffu), pr.a - 2);
  Lvl l0 = D.lvl(ipr), l1 = D.lvl(ipr + 1);
  if ((p < l0.p) | (p >= l1.p)) {
    while (p < l0.p && ipr > 0) { --ipr; l1 = l0; l0 = D.lvl(ipr); }
    while (p >= l1.p && ipr < pr.a - 2) { ++ipr; l0 = l1; l1 = D.lvl(ipr + 1); }
  }
  br = (br & ~0xffu) | (unsigned)ipr;
  if (l0.nt < 2 || l1.nt < 2) return one;
  unsigned const k0 = (unsigned)l0.c0, k1 = (unsigned)l1.c0;
  int it0 = min((int)((br >> 8) & 0xffu), l0.nt - 2), it1 = min((int)((br >> 16) & 0xffu), l1.nt - 2);
  Crv c00 = D.crv(k0 + it0), c01_ = D.crv(k0 + it0 + 1), c10 = D.crv(k1 + it1), c11 = D.crv(k1 + it1 + 1);
  if ((t < c00.t) | (t >= c01_.t) | (t < c10.t) | (t >= c11.t)) {
    while (t < c00.t && it0 > 0) { --it0; c01_ = c00; c00 = D.crv(k0 + it0); }
    while (t >= c01_.t && it0 < l0.nt - 2) { ++it0; c00 = c01_; c01_ = D.crv(k0 + it0 + 1); }
    while (t < c10.t && it1 > 0) { --it1; c11 = c10; c10 = D.crv(k1 + it1); }
    while (t >= c11.t && it1 < l1.nt - 2) { ++it1; c10 = c11; c11 = D.crv(k1 + it1 + 1); }
  }
  br = (unsigned)ipr | ((unsigned)it0 << 8) | ((unsigned)it1 << 16);
  if (c00.nu < 2 || c01_.nu < 2 || c10.nu < 2 || c11.nu < 2) return one;
  int const q = threadIdx.x & 3;
  Crv const mine = (q == 0) ? c00 : (q == 1) ? c01_ : (q == 2) ? c10 : c11;
  unsigned const e0 = (unsigned)mine.e0;
  int const n = mine.nu;
  int i = min((int)ix, n - 2);
  Ue a, b;
  if (!(FAST && D.recb)) ld_pair(ueb, e0 + i, a, b);
  double const eps = 1 - tau;
  if constexpr (FAST) {
    bool const nan_in = (tau != tau || t != t || u != u || p != p);   // as ega_eps_warm: min/max clamps below
    if (D.recb) {   // bracket records (round 4): one fetch for the curve's keys and both slopes -- the same doubles
      Rec r = ld_rec(D.recb, e0 + i);
      double ka, kb;
      seek_rec<true, false, false>(D.recb, e0, n, eps, i, r, ka, kb);
      double const ya = (double)r.u0;
      double const x = lip_slope(ka, ya, r.du_de, eps) + u;
      ka = ya; kb = (double)r.u1;
      seek_rec<false, true, false>(D.recb, e0, n, x, i, r, ka, kb);
      double const ec = c01_num(lip_slope(ka, (double)r.e0, r.de_du, x));
      ix = (unsigned)i;
      double const ec0 = quad_bcast<0>(ec), ec1 = quad_bcast<1>(ec), ec2 = quad_bcast<2>(ec), ec3 = quad_bcast<3>(ec);
      double const eps_p0 = c01_num(lip_mulr(c00.t, ec0, ec1, t, 1. / (c01_.t - c00.t)));
      double const eps_p1 = c01_num(lip_mulr(c10.t, ec2, ec3, t, 1. / (c11.t - c10.t)));
      double const tau_new = 1. - c01_num(lip_mulr(l0.p, eps_p0, eps_p1, p, 1. / (l1.p - l0.p)));
      return nan_in ? __builtin_nan("") : tau_new;
    }
    seek_curve<true>(ueb, e0, n, eps, i, a, b);
    double const x = lip_slope((double)a.eps, (double)a.u, ld_slope<0>(D.slb, e0 + i), eps) + u;
    seek_curve<false>(ueb, e0, n, x, i, a, b);
    double const ec = c01_num(lip_slope((double)a.u, (double)a.eps, ld_slope<1>(D.slb, e0 + i), x));
    ix = (unsigned)i;
    double const ec0 = quad_bcast<0>(ec), ec1 = quad_bcast<1>(ec), ec2 = quad_bcast<2>(ec), ec3 = quad_bcast<3>(ec);
    double const eps_p0 = c01_num(lip_mulr(c00.t, ec0, ec1, t, 1. / (c01_.t - c00.t)));
    double const eps_p1 = c01_num(lip_mulr(c10.t, ec2, ec3, t, 1. / (c11.t - c10.t)));
    double const tau_new = 1. - c01_num(lip_mulr(l0.p, eps_p0, eps_p1, p, 1. / (l1.p - l0.p)));
    return nan_in ? __builtin_nan("") : tau_new;
  } else {
    seek_curve<true>(ueb, e0, n, eps, i, a, b);
    double const x = lip((double)a.eps, (double)a.u, (double)b.eps, (double)b.u, eps) + u;
    seek_curve<false>(ueb, e0, n, x, i, a, b);
    double const ec = c01(lip((double)a.u, (double)a.eps, (double)b.u, (double)b.eps, x));
    ix = (unsigned)i;
    double const ec0 = quad_bcast<0>(ec), ec1 = quad_bcast<1>(ec), ec2 = quad_bcast<2>(ec), ec3 = quad_bcast<3>(ec);
    double const eps_p0 = c01(lip(c00.t, ec0, c01_.t, ec1, t));
    double const eps_p1 = c01(lip(c10.t, ec2, c11.t, ec3, t));
    return (1. - c01(lip(l0.p, eps_p0, l1.p, eps_p1, p))) / tau;
  }
}

// ---------------------------------------------------------------------------------------
// continua; the channel-only factors come precomputed in jur_chan_t
// ---------------------------------------------------------------------------------------
// a / C for a compile-time constant C: the quotient through r = RN(1/C) (div_rcp), folded by the compiler
struct CO2_DEN { static constexpr double v = JUR_AVOGADRO * 1000 * JUR_P0; };
struct P0_DEN { static constexpr double v = JUR_P0; };
struct T36_DEN { static constexpr double v = 296. - 260.; };
struct TR_DEN { static constexpr double v = 296.; };
template <class C>
__device__ __forceinline__ double div_const(double a) { return div_rcp(a, C::v, 1. / C::v); }

__device__ __forceinline__ double ctm_co2(jur_chan_t const &ch, double p, double t, double u) {
  double const dt230 = t - 230;
  double const dt260 = t - 260;
  double const dt296 = t - 296;
  double const ctw = dt260 * 5.050505e-4 * dt296 * ch.co2_cw230 - dt230 * 9.259259e-4 * dt296 * ch.co2_cw260
                   + dt230 * 4.208754e-4 * dt260 * ch.co2_cw296;
  return div_const<CO2_DEN>(u * p * ctw);
}

// ratio^y for a per-channel constant ratio: exp(y ln ratio) with ln ratio = hi + lo prepared on the host
// (64-bit logarithm) and the product y ln ratio carried with its rounding error -- about 1 ulp like the
// library's pow, which spends most of its ~180 instructions on that logarithm.
template <class Tab>
__device__ __forceinline__ double pow_const(Tab const &tab, double lnr_hi, double lnr_lo, double y) {
  double const ph = y * lnr_hi;
  double const pl = __builtin_fma(y, lnr_hi, -ph) + y * lnr_lo;
  double const e = exp_tab(tab, ph);
  return __builtin_fma(e, pl, e);
}

// 1 / t for the continua of one segment: 296/T, 0.7193876/T, 273/T and 1/T (jr_common.h:352-355, 374, 388) share
// it as multiplications.  rcp + two Newton steps: within an ulp of the correctly rounded reciprocal, so every product
// below is within ~2 ulp of the reference's quotient -- 1e-16 relative where the contract is 1e-6 -- for 5
// instructions instead of 3 .. 4 full divisions (~45).  (Round 2 measured the bit-identical form of this sharing
// slower because of two spilled registers; without the correctly-rounded residual steps it is not.)
__device__ __forceinline__ double rcp_t(double t) {
  double r = __builtin_amdgcn_rcp(t);
  r = __builtin_fma(r, __builtin_fma(-t, r, 1.0), r);
  return __builtin_fma(r, __builtin_fma(-t, r, 1.0), r);
}

// tanh(x) for the x = 0.7193876 nu / T > 0 of the H2O continuum (jr_common.h:354): (1 - e) / (1 + e) with
// e = exp(-2x) -- one exp and one division instead of the library's tanh (which branches on the size of x, costs about
// twice as much and keeps a second set of polynomial coefficients in registers across the segment loop).  1 - e loses
// log2(1 / 2x) bits to cancellation: relative error ~1e-16 / 2x, i.e. <= 2e-15 for nu >= 100 cm^-1 at any T <= 1000 K
// and 2e-14 for a channel at 1 cm^-1 -- eight orders inside the contract.
template <class Tab>
__device__ __forceinline__ double tanh_pos(Tab const &tab, double x) {
  double const e = exp_tab(tab, -2. * x);
  return div_finite(1. - e, 1. + e);
}

template <class Tab>
__device__ __forceinline__ double ctm_h2o(Tab const &tab, jur_chan_t const &ch, double p, double t, double rt, double q, double u) {
  double const y = div_const<T36_DEN>(296. - t);
  double const ctwslf = ch.h2o_sc * pow_const(tab, ch.h2o_lnr_hi, ch.h2o_lnr_lo, y);
  double const x = .7193876 * rt * ch.nu;
  double const a1 = ch.nu * u * tanh_pos(tab, x);
  double const a2 = 296. * rt;
  double const a3 = div_const<P0_DEN>(p) * (q * ctwslf + (1 - q) * ch.h2o_ctwfrn) * 1e-20;
  return a1 * a2 * a3;
}

template <class Tab>
__device__ __forceinline__ double ctm_n2(Tab const &tab, jur_chan_t const &ch, double p, double t, double rt) {
  double const q_n2 = 0.79, t0 = 273, tr = 296;
  double const pr = div_const<P0_DEN>(p), s = t0 * rt;
  return 0.1 * pr * pr * s * s * exp_tab(tab, ch.n2_beta * (1 / tr - rt)) * q_n2 * ch.n2_b
         * (q_n2 + (1 - q_n2) * (1.294 - div_const<TR_DEN>(0.4545 * t)));
}

template <class Tab>
__device__ __forceinline__ double ctm_o2(Tab const &tab, jur_chan_t const &ch, double p, double t, double rt) {
  double const q_o2 = 0.21, t0 = 273, tr = 296;
  double const pr = div_const<P0_DEN>(p), s = t0 * rt;
  return 0.1 * pr * pr * s * s * exp_tab(tab, ch.o2_beta * (1 / tr - rt)) * q_o2 * ch.o2_b;
}

__device__ __forceinline__ double planck_src(double const *__restrict__ sr, double t) {
  // 0.25 K grid from 100 K (locate_st, jr_common.h:82-84).  Upstream has no range check and reads
  // outside the table for T outside [100, 400) K; here the index is clamped (end intervals extrapolate).
  int const it = min(max((int)(4 * t) - 400, 0), TBLNS - 2);
  // grid points 100 + it * 300 / 1200 are multiples of 0.25: exact doubles, and the bracket width
  // st1 - st0 is exactly 0.25, so the interpolation's division is the exact multiplication by 4
  double const st0 = 100 + 0.25 * (double)it;
  return sr[it] + (t - st0) * (sr[it + 1] - sr[it]) * 4.0;
}

// Transmittance of all gases over ONE segment (apply_ega_core's product of the per-gas segment transmittances,
// jr_common.h:270-280) from the products of the gases' PATH transmittances after this segment (pcur) and after
// the previous one (pprev): one division per (channel, segment) instead of one per gas.  A gas whose path has
// gone opaque (path transmittance 0: jr_common.h:239 answers 0 from there on) makes pcur, and from the next segment
// pprev, 0: the reference's product is 0 there too.  (A pprev below 1e-280 -- twenty gases each within a hair of
// opaque -- is treated like 0: the ray's own transmittance is below it, nothing it could still add is representable.)
__device__ __forceinline__ double segment_tau_gas(double pcur, double pprev) {
  return (pprev > 1e-280) ? div_finite(pcur, pprev) : 0.;
}

// radiance update of one segment (new_obs_core, jr_common.h:293-300)
template <class Tab>
__device__ __forceinline__ void new_obs_step(Tab const &tab, double tau_gas, double beta_ds, double src, double &rad, double &tau) {
  if (tau_gas > 1e-50) {
    double const eps = 1. - tau_gas * exp_tab(tab, -beta_ds);
    rad += src * eps * tau;
    tau *= (1. - eps);
  }
}

// after the last segment: surface emission if the ray hit the ground (add_surface_core, jr_common.h:227-234),
// brightness temperature if asked for (brightness_core, :187-190)
__device__ __forceinline__ void ray_epilogue(double const *__restrict__ sr, double nu, double tsurf, int write_bbt, double &rad,
                                             double const tau) {
  if (tsurf > 0.) rad += planck_src(sr, tsurf) * tau;
  if (write_bbt) rad = JUR_C2 * nu / log1p((JUR_C1 * nu * nu * nu) / rad);
}

// ---------------------------------------------------------------------------------------
// Workgroup -> (ray block, item) for the kernels whose workgroups of one ray block share its LOS rows: the hardware
// deals workgroups to the 8 XCDs round robin (b % 8), each XCD has its own L2, so the `nitem` workgroups of a ray block
// follow each other on ONE XCD: b -> xcd = b % 8, s = b / 8, ray block = (s / nitem) * 8 + xcd, item = s % nitem --
// eight ray blocks in flight, one per XCD.  The last, incomplete group of k < 8 ray blocks would leave 8 - k XCDs idle for
// as long as a whole group takes (a launch of 49 ray blocks x 7134 pairs: 12 % of its time): there the XCDs x with
// x % k == j share ray block j and split its items between them.  Placement only; every (ray block, item) runs once.
// ---------------------------------------------------------------------------------------
struct BlockItem { int rb, item; };
__host__ __device__ inline unsigned xcd_grid(int nrb, int nitem) {
  int const full = nrb >> 3, k = nrb & 7;
  return (unsigned)(full * 8 * nitem + (k ? 8 * ((nitem + (8 / k) - 1) / (8 / k)) : 0));
}
__device__ __forceinline__ BlockItem xcd_block_item(int b, int nrb, int nitem) {      // rb < 0: nothing to do
  int const full = nrb >> 3, k = nrb & 7, nfull = full * 8 * nitem;
  if (b < nfull) {
    int const xcd = b & 7, s = b >> 3;
    return {(s / nitem) * 8 + xcd, s - (s / nitem) * nitem};
  }
  if (k == 0) return {-1, 0};
  int const bb = b - nfull, xcd = bb & 7, s = bb >> 3;
  int const j = xcd % k, cj = (8 - j + k - 1) / k, q = xcd / k;          // XCDs j, j + k, j + 2k .. share ray block j
  int const item = s * cj + q;
  return {item < nitem ? full * 8 + j : -1, item};
}

// Stage the level and curve descriptors (16 B each) of one (gas, channel) pair in LDS, once per workgroup, with
// the reciprocal widths of the p and T brackets when the tables are strictly increasing (RCPB).  Ends in a barrier.
template <bool LDS, bool RCPB>
__device__ __forceinline__ void stage_pair(jur_view_t const &v, jur_int2 const pd, PairDesc<LDS> &D) {
  if (LDS) {
    Lvl const *const gl = reinterpret_cast<Lvl const *>(v.lvl) + pd.b;
    Lvl const first = gl[0], last = gl[pd.a - 1];
    D.kbase = (unsigned)first.c0;
    int const ncrv = last.c0 + last.nt - first.c0;
    Lvl *const sl = reinterpret_cast<Lvl *>(jur_lds);
    Crv *const sc = reinterpret_cast<Crv *>(jur_lds + JUR_TBLNP * sizeof(Lvl));
    Crv const *const gc = reinterpret_cast<Crv const *>(v.crv) + first.c0;
    for (int i = threadIdx.x; i < pd.a; i += blockDim.x) sl[i] = gl[i];
    for (int i = threadIdx.x; i < ncrv; i += blockDim.x) sc[i] = gc[i];
    if (RCPB) {  // reciprocal widths of the p and T brackets (entries that straddle two axes are never used)
      D.rp_off = (unsigned)((JUR_TBLNP + v.max_pair_curves) * 16);
      D.rt_off = D.rp_off + (unsigned)(JUR_TBLNP * 8);
      double *const rp = reinterpret_cast<double *>(jur_lds + D.rp_off), *const rt = reinterpret_cast<double *>(jur_lds + D.rt_off);
      for (int i = threadIdx.x; i + 1 < pd.a; i += blockDim.x) rp[i] = 1. / (gl[i + 1].p - gl[i].p);
      for (int i = threadIdx.x; i + 1 < ncrv; i += blockDim.x) rt[i] = 1. / (gc[i + 1].t - gc[i].t);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// along-path integration in two kernels.  The emissivity-growth recurrence of every (ray, channel,
// gas) triple is an independent sequential chain, so it gets its own lane -- ng x more lanes than
// one lane per (ray, channel), half the registers, 7 - 8 waves per SIMD to hide the dependent table
// loads -- and hands its path transmittance after every segment to the combine kernel through HBM
// (tiles of 64 ray slots, [tile][point][pair][64]).  (A fused single kernel with the gases unrolled in one lane
// was measured 1.7x slower: 166+ VGPRs, 2-3 waves per SIMD.)
//
// jur_ega_kernel: one lane per ray, one (channel, gas) pair per workgroup.  Workgroups that
// share a block of rays are made consecutive on one XCD (they re-read the same p, T, u lines
// from that XCD's L2): b -> xcd = b % 8, s = b / 8, ray block = (s / npair) * 8 + xcd,
// pair = s % npair.  Placement only affects speed.
// ---------------------------------------------------------------------------------------
#ifndef JUR_REC_WAVES
#define JUR_REC_WAVES 7
#endif
template <bool WARM, bool LDS, bool RCPB, bool REC = false>
__global__ __launch_bounds__(256, REC ? JUR_REC_WAVES : 5) void jur_ega_kernel(jur_view_t v, jur_chunk_t c, int nrb) {
  static_assert(WARM || !RCPB, "reciprocal widths need strictly increasing axes");
  int const npair = v.nd * v.ng;
  BlockItem const bi = xcd_block_item((int)blockIdx.x, nrb, npair);
  int const rb = bi.rb, pr = bi.item;                                       // ray block, pair: uniform
  if (rb < 0) return;
  int const d = pr / v.ng, g = pr - d * v.ng;
  int const r = rb * blockDim.x + threadIdx.x;
  int const pair_idx = g * v.nd + d;
  jur_int2 const pd = v.pair[pair_idx];
  if (pd.a < 2) return;                          // no table: transmittance 1, the combine kernel knows
  PairDesc<LDS> D{v.lvl, v.crv, v.ue + v.pair_e0[pair_idx], v.sl + v.pair_e0[pair_idx], (unsigned)pd.b, 0u, 0u, 0u};
  if constexpr (REC) D.recb = v.rec + v.pair_e0[pair_idx];
  stage_pair<LDS, RCPB>(v, pd, D);
  if (r >= c.n) return;
  // the workspaces are addressed as (wave-uniform pointer into this wavefront's tile) + lane offset
  int const nfield = JUR_F_K + v.nw + v.ng;
  int const tile = __builtin_amdgcn_readfirstlane(r >> 6);
  unsigned const lane = (unsigned)(r & 63);
  size_t const R = (size_t)nfield * 64;                              // doubles from one LOS point to the next
  double const *const los_tile = c.los + (size_t)tile * los_tile_doubles(nfield);
  double const *const los_p = los_tile + JUR_F_P * 64, *const los_t = los_tile + JUR_F_T * 64,
               *const los_u = los_tile + (size_t)(JUR_F_K + v.nw + g) * 64;
  size_t const Re = (size_t)npair * 64;                              // ... and from one point's transmittances to the next
  double *const out = c.eps + eps_tile_point0(c, tile) * Re + (size_t)pr * 64;
  int const np = c.np[r];
  double tau_path = 1.0;
  unsigned br = 0, ia = 0, ib = 0;
  // p and T of a segment -- what the look-up needs first, for its bracket tests -- are requested one segment ahead
  // (4 VGPRs; -1.5 % kernel time); u, needed only after the first curve search, is not: in round 2 it cost a wavefront
  // of occupancy (62.6 vs 60.9 ms), with the slopes it fits (64 VGPRs) and changes nothing (39.5 vs 39.8 ms).  The warm
  // variants without reciprocal widths sit at 70 VGPRs and keep loading at the point of use.
  constexpr bool AHEAD = RCPB || !WARM;
  double p_next = 0., t_next = 0.;
  if (AHEAD && np > 0) { p_next = ldg<double>(los_p, lane); t_next = ldg<double>(los_t, lane); }

  for (int ip = 0; ip < np; ++ip) {
    size_t const o = (size_t)ip * R;
    double p, t;
    if constexpr (AHEAD) {
      p = p_next; t = t_next;
      if (ip + 1 < np) { p_next = ldg<double>(los_p + o + R, lane); t_next = ldg<double>(los_t + o + R, lane); }
    } else { p = ldg<double>(los_p + o, lane); t = ldg<double>(los_t + o, lane); }
    double const u = ldg<double>(los_u + o, lane);
    // what is carried and written is the gas's transmittance of the path up to and including this segment
    if constexpr (RCPB) tau_path = ega_eps_warm<LDS, true, true, REC>(v, pd, D, tau_path, t, u, p, br, ia, ib);
    else if constexpr (WARM) tau_path *= ega_eps_warm<LDS, false>(v, pd, D, tau_path, t, u, p, br, ia, ib);
    else tau_path *= ega_eps_exact<LDS>(v, pd, D, tau_path, t, u, p);
    st_stream(out + (size_t)ip * Re, lane, tau_path);
  }
}

// ---------------------------------------------------------------------------------------
// jur_ega_group_kernel (round 4): one lane per (ray, gas), walking the channels of an ITEM -- up to JUR_EGA_NCH
// channels whose tables of this gas stand on the same (p, T) grid (jur_flat_group_items) -- inside the segment loop.
// What depends on the ray, the gas and the grid only is done once per segment and item instead of once per
// (channel, gas) pair: the three LOS loads (p, T, u[g]), the pressure bracket, the two temperature brackets
// (jr_common.h:241-246 -- the brackets do not depend on the curve), the NaN test of the inputs, the differences
// p - p0, T - T0 of the three blends.  Per channel there remain the four curve walks with their blends: the same
// operations on the same operands in the same order as ega_eps_warm<.., true, true>, hence the same doubles (the
// test suite compares the two kernels bit for bit).
// LDS per workgroup: the grid (levels {p, nt, first curve RELATIVE to the pair}, curve temperatures, reciprocal
// bracket widths), per channel the curves' {nu, first entry}, and the chain state {path transmittance, curve
// positions} of every (lane, channel): the channel loop is rolled, its state cannot live in registers.
// Strict tables only (the arithmetic of lip_slope / lip_mulr); everything else keeps jur_ega_kernel.
// ---------------------------------------------------------------------------------------

// LDS of a workgroup of `block` lanes: grid {levels, curve temperatures, reciprocal T widths, reciprocal p widths},
// first entries of the curves per channel (4 B each: a curve's length is the distance to the next one's start),
// chain state [channel][lane] as two 8-byte arrays
__host__ __device__ inline size_t ega_group_lds_bytes(int max_pair_curves, int nch, int block) {
  size_t const capC = (size_t)max_pair_curves + 4;
  return 16 * (size_t)JUR_TBLNP + 8 * capC * 2 + 8 * (size_t)JUR_TBLNP + ((4 * capC * nch + 15) & ~(size_t)15) + 16 * (size_t)block * nch;
}

template <int WAVES>
__global__ __launch_bounds__(1024, WAVES) void jur_ega_group_kernel(jur_view_t v, jur_chunk_t c, int nrb) {
  int const nitems = v.ega_nitems;
  int const BLOCK = (int)blockDim.x;
  BlockItem const bi = xcd_block_item((int)blockIdx.x, nrb, nitems);
  int const rb = bi.rb, it = bi.item;                                       // ray block, item: uniform
  if (rb < 0) return;
  jur_item_t const *const item = v.ega_items + it;
  int const g = item->g, nch = item->nch;
  bool const all_curves = item->flags & 1;         // every curve of the item's tables has >= 2 entries (uniform)
  int const capC = v.max_pair_curves + 4;
  Lvl *const sL = reinterpret_cast<Lvl *>(jur_lds);
  double *const sT = reinterpret_cast<double *>(jur_lds + 16 * JUR_TBLNP), *const sRT = sT + capC, *const sRP = sRT + capC;
  int *const sE0 = reinterpret_cast<int *>(sRP + JUR_TBLNP);
  // chain state, [channel][lane] with 8-byte elements: conflict-free 64-bit LDS accesses
  double *const sTau = reinterpret_cast<double *>(jur_lds + 16 * JUR_TBLNP + 16 * (size_t)capC + 8 * JUR_TBLNP +
                                                  ((4 * (size_t)capC * v.ega_nch + 15) & ~(size_t)15));
  unsigned *const sPos = reinterpret_cast<unsigned *>(sTau + (size_t)v.ega_nch * BLOCK);   // [level 0 / 1][channel][lane]
  jur_int2 const pd0 = v.pair[g * v.nd + item->d[0]];
  int const npl = pd0.a;                           // >= 2: pairs without a table are in no item
  {
    Lvl const *const gl = reinterpret_cast<Lvl const *>(v.lvl) + pd0.b;
    int const kb0 = gl[0].c0;
    Lvl const last = gl[npl - 1];
    int const ncrv = last.c0 + last.nt - kb0;
    Crv const *const gc = reinterpret_cast<Crv const *>(v.crv) + kb0;
    for (int i = threadIdx.x; i < npl; i += BLOCK) { Lvl l = gl[i]; l.c0 -= kb0; sL[i] = l; }
    for (int i = threadIdx.x; i < ncrv + 2; i += BLOCK) sT[i] = i < ncrv ? gc[i].t : 0.;
    // reciprocal widths exactly as stage_pair forms them (entries that straddle two axes are never used)
    for (int i = threadIdx.x; i + 1 < npl; i += BLOCK) sRP[i] = 1. / (gl[i + 1].p - gl[i].p);
    for (int i = threadIdx.x; i + 1 < ncrv; i += BLOCK) sRT[i] = 1. / (gc[i + 1].t - gc[i].t);
    for (int k = 0; k < nch; k++) {
      jur_int2 const pdk = v.pair[g * v.nd + item->d[k]];
      Crv const *const gck = reinterpret_cast<Crv const *>(v.crv) + reinterpret_cast<Lvl const *>(v.lvl)[pdk.b].c0;
      int const end = gck[ncrv - 1].e0 + gck[ncrv - 1].nu;            // the curves of a pair follow each other in ue
      for (int i = threadIdx.x; i < ncrv + 4; i += BLOCK) sE0[(size_t)k * capC + i] = i < ncrv ? gck[i].e0 : end;
    }
    __syncthreads();
  }
  int const r = rb * BLOCK + threadIdx.x;
  if (r >= c.n) return;
  int const nfield = JUR_F_K + v.nw + v.ng;
  int const tile = __builtin_amdgcn_readfirstlane(r >> 6);
  unsigned const lane = (unsigned)(r & 63);
  size_t const R = (size_t)nfield * 64;
  double const *const los_tile = c.los + (size_t)tile * los_tile_doubles(nfield);
  double const *const los_p = los_tile + JUR_F_P * 64, *const los_t = los_tile + JUR_F_T * 64,
               *const los_u = los_tile + (size_t)(JUR_F_K + v.nw + g) * 64;
  size_t const Re = (size_t)v.nd * v.ng * 64;
  double *const out_tile = c.eps + eps_tile_point0(c, tile) * Re;
  int const np = c.np[r];
  double *const mytau = sTau + threadIdx.x;
  unsigned *const mypos = sPos + threadIdx.x;
  int const HS = v.ega_nch * BLOCK;                 // from a chain's positions on level l0 to those on l1
  for (int k = 0; k < nch; k++) { mytau[k * BLOCK] = 1.0; mypos[k * BLOCK] = 0u; mypos[HS + k * BLOCK] = 0u; }
  // the item's fields inside the segment loop come through the scalar cache (constant address space: a plain load
  // behind the loop's stores is a vector load of a uniform value)
  unsigned const item_byte = (unsigned)it * (unsigned)sizeof(jur_item_t);
  unsigned br = 0;

  for (int ip = 0; ip < np; ++ip) {
    size_t const o = (size_t)ip * R;
    // (no request one segment ahead as in jur_ega_kernel: the row's latency is spread over the item's channels)
    double const p = ldg<double>(los_p + o, lane), t = ldg<double>(los_t + o, lane), u = ldg<double>(los_u + o, lane);
    // ---- once per segment: the brackets of the shared grid
    int ipr = min((int)(br & 0xffu), npl - 2);
    Lvl l0 = sL[ipr], l1 = sL[ipr + 1];
    if ((p < l0.p) | (p >= l1.p)) {
      while (p < l0.p && ipr > 0) { --ipr; l1 = l0; l0 = sL[ipr]; }
      while (p >= l1.p && ipr < npl - 2) { ++ipr; l0 = l1; l1 = sL[ipr + 1]; }
    }
    bool const valid = (l0.nt >= 2) & (l1.nt >= 2);                 // else the look-up answers "no change"
    int it0 = max(min((int)((br >> 8) & 0xffu), l0.nt - 2), 0), it1 = max(min((int)((br >> 16) & 0xffu), l1.nt - 2), 0);
    double T00 = sT[l0.c0 + it0], T01 = sT[l0.c0 + it0 + 1], T10 = sT[l1.c0 + it1], T11 = sT[l1.c0 + it1 + 1];
    if (valid && ((t < T00) | (t >= T01) | (t < T10) | (t >= T11))) {
      while (t < T00 && it0 > 0) { --it0; T01 = T00; T00 = sT[l0.c0 + it0]; }
      while (t >= T01 && it0 < l0.nt - 2) { ++it0; T00 = T01; T01 = sT[l0.c0 + it0 + 1]; }
      while (t < T10 && it1 > 0) { --it1; T11 = T10; T10 = sT[l1.c0 + it1]; }
      while (t >= T11 && it1 < l1.nt - 2) { ++it1; T10 = T11; T11 = sT[l1.c0 + it1 + 1]; }
    }
    br = (unsigned)ipr | ((unsigned)it0 << 8) | ((unsigned)it1 << 16);
    int const kc0 = l0.c0 + it0, kc1 = l1.c0 + it1;
    double const dp = p - l0.p, dt0 = t - T00, dt1 = t - T10;      // (x - x0) of the three blends
    bool const nan_in = (t != t) | (u != u) | (p != p);
    // ---- per channel: the four curve walks and the blends
#pragma unroll 1
    for (int k = 0; k < nch; k++) {
      double const tau = mytau[k * BLOCK];
      unsigned *const pos = mypos + k * BLOCK;       // [0]: the chain's positions in level l0's curve pair, [HS]: l1's
      unsigned const ku = (unsigned)__builtin_amdgcn_readfirstlane(k);
      double tau_new = tau;
      if (tau < 1e-9) tau_new = 0.;
      else if (valid) {
        int const *const ne = sE0 + (size_t)k * capC;
        bool curves_ok = true;
        if (!all_curves) {
          int const a0 = ne[kc0], a1 = ne[kc0 + 1], a2 = ne[kc0 + 2], b0 = ne[kc1], b1 = ne[kc1 + 1], b2 = ne[kc1 + 2];
          curves_ok = (a1 - a0 >= 2) & (a2 - a1 >= 2) & (b1 - b0 >= 2) & (b2 - b1 >= 2);
        }
        if (curves_ok) {
          if (nan_in || tau != tau) tau_new = __builtin_nan("");
          else {
            long long const pe0 = ld_scalar<long long>(v.ega_items, item_byte + (unsigned)offsetof(jur_item_t, e0) + ku * 8u);
            void const *const recb = v.rec + pe0;
            double const eps = 1 - tau;
            double eps_p0 = 0, eps_p1 = 0;
#pragma unroll 1
            for (int h = 0; h < 2; h++) {
              int const kc = h ? kc1 : kc0;
              int const ea = ne[kc], eb = ne[kc + 1], ec_ = ne[kc + 2];
              unsigned const packed = pos[h ? HS : 0];
              unsigned const e0[2] = {(unsigned)ea, (unsigned)eb};
              int const n[2] = {eb - ea, ec_ - eb};
              int i[2] = {(int)(packed & 0xffffu), (int)(packed >> 16)};
#pragma unroll
              for (int q = 0; q < 2; q++) i[q] = min(i[q], n[q] - 2);
              double x[2], ec[2];
              {   // bracket records (one fetch per curve), as jur_ega_kernel
                Rec r[2];
                ld_rec2(recb, e0[0] + i[0], e0[1] + i[1], r[0], r[1]);
#pragma unroll
                for (int q = 0; q < 2; q++) {
                  double ka, kb;
                  seek_rec<true, false>(recb, e0[q], n[q], eps, i[q], r[q], ka, kb);
                  double const ya = cvt_keep(r[q].u0);
                  x[q] = lip_slope(ka, ya, r[q].du_de, eps) + u;
                  ka = ya; kb = cvt_keep(r[q].u1);
                  seek_rec<false, true>(recb, e0[q], n[q], x[q], i[q], r[q], ka, kb);
                  ec[q] = c01_num(lip_slope(ka, (double)r[q].e0, r[q].de_du, x[q]));
                }
              }
              pos[h ? HS : 0] = (unsigned)i[0] | ((unsigned)i[1] << 16);
              // lip_mulr(T0, ec0, ec1, t, rt) with its (t - T0) formed above
              double const e = c01_num(ec[0] + ((h ? dt1 : dt0) * (ec[1] - ec[0])) * sRT[kc]);
              if (h) eps_p1 = e; else eps_p0 = e;
            }
            tau_new = 1. - c01_num(eps_p0 + (dp * (eps_p1 - eps_p0)) * sRP[br & 0xffu]);
          }
        }
      }
      mytau[k * BLOCK] = tau_new;
      int const dk = ld_scalar<int>(v.ega_items, item_byte + (unsigned)offsetof(jur_item_t, d) + ku * 4u);
      double *const out = out_tile + (size_t)(dk * v.ng + g) * 64;
      st_stream(out + (size_t)ip * Re, lane, tau_new);
    }
  }
}

// Everything the radiance update reads for one segment, requested together: the LOS fields of the point (one
// contiguous run of the tile) and the path transmittances of the channel's gases, multiplied in the reference's gas
// order (jr_common.h:272-278).  All loads are issued before the first value is used -- one memory latency per segment.
// (Rounds 1-2 loaded each gas's transmittance inside the product loop, and the continua's columns inside their
// branches: the compiler waited for each in turn, seven dependent round trips to HBM per segment, and the kernel sat
// at half the bandwidth it asks for with idle vector units.)  Gases beyond the first eight take a second batch.
struct SegmentIn { double p, t, ds, k, u_co2, q_h2o, u_h2o, pcur; };
__device__ __forceinline__ SegmentIn load_segment(double const *__restrict__ row, double const *__restrict__ trow, unsigned lane,
                                                  int f_k, int f_co2, int f_h2o, int ng, unsigned has_table) {
  SegmentIn in;
  in.p = ldg<double>(row + JUR_F_P * 64, lane);
  in.t = ldg<double>(row + JUR_F_T * 64, lane);
  in.ds = ldg<double>(row + JUR_F_DS * 64, lane);
  in.k = ldg<double>(row + (size_t)f_k * 64, lane);
  in.u_co2 = ldg<double>(row + (size_t)f_co2 * 64, lane);
  in.q_h2o = ldg<double>(row + JUR_F_QH2O * 64, lane);
  in.u_h2o = ldg<double>(row + (size_t)f_h2o * 64, lane);
  double tg[8];
#pragma unroll
  for (int k = 0; k < 8; k++) tg[k] = (k < ng && ((has_table >> k) & 1u)) ? ld_stream(trow + (size_t)k * 64, lane) : 1.0;
  double pcur = 1.0;
#pragma unroll
  for (int k = 0; k < 8; k++) pcur *= tg[k];           // (x * 1.0 == x: absent gases do not change the product)
  for (int g0 = 8; g0 < ng; g0 += 8) {
#pragma unroll
    for (int k = 0; k < 8; k++) tg[k] = (g0 + k < ng && g0 + k < 32 && ((has_table >> (g0 + k)) & 1u)) ? ld_stream(trow + (size_t)(g0 + k) * 64, lane) : 1.0;
#pragma unroll
    for (int k = 0; k < 8; k++) pcur *= tg[k];
  }
  in.pcur = pcur;
  return in;
}

// jur_combine_kernel: one lane per ray, one channel per workgroup: continua, product of the gas
// transmittances in the reference's order, Planck source, radiance update, epilogue.
// (59 VGPRs since the exponentials go through exp_tab: 8 waves per SIMD)
__global__ __launch_bounds__(256, 6) void jur_combine_kernel(jur_view_t v, jur_chunk_t c, int nrb) {
  int const nd = v.nd, ng = v.ng;
  // same XCD-aware order as jur_ega_kernel: the nd workgroups of one ray block follow each other
  // on one XCD and share the block's LOS rows in that L2
  BlockItem const bi = xcd_block_item((int)blockIdx.x, nrb, nd);
  int const rb = bi.rb, d = bi.item;                             // ray block, channel: uniform
  if (rb < 0) return;
  int const r = rb * blockDim.x + threadIdx.x;
  // the channel's source-function table (1201 doubles) is staged in LDS: its two reads per segment are
  // gathers by temperature, everything else this kernel loads is a coalesced stream
  double *const sr = reinterpret_cast<double *>(jur_lds);
  Exp2Lds const e2t{sr + TBLNS};                       // 2^(j/64), 64 doubles behind the source-function table
  {
    double const *const gsr = v.sr + (size_t)d * TBLNS;
    for (int i = threadIdx.x; i < TBLNS; i += blockDim.x) sr[i] = gsr[i];
    if (threadIdx.x < 64) sr[TBLNS + threadIdx.x] = JUR_EXP2_64[threadIdx.x];
    __syncthreads();
  }
  if (r >= c.n) return;
  long const ray = c.order ? (long)c.order[r] : c.first + r;
  int const nfield = JUR_F_K + v.nw + ng;
  int const tile = __builtin_amdgcn_readfirstlane(r >> 6);
  unsigned const lane = (unsigned)(r & 63);
  size_t const R = (size_t)nfield * 64, Re = (size_t)nd * ng * 64;   // doubles from one LOS point to the next
  double const *const los = c.los + (size_t)tile * los_tile_doubles(nfield);
  double const *const epsb = c.eps + eps_tile_point0(c, tile) * Re + (size_t)d * ng * 64;
  jur_chan_t const ch = v.chan[d];
  int const f_k = JUR_F_K + ch.window, f_u = JUR_F_K + v.nw;
  bool const do_co2 = (v.fourbit & 8) && ch.co2_on, do_h2o = (v.fourbit & 4) && ch.h2o_on,
             do_n2 = (v.fourbit & 2) && ch.n2_on, do_o2 = (v.fourbit & 1) && ch.o2_on;
  size_t const oidx = (size_t)ray * nd + d;
  bool const masked = !isfinite(c.rad[oidx]);
  double rad = 0.0, tau = 1.0, pprev = 1.0;
  int const np = c.np[r];
  unsigned has_table = 0;                              // gases with a table for this channel (uniform)
  for (int g = 0; g < ng && g < 32; g++) has_table |= (v.pair[g * nd + d].a >= 2 ? 1u : 0u) << g;
  int const f_co2 = do_co2 ? f_u + v.ig_co2 : JUR_F_P, f_h2o = do_h2o ? f_u + v.ig_h2o : JUR_F_P;   // (a field that always exists when the continuum is off)
  for (int ip = 0; ip < np; ++ip) {
    SegmentIn const in = load_segment(los + (size_t)ip * R, epsb + (size_t)ip * Re, lane, f_k, f_co2, f_h2o, ng, has_table);
    double beta_ds = in.k * in.ds;
    if (do_co2) beta_ds += ctm_co2(ch, in.p, in.t, in.u_co2);
    double const rt = rcp_t(in.t);
    if (do_h2o) beta_ds += ctm_h2o(e2t, ch, in.p, in.t, rt, in.q_h2o, in.u_h2o);
    if (do_n2) beta_ds += ctm_n2(e2t, ch, in.p, in.t, rt) * in.ds;
    if (do_o2) beta_ds += ctm_o2(e2t, ch, in.p, in.t, rt) * in.ds;
    double const tau_gas = segment_tau_gas(in.pcur, pprev);
    pprev = in.pcur;
    new_obs_step(e2t, tau_gas, beta_ds, planck_src(sr, in.t), rad, tau);
  }
  ray_epilogue(sr, ch.nu, c.tsurf[r], v.write_bbt, rad, tau);
  if (masked) rad = __builtin_nan("");
  c.rad[oidx] = rad;
  c.tau[oidx] = tau;
}

// jur_combine_group_kernel: the same lane program with the channels of a ray block as wavefronts of ONE workgroup --
// wave w works on channel cg * CG + w % CG for the 64 rays of sub-block w / CG (CG = min(nd, 4) channels per group,
// 8 / CG sub-blocks, 6 or 8 waves) -- and a barrier every SYNC segments that keeps the waves within reach of each
// other: the LOS rows of a segment, which every channel reads, are then found in L2 by all but the first reader
// (FETCH_SIZE of the kernel -29 %).  The kernel waits on its dependent chains more than on HBM, so this is worth
// 4 % of its time, not 29 (17.3 against 18.1 ms per 1e6 limb rays); one channel per workgroup remains for nd = 1.
// The wave index is made uniform (readfirstlane) so that the channel constants stay in scalar registers.
// 8 waves per SIMD (56 VGPRs, no scratch): a workgroup is 8 wavefronts, so anything above 64 VGPRs runs three
// workgroups per CU instead of four; 13.1 ms per 1e6 limb rays against 14.7 ms at 69 VGPRs (A/B: -DJUR_COMBINE_WAVES=6)
#ifndef JUR_COMBINE_WAVES
#define JUR_COMBINE_WAVES 8
#endif
__global__ __launch_bounds__(512, JUR_COMBINE_WAVES) void jur_combine_group_kernel(jur_view_t v, jur_chunk_t c, int nsb, int CG, int SYNC) {   // SYNC: mask, see the loop
  int const nd = v.nd, ng = v.ng;
  int const ncg = (nd + CG - 1) / CG, SUB = (int)(blockDim.x >> 6) / CG;
  BlockItem const bi = xcd_block_item((int)blockIdx.x, nsb, ncg);
  int const sb = bi.rb, cg = bi.item;                                // ray super-block, channel group: uniform
  if (sb < 0) return;
  int const w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);   // uniform: channel constants stay in SGPRs
  int const cw = w % CG, sub = w / CG;
  int const d = cg * CG + cw;
  bool const live_wave = d < nd;
  double *const sr = reinterpret_cast<double *>(jur_lds) + (size_t)cw * TBLNS;
  int *const npmax_sh = reinterpret_cast<int *>(reinterpret_cast<double *>(jur_lds) + (size_t)CG * TBLNS);
  Exp2Lds const e2t{reinterpret_cast<double *>(jur_lds) + (size_t)CG * TBLNS + 2};   // 2^(j/64), 64 doubles
  if (threadIdx.x == 0) *npmax_sh = 0;
  if (threadIdx.x < 64) reinterpret_cast<double *>(jur_lds)[(size_t)CG * TBLNS + 2 + threadIdx.x] = JUR_EXP2_64[threadIdx.x];
  for (int k = 0; k < CG; k++) {
    if (cg * CG + k >= nd) break;
    double const *const gsr = v.sr + (size_t)(cg * CG + k) * TBLNS;
    double *const dst = reinterpret_cast<double *>(jur_lds) + (size_t)k * TBLNS;
    for (int i = threadIdx.x; i < TBLNS; i += blockDim.x) dst[i] = gsr[i];
  }
  __syncthreads();
  int const r = (sb * SUB + sub) * 64 + lane;
  bool const live = live_wave && r < c.n;
  int const np = live ? c.np[r] : 0;
  atomicMax(npmax_sh, np);
  __syncthreads();
  int const npmax = *npmax_sh;
  long const ray = live ? (c.order ? (long)c.order[r] : c.first + r) : 0;
  int const nfield = JUR_F_K + v.nw + ng;
  int const tile = sb * SUB + sub;                                   // (uniform: w is)
  size_t const R = (size_t)nfield * 64, Re = (size_t)nd * ng * 64;   // doubles from one LOS point to the next
  double const *const los = c.los + (size_t)tile * los_tile_doubles(nfield);
  int const dd = live_wave ? d : 0;
  double const *const epsb = c.eps + eps_tile_point0(c, tile) * Re + (size_t)dd * ng * 64;
  jur_chan_t const ch = v.chan[dd];
  int const f_k = JUR_F_K + ch.window, f_u = JUR_F_K + v.nw;
  bool const do_co2 = (v.fourbit & 8) && ch.co2_on, do_h2o = (v.fourbit & 4) && ch.h2o_on,
             do_n2 = (v.fourbit & 2) && ch.n2_on, do_o2 = (v.fourbit & 1) && ch.o2_on;
  size_t const oidx = (size_t)ray * nd + dd;
  bool const masked = live && !isfinite(c.rad[oidx]);
  double rad = 0.0, tau = 1.0, pprev = 1.0;
  unsigned has_table = 0;
  for (int g = 0; g < ng && g < 32; g++) has_table |= (v.pair[g * nd + dd].a >= 2 ? 1u : 0u) << g;
  int const f_co2 = do_co2 ? f_u + v.ig_co2 : JUR_F_P, f_h2o = do_h2o ? f_u + v.ig_h2o : JUR_F_P;   // (a field that always exists when the continuum is off)
  for (int ip = 0; ip < npmax; ++ip) {
    if (ip < np) {
      SegmentIn const in = load_segment(los + (size_t)ip * R, epsb + (size_t)ip * Re, (unsigned)lane, f_k, f_co2, f_h2o, ng, has_table);
      double beta_ds = in.k * in.ds;
      if (do_co2) beta_ds += ctm_co2(ch, in.p, in.t, in.u_co2);
      double const rt = rcp_t(in.t);
      if (do_h2o) beta_ds += ctm_h2o(e2t, ch, in.p, in.t, rt, in.q_h2o, in.u_h2o);
      if (do_n2) beta_ds += ctm_n2(e2t, ch, in.p, in.t, rt) * in.ds;
      if (do_o2) beta_ds += ctm_o2(e2t, ch, in.p, in.t, rt) * in.ds;
      double const tau_gas = segment_tau_gas(in.pcur, pprev);
      pprev = in.pcur;
      new_obs_step(e2t, tau_gas, beta_ds, planck_src(sr, in.t), rad, tau);
    }
    if ((ip & SYNC) == SYNC) __syncthreads();      // SYNC = 2^k - 1 (0: a barrier after every segment; -1 never matches: none)
  }
  if (!live) return;
  ray_epilogue(sr, ch.nu, c.tsurf[r], v.write_bbt, rad, tau);
  if (masked) rad = __builtin_nan("");
  c.rad[oidx] = rad;
  c.tau[oidx] = tau;
}

// ---------------------------------------------------------------------------------------
// jur_pencil_kernel: the whole path of a ray pencil inside ONE workgroup -- for calls of the size the
// reference's callers make (packages of <= NR = 1088 rays, formod.c:100, kernel() jurassic.c:844), which cannot
// fill 256 CUs with one lane per ray and would run the three batched kernels as three serial <= 400-step chains.
//
// A workgroup owns RB rays.  Its wavefronts take roles and hand the line of sight from one to the next through
// LDS, point by point, as it is being traced:
//   wave 0              traces the rays (trace_ray) into a ring of LOS points in LDS -- up to 16 rays per workgroup
//                       with a quad of lanes per ray, one refraction probe each;
//   waves 1 .. NE       the emissivity-growth recurrence of every (ray, channel, gas) chain on the points the tracer
//                       has released -- sorted tables: a quad of lanes per chain, one (p, T) corner curve each
//                       (ega_eps_warm_quad); search state and accumulated transmittance of the chain in LDS;
//                       segment transmittances go to a second ring;
//   waves NE+1 .. +NC   one lane per (ray, channel): continua, product over the gases, Planck source, radiance
//                       update, epilogue -- the body of jur_combine_kernel.
// The roles overlap along the ray: a call takes about as long as its longest ray takes to TRACE, the LOS state
// and the segment transmittances never touch HBM, and nothing is sorted.  Same device functions, same operand
// order, same doubles as the batched kernels (tests compare the two paths bit for bit).
//
// Hand-over: cnt_trace = number of points of every still-running ray that will not change any more (the lanes of
// wave 0 step in lockstep), npr[ray] = 1 + its point count once the ray has left the atmosphere; every consumer wave
// publishes the number of points it has finished.  All counters live in LDS, release/acquire at workgroup scope.
// Every wait is for a wave of the same workgroup that is resident and never waits for the waiter in turn
// (tracer <- combine: ring slot free; ega <- tracer: point released, <- combine: eps slot free; combine <- ega),
// so every wave reaches its exit.
// ---------------------------------------------------------------------------------------
#define PEN_RING 16       // LOS points in flight (power of two)
#define PEN_RINGE 8       // segment transmittances in flight (power of two, <= PEN_RING)
#define PEN_MAXE 8        // at most this many ega / combine waves per workgroup
#define PEN_MAXC 4

// doubles of dynamic LDS in front of the optional profile slab (rings, chain and radiance state, per-ray results,
// the 32-bit state rounded up to whole doubles)
__host__ __device__ inline long pen_lds_doubles(int nd, int ng, int nw, int RB) {
  long const npair = (long)nd * ng, nfield = JUR_F_K + nw + ng, nchain = RB * npair, nitem = (long)RB * nd;
  long const n1 = nchain > 0 ? nchain : 1;
  return (long)PEN_RING * nfield * RB + (long)PEN_RINGE * (npair > 0 ? npair : 1) * RB + n1 + 3 * nitem + RB + (5 * n1 + RB + 1) / 2 + 2;
}

struct PenCtl {            // LDS, one per workgroup
  int cnt_trace, done;
  int cnt_ega[PEN_MAXE];
  int cnt_comb[PEN_MAXC];
};

__device__ __forceinline__ int ld_acq(int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_rel(int *p, int x) { __hip_atomic_store(p, x, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int min_cnt(int *p, int n) {
  int m = ld_acq(p);
  for (int i = 1; i < n; i++) m = min(m, ld_acq(p + i));
  return m;
}
__device__ __forceinline__ bool first_active_lane() {
  return (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1;
}

// LOS fields of the workgroup's rays in the LDS ring [point % PEN_RING][field][ray]
struct LosRing {
  double *ring;
  PenCtl *ctl;
  int *npr;
  double *tsurf;
  int nfield, RB, r, nc;
  int shift;               // lanes per ray in the tracer wavefront = 1 << shift (ray slot = lane >> shift)
  __device__ __forceinline__ double &at(int field, int ip) const { return ring[((ip & (PEN_RING - 1)) * nfield + field) * RB + r]; }
  __device__ __forceinline__ void put(int field, int ip, double x) const { at(field, ip) = x; }
  __device__ __forceinline__ size_t field_stride() const { return (size_t)RB; }
  __device__ __forceinline__ void begin_point(int ip) const {   // the slot still holds point ip - PEN_RING
    if (ip >= PEN_RING)
      while (min_cnt(ctl->cnt_comb, nc) + PEN_RING <= ip) __builtin_amdgcn_s_sleep(2);
  }
  // Lanes whose rays left the atmosphere in this step have dropped out of the loop; their own announcement
  // (ray_final) may be scheduled after the loop of the others, so a lane that is still stepping says it for them,
  // in its own program order before the release of the next point: they stopped at point np, i.e. have np + 1 points.
  __device__ __forceinline__ void points_final(int np) {
    unsigned long long const cur = __ballot(1);
    unsigned long long gone = running & ~cur;
    running = cur;
    if (first_active_lane()) {
      for (; gone; gone &= gone - 1) {
        int const l = __ffsll((long long)gone) - 1;
        if ((l & ((1 << shift) - 1)) == 0) st_rel(&npr[l >> shift], np + 2);
      }
      st_rel(&ctl->cnt_trace, np);
    }
  }
  __device__ __forceinline__ void ray_final(int n, double ts) const {
    tsurf[r] = ts;
    st_rel(&npr[r], n + 1);
  }
  __device__ __forceinline__ void never_enter(unsigned long long mask) {   // lane number >> shift == ray slot in wave 0
    running = __ballot(1) & ~mask;
    if (first_active_lane())
      for (; mask; mask &= mask - 1) {
        int const l = __ffsll((long long)mask) - 1;
        if ((l & ((1 << shift) - 1)) == 0) st_rel(&npr[l >> shift], 1);
      }
  }
  unsigned long long running;
};

// wait until point ip is released (returns the released count c > ip) or the tracer is through (returns its final
// count, possibly <= ip: no such point)
__device__ __forceinline__ int wait_point(PenCtl *ctl, int ip) {
  for (;;) {
    int c = ld_acq(&ctl->cnt_trace);
    if (c > ip) return c;
    if (ld_acq(&ctl->done)) return ld_acq(&ctl->cnt_trace);
    __builtin_amdgcn_s_sleep(2);
  }
}

// first point of the profile slice with this time stamp (the first search of locate_atm, jr_common.h:130-140)
__device__ __forceinline__ int slice_start(double const *__restrict__ atm_time, int atm_np, double time) {
  int lo = 0, hi = atm_np - 1;
  while (hi > lo + 1) {
    int const i = (lo + hi) / 2;
    if (atm_time[i] < time) lo = i; else hi = i;
  }
  return (0 == lo) ? lo : hi;
}

// atm_cap > 0: the dynamic LDS block ends in room for (7 + ng + nw) rows of atm_cap doubles -- the profile slice of
// the workgroup's rays (time, z, lon, lat, p, T, q[], k[], ln-p slope), copied there when all its rays use the same
// slice.  The tracer's per-step profile gathers (a chain of dependent loads at L2 latency when one wave per SIMD
// has nothing to hide them behind) then stay inside the CU.  Same numbers from another place: same results.
template <bool WARM, bool QUAD>
__global__ __launch_bounds__(1024) void jur_pencil_kernel(jur_view_t v, jur_chunk_t c, int RB, int NE, int NC, int atm_cap) {
  __shared__ double tr_sh[15][64];
  __shared__ PenCtl ctl;
  __shared__ int sl_lo, sl_hi;
  int const nd = v.nd, ng = v.ng, npair = nd * ng, nfield = JUR_F_K + v.nw + ng;
  int const nchain = RB * npair, nitem = RB * nd;
  // dynamic LDS: rings, chain state, per-(ray, channel) radiance state, per-ray results of the tracer
  double *const ring = reinterpret_cast<double *>(jur_lds);
  double *const epsr = ring + (size_t)PEN_RING * nfield * RB;
  double *const st_tau = epsr + (size_t)PEN_RINGE * (npair > 0 ? npair : 1) * RB;
  double *const c_rad = st_tau + (nchain > 0 ? nchain : 1);
  double *const c_tau = c_rad + nitem;
  double *const c_pp = c_tau + nitem;             // product of the gases' path transmittances after the previous segment
  double *const tsurf = c_pp + nitem;
  unsigned *const st_br = reinterpret_cast<unsigned *>(tsurf + RB);
  unsigned *const st_ix = st_br + (nchain > 0 ? nchain : 1);     // [chain][4]: curve positions (ia, ib packed in [0], [1] without quads)
  int *const npr = reinterpret_cast<int *>(st_ix + 4 * (nchain > 0 ? nchain : 1));
  int const tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  long const ray0 = (long)blockIdx.x * RB;                         // first slot of this workgroup
  int const nray = (int)((c.n - ray0 < RB) ? c.n - ray0 : RB);    // rays it really has

  for (int i = tid; i < nchain; i += blockDim.x) { st_tau[i] = 1.0; st_br[i] = 0; st_ix[4 * i] = st_ix[4 * i + 1] = st_ix[4 * i + 2] = st_ix[4 * i + 3] = 0; }
  for (int i = tid; i < nitem; i += blockDim.x) { c_rad[i] = 0.0; c_tau[i] = 1.0; c_pp[i] = 1.0; }
  for (int i = tid; i < RB; i += blockDim.x) { npr[i] = (i < nray) ? 0 : 1; tsurf[i] = -999; }   // 1: through, no points
  if (tid == 0) {
    ctl.cnt_trace = 0; ctl.done = 0;
    for (int i = 0; i < PEN_MAXE; i++) ctl.cnt_ega[i] = 0;
    for (int i = 0; i < PEN_MAXC; i++) ctl.cnt_comb[i] = 0;
    sl_lo = 0x7fffffff; sl_hi = -1;
  }
  __syncthreads();
  jur_view_t vt = v;                            // the tracer's view of the atmosphere
  if (atm_cap > 0) {
    if (tid < nray) {                           // do all rays of the workgroup use one slice?
      double const time = c.geom[0][c.first + ray0 + tid];
      int const a0 = slice_start(v.atm_time, v.atm_np, time);
      atomicMin(&sl_lo, a0);
      atomicMax(&sl_hi, (v.atm_time[a0] == time) ? a0 : 0x7ffffffe);   // a time stamp without a profile: no copy
    }
    __syncthreads();
    int const a0 = sl_lo;
    if (a0 == sl_hi) {
      // its extent: the points that share the first one's time stamp (time stamps are sorted when atm_sorted;
      // otherwise no copy is made)
      int n = 0;
      if (v.atm_sorted) {
        double const t0 = v.atm_time[a0];
        int lo = a0, hi = v.atm_np;              // first index in (a0, atm_np] whose time differs
        while (hi > lo + 1) {
          int const i = (lo + hi) / 2;
          if (v.atm_time[i] == t0) lo = i; else hi = i;
        }
        n = hi - a0;
      }
      if (n >= 2 && n <= atm_cap) {
        int const nrow = 7 + ng + v.nw;
        double *const slab = reinterpret_cast<double *>(jur_lds) + pen_lds_doubles(nd, ng, v.nw, RB);
        double const *const rows[7] = {v.atm_time, v.atm_z, v.atm_lon, v.atm_lat, v.atm_p, v.atm_t, v.atm_pslope};
        for (int i = tid; i < nrow * n; i += blockDim.x) {
          int const row = i / n, k = i - row * n;
          double x;
          if (row < 7) x = rows[row][a0 + k];
          else if (row < 7 + ng) x = v.atm_q[(size_t)(row - 7) * v.atm_np + a0 + k];
          else x = v.atm_k[(size_t)(row - 7 - ng) * v.atm_np + a0 + k];
          slab[(size_t)row * n + k] = x;
        }
        vt.atm_np = n;
        vt.atm_time = slab; vt.atm_z = slab + n; vt.atm_lon = slab + 2 * n; vt.atm_lat = slab + 3 * n;
        vt.atm_p = slab + 4 * n; vt.atm_t = slab + 5 * n; vt.atm_pslope = slab + 6 * n;
        vt.atm_q = slab + 7 * (size_t)n; vt.atm_k = slab + (7 + (size_t)ng) * n;
        __syncthreads();                          // (uniform branch: every thread of the workgroup is here)
      }
    }
  }

  if (wave == 0) {
    // ---- tracer ----  (QUAD: four lanes per ray, RB <= 16)
    constexpr int SH = QUAD ? 2 : 0;
    int const slot = lane >> SH;
    if (slot < nray) {
      long const ray = c.first + ray0 + slot;
      LosRing L{ring, &ctl, npr, tsurf, nfield, RB, slot, NC, SH, 0ull};
      TraceResult const t = trace_ray<LosRing, QUAD ? 4 : 1>(vt, c.geom[0][ray], c.geom[1][ray], c.geom[2][ray], c.geom[3][ray],
                                                     c.geom[4][ray], c.geom[5][ray], c.geom[6][ray], L, tr_sh, c.status);
      if ((lane & ((1 << SH) - 1)) == 0) {
        if (c.np_out) c.np_out[ray] = t.np;
        c.tp[0][ray] = t.tpz;
        c.tp[1][ray] = t.tplon;
        c.tp[2][ray] = t.tplat;
        tsurf[slot] = t.tsurf;
        st_rel(&npr[slot], t.np + 1);        // also the rays that never entered the atmosphere or ran into NLOS
      }
    }
    if (lane == 0) {                         // every ray is through: its point count governs from here on
      int m = 0;
      for (int i = 0; i < nray; i++) m = max(m, ld_acq(&npr[i]) - 1);
      st_rel(&ctl.cnt_trace, m);
      st_rel(&ctl.done, 1);
    }
  } else if (wave <= NE) {
    // ---- emissivity growth: one chain per (ray, channel, gas); with QUAD (sorted tables) a quad of lanes per
    // chain, one corner curve each ----
    constexpr int CS = (QUAD && WARM) ? 2 : 0;                  // lanes per chain = 1 << CS
    int const w = wave - 1, nl = (NE * 64) >> CS, me = (w * 64 + lane) >> CS;
    PairDesc<false> D{v.lvl, v.crv, nullptr, nullptr, 0u, 0u, 0u, 0u};
    for (int ip = 0;; ++ip) {
      int const cnt = wait_point(&ctl, ip);
      if (cnt <= ip) break;
      if (ip >= PEN_RINGE)
        while (min_cnt(ctl.cnt_comb, NC) + PEN_RINGE <= ip) __builtin_amdgcn_s_sleep(2);
      double const *const slot = ring + (size_t)(ip & (PEN_RING - 1)) * nfield * RB;
      double *const eslot = epsr + (size_t)(ip & (PEN_RINGE - 1)) * npair * RB;
      for (int e = me; e < nchain; e += nl) {
        int const pr = e / RB, r = e - pr * RB, d = pr / ng, g = pr - d * ng;
        int const n1 = ld_acq(&npr[r]);
        if (n1 ? (ip >= n1 - 1) : (ip >= cnt)) continue;          // this ray has no such point
        jur_int2 const pd = v.pair[g * nd + d];
        if (pd.a < 2) continue;                                    // no table: the combine role knows
        D.l0 = (unsigned)pd.b;
        D.ueb = v.ue + v.pair_e0[g * nd + d];
        D.slb = v.sl + v.pair_e0[g * nd + d];      // only read by the strict-table arithmetic, where v.sl is set
        D.recb = (v.fast_arith && v.rec) ? static_cast<void const *>(v.rec + v.pair_e0[g * nd + d]) : nullptr;
        double const p = slot[JUR_F_P * RB + r], t = slot[JUR_F_T * RB + r], u = slot[(JUR_F_K + v.nw + g) * RB + r];
        double const tau_path = st_tau[e];
        double tau_new;                                            // the gas's path transmittance after this segment
        if constexpr (WARM && QUAD) {
          unsigned br = st_br[e], ix = st_ix[4 * e + (lane & 3)];
          if (v.fast_arith) tau_new = ega_eps_warm_quad<true>(v, pd, D, tau_path, t, u, p, br, ix);
          else tau_new = tau_path * ega_eps_warm_quad<false>(v, pd, D, tau_path, t, u, p, br, ix);
          st_br[e] = br; st_ix[4 * e + (lane & 3)] = ix;
        } else if constexpr (WARM) {
          unsigned br = st_br[e], ia = st_ix[4 * e], ib = st_ix[4 * e + 1];
          if (v.fast_arith && v.rec) tau_new = ega_eps_warm<false, true, true, true>(v, pd, D, tau_path, t, u, p, br, ia, ib);
          else if (v.fast_arith) tau_new = ega_eps_warm<false, true, true>(v, pd, D, tau_path, t, u, p, br, ia, ib);
          else tau_new = tau_path * ega_eps_warm<false, false>(v, pd, D, tau_path, t, u, p, br, ia, ib);
          st_br[e] = br; st_ix[4 * e] = ia; st_ix[4 * e + 1] = ib;
        } else tau_new = tau_path * ega_eps_exact<false>(v, pd, D, tau_path, t, u, p);
        st_tau[e] = tau_new;
        eslot[pr * RB + r] = tau_new;
      }
      if (first_active_lane()) st_rel(&ctl.cnt_ega[w], ip + 1);
    }
  } else {
    // ---- combine: one lane per (ray, channel) ----
    int const w = wave - 1 - NE, nl = NC * 64, me = w * 64 + lane;
    int const f_u = JUR_F_K + v.nw;
    Exp2Global const e2t;
    for (int ip = 0;; ++ip) {
      int const cnt = wait_point(&ctl, ip);
      if (cnt <= ip) break;
      while (min_cnt(ctl.cnt_ega, NE) <= ip) __builtin_amdgcn_s_sleep(2);
      double const *const slot = ring + (size_t)(ip & (PEN_RING - 1)) * nfield * RB;
      double const *const eslot = epsr + (size_t)(ip & (PEN_RINGE - 1)) * npair * RB;
      for (int i = me; i < nitem; i += nl) {
        int const d = i / RB, r = i - d * RB;
        int const n1 = ld_acq(&npr[r]);
        if (n1 ? (ip >= n1 - 1) : (ip >= cnt)) continue;
        jur_chan_t const ch = v.chan[d];
        auto L = [&](int field) { return slot[field * RB + r]; };
        double const p = L(JUR_F_P), t = L(JUR_F_T), ds = L(JUR_F_DS);
        double beta_ds = L(JUR_F_K + ch.window) * ds;
        if ((v.fourbit & 8) && ch.co2_on) beta_ds += ctm_co2(ch, p, t, L(f_u + v.ig_co2));
        double const rt = rcp_t(t);
        if ((v.fourbit & 4) && ch.h2o_on) beta_ds += ctm_h2o(e2t, ch, p, t, rt, L(JUR_F_QH2O), L(f_u + v.ig_h2o));
        if ((v.fourbit & 2) && ch.n2_on) beta_ds += ctm_n2(e2t, ch, p, t, rt) * ds;
        if ((v.fourbit & 1) && ch.o2_on) beta_ds += ctm_o2(e2t, ch, p, t, rt) * ds;
        double pcur = 1.0;
        for (int g = 0; g < ng; g++)                       // jr_common.h:272-278
          if (v.pair[g * nd + d].a >= 2) pcur *= eslot[(d * ng + g) * RB + r];
        double const tau_gas = segment_tau_gas(pcur, c_pp[i]);
        c_pp[i] = pcur;
        double rad = c_rad[i], tau = c_tau[i];
        new_obs_step(e2t, tau_gas, beta_ds, planck_src(v.sr + (size_t)d * TBLNS, t), rad, tau);
        c_rad[i] = rad;
        c_tau[i] = tau;
      }
      if (first_active_lane()) st_rel(&ctl.cnt_comb[w], ip + 1);
    }
    for (int i = me; i < nitem; i += nl) {                 // every ray is through (done was seen): epilogue
      int const d = i / RB, r = i - d * RB;
      if (r >= nray) continue;
      size_t const oidx = (size_t)(c.first + ray0 + r) * nd + d;
      bool const masked = !isfinite(c.rad[oidx]);
      double rad = c_rad[i];
      double const tau = c_tau[i];
      ray_epilogue(v.sr + (size_t)d * TBLNS, v.chan[d].nu, tsurf[r], v.write_bbt, rad, tau);
      if (masked) rad = __builtin_nan("");
      c.rad[oidx] = rad;
      c.tau[oidx] = tau;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Curtis-Godson means along the path (curtis_godson, jr_common.h:455-473; upstream compiles it only
// with -DCURTIS_GODSON for FORMOD=1 and never consumes the result): per gas, the inclusive prefix
// sums over the LOS points of u p, u T and u, then cgp = S(u p)/S(u), cgt = S(u T)/S(u), cgu = S(u).
// One wavefront per (ray, gas) pencil, lanes = LOS points: the along-path prefix is a wave scan
// (shuffles), carried from one 64-point chunk to the next.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_inclusive_scan(double x, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    double const y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  return x;
}

__global__ __launch_bounds__(256) void jur_cg_kernel(jur_view_t v, jur_chunk_t c, double *__restrict__ cgp,
                                                     double *__restrict__ cgt, double *__restrict__ cgu) {
  int const lane = threadIdx.x & 63;
  long const wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;   // (slot, gas) pencil
  int const ng = v.ng;
  if (wave >= (long)c.n * ng) return;
  int const r = (int)(wave / ng), g = (int)(wave - (long)r * ng);
  long const ray = c.order ? (long)c.order[r] : c.first + r;
  int const nfield = JUR_F_K + v.nw + ng;
  size_t const R = (size_t)nfield * 64;                              // doubles from one LOS point of a ray to the next
  double const *const tile = c.los + (size_t)(r >> 6) * los_tile_doubles(nfield) + (r & 63);
  double const *const lp = tile + JUR_F_P * 64, *const lt = tile + JUR_F_T * 64,
               *const lu = tile + (size_t)(JUR_F_K + v.nw + g) * 64;
  int const np = c.np[r];
  size_t const out = ((size_t)ray * ng + g) * NLOS;
  double cp = 0, ct = 0, cu = 0;                                         // carry of the previous chunks
  for (int base = 0; base < NLOS; base += 64) {
    int const ip = base + lane;
    double a = 0, b = 0, w = 0;
    if (ip < np) {
      double const u = lu[(size_t)ip * R];
      a = u * lp[(size_t)ip * R];
      b = u * lt[(size_t)ip * R];
      w = u;
    }
    a = wave_inclusive_scan(a, lane) + cp;
    b = wave_inclusive_scan(b, lane) + ct;
    w = wave_inclusive_scan(w, lane) + cu;
    cp = __shfl(a, 63, 64); ct = __shfl(b, 63, 64); cu = __shfl(w, 63, 64);
    if (ip < np) { cgp[out + ip] = a / w; cgt[out + ip] = b / w; cgu[out + ip] = w; }
    else if (ip < NLOS) { cgp[out + ip] = 0; cgt[out + ip] = 0; cgu[out + ip] = 0; }   // NLOS is not a multiple of 64
  }
}

// ---------------------------------------------------------------------------------------
// ray ordering key: altitude of the straight line's closest approach to the Earth's centre;
// optionally grouped by the atmosphere slice the ray uses (neighbouring lanes then walk through
// the same profile, i.e. the same table brackets)
// ---------------------------------------------------------------------------------------
// Bracket slopes of the emissivity curves, once per model: entry j gets the slopes of [entry j, entry j+1] in both
// directions, formed from the fp32 entries with fp64 differences and IEEE divisions.  The last entry of a curve pairs
// with the first of the next one; the look-up never reads that slot (its bracket index ends at nu - 2).
// bracket records from the entries and their slopes (entry i and i + 1, slopes of bracket [i, i+1]); once per model
__global__ __launch_bounds__(256) void jur_records_kernel(long long n, jur_ue_t const *__restrict__ ue, jur_sl_t const *__restrict__ sl,
                                                          jur_rec_t *__restrict__ rec) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    rec[i] = jur_rec_t{ue[i].u, ue[i].eps, ue[i + 1].u, ue[i + 1].eps, sl[i].du_de, sl[i].de_du};
}

__global__ __launch_bounds__(256) void jur_slopes_kernel(long long n, jur_ue_t const *__restrict__ ue, jur_sl_t *__restrict__ sl) {
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j + 1 < n; j += (long long)gridDim.x * blockDim.x) {
    double const du = (double)ue[j + 1].u - (double)ue[j].u, de = (double)ue[j + 1].eps - (double)ue[j].eps;
    sl[j].du_de = du / de;
    sl[j].de_du = de / du;
  }
}

// Difference quotients of the batched Jacobian (kernel(), jurassic.c:853-855): the radiances of the n perturbed copies
// of the nr rays stand behind the unperturbed ones, rad[(e + 1) * nr + i][id]; dense rows q = i * nd + id, columns e:
// kq[q][e] = (y1 - y0) / h[e].  One lane per element, columns fastest.
__global__ __launch_bounds__(256) void jur_kquot_kernel(long nq, long n, long nrnd, double const *__restrict__ rad,
                                                        double const *__restrict__ h, double *__restrict__ kq) {
  long const t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nq * n) return;
  long const q = t / n, e = t - q * n;
  kq[t] = (rad[(e + 1) * nrnd + q] - rad[q]) / h[e];
}

// longest path of every tile of 64 ray slots (what the tile needs of the transmittance workspace), one wavefront per tile
__global__ __launch_bounds__(256) void jur_tilemax_kernel(int n, int const *__restrict__ np, int *__restrict__ tile_np) {
  int const r = blockIdx.x * blockDim.x + threadIdx.x;
  int m = r < n ? np[r] : 0;
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && (r >> 6) < (n + 63) / 64) tile_np[r >> 6] = m;
}

__global__ __launch_bounds__(256) void jur_raykey_kernel(long nr, long ld, double const *__restrict__ geom,
                                                         double const *__restrict__ atm_time, int atm_np, int by_profile,
                                                         unsigned long long *__restrict__ key, int *__restrict__ id) {
  long const r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nr) return;
  double xo[3], xv[3], e[3];
  geo2cart(geom[1 * ld + r], geom[2 * ld + r], geom[3 * ld + r], xo);      // field k of ray r: geom[k * ld + r]
  geo2cart(geom[4 * ld + r], geom[5 * ld + r], geom[6 * ld + r], xv);
  for (int i = 0; i < 3; i++) e[i] = xv[i] - xo[i];
  double const n = norm3(e);
  double h = norm3(xo) - JUR_RE;
  if (n > 0) {
    double s = 0;
    for (int i = 0; i < 3; i++) s -= xo[i] * e[i] / n;
    if (s > 0) h = sqrt(fmax(xo[0] * xo[0] + xo[1] * xo[1] + xo[2] * xo[2] - s * s, 0.)) - JUR_RE;
  }
  unsigned const fb = __float_as_uint((float)h);
  unsigned const ord = fb ^ ((fb >> 31) ? 0xffffffffu : 0x80000000u);   // unsigned order == float order
  unsigned long long slice = 0;
  if (by_profile) {  // first point of the profile slice with this ray's time stamp (jr_common.h:130-140)
    double const time = geom[r];
    int lo = 0, hi = atm_np - 1;
    while (hi > lo + 1) {
      int const i = (lo + hi) / 2;
      if (atm_time[i] < time) lo = i; else hi = i;
    }
    slice = (unsigned long long)((0 == lo) ? lo : hi);
  }
  key[r] = (slice << 32) | ord;
  id[r] = (int)r;
}

// ---------------------------------------------------------------------------------------
// field-of-view convolution on device arrays (formod_fov, jurassic.c:214-258): for callers that keep the
// radiances in HBM.  One lane per (ray, channel); the neighbours' profile is gathered per lane, the weights are
// walked in the reference's order with its arithmetic (jur_fov.c is the host twin, bit-identical).
// rad0 / tau0: the forward model's results [nr][nd]; rad / tau: the convolved ones [nr][ld].
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void jur_fov_kernel(long nr, int nd, double const *__restrict__ time, double const *__restrict__ vpz,
                                                      double const *__restrict__ rad0, double const *__restrict__ tau0,
                                                      double *__restrict__ rad, double *__restrict__ tau, long ld, int n,
                                                      double const *__restrict__ dz, double const *__restrict__ w,
                                                      int *__restrict__ status) {
  long const i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nr * nd) return;
  long const ir = i / nd;
  int const id = (int)(i - ir * nd);
  double z[2 * JUR_NFOV + 1];
  long src[2 * JUR_NFOV + 1];
  int nz = 0;
  long const lo = ir - JUR_NFOV > 0 ? ir - JUR_NFOV : 0, hi = ir + 1 + JUR_NFOV < nr ? ir + 1 + JUR_NFOV : nr;
  for (long ir2 = lo; ir2 < hi; ir2++)
    if (time[ir2] == time[ir]) { z[nz] = vpz[ir2]; src[nz] = ir2; nz++; }
  if (nz < 2) { atomicOr(status, 2); return; }          // "Cannot apply FOV convolution!" upstream
  double r = 0, t = 0, wsum = 0;
  for (int k = 0; k < n; k++) {
    double const zfov = vpz[ir] + dz[k];
    int const idx = locate_axis(z, nz, zfov);
    double const r0 = rad0[src[idx] * nd + id], r1 = rad0[src[idx + 1] * nd + id];
    double const t0 = tau0[src[idx] * nd + id], t1 = tau0[src[idx + 1] * nd + id];
    r += w[k] * (r0 + (zfov - z[idx]) * (r1 - r0) / (z[idx + 1] - z[idx]));
    t += w[k] * (t0 + (zfov - z[idx]) * (t1 - t0) / (z[idx + 1] - z[idx]));
    wsum += w[k];
  }
  rad[ir * ld + id] = r / wsum;
  tau[ir * ld + id] = t / wsum;
}

// ---------------------------------------------------------------------------------------
// atmosphere regridding (intpol_atm, jurassic.c:675-804): the step in front of the path for atmospheres that
// are not one profile.  One lane per destination point; ip = 1 one profile, 2 the nearest two profiles of a
// track, 3 the distance-weighted mean of a point cloud.  The Cartesian positions of source and destination points
// come from the host (they are what upstream caches), the kernel does the searches and the interpolation.
// src rows: z, lon, lat, p, T, q[ng], k[nw], each [ns]; out rows: p, T, q[ng], k[nw], each [nd_].
// ---------------------------------------------------------------------------------------
struct IntpolArgs {
  int ip, ng, nw, ns, nd_, nx;
  double cx, cz;
  double const *src;        // [5 + ng + nw][ns]
  double const *x1;         // ip 2: [nx][3] profile positions; ip 3: [ns][3] point positions
  int const *idx, *nz;      // ip 2: first point and length of every profile
  double const *dst;        // [3][nd_]: z, lon, lat of the destination points
  double const *x0;         // [nd_][3] their positions
  double *out;              // [2 + ng + nw][nd_]
};

__device__ __forceinline__ double eip_dev(double x0, double y0, double x1, double y1, double x) {
  if ((y0 > 0) && (y1 > 0)) return y0 * exp(log(y1 / y0) / (x1 - x0) * (x - x0));
  return lip(x0, y0, x1, y1, x);
}

// intpol_atm_1d on source points [i0, i0 + n): row r of the source interpolated to z0
__device__ __forceinline__ double ip1d(IntpolArgs const &a, int ipt, int row, double z0) {
  double const *const z = a.src, *const y = a.src + (size_t)row * a.ns;
  return (row == 3) ? eip_dev(z[ipt], y[ipt], z[ipt + 1], y[ipt + 1], z0) : lip(z[ipt], y[ipt], z[ipt + 1], y[ipt + 1], z0);
}

__global__ __launch_bounds__(128) void jur_intpol_kernel(IntpolArgs a) {
  int const id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= a.nd_) return;
  int const nrow = 2 + a.ng + a.nw;                       // p, T, q[], k[]: source rows 3 ..
  double const z0 = a.dst[id], lat0 = a.dst[2 * (size_t)a.nd_ + id];
  double const x0[3] = {a.x0[3 * (size_t)id], a.x0[3 * (size_t)id + 1], a.x0[3 * (size_t)id + 2]};
  auto OUT = [&](int r) -> double & { return a.out[(size_t)r * a.nd_ + id]; };
  if (a.ip == 1) {
    int const ipt = locate_axis(a.src, a.ns, z0);
    for (int r = 0; r < nrow; r++) OUT(r) = ip1d(a, ipt, 3 + r, z0);
  } else if (a.ip == 2) {
    double dhmin0 = 1e99, dhmin1 = 1e99;
    int ix0 = 0, ix1 = 0;
    double const *const lat = a.src + 2 * (size_t)a.ns;
    for (int ix = 0; ix < a.nx; ix++)
      if (fabs(lat0 - lat[a.idx[ix]]) <= 10) {
        double const *const x1 = a.x1 + 3 * (size_t)ix;
        double const dh = (x0[0] - x1[0]) * (x0[0] - x1[0]) + (x0[1] - x1[1]) * (x0[1] - x1[1]) + (x0[2] - x1[2]) * (x0[2] - x1[2]);
        if (dh <= dhmin0) { dhmin1 = dhmin0; ix1 = ix0; dhmin0 = dh; ix0 = ix; }
        else if (dh <= dhmin1) { dhmin1 = dh; ix1 = ix; }
      }
    int const i0 = a.idx[ix0] + locate_axis(a.src + a.idx[ix0], a.nz[ix0], z0);
    int const i1 = a.idx[ix1] + locate_axis(a.src + a.idx[ix1], a.nz[ix1], z0);
    double const *const xa = a.x1 + 3 * (size_t)ix0, *const xb = a.x1 + 3 * (size_t)ix1;
    double const x2 = (xa[0] - xb[0]) * (xa[0] - xb[0]) + (xa[1] - xb[1]) * (xa[1] - xb[1]) + (xa[2] - xb[2]) * (xa[2] - xb[2]);
    double const x = sqrt(x2);
    double const r0 = (dhmin0 - dhmin1 + x2) / (2 * x);
    double const r1 = x - r0;
    double r;
    if (r0 <= 0) r = 0;
    else r = (r1 <= 0) ? 1 : r0 / (r0 + r1);
    for (int k = 0; k < nrow; k++) OUT(k) = (1 - r) * ip1d(a, i0, 3 + k, z0) + r * ip1d(a, i1, 3 + k, z0);
  } else {
    double const rm2 = a.cx * a.cx;
    double wsum = 0;
    for (int r = 0; r < nrow; r++) OUT(r) = 0;
    double const *const lat = a.src + 2 * (size_t)a.ns;
    for (int ipt = 0; ipt < a.ns; ipt++) {
      double const dz = fabs(a.src[ipt] - z0);
      if (dz >= a.cz) continue;
      if (fabs(lat[ipt] - lat0) * 111.13 >= a.cx) continue;
      double const *const x1 = a.x1 + 3 * (size_t)ipt;
      double const dx2 = (x0[0] - x1[0]) * (x0[0] - x1[0]) + (x0[1] - x1[1]) * (x0[1] - x1[1]) + (x0[2] - x1[2]) * (x0[2] - x1[2]);
      if (dx2 >= rm2) continue;
      double const w = (1 - dz / a.cz) * (rm2 - dx2) / (rm2 + dx2);
      wsum += w;
      for (int r = 0; r < nrow; r++) OUT(r) += w * a.src[(size_t)(3 + r) * a.ns + ipt];
    }
    if (wsum >= 1e-6) for (int r = 0; r < nrow; r++) OUT(r) /= wsum;
    else for (int r = 0; r < nrow; r++) OUT(r) = __builtin_nan("");
  }
}

// ---------------------------------------------------------------------------------------
// known-answer hooks (tests): the device functions above on arrays of inputs, one element per lane,
// so that each can be compared with its reference counterpart at thresholds and range edges
// (jr_common.h:239 tau < 1e-9, :295 tau_gas > 1e-50, continuum windows :318,345,367,381, extrapolation
// beyond both ends of a curve) without a ray around it.  Not on the product path.
// ---------------------------------------------------------------------------------------
// chain != 0: ONE lane walks the n inputs in order and carries the warm-start state (br, ia, ib) from element
// to element as jur_ega_kernel carries it from segment to segment -- the result of a look-up must not depend on
// where the previous one left the brackets.
template <bool WARM, bool LDS, bool RCPB, bool REC = false>
__global__ __launch_bounds__(256) void jur_kat_ega_kernel(jur_view_t v, int g, int d, long n, double const *__restrict__ tau,
                                                          double const *__restrict__ t, double const *__restrict__ u,
                                                          double const *__restrict__ p, int chain, double *__restrict__ out) {
  jur_int2 const pd = v.pair[g * v.nd + d];
  PairDesc<LDS> D{v.lvl, v.crv, v.ue + v.pair_e0[g * v.nd + d], v.sl + v.pair_e0[g * v.nd + d], (unsigned)pd.b, 0u, 0u, 0u};
  if constexpr (REC) D.recb = v.rec + v.pair_e0[g * v.nd + d];
  if (pd.a >= 2) stage_pair<LDS, RCPB>(v, pd, D);       // uniform branch; stage_pair ends in a barrier
  unsigned br = 0, ia = 0, ib = 0;
  auto one = [&](long i) {
    if constexpr (WARM) out[i] = ega_eps_warm<LDS, RCPB, false, REC>(v, pd, D, tau[i], t[i], u[i], p[i], br, ia, ib);
    else out[i] = ega_eps_exact<LDS>(v, pd, D, tau[i], t[i], u[i], p[i]);
  };
  if (chain) {
    if (blockIdx.x == 0 && threadIdx.x == 0)
      for (long i = 0; i < n; i++) one(i);
  } else {
    long const i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) one(i);
  }
}

// out[0..3][n]: CO2, H2O, N2, O2 continuum terms of channel d as jur_combine_kernel adds them (0 outside a window)
__global__ __launch_bounds__(256) void jur_kat_continua_kernel(jur_view_t v, int d, long n, double const *__restrict__ p,
                                                               double const *__restrict__ t, double const *__restrict__ q,
                                                               double const *__restrict__ u_co2, double const *__restrict__ u_h2o,
                                                               double *__restrict__ out) {
  long const i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  jur_chan_t const ch = v.chan[d];
  out[i] = ch.co2_on ? ctm_co2(ch, p[i], t[i], u_co2[i]) : 0.;
  double const rt = rcp_t(t[i]);
  Exp2Global const e2t;
  out[n + i] = ch.h2o_on ? ctm_h2o(e2t, ch, p[i], t[i], rt, q[i], u_h2o[i]) : 0.;
  out[2 * n + i] = ch.n2_on ? ctm_n2(e2t, ch, p[i], t[i], rt) : 0.;
  out[3 * n + i] = ch.o2_on ? ctm_o2(e2t, ch, p[i], t[i], rt) : 0.;
}

// what == 0: src = source function at t = a[i]; (rad, tau) updated by one segment with tau_gas = b[i], beta_ds = c[i]
// what == 1: epilogue with surface temperature a[i] and brightness conversion if b[i] != 0; src = source at a[i]
__global__ __launch_bounds__(256) void jur_kat_update_kernel(jur_view_t v, int d, long n, int what, double const *__restrict__ a,
                                                             double const *__restrict__ b, double const *__restrict__ c,
                                                             double *__restrict__ rad, double *__restrict__ tau,
                                                             double *__restrict__ src) {
  long const i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double const *const sr = v.sr + (size_t)d * TBLNS;
  double r = rad[i], tt = tau[i];
  src[i] = planck_src(sr, a[i]);
  if (what == 0) new_obs_step(Exp2Global(), b[i], c[i], src[i], r, tt);
  else ray_epilogue(sr, v.chan[d].nu, a[i], b[i] != 0., r, tt);
  rad[i] = r;
  tau[i] = tt;
}

}  // namespace

extern "C" int jurk_prepare_atm(jur_view_t const *v, double *d_pslope, void *stream) {
  int const block = 256;
  hipLaunchKernelGGL(jur_pslope_kernel, dim3((v->atm_np + block - 1) / block), dim3(block), 0, (hipStream_t)stream,
                     v->atm_np, v->atm_z, v->atm_p, d_pslope);
  return (int)hipGetLastError();
}

static int g_trace_lanes = 0;          // 0: by launch size
extern "C" void jurk_tune_trace(int lanes) { g_trace_lanes = (lanes == 1 || lanes == 4) ? lanes : 0; }

extern "C" int jurk_launch_trace(jur_view_t const *v, jur_chunk_t const *c, void *stream) {
  if (c->n <= 0) return 0;
  int const block = 64;
  // a quad of lanes per ray where the launch stays within the chip's 4096 tracer wavefront slots with it AND the rays
  // have several emitters' columns to share (see jur_trace_lanes_kernel)
  static int const env_lanes = getenv("JUR_TRACE_LANES") ? atoi(getenv("JUR_TRACE_LANES")) : 0;     // A/B switch, read once
  int const forced = g_trace_lanes ? g_trace_lanes : env_lanes;
  int const lanes = forced == 1 || forced == 4 ? forced : ((c->n <= 65536 && v->ng >= 3) ? 4 : 1);
  int const grid = (int)(((long)c->n * lanes + block - 1) / block);
  if (lanes == 4) hipLaunchKernelGGL(jur_trace_lanes_kernel<4>, dim3(grid), dim3(block), 0, (hipStream_t)stream, *v, *c);
  else hipLaunchKernelGGL(jur_trace_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, *v, *c);
  return (int)hipGetLastError();
}

// dynamic LDS beyond 64 KB needs the attribute once per kernel AND per device (models may live on several GPUs of one
// process, their lanes launch from several threads); `raised` holds one bit per device for the caller's kernel set
static int raise_lds_limit(std::mutex &mu, unsigned long long &raised, void const *const *funcs, int nfunc, bool needed) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return (int)hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  if ((raised >> dev) & 1ull) return 0;
  hipError_t e = hipSuccess;
  for (int i = 0; i < nfunc && e == hipSuccess; i++) e = hipFuncSetAttribute(funcs[i], hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  if (e != hipSuccess) { (void)hipGetLastError(); return needed ? (int)e : 0; }
  raised |= 1ull << dev;
  return 0;
}

// Channel groups on a shared (p, T) grid (jur_ega_group_kernel) when the model has any and its tables are strictly
// increasing; workgroup size from the LDS the chain states need (JUR_EGA_BLOCK overrides; JUR_EGA_GROUP=0 switches the
// kernel off).  Returns -1 when the call is not for it.
static int launch_ega_group(jur_view_t const *v, jur_chunk_t const *c, hipStream_t s) {
  static int const env_group = getenv("JUR_EGA_GROUP") ? atoi(getenv("JUR_EGA_GROUP")) : -1;      // A/B switches, read once
  static int const env_block = getenv("JUR_EGA_BLOCK") ? atoi(getenv("JUR_EGA_BLOCK")) : 0;
  static int const env_waves = getenv("JUR_EGA_WAVES") ? atoi(getenv("JUR_EGA_WAVES")) : 0;
  if (!v->ega_items || v->ega_nitems <= 0 || v->ega_nch < 2 || !v->fast_arith || !v->rec || env_group == 0 || getenv("JUR_EGA_NO_LDS")) return -1;
  int const block = (env_block >= 64 && env_block <= 1024 && env_block % 64 == 0) ? env_block : 256;
  int const waves = env_waves >= 6 && env_waves <= 8 ? env_waves : 7;
  size_t const lds = ega_group_lds_bytes(v->max_pair_curves, v->ega_nch, block);
  if (lds > 128 * 1024) return -1;
  static std::mutex mu;
  static unsigned long long raised = 0;
  void const *const funcs[] = {reinterpret_cast<void const *>(&jur_ega_group_kernel<6>), reinterpret_cast<void const *>(&jur_ega_group_kernel<7>),
                               reinterpret_cast<void const *>(&jur_ega_group_kernel<8>)};
  int const e = raise_lds_limit(mu, raised, funcs, 3, lds > 64 * 1024);
  if (e) return e;
  int const nrb = (c->n + block - 1) / block;
  unsigned const grid = xcd_grid(nrb, v->ega_nitems);
  if (waves == 6) hipLaunchKernelGGL((jur_ega_group_kernel<6>), dim3(grid), dim3(block), lds, s, *v, *c, nrb);
  else if (waves == 7) hipLaunchKernelGGL((jur_ega_group_kernel<7>), dim3(grid), dim3(block), lds, s, *v, *c, nrb);
  else hipLaunchKernelGGL((jur_ega_group_kernel<8>), dim3(grid), dim3(block), lds, s, *v, *c, nrb);
  return (int)hipGetLastError();
}

extern "C" int jurk_launch_ega(jur_view_t const *v, jur_chunk_t const *c, void *stream) {
  if (c->n <= 0 || v->ng <= 0) return 0;
  {
    int const e = launch_ega_group(v, c, (hipStream_t)stream);
    if (e >= 0) return e;
  }
  int const block = 256;   // (128 rays per workgroup: 34.6 against 33.7 ms; 64: the staged descriptors limit the occupancy, 45 ms)
  int const nrb = (c->n + block - 1) / block, npair = v->nd * v->ng;
  unsigned const grid = xcd_grid(nrb, npair);
  hipStream_t s = (hipStream_t)stream;
  // LDS copy of one pair's descriptors per workgroup: 16 B x (levels + curves of the largest pair)
  // (+ 8 B x the same counts for the reciprocal bracket widths of strictly increasing tables)
  size_t const lds = (sizeof(jur_lvl_t) + 8) * JUR_TBLNP + (sizeof(jur_crv_t) + 8) * (size_t)v->max_pair_curves;
  bool const use_lds = v->max_pair_curves > 0 && lds <= 48 * 1024 && !getenv("JUR_EGA_NO_LDS");
  bool const rcpb = use_lds && v->fast_arith;
  if (v->sorted_tables) {
    if (rcpb && v->rec) hipLaunchKernelGGL((jur_ega_kernel<true, true, true, true>), dim3(grid), dim3(block), lds, s, *v, *c, nrb);
    else if (rcpb) hipLaunchKernelGGL((jur_ega_kernel<true, true, true>), dim3(grid), dim3(block), lds, s, *v, *c, nrb);
    else if (use_lds) hipLaunchKernelGGL((jur_ega_kernel<true, true, false>), dim3(grid), dim3(block), lds, s, *v, *c, nrb);
    else hipLaunchKernelGGL((jur_ega_kernel<true, false, false>), dim3(grid), dim3(block), 0, s, *v, *c, nrb);
  } else {
    if (use_lds) hipLaunchKernelGGL((jur_ega_kernel<false, true, false>), dim3(grid), dim3(block), lds, s, *v, *c, nrb);
    else hipLaunchKernelGGL((jur_ega_kernel<false, false, false>), dim3(grid), dim3(block), 0, s, *v, *c, nrb);
  }
  return (int)hipGetLastError();
}

// how jur_combine runs: channels per workgroup (0: always one channel per workgroup), segments between barriers,
// smallest launch (rays x channels) that takes the grouped kernel
static int g_combine_group = -1, g_combine_sync = 8, g_combine_forced = 0;
static long g_combine_min_lanes = 1000000L;
extern "C" void jurk_tune_combine(int group, int sync, long min_lanes) {
  g_combine_forced = group >= 0;                 // < 0: back to the defaults and the default rule
  g_combine_group = group < 0 ? 4 : (group > 6 ? 6 : group);   // 6 source-function tables fill the 64 KB of LDS a launch may ask for
  g_combine_sync = sync;
  g_combine_min_lanes = min_lanes;
}

extern "C" int jurk_launch_combine(jur_view_t const *v, jur_chunk_t const *c, void *stream) {
  if (c->n <= 0) return 0;
  int const block = 256;
  int const nrb = (c->n + block - 1) / block;
  unsigned const grid = xcd_grid(nrb, v->nd);
  static std::once_flag env_once;                  // first launch: the environment may override the defaults (A/B
  std::call_once(env_once, [] {                    // switch); once, whichever lane's thread comes first
    if (g_combine_group >= 0) return;              // jur_tune_combine has spoken already
    g_combine_group = getenv("JUR_COMBINE_GROUP") ? atoi(getenv("JUR_COMBINE_GROUP")) : 4;
    g_combine_forced = getenv("JUR_COMBINE_GROUP") != NULL;
    if (g_combine_group > 6) g_combine_group = 6;
    if (getenv("JUR_COMBINE_SYNC")) g_combine_sync = atoi(getenv("JUR_COMBINE_SYNC"));
    if (getenv("JUR_COMBINE_MIN_LANES")) g_combine_min_lanes = atol(getenv("JUR_COMBINE_MIN_LANES"));
  });
  if (g_combine_group < 0) g_combine_group = 4;
  int const group = g_combine_group, sync = g_combine_sync;
  // grouped only when the launch fills the chip several times over -- below that a call is as long as its longest
  // chain, and barriers between wavefronts lengthen it (nadir_1e5, 3e5 lanes: 0.66 against 0.37 ms) -- and, by
  // default, only when the channels fall into full groups of four: groups of two or three and ragged last groups were
  // measured slower than one channel per workgroup (tools/bench_combine_groups.py: 2 channels +7 %, 3 +17 %,
  // 6 = 4 + 2 +27 %).  A group size set through jur_tune_combine / JUR_COMBINE_GROUP is taken as it is.
  // (with many channels the one ragged group at the end does not matter: 2378 = 594 x 4 + 2)
  bool const fits = g_combine_forced || (group == 4 && (v->nd % 4 == 0 || v->nd >= 64));
  if (group > 0 && v->nd > 1 && fits && (long)c->n * v->nd >= g_combine_min_lanes) {
    int const CG = v->nd < group ? v->nd : group, SUB = 8 / CG, W = CG * SUB;
    int const nsb = (c->n + SUB * 64 - 1) / (SUB * 64), ncg = (v->nd + CG - 1) / CG;
    unsigned const g2 = xcd_grid(nsb, ncg);
    int mask = -1;                                 // barrier every 2^k segments, k from JUR_COMBINE_SYNC (<= 0: none)
    if (sync > 0) { mask = 1; while (mask * 2 <= sync) mask *= 2; mask -= 1; }
    hipLaunchKernelGGL(jur_combine_group_kernel, dim3(g2), dim3(64 * W), sizeof(double) * (JUR_TBLNS * CG + 2 + 64), (hipStream_t)stream,
                       *v, *c, nsb, CG, mask);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(jur_combine_kernel, dim3(grid), dim3(block), sizeof(double) * (JUR_TBLNS + 64), (hipStream_t)stream, *v, *c,
                     nrb);
  return (int)hipGetLastError();
}

extern "C" int jurk_launch_kquot(long nq, long n, double const *d_rad, double const *d_h, double *d_kq, void *stream) {
  if (nq <= 0 || n <= 0) return 0;
  hipLaunchKernelGGL(jur_kquot_kernel, dim3((unsigned)((nq * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nq, n, nq, d_rad, d_h, d_kq);
  return (int)hipGetLastError();
}

extern "C" int jurk_tile_max(int n, int const *d_np, int *d_tile_np, void *stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(jur_tilemax_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, d_np, d_tile_np);
  return (int)hipGetLastError();
}

extern "C" int jurk_launch_cg(jur_view_t const *v, jur_chunk_t const *c, double *cgp, double *cgt, double *cgu,
                              void *stream) {
  if (c->n <= 0 || v->ng <= 0) return 0;
  int const block = 256;                                   // 4 pencils per workgroup
  long const pencils = (long)c->n * v->ng;
  unsigned const grid = (unsigned)((pencils * 64 + block - 1) / block);
  hipLaunchKernelGGL(jur_cg_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, *v, *c, cgp, cgt, cgu);
  return (int)hipGetLastError();
}

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" long jurk_sort_tmp_bytes(long nr) {
  size_t cub = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, cub, (unsigned long long const *)nullptr, (unsigned long long *)nullptr,
                                         (int const *)nullptr, (int *)nullptr, (int)nr) != hipSuccess)
    cub = 64 * (size_t)nr + (1 << 20);
  return (long)(2 * align_up(8 * (size_t)nr) + align_up(4 * (size_t)nr) + align_up(cub) + 256);
}

extern "C" int jurk_sort_rays(jur_view_t const *v, int by_profile, long nr, double const *d_geom, long ld, int *d_order, void *tmp,
                              long tmp_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  size_t const seg8 = align_up(8 * (size_t)nr), seg4 = align_up(4 * (size_t)nr);
  char *base = (char *)tmp;
  unsigned long long *key_in = (unsigned long long *)base, *key_out = (unsigned long long *)(base + seg8);
  int *id_in = (int *)(base + 2 * seg8);
  void *cub_tmp = base + 2 * seg8 + seg4;
  size_t cub_bytes = (size_t)tmp_bytes - 2 * seg8 - seg4;
  int const block = 256;
  hipLaunchKernelGGL(jur_raykey_kernel, dim3((unsigned)((nr + block - 1) / block)), dim3(block), 0, s, nr, ld, d_geom,
                     v->atm_time, v->atm_np, by_profile, key_in, id_in);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  e = hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, key_in, key_out, id_in, d_order, (int)nr, 0,
                                         by_profile ? 64 : 32, s);
  return (int)e;
}

// ---- known-answer hooks ----
// mode 0: the reference's bisections; 1: warm-started searches, descriptors from global memory; 2: ... from LDS;
// 3: LDS + reciprocal bracket widths (the variant the bench runs).  Returns hipErrorInvalidValue (1) when the
// tables do not admit the mode.
extern "C" int jurk_kat_ega(jur_view_t const *v, int g, int d, long n, double const *tau, double const *t, double const *u,
                            double const *p, int mode, int chain, double *out, void *stream) {
  if (n <= 0) return 0;
  size_t const lds = (sizeof(jur_lvl_t) + 8) * JUR_TBLNP + (sizeof(jur_crv_t) + 8) * (size_t)v->max_pair_curves;
  bool const lds_ok = v->max_pair_curves > 0 && lds <= 48 * 1024;
  if ((mode >= 1 && !v->sorted_tables) || (mode >= 2 && !lds_ok) || (mode == 3 && !v->strict_tables) || mode < 0 || mode > 3)
    return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  dim3 const grid(chain ? 1u : (unsigned)((n + 255) / 256)), block(256);
  if (mode == 0) {
    if (lds_ok) hipLaunchKernelGGL((jur_kat_ega_kernel<false, true, false>), grid, block, lds, s, *v, g, d, n, tau, t, u, p, chain, out);
    else hipLaunchKernelGGL((jur_kat_ega_kernel<false, false, false>), grid, block, 0, s, *v, g, d, n, tau, t, u, p, chain, out);
  } else if (mode == 1) hipLaunchKernelGGL((jur_kat_ega_kernel<true, false, false>), grid, block, 0, s, *v, g, d, n, tau, t, u, p, chain, out);
  else if (mode == 2) hipLaunchKernelGGL((jur_kat_ega_kernel<true, true, false>), grid, block, lds, s, *v, g, d, n, tau, t, u, p, chain, out);
  else if (v->rec) hipLaunchKernelGGL((jur_kat_ega_kernel<true, true, true, true>), grid, block, lds, s, *v, g, d, n, tau, t, u, p, chain, out);   // what the batched kernel runs
  else hipLaunchKernelGGL((jur_kat_ega_kernel<true, true, true>), grid, block, lds, s, *v, g, d, n, tau, t, u, p, chain, out);
  return (int)hipGetLastError();
}

extern "C" int jurk_kat_continua(jur_view_t const *v, int d, long n, double const *p, double const *t, double const *q,
                                 double const *u_co2, double const *u_h2o, double *out, void *stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(jur_kat_continua_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *v, d, n, p, t,
                     q, u_co2, u_h2o, out);
  return (int)hipGetLastError();
}

extern "C" int jurk_kat_update(jur_view_t const *v, int d, long n, int what, double const *a, double const *b, double const *c,
                               double *rad, double *tau, double *src, void *stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(jur_kat_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *v, d, n, what,
                     a, b, c, rad, tau, src);
  return (int)hipGetLastError();
}

// ---- fused kernel for small calls ----
// LDS bytes jur_pencil_kernel needs for RB rays per workgroup (0: the configuration does not fit)
extern "C" long jurk_pencil_lds_bytes(jur_view_t const *v, int RB) {
  long const bytes = 8 * pen_lds_doubles(v->nd, v->ng, v->nw, RB);
  return bytes <= 96 * 1024 ? bytes : 0;
}

// room for the profile slab: the longest slice of the atmosphere, if that stays inside 32 KB
static int pencil_atm_cap(jur_view_t const *v) {
  long const nrow = 7 + v->ng + v->nw;
  static int const off = getenv("JUR_PENCIL_NO_ATM_LDS") ? 1 : 0;      // A/B switch, read once
  if (!v->atm_sorted || v->atm_maxslice < 2 || off) return 0;
  return (nrow * v->atm_maxslice * 8 <= 32 * 1024) ? v->atm_maxslice : 0;
}

extern "C" int jurk_launch_pencil(jur_view_t const *v, jur_chunk_t const *c, int RB, void *stream) {
  if (c->n <= 0) return 0;
  long const lds = jurk_pencil_lds_bytes(v, RB);
  if (RB < 1 || RB > 64 || lds <= 0) return (int)hipErrorInvalidValue;
  int const npair = v->nd * v->ng;
  // up to 16 rays per workgroup the tracer wavefront has four lanes per ray (one refraction probe each) and, for
  // sorted tables, every chain four lanes (one corner curve each)
  static bool const no_quad = getenv("JUR_PENCIL_NO_QUAD") != nullptr;   // A/B switch, read once
  bool const quad = RB <= 16 && !no_quad;
  int NE = (RB * npair * ((quad && v->sorted_tables) ? 4 : 1) + 63) / 64, NC = (RB * v->nd + 63) / 64;
  NE = NE < 1 ? 1 : (NE > PEN_MAXE ? PEN_MAXE : NE);
  NC = NC < 1 ? 1 : (NC > PEN_MAXC ? PEN_MAXC : NC);
  dim3 const grid((unsigned)((c->n + RB - 1) / RB)), block((unsigned)(64 * (1 + NE + NC)));
  hipStream_t s = (hipStream_t)stream;
  int const atm_cap = pencil_atm_cap(v);
  long const lds_all = lds + 8L * (7 + v->ng + v->nw) * atm_cap;
  {
    static std::mutex mu;
    static unsigned long long raised = 0;                 // one bit per device, read and written under mu
    void const *const funcs[] = {reinterpret_cast<void const *>(&jur_pencil_kernel<true, true>), reinterpret_cast<void const *>(&jur_pencil_kernel<true, false>),
                                 reinterpret_cast<void const *>(&jur_pencil_kernel<false, true>), reinterpret_cast<void const *>(&jur_pencil_kernel<false, false>)};
    int const e = raise_lds_limit(mu, raised, funcs, 4, lds_all > 64 * 1024);
    if (e) return e;
  }
  if (v->sorted_tables) {
    if (quad) hipLaunchKernelGGL((jur_pencil_kernel<true, true>), grid, block, (size_t)lds_all, s, *v, *c, RB, NE, NC, atm_cap);
    else hipLaunchKernelGGL((jur_pencil_kernel<true, false>), grid, block, (size_t)lds_all, s, *v, *c, RB, NE, NC, atm_cap);
  } else {
    if (quad) hipLaunchKernelGGL((jur_pencil_kernel<false, true>), grid, block, (size_t)lds_all, s, *v, *c, RB, NE, NC, atm_cap);
    else hipLaunchKernelGGL((jur_pencil_kernel<false, false>), grid, block, (size_t)lds_all, s, *v, *c, RB, NE, NC, atm_cap);
  }
  return (int)hipGetLastError();
}

extern "C" int jurk_fill_records(jur_ue_t const *ue, jur_sl_t const *sl, jur_rec_t *rec, long long n, void *stream) {
  long long const nb = (n + 255) / 256;
  hipLaunchKernelGGL(jur_records_kernel, dim3((unsigned)(nb < 65536 ? (nb > 0 ? nb : 1) : 65536)), dim3(256), 0, (hipStream_t)stream, n, ue, sl, rec);
  return (int)hipGetLastError();
}

extern "C" int jurk_fill_slopes(jur_ue_t const *ue, jur_sl_t *sl, long long n, void *stream) {
  long long const nb = (n + 255) / 256;
  hipLaunchKernelGGL(jur_slopes_kernel, dim3((unsigned)(nb < 65536 ? (nb > 0 ? nb : 1) : 65536)), dim3(256), 0, (hipStream_t)stream, n, ue, sl);
  return (int)hipGetLastError();
}

extern "C" int jurk_launch_fov(long nr, int nd, double const *time, double const *vpz, double const *rad0, double const *tau0,
                               double *rad, double *tau, long ld, int n, double const *dz, double const *w, int *status, void *stream) {
  if (nr <= 0) return 0;
  long const total = nr * nd;
  hipLaunchKernelGGL(jur_fov_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nr, nd, time, vpz, rad0,
                     tau0, rad, tau, ld, n, dz, w, status);
  return (int)hipGetLastError();
}

extern "C" int jurk_launch_intpol(int ip, int ng, int nw, int ns, int nd_, int nx, double cx, double cz, double const *src,
                                  double const *x1, int const *idx, int const *nz, double const *dst, double const *x0, double *out,
                                  void *stream) {
  if (nd_ <= 0) return 0;
  IntpolArgs const a{ip, ng, nw, ns, nd_, nx, cx, cz, src, x1, idx, nz, dst, x0, out};
  hipLaunchKernelGGL(jur_intpol_kernel, dim3((unsigned)((nd_ + 127) / 128)), dim3(128), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
