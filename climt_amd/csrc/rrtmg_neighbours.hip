// rrtmg_neighbours.hip -- the steps either side of the radiation path, so that a radiation step can stay
// device-resident:
//   upstream   rrtmg_hip_zenith_angle     climt Instellation          (instellation/component.py:85-191)
//              rrtmg_hip_solar_insolation climt BergerSolarInsolation (berger_solar_insolation.py:671-676; orbital series on the host)
//   downstream rrtmg_hip_slab_surface     climt SlabSurface           (slab_surface.py:440-517)
//
// Instellation: zenith angle from latitude, longitude and time.
//   host  : sun position for the time of the call -- obliquity (:138-152), ecliptic longitude of the sun (:155-179),
//           declination / right ascension (:90-99), Greenwich mean sidereal time (:182-191); scalars, same arithmetic
//   device: one thread per column -- local hour angle, cos(mu) clamped to [-1, 1], arccos clamped to [-pi/2, pi/2]
//           (:113-133; the reference clamps night-side columns to pi/2)
#include <cmath>

#include "rrtmg_ctx.h"

namespace rrtmg {

struct SunPos { double sin_dec, cos_dec, ra, gmst; };

static double deg2rad(double x) { return x * (M_PI / 180.0); }

static SunPos sun_position(double t) {
  const double eps = deg2rad(23.0 + 26.0 / 60 + 21.406 / 3600.0 -
                             (46.836769 * t - 0.0001831 * (t * t) + 0.00200340 * (t * t * t) - 0.576e-6 * (t * t * t * t) -
                              4.34e-8 * (t * t * t * t * t)) / 3600.0);
  const double mean_anomaly = deg2rad(357.52910 + 35999.05030 * t - 0.0001559 * t * t - 0.00000048 * t * t * t);
  const double mean_longitude = deg2rad(280.46645 + 36000.76983 * t + 0.0003032 * (t * t));
  const double d_l = deg2rad((1.914600 - 0.004817 * t - 0.000014 * (t * t)) * sin(mean_anomaly) +
                             (0.019993 - 0.000101 * t) * sin(2 * mean_anomaly) + 0.000290 * sin(3 * mean_anomaly));
  const double eclon = mean_longitude + d_l;
  const double x = cos(eclon), y = cos(eps) * sin(eclon), z = sin(eps) * sin(eclon);
  const double r = sqrt(1.0 - z * z);
  const double declination = atan2(z, r);
  SunPos s;
  s.sin_dec = sin(declination); s.cos_dec = cos(declination);
  s.ra = 2.0 * atan2(y, (x + r));
  // "6.2 * 10e-6" is the reference's literal (component.py:186)
  const double theta = 67310.54841 + t * (876600.0 * 3600 + 8640184.812866 + t * (0.093104 - t * 6.2 * 10e-6));
  double g = fmod(deg2rad(theta / 240.0), 2.0 * M_PI);
  if (g < 0) g += 2.0 * M_PI;   // numpy's % is non-negative for a positive modulus
  s.gmst = g;
  return s;
}

__global__ void __launch_bounds__(256) zenith_kernel(int n, const double *lat_deg, const double *lon_deg, SunPos s, double *zenith) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double pi = 3.14159265358979323846;
  const double lat = lat_deg[i] * (pi / 180.0);
  const double h_angle = s.gmst + lon_deg[i] * (pi / 180.0) - s.ra;
  double cos_mu = sin(lat) * s.sin_dec + cos(lat) * s.cos_dec * cos(h_angle);
  if (cos_mu > 1.0) cos_mu = 1.0; else if (cos_mu < -1.0) cos_mu = -1.0;
  double z = acos(cos_mu);
  if (z > pi / 2.0) z = pi / 2.0; else if (z < -pi / 2.0) z = -pi / 2.0;
  zenith[i] = z;
}

// climt BergerSolarInsolation, per-column part (berger_solar_insolation.py:671-676): hour angle from the fraction of
// the day and the longitude, cos(mu), zenith angle and insolation.  The latitude VALUE (degrees) goes into sin/cos as
// it is -- the reference does so (:673) and its golden caches pin it.
__global__ void __launch_bounds__(256) insolation_kernel(int n, const double *lat, const double *lon, double sin_delta, double cos_delta,
                                                         double fractional_day, double irradiance, double *zenith, double *insolation) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double pi = 3.14159265358979323846;
  const double H = 2 * pi * (fractional_day + lon[i] / 360.0);
  const double cos_mu = sin(lat[i]) * sin_delta - cos(lat[i]) * cos_delta * cos(H);
  zenith[i] = acos(cos_mu);
  insolation[i] = irradiance * cos_mu;
}

// Downstream of the radiation path: kernel of climt's SlabSurface (slab_surface.py:440-517, default configuration),
// one thread per column.  The four flux arguments are the SURFACE rows of the radiation outputs -- row 0 of the
// [level][column] arrays, so a device-resident radiation step feeds it without a copy.
struct SlabArgs {
  const double *sw_down, *lw_down, *sw_up, *lw_up, *lh, *sh;
  const int32_t *area_type;   // land 0, land_ice 1, sea 2, sea_ice 3
  const double *up_heat_soil, *heat_flux_sea_ice, *sea_water_dens, *surf_dens, *heat_cap_soil, *surf_therm_cap;
  const double *ocean_mix_thick, *soil_layer_thick, *ocean_heat_transport;
  double *tend_ts, *depth;
};
__global__ void __launch_bounds__(256) slab_surface_kernel(int n, SlabArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double net_heat_flux = a.sw_down[i] + a.lw_down[i] - a.sw_up[i] - a.lw_up[i] - a.sh[i] - a.lh[i];
  const int at = a.area_type[i];
  const bool land_mask = (at == 0) || (at == 1), sea_mask = (at == 2) || (at == 3), land_ice_mask = at == 1, sea_ice_mask = at == 3;
  if (land_ice_mask) net_heat_flux = -a.up_heat_soil[i];
  else if (sea_ice_mask) net_heat_flux = a.heat_flux_sea_ice[i];
  if (sea_mask && !sea_ice_mask) net_heat_flux = net_heat_flux + a.ocean_heat_transport[i];
  double final_dens, d;
  if (sea_mask) { final_dens = a.sea_water_dens[i]; d = a.ocean_mix_thick[i]; }
  else { final_dens = a.surf_dens[i]; d = land_mask ? a.soil_layer_thick[i] : 0.0; }
  const double final_therm_cap = land_mask ? a.heat_cap_soil[i] : a.surf_therm_cap[i];
  a.depth[i] = d;
  const double heat_cap_slab = (final_dens * d) * final_therm_cap;
  double val = heat_cap_slab != 0 ? net_heat_flux / heat_cap_slab : 0.0;
  if (land_ice_mask || sea_ice_mask) val = 0.0;
  a.tend_ts[i] = val;
}

}  // namespace rrtmg

using namespace rrtmg;

extern "C" int rrtmg_hip_slab_surface(rrtmg_ctx *ctx, int ncol, int memspace, const rrtmg_slab_args *h) {
  if (!ctx) return RRTMG_ERR_ARG;
  if (ncol <= 0 || !h) return ctx->fail(RRTMG_ERR_ARG, "slab_surface: bad argument");
  const double *const in[15] = {h->sw_down, h->lw_down, h->sw_up, h->lw_up, h->lh, h->sh, h->up_heat_soil, h->heat_flux_sea_ice,
                                h->sea_water_dens, h->surf_dens, h->heat_cap_soil, h->surf_therm_cap, h->ocean_mix_thick,
                                h->soil_layer_thick, h->ocean_heat_transport};
  for (int k = 0; k < 15; ++k)
    if (!in[k]) return ctx->fail(RRTMG_ERR_ARG, "slab_surface: input array %d is NULL", k);
  if (!h->area_type || !h->tend_ts || !h->depth) return ctx->fail(RRTMG_ERR_ARG, "slab_surface: area_type / output array is NULL");
  int rc = ctx_prepare_device(ctx);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const size_t bytes = (size_t)ncol * sizeof(double);
  const double *dev[15];
  const int32_t *dat = h->area_type;
  double *dt = h->tend_ts, *dd = h->depth;
  if (memspace == 0) {
    double *slab = (double *)ctx->buf("slab.io", 17 * bytes);
    int32_t *at = (int32_t *)ctx->buf("slab.at", (size_t)ncol * 4);
    if (!slab || !at) return ctx->status;
    for (int k = 0; k < 15; ++k) {
      RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(slab + (size_t)k * ncol, in[k], bytes, hipMemcpyHostToDevice, s));
      dev[k] = slab + (size_t)k * ncol;
    }
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(at, h->area_type, (size_t)ncol * 4, hipMemcpyHostToDevice, s));
    dat = at; dt = slab + (size_t)15 * ncol; dd = slab + (size_t)16 * ncol;
  } else {
    for (int k = 0; k < 15; ++k) dev[k] = in[k];
  }
  SlabArgs a{dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], dat, dev[6], dev[7], dev[8], dev[9], dev[10], dev[11], dev[12], dev[13], dev[14], dt, dd};
  hipLaunchKernelGGL(slab_surface_kernel, dim3((ncol + 255) / 256), dim3(256), 0, s, ncol, a);
  RRTMG_HIP_CHECK(ctx, hipGetLastError());
  if (memspace == 0) {
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(h->tend_ts, dt, bytes, hipMemcpyDeviceToHost, s));
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(h->depth, dd, bytes, hipMemcpyDeviceToHost, s));
  }
  if (ctx->deferred && memspace == 1) return RRTMG_OK;
  RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(s));
  return RRTMG_OK;
}

extern "C" int rrtmg_hip_solar_insolation(rrtmg_ctx *ctx, int ncol, int memspace, const double *lat, const double *lon, double sin_delta,
                                          double cos_delta, double fractional_day, double irradiance, double *zenith, double *insolation) {
  if (!ctx) return RRTMG_ERR_ARG;
  if (ncol <= 0 || !lat || !lon || !zenith || !insolation) return ctx->fail(RRTMG_ERR_ARG, "solar_insolation: bad argument");
  int rc = ctx_prepare_device(ctx);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const size_t bytes = (size_t)ncol * sizeof(double);
  const double *dlat = lat, *dlon = lon;
  double *dz = zenith, *di = insolation;
  if (memspace == 0) {
    double *a = (double *)ctx->buf("zen.lat", bytes), *b = (double *)ctx->buf("zen.lon", bytes);
    dz = (double *)ctx->buf("zen.out", bytes); di = (double *)ctx->buf("zen.ins", bytes);
    if (!a || !b || !dz || !di) return ctx->status;
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(a, lat, bytes, hipMemcpyHostToDevice, s));
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(b, lon, bytes, hipMemcpyHostToDevice, s));
    dlat = a; dlon = b;
  }
  hipLaunchKernelGGL(insolation_kernel, dim3((ncol + 255) / 256), dim3(256), 0, s, ncol, dlat, dlon, sin_delta, cos_delta, fractional_day,
                     irradiance, dz, di);
  RRTMG_HIP_CHECK(ctx, hipGetLastError());
  if (memspace == 0) {
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(zenith, dz, bytes, hipMemcpyDeviceToHost, s));
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(insolation, di, bytes, hipMemcpyDeviceToHost, s));
  }
  if (ctx->deferred && memspace == 1) return RRTMG_OK;
  RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(s));
  return RRTMG_OK;
}

extern "C" int rrtmg_hip_zenith_angle(rrtmg_ctx *ctx, int ncol, int memspace, const double *lat_deg, const double *lon_deg,
                                      double julian_centuries, double *zenith) {
  if (!ctx) return RRTMG_ERR_ARG;
  if (ncol <= 0 || !lat_deg || !lon_deg || !zenith) return ctx->fail(RRTMG_ERR_ARG, "zenith_angle: bad argument");
  int rc = ctx_prepare_device(ctx);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const size_t bytes = (size_t)ncol * sizeof(double);
  const double *dlat = lat_deg, *dlon = lon_deg;
  double *dz = zenith;
  if (memspace == 0) {
    double *a = (double *)ctx->buf("zen.lat", bytes), *b = (double *)ctx->buf("zen.lon", bytes);
    dz = (double *)ctx->buf("zen.out", bytes);
    if (!a || !b || !dz) return ctx->status;
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(a, lat_deg, bytes, hipMemcpyHostToDevice, s));
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(b, lon_deg, bytes, hipMemcpyHostToDevice, s));
    dlat = a; dlon = b;
  }
  hipLaunchKernelGGL(zenith_kernel, dim3((ncol + 255) / 256), dim3(256), 0, s, ncol, dlat, dlon, sun_position(julian_centuries), dz);
  RRTMG_HIP_CHECK(ctx, hipGetLastError());
  if (memspace == 0) RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(zenith, dz, bytes, hipMemcpyDeviceToHost, s));
  if (ctx->deferred && memspace == 1) return RRTMG_OK;   // ordered before later shortwave work on the same stream
  RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(s));
  return RRTMG_OK;
}

// ---- glue of the device-resident radiation step --------------------------------------------------------------------------
// The host-side numpy of the component classes between the kernels -- so that a model loop never leaves HBM:
//   interface temperatures     get_interface_values            climt/_core/util.py:89-142 (lw/component.py:378-384)
//   q -> volume mixing ratio   mass_to_volume_mixing_ratio     climt/_core/util.py:47-86  (q * 28.964 / 18.02: two roundings)
//   cos(zenith)                np.cos(state['zenith_angle'])   sw/component.py:567
//   tendency sums and the Adams-Bashforth update (sympl AdamsBashforth around the components, tests/test_components.py:123-160)
namespace rrtmg {
__global__ void __launch_bounds__(256) interface_values_kernel(int ncol, int nlay, const double *mid, const double *surf, const double *pmid,
                                                               const double *pint, double *out) {
  const int col = blockIdx.x * 256 + threadIdx.x, lev = blockIdx.y;
  if (col >= ncol) return;
  const long N = ncol;
  double v;
  if (lev == 0) v = surf[col];
  else if (lev == nlay) v = mid[(long)(nlay - 1) * N + col];
  else {
    const double lp1 = log(pmid[(long)lev * N + col]), lp0 = log(pmid[(long)(lev - 1) * N + col]);
    const double weight = (log(pint[(long)lev * N + col]) - lp1) / (lp0 - lp1);
    const double m1 = mid[(long)lev * N + col], m0 = mid[(long)(lev - 1) * N + col];
    v = m1 - weight * (m1 - m0);
  }
  out[(long)lev * N + col] = v;
}
// op 0: out = alpha * a (+ beta * b);  1: out = cos(a);  2: out = a * alpha / beta (the two roundings of util.py:86)
__global__ void __launch_bounds__(256) elementwise_kernel(int op, long n, const double *a, const double *b, double alpha, double beta, double *out) {
#pragma clang fp contract(off)   // the host numpy this replaces rounds every product and sum
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (op == 0) { const double x = alpha * a[i]; out[i] = b ? x + beta * b[i] : x; }
  else if (op == 1) out[i] = cos(a[i]);
  else out[i] = a[i] * alpha / beta;
}
// x_out = x + dt * (w0 f0 + w1 f1 + w2 f2 + w3 f3), the sum formed left to right as sympl's stepper does (0 + w0 f0 + ...)
__global__ void __launch_bounds__(256) ab_step_kernel(long n, int order, const double *x, const double *f0, const double *f1, const double *f2,
                                                      const double *f3, double w0, double w1, double w2, double w3, double dt, double *out) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double incr = 0.0 + w0 * f0[i];
  if (order > 1) incr = incr + w1 * f1[i];
  if (order > 2) incr = incr + w2 * f2[i];
  if (order > 3) incr = incr + w3 * f3[i];
  out[i] = x[i] + dt * incr;
}
// strided 2-d block copies (rrtmg_hip_copy_blocks): grid (column tiles of 256, row, block); rows of a block beyond its own
// count and columns beyond its own width fall out.  Each row is a contiguous run on both sides: coalesced 8-byte accesses,
// non-temporal (the gathered buffer is read once, the destination is not read by this kernel).
__global__ void __launch_bounds__(256) copy_blocks_kernel(const int64_t *desc, const double *src, double *dst) {
  const int64_t *q = desc + 6 * (long)blockIdx.z;
  const long r = blockIdx.y, c = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= q[2] || c >= q[3]) return;
  __builtin_nontemporal_store(__builtin_nontemporal_load(src + q[0] + r * q[4] + c), dst + q[1] + r * q[5] + c);
}
// inputs of host-pointer calls that did not have to cross PCIe (rrtmg_host_inputs.h): a fill, and the unit factor the caller
// would have applied on the host (one rounding per operation, as numpy: contraction off)
__global__ void __launch_bounds__(256) fill_kernel(double *p, size_t n, double value) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = value;
}
__global__ void __launch_bounds__(256) scale_kernel(double *p, size_t n, double mul, double div) {
#pragma clang fp contract(off)
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double x = p[i] * mul;
  if (div != 0.0) x = x / div;
  p[i] = x;
}
void launch_fill(hipStream_t s, double *p, size_t n, double value) {
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, value);
}
void launch_scale(hipStream_t s, double *p, size_t n, double mul, double div) {
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, mul, div);
}
void launch_interface_values(hipStream_t s, int ncol, int nlay, const double *mid, const double *surf, const double *pmid, const double *pint, double *out) {
  hipLaunchKernelGGL(interface_values_kernel, dim3((ncol + 255) / 256, nlay + 1), dim3(256), 0, s, ncol, nlay, mid, surf, pmid, pint, out);
}
}  // namespace rrtmg

extern "C" int rrtmg_hip_interface_values(rrtmg_ctx *ctx, int ncol, int nlay, const double *mid, const double *surf, const double *pmid,
                                          const double *pint, double *out) {
  if (!ctx) return RRTMG_ERR_ARG;
  if (ncol <= 0 || nlay <= 0 || !mid || !surf || !pmid || !pint || !out) return ctx->fail(RRTMG_ERR_ARG, "interface_values: bad argument");
  int rc = ctx_prepare_device(ctx);
  if (rc) return rc;
  hipLaunchKernelGGL(interface_values_kernel, dim3((ncol + 255) / 256, nlay + 1), dim3(256), 0, ctx->stream, ncol, nlay, mid, surf, pmid, pint, out);
  RRTMG_HIP_CHECK(ctx, hipGetLastError());
  if (!ctx->deferred) RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return RRTMG_OK;
}
extern "C" int rrtmg_hip_elementwise(rrtmg_ctx *ctx, int op, long n, const double *a, const double *b, double alpha, double beta, double *out) {
  if (!ctx) return RRTMG_ERR_ARG;
  if (n <= 0 || op < 0 || op > 2 || !a || !out) return ctx->fail(RRTMG_ERR_ARG, "elementwise: bad argument");
  int rc = ctx_prepare_device(ctx);
  if (rc) return rc;
  hipLaunchKernelGGL(elementwise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, op, n, a, b, alpha, beta, out);
  RRTMG_HIP_CHECK(ctx, hipGetLastError());
  if (!ctx->deferred) RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return RRTMG_OK;
}
extern "C" int rrtmg_hip_ab_step(rrtmg_ctx *ctx, long n, int order, const double *x, const double *const *f, const double *w, double dt, double *out) {
  if (!ctx) return RRTMG_ERR_ARG;
  if (n <= 0 || order < 1 || order > 4 || !x || !f || !w || !out) return ctx->fail(RRTMG_ERR_ARG, "ab_step: bad argument");
  for (int k = 0; k < order; ++k)
    if (!f[k]) return ctx->fail(RRTMG_ERR_ARG, "ab_step: tendency %d is NULL", k);
  int rc = ctx_prepare_device(ctx);
  if (rc) return rc;
  hipLaunchKernelGGL(ab_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, order, x, f[0], order > 1 ? f[1] : nullptr,
                     order > 2 ? f[2] : nullptr, order > 3 ? f[3] : nullptr, w[0], order > 1 ? w[1] : 0.0, order > 2 ? w[2] : 0.0, order > 3 ? w[3] : 0.0, dt, out);
  RRTMG_HIP_CHECK(ctx, hipGetLastError());
  if (!ctx->deferred) RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return RRTMG_OK;
}
// direction 0: the longwave stream waits for everything enqueued so far on the main stream (inputs prepared there);
// direction 1: the main stream waits for the longwave stream (its outputs are consumed there).  Host does not block.
extern "C" int rrtmg_hip_order_streams(rrtmg_ctx *ctx, int direction) {
  if (!ctx || !ctx->stream || !ctx->stream_lw) return RRTMG_ERR_ARG;
  hipStream_t from = direction == 0 ? ctx->stream : ctx->stream_lw, to = direction == 0 ? ctx->stream_lw : ctx->stream;
  RRTMG_HIP_CHECK(ctx, hipEventRecord(ctx->sync_ev[direction ? 1 : 0], from));
  RRTMG_HIP_CHECK(ctx, hipStreamWaitEvent(to, ctx->sync_ev[direction ? 1 : 0], 0));
  return RRTMG_OK;
}

extern "C" int rrtmg_hip_copy_blocks(rrtmg_ctx *ctx, int nblk, const int64_t *desc, long max_rows, long max_cols, const double *src, double *dst,
                                     void *stream) {
  if (!ctx) return RRTMG_ERR_ARG;
  if (nblk <= 0 || nblk > 65535 || max_rows <= 0 || max_rows > 65535 || max_cols <= 0 || !desc || !src || !dst)
    return ctx->fail(RRTMG_ERR_ARG, "copy_blocks: bad argument");
  int rc = ctx_prepare_device(ctx);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  hipLaunchKernelGGL(copy_blocks_kernel, dim3((unsigned)((max_cols + 255) / 256), (unsigned)max_rows, (unsigned)nblk), dim3(256), 0, s, desc, src, dst);
  RRTMG_HIP_CHECK(ctx, hipGetLastError());
  if (!ctx->deferred && !stream) RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(s));
  return RRTMG_OK;
}
