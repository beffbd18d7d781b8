// rrtmg_zenith.hip -- the producer on the upstream side of the shortwave path: climt's `Instellation` component
// (zenith angle from latitude, longitude and time; /root/reference/climt/_components/instellation/component.py:85-191)
// on the device, so that a radiation step can stay device-resident.
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

}  // namespace rrtmg

using namespace rrtmg;

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
