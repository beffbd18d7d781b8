// TEST INFRASTRUCTURE ONLY -- host emulation of the longwave DEVICE functions (see emu_sw.hip).
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../climt_amd/csrc/rrtmg_lw_device.h"
#include "../../climt_amd/csrc/rrtmg_lw_host.h"
#include "../../climt_amd/csrc/rrtmg_sw_device.h"
#include "../../include/rrtmg_hip.h"

using namespace rrtmg;

namespace rrtmg {
void mt_mask_host(int ncol, int nlay, int nsub, int icld, int seed, const double *cldfr, std::vector<uint64_t> &mask, int nw, int col0 = 0, int ncol_total = 0);
}

// the clear-sky variant for cloud-free columns, as the device picks it per tile
static bool emu_lw_cloudy(const LwDev &d, int col) {
  bool cld = false;
  if (d.icld >= 1 && d.cldfr) for (int l = 0; l < d.nlay; ++l) cld = cld || d.cldfr[(size_t)l * d.ncol + col] > 0.0;
  return cld;
}
static void emu_lw_solve(const LwDev &d, const LwTab &T) {
  std::vector<double> scr((size_t)LF_N * d.nlay * 4);
  for (int slot = 0; slot < T.nitem; ++slot)
    for (int col = 0; col < d.ncol; ++col) {
      LwPartSink sink = lw_part_sink(d, slot, col);
      const bool cld = emu_lw_cloudy(d, col);
      if (cld && !d.mcica && d.icld >= 2) lw_solve_item<true, true>(d, T, T.item[slot], col, scr.data(), 1, sink);
      else if (cld) lw_solve_item<true, false>(d, T, T.item[slot], col, scr.data(), 1, sink);
      else lw_solve_item<false, false>(d, T, T.item[slot], col, scr.data(), 1, sink);
    }
}

extern "C" int emu_lw_fluxes(const rrtmg_lw_args *a, const char *blob_path, double cpdair, const double *consts, char *errbuf, int errlen) {
  auto fail = [&](int code, const std::string &m) { if (errbuf) { strncpy(errbuf, m.c_str(), errlen - 1); errbuf[errlen - 1] = 0; } return code; };
  Blob blob;
  std::string err;
  if (!blob.load(blob_path, err)) return fail(3, err);
  TableSet ts;
  Constants k{};
  k.pi = consts[0]; k.grav = consts[1]; k.planck = consts[2]; k.boltz = consts[3]; k.clight = consts[4];
  k.avogad = consts[5]; k.alosmt = consts[6]; k.gascon = consts[7]; k.sbcnst = consts[8]; k.secdy = consts[9];
  if (!build_tables(blob, "lw", cpdair, k.grav, k.secdy, ts, err)) return fail(3, err);
  LwTab T{};
  if (!build_lw_tab(ts, T, err)) return fail(3, err);
  T.t = ts.flat.data();
  const int N = a->ncol, L = a->nlay;
  const size_t nl = (size_t)N * L, nl1 = (size_t)N * (L + 1);
  LwDev d{};
  d.ncol = N; d.nlay = L; d.icld = a->icld;
  if (d.icld < 0 || d.icld > 3) d.icld = 2;
  d.idrv = a->idrv ? 1 : 0;
  d.inflag = a->inflglw; d.iceflag = a->iceflglw; d.liqflag = a->liqflglw; d.mcica = a->mcica ? 1 : 0;
  d.k = k;
  d.fluxfac = (2.0 * asin(1.0)) * 2.e4;
  d.play = a->play; d.plev = a->plev; d.tlay = a->tlay; d.tlev = a->tlev; d.tsfc = a->tsfc; d.h2o = a->h2ovmr; d.o3 = a->o3vmr;
  d.co2 = a->co2vmr; d.ch4 = a->ch4vmr; d.n2o = a->n2ovmr; d.o2 = a->o2vmr; d.cfc11 = a->cfc11vmr; d.cfc12 = a->cfc12vmr;
  d.cfc22 = a->cfc22vmr; d.ccl4 = a->ccl4vmr; d.emis = a->emis; d.tauaer = a->tauaer;
  std::vector<double> tlev_host;
  if (!d.tlev) {   // interface temperatures not given: the interpolation the library does on the device (util.py:89-142)
    tlev_host.resize(nl1);
    for (int c = 0; c < N; ++c) {
      tlev_host[c] = d.tsfc[c];
      tlev_host[(size_t)L * N + c] = d.tlay[(size_t)(L - 1) * N + c];
      for (int lev = 1; lev < L; ++lev) {
        const double lp1 = log(d.play[(size_t)lev * N + c]), lp0 = log(d.play[(size_t)(lev - 1) * N + c]);
        const double w = (log(d.plev[(size_t)lev * N + c]) - lp1) / (lp0 - lp1);
        const double m1 = d.tlay[(size_t)lev * N + c], m0 = d.tlay[(size_t)(lev - 1) * N + c];
        tlev_host[(size_t)lev * N + c] = m1 - w * (m1 - m0);
      }
    }
    d.tlev = tlev_host.data();
  }
  const bool clouds = d.icld >= 1;
  if (clouds) { d.cldfr = a->cldfr; d.taucld = a->taucld; d.cicewp = a->cicewp; d.cliqwp = a->cliqwp; d.reice = a->reice; d.reliq = a->reliq; }
  std::vector<std::vector<double>> keep;
  auto wd = [&](size_t n) { keep.emplace_back(n, 0.0); return keep.back().data(); };
  d.prep = wd(lw_prep_size(N, L)); d.secdiff = wd((size_t)N * 16);
  std::vector<int32_t> laytrop(N), ncb(N, 1);
  d.laytrop = laytrop.data(); d.ncbands = ncb.data();
  if (clouds) d.ctau = wd(nl * 16);
  d.nw = (L + 63) / 64;
  std::vector<uint64_t> mask, anym;
  const int nk = d.idrv ? 6 : 4;
  d.col0 = 0; d.pcols = N;
  d.part = wd((size_t)kLwNGpt * nk * nl1);
  d.uflx = a->uflx; d.dflx = a->dflx; d.hr = a->hr; d.uflxc = a->uflxc; d.dflxc = a->dflxc; d.hrc = a->hrc;
  d.duflx_dt = a->duflx_dt; d.duflxc_dt = a->duflxc_dt;
  int errflag = 0;
  d.err = &errflag;
  for (int c = 0; c < N; ++c) { for (int l = 0; l < L; ++l) lw_prep_layer(d, T, c, l); lw_prep_column(d, T, c); }
  if (clouds) {
    if (!d.mcica) {
      for (int c = 0; c < N; ++c) lw_cloud_column(d, T, c);
      if (d.icld >= 2) { d.mr = wd(lw_mr_size(N, L)); for (int c = 0; c < N; ++c) lw_mr_column(d, c); }
    } else {
      for (int l = 0; l < L; ++l) for (int c = 0; c < N; ++c) lw_cloudmc_layer(d, T, c, l);
      mask.assign((size_t)kLwNGpt * d.nw * N, 0);
      anym.assign((size_t)d.nw * N, 0);
      d.mask = mask.data(); d.anymask = anym.data();
      if (a->cldfmcl) {
        for (int g = 0; g < kLwNGpt; ++g) for (int l = 0; l < L; ++l) for (int c = 0; c < N; ++c)
          if (a->cldfmcl[((size_t)l * N + c) * kLwNGpt + g] > 1.e-12) mask[((size_t)g * d.nw + (l >> 6)) * N + c] |= 1ull << (l & 63);
      } else if (a->irng == 0) {
        for (int c = 0; c < N; ++c) kiss_mask_column(N, L, kLwNGpt, d.icld, a->permuteseed, d.play, d.cldfr, d.mask, d.nw, d.err, c);
      } else {
        mt_mask_host(N, L, kLwNGpt, d.icld, a->permuteseed, a->cldfr, mask, d.nw, a->shard_col0, a->shard_ncol);
        d.mask = mask.data();
      }
      for (int c = 0; c < N; ++c) lw_anymask_column(d, c);
    }
  }
  emu_lw_solve(d, T);
  for (int lev = 0; lev <= L; ++lev) for (int c = 0; c < N; ++c) lw_flux_level(d, T, c, lev, T.nitem, emu_lw_cloudy(d, c));
  for (int l = 0; l < L; ++l) for (int c = 0; c < N; ++c) lw_heat_layer(d, T, c, l);
  if (errflag) return fail(errflag, "device-side error flag " + std::to_string(errflag));
  return 0;
}

// stage check: taug / fracs of every (layer, g-point) of ONE column through lw_prep_column + lw_taug<>
template <int BAND>
static void emu_taug_band(const LwDev &d, const LwTab &T, double *taug, double *fracs) {
  const int L = d.nlay;
  for (int ig = 0; ig < T.b[BAND - 1].ng; ig += 2)
    for (int l = 0; l < L; ++l) {
      LwLayerIn s;
      lw_load_layer(d, 0, l, s);
      V<2> fr;
      const V<2> tg = lw_taug<BAND, 2>(T, s, (l + 1) <= d.laytrop[0], ig, fr);
      for (int j = 0; j < 2; ++j) {
        taug[(size_t)(T.b[BAND - 1].gs + ig + j) * L + l] = tg[j];    // Fortran (nlay, ngpt) order
        fracs[(size_t)(T.b[BAND - 1].gs + ig + j) * L + l] = fr[j];
      }
    }
}

extern "C" int emu_lw_taumol(const rrtmg_lw_args *a, const char *blob_path, double cpdair, const double *consts, double *taug, double *fracs) {
  Blob blob;
  std::string err;
  if (!blob.load(blob_path, err)) return 3;
  TableSet ts;
  Constants k{};
  k.pi = consts[0]; k.grav = consts[1]; k.avogad = consts[5]; k.secdy = consts[9];
  if (!build_tables(blob, "lw", cpdair, k.grav, k.secdy, ts, err)) return 3;
  LwTab T{};
  if (!build_lw_tab(ts, T, err)) return 3;
  T.t = ts.flat.data();
  const int L = a->nlay;
  LwDev d{};
  d.ncol = 1; d.nlay = L; d.k = k;
  d.play = a->play; d.plev = a->plev; d.tlay = a->tlay; d.tlev = a->tlev; d.tsfc = a->tsfc; d.h2o = a->h2ovmr; d.o3 = a->o3vmr;
  d.co2 = a->co2vmr; d.ch4 = a->ch4vmr; d.n2o = a->n2ovmr; d.o2 = a->o2vmr; d.cfc11 = a->cfc11vmr; d.cfc12 = a->cfc12vmr;
  d.cfc22 = a->cfc22vmr; d.ccl4 = a->ccl4vmr; d.emis = a->emis;
  std::vector<std::vector<double>> keep;
  auto wd = [&](size_t n) { keep.emplace_back(n, 0.0); return keep.back().data(); };
  d.prep = wd(lw_prep_size(1, L)); d.secdiff = wd(16);
  std::vector<int32_t> laytrop(1);
  d.laytrop = laytrop.data();
  int errflag = 0;
  d.err = &errflag;
  for (int l = 0; l < L; ++l) lw_prep_layer(d, T, 0, l);
  lw_prep_column(d, T, 0);
  emu_taug_band<1>(d, T, taug, fracs); emu_taug_band<2>(d, T, taug, fracs); emu_taug_band<3>(d, T, taug, fracs); emu_taug_band<4>(d, T, taug, fracs);
  emu_taug_band<5>(d, T, taug, fracs); emu_taug_band<6>(d, T, taug, fracs); emu_taug_band<7>(d, T, taug, fracs); emu_taug_band<8>(d, T, taug, fracs);
  emu_taug_band<9>(d, T, taug, fracs); emu_taug_band<10>(d, T, taug, fracs); emu_taug_band<11>(d, T, taug, fracs); emu_taug_band<12>(d, T, taug, fracs);
  emu_taug_band<13>(d, T, taug, fracs); emu_taug_band<14>(d, T, taug, fracs); emu_taug_band<15>(d, T, taug, fracs); emu_taug_band<16>(d, T, taug, fracs);
  return laytrop[0];
}

// reduced table read-back for the reduction tests: builds tables on the host only
extern "C" long emu_get_table(const char *which, const char *blob_path, double cpdair, const char *name, double *out, long cap) {
  Blob blob;
  std::string err;
  if (!blob.load(blob_path, err)) return -3;
  TableSet ts;
  if (!build_tables(blob, which, cpdair, 9.80665, 86400.0, ts, err)) return -3;
  auto it = ts.reg.find(name);
  if (it == ts.reg.end()) return -1;
  if (out) { if (cap < it->second.n) return -2; memcpy(out, ts.flat.data() + it->second.off, (size_t)it->second.n * 8); }
  return it->second.n;
}
