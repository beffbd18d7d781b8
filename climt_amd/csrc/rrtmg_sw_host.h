// rrtmg_sw_host.h -- host-side setup shared by the SW launch path: table descriptor construction and
// the per-call scalar part of inatm_sw (Earth-Sun distance, solar-variability multipliers).
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "rrtmg_sw_device.h"
#include "rrtmg_tables.h"

namespace rrtmg {

// Fill the SwTab offsets from the reduced-table registry.  Scalar Rayleigh coefficients are
// replicated per g-point so the kernel reads rayl[ig] for every band.
inline bool build_sw_tab(TableSet &ts, SwTab &T, std::string &err) {
  const std::vector<int32_t> *ngc = ts.ints("sw/wvn/ngc"), *ngs = ts.ints("sw/wvn/ngs");
  if (!ngc || !ngs) { err = "sw/wvn/ngc missing"; return false; }
  auto off = [&](const std::string &n, bool required) -> long {
    long o = ts.off(n);
    if (o < 0 && required) err = "reduced table '" + n + "' missing";
    return o < 0 ? 0 : o;
  };
  const int kNg[kSwNBand] = {SwBandCfg<16>::ng, SwBandCfg<17>::ng, SwBandCfg<18>::ng, SwBandCfg<19>::ng, SwBandCfg<20>::ng,
                             SwBandCfg<21>::ng, SwBandCfg<22>::ng, SwBandCfg<23>::ng, SwBandCfg<24>::ng, SwBandCfg<25>::ng,
                             SwBandCfg<26>::ng, SwBandCfg<27>::ng, SwBandCfg<28>::ng, SwBandCfg<29>::ng};
  for (int b = 0; b < kSwNBand; ++b) {
    SwBandTab &B = T.b[b];
    const std::string p = "sw/kg" + std::to_string(16 + b) + "/";
    B.ng = (*ngc)[b];
    B.gs = b == 0 ? 0 : (*ngs)[b - 1];
    if (B.ng != kNg[b]) { err = "reduced g-point count of band " + std::to_string(16 + b) + " differs from the compiled-in one"; return false; }
    B.nfor = 4;
    { auto it = ts.reg.find(p + "forref"); if (it != ts.reg.end()) B.nfor = (int)it->second.dims[0]; }
    // ONE g-point-fastest slab [nrows][ng] per band holding all the per-g-point tables of taumol (the blob stores the
    // k-tables [ng][row], the Rayleigh and extra-absorber tables [row][ng] already); SwBandTab::r_* = first rows
    std::vector<double> slab;
    auto append = [&](const std::string &n, bool required, bool transposed) -> int {
      auto it = ts.reg.find(p + n);
      if (it == ts.reg.end()) { if (required) err = "reduced table '" + p + n + "' missing"; return 0; }
      const int r0 = (int)(slab.size() / B.ng);
      if (it->second.n == 1) {   // a band constant (rayl of most bands): replicated per g-point
        slab.insert(slab.end(), (size_t)B.ng, ts.flat[(size_t)it->second.off]);
        return r0;
      }
      const long o = it->second.off, rows = it->second.n / B.ng;
      slab.resize(slab.size() + (size_t)rows * B.ng);
      for (int ig = 0; ig < B.ng; ++ig)
        for (long r = 0; r < rows; ++r)
          slab[(size_t)(r0 + r) * B.ng + ig] = ts.flat[(size_t)o + (transposed ? (size_t)ig * rows + r : (size_t)r * B.ng + ig)];
      return r0;
    };
    B.r_absa = append("absa", false, true); B.r_absb = append("absb", false, true);
    B.r_self = append("selfref", false, true); B.r_forr = append("forref", false, true);
    B.sflux = off(p + "sfluxref", true); B.irr = off(p + "irradnce", true);
    B.fac = off(p + "facbrght", true); B.sns = off(p + "snsptdrk", true);
    { auto it = ts.reg.find(p + "sfluxref"); B.nsrc = it->second.dims.size() > 1 ? (int)it->second.dims[1] : 1; }
    B.r_raylb = 0; B.r_ex1 = 0; B.r_ex2 = 0;
    const int band = 16 + b;
    if (band == 24) {
      B.r_rayl = append("rayla", true, false); B.r_raylb = append("raylb", true, false);
      B.r_ex1 = append("abso3a", true, false); B.r_ex2 = append("abso3b", true, false);
    } else {
      B.r_rayl = append("rayl", true, false);
      if (band == 20) { B.r_ex1 = append("absch4", true, false); }
      if (band == 25) { B.r_ex1 = append("abso3a", true, false); B.r_ex2 = append("abso3b", true, false); }
      if (band == 29) { B.r_ex1 = append("absco2", true, false); B.r_ex2 = append("absh2o", true, false); }
    }
    if (!err.empty()) return false;
    B.nrows = (int)(slab.size() / B.ng);
    if (B.nrows > kSwSlabMaxRows) { err = "band " + std::to_string(band) + " table slab has more rows than kSwSlabMaxRows"; return false; }
    B.slab = ts.add(p + "slab_g", slab.data(), (long)slab.size(), {(uint32_t)B.nrows, (uint32_t)B.ng});
    if (!err.empty()) return false;
  }
  T.preflog = off("sw/ref/preflog", true); T.tref = off("sw/ref/tref", true); T.exp_tbl = off("sw/tbl/exp_tbl", true);
  T.extliq1 = off("sw/cld/extliq1", true); T.ssaliq1 = off("sw/cld/ssaliq1", true); T.asyliq1 = off("sw/cld/asyliq1", true);
  T.extice2 = off("sw/cld/extice2", true); T.ssaice2 = off("sw/cld/ssaice2", true); T.asyice2 = off("sw/cld/asyice2", true);
  T.extice3 = off("sw/cld/extice3", true); T.ssaice3 = off("sw/cld/ssaice3", true); T.asyice3 = off("sw/cld/asyice3", true);
  T.fdlice3 = off("sw/cld/fdlice3", true);
  T.abari = off("sw/cld/abari", true); T.bbari = off("sw/cld/bbari", true); T.cbari = off("sw/cld/cbari", true);
  T.dbari = off("sw/cld/dbari", true); T.ebari = off("sw/cld/ebari", true); T.fbari = off("sw/cld/fbari", true);
  T.wavenum2 = off("sw/wvn/wavenum2", true);
  T.rsrtaua = off("sw/aer/rsrtaua", true); T.rsrpiza = off("sw/aer/rsrpiza", true); T.rsrasya = off("sw/aer/rsrasya", true);
  T.heatfac = ts.heatfac;
  // work items: chunks of 4 (then 2) consecutive g-points of a band; launch order heaviest first
  const int nspa[kSwNBand] = {9, 9, 9, 9, 1, 9, 9, 1, 9, 1, 0, 1, 9, 1};
  {
    int &n = T.nitem;
    n = 0;
    double cost[kSwMaxItem];
    for (int b = 0; b < kSwNBand; ++b) {
      int ig = 0;
      while (ig < T.b[b].ng) {
        const int g = (T.b[b].ng - ig >= 4) ? 4 : 2;
        if (n >= kSwMaxItem) { err = "too many work items"; return false; }
        cost[n] = (nspa[b] == 9 ? 1.0 : 0.6) + g * 1.0;
        T.item[n] = b | (ig << 8) | (g << 16) | ((T.b[b].gs + ig) << 20);
        T.sched[n] = n;
        ++n;
        ig += g;
      }
    }
    for (int i = 1; i < n; ++i)   // stable insertion sort, descending cost
      for (int j = i; j > 0 && cost[T.sched[j]] > cost[T.sched[j - 1]]; --j) { const int t = T.sched[j]; T.sched[j] = T.sched[j - 1]; T.sched[j - 1] = t; }
  }
  return err.empty();
}

// earth_sun(idn) -- rrtmg_sw_rad.nomcica.f90:819-843
inline double sw_earth_sun(int idn, double pi) {
  const double gamma = 2.0 * pi * (idn - 1) / 365.0;
  return 1.000110 + .034221 * cos(gamma) + .001289 * sin(gamma) + .000719 * cos(2.0 * gamma) + .000077 * sin(2.0 * gamma);
}

// Scalar part of inatm_sw: adjflux and the solar-variability multipliers (rrtmg_sw_rad.nomcica.f90:1196-1428).
// The reference calls inatm_sw INSIDE its column loop and rescales the facular/sunspot amplitudes `indsolvar`
// IN PLACE there (:1199-1215): when they differ from 1, column k sees amplitudes that were already pulled k times
// towards the solar-cycle weight, and the caller's array comes back changed.  That is reproduced: `svar_col`
// receives per-column (svar_f, svar_s, svar_i) -- [3][ncol] -- whenever the multipliers differ between columns
// (it stays empty otherwise and the scalars in `d` apply), and `indsolvar` is updated as the reference leaves it.
// mgavgcyc / sbavgcyc: the 132-entry NRLSSI2 mean-cycle index tables (isolvar = 1).
inline int sw_scalar_setup(SwDev &d, int ncol, int isolvar, double adjes, int dyofyr, double scon, double solcycfrac,
                           const double *bndsolvar, double *indsolvar, const double *mgavgcyc, const double *sbavgcyc,
                           std::vector<double> &svar_col, std::string &err) {
  const double rrsw_scon = (double)1.36822e+03f;   // parrrsw.f90:115 -- a default-real (single precision) literal
  const double Iint = 1360.37, Fint = 0.996047, Sint = -0.511590;
  const double Foffset = 0.14959542, Soffset = 0.00066696, svar_f_avg = 0.1568113, svar_s_avg = 909.21910;
  const int nsolfrac = 132;
  double solvar[kSwNBand];
  for (int b = 0; b < kSwNBand; ++b) { solvar[b] = 1.0; d.svar_b[b] = 1.0; }
  d.svar_f = d.svar_s = d.svar_i = 1.0;
  d.isolvar = isolvar;
  svar_col.clear();
  if (isolvar < -1 || isolvar > 3) { err = "isolvar=" + std::to_string(isolvar) + " is not a solar variability method"; return 20; }
  if (isolvar == 1 && (!mgavgcyc || !sbavgcyc)) { err = "solar-cycle index tables missing from the shortwave data file"; return 3; }
  double adjflx = adjes;
  if (dyofyr > 0) adjflx = sw_earth_sun(dyofyr, d.k.pi);

  // interpolated mean-cycle indices (isolvar = 1), the same for every column
  double svar_f_0 = 0.0, svar_s_0 = 0.0;
  if (isolvar == 1) {
    if (solcycfrac <= 0.0) { svar_f_0 = mgavgcyc[0]; svar_s_0 = sbavgcyc[0]; }
    else if (solcycfrac >= 1.0) { svar_f_0 = mgavgcyc[nsolfrac - 1]; svar_s_0 = sbavgcyc[nsolfrac - 1]; }
    else {
      const int sfid = (int)floor(solcycfrac * (nsolfrac - 1)) + 1;
      const double nsfm1_inv = 1.0 / (nsolfrac - 1);
      const double fraclo = (sfid - 1) * nsfm1_inv, frachi = sfid * nsfm1_inv;
      const double intfrac = (solcycfrac - fraclo) / (frachi - fraclo);
      svar_f_0 = mgavgcyc[sfid - 1] + intfrac * (mgavgcyc[sfid] - mgavgcyc[sfid - 1]);
      svar_s_0 = sbavgcyc[sfid - 1] + intfrac * (sbavgcyc[sfid] - sbavgcyc[sfid - 1]);
    }
  }
  // per-column multipliers from the amplitudes as column `k` sees them
  double i1 = indsolvar ? indsolvar[0] : 1.0, i2 = indsolvar ? indsolvar[1] : 1.0;
  auto multipliers = [&](double a1, double a2, double &f, double &s, double &i) {
    f = s = i = 1.0;
    if (scon == 0.0) {
      if (isolvar == 1) { f = a1 * (svar_f_0 - Foffset) / (svar_f_avg - Foffset); s = a2 * (svar_s_0 - Soffset) / (svar_s_avg - Soffset); i = 1.0; }
      if (isolvar == 2) { f = (a1 - Foffset) / (svar_f_avg - Foffset); s = (a2 - Soffset) / (svar_s_avg - Soffset); i = 1.0; }
    } else if (scon > 0.0) {
      if (isolvar == 0) { const double r = scon / (Fint + Sint + Iint); f = s = i = r; }
      if (isolvar == 1) {
        i = (scon - (a1 * Fint + a2 * Sint)) / Iint;
        f = a1 * (svar_f_0 - Foffset) / (svar_f_avg - Foffset);
        s = a2 * (svar_s_0 - Soffset) / (svar_s_avg - Soffset);
      }
    }
  };
  const bool varies = (i1 != 1.0 || i2 != 1.0) && solcycfrac >= 0.0 && solcycfrac <= 1.0;
  const bool per_column = varies && (isolvar == 1 || (isolvar == 2 && scon == 0.0));
  if (per_column) svar_col.assign((size_t)3 * ncol, 1.0);
  for (int k = 0; k < (varies ? ncol : 1); ++k) {
    if (i1 != 1.0 || i2 != 1.0) {   // rrtmg_sw_rad.nomcica.f90:1199-1215, once per column
      if (solcycfrac >= 0.0 && solcycfrac < 0.0229) {
        const double wgt = (solcycfrac + 1.0 - 0.3817) / (1.0229 - 0.3817);
        i1 = i1 + wgt * (1.0 - i1); i2 = i2 + wgt * (1.0 - i2);
      }
      if (solcycfrac >= 0.0229 && solcycfrac <= 0.3817) {
        const double wgt = (solcycfrac - 0.0229) / (0.3817 - 0.0229);
        i1 = 1.0 + wgt * (i1 - 1.0); i2 = 1.0 + wgt * (i2 - 1.0);
      }
      if (solcycfrac > 0.3817 && solcycfrac <= 1.0) {
        const double wgt = (solcycfrac - 0.3817) / (1.0229 - 0.3817);
        i1 = i1 + wgt * (1.0 - i1); i2 = i2 + wgt * (1.0 - i2);
      }
    }
    double f, s, i;
    multipliers(i1, i2, f, s, i);
    if (per_column) { svar_col[k] = f; svar_col[(size_t)ncol + k] = s; svar_col[(size_t)2 * ncol + k] = i; }
    if (k == 0) { d.svar_f = f; d.svar_s = s; d.svar_i = i; }
  }
  if (indsolvar) { indsolvar[0] = i1; indsolvar[1] = i2; }

  if (scon == 0.0) {
    if (isolvar == -1 && bndsolvar) for (int b = 0; b < kSwNBand; ++b) solvar[b] = bndsolvar[b];
    if (isolvar == 3) for (int b = 0; b < kSwNBand; ++b) { solvar[b] = bndsolvar ? bndsolvar[b] : 1.0; d.svar_b[b] = solvar[b]; }
  } else if (scon > 0.0) {
    if (isolvar == -1) for (int b = 0; b < kSwNBand; ++b) solvar[b] = bndsolvar ? bndsolvar[b] * scon / rrsw_scon : scon / rrsw_scon;
    if (isolvar == 3) {
      const double c = Fint + Sint + Iint;
      for (int b = 0; b < kSwNBand; ++b) { solvar[b] = bndsolvar ? bndsolvar[b] * scon / c : scon / c; d.svar_b[b] = solvar[b]; }
    }
  }
  d.adjflux = adjflx;
  for (int b = 0; b < kSwNBand; ++b) d.adjflux_b[b] = (isolvar < 0) ? adjflx * solvar[b] : adjflx;
  return 0;
}

}  // namespace rrtmg
