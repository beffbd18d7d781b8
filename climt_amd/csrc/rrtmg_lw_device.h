// rrtmg_lw_device.h -- RRTMG longwave hot path as per-thread device functions (gfx950).
//
// Decomposition (see rrtmg_sw_device.h for the rationale of the wave = 64 columns x one g-point map):
//   lw_prep_column   one thread per column : inatm + setcoef (interpolation indices/fractions,
//                    column amounts, pwvcm -> secdiff(band)), laytrop
//   lw_cloud_column  one thread per column : cldprop (non-McICA; the reference's layer-order dependent
//                    ncbands bookkeeping is reproduced)   /  lw_cloudmc_layer: cldprmc band optics
//   lw_solve_thread  one thread per (column, g-point): downward sweep (taumol for the layer, Planck
//                    terms, no-scattering recurrence; layer state spilled to a [layer][field][lane]
//                    scratch slab), surface reflection, upward sweep.
//   lw_flux_level / lw_heat_layer  band / g-point integration, flux scaling; heating rates.
//
// Reference followed (climt/_lib/rrtmg_lw/): rrtmg_lw_rad.nomcica.f90:80-569,:572-900 (driver, inatm),
// rrtmg_lw_rad.f90, rrtmg_lw_setcoef.f90:31-415, rrtmg_lw_taumol.f90:31-3147 (taugb1..16),
// rrtmg_lw_cldprop.f90:31-276, rrtmg_lw_cldprmc.f90:32-254, rrtmg_lw_rtrn.f90:261-587,
// rrtmg_lw_rtrnmc.f90:32-576.
#pragma once
#include "rrtmg_common.h"

namespace rrtmg {

constexpr int kLwNBand = 16;
constexpr int kLwNGpt = 140;

struct LwBandTab {
  int ng, gs;
  int nfraca, nfracb;      // Planck-fraction mixtures (1, 9 / 1, 5)
  // Every per-g-point table of the band lives in ONE g-point-fastest slab [nrows][ng] at T.t + slab (built at init);
  // r_* = first row of each table in it.  A work item's slice of the slab -- columns ig0 .. ig0+G-1, [nrows][G] --
  // is what the solve kernel stages in LDS, rows unchanged.
  long slab;
  int nrows;
  int r_absa, r_absb, r_self, r_forr, r_fraca, r_fracb;
  int r_ma[3];             // lower-atmosphere minor-gas tables [19 * nm] rows
  int r_mb[2];             // upper-atmosphere minor-gas tables
  int r_x[2];              // cross-section rows (ccl4 | cfc11adj, cfc12 | cfc12, cfc22adj)
};
constexpr int kLwSlabMaxRows = 2056;   // band 3: 585 + 1175 + 10 + 4 + 171 + 95 + 9 + 5 rows (checked at init)

// work items of the solve kernel: see SwTab (packed band | ig0 << 8 | G << 16 | first g-point << 20)
constexpr int kLwMaxItem = 72;

struct LwTab {
  const double *t;
  LwBandTab b[kLwNBand];
  int nitem;
  int32_t item[kLwMaxItem], sched[kLwMaxItem];
  long chirat;             // [CR_N][59] species ratios chi_mls(x, j)/chi_mls(y, j)
  long preflog, tref, chi_mls, totplnk, totplk16, totplnkderiv, totplk16deriv;
  long exp_tbl, tau_tbl, tfn_tbl, delwave;
  long abscld1, absice0, absice1, absice2, absice3, absliq0, absliq1;
  double heatfac;
};

struct LwDev {
  int ncol, nlay;
  int icld, idrv, inflag, iceflag, liqflag, mcica;
  Constants k;
  double fluxfac;
  // inputs
  const double *play, *plev, *tlay, *tlev, *tsfc, *h2o, *o3, *co2, *ch4, *n2o, *o2;
  const double *cfc11, *cfc12, *cfc22, *ccl4, *emis;
  const double *cldfr, *taucld, *cicewp, *cliqwp, *reice, *reliq, *tauaer;
  // prep products: ONE slab [tile][layer][LP_N fields][64 lanes] -- a wavefront reads the rows of its
  // (tile, layer) at constant offsets from a single address (lw_prep_off), see enum LwPrepField
  double *prep;
  int32_t *laytrop;    // [col]
  double *secdiff;     // [16][col]
  // clouds
  double *ctau;        // [16][lay][col]  (nomcica: taucloud(lay, ib); mcica: per-band cloudy-sub-column tau)
  int32_t *ncbands;    // [col]
  double *mr;          // rtrnmr overlap factors [tile][index 0..L+1][MR_N][64] (non-McICA icld >= 2), see lw_mr_column
  int32_t *tile_cld;   // [tile] 1 if any column of the 64-column tile has cldfr > 0 (selects the solve kernel variant)
  const int32_t *tlist, *tcnt;   // the chunk's tiles by variant, compacted (see SwDev)
  int tcap;
  int32_t *ncloudy;    // number of tiles with tile_cld set, counted by the preparation kernels (rrtmg_ctx::CallHint) ...
  int32_t *hint_out;   // ... and where the call's LAST integration launch leaves it for the host (page-locked; nullptr in the others)
  uint64_t *mask;      // [140][nw][col]
  uint64_t *anymask;   // [nw][col]   OR over the sub-columns (icldlyr of rtrnmc)
  int nw;
  double *scratch;
  double *part;        // [item][nk][nlay+1][pcols], nk = 4 (+2 with idrv): radlu, radld, radclru, radclrd summed over the item
  int col0, pcols;     // column chunk the solve / flux kernels are working on (scratch and part are per chunk)
  RRTMG_PROFILE_FIELDS
  int *err;
  double *uflx, *dflx, *hr, *uflxc, *dflxc, *hrc, *duflx_dt, *duflxc_dt;
};

// rows of the prep slab; LP_IDX holds jp | jt<<8 | jt1<<12 | indself<<16 | indfor<<20 | indminor<<24 as an
// (exact) double, LP_PAVEL a copy of the layer pressure
enum LwPrepField { LP_FAC00 = 0, LP_FAC01, LP_FAC10, LP_FAC11, LP_SELFFAC, LP_SELFFRAC, LP_FORFAC, LP_FORFRAC,
                   LP_MINORFRAC, LP_SCALEMINOR, LP_SCALEMINORN2, LP_COLH2O, LP_COLCO2, LP_COLO3, LP_COLN2O,
                   LP_COLCO, LP_COLCH4, LP_COLO2, LP_COLBRD, LP_COLDRY, LP_WX1, LP_WX2, LP_WX3, LP_WX4, LP_PAVEL,
                   LP_IDX, LP_N };
RRTMG_HD long lw_prep_off(int nlay, int col, int lay) {
  return ((long)(col >> 6) * nlay + lay) * (LP_N * 64) + (col & 63);
}
RRTMG_HD size_t lw_prep_size(int ncol, int nlay) { return (size_t)((ncol + 63) / 64) * nlay * LP_N * 64; }

// ------------------------------------------------------------------------------------------
// inatm (rrtmg_lw_rad.nomcica.f90:744-880) + setcoef indices (rrtmg_lw_setcoef.f90:253-411)
// ------------------------------------------------------------------------------------------
// layer part: one thread per (column, layer)
// keep (optional): keep[0] = coldry, keep[1] = h2o vmr, keep[2] = 1.0 if the layer is in the lower atmosphere -- what the column
// part reads back, so that a fused kernel can hand it over in LDS
RRTMG_HD void lw_prep_layer(const LwDev &d, const LwTab &T, int col, int l, double *keep = nullptr, int keep_stride = 0) {
  const int L = d.nlay, N = d.ncol;
  const double *preflog = T.t + T.preflog, *tref = T.t + T.tref;
  const double amd = 28.9660, amw = 18.0160;
  const double stpfac = 296.0 / 1013.0;
  int laytrop = 0;   // 1 if this layer is in the lower atmosphere
  {
    const long i = (long)l * N + col;
    const double pz0 = d.plev[i], pz1 = d.plev[i + N];
    const double pavel = d.play[i], tavel = d.tlay[i];
    const double v1 = d.h2o[i], v2 = d.co2[i], v3 = d.o3[i], v4 = d.n2o[i], v6 = d.ch4[i], v7 = d.o2[i];
    const double amm = (1.0 - v1) * amd + v1 * amw;
    const double coldry = (pz0 - pz1) * 1.e3 * d.k.avogad / (1.e2 * d.k.grav * amm * (1.0 + v1));
    // summol over molecules 2..7 (wkl(5) = CO is zero)
    double summol = 0.0;
    summol = summol + v2; summol = summol + v3; summol = summol + v4; summol = summol + 0.0; summol = summol + v6; summol = summol + v7;
    const double wbroad = coldry * (1.0 - summol);
    const double w1 = coldry * v1, w2 = coldry * v2, w3 = coldry * v3, w4 = coldry * v4, w5 = coldry * 0.0, w6 = coldry * v6, w7 = coldry * v7;
    double *q = d.prep + lw_prep_off(L, col, l);
    q[LP_WX1 * 64] = coldry * (d.ccl4 ? d.ccl4[i] : 0.0) * 1.e-20;
    q[LP_WX2 * 64] = coldry * (d.cfc11 ? d.cfc11[i] : 0.0) * 1.e-20;
    q[LP_WX3 * 64] = coldry * (d.cfc12 ? d.cfc12[i] : 0.0) * 1.e-20;
    q[LP_WX4 * 64] = coldry * (d.cfc22 ? d.cfc22[i] : 0.0) * 1.e-20;

    const double plog = log(pavel);
    int jp = (int)(36.0 - 5 * (plog + 0.04));
    if (jp < 1) jp = 1; else if (jp > 58) jp = 58;
    const double fp = 5.0 * (preflog[jp - 1] - plog);
    int jt = (int)(3.0 + (tavel - tref[jp - 1]) / 15.0);
    if (jt < 1) jt = 1; else if (jt > 4) jt = 4;
    const double ft = ((tavel - tref[jp - 1]) / 15.0) - (double)(jt - 3);
    int jt1 = (int)(3.0 + (tavel - tref[jp]) / 15.0);
    if (jt1 < 1) jt1 = 1; else if (jt1 > 4) jt1 = 4;
    const double ft1 = ((tavel - tref[jp]) / 15.0) - (double)(jt1 - 3);
    const double water = w1 / coldry;
    const double scalefac = pavel * stpfac / tavel;
    int indself = 0, indfor;
    double forfac, forfrac, selffac, selffrac = 0.0;
    if (plog > 4.56) {
      laytrop++;
      forfac = scalefac / (1. + water);
      double factor = (332.0 - tavel) / 36.0;
      int ifac = (int)factor;
      indfor = ifac < 1 ? 1 : (ifac > 2 ? 2 : ifac);
      forfrac = factor - (double)indfor;
      selffac = water * forfac;
      factor = (tavel - 188.0) / 7.2;
      ifac = (int)factor - 7;
      indself = ifac < 1 ? 1 : (ifac > 9 ? 9 : ifac);
      selffrac = factor - (double)(indself + 7);
    } else {
      forfac = scalefac / (1. + water);
      const double factor = (tavel - 188.0) / 36.0;
      indfor = 3;
      forfrac = factor - 1.0;
      selffac = water * forfac;
    }
    const double scaleminor = pavel / tavel;
    const double scaleminorn2 = (pavel / tavel) * (wbroad / (coldry + w1));
    const double fm = (tavel - 180.8) / 7.2;
    int im = (int)fm;
    const int indminor = im < 1 ? 1 : (im > 18 ? 18 : im);
    const double minorfrac = fm - (double)indminor;
    double colh2o = 1.e-20 * w1, colco2 = 1.e-20 * w2, colo3 = 1.e-20 * w3, coln2o = 1.e-20 * w4;
    double colco = 1.e-20 * w5, colch4 = 1.e-20 * w6, colo2 = 1.e-20 * w7;
    if (colco2 == 0.0) colco2 = 1.e-32 * coldry;
    if (colo3 == 0.0) colo3 = 1.e-32 * coldry;
    if (coln2o == 0.0) coln2o = 1.e-32 * coldry;
    if (colco == 0.0) colco = 1.e-32 * coldry;
    if (colch4 == 0.0) colch4 = 1.e-32 * coldry;
    const double colbrd = 1.e-20 * wbroad;
    const double compfp = 1. - fp;
    q[LP_FAC10 * 64] = compfp * ft;
    q[LP_FAC00 * 64] = compfp * (1.0 - ft);
    q[LP_FAC11 * 64] = fp * ft1;
    q[LP_FAC01 * 64] = fp * (1.0 - ft1);
    q[LP_SELFFAC * 64] = colh2o * selffac; q[LP_FORFAC * 64] = colh2o * forfac;
    q[LP_SELFFRAC * 64] = selffrac; q[LP_FORFRAC * 64] = forfrac; q[LP_MINORFRAC * 64] = minorfrac;
    q[LP_SCALEMINOR * 64] = scaleminor; q[LP_SCALEMINORN2 * 64] = scaleminorn2;
    q[LP_COLH2O * 64] = colh2o; q[LP_COLCO2 * 64] = colco2; q[LP_COLO3 * 64] = colo3; q[LP_COLN2O * 64] = coln2o;
    q[LP_COLCO * 64] = colco; q[LP_COLCH4 * 64] = colch4; q[LP_COLO2 * 64] = colo2; q[LP_COLBRD * 64] = colbrd;
    q[LP_COLDRY * 64] = coldry; q[LP_PAVEL * 64] = pavel;
    // bit 30: layer is in the lower atmosphere (counted into laytrop by the column part)
    q[LP_IDX * 64] = (double)(jp | (jt << 8) | (jt1 << 12) | (indself << 16) | (indfor << 20) | (indminor << 24) | (laytrop << 30));
    if (keep) { keep[0] = coldry; keep[keep_stride] = v1; keep[2 * keep_stride] = (double)laytrop; }
  }
}

// column part (after every layer of the column is done): laytrop, precipitable water -> diffusivity angle by band
// kept (optional): the layers' (coldry, h2o, lower flag) as lw_prep_layer hands them out: kept[(3 l + k) * kept_stride]
RRTMG_HD void lw_prep_column(const LwDev &d, const LwTab &T, int col, const double *kept = nullptr, int kept_stride = 0) {
  (void)T;
  const int L = d.nlay, N = d.ncol;
  const double amd = 28.9660, amw = 18.0160;
  int laytrop = 0;
  double amttl = 0.0, wvttl = 0.0;
#pragma unroll 8   // independent loads: keep several layers in flight
  for (int l = 0; l < L; ++l) {
    double coldry, h2o;
    if (kept) {
      coldry = kept[(3 * l) * kept_stride]; h2o = kept[(3 * l + 1) * kept_stride];
      laytrop += (int)kept[(3 * l + 2) * kept_stride];
    } else {
      const double *q = d.prep + lw_prep_off(L, col, l);
      laytrop += ((int)q[LP_IDX * 64] >> 30) & 1;
      coldry = q[LP_COLDRY * 64]; h2o = d.h2o[(long)l * N + col];
    }
    const double w1 = coldry * h2o;
    amttl = amttl + coldry + w1;
    wvttl = wvttl + w1;
  }
  d.laytrop[col] = laytrop;
  const double wvsh = (amw * wvttl) / (amd * amttl);
  const double pwvcm = wvsh * (1.e3 * d.plev[col]) / (1.e2 * d.k.grav);
  // diffusivity angle by band (rrtmg_lw_rtrn.f90:261-269)
  const double a0[16] = {1.66, 1.55, 1.58, 1.66, 1.54, 1.454, 1.89, 1.33, 1.668, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66};
  const double a1[16] = {0.00, 0.25, 0.22, 0.00, 0.13, 0.446, -0.10, 0.40, -0.006, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
  const double a2[16] = {0.00, -12.0, -11.7, 0.00, -0.72, -0.243, 0.19, -0.062, 0.414, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
  for (int b = 0; b < kLwNBand; ++b) {
    double s;
    if (b == 0 || b == 3 || b >= 9) {
      s = 1.66;
    } else {
      s = a0[b] + a1[b] * exp(a2[b] * pwvcm);
      if (s > 1.80) s = 1.80;
      if (s < 1.50) s = 1.50;
    }
    d.secdiff[(long)b * N + col] = s;
  }
}

// ------------------------------------------------------------------------------------------
// cldprop for one column, non-McICA (rrtmg_lw_cldprop.f90:118-272).  `ncbands` is carried from layer
// to layer exactly as in the reference: taucloud(lay, 1..ncbands-at-that-time) is written, and the
// final value selects the band pattern used by rtrn.
// ------------------------------------------------------------------------------------------
RRTMG_HD void lw_cloud_column(const LwDev &d, const LwTab &T, int col) {
  const int L = d.nlay, N = d.ncol;
  const double *t = T.t;
  const double cldmin = 1.e-20;
  const int icb1[16] = {1, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 5};
  int ncbands = 1;
  // first failed check in program order (see sw_cloud_layer); one code per distinct `stop` message of rrtmg_lw_cldprop.f90
  int e = 0;
  auto chk = [&](bool bad, int code) { if (e == 0 && bad) e = code; };
  for (int l = 0; l < L; ++l) {
    const long i = (long)l * N + col;
    for (int ib = 0; ib < 16; ++ib) d.ctau[((long)ib * L + l) * N + col] = 0.0;
    double tauctot = 0.0;
    if (d.taucld) for (int ib = 0; ib < 16; ++ib) tauctot = tauctot + d.taucld[i * 16 + ib];
    const double ciwp = d.cicewp ? d.cicewp[i] : 0.0, clwp = d.cliqwp ? d.cliqwp[i] : 0.0;
    const double cwp = ciwp + clwp;
    if (!(d.cldfr[i] >= cldmin && (cwp >= cldmin || tauctot >= cldmin))) continue;
    if (d.inflag == 0) {
      ncbands = 16;
      for (int ib = 0; ib < 16; ++ib) d.ctau[((long)ib * L + l) * N + col] = d.taucld[i * 16 + ib];
    } else if (d.inflag == 1) {
      ncbands = 16;
      for (int ib = 0; ib < 16; ++ib) d.ctau[((long)ib * L + l) * N + col] = t[T.abscld1] * cwp;
    } else if (d.inflag == 2) {
      double abscoice[16], abscoliq[16];
      for (int ib = 0; ib < 16; ++ib) { abscoice[ib] = 0.0; abscoliq[ib] = 0.0; }
      int iceind = 0, liqind = 0;
      const double radice = d.reice[i];
      if (ciwp == 0.0) {
        abscoice[0] = 0.0; iceind = 0;
      } else if (d.iceflag == 0) {
        chk(radice < 10.0, RRTMG_ERR_ICE_RADIUS_SMALL);   // rrtmg_lw_cldprop.f90:193
        abscoice[0] = t[T.absice0] + t[T.absice0 + 1] / radice;
        iceind = 0;
      } else if (d.iceflag == 1) {
        chk(radice < 13.0 || radice > 130., RRTMG_ERR_ICE_RADIUS);   // :198
        ncbands = 5;
        for (int ib = 0; ib < 5; ++ib) abscoice[ib] = t[T.absice1 + 2 * ib] + t[T.absice1 + 2 * ib + 1] / radice;
        iceind = 1;
      } else if (d.iceflag == 2) {
        chk(radice < 5.0 || radice > 131.0, RRTMG_ERR_ICE_RADIUS);   // :209
        ncbands = 16;
        const double factor = e ? 1.0 : (radice - 2.0) / 3.0;
        int index = (int)factor;
        if (index == 43) index = 42;
        if (index < 1) index = 1;
        const double fint = factor - (double)index;
        for (int ib = 0; ib < 16; ++ib) {
          const long k = T.absice2 + (index - 1) + 43 * ib;
          abscoice[ib] = t[k] + fint * (t[k + 1] - (t[k]));
        }
        iceind = 2;
      } else if (d.iceflag == 3) {
        chk(radice < 5.0 || radice > 140.0, RRTMG_ERR_ICE_GEN_SIZE);   // :225
        ncbands = 16;
        const double factor = e ? 1.0 : (radice - 2.0) / 3.0;
        int index = (int)factor;
        if (index == 46) index = 45;
        if (index < 1) index = 1;
        const double fint = factor - (double)index;
        for (int ib = 0; ib < 16; ++ib) {
          const long k = T.absice3 + (index - 1) + 46 * ib;
          abscoice[ib] = t[k] + fint * (t[k + 1] - (t[k]));
        }
        iceind = 2;
      }
      if (clwp == 0.0) {
        abscoliq[0] = 0.0; liqind = 0;
        if (iceind == 1) iceind = 2;
      } else if (d.liqflag == 0) {
        abscoliq[0] = t[T.absliq0]; liqind = 0;
        if (iceind == 1) iceind = 2;
      } else if (d.liqflag == 1) {
        const double radliq = d.reliq[i];
        chk(radliq < 2.5 || radliq > 60., RRTMG_ERR_LIQ_RADIUS);   // :253
        int index = (int)(radliq - 1.5);
        if (index == 0) index = 1;
        if (index == 58) index = 57;
        if (index < 1) index = 1;
        if (index > 57) index = 57;
        const double fint = radliq - 1.5 - (double)index;
        ncbands = 16;
        for (int ib = 0; ib < 16; ++ib) {
          const long k = T.absliq1 + (index - 1) + 58 * ib;
          abscoliq[ib] = t[k] + fint * (t[k + 1] - (t[k]));
        }
        liqind = 2;
      }
      for (int ib = 0; ib < ncbands; ++ib) {
        const int ii = iceind == 0 ? 0 : (iceind == 1 ? icb1[ib] - 1 : ib);
        const int il = liqind == 0 ? 0 : ib;
        // (behind a failed check: no cloud optical depth -- the solve kernels run to the end of the call whatever the flag
        // says, and an extrapolated, negative optical depth would take their table lookups out of bounds)
        d.ctau[((long)ib * L + l) * N + col] = e ? 0.0 : ciwp * abscoice[ii] + clwp * abscoliq[il];
      }
    }
  }
  d.ncbands[col] = ncbands;
  if (e) report_error(d.err, e);
}

// cldprmc band values for one (column, layer): every cloudy sub-column of band ib gets this tau
// (rrtmg_lw_cldprmc.f90:103-250)
RRTMG_HD void lw_cloudmc_layer(const LwDev &d, const LwTab &T, int col, int lay) {
  const int L = d.nlay, N = d.ncol;
  const double *t = T.t;
  const long i = (long)lay * N + col;
  const double cldmin = 1.e-20;
  const int icb1[16] = {1, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 5};
  const double ciwp = d.cicewp ? d.cicewp[i] : 0.0, clwp = d.cliqwp ? d.cliqwp[i] : 0.0;
  const double cwp = ciwp + clwp;
  int e = 0;
  auto chk = [&](bool bad, int code) { if (e == 0 && bad) e = code; };
  chk(d.inflag == 1, RRTMG_ERR_INFLAG1_MCICA);   // rrtmg_lw_cldprmc.f90:172
  for (int ib = 0; ib < 16; ++ib) {
    const double tin = d.taucld ? d.taucld[i * 16 + ib] : 0.0;
    double tau = tin;
    // (a sub-column can only be cloudy where cldfrac >= cldmin)
    if (d.inflag == 2 && d.cldfr[i] >= cldmin && (cwp >= cldmin || tin >= cldmin)) {
      double abscoice = 0.0, abscoliq = 0.0;
      const double radice = d.reice[i];
      if (ciwp == 0.0) {
      } else if (d.iceflag == 0) {
        chk(radice < 10.0, RRTMG_ERR_ICE_RADIUS_SMALL);   // :185
        abscoice = t[T.absice0] + t[T.absice0 + 1] / radice;
      } else if (d.iceflag == 1) {
        chk(radice < 13.0 || radice > 130., RRTMG_ERR_ICE_RADIUS);   // :189
        const int jb = icb1[ib] - 1;
        abscoice = t[T.absice1 + 2 * jb] + t[T.absice1 + 2 * jb + 1] / radice;
      } else if (d.iceflag == 2 || d.iceflag == 3) {
        const int nr = d.iceflag == 2 ? 43 : 46;
        if (d.iceflag == 2) chk(radice < 5.0 || radice > 131.0, RRTMG_ERR_ICE_RADIUS);   // :198
        else chk(radice < 5.0 || radice > 140.0, RRTMG_ERR_ICE_GEN_SIZE);                // :212
        const double factor = e ? 1.0 : (radice - 2.0) / 3.0;
        int index = (int)factor;
        if (index == nr) index = nr - 1;
        if (index < 1) index = 1;
        const double fint = factor - (double)index;
        const long k = (d.iceflag == 2 ? T.absice2 : T.absice3) + (index - 1) + nr * ib;
        abscoice = t[k] + fint * (t[k + 1] - (t[k]));
      }
      if (clwp == 0.0) {
      } else if (d.liqflag == 0) {
        abscoliq = t[T.absliq0];
      } else if (d.liqflag == 1) {
        const double radliq = d.reliq[i];
        chk(radliq < 2.5 || radliq > 60., RRTMG_ERR_LIQ_RADIUS);   // :234
        int index = (int)(radliq - 1.5);
        if (index == 0) index = 1;
        if (index == 58) index = 57;
        if (index < 1) index = 1;
        if (index > 57) index = 57;
        const double fint = radliq - 1.5 - (double)index;
        const long k = T.absliq1 + (index - 1) + 58 * ib;
        abscoliq = t[k] + fint * (t[k + 1] - (t[k]));
      }
      tau = e ? 0.0 : ciwp * abscoice + clwp * abscoliq;   // (see lw_cloud_column)
    }
    d.ctau[((long)ib * L + lay) * N + col] = tau;
  }
  if (e) report_error(d.err, e);
}

// OR of the sub-column masks: icldlyr of rtrnmc (rrtmg_lw_rtrnmc.f90:298-312)
RRTMG_HD void lw_anymask_column(const LwDev &d, int col) {
  for (int w = 0; w < d.nw; ++w) {
    uint64_t m = 0;
    for (int g = 0; g < kLwNGpt; ++g) m |= d.mask[((long)g * d.nw + w) * d.ncol + col];
    d.anymask[(long)w * d.ncol + col] = m;
  }
}

// ------------------------------------------------------------------------------------------
// Maximum/random overlap factors of rtrnmr for one column (rrtmg_lw_rtrnmr.f90:318-452).  Every array keeps the
// reference's own index (0 .. nlayers+1): mr[tile][index][field][lane].  The radiative loops read, for layer lev,
// the upward factors at index lev+1 and the downward ones at index lev-1.  cldfrac(0) and cldfrac(nlayers+1),
// which the reference reads out of bounds, only ever multiply factors that are zero there; they are taken as 0.
// ------------------------------------------------------------------------------------------
enum LwMrField { MR_FACCLD1 = 0, MR_FACCLD2, MR_FACCLR1, MR_FACCLR2, MR_FACCMB1, MR_FACCMB2, MR_ISTCLD,
                 MR_FACCLD1D, MR_FACCLD2D, MR_FACCLR1D, MR_FACCLR2D, MR_FACCMB1D, MR_FACCMB2D, MR_ISTCLDD, MR_N };
RRTMG_HD long lw_mr_off(int nlay, int col, int index) {
  return ((long)(col >> 6) * (nlay + 2) + index) * (MR_N * 64) + (col & 63);
}
RRTMG_HD size_t lw_mr_size(int ncol, int nlay) { return (size_t)((ncol + 63) / 64) * (nlay + 2) * MR_N * 64; }

RRTMG_HD void lw_mr_column(const LwDev &d, int col) {
  const int L = d.nlay, N = d.ncol;
  auto M = [&](int f, int index) -> double & { return d.mr[lw_mr_off(L, col, index) + f * 64]; };
  auto cf = [&](int lev) { return (lev >= 1 && lev <= L) ? d.cldfr[(long)(lev - 1) * N + col] : 0.0; };
  auto cloudy = [&](int lev) { return cf(lev) >= 1.e-6; };
  for (int i = 0; i <= L + 1; ++i)
    for (int f = 0; f < MR_N; ++f) M(f, i) = 0.0;
  double rat1 = 0.0, rat2 = 0.0;
  M(MR_ISTCLD, 1) = 1.0;
  M(MR_ISTCLDD, L) = 1.0;
  for (int lev = 1; lev <= L; ++lev) {
    if (cloudy(lev)) {
      M(MR_ISTCLD, lev + 1) = 0.0;
      if (lev == L) {
        M(MR_FACCLD1, lev + 1) = 0.0; M(MR_FACCLD2, lev + 1) = 0.0; M(MR_FACCLR1, lev + 1) = 0.0;
        M(MR_FACCLR2, lev + 1) = 0.0; M(MR_FACCMB1, lev + 1) = 0.0; M(MR_FACCMB2, lev + 1) = 0.0;
      } else if (cf(lev + 1) >= cf(lev)) {
        M(MR_FACCLD1, lev + 1) = 0.0; M(MR_FACCLD2, lev + 1) = 0.0;
        if (M(MR_ISTCLD, lev) == 1.0) {
          M(MR_FACCLR1, lev + 1) = 0.0; M(MR_FACCLR2, lev + 1) = 0.0;
          if (cf(lev) < 1.0) M(MR_FACCLR2, lev + 1) = (cf(lev + 1) - cf(lev)) / (1.0 - cf(lev));
          M(MR_FACCLR2, lev) = 0.0; M(MR_FACCLD2, lev) = 0.0;
        } else {
          const double fmax = cf(lev) > cf(lev - 1) ? cf(lev) : cf(lev - 1);
          if (cf(lev + 1) > fmax) {
            M(MR_FACCLR1, lev + 1) = rat2;
            M(MR_FACCLR2, lev + 1) = (cf(lev + 1) - fmax) / (1.0 - fmax);
          } else if (cf(lev + 1) < fmax) {
            M(MR_FACCLR1, lev + 1) = (cf(lev + 1) - cf(lev)) / (cf(lev - 1) - cf(lev));
            M(MR_FACCLR2, lev + 1) = 0.0;
          } else {
            M(MR_FACCLR1, lev + 1) = rat2;
            M(MR_FACCLR2, lev + 1) = 0.0;
          }
        }
        if (M(MR_FACCLR1, lev + 1) > 0.0 || M(MR_FACCLR2, lev + 1) > 0.0) { rat1 = 1.0; rat2 = 0.0; }
        else { rat1 = 0.0; rat2 = 0.0; }
      } else {
        M(MR_FACCLR1, lev + 1) = 0.0; M(MR_FACCLR2, lev + 1) = 0.0;
        if (M(MR_ISTCLD, lev) == 1.0) {
          M(MR_FACCLD1, lev + 1) = 0.0;
          M(MR_FACCLD2, lev + 1) = (cf(lev) - cf(lev + 1)) / cf(lev);
          M(MR_FACCLR2, lev) = 0.0; M(MR_FACCLD2, lev) = 0.0;
        } else {
          const double fmin = cf(lev) < cf(lev - 1) ? cf(lev) : cf(lev - 1);
          if (cf(lev + 1) <= fmin) {
            M(MR_FACCLD1, lev + 1) = rat1;
            M(MR_FACCLD2, lev + 1) = (fmin - cf(lev + 1)) / fmin;
          } else {
            M(MR_FACCLD1, lev + 1) = (cf(lev) - cf(lev + 1)) / (cf(lev) - fmin);
            M(MR_FACCLD2, lev + 1) = 0.0;
          }
        }
        if (M(MR_FACCLD1, lev + 1) > 0.0 || M(MR_FACCLD2, lev + 1) > 0.0) { rat1 = 0.0; rat2 = 1.0; }
        else { rat1 = 0.0; rat2 = 0.0; }
      }
      M(MR_FACCMB1, lev + 1) = M(MR_FACCLR1, lev + 1) * M(MR_FACCLD2, lev) * cf(lev - 1);
      M(MR_FACCMB2, lev + 1) = M(MR_FACCLD1, lev + 1) * M(MR_FACCLR2, lev) * (1.0 - cf(lev - 1));
    } else {
      M(MR_ISTCLD, lev + 1) = 1.0;
    }
  }
  for (int lev = L; lev >= 1; --lev) {
    if (cloudy(lev)) {
      M(MR_ISTCLDD, lev - 1) = 0.0;
      if (lev == 1) {
        M(MR_FACCLD1D, lev - 1) = 0.0; M(MR_FACCLD2D, lev - 1) = 0.0; M(MR_FACCLR1D, lev - 1) = 0.0;
        M(MR_FACCLR2D, lev - 1) = 0.0; M(MR_FACCMB1D, lev - 1) = 0.0; M(MR_FACCMB2D, lev - 1) = 0.0;
      } else if (cf(lev - 1) >= cf(lev)) {
        M(MR_FACCLD1D, lev - 1) = 0.0; M(MR_FACCLD2D, lev - 1) = 0.0;
        if (M(MR_ISTCLDD, lev) == 1.0) {
          M(MR_FACCLR1D, lev - 1) = 0.0; M(MR_FACCLR2D, lev - 1) = 0.0;
          if (cf(lev) < 1.0) M(MR_FACCLR2D, lev - 1) = (cf(lev - 1) - cf(lev)) / (1.0 - cf(lev));
          M(MR_FACCLR2D, lev) = 0.0; M(MR_FACCLD2D, lev) = 0.0;
        } else {
          const double fmax = cf(lev) > cf(lev + 1) ? cf(lev) : cf(lev + 1);
          if (cf(lev - 1) > fmax) {
            M(MR_FACCLR1D, lev - 1) = rat2;
            M(MR_FACCLR2D, lev - 1) = (cf(lev - 1) - fmax) / (1.0 - fmax);
          } else if (cf(lev - 1) < fmax) {
            M(MR_FACCLR1D, lev - 1) = (cf(lev - 1) - cf(lev)) / (cf(lev + 1) - cf(lev));
            M(MR_FACCLR2D, lev - 1) = 0.0;
          } else {
            M(MR_FACCLR1D, lev - 1) = rat2;
            M(MR_FACCLR2D, lev - 1) = 0.0;
          }
        }
        if (M(MR_FACCLR1D, lev - 1) > 0.0 || M(MR_FACCLR2D, lev - 1) > 0.0) { rat1 = 1.0; rat2 = 0.0; }
        else { rat1 = 0.0; rat2 = 0.0; }
      } else {
        M(MR_FACCLR1D, lev - 1) = 0.0; M(MR_FACCLR2D, lev - 1) = 0.0;
        if (M(MR_ISTCLDD, lev) == 1.0) {
          M(MR_FACCLD1D, lev - 1) = 0.0;
          M(MR_FACCLD2D, lev - 1) = (cf(lev) - cf(lev - 1)) / cf(lev);
          M(MR_FACCLR2D, lev) = 0.0; M(MR_FACCLD2D, lev) = 0.0;
        } else {
          const double fmin = cf(lev) < cf(lev + 1) ? cf(lev) : cf(lev + 1);
          if (cf(lev - 1) <= fmin) {
            M(MR_FACCLD1D, lev - 1) = rat1;
            M(MR_FACCLD2D, lev - 1) = (fmin - cf(lev - 1)) / fmin;
          } else {
            M(MR_FACCLD1D, lev - 1) = (cf(lev) - cf(lev - 1)) / (cf(lev) - fmin);
            M(MR_FACCLD2D, lev - 1) = 0.0;
          }
        }
        if (M(MR_FACCLD1D, lev - 1) > 0.0 || M(MR_FACCLD2D, lev - 1) > 0.0) { rat1 = 0.0; rat2 = 1.0; }
        else { rat1 = 0.0; rat2 = 0.0; }
      }
      M(MR_FACCMB1D, lev - 1) = M(MR_FACCLR1D, lev - 1) * M(MR_FACCLD2D, lev) * cf(lev + 1);
      M(MR_FACCMB2D, lev - 1) = M(MR_FACCLD1D, lev - 1) * M(MR_FACCLR2D, lev) * (1.0 - cf(lev + 1));
    } else {
      M(MR_ISTCLDD, lev - 1) = 1.0;
    }
  }
}

// ------------------------------------------------------------------------------------------
// taumol pieces
// ------------------------------------------------------------------------------------------
struct LwLayerIn {
  double fac00, fac01, fac10, fac11, selffac, selffrac, forfac, forfrac, minorfrac, scaleminor, scaleminorn2;
  double colh2o, colco2, colo3, coln2o, colco, colch4, colo2, colbrd, coldry, pavel, wx1, wx2, wx3, wx4;
  int jp, jt, jt1, indself, indfor, indminor;
};

RRTMG_HD void lw_load_layer(const LwDev &d, int col, int lay, LwLayerIn &s) {
  const double *q = d.prep + lw_prep_off(d.nlay, col, lay);
  s.fac00 = q[LP_FAC00 * 64]; s.fac01 = q[LP_FAC01 * 64]; s.fac10 = q[LP_FAC10 * 64]; s.fac11 = q[LP_FAC11 * 64];
  s.selffac = q[LP_SELFFAC * 64]; s.selffrac = q[LP_SELFFRAC * 64]; s.forfac = q[LP_FORFAC * 64]; s.forfrac = q[LP_FORFRAC * 64];
  s.minorfrac = q[LP_MINORFRAC * 64]; s.scaleminor = q[LP_SCALEMINOR * 64]; s.scaleminorn2 = q[LP_SCALEMINORN2 * 64];
  s.colh2o = q[LP_COLH2O * 64]; s.colco2 = q[LP_COLCO2 * 64]; s.colo3 = q[LP_COLO3 * 64]; s.coln2o = q[LP_COLN2O * 64];
  s.colco = q[LP_COLCO * 64]; s.colch4 = q[LP_COLCH4 * 64]; s.colo2 = q[LP_COLO2 * 64]; s.colbrd = q[LP_COLBRD * 64];
  s.coldry = q[LP_COLDRY * 64]; s.pavel = q[LP_PAVEL * 64];
  s.wx1 = q[LP_WX1 * 64]; s.wx2 = q[LP_WX2 * 64]; s.wx3 = q[LP_WX3 * 64]; s.wx4 = q[LP_WX4 * 64];
  const int p = (int)q[LP_IDX * 64];
  s.jp = p & 0xff; s.jt = (p >> 8) & 0xf; s.jt1 = (p >> 12) & 0xf; s.indself = (p >> 16) & 0xf; s.indfor = (p >> 20) & 0xf;
  s.indminor = (p >> 24) & 0x1f;
}

// End of one term of taumol (a group of table-row reads and the arithmetic on them): nothing is scheduled across, so that a
// term's rows are dead before the next term's rows are requested.  Left alone the scheduler keeps ~20 rows (8 registers each
// at G = 4) in flight to hide the LDS latency: 245 VGPRs in the clear-sky kernel, 97 spilled in the McICA one; with the fences
// 213 / 10 spilled, the cloudy kernel 5 % faster, the clear-sky one unchanged (2 waves/SIMD either way; 168 for a third wave
// is out of reach: the layer's 26 prep values alone are 56 registers).
#if defined(__HIP_DEVICE_COMPILE__)
#define LW_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define LW_FENCE
#endif
struct LwSpec { double speccomb, specparm, fs; int js; };
RRTMG_HD LwSpec lw_spec(double colx, double rat, double coly, double mult) {
  LwSpec r;
  r.speccomb = colx + rat * coly;
  r.specparm = colx / r.speccomb;
  const double oneminus = 1.0 - 1.e-6;
  if (r.specparm >= oneminus) r.specparm = oneminus;
  const double specmult = mult * r.specparm;
  r.js = 1 + (int)specmult;
  r.fs = specmult - (double)(int)specmult;
  return r;
}

// chi_mls(m, j) with 1-based (m, j): reference mixing ratios (rrlw_ref)
RRTMG_HD double lw_chi(const LwTab &T, int m, int j) { return T.t[T.chi_mls + (m - 1) + 7 * (j - 1)]; }

// lower-atmosphere binary-species major term with the 3-point end-zone blend
// (pattern at rrtmg_lw_taumol.f90:550-609 / :622-668); k -> g-point-fastest table view, ind 0-based row
// of (js, jt, jp) ; f0/f1 = (fac00, fac10) for the jp side or (fac01, fac11) for the jp+1 side.
template <int G, int NG>
RRTMG_HD V<G> lw_major_lower(const KTab<G, NG> &k, int ind, const LwSpec &sp, double f0, double f1) {
  if (sp.specparm < 0.125) {
    const double p = sp.fs - 1;
    const double p2 = p * p, p4 = p2 * p2;
    const double fk0 = p4, fk1 = 1 - p - 2.0 * p4, fk2 = p + p4;
    { const V<G> r__ = sp.speccomb * ((fk0 * f0) * k[ind] + (fk1 * f0) * k[ind + 1] + (fk2 * f0) * k[ind + 2] + (fk0 * f1) * k[ind + 9] +
                          (fk1 * f1) * k[ind + 10] + (fk2 * f1) * k[ind + 11]); LW_FENCE; return r__; }
  } else if (sp.specparm > 0.875) {
    const double p = -sp.fs;
    const double p2 = p * p, p4 = p2 * p2;
    const double fk0 = p4, fk1 = 1 - p - 2.0 * p4, fk2 = p + p4;
    { const V<G> r__ = sp.speccomb * ((fk2 * f0) * k[ind - 1] + (fk1 * f0) * k[ind] + (fk0 * f0) * k[ind + 1] + (fk2 * f1) * k[ind + 8] +
                          (fk1 * f1) * k[ind + 9] + (fk0 * f1) * k[ind + 10]); LW_FENCE; return r__; }
  }
  { const V<G> r__ = sp.speccomb * (((1.0 - sp.fs) * f0) * k[ind] + (sp.fs * f0) * k[ind + 1] + ((1.0 - sp.fs) * f1) * k[ind + 9] +
                        (sp.fs * f1) * k[ind + 10]); LW_FENCE; return r__; }
}
// upper-atmosphere binary-species term (4 points, nspb = 5)
template <int G, int NG>
RRTMG_HD V<G> lw_major_upper(const KTab<G, NG> &k, int ind, const LwSpec &sp, double f0, double f1) {
  { const V<G> r__ = sp.speccomb * (((1.0 - sp.fs) * f0) * k[ind] + (sp.fs * f0) * k[ind + 1] + ((1.0 - sp.fs) * f1) * k[ind + 5] + (sp.fs * f1) * k[ind + 6]); LW_FENCE; return r__; }
}
template <int G, int NG>
RRTMG_HD V<G> lw_m4(const KTab<G, NG> &k, int i0, int i1, const LwLayerIn &s) {
  { const V<G> r__ = s.fac00 * k[i0] + s.fac10 * k[i0 + 1] + s.fac01 * k[i1] + s.fac11 * k[i1 + 1]; LW_FENCE; return r__; }
}
template <int G, int NG>
RRTMG_HD V<G> lw_tauself(const KTab<G, NG> &selfref, const LwLayerIn &s) {
  const V<G> a = selfref[s.indself - 1], b = selfref[s.indself];
  { const V<G> r__ = s.selffac * (a + s.selffrac * (b - a)); LW_FENCE; return r__; }
}
template <int G, int NG>
RRTMG_HD V<G> lw_taufor(const KTab<G, NG> &forref, const LwLayerIn &s) {
  const V<G> a = forref[s.indfor - 1], b = forref[s.indfor];
  { const V<G> r__ = s.forfac * (a + s.forfrac * (b - a)); LW_FENCE; return r__; }
}
// minor-gas coefficient, temperature-interpolated: table (19, ng) stored [19][ng]
template <int G, int NG>
RRTMG_HD V<G> lw_minor1(const KTab<G, NG> &m, const LwLayerIn &s) {
  const V<G> a = m[s.indminor - 1], b = m[s.indminor];
  { const V<G> r__ = a + s.minorfrac * (b - a); LW_FENCE; return r__; }
}
// minor-gas coefficient, (mixture, temperature)-interpolated: table (nm, 19, ng) stored [19*nm][ng]
template <int G, int NG>
RRTMG_HD V<G> lw_minor2(const KTab<G, NG> &m, int nm, int jm, double fm, const LwLayerIn &s) {
  const int r = nm * (s.indminor - 1) + (jm - 1);
  const V<G> a0 = m[r], a1 = m[r + 1], b0 = m[r + nm], b1 = m[r + nm + 1];
  const V<G> m1 = a0 + fm * (a1 - a0);
  const V<G> m2 = b0 + fm * (b1 - b0);
  { const V<G> r__ = m1 + s.minorfrac * (m2 - m1); LW_FENCE; return r__; }
}
// Planck fraction interpolated in the reference mixture: table (ng, nmix) = [nmix][ng]
template <int G, int NG>
RRTMG_HD V<G> lw_frac2(const KTab<G, NG> &tab, const LwSpec &pl) {
  const V<G> a = tab[pl.js - 1], b = tab[pl.js];
  { const V<G> r__ = a + pl.fs * (b - a); LW_FENCE; return r__; }
}
// "too abundant" minor-gas column adjustment: adjfac = a + (rat - a)**e  
RRTMG_HD double lw_adjcol(double col, double coldry, double chiref, double e20, double thresh, double a, double e, double chimul) {
  const double chi = col / coldry;
  const double rat = e20 * chi / chiref;
  if (rat > thresh) {
    const double adjfac = a + pow(rat - a, e);
    return adjfac * chimul * coldry * 1.e-20;
  }
  return col;
}

constexpr double kE20f = (double)1.e20f;   // `1.e20` default-real literals at rrtmg_lw_taumol.f90:715,:1462,:1618

// reduced g-points per band (rrtmg_lw parrrtm.f90 ng1..ng16)
constexpr int kLwNg[16] = {10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2};
// species-ratio pairs of the binary bands, rows of the chi ratio table built at init (LwTab::chirat)
enum { CR_12 = 0, CR_32, CR_13, CR_16, CR_14, CR_42, CR_N };

// gas optical depths and Planck fractions of the G g-points ig0 .. ig0+G-1 of band BAND (1..16) in one layer
// LDSK = true: kb -> the item's slice of the band slab, [nrows][G] (the workgroup's LDS copy); LDSK = false: kb is
// ignored and the slab is read in place, [nrows][ng], through the vector L1.
template <int BAND, int G, bool LDSK = false>
RRTMG_HD V<G> lw_taug(const LwTab &T, const LwLayerIn &s, bool lower, int ig0, V<G> &fracs, const double *kb = nullptr) {
  const LwBandTab &B = T.b[BAND - 1];
  const double *t = T.t;
  constexpr int NG = kLwNg[BAND - 1];
  constexpr int ST = LDSK ? G : NG;
  if (!LDSK) kb = t + B.slab + ig0;
  // g-point-fastest table views ([row][ST], rows as in the reference's first dimensions)
  auto view = [&](int r) { return KTab<G, ST>{kb + (long)r * ST}; };
  const KTab<G, ST> absa = view(B.r_absa), absb = view(B.r_absb), selfref = view(B.r_self), forref = view(B.r_forr);
  const KTab<G, ST> ma0 = view(B.r_ma[0]), ma1 = view(B.r_ma[1]), ma2 = view(B.r_ma[2]), mb0 = view(B.r_mb[0]), mb1 = view(B.r_mb[1]);
  const KTab<G, ST> fraca = view(B.r_fraca), fracb = view(B.r_fracb);
  auto row = [&](int r) { return vload<G>(kb + (long)r * ST); };   // a one-row ([ng]) table
  V<G> taug = vsplat<G>(0.0);
  const int i0s = ((s.jp - 1) * 5 + (s.jt - 1)), i1s = (s.jp * 5 + (s.jt1 - 1));         // lower, nspa = 1
  // band 16 has a kb table but nspb(16) = 0 in lwdatinit, so the reference's index
  // ((jp-13)*5+(jt-1))*nspb(16) + 1 collapses to 1 for every upper layer (rrtmg_lw_taumol.f90 taugb16)
  const int u0s = ((s.jp - 13) * 5 + (s.jt - 1)) * (BAND == 16 ? 0 : 1), u1s = ((s.jp - 12) * 5 + (s.jt1 - 1)) * (BAND == 16 ? 0 : 1);  // upper, nspb = 1
  // species pair of the binary bands: chi_x/chi_y at reference level j (quotients formed at init)
  auto chirat = [&](int pair, int j) { return t[T.chirat + pair * 59 + (j - 1)]; };
  (void)ma1; (void)ma2; (void)mb1; (void)fracb; (void)absb; (void)chirat; (void)u0s; (void)u1s;

  if constexpr (BAND == 1) {
    const double scalen2 = s.colbrd * s.scaleminorn2;
    if (lower) {
      double corradj = 1.;
      if (s.pavel < 250.0) corradj = 1.0 - 0.15 * (250.0 - s.pavel) / 154.4;
      const V<G> taun2 = scalen2 * lw_minor1(ma0, s);
      taug = corradj * (s.colh2o * lw_m4(absa, i0s, i1s, s) + lw_tauself(selfref, s) + lw_taufor(forref, s) + taun2);
      fracs = fraca[0];
    } else {
      const double corradj = 1.0 - 0.15 * (s.pavel / 95.6);
      const V<G> taun2 = scalen2 * lw_minor1(mb0, s);
      taug = corradj * (s.colh2o * lw_m4(absb, u0s, u1s, s) + lw_taufor(forref, s) + taun2);
      fracs = fracb[0];
    }
  } else if constexpr (BAND == 2) {
    if (lower) {
      const double corradj = 1.0 - .05 * (s.pavel - 100.0) / 900.0;
      taug = corradj * (s.colh2o * lw_m4(absa, i0s, i1s, s) + lw_tauself(selfref, s) + lw_taufor(forref, s));
      fracs = fraca[0];
    } else {
      taug = s.colh2o * lw_m4(absb, u0s, u1s, s) + lw_taufor(forref, s);
      fracs = fracb[0];
    }
  } else if constexpr (BAND == 3) {
    // h2o/co2; minor n2o
    if (lower) {
      const LwSpec sp = lw_spec(s.colh2o, chirat(CR_12, s.jp), s.colco2, 8.0), sp1 = lw_spec(s.colh2o, chirat(CR_12, s.jp + 1), s.colco2, 8.0);
      const LwSpec sm = lw_spec(s.colh2o, chirat(CR_12, 3), s.colco2, 8.0), pl = lw_spec(s.colh2o, chirat(CR_12, 9), s.colco2, 8.0);
      const double adjcoln2o = lw_adjcol(s.coln2o, s.coldry, lw_chi(T, 4, s.jp + 1), 1.e20, 1.5, 0.5, 0.65, lw_chi(T, 4, s.jp + 1));
      const V<G> absn2o = lw_minor2(ma0, 9, sm.js, sm.fs, s);
      taug = lw_major_lower(absa, i0s * 9 + sp.js - 1, sp, s.fac00, s.fac10) + lw_major_lower(absa, i1s * 9 + sp1.js - 1, sp1, s.fac01, s.fac11) +
             lw_tauself(selfref, s) + lw_taufor(forref, s) + adjcoln2o * absn2o;
      fracs = lw_frac2(fraca, pl);
    } else {
      const LwSpec sp = lw_spec(s.colh2o, chirat(CR_12, s.jp), s.colco2, 4.0), sp1 = lw_spec(s.colh2o, chirat(CR_12, s.jp + 1), s.colco2, 4.0);
      const LwSpec sm = lw_spec(s.colh2o, chirat(CR_12, 13), s.colco2, 4.0), pl = lw_spec(s.colh2o, chirat(CR_12, 13), s.colco2, 4.0);
      const double adjcoln2o = lw_adjcol(s.coln2o, s.coldry, lw_chi(T, 4, s.jp + 1), kE20f, 1.5, 0.5, 0.65, lw_chi(T, 4, s.jp + 1));
      const V<G> absn2o = lw_minor2(mb0, 5, sm.js, sm.fs, s);
      taug = lw_major_upper(absb, u0s * 5 + sp.js - 1, sp, s.fac00, s.fac10) + lw_major_upper(absb, u1s * 5 + sp1.js - 1, sp1, s.fac01, s.fac11) +
             lw_taufor(forref, s) + adjcoln2o * absn2o;
      fracs = lw_frac2(fracb, pl);
    }
  } else if constexpr (BAND == 4 || BAND == 5) {
    // lower h2o/co2 ; upper o3/co2.  band 5: minor o3 (lower), ccl4 (both)
    if (lower) {
      const LwSpec sp = lw_spec(s.colh2o, chirat(CR_12, s.jp), s.colco2, 8.0), sp1 = lw_spec(s.colh2o, chirat(CR_12, s.jp + 1), s.colco2, 8.0);
      const LwSpec pl = lw_spec(s.colh2o, chirat(CR_12, BAND == 4 ? 11 : 5), s.colco2, 8.0);
      taug = lw_major_lower(absa, i0s * 9 + sp.js - 1, sp, s.fac00, s.fac10) + lw_major_lower(absa, i1s * 9 + sp1.js - 1, sp1, s.fac01, s.fac11) +
             lw_tauself(selfref, s) + lw_taufor(forref, s);
      if constexpr (BAND == 5) {
        const LwSpec sm = lw_spec(s.colh2o, chirat(CR_12, 7), s.colco2, 8.0);
        const V<G> abso3 = lw_minor2(ma0, 9, sm.js, sm.fs, s);
        taug = taug + abso3 * s.colo3 + s.wx1 * row(B.r_x[0]);
      }
      fracs = lw_frac2(fraca, pl);
    } else {
      const LwSpec sp = lw_spec(s.colo3, chirat(CR_32, s.jp), s.colco2, 4.0), sp1 = lw_spec(s.colo3, chirat(CR_32, s.jp + 1), s.colco2, 4.0);
      const LwSpec pl = lw_spec(s.colo3, chirat(CR_32, BAND == 4 ? 13 : 43), s.colco2, 4.0);
      taug = lw_major_upper(absb, u0s * 5 + sp.js - 1, sp, s.fac00, s.fac10) + lw_major_upper(absb, u1s * 5 + sp1.js - 1, sp1, s.fac01, s.fac11);
      if constexpr (BAND == 5) taug = taug + s.wx1 * row(B.r_x[0]);
      fracs = lw_frac2(fracb, pl);
      if constexpr (BAND == 4) {
        // empirical stratospheric scalings, default-real literals (rrtmg_lw_taumol.f90:1009-1015)
        const double sc[7] = {(double)0.92f, (double)0.88f, (double)1.07f, (double)1.1f, (double)0.99f, (double)0.88f, (double)0.943f};
#pragma unroll
        for (int j = 0; j < G; ++j) if (ig0 + j >= 7) taug[j] = taug[j] * sc[ig0 + j - 7];
      }
    }
  } else if constexpr (BAND == 6) {
    if (lower) {
      const double adjcolco2 = lw_adjcol(s.colco2, s.coldry, lw_chi(T, 2, s.jp + 1), 1.e20, 3.0, 2.0, 0.77, lw_chi(T, 2, s.jp + 1));
      const V<G> absco2 = lw_minor1(ma0, s);
      taug = s.colh2o * lw_m4(absa, i0s, i1s, s) + lw_tauself(selfref, s) + lw_taufor(forref, s) + adjcolco2 * absco2 +
             s.wx2 * row(B.r_x[0]) + s.wx3 * row(B.r_x[1]);
    } else {
      taug = 0.0 + s.wx2 * row(B.r_x[0]) + s.wx3 * row(B.r_x[1]);
    }
    fracs = fraca[0];
  } else if constexpr (BAND == 7) {
    // lower h2o/o3, minor co2 ; upper o3, minor co2
    if (lower) {
      const LwSpec sp = lw_spec(s.colh2o, chirat(CR_13, s.jp), s.colo3, 8.0), sp1 = lw_spec(s.colh2o, chirat(CR_13, s.jp + 1), s.colo3, 8.0);
      const LwSpec sm = lw_spec(s.colh2o, chirat(CR_13, 3), s.colo3, 8.0), pl = lw_spec(s.colh2o, chirat(CR_13, 3), s.colo3, 8.0);
      const double adjcolco2 = lw_adjcol(s.colco2, s.coldry, lw_chi(T, 2, s.jp + 1), kE20f, 3.0, 3.0, 0.79, lw_chi(T, 2, s.jp + 1));
      const V<G> absco2 = lw_minor2(ma0, 9, sm.js, sm.fs, s);
      taug = lw_major_lower(absa, i0s * 9 + sp.js - 1, sp, s.fac00, s.fac10) + lw_major_lower(absa, i1s * 9 + sp1.js - 1, sp1, s.fac01, s.fac11) +
             lw_tauself(selfref, s) + lw_taufor(forref, s) + adjcolco2 * absco2;
      fracs = lw_frac2(fraca, pl);
    } else {
      const double adjcolco2 = lw_adjcol(s.colco2, s.coldry, lw_chi(T, 2, s.jp + 1), kE20f, 3.0, 2.0, 0.79, lw_chi(T, 2, s.jp + 1));
      const V<G> absco2 = lw_minor1(mb0, s);
      taug = s.colo3 * lw_m4(absb, u0s, u1s, s) + adjcolco2 * absco2;
      fracs = fracb[0];
      const double sc[6] = {0.92, 0.88, 1.07, 1.1, 0.99, 0.855};   // rrtmg_lw_taumol.f90:1645-1650 (_rb literals)
#pragma unroll
      for (int j = 0; j < G; ++j) if (ig0 + j >= 5 && ig0 + j <= 10) taug[j] = taug[j] * sc[ig0 + j - 5];
    }
  } else if constexpr (BAND == 8) {
    const double adjcolco2 = lw_adjcol(s.colco2, s.coldry, lw_chi(T, 2, s.jp + 1), 1.e20, 3.0, 2.0, 0.65, lw_chi(T, 2, s.jp + 1));
    if (lower) {
      const V<G> absco2 = lw_minor1(ma0, s), abso3 = lw_minor1(ma1, s), absn2o = lw_minor1(ma2, s);
      taug = s.colh2o * lw_m4(absa, i0s, i1s, s) + lw_tauself(selfref, s) + lw_taufor(forref, s) + adjcolco2 * absco2 + s.colo3 * abso3 +
             s.coln2o * absn2o + s.wx3 * row(B.r_x[0]) + s.wx4 * row(B.r_x[1]);
      fracs = fraca[0];
    } else {
      const V<G> absco2 = lw_minor1(mb0, s), absn2o = lw_minor1(mb1, s);
      taug = s.colo3 * lw_m4(absb, u0s, u1s, s) + adjcolco2 * absco2 + s.coln2o * absn2o + s.wx3 * row(B.r_x[0]) + s.wx4 * row(B.r_x[1]);
      fracs = fracb[0];
    }
  } else if constexpr (BAND == 9) {
    const double adjcoln2o = lw_adjcol(s.coln2o, s.coldry, lw_chi(T, 4, s.jp + 1), 1.e20, 1.5, 0.5, 0.65, lw_chi(T, 4, s.jp + 1));
    if (lower) {
      const LwSpec sp = lw_spec(s.colh2o, chirat(CR_16, s.jp), s.colch4, 8.0), sp1 = lw_spec(s.colh2o, chirat(CR_16, s.jp + 1), s.colch4, 8.0);
      const LwSpec sm = lw_spec(s.colh2o, chirat(CR_16, 3), s.colch4, 8.0), pl = lw_spec(s.colh2o, chirat(CR_16, 9), s.colch4, 8.0);
      const V<G> absn2o = lw_minor2(ma0, 9, sm.js, sm.fs, s);
      taug = lw_major_lower(absa, i0s * 9 + sp.js - 1, sp, s.fac00, s.fac10) + lw_major_lower(absa, i1s * 9 + sp1.js - 1, sp1, s.fac01, s.fac11) +
             lw_tauself(selfref, s) + lw_taufor(forref, s) + adjcoln2o * absn2o;
      fracs = lw_frac2(fraca, pl);
    } else {
      const V<G> absn2o = lw_minor1(mb0, s);
      taug = s.colch4 * lw_m4(absb, u0s, u1s, s) + adjcoln2o * absn2o;
      fracs = fracb[0];
    }
  } else if constexpr (BAND == 10) {
    if (lower) {
      taug = s.colh2o * lw_m4(absa, i0s, i1s, s) + lw_tauself(selfref, s) + lw_taufor(forref, s);
      fracs = fraca[0];
    } else {
      taug = s.colh2o * lw_m4(absb, u0s, u1s, s) + lw_taufor(forref, s);
      fracs = fracb[0];
    }
  } else if constexpr (BAND == 11) {
    const double scaleo2 = s.colo2 * s.scaleminor;
    if (lower) {
      const V<G> tauo2 = scaleo2 * lw_minor1(ma0, s);
      taug = s.colh2o * lw_m4(absa, i0s, i1s, s) + lw_tauself(selfref, s) + lw_taufor(forref, s) + tauo2;
      fracs = fraca[0];
    } else {
      const V<G> tauo2 = scaleo2 * lw_minor1(mb0, s);
      taug = s.colh2o * lw_m4(absb, u0s, u1s, s) + lw_taufor(forref, s) + tauo2;
      fracs = fracb[0];
    }
  } else if constexpr (BAND == 12) {
    if (lower) {
      const LwSpec sp = lw_spec(s.colh2o, chirat(CR_12, s.jp), s.colco2, 8.0), sp1 = lw_spec(s.colh2o, chirat(CR_12, s.jp + 1), s.colco2, 8.0);
      const LwSpec pl = lw_spec(s.colh2o, chirat(CR_12, 10), s.colco2, 8.0);
      taug = lw_major_lower(absa, i0s * 9 + sp.js - 1, sp, s.fac00, s.fac10) + lw_major_lower(absa, i1s * 9 + sp1.js - 1, sp1, s.fac01, s.fac11) +
             lw_tauself(selfref, s) + lw_taufor(forref, s);
      fracs = lw_frac2(fraca, pl);
    } else {
      taug = vsplat<G>(0.0); fracs = vsplat<G>(0.0);
    }
  } else if constexpr (BAND == 13) {
    // lower h2o/n2o, minor co2 and co ; upper o3 minor only
    if (lower) {
      const LwSpec sp = lw_spec(s.colh2o, chirat(CR_14, s.jp), s.coln2o, 8.0), sp1 = lw_spec(s.colh2o, chirat(CR_14, s.jp + 1), s.coln2o, 8.0);
      const LwSpec sm = lw_spec(s.colh2o, chirat(CR_14, 1), s.coln2o, 8.0), sm3 = lw_spec(s.colh2o, chirat(CR_14, 3), s.coln2o, 8.0);
      const LwSpec pl = lw_spec(s.colh2o, chirat(CR_14, 5), s.coln2o, 8.0);
      // adjcolco2 = adjfac*3.55e-4*coldry*1e-20 with a default-real 3.55e-4 (rrtmg_lw_taumol.f90:2479)
      const double adjcolco2 = lw_adjcol(s.colco2, s.coldry, 3.55e-4, 1.e20, 3.0, 2.0, 0.68, (double)3.55e-4f);
      const V<G> absco2 = lw_minor2(ma0, 9, sm.js, sm.fs, s);
      const V<G> absco = lw_minor2(ma1, 9, sm3.js, sm3.fs, s);
      taug = lw_major_lower(absa, i0s * 9 + sp.js - 1, sp, s.fac00, s.fac10) + lw_major_lower(absa, i1s * 9 + sp1.js - 1, sp1, s.fac01, s.fac11) +
             lw_tauself(selfref, s) + lw_taufor(forref, s) + adjcolco2 * absco2 + s.colco * absco;
      fracs = lw_frac2(fraca, pl);
    } else {
      const V<G> abso3 = lw_minor1(mb0, s);
      taug = s.colo3 * abso3;
      fracs = fracb[0];
    }
  } else if constexpr (BAND == 14) {
    if (lower) {
      taug = s.colco2 * lw_m4(absa, i0s, i1s, s) + lw_tauself(selfref, s) + lw_taufor(forref, s);
      fracs = fraca[0];
    } else {
      taug = s.colco2 * lw_m4(absb, u0s, u1s, s);
      fracs = fracb[0];
    }
  } else if constexpr (BAND == 15) {
    // lower n2o/co2, minor n2 ; nothing above
    if (lower) {
      const LwSpec sp = lw_spec(s.coln2o, chirat(CR_42, s.jp), s.colco2, 8.0), sp1 = lw_spec(s.coln2o, chirat(CR_42, s.jp + 1), s.colco2, 8.0);
      const LwSpec sm = lw_spec(s.coln2o, chirat(CR_42, 1), s.colco2, 8.0), pl = lw_spec(s.coln2o, chirat(CR_42, 1), s.colco2, 8.0);
      const double scalen2 = s.colbrd * s.scaleminor;
      const V<G> taun2 = scalen2 * lw_minor2(ma0, 9, sm.js, sm.fs, s);
      taug = lw_major_lower(absa, i0s * 9 + sp.js - 1, sp, s.fac00, s.fac10) + lw_major_lower(absa, i1s * 9 + sp1.js - 1, sp1, s.fac01, s.fac11) +
             lw_tauself(selfref, s) + lw_taufor(forref, s) + taun2;
      fracs = lw_frac2(fraca, pl);
    } else {
      taug = vsplat<G>(0.0); fracs = vsplat<G>(0.0);
    }
  } else {  // 16
    if (lower) {
      const LwSpec sp = lw_spec(s.colh2o, chirat(CR_16, s.jp), s.colch4, 8.0), sp1 = lw_spec(s.colh2o, chirat(CR_16, s.jp + 1), s.colch4, 8.0);
      const LwSpec pl = lw_spec(s.colh2o, chirat(CR_16, 6), s.colch4, 8.0);
      taug = lw_major_lower(absa, i0s * 9 + sp.js - 1, sp, s.fac00, s.fac10) + lw_major_lower(absa, i1s * 9 + sp1.js - 1, sp1, s.fac01, s.fac11) +
             lw_tauself(selfref, s) + lw_taufor(forref, s);
      fracs = lw_frac2(fraca, pl);
    } else {
      taug = s.colch4 * lw_m4(absb, u0s, u1s, s);
      fracs = fracb[0];
    }
  }
  return taug;
}

// Planck function integrated over band ib (0-based) at temperature tt: linear interpolation in
// totplnk(181,16) (rrtmg_lw_setcoef.f90:147-250, istart = 1)
RRTMG_HD double lw_planck(const LwTab &T, int ib, double tt) {
  int ind = (int)(tt - 159.0);
  if (ind < 1) ind = 1; else if (ind > 180) ind = 180;
  const double frac = tt - 159.0 - (double)ind;
  const double *p = T.t + T.totplnk + 181 * ib + (ind - 1);
  return p[0] + frac * (p[1] - p[0]);
}
RRTMG_HD double lw_planck_deriv(const LwTab &T, int ib, double tt) {
  int ind = (int)(tt - 159.0);
  if (ind < 1) ind = 1; else if (ind > 180) ind = 180;
  const double frac = tt - 159.0 - (double)ind;
  const double *p = T.t + T.totplnkderiv + 181 * ib + (ind - 1);
  return p[0] + frac * (p[1] - p[0]);
}

// the same in two halves: the two table entries are requested where the temperature is known, the interpolation is formed
// where the value is needed -- with the layer's taumol between the two in lw_solve_thread, the gathers fly under it
struct LwPlanckRaw { double p0, p1, frac; };
RRTMG_HD LwPlanckRaw lw_planck_fetch(const LwTab &T, int ib, double tt) {
  int ind = (int)(tt - 159.0);
  if (ind < 1) ind = 1; else if (ind > 180) ind = 180;
  const double *p = T.t + T.totplnk + 181 * ib + (ind - 1);
  return LwPlanckRaw{p[0], p[1], tt - 159.0 - (double)ind};
}
RRTMG_HD double lw_planck_finish(const LwPlanckRaw &r) { return r.p0 + r.frac * (r.p1 - r.p0); }

// Scratch rows per (layer, item): atrans and bbugas (+ atot, bbutot in cloudy layers) as computed by the downward sweep,
// read back by the upward sweep.  (Keeping only the gas optical depth and re-forming the terms going up halves the slab
// and was measured 14 % slower: DESIGN.md 5.)
enum { LF_ATRANS = 0, LF_BBUGAS, LF_ATOT, LF_BBUTOT, LF_N };

// Per-item radiance sink (host emulation, tests and the device kernel): the band-weighted radiances, summed over
// the item's g-points, go to part[item][k][level][column], k = 0 up, 1 down, 2 clear up, 3 clear down,
// (4, 5 = d/dTs of 0, 2 with idrv).
struct LwPartSink {
  double *p;       // part + (slot*nk*(L+1))*N + col
  long N, st;      // st = (L+1)*N
  bool idrv;
  RRTMG_HD void dn(int lev, double rd, double rcd) { part_store(p + st + (long)lev * N, rd); part_store(p + 3 * st + (long)lev * N, rcd); }
  RRTMG_HD void up(int lev, double ru, double rcu, double du, double dcu) {
    part_store(p + (long)lev * N, ru); part_store(p + 2 * st + (long)lev * N, rcu);
    if (idrv) { part_store(p + 4 * st + (long)lev * N, du); part_store(p + 5 * st + (long)lev * N, dcu); }
  }
  // cloud-free column (CLD = false variant): the clear-sky radiances ARE the total ones, so only the total planes
  // are written and lw_flux_level(cld = false) reads them for both outputs (half the partial-plane traffic)
  RRTMG_HD void dn_clear(int lev, double rd) { part_store(p + st + (long)lev * N, rd); }
  RRTMG_HD void up_clear(int lev, double ru, double du) {
    part_store(p + (long)lev * N, ru);
    if (idrv) part_store(p + 4 * st + (long)lev * N, du);
  }
};
RRTMG_HD LwPartSink lw_part_sink(const LwDev &d, int slot, int col) {
  LwPartSink s;
  const int nk = d.idrv ? 6 : 4;
  s.N = d.pcols; s.st = (long)(d.nlay + 1) * d.pcols; s.idrv = d.idrv != 0;
  s.p = d.part + ((long)slot * nk * (d.nlay + 1)) * d.pcols + (col - d.col0);
  return s;
}

// quotient of the Pade table index x/(bpade + x): the IEEE division (the index decides which table entry is read)
#define LW_TDIV(a, b) ((a) / (b))
// (clamped to the table, 0 .. ntbl: these are per-lane GLOBAL gathers, and an optical depth that is negative or not a number
//  -- input the reference would index out of bounds with -- must not take them out of the allocation; one v_med3_i32)
RRTMG_HD int lw_tblidx(double x) { const int i = (int)x; return i < 0 ? 0 : (i > 10000 ? 10000 : i); }
#define LW_TBLIDX(x) lw_tblidx(x)
// Transmittance and Planck source terms of ONE (layer, g-point) cell from its gas optical depth (already times the
// diffusivity angle, clamped at zero) -- rrtmg_lw_rtrn.f90:342-447 / rrtmg_lw_rtrnmc.f90:342-456.  Called by the downward
// sweep only; the upward sweep reads the terms back from the scratch rows (LF_*).
struct LwLut { const double *exp_tbl, *tau_tbl, *tfn_tbl; };
struct LwCell { double atrans, bbd, bbugas, gassrc, atot, bbdtot, bbutot; };
// The G cells of one layer at once, WITHOUT data-dependent branches around the table lookups: the reference's
// "optically thin: series, else: table" (rrtmg_lw_rtrn.f90:426-447) becomes index 0 for the thin lanes, an unconditional
// gather and a select, so that the G g-points' divisions and gathers are issued back to back and cost ONE trip to the
// tables per layer instead of one per g-point (a branch with loads in it is not if-converted, and a wave waits for its
// loads in order).  Same operations on the same operands per lane: identical values.
// gas part: atrans, bbd, bbugas of the gas optical depth (every layer); cloudy part (CLD, lanes with icldlyr): the terms of
// gas + cloud (rrtmg_lw_rtrnmc.f90:342-456), whose second lookup depends on the first in the optically thick case.
template <int G, bool CLD>
RRTMG_HD void lw_cells(const LwLut &lut, bool icldlyr, const double *odepth_in, const double *odcld, const V<G> &plfv, double blay, double dplankdn,
                       double dplankup, LwCell *c) {
  const double rec_6 = 0.166667;
  bool thin_g[G];
  int itgas[G];
  double facgas[G], od_eff[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const double odepth = odepth_in[g];
    thin_g[g] = odepth <= 0.06;
    const double tblind = LW_TDIV(odepth, kBpade + odepth);
    itgas[g] = thin_g[g] ? 0 : LW_TBLIDX(kTblInt * tblind + 0.5);
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const double odepth = odepth_in[g], plf = plfv[g];
    const double transc = lut.exp_tbl[itgas[g]], tausfac = lut.tfn_tbl[itgas[g]];
    c[g].atrans = thin_g[g] ? (odepth - 0.5 * odepth * odepth) : (1.0 - transc);
    facgas[g] = thin_g[g] ? (rec_6 * odepth) : tausfac;
    c[g].bbd = plf * (blay + facgas[g] * dplankdn);
    c[g].bbugas = plf * (blay + facgas[g] * dplankup);
    od_eff[g] = odepth;
    if constexpr (CLD) od_eff[g] = thin_g[g] ? odepth : lut.tau_tbl[itgas[g]];
    c[g].gassrc = 0.0; c[g].atot = 0.0; c[g].bbdtot = 0.0; c[g].bbutot = 0.0;
  }
  if constexpr (CLD) {
    if (icldlyr) {
      bool thin_t[G];
      int ittot[G];
      double odtot[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        // (odtot < 0.06 implies odepth < 0.06: the three cases of the reference are thin/thin, thin/thick, thick/thick)
        odtot[g] = od_eff[g] + odcld[g];
        thin_t[g] = thin_g[g] && (odtot[g] < 0.06);
        const double tblind = LW_TDIV(odtot[g], kBpade + odtot[g]);
        ittot[g] = thin_t[g] ? 0 : LW_TBLIDX(kTblInt * tblind + 0.5);
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const double plf = plfv[g];
        const double transt = lut.exp_tbl[ittot[g]], tfactot = lut.tfn_tbl[ittot[g]];
        c[g].atot = thin_t[g] ? (odtot[g] - 0.5 * odtot[g] * odtot[g]) : (1.0 - transt);
        const double factot = thin_t[g] ? (rec_6 * odtot[g]) : tfactot;
        const double srcdn = blay + facgas[g] * dplankdn;
        // gassrc = plf*(...)*atrans in the two optically thin cases, atrans*plf*(...) in the thick one: kept as written
        c[g].gassrc = thin_g[g] ? ((plf * srcdn) * c[g].atrans) : ((c[g].atrans * plf) * srcdn);
        c[g].bbdtot = plf * (blay + factot * dplankdn);
        c[g].bbutot = plf * (blay + factot * dplankup);
      }
    }
  }
}
// One (column, work item): rtrn / rtrnmc for the item's G g-points (rrtmg_lw_rtrn.f90:324-525).  The layer
// state, the species mixtures, the Planck functions and the cloud optics are evaluated once for the G g-points.
// Radiances leave through `sink` weighted by wtdiff*delwave(band) (rrtmg_lw_rtrn.f90:530-543) and summed over
// the item's g-points in g-point order.
// CLD = false: the caller guarantees a cloud-free column (the cloud code is compiled out).
// MR = true (non-McICA icld >= 2): rtrnmr, maximum/random overlap of the cloudy layers (rrtmg_lw_rtrnmr.f90:454-700)
// with the column's overlap factors from lw_mr_column; MR = false: rtrn / rtrnmc.
template <int BAND, int G, bool CLD, bool MR, bool LDSK, class Sink>
RRTMG_HD void lw_solve_thread(const LwDev &d, const LwTab &T, int col, int ig0, double *scr, long stride, Sink &sink, const double *kb) {
  const int L = d.nlay, N = d.ncol;
  const int ib = BAND - 1;
  const int iw0 = T.b[ib].gs + ig0;
  const double *t = T.t;
  const LwLut lut{t + T.exp_tbl, t + T.tau_tbl, t + T.tfn_tbl};
  const int laytrop = d.laytrop[col];
  const double secd = d.secdiff[(long)ib * N + col];
  const double wtdiff = 0.5, delw = t[T.delwave + ib];
  // scratch slab of this (tile, item): [layer][field][lane][G] -- the G values of a lane are one 16/32-byte access;
  // scr points at this lane's first element, stride = lanes per row (64 on the device, 1 in the host emulation)
  auto SP = [&](int f, int l) -> double * { return scr + ((long)l * LF_N + f) * stride * G; };
  auto W = [&](double r) { return (r * wtdiff) * delw; };

  // cloud bookkeeping
  const bool clouds = CLD && d.icld >= 1 && d.cldfr != nullptr;
  int cb = 0;           // cloud band index used for odcld / efclfrac (non-McICA)
  double secd_cb = secd;
  if (clouds && !d.mcica) {
    const int ncb = d.ncbands[col];
    const int ipat1[16] = {1, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 5};
    cb = ncb == 1 ? 0 : (ncb == 5 ? ipat1[ib] - 1 : ib);
    secd_cb = d.secdiff[(long)cb * N + col];   // odcld(lay,ib) = secdiff(ib)*taucloud(lay,ib): cloud-band index
  }

  // McICA cloud-mask words of the 64-layer block the sweeps are in (the G sub-columns of the item + the OR over
  // all sub-columns): one 8-byte read per 64 layers instead of one per layer
  uint64_t mw[G], aw = 0;
  int mword = -1;
#pragma unroll
  for (int g = 0; g < G; ++g) mw[g] = 0;
  auto mask_words = [&](int l) {
    const int w = l >> 6;
    if (w != mword) {
#pragma unroll
      for (int g = 0; g < G; ++g) mw[g] = d.mask[((long)(iw0 + g) * d.nw + w) * N + col];
      aw = d.anymask[(long)w * N + col];
      mword = w;
    }
  };

  // ---- downward sweep, lev = L .. 1 ---------------------------------------------------------
  double radld[G], radclrd[G], plfrac_bot[G];
  double cldrad[G], clrrad[G], radmr[G];   // rtrnmr: cloudy / clear parts of the radiance and the overlap carry `rad`
  int iclddn[G];
#pragma unroll
  for (int g = 0; g < G; ++g) { radld[g] = 0.0; radclrd[g] = 0.0; plfrac_bot[g] = 0.0; iclddn[g] = 0; cldrad[g] = 0.0; clrrad[g] = 0.0; radmr[g] = 0.0; }
  if constexpr (CLD) sink.dn(L, 0.0, 0.0); else sink.dn_clear(L, 0.0);
  double plev_up = lw_planck_finish(lw_planck_fetch(T, ib, d.tlev[(long)L * N + col]));   // Planck function at the interface above the layer
  RRTMG_PH_DECL
  RRTMG_PH_MARK(0, plev_up)      // 0: setup before the sweep
  for (int lev = L; lev >= 1; --lev) {
    const int l = lev - 1;
    const long i = (long)l * N + col;
    LwLayerIn s;
    lw_load_layer(d, col, l, s);
    RRTMG_PH_MARK(1, s.fac00 + s.colh2o + (double)s.jp)      // 1: the layer's prep rows have arrived
    V<G> plfrac;
    // the layer's aerosol / temperature rows and the two Planck-table entries of each temperature are REQUESTED here and used
    // after taumol: the gathers fly under taumol's table rows (LDS) and arithmetic.  The interface value is carried down from
    // the layer above (the same two entries, the same arithmetic as interpolating it again).
    const double taua = d.tauaer ? d.tauaer[((long)ib * L + l) * N + col] : 0.0;
    const LwPlanckRaw q_lay = lw_planck_fetch(T, ib, d.tlay[i]), q_dn = lw_planck_fetch(T, ib, d.tlev[i]);
    const V<G> taug = lw_taug<BAND, G, LDSK>(T, s, lev <= laytrop, ig0, plfrac, kb);
    RRTMG_PH_MARK(2, taug[0] + taug[G - 1] + plfrac[0])      // 2: taumol (LDS row gathers + arithmetic)
    const double blay = lw_planck_finish(q_lay);
    const double plev_dn = lw_planck_finish(q_dn);
    const double dplankup = plev_up - blay;
    const double dplankdn = plev_dn - blay;
    plev_up = plev_dn;
    RRTMG_PH_MARK(3, blay + dplankup + dplankdn + taua)      // 3: aerosol / temperature rows and the three Planck interpolations
    // band-level cloud state of this layer (shared by the g-points)
    bool icldlyr = false, cld_band = false;
    double cfrac_band = 0.0, odcld_band = 0.0, efcl_band = 0.0;
    if (clouds) {
      if (d.mcica) {
        mask_words(l);
        icldlyr = (aw >> (l & 63)) & 1ull;
        if (icldlyr) {
          odcld_band = secd * d.ctau[((long)ib * L + l) * N + col];
          efcl_band = (1.0 - exp(-odcld_band)) * 1.0;
        }
      } else {
        cfrac_band = d.cldfr[i];
        if (cfrac_band >= 1.e-6) {
          icldlyr = true; cld_band = true;
          odcld_band = secd_cb * d.ctau[((long)cb * L + l) * N + col];
          efcl_band = (1. - exp(-odcld_band)) * cfrac_band;
        }
      }
    }
    double mr_start = 0.0, mr_clr1 = 0.0, mr_cld1 = 0.0, mr_cmb1 = 0.0, mr_cmb2 = 0.0, mr_clr2 = 0.0, mr_cld2 = 0.0;
    if (MR && icldlyr) {
      mr_start = d.mr[lw_mr_off(L, col, lev) + MR_ISTCLDD * 64];
      const double *q = d.mr + lw_mr_off(L, col, lev - 1);
      mr_clr1 = q[MR_FACCLR1D * 64]; mr_cld1 = q[MR_FACCLD1D * 64]; mr_cmb1 = q[MR_FACCMB1D * 64]; mr_cmb2 = q[MR_FACCMB2D * 64];
      mr_clr2 = q[MR_FACCLR2D * 64]; mr_cld2 = q[MR_FACCLD2D * 64];
    }
    double srd = 0.0, srcd = 0.0;
    V<G> v_atrans, v_bbugas, v_atot, v_bbutot, v_od;
    double cfrac_g[G], odcld_g[G], efcl_g[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const double taut = taug[g] + taua;
      double odepth = secd * taut;
      if (odepth < 0.0) odepth = 0.0;
      v_od[g] = odepth;
      cfrac_g[g] = 0.0; odcld_g[g] = 0.0; efcl_g[g] = 0.0;
      if (icldlyr) {
        if (d.mcica) {
          if ((mw[g] >> (l & 63)) & 1ull) { cfrac_g[g] = 1.0; odcld_g[g] = odcld_band; efcl_g[g] = efcl_band; }
        } else if (cld_band) {
          cfrac_g[g] = cfrac_band; odcld_g[g] = odcld_band; efcl_g[g] = efcl_band;
        }
      }
    }
    LwCell cells[G];
    lw_cells<G, CLD>(lut, icldlyr, v_od.v, odcld_g, plfrac, blay, dplankdn, dplankup, cells);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const double plf = plfrac[g];
      const double cfrac = cfrac_g[g], efclfrac = efcl_g[g];
      const LwCell &c = cells[g];
      const double atrans = c.atrans, bbd = c.bbd, bbugas = c.bbugas;
      if (icldlyr) {
        iclddn[g] = 1;
        const double gassrc = c.gassrc, atot = c.atot, bbdtot = c.bbdtot, bbutot = c.bbutot;
        if constexpr (MR) {
          if (mr_start == 1.0) { cldrad[g] = cfrac * radld[g]; clrrad[g] = radld[g] - cldrad[g]; radmr[g] = 0.0; }
          const double ttot = 1.0 - atot;
          const double cldsrc = bbdtot * atot;
          cldrad[g] = cldrad[g] * ttot + cfrac * cldsrc;
          clrrad[g] = clrrad[g] * (1.0 - atrans) + (1.0 - cfrac) * gassrc;
          radld[g] = cldrad[g] + clrrad[g];
          const double radmod = radmr[g] * (mr_clr1 * (1. - atrans) + mr_cld1 * ttot) - mr_cmb1 * gassrc + mr_cmb2 * cldsrc;
          const double oldcld = cldrad[g] - radmod, oldclr = clrrad[g] + radmod;
          radmr[g] = -radmod + mr_clr2 * oldclr - mr_cld2 * oldcld;
          cldrad[g] = cldrad[g] + radmr[g];
          clrrad[g] = clrrad[g] - radmr[g];
        } else {
          radld[g] = radld[g] - radld[g] * (atrans + efclfrac * (1. - atrans)) + gassrc + cfrac * (bbdtot * atot - gassrc);
        }
        v_atot[g] = atot;
        v_bbutot[g] = bbutot;
      } else {
        radld[g] = radld[g] + (bbd - radld[g]) * atrans;
      }
      v_atrans[g] = atrans;
      v_bbugas[g] = bbugas;
      if (iclddn[g] == 1) {
        radclrd[g] = radclrd[g] + (bbd - radclrd[g]) * atrans;
      } else {
        radclrd[g] = radld[g];
      }
      srd = srd + W(radld[g]); srcd = srcd + W(radclrd[g]);
      plfrac_bot[g] = plf;
    }
    RRTMG_PH_MARK(4, radld[0] + radld[G - 1] + srd)      // 4: table index, exp/tfn lookups, recurrence of the G g-points
    scr_store<G>(SP(LF_ATRANS, l), stride, v_atrans);
    scr_store<G>(SP(LF_BBUGAS, l), stride, v_bbugas);
    if (icldlyr) { scr_store<G>(SP(LF_ATOT, l), stride, v_atot); scr_store<G>(SP(LF_BBUTOT, l), stride, v_bbutot); }
    (void)v_od;
    if constexpr (CLD) sink.dn(lev - 1, srd, srcd); else sink.dn_clear(lev - 1, srd);
    RRTMG_PH_MARK(5, srd)      // 5: scratch rows and partial sums stored
  }

  // ---- surface ------------------------------------------------------------------------------
  const double semiss = d.emis[(long)ib * N + col];
  const double tbound = d.tsfc[col];
  const double plankbnd = semiss * lw_planck(T, ib, tbound);
  const double reflect = 1.0 - semiss;
  double radlu[G], radclru[G], d_radlu_dt[G], d_radclru_dt[G];
  {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const double rad0 = plfrac_bot[g] * plankbnd;
      radlu[g] = rad0 + reflect * radld[g];
      radclru[g] = rad0 + reflect * radclrd[g];
      d_radlu_dt[g] = 0.0; d_radclru_dt[g] = 0.0;
      if (d.idrv) {
        const double d_rad0_dt = plfrac_bot[g] * (semiss * lw_planck_deriv(T, ib, tbound));
        d_radlu_dt[g] = d_rad0_dt; d_radclru_dt[g] = d_rad0_dt;
      }
      s0 = s0 + W(radlu[g]); s1 = s1 + W(radclru[g]); s2 = s2 + W(d_radlu_dt[g]); s3 = s3 + W(d_radclru_dt[g]);
    }
    if constexpr (CLD) sink.up(0, s0, s1, s2, s3); else sink.up_clear(0, s0, s2);
  }

  // ---- upward sweep: kU layers at a time, their scratch rows are loaded before the first is used ----------
  constexpr int kU = 4;
  for (int lev0 = 1; lev0 <= L; lev0 += kU) {
    V<G> r_atrans[kU], r_bbugas[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (lev0 + u <= L) { r_atrans[u] = scr_load<G>(SP(LF_ATRANS, lev0 + u - 1), stride); r_bbugas[u] = scr_load<G>(SP(LF_BBUGAS, lev0 + u - 1), stride); }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int lev = lev0 + u;
      if (lev > L) break;
      const int l = lev - 1;
      bool icldlyr = false, cld_band = false;
      double cfrac_band = 0.0, efcl_band = 0.0, odcld_band = 0.0;
      if (clouds) {
        if (d.mcica) {
          mask_words(l);
          icldlyr = (aw >> (l & 63)) & 1ull;
          if (icldlyr) {
            odcld_band = secd * d.ctau[((long)ib * L + l) * N + col];
            efcl_band = (1.0 - exp(-odcld_band)) * 1.0;
          }
        } else {
          cfrac_band = d.cldfr[(long)l * N + col];
          if (cfrac_band >= 1.e-6) {
            icldlyr = true; cld_band = true;
            odcld_band = secd_cb * d.ctau[((long)cb * L + l) * N + col];
            efcl_band = (1. - exp(-odcld_band)) * cfrac_band;
          }
        }
      }
      V<G> r_atot, r_bbutot;
      if (icldlyr) { r_atot = scr_load<G>(SP(LF_ATOT, l), stride); r_bbutot = scr_load<G>(SP(LF_BBUTOT, l), stride); }
      double mr_start = 0.0, mr_clr1 = 0.0, mr_cld1 = 0.0, mr_cmb1 = 0.0, mr_cmb2 = 0.0, mr_clr2 = 0.0, mr_cld2 = 0.0;
      if (MR && icldlyr) {
        mr_start = d.mr[lw_mr_off(L, col, lev) + MR_ISTCLD * 64];
        const double *q = d.mr + lw_mr_off(L, col, lev + 1);
        mr_clr1 = q[MR_FACCLR1 * 64]; mr_cld1 = q[MR_FACCLD1 * 64]; mr_cmb1 = q[MR_FACCMB1 * 64]; mr_cmb2 = q[MR_FACCMB2 * 64];
        mr_clr2 = q[MR_FACCLR2 * 64]; mr_cld2 = q[MR_FACCLD2 * 64];
      }
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      double cfrac_g[G], odcld_g[G], efcl_g[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        cfrac_g[g] = 0.0; odcld_g[g] = 0.0; efcl_g[g] = 0.0;
        if (icldlyr) {
          if (d.mcica) {
            if ((mw[g] >> (l & 63)) & 1ull) { cfrac_g[g] = 1.0; efcl_g[g] = efcl_band; odcld_g[g] = odcld_band; }
          } else if (cld_band) {
            cfrac_g[g] = cfrac_band; efcl_g[g] = efcl_band; odcld_g[g] = odcld_band;
          }
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const double cfrac = cfrac_g[g], efclfrac = efcl_g[g];
        const double atrans = r_atrans[u][g], bbugas = r_bbugas[u][g];
        if (icldlyr) {
          const double atot = r_atot[g], bbutot = r_bbutot[g];
          const double gassrc = bbugas * atrans;
          if constexpr (MR) {
            if (mr_start == 1.0) { cldrad[g] = cfrac * radlu[g]; clrrad[g] = radlu[g] - cldrad[g]; radmr[g] = 0.0; }
            const double ttot = 1.0 - atot;
            const double cldsrc = bbutot * atot;
            cldrad[g] = cldrad[g] * ttot + cfrac * cldsrc;
            clrrad[g] = clrrad[g] * (1.0 - atrans) + (1.0 - cfrac) * gassrc;
            radlu[g] = cldrad[g] + clrrad[g];
            const double radmod = radmr[g] * (mr_clr1 * (1.0 - atrans) + mr_cld1 * ttot) - mr_cmb1 * gassrc + mr_cmb2 * cldsrc;
            const double oldcld = cldrad[g] - radmod, oldclr = clrrad[g] + radmod;
            radmr[g] = -radmod + mr_clr2 * oldclr - mr_cld2 * oldcld;
            cldrad[g] = cldrad[g] + radmr[g];
            clrrad[g] = clrrad[g] - radmr[g];
          } else {
            radlu[g] = radlu[g] - radlu[g] * (atrans + efclfrac * (1.0 - atrans)) + gassrc + cfrac * (bbutot * atot - gassrc);
          }
          if (d.idrv) d_radlu_dt[g] = d_radlu_dt[g] * cfrac * (1.0 - atot) + d_radlu_dt[g] * (1.0 - cfrac) * (1.0 - atrans);
        } else {
          radlu[g] = radlu[g] + (bbugas - radlu[g]) * atrans;
          if (d.idrv) d_radlu_dt[g] = d_radlu_dt[g] * (1.0 - atrans);
        }
        if (iclddn[g] == 1) {
          radclru[g] = radclru[g] + (bbugas - radclru[g]) * atrans;
          if (d.idrv) d_radclru_dt[g] = d_radclru_dt[g] * (1.0 - atrans);
        } else {
          radclru[g] = radlu[g];
          if (d.idrv) d_radclru_dt[g] = d_radlu_dt[g];
        }
        s0 = s0 + W(radlu[g]); s1 = s1 + W(radclru[g]); s2 = s2 + W(d_radlu_dt[g]); s3 = s3 + W(d_radclru_dt[g]);
      }
      if constexpr (CLD) sink.up(lev, s0, s1, s2, s3); else sink.up_clear(lev, s0, s2);
    }
  }
  RRTMG_PH_MARK(6, radlu[0])      // 6: surface + the whole upward sweep
  RRTMG_PH_FLUSH(d)
}

// Dispatch of one work item (packed, see LwTab) for one column: band switch + G in {4, 2}.
template <int BAND, bool CLD, bool MR, bool LDSK, class Sink>
RRTMG_HD void lw_solve_band(const LwDev &d, const LwTab &T, int g, int col, int ig0, double *scr, long stride, Sink &sink, const double *kb) {
  constexpr int ng = kLwNg[BAND - 1];
  if constexpr (ng >= 4) {
    if (g == 4) { lw_solve_thread<BAND, 4, CLD, MR, LDSK>(d, T, col, ig0, scr, stride, sink, kb); return; }
  }
  if constexpr (ng % 4 != 0) lw_solve_thread<BAND, 2, CLD, MR, LDSK>(d, T, col, ig0, scr, stride, sink, kb);
}
// LDSK / kb: see lw_taug (kb = the workgroup's LDS slice of the item's band slab, or nullptr with LDSK = false)
template <bool CLD, bool MR, bool LDSK = false, class Sink>
RRTMG_HD void lw_solve_item(const LwDev &d, const LwTab &T, int item, int col, double *scr, long stride, Sink &sink, const double *kb = nullptr) {
  const int g = (item >> 16) & 0xf, ig0 = (item >> 8) & 0xff;
  switch ((item & 0xff) + 1) {
    case 1: lw_solve_band<1, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    case 2: lw_solve_band<2, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    case 3: lw_solve_band<3, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    case 4: lw_solve_band<4, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    case 5: lw_solve_band<5, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    case 6: lw_solve_band<6, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    case 7: lw_solve_band<7, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    case 8: lw_solve_band<8, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    case 9: lw_solve_band<9, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    case 10: lw_solve_band<10, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    case 11: lw_solve_band<11, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    case 12: lw_solve_band<12, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    case 13: lw_solve_band<13, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    case 14: lw_solve_band<14, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    case 15: lw_solve_band<15, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
    default: lw_solve_band<16, CLD, MR, LDSK>(d, T, g, col, ig0, scr, stride, sink, kb); break;
  }
}

// band / g-point integration and heating rates (rrtmg_lw_rtrn.f90:528-585)
// one thread per (column, interface level)
// nparts = number of work items (T.nitem); each partial is the sum over its item's g-points and already carries
// wtdiff*delwave(band)
// out[0..5] = uflx, dflx, uflxc, dflxc, duflx_dt, duflxc_dt of (column, level), scaled by fluxfac
RRTMG_HD void lw_flux_sums(const LwDev &d, int col, int lev, int nparts, bool cld, double *out) {
  const int L = d.nlay;
  const int nk = d.idrv ? 6 : 4;
  const long st = (long)(L + 1) * d.pcols;
  double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0, t4 = 0.0, t5 = 0.0;
  for (int iw = 0; iw < nparts; ++iw) {
    const double *p = d.part + ((long)iw * nk * (L + 1) + lev) * d.pcols + (col - d.col0);
    t0 = t0 + part_load(p); t1 = t1 + part_load(p + st);
    if (d.idrv) t4 = t4 + part_load(p + 4 * st);
    if (cld) {
      t2 = t2 + part_load(p + 2 * st); t3 = t3 + part_load(p + 3 * st);
      if (d.idrv) t5 = t5 + part_load(p + 5 * st);
    }
  }
  if (!cld) { t2 = t0; t3 = t1; t5 = t4; }   // the clear-sky variant wrote the total planes only (LwPartSink::dn_clear)
  out[0] = t0 * d.fluxfac; out[1] = t1 * d.fluxfac; out[2] = t2 * d.fluxfac; out[3] = t3 * d.fluxfac;
  out[4] = t4 * d.fluxfac; out[5] = t5 * d.fluxfac;
}
RRTMG_HD void lw_flux_level(const LwDev &d, const LwTab &T, int col, int lev, int nparts, bool cld) {
  (void)T;
  double f[6];
  lw_flux_sums(d, col, lev, nparts, cld, f);
  const long o = (long)lev * d.ncol + col;
  d.uflx[o] = f[0]; d.dflx[o] = f[1]; d.uflxc[o] = f[2]; d.dflxc[o] = f[3];
  if (d.idrv) { d.duflx_dt[o] = f[4]; d.duflxc_dt[o] = f[5]; }
}
// one thread per (column, layer)
RRTMG_HD void lw_heat_layer(const LwDev &d, const LwTab &T, int col, int lay) {
  const int N = d.ncol;
  const long o0 = (long)lay * N + col, o1 = o0 + N;
  const double fnet0 = d.uflx[o0] - d.dflx[o0], fnet1 = d.uflx[o1] - d.dflx[o1];
  const double fnetc0 = d.uflxc[o0] - d.dflxc[o0], fnetc1 = d.uflxc[o1] - d.dflxc[o1];
  const double dp = d.plev[o0] - d.plev[o1];
  d.hr[o0] = T.heatfac * (fnet0 - fnet1) / dp;
  d.hrc[o0] = T.heatfac * (fnetc0 - fnetc1) / dp;
}

}  // namespace rrtmg
