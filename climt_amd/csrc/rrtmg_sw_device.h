// rrtmg_sw_device.h -- RRTMG shortwave hot path as per-thread device functions (gfx950).
//
// Decomposition (MI355X-first, not the reference's per-column serial loops); kernels in rrtmg_sw.hip:
//   sw_prep_layer    one thread per (column, layer): inatm_sw + setcoef_sw -> one row set of the prep slab
//   sw_prep_column   one thread per (column, band): laytrop, cloud flags, the layer each band takes its solar source from
//   sw_cloud_layer   one thread per (column, layer): cldprop_sw / cldprmc_sw band optics (delta-scaled)
//   kiss_mask_jump   one thread per (column, sub-column): kissvec sub-column cloud mask, bit-packed, by jump-ahead
//   sw_solve_thread  one thread per (column, work item); a work item = 4 or 2 consecutive g-points of ONE band carried
//                    through both sweeps by the thread, a wavefront = 64 columns x one item: band control flow is
//                    wave-uniform, every per-column input is a coalesced 512-B row, the item's k-distribution slice
//                    [row][G] sits in LDS.  Sweep 1 (bottom-up): taumol + delta scaling + reftra + upward adding
//                    recurrence, (rup, rupd) spilled to a [layer][field][lane][G] scratch slab; sweep 2 (top-down):
//                    the layer optics again (recomputed, not stored), downward recurrence and the fluxes at every
//                    interface, summed over the item's g-points into part[slot][k][level][column].
//   sw_flux_level / sw_heat_layer  spectral integration in g-point order; heating rates.
//
// Reference followed (climt/_lib/rrtmg_sw/): rrtmg_sw_rad.nomcica.f90:587-816 (driver),
// :846-1539 (inatm_sw), rrtmg_sw_setcoef.f90:49-305, rrtmg_sw_taumol.f90:50-1790,
// rrtmg_sw_cldprop.f90:53-365, rrtmg_sw_cldprmc.f90:53-349, rrtmg_sw_spcvrt.f90:53-667,
// rrtmg_sw_spcvmc.f90, rrtmg_sw_reftra.f90:48-324, rrtmg_sw_vrtqdr.f90:47-171,
// mcica_subcol_gen_sw.f90:182-591.
#pragma once
#include "rrtmg_common.h"

namespace rrtmg {

constexpr int kSwNBand = 14;
constexpr int kSwNGpt = 112;

struct SwBandTab {
  int ng, gs;       // g-points in band, first g-point (0-based) of band
  int nfor, nsrc;   // rows of forref (3|4); source mixtures (1, 5 or 9)
  // Every per-g-point table of taumol lives in ONE g-point-fastest slab [nrows][ng] at T.t + slab (built at init);
  // r_* = first row of each table in it.  A work item's slice of the slab -- columns ig0 .. ig0+G-1, [nrows][G] --
  // is what the solve kernels stage in LDS, rows unchanged.
  long slab;
  int nrows;
  int r_absa, r_absb, r_self, r_forr;
  int r_rayl, r_raylb;         // rayl: 1 row (or 9 rows, band 24); raylb band 24 upper
  int r_ex1, r_ex2;            // extra absorber rows (lower / upper)
  long sflux, irr, fac, sns;   // [js][ig]
};
constexpr int kSwSlabMaxRows = 1800;   // bands 17 / 21 / 28: 585 + 1175 + 10 + 4 + ... rows (checked at init)

// Work item of the solve kernel: G consecutive g-points of one band -- chunks of 4, then 2 where a band's count is not a
// multiple of 4 -- carried by one thread per column.  Packed band | ig0 << 8 | G << 16 | (first g-point of the whole
// spectrum) << 20.  sched[] lists the items heaviest first (launch order).  Partial fluxes: one slot per item, holding
// the sum of its pairs, (pair0 + pair1); the flux kernel adds the slots in item order -- the 112 g-points are added in the
// same order whichever kernel variant (cloud-free / cloudy tile) a tile ran.
constexpr int kSwMaxItem = 32;
constexpr int kSwNSlot = 32;
RRTMG_HD int item_band(int it) { return it & 0xff; }
RRTMG_HD int item_ig0(int it) { return (it >> 8) & 0xff; }
RRTMG_HD int item_g(int it) { return (it >> 16) & 0xf; }
RRTMG_HD int item_iw0(int it) { return (it >> 20) & 0xff; }

struct SwTab {
  const double *t;
  SwBandTab b[kSwNBand];
  int nitem;
  int32_t item[kSwMaxItem], sched[kSwMaxItem];
  long preflog, tref, exp_tbl;
  long extliq1, ssaliq1, asyliq1, extice2, ssaice2, asyice2, extice3, ssaice3, asyice3, fdlice3;
  long abari, bbari, cbari, dbari, ebari, fbari, wavenum2;
  long rsrtaua, rsrpiza, rsrasya;
  double heatfac;
};

// everything a launch needs; all arrays device-resident, [layer][column] with column fastest
struct SwDev {
  int ncol, nlay;
  int icld, iaer, inflag, iceflag, liqflag, mcica, isolvar;
  double adjflux;           // Earth-Sun factor (isolvar >= 0) -- per band array below for isolvar < 0
  double adjflux_b[kSwNBand];
  double svar_f, svar_s, svar_i;
  double svar_b[kSwNBand];  // isolvar == 3: per-band multiplier (same for f, s, i)
  const double *svar_col;   // [3][ncol] per-column (svar_f, svar_s, svar_i) or null (see sw_scalar_setup)
  Constants k;
  // inputs
  const double *play, *plev, *tlay, *h2o, *o3, *co2, *ch4, *n2o, *o2;
  const double *asdir, *asdif, *aldir, *aldif, *coszen;
  const double *cldfr, *taucld, *ssacld, *asmcld, *fsfcld, *cicewp, *cliqwp, *reice, *reliq;
  const double *tauaer, *ssaaer, *asmaer;   // effective per-band aerosol [14][lay][col] or null
  // prep products: ONE slab [tile][layer][SP_N fields][64 lanes] -- a wavefront reads the SP_N rows of its
  // (tile, layer) at constant offsets from a single address (sw_prep_off), see enum SwPrepField
  double *prep;
  int32_t *laytrop;    // [col]
  int32_t *laysolfr;   // [14][col], 1-based layer, 0 = source never set
  int32_t *anycld;     // [col] 1 if any layer has cldfr > 0
  int32_t *tile_cld;   // [tile] 1 if any column of the 64-column tile has a cloud (selects the solve kernel variant)
  // The chunk's tiles by variant, compacted in tile order (tile_lists_kernel, behind the chunk's preparation): tlist[v * tcap + i]
  // = i-th tile (index within the chunk) of variant v (0 cloud-free, 1 cloudy), tcnt[v] of them.  A solve workgroup takes
  // consecutive LIST entries, so that all of its wavefronts have work wherever the two kinds of tiles interleave.
  const int32_t *tlist, *tcnt;
  int tcap;
  int32_t *ncloudy;    // number of tiles with tile_cld set, counted by the preparation kernels (rrtmg_ctx::CallHint) ...
  int32_t *hint_out;   // ... and where the call's LAST integration launch leaves it for the host (page-locked; nullptr in the others)
  double *cossza;      // [col]
  double *pdp;         // [lay][col]
  double *ctau, *cssa, *casm;   // delta-scaled cloud optics [14][lay][col]
  uint64_t *mask;      // McICA cloud mask bits [112][nw][col]
  int nw;
  double *scratch;     // [tile][item: first g-point * ...][lay][field][G][64]
  double *part;        // [slot][4][nlay+1][pcols]  weighted (fu, fd, cu, cd) of the columns col0 .. col0+pcols-1
  int col0, pcols;     // column chunk the solve / flux kernels are working on (scratch and part are per chunk)
  RRTMG_PROFILE_FIELDS
  int *err;
  // outputs
  double *swuflx, *swdflx, *swhr, *swuflxc, *swdflxc, *swhrc;
};

enum { SP_H2O = 0, SP_CO2 = 1, SP_O3 = 2, SP_CH4 = 3, SP_O2 = 4 };

// rows of the prep slab; P_IDX holds jp | jt<<8 | jt1<<12 | indself<<16 | indfor<<24 as an (exact) double
enum SwPrepField { P_FAC00 = 0, P_FAC01, P_FAC10, P_FAC11, P_SELFFAC, P_SELFFRAC, P_FORFAC, P_FORFRAC,
                   P_COLH2O, P_COLCO2, P_COLO3, P_COLCH4, P_COLO2, P_COLMOL, P_IDX, SP_N };
RRTMG_HD long sw_prep_off(int nlay, int col, int lay) {
  return ((long)(col >> 6) * nlay + lay) * (SP_N * 64) + (col & 63);
}
RRTMG_HD size_t sw_prep_size(int ncol, int nlay) { return (size_t)((ncol + 63) / 64) * nlay * SP_N * 64; }

// ------------------------------------------------------------------------------------------
// inatm_sw (rrtmg_sw_rad.nomcica.f90:1441-1465) + setcoef_sw (rrtmg_sw_setcoef.f90:137-303)
// ------------------------------------------------------------------------------------------
// layer part: one thread per (column, layer)
// returns the layer's packed index word (P_IDX) so that a caller can keep it at hand for the column part
RRTMG_HD int sw_prep_layer(const SwDev &d, const SwTab &T, int col, int l) {
  const int L = d.nlay, N = d.ncol;
  const double *preflog = T.t + T.preflog, *tref = T.t + T.tref;
  const double amd = 28.9660, amw = 18.0160;
  const double stpfac = 296.0 / 1013.0;
  int lower = 0;
  {
    const long i = (long)l * N + col;
    const double pz0 = d.plev[i], pz1 = d.plev[i + N];
    const double pavel = d.play[i], tavel = d.tlay[i];
    const double wh2o = d.h2o[i];
    const double amm = (1.0 - wh2o) * amd + wh2o * amw;
    const double coldry = (pz0 - pz1) * 1.e3 * d.k.avogad / (1.e2 * d.k.grav * amm * (1.0 + wh2o));
    d.pdp[i] = pz0 - pz1;
    const double w1 = coldry * wh2o, w2 = coldry * d.co2[i], w3 = coldry * d.o3[i];
    const double w4 = coldry * d.n2o[i], w6 = coldry * d.ch4[i], w7 = coldry * d.o2[i];
    (void)w4;

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
    int indself, indfor;
    double forfac, forfrac, selffac, selffrac;
    if (plog > 4.56) {
      lower = 1;
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
      double factor = (tavel - 188.0) / 36.0;
      indfor = 3;
      forfrac = factor - 1.0;
      selffac = 0.0;
      selffrac = 0.0;
      indself = 0;
    }
    double colh2o = 1.e-20 * w1, colco2 = 1.e-20 * w2, colo3 = 1.e-20 * w3;
    double colch4 = 1.e-20 * w6, colo2 = 1.e-20 * w7;
    const double colmol = 1.e-20 * coldry + colh2o;
    if (colco2 == 0.0) colco2 = 1.e-32 * coldry;
    if (colch4 == 0.0) colch4 = 1.e-32 * coldry;
    if (colo2 == 0.0) colo2 = 1.e-32 * coldry;
    const double compfp = 1.0 - fp;
    double *q = d.prep + sw_prep_off(L, col, l);
    q[P_FAC10 * 64] = compfp * ft;
    q[P_FAC00 * 64] = compfp * (1.0 - ft);
    q[P_FAC11 * 64] = fp * ft1;
    q[P_FAC01 * 64] = fp * (1.0 - ft1);
    q[P_SELFFAC * 64] = selffac; q[P_SELFFRAC * 64] = selffrac; q[P_FORFAC * 64] = forfac; q[P_FORFRAC * 64] = forfrac;
    q[P_COLH2O * 64] = colh2o; q[P_COLCO2 * 64] = colco2; q[P_COLO3 * 64] = colo3; q[P_COLCH4 * 64] = colch4;
    q[P_COLO2 * 64] = colo2; q[P_COLMOL * 64] = colmol;
    // bit 28: layer is in the lower atmosphere (counted into laytrop by the column part)
    const int packed = jp | (jt << 8) | (jt1 << 12) | (indself << 16) | (indfor << 24) | (lower << 28);
    q[P_IDX * 64] = (double)packed;
    if (d.icld >= 1 && d.cldfr) {
      const double cf = d.cldfr[i];
      // rrtmg_sw_rad.nomcica.f90:616-620
      if (!d.mcica && cf > 1.e-6 && cf < 1.0 - 1.e-6) report_error(d.err, RRTMG_ERR_PARTIAL_CLOUD);
    }
    return packed;
  }
}

// column part (after every layer of the column is done): laytrop, cloud flag, zenith angle, solar-source layers
// Bands b0 .. b1-1 of the solar-source bookkeeping; the column scalars (laytrop, cloud flag, clamped cos(zenith)) are
// written by the caller that owns band 0.  The device runs one thread per (column, band) -- the per-band state
// machines are ~500 instructions each, one thread for all 14 was the whole kernel time.
// idx: optional copy of the column's packed index words, idx[l * idx_stride] (the fused kernel's LDS copy); else the slab.
RRTMG_HD void sw_prep_column(const SwDev &d, const SwTab &T, int col, int b0 = 0, int b1 = kSwNBand, const int *idx = nullptr, int idx_stride = 0) {
  (void)T;
  const int L = d.nlay, N = d.ncol;
  int laytrop = 0, anycld = 0;
  auto packed = [&](int l) { return idx ? idx[l * idx_stride] : (int)d.prep[sw_prep_off(L, col, l) + P_IDX * 64]; };
#pragma unroll 8   // independent loads: keep several layers in flight
  for (int l = 0; l < L; ++l) {
    laytrop += (packed(l) >> 28) & 1;
    if (b0 == 0 && d.icld >= 1 && d.cldfr && d.cldfr[(long)l * N + col] > 0.0) anycld = 1;
  }
  if (b0 == 0) {
    d.laytrop[col] = laytrop;
    d.anycld[col] = anycld;
    double cz = d.coszen[col];
    if (cz < 1.e-10) cz = 1.e-10;   // rrtmg_sw_rad.nomcica.f90:641-642
    d.cossza[col] = cz;
  }

  // layer at which each band takes its solar source term (rrtmg_sw_taumol.f90, per-band
  // laysolfr logic; table SURVEY.md A.2).  Emulates the sequential update-and-test of the
  // reference loops, result = last layer for which (lay == laysolfr) held.
  const int layreffr[kSwNBand] = {18, 30, 6, 3, 3, 8, 2, 6, 1, 2, 0, 32, 58, 49};
  const bool upper[kSwNBand] = {true, true, false, false, false, false, false, false, false, false, false, true, true, true};
  auto jp_of = [&](int lay0) { return packed(lay0) & 0xff; };
  for (int b = b0; b < b1; ++b) {
    // one pass over the layers with a sliding (previous, current, next) window of jp
    const int ref = layreffr[b];
    const bool up = upper[b];
    int ls = up ? L : laytrop, fin = 0;
    int jpm = 0, jpc = jp_of(0);
#pragma unroll 8
    for (int lay = 1; lay <= L; ++lay) {
      const int jpn = (lay < L) ? jp_of(lay) : 0;
      if (up) {
        if (lay > laytrop) {
          if (jpm < ref && jpc >= ref) ls = lay;
          if (lay == ls) fin = lay;
        }
      } else if (lay <= laytrop) {
        if (b != 10) {  // band 26 has no layreffr test
          if (jpc < ref && jpn >= ref) ls = (lay + 1 < laytrop) ? lay + 1 : laytrop;
        }
        if (lay == ls) fin = lay;
      }
      jpm = jpc; jpc = jpn;
    }
    d.laysolfr[(long)b * N + col] = fin;
  }
}

// ------------------------------------------------------------------------------------------
// cloud optics by band for one (column, layer): cldprop_sw (rrtmg_sw_cldprop.f90:113-360);
// cldprmc_sw computes the same numbers per g-point with ib = ngb(ig) (rrtmg_sw_cldprmc.f90:103-345)
// so one band value serves every cloudy sub-column of the band.
// ------------------------------------------------------------------------------------------
RRTMG_HD void sw_cloud_layer(const SwDev &d, const SwTab &T, int col, int lay) {
  const int L = d.nlay, N = d.ncol;
  const long i = (long)lay * N + col;
  const double cldmin = 1.e-20, eps = 1.e-06;
  const double *t = T.t;
  const double ciwp = d.cicewp ? d.cicewp[i] : 0.0, clwp = d.cliqwp ? d.cliqwp[i] : 0.0;
  const double cwp = ciwp + clwp;
  const double cf = d.cldfr[i];
  double tauctot = 0.0;
  if (d.taucld) {
    for (int b = 0; b < kSwNBand; ++b) tauctot = tauctot + d.taucld[i * kSwNBand + b];
  }
  // McICA gate is per sub-column (cldfmc >= cldmin and (cwp >= cldmin or taucmc >= cldmin)) with the
  // band's tauc; nomcica gate uses the band-summed tauctot.
  // The reference stops at the FIRST failed check in program order (layer, then g-point / band, then the order of the
  // source lines); a thread keeps the first code it meets (`chk`) and skips the arithmetic behind it, across threads the
  // largest code wins (report_error).  One code per distinct `stop` message: include/rrtmg_hip.h.
  int e = 0;
  auto chk = [&](bool bad, int code) { if (e == 0 && bad) e = code; };
  for (int b = 0; b < kSwNBand; ++b) {
    const long o = ((long)b * L + lay) * N + col;
    double tau = 0.0, ssa = 1.0, asy = 0.0;
    const double tcb = d.taucld ? d.taucld[i * kSwNBand + b] : 0.0;
    // (McICA: a sub-column can only be cloudy where cldfrac >= cldmin, mcica_subcol_gen_sw.f90:474-497)
    const bool gate = cf >= cldmin && (d.mcica ? (cwp >= cldmin || tcb >= cldmin) : (cwp >= cldmin || tauctot >= cldmin));
    if (d.mcica) tau = tcb, ssa = d.ssacld ? d.ssacld[i * kSwNBand + b] : 1.0, asy = d.asmcld ? d.asmcld[i * kSwNBand + b] : 0.0;
    if (gate) {
      if (d.inflag == 0) {
        const double ffp = d.fsfcld[i * kSwNBand + b];
        const double ssac = d.ssacld[i * kSwNBand + b];
        const double ffp1 = 1.0 - ffp, ffpssa = 1.0 - ffp * ssac;
        ssa = ffp1 * ssac / ffpssa;
        tau = ffpssa * tcb;
        asy = (d.asmcld[i * kSwNBand + b] - ffp) / ffp1;
      } else if (d.inflag == 2) {
        double extcoice = 0, ssacoice = 0, gice = 0, forwice = 0;
        double extcoliq = 0, ssacoliq = 0, gliq = 0, forwliq = 0;
        const double radice = d.reice[i];
        if (ciwp == 0.0) {
        } else if (d.iceflag == 1) {
          chk(radice < 13.0 || radice > 130., RRTMG_ERR_ICE_RADIUS);   // rrtmg_sw_cldprop.f90:194, cldprmc:183
          const double wn2 = t[T.wavenum2 + b];
          int icx = 5;
          if (wn2 > 1.43e04) icx = 1; else if (wn2 > 7.7e03) icx = 2; else if (wn2 > 5.3e03) icx = 3; else if (wn2 > 4.0e03) icx = 4;
          extcoice = t[T.abari + icx - 1] + t[T.bbari + icx - 1] / radice;
          ssacoice = 1.0 - t[T.cbari + icx - 1] - t[T.dbari + icx - 1] * radice;
          gice = t[T.ebari + icx - 1] + t[T.fbari + icx - 1] * radice;
          if (gice >= 1.0) gice = 1.0 - eps;
          forwice = gice * gice;
        } else if (d.iceflag == 2) {
          chk(radice < 5.0 || radice > 131.0, RRTMG_ERR_ICE_RADIUS);   // :226 / :213
          const double factor = e ? 1.0 : (radice - 2.0) / 3.0;
          int index = (int)factor;
          if (index == 43) index = 42;
          if (index < 1) index = 1;
          const double fint = factor - (double)index;
          const long k = (long)(index - 1) + 43 * b;
          extcoice = t[T.extice2 + k] + fint * (t[T.extice2 + k + 1] - t[T.extice2 + k]);
          ssacoice = t[T.ssaice2 + k] + fint * (t[T.ssaice2 + k + 1] - t[T.ssaice2 + k]);
          gice = t[T.asyice2 + k] + fint * (t[T.asyice2 + k + 1] - t[T.asyice2 + k]);
          forwice = gice * gice;
        } else if (d.iceflag == 3) {
          chk(radice < 5.0 || radice > 140.0, RRTMG_ERR_ICE_GEN_SIZE);   // :250 / :236
          const double factor = e ? 1.0 : (radice - 2.0) / 3.0;
          int index = (int)factor;
          if (index == 46) index = 45;
          if (index < 1) index = 1;
          const double fint = factor - (double)index;
          const long k = (long)(index - 1) + 46 * b;
          extcoice = t[T.extice3 + k] + fint * (t[T.extice3 + k + 1] - t[T.extice3 + k]);
          ssacoice = t[T.ssaice3 + k] + fint * (t[T.ssaice3 + k + 1] - t[T.ssaice3 + k]);
          gice = t[T.asyice3 + k] + fint * (t[T.asyice3 + k + 1] - t[T.asyice3 + k]);
          const double fdelta = t[T.fdlice3 + k] + fint * (t[T.fdlice3 + k + 1] - t[T.fdlice3 + k]);
          chk(fdelta < 0.0, RRTMG_ERR_FDELTA_NEG); chk(fdelta > 1.0, RRTMG_ERR_FDELTA_GT1);   // :264-265 / :250-251
          forwice = fdelta + 0.5 / ssacoice;
          if (forwice > gice) forwice = gice;
        } else {
          chk(true, RRTMG_ERR_UNSUPPORTED);
        }
        if (ciwp != 0.0) {   // :216-220, :240-244, :270-274 / cldprmc :204-208, :227-231, :256-260 (after each parameterisation)
          chk(extcoice < 0.0, RRTMG_ERR_ICE_EXT_NEG); chk(ssacoice > 1.0, RRTMG_ERR_ICE_SSA_GT1); chk(ssacoice < 0.0, RRTMG_ERR_ICE_SSA_NEG);
          chk(gice > 1.0, RRTMG_ERR_ICE_ASYM_GT1); chk(gice < 0.0, RRTMG_ERR_ICE_ASYM_NEG);
        }
        if (clwp == 0.0) {
        } else if (d.liqflag == 1) {
          const double radliq = d.reliq[i];
          chk(radliq < 2.5 || radliq > 60., RRTMG_ERR_LIQ_RADIUS);   // :290 / :273
          int index = (int)(radliq - 1.5);
          if (index == 0) index = 1;
          if (index == 58) index = 57;
          if (index < 1) index = 1;
          if (index > 57) index = 57;
          const double fint = radliq - 1.5 - (double)index;
          const long k = (long)(index - 1) + 58 * b;
          extcoliq = t[T.extliq1 + k] + fint * (t[T.extliq1 + k + 1] - t[T.extliq1 + k]);
          ssacoliq = t[T.ssaliq1 + k] + fint * (t[T.ssaliq1 + k + 1] - t[T.ssaliq1 + k]);
          if (fint < 0. && ssacoliq > 1.) ssacoliq = t[T.ssaliq1 + k];
          gliq = t[T.asyliq1 + k] + fint * (t[T.asyliq1 + k + 1] - t[T.asyliq1 + k]);
          forwliq = gliq * gliq;
          // :307-311 / :290-294
          chk(extcoliq < 0.0, RRTMG_ERR_LIQ_EXT_NEG); chk(ssacoliq > 1.0, RRTMG_ERR_LIQ_SSA_GT1); chk(ssacoliq < 0.0, RRTMG_ERR_LIQ_SSA_NEG);
          chk(gliq > 1.0, RRTMG_ERR_LIQ_ASYM_GT1); chk(gliq < 0.0, RRTMG_ERR_LIQ_ASYM_NEG);
        } else {
          chk(true, RRTMG_ERR_UNSUPPORTED);
        }
        const double tauliqorig = clwp * extcoliq, tauiceorig = ciwp * extcoice;
        const double ssaliq = ssacoliq * (1.0 - forwliq) / (1.0 - forwliq * ssacoliq);
        const double tauliq = (1.0 - forwliq * ssacoliq) * tauliqorig;
        const double ssaice = ssacoice * (1.0 - forwice) / (1.0 - forwice * ssacoice);
        const double tauice = (1.0 - forwice * ssacoice) * tauiceorig;
        const double scatliq = ssaliq * tauliq;
        double scatice = ssaice * tauice;
        tau = tauliq + tauice;
        if (tau == 0.0) tau = cldmin;
        if (scatice == 0.0) scatice = cldmin;
        ssa = (scatliq + scatice) / tau;
        if (d.iceflag == 3) {
          asy = (1.0 / (scatliq + scatice)) *
                (scatliq * (gliq - forwliq) / (1.0 - forwliq) + scatice * ((gice - forwice) / (1.0 - forwice)));
        } else {
          asy = (scatliq * (gliq - forwliq) / (1.0 - forwliq) + scatice * (gice - forwice) / (1.0 - forwice)) /
                (scatliq + scatice);
        }
      }
    }
    // (behind a failed check the band gets the optics of no cloud: the solve kernels run to the end of the call whatever the
    // flag says, and a negative optical depth would take their table lookups out of bounds)
    if (e) { tau = 0.0; ssa = 1.0; asy = 0.0; }
    d.ctau[o] = tau; d.cssa[o] = ssa; d.casm[o] = asy;
  }
  if (e) report_error(d.err, e);
}

// ------------------------------------------------------------------------------------------
// kissvec sub-column generator for one column (mcica_subcol_gen_sw.f90:316-470, :557-591).
// Draw order of the reference: sub-column outer, layer inner, after `changeSeed` warm-up draws.
// Output: bit (lay & 63) of mask[(ig*nw + lay/64)*ncol + col] set when the sub-column is cloudy.
// nsub = 112 (SW) or 140 (LW).
// ------------------------------------------------------------------------------------------
struct Kiss { int32_t s1, s2, s3, s4; };
RRTMG_HD double kiss_next(Kiss &k) {
#pragma clang fp contract(off)   // kiss*2.328306e-10 + 0.5 is compared with 1-cldf: keep the reference's rounding
  uint32_t a = (uint32_t)k.s1, b = (uint32_t)k.s2, c = (uint32_t)k.s3, e = (uint32_t)k.s4;
  a = 69069u * a + 1327217885u;
  b ^= b << 13; b ^= b >> 17; b ^= b << 5;
  // ishft(seed,-16) is a LOGICAL shift; iand(seed,65535)
  c = 18000u * (c & 65535u) + (c >> 16);
  e = 30903u * (e & 65535u) + (e >> 16);
  k.s1 = (int32_t)a; k.s2 = (int32_t)b; k.s3 = (int32_t)c; k.s4 = (int32_t)e;
  const int32_t kiss = (int32_t)(a + b + (c << 16) + e);
  return (double)kiss * 2.328306e-10 + 0.5;
}

// seeds of one column from the fractional digits of the four lowest mid-layer pressures (mcica_subcol_gen_sw.f90:340-356)
RRTMG_HD bool kiss_seed_column(int ncol, const double *play, int *err, int col, Kiss &k) {
  // integer seeds come from the fractional part of pmid = play*100: the product must be ROUNDED before the
  // subtraction (as the reference, which stores pmid), so no fused multiply-add here
#pragma clang fp contract(off)
  const long N = ncol;
  const double p1 = play[col] * 1.e2, p2 = play[N + col] * 1.e2;
  const double p3 = play[2l * N + col] * 1.e2, p4 = play[3l * N + col] * 1.e2;
  if (p1 < p2) { report_error(err, RRTMG_ERR_KISS_PRESSURE); return false; }
  k.s1 = (int32_t)((p1 - (double)(int)p1) * 1000000000.0);
  k.s2 = (int32_t)((p2 - (double)(int)p2) * 1000000000.0);
  k.s3 = (int32_t)((p3 - (double)(int)p3) * 1000000000.0);
  k.s4 = (int32_t)((p4 - (double)(int)p4) * 1000000000.0);
  return true;
}

// one sub-column of one column from the generator state k positioned at the sub-column's first draw
RRTMG_HD void kiss_mask_subcolumn(int ncol, int nlay, int icld, const double *cldfr, uint64_t *mask, int nw, int col, int g, Kiss &k) {
  const int N = ncol, L = nlay;
  const double cldmin = 1.0e-20;
  double cdf_prev = 0.0, cmax = 0.0;
  if (icld == 3) cmax = kiss_next(k);
  uint64_t word = 0;
  for (int l = 0; l < L; ++l) {
    double cf = cldfr[(long)l * N + col];
    if (cf < cldmin) cf = 0.0;
    double cdf;
    if (icld == 3) {
      cdf = cmax;
    } else {
      cdf = kiss_next(k);
      if (icld == 2 && l > 0) {
        double cfm = cldfr[(long)(l - 1) * N + col];
        if (cfm < cldmin) cfm = 0.0;
        if (cdf_prev > 1.0 - cfm) cdf = cdf_prev; else cdf = cdf * (1.0 - cfm);
      }
    }
    cdf_prev = cdf;
    if (cdf >= 1.0 - cf) word |= (1ull << (l & 63));
    if ((l & 63) == 63 || l == L - 1) { mask[((long)g * nw + (l >> 6)) * N + col] = word; word = 0; }
  }
}

// Reference order, one thread per column: sub-columns drawn one after the other from ONE stream (this is the
// definition the jump-ahead kernel below is tested against).
RRTMG_HD void kiss_mask_column(int ncol, int nlay, int nsub, int icld, int changeSeed, const double *play,
                               const double *cldfr, uint64_t *mask, int nw, int *err, int col) {
  const int N = ncol, L = nlay;
  for (int g = 0; g < nsub; ++g)
    for (int w = 0; w < nw; ++w) mask[((long)g * nw + w) * N + col] = 0ull;
  if (icld == 0) return;
  if (L < 4) { report_error(err, RRTMG_ERR_ARG); return; }
  Kiss k;
  if (!kiss_seed_column(ncol, play, err, col, k)) return;
  for (int i = 0; i < changeSeed; ++i) (void)kiss_next(k);
  for (int g = 0; g < nsub; ++g) kiss_mask_subcolumn(ncol, nlay, icld, cldfr, mask, nw, col, g, k);
}

// ---- jump-ahead: sub-column g starts n_g = changeSeed + g * (draws per sub-column) draws into the column's
// stream.  The four component generators of KISS can each be advanced n steps in closed form:
//   congruential  x -> 69069 x + 1327217885 (mod 2^32)      : affine map (A_n, C_n)
//   xorshift      13 / 17 / 5                                 : 32x32 matrix over GF(2), M^n (32 column words)
//   multiply-with-carry  s -> a (s & 65535) + (s >> 16)       : after two real steps s lies in [0, m], m = a 2^16 - 1,
//                                                               and the step is s -> a s mod m (fixed points 0 and m)
// so one thread per (column, sub-column) reproduces the reference's sequential stream exactly.  The per-sub-column
// jump operators are the same for every column; the host builds them (rrtmg_kiss_host.h) -- kKissJumpWords words
// per sub-column: [0] n, [1] A_n, [2] C_n, [3] 18000^(n-2) mod m3, [4] 30903^(n-2) mod m4, [8..39] columns of M^n.
constexpr int kKissJumpWords = 40;
constexpr uint32_t kKissM3 = 18000u * 65536u - 1u, kKissM4 = 30903u * 65536u - 1u;
RRTMG_HD uint32_t kiss_mwc_jump(uint32_t s, uint32_t a, uint32_t m, uint32_t n, uint32_t apow) {
  const uint32_t real = n < 2u ? n : 2u;
  for (uint32_t i = 0; i < real; ++i) s = a * (s & 65535u) + (s >> 16);
  if (n > 2u && s != 0u && s != m) s = (uint32_t)(((uint64_t)s * (uint64_t)apow) % (uint64_t)m);
  return s;
}
RRTMG_HD void kiss_jump(Kiss &k, const uint32_t *J) {
  const uint32_t n = J[0];
  uint32_t a = (uint32_t)k.s1, b = (uint32_t)k.s2;
  a = J[1] * a + J[2];
  uint32_t r = 0;
  for (int i = 0; i < 32; ++i) r ^= ((b >> i) & 1u) ? J[8 + i] : 0u;
  k.s1 = (int32_t)a; k.s2 = (int32_t)r;
  k.s3 = (int32_t)kiss_mwc_jump((uint32_t)k.s3, 18000u, kKissM3, n, J[3]);
  k.s4 = (int32_t)kiss_mwc_jump((uint32_t)k.s4, 30903u, kKissM4, n, J[4]);
}
// one thread per (column, sub-column)
RRTMG_HD void kiss_mask_jump(int ncol, int nlay, int icld, const double *play, const double *cldfr, uint64_t *mask, int nw,
                             int *err, const uint32_t *jumps, int col, int g) {
  for (int w = 0; w < nw; ++w) mask[((long)g * nw + w) * ncol + col] = 0ull;
  if (icld == 0) return;
  if (nlay < 4) { report_error(err, RRTMG_ERR_ARG); return; }
  Kiss k;
  if (!kiss_seed_column(ncol, play, err, col, k)) return;
  kiss_jump(k, jumps + (long)g * kKissJumpWords);
  kiss_mask_subcolumn(ncol, nlay, icld, cldfr, mask, nw, col, g, k);
}

// ------------------------------------------------------------------------------------------
// transmittance table / two-stream layer operators
// ------------------------------------------------------------------------------------------
// (the quotient is the quick division: a last-place difference moves the rounded index with probability ~1e-12
//  per lookup, to a neighbouring entry 1e-4 away -- far inside the 0.01 W m-2 bar)
RRTMG_HD double sw_exp_lookup(const double *exp_tbl, double x) {
  const double tblind = qdiv(x, kBpade + x);
  const int itind = (int)(kTblInt * tblind + 0.5);
  return exp_tbl[itind];
}

// min(x, 500), the clamp of reftra's exponents (rrtmg_sw_reftra.f90:198,262-263): ONE v_min_f64 on the device instead of a compare and
// two selects (x is the result of a multiplication, never a signalling NaN; a NaN -- the reference keeps it -- becomes 500)
RRTMG_HD double cap500(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_fmin(x, 500.0);
#else
  return x > 500.0 ? 500.0 : x;
#endif
}

// direct-beam transmittance of a layer (rrtmg_sw_spcvrt.f90:562-574)
// rmu0 = 1/prmu0, formed once per column: tau/prmu0 is evaluated as tau*rmu0 (<= 1 ulp apart)
RRTMG_HD double sw_dbt(const double *exp_tbl, double tau, double rmu0) {
  const double ze1 = tau * rmu0;
  if (ze1 <= 0.06) return 1.0 - ze1 + 0.5 * ze1 * ze1;
  return sw_exp_lookup(exp_tbl, ze1);
}

// reftra_sw for one layer, kmodts = 2 (rrtmg_sw_reftra.f90:148-316)
// ZG0: the asymmetry parameter is exactly zero (clear sky without aerosol) -- the same arithmetic with the
// terms that are then exactly 0 / 1 folded at compile time (identical results).  rmuz = 1/prmuz.
// pdbt: the direct-beam transmittance exp(-zto1/prmuz) of the layer (rrtmg_sw_spcvrt.f90:562-574, sw_dbt) -- it is the
// very table entry reftra looks up for its own exp(-zto1/prmuz) whenever zto1/prmuz <= 500 (reftra clamps there, the
// direct beam does not), so it is handed out instead of being looked up a second time.
template <bool ZG0 = false>
RRTMG_HD void sw_reftra(const double *exp_tbl, double zg, double prmuz, double rmuz, double zto1, double zw, double &pref,
                        double &prefd, double &ptra, double &ptrad, double &pdbt) {
  const double eps = 1.e-08, zwcrit = 0.9999995, od_lo = 0.06;
  double zgamma1, zgamma2, zgamma3, zwo;
  if constexpr (ZG0) {
#if defined(__HIP_DEVICE_COMPILE__)
    // (8 - 5 w) / 4 and 3 w / 4 with the exact scaling by 1/4 folded into the constants: the SAME bits (a power of two commutes
    // with the rounding) in one instruction each instead of four and two (an accumulator to initialise, a v_ldexp_f64)
    zgamma1 = __builtin_fma(-1.25, zw, 2.0);
    zgamma2 = 0.75 * zw;
#else
    zgamma1 = (8.0 - zw * 5.0) * 0.25;
    zgamma2 = 3.0 * zw * 0.25;
#endif
    zgamma3 = 0.5;
    zwo = zw;
  } else {
    const double zg3 = 3.0 * zg;
    zgamma1 = (8.0 - zw * (5.0 + zg3)) * 0.25;
    zgamma2 = 3.0 * (zw * (1.0 - zg)) * 0.25;
    zgamma3 = (2.0 - zg3 * prmuz) * 0.25;
    const double zq = qdiv(zg, 1.0 - zg);
    zwo = qdiv(zw, 1.0 - (1.0 - zw) * (zq * zq));
  }
  const double zgamma4 = 1.0 - zgamma3;
  if (zwo >= zwcrit) {
    const double za = zgamma1 * prmuz;
    const double za1 = za - zgamma3;
    const double zgt = zgamma1 * zto1;
    const double zeu = zto1 * rmuz;
    double ze1 = zeu;
    ze1 = cap500(ze1);
    double ze2;
    if (ze1 <= od_lo) ze2 = 1.0 - ze1 + 0.5 * ze1 * ze1; else ze2 = sw_exp_lookup(exp_tbl, ze1);
    pdbt = zeu > 500.0 ? sw_exp_lookup(exp_tbl, zeu) : ze2;
    pref = qdiv(zgt - za1 * (1.0 - ze2), 1.0 + zgt);   // same denominator below: one rcp on the device
    ptra = 1.0 - pref;
    prefd = qdiv(zgt, 1.0 + zgt);
    ptrad = 1.0 - prefd;
    if (ze2 == 1.0) { pref = 0.0; ptra = 1.0; prefd = 0.0; ptrad = 1.0; }
  } else {
    const double za1 = zgamma1 * zgamma4 + zgamma2 * zgamma3;
    const double za2 = zgamma1 * zgamma3 + zgamma2 * zgamma4;
    const double zrk = qsqrt(zgamma1 * zgamma1 - zgamma2 * zgamma2);   // > 0 here: (g1 - g2) = 2 (1 - w) >= 1e-6
    const double zrp = zrk * prmuz;
    const double zrp1 = 1.0 + zrp, zrm1 = 1.0 - zrp;
    const double zrk2 = 2.0 * zrk;
    const double zrpp = 1.0 - zrp * zrp;
    const double zrkg = zrk + zgamma1;
    const double zr1 = zrm1 * (za2 + zrk * zgamma3);
    const double zr2 = zrp1 * (za2 - zrk * zgamma3);
    const double zr3 = zrk2 * (zgamma3 - za2 * prmuz);
    const double zr4 = zrpp * zrkg;
    const double zr5 = zrpp * (zrk - zgamma1);
    const double zt1 = zrp1 * (za1 + zrk * zgamma4);
    const double zt2 = zrm1 * (za1 - zrk * zgamma4);
    const double zt3 = zrk2 * (zgamma4 + za1 * prmuz);
    const double ze1 = cap500(zrk * zto1);
    const double zeu = zto1 * rmuz;
    const double ze2 = cap500(zeu);
    double zem1, zem2;
    if (ze1 <= od_lo) zem1 = 1.0 - ze1 + 0.5 * ze1 * ze1; else zem1 = sw_exp_lookup(exp_tbl, ze1);
    if (ze2 <= od_lo) zem2 = 1.0 - ze2 + 0.5 * ze2 * ze2; else zem2 = sw_exp_lookup(exp_tbl, ze2);
    pdbt = zeu > 500.0 ? sw_exp_lookup(exp_tbl, zeu) : zem2;
    const double zemm = zem1 * zem1;
#if defined(__HIP_DEVICE_COMPILE__)
    // The reference's quotients with numerator and denominator multiplied through by zem1 = exp(-k tau) (and
    // zem2 * zep2 = 1 used): ONE reciprocal instead of its three (1/zem1, 1/zem2, 1/zdenr), and the diffuse pair
    // without forming zbeta -- two reciprocals per layer operator instead of five.  Same quantities, last-place
    // differences (the host build below keeps the reference's operation order; GPU vs reference <= 1e-9 W m-2).
    const double zdp = zr4 + zr5 * zemm;   // = zdenr * zem1
    if (fabs(zdp) <= eps * zem1) {
      pref = eps;
      ptra = zem2;
    } else {
      const double rd = qrcp(zdp);
      pref = (zw * (zr1 - zr2 * zemm - (zr3 * zem2) * zem1)) * rd;
      ptra = zem2 - (zw * (zem2 * (zt1 - zt2 * zemm) - zt3 * zem1)) * rd;
    }
    const double zdend = qrcp(zrkg - (zgamma1 - zrk) * zemm);
#else
    const double zbeta = qdiv(zgamma1 - zrk, zrkg);
    const double zep1 = qrcp(zem1);
    const double zep2 = qrcp(zem2);
    const double zdenr = zr4 * zep1 + zr5 * zem1;
    if (zdenr >= -eps && zdenr <= eps) {
      pref = eps;
      ptra = zem2;
    } else {
      pref = qdiv(zw * (zr1 * zep1 - zr2 * zem1 - zr3 * zem2), zdenr);
      ptra = zem2 - qdiv(zem2 * zw * (zt1 * zep1 - zt2 * zem1 - zt3 * zep2), zdenr);   // zdent == zdenr
    }
    const double zdend = qrcp((1.0 - zbeta * zemm) * zrkg);
#endif
    prefd = zgamma2 * (1.0 - zemm) * zdend;
    ptrad = zrk2 * zem1 * zdend;
  }
}

// ------------------------------------------------------------------------------------------
// taumol_sw for one (layer, g-point) of band BAND (rrtmg_sw_taumol.f90 taumol16..29).
// ------------------------------------------------------------------------------------------
struct SwLayerIn {
  double fac00, fac01, fac10, fac11, selffac, selffrac, forfac, forfrac;
  double colh2o, colco2, colo3, colch4, colo2, colmol;
  int jp, jt, jt1, indself, indfor;
};

RRTMG_HD void sw_load_layer(const SwDev &d, int col, int lay, SwLayerIn &s) {
  const double *q = d.prep + sw_prep_off(d.nlay, col, lay);
  s.fac00 = q[P_FAC00 * 64]; s.fac01 = q[P_FAC01 * 64]; s.fac10 = q[P_FAC10 * 64]; s.fac11 = q[P_FAC11 * 64];
  s.selffac = q[P_SELFFAC * 64]; s.selffrac = q[P_SELFFRAC * 64]; s.forfac = q[P_FORFAC * 64]; s.forfrac = q[P_FORFRAC * 64];
  s.colh2o = q[P_COLH2O * 64]; s.colco2 = q[P_COLCO2 * 64]; s.colo3 = q[P_COLO3 * 64]; s.colch4 = q[P_COLCH4 * 64];
  s.colo2 = q[P_COLO2 * 64]; s.colmol = q[P_COLMOL * 64];
  const int p = (int)q[P_IDX * 64];
  s.jp = p & 0xff; s.jt = (p >> 8) & 0xf; s.jt1 = (p >> 12) & 0xf; s.indself = (p >> 16) & 0xff; s.indfor = (p >> 24) & 0xf;
}

struct SwSpec { int js; double fs; double speccomb; };

RRTMG_HD SwSpec sw_specparm(double colx, double coly, double strrat, double mult) {
  SwSpec r;
  r.speccomb = colx + strrat * coly;
  const double oneminus = 1.0 - 1.e-6;
#if defined(__HIP_DEVICE_COMPILE__)
  // the quick quotient and ONE v_min_f64: a last-place difference in specparm can move js across an integer only together with
  // fs across 0 / 1, and the interpolation between rows js, js + 1 is continuous there (unlike a nearest-entry table lookup)
  const double specparm = __builtin_fmin(qdiv(colx, r.speccomb), oneminus);
#else
  double specparm = colx / r.speccomb;
  if (specparm >= oneminus) specparm = oneminus;
#endif
  const double specmult = mult * specparm;
  r.js = 1 + (int)specmult;
  r.fs = specmult - (double)(int)specmult;   // mod(specmult, 1)
  return r;
}

// 8-point (binary species) major-gas sum, dT = offset of the next-temperature rows
template <int G, int NG>
RRTMG_HD V<G> sw_m8(const KTab<G, NG> &k, int i0, int i1, int dT, const SwLayerIn &s, double fs) {
  const double fac000 = (1.0 - fs) * s.fac00, fac010 = (1.0 - fs) * s.fac10;
  const double fac100 = fs * s.fac00, fac110 = fs * s.fac10;
  const double fac001 = (1.0 - fs) * s.fac01, fac011 = (1.0 - fs) * s.fac11;
  const double fac101 = fs * s.fac01, fac111 = fs * s.fac11;
  return fac000 * k[i0] + fac100 * k[i0 + 1] + fac010 * k[i0 + dT] + fac110 * k[i0 + dT + 1] +
         fac001 * k[i1] + fac101 * k[i1 + 1] + fac011 * k[i1 + dT] + fac111 * k[i1 + dT + 1];
}
template <int G, int NG>
RRTMG_HD V<G> sw_m4(const KTab<G, NG> &k, int i0, int i1, const SwLayerIn &s) {
  return s.fac00 * k[i0] + s.fac10 * k[i0 + 1] + s.fac01 * k[i1] + s.fac11 * k[i1 + 1];
}
template <int G, int NG>
RRTMG_HD V<G> sw_selfterm(const KTab<G, NG> &selfref, const SwLayerIn &s) {  // selffac*(selfref + selffrac*(d))
  const V<G> a = selfref[s.indself - 1], b = selfref[s.indself];
  return s.selffac * (a + s.selffrac * (b - a));
}
template <int G, int NG>
RRTMG_HD V<G> sw_forinterp(const KTab<G, NG> &forref, const SwLayerIn &s) {  // forref + forfrac*(d)
  const V<G> a = forref[s.indfor - 1], b = forref[s.indfor];
  return a + s.forfrac * (b - a);
}

template <int BAND> struct SwBandCfg;
#define SW_CFG(B, NG, NSPA, NSPB, LOX, LOY, STR, UPPERSRC, BINSRC)                 \
  template <> struct SwBandCfg<B> {                                                \
    static constexpr int ng = NG, nspa = NSPA, nspb = NSPB, lox = LOX, loy = LOY;  \
    static constexpr double strrat = STR;                                          \
    static constexpr bool upper_src = UPPERSRC, bin_src = BINSRC;                  \
  };
//      band ng(reduced, rrtmg_sw parrrsw.f90 ng16..ng29) nspa nspb key-lo-x  key-lo-y  strrat  src-in-upper  binary-src
SW_CFG(16, 6, 9, 1, SP_H2O, SP_CH4, 252.131, true, false)
SW_CFG(17, 12, 9, 5, SP_H2O, SP_CO2, 0.364641, true, true)
SW_CFG(18, 8, 9, 1, SP_H2O, SP_CH4, 38.9589, false, true)
SW_CFG(19, 8, 9, 1, SP_H2O, SP_CO2, 5.49281, false, true)
SW_CFG(20, 10, 1, 1, SP_H2O, -1, 0.0, false, false)
SW_CFG(21, 10, 9, 5, SP_H2O, SP_CO2, 0.0045321, false, true)
SW_CFG(22, 2, 9, 1, SP_H2O, SP_O2, 1.6 * 0.022708, false, true)
SW_CFG(23, 10, 1, 0, SP_H2O, -1, 0.0, false, false)
SW_CFG(24, 8, 9, 1, SP_H2O, SP_O2, 0.124692, false, true)
SW_CFG(25, 6, 1, 0, SP_H2O, -1, 0.0, false, false)
SW_CFG(26, 6, 0, 0, -1, -1, 0.0, false, false)
SW_CFG(27, 8, 1, 1, SP_O3, -1, 0.0, true, false)
SW_CFG(28, 6, 9, 5, SP_O3, SP_O2, 6.67029e-07, true, true)
SW_CFG(29, 12, 1, 1, SP_H2O, -1, 0.0, true, false)
#undef SW_CFG

RRTMG_HD double sw_col(const SwLayerIn &s, int sp) {
  return sp == SP_H2O ? s.colh2o : sp == SP_CO2 ? s.colco2 : sp == SP_O3 ? s.colo3 : sp == SP_CH4 ? s.colch4 : s.colo2;
}

// Gas optical depths of the G g-points ig0 .. ig0+G-1; sets the Rayleigh optical depths.
// `lower` = layer index <= laytrop.
// LDSK = true: kb -> the item's slice of the band slab, [nrows][G] (the workgroup's LDS copy); LDSK = false: kb is
// ignored and the slab is read in place, [nrows][ng], through the vector L1.
template <int BAND, int G, bool LDSK = false>
RRTMG_HD V<G> sw_taug(const SwTab &T, const SwLayerIn &s, bool lower, int ig0, V<G> &taur, const double *kb = nullptr) {
  using C = SwBandCfg<BAND>;
  constexpr int NG = C::ng;
  constexpr int ST = LDSK ? G : NG;
  const SwBandTab &B = T.b[BAND - 16];
  if (!LDSK) kb = T.t + B.slab + ig0;
  auto view = [&](int r) { return KTab<G, ST>{kb + (long)r * ST}; };
  const KTab<G, ST> absa = view(B.r_absa), absb = view(B.r_absb), selfref = view(B.r_self), forref = view(B.r_forr);
  auto row = [&](int r) { return vload<G>(kb + (long)r * ST); };   // a one-row ([ng]) table
  V<G> taug = vsplat<G>(0.0);
  // Rayleigh: scalar per band (replicated per g at init), per g, or band 24's mixture-dependent form
  V<G> rayl = row(B.r_rayl);
  if (lower) {
    if constexpr (C::nspa == 9) {
      const SwSpec sp = sw_specparm(sw_col(s, C::lox), sw_col(s, C::loy), C::strrat, 8.0);
      const int i0 = ((s.jp - 1) * 5 + (s.jt - 1)) * 9 + sp.js - 1;
      const int i1 = (s.jp * 5 + (s.jt1 - 1)) * 9 + sp.js - 1;
      const V<G> major = sp.speccomb * sw_m8(absa, i0, i1, 9, s, sp.fs);
      if constexpr (BAND == 28) {
        taug = major;
      } else if constexpr (BAND == 24) {
        taug = major + s.colo3 * row(B.r_ex1) +
               s.colh2o * (sw_selfterm(selfref, s) + s.forfac * sw_forinterp(forref, s));
        const KTab<G, ST> ra = view(B.r_rayl);   // rayla(ig, js): [js][ig]
        const V<G> r0 = ra[sp.js - 1], r1 = ra[sp.js];
        rayl = r0 + sp.fs * (r1 - r0);
      } else {
        taug = major + s.colh2o * (sw_selfterm(selfref, s) + s.forfac * sw_forinterp(forref, s));
        if constexpr (BAND == 22) taug = taug + 4.35e-4 * s.colo2 / (350.0 * 2.0);
      }
    } else if constexpr (C::nspa == 1) {
      const int i0 = ((s.jp - 1) * 5 + (s.jt - 1));
      const int i1 = (s.jp * 5 + (s.jt1 - 1));
      const V<G> m4 = sw_m4(absa, i0, i1, s);
      if constexpr (BAND == 20) {
        taug = s.colh2o * (m4 + sw_selfterm(selfref, s) + s.forfac * sw_forinterp(forref, s)) + s.colch4 * row(B.r_ex1);
      } else if constexpr (BAND == 29) {
        taug = s.colh2o * (m4 + sw_selfterm(selfref, s) + s.forfac * sw_forinterp(forref, s)) + s.colco2 * row(B.r_ex1);
      } else if constexpr (BAND == 23) {
        taug = s.colh2o * (1.029 * m4 + sw_selfterm(selfref, s) + s.forfac * sw_forinterp(forref, s));
      } else if constexpr (BAND == 25) {
        taug = s.colh2o * m4 + s.colo3 * row(B.r_ex1);
      } else {  // 27
        taug = s.colo3 * m4;
      }
    } else {
      taug = vsplat<G>(0.0);  // band 26
    }
  } else {
    if constexpr (C::nspb == 5) {
      const SwSpec sp = sw_specparm(sw_col(s, C::lox), sw_col(s, C::loy), C::strrat, 4.0);
      const int i0 = ((s.jp - 13) * 5 + (s.jt - 1)) * 5 + sp.js - 1;
      const int i1 = ((s.jp - 12) * 5 + (s.jt1 - 1)) * 5 + sp.js - 1;
      const V<G> major = sp.speccomb * sw_m8(absb, i0, i1, 5, s, sp.fs);
      if constexpr (BAND == 28) taug = major;
      else taug = major + s.colh2o * s.forfac * sw_forinterp(forref, s);
    } else if constexpr (C::nspb == 1) {
      const int i0 = ((s.jp - 13) * 5 + (s.jt - 1));
      const int i1 = ((s.jp - 12) * 5 + (s.jt1 - 1));
      if constexpr (BAND == 16 || BAND == 18) taug = s.colch4 * sw_m4(absb, i0, i1, s);
      else if constexpr (BAND == 19) taug = s.colco2 * sw_m4(absb, i0, i1, s);
      else if constexpr (BAND == 20)
        taug = s.colh2o * (s.fac00 * absb[i0] + s.fac10 * absb[i0 + 1] + s.fac01 * absb[i1] + s.fac11 * absb[i1 + 1] +
                           s.forfac * sw_forinterp(forref, s)) + s.colch4 * row(B.r_ex1);
      else if constexpr (BAND == 22) taug = s.colo2 * 1.6 * sw_m4(absb, i0, i1, s) + 4.35e-4 * s.colo2 / (350.0 * 2.0);
      else if constexpr (BAND == 24) { taug = s.colo2 * sw_m4(absb, i0, i1, s) + s.colo3 * row(B.r_ex2); rayl = row(B.r_raylb); }
      else if constexpr (BAND == 27) taug = s.colo3 * sw_m4(absb, i0, i1, s);
      else taug = s.colco2 * sw_m4(absb, i0, i1, s) + s.colh2o * row(B.r_ex2);  // 29
    } else {
      if constexpr (BAND == 25) taug = s.colo3 * row(B.r_ex2);
      else taug = vsplat<G>(0.0);  // 23, 26
    }
  }
  taur = s.colmol * rayl;
  return taug;
}

// incoming solar flux of one g-point: zincflx = adjflux * (ssi | sfluxzen) * prmu0
// (rrtmg_sw_spcvrt.f90:335-343; source selection rrtmg_sw_taumol.f90 "lay .eq. laysolfr" blocks)
template <int BAND>
RRTMG_HD double sw_incflux(const SwDev &d, const SwTab &T, int col, int ig, double prmu0) {
  using C = SwBandCfg<BAND>;
  const SwBandTab &B = T.b[BAND - 16];
  const double *t = T.t;
  const int b = BAND - 16;
  const int ls = d.laysolfr[(long)b * d.ncol + col];
  if (ls <= 0) return 0.0;
  int js = 1;
  double fs = 0.0;
  if constexpr (C::bin_src) {
    SwLayerIn s;
    sw_load_layer(d, col, ls - 1, s);
    const SwSpec sp = sw_specparm(sw_col(s, C::lox), sw_col(s, C::loy), C::strrat, C::upper_src ? 4.0 : 8.0);
    js = sp.js; fs = sp.fs;
  }
  auto src = [&](long base) {
    if constexpr (C::bin_src) {
      const double a = t[base + ig + B.ng * (js - 1)], c = t[base + ig + B.ng * js];
      return a + fs * (c - a);
    } else {
      return t[base + ig];
    }
  };
  double s;
  if (d.isolvar < 0) {
    s = src(B.sflux);
    if constexpr (BAND == 27) s = (50.15 / 48.37) * t[B.sflux + ig];
    return d.adjflux_b[b] * s * prmu0;
  }
  if (d.isolvar == 3)
    s = d.svar_b[b] * src(B.fac) + d.svar_b[b] * src(B.sns) + d.svar_b[b] * src(B.irr);
  else if (d.svar_col)
    s = d.svar_col[col] * src(B.fac) + d.svar_col[(long)d.ncol + col] * src(B.sns) + d.svar_col[2l * d.ncol + col] * src(B.irr);
  else
    s = d.svar_f * src(B.fac) + d.svar_s * src(B.sns) + d.svar_i * src(B.irr);
  return d.adjflux * s * prmu0;
}

enum { F_RUP = 0, F_RUPD, F_NCLR, F_NTOT = 2 * F_NCLR };

// optical properties of one layer for one g-point: clear sky and (if requested) total sky
struct SwLayerOpt { double ref, refd, tra, trad, dbt; };

// Flux sink of the host emulation, of tests and of the device kernel: the weighted (fu, fd, cu, cd), summed over
// the item's g-points (pair sums first), go to part[slot][k][level][column].
struct SwPartSink {
  double *pfu, *pfd, *pcu, *pcd;   // slot of the item's first pair
  long N, slot_stride;             // slot_stride = 4 * (nlay + 1) * ncol
  RRTMG_HD void emit(int pair, int lev, double fu, double fd, double cu, double cd) {
    const long o = pair * slot_stride + (long)lev * N;
    part_store(pfu + o, fu); part_store(pfd + o, fd); part_store(pcu + o, cu); part_store(pcd + o, cd);
  }
  // cloud-free column (CLD = false variant): clear-sky == total, only the total planes are written and
  // sw_flux_level(pairs = false) reads them for both outputs
  RRTMG_HD void emit_clear(int lev, double fu, double fd) {
    const long o = (long)lev * N;
    part_store(pfu + o, fu); part_store(pfd + o, fd);
  }
};
RRTMG_HD SwPartSink sw_part_sink(const SwDev &d, int slot, int col) {
  const long N = d.pcols, L1 = d.nlay + 1;
  col -= d.col0;
  SwPartSink s;
  s.N = N; s.slot_stride = 4 * L1 * N;
  s.pfu = d.part + (((long)slot * 4 + 0) * L1) * N + col; s.pfd = d.part + (((long)slot * 4 + 1) * L1) * N + col;
  s.pcu = d.part + (((long)slot * 4 + 2) * L1) * N + col; s.pcd = d.part + (((long)slot * 4 + 3) * L1) * N + col;
  return s;
}

// per-thread constants of a (column, work item)
template <int G> struct SwThreadCtx {
  int b, iw0, ig0, laytrop;
  double prmu0, rmu0;      // cosine of the solar zenith angle and its reciprocal
  const double *exp_tbl;   // transmittance table: the workgroup's LDS copy on the device, T.t + T.exp_tbl on the host
  const double *kb;        // the item's k-distribution slice in LDS (LDSK) or nullptr
  bool cloudy[G];   // any cloud in (sub-)column g
  bool any_cloudy;
  uint64_t mw[G];   // McICA cloud-mask words of the 64-layer block the sweep is in (one read per 64 layers)
  int mword;
};

// taumol + delta scaling + reftra (+ cloud) for layer l and the G g-points of the item: everything the two
// adding-method sweeps need.  The layer state, the species mixture, the interpolation weights and the table
// rows are evaluated ONCE for the G g-points.  Called in BOTH sweeps: recomputing it is cheaper than spilling
// five more level arrays per g-point through HBM (profiles/r01_pmc_*.txt).
// consume(g, clear, total) is called for each g-point right after its optics are ready, so that only ONE g-point's
// ten layer operators are live at a time (register pressure).
template <int BAND, int G, bool CLD, bool LDSK, class Consume>
RRTMG_HD void sw_layer_optics(const SwDev &d, const SwTab &T, SwThreadCtx<G> &c, int col, int l, Consume &&consume) {
  const int L = d.nlay, N = d.ncol;
  const double *exp_tbl = c.exp_tbl;
  const double prmu0 = c.prmu0, rmu0 = c.rmu0;
  const long i = (long)l * N + col;
  SwLayerIn s;
  sw_load_layer(d, col, l, s);
  V<G> taur;
  const V<G> taug = sw_taug<BAND, G, LDSK>(T, s, (l + 1) <= c.laytrop, c.ig0, taur, c.kb);
  double taua = 0.0, omga = 1.0, asya = 0.0;
  const long o = ((long)c.b * L + l) * N + col;
  if (d.tauaer) { taua = d.tauaer[o]; omga = d.ssaaer[o]; asya = d.asmaer[o]; }
  // band cloud optics of this layer (shared by the g-points)
  double zcloud = 0.0, ptauc = 0.0, pomgc = 0.0, pasyc = 0.0;
  bool lcld_band = false;
  if (CLD && c.any_cloudy) {
    if (!d.mcica) { zcloud = d.cldfr[i]; lcld_band = zcloud > 1.e-12; }
    else if ((l >> 6) != c.mword) {
      c.mword = l >> 6;
#pragma unroll
      for (int g = 0; g < G; ++g) c.mw[g] = d.mask[((long)(c.iw0 + g) * d.nw + c.mword) * N + col];
    }
    ptauc = d.ctau[o];
    pomgc = d.cssa[o]; pasyc = d.casm[o];
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    SwLayerOpt oc, ot;
    // clear-sky optical properties and delta scaling (rrtmg_sw_spcvrt.f90:447-498)
    double ztauc, zomcc, zgcc;
    if (d.tauaer) {
      ztauc = taur[g] + taug[g] + taua;
      zomcc = taur[g] * 1.0 + taua * omga;
      zgcc = qdiv(asya * omga * taua, zomcc);
      zomcc = qdiv(zomcc, ztauc);
      const double zf = zgcc * zgcc, zwf = zomcc * zf;
      ztauc = (1.0 - zwf) * ztauc;
      zomcc = qdiv(zomcc - zwf, 1.0 - zwf);
      zgcc = qdiv(zgcc - zf, 1.0 - zf);
      sw_reftra<false>(exp_tbl, zgcc, prmu0, rmu0, ztauc, zomcc, oc.ref, oc.refd, oc.tra, oc.trad, oc.dbt);
    } else {
      // no aerosol: taua = 0, omga = 1, asya = 0 -> zgcc = 0 and the delta scaling is the identity
      ztauc = taur[g] + taug[g];
      zomcc = qdiv(taur[g], ztauc);
      zgcc = 0.0;
      sw_reftra<true>(exp_tbl, 0.0, prmu0, rmu0, ztauc, zomcc, oc.ref, oc.refd, oc.tra, oc.trad, oc.dbt);
    }
    if (!CLD || !c.cloudy[g]) { consume(g, oc, oc); continue; }
    ot = oc;
    bool lcld;
    double zc;
    if (d.mcica) { lcld = (c.mw[g] >> (l & 63)) & 1ull; zc = lcld ? 1.0 : 0.0; }
    else { zc = zcloud; lcld = lcld_band; }
    const double ptc = (lcld || !d.mcica) ? ptauc : 0.0;
    if (lcld) {
      // icpr = 1 branch (rrtmg_sw_spcvrt.f90:503-509)
      const double ztauo = ztauc + ptc;
      double zomco = ztauc * zomcc + ptc * pomgc;
      const double zgco = qdiv(ptc * pomgc * pasyc + ztauc * zomcc * zgcc, zomco);
      zomco = qdiv(zomco, ztauo);
      double refo, refdo, trao, trado, dbto;
      sw_reftra<false>(exp_tbl, zgco, prmu0, rmu0, ztauo, zomco, refo, refdo, trao, trado, dbto);
      if (d.mcica) {
        ot.ref = refo; ot.refd = refdo; ot.tra = trao; ot.trad = trado; ot.dbt = dbto;
      } else {
        const double zclear = 1.0 - zc;
        ot.ref = zclear * oc.ref + zc * refo; ot.refd = zclear * oc.refd + zc * refdo;
        ot.tra = zclear * oc.tra + zc * trao; ot.trad = zclear * oc.trad + zc * trado;
        ot.dbt = zclear * oc.dbt + zc * dbto;
      }
    } else if (!d.mcica && zc != 0.0) {
      // cloud fraction in (0, 1e-12]: lrtchkcld false -> (0,0,1,1) mixed with weight zcloud
      const double zclear = 1.0 - zc;
      const double dbto = sw_dbt(exp_tbl, ztauc + ptc, rmu0);
      ot.ref = zclear * oc.ref; ot.refd = zclear * oc.refd; ot.tra = zclear * oc.tra + zc;
      ot.trad = zclear * oc.trad + zc;
      ot.dbt = zclear * oc.dbt + zc * dbto;
    }
    consume(g, oc, ot);
   
  }
}

// One (column, work item): both sweeps for the item's G g-points.  scr -> this thread's element of a
// [layer][field][G][stride] slab holding the upward-sweep results (rup, rupd) for the clear and -- in cloudy
// (sub-)columns -- the total sky.  The weighted fluxes of the G g-points are added in g-point order before they
// leave through `sink`.
// CLD = false: the caller guarantees a cloud-free column (the cloud code is compiled out: fewer registers).
template <int BAND, int G, bool CLD, bool LDSK, class Sink>
RRTMG_HD void sw_solve_thread(const SwDev &d, const SwTab &T, const double *exp_tbl, int col, int ig0, double *scr, long stride, Sink &sink, const double *kb) {
  const int L = d.nlay, N = d.ncol;
  SwThreadCtx<G> c;
  c.exp_tbl = exp_tbl;
  c.kb = kb;
  c.b = BAND - 16;
  c.ig0 = ig0;
  c.iw0 = T.b[c.b].gs + ig0;
  c.prmu0 = d.cossza[col];
  c.rmu0 = 1.0 / c.prmu0;
  c.laytrop = d.laytrop[col];
  // albedo by band: bands 1-9 and 14 near-IR, 10-13 UV/vis (rrtmg_sw_rad.nomcica.f90:648-659)
  const bool vis = (c.b >= 9 && c.b <= 12);
  const double albp = vis ? d.asdir[col] : d.aldir[col];
  const double albd = vis ? d.asdif[col] : d.aldif[col];
  c.any_cloudy = false;
  c.mword = -1;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    c.mw[g] = 0;
    c.cloudy[g] = false;
    if (CLD && d.icld >= 1) {
      if (d.mcica) {
        for (int w = 0; w < d.nw; ++w) c.cloudy[g] |= (d.mask[((long)(c.iw0 + g) * d.nw + w) * N + col] != 0);
      } else {
        c.cloudy[g] = d.anycld[col] != 0;
      }
    }
    c.any_cloudy |= c.cloudy[g];
  }
  // scratch slab of this (tile, item): [layer][field][lane][G] -- the G values of a lane are one 16-byte access;
  // scr points at this lane's first element, stride = lanes per row (64 on the device, 1 in the host emulation)
  auto SP = [&](int f, int l) -> double * { return scr + ((long)l * F_NTOT + f) * stride * G; };

  // ---- sweep 1: bottom -> top, upward adding recurrence (rrtmg_sw_vrtqdr.f90:114-140) ---------------
  double rupc[G], rupdc[G], rup[G], rupd[G];
#pragma unroll
  for (int g = 0; g < G; ++g) { rupc[g] = albp; rupdc[g] = albd; rup[g] = albp; rupd[g] = albd; }
  for (int l = 0; l < L; ++l) {
    sw_layer_optics<BAND, G, CLD, LDSK>(d, T, c, col, l, [&](int g, const SwLayerOpt &oc, const SwLayerOpt &ot) {
      {
        const double zr = qrcp(1.0 - rupdc[g] * oc.refd);
        const double nrup = oc.ref + (oc.trad * ((oc.tra - oc.dbt) * rupdc[g] + oc.dbt * rupc[g])) * zr;
        const double nrupd = oc.refd + oc.trad * oc.trad * rupdc[g] * zr;
        rupc[g] = nrup; rupdc[g] = nrupd;
      }
      if (CLD && c.cloudy[g]) {
        const double zr = qrcp(1.0 - rupd[g] * ot.refd);
        const double nrup = ot.ref + (ot.trad * ((ot.tra - ot.dbt) * rupd[g] + ot.dbt * rup[g])) * zr;
        const double nrupd = ot.refd + ot.trad * ot.trad * rupd[g] * zr;
        rup[g] = nrup; rupd[g] = nrupd;
      }
    });
    {
      V<G> v0, v1;
#pragma unroll
      for (int g = 0; g < G; ++g) { v0[g] = rupc[g]; v1[g] = rupdc[g]; }
      scr_store<G>(SP(F_RUP, l), stride, v0); scr_store<G>(SP(F_RUPD, l), stride, v1);
      if (CLD && c.any_cloudy) {
#pragma unroll
        for (int g = 0; g < G; ++g) { v0[g] = rup[g]; v1[g] = rupd[g]; }
        scr_store<G>(SP(F_NCLR + F_RUP, l), stride, v0); scr_store<G>(SP(F_NCLR + F_RUPD, l), stride, v1);
      }
    }
  }

  // ---- sweep 2: top -> bottom; downward recurrence + fluxes at every interface (:142-169) -------------
  double tdnc[G], rdndc[G], tdbtc[G], tdn[G], rdnd[G], tdbt[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
#if defined(__HIP_DEVICE_COMPILE__)
    // the downward recurrence and the flux formula are linear in (tdn, tdbt): starting them at the g-point's incoming flux instead
    // of 1 gives the fluxes already weighted (rrtmg_sw_spcvrt.f90:623-627: zincflx * zfd) -- the weights need no registers through
    // the sweep (8 of the clear-sky kernel's 128), at last-place differences in the products
    const double zinc = sw_incflux<BAND>(d, T, col, ig0 + g, c.prmu0);
#else
    const double zinc = 1.0;
#endif
    tdnc[g] = zinc; rdndc[g] = 0.0; tdbtc[g] = zinc; tdn[g] = zinc; rdnd[g] = 0.0; tdbt[g] = zinc;
  }
#if !defined(__HIP_DEVICE_COMPILE__)
  double zinc[G];   // host: the reference's order, the weights applied to the unit fluxes
  for (int g = 0; g < G; ++g) zinc[g] = sw_incflux<BAND>(d, T, col, ig0 + g, c.prmu0);
#endif
  for (int lev = L; lev >= 0; --lev) {
    double sfu[G / 2], sfd[G / 2], scu[G / 2], scd[G / 2];
#pragma unroll
    for (int h = 0; h < G / 2; ++h) { sfu[h] = 0.0; sfd[h] = 0.0; scu[h] = 0.0; scd[h] = 0.0; }
    // (fetching these rows a level ahead costs more in registers than the latency it hides: measured)
    V<G> c_rc, c_rdc, c_r, c_rd;
    if (lev > 0) {
      c_rc = scr_load<G>(SP(F_RUP, lev - 1), stride); c_rdc = scr_load<G>(SP(F_RUPD, lev - 1), stride);
      if (CLD && c.any_cloudy) { c_r = scr_load<G>(SP(F_NCLR + F_RUP, lev - 1), stride); c_rd = scr_load<G>(SP(F_NCLR + F_RUPD, lev - 1), stride); }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const double rc = (lev > 0) ? c_rc[g] : albp;
      const double rdc = (lev > 0) ? c_rdc[g] : albd;
      double zr = qrcp(1.0 - rdndc[g] * rdc);
      const double cu = (tdbtc[g] * rc + (tdnc[g] - tdbtc[g]) * rdc) * zr;
      const double cd = tdbtc[g] + (tdnc[g] - tdbtc[g] + tdbtc[g] * rc * rdndc[g]) * zr;
      double fu = cu, fd = cd;
      if (CLD && c.cloudy[g]) {
        const double r = (lev > 0) ? c_r[g] : albp;
        const double rd = (lev > 0) ? c_rd[g] : albd;
        zr = qrcp(1.0 - rdnd[g] * rd);
        fu = (tdbt[g] * r + (tdn[g] - tdbt[g]) * rd) * zr;
        fd = tdbt[g] + (tdn[g] - tdbt[g] + tdbt[g] * r * rdnd[g]) * zr;
      }
      const int h = g >> 1;
#if defined(__HIP_DEVICE_COMPILE__)
      sfu[h] = sfu[h] + fu; sfd[h] = sfd[h] + fd; scu[h] = scu[h] + cu; scd[h] = scd[h] + cd;
#else
      sfu[h] = sfu[h] + zinc[g] * fu; sfd[h] = sfd[h] + zinc[g] * fd; scu[h] = scu[h] + zinc[g] * cu; scd[h] = scd[h] + zinc[g] * cd;
#endif
    }
    if constexpr (CLD && G == 4) {
      // one slot per chunk: the two pairs' sums added here (pair0 + pair1, the association the flux sums have had since round 1)
      sink.emit(0, lev, sfu[0] + sfu[1], sfd[0] + sfd[1], scu[0] + scu[1], scd[0] + scd[1]);
    } else if constexpr (CLD) {
      sink.emit(0, lev, sfu[0], sfd[0], scu[0], scd[0]);
    } else if constexpr (G == 4) {
      sink.emit_clear(lev, sfu[0] + sfu[1], sfd[0] + sfd[1]);   // one slot per chunk
    } else {
      sink.emit_clear(lev, sfu[0], sfd[0]);
    }
    if (lev > 0) {
      const int l = lev - 1;
      auto down = [&](int g, const SwLayerOpt &oc, const SwLayerOpt &ot) {
        {
          const double zr = qrcp(1.0 - oc.refd * rdndc[g]);
          const double ntdn = tdbtc[g] * oc.tra + (oc.trad * ((tdnc[g] - tdbtc[g]) + tdbtc[g] * oc.ref * rdndc[g])) * zr;
          const double nrdnd = oc.refd + oc.trad * oc.trad * rdndc[g] * zr;
          tdnc[g] = ntdn; rdndc[g] = nrdnd; tdbtc[g] = oc.dbt * tdbtc[g];
        }
        if (CLD && c.cloudy[g]) {
          const double zr = qrcp(1.0 - ot.refd * rdnd[g]);
          const double ntdn = tdbt[g] * ot.tra + (ot.trad * ((tdn[g] - tdbt[g]) + tdbt[g] * ot.ref * rdnd[g])) * zr;
          const double nrdnd = ot.refd + ot.trad * ot.trad * rdnd[g] * zr;
          tdn[g] = ntdn; rdnd[g] = nrdnd; tdbt[g] = ot.dbt * tdbt[g];
        }
      };
      sw_layer_optics<BAND, G, CLD, LDSK>(d, T, c, col, l, down);
    }
  }
}

// Dispatch of one work item (packed, see SwTab) for one column: band switch + G in {4, 2}.
template <int BAND, bool CLD, bool LDSK, class Sink>
RRTMG_HD void sw_solve_band(const SwDev &d, const SwTab &T, const double *exp_tbl, int g, int col, int ig0, double *scr, long stride, Sink &sink, const double *kb) {
  constexpr int ng = SwBandCfg<BAND>::ng;
  if constexpr (ng >= 4) {
    if (g == 4) { sw_solve_thread<BAND, 4, CLD, LDSK>(d, T, exp_tbl, col, ig0, scr, stride, sink, kb); return; }
  }
  if constexpr (ng % 4 != 0) sw_solve_thread<BAND, 2, CLD, LDSK>(d, T, exp_tbl, col, ig0, scr, stride, sink, kb);
}
// LDSK / kb: see sw_taug (kb = the workgroup's LDS slice of the item's band slab, or nullptr with LDSK = false)
template <bool CLD, bool LDSK = false, class Sink>
RRTMG_HD void sw_solve_item(const SwDev &d, const SwTab &T, const double *exp_tbl, int item, int col, double *scr, long stride, Sink &sink, const double *kb = nullptr) {
  const int g = item_g(item), ig0 = item_ig0(item);
  switch (item_band(item) + 16) {
    case 16: sw_solve_band<16, CLD, LDSK>(d, T, exp_tbl, g, col, ig0, scr, stride, sink, kb); break;
    case 17: sw_solve_band<17, CLD, LDSK>(d, T, exp_tbl, g, col, ig0, scr, stride, sink, kb); break;
    case 18: sw_solve_band<18, CLD, LDSK>(d, T, exp_tbl, g, col, ig0, scr, stride, sink, kb); break;
    case 19: sw_solve_band<19, CLD, LDSK>(d, T, exp_tbl, g, col, ig0, scr, stride, sink, kb); break;
    case 20: sw_solve_band<20, CLD, LDSK>(d, T, exp_tbl, g, col, ig0, scr, stride, sink, kb); break;
    case 21: sw_solve_band<21, CLD, LDSK>(d, T, exp_tbl, g, col, ig0, scr, stride, sink, kb); break;
    case 22: sw_solve_band<22, CLD, LDSK>(d, T, exp_tbl, g, col, ig0, scr, stride, sink, kb); break;
    case 23: sw_solve_band<23, CLD, LDSK>(d, T, exp_tbl, g, col, ig0, scr, stride, sink, kb); break;
    case 24: sw_solve_band<24, CLD, LDSK>(d, T, exp_tbl, g, col, ig0, scr, stride, sink, kb); break;
    case 25: sw_solve_band<25, CLD, LDSK>(d, T, exp_tbl, g, col, ig0, scr, stride, sink, kb); break;
    case 26: sw_solve_band<26, CLD, LDSK>(d, T, exp_tbl, g, col, ig0, scr, stride, sink, kb); break;
    case 27: sw_solve_band<27, CLD, LDSK>(d, T, exp_tbl, g, col, ig0, scr, stride, sink, kb); break;
    case 28: sw_solve_band<28, CLD, LDSK>(d, T, exp_tbl, g, col, ig0, scr, stride, sink, kb); break;
    default: sw_solve_band<29, CLD, LDSK>(d, T, exp_tbl, g, col, ig0, scr, stride, sink, kb); break;
  }
}

// spectral integration in g-point order + heating rates (rrtmg_sw_spcvrt.f90:623-627,
// rrtmg_sw_rad.nomcica.f90:777-806)
// one thread per (column, interface level): g-point sum in reference order
// cld = false: the column's tile ran the clear-sky kernel variant, which writes the total-sky planes only
// (SwPartSink::emit_clear); true: the cloudy variant, four planes per slot.  One slot per work item either way.
RRTMG_HD void sw_flux_sums(const SwDev &d, const SwTab &T, int col, int lev, bool cld, double &fu, double &fd, double &cu, double &cd) {
  const int L = d.nlay, P = d.pcols;
  fu = 0.0; fd = 0.0; cu = 0.0; cd = 0.0;
  const long st = (long)(L + 1) * P, slot = 4 * st;
  for (int c = 0; c < T.nitem; ++c) {
    const double *p = d.part + (long)c * slot + (long)lev * P + (col - d.col0);
    fu = fu + part_load(p); fd = fd + part_load(p + st);
    if (cld) { cu = cu + part_load(p + 2 * st); cd = cd + part_load(p + 3 * st); }
  }
  if (!cld) { cu = fu; cd = fd; }
}
RRTMG_HD void sw_flux_level(const SwDev &d, const SwTab &T, int col, int lev, bool cld) {
  double fu, fd, cu, cd;
  sw_flux_sums(d, T, col, lev, cld, fu, fd, cu, cd);
  const long o = (long)lev * d.ncol + col;
  d.swuflx[o] = fu; d.swdflx[o] = fd; d.swuflxc[o] = cu; d.swdflxc[o] = cd;
}
// one thread per (column, layer): heating rates from the net-flux divergence
RRTMG_HD void sw_heat_layer(const SwDev &d, const SwTab &T, int col, int lay) {
  const int N = d.ncol;
  const long o0 = (long)lay * N + col, o1 = o0 + N;
  const double net0 = d.swdflx[o0] - d.swuflx[o0], net1 = d.swdflx[o1] - d.swuflx[o1];
  const double netc0 = d.swdflxc[o0] - d.swuflxc[o0], netc1 = d.swdflxc[o1] - d.swuflxc[o1];
  const double zdpgcp = T.heatfac / d.pdp[o0];
  d.swhrc[o0] = (netc1 - netc0) * zdpgcp;
  d.swhr[o0] = (net1 - net0) * zdpgcp;
}

}  // namespace rrtmg
