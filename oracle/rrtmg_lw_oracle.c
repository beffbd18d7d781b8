/* TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference RRTMG LONGWAVE column algorithm.
 * Plain C, one column at a time, arrays per column as in the Fortran; each function cites what it follows under
 * /root/reference/climt/_lib/rrtmg_lw/.  PARITY UNPINNED (physical): the reference's LW k-data file is a missing
 * blob, so this oracle -- like the product -- is pinned only as algorithm parity against the reference Fortran
 * running on the same SYNTHETIC k-tables (tests/golden/ref_*.npz, oracle/_ref). */
#include "oracle_common.h"

#define NB 16
#define NG 140
static const int ngc_[NB] = {10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2};
static const int nspa_[NB] = {1, 1, 9, 9, 9, 1, 9, 1, 9, 1, 1, 9, 9, 1, 9, 9};
static const int nspb_[NB] = {1, 1, 5, 5, 5, 0, 1, 1, 1, 1, 1, 0, 0, 1, 0, 0};

typedef struct {
  or_store st;
  double exp_tbl[10001], tau_tbl[10001], tfn_tbl[10001];
  double heatfac, grav, avogad;
  int ready;
} lw_oracle;
static lw_oracle G;

int lw_oracle_init(const char *blob, double cpdair, double grav, double avogad, double secdy) {
  if (G.ready) return 0;
  int rc = or_load_blob(&G.st, blob);
  if (rc) return rc;
  or_reduce(&G.st, "lw", NB, 1);
  or_lookup_tables(1, G.exp_tbl, G.tau_tbl, G.tfn_tbl);
  G.heatfac = grav * secdy / (cpdair * 1.e2); /* lwdatinit, rrtmg_lw_init.f90:279 */
  G.grav = grav; G.avogad = avogad;
  G.ready = 1;
  return 0;
}
long lw_oracle_table(const char *name, double *out, long cap) {
  const double *t = !strcmp(name, "lw/tbl/exp_tbl") ? G.exp_tbl : !strcmp(name, "lw/tbl/tau_tbl") ? G.tau_tbl : !strcmp(name, "lw/tbl/tfn_tbl") ? G.tfn_tbl : NULL;
  if (t) { if (out) memcpy(out, t, 10001 * 8); return 10001; }
  or_entry *e = or_find(&G.st, name);
  if (!e || e->dtype) return -1;
  if (out) { if (cap < e->n) return -2; memcpy(out, e->f, (size_t)e->n * 8); }
  return e->n;
}
static double *T(int band, const char *leaf) {
  char nm[64];
  snprintf(nm, sizeof nm, "lw/kg%02d/%s", band, leaf);
  return or_f(&G.st, nm);
}
static double chi(int m, int j) { return or_f(&G.st, "lw/ref/chi_mls")[(m - 1) + 7 * (j - 1)]; }

typedef struct {
  int nlay, laytrop;
  double pavel[OR_MAXL], tavel[OR_MAXL], pz[OR_MAXL + 1], tz[OR_MAXL + 1], tbound, coldry[OR_MAXL], pwvcm, semiss[NB];
  double colh2o[OR_MAXL], colco2[OR_MAXL], colo3[OR_MAXL], coln2o[OR_MAXL], colco[OR_MAXL], colch4[OR_MAXL], colo2[OR_MAXL], colbrd[OR_MAXL];
  double wx[4][OR_MAXL];
  double fac00[OR_MAXL], fac01[OR_MAXL], fac10[OR_MAXL], fac11[OR_MAXL], selffac[OR_MAXL], selffrac[OR_MAXL], forfac[OR_MAXL], forfrac[OR_MAXL];
  double minorfrac[OR_MAXL], scaleminor[OR_MAXL], scaleminorn2[OR_MAXL];
  int jp[OR_MAXL], jt[OR_MAXL], jt1[OR_MAXL], indself[OR_MAXL], indfor[OR_MAXL], indminor[OR_MAXL];
  double planklay[OR_MAXL][NB], planklev[OR_MAXL + 1][NB], plankbnd[NB], dplankbnd_dt[NB];
} lw_col;

/* inatm (rrtmg_lw_rad.nomcica.f90:744-880) + setcoef (rrtmg_lw_setcoef.f90:140-411) */
static void lw_setcoef(lw_col *c, int idrv, const double *h2o, const double *co2, const double *o3, const double *n2o, const double *ch4,
                       const double *o2, const double *cfc11, const double *cfc12, const double *cfc22, const double *ccl4) {
  const double amd = 28.9660, amw = 18.0160, stpfac = 296.0 / 1013.0;
  const double *preflog = or_f(&G.st, "lw/ref/preflog"), *tref = or_f(&G.st, "lw/ref/tref");
  const double *totplnk = or_f(&G.st, "lw/wvn/totplnk"), *totplnkderiv = or_f(&G.st, "lw/wvn/totplnkderiv");
  double amttl = 0.0, wvttl = 0.0;
  const int L = c->nlay;
#define TP(i, ib) totplnk[((i)-1) + 181 * (ib)]
  int indbound = (int)(c->tbound - 159.0);
  if (indbound < 1) indbound = 1; else if (indbound > 180) indbound = 180;
  double tbndfrac = c->tbound - 159.0 - (double)indbound;
  int indlev0 = (int)(c->tz[0] - 159.0);
  if (indlev0 < 1) indlev0 = 1; else if (indlev0 > 180) indlev0 = 180;
  double t0frac = c->tz[0] - 159.0 - (double)indlev0;
  c->laytrop = 0;
  for (int l = 0; l < L; ++l) {
    double amm = (1.0 - h2o[l]) * amd + h2o[l] * amw;
    c->coldry[l] = (c->pz[l] - c->pz[l + 1]) * 1.e3 * G.avogad / (1.e2 * G.grav * amm * (1.0 + h2o[l]));
    double summol = 0.0;
    summol = summol + co2[l]; summol = summol + o3[l]; summol = summol + n2o[l]; summol = summol + 0.0; summol = summol + ch4[l]; summol = summol + o2[l];
    double wbroad = c->coldry[l] * (1.0 - summol);
    double wkl1 = c->coldry[l] * h2o[l], wkl2 = c->coldry[l] * co2[l], wkl3 = c->coldry[l] * o3[l], wkl4 = c->coldry[l] * n2o[l];
    double wkl5 = c->coldry[l] * 0.0, wkl6 = c->coldry[l] * ch4[l], wkl7 = c->coldry[l] * o2[l];
    amttl = amttl + c->coldry[l] + wkl1; wvttl = wvttl + wkl1;
    c->wx[0][l] = c->coldry[l] * (ccl4 ? ccl4[l] : 0.0) * 1.e-20; c->wx[1][l] = c->coldry[l] * (cfc11 ? cfc11[l] : 0.0) * 1.e-20;
    c->wx[2][l] = c->coldry[l] * (cfc12 ? cfc12[l] : 0.0) * 1.e-20; c->wx[3][l] = c->coldry[l] * (cfc22 ? cfc22[l] : 0.0) * 1.e-20;
    /* Planck functions */
    int indlay = (int)(c->tavel[l] - 159.0);
    if (indlay < 1) indlay = 1; else if (indlay > 180) indlay = 180;
    double tlayfrac = c->tavel[l] - 159.0 - (double)indlay;
    int indlev = (int)(c->tz[l + 1] - 159.0);
    if (indlev < 1) indlev = 1; else if (indlev > 180) indlev = 180;
    double tlevfrac = c->tz[l + 1] - 159.0 - (double)indlev;
    for (int ib = 0; ib < NB; ++ib) {
      if (l == 0) {
        double dbdtlev = TP(indbound + 1, ib) - TP(indbound, ib);
        c->plankbnd[ib] = c->semiss[ib] * (TP(indbound, ib) + tbndfrac * dbdtlev);
        dbdtlev = TP(indlev0 + 1, ib) - TP(indlev0, ib);
        c->planklev[0][ib] = TP(indlev0, ib) + t0frac * dbdtlev;
        if (idrv) {
          dbdtlev = totplnkderiv[indbound + 181 * ib] - totplnkderiv[indbound - 1 + 181 * ib];
          c->dplankbnd_dt[ib] = c->semiss[ib] * (totplnkderiv[indbound - 1 + 181 * ib] + tbndfrac * dbdtlev);
        }
      }
      double dbdtlev = TP(indlev + 1, ib) - TP(indlev, ib), dbdtlay = TP(indlay + 1, ib) - TP(indlay, ib);
      c->planklay[l][ib] = TP(indlay, ib) + tlayfrac * dbdtlay;
      c->planklev[l + 1][ib] = TP(indlev, ib) + tlevfrac * dbdtlev;
    }
    double plog = log(c->pavel[l]);
    int jp = (int)(36.0 - 5 * (plog + 0.04));
    if (jp < 1) jp = 1; else if (jp > 58) jp = 58;
    c->jp[l] = jp;
    double fp = 5.0 * (preflog[jp - 1] - plog);
    int jt = (int)(3.0 + (c->tavel[l] - tref[jp - 1]) / 15.0);
    if (jt < 1) jt = 1; else if (jt > 4) jt = 4;
    c->jt[l] = jt;
    double ft = ((c->tavel[l] - tref[jp - 1]) / 15.0) - (double)(jt - 3);
    int jt1 = (int)(3.0 + (c->tavel[l] - tref[jp]) / 15.0);
    if (jt1 < 1) jt1 = 1; else if (jt1 > 4) jt1 = 4;
    c->jt1[l] = jt1;
    double ft1 = ((c->tavel[l] - tref[jp]) / 15.0) - (double)(jt1 - 3);
    double water = wkl1 / c->coldry[l], scalefac = c->pavel[l] * stpfac / c->tavel[l];
    if (plog > 4.56) {
      c->laytrop++;
      c->forfac[l] = scalefac / (1. + water);
      double factor = (332.0 - c->tavel[l]) / 36.0;
      int k = (int)factor; c->indfor[l] = k < 1 ? 1 : (k > 2 ? 2 : k);
      c->forfrac[l] = factor - (double)c->indfor[l];
      c->selffac[l] = water * c->forfac[l];
      factor = (c->tavel[l] - 188.0) / 7.2;
      k = (int)factor - 7; c->indself[l] = k < 1 ? 1 : (k > 9 ? 9 : k);
      c->selffrac[l] = factor - (double)(c->indself[l] + 7);
    } else {
      c->forfac[l] = scalefac / (1. + water);
      double factor = (c->tavel[l] - 188.0) / 36.0;
      c->indfor[l] = 3; c->forfrac[l] = factor - 1.0;
      c->selffac[l] = water * c->forfac[l]; c->selffrac[l] = 0.0; c->indself[l] = 1;
    }
    c->scaleminor[l] = c->pavel[l] / c->tavel[l];
    c->scaleminorn2[l] = (c->pavel[l] / c->tavel[l]) * (wbroad / (c->coldry[l] + wkl1));
    double factor = (c->tavel[l] - 180.8) / 7.2;
    int k = (int)factor; c->indminor[l] = k < 1 ? 1 : (k > 18 ? 18 : k);
    c->minorfrac[l] = factor - (double)c->indminor[l];
    c->colh2o[l] = 1.e-20 * wkl1; c->colco2[l] = 1.e-20 * wkl2; c->colo3[l] = 1.e-20 * wkl3; c->coln2o[l] = 1.e-20 * wkl4;
    c->colco[l] = 1.e-20 * wkl5; c->colch4[l] = 1.e-20 * wkl6; c->colo2[l] = 1.e-20 * wkl7;
    if (c->colco2[l] == 0.0) c->colco2[l] = 1.e-32 * c->coldry[l];
    if (c->colo3[l] == 0.0) c->colo3[l] = 1.e-32 * c->coldry[l];
    if (c->coln2o[l] == 0.0) c->coln2o[l] = 1.e-32 * c->coldry[l];
    if (c->colco[l] == 0.0) c->colco[l] = 1.e-32 * c->coldry[l];
    if (c->colch4[l] == 0.0) c->colch4[l] = 1.e-32 * c->coldry[l];
    c->colbrd[l] = 1.e-20 * wbroad;
    double compfp = 1. - fp;
    c->fac10[l] = compfp * ft; c->fac00[l] = compfp * (1.0 - ft); c->fac11[l] = fp * ft1; c->fac01[l] = fp * (1.0 - ft1);
    c->selffac[l] = c->colh2o[l] * c->selffac[l];
    c->forfac[l] = c->colh2o[l] * c->forfac[l];
  }
  double wvsh = (amw * wvttl) / (amd * amttl);
  c->pwvcm = wvsh * (1.e3 * c->pz[0]) / (1.e2 * G.grav);
#undef TP
}

/* ---- taumol (rrtmg_lw_taumol.f90:287-3147) ------------------------------------------------------------------------ */
typedef struct { double taug[OR_MAXL][NG], fracs[OR_MAXL][NG]; } lw_tau;
typedef struct { double speccomb, specparm, fs; int js; } spec_t;
static spec_t mkspec(double colx, double rat, double coly, double mult) {
  spec_t r;
  r.speccomb = colx + rat * coly;
  r.specparm = colx / r.speccomb;
  if (r.specparm >= 1.0 - 1.e-6) r.specparm = 1.0 - 1.e-6;
  double specmult = mult * r.specparm;
  r.js = 1 + (int)specmult;
  r.fs = fmod(specmult, 1.0);
  return r;
}
/* lower-atmosphere major term with 3-point end-zone blending (rrtmg_lw_taumol.f90:550-609, :622-668) */
static double major_lo(const double *absa, int ind, spec_t sp, double f0, double f1) {
  if (sp.specparm < 0.125) {
    double p = sp.fs - 1, p4 = (p * p) * (p * p), fk0 = p4, fk1 = 1 - p - 2.0 * p4, fk2 = p + p4;
    double fac0 = fk0 * f0, fac1 = fk1 * f0, fac2 = fk2 * f0, fac0t = fk0 * f1, fac1t = fk1 * f1, fac2t = fk2 * f1;
    return sp.speccomb * (fac0 * absa[ind] + fac1 * absa[ind + 1] + fac2 * absa[ind + 2] + fac0t * absa[ind + 9] + fac1t * absa[ind + 10] + fac2t * absa[ind + 11]);
  } else if (sp.specparm > 0.875) {
    double p = -sp.fs, p4 = (p * p) * (p * p), fk0 = p4, fk1 = 1 - p - 2.0 * p4, fk2 = p + p4;
    double fac0 = fk0 * f0, fac1 = fk1 * f0, fac2 = fk2 * f0, fac0t = fk0 * f1, fac1t = fk1 * f1, fac2t = fk2 * f1;
    return sp.speccomb * (fac2 * absa[ind - 1] + fac1 * absa[ind] + fac0 * absa[ind + 1] + fac2t * absa[ind + 8] + fac1t * absa[ind + 9] + fac0t * absa[ind + 10]);
  }
  double fac0 = (1.0 - sp.fs) * f0, fac0t = (1.0 - sp.fs) * f1, fac1 = sp.fs * f0, fac1t = sp.fs * f1;
  return sp.speccomb * (fac0 * absa[ind] + fac1 * absa[ind + 1] + fac0t * absa[ind + 9] + fac1t * absa[ind + 10]);
}
static double major_up(const double *absb, int ind, spec_t sp, double f0, double f1) {
  double fac0 = (1.0 - sp.fs) * f0, fac0t = (1.0 - sp.fs) * f1, fac1 = sp.fs * f0, fac1t = sp.fs * f1;
  return sp.speccomb * (fac0 * absb[ind] + fac1 * absb[ind + 1] + fac0t * absb[ind + 5] + fac1t * absb[ind + 6]);
}
static double adjcol(double col, double coldry, double chiref, double e20, double thresh, double a, double e, double chimul) {
  double chi_ = col / coldry, rat = e20 * chi_ / chiref;
  if (rat > thresh) { double adjfac = a + pow(rat - a, e); return adjfac * chimul * coldry * 1.e-20; }
  return col;
}

static void lw_taumol(const lw_col *c, lw_tau *o) {
  const int L = c->nlay;
  const double E20F = (double)1.e20f;
  int gs = 0;
  for (int b = 0; b < NB; ++b) {
    const int band = b + 1, ng = ngc_[b];
    const double *absa_ = T(band, "absa"), *absb_ = (nspb_[b] || band == 16) ? T(band, "absb") : NULL;
    const double *selfref_ = T(band, "selfref"), *forref_ = T(band, "forref"), *fraca = T(band, "fracrefa");
    const double *fracb = (band == 6 || band == 12 || band == 15) ? NULL : T(band, "fracrefb");
    /* key species pair (x, y) of the lower-atmosphere binary bands; chi indices */
    int mx = 1, my = 2;
    if (band == 7) my = 3; if (band == 9 || band == 16) my = 6; if (band == 13) my = 4; if (band == 15) { mx = 4; my = 2; }
    for (int l = 0; l < L; ++l) {
      const int lower = (l + 1) <= c->laytrop, jp = c->jp[l];
      const double colx = mx == 1 ? c->colh2o[l] : c->coln2o[l];
      const double coly = my == 2 ? c->colco2[l] : my == 3 ? c->colo3[l] : my == 6 ? c->colch4[l] : c->coln2o[l];
      const int i0 = (jp - 1) * 5 + (c->jt[l] - 1), i1 = jp * 5 + (c->jt1[l] - 1);
      const int u0 = ((jp - 13) * 5 + (c->jt[l] - 1)) * (band == 16 ? 0 : 1), u1 = ((jp - 12) * 5 + (c->jt1[l] - 1)) * (band == 16 ? 0 : 1);
      const int indm = c->indminor[l];
      for (int ig = 0; ig < ng; ++ig) {
        const double *absa = absa_ + (long)ig * 65 * nspa_[b], *absb = absb_ ? absb_ + (long)ig * 235 * (band == 16 ? 1 : nspb_[b]) : NULL;
        const double *selfref = selfref_ + ig * 10, *forref = forref_ + ig * 4;
        const double tauself = c->selffac[l] * (selfref[c->indself[l] - 1] + c->selffrac[l] * (selfref[c->indself[l]] - selfref[c->indself[l] - 1]));
        const double taufor = c->forfac[l] * (forref[c->indfor[l] - 1] + c->forfrac[l] * (forref[c->indfor[l]] - forref[c->indfor[l] - 1]));
#define M4(k, a0, a1) (c->fac00[l] * (k)[a0] + c->fac10[l] * (k)[(a0) + 1] + c->fac01[l] * (k)[a1] + c->fac11[l] * (k)[(a1) + 1])
#define MIN1(name) (T(band, name)[19 * ig + indm - 1] + c->minorfrac[l] * (T(band, name)[19 * ig + indm] - T(band, name)[19 * ig + indm - 1]))
        double tg = 0.0, fr = 0.0;
        if (lower && nspa_[b] == 9) {
          spec_t sp = mkspec(colx, chi(mx, jp) / chi(my, jp), coly, 8.0), sp1 = mkspec(colx, chi(mx, jp + 1) / chi(my, jp + 1), coly, 8.0);
          static const int jpl_ref[NB] = {0, 0, 9, 11, 5, 0, 3, 0, 9, 0, 0, 10, 5, 0, 1, 6};
          spec_t pl = mkspec(colx, chi(mx, jpl_ref[b]) / chi(my, jpl_ref[b]), coly, 8.0);
          tg = major_lo(absa, i0 * 9 + sp.js - 1, sp, c->fac00[l], c->fac10[l]) + major_lo(absa, i1 * 9 + sp1.js - 1, sp1, c->fac01[l], c->fac11[l]) + tauself + taufor;
          fr = fraca[ig + ng * (pl.js - 1)] + pl.fs * (fraca[ig + ng * pl.js] - fraca[ig + ng * (pl.js - 1)]);
          /* minor species interpolated in their own reference mixture */
          static const int jm_ref[NB] = {0, 0, 3, 0, 7, 0, 3, 0, 3, 0, 0, 0, 1, 0, 1, 0};
          if (jm_ref[b]) {
            spec_t sm = mkspec(colx, chi(mx, jm_ref[b]) / chi(my, jm_ref[b]), coly, 8.0);
            const char *mn = band == 3 || band == 9 ? "ka_mn2o" : band == 5 ? "ka_mo3" : band == 15 ? "ka_mn2" : "ka_mco2";
            const double *m = T(band, mn) + (long)9 * 19 * ig + 9 * (indm - 1) + (sm.js - 1);
            double m1 = m[0] + sm.fs * (m[1] - m[0]), m2 = m[9] + sm.fs * (m[10] - m[9]);
            double absm = m1 + c->minorfrac[l] * (m2 - m1);
            if (band == 3 || band == 9) tg = tg + adjcol(c->coln2o[l], c->coldry[l], chi(4, jp + 1), 1.e20, 1.5, 0.5, 0.65, chi(4, jp + 1)) * absm;
            if (band == 5) tg = tg + absm * c->colo3[l] + c->wx[0][l] * T(5, "ccl4")[ig];
            if (band == 7) tg = tg + adjcol(c->colco2[l], c->coldry[l], chi(2, jp + 1), E20F, 3.0, 3.0, 0.79, chi(2, jp + 1)) * absm;
            if (band == 13) {
              spec_t s3 = mkspec(colx, chi(1, 3) / chi(4, 3), coly, 8.0);
              const double *mc = T(13, "ka_mco") + (long)9 * 19 * ig + 9 * (indm - 1) + (s3.js - 1);
              double c1 = mc[0] + s3.fs * (mc[1] - mc[0]), c2 = mc[9] + s3.fs * (mc[10] - mc[9]);
              tg = tg + adjcol(c->colco2[l], c->coldry[l], 3.55e-4, 1.e20, 3.0, 2.0, 0.68, (double)3.55e-4f) * absm + c->colco[l] * (c1 + c->minorfrac[l] * (c2 - c1));
            }
            if (band == 15) tg = tg + (c->colbrd[l] * c->scaleminor[l]) * absm;
          }
        } else if (lower) {
          double m4 = M4(absa, i0, i1);
          if (band == 1) {
            double corradj = 1.;
            if (c->pavel[l] < 250.0) corradj = 1.0 - 0.15 * (250.0 - c->pavel[l]) / 154.4;
            tg = corradj * (c->colh2o[l] * m4 + tauself + taufor + (c->colbrd[l] * c->scaleminorn2[l]) * MIN1("ka_mn2"));
          } else if (band == 2) tg = (1.0 - .05 * (c->pavel[l] - 100.0) / 900.0) * (c->colh2o[l] * m4 + tauself + taufor);
          else if (band == 6) tg = c->colh2o[l] * m4 + tauself + taufor + adjcol(c->colco2[l], c->coldry[l], chi(2, jp + 1), 1.e20, 3.0, 2.0, 0.77, chi(2, jp + 1)) * MIN1("ka_mco2") +
                                   c->wx[1][l] * T(6, "cfc11adj")[ig] + c->wx[2][l] * T(6, "cfc12")[ig];
          else if (band == 8) tg = c->colh2o[l] * m4 + tauself + taufor + adjcol(c->colco2[l], c->coldry[l], chi(2, jp + 1), 1.e20, 3.0, 2.0, 0.65, chi(2, jp + 1)) * MIN1("ka_mco2") +
                                   c->colo3[l] * MIN1("ka_mo3") + c->coln2o[l] * MIN1("ka_mn2o") + c->wx[2][l] * T(8, "cfc12")[ig] + c->wx[3][l] * T(8, "cfc22adj")[ig];
          else if (band == 10) tg = c->colh2o[l] * m4 + tauself + taufor;
          else if (band == 11) tg = c->colh2o[l] * m4 + tauself + taufor + (c->colo2[l] * c->scaleminor[l]) * MIN1("ka_mo2");
          else tg = c->colco2[l] * m4 + tauself + taufor; /* 14 */
          fr = fraca[ig];
        } else if (nspb_[b] == 5) {
          const double ux = band == 3 ? c->colh2o[l] : c->colo3[l];
          const int umx = band == 3 ? 1 : 3;
          spec_t sp = mkspec(ux, chi(umx, jp) / chi(2, jp), c->colco2[l], 4.0), sp1 = mkspec(ux, chi(umx, jp + 1) / chi(2, jp + 1), c->colco2[l], 4.0);
          const int jr = band == 5 ? 43 : 13;
          spec_t pl = mkspec(ux, chi(umx, jr) / chi(2, jr), c->colco2[l], 4.0);
          tg = major_up(absb, u0 * 5 + sp.js - 1, sp, c->fac00[l], c->fac10[l]) + major_up(absb, u1 * 5 + sp1.js - 1, sp1, c->fac01[l], c->fac11[l]);
          if (band == 3) {
            spec_t sm = mkspec(ux, chi(1, 13) / chi(2, 13), c->colco2[l], 4.0);
            const double *m = T(3, "kb_mn2o") + (long)5 * 19 * ig + 5 * (indm - 1) + (sm.js - 1);
            double m1 = m[0] + sm.fs * (m[1] - m[0]), m2 = m[5] + sm.fs * (m[6] - m[5]);
            tg = tg + taufor + adjcol(c->coln2o[l], c->coldry[l], chi(4, jp + 1), E20F, 1.5, 0.5, 0.65, chi(4, jp + 1)) * (m1 + c->minorfrac[l] * (m2 - m1));
          }
          if (band == 5) tg = tg + c->wx[0][l] * T(5, "ccl4")[ig];
          if (band == 4) {
            static const float sc[7] = {0.92f, 0.88f, 1.07f, 1.1f, 0.99f, 0.88f, 0.943f};
            if (ig >= 7) tg = tg * (double)sc[ig - 7];
          }
          fr = fracb[ig + ng * (pl.js - 1)] + pl.fs * (fracb[ig + ng * pl.js] - fracb[ig + ng * (pl.js - 1)]);
        } else {
          double m4 = absb ? M4(absb, u0, u1) : 0.0;
          fr = fracb ? fracb[ig] : 0.0;
          if (band == 1) tg = (1.0 - 0.15 * (c->pavel[l] / 95.6)) * (c->colh2o[l] * m4 + taufor + (c->colbrd[l] * c->scaleminorn2[l]) * MIN1("kb_mn2"));
          else if (band == 2 || band == 10) tg = c->colh2o[l] * m4 + taufor;
          else if (band == 6) { tg = 0.0 + c->wx[1][l] * T(6, "cfc11adj")[ig] + c->wx[2][l] * T(6, "cfc12")[ig]; fr = fraca[ig]; }
          else if (band == 7) {
            tg = c->colo3[l] * m4 + adjcol(c->colco2[l], c->coldry[l], chi(2, jp + 1), E20F, 3.0, 2.0, 0.79, chi(2, jp + 1)) * MIN1("kb_mco2");
            static const double sc[6] = {0.92, 0.88, 1.07, 1.1, 0.99, 0.855};
            if (ig >= 5 && ig <= 10) tg = tg * sc[ig - 5];
          } else if (band == 8) tg = c->colo3[l] * m4 + adjcol(c->colco2[l], c->coldry[l], chi(2, jp + 1), 1.e20, 3.0, 2.0, 0.65, chi(2, jp + 1)) * MIN1("kb_mco2") +
                                    c->coln2o[l] * MIN1("kb_mn2o") + c->wx[2][l] * T(8, "cfc12")[ig] + c->wx[3][l] * T(8, "cfc22adj")[ig];
          else if (band == 9) tg = c->colch4[l] * m4 + adjcol(c->coln2o[l], c->coldry[l], chi(4, jp + 1), 1.e20, 1.5, 0.5, 0.65, chi(4, jp + 1)) * MIN1("kb_mn2o");
          else if (band == 11) tg = c->colh2o[l] * m4 + taufor + (c->colo2[l] * c->scaleminor[l]) * MIN1("kb_mo2");
          else if (band == 13) tg = c->colo3[l] * MIN1("kb_mo3");
          else if (band == 14) tg = c->colco2[l] * m4;
          else if (band == 16) tg = c->colch4[l] * m4;
          else { tg = 0.0; fr = 0.0; } /* 12, 15 */
        }
#undef M4
#undef MIN1
        o->taug[l][gs + ig] = tg;
        o->fracs[l][gs + ig] = fr;
      }
    }
    gs += ng;
  }
}

typedef struct {
  int ncol, nlay, mcica, icld, idrv, inflag, iceflag, liqflag, irng, permuteseed;
  const double *play, *plev, *tlay, *tlev, *tsfc, *h2o, *o3, *co2, *ch4, *n2o, *o2, *cfc11, *cfc12, *cfc22, *ccl4, *emis;
  const double *cldfr, *taucld, *cicewp, *cliqwp, *reice, *reliq, *tauaer;
  double *uflx, *dflx, *hr, *uflxc, *dflxc, *hrc, *duflx_dt, *duflxc_dt;
} lw_args;

/* ice / liquid absorption coefficients of one layer (shared by cldprop and cldprmc) */
static int lw_abscoef(int iceflag, int liqflag, double ciwp, double clwp, double radice, double radliq, double *abscoice, double *abscoliq,
                      int *iceind, int *liqind, int *ncbands) {
  or_store *s = &G.st;
  *iceind = 0; *liqind = 0;
  if (ciwp == 0.0) { abscoice[0] = 0.0; }
  else if (iceflag == 0) { if (radice < 10.0) return 11; abscoice[0] = or_f(s, "lw/cld/absice0")[0] + or_f(s, "lw/cld/absice0")[1] / radice; }
  else if (iceflag == 1) {
    if (radice < 13.0 || radice > 130.) return 11;
    *ncbands = 5;
    for (int ib = 0; ib < 5; ++ib) abscoice[ib] = or_f(s, "lw/cld/absice1")[2 * ib] + or_f(s, "lw/cld/absice1")[2 * ib + 1] / radice;
    *iceind = 1;
  } else if (iceflag == 2 || iceflag == 3) {
    int nr = iceflag == 2 ? 43 : 46;
    if (radice < 5.0 || radice > (iceflag == 2 ? 131.0 : 140.0)) return 11;
    *ncbands = 16;
    double factor = (radice - 2.0) / 3.0;
    int index = (int)factor;
    if (index == nr) index = nr - 1;
    double fint = factor - (double)index;
    const double *t = or_f(s, iceflag == 2 ? "lw/cld/absice2" : "lw/cld/absice3");
    for (int ib = 0; ib < 16; ++ib) abscoice[ib] = t[(index - 1) + nr * ib] + fint * (t[index + nr * ib] - (t[(index - 1) + nr * ib]));
    *iceind = 2;
  }
  if (clwp == 0.0) { abscoliq[0] = 0.0; if (*iceind == 1) *iceind = 2; }
  else if (liqflag == 0) { abscoliq[0] = or_f(s, "lw/cld/absliq0")[0]; if (*iceind == 1) *iceind = 2; }
  else if (liqflag == 1) {
    if (radliq < 2.5 || radliq > 60.) return 12;
    int index = (int)(radliq - 1.5);
    if (index == 0) index = 1;
    if (index == 58) index = 57;
    double fint = radliq - 1.5 - (double)index;
    *ncbands = 16;
    const double *t = or_f(s, "lw/cld/absliq1");
    for (int ib = 0; ib < 16; ++ib) abscoliq[ib] = t[(index - 1) + 58 * ib] + fint * (t[index + 58 * ib] - (t[(index - 1) + 58 * ib]));
    *liqind = 2;
  }
  return 0;
}

/* Maximum/random overlap factors of one column, rrtmg_lw_rtrnmr.f90:326-479.  Arrays keep the reference's own index
 * 0 .. nlayers+1; cf[] is cldfrac with cf[0] = cf[nlayers+1] = 0: the reference reads cldfrac(0) and cldfrac(nlayers+1)
 * out of bounds there, but only as a factor of products whose other factors are zero at those levels. */
typedef struct {
  double faccld1[OR_MAXL + 2], faccld2[OR_MAXL + 2], facclr1[OR_MAXL + 2], facclr2[OR_MAXL + 2], faccmb1[OR_MAXL + 2], faccmb2[OR_MAXL + 2];
  double faccld1d[OR_MAXL + 2], faccld2d[OR_MAXL + 2], facclr1d[OR_MAXL + 2], facclr2d[OR_MAXL + 2], faccmb1d[OR_MAXL + 2], faccmb2d[OR_MAXL + 2];
  int istcld[OR_MAXL + 2], istcldd[OR_MAXL + 2];
} lw_mrfac;
static void lw_mr_factors(int nlayers, const double *cf, const int *icldlyr /* 1-based */, lw_mrfac *m) {
  double rat1 = 0.0, rat2 = 0.0, fmax, fmin;
  memset(m, 0, sizeof *m);
  m->istcld[1] = 1;
  m->istcldd[nlayers] = 1;
  for (int lev = 1; lev <= nlayers; ++lev) {
    if (icldlyr[lev] == 1) {
      m->istcld[lev + 1] = 0;
      if (lev == nlayers) {
        m->faccld1[lev + 1] = 0.; m->faccld2[lev + 1] = 0.; m->facclr1[lev + 1] = 0.; m->facclr2[lev + 1] = 0.; m->faccmb1[lev + 1] = 0.; m->faccmb2[lev + 1] = 0.;
      } else if (cf[lev + 1] >= cf[lev]) {
        m->faccld1[lev + 1] = 0.; m->faccld2[lev + 1] = 0.;
        if (m->istcld[lev] == 1) {
          m->facclr1[lev + 1] = 0.; m->facclr2[lev + 1] = 0.;
          if (cf[lev] < 1.) m->facclr2[lev + 1] = (cf[lev + 1] - cf[lev]) / (1. - cf[lev]);
          m->facclr2[lev] = 0.; m->faccld2[lev] = 0.;
        } else {
          fmax = cf[lev] > cf[lev - 1] ? cf[lev] : cf[lev - 1];
          if (cf[lev + 1] > fmax) { m->facclr1[lev + 1] = rat2; m->facclr2[lev + 1] = (cf[lev + 1] - fmax) / (1. - fmax); }
          else if (cf[lev + 1] < fmax) { m->facclr1[lev + 1] = (cf[lev + 1] - cf[lev]) / (cf[lev - 1] - cf[lev]); m->facclr2[lev + 1] = 0.; }
          else { m->facclr1[lev + 1] = rat2; m->facclr2[lev + 1] = 0.; }
        }
        if (m->facclr1[lev + 1] > 0. || m->facclr2[lev + 1] > 0.) { rat1 = 1.; rat2 = 0.; } else { rat1 = 0.; rat2 = 0.; }
      } else {
        m->facclr1[lev + 1] = 0.; m->facclr2[lev + 1] = 0.;
        if (m->istcld[lev] == 1) {
          m->faccld1[lev + 1] = 0.; m->faccld2[lev + 1] = (cf[lev] - cf[lev + 1]) / cf[lev];
          m->facclr2[lev] = 0.; m->faccld2[lev] = 0.;
        } else {
          fmin = cf[lev] < cf[lev - 1] ? cf[lev] : cf[lev - 1];
          if (cf[lev + 1] <= fmin) { m->faccld1[lev + 1] = rat1; m->faccld2[lev + 1] = (fmin - cf[lev + 1]) / fmin; }
          else { m->faccld1[lev + 1] = (cf[lev] - cf[lev + 1]) / (cf[lev] - fmin); m->faccld2[lev + 1] = 0.; }
        }
        if (m->faccld1[lev + 1] > 0. || m->faccld2[lev + 1] > 0.) { rat1 = 0.; rat2 = 1.; } else { rat1 = 0.; rat2 = 0.; }
      }
      m->faccmb1[lev + 1] = m->facclr1[lev + 1] * m->faccld2[lev] * cf[lev - 1];
      m->faccmb2[lev + 1] = m->faccld1[lev + 1] * m->facclr2[lev] * (1. - cf[lev - 1]);
    } else {
      m->istcld[lev + 1] = 1;
    }
  }
  for (int lev = nlayers; lev >= 1; --lev) {
    if (icldlyr[lev] == 1) {
      m->istcldd[lev - 1] = 0;
      if (lev == 1) {
        m->faccld1d[lev - 1] = 0.; m->faccld2d[lev - 1] = 0.; m->facclr1d[lev - 1] = 0.; m->facclr2d[lev - 1] = 0.; m->faccmb1d[lev - 1] = 0.; m->faccmb2d[lev - 1] = 0.;
      } else if (cf[lev - 1] >= cf[lev]) {
        m->faccld1d[lev - 1] = 0.; m->faccld2d[lev - 1] = 0.;
        if (m->istcldd[lev] == 1) {
          m->facclr1d[lev - 1] = 0.; m->facclr2d[lev - 1] = 0.;
          if (cf[lev] < 1.) m->facclr2d[lev - 1] = (cf[lev - 1] - cf[lev]) / (1. - cf[lev]);
          m->facclr2d[lev] = 0.; m->faccld2d[lev] = 0.;
        } else {
          fmax = cf[lev] > cf[lev + 1] ? cf[lev] : cf[lev + 1];
          if (cf[lev - 1] > fmax) { m->facclr1d[lev - 1] = rat2; m->facclr2d[lev - 1] = (cf[lev - 1] - fmax) / (1. - fmax); }
          else if (cf[lev - 1] < fmax) { m->facclr1d[lev - 1] = (cf[lev - 1] - cf[lev]) / (cf[lev + 1] - cf[lev]); m->facclr2d[lev - 1] = 0.; }
          else { m->facclr1d[lev - 1] = rat2; m->facclr2d[lev - 1] = 0.; }
        }
        if (m->facclr1d[lev - 1] > 0. || m->facclr2d[lev - 1] > 0.) { rat1 = 1.; rat2 = 0.; } else { rat1 = 0.; rat2 = 0.; }
      } else {
        m->facclr1d[lev - 1] = 0.; m->facclr2d[lev - 1] = 0.;
        if (m->istcldd[lev] == 1) {
          m->faccld1d[lev - 1] = 0.; m->faccld2d[lev - 1] = (cf[lev] - cf[lev - 1]) / cf[lev];
          m->facclr2d[lev] = 0.; m->faccld2d[lev] = 0.;
        } else {
          fmin = cf[lev] < cf[lev + 1] ? cf[lev] : cf[lev + 1];
          if (cf[lev - 1] <= fmin) { m->faccld1d[lev - 1] = rat1; m->faccld2d[lev - 1] = (fmin - cf[lev - 1]) / fmin; }
          else { m->faccld1d[lev - 1] = (cf[lev] - cf[lev - 1]) / (cf[lev] - fmin); m->faccld2d[lev - 1] = 0.; }
        }
        if (m->faccld1d[lev - 1] > 0. || m->faccld2d[lev - 1] > 0.) { rat1 = 0.; rat2 = 1.; } else { rat1 = 0.; rat2 = 0.; }
      }
      m->faccmb1d[lev - 1] = m->facclr1d[lev - 1] * m->faccld2d[lev] * cf[lev + 1];
      m->faccmb2d[lev - 1] = m->faccld1d[lev - 1] * m->facclr2d[lev] * (1. - cf[lev + 1]);
    } else {
      m->istcldd[lev - 1] = 1;
    }
  }
}

/* rrtmg_lw driver (rrtmg_lw_rad.nomcica.f90:453-567) with cldprop (rrtmg_lw_cldprop.f90:118-272) / cldprmc
 * (rrtmg_lw_cldprmc.f90:103-250) and rtrn / rtrnmc (rrtmg_lw_rtrn.f90:261-587, rrtmg_lw_rtrnmc.f90) */
int lw_oracle_fluxes(const lw_args *a) {
  const int N = a->ncol, L = a->nlay;
  int icld = a->icld;
  if (icld < 0 || icld > 3) icld = 2;
  const int mr = !a->mcica && icld >= 2; /* maximum/random overlap of the band cloud fractions: rtrnmr (rrtmg_lw_rad.nomcica.f90:527-544) */
  const double fluxfac = (2.0 * asin(1.0)) * 2.e4, wtdiff = 0.5, rec_6 = 0.166667, bpade = 1.0 / 0.278;
  const double *delwave = or_f(&G.st, "lw/wvn/delwave");
  static const int icb1[16] = {1, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 5};
  unsigned char *cloudy = NULL;
  if (a->mcica && icld >= 1) {
    if (a->inflag == 1) return 20;
    cloudy = (unsigned char *)malloc((size_t)NG * N * L);
    int rc = or_subcol_mask(N, L, NG, icld, a->irng, a->permuteseed, a->play, a->cldfr, cloudy);
    if (rc) { free(cloudy); return rc; }
  }
  lw_col *c = (lw_col *)malloc(sizeof(lw_col));
  lw_tau *tm = (lw_tau *)malloc(sizeof(lw_tau));
  int err = 0;
  for (int col = 0; col < N && !err; ++col) {
    double h2o[OR_MAXL], co2[OR_MAXL], o3[OR_MAXL], n2o[OR_MAXL], ch4[OR_MAXL], o2[OR_MAXL], f11[OR_MAXL], f12[OR_MAXL], f22[OR_MAXL], cc4[OR_MAXL];
    double cldfrac[OR_MAXL];
    static double taucloud[OR_MAXL][NB];
    c->nlay = L; c->tbound = a->tsfc[col]; c->pz[0] = a->plev[col]; c->tz[0] = a->tlev[col];
    for (int l = 0; l < L; ++l) {
      long i = (long)l * N + col;
      c->pavel[l] = a->play[i]; c->tavel[l] = a->tlay[i]; c->pz[l + 1] = a->plev[i + N]; c->tz[l + 1] = a->tlev[i + N];
      h2o[l] = a->h2o[i]; co2[l] = a->co2[i]; o3[l] = a->o3[i]; n2o[l] = a->n2o[i]; ch4[l] = a->ch4[i]; o2[l] = a->o2[i];
      f11[l] = a->cfc11 ? a->cfc11[i] : 0; f12[l] = a->cfc12 ? a->cfc12[i] : 0; f22[l] = a->cfc22 ? a->cfc22[i] : 0; cc4[l] = a->ccl4 ? a->ccl4[i] : 0;
      cldfrac[l] = icld >= 1 && a->cldfr ? a->cldfr[i] : 0.0;
    }
    for (int ib = 0; ib < NB; ++ib) c->semiss[ib] = a->emis[(long)ib * N + col];
    lw_setcoef(c, a->idrv, h2o, co2, o3, n2o, ch4, o2, f11, f12, f22, cc4);
    lw_taumol(c, tm);
    /* cloud optics */
    int ncbands = 1;
    for (int l = 0; l < L; ++l) {
      long i = (long)l * N + col;
      for (int ib = 0; ib < NB; ++ib) taucloud[l][ib] = 0.0;
      if (icld == 0) continue;
      double ciwp = a->cicewp ? a->cicewp[i] : 0.0, clwp = a->cliqwp ? a->cliqwp[i] : 0.0, cwp = ciwp + clwp;
      if (a->mcica) {
        for (int ib = 0; ib < NB; ++ib) {
          double tin = a->taucld ? a->taucld[i * NB + ib] : 0.0;
          taucloud[l][ib] = tin;
          if (a->inflag == 2 && (cwp >= 1.e-20 || tin >= 1.e-20)) {
            double aci[16] = {0}, acl[16] = {0}; int ii, li, nc = 1;
            int rc = lw_abscoef(a->iceflag, a->liqflag, ciwp, clwp, a->reice[i], a->reliq[i], aci, acl, &ii, &li, &nc);
            if (rc) { err = rc; break; }
            double ai = ciwp == 0.0 ? 0.0 : (a->iceflag == 0 ? aci[0] : a->iceflag == 1 ? aci[icb1[ib] - 1] : aci[ib]);
            double al = clwp == 0.0 ? 0.0 : (a->liqflag == 0 ? acl[0] : acl[ib]);
            taucloud[l][ib] = ciwp * ai + clwp * al;
          }
        }
      } else {
        double tauctot = 0.0;
        for (int ib = 0; ib < NB; ++ib) tauctot = tauctot + (a->taucld ? a->taucld[i * NB + ib] : 0.0);
        if (!(cldfrac[l] >= 1.e-20 && (cwp >= 1.e-20 || tauctot >= 1.e-20))) continue;
        if (a->inflag == 0) { ncbands = 16; for (int ib = 0; ib < 16; ++ib) taucloud[l][ib] = a->taucld[i * NB + ib]; }
        else if (a->inflag == 1) { ncbands = 16; for (int ib = 0; ib < 16; ++ib) taucloud[l][ib] = or_f(&G.st, "lw/cld/abscld1")[0] * cwp; }
        else if (a->inflag == 2) {
          double aci[16] = {0}, acl[16] = {0}; int ii, li;
          int rc = lw_abscoef(a->iceflag, a->liqflag, ciwp, clwp, a->reice[i], a->reliq[i], aci, acl, &ii, &li, &ncbands);
          if (rc) { err = rc; break; }
          for (int ib = 0; ib < ncbands; ++ib)
            taucloud[l][ib] = ciwp * aci[ii == 0 ? 0 : ii == 1 ? icb1[ib] - 1 : ib] + clwp * acl[li == 0 ? 0 : ib];
        }
      }
    }
    if (err) break;
    /* rtrn */
    double secdiff[NB];
    {
      static const double a0[16] = {1.66, 1.55, 1.58, 1.66, 1.54, 1.454, 1.89, 1.33, 1.668, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66};
      static const double a1[16] = {0.00, 0.25, 0.22, 0.00, 0.13, 0.446, -0.10, 0.40, -0.006, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
      static const double a2[16] = {0.00, -12.0, -11.7, 0.00, -0.72, -0.243, 0.19, -0.062, 0.414, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
      for (int ib = 0; ib < NB; ++ib) {
        if (ib == 0 || ib == 3 || ib >= 9) secdiff[ib] = 1.66;
        else { secdiff[ib] = a0[ib] + a1[ib] * exp(a2[ib] * c->pwvcm); if (secdiff[ib] > 1.80) secdiff[ib] = 1.80; if (secdiff[ib] < 1.50) secdiff[ib] = 1.50; }
      }
    }
    int icldlyr[OR_MAXL];
    for (int l = 0; l < L; ++l) {
      icldlyr[l] = 0;
      if (icld == 0) continue;
      if (a->mcica) { for (int g = 0; g < NG; ++g) if (cloudy[g + (size_t)NG * (col + (size_t)N * l)]) icldlyr[l] = 1; }
      else icldlyr[l] = cldfrac[l] >= 1.e-6;
    }
    static lw_mrfac mf;
    if (mr) {
      double cf1[OR_MAXL + 2]; int ic1[OR_MAXL + 2];
      cf1[0] = 0.0; cf1[L + 1] = 0.0; ic1[0] = 0; ic1[L + 1] = 0;
      for (int l = 0; l < L; ++l) { cf1[l + 1] = cldfrac[l]; ic1[l + 1] = icldlyr[l]; }
      lw_mr_factors(L, cf1, ic1, &mf);
    }
    double totuflux[OR_MAXL + 1] = {0}, totdflux[OR_MAXL + 1] = {0}, totuclfl[OR_MAXL + 1] = {0}, totdclfl[OR_MAXL + 1] = {0};
    double dtotu[OR_MAXL + 1] = {0}, dtotuc[OR_MAXL + 1] = {0};
    int igc = 0;
    for (int ib = 0; ib < NB; ++ib) {
      double urad[OR_MAXL + 1] = {0}, drad[OR_MAXL + 1] = {0}, clrurad[OR_MAXL + 1] = {0}, clrdrad[OR_MAXL + 1] = {0};
      double d_urad[OR_MAXL + 1] = {0}, d_clrurad[OR_MAXL + 1] = {0};
      const int cb = ncbands == 1 ? 0 : ncbands == 5 ? icb1[ib] - 1 : ib;
      for (int jg = 0; jg < ngc_[ib]; ++jg, ++igc) {
        double atrans[OR_MAXL], bbugas[OR_MAXL], atot[OR_MAXL], bbutot[OR_MAXL], cf_[OR_MAXL], efcl_[OR_MAXL];
        double radld = 0.0, radclrd = 0.0;
        double cldradd = 0.0, clrradd = 0.0, cldradu = 0.0, clrradu = 0.0, rad = 0.0;   /* rtrnmr: cloudy / clear parts, overlap carry */
        int iclddn = 0;
        for (int lev = L; lev >= 1; --lev) {
          const int l = lev - 1;
          double plfrac = tm->fracs[l][igc], blay = c->planklay[l][ib];
          double dplankup = c->planklev[lev][ib] - blay, dplankdn = c->planklev[lev - 1][ib] - blay;
          double taua = a->tauaer ? a->tauaer[((long)ib * L + l) * N + col] : 0.0;
          double odepth = secdiff[ib] * (tm->taug[l][igc] + taua);
          if (odepth < 0.0) odepth = 0.0;
          double cf = 0.0, odcld = 0.0, efclfrac = 0.0, bbd;
          if (icldlyr[l]) {
            if (a->mcica) {
              if (cloudy[igc + (size_t)NG * (col + (size_t)N * l)]) { cf = 1.0; odcld = secdiff[ib] * taucloud[l][ib]; efclfrac = (1.0 - exp(-odcld)) * cf; }
            } else { cf = cldfrac[l]; odcld = secdiff[cb] * taucloud[l][cb]; efclfrac = (1. - exp(-odcld)) * cf; }
          }
          cf_[l] = cf; efcl_[l] = efclfrac;
          if (icldlyr[l]) {
            iclddn = 1;
            double odtot = odepth + odcld, gassrc, bbdtot;
            if (odtot < 0.06) {
              atrans[l] = odepth - 0.5 * odepth * odepth;
              double odepth_rec = rec_6 * odepth;
              gassrc = plfrac * (blay + dplankdn * odepth_rec) * atrans[l];
              atot[l] = odtot - 0.5 * odtot * odtot;
              double odtot_rec = rec_6 * odtot;
              bbdtot = plfrac * (blay + dplankdn * odtot_rec);
              bbd = plfrac * (blay + dplankdn * odepth_rec);
              bbugas[l] = plfrac * (blay + dplankup * odepth_rec);
              bbutot[l] = plfrac * (blay + dplankup * odtot_rec);
            } else if (odepth <= 0.06) {
              atrans[l] = odepth - 0.5 * odepth * odepth;
              double odepth_rec = rec_6 * odepth;
              gassrc = plfrac * (blay + dplankdn * odepth_rec) * atrans[l];
              odtot = odepth + odcld;
              int ittot = (int)(10000.0 * (odtot / (bpade + odtot)) + 0.5);
              double tfactot = G.tfn_tbl[ittot];
              bbdtot = plfrac * (blay + tfactot * dplankdn);
              bbd = plfrac * (blay + dplankdn * odepth_rec);
              atot[l] = 1.0 - G.exp_tbl[ittot];
              bbugas[l] = plfrac * (blay + dplankup * odepth_rec);
              bbutot[l] = plfrac * (blay + tfactot * dplankup);
            } else {
              int itgas = (int)(10000.0 * (odepth / (bpade + odepth)) + 0.5);
              odepth = G.tau_tbl[itgas];
              atrans[l] = 1.0 - G.exp_tbl[itgas];
              double tfacgas = G.tfn_tbl[itgas];
              gassrc = atrans[l] * plfrac * (blay + tfacgas * dplankdn);
              odtot = odepth + odcld;
              int ittot = (int)(10000.0 * (odtot / (bpade + odtot)) + 0.5);
              double tfactot = G.tfn_tbl[ittot];
              bbdtot = plfrac * (blay + tfactot * dplankdn);
              bbd = plfrac * (blay + tfacgas * dplankdn);
              atot[l] = 1.0 - G.exp_tbl[ittot];
              bbugas[l] = plfrac * (blay + tfacgas * dplankup);
              bbutot[l] = plfrac * (blay + tfactot * dplankup);
            }
            if (mr) {   /* rrtmg_lw_rtrnmr.f90:572-600 */
              if (mf.istcldd[lev] == 1) { cldradd = cldfrac[l] * radld; clrradd = radld - cldradd; rad = 0.; }
              const double ttot = 1. - atot[l], cldsrc = bbdtot * atot[l];
              cldradd = cldradd * ttot + cldfrac[l] * cldsrc;
              clrradd = clrradd * (1. - atrans[l]) + (1. - cldfrac[l]) * gassrc;
              radld = cldradd + clrradd;
              const double radmod = rad * (mf.facclr1d[lev - 1] * (1. - atrans[l]) + mf.faccld1d[lev - 1] * ttot) - mf.faccmb1d[lev - 1] * gassrc + mf.faccmb2d[lev - 1] * cldsrc;
              const double oldcld = cldradd - radmod, oldclr = clrradd + radmod;
              rad = -radmod + mf.facclr2d[lev - 1] * oldclr - mf.faccld2d[lev - 1] * oldcld;
              cldradd = cldradd + rad;
              clrradd = clrradd - rad;
            } else
            radld = radld - radld * (atrans[l] + efclfrac * (1. - atrans[l])) + gassrc + cf * (bbdtot * atot[l] - gassrc);
          } else {
            if (odepth <= 0.06) {
              atrans[l] = odepth - 0.5 * odepth * odepth;
              odepth = rec_6 * odepth;
              bbd = plfrac * (blay + dplankdn * odepth);
              bbugas[l] = plfrac * (blay + dplankup * odepth);
            } else {
              int itr = (int)(10000.0 * (odepth / (bpade + odepth)) + 0.5);
              atrans[l] = 1.0 - G.exp_tbl[itr];
              double tausfac = G.tfn_tbl[itr];
              bbd = plfrac * (blay + tausfac * dplankdn);
              bbugas[l] = plfrac * (blay + tausfac * dplankup);
            }
            radld = radld + (bbd - radld) * atrans[l];
          }
          drad[lev - 1] = drad[lev - 1] + radld;
          if (iclddn) { radclrd = radclrd + (bbd - radclrd) * atrans[l]; clrdrad[lev - 1] = clrdrad[lev - 1] + radclrd; }
          else { radclrd = radld; clrdrad[lev - 1] = clrdrad[lev - 1] + radclrd; }
        }
        double rad0 = tm->fracs[0][igc] * c->plankbnd[ib], reflect = 1.0 - c->semiss[ib];
        double radlu = rad0 + reflect * radld, radclru = rad0 + reflect * radclrd;
        urad[0] = urad[0] + radlu; clrurad[0] = clrurad[0] + radclru;
        double d_radlu = 0, d_radclru = 0;
        if (a->idrv) { d_radlu = tm->fracs[0][igc] * c->dplankbnd_dt[ib]; d_radclru = d_radlu; d_urad[0] += d_radlu; d_clrurad[0] += d_radclru; }
        for (int lev = 1; lev <= L; ++lev) {
          const int l = lev - 1;
          if (icldlyr[l]) {
            double gassrc = bbugas[l] * atrans[l];
            if (mr) {   /* rrtmg_lw_rtrnmr.f90:657-682 */
              if (mf.istcld[lev] == 1) { cldradu = cldfrac[l] * radlu; clrradu = radlu - cldradu; rad = 0.; }
              const double ttot = 1. - atot[l], cldsrc = bbutot[l] * atot[l];
              cldradu = cldradu * ttot + cldfrac[l] * cldsrc;
              clrradu = clrradu * (1.0 - atrans[l]) + (1. - cldfrac[l]) * gassrc;
              radlu = cldradu + clrradu;
              const double radmod = rad * (mf.facclr1[lev + 1] * (1.0 - atrans[l]) + mf.faccld1[lev + 1] * ttot) - mf.faccmb1[lev + 1] * gassrc + mf.faccmb2[lev + 1] * cldsrc;
              const double oldcld = cldradu - radmod, oldclr = clrradu + radmod;
              rad = -radmod + mf.facclr2[lev + 1] * oldclr - mf.faccld2[lev + 1] * oldcld;
              cldradu = cldradu + rad;
              clrradu = clrradu - rad;
            } else
            radlu = radlu - radlu * (atrans[l] + efcl_[l] * (1.0 - atrans[l])) + gassrc + cf_[l] * (bbutot[l] * atot[l] - gassrc);
            if (a->idrv) d_radlu = d_radlu * cf_[l] * (1.0 - atot[l]) + d_radlu * (1.0 - cf_[l]) * (1.0 - atrans[l]);
          } else {
            radlu = radlu + (bbugas[l] - radlu) * atrans[l];
            if (a->idrv) d_radlu = d_radlu * (1.0 - atrans[l]);
          }
          urad[lev] = urad[lev] + radlu;
          if (iclddn) { radclru = radclru + (bbugas[l] - radclru) * atrans[l]; if (a->idrv) d_radclru = d_radclru * (1.0 - atrans[l]); }
          else { radclru = radlu; if (a->idrv) d_radclru = d_radlu; }
          clrurad[lev] = clrurad[lev] + radclru;
          if (a->idrv) { d_urad[lev] += d_radlu; d_clrurad[lev] += d_radclru; }
        }
      }
      for (int lev = L; lev >= 0; --lev) {
        totuflux[lev] = totuflux[lev] + (urad[lev] * wtdiff) * delwave[ib]; totdflux[lev] = totdflux[lev] + (drad[lev] * wtdiff) * delwave[ib];
        totuclfl[lev] = totuclfl[lev] + (clrurad[lev] * wtdiff) * delwave[ib]; totdclfl[lev] = totdclfl[lev] + (clrdrad[lev] * wtdiff) * delwave[ib];
        if (a->idrv) { dtotu[lev] = dtotu[lev] + (d_urad[lev] * wtdiff) * delwave[ib] * fluxfac; dtotuc[lev] = dtotuc[lev] + (d_clrurad[lev] * wtdiff) * delwave[ib] * fluxfac; }
      }
    }
    double fnetp = 0, fnetcp = 0;
    for (int lev = 0; lev <= L; ++lev) {
      long o = (long)lev * N + col;
      double uf = totuflux[lev] * fluxfac, df = totdflux[lev] * fluxfac, ucf = totuclfl[lev] * fluxfac, dcf = totdclfl[lev] * fluxfac;
      a->uflx[o] = uf; a->dflx[o] = df; a->uflxc[o] = ucf; a->dflxc[o] = dcf;
      if (a->idrv) { a->duflx_dt[o] = dtotu[lev]; a->duflxc_dt[o] = dtotuc[lev]; }
      double fnet = uf - df, fnetc = ucf - dcf;
      if (lev > 0) {
        long ol = (long)(lev - 1) * N + col;
        a->hr[ol] = G.heatfac * (fnetp - fnet) / (c->pz[lev - 1] - c->pz[lev]);
        a->hrc[ol] = G.heatfac * (fnetcp - fnetc) / (c->pz[lev - 1] - c->pz[lev]);
      }
      fnetp = fnet; fnetcp = fnetc;
    }
  }
  free(c); free(tm); free(cloudy);
  return err;
}
