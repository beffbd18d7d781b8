/* TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference RRTMG SHORTWAVE column algorithm.
 * Plain C, one column at a time, array-per-column as in the Fortran; each function cites what it follows under
 * /root/reference/climt/_lib/rrtmg_sw/.  Build: oracle/Makefile (gcc -O2 -ffp-contract=off).
 * Pinned against reference-Fortran outputs (tests/golden/ref_*.npz; oracle/_ref when present) and the reference's
 * own golden caches through tests/test_oracle.py. */
#include "oracle_common.h"

#define NB 14
#define NG 112
static const int ngc_[NB] = {6, 12, 8, 8, 10, 10, 2, 10, 8, 6, 6, 8, 6, 12};
static const int nspa_[NB] = {9, 9, 9, 9, 1, 9, 9, 1, 9, 1, 0, 1, 9, 1};
static const int nspb_[NB] = {1, 5, 1, 1, 1, 5, 1, 0, 1, 0, 0, 1, 5, 1};

typedef struct {
  or_store st;
  double exp_tbl[10001], tau_tbl[10001], tfn_tbl[10001];
  double heatfac, grav, avogad, pi;
  int ready;
} sw_oracle;
static sw_oracle G;

int sw_oracle_init(const char *blob, double cpdair, double grav, double avogad, double secdy, double pi) {
  if (G.ready) return 0;
  int rc = or_load_blob(&G.st, blob);
  if (rc) return rc;
  or_reduce(&G.st, "sw", NB, 16);
  or_lookup_tables(0, G.exp_tbl, G.tau_tbl, G.tfn_tbl);
  G.heatfac = grav * secdy / (cpdair * 1.e2); /* swdatinit, rrtmg_sw_init.f90:258 */
  G.grav = grav; G.avogad = avogad; G.pi = pi;
  G.ready = 1;
  return 0;
}
long sw_oracle_table(const char *name, double *out, long cap) {
  or_entry *e = or_find(&G.st, name);
  if (!strcmp(name, "sw/tbl/exp_tbl")) { if (out) memcpy(out, G.exp_tbl, 10001 * 8); return 10001; }
  if (!e || e->dtype) return -1;
  if (out) { if (cap < e->n) return -2; memcpy(out, e->f, (size_t)e->n * 8); }
  return e->n;
}
static double *T(int band, const char *leaf) {
  char nm[64];
  snprintf(nm, sizeof nm, "sw/kg%02d/%s", band, leaf);
  return or_f(&G.st, nm);
}

typedef struct {
  int nlay, laytrop;
  double pavel[OR_MAXL], tavel[OR_MAXL], pz[OR_MAXL + 1], pdp[OR_MAXL], coldry[OR_MAXL];
  double colh2o[OR_MAXL], colco2[OR_MAXL], colo3[OR_MAXL], colch4[OR_MAXL], colo2[OR_MAXL], colmol[OR_MAXL];
  double fac00[OR_MAXL], fac01[OR_MAXL], fac10[OR_MAXL], fac11[OR_MAXL];
  double selffac[OR_MAXL], selffrac[OR_MAXL], forfac[OR_MAXL], forfrac[OR_MAXL];
  int jp[OR_MAXL + 1], jt[OR_MAXL], jt1[OR_MAXL], indself[OR_MAXL], indfor[OR_MAXL];
} sw_col;

/* inatm_sw (rrtmg_sw_rad.nomcica.f90:1441-1465) + setcoef_sw (rrtmg_sw_setcoef.f90:137-303) */
static void sw_setcoef(sw_col *c, const double *h2o, const double *co2, const double *o3, const double *ch4, const double *o2) {
  const double amd = 28.9660, amw = 18.0160, stpfac = 296.0 / 1013.0;
  const double *preflog = or_f(&G.st, "sw/ref/preflog"), *tref = or_f(&G.st, "sw/ref/tref");
  c->laytrop = 0;
  for (int l = 0; l < c->nlay; ++l) {
    c->pdp[l] = c->pz[l] - c->pz[l + 1];
    double amm = (1.0 - h2o[l]) * amd + h2o[l] * amw;
    c->coldry[l] = (c->pz[l] - c->pz[l + 1]) * 1.e3 * G.avogad / (1.e2 * G.grav * amm * (1.0 + h2o[l]));
    double wkl1 = c->coldry[l] * h2o[l], wkl2 = c->coldry[l] * co2[l], wkl3 = c->coldry[l] * o3[l];
    double wkl6 = c->coldry[l] * ch4[l], wkl7 = c->coldry[l] * o2[l];
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
    double water = wkl1 / c->coldry[l];
    double scalefac = c->pavel[l] * stpfac / c->tavel[l];
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
      c->indfor[l] = 3;
      c->forfrac[l] = factor - 1.0;
      c->selffac[l] = 0.0; c->selffrac[l] = 0.0; c->indself[l] = 0;
    }
    c->colh2o[l] = 1.e-20 * wkl1; c->colco2[l] = 1.e-20 * wkl2; c->colo3[l] = 1.e-20 * wkl3;
    c->colch4[l] = 1.e-20 * wkl6; c->colo2[l] = 1.e-20 * wkl7;
    c->colmol[l] = 1.e-20 * c->coldry[l] + c->colh2o[l];
    if (c->colco2[l] == 0.0) c->colco2[l] = 1.e-32 * c->coldry[l];
    if (c->colch4[l] == 0.0) c->colch4[l] = 1.e-32 * c->coldry[l];
    if (c->colo2[l] == 0.0) c->colo2[l] = 1.e-32 * c->coldry[l];
    double compfp = 1.0 - fp;
    c->fac10[l] = compfp * ft; c->fac00[l] = compfp * (1.0 - ft);
    c->fac11[l] = fp * ft1; c->fac01[l] = fp * (1.0 - ft1);
  }
  c->jp[c->nlay] = 0;
}

/* ---- taumol_sw (rrtmg_sw_taumol.f90:50-1790): taug(lay, ig), taur(lay, ig), ssi/sfluxzen(ig) ------------------ */
typedef struct { double taug[OR_MAXL][NG], taur[OR_MAXL][NG], src[NG]; } sw_tau;

static double selfterm(const sw_col *c, int l, const double *selfref) {
  return c->selffac[l] * (selfref[c->indself[l] - 1] + c->selffrac[l] * (selfref[c->indself[l]] - selfref[c->indself[l] - 1]));
}
static double forint(const sw_col *c, int l, const double *forref) {
  return forref[c->indfor[l] - 1] + c->forfrac[l] * (forref[c->indfor[l]] - forref[c->indfor[l] - 1]);
}
/* source selection: (svar_f*facbrght + svar_s*snsptdrk + svar_i*irradnce) or sfluxref, optionally mixture-interpolated */
typedef struct { int isolvar; double svar_f, svar_s, svar_i, svar_b[NB]; } sw_solar;
static double sw_source(const sw_solar *so, int b, int band, int ig, int ng, int binary, int js, double fs) {
  const double *sf = T(band, "sfluxref"), *fb = T(band, "facbrght"), *sn = T(band, "snsptdrk"), *ir = T(band, "irradnce");
#define SRC(a) (binary ? ((a)[ig + ng * (js - 1)] + fs * ((a)[ig + ng * js] - (a)[ig + ng * (js - 1)])) : (a)[ig])
  if (so->isolvar < 0) {
    double s = SRC(sf);
    if (band == 27) s = (50.15 / 48.37) * sf[ig];
    return s;
  }
  if (so->isolvar == 3) return so->svar_b[b] * SRC(fb) + so->svar_b[b] * SRC(sn) + so->svar_b[b] * SRC(ir);
  return so->svar_f * SRC(fb) + so->svar_s * SRC(sn) + so->svar_i * SRC(ir);
#undef SRC
}

static void sw_taumol(const sw_col *c, const sw_solar *so, sw_tau *o) {
  const double oneminus = 1.0 - 1.e-6;
  const int L = c->nlay, laytrop = c->laytrop;
  /* per band: key species of the lower/upper atmosphere and strrat */
  static const int layreffr[NB] = {18, 30, 6, 3, 3, 8, 2, 6, 1, 2, 0, 32, 58, 49};
  static const int upper_src[NB] = {1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1};
  static const double strrat[NB] = {252.131, 0.364641, 38.9589, 5.49281, 0, 0.0045321, 0.022708, 0, 0.124692, 0, 0, 0, 6.67029e-07, 0};
  int gs = 0;
  for (int b = 0; b < NB; ++b) {
    const int band = 16 + b, ng = ngc_[b];
    const double *absa = nspa_[b] ? T(band, "absa") : NULL, *absb = nspb_[b] ? T(band, "absb") : NULL;
    const double *selfref = (band <= 24 || band == 29) ? T(band, "selfref") : NULL;
    const double *forref = (band <= 24 || band == 29) ? T(band, "forref") : NULL;
    const int nfor = (band >= 22 && band <= 24) || band == 16 || band == 18 || band == 19 ? 3 : 4;
    const int nA = 65 * nspa_[b], nB2 = 235 * nspb_[b];
    const double *rayl = band == 24 ? NULL : T(band, "rayl");
    const int rayl_per_g = (band == 23 || band == 25 || band == 26 || band == 27);
    int laysolfr = upper_src[b] ? L : laytrop;
    for (int lay = 1; lay <= L; ++lay) {
      const int l = lay - 1, lower = lay <= laytrop;
      /* second key species of the binary bands */
      double colx = c->colh2o[l], coly = 0.0, sr = strrat[b];
      if (band == 16 || band == 18) coly = c->colch4[l];
      if (band == 17 || band == 19 || band == 21) coly = c->colco2[l];
      if (band == 22) { coly = c->colo2[l]; sr = 1.6 * strrat[b]; }
      if (band == 24) coly = c->colo2[l];
      if (band == 28) { colx = c->colo3[l]; coly = c->colo2[l]; }
      const int binary = lower ? nspa_[b] == 9 : nspb_[b] == 5;
      double speccomb = 0, fs = 0;
      int js = 1;
      if (binary) {
        speccomb = colx + sr * coly;
        double specparm = colx / speccomb;
        if (specparm >= oneminus) specparm = oneminus;
        double specmult = (lower ? 8.0 : 4.0) * specparm;
        js = 1 + (int)specmult;
        fs = fmod(specmult, 1.0);
      }
      /* laysolfr bookkeeping exactly in loop order */
      if (lower && !upper_src[b] && band != 26) {
        if (c->jp[l] < layreffr[b] && c->jp[l + 1] >= layreffr[b]) laysolfr = (lay + 1 < laytrop) ? lay + 1 : laytrop;
      }
      if (!lower && upper_src[b]) {
        int jpm = lay >= 2 ? c->jp[l - 1] : 0;
        if (jpm < layreffr[b] && c->jp[l] >= layreffr[b]) laysolfr = lay;
      }
      int ind0, ind1;
      if (lower) { ind0 = ((c->jp[l] - 1) * 5 + (c->jt[l] - 1)) * nspa_[b] + js - 1; ind1 = (c->jp[l] * 5 + (c->jt1[l] - 1)) * nspa_[b] + js - 1; }
      else { ind0 = ((c->jp[l] - 13) * 5 + (c->jt[l] - 1)) * nspb_[b] + js - 1; ind1 = ((c->jp[l] - 12) * 5 + (c->jt1[l] - 1)) * nspb_[b] + js - 1; }
      for (int ig = 0; ig < ng; ++ig) {
        const double *ka = absa ? absa + (long)ig * nA : NULL, *kb = absb ? absb + (long)ig * nB2 : NULL;
        const double *sref = selfref ? selfref + ig * 10 : NULL, *fref = forref ? forref + ig * nfor : NULL;
        double tg = 0.0, tauray;
        if (band == 24) {
          if (lower) { const double *ra = T(24, "rayla"); tauray = c->colmol[l] * (ra[ig + ng * (js - 1)] + fs * (ra[ig + ng * js] - ra[ig + ng * (js - 1)])); }
          else tauray = c->colmol[l] * T(24, "raylb")[ig];
        } else tauray = c->colmol[l] * (rayl_per_g ? rayl[ig] : rayl[0]);
        if (binary) {
          const double *k = lower ? ka : kb;
          const int dT = lower ? 9 : 5;
          double fac000 = (1.0 - fs) * c->fac00[l], fac010 = (1.0 - fs) * c->fac10[l], fac100 = fs * c->fac00[l], fac110 = fs * c->fac10[l];
          double fac001 = (1.0 - fs) * c->fac01[l], fac011 = (1.0 - fs) * c->fac11[l], fac101 = fs * c->fac01[l], fac111 = fs * c->fac11[l];
          double major = speccomb * (fac000 * k[ind0] + fac100 * k[ind0 + 1] + fac010 * k[ind0 + dT] + fac110 * k[ind0 + dT + 1] +
                                     fac001 * k[ind1] + fac101 * k[ind1 + 1] + fac011 * k[ind1 + dT] + fac111 * k[ind1 + dT + 1]);
          if (band == 28) tg = major;
          else if (lower && band == 24) tg = major + c->colo3[l] * T(24, "abso3a")[ig] + c->colh2o[l] * (selfterm(c, l, sref) + c->forfac[l] * forint(c, l, fref));
          else if (lower) { tg = major + c->colh2o[l] * (selfterm(c, l, sref) + c->forfac[l] * forint(c, l, fref)); if (band == 22) tg = tg + 4.35e-4 * c->colo2[l] / (350.0 * 2.0); }
          else tg = major + c->colh2o[l] * c->forfac[l] * forint(c, l, fref);
        } else if (lower) {
          double m4 = nspa_[b] ? c->fac00[l] * ka[ind0] + c->fac10[l] * ka[ind0 + 1] + c->fac01[l] * ka[ind1] + c->fac11[l] * ka[ind1 + 1] : 0.0;
          if (band == 20) tg = c->colh2o[l] * (m4 + selfterm(c, l, sref) + c->forfac[l] * forint(c, l, fref)) + c->colch4[l] * T(20, "absch4")[ig];
          else if (band == 29) tg = c->colh2o[l] * (m4 + selfterm(c, l, sref) + c->forfac[l] * forint(c, l, fref)) + c->colco2[l] * T(29, "absco2")[ig];
          else if (band == 23) tg = c->colh2o[l] * (1.029 * m4 + selfterm(c, l, sref) + c->forfac[l] * forint(c, l, fref));
          else if (band == 25) tg = c->colh2o[l] * m4 + c->colo3[l] * T(25, "abso3a")[ig];
          else if (band == 27) tg = c->colo3[l] * m4;
          else tg = 0.0;
        } else {
          double m4 = nspb_[b] ? c->fac00[l] * kb[ind0] + c->fac10[l] * kb[ind0 + 1] + c->fac01[l] * kb[ind1] + c->fac11[l] * kb[ind1 + 1] : 0.0;
          if (band == 16 || band == 18) tg = c->colch4[l] * m4;
          else if (band == 19) tg = c->colco2[l] * m4;
          else if (band == 20) tg = c->colh2o[l] * (c->fac00[l] * kb[ind0] + c->fac10[l] * kb[ind0 + 1] + c->fac01[l] * kb[ind1] + c->fac11[l] * kb[ind1 + 1] +
                                                   c->forfac[l] * forint(c, l, fref)) + c->colch4[l] * T(20, "absch4")[ig];
          else if (band == 22) tg = c->colo2[l] * 1.6 * m4 + 4.35e-4 * c->colo2[l] / (350.0 * 2.0);
          else if (band == 24) tg = c->colo2[l] * m4 + c->colo3[l] * T(24, "abso3b")[ig];
          else if (band == 25) tg = c->colo3[l] * T(25, "abso3b")[ig];
          else if (band == 27) tg = c->colo3[l] * m4;
          else if (band == 29) tg = c->colco2[l] * m4 + c->colh2o[l] * T(29, "absh2o")[ig];
          else tg = 0.0;
        }
        o->taug[l][gs + ig] = tg;
        o->taur[l][gs + ig] = tauray;
        if (lay == laysolfr && ((lower && !upper_src[b]) || (!lower && upper_src[b]))) {
          const int bin_src = (band == 17 || band == 18 || band == 19 || band == 21 || band == 22 || band == 24 || band == 28);
          o->src[gs + ig] = sw_source(so, b, band, ig, ng, bin_src, js, fs);
        }
      }
    }
    gs += ng;
  }
}

/* reftra_sw (rrtmg_sw_reftra.f90:148-316), kmodts = 2 */
static double tbl(double x) {
  double tblind = x / (1.0 / 0.278 + x);
  int itind = (int)(10000.0 * tblind + 0.5);
  return G.exp_tbl[itind];
}
static void sw_reftra(int nlay, const int *lrtchk, const double *pgg, double prmuz, const double *ptau, const double *pw, double *pref,
                      double *prefd, double *ptra, double *ptrad) {
  const double eps = 1.e-08, zwcrit = 0.9999995, od_lo = 0.06;
  for (int jk = 0; jk < nlay; ++jk) {
    if (!lrtchk[jk]) { pref[jk] = 0.0; ptra[jk] = 1.0; prefd[jk] = 0.0; ptrad[jk] = 1.0; continue; }
    double zto1 = ptau[jk], zw = pw[jk], zg = pgg[jk], zg3 = 3.0 * zg;
    double zgamma1 = (8.0 - zw * (5.0 + zg3)) * 0.25, zgamma2 = 3.0 * (zw * (1.0 - zg)) * 0.25, zgamma3 = (2.0 - zg3 * prmuz) * 0.25;
    double zgamma4 = 1.0 - zgamma3;
    double q = zg / (1.0 - zg);
    double zwo = zw / (1.0 - (1.0 - zw) * (q * q));
    if (zwo >= zwcrit) {
      double za = zgamma1 * prmuz, za1 = za - zgamma3, zgt = zgamma1 * zto1;
      double ze1 = zto1 / prmuz; if (ze1 > 500.0) ze1 = 500.0;
      double ze2 = ze1 <= od_lo ? 1.0 - ze1 + 0.5 * ze1 * ze1 : tbl(ze1);
      pref[jk] = (zgt - za1 * (1.0 - ze2)) / (1.0 + zgt);
      ptra[jk] = 1.0 - pref[jk];
      prefd[jk] = zgt / (1.0 + zgt);
      ptrad[jk] = 1.0 - prefd[jk];
      if (ze2 == 1.0) { pref[jk] = 0.0; ptra[jk] = 1.0; prefd[jk] = 0.0; ptrad[jk] = 1.0; }
    } else {
      double za1 = zgamma1 * zgamma4 + zgamma2 * zgamma3, za2 = zgamma1 * zgamma3 + zgamma2 * zgamma4;
      double zrk = sqrt(zgamma1 * zgamma1 - zgamma2 * zgamma2), zrp = zrk * prmuz, zrp1 = 1.0 + zrp, zrm1 = 1.0 - zrp, zrk2 = 2.0 * zrk;
      double zrpp = 1.0 - zrp * zrp, zrkg = zrk + zgamma1;
      double zr1 = zrm1 * (za2 + zrk * zgamma3), zr2 = zrp1 * (za2 - zrk * zgamma3), zr3 = zrk2 * (zgamma3 - za2 * prmuz);
      double zr4 = zrpp * zrkg, zr5 = zrpp * (zrk - zgamma1);
      double zt1 = zrp1 * (za1 + zrk * zgamma4), zt2 = zrm1 * (za1 - zrk * zgamma4), zt3 = zrk2 * (zgamma4 + za1 * prmuz);
      double zbeta = (zgamma1 - zrk) / zrkg;
      double ze1 = zrk * zto1; if (ze1 > 500.0) ze1 = 500.0;
      double ze2 = zto1 / prmuz; if (ze2 > 500.0) ze2 = 500.0;
      double zem1 = ze1 <= od_lo ? 1.0 - ze1 + 0.5 * ze1 * ze1 : tbl(ze1), zep1 = 1.0 / zem1;
      double zem2 = ze2 <= od_lo ? 1.0 - ze2 + 0.5 * ze2 * ze2 : tbl(ze2), zep2 = 1.0 / zem2;
      double zdenr = zr4 * zep1 + zr5 * zem1, zdent = zr4 * zep1 + zr5 * zem1;
      if (zdenr >= -eps && zdenr <= eps) { pref[jk] = eps; ptra[jk] = zem2; }
      else {
        pref[jk] = zw * (zr1 * zep1 - zr2 * zem1 - zr3 * zem2) / zdenr;
        ptra[jk] = zem2 - zem2 * zw * (zt1 * zep1 - zt2 * zem1 - zt3 * zep2) / zdent;
      }
      double zemm = zem1 * zem1, zdend = 1.0 / ((1.0 - zbeta * zemm) * zrkg);
      prefd[jk] = zgamma2 * (1.0 - zemm) * zdend;
      ptrad[jk] = zrk2 * zem1 * zdend;
    }
  }
}

/* vrtqdr_sw (rrtmg_sw_vrtqdr.f90:114-169); arrays 0-based, index klev = surface */
static void sw_vrtqdr(int klev, const double *pref, const double *prefd, const double *ptra, const double *ptrad, const double *pdbt,
                      double *prdnd, double *prup, double *prupd, const double *ptdbt, double *pfd, double *pfu) {
  double ztdn[OR_MAXL + 1];
  double zreflect = 1.0 / (1.0 - prefd[klev] * prefd[klev - 1]);
  prup[klev - 1] = pref[klev - 1] + (ptrad[klev - 1] * ((ptra[klev - 1] - pdbt[klev - 1]) * prefd[klev] + pdbt[klev - 1] * pref[klev])) * zreflect;
  prupd[klev - 1] = prefd[klev - 1] + ptrad[klev - 1] * ptrad[klev - 1] * prefd[klev] * zreflect;
  for (int jk = 1; jk <= klev - 1; ++jk) {
    int ikp = klev - jk, ikx = ikp - 1;
    zreflect = 1.0 / (1.0 - prupd[ikp] * prefd[ikx]);
    prup[ikx] = pref[ikx] + (ptrad[ikx] * ((ptra[ikx] - pdbt[ikx]) * prupd[ikp] + pdbt[ikx] * prup[ikp])) * zreflect;
    prupd[ikx] = prefd[ikx] + ptrad[ikx] * ptrad[ikx] * prupd[ikp] * zreflect;
  }
  ztdn[0] = 1.0; prdnd[0] = 0.0; ztdn[1] = ptra[0]; prdnd[1] = prefd[0];
  for (int jk = 1; jk < klev; ++jk) {
    int ikp = jk + 1;
    zreflect = 1.0 / (1.0 - prefd[jk] * prdnd[jk]);
    ztdn[ikp] = ptdbt[jk] * ptra[jk] + (ptrad[jk] * ((ztdn[jk] - ptdbt[jk]) + ptdbt[jk] * pref[jk] * prdnd[jk])) * zreflect;
    prdnd[ikp] = prefd[jk] + ptrad[jk] * ptrad[jk] * prdnd[jk] * zreflect;
  }
  for (int jk = 0; jk <= klev; ++jk) {
    zreflect = 1.0 / (1.0 - prdnd[jk] * prupd[jk]);
    pfu[jk] = (ptdbt[jk] * prup[jk] + (ztdn[jk] - ptdbt[jk]) * prupd[jk]) * zreflect;
    pfd[jk] = ptdbt[jk] + (ztdn[jk] - ptdbt[jk] + ptdbt[jk] * prup[jk] * prdnd[jk]) * zreflect;
  }
}

static double dbt_of(double tau, double prmu0) {
  double ze1 = tau / prmu0;
  return ze1 <= 0.06 ? 1.0 - ze1 + 0.5 * ze1 * ze1 : tbl(ze1);
}

/* cloud optics for one layer and band: cldprop_sw (rrtmg_sw_cldprop.f90:139-360) == cldprmc_sw per g-point.
 * returns 0 or an error code (the reference's `stop`s) */
static int sw_cldopt(int b, int inflag, int iceflag, int liqflag, double ciwp, double clwp, double radice, double radliq, double tauc,
                     double ssac, double asmc, double fsfc, double *tau, double *ssa, double *asy) {
  const double eps = 1.e-06, cldmin = 1.e-20;
  or_store *s = &G.st;
  if (inflag == 0) {
    double ffp = fsfc, ffp1 = 1.0 - ffp, ffpssa = 1.0 - ffp * ssac;
    *ssa = ffp1 * ssac / ffpssa; *tau = ffpssa * tauc; *asy = (asmc - ffp) / ffp1;
    return 0;
  }
  if (inflag != 2) return 0;
  double extcoice = 0, ssacoice = 0, gice = 0, forwice = 0, extcoliq = 0, ssacoliq = 0, gliq = 0, forwliq = 0;
  if (ciwp == 0.0) {
  } else if (iceflag == 1) {
    if (radice < 13.0 || radice > 130.) return 11;
    double wn2 = or_f(s, "sw/wvn/wavenum2")[b];
    int icx = 5;
    if (wn2 > 1.43e04) icx = 1; else if (wn2 > 7.7e03) icx = 2; else if (wn2 > 5.3e03) icx = 3; else if (wn2 > 4.0e03) icx = 4;
    extcoice = or_f(s, "sw/cld/abari")[icx - 1] + or_f(s, "sw/cld/bbari")[icx - 1] / radice;
    ssacoice = 1.0 - or_f(s, "sw/cld/cbari")[icx - 1] - or_f(s, "sw/cld/dbari")[icx - 1] * radice;
    gice = or_f(s, "sw/cld/ebari")[icx - 1] + or_f(s, "sw/cld/fbari")[icx - 1] * radice;
    if (gice >= 1.0) gice = 1.0 - eps;
    forwice = gice * gice;
  } else if (iceflag == 2 || iceflag == 3) {
    int nr = iceflag == 2 ? 43 : 46;
    if (radice < 5.0 || radice > (iceflag == 2 ? 131.0 : 140.0)) return 11;
    double factor = (radice - 2.0) / 3.0;
    int index = (int)factor;
    if (index == nr) index = nr - 1;
    double fint = factor - (double)index;
    const double *e = or_f(s, iceflag == 2 ? "sw/cld/extice2" : "sw/cld/extice3") + (index - 1) + nr * b;
    const double *w = or_f(s, iceflag == 2 ? "sw/cld/ssaice2" : "sw/cld/ssaice3") + (index - 1) + nr * b;
    const double *g = or_f(s, iceflag == 2 ? "sw/cld/asyice2" : "sw/cld/asyice3") + (index - 1) + nr * b;
    extcoice = e[0] + fint * (e[1] - e[0]); ssacoice = w[0] + fint * (w[1] - w[0]); gice = g[0] + fint * (g[1] - g[0]);
    if (iceflag == 2) forwice = gice * gice;
    else {
      const double *f = or_f(s, "sw/cld/fdlice3") + (index - 1) + 46 * b;
      double fdelta = f[0] + fint * (f[1] - f[0]);
      if (fdelta < 0.0 || fdelta > 1.0) return 13;
      forwice = fdelta + 0.5 / ssacoice;
      if (forwice > gice) forwice = gice;
    }
  } else return 20;
  if (ciwp != 0.0 && (extcoice < 0.0 || ssacoice > 1.0 || ssacoice < 0.0 || gice > 1.0 || gice < 0.0)) return 13;
  if (clwp == 0.0) {
  } else if (liqflag == 1) {
    if (radliq < 2.5 || radliq > 60.) return 12;
    int index = (int)(radliq - 1.5);
    if (index == 0) index = 1;
    if (index == 58) index = 57;
    double fint = radliq - 1.5 - (double)index;
    const double *e = or_f(s, "sw/cld/extliq1") + (index - 1) + 58 * b, *w = or_f(s, "sw/cld/ssaliq1") + (index - 1) + 58 * b;
    const double *g = or_f(s, "sw/cld/asyliq1") + (index - 1) + 58 * b;
    extcoliq = e[0] + fint * (e[1] - e[0]);
    ssacoliq = w[0] + fint * (w[1] - w[0]);
    if (fint < 0. && ssacoliq > 1.) ssacoliq = w[0];
    gliq = g[0] + fint * (g[1] - g[0]);
    forwliq = gliq * gliq;
    if (extcoliq < 0.0 || ssacoliq > 1.0 || ssacoliq < 0.0 || gliq > 1.0 || gliq < 0.0) return 13;
  } else return 20;
  double tauliqorig = clwp * extcoliq, tauiceorig = ciwp * extcoice;
  double ssaliq = ssacoliq * (1.0 - forwliq) / (1.0 - forwliq * ssacoliq), tauliq = (1.0 - forwliq * ssacoliq) * tauliqorig;
  double ssaice = ssacoice * (1.0 - forwice) / (1.0 - forwice * ssacoice), tauice = (1.0 - forwice * ssacoice) * tauiceorig;
  double scatliq = ssaliq * tauliq, scatice = ssaice * tauice;
  *tau = tauliq + tauice;
  if (*tau == 0.0) *tau = cldmin;
  if (scatice == 0.0) scatice = cldmin;
  *ssa = (scatliq + scatice) / *tau;
  if (iceflag == 3) *asy = (1.0 / (scatliq + scatice)) * (scatliq * (gliq - forwliq) / (1.0 - forwliq) + scatice * ((gice - forwice) / (1.0 - forwice)));
  else *asy = (scatliq * (gliq - forwliq) / (1.0 - forwliq) + scatice * (gice - forwice) / (1.0 - forwice)) / (scatliq + scatice);
  return 0;
}

typedef struct {
  int ncol, nlay, mcica, icld, iaer, inflag, iceflag, liqflag, dyofyr, isolvar, irng, permuteseed;
  double adjes, scon;
  const double *bndsolvar, *play, *plev, *tlay, *h2o, *o3, *co2, *ch4, *n2o, *o2, *asdir, *asdif, *aldir, *aldif, *coszen;
  const double *cldfr, *taucld, *ssacld, *asmcld, *fsfcld, *cicewp, *cliqwp, *reice, *reliq, *tauaer, *ssaaer, *asmaer, *ecaer;
  double *swuflx, *swdflx, *swhr, *swuflxc, *swdflxc, *swhrc;
} sw_args;

/* rrtmg_sw driver (rrtmg_sw_rad.nomcica.f90:587-816 / rrtmg_sw_rad.f90) + spcvrt_sw / spcvmc_sw
 * (rrtmg_sw_spcvrt.f90:290-660).  Returns 0 or the code of the reference's `stop`. */
int sw_oracle_fluxes(const sw_args *a) {
  const int N = a->ncol, L = a->nlay;
  int icld = a->icld, iaer = a->iaer;
  if (icld < 0 || icld > 3) icld = 2;
  if (iaer != 0 && iaer != 6 && iaer != 10) iaer = 0;
  /* inatm_sw scalar part (rrtmg_sw_rad.nomcica.f90:1222-1428) */
  sw_solar so;
  double adjflux[NB], solvar[NB];
  {
    const double rrsw_scon = (double)1.36822e+03f, Iint = 1360.37, Fint = 0.996047, Sint = -0.511590;
    for (int b = 0; b < NB; ++b) { solvar[b] = 1.0; so.svar_b[b] = 1.0; }
    so.isolvar = a->isolvar; so.svar_f = so.svar_s = so.svar_i = 1.0;
    double adjflx = a->adjes;
    if (a->dyofyr > 0) {
      double gamma = 2.0 * G.pi * (a->dyofyr - 1) / 365.0;
      adjflx = 1.000110 + .034221 * cos(gamma) + .001289 * sin(gamma) + .000719 * cos(2.0 * gamma) + .000077 * sin(2.0 * gamma);
    }
    if (a->scon == 0.0) {
      if (a->isolvar == -1 && a->bndsolvar) for (int b = 0; b < NB; ++b) solvar[b] = a->bndsolvar[b];
      if (a->isolvar == 3) for (int b = 0; b < NB; ++b) { solvar[b] = a->bndsolvar ? a->bndsolvar[b] : 1.0; so.svar_b[b] = solvar[b]; }
    } else if (a->scon > 0.0) {
      if (a->isolvar == -1) for (int b = 0; b < NB; ++b) solvar[b] = a->bndsolvar ? a->bndsolvar[b] * a->scon / rrsw_scon : a->scon / rrsw_scon;
      if (a->isolvar == 0) { double r = a->scon / (Fint + Sint + Iint); so.svar_f = so.svar_s = so.svar_i = r; }
      if (a->isolvar == 3) { double cc = Fint + Sint + Iint; for (int b = 0; b < NB; ++b) { solvar[b] = a->bndsolvar ? a->bndsolvar[b] * a->scon / cc : a->scon / cc; so.svar_b[b] = solvar[b]; } }
    }
    for (int b = 0; b < NB; ++b) adjflux[b] = a->isolvar < 0 ? adjflx * solvar[b] : adjflx;
  }
  unsigned char *cloudy = NULL;
  if (a->mcica && icld >= 1) {
    cloudy = (unsigned char *)malloc((size_t)NG * N * L);
    int rc = or_subcol_mask(N, L, NG, icld, a->irng, a->permuteseed, a->play, a->cldfr, cloudy);
    if (rc) { free(cloudy); return rc; }
  }
  sw_col *c = (sw_col *)malloc(sizeof(sw_col));
  sw_tau *tm = (sw_tau *)malloc(sizeof(sw_tau));
  int err = 0;
  for (int col = 0; col < N && !err; ++col) {
    double h2o[OR_MAXL], co2[OR_MAXL], o3[OR_MAXL], ch4[OR_MAXL], o2[OR_MAXL], cldfrac[OR_MAXL];
    c->nlay = L;
    c->pz[0] = a->plev[col];
    for (int l = 0; l < L; ++l) {
      long i = (long)l * N + col;
      c->pavel[l] = a->play[i]; c->tavel[l] = a->tlay[i]; c->pz[l + 1] = a->plev[i + N];
      h2o[l] = a->h2o[i]; co2[l] = a->co2[i]; o3[l] = a->o3[i]; ch4[l] = a->ch4[i]; o2[l] = a->o2[i];
      cldfrac[l] = icld >= 1 && a->cldfr ? a->cldfr[i] : 0.0;
      if (!a->mcica && cldfrac[l] > 1.e-6 && cldfrac[l] < 1.0 - 1.e-6) err = 10;   /* 'PARTIAL CLOUD NOT ALLOWED' */
    }
    if (err) break;
    sw_setcoef(c, h2o, co2, o3, ch4, o2);
    sw_taumol(c, &so, tm);
    double cossza = a->coszen[col];
    if (cossza < 1.e-10) cossza = 1.e-10;
    /* band cloud optics (cldprop_sw / cldprmc_sw) */
    static double ctau[OR_MAXL][NB], cssa[OR_MAXL][NB], casm[OR_MAXL][NB];
    for (int l = 0; l < L && icld >= 1; ++l) {
      long i = (long)l * N + col;
      double ciwp = a->cicewp ? a->cicewp[i] : 0.0, clwp = a->cliqwp ? a->cliqwp[i] : 0.0, cwp = ciwp + clwp, tauctot = 0.0;
      for (int b = 0; b < NB; ++b) tauctot = tauctot + (a->taucld ? a->taucld[i * NB + b] : 0.0);
      for (int b = 0; b < NB; ++b) {
        double tcb = a->taucld ? a->taucld[i * NB + b] : 0.0;
        ctau[l][b] = 0.0; cssa[l][b] = 1.0; casm[l][b] = 0.0;
        if (a->mcica) { ctau[l][b] = tcb; cssa[l][b] = a->ssacld ? a->ssacld[i * NB + b] : 1.0; casm[l][b] = a->asmcld ? a->asmcld[i * NB + b] : 0.0; }
        int gate = a->mcica ? (cwp >= 1.e-20 || tcb >= 1.e-20) : (cldfrac[l] >= 1.e-20 && (cwp >= 1.e-20 || tauctot >= 1.e-20));
        if (gate) {
          int rc = sw_cldopt(b, a->inflag, a->iceflag, a->liqflag, ciwp, clwp, a->reice ? a->reice[i] : 0.0, a->reliq ? a->reliq[i] : 0.0, tcb,
                             a->ssacld ? a->ssacld[i * NB + b] : 1.0, a->asmcld ? a->asmcld[i * NB + b] : 0.0, a->fsfcld ? a->fsfcld[i * NB + b] : 0.0,
                             &ctau[l][b], &cssa[l][b], &casm[l][b]);
          if (rc) err = rc;
        }
      }
    }
    if (err) break;
    /* aerosol optics by band: iaer = 10 as given, iaer = 6 mixed from the six ECMWF types (rrtmg_sw_rad.nomcica.f90:693-727) */
    static double ztaua[OR_MAXL][NB], zomga[OR_MAXL][NB], zasya[OR_MAXL][NB];
    for (int l = 0; l < L; ++l)
      for (int b = 0; b < NB; ++b) {
        const long o = ((long)b * L + l) * N + col;
        ztaua[l][b] = iaer == 10 && a->tauaer ? a->tauaer[o] : 0.0;
        zomga[l][b] = iaer == 10 && a->ssaaer ? a->ssaaer[o] : 1.0;
        zasya[l][b] = iaer == 10 && a->asmaer ? a->asmaer[o] : 0.0;
        if (iaer == 6) {
          const double *rt = or_f(&G.st, "sw/aer/rsrtaua"), *rp = or_f(&G.st, "sw/aer/rsrpiza"), *ra = or_f(&G.st, "sw/aer/rsrasya");
          double t = 0.0, w = 0.0, g = 0.0;
          for (int ia = 0; ia < 6; ++ia) {
            const double e = a->ecaer ? a->ecaer[((long)ia * L + l) * N + col] : 0.0;
            t = t + rt[b + NB * ia] * e;
            w = w + rt[b + NB * ia] * e * rp[b + NB * ia];
            g = g + rt[b + NB * ia] * e * rp[b + NB * ia] * ra[b + NB * ia];
          }
          if (t == 0.0) { t = 0.0; g = 0.0; w = 1.0; }
          else { if (w != 0.0) g = g / w; if (t != 0.0) w = w / t; }
          ztaua[l][b] = t; zomga[l][b] = w; zasya[l][b] = g;
        }
      }
    double bbfu[OR_MAXL + 1] = {0}, bbfd[OR_MAXL + 1] = {0}, bbcu[OR_MAXL + 1] = {0}, bbcd[OR_MAXL + 1] = {0};
    int iw = 0;
    for (int b = 0; b < NB; ++b) {
      const int vis = (b >= 9 && b <= 12);
      const double albp = vis ? a->asdir[col] : a->aldir[col], albd = vis ? a->asdif[col] : a->aldif[col];
      for (int jg = 0; jg < ngc_[b]; ++jg, ++iw) {
        const double zincflx = adjflux[b] * tm->src[iw] * cossza;
        double ztauc[OR_MAXL], zomcc[OR_MAXL], zgcc[OR_MAXL], ztauo[OR_MAXL], zomco[OR_MAXL], zgco[OR_MAXL];
        double zrefc[OR_MAXL + 1], zrefdc[OR_MAXL + 1], ztrac[OR_MAXL + 1], ztradc[OR_MAXL + 1], zrefo[OR_MAXL + 1], zrefdo[OR_MAXL + 1], ztrao[OR_MAXL + 1], ztrado[OR_MAXL + 1];
        double zref[OR_MAXL + 1], zrefd[OR_MAXL + 1], ztra[OR_MAXL + 1], ztrad[OR_MAXL + 1];
        double zdbtc[OR_MAXL + 1], ztdbtc[OR_MAXL + 1], zdbt[OR_MAXL + 1], ztdbt[OR_MAXL + 1];
        double zrdndc[OR_MAXL + 1], zrupc[OR_MAXL + 1], zrupdc[OR_MAXL + 1], zrdnd[OR_MAXL + 1], zrup[OR_MAXL + 1], zrupd[OR_MAXL + 1];
        double zcd[OR_MAXL + 1], zcu[OR_MAXL + 1], zfd[OR_MAXL + 1], zfu[OR_MAXL + 1], pclfr[OR_MAXL];
        int lrtclr[OR_MAXL], lrtcld[OR_MAXL];
        ztdbtc[0] = 1.0; ztdbt[0] = 1.0;
        zdbtc[L] = 0.0; ztrac[L] = 0.0; ztradc[L] = 0.0; zrefc[L] = albp; zrefdc[L] = albd; zrupc[L] = albp; zrupdc[L] = albd;
        ztrao[L] = 0.0; ztrado[L] = 0.0; zrefo[L] = albp; zrefdo[L] = albd;
        zdbt[L] = 0.0; ztra[L] = 0.0; ztrad[L] = 0.0; zref[L] = albp; zrefd[L] = albd; zrup[L] = albp; zrupd[L] = albd;
        for (int jk = 0; jk < L; ++jk) {
          const int ikl = L - 1 - jk;
          const double ptaua = ztaua[ikl][b], pomga = zomga[ikl][b], pasya = zasya[ikl][b];
          double cf, ptauc = 0.0, pomgc = 1.0, pasyc = 0.0;
          if (icld == 0) cf = 0.0;
          else if (a->mcica) { cf = cloudy[iw + (size_t)NG * (col + (size_t)N * ikl)] ? 1.0 : 0.0; if (cf > 0) { ptauc = ctau[ikl][b]; pomgc = cssa[ikl][b]; pasyc = casm[ikl][b]; } }
          else { cf = cldfrac[ikl]; ptauc = ctau[ikl][b]; pomgc = cssa[ikl][b]; pasyc = casm[ikl][b]; }
          pclfr[jk] = cf;
          lrtclr[jk] = 1; lrtcld[jk] = cf > 1.e-12;
          ztauc[jk] = tm->taur[ikl][iw] + tm->taug[ikl][iw] + ptaua;
          zomcc[jk] = tm->taur[ikl][iw] * 1.0 + ptaua * pomga;
          zgcc[jk] = pasya * pomga * ptaua / zomcc[jk];
          zomcc[jk] = zomcc[jk] / ztauc[jk];
          double zf = zgcc[jk] * zgcc[jk], zwf = zomcc[jk] * zf;
          ztauc[jk] = (1.0 - zwf) * ztauc[jk];
          zomcc[jk] = (zomcc[jk] - zwf) / (1.0 - zwf);
          zgcc[jk] = (zgcc[jk] - zf) / (1.0 - zf);
          ztauo[jk] = ztauc[jk] + ptauc;
          zomco[jk] = ztauc[jk] * zomcc[jk] + ptauc * pomgc;
          zgco[jk] = (ptauc * pomgc * pasyc + ztauc[jk] * zomcc[jk] * zgcc[jk]) / zomco[jk];
          zomco[jk] = zomco[jk] / ztauo[jk];
        }
        sw_reftra(L, lrtclr, zgcc, cossza, ztauc, zomcc, zrefc, zrefdc, ztrac, ztradc);
        sw_reftra(L, lrtcld, zgco, cossza, ztauo, zomco, zrefo, zrefdo, ztrao, ztrado);
        for (int jk = 0; jk < L; ++jk) {
          double zclear = 1.0 - pclfr[jk], zcloud = pclfr[jk];
          zref[jk] = zclear * zrefc[jk] + zcloud * zrefo[jk]; zrefd[jk] = zclear * zrefdc[jk] + zcloud * zrefdo[jk];
          ztra[jk] = zclear * ztrac[jk] + zcloud * ztrao[jk]; ztrad[jk] = zclear * ztradc[jk] + zcloud * ztrado[jk];
          double zdbtmc = dbt_of(ztauc[jk], cossza), zdbtmo = dbt_of(ztauo[jk], cossza);
          zdbtc[jk] = zdbtmc; ztdbtc[jk + 1] = zdbtc[jk] * ztdbtc[jk];
          zdbt[jk] = zclear * zdbtmc + zcloud * zdbtmo; ztdbt[jk + 1] = zdbt[jk] * ztdbt[jk];
        }
        sw_vrtqdr(L, zrefc, zrefdc, ztrac, ztradc, zdbtc, zrdndc, zrupc, zrupdc, ztdbtc, zcd, zcu);
        sw_vrtqdr(L, zref, zrefd, ztra, ztrad, zdbt, zrdnd, zrup, zrupd, ztdbt, zfd, zfu);
        for (int jk = 0; jk <= L; ++jk) {
          int ikl = L - jk;
          bbfu[ikl] = bbfu[ikl] + zincflx * zfu[jk]; bbfd[ikl] = bbfd[ikl] + zincflx * zfd[jk];
          bbcu[ikl] = bbcu[ikl] + zincflx * zcu[jk]; bbcd[ikl] = bbcd[ikl] + zincflx * zcd[jk];
        }
      }
    }
    double netp = 0, netcp = 0;
    for (int i = 0; i <= L; ++i) {
      long o = (long)i * N + col;
      a->swuflxc[o] = bbcu[i]; a->swdflxc[o] = bbcd[i]; a->swuflx[o] = bbfu[i]; a->swdflx[o] = bbfd[i];
      double net = bbfd[i] - bbfu[i], netc = bbcd[i] - bbcu[i];
      if (i > 0) {
        long ol = (long)(i - 1) * N + col;
        double zdpgcp = G.heatfac / c->pdp[i - 1];
        a->swhrc[ol] = (netc - netcp) * zdpgcp; a->swhr[ol] = (net - netp) * zdpgcp;
      }
      netp = net; netcp = netc;
    }
  }
  free(c); free(tm); free(cloudy);
  return err;
}
