// rrtmg_compat.hip -- reference-compatible entry points (same symbols, argument order and array layouts as
// the Fortran bind(c) wrappers bound by climt's Cython shims; see include/rrtmg_hip.h layer (1)).
// They run on a process-global default context, mirroring the reference's module-global state, take HOST
// pointers, and turn the reference's `stop` aborts into a retrievable status.
#include <cstdlib>
#include <mutex>
#include <vector>

#include "rrtmg_ctx.h"

namespace {
rrtmg_ctx *g_ctx = nullptr;
std::mutex g_mu;
int g_status = 0;
std::string g_error;

rrtmg_ctx *default_ctx() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx) {
    const char *dev = getenv("RRTMG_HIP_DEVICE");
    int rc = rrtmg_hip_create(&g_ctx, dev ? atoi(dev) : 0);
    if (rc) { g_status = rc; g_error = g_ctx ? g_ctx->err : "context creation failed"; }
  }
  return g_ctx;
}
void note(int rc) {
  if (rc) { g_status = rc; g_error = g_ctx ? g_ctx->err : "error"; fprintf(stderr, "librrtmg_hip: error %d: %s\n", rc, g_error.c_str()); }
  else g_status = 0;
}

// first cloudy sub-column of band [g0, g1) at (lay, col), or -1
inline int first_cloudy(const double *cldfmcl, size_t cell, int ngpt, int g0, int g1) {
  for (int g = g0; g < g1; ++g) if (cldfmcl[cell * ngpt + g] > 1.e-12) return g;
  return -1;
}
const int kSwGs[15] = {0, 6, 18, 26, 34, 44, 54, 56, 66, 74, 80, 86, 94, 100, 112};
const int kLwGs[17] = {0, 10, 22, 38, 52, 68, 76, 88, 96, 108, 114, 122, 130, 134, 136, 138, 140};
}  // namespace

extern "C" {

int rrtmg_hip_default_status(void) { return g_status; }
const char *rrtmg_hip_default_error(void) { return g_error.c_str(); }

// ---- shortwave --------------------------------------------------------------------------------
void rrtmg_sw_set_constants(double *pi, double *grav, double *planck, double *boltz, double *clight, double *avogad,
                            double *alosmt, double *gascon, double *sbcnst, double *secdy) {
  rrtmg_ctx *c = default_ctx();
  if (c) note(rrtmg_hip_set_constants(c, *pi, *grav, *planck, *boltz, *clight, *avogad, *alosmt, *gascon, *sbcnst, *secdy));
}
void rrtmg_sw_ini_wrapper(double *cpdair) {
  rrtmg_ctx *c = default_ctx();
  if (c) note(rrtmg_hip_sw_init(c, *cpdair, nullptr));
}

void mcica_subcol_sw_wrapper(int32_t *iplon, int32_t *ncol, int32_t *nlay, int32_t *icld, int32_t *permuteseed, int32_t *irng,
                             double *play, double *cldfrac, double *ciwp, double *clwp, double *rei, double *rel, double *tauc,
                             double *ssac, double *asmc, double *fsfc, double *cldfmcl, double *ciwpmcl, double *clwpmcl,
                             double *reicmcl, double *relqmcl, double *taucmcl, double *ssacmcl, double *asmcmcl, double *fsfcmcl) {
  (void)iplon;
  rrtmg_ctx *c = default_ctx();
  if (!c) return;
  if (*icld == 0) return;                       // mcica_subcol_gen_sw.f90:146
  if (*irng != 0) *irng = 1;
  const int N = *ncol, L = *nlay, G = 112;
  int rc = rrtmg_hip_mcica_mask(c, 0, N, L, *icld, *permuteseed, *irng, play, cldfrac, cldfmcl);
  note(rc);
  if (rc) return;
  for (size_t cell = 0; cell < (size_t)N * L; ++cell) {
    reicmcl[cell] = rei[cell];
    relqmcl[cell] = rel[cell];
    for (int b = 0; b < 14; ++b)
      for (int g = kSwGs[b]; g < kSwGs[b + 1]; ++g) {
        const bool cl = cldfmcl[cell * G + g] > 0.5;
        clwpmcl[cell * G + g] = cl ? clwp[cell] : 0.0;
        ciwpmcl[cell * G + g] = cl ? ciwp[cell] : 0.0;
        taucmcl[cell * G + g] = cl ? tauc[cell * 14 + b] : 0.0;
        ssacmcl[cell * G + g] = cl ? ssac[cell * 14 + b] : 1.0;
        asmcmcl[cell * G + g] = cl ? asmc[cell * 14 + b] : 0.0;
        fsfcmcl[cell * G + g] = cl ? fsfc[cell * 14 + b] : 0.0;
      }
  }
}

static void sw_common(rrtmg_sw_args &a, int32_t *ncol, int32_t *nlay, int32_t *icld, int32_t *iaer, double *play, double *plev,
                      double *tlay, double *tlev, double *tsfc, double *h2ovmr, double *o3vmr, double *co2vmr, double *ch4vmr,
                      double *n2ovmr, double *o2vmr, double *asdir, double *asdif, double *aldir, double *aldif, double *coszen,
                      double *adjes, int32_t *dyofyr, double *scon, int32_t *isolvar, int32_t *inflgsw, int32_t *iceflgsw,
                      int32_t *liqflgsw, double *tauaer, double *ssaaer, double *asmaer, double *ecaer, double *swuflx,
                      double *swdflx, double *swhr, double *swuflxc, double *swdflxc, double *swhrc, double *bndsolvar,
                      double *indsolvar, double *solcycfrac) {
  a = rrtmg_sw_args{};
  a.struct_size = (int32_t)sizeof a;
  a.ncol = *ncol; a.nlay = *nlay; a.memspace = 0;
  if (*icld < 0 || *icld > 3) *icld = 2;                       // intent(inout), rrtmg_sw_rad.nomcica.f90:563
  if (*iaer != 0 && *iaer != 6 && *iaer != 10) *iaer = 0;
  a.icld = *icld; a.iaer = *iaer; a.inflgsw = *inflgsw; a.iceflgsw = *iceflgsw; a.liqflgsw = *liqflgsw;
  a.dyofyr = *dyofyr; a.isolvar = *isolvar; a.adjes = *adjes; a.scon = *scon; a.solcycfrac = solcycfrac ? *solcycfrac : 0.0;
  a.bndsolvar = bndsolvar; a.indsolvar = indsolvar;
  a.play = play; a.plev = plev; a.tlay = tlay; a.tlev = tlev; a.tsfc = tsfc; a.h2ovmr = h2ovmr; a.o3vmr = o3vmr; a.co2vmr = co2vmr;
  a.ch4vmr = ch4vmr; a.n2ovmr = n2ovmr; a.o2vmr = o2vmr; a.asdir = asdir; a.asdif = asdif; a.aldir = aldir; a.aldif = aldif; a.coszen = coszen;
  a.tauaer = tauaer; a.ssaaer = ssaaer; a.asmaer = asmaer; a.ecaer = ecaer;
  a.swuflx = swuflx; a.swdflx = swdflx; a.swhr = swhr; a.swuflxc = swuflxc; a.swdflxc = swdflxc; a.swhrc = swhrc;
}

void rrtmg_sw_nomcica_wrapper(int32_t *ncol, int32_t *nlay, int32_t *icld, int32_t *iaer, double *play, double *plev, double *tlay,
                              double *tlev, double *tsfc, double *h2ovmr, double *o3vmr, double *co2vmr, double *ch4vmr,
                              double *n2ovmr, double *o2vmr, double *asdir, double *asdif, double *aldir, double *aldif,
                              double *coszen, double *adjes, int32_t *dyofyr, double *scon, int32_t *isolvar, int32_t *inflgsw,
                              int32_t *iceflgsw, int32_t *liqflgsw, double *cldfr, double *taucld, double *ssacld, double *asmcld,
                              double *fsfcld, double *cicewp, double *cliqwp, double *reice, double *reliq, double *tauaer,
                              double *ssaaer, double *asmaer, double *ecaer, double *swuflx, double *swdflx, double *swhr,
                              double *swuflxc, double *swdflxc, double *swhrc, double *bndsolvar, double *indsolvar, double *solcycfrac) {
  rrtmg_ctx *c = default_ctx();
  if (!c) return;
  rrtmg_sw_args a;
  sw_common(a, ncol, nlay, icld, iaer, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, asdir, asdif, aldir,
            aldif, coszen, adjes, dyofyr, scon, isolvar, inflgsw, iceflgsw, liqflgsw, tauaer, ssaaer, asmaer, ecaer, swuflx, swdflx,
            swhr, swuflxc, swdflxc, swhrc, bndsolvar, indsolvar, solcycfrac);
  a.mcica = 0;
  a.cldfr = cldfr; a.taucld = taucld; a.ssacld = ssacld; a.asmcld = asmcld; a.fsfcld = fsfcld;
  a.cicewp = cicewp; a.cliqwp = cliqwp; a.reice = reice; a.reliq = reliq;
  note(rrtmg_hip_sw_fluxes(c, &a));
}

void rrtmg_sw_mcica_wrapper(int32_t *ncol, int32_t *nlay, int32_t *icld, int32_t *iaer, double *play, double *plev, double *tlay,
                            double *tlev, double *tsfc, double *h2ovmr, double *o3vmr, double *co2vmr, double *ch4vmr, double *n2ovmr,
                            double *o2vmr, double *asdir, double *asdif, double *aldir, double *aldif, double *coszen, double *adjes,
                            int32_t *dyofyr, double *scon, int32_t *isolvar, int32_t *inflgsw, int32_t *iceflgsw, int32_t *liqflgsw,
                            double *cldfmcl, double *taucmcl, double *ssacmcl, double *asmcmcl, double *fsfcmcl, double *ciwpmcl,
                            double *clwpmcl, double *reicmcl, double *relqmcl, double *tauaer, double *ssaaer, double *asmaer,
                            double *ecaer, double *swuflx, double *swdflx, double *swhr, double *swuflxc, double *swdflxc,
                            double *swhrc, double *bndsolvar, double *indsolvar, double *solcycfrac) {
  rrtmg_ctx *c = default_ctx();
  if (!c) return;
  rrtmg_sw_args a;
  sw_common(a, ncol, nlay, icld, iaer, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, asdir, asdif, aldir,
            aldif, coszen, adjes, dyofyr, scon, isolvar, inflgsw, iceflgsw, liqflgsw, tauaer, ssaaer, asmaer, ecaer, swuflx, swdflx,
            swhr, swuflxc, swdflxc, swhrc, bndsolvar, indsolvar, solcycfrac);
  a.mcica = 1;
  // band-level quantities are recovered from the first cloudy sub-column of each band (the generator writes
  // the band value into every cloudy sub-column, mcica_subcol_gen_sw.f90:474-497)
  const int N = *ncol, L = *nlay, G = 112;
  const size_t nl = (size_t)N * L;
  std::vector<double> cf(nl, 0.0), ci(nl, 0.0), cl(nl, 0.0), tc(nl * 14, 0.0), sc(nl * 14, 1.0), ac(nl * 14, 0.0), fc(nl * 14, 0.0);
  for (size_t cell = 0; cell < nl; ++cell)
    for (int b = 0; b < 14; ++b) {
      const int g = first_cloudy(cldfmcl, cell, G, kSwGs[b], kSwGs[b + 1]);
      if (g < 0) continue;
      cf[cell] = 1.0;
      ci[cell] = ciwpmcl[cell * G + g]; cl[cell] = clwpmcl[cell * G + g];
      tc[cell * 14 + b] = taucmcl[cell * G + g]; sc[cell * 14 + b] = ssacmcl[cell * G + g];
      ac[cell * 14 + b] = asmcmcl[cell * G + g]; fc[cell * 14 + b] = fsfcmcl[cell * G + g];
    }
  a.cldfr = cf.data(); a.cicewp = ci.data(); a.cliqwp = cl.data(); a.reice = reicmcl; a.reliq = relqmcl;
  a.taucld = tc.data(); a.ssacld = sc.data(); a.asmcld = ac.data(); a.fsfcld = fc.data();
  a.cldfmcl = cldfmcl;
  note(rrtmg_hip_sw_fluxes(c, &a));
}

// ---- longwave ----------------------------------------------------------------------------------
void rrtmg_set_constants(double *pi, double *grav, double *planck, double *boltz, double *clight, double *avogad, double *alosmt,
                         double *gascon, double *sbcnst, double *secdy) {
  rrtmg_sw_set_constants(pi, grav, planck, boltz, clight, avogad, alosmt, gascon, sbcnst, secdy);
}
void rrtmg_lw_set_constants(double *pi, double *grav, double *planck, double *boltz, double *clight, double *avogad, double *alosmt,
                            double *gascon, double *sbcnst, double *secdy) {
  rrtmg_sw_set_constants(pi, grav, planck, boltz, clight, avogad, alosmt, gascon, sbcnst, secdy);
}
void rrtmg_lw_ini_wrapper(double *cpdair) {
  rrtmg_ctx *c = default_ctx();
  if (c) note(rrtmg_hip_lw_init(c, *cpdair, nullptr));
}

void mcica_subcol_lw_wrapper(int32_t *iplon, int32_t *ncol, int32_t *nlay, int32_t *icld, int32_t *permuteseed, int32_t *irng,
                             double *play, double *cldfrac, double *ciwp, double *clwp, double *rei, double *rel, double *tauc,
                             double *cldfmcl, double *ciwpmcl, double *clwpmcl, double *reicmcl, double *relqmcl, double *taucmcl) {
  (void)iplon;
  rrtmg_ctx *c = default_ctx();
  if (!c) return;
  if (*icld == 0) return;
  if (*irng != 0) *irng = 1;
  const int N = *ncol, L = *nlay, G = 140;
  int rc = rrtmg_hip_mcica_mask(c, 1, N, L, *icld, *permuteseed, *irng, play, cldfrac, cldfmcl);
  note(rc);
  if (rc) return;
  for (size_t cell = 0; cell < (size_t)N * L; ++cell) {
    reicmcl[cell] = rei[cell];
    relqmcl[cell] = rel[cell];
    for (int b = 0; b < 16; ++b)
      for (int g = kLwGs[b]; g < kLwGs[b + 1]; ++g) {
        const bool cl = cldfmcl[cell * G + g] > 0.5;
        clwpmcl[cell * G + g] = cl ? clwp[cell] : 0.0;
        ciwpmcl[cell * G + g] = cl ? ciwp[cell] : 0.0;
        taucmcl[cell * G + g] = cl ? tauc[cell * 16 + b] : 0.0;
      }
  }
}

static void lw_common(rrtmg_lw_args &a, int32_t *ncol, int32_t *nlay, int32_t *icld, int32_t *idrv, double *play, double *plev,
                      double *tlay, double *tlev, double *tsfc, double *h2ovmr, double *o3vmr, double *co2vmr, double *ch4vmr,
                      double *n2ovmr, double *o2vmr, double *cfc11vmr, double *cfc12vmr, double *cfc22vmr, double *ccl4vmr, double *emis,
                      int32_t *inflglw, int32_t *iceflglw, int32_t *liqflglw, double *tauaer, double *uflx, double *dflx, double *hr,
                      double *uflxc, double *dflxc, double *hrc, double *duflx_dt, double *duflxc_dt) {
  a = rrtmg_lw_args{};
  a.struct_size = (int32_t)sizeof a;
  a.ncol = *ncol; a.nlay = *nlay; a.memspace = 0;
  if (*icld < 0 || *icld > 3) *icld = 2;
  a.icld = *icld; a.idrv = *idrv; a.inflglw = *inflglw; a.iceflglw = *iceflglw; a.liqflglw = *liqflglw;
  a.play = play; a.plev = plev; a.tlay = tlay; a.tlev = tlev; a.tsfc = tsfc; a.h2ovmr = h2ovmr; a.o3vmr = o3vmr; a.co2vmr = co2vmr;
  a.ch4vmr = ch4vmr; a.n2ovmr = n2ovmr; a.o2vmr = o2vmr; a.cfc11vmr = cfc11vmr; a.cfc12vmr = cfc12vmr; a.cfc22vmr = cfc22vmr;
  a.ccl4vmr = ccl4vmr; a.emis = emis; a.tauaer = tauaer;
  a.uflx = uflx; a.dflx = dflx; a.hr = hr; a.uflxc = uflxc; a.dflxc = dflxc; a.hrc = hrc; a.duflx_dt = duflx_dt; a.duflxc_dt = duflxc_dt;
}

void rrtmg_lw_nomcica_wrapper(int32_t *ncol, int32_t *nlay, int32_t *icld, int32_t *idrv, double *play, double *plev, double *tlay,
                              double *tlev, double *tsfc, double *h2ovmr, double *o3vmr, double *co2vmr, double *ch4vmr, double *n2ovmr,
                              double *o2vmr, double *cfc11vmr, double *cfc12vmr, double *cfc22vmr, double *ccl4vmr, double *emis,
                              int32_t *inflglw, int32_t *iceflglw, int32_t *liqflglw, double *cldfr, double *taucld, double *cicewp,
                              double *cliqwp, double *reice, double *reliq, double *tauaer, double *uflx, double *dflx, double *hr,
                              double *uflxc, double *dflxc, double *hrc, double *duflx_dt, double *duflxc_dt) {
  rrtmg_ctx *c = default_ctx();
  if (!c) return;
  rrtmg_lw_args a;
  lw_common(a, ncol, nlay, icld, idrv, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr,
            cfc22vmr, ccl4vmr, emis, inflglw, iceflglw, liqflglw, tauaer, uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt);
  a.mcica = 0;
  a.cldfr = cldfr; a.taucld = taucld; a.cicewp = cicewp; a.cliqwp = cliqwp; a.reice = reice; a.reliq = reliq;
  note(rrtmg_hip_lw_fluxes(c, &a));
}

void rrtmg_lw_mcica_wrapper(int32_t *ncol, int32_t *nlay, int32_t *icld, int32_t *idrv, double *play, double *plev, double *tlay,
                            double *tlev, double *tsfc, double *h2ovmr, double *o3vmr, double *co2vmr, double *ch4vmr, double *n2ovmr,
                            double *o2vmr, double *cfc11vmr, double *cfc12vmr, double *cfc22vmr, double *ccl4vmr, double *emis,
                            int32_t *inflglw, int32_t *iceflglw, int32_t *liqflglw, double *cldfmcl, double *taucmcl, double *ciwpmcl,
                            double *clwpmcl, double *reicmcl, double *relqmcl, double *tauaer, double *uflx, double *dflx, double *hr,
                            double *uflxc, double *dflxc, double *hrc, double *duflx_dt, double *duflxc_dt) {
  rrtmg_ctx *c = default_ctx();
  if (!c) return;
  rrtmg_lw_args a;
  lw_common(a, ncol, nlay, icld, idrv, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr,
            cfc22vmr, ccl4vmr, emis, inflglw, iceflglw, liqflglw, tauaer, uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt);
  a.mcica = 1;
  const int N = *ncol, L = *nlay, G = 140;
  const size_t nl = (size_t)N * L;
  std::vector<double> cf(nl, 0.0), ci(nl, 0.0), cl(nl, 0.0), tc(nl * 16, 0.0);
  for (size_t cell = 0; cell < nl; ++cell)
    for (int b = 0; b < 16; ++b) {
      const int g = first_cloudy(cldfmcl, cell, G, kLwGs[b], kLwGs[b + 1]);
      if (g < 0) continue;
      cf[cell] = 1.0;
      ci[cell] = ciwpmcl[cell * G + g]; cl[cell] = clwpmcl[cell * G + g];
      tc[cell * 16 + b] = taucmcl[cell * G + g];
    }
  a.cldfr = cf.data(); a.cicewp = ci.data(); a.cliqwp = cl.data(); a.reice = reicmcl; a.reliq = relqmcl; a.taucld = tc.data();
  a.cldfmcl = cldfmcl;
  note(rrtmg_hip_lw_fluxes(c, &a));
}

}  // extern "C"
