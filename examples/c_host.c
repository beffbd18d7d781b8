/* A plain-C host of librrtmg_hip.so: what a non-Python caller of the C-ABI (include/rrtmg_hip.h) looks like.
 * 64 clear-sky columns x 30 layers on host memory, shortwave + longwave, status codes checked the way a caller would.
 *
 *   gcc -std=c99 -Iinclude examples/c_host.c -Lclimt_amd/_lib -lrrtmg_hip -Wl,-rpath,$PWD/climt_amd/_lib -o /tmp/c_host
 *   /tmp/c_host            (tables are found next to the library; exits 2 with the library's message when no GPU is there)
 *
 * The inputs use only + - * / so that tests/test_gpu_parity.py can rebuild them bit for bit in numpy. */
#include <stdio.h>
#include <stdlib.h>

#include "rrtmg_hip.h"

#define NCOL 64
#define NLAY 30

static int fail(rrtmg_ctx *ctx, const char *what, int rc) {
  fprintf(stderr, "%s: status %d: %s\n", what, rc, ctx ? rrtmg_hip_last_error(ctx) : rrtmg_hip_default_error());
  if (ctx) rrtmg_hip_destroy(ctx);
  return 2;
}

int main(void) {
  static double play[NLAY][NCOL], plev[NLAY + 1][NCOL], tlay[NLAY][NCOL], tlev[NLAY + 1][NCOL], tsfc[NCOL];
  static double h2o[NLAY][NCOL], o3[NLAY][NCOL], co2[NLAY][NCOL], ch4[NLAY][NCOL], n2o[NLAY][NCOL], o2[NLAY][NCOL], zero[NLAY][NCOL];
  static double alb[NCOL], coszen[NCOL], emis[RRTMG_NBNDLW][NCOL];
  static double swu[NLAY + 1][NCOL], swd[NLAY + 1][NCOL], swh[NLAY][NCOL], swuc[NLAY + 1][NCOL], swdc[NLAY + 1][NCOL], swhc[NLAY][NCOL];
  static double lwu[NLAY + 1][NCOL], lwd[NLAY + 1][NCOL], lwh[NLAY][NCOL], lwuc[NLAY + 1][NCOL], lwdc[NLAY + 1][NCOL], lwhc[NLAY][NCOL];
  rrtmg_ctx *ctx = NULL;
  rrtmg_sw_args sw = {0};
  rrtmg_lw_args lw = {0};
  int rc, c, k, b;
  sw.struct_size = (int32_t)sizeof sw;   /* the header this file was compiled against: the library refuses any other */
  lw.struct_size = (int32_t)sizeof lw;

  /* layer k between interfaces k (below) and k+1 (above); pressures in hPa, level 0 = surface */
  for (c = 0; c < NCOL; ++c) {
    const double ps = 1000.0 + 0.25 * c;
    for (k = 0; k <= NLAY; ++k) {
      const double x = 1.0 - (double)k / NLAY;
      plev[k][c] = 0.5 + (ps - 0.5) * x * x;
      tlev[k][c] = 210.0 + 78.0 * x;
    }
    for (k = 0; k < NLAY; ++k) {
      const double x = 1.0 - (k + 0.5) / NLAY;
      play[k][c] = 0.5 * (plev[k][c] + plev[k + 1][c]);
      tlay[k][c] = 0.5 * (tlev[k][c] + tlev[k + 1][c]);
      h2o[k][c] = 1.0e-6 + 0.012 * x * x * x * x;
      o3[k][c] = 4.0e-8 + 6.0e-6 * (1.0 - x) * (1.0 - x);
      co2[k][c] = 400.0e-6; ch4[k][c] = 1.8e-6; n2o[k][c] = 0.32e-6; o2[k][c] = 0.209;
    }
    tsfc[c] = 289.0;
    alb[c] = 0.1 + 0.002 * c;
    coszen[c] = 0.2 + 0.0125 * c;
    for (b = 0; b < RRTMG_NBNDLW; ++b) emis[b][c] = 0.98;
  }

  rc = rrtmg_hip_create(&ctx, 0);
  if (rc) return fail(ctx, "rrtmg_hip_create", rc);
  rc = rrtmg_hip_set_constants(ctx, 3.14159265358979323846, 9.80665, 6.62607004e-27, 1.38064852e-16, 2.99792458e10, 6.022140857e23,
                               2.6867774e19, 8.3144598e7, 5.670367e-12, 86400.0);
  if (rc) return fail(ctx, "rrtmg_hip_set_constants", rc);
  if ((rc = rrtmg_hip_sw_init(ctx, 1004.64, NULL))) return fail(ctx, "rrtmg_hip_sw_init", rc);
  if ((rc = rrtmg_hip_lw_init(ctx, 1004.64, NULL))) return fail(ctx, "rrtmg_hip_lw_init", rc);

  sw.ncol = NCOL; sw.nlay = NLAY; sw.memspace = 0; sw.mcica = 0; sw.icld = 0; sw.iaer = 0;
  sw.inflgsw = 2; sw.iceflgsw = 1; sw.liqflgsw = 1; sw.dyofyr = 1; sw.isolvar = 0;
  sw.adjes = 1.0; sw.scon = 1367.0; sw.solcycfrac = 0.0;
  sw.play = &play[0][0]; sw.plev = &plev[0][0]; sw.tlay = &tlay[0][0]; sw.tlev = &tlev[0][0]; sw.tsfc = tsfc;
  sw.h2ovmr = &h2o[0][0]; sw.o3vmr = &o3[0][0]; sw.co2vmr = &co2[0][0]; sw.ch4vmr = &ch4[0][0]; sw.n2ovmr = &n2o[0][0]; sw.o2vmr = &o2[0][0];
  sw.asdir = alb; sw.asdif = alb; sw.aldir = alb; sw.aldif = alb; sw.coszen = coszen;
  sw.swuflx = &swu[0][0]; sw.swdflx = &swd[0][0]; sw.swhr = &swh[0][0]; sw.swuflxc = &swuc[0][0]; sw.swdflxc = &swdc[0][0]; sw.swhrc = &swhc[0][0];
  if ((rc = rrtmg_hip_sw_fluxes(ctx, &sw))) return fail(ctx, "rrtmg_hip_sw_fluxes", rc);

  lw.ncol = NCOL; lw.nlay = NLAY; lw.memspace = 0; lw.mcica = 0; lw.icld = 0; lw.idrv = 0;
  lw.inflglw = 2; lw.iceflglw = 1; lw.liqflglw = 1;
  lw.play = &play[0][0]; lw.plev = &plev[0][0]; lw.tlay = &tlay[0][0]; lw.tlev = &tlev[0][0]; lw.tsfc = tsfc;
  lw.h2ovmr = &h2o[0][0]; lw.o3vmr = &o3[0][0]; lw.co2vmr = &co2[0][0]; lw.ch4vmr = &ch4[0][0]; lw.n2ovmr = &n2o[0][0]; lw.o2vmr = &o2[0][0];
  lw.cfc11vmr = &zero[0][0]; lw.cfc12vmr = &zero[0][0]; lw.cfc22vmr = &zero[0][0]; lw.ccl4vmr = &zero[0][0];
  lw.emis = &emis[0][0];
  lw.uflx = &lwu[0][0]; lw.dflx = &lwd[0][0]; lw.hr = &lwh[0][0]; lw.uflxc = &lwuc[0][0]; lw.dflxc = &lwdc[0][0]; lw.hrc = &lwhc[0][0];
  if ((rc = rrtmg_hip_lw_fluxes(ctx, &lw))) return fail(ctx, "rrtmg_hip_lw_fluxes", rc);

  printf("%s%s\n", rrtmg_hip_version(), rrtmg_hip_lw_tables_synthetic(ctx) ? " (longwave k-tables: synthetic)" : "");
  for (c = 0; c < NCOL; c += 21)
    printf("column %2d  sw toa_down %.10f toa_up %.10f sfc_down %.10f hr_top %.10f  lw olr %.10f sfc_down %.10f hr_bottom %.10f\n", c,
           swd[NLAY][c], swu[NLAY][c], swd[0][c], swh[NLAY - 1][c], lwu[NLAY][c], lwd[0][c], lwh[0][c]);

  /* a former Fortran `stop`: the call returns a status and a message instead of ending the process */
  sw.ncol = 0;
  rc = rrtmg_hip_sw_fluxes(ctx, &sw);
  printf("ncol = 0 -> status %d (%s)\n", rc, rrtmg_hip_last_error(ctx));
  rrtmg_hip_destroy(ctx);
  return 0;
}
