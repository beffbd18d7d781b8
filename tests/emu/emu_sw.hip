// TEST INFRASTRUCTURE ONLY -- host emulation of the shortwave DEVICE functions.
//
// Runs the very same __host__ __device__ per-thread functions the gfx950 kernels run
// (climt_amd/csrc/rrtmg_sw_device.h), thread by thread on the CPU, so the device arithmetic can be
// parity-checked in the build container (which has no GPU).  It is compiled into
// tests/_emu/librrtmg_emu.so by tests/emu/build.sh, is never loaded by the product, and is not a
// fallback: librrtmg_hip.so fails loudly without a GPU.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../climt_amd/csrc/rrtmg_sw_device.h"
#include "../../climt_amd/csrc/rrtmg_sw_host.h"
#include "../../climt_amd/csrc/rrtmg_kiss_host.h"
#include "../../include/rrtmg_hip.h"

using namespace rrtmg;

namespace rrtmg {
void mt_mask_host(int ncol, int nlay, int nsub, int icld, int seed, const double *cldfr, std::vector<uint64_t> &mask, int nw, int col0 = 0, int ncol_total = 0);
}

static void emu_solve(const SwDev &d, const SwTab &T) {
  std::vector<double> scr((size_t)F_NTOT * d.nlay * 4);
  for (int col = 0; col < d.ncol; ++col) {
    // the clear-sky variant for cloud-free columns, as the device picks it per tile
    const bool cld = d.anycld[col] != 0;
    for (int i = 0; i < T.nitem; ++i) {
      const int item = T.item[i];
      SwPartSink sink = sw_part_sink(d, i, col);
      if (cld) sw_solve_item<true>(d, T, T.t + T.exp_tbl, item, col, scr.data(), 1, sink);
      else sw_solve_item<false>(d, T, T.t + T.exp_tbl, item, col, scr.data(), 1, sink);
    }
  }
}

extern "C" int emu_sw_fluxes(const rrtmg_sw_args *a, const char *blob_path, double cpdair, const double *consts, char *errbuf, int errlen) {
  auto fail = [&](int code, const std::string &m) { if (errbuf) { strncpy(errbuf, m.c_str(), errlen - 1); errbuf[errlen - 1] = 0; } return code; };
  Blob blob;
  std::string err;
  if (!blob.load(blob_path, err)) return fail(3, err);
  TableSet ts;
  Constants k{};
  k.pi = consts[0]; k.grav = consts[1]; k.planck = consts[2]; k.boltz = consts[3]; k.clight = consts[4];
  k.avogad = consts[5]; k.alosmt = consts[6]; k.gascon = consts[7]; k.sbcnst = consts[8]; k.secdy = consts[9];
  if (!build_tables(blob, "sw", cpdair, k.grav, k.secdy, ts, err)) return fail(3, err);
  SwTab T{};
  if (!build_sw_tab(ts, T, err)) return fail(3, err);
  T.t = ts.flat.data();
  const int N = a->ncol, L = a->nlay;
  const size_t nl = (size_t)N * L, nl1 = (size_t)N * (L + 1);
  SwDev d{};
  d.ncol = N; d.nlay = L; d.icld = a->icld; d.iaer = a->iaer;
  if (d.icld < 0 || d.icld > 3) d.icld = 2;
  if (d.iaer != 0 && d.iaer != 6 && d.iaer != 10) d.iaer = 0;
  d.inflag = a->inflgsw; d.iceflag = a->iceflgsw; d.liqflag = a->liqflgsw; d.mcica = a->mcica ? 1 : 0;
  d.k = k;
  std::vector<double> svar_col;
  const long omg = ts.off("sw/sol/mgavgcyc"), osb = ts.off("sw/sol/sbavgcyc");
  int rc = sw_scalar_setup(d, a->ncol, a->isolvar, a->adjes, a->dyofyr, a->scon, a->solcycfrac, a->bndsolvar, a->indsolvar,
                           omg >= 0 ? ts.flat.data() + omg : nullptr, osb >= 0 ? ts.flat.data() + osb : nullptr, svar_col, err);
  if (!svar_col.empty()) d.svar_col = svar_col.data();
  if (rc) return fail(rc, err);
  d.play = a->play; d.plev = a->plev; d.tlay = a->tlay; d.h2o = a->h2ovmr; d.o3 = a->o3vmr; d.co2 = a->co2vmr;
  d.ch4 = a->ch4vmr; d.n2o = a->n2ovmr; d.o2 = a->o2vmr; d.asdir = a->asdir; d.asdif = a->asdif; d.aldir = a->aldir;
  d.aldif = a->aldif; d.coszen = a->coszen;
  if (d.icld >= 1) {
    d.cldfr = a->cldfr; d.taucld = a->taucld; d.ssacld = a->ssacld; d.asmcld = a->asmcld; d.fsfcld = a->fsfcld;
    d.cicewp = a->cicewp; d.cliqwp = a->cliqwp; d.reice = a->reice; d.reliq = a->reliq;
  }
  if (d.iaer == 10) { d.tauaer = a->tauaer; d.ssaaer = a->ssaaer; d.asmaer = a->asmaer; }
  std::vector<std::vector<double>> keep;
  auto wd = [&](size_t n) { keep.emplace_back(n, 0.0); return keep.back().data(); };
  d.prep = wd(sw_prep_size(N, L)); d.pdp = wd(nl); d.cossza = wd(N);
  std::vector<int32_t> laytrop(N), laysolfr((size_t)N * kSwNBand), anycld(N);
  d.laytrop = laytrop.data(); d.laysolfr = laysolfr.data(); d.anycld = anycld.data();
  if (d.icld >= 1) { d.ctau = wd(nl * kSwNBand); d.cssa = wd(nl * kSwNBand); d.casm = wd(nl * kSwNBand); }
  d.nw = (L + 63) / 64;
  std::vector<uint64_t> mask;
  d.col0 = 0; d.pcols = N;
  d.part = wd((size_t)kSwNSlot * 4 * nl1);
  d.swuflx = a->swuflx; d.swdflx = a->swdflx; d.swhr = a->swhr; d.swuflxc = a->swuflxc; d.swdflxc = a->swdflxc; d.swhrc = a->swhrc;
  int errflag = 0;
  d.err = &errflag;
  for (int c = 0; c < N; ++c) { for (int l = 0; l < L; ++l) sw_prep_layer(d, T, c, l); sw_prep_column(d, T, c); }
  std::vector<double> ta, om, as;
  if (d.iaer == 6) {
    ta.assign(nl * kSwNBand, 0); om.assign(nl * kSwNBand, 0); as.assign(nl * kSwNBand, 0);
    const double *t = T.t;
    for (int lay = 0; lay < L; ++lay) for (int col = 0; col < N; ++col) for (int ib = 0; ib < kSwNBand; ++ib) {
      double ztaua = 0.0, zasya = 0.0, zomga = 0.0;
      for (int ia = 0; ia < 6; ++ia) {
        const double e = a->ecaer[((size_t)ia * L + lay) * N + col];
        const double rt = t[T.rsrtaua + ib + kSwNBand * ia], rp = t[T.rsrpiza + ib + kSwNBand * ia], ra = t[T.rsrasya + ib + kSwNBand * ia];
        ztaua = ztaua + rt * e; zomga = zomga + rt * e * rp; zasya = zasya + rt * e * rp * ra;
      }
      if (ztaua == 0.0) { zasya = 0.0; zomga = 1.0; } else { if (zomga != 0.0) zasya = zasya / zomga; zomga = zomga / ztaua; }
      const size_t o = ((size_t)ib * L + lay) * N + col;
      ta[o] = ztaua; om[o] = zomga; as[o] = zasya;
    }
    d.tauaer = ta.data(); d.ssaaer = om.data(); d.asmaer = as.data();
  }
  if (d.icld >= 1) {
    for (int lay = 0; lay < L; ++lay) for (int c = 0; c < N; ++c) sw_cloud_layer(d, T, c, lay);
    if (d.mcica) {
      mask.assign((size_t)kSwNGpt * d.nw * N, 0);
      d.mask = mask.data();
      if (a->cldfmcl) {
        for (int g = 0; g < kSwNGpt; ++g) for (int l = 0; l < L; ++l) for (int c = 0; c < N; ++c)
          if (a->cldfmcl[((size_t)l * N + c) * kSwNGpt + g] > 1.e-12) mask[((size_t)g * d.nw + (l >> 6)) * N + c] |= 1ull << (l & 63);
      } else if (a->irng == 0) {
        for (int c = 0; c < N; ++c) kiss_mask_column(N, L, kSwNGpt, d.icld, a->permuteseed, d.play, d.cldfr, d.mask, d.nw, d.err, c);
      } else {
        mt_mask_host(N, L, kSwNGpt, d.icld, a->permuteseed, a->cldfr, mask, d.nw, a->shard_col0, a->shard_ncol);
        d.mask = mask.data();
      }
    }
  }
  emu_solve(d, T);
  for (int lev = 0; lev <= L; ++lev) for (int c = 0; c < N; ++c) sw_flux_level(d, T, c, lev, d.anycld[c] != 0);
  for (int l = 0; l < L; ++l) for (int c = 0; c < N; ++c) sw_heat_layer(d, T, c, l);
  if (errflag) return fail(errflag, "device-side error flag " + std::to_string(errflag));
  return 0;
}

extern "C" int emu_mask(int which, int ncol, int nlay, int icld, int seed, int irng, const double *play, const double *cldfr, double *cldfmcl) {
  const int nsub = which == 0 ? 112 : 140, nw = (nlay + 63) / 64;
  std::vector<uint64_t> mask((size_t)nsub * nw * ncol, 0);
  int err = 0;
  if (irng == 0) { for (int c = 0; c < ncol; ++c) kiss_mask_column(ncol, nlay, nsub, icld, seed, play, cldfr, mask.data(), nw, &err, c); }
  else if (irng == -1) {
    // the device kernel's decomposition: one (column, sub-column) at a time through the jump-ahead operators
    std::vector<uint32_t> jumps;
    kiss_build_jumps(nsub, nlay, icld, seed, jumps);
    for (int g = 0; g < nsub; ++g) for (int c = 0; c < ncol; ++c) kiss_mask_jump(ncol, nlay, icld, play, cldfr, mask.data(), nw, &err, jumps.data(), c, g);
  }
  else mt_mask_host(ncol, nlay, nsub, icld, seed, cldfr, mask, nw);
  for (int l = 0; l < nlay; ++l) for (int c = 0; c < ncol; ++c) for (int g = 0; g < nsub; ++g)
    cldfmcl[((size_t)l * ncol + c) * nsub + g] = ((mask[((size_t)g * nw + (l >> 6)) * ncol + c] >> (l & 63)) & 1ull) ? 1.0 : 0.0;
  return err;
}
