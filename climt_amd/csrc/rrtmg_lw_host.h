// rrtmg_lw_host.h -- host-side construction of the longwave table descriptor.
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "rrtmg_lw_device.h"
#include "rrtmg_tables.h"

namespace rrtmg {

inline bool build_lw_tab(TableSet &ts, LwTab &T, std::string &err) {
  const std::vector<int32_t> *ngc = ts.ints("lw/wvn/ngc"), *ngs = ts.ints("lw/wvn/ngs");
  if (!ngc || !ngs) { err = "lw/wvn/ngc missing"; return false; }
  auto off = [&](const std::string &n, bool required) -> long {
    long o = ts.off(n);
    if (o < 0 && required) err = "reduced table '" + n + "' missing";
    return o < 0 ? 0 : o;
  };
  // minor-gas / cross-section tables by band, in the slot order lw_taug<> expects
  static const char *MA[16][3] = {{"ka_mn2", 0, 0}, {0, 0, 0}, {"ka_mn2o", 0, 0}, {0, 0, 0}, {"ka_mo3", 0, 0}, {"ka_mco2", 0, 0},
                                  {"ka_mco2", 0, 0}, {"ka_mco2", "ka_mo3", "ka_mn2o"}, {"ka_mn2o", 0, 0}, {0, 0, 0},
                                  {"ka_mo2", 0, 0}, {0, 0, 0}, {"ka_mco2", "ka_mco", 0}, {0, 0, 0}, {"ka_mn2", 0, 0}, {0, 0, 0}};
  static const char *MB[16][2] = {{"kb_mn2", 0}, {0, 0}, {"kb_mn2o", 0}, {0, 0}, {0, 0}, {0, 0}, {"kb_mco2", 0}, {"kb_mco2", "kb_mn2o"},
                                  {"kb_mn2o", 0}, {0, 0}, {"kb_mo2", 0}, {0, 0}, {"kb_mo3", 0}, {0, 0}, {0, 0}, {0, 0}};
  static const char *X[16][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {"ccl4", 0}, {"cfc11adj", "cfc12"}, {0, 0}, {"cfc12", "cfc22adj"},
                                 {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
  for (int b = 0; b < kLwNBand; ++b) {
    LwBandTab &B = T.b[b];
    char buf[32];
    snprintf(buf, sizeof buf, "lw/kg%02d/", b + 1);
    const std::string p = buf;
    B.ng = (*ngc)[b];
    B.gs = b == 0 ? 0 : (*ngs)[b - 1];
    if (B.ng != kLwNg[b]) { err = "reduced g-point count of band " + std::to_string(b + 1) + " differs from the compiled-in one"; return false; }
    // ONE g-point-fastest slab [nrows][ng] per band holding all its per-g-point tables (the blob stores them [ng][row],
    // the Planck-fraction and cross-section tables [row][ng] already); LwBandTab::r_* = first row of each table
    std::vector<double> slab;
    auto append = [&](const std::string &n, bool required, bool transposed) -> int {
      auto it = ts.reg.find(p + n);
      if (it == ts.reg.end()) { if (required) err = "reduced table '" + p + n + "' missing"; return 0; }
      const long o = it->second.off, rows = it->second.n / B.ng;
      const int r0 = (int)(slab.size() / B.ng);
      slab.resize(slab.size() + (size_t)rows * B.ng);
      for (int ig = 0; ig < B.ng; ++ig)
        for (long r = 0; r < rows; ++r)
          slab[(size_t)(r0 + r) * B.ng + ig] = ts.flat[(size_t)o + (transposed ? (size_t)ig * rows + r : (size_t)r * B.ng + ig)];
      return r0;
    };
    B.r_absa = append("absa", true, true); B.r_absb = append("absb", false, true);
    B.r_self = append("selfref", true, true); B.r_forr = append("forref", true, true);
    B.r_fraca = append("fracrefa", true, false); B.r_fracb = append("fracrefb", false, false);
    { auto it = ts.reg.find(p + "fracrefa"); B.nfraca = it != ts.reg.end() && it->second.dims.size() > 1 ? (int)it->second.dims[1] : 1; }
    { auto it = ts.reg.find(p + "fracrefb"); B.nfracb = it != ts.reg.end() && it->second.dims.size() > 1 ? (int)it->second.dims[1] : 1; }
    for (int k = 0; k < 3; ++k) B.r_ma[k] = MA[b][k] ? append(MA[b][k], true, true) : 0;
    for (int k = 0; k < 2; ++k) B.r_mb[k] = MB[b][k] ? append(MB[b][k], true, true) : 0;
    for (int k = 0; k < 2; ++k) B.r_x[k] = X[b][k] ? append(X[b][k], true, false) : 0;
    if (!err.empty()) return false;
    B.nrows = (int)(slab.size() / B.ng);
    if (B.nrows > kLwSlabMaxRows) { err = "band " + std::to_string(b + 1) + " table slab has more rows than kLwSlabMaxRows"; return false; }
    B.slab = ts.add(p + "slab_g", slab.data(), (long)slab.size(), {(uint32_t)B.nrows, (uint32_t)B.ng});
    if (!err.empty()) return false;
  }
  T.preflog = off("lw/ref/preflog", true); T.tref = off("lw/ref/tref", true); T.chi_mls = off("lw/ref/chi_mls", true);
  T.totplnk = off("lw/wvn/totplnk", true); T.totplk16 = off("lw/wvn/totplk16", true);
  T.totplnkderiv = off("lw/wvn/totplnkderiv", true); T.totplk16deriv = off("lw/wvn/totplk16deriv", true);
  T.exp_tbl = off("lw/tbl/exp_tbl", true); T.tau_tbl = off("lw/tbl/tau_tbl", true); T.tfn_tbl = off("lw/tbl/tfn_tbl", true);
  T.delwave = off("lw/wvn/delwave", true);
  T.abscld1 = off("lw/cld/abscld1", true); T.absice0 = off("lw/cld/absice0", true); T.absice1 = off("lw/cld/absice1", true);
  T.absice2 = off("lw/cld/absice2", true); T.absice3 = off("lw/cld/absice3", true); T.absliq0 = off("lw/cld/absliq0", true);
  T.absliq1 = off("lw/cld/absliq1", true);
  T.heatfac = ts.heatfac;
  // chi_mls(x, j)/chi_mls(y, j) of the binary-species bands (rrtmg_lw_setcoef.f90 rat_* quotients), rows CR_*
  {
    const int pair[CR_N][2] = {{1, 2}, {3, 2}, {1, 3}, {1, 6}, {1, 4}, {4, 2}};
    std::vector<double> cr((size_t)CR_N * 59);
    for (int q = 0; q < CR_N; ++q)
      for (int j = 1; j <= 59; ++j)
        cr[(size_t)q * 59 + (j - 1)] = ts.flat[(size_t)T.chi_mls + (pair[q][0] - 1) + 7 * (j - 1)] / ts.flat[(size_t)T.chi_mls + (pair[q][1] - 1) + 7 * (j - 1)];
    T.chirat = ts.add("lw/ref/chirat", cr.data(), (long)cr.size(), {59u, (uint32_t)CR_N});
  }
  // work items: chunks of 4 (then 2) consecutive g-points of a band; launch order heaviest first
  T.nitem = 0;
  double cost[kLwMaxItem];
  const int nspa[kLwNBand] = {1, 1, 9, 9, 9, 1, 9, 1, 9, 1, 1, 9, 9, 1, 9, 9};
  for (int b = 0; b < kLwNBand; ++b) {
    int ig = 0;
    while (ig < T.b[b].ng) {
      // (chunks of 8 for the bands without a binary species mixture were measured slower: DESIGN.md 5)
      const int g = T.b[b].ng - ig >= 4 ? 4 : 2;
      if ((long)T.b[b].nrows * g > (long)kLwSlabMaxRows * 4) { err = "work item slice does not fit the LDS buffer"; return false; }
      if (T.nitem >= kLwMaxItem) { err = "too many work items"; return false; }
      cost[T.nitem] = (nspa[b] == 9 ? 2.0 : 1.0) + g * 0.7;
      T.item[T.nitem] = b | (ig << 8) | (g << 16) | ((T.b[b].gs + ig) << 20);
      T.sched[T.nitem] = T.nitem;
      ++T.nitem;
      ig += g;
    }
  }
  for (int i = 1; i < T.nitem; ++i)   // stable insertion sort, descending cost
    for (int j = i; j > 0 && cost[T.sched[j]] > cost[T.sched[j - 1]]; --j) { const int t = T.sched[j]; T.sched[j] = T.sched[j - 1]; T.sched[j - 1] = t; }
  return err.empty();
}

}  // namespace rrtmg
