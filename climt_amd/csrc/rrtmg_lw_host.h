// rrtmg_lw_host.h -- host-side construction of the longwave table descriptor.
#pragma once
#include <cmath>
#include <string>

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
    B.absa = off(p + "absa", true); B.absb = off(p + "absb", false);
    B.self = off(p + "selfref", true); B.forr = off(p + "forref", true);
    B.fraca = off(p + "fracrefa", true); B.fracb = off(p + "fracrefb", false);
    { auto it = ts.reg.find(p + "fracrefa"); B.nfraca = it != ts.reg.end() && it->second.dims.size() > 1 ? (int)it->second.dims[1] : 1; }
    { auto it = ts.reg.find(p + "fracrefb"); B.nfracb = it != ts.reg.end() && it->second.dims.size() > 1 ? (int)it->second.dims[1] : 1; }
    for (int k = 0; k < 3; ++k) B.ma[k] = MA[b][k] ? off(p + MA[b][k], true) : 0;
    for (int k = 0; k < 2; ++k) B.mb[k] = MB[b][k] ? off(p + MB[b][k], true) : 0;
    for (int k = 0; k < 2; ++k) B.x[k] = X[b][k] ? off(p + X[b][k], true) : 0;
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
  return err.empty();
}

}  // namespace rrtmg
