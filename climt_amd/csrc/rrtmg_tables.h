// rrtmg_tables.h -- host-side table store for librrtmg_hip.so.
//
// Reads the neutral data blob written by tools/pack_tables.py (raw 16-g k-distribution tables and
// the small reference tables) and performs the one-time construction the reference does in
// rrtmg_sw_ini (climt/_lib/rrtmg_sw/rrtmg_sw_init.f90:47-173, cmbgb16s..29 :492-1689) and
// rrtmg_lw_ini (climt/_lib/rrtmg_lw/rrtmg_lw_init.f90:28-175, cmbgb1..16 :366-2015):
//   * relative g-point weights rwgt, 224->112 / 256->140 g-point reduction of every table,
//   * transmittance / Pade lookup tables (exp_tbl, tau_tbl, tfn_tbl), heatfac.
// Layout of every reduced table in the flat array is [g][inner...] for k-like tables (g slowest,
// interpolation index fastest) and [j][g] for the (16,n) "g-first" tables -- i.e. exactly the
// Fortran storage order of the reference's reduced arrays, so one wavefront that works on one
// g-point gathers from one contiguous slice.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace rrtmg {

struct BlobEntry {
  int dtype = 0;  // 0 f64, 1 i32
  std::vector<uint32_t> dims;
  std::vector<double> f;
  std::vector<int32_t> i;
  size_t size() const { return dtype == 0 ? f.size() : i.size(); }
};

struct Blob {
  std::map<std::string, BlobEntry> e;
  bool load(const std::string &path, std::string &err);
  const BlobEntry *find(const std::string &name) const {
    auto it = e.find(name);
    return it == e.end() ? nullptr : &it->second;
  }
};

struct TableRef {
  long off = -1;  // offset (doubles) into flat
  long n = 0;
  std::vector<uint32_t> dims;
};

// A flat fp64 table arena + name registry (names like "sw/kg16/absa", "sw/tbl/exp_tbl").
struct TableSet {
  std::vector<double> flat;
  std::map<std::string, TableRef> reg;
  std::map<std::string, std::vector<int32_t>> ireg;
  double heatfac = 0.0;
  bool synthetic = false;
  long add(const std::string &name, const double *p, long n, const std::vector<uint32_t> &dims);
  long off(const std::string &name) const {
    auto it = reg.find(name);
    return it == reg.end() ? -1 : it->second.off;
  }
  const std::vector<int32_t> *ints(const std::string &name) const {
    auto it = ireg.find(name);
    return it == ireg.end() ? nullptr : &it->second;
  }
};

// which: "sw" or "lw".  grav/secdy enter heatfac exactly as in swdatinit/lwdatinit.
bool build_tables(const Blob &blob, const std::string &which, double cpdair, double grav, double secdy,
                  TableSet &out, std::string &err);

}  // namespace rrtmg
