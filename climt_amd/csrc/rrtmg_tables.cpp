// rrtmg_tables.cpp -- blob reader and init-time table construction (see rrtmg_tables.h).
#include "rrtmg_tables.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace rrtmg {

bool Blob::load(const std::string &path, std::string &err) {
  FILE *fp = fopen(path.c_str(), "rb");
  if (!fp) {
    err = "cannot open table blob '" + path + "'";
    return false;
  }
  auto rd = [&](void *p, size_t n) { return fread(p, 1, n, fp) == n; };
  char magic[8];
  uint32_t count = 0;
  if (!rd(magic, 8) || memcmp(magic, "RRTBL001", 8) != 0 || !rd(&count, 4)) {
    err = "bad table blob header in '" + path + "'";
    fclose(fp);
    return false;
  }
  for (uint32_t k = 0; k < count; ++k) {
    uint32_t len = 0, code = 0, nd = 0;
    if (!rd(&len, 4) || len > 4096) goto bad;
    {
      std::string name(len, '\0');
      if (!rd(&name[0], len) || !rd(&code, 4) || !rd(&nd, 4) || nd > 8) goto bad;
      BlobEntry en;
      en.dtype = (int)code;
      en.dims.resize(nd);
      if (nd && !rd(en.dims.data(), 4 * nd)) goto bad;
      uint64_t nbytes = 0;
      if (!rd(&nbytes, 8)) goto bad;
      if (code == 0) {
        en.f.resize(nbytes / 8);
        if (nbytes && !rd(en.f.data(), nbytes)) goto bad;
      } else {
        en.i.resize(nbytes / 4);
        if (nbytes && !rd(en.i.data(), nbytes)) goto bad;
      }
      long pos = ftell(fp);
      long pad = (8 - (nbytes % 8)) % 8;
      pos += pad;
      pad += (8 - (pos % 8)) % 8;
      if (pad) fseek(fp, pad, SEEK_CUR);
      e[name] = std::move(en);
    }
  }
  fclose(fp);
  return true;
bad:
  err = "truncated/malformed table blob '" + path + "'";
  fclose(fp);
  return false;
}

long TableSet::add(const std::string &name, const double *p, long n, const std::vector<uint32_t> &dims) {
  // keep every table 16-byte aligned in the arena
  if (flat.size() & 1) flat.push_back(0.0);
  long o = (long)flat.size();
  flat.insert(flat.end(), p, p + n);
  TableRef r;
  r.off = o;
  r.n = n;
  r.dims = dims;
  reg[name] = r;
  return o;
}

static bool ends_with(const std::string &s, const std::string &suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

// names of raw tables whose FIRST dimension is the 16 original g-points and which are combined by
// plain summation (solar source terms, Planck fractions); everything else in a kg module is an
// absorption-like table with g LAST, combined with the relative weights rwgt.
static bool is_source_like(const std::string &leaf) {
  return leaf == "sfluxrefo" || leaf == "irradnceo" || leaf == "facbrghto" || leaf == "snsptdrko" ||
         leaf == "fracrefao" || leaf == "fracrefbo";
}
static bool is_gfirst(const std::string &leaf, const std::vector<uint32_t> &dims) {
  if (dims.size() == 1) return true;
  return is_source_like(leaf) || leaf == "raylao";
}

static std::string reduced_name(const std::string &leaf) {
  if (leaf == "kao") return "absa";
  if (leaf == "kbo") return "absb";
  if (leaf.compare(0, 4, "kao_") == 0) return "ka_" + leaf.substr(4);
  if (leaf.compare(0, 4, "kbo_") == 0) return "kb_" + leaf.substr(4);
  return leaf.substr(0, leaf.size() - 1);  // strip trailing 'o'
}

bool build_tables(const Blob &blob, const std::string &which, double cpdair, double grav, double secdy,
                  TableSet &out, std::string &err) {
  const bool sw = (which == "sw");
  const int nbnd = sw ? 14 : 16;
  const int band0 = sw ? 16 : 1;
  const std::string P = which + "/";
  auto need = [&](const std::string &n) -> const BlobEntry * {
    const BlobEntry *x = blob.find(P + n);
    if (!x) err = "table blob lacks entry '" + P + n + "'";
    return x;
  };
  const BlobEntry *ngc = need("wvn/ngc"), *ngn = need("wvn/ngn"), *ngm = need("wvn/ngm"),
                  *wt = need("wvn/wt"), *ngs = need("wvn/ngs");
  if (!ngc || !ngn || !ngm || !wt || !ngs) return false;
  const int mg = 16;

  // ---- rwgt (rrtmg_sw_init.f90:130-153 / rrtmg_lw_init.f90:130-154) ----------------------
  std::vector<double> rwgt((size_t)nbnd * mg, 0.0);
  {
    int igcsm = 0;
    for (int ib = 0; ib < nbnd; ++ib) {
      int iprsm = 0;
      if (ngc->i[ib] < mg) {
        double wtsm[16];
        for (int igc = 0; igc < ngc->i[ib]; ++igc) {
          double wtsum = 0.0;
          for (int ipr = 0; ipr < ngn->i[igcsm]; ++ipr) wtsum = wtsum + wt->f[iprsm++];
          ++igcsm;
          wtsm[igc] = wtsum;
        }
        for (int ig = 0; ig < mg; ++ig) {
          int ind = ib * mg + ig;
          rwgt[ind] = wt->f[ig] / wtsm[ngm->i[ind] - 1];
        }
      } else {
        for (int ig = 0; ig < mg; ++ig) {
          ++igcsm;
          rwgt[ib * mg + ig] = 1.0;
        }
      }
    }
  }
  out.add(P + "wvn/rwgt", rwgt.data(), (long)rwgt.size(), {(uint32_t)rwgt.size()});

  // ---- g-point reduction of every raw kg table ---------------------------------------------
  for (const auto &kv : blob.e) {
    const std::string &full = kv.first;
    if (full.compare(0, P.size() + 2, P + "kg") != 0) continue;
    const BlobEntry &en = kv.second;
    size_t slash = full.rfind('/');
    std::string mod = full.substr(P.size(), slash - P.size());  // "kg16"
    std::string leaf = full.substr(slash + 1);
    int band = atoi(mod.c_str() + 2);
    int ib = band - band0;
    if (ib < 0 || ib >= nbnd) continue;
    if (en.dtype != 0) continue;
    bool has_g = false;
    for (uint32_t d : en.dims) has_g |= (d == 16);
    if (!has_g || !ends_with(leaf, "o") && leaf.find("o_") == std::string::npos) {
      // scalar / non-g table (rayl, refparam ...): copy through
      out.add(P + mod + "/" + leaf, en.f.data(), (long)en.f.size(), en.dims);
      continue;
    }
    const int ng = ngc->i[ib];
    const int g0 = (ib == 0) ? 0 : ngs->i[ib - 1];  // first reduced g-point of the band (0-based)
    const bool gfirst = is_gfirst(leaf, en.dims);
    const bool weighted = !is_source_like(leaf);
    long total = (long)en.f.size();
    long inner = total / 16;  // elements per original g-point
    std::vector<double> red((size_t)inner * ng, 0.0);
    std::vector<uint32_t> rdims = en.dims;
    if (gfirst) {
      rdims[0] = (uint32_t)ng;
      // raw(g, j) flat g + 16 j   ->   red(igc, j) flat igc + ng j
      for (long j = 0; j < inner; ++j) {
        int iprsm = 0;
        for (int igc = 0; igc < ng; ++igc) {
          double s = 0.0;
          for (int ipr = 0; ipr < ngn->i[g0 + igc]; ++ipr, ++iprsm) {
            double v = en.f[(size_t)iprsm + 16 * j];
            s = s + (weighted ? v * rwgt[ib * mg + iprsm] : v);
          }
          red[(size_t)igc + (size_t)ng * j] = s;
        }
      }
    } else {
      rdims.back() = (uint32_t)ng;
      // raw(i, g) flat i + inner g   ->   red(i, igc) flat i + inner igc
      for (long i = 0; i < inner; ++i) {
        int iprsm = 0;
        for (int igc = 0; igc < ng; ++igc) {
          double s = 0.0;
          for (int ipr = 0; ipr < ngn->i[g0 + igc]; ++ipr, ++iprsm) {
            double v = en.f[(size_t)i + (size_t)inner * iprsm];
            s = s + (weighted ? v * rwgt[ib * mg + iprsm] : v);
          }
          red[(size_t)i + (size_t)inner * igc] = s;
        }
      }
    }
    out.add(P + mod + "/" + reduced_name(leaf), red.data(), (long)red.size(), rdims);
  }

  // ---- small tables copied through ---------------------------------------------------------
  for (const auto &kv : blob.e) {
    const std::string &full = kv.first;
    if (full.compare(0, P.size(), P) != 0) continue;
    if (full.compare(0, P.size() + 2, P + "kg") == 0) continue;
    const BlobEntry &en = kv.second;
    if (en.dtype == 0)
      out.add(full, en.f.data(), (long)en.f.size(), en.dims);
    else
      out.ireg[full] = en.i;
  }
  if (const BlobEntry *syn = blob.find(P + "meta/synthetic")) out.synthetic = !syn->i.empty() && syn->i[0] != 0;

  // ---- lookup tables -------------------------------------------------------------------------
  const int ntbl = 10000;
  const double pade = 0.278, bpade = 1.0 / pade, expeps = 1.e-20;
  std::vector<double> exp_tbl(ntbl + 1), tau_tbl(ntbl + 1), tfn_tbl(ntbl + 1);
  exp_tbl[0] = 1.0;
  exp_tbl[ntbl] = expeps;
  tau_tbl[0] = 0.0;
  tau_tbl[ntbl] = 1.e10;
  tfn_tbl[0] = 0.0;
  tfn_tbl[ntbl] = 1.0;
  for (int itr = 1; itr < ntbl; ++itr) {
    // SW: real(itr,kind=rb)/real(ntbl,kind=rb) (rrtmg_sw_init.f90:118); LW: real(itr)/real(ntbl),
    // a single-precision quotient (rrtmg_lw_init.f90:112)
    double tfn = sw ? (double)itr / (double)ntbl : (double)((float)itr / (float)ntbl);
    tau_tbl[itr] = bpade * tfn / (1.0 - tfn);
    exp_tbl[itr] = exp(-tau_tbl[itr]);
    if (exp_tbl[itr] <= expeps) exp_tbl[itr] = expeps;
    if (tau_tbl[itr] < 0.06)
      tfn_tbl[itr] = tau_tbl[itr] / 6.0;
    else
      tfn_tbl[itr] = 1.0 - 2.0 * ((1.0 / tau_tbl[itr]) - (exp_tbl[itr] / (1.0 - exp_tbl[itr])));
  }
  out.add(P + "tbl/exp_tbl", exp_tbl.data(), ntbl + 1, {(uint32_t)ntbl + 1});
  if (!sw) {
    out.add(P + "tbl/tau_tbl", tau_tbl.data(), ntbl + 1, {(uint32_t)ntbl + 1});
    out.add(P + "tbl/tfn_tbl", tfn_tbl.data(), ntbl + 1, {(uint32_t)ntbl + 1});
  }
  out.heatfac = grav * secdy / (cpdair * 1.e2);
  return true;
}

}  // namespace rrtmg
