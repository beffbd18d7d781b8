/* TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference RRTMG algorithm, plain C.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.  The product
 * (climt_amd) never links, loads or imports anything under oracle/.
 *
 * This header: table store.  Reads the neutral data blob (tools/pack_tables.py) and restates the reference's
 * init (rrtmg_sw_init.f90:47-173 + cmbgb16s..29 :492-1689; rrtmg_lw_init.f90:28-175 + cmbgb1..16 :366-2015):
 * relative g-point weights, 16-g -> reduced-g combination, exponential / Pade lookup tables.
 * Pinned by tests/test_oracle.py against the reference's own post-init tables (tests/golden/{sw,lw}_reduced_tables.npz)
 * and, for the whole path, against reference-Fortran outputs (tests/golden/ref_*.npz, oracle/_ref when present).
 */
#ifndef ORACLE_COMMON_H
#define ORACLE_COMMON_H
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define OR_MAXENT 512
#define OR_MAXL 260

typedef struct {
  char name[64];
  int dtype; /* 0 f64, 1 i32 */
  int nd;
  uint32_t dims[8];
  long n;
  double *f;
  int32_t *i;
} or_entry;

typedef struct {
  or_entry e[OR_MAXENT];
  int n;
} or_store;

static or_entry *or_find(or_store *s, const char *name) {
  for (int k = 0; k < s->n; ++k)
    if (strcmp(s->e[k].name, name) == 0) return &s->e[k];
  return NULL;
}
static double *or_f(or_store *s, const char *name) {
  or_entry *e = or_find(s, name);
  if (!e || e->dtype != 0) {
    fprintf(stderr, "oracle: missing table %s\n", name);
    abort();
  }
  return e->f;
}
static int32_t *or_i(or_store *s, const char *name) {
  or_entry *e = or_find(s, name);
  if (!e || e->dtype != 1) {
    fprintf(stderr, "oracle: missing int table %s\n", name);
    abort();
  }
  return e->i;
}
static or_entry *or_add(or_store *s, const char *name, int dtype, int nd, const uint32_t *dims, long n) {
  or_entry *e = &s->e[s->n++];
  memset(e, 0, sizeof *e);
  strncpy(e->name, name, 63);
  e->dtype = dtype;
  e->nd = nd;
  for (int k = 0; k < nd; ++k) e->dims[k] = dims[k];
  e->n = n;
  if (dtype == 0) e->f = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
  else e->i = (int32_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int32_t));
  return e;
}

static int or_load_blob(or_store *s, const char *path) {
  FILE *fp = fopen(path, "rb");
  char magic[8];
  uint32_t count;
  if (!fp) return -1;
  if (fread(magic, 1, 8, fp) != 8 || memcmp(magic, "RRTBL001", 8) || fread(&count, 4, 1, fp) != 1) { fclose(fp); return -2; }
  for (uint32_t k = 0; k < count; ++k) {
    uint32_t len, code, nd, dims[8];
    char name[256];
    uint64_t nbytes;
    if (fread(&len, 4, 1, fp) != 1 || len > 255) { fclose(fp); return -3; }
    if (fread(name, 1, len, fp) != len) { fclose(fp); return -3; }
    name[len] = 0;
    if (fread(&code, 4, 1, fp) != 1 || fread(&nd, 4, 1, fp) != 1 || nd > 8) { fclose(fp); return -3; }
    if (nd && fread(dims, 4, nd, fp) != nd) { fclose(fp); return -3; }
    if (fread(&nbytes, 8, 1, fp) != 1) { fclose(fp); return -3; }
    long n = (long)(nbytes / (code == 0 ? 8 : 4));
    or_entry *e = or_add(s, name, (int)code, (int)nd, dims, n);
    if (nbytes && fread(code == 0 ? (void *)e->f : (void *)e->i, 1, nbytes, fp) != nbytes) { fclose(fp); return -3; }
    long pos = ftell(fp), pad = (8 - (long)(nbytes % 8)) % 8;
    pos += pad;
    pad += (8 - (pos % 8)) % 8;
    if (pad) fseek(fp, pad, SEEK_CUR);
  }
  fclose(fp);
  return 0;
}

/* g-point combination of every raw table of bands [band0, band0+nbnd): absorption-like tables are weighted by
 * rwgt (g is the LAST dimension), source-like tables (solar source terms, Planck fractions; g FIRST) are summed. */
static void or_reduce(or_store *s, const char *pfx, int nbnd, int band0) {
  char nm[96];
  snprintf(nm, sizeof nm, "%s/wvn/ngc", pfx); int32_t *ngc = or_i(s, nm);
  snprintf(nm, sizeof nm, "%s/wvn/ngn", pfx); int32_t *ngn = or_i(s, nm);
  snprintf(nm, sizeof nm, "%s/wvn/ngm", pfx); int32_t *ngm = or_i(s, nm);
  snprintf(nm, sizeof nm, "%s/wvn/ngs", pfx); int32_t *ngs = or_i(s, nm);
  snprintf(nm, sizeof nm, "%s/wvn/wt", pfx); double *wt = or_f(s, nm);
  uint32_t d1 = (uint32_t)(nbnd * 16);
  snprintf(nm, sizeof nm, "%s/wvn/rwgt", pfx);
  double *rwgt = or_add(s, nm, 0, 1, &d1, nbnd * 16)->f;
  int igcsm = 0;
  for (int ib = 0; ib < nbnd; ++ib) {
    int iprsm = 0;
    if (ngc[ib] < 16) {
      double wtsm[16];
      for (int igc = 0; igc < ngc[ib]; ++igc) {
        double wtsum = 0.0;
        for (int ipr = 0; ipr < ngn[igcsm]; ++ipr) wtsum = wtsum + wt[iprsm++];
        igcsm++;
        wtsm[igc] = wtsum;
      }
      for (int ig = 0; ig < 16; ++ig) rwgt[ib * 16 + ig] = wt[ig] / wtsm[ngm[ib * 16 + ig] - 1];
    } else {
      for (int ig = 0; ig < 16; ++ig) { igcsm++; rwgt[ib * 16 + ig] = 1.0; }
    }
  }
  int nraw = s->n;
  for (int k = 0; k < nraw; ++k) {
    or_entry *e = &s->e[k];
    size_t lp = strlen(pfx);
    if (strncmp(e->name, pfx, lp) || strncmp(e->name + lp, "/kg", 3) || e->dtype != 0) continue;
    int band = atoi(e->name + lp + 3);
    int ib = band - band0;
    const char *leaf = strrchr(e->name, '/') + 1;
    int hasg = 0;
    for (int q = 0; q < e->nd; ++q) hasg |= (e->dims[q] == 16);
    size_t ll = strlen(leaf);
    int rawlike = (ll > 0 && leaf[ll - 1] == 'o') || strstr(leaf, "o_m") != NULL;
    if (!hasg || !rawlike) continue;
    int src = !strcmp(leaf, "sfluxrefo") || !strcmp(leaf, "irradnceo") || !strcmp(leaf, "facbrghto") || !strcmp(leaf, "snsptdrko") ||
              !strcmp(leaf, "fracrefao") || !strcmp(leaf, "fracrefbo");
    int gfirst = (e->nd == 1) || src || !strcmp(leaf, "raylao");
    char red[96], base[64];
    if (!strcmp(leaf, "kao")) strcpy(base, "absa");
    else if (!strcmp(leaf, "kbo")) strcpy(base, "absb");
    else if (!strncmp(leaf, "kao_", 4)) snprintf(base, sizeof base, "ka_%s", leaf + 4);
    else if (!strncmp(leaf, "kbo_", 4)) snprintf(base, sizeof base, "kb_%s", leaf + 4);
    else { strcpy(base, leaf); base[ll - 1] = 0; }
    snprintf(red, sizeof red, "%s/kg%02d/%s", pfx, band, base);
    int ng = ngc[ib], g0 = ib == 0 ? 0 : ngs[ib - 1];
    long inner = e->n / 16;
    uint32_t rd[8];
    for (int q = 0; q < e->nd; ++q) rd[q] = e->dims[q];
    if (gfirst) rd[0] = (uint32_t)ng; else rd[e->nd - 1] = (uint32_t)ng;
    double *raw = e->f;
    double *out = or_add(s, red, 0, e->nd, rd, inner * ng)->f;
    e = &s->e[k];
    for (long j = 0; j < inner; ++j) {
      int iprsm = 0;
      for (int igc = 0; igc < ng; ++igc) {
        double sum = 0.0;
        for (int ipr = 0; ipr < ngn[g0 + igc]; ++ipr, ++iprsm) {
          double v = gfirst ? raw[iprsm + 16 * j] : raw[j + inner * iprsm];
          sum = sum + (src ? v : v * rwgt[ib * 16 + iprsm]);
        }
        if (gfirst) out[igc + (long)ng * j] = sum; else out[j + inner * igc] = sum;
      }
    }
  }
}

/* exp / tau / tfn lookup tables (rrtmg_sw_init.f90:113-123; rrtmg_lw_init.f90:103-123) */
static void or_lookup_tables(int lw, double *exp_tbl, double *tau_tbl, double *tfn_tbl) {
  const int ntbl = 10000;
  const double bpade = 1.0 / 0.278, expeps = 1.e-20;
  exp_tbl[0] = 1.0; exp_tbl[ntbl] = expeps;
  tau_tbl[0] = 0.0; tau_tbl[ntbl] = 1.e10;
  tfn_tbl[0] = 0.0; tfn_tbl[ntbl] = 1.0;
  for (int itr = 1; itr < ntbl; ++itr) {
    double tfn = lw ? (double)((float)itr / (float)ntbl) : (double)itr / (double)ntbl;
    tau_tbl[itr] = bpade * tfn / (1.0 - tfn);
    exp_tbl[itr] = exp(-tau_tbl[itr]);
    if (exp_tbl[itr] <= expeps) exp_tbl[itr] = expeps;
    if (tau_tbl[itr] < 0.06) tfn_tbl[itr] = tau_tbl[itr] / 6.0;
    else tfn_tbl[itr] = 1.0 - 2.0 * ((1.0 / tau_tbl[itr]) - (exp_tbl[itr] / (1.0 - exp_tbl[itr])));
  }
}

/* kissvec for one column (mcica_subcol_gen_sw.f90:557-591) and MT19937 (mcica_random_numbers.f90:77-302) */
typedef struct { int32_t s1, s2, s3, s4; } or_kiss;
static double or_kiss_next(or_kiss *k) {
  uint32_t a = (uint32_t)k->s1, b = (uint32_t)k->s2, c = (uint32_t)k->s3, d = (uint32_t)k->s4;
  a = 69069u * a + 1327217885u;
  b ^= b << 13; b ^= b >> 17; b ^= b << 5;
  c = 18000u * (c & 65535u) + (c >> 16);
  d = 30903u * (d & 65535u) + (d >> 16);
  k->s1 = (int32_t)a; k->s2 = (int32_t)b; k->s3 = (int32_t)c; k->s4 = (int32_t)d;
  int32_t kiss = (int32_t)(a + b + (c << 16) + d);
  volatile double r = (double)kiss * 2.328306e-10;
  return r + 0.5;
}
typedef struct { uint32_t st[624]; int cur; } or_mt;
static void or_mt_init(or_mt *m, int32_t seed) {
  m->st[0] = (uint32_t)seed;
  for (int i = 1; i < 624; ++i) m->st[i] = 1812433253u * (m->st[i - 1] ^ (m->st[i - 1] >> 30)) + (uint32_t)i;
  m->cur = 624;
}
static uint32_t or_mt_twist(uint32_t u, uint32_t v) {
  uint32_t mix = (u & 0x80000000u) | (v & 0x7fffffffu);
  return (mix >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
static double or_mt_real(or_mt *m) {
  if (m->cur >= 624) {
    for (int k = 0; k < 227; ++k) m->st[k] = m->st[k + 397] ^ or_mt_twist(m->st[k], m->st[k + 1]);
    for (int k = 227; k < 623; ++k) m->st[k] = m->st[k - 227] ^ or_mt_twist(m->st[k], m->st[k + 1]);
    m->st[623] = m->st[396] ^ or_mt_twist(m->st[623], m->st[0]);
    m->cur = 0;
  }
  uint32_t y = m->st[m->cur++];
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  int32_t li = (int32_t)y;
  /* (localInt + 2.0**32_rb) is single-precision arithmetic in the reference (mcica_random_numbers.f90:288-292) */
  if (li < 0) { volatile float f = (float)li; f = f + 4294967296.0f; return (double)f / 4294967295.0; }
  return (double)li / 4294967295.0;
}

/* generate_stochastic_clouds: cldf -> cloudy flag per (sub-column, column, layer), Fortran order (nsub, ncol, nlay)
 * (mcica_subcol_gen_sw.f90:316-470; the LW generator mcica_subcol_gen_lw.f90:296-440 is the same algorithm) */
static int or_subcol_mask(int ncol, int nlay, int nsub, int icld, int irng, int seed, const double *play, const double *cldfr,
                          unsigned char *cloudy) {
  const double cldmin = 1.0e-20;
  memset(cloudy, 0, (size_t)nsub * ncol * nlay);
  if (icld == 0) return 0;
  double *cdf = (double *)malloc(sizeof(double) * (size_t)nsub * ncol * nlay);
#define CDF(is, ic, il) cdf[(is) + (size_t)nsub * ((ic) + (size_t)ncol * (il))]
  if (irng == 0) {
    for (int c = 0; c < ncol; ++c) {
      or_kiss k;
      double p[4];
      for (int q = 0; q < 4; ++q) { volatile double pm = play[(size_t)q * ncol + c] * 1.e2; p[q] = pm; }
      if (p[0] < p[1]) { free(cdf); return 14; }
      k.s1 = (int32_t)((p[0] - (double)(int)p[0]) * 1000000000.0);
      k.s2 = (int32_t)((p[1] - (double)(int)p[1]) * 1000000000.0);
      k.s3 = (int32_t)((p[2] - (double)(int)p[2]) * 1000000000.0);
      k.s4 = (int32_t)((p[3] - (double)(int)p[3]) * 1000000000.0);
      for (int i = 0; i < seed; ++i) (void)or_kiss_next(&k);
      for (int is = 0; is < nsub; ++is) {
        if (icld == 3) {
          double r = or_kiss_next(&k);
          for (int l = 0; l < nlay; ++l) CDF(is, c, l) = r;
        } else {
          for (int l = 0; l < nlay; ++l) CDF(is, c, l) = or_kiss_next(&k);
        }
      }
    }
  } else {
    or_mt m;
    or_mt_init(&m, seed);
    for (int is = 0; is < nsub; ++is)
      for (int c = 0; c < ncol; ++c) {
        if (icld == 3) {
          double r = or_mt_real(&m);
          for (int l = 0; l < nlay; ++l) CDF(is, c, l) = r;
        } else {
          for (int l = 0; l < nlay; ++l) CDF(is, c, l) = or_mt_real(&m);
        }
      }
  }
  if (icld == 2) {
    for (int l = 1; l < nlay; ++l)
      for (int c = 0; c < ncol; ++c) {
        double cfm = cldfr[(size_t)(l - 1) * ncol + c];
        if (cfm < cldmin) cfm = 0.0;
        for (int is = 0; is < nsub; ++is) {
          if (CDF(is, c, l - 1) > 1.0 - cfm) CDF(is, c, l) = CDF(is, c, l - 1);
          else CDF(is, c, l) = CDF(is, c, l) * (1.0 - cfm);
        }
      }
  }
  for (int l = 0; l < nlay; ++l)
    for (int c = 0; c < ncol; ++c) {
      double cf = cldfr[(size_t)l * ncol + c];
      if (cf < cldmin) cf = 0.0;
      for (int is = 0; is < nsub; ++is) cloudy[is + (size_t)nsub * (c + (size_t)ncol * l)] = CDF(is, c, l) >= 1.0 - cf;
    }
#undef CDF
  free(cdf);
  return 0;
}
#endif
