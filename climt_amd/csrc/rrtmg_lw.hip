// rrtmg_lw.hip -- longwave kernels and launch sequence (gfx950).
//
// Launch sequence of one rrtmg_hip_lw_fluxes call (all on the context's longwave stream):
//   kiss_mask_kernel / mask upload + lw_anymask_kernel (McICA, whole grid)
//   per column chunk (<= RRTMG_HIP_CHUNK_TILES tiles), so that a chunk's rows are still cached when its solve reads them:
//     lw_prep_fused_kernel <<<tiles, 16 waves>>>  inatm + setcoef per (column, layer), then the column part (laytrop,
//                          precipitable water -> secdiff, tile cloud flag) on what the layer part left in LDS; non-McICA cloudy
//                          tiles: cldprop and the rtrnmr overlap factors
//     lw_cloudmc_kernel    (McICA)                cldprmc band optics per (column, layer)
//     tile_lists_kernel    <<<1, 64>>>            the chunk's tiles by solve variant, compacted in tile order (LwDev::tlist)
//     lw_solve_all_kernel  one launch per variant (cloud-free / cloudy tiles): wavefront = tile(64 columns) x work item (4|2
//                          g-points of a band), workgroup = 4 tiles of one item sharing its k-distribution slice in LDS
//     lw_fluxheat_kernel   <<<(tiles, levels/15), 16 waves>>>  band / g-point integration per interface + heating rates
#include <future>

#include "rrtmg_ctx.h"
#include "rrtmg_lw_device.h"
#include "rrtmg_lw_host.h"
#include "rrtmg_mcica_kernels.h"
#include "rrtmg_sort.h"

namespace rrtmg {

// Preparation in ONE launch (see sw_prep_fused_kernel): phase 1 the layer part, layers strided over the 16 waves; phase 2
// wave 0: the column scan (laytrop, precipitable water -> diffusivity angles) on the rows just written, and the tile's
// cloud flag; phase 3, cloudy tiles only: cldprop / the rtrnmr overlap factors (one wave each, sequential in the layers as
// the reference); with McICA the cldprmc band optics stay a launch of their own (see sw_prep_fused_kernel).
constexpr int kPrepWaves = 16;
constexpr int kLwKeepLayers = 104;   // 104 x 3 x 64 doubles = 156 KB of the 160 KB a gfx950 workgroup can have
static_assert(kLwKeepLayers * 3 * 64 * sizeof(double) + 1024 <= 160 * 1024, "lw_prep_fused_kernel: LDS budget of gfx950");
__global__ void __launch_bounds__(64 * kPrepWaves) lw_prep_fused_kernel(LwDev d, LwTab T, int clouds, int maxrand, int keep_layers, int tile0) {
  const int tile = tile0 + blockIdx.x;   // (launched per column chunk, see sw_prep_fused_kernel)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = tile * 64 + lane;
  const bool act = col < d.ncol;
  __shared__ int sh_cld;
  // what the column scan reads back from the layer part -- [layer][coldry | h2o | lower flag][lane] -- in DYNAMIC LDS sized by
  // the launch for the grid's layer count (92 KB at 60 layers: a second workgroup, or a longwave solve workgroup, still fits
  // the CU); grids deeper than kLwKeepLayers get none (keep_layers = 0) and re-read the slab
  extern __shared__ __attribute__((aligned(16))) double sh_keep[];
  __shared__ int sh_any[kPrepWaves];
  const bool keep = keep_layers > 0;
  bool cld = false;
  if (act)
    for (int l = w; l < d.nlay; l += kPrepWaves) {
      lw_prep_layer(d, T, col, l, keep ? sh_keep + (3 * l) * 64 + lane : nullptr, 64);
      if (d.icld >= 1 && d.cldfr) cld = cld || d.cldfr[(long)l * d.ncol + col] > 0.0;
    }
  {
    const unsigned long long any = __ballot(cld);
    if (lane == 0) sh_any[w] = any != 0ull;
  }
  __syncthreads();
  if (w == 0) {
    if (act) lw_prep_column(d, T, col, keep ? sh_keep + lane : nullptr, 64);
    if (lane == 0) {
      int any = 0;
      for (int k = 0; k < kPrepWaves; ++k) any |= sh_any[k];
      d.tile_cld[tile] = any; sh_cld = any;
      if (any) atomicAdd(d.ncloudy, 1);
    }
  }
  if (!clouds) return;
  __syncthreads();
  if (!sh_cld || !act) return;
  if (w == 0) lw_cloud_column(d, T, col);
  if (w == kPrepWaves - 1 && maxrand) lw_mr_column(d, col);
}

__global__ void __launch_bounds__(64) lw_cloudmc_kernel(LwDev d, LwTab T, int tile0) {
  const int tile = tile0 + blockIdx.x;
  if (!d.tile_cld[tile]) return;
  const int col = tile * 64 + threadIdx.x;
  if (col < d.ncol) lw_cloudmc_layer(d, T, col, blockIdx.y);
}
__global__ void __launch_bounds__(64) lw_anymask_kernel(LwDev d) {
  const int col = blockIdx.x * 64 + threadIdx.x;
  if (col < d.ncol) lw_anymask_column(d, col);
}

// All 140 g-points in ONE launch.  Wavefront = 64 columns of one tile x one work item (4 or 2 consecutive g-points
// of a band, LwTab::item).  The thread carries the item's g-points through both sweeps: the layer state, the species
// mixtures (specparm/js/fs of the major, minor and Planck mixtures, adjusted columns), the Planck functions and the
// cloud optics -- more than half of the per-g-point work of rtrnmc+taumol -- are evaluated once per item.  The item's
// band-weighted radiances are summed in registers: part[item][k][level][column].
// Workgroup = kLwWgWaves wavefronts = the same item for kLwWgWaves consecutive tiles, sharing ONE copy of the item's
// k-distribution slice in LDS: columns ig0..ig0+G-1 of the band's table slab, [nrows][G], <= 66 KB, two workgroups per
// CU.  Every absorption-coefficient / Planck-fraction row a lane needs is then a 16/32-byte LDS read at a per-lane
// row (bank conflicts only) instead of a per-lane gather through the vector L1, whose return path (64 B/clk/CU) the
// ~30 row gathers per layer saturated: with the rows through the scalar cache (an ablation) the kernel ran 23 %
// faster, which bounded what staging could win.
// Launch order: tile groups of kLwTileGroup, within a group items heaviest first (LwTab::sched), tile blocks fastest
// -- the group's prep rows stay L2-resident while its items run.  Speed only, never correctness.
constexpr int kLwWgWaves = 4;
constexpr int kLwTileGroup = 32;
constexpr int kLwGroupBlocks = kLwTileGroup / kLwWgWaves;
static_assert(kLwTileGroup % kLwWgWaves == 0, "tile group must be a whole number of workgroups");
// Two variants are launched back to back (see sw_solve_all_kernel): CLD = false for the cloud-free tiles.
// MR = true: non-McICA maximum/random overlap (rtrnmr).
template <bool CLD, bool MR>
__global__ void __launch_bounds__(64 * kLwWgWaves) __attribute__((amdgpu_waves_per_eu(2))) lw_solve_all_kernel(LwDev d, LwTab T, int tile0, int ntile) {   // tiles tile0 .. tile0 + ntile - 1 (one column chunk)
  // this variant's tiles, compacted (LwDev::tlist): nblk workgroups of kLwWgWaves list entries have work; they are the FIRST
  // nblk x nitem of the dispatch order and dense in it (see sw_solve_all_kernel), in tile groups of kLwGroupBlocks workgroups
  // -- the last group holds the remaining ones
  const int nmine = d.tcnt[CLD ? 1 : 0];
  const int nblk = (nmine + kLwWgWaves - 1) / kLwWgWaves;
  const int q = blockIdx.x;
  if (q >= nblk * T.nitem) return;   // workgroup-uniform exit before the slice is staged
  const int per = kLwGroupBlocks * T.nitem, nfull = nblk / kLwGroupBlocks;
  const int bpg = q < nfull * per ? kLwGroupBlocks : nblk - nfull * kLwGroupBlocks, r = q < nfull * per ? q % per : q - nfull * per;
  const int grp = q < nfull * per ? q / per : nfull;
  const int k = r / bpg;
  const int first = grp * kLwTileGroup + (r % bpg) * kLwWgWaves;
  RRTMG_PROFILE_ONLY_ITEM(d, k)
  const int slot = T.sched[k], item = T.item[slot];
  const int g = (item >> 16) & 0xf, ig0 = (item >> 8) & 0xff;
  constexpr bool kLdsK = true;
  __shared__ __attribute__((aligned(16))) double sh_k[kLwSlabMaxRows * 4];   // rows are read 16 bytes at a time
  {
    const LwBandTab &B = T.b[item & 0xff];
    const double *src = T.t + B.slab + ig0;
    const int ng = B.ng, sh = g == 4 ? 2 : 1, n = B.nrows << sh;
    for (int i = threadIdx.x; i < n; i += 64 * kLwWgWaves) sh_k[i] = src[(long)(i >> sh) * ng + (i & (g - 1))];
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (first + wave >= nmine) return;
  const int ctile = d.tlist[(CLD ? d.tcap : 0) + first + wave], tile = tile0 + ctile;
  const int lane = threadIdx.x & 63;
  const int col = tile * 64 + lane;
  if (col >= d.ncol) return;
  double *scr = d.scratch + ((long)ctile * kLwNGpt + ((item >> 20) & 0xff)) * (long)LF_N * d.nlay * 64 + lane * 2;
  LwPartSink sink = lw_part_sink(d, slot, col);
  lw_solve_item<CLD, MR, kLdsK>(d, T, item, col, scr, 64, sink, sh_k);
}


// band integration AND heating rates in one launch (see sw_fluxheat_kernel)
constexpr int kFluxLev = 15;   // 16 waves per workgroup: the halo level is 1 in 16 of the partial-plane reads
__global__ void __launch_bounds__(64 * (kFluxLev + 1)) lw_fluxheat_kernel(LwDev d, LwTab T, int tile0) {
  // (the call's last launch leaves the preparation kernels' cloudy-tile count where the host will look for it, and clears it)
  if (d.hint_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { *d.hint_out = *d.ncloudy; *d.ncloudy = 0; }
  const int tile = tile0 + blockIdx.x, lane = threadIdx.x & 63, j = threadIdx.x >> 6;
  const int col = tile * 64 + lane, lev = blockIdx.y * kFluxLev + j;
  __shared__ double net[kFluxLev + 1][64], netc[kFluxLev + 1][64];
  const bool act = col < d.ncol && lev <= d.nlay;
  if (act) {
    double f[6];
    lw_flux_sums(d, col, lev, T.nitem, d.tile_cld[tile] != 0, f);
    if (j < kFluxLev || lev == d.nlay) {
      const long o = (long)lev * d.ncol + col;
      d.uflx[o] = f[0]; d.dflx[o] = f[1]; d.uflxc[o] = f[2]; d.dflxc[o] = f[3];
      if (d.idrv) { d.duflx_dt[o] = f[4]; d.duflxc_dt[o] = f[5]; }
    }
    net[j][lane] = f[0] - f[1]; netc[j][lane] = f[2] - f[3];
  }
  __syncthreads();
  if (col < d.ncol && j < kFluxLev && lev < d.nlay) {
    const long o0 = (long)lev * d.ncol + col;
    const double dp = d.plev[o0] - d.plev[o0 + d.ncol];
    d.hr[o0] = T.heatfac * (net[j][lane] - net[j + 1][lane]) / dp;
    d.hrc[o0] = T.heatfac * (netc[j][lane] - netc[j + 1][lane]) / dp;
  }
}

void free_lw_desc(rrtmg_ctx *ctx) {
  delete (LwTab *)ctx->lw_desc;
  ctx->lw_desc = nullptr;
}

int lw_init_impl(rrtmg_ctx *ctx, double cpdair, const char *blob_path) {
  if (!ctx->have_constants) return ctx->fail(RRTMG_ERR_NOT_INITIALISED, "set_constants must be called before lw_init");
  std::string path = blob_path ? std::string(blob_path) : default_blob_path("lw");
  Blob blob;
  std::string err;
  if (!blob.load(path, err)) return ctx->fail(RRTMG_ERR_TABLES, "%s", err.c_str());
  ctx->lw_ts = TableSet();
  if (!build_tables(blob, "lw", cpdair, ctx->k.grav, ctx->k.secdy, ctx->lw_ts, err)) return ctx->fail(RRTMG_ERR_TABLES, "%s", err.c_str());
  LwTab *T = ctx->lw_desc ? (LwTab *)ctx->lw_desc : new LwTab();
  ctx->lw_desc = T;
  if (!build_lw_tab(ctx->lw_ts, *T, err)) return ctx->fail(RRTMG_ERR_TABLES, "%s", err.c_str());
  int rc = ctx_prepare_device(ctx);
  if (rc) return rc;
  if (ctx->lw_tab_dev) (void)hipFree(ctx->lw_tab_dev);
  ctx->lw_tab_dev = nullptr;
  RRTMG_HIP_CHECK(ctx, hipMalloc((void **)&ctx->lw_tab_dev, ctx->lw_ts.flat.size() * sizeof(double)));
  RRTMG_HIP_CHECK(ctx, hipMemcpy(ctx->lw_tab_dev, ctx->lw_ts.flat.data(), ctx->lw_ts.flat.size() * sizeof(double), hipMemcpyHostToDevice));
  T->t = ctx->lw_tab_dev;
  ctx->lw_ready = true;
  return RRTMG_OK;
}

// the call on an internal copy of its inputs, cloud-free columns first (rrtmg_sort.h; see sw_sorted_call)
static int lw_sorted_call(rrtmg_ctx *ctx, const rrtmg_lw_args *a) {
  int rc = ctx_prepare_device(ctx);
  if (rc) return rc;
  hipStream_t s = ctx->deferred ? ctx->stream_lw : ctx->stream;
  const int N = a->ncol, L = a->nlay;
  ColumnSort cs(ctx, s, N, L, "lw.sort.");
  if (!cs.prepare(a->cldfr)) return ctx->status;
  rrtmg_lw_args b = *a;
  b.ncol = cs.Np; b.shard_col0 = 0; b.shard_ncol = 0;
  const size_t l = (size_t)L, l1 = l + 1;
  b.play = cs.gather("play", a->play, l); b.plev = cs.gather("plev", a->plev, l1); b.tlay = cs.gather("tlay", a->tlay, l);
  b.tlev = cs.gather("tlev", a->tlev, l1); b.tsfc = cs.gather("tsfc", a->tsfc, 1);
  b.h2ovmr = cs.gather("h2o", a->h2ovmr, l); b.o3vmr = cs.gather("o3", a->o3vmr, l); b.co2vmr = cs.gather("co2", a->co2vmr, l);
  b.ch4vmr = cs.gather("ch4", a->ch4vmr, l); b.n2ovmr = cs.gather("n2o", a->n2ovmr, l); b.o2vmr = cs.gather("o2", a->o2vmr, l);
  b.cfc11vmr = cs.gather("cfc11", a->cfc11vmr, l); b.cfc12vmr = cs.gather("cfc12", a->cfc12vmr, l);
  b.cfc22vmr = cs.gather("cfc22", a->cfc22vmr, l); b.ccl4vmr = cs.gather("ccl4", a->ccl4vmr, l);
  b.emis = cs.gather("emis", a->emis, 16);
  b.cldfr = cs.gather("cldfr", a->cldfr, l); b.taucld = cs.gather("taucld", a->taucld, l, 16);
  b.cicewp = cs.gather("cicewp", a->cicewp, l); b.cliqwp = cs.gather("cliqwp", a->cliqwp, l);
  b.reice = cs.gather("reice", a->reice, l); b.reliq = cs.gather("reliq", a->reliq, l);
  b.tauaer = cs.gather("tauaer", a->tauaer, l * 16);
  b.cldfmcl = cs.gather("cldfmcl", a->cldfmcl, l, kLwNGpt);
  const bool dr = a->idrv != 0;
  double *o[8] = {cs.out("o0", l1), cs.out("o1", l1), cs.out("o2", l), cs.out("o3", l1), cs.out("o4", l1), cs.out("o5", l),
                  dr ? cs.out("o6", l1) : nullptr, dr ? cs.out("o7", l1) : nullptr};
  if (!cs.ok) return ctx->status;
  if (!a->uflx || !a->dflx || !a->hr || !a->uflxc || !a->dflxc || !a->hrc) return ctx->fail(RRTMG_ERR_ARG, "output array is NULL");
  if (dr && (!a->duflx_dt || !a->duflxc_dt)) return ctx->fail(RRTMG_ERR_ARG, "idrv=1 needs duflx_dt/duflxc_dt");
  b.uflx = o[0]; b.dflx = o[1]; b.hr = o[2]; b.uflxc = o[3]; b.dflxc = o[4]; b.hrc = o[5]; b.duflx_dt = o[6]; b.duflxc_dt = o[7];
  ctx->sorting = true;
  rc = lw_fluxes_impl(ctx, &b);
  ctx->sorting = false;
  if (rc) return rc;
  double *u[8] = {a->uflx, a->dflx, a->hr, a->uflxc, a->dflxc, a->hrc, a->duflx_dt, a->duflxc_dt};
  for (int k = 0; k < (dr ? 8 : 6); ++k) cs.scatter(o[k], u[k], (k == 2 || k == 5) ? l : l1);
  RRTMG_HIP_CHECK(ctx, hipGetLastError());
  if (!ctx->deferred) RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(s));
  return RRTMG_OK;
}

int lw_fluxes_impl(rrtmg_ctx *ctx, const rrtmg_lw_args *a) {
  if (ctx->lw_ready && a && ctx->sort_columns && !ctx->sorting && a->memspace == 1 && a->icld != 0 && a->cldfr && a->ncol >= 128 && a->nlay > 0 && a->nlay <= 256 &&
      !(a->mcica && a->irng != 0))
    return lw_sorted_call(ctx, a);
  if (!ctx->lw_ready) return ctx->fail(RRTMG_ERR_NOT_INITIALISED, "rrtmg_hip_lw_init has not been called");
  if (!a || a->ncol <= 0 || a->nlay <= 0) return ctx->fail(RRTMG_ERR_ARG, "ncol/nlay must be positive");
  if (a->nlay > 256) return ctx->fail(RRTMG_ERR_ARG, "nlay > 256 not supported (cloud-mask words)");
  if (a->shard_ncol != 0 && (a->shard_col0 < 0 || a->shard_col0 + a->ncol > a->shard_ncol)) return ctx->fail(RRTMG_ERR_ARG, "shard_col0/shard_ncol do not contain ncol columns");
  int rc = ctx_prepare_device(ctx);
  if (rc) return rc;
  hipStream_t s = (ctx->deferred && a->memspace == 1) ? ctx->stream_lw : ctx->stream;
  const int N = a->ncol, L = a->nlay;
  const size_t nl = (size_t)N * L, nl1 = (size_t)N * (L + 1);
  const LwTab &T = *(LwTab *)ctx->lw_desc;
  LwDev d{};
  d.ncol = N; d.nlay = L;
  d.icld = a->icld;
  if (d.icld < 0 || d.icld > 3) d.icld = 2;   // rrtmg_lw_rad.nomcica.f90:436
  d.idrv = a->idrv ? 1 : 0;
  d.inflag = a->inflglw; d.iceflag = a->iceflglw; d.liqflag = a->liqflglw; d.mcica = a->mcica ? 1 : 0;
  d.k = ctx->k;
  RRTMG_PROFILE_READ_ONLY_ITEM(d)
  d.fluxfac = (2.0 * asin(1.0)) * 2.e4;       // rrtmg_lw_rad.nomcica.f90:420-421
  const bool maxrand = !d.mcica && d.icld >= 2;   // rtrnmr (rrtmg_lw_rad.nomcica.f90:527-544)
  if (d.mcica && d.icld >= 1 && d.inflag == 1) return ctx->fail(RRTMG_ERR_INFLAG1_MCICA, "longwave: %s", status_message(RRTMG_ERR_INFLAG1_MCICA));   // rrtmg_lw_cldprmc.f90:172

  // ---- inputs (rrtmg_host_inputs.h: uniform arrays are filled on the device, all-zero band arrays are absent) ----------------
  bool ok = true;
  const double ps = a->pressure_scale, ws = a->water_path_scale;
  HostInputs hi(ctx, s, "lw.in.", a->memspace);
  hi.add(&d.play, a->play, nl, "play", true, InPolicy::Plain, ps); hi.add(&d.plev, a->plev, nl1, "plev", true, InPolicy::Plain, ps);
  hi.add(&d.tlay, a->tlay, nl, "tlay", true); hi.add(&d.tlev, a->tlev, nl1, "tlev", false); hi.add(&d.tsfc, a->tsfc, N, "tsfc", true);
  hi.add(&d.h2o, a->h2ovmr, nl, "h2o", true, InPolicy::Plain, a->h2o_mul, a->h2o_div); hi.add(&d.o3, a->o3vmr, nl, "o3", true);
  hi.add(&d.co2, a->co2vmr, nl, "co2", true); hi.add(&d.ch4, a->ch4vmr, nl, "ch4", true); hi.add(&d.n2o, a->n2ovmr, nl, "n2o", true);
  hi.add(&d.o2, a->o2vmr, nl, "o2", true);
  hi.add(&d.cfc11, a->cfc11vmr, nl, "cfc11", false); hi.add(&d.cfc12, a->cfc12vmr, nl, "cfc12", false);
  hi.add(&d.cfc22, a->cfc22vmr, nl, "cfc22", false); hi.add(&d.ccl4, a->ccl4vmr, nl, "ccl4", false);
  hi.add(&d.emis, a->emis, (size_t)N * 16, "emis", true);
  const bool clouds = d.icld >= 1;
  if (clouds) {
    hi.add(&d.cldfr, a->cldfr, nl, "cldfr", true);
    // (given directly -- inflag 0 -- the cloud optical depth is used as it is; otherwise zeros mean there is none to add)
    hi.add(&d.taucld, a->taucld, nl * 16, "taucld", d.inflag == 0, d.inflag == 0 ? InPolicy::Plain : InPolicy::ZeroAbsent);
    hi.add(&d.cicewp, a->cicewp, nl, "cicewp", d.inflag >= 1, InPolicy::Plain, ws); hi.add(&d.cliqwp, a->cliqwp, nl, "cliqwp", d.inflag >= 1, InPolicy::Plain, ws);
    hi.add(&d.reice, a->reice, nl, "reice", d.inflag == 2); hi.add(&d.reliq, a->reliq, nl, "reliq", d.inflag == 2);
  }
  hi.add(&d.tauaer, a->tauaer, nl * 16, "tauaer", false, InPolicy::ZeroAbsent);
  const double *cldfmcl_dev = nullptr;
  if (clouds && d.mcica && a->cldfmcl) hi.add(&cldfmcl_dev, a->cldfmcl, nl * kLwNGpt, "cldfmcl", true);
  if (!hi.finish()) return ctx->status;

  auto wd = [&](const char *name, size_t n) -> double * { double *p = (double *)ctx->buf(std::string("lw.w.") + name, n * sizeof(double)); if (!p) ok = false; return p; };
  d.prep = wd("prep", lw_prep_size(N, L));
  d.secdiff = wd("secdiff", (size_t)N * 16);
  d.laytrop = (int32_t *)ctx->buf("lw.w.laytrop", (size_t)N * 4);
  d.ncbands = (int32_t *)ctx->buf("lw.w.ncbands", (size_t)N * 4);
  d.tile_cld = (int32_t *)ctx->buf("lw.w.tilecld", (size_t)((N + 63) / 64) * 4);
  d.ncloudy = ctx->ncloudy_dev + 1;
  if (maxrand) d.mr = wd("mr", lw_mr_size(N, L));
  if (!d.laytrop || !d.ncbands || !d.tile_cld) ok = false;
  if (clouds) d.ctau = wd("ctau", nl * 16);
  d.nw = (L + 63) / 64;
  if (clouds && d.mcica) {
    d.mask = (uint64_t *)ctx->buf("lw.w.mask", (size_t)kLwNGpt * d.nw * N * 8);
    d.anymask = (uint64_t *)ctx->buf("lw.w.anymask", (size_t)d.nw * N * 8);
    if (!d.mask || !d.anymask) ok = false;
  }
  const int ntile = (N + 63) / 64;
  const int nk = d.idrv ? 6 : 4;
  // what the previous call found (rrtmg_ctx::CallHint): read without waiting, used for speed only
  const int hint_cloudy = (ctx->hint[1].ntile == ntile && ctx->hint[1].nlay == L) ? ctx->hint[1].ncloudy : -1;
  int chunk_tiles = ctx->chunk_tiles;
  if (ctx->chunk_auto && L > 80 && hint_cloudy >= 0 && 10 * hint_cloudy >= 9 * ntile) chunk_tiles = 64;   // deep cloudy grid: DESIGN.md 5
  chunk_tiles = ctx->plan_chunks(1, chunk_tiles, ntile, L, (clouds && !ctx->sorting) ? hint_cloudy : -1,   /* (a sorted grid keeps the small chunks: its tiles are segregated by kind, every chunk but one is of one kind) */ (size_t)kLwNGpt * LF_N * L * 64 * sizeof(double), "lw.w.scratch");
  const int ctile = ntile < chunk_tiles ? ntile : chunk_tiles;   // tiles per solve chunk
  int32_t *tlist = (int32_t *)ctx->buf("lw.w.tilelist", (size_t)(2 * ctile + 2) * 4);
  if (!tlist) ok = false;
  d.tcap = ctile; d.tlist = tlist; d.tcnt = tlist ? tlist + 2 * d.tcap : nullptr;
  d.scratch = wd("scratch", (size_t)ctile * kLwNGpt * LF_N * L * 64);
  d.part = wd("part", (size_t)T.nitem * nk * (L + 1) * ctile * 64);
  if (!a->uflx || !a->dflx || !a->hr || !a->uflxc || !a->dflxc || !a->hrc) return ctx->fail(RRTMG_ERR_ARG, "output array is NULL");
  if (d.idrv && (!a->duflx_dt || !a->duflxc_dt)) return ctx->fail(RRTMG_ERR_ARG, "idrv=1 needs duflx_dt/duflxc_dt");
  if (a->memspace == 1) {
    d.uflx = a->uflx; d.dflx = a->dflx; d.hr = a->hr; d.uflxc = a->uflxc; d.dflxc = a->dflxc; d.hrc = a->hrc;
    d.duflx_dt = a->duflx_dt; d.duflxc_dt = a->duflxc_dt;
  } else {
    d.uflx = wd("o.uflx", nl1); d.dflx = wd("o.dflx", nl1); d.hr = wd("o.hr", nl); d.uflxc = wd("o.uflxc", nl1);
    d.dflxc = wd("o.dflxc", nl1); d.hrc = wd("o.hrc", nl);
    if (d.idrv) { d.duflx_dt = wd("o.du", nl1); d.duflxc_dt = wd("o.duc", nl1); }
  }
  if (!ok) return ctx->status;
#ifdef RRTMG_PROFILE
  d.phase = (unsigned long long *)ctx->buf("lw.w.phase", 16 * 8);
  if (!d.phase) return ctx->status;
  RRTMG_HIP_CHECK(ctx, hipMemsetAsync(d.phase, 0, 16 * 8, s));
#endif
  d.err = ctx->err_dev + 1;   // [0] shortwave, [1] longwave
  const bool deferred_call = ctx->deferred && a->memspace == 1;
  if (!deferred_call) {
    // a synchronous call owns its flag; flags of calls still pending from deferred mode are collected first
    if (ctx->pending[0] || ctx->pending[1]) { const int prc = rrtmg_hip_synchronize(ctx); if (prc) return prc; }
    RRTMG_HIP_CHECK(ctx, hipMemsetAsync(d.err, 0, sizeof(int), s));
  }   // deferred: the flag accumulates (atomicMax) until rrtmg_hip_synchronize collects and clears it

  const dim3 gcol(ntile), blk(64);
  if (!d.tlev) {
    // no interface temperatures given: log-pressure interpolation of the layer temperatures on the device, as climt's
    // host does before the call when calculate_interface_temperature is set (lw/component.py:378-384, util.py:89-142)
    double *tl = wd("tlev", nl1);
    if (!ok) return ctx->status;
    launch_interface_values(s, N, L, d.tlay, d.tsfc, d.play, d.plev, tl);
    d.tlev = tl;
  }
  // (more than 64 KB of dynamic LDS has to be allowed per kernel once; if the runtime refuses, the scan re-reads the slab)
  const bool big_lds = ctx->allow_dynamic_lds(0, (const void *)lw_prep_fused_kernel, kLwKeepLayers * 3 * 64 * (int)sizeof(double));
  const int keep_layers = (L <= kLwKeepLayers && (big_lds || (size_t)L * 3 * 64 * sizeof(double) <= 64 * 1024)) ? L : 0;
  if (clouds) {
    if (d.mcica) {
      if (a->cldfmcl) {
        hipLaunchKernelGGL(mask_from_cldfmcl_kernel, dim3(ntile, kLwNGpt), blk, 0, s, N, L, kLwNGpt, cldfmcl_dev, d.mask, d.nw);
      } else if (a->irng == 0) {
        const uint32_t *jumps = kiss_jumps_device(ctx, 1, kLwNGpt, L, d.icld, a->permuteseed, s);
        if (!jumps) return ctx->status;
        hipLaunchKernelGGL(kiss_mask_kernel, dim3(kLwNGpt, ntile), blk, 0, s, N, L, d.icld, d.play, d.cldfr, d.mask, d.nw, d.err, jumps);
      } else {
        rc = mt_mask_device(ctx, 1, N, L, kLwNGpt, d.icld, a->permuteseed, d.cldfr, d.mask, d.nw, a->shard_col0, a->shard_ncol, s);
        if (rc) return rc;
      }
      hipLaunchKernelGGL(lw_anymask_kernel, gcol, blk, 0, s, d);
    }
  }
  // preparation, solve and band integration, one column chunk at a time (see sw_fluxes_impl)
  for (int t0 = 0; t0 < ntile; t0 += ctile) {
    const int nt = ntile - t0 < ctile ? ntile - t0 : ctile;
    d.col0 = t0 * 64; d.pcols = ctile * 64;
    hipLaunchKernelGGL(lw_prep_fused_kernel, dim3(nt), dim3(64 * kPrepWaves), (size_t)keep_layers * 3 * 64 * sizeof(double), s, d, T,
                       clouds && !d.mcica ? 1 : 0, maxrand ? 1 : 0, keep_layers, t0);
    if (clouds && d.mcica) hipLaunchKernelGGL(lw_cloudmc_kernel, dim3(nt, L), blk, 0, s, d, T, t0);
    hipLaunchKernelGGL(tile_lists_kernel, dim3(1), blk, 0, s, d.tile_cld + t0, nt, tlist, tlist + 2 * d.tcap, d.tcap);
    const dim3 lwwg(64 * kLwWgWaves);
    const int lwgrid = (nt + kLwTileGroup - 1) / kLwTileGroup * kLwGroupBlocks * T.nitem;
    const int ci = t0 / ctile;
    auto clear_variant = [&]() {
      (void)hipEventRecord(ctx->chunk_event(1, ci, 0), s);
      hipLaunchKernelGGL((lw_solve_all_kernel<false, false>), dim3(lwgrid), lwwg, 0, s, d, T, t0, nt);
      (void)hipEventRecord(ctx->chunk_event(1, ci, 1), s);
    };
    auto cloudy_variant = [&]() {
      (void)hipEventRecord(ctx->chunk_event(3, ci, 0), s);
      if (maxrand) hipLaunchKernelGGL((lw_solve_all_kernel<true, true>), dim3(lwgrid), lwwg, 0, s, d, T, t0, nt);
      else hipLaunchKernelGGL((lw_solve_all_kernel<true, false>), dim3(lwgrid), lwwg, 0, s, d, T, t0, nt);
      (void)hipEventRecord(ctx->chunk_event(3, ci, 1), s);
    };
    // the variant expected to find nothing goes first (see sw_fluxes_impl)
    // (a sorted grid -- rrtmg_sort.h -- has its cloud-free tiles first: the chunks in front of the previous call's cloudy-tile count
    //  are expected to hold no cloudy tile)
    const bool expect_clear = clouds && hint_cloudy >= 0 && (hint_cloudy == 0 || (ctx->sorting && t0 + nt <= ntile - hint_cloudy));
    if (expect_clear) { cloudy_variant(); clear_variant(); }
    else { clear_variant(); if (clouds) cloudy_variant(); }
    d.hint_out = t0 + ctile >= ntile ? (int32_t *)&ctx->hint[1].ncloudy : nullptr;
    hipLaunchKernelGGL(lw_fluxheat_kernel, dim3(nt, (L + kFluxLev) / kFluxLev), dim3(64 * (kFluxLev + 1)), 0, s, d, T, t0);
  }
  ctx->hint[1].ntile = ntile; ctx->hint[1].nlay = L;
  ctx->ev_chunks[1] = (ntile + ctile - 1) / ctile; ctx->ev_chunks[3] = clouds ? ctx->ev_chunks[1] : 0;
  RRTMG_HIP_CHECK(ctx, hipGetLastError());
#ifdef RRTMG_PROFILE
  {
    unsigned long long ph[16];
    RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(s));
    RRTMG_HIP_CHECK(ctx, hipMemcpy(ph, d.phase, sizeof ph, hipMemcpyDeviceToHost));
    static const char *nm[8] = {"setup", "prep rows", "taumol", "planck+rows", "lookups+recurrence", "stores", "surface+up sweep", "-"};
    const double w = ph[8] ? (double)ph[8] : 1.0;
    fprintf(stderr, "lw phases (s_memtime ticks per wave-item, %llu waves, %d layers):", ph[8], L);
    for (int k = 0; k < 7; ++k) fprintf(stderr, " %s %.0f;", nm[k], ph[k] / w);
    fprintf(stderr, "\n  per layer of the downward sweep:");
    for (int k = 1; k <= 5; ++k) fprintf(stderr, " %s %.0f", nm[k], ph[k] / w / L);
    fprintf(stderr, "; up sweep per layer %.0f\n", ph[6] / w / L);
  }
#endif

  if (ctx->deferred && a->memspace == 1) { ctx->pending[1] = true; ctx->status = 0; return RRTMG_OK; }
  int herr = 0;
  if (a->memspace == 0) {
    const OutCopy oc[8] = {{a->uflx, d.uflx, nl1}, {a->dflx, d.dflx, nl1}, {a->uflxc, d.uflxc, nl1}, {a->dflxc, d.dflxc, nl1},
                           {a->hr, d.hr, nl}, {a->hrc, d.hrc, nl}, {a->duflx_dt, d.duflx_dt, nl1}, {a->duflxc_dt, d.duflxc_dt, nl1}};
    rc = copy_out(ctx, s, oc, d.idrv ? 8 : 6, d.err, &herr);
    if (rc) return rc;
  } else {
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(&herr, d.err, sizeof(int), hipMemcpyDeviceToHost, s));
    RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(s));
  }
  if (herr) return ctx->fail(herr, "longwave: %s", status_message(herr));
  ctx->status = 0;
  return RRTMG_OK;
}

}  // namespace rrtmg
