// rrtmg_sw.hip -- shortwave kernels and launch sequence (gfx950).
//
// Launch sequence of one rrtmg_hip_sw_fluxes call (all on the context's shortwave stream):
//   sw_aer_kernel       (iaer == 6)              ECMWF aerosol mixing per (column, layer)
//   kiss_mask_kernel / mask upload (McICA)       sub-column cloud mask
//   per column chunk (<= RRTMG_HIP_CHUNK_TILES tiles), so that a chunk's rows are still cached when its solve reads them:
//     sw_prep_fused_kernel <<<tiles, 16 waves>>>  inatm_sw + setcoef_sw per (column, layer), then the column part from the
//                         index words in LDS (laytrop, cloud flag, solar-source layer per band); non-McICA cloudy tiles: the
//                         band cloud optics
//     sw_cloud_kernel     (McICA)                 band cloud optics per (column, layer)
//     tile_lists_kernel   <<<1, 64>>>             the chunk's tiles by solve variant, compacted in tile order (SwDev::tlist)
//     sw_solve_all_kernel<false> (cloud-free tiles) + sw_solve_cloudy_kernel (cloudy tiles): wavefront = tile(64 columns) x work
//                         item (4|2 g-points of a band), workgroup = 16 | 8 tiles of one item sharing its tables in LDS;
//                         a workgroup's tiles are consecutive entries of ITS variant's list, so all of its wavefronts have
//                         work wherever cloud-free and cloudy tiles interleave (every fourth tile cloud-free: 6 of 8 and
//                         4 of 16 wavefronts otherwise, in workgroups that hold a whole CU either way)
//     sw_fluxheat_kernel  <<<(tiles, levels/15), 16 waves>>>  g-point sum per interface + heating rates
#include <future>

#include "rrtmg_ctx.h"
#include "rrtmg_sw_device.h"
#include "rrtmg_sw_host.h"
#include "rrtmg_mcica_kernels.h"
#include "rrtmg_sort.h"

namespace rrtmg {

// Preparation in ONE launch (round 1: three kernels): a workgroup = one 64-column tile, 16 wavefronts.  Phase 1: wave w prepares layers w, w+16, ...; phase 2, behind a barrier: wave b runs
// band b's column bookkeeping on the rows just written (L2-hot) and wave 0 sets the tile's cloud flag; phase 3, in cloudy
// tiles of a non-McICA call: the band cloud optics, again layers strided over the waves (with McICA, where most tiles are
// cloudy, the optics stay a launch of their own over (tile, layer): one workgroup per tile was measured 5 % slower on the
// whole McICA step).  Same per-thread functions, same results.
constexpr int kPrepWaves = 16;
__global__ void __launch_bounds__(64 * kPrepWaves) sw_prep_fused_kernel(SwDev d, SwTab T, int clouds, int tile0) {
  const int tile = tile0 + blockIdx.x;   // (launched per column chunk, right before the chunk's solve: its rows are still cached)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = tile * 64 + lane;
  const bool act = col < d.ncol;
  __shared__ int sh_cld;
  extern __shared__ int sh_idx[];    // the tile's packed index words [layer][lane] (dynamic LDS, nlay x 64 ints): the 14 band scans of phase 2 read them here
  if (act)
    for (int l = w; l < d.nlay; l += kPrepWaves) sh_idx[l * 64 + lane] = sw_prep_layer(d, T, col, l);
  __syncthreads();
  if (act)
    for (int b = w; b < kSwNBand; b += kPrepWaves) sw_prep_column(d, T, col, b, b + 1, sh_idx + lane, 64);
  if (w == 0) {
    const unsigned long long any = __ballot(act && d.anycld[col] != 0);
    if (lane == 0) { d.tile_cld[tile] = any != 0ull; sh_cld = any != 0ull; if (any) atomicAdd(d.ncloudy, 1); }
  }
  if (!clouds) return;
  __syncthreads();
  if (!sh_cld || !act) return;
  for (int l = w; l < d.nlay; l += kPrepWaves) sw_cloud_layer(d, T, col, l);
}

__global__ void __launch_bounds__(64) sw_cloud_kernel(SwDev d, SwTab T, int tile0) {
  const int tile = tile0 + blockIdx.x;
  if (!d.tile_cld[tile]) return;   // cloud-free tile: the clear-sky solve variant never reads the cloud optics
  const int col = tile * 64 + threadIdx.x;
  const int lay = blockIdx.y;
  if (col < d.ncol) sw_cloud_layer(d, T, col, lay);
}

// ECMWF aerosol mixing (iaer = 6), rrtmg_sw_rad.nomcica.f90:693-727 -> per-band tau/ssa/asm
__global__ void __launch_bounds__(64) sw_aer_kernel(SwDev d, SwTab T, const double *ecaer, double *ta, double *om, double *as) {
  const int col = blockIdx.x * 64 + threadIdx.x;
  const int lay = blockIdx.y;
  if (col >= d.ncol) return;
  const int L = d.nlay, N = d.ncol;
  const double *t = T.t;
  for (int ib = 0; ib < kSwNBand; ++ib) {
    double ztaua = 0.0, zasya = 0.0, zomga = 0.0;
    for (int ia = 0; ia < 6; ++ia) {
      const double e = ecaer[((long)ia * L + lay) * N + col];
      const double rt = t[T.rsrtaua + ib + kSwNBand * ia], rp = t[T.rsrpiza + ib + kSwNBand * ia], ra = t[T.rsrasya + ib + kSwNBand * ia];
      ztaua = ztaua + rt * e;
      zomga = zomga + rt * e * rp;
      zasya = zasya + rt * e * rp * ra;
    }
    if (ztaua == 0.0) {
      ztaua = 0.0; zasya = 0.0; zomga = 1.0;
    } else {
      if (zomga != 0.0) zasya = zasya / zomga;
      if (ztaua != 0.0) zomga = zomga / ztaua;
    }
    const long o = ((long)ib * L + lay) * N + col;
    ta[o] = ztaua; om[o] = zomga; as[o] = zasya;
  }
}

// All 112 g-points in ONE launch.  Wavefront = 64 columns of one tile x one work item (4 or 2 consecutive g-points
// of a band, SwTab::item): the thread carries the item's g-points through both sweeps, so the layer state, species
// mixtures and interpolation weights are evaluated once per item; the item's weighted fluxes are summed in
// registers: part[item][k][level][column].
// Workgroup = 16 wavefronts = the same item for 16 consecutive tiles (equal run times), one workgroup per CU,
// sharing in LDS (a) ONE copy of the 10001-entry transmittance table (80 KB): its lookups are per-lane random and
// cost a tag lookup per lane in the vector L1, but only bank conflicts in LDS; (b) the item's k-distribution slice,
// columns ig0..ig0+G-1 of the band's table slab, [nrows][G] (<= 58 KB): every absorption-coefficient row a lane
// needs is a 32-byte LDS read instead of a per-lane gather through the vector L1's 64 B/clk return path
// (measured: -6 % kernel time; with the rows through the scalar cache, an ablation, -10 % was the bound).
// Launch order: items heaviest first (SwTab::sched), tile groups fastest.  Speed only, never correctness.
constexpr int kSwWgWaves = 16;
constexpr int kSwGroupsPerBlock = 8;    // x 16 tiles = 128 tiles per block of the launch order
constexpr int kExpTblN = 10001;
// the item's slice of its band's table slab -> LDS: columns ig0 .. ig0+G-1, [nrows][G] (see SwBandTab)
__device__ __forceinline__ void sw_stage_slice(const SwTab &T, int item, double *sh_k, int nthreads) {
  const SwBandTab &B = T.b[item_band(item)];
  const int g = item_g(item);
  const double *src = T.t + B.slab + item_ig0(item);
  const int ng = B.ng, sh = g == 4 ? 2 : 1, n = B.nrows << sh;
  for (int i = threadIdx.x; i < n; i += nthreads) sh_k[i] = src[(long)(i >> sh) * ng + (i & (g - 1))];
}
// Two kernels are launched back to back: this one handles the cloud-free tiles with the cloud code compiled out
// (CLD = false: no spills, chunks of 4 g-points), sw_solve_cloudy_kernel the tiles flagged by sw_prep_kernel; a
// wavefront whose tile belongs to the other kernel exits at once.
template <bool CLD>
__global__ void __launch_bounds__(64 * kSwWgWaves) __attribute__((amdgpu_waves_per_eu(4))) sw_solve_all_kernel(SwDev d, SwTab T, int tile0, int ntile) {   // tiles tile0 .. tile0 + ntile - 1 (one column chunk)
  // Launch order: BLOCKS of kSwGroupsPerBlock tile groups (128 tiles); within a block work items heaviest first, tile groups
  // fastest -- a block's prep rows (58 MB at 60 layers) are read by its 32 work items while they are still cached, however many
  // tiles the launch covers (a large chunk of a grid with both kinds of tiles: 2048 tiles are 0.94 GB of prep rows, re-read
  // from HBM by every work item in item-major order).  Up to 128 tiles there is one block: the order of rounds 1-4.
  // this variant's tiles, compacted (SwDev::tlist): ngrp groups of kSwWgWaves list entries HAVE work.  The launch was sized for
  // every tile of the chunk (the host does not know the counts); the workgroups with work are the FIRST ngrp x nitem of the
  // dispatch order and dense in it, the others exit at once behind them.  (A workgroup that exits at once still has to be
  // PLACED with its 138 KB of LDS: interleaved with real ones -- a padded grid in round 2, or 8192 McICA columns with every
  // fourth tile cloud-free before this mapping, 3.64 ms against 2.93 with clouds everywhere -- half the real workgroups wait
  // for a CU that still holds another.)  The last block of the order holds the remaining groups.
  const int nmine = d.tcnt[CLD ? 1 : 0];
  const int ngrp = (nmine + kSwWgWaves - 1) / kSwWgWaves;
  const int q = blockIdx.x, per = kSwGroupsPerBlock * T.nitem, nfull = ngrp / kSwGroupsPerBlock;
  if (q >= ngrp * T.nitem) return;   // workgroup-uniform exit before the tables are staged
  const int gpb = q < nfull * per ? kSwGroupsPerBlock : ngrp - nfull * kSwGroupsPerBlock, r = q < nfull * per ? q % per : q - nfull * per;
  const int grp = (q < nfull * per ? q / per : nfull) * kSwGroupsPerBlock + r % gpb;
  const int first = grp * kSwWgWaves;
  __shared__ double sh_exp[kExpTblN];
  for (int i = threadIdx.x; i < kExpTblN; i += 64 * kSwWgWaves) sh_exp[i] = T.t[T.exp_tbl + i];
  const int k = r / gpb;
  RRTMG_PROFILE_ONLY_ITEM(d, k)
  const int id = T.sched[k], item = T.item[id], slot = id;
  constexpr bool kLdsK = true;
  __shared__ __attribute__((aligned(16))) double sh_k[kSwSlabMaxRows * 4];   // rows are read 16 bytes at a time
  sw_stage_slice(T, item, sh_k, 64 * kSwWgWaves);
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (first + wave >= nmine) return;
  const int ctile = d.tlist[(CLD ? d.tcap : 0) + first + wave];   // tile within the chunk
  const int tile = tile0 + ctile;
  const int col = tile * 64 + (threadIdx.x & 63);
  if (col >= d.ncol) return;
  double *scr = d.scratch + ((long)ctile * kSwNGpt + item_iw0(item)) * (long)F_NTOT * d.nlay * 64 + (threadIdx.x & 63) * 2;
  SwPartSink sink = sw_part_sink(d, slot, col);
  sw_solve_item<CLD, kLdsK>(d, T, sh_exp, item, col, scr, 64, sink, sh_k);
}


// The cloudy tiles (flagged by the preparation kernel): both sky streams per g-point.  They run the CLEAR kernel's item set
// -- chunks of 4 g-points, so the item-invariant work (layer state, species mixtures, weights, row indices) is paid once per
// 4 g-points, not per pair -- at 2 waves/SIMD: 238 VGPRs, no spills; 8-wave workgroups, one per CU, sharing the
// transmittance table and the chunk's slice in LDS.  Against the round-1 kernel (pairs, 170 VGPRs, 3 waves/SIMD, 12-wave
// workgroups): 1.91 -> 1.73 ms at 8192 columns.  The partial sums leave per chunk, the two pairs' sums added in the order the
// flux kernel added the pair slots of the round-1 kernel: bit-identical, half the partial-plane traffic.
constexpr int kC4Waves = 8;
constexpr int kC4GroupsPerBlock = 16;   // x 8 tiles = 128 tiles per block of the launch order
__global__ void __launch_bounds__(64 * kC4Waves) __attribute__((amdgpu_waves_per_eu(2, 2))) sw_solve_cloudy_kernel(SwDev d, SwTab T, int tile0, int ntile) {
  const int nmine = d.tcnt[1];   // the cloudy tiles, compacted (SwDev::tlist); the workgroups with work first and dense: see sw_solve_all_kernel
  const int ngrp = (nmine + kC4Waves - 1) / kC4Waves;
  const int q = blockIdx.x, per = kC4GroupsPerBlock * T.nitem, nfull = ngrp / kC4GroupsPerBlock;      // blocks of 128 tiles
  if (q >= ngrp * T.nitem) return;
  const int gpb = q < nfull * per ? kC4GroupsPerBlock : ngrp - nfull * kC4GroupsPerBlock, r = q < nfull * per ? q % per : q - nfull * per;
  const int grp = (q < nfull * per ? q / per : nfull) * kC4GroupsPerBlock + r % gpb, first = grp * kC4Waves, k = r / gpb;
  RRTMG_PROFILE_ONLY_ITEM(d, k)
  const int id = T.sched[k], item = T.item[id], slot = id;      // one slot per chunk
  __shared__ __attribute__((aligned(16))) double sh_k[kSwSlabMaxRows * 4];
  sw_stage_slice(T, item, sh_k, 64 * kC4Waves);
  __shared__ double sh_exp[kExpTblN];
  for (int i = threadIdx.x; i < kExpTblN; i += 64 * kC4Waves) sh_exp[i] = T.t[T.exp_tbl + i];
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (first + wave >= nmine) return;
  const int ctile = d.tlist[d.tcap + first + wave], tile = tile0 + ctile;
  const int lane = threadIdx.x & 63;
  const int col = tile * 64 + lane;
  if (col >= d.ncol) return;
  double *scr = d.scratch + ((long)ctile * kSwNGpt + item_iw0(item)) * (long)F_NTOT * d.nlay * 64 + lane * 2;
  SwPartSink sink = sw_part_sink(d, slot, col);
  sw_solve_item<true, true>(d, T, sh_exp, item, col, scr, 64, sink, sh_k);
}

// Spectral integration AND heating rates in one launch: a workgroup = one tile x kFluxLev layers; wave j sums the partial
// planes of interface level l0 + j (the extra wave kFluxLev: the halo level on top, recomputed by the next workgroup, which
// owns and stores it), the net fluxes meet in LDS, waves j < kFluxLev form the layer's heating rates from levels j and j + 1
// -- the same differences of the same doubles as sw_heat_layer reads back from memory.
constexpr int kFluxLev = 15;   // 16 waves per workgroup: the halo level is 1 in 16 of the partial-plane reads
__global__ void __launch_bounds__(64 * (kFluxLev + 1)) sw_fluxheat_kernel(SwDev d, SwTab T, int tile0) {
  // (the call's last launch leaves the preparation kernels' cloudy-tile count where the host will look for it, and clears it)
  if (d.hint_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { *d.hint_out = *d.ncloudy; *d.ncloudy = 0; }
  const int tile = tile0 + blockIdx.x, lane = threadIdx.x & 63, j = threadIdx.x >> 6;
  const int col = tile * 64 + lane, lev = blockIdx.y * kFluxLev + j;
  __shared__ double net[kFluxLev + 1][64], netc[kFluxLev + 1][64];
  const bool act = col < d.ncol && lev <= d.nlay;
  if (act) {
    double fu, fd, cu, cd;
    sw_flux_sums(d, T, col, lev, d.tile_cld[tile] != 0, fu, fd, cu, cd);
    if (j < kFluxLev || lev == d.nlay) {
      const long o = (long)lev * d.ncol + col;
      d.swuflx[o] = fu; d.swdflx[o] = fd; d.swuflxc[o] = cu; d.swdflxc[o] = cd;
    }
    net[j][lane] = fd - fu; netc[j][lane] = cd - cu;
  }
  __syncthreads();
  if (col < d.ncol && j < kFluxLev && lev < d.nlay) {
    const long o0 = (long)lev * d.ncol + col;
    const double zdpgcp = T.heatfac / d.pdp[o0];
    d.swhrc[o0] = (netc[j + 1][lane] - netc[j][lane]) * zdpgcp;
    d.swhr[o0] = (net[j + 1][lane] - net[j][lane]) * zdpgcp;
  }
}

void free_sw_desc(rrtmg_ctx *ctx) {
  delete (SwTab *)ctx->sw_desc;
  ctx->sw_desc = nullptr;
}

// stand-alone sub-column generator (host pointers): the mask is built on the device by either generator
int mcica_mask_impl(rrtmg_ctx *ctx, int which, int ncol, int nlay, int icld, int permuteseed, int irng,
                    const double *play, const double *cldfrac, double *cldfmcl) {
  if (ncol <= 0 || nlay <= 0 || !play || !cldfrac || !cldfmcl) return ctx->fail(RRTMG_ERR_ARG, "mcica_mask: bad argument");
  if (icld < 0 || icld > 3) return ctx->fail(RRTMG_ERR_ICLD, "%s", status_message(RRTMG_ERR_ICLD));
  const int nsub = which == 0 ? kSwNGpt : 140;
  const int nw = (nlay + 63) / 64;
  const size_t nl = (size_t)ncol * nlay;
  if (icld == 0) return RRTMG_OK;   // mcica_subcol_*: "if (icld.eq.0) return" -- outputs untouched
  int rc = ctx_prepare_device(ctx);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  double *dp = (double *)ctx->buf("mm.play", nl * 8), *dc = (double *)ctx->buf("mm.cld", nl * 8);
  double *dm = (double *)ctx->buf("mm.out", nl * nsub * 8);
  uint64_t *mk = (uint64_t *)ctx->buf("mm.mask", (size_t)nsub * nw * ncol * 8);
  if (!dp || !dc || !dm || !mk) return ctx->status;
  RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(dp, play, nl * 8, hipMemcpyHostToDevice, s));
  RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(dc, cldfrac, nl * 8, hipMemcpyHostToDevice, s));
  RRTMG_HIP_CHECK(ctx, hipMemsetAsync(ctx->err_dev, 0, sizeof(int), s));
  const int ntile = (ncol + 63) / 64;
  if (irng != 0) {
    rc = mt_mask_device(ctx, which == 0 ? 0 : 1, ncol, nlay, nsub, icld, permuteseed, dc, mk, nw, 0, 0, s);
    if (rc) return rc;
  } else {
    const uint32_t *jumps = kiss_jumps_device(ctx, which == 0 ? 0 : 1, nsub, nlay, icld, permuteseed, s);
    if (!jumps) return ctx->status;
    hipLaunchKernelGGL(kiss_mask_kernel, dim3(nsub, ntile), dim3(64), 0, s, ncol, nlay, icld, dp, dc, mk, nw, ctx->err_dev, jumps);
  }
  hipLaunchKernelGGL(cldfmcl_from_mask_kernel, dim3(ntile, nsub), dim3(64), 0, s, ncol, nlay, nsub, mk, nw, dm);
  int herr = 0;
  RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(&herr, ctx->err_dev, sizeof(int), hipMemcpyDeviceToHost, s));
  RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(cldfmcl, dm, nl * nsub * 8, hipMemcpyDeviceToHost, s));
  RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(s));
  if (herr) return ctx->fail(herr, "mcica_mask: %s", status_message(herr));
  return RRTMG_OK;
}

int sw_init_impl(rrtmg_ctx *ctx, double cpdair, const char *blob_path) {
  if (!ctx->have_constants) return ctx->fail(RRTMG_ERR_NOT_INITIALISED, "set_constants must be called before sw_init");
  std::string path = blob_path ? std::string(blob_path) : default_blob_path("sw");
  Blob blob;
  std::string err;
  if (!blob.load(path, err)) return ctx->fail(RRTMG_ERR_TABLES, "%s", err.c_str());
  ctx->sw_ts = TableSet();
  if (!build_tables(blob, "sw", cpdair, ctx->k.grav, ctx->k.secdy, ctx->sw_ts, err)) return ctx->fail(RRTMG_ERR_TABLES, "%s", err.c_str());
  SwTab *T = ctx->sw_desc ? (SwTab *)ctx->sw_desc : new SwTab();
  ctx->sw_desc = T;
  if (!build_sw_tab(ctx->sw_ts, *T, err)) return ctx->fail(RRTMG_ERR_TABLES, "%s", err.c_str());
  int rc = ctx_prepare_device(ctx);
  if (rc) return rc;
  if (ctx->sw_tab_dev) (void)hipFree(ctx->sw_tab_dev);
  ctx->sw_tab_dev = nullptr;
  RRTMG_HIP_CHECK(ctx, hipMalloc((void **)&ctx->sw_tab_dev, ctx->sw_ts.flat.size() * sizeof(double)));
  RRTMG_HIP_CHECK(ctx, hipMemcpy(ctx->sw_tab_dev, ctx->sw_ts.flat.data(), ctx->sw_ts.flat.size() * sizeof(double), hipMemcpyHostToDevice));
  T->t = ctx->sw_tab_dev;
  ctx->sw_ready = true;
  return RRTMG_OK;
}

// the call on an internal copy of its inputs, cloud-free columns first (rrtmg_sort.h; opt-in, device pointers, kissvec or no McICA)
static int sw_sorted_call(rrtmg_ctx *ctx, const rrtmg_sw_args *a) {
  int rc = ctx_prepare_device(ctx);
  if (rc) return rc;
  const int N = a->ncol, L = a->nlay;
  ColumnSort cs(ctx, ctx->stream, N, L, "sw.sort.");
  if (!cs.prepare(a->cldfr)) return ctx->status;
  rrtmg_sw_args b = *a;
  b.ncol = cs.Np; b.shard_col0 = 0; b.shard_ncol = 0;
  const size_t l = (size_t)L, l1 = l + 1;
  b.play = cs.gather("play", a->play, l); b.plev = cs.gather("plev", a->plev, l1); b.tlay = cs.gather("tlay", a->tlay, l);
  b.tlev = nullptr; b.tsfc = nullptr;   // (the shortwave reads neither)
  b.h2ovmr = cs.gather("h2o", a->h2ovmr, l); b.o3vmr = cs.gather("o3", a->o3vmr, l); b.co2vmr = cs.gather("co2", a->co2vmr, l);
  b.ch4vmr = cs.gather("ch4", a->ch4vmr, l); b.n2ovmr = cs.gather("n2o", a->n2ovmr, l); b.o2vmr = cs.gather("o2", a->o2vmr, l);
  b.asdir = cs.gather("asdir", a->asdir, 1); b.asdif = cs.gather("asdif", a->asdif, 1); b.aldir = cs.gather("aldir", a->aldir, 1);
  b.aldif = cs.gather("aldif", a->aldif, 1); b.coszen = cs.gather("coszen", a->coszen, 1);
  b.cldfr = cs.gather("cldfr", a->cldfr, l);
  b.taucld = cs.gather("taucld", a->taucld, l, kSwNBand); b.ssacld = cs.gather("ssacld", a->ssacld, l, kSwNBand);
  b.asmcld = cs.gather("asmcld", a->asmcld, l, kSwNBand); b.fsfcld = cs.gather("fsfcld", a->fsfcld, l, kSwNBand);
  b.cicewp = cs.gather("cicewp", a->cicewp, l); b.cliqwp = cs.gather("cliqwp", a->cliqwp, l);
  b.reice = cs.gather("reice", a->reice, l); b.reliq = cs.gather("reliq", a->reliq, l);
  b.tauaer = cs.gather("tauaer", a->tauaer, l * kSwNBand); b.ssaaer = cs.gather("ssaaer", a->ssaaer, l * kSwNBand);
  b.asmaer = cs.gather("asmaer", a->asmaer, l * kSwNBand); b.ecaer = cs.gather("ecaer", a->ecaer, l * 6);
  b.cldfmcl = cs.gather("cldfmcl", a->cldfmcl, l, kSwNGpt);
  double *o[6] = {cs.out("o0", l1), cs.out("o1", l1), cs.out("o2", l), cs.out("o3", l1), cs.out("o4", l1), cs.out("o5", l)};
  if (!cs.ok) return ctx->status;
  if (!a->swuflx || !a->swdflx || !a->swhr || !a->swuflxc || !a->swdflxc || !a->swhrc) return ctx->fail(RRTMG_ERR_ARG, "output array is NULL");
  b.swuflx = o[0]; b.swdflx = o[1]; b.swhr = o[2]; b.swuflxc = o[3]; b.swdflxc = o[4]; b.swhrc = o[5];
  ctx->sorting = true;
  rc = sw_fluxes_impl(ctx, &b);
  ctx->sorting = false;
  if (rc) return rc;
  double *u[6] = {a->swuflx, a->swdflx, a->swhr, a->swuflxc, a->swdflxc, a->swhrc};
  for (int k = 0; k < 6; ++k) cs.scatter(o[k], u[k], (k == 2 || k == 5) ? l : l1);
  RRTMG_HIP_CHECK(ctx, hipGetLastError());
  if (!ctx->deferred) RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return RRTMG_OK;
}

int sw_fluxes_impl(rrtmg_ctx *ctx, const rrtmg_sw_args *a) {
  if (ctx->sw_ready && a && ctx->sort_columns && !ctx->sorting && a->memspace == 1 && a->icld != 0 && a->cldfr && a->ncol >= 128 && a->nlay > 0 && a->nlay <= 256 &&
      !(a->mcica && a->irng != 0))
    return sw_sorted_call(ctx, a);
  if (!ctx->sw_ready) return ctx->fail(RRTMG_ERR_NOT_INITIALISED, "rrtmg_hip_sw_init has not been called");
  if (!a || a->ncol <= 0 || a->nlay <= 0) return ctx->fail(RRTMG_ERR_ARG, "ncol/nlay must be positive");
  if (a->nlay > 256) return ctx->fail(RRTMG_ERR_ARG, "nlay > 256 not supported (cloud-mask words)");
  if (a->shard_ncol != 0 && (a->shard_col0 < 0 || a->shard_col0 + a->ncol > a->shard_ncol)) return ctx->fail(RRTMG_ERR_ARG, "shard_col0/shard_ncol do not contain ncol columns");
  int rc = ctx_prepare_device(ctx);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const int N = a->ncol, L = a->nlay;
  const size_t nl = (size_t)N * L, nl1 = (size_t)N * (L + 1);
  const SwTab &T = *(SwTab *)ctx->sw_desc;
  SwDev d{};
  d.ncol = N; d.nlay = L;
  d.icld = a->icld; d.iaer = a->iaer;
  if (d.icld < 0 || d.icld > 3) d.icld = 2;                 // rrtmg_sw_rad.nomcica.f90:563
  if (d.iaer != 0 && d.iaer != 6 && d.iaer != 10) d.iaer = 0;
  d.inflag = a->inflgsw; d.iceflag = a->iceflgsw; d.liqflag = a->liqflgsw; d.mcica = a->mcica ? 1 : 0;
  d.k = ctx->k;
  RRTMG_PROFILE_READ_ONLY_ITEM(d)
  std::string err;
  std::vector<double> svar_col;
  {
    const long omg = ctx->sw_ts.off("sw/sol/mgavgcyc"), osb = ctx->sw_ts.off("sw/sol/sbavgcyc");
    rc = sw_scalar_setup(d, N, a->isolvar, a->adjes, a->dyofyr, a->scon, a->solcycfrac, a->bndsolvar, a->indsolvar,
                         omg >= 0 ? ctx->sw_ts.flat.data() + omg : nullptr, osb >= 0 ? ctx->sw_ts.flat.data() + osb : nullptr, svar_col, err);
  }
  if (rc) return ctx->fail(rc, "%s", err.c_str());
  if (d.icld >= 1 && d.inflag == 1 && d.mcica) return ctx->fail(RRTMG_ERR_INFLAG1_MCICA, "shortwave: %s", status_message(RRTMG_ERR_INFLAG1_MCICA));   // rrtmg_sw_cldprmc.f90:166
  if (d.icld >= 1 && d.inflag == 1) return ctx->fail(RRTMG_ERR_UNSUPPORTED, "inflgsw=1 has no shortwave implementation in RRTMG_SW (cldprop_sw handles 0 and 2)");

  // ---- inputs (rrtmg_host_inputs.h: uniform arrays are filled on the device, all-zero band arrays are absent) ----------------
  bool ok = true;
  const double ps = a->pressure_scale, ws = a->water_path_scale;
  HostInputs hi(ctx, s, "sw.in.", a->memspace);
  hi.add(&d.play, a->play, nl, "play", true, InPolicy::Plain, ps); hi.add(&d.plev, a->plev, nl1, "plev", true, InPolicy::Plain, ps);
  hi.add(&d.tlay, a->tlay, nl, "tlay", true);
  hi.add(&d.h2o, a->h2ovmr, nl, "h2o", true, InPolicy::Plain, a->h2o_mul, a->h2o_div); hi.add(&d.o3, a->o3vmr, nl, "o3", true);
  hi.add(&d.co2, a->co2vmr, nl, "co2", true); hi.add(&d.ch4, a->ch4vmr, nl, "ch4", true); hi.add(&d.n2o, a->n2ovmr, nl, "n2o", true);
  hi.add(&d.o2, a->o2vmr, nl, "o2", true);
  hi.add(&d.asdir, a->asdir, N, "asdir", true); hi.add(&d.asdif, a->asdif, N, "asdif", true);
  hi.add(&d.aldir, a->aldir, N, "aldir", true); hi.add(&d.aldif, a->aldif, N, "aldif", true);
  hi.add(&d.coszen, a->coszen, N, "coszen", true);
  const bool clouds = d.icld >= 1;
  if (clouds) {
    hi.add(&d.cldfr, a->cldfr, nl, "cldfr", true);
    const bool optics = (d.inflag == 0);
    // single-scattering albedo / asymmetry / forward fraction are read only where the optics are given directly -- under
    // inflag 2 they would multiply an optical depth below cldmin = 1e-20 at most -- so host copies are not uploaded then
    const bool up = optics || a->memspace == 1;
    if (up) {
      hi.add(&d.ssacld, a->ssacld, nl * kSwNBand, "ssacld", optics); hi.add(&d.asmcld, a->asmcld, nl * kSwNBand, "asmcld", optics);
      hi.add(&d.fsfcld, a->fsfcld, nl * kSwNBand, "fsfcld", optics);
    }
    hi.add(&d.cicewp, a->cicewp, nl, "cicewp", d.inflag == 2, InPolicy::Plain, ws); hi.add(&d.cliqwp, a->cliqwp, nl, "cliqwp", d.inflag == 2, InPolicy::Plain, ws);
    hi.add(&d.reice, a->reice, nl, "reice", d.inflag == 2); hi.add(&d.reliq, a->reliq, nl, "reliq", d.inflag == 2);
    // (stays live under inflag 2: the tauctot gate of cldprop_sw; given directly -- inflag 0 -- it is used as it is)
    hi.add(&d.taucld, a->taucld, nl * kSwNBand, "taucld", optics, optics ? InPolicy::Plain : InPolicy::ZeroAbsent);
  }
  const double *ecaer = nullptr;
  if (d.iaer == 10) {
    hi.add(&d.tauaer, a->tauaer, nl * kSwNBand, "tauaer", true); hi.add(&d.ssaaer, a->ssaaer, nl * kSwNBand, "ssaaer", true);
    hi.add(&d.asmaer, a->asmaer, nl * kSwNBand, "asmaer", true);
  } else if (d.iaer == 6) {
    hi.add(&ecaer, a->ecaer, nl * 6, "ecaer", true);
  }
  const double *cldfmcl_dev = nullptr;
  if (clouds && d.mcica && a->cldfmcl) hi.add(&cldfmcl_dev, a->cldfmcl, nl * kSwNGpt, "cldfmcl", true);
  if (!hi.finish()) return ctx->status;

  // ---- work buffers -------------------------------------------------------------------------
  auto wd = [&](const char *name, size_t n) -> double * { double *p = (double *)ctx->buf(std::string("sw.w.") + name, n * sizeof(double)); if (!p) ok = false; return p; };
  d.prep = wd("prep", sw_prep_size(N, L));
  d.pdp = wd("pdp", nl); d.cossza = wd("cossza", N);
  d.laytrop = (int32_t *)ctx->buf("sw.w.laytrop", (size_t)N * 4);
  d.laysolfr = (int32_t *)ctx->buf("sw.w.laysolfr", (size_t)N * 4 * kSwNBand); d.anycld = (int32_t *)ctx->buf("sw.w.anycld", (size_t)N * 4);
  d.tile_cld = (int32_t *)ctx->buf("sw.w.tilecld", (size_t)((N + 63) / 64) * 4);
  d.ncloudy = ctx->ncloudy_dev;
  if (!d.laytrop || !d.laysolfr || !d.anycld || !d.tile_cld) ok = false;
  if (clouds) { d.ctau = wd("ctau", nl * kSwNBand); d.cssa = wd("cssa", nl * kSwNBand); d.casm = wd("casm", nl * kSwNBand); }
  d.nw = (L + 63) / 64;
  if (clouds && d.mcica) { d.mask = (uint64_t *)ctx->buf("sw.w.mask", (size_t)kSwNGpt * d.nw * N * 8); if (!d.mask) ok = false; }
  const int ntile = (N + 63) / 64;
  // what the previous call found (rrtmg_ctx::CallHint): read without waiting, used for speed only
  const int hint_cloudy = (ctx->hint[0].ntile == ntile && ctx->hint[0].nlay == L) ? ctx->hint[0].ncloudy : -1;
  int chunk_tiles = ctx->chunk_tiles;
  if (ctx->chunk_auto && L > 80 && hint_cloudy >= 0 && 10 * hint_cloudy >= 9 * ntile) chunk_tiles = 64;   // deep cloudy grid: DESIGN.md 5
  chunk_tiles = ctx->plan_chunks(0, chunk_tiles, ntile, L, (clouds && !ctx->sorting) ? hint_cloudy : -1,   /* (a sorted grid keeps the small chunks: its tiles are segregated by kind, every chunk but one is of one kind) */ (size_t)kSwNGpt * F_NTOT * L * 64 * sizeof(double), "sw.w.scratch");
  const int ctile = ntile < chunk_tiles ? ntile : chunk_tiles;   // tiles per solve chunk
  int32_t *tlist = (int32_t *)ctx->buf("sw.w.tilelist", (size_t)(2 * ctile + 2) * 4);
  if (!tlist) ok = false;
  d.tcap = ctile; d.tlist = tlist; d.tcnt = tlist ? tlist + 2 * d.tcap : nullptr;
  d.scratch = wd("scratch", (size_t)ctile * kSwNGpt * F_NTOT * L * 64);
  d.part = wd("part", (size_t)kSwNSlot * 4 * (L + 1) * ctile * 64);
  if (!svar_col.empty()) {   // per-column solar-variability multipliers (rare: facular/sunspot amplitudes != 1)
    double *p = wd("svarcol", svar_col.size());
    if (!ok) return ctx->status;
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(p, svar_col.data(), svar_col.size() * sizeof(double), hipMemcpyHostToDevice, s));
    RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(s));   // svar_col is a local
    d.svar_col = p;
  }
  if (a->memspace == 1) {
    d.swuflx = a->swuflx; d.swdflx = a->swdflx; d.swhr = a->swhr; d.swuflxc = a->swuflxc; d.swdflxc = a->swdflxc; d.swhrc = a->swhrc;
  } else {
    d.swuflx = wd("o.uflx", nl1); d.swdflx = wd("o.dflx", nl1); d.swhr = wd("o.hr", nl); d.swuflxc = wd("o.uflxc", nl1); d.swdflxc = wd("o.dflxc", nl1); d.swhrc = wd("o.hrc", nl);
  }
  if (!ok) return ctx->status;
  if (!a->swuflx || !a->swdflx || !a->swhr || !a->swuflxc || !a->swdflxc || !a->swhrc) return ctx->fail(RRTMG_ERR_ARG, "output array is NULL");
  d.err = ctx->err_dev;
  const bool deferred_call = ctx->deferred && a->memspace == 1;
  if (!deferred_call) {
    // a synchronous call owns its flag; flags of calls still pending from deferred mode are collected first
    if (ctx->pending[0] || ctx->pending[1]) { const int prc = rrtmg_hip_synchronize(ctx); if (prc) return prc; }
    RRTMG_HIP_CHECK(ctx, hipMemsetAsync(d.err, 0, sizeof(int), s));
  }   // deferred: the flag accumulates (atomicMax) until rrtmg_hip_synchronize collects and clears it

  // ---- launches ---------------------------------------------------------------------------
  const dim3 gcl(ntile, L), blk(64);
  if (d.iaer == 6) {
    double *ta = wd("aer.tau", nl * kSwNBand), *om = wd("aer.ssa", nl * kSwNBand), *as = wd("aer.asm", nl * kSwNBand);
    if (!ok) return ctx->status;
    hipLaunchKernelGGL(sw_aer_kernel, gcl, blk, 0, s, d, T, ecaer, ta, om, as);
    d.tauaer = ta; d.ssaaer = om; d.asmaer = as;
  }
  if (clouds) {
    if (d.mcica) {
      if (a->cldfmcl) {
        hipLaunchKernelGGL(mask_from_cldfmcl_kernel, dim3(ntile, kSwNGpt), blk, 0, s, N, L, kSwNGpt, cldfmcl_dev, d.mask, d.nw);
      } else if (a->irng == 0) {
        const uint32_t *jumps = kiss_jumps_device(ctx, 0, kSwNGpt, L, d.icld, a->permuteseed, s);
        if (!jumps) return ctx->status;
        hipLaunchKernelGGL(kiss_mask_kernel, dim3(kSwNGpt, ntile), blk, 0, s, N, L, d.icld, d.play, d.cldfr, d.mask, d.nw, d.err, jumps);
      } else {
        rc = mt_mask_device(ctx, 0, N, L, kSwNGpt, d.icld, a->permuteseed, d.cldfr, d.mask, d.nw, a->shard_col0, a->shard_ncol, s);
        if (rc) return rc;
      }
    }
  }
  // preparation, solve and spectral integration, one column chunk at a time: the chunk's prep rows (58 MB at 8192 columns x
  // 60 layers) are read by its 32 work items while still in the L2s / the Infinity Cache, not streamed back from HBM after
  // the preparation of the whole grid (every solve launch of every chunk has its own event pair)
  // (Two chunks in flight at once -- even and odd chunks on two streams of the spectrum, each with its own work space -- were
  // built and measured in round 6: 131 072 clear-sky columns 23.2 -> 24.1-24.5 ms, config-5 shard 72.9-73.7 -> 74.1-74.9,
  // config-4 shard 5.83-5.94 -> 5.79-5.89: the other spectrum's solve already runs over a chunk's preparation and
  // integration, and two solves of one spectrum sharing the CUs take 1.7 x as long each.  docs/EXPERIMENTS.md E.)
  for (int t0 = 0; t0 < ntile; t0 += ctile) {
    const int nt = ntile - t0 < ctile ? ntile - t0 : ctile;
    d.col0 = t0 * 64; d.pcols = ctile * 64;
    hipLaunchKernelGGL(sw_prep_fused_kernel, dim3(nt), dim3(64 * kPrepWaves), (size_t)L * 64 * sizeof(int), s, d, T, clouds && !d.mcica ? 1 : 0, t0);
    if (clouds && d.mcica) hipLaunchKernelGGL(sw_cloud_kernel, dim3(nt, L), blk, 0, s, d, T, t0);
    hipLaunchKernelGGL(tile_lists_kernel, dim3(1), blk, 0, s, d.tile_cld + t0, nt, tlist, tlist + 2 * d.tcap, d.tcap);
    const int ngrp = (nt + kSwWgWaves - 1) / kSwWgWaves;
    const dim3 wg(64 * kSwWgWaves);
    const int ci = t0 / ctile;
    auto clear_variant = [&]() {
      (void)hipEventRecord(ctx->chunk_event(0, ci, 0), s);
      hipLaunchKernelGGL(sw_solve_all_kernel<false>, dim3(ngrp * T.nitem), wg, 0, s, d, T, t0, nt);
      (void)hipEventRecord(ctx->chunk_event(0, ci, 1), s);
    };
    auto cloudy_variant = [&]() {
      (void)hipEventRecord(ctx->chunk_event(2, ci, 0), s);
      hipLaunchKernelGGL(sw_solve_cloudy_kernel, dim3((nt + kC4Waves - 1) / kC4Waves * T.nitem), dim3(64 * kC4Waves), 0, s, d, T, t0, nt);
      (void)hipEventRecord(ctx->chunk_event(2, ci, 1), s);
    };
    // the variant expected to find nothing goes first (order is speed only: each tile belongs to exactly one of them)
    // (a sorted grid -- rrtmg_sort.h -- has its cloud-free tiles first: the chunks in front of the previous call's cloudy-tile count
    //  are expected to hold no cloudy tile)
    const bool expect_clear = clouds && hint_cloudy >= 0 && (hint_cloudy == 0 || (ctx->sorting && t0 + nt <= ntile - hint_cloudy));
    if (expect_clear) { cloudy_variant(); clear_variant(); }
    else { clear_variant(); if (clouds) cloudy_variant(); }
    d.hint_out = t0 + ctile >= ntile ? (int32_t *)&ctx->hint[0].ncloudy : nullptr;
    hipLaunchKernelGGL(sw_fluxheat_kernel, dim3(nt, (L + kFluxLev) / kFluxLev), dim3(64 * (kFluxLev + 1)), 0, s, d, T, t0);
  }
  ctx->hint[0].ntile = ntile; ctx->hint[0].nlay = L;
  ctx->ev_chunks[0] = (ntile + ctile - 1) / ctile; ctx->ev_chunks[2] = clouds ? ctx->ev_chunks[0] : 0;
  RRTMG_HIP_CHECK(ctx, hipGetLastError());

  // ---- status + outputs -------------------------------------------------------------------
  if (ctx->deferred && a->memspace == 1) { ctx->pending[0] = true; ctx->status = 0; return RRTMG_OK; }
  int herr = 0;
  if (a->memspace == 0) {
    const OutCopy oc[6] = {{a->swuflx, d.swuflx, nl1}, {a->swdflx, d.swdflx, nl1}, {a->swuflxc, d.swuflxc, nl1}, {a->swdflxc, d.swdflxc, nl1},
                           {a->swhr, d.swhr, nl}, {a->swhrc, d.swhrc, nl}};
    rc = copy_out(ctx, s, oc, 6, d.err, &herr);
    if (rc) return rc;
  } else {
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(&herr, d.err, sizeof(int), hipMemcpyDeviceToHost, s));
    RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(s));
  }
  if (herr) return ctx->fail(herr, "shortwave: %s", status_message(herr));
  ctx->status = 0;
  return RRTMG_OK;
}

}  // namespace rrtmg
