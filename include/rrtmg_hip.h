/*
 * rrtmg_hip.h -- C-ABI of librrtmg_hip.so: MI355X (gfx950) RRTMG longwave + shortwave.
 *
 * Drop-in boundary for CliMT/climt's RRTMG path.  Two layers are exported:
 *
 *  (1) REFERENCE-COMPATIBLE ENTRY POINTS -- same symbol names, argument order, pass-by-pointer
 *      scalars and array layouts as the Fortran bind(c) wrappers that climt's Cython shims bind
 *      (climt/_components/rrtmg/sw/_rrtmg_sw.pyx:22-104, lw/_rrtmg_lw.pyx:19-80):
 *        rrtmg_sw_set_constants      rrtmg_sw_c_binder.f90:19-46
 *        rrtmg_sw_ini_wrapper        rrtmg_sw_c_binder.f90:48-57
 *        mcica_subcol_sw_wrapper     rrtmg_sw_c_binder.f90:59-107
 *        rrtmg_sw_mcica_wrapper      rrtmg_sw_c_binder.f90:109-200
 *        rrtmg_sw_nomcica_wrapper    rrtmg_sw_c_binder.f90:202-294
 *        rrtmg_set_constants         rrlw_con.f90:46-71   (the symbol _rrtmg_lw.pyx:20 binds)
 *        rrtmg_lw_set_constants      rrtmg_lw_c_binder.f90:10-37
 *        rrtmg_lw_ini_wrapper        rrtmg_lw_c_binder.f90:39-48
 *        mcica_subcol_lw_wrapper     rrtmg_lw_c_binder.f90:50-92
 *        rrtmg_lw_mcica_wrapper      rrtmg_lw_c_binder.f90:94-174
 *        rrtmg_lw_nomcica_wrapper    rrtmg_lw_c_binder.f90:176-256
 *      They operate on a process-global default context (the reference keeps the same state in
 *      Fortran module variables) and take HOST pointers.  The reference aborts the process
 *      (Fortran `stop`) on invalid input; these record an error instead, retrievable with
 *      rrtmg_hip_default_status() / rrtmg_hip_default_error().
 *
 *  (2) CONTEXT API -- explicit context (one per GPU / per component instance), host or device
 *      pointers, int status returns.  This is what climt_amd's Python host uses via ctypes.
 *
 * Array layout (both layers), identical to the reference boundary:
 *   layer arrays      double[nlay][ncol]      (Fortran (ncol,nlay)); layer 0 = surface
 *   interface arrays  double[nlay+1][ncol]
 *   per-column        double[ncol]
 *   cloud optics      double[nlay][ncol][nbnd]  (Fortran (nbnd,ncol,nlay))
 *   aerosol           double[nbnd][nlay][ncol]  (Fortran (ncol,nlay,nbnd)); ecaer [6][nlay][ncol]
 *   LW emissivity     double[16][ncol]
 *   McICA sub-columns double[nlay][ncol][ngpt]  (Fortran (ngpt,ncol,nlay))
 * No torch types, no ownership transfer: the caller owns every buffer.
 */
#ifndef RRTMG_HIP_H
#define RRTMG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRTMG_NBNDSW 14
#define RRTMG_NGPTSW 112
#define RRTMG_NBNDLW 16
#define RRTMG_NGPTLW 140

/* Status codes: 0 ok.  Every distinct `stop` MESSAGE of the reference (the 61 sites of rrtmg_{sw,lw}_cldprop.f90,
 * rrtmg_{sw,lw}_cldprmc.f90, mcica_subcol_gen_{sw,lw}.f90, rrtmg_sw_rad.nomcica.f90 carry 20 different texts) has its own
 * code, and rrtmg_hip_last_error() ends with the reference's text.  Where the reference stops at the first failed check, a
 * call here returns the first failed check of a (column, layer) in the reference's program order, and the largest such code
 * over the grid (the checks run in parallel); the context stays usable.  Code 13 (rounds 1-4: "any cloud optical property
 * out of range") is no longer returned: 30-41 say which. */
enum {
  RRTMG_OK = 0,
  RRTMG_ERR_HIP = 1,               /* HIP runtime failure (message has the hipError string) */
  RRTMG_ERR_NOT_INITIALISED = 2,   /* *_init not called / tables missing */
  RRTMG_ERR_TABLES = 3,            /* data blob unreadable or malformed */
  RRTMG_ERR_ARG = 4,               /* bad argument (null pointer, nlay<=0, struct_size, ...) */
  RRTMG_ERR_PARTIAL_CLOUD = 10,    /* rrtmg_sw_rad.nomcica.f90:618 'PARTIAL CLOUD NOT ALLOWED' */
  RRTMG_ERR_ICE_RADIUS = 11,       /* 'ICE RADIUS OUT OF BOUNDS' (sw_cldprop:194,226 sw_cldprmc:183,213 lw_cldprop:198,209 lw_cldprmc:189,198) */
  RRTMG_ERR_LIQ_RADIUS = 12,       /* 'LIQUID EFFECTIVE RADIUS OUT OF BOUNDS' (sw_cldprop:290 sw_cldprmc:273 lw_cldprop:253 lw_cldprmc:234) */
  RRTMG_ERR_KISS_PRESSURE = 14,    /* 'MCICA_SUBCOL: KISSVEC SEED GENERATOR REQUIRES PMID FROM BOTTOM FOUR LAYERS.' (mcica_subcol_gen_sw:355, _lw:328) */
  RRTMG_ERR_ICLD = 15,             /* 'MCICA_SUBCOL: INVALID ICLD' (mcica_subcol_gen_sw:145, _lw:122) */
  RRTMG_ERR_INFLAG1_MCICA = 16,    /* 'INFLAG = 1 OPTION NOT AVAILABLE WITH MCICA' (sw_cldprmc:166, lw_cldprmc:172) */
  RRTMG_ERR_ICE_GEN_SIZE = 17,     /* 'ICE GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS' (sw_cldprop:250 sw_cldprmc:236 lw_cldprop:225 lw_cldprmc:212) */
  RRTMG_ERR_ICE_RADIUS_SMALL = 18, /* 'ICE RADIUS TOO SMALL' (lw_cldprop:193, lw_cldprmc:185) */
  RRTMG_ERR_UNSUPPORTED = 20,      /* option without an implementation in RRTMG itself (e.g. shortwave inflag = 1, iceflag = 0) */
  /* shortwave cloud-optics checks behind each parameterisation (sw_cldprop:216-220,240-244,264-265,270-274,307-311;
   * sw_cldprmc:204-208,227-231,250-251,256-260,290-294) */
  RRTMG_ERR_ICE_EXT_NEG = 30,      /* 'ICE EXTINCTION LESS THAN 0.0' */
  RRTMG_ERR_ICE_SSA_GT1 = 31,      /* 'ICE SSA GRTR THAN 1.0' */
  RRTMG_ERR_ICE_SSA_NEG = 32,      /* 'ICE SSA LESS THAN 0.0' */
  RRTMG_ERR_ICE_ASYM_GT1 = 33,     /* 'ICE ASYM GRTR THAN 1.0' */
  RRTMG_ERR_ICE_ASYM_NEG = 34,     /* 'ICE ASYM LESS THAN 0.0' */
  RRTMG_ERR_FDELTA_NEG = 35,       /* 'FDELTA LESS THAN 0.0' */
  RRTMG_ERR_FDELTA_GT1 = 36,       /* 'FDELTA GT THAN 1.0' */
  RRTMG_ERR_LIQ_EXT_NEG = 37,      /* 'LIQUID EXTINCTION LESS THAN 0.0' */
  RRTMG_ERR_LIQ_SSA_GT1 = 38,      /* 'LIQUID SSA GRTR THAN 1.0' */
  RRTMG_ERR_LIQ_SSA_NEG = 39,      /* 'LIQUID SSA LESS THAN 0.0' */
  RRTMG_ERR_LIQ_ASYM_GT1 = 40,     /* 'LIQUID ASYM GRTR THAN 1.0' */
  RRTMG_ERR_LIQ_ASYM_NEG = 41      /* 'LIQUID ASYM LESS THAN 0.0' */
};

typedef struct rrtmg_ctx rrtmg_ctx;

/* ---- context lifecycle -------------------------------------------------------------- */
int rrtmg_hip_create(rrtmg_ctx **out, int device_ordinal);
void rrtmg_hip_destroy(rrtmg_ctx *ctx);
const char *rrtmg_hip_last_error(const rrtmg_ctx *ctx);
const char *rrtmg_hip_version(void);
/* Version of the argument structs below (rrtmg_sw_args, rrtmg_lw_args, rrtmg_slab_args): bumped whenever a field is added.
 * 5 = this header.  A caller can compare it with RRTMG_HIP_ABI_VERSION of the header it was built with; the flux calls
 * check `struct_size` themselves. */
#define RRTMG_HIP_ABI_VERSION 5
int rrtmg_hip_abi_version(void);
/* HIP stream (hipStream_t) the work of this context is enqueued on (longwave uses a second one in deferred mode). */
void *rrtmg_hip_stream(rrtmg_ctx *ctx);
int rrtmg_hip_synchronize(rrtmg_ctx *ctx);
/* Device-side ordering: `other_stream` (a hipStream_t of the caller, e.g. the one an RCCL gather of the outputs is issued on)
 * waits for everything enqueued so far on this context's streams; the host does not block. */
int rrtmg_hip_stream_wait(rrtmg_ctx *ctx, void *other_stream);
/* Deferred mode (off by default).  When on, rrtmg_hip_{sw,lw}_fluxes calls with memspace == 1 (device-resident
 * arrays) return as soon as their kernels are enqueued -- shortwave and longwave on separate streams, so the two
 * overlap on the GPU -- and the device-side error flags (the reference's `stop` conditions) are reported by the
 * next rrtmg_hip_synchronize / rrtmg_hip_set_deferred call instead.  Host-memory calls stay synchronous. */
int rrtmg_hip_set_deferred(rrtmg_ctx *ctx, int on);
/* OPT-IN internal column order for device-resident calls (memspace 1) with clouds (env RRTMG_HIP_SORT_COLUMNS=1 sets it at
 * create): the call runs on an internal copy of its inputs in which the cloud-free columns come first and the cloudy ones behind,
 * each block padded to a 64-column tile, and scatters its outputs back -- so that a cloud-free column never shares a tile
 * (= a solve-kernel variant that computes both sky streams) with a cloudy one.  It pays where cloud-free columns are interleaved
 * with cloudy ones more finely than 64 columns AND the grid is large (>= 32 768 columns); it costs a copy of the inputs (as much
 * device memory again) and, on small grids, a clear-sky launch of its own.  Columns are independent, the kissvec masks are seeded
 * per column: a cloudy column's results are the same bits as without the sort; a cloud-free column that used to sit in a cloudy
 * tile now runs in the clear-sky variant, whose shortwave differs from the cloudy variant's clear-sky stream by ~1e-12 W m^-2 --
 * which is why this is not the default (tile-aligned shards == the whole grid bit for bit only when a column's variant is a
 * function of its tile).  Calls with the Mersenne twister (one positional stream) and host-pointer calls are not sorted. */
int rrtmg_hip_set_column_sort(rrtmg_ctx *ctx, int on);
/* Duration (ms, HIP events recorded on the stream the kernel is launched on) of a solve kernel in the last completed call:
 * which = 0 -> sw_solve_all_kernel<false> (clear-sky tiles), 1 -> lw_solve_all_kernel<false,..>, 2 -> sw_solve_cloudy_kernel,
 * 3 -> lw_solve_all_kernel<true,..>.  A call launches that kernel once per column chunk (RRTMG_HIP_CHUNK_TILES tiles of 64
 * columns; one chunk up to 8192 columns by default); every launch has its own event bracket and the value is their SUM -- the time the
 * kernel took for ALL the call's columns.  rrtmg_hip_kernel_launches returns the number of launches (chunks), so that
 * sum / launches is the average launch duration rocprofv3 reports for that kernel (when the GPU is not shared with another
 * stream: a bracket also contains the time its workgroups waited for compute units another stream's kernel held).
 * RRTMG_ERR_ARG / 0 launches if that kernel was not launched by the last call. */
int rrtmg_hip_kernel_ms(rrtmg_ctx *ctx, int which, double *ms);
int rrtmg_hip_kernel_launches(rrtmg_ctx *ctx, int which);

/* physical constants (cgs, as climt passes them): replaces rrtmg[_sw]_set_constants */
int rrtmg_hip_set_constants(rrtmg_ctx *ctx, double pi, double grav, double planck, double boltz,
                            double clight, double avogad, double alosmt, double gascon,
                            double sbcnst, double secdy);

/* table construction + g-point reduction + upload; blob_path NULL -> "<dir of .so>/../data/rrtmg_{sw,lw}_data.bin"
 * replaces rrtmg_sw_ini (rrtmg_sw_init.f90:47-173) / rrtmg_lw_ini (rrtmg_lw_init.f90:28-175) */
int rrtmg_hip_sw_init(rrtmg_ctx *ctx, double cpdair, const char *blob_path);
int rrtmg_hip_lw_init(rrtmg_ctx *ctx, double cpdair, const char *blob_path);
/* 1 if the loaded LW k-distribution tables are synthetic (reference data file missing) */
int rrtmg_hip_lw_tables_synthetic(const rrtmg_ctx *ctx);

/* read back a reduced table built at init (for tests): name e.g. "sw/kg16/absa"; returns element count, or <0 */
long rrtmg_hip_get_table(rrtmg_ctx *ctx, const char *name, double *out, long capacity);

/* ---- upstream of the shortwave: zenith angle (climt Instellation) ----------------------------- */
/* zenith[i] (radians, clamped to pi/2 on the night side) of column i at `julian_centuries` (days since
 * 2000-01-01 12:00 / 36525): replaces climt/_components/instellation/component.py:85-135 (_instellation_kernel_np; sun
 * position helpers :138-191 are evaluated on the host).  lat/lon in degrees; memspace as in the flux calls. */
int rrtmg_hip_zenith_angle(rrtmg_ctx *ctx, int ncol, int memspace, const double *lat_deg, const double *lon_deg,
                           double julian_centuries, double *zenith);

/* Per-column part of climt's BergerSolarInsolation (climt/_components/berger_solar_insolation.py:671-676):
 * zenith[i] = arccos(cos_mu), insolation[i] = irradiance * cos_mu with cos_mu = sin(lat) sin_delta - cos(lat) cos_delta
 * cos(2 pi (fractional_day + lon / 360)).  lat is used as given (the reference passes degrees into sin/cos, :673);
 * sin_delta / cos_delta of the solar declination and irradiance = S0 / rho^2 come from the host-side orbital series
 * (:579-668, climt_amd/berger.py). */
int rrtmg_hip_solar_insolation(rrtmg_ctx *ctx, int ncol, int memspace, const double *lat, const double *lon, double sin_delta,
                               double cos_delta, double fractional_day, double irradiance, double *zenith, double *insolation);

/* ---- downstream of the radiation path: slab surface energy balance (climt SlabSurface) ---------- */
/* Kernel of climt/_components/slab_surface.py:440-517 (default configuration, include_ekman=False): surface
 * temperature tendency (K s^-1) and slab depth (m) per column.  The four flux pointers are the SURFACE rows (row 0 of
 * the [level][column] radiation outputs); area_type codes: land 0, land_ice 1, sea 2, sea_ice 3 (slab_surface.py:9). */
typedef struct rrtmg_slab_args {
  const double *sw_down, *lw_down, *sw_up, *lw_up;      /* surface fluxes W m^-2 */
  const double *lh, *sh;                                /* surface upward latent / sensible heat flux */
  const int32_t *area_type;
  const double *up_heat_soil, *heat_flux_sea_ice, *sea_water_dens, *surf_dens, *heat_cap_soil, *surf_therm_cap;
  const double *ocean_mix_thick, *soil_layer_thick, *ocean_heat_transport;
  double *tend_ts, *depth;                              /* outputs */
} rrtmg_slab_args;
int rrtmg_hip_slab_surface(rrtmg_ctx *ctx, int ncol, int memspace, const rrtmg_slab_args *args);

/* ---- glue of a device-resident radiation step (device pointers only; enqueued on the context's main stream) ----------
 * The numpy that climt's component classes run on the host between the kernels, so that a model loop can stay in HBM. */
/* interface values of a mid-level quantity by log-pressure interpolation, [nlay+1][ncol]: climt/_core/util.py:89-142 */
int rrtmg_hip_interface_values(rrtmg_ctx *ctx, int ncol, int nlay, const double *mid, const double *surf, const double *pmid,
                               const double *pint, double *out);
/* op 0: out = alpha*a (+ beta*b if b != NULL); op 1: out = cos(a) (sw/component.py:567); op 2: out = a*alpha/beta
 * (mass_to_volume_mixing_ratio, util.py:86, with its two roundings) */
int rrtmg_hip_elementwise(rrtmg_ctx *ctx, int op, long n, const double *a, const double *b, double alpha, double beta, double *out);
/* Adams-Bashforth update out = x + dt * sum_k w[k] f[k], k < order <= 4 (f, w: host arrays of device pointers / weights) */
int rrtmg_hip_ab_step(rrtmg_ctx *ctx, long n, int order, const double *x, const double *const *f, const double *w, double dt, double *out);
/* deferred mode: 0 = the longwave stream waits for the main stream's work so far, 1 = the main stream waits for the longwave's */
int rrtmg_hip_order_streams(rrtmg_ctx *ctx, int direction);
/* Strided copies of nblk two-dimensional blocks of doubles on the device, enqueued on `stream` (a hipStream_t of the caller;
 * NULL = the context's main stream): block b copies rows x cols elements, dst[dst_off + r*dst_stride + c] = src[src_off +
 * r*src_stride + c].  desc: DEVICE array of 6 int64 per block {src_off, dst_off, rows, cols, src_stride, dst_stride};
 * max_rows / max_cols bound the launch.  Used to put an all-gathered output buffer -- [rank][array][level][local column] --
 * into the boundary layout [array][level][column] (column fastest, rrtmg_lw_c_binder.f90:198-202) without leaving the GPU:
 * climt_amd/distributed.py. */
int rrtmg_hip_copy_blocks(rrtmg_ctx *ctx, int nblk, const int64_t *desc, long max_rows, long max_cols, const double *src, double *dst,
                          void *stream);

/* ---- shortwave ------------------------------------------------------------------------ */
typedef struct rrtmg_sw_args {
  int32_t ncol, nlay;
  int32_t memspace;     /* 0: all pointers are host memory; 1: all pointers are device memory */
  int32_t mcica;        /* 0: rrtmg_sw_rad.nomcica.f90 path; 1: McICA path (rrtmg_sw_rad.f90) */
  int32_t icld, iaer;   /* as the reference (icld 0..3; iaer 0/6/10) */
  int32_t inflgsw, iceflgsw, liqflgsw;
  int32_t dyofyr, isolvar;
  int32_t irng;         /* McICA RNG: 0 kissvec, 1 Mersenne twister */
  int32_t permuteseed;  /* McICA changeSeed */
  /* Column shard of a larger grid (multi-GPU): this call's columns are columns shard_col0 .. shard_col0+ncol-1 of a grid of
   * shard_ncol columns; 0, 0 = not sharded.  Only the Mersenne twister needs it: the reference draws ONE stream in
   * (sub-column, column, layer) order (mcica_subcol_gen_sw.f90:360-367), so a shard skips the other shards' draws and
   * reproduces the unsharded masks bit for bit.  kissvec seeds are per column and ignore it. */
  int32_t shard_col0, shard_ncol;
  /* sizeof(rrtmg_sw_args) of the header the CALLER was compiled against: REQUIRED.  Any value that is not the library's own
   * sizeof -- 0 included -- is refused with RRTMG_ERR_ARG: a caller built against another header never has fields dropped
   * or read past its struct.  (This slot was `reserved0` = 0 in two earlier layouts, with and without the unit factors at
   * the end; a zero cannot tell them apart, so it is not guessed.) */
  int32_t struct_size;
  double adjes, scon, solcycfrac;
  const double *bndsolvar;   /* [14] (host) or NULL -> ones */
  double *indsolvar;         /* [2]  (host) or NULL -> ones; IN/OUT: amplitudes != 1 are rescaled in place once per
                              * column, as the reference does (rrtmg_sw_rad.nomcica.f90:1199-1215) */
  /* state */
  const double *play, *plev, *tlay, *tlev, *tsfc;
  const double *h2ovmr, *o3vmr, *co2vmr, *ch4vmr, *n2ovmr, *o2vmr;
  const double *asdir, *asdif, *aldir, *aldif, *coszen;
  /* clouds (NULL allowed when icld == 0; optics arrays NULL allowed when inflgsw != 0) */
  const double *cldfr;
  const double *taucld, *ssacld, *asmcld, *fsfcld;   /* [nlay][ncol][14] */
  const double *cicewp, *cliqwp, *reice, *reliq;
  /* aerosol (NULL allowed when iaer == 0) */
  const double *tauaer, *ssaaer, *asmaer;            /* [14][nlay][ncol] */
  const double *ecaer;                               /* [6][nlay][ncol]  */
  /* McICA: optional externally generated sub-column cloud mask, [nlay][ncol][112] of 0.0/1.0
   * (e.g. cldfmcl from mcica_subcol_sw_wrapper). NULL -> generated on the device by either
   * generator (irng 0 kissvec, 1 Mersenne twister: its one stream by jump-ahead). */
  const double *cldfmcl;
  /* outputs */
  double *swuflx, *swdflx, *swhr, *swuflxc, *swdflxc, *swhrc;
  /* Unit factors for HOST arrays (memspace 0), applied by the library on the device after the upload instead of by the
   * caller on the host; 0 = the array is in the unit of the reference already.  play, plev *= pressure_scale (Pa -> mbar:
   * 0.01); cicewp, cliqwp *= water_path_scale (kg m^-2 -> g m^-2: 1000); h2ovmr = h2ovmr * h2o_mul / h2o_div (specific
   * humidity -> volume mixing ratio: 28.964 / 18.02, climt/_core/util.py:86).  One rounding per operation, as numpy.
   * A struct that was zero-initialised gets none of it.  With device
   * pointers (memspace 1) a non-zero factor is an error (RRTMG_ERR_ARG): the caller's device arrays are never modified. */
  double pressure_scale, water_path_scale, h2o_mul, h2o_div;
} rrtmg_sw_args;

int rrtmg_hip_sw_fluxes(rrtmg_ctx *ctx, const rrtmg_sw_args *a);

/* ---- longwave ------------------------------------------------------------------------- */
typedef struct rrtmg_lw_args {
  int32_t ncol, nlay;
  int32_t memspace;
  int32_t mcica;
  int32_t icld, idrv;
  int32_t inflglw, iceflglw, liqflglw;
  int32_t irng, permuteseed;
  int32_t shard_col0, shard_ncol;                    /* see rrtmg_sw_args */
  int32_t struct_size;                               /* sizeof(rrtmg_lw_args) of the caller's header: required (see rrtmg_sw_args) */
  const double *play, *plev, *tlay, *tlev, *tsfc;      /* tlev NULL: interpolated on the device from tlay, tsfc, play, plev as
                                                         * climt's get_interface_values does (util.py:89-142) */
  const double *h2ovmr, *o3vmr, *co2vmr, *ch4vmr, *n2ovmr, *o2vmr;
  const double *cfc11vmr, *cfc12vmr, *cfc22vmr, *ccl4vmr;
  const double *emis;                                /* [16][ncol] */
  const double *cldfr;
  const double *taucld;                              /* [nlay][ncol][16] */
  const double *cicewp, *cliqwp, *reice, *reliq;
  const double *tauaer;                              /* [16][nlay][ncol]; NULL -> 0 */
  const double *cldfmcl;                             /* optional [nlay][ncol][140] mask */
  double *uflx, *dflx, *hr, *uflxc, *dflxc, *hrc;
  double *duflx_dt, *duflxc_dt;                      /* idrv==1 only, [nlay+1][ncol] */
  double pressure_scale, water_path_scale, h2o_mul, h2o_div;   /* see rrtmg_sw_args */
} rrtmg_lw_args;

int rrtmg_hip_lw_fluxes(rrtmg_ctx *ctx, const rrtmg_lw_args *a);

/* sub-column generators on their own (mcica_subcol_gen_{sw,lw}.f90); host pointers.
 * which: 0 = SW (112 sub-columns), 1 = LW (140).  cldfmcl out: [nlay][ncol][ngpt] of 0/1. */
int rrtmg_hip_mcica_mask(rrtmg_ctx *ctx, int which, int ncol, int nlay, int icld, int permuteseed,
                         int irng, const double *play, const double *cldfrac, double *cldfmcl);

/* ---- reference-compatible entry points (host pointers, default context) ---------------- */
int rrtmg_hip_default_status(void);
const char *rrtmg_hip_default_error(void);

void rrtmg_sw_set_constants(double *pi, double *grav, double *planck, double *boltz, double *clight,
                            double *avogad, double *alosmt, double *gascon, double *sbcnst, double *secdy);
void rrtmg_sw_ini_wrapper(double *cpdair);
void mcica_subcol_sw_wrapper(int32_t *iplon, int32_t *ncol, int32_t *nlay, int32_t *icld,
                             int32_t *permuteseed, int32_t *irng, double *play, double *cldfrac,
                             double *ciwp, double *clwp, double *rei, double *rel, double *tauc,
                             double *ssac, double *asmc, double *fsfc, double *cldfmcl,
                             double *ciwpmcl, double *clwpmcl, double *reicmcl, double *relqmcl,
                             double *taucmcl, double *ssacmcl, double *asmcmcl, double *fsfcmcl);
void rrtmg_sw_mcica_wrapper(int32_t *ncol, int32_t *nlay, int32_t *icld, int32_t *iaer, double *play,
                            double *plev, double *tlay, double *tlev, double *tsfc, double *h2ovmr,
                            double *o3vmr, double *co2vmr, double *ch4vmr, double *n2ovmr, double *o2vmr,
                            double *asdir, double *asdif, double *aldir, double *aldif, double *coszen,
                            double *adjes, int32_t *dyofyr, double *scon, int32_t *isolvar,
                            int32_t *inflgsw, int32_t *iceflgsw, int32_t *liqflgsw, double *cldfmcl,
                            double *taucmcl, double *ssacmcl, double *asmcmcl, double *fsfcmcl,
                            double *ciwpmcl, double *clwpmcl, double *reicmcl, double *relqmcl,
                            double *tauaer, double *ssaaer, double *asmaer, double *ecaer,
                            double *swuflx, double *swdflx, double *swhr, double *swuflxc,
                            double *swdflxc, double *swhrc, double *bndsolvar, double *indsolvar,
                            double *solcycfrac);
void rrtmg_sw_nomcica_wrapper(int32_t *ncol, int32_t *nlay, int32_t *icld, int32_t *iaer, double *play,
                              double *plev, double *tlay, double *tlev, double *tsfc, double *h2ovmr,
                              double *o3vmr, double *co2vmr, double *ch4vmr, double *n2ovmr, double *o2vmr,
                              double *asdir, double *asdif, double *aldir, double *aldif, double *coszen,
                              double *adjes, int32_t *dyofyr, double *scon, int32_t *isolvar,
                              int32_t *inflgsw, int32_t *iceflgsw, int32_t *liqflgsw, double *cldfr,
                              double *taucld, double *ssacld, double *asmcld, double *fsfcld,
                              double *cicewp, double *cliqwp, double *reice, double *reliq,
                              double *tauaer, double *ssaaer, double *asmaer, double *ecaer,
                              double *swuflx, double *swdflx, double *swhr, double *swuflxc,
                              double *swdflxc, double *swhrc, double *bndsolvar, double *indsolvar,
                              double *solcycfrac);

void rrtmg_set_constants(double *pi, double *grav, double *planck, double *boltz, double *clight,
                         double *avogad, double *alosmt, double *gascon, double *sbcnst, double *secdy);
void rrtmg_lw_set_constants(double *pi, double *grav, double *planck, double *boltz, double *clight,
                            double *avogad, double *alosmt, double *gascon, double *sbcnst, double *secdy);
void rrtmg_lw_ini_wrapper(double *cpdair);
void mcica_subcol_lw_wrapper(int32_t *iplon, int32_t *ncol, int32_t *nlay, int32_t *icld,
                             int32_t *permuteseed, int32_t *irng, double *play, double *cldfrac,
                             double *ciwp, double *clwp, double *rei, double *rel, double *tauc,
                             double *cldfmcl, double *ciwpmcl, double *clwpmcl, double *reicmcl,
                             double *relqmcl, double *taucmcl);
void rrtmg_lw_mcica_wrapper(int32_t *ncol, int32_t *nlay, int32_t *icld, int32_t *idrv, double *play,
                            double *plev, double *tlay, double *tlev, double *tsfc, double *h2ovmr,
                            double *o3vmr, double *co2vmr, double *ch4vmr, double *n2ovmr, double *o2vmr,
                            double *cfc11vmr, double *cfc12vmr, double *cfc22vmr, double *ccl4vmr,
                            double *emis, int32_t *inflglw, int32_t *iceflglw, int32_t *liqflglw,
                            double *cldfmcl, double *taucmcl, double *ciwpmcl, double *clwpmcl,
                            double *reicmcl, double *relqmcl, double *tauaer, double *uflx, double *dflx,
                            double *hr, double *uflxc, double *dflxc, double *hrc, double *duflx_dt,
                            double *duflxc_dt);
void rrtmg_lw_nomcica_wrapper(int32_t *ncol, int32_t *nlay, int32_t *icld, int32_t *idrv, double *play,
                              double *plev, double *tlay, double *tlev, double *tsfc, double *h2ovmr,
                              double *o3vmr, double *co2vmr, double *ch4vmr, double *n2ovmr, double *o2vmr,
                              double *cfc11vmr, double *cfc12vmr, double *cfc22vmr, double *ccl4vmr,
                              double *emis, int32_t *inflglw, int32_t *iceflglw, int32_t *liqflglw,
                              double *cldfr, double *taucld, double *cicewp, double *cliqwp,
                              double *reice, double *reliq, double *tauaer, double *uflx, double *dflx,
                              double *hr, double *uflxc, double *dflxc, double *hrc, double *duflx_dt,
                              double *duflxc_dt);

#ifdef __cplusplus
}
#endif
#endif /* RRTMG_HIP_H */
