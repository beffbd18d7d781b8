#!/usr/bin/env bash
# Ingest the REAL longwave k-distribution data: the reference's rrtmg_lw_k_g.f90 (a missing blob in the reference
# checkout this package was built against, /root/reference/.MISSING_LARGE_BLOBS:3).  Run in the build container (flang,
# the reference checkout); nothing of the file is copied into the repository -- what ships is the packed table blob.
#
#   tools/ingest_lw_data.sh /path/to/rrtmg_lw_k_g.f90      (the Fortran data file)
#   tools/ingest_lw_data.sh /path/to/rrtmg_lw.nc           (AER's netCDF form of the same data, read by tools/lw_netcdf.py following
#                                                           rrtmg_lw_read_nc.f90; step 1 is skipped: the tables go into the module
#                                                           arrays of the stub-linked library before rrtmg_lw_ini)
#
#   1. oracle/build_ref.sh lw   compiles the file where it lies into oracle/_ref/librrtmg_lw_ref.so (instead of the empty
#                               loaders of oracle/lw_kg_stub.f90); oracle/_ref/lw_kdata.txt records "file <path> <sha256>"
#   2. tools/pack_tables.py lw  dumps the tables the reference loaded -> climt_amd/data/rrtmg_lw_data.bin with
#                               lw/meta/synthetic = 0, and the reference's reduced tables -> tests/golden/lw_reduced_tables.npz
#   3. tests/golden/make_golden.py   regenerates the longwave fixtures from the reference running on the real tables
#   4. the CPU suite: RRTMGLongwave() no longer needs allow_synthetic_tables, and the four longwave cache classes of the
#      reference's tests are compared at its own criterion, 1e-8 (tests/test_components_host.py, tests/test_gpu_parity.py)
#
# The chain is exercised without the real file by tests/test_lw_ingest.py: a stand-in in the same syntax
# (tools/write_lw_k_g.py) goes through steps 1-2 into a side directory and must reproduce the shipped blob bit for bit.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
KG="${1:?usage: tools/ingest_lw_data.sh <rrtmg_lw_k_g.f90 | rrtmg_lw.nc>}"
[ -f "$KG" ] || { echo "no such file: $KG" >&2; exit 1; }
cd "$ROOT"
case "$KG" in
  *.nc)
    bash oracle/build_ref.sh lw                    # (stub loaders: the netCDF tables are written into the module arrays)
    python tools/pack_tables.py lw --from-nc "$(readlink -f "$KG")" ;;
  *)
    RRTMG_LW_K_G="$(readlink -f "$KG")" bash oracle/build_ref.sh lw
    grep -q '^file ' oracle/_ref/lw_kdata.txt || { echo "build_ref.sh did not link the data file" >&2; exit 1; }
    python tools/pack_tables.py lw ;;
esac
python tests/golden/make_golden.py          # (regenerates every fixture; the shortwave ones come out as they are)
python -m pytest tests -x -q -m "not gpu"
echo "longwave tables ingested from $KG (reference library: $(cat oracle/_ref/lw_kdata.txt))"
echo "now run the GPU suite on an MI355X box:  python -m pytest tests -m gpu -x -q"
