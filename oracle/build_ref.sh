#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- builds the *reference* RRTMG Fortran (CliMT/climt) into
# oracle/_ref/ straight from the sources where they lie under /root/reference.
# Nothing from the reference is copied into the repository; outputs (.o/.mod/.so) go to
# oracle/_ref/ only, which is git-ignored (but travels to the GPU box with gpurun).
#
# Recipe follows the order of climt/_lib/rrtmg_sw/Makefile:5-42 and
# climt/_lib/rrtmg_lw/Makefile:5-44 (we do not run the reference's own build system).
#
#   oracle/_ref/librrtmg_sw_ref.so   full SW reference (k-data present)      -> pins SW parity
#   oracle/_ref/librrtmg_lw_ref.so   LW reference.  Its k-data file rrtmg_lw_k_g.f90 is a missing blob in the
#                                    reference checkout.  When the file EXISTS -- at its place in the reference
#                                    tree or wherever RRTMG_LW_K_G points -- it is compiled and linked like the
#                                    reference's Makefile does (:44-51, -O0) and oracle/_ref/lw_kdata.txt says
#                                    "file <path> <sha256>".  Otherwise oracle/lw_kg_stub.f90 (empty loaders) is
#                                    linked, lw_kdata.txt says "stub", and tests fill the rrlw_kgNN module arrays
#                                    with SYNTHETIC tables before rrtmg_lw_ini -> LW "algorithm parity on synthetic
#                                    k-tables", physical parity UNPINNED.
#
# Environment: CLIMT_REFERENCE (reference checkout), RRTMG_LW_K_G (the LW data file, if it lives elsewhere),
# RRTMG_REF_OUT (output directory instead of oracle/_ref), FC (flang).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${CLIMT_REFERENCE:-/root/reference}"
FC="${FC:-/opt/rocm/lib/llvm/bin/flang}"
OUT="${RRTMG_REF_OUT:-$HERE/_ref}"
WHAT="${1:-all}"
mkdir -p "$OUT/sw" "$OUT/lw"

if [ ! -d "$REF/climt/_lib/rrtmg_sw" ]; then
  echo "build_ref: reference tree not present at $REF -- keeping prebuilt oracle/_ref" >&2
  exit 0
fi

compile() {  # $1=dir-tag $2=src-dir $3=name $4=optflag
  local tag="$1" src="$2/$3.f90" obj="$OUT/$1/$3.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ]; then
    (cd "$OUT/$tag" && "$FC" -fPIC "$4" -c "$src" -o "$obj" -module-dir "$OUT/$tag" 2>"$obj.log") \
      || { cat "$obj.log" >&2; exit 1; }
  fi
}

build_sw() {
  local S="$REF/climt/_lib/rrtmg_sw"
  local mods="parkind parrrsw rrsw_cld rrsw_con rrsw_kg16 rrsw_kg17 rrsw_kg18 rrsw_kg19 rrsw_kg20 rrsw_kg21
    rrsw_kg22 rrsw_kg23 rrsw_kg24 rrsw_kg25 rrsw_kg26 rrsw_kg27 rrsw_kg28 rrsw_kg29 rrsw_ncpar rrsw_ref
    rrsw_tbl rrsw_vsn rrsw_aer rrsw_wvn"
  local code="rrtmg_sw_cldprop rrtmg_sw_cldprmc rrtmg_sw_taumol rrtmg_sw_vrtqdr rrtmg_sw_reftra rrtmg_sw_spcvmc
    rrtmg_sw_setcoef rrtmg_sw_spcvrt rrtmg_sw_rad.nomcica mcica_random_numbers rrtmg_sw_init
    mcica_subcol_gen_sw rrtmg_sw_rad rrtmg_sw_c_binder"
  for m in $mods; do compile sw "$S" "$m" -O2; done
  # the 3.9 MB data file: -O0 as in the reference Makefile; slow (minutes) -> run beside the rest
  compile sw "$S" rrtmg_sw_k_g -O0 &
  local kpid=$!
  for m in $code; do compile sw "$S" "$m" -O2; done
  wait $kpid
  "$FC" -shared -fPIC -o "$OUT/librrtmg_sw_ref.so" "$OUT"/sw/*.o
  echo "built $OUT/librrtmg_sw_ref.so"
}

build_lw() {
  local S="$REF/climt/_lib/rrtmg_lw"
  local mods="parkind parrrtm rrlw_cld rrlw_con rrlw_kg01 rrlw_kg02 rrlw_kg03 rrlw_kg04 rrlw_kg05 rrlw_kg06
    rrlw_kg07 rrlw_kg08 rrlw_kg09 rrlw_kg10 rrlw_kg11 rrlw_kg12 rrlw_kg13 rrlw_kg14 rrlw_kg15 rrlw_kg16
    rrlw_ncpar rrlw_ref rrlw_tbl rrlw_vsn rrlw_wvn"
  local code="rrtmg_lw_cldprop rrtmg_lw_cldprmc rrtmg_lw_rtrn rrtmg_lw_rtrnmr rrtmg_lw_rtrnmc rrtmg_lw_setcoef
    rrtmg_lw_taumol rrtmg_lw_rad.nomcica mcica_random_numbers rrtmg_lw_init mcica_subcol_gen_lw
    rrtmg_lw_rad rrtmg_lw_c_binder"
  local KG="${RRTMG_LW_K_G:-$S/rrtmg_lw_k_g.f90}"
  if [ -f "$KG" ]; then
    # The data file is there: compiled where it lies, -O0 as in the reference Makefile (:51-52).  It takes minutes, so it
    # runs beside the rest and is kept while the file's hash stays the same.
    local tag="file $KG $(sha256sum "$KG" | cut -d' ' -f1)"
    rm -f "$OUT/lw/lw_kg_stub.o"                       # exactly one definition of lw_kgb01..16 gets linked
    for m in parkind parrrtm rrlw_kg01 rrlw_kg02 rrlw_kg03 rrlw_kg04 rrlw_kg05 rrlw_kg06 rrlw_kg07 rrlw_kg08 rrlw_kg09 \
        rrlw_kg10 rrlw_kg11 rrlw_kg12 rrlw_kg13 rrlw_kg14 rrlw_kg15 rrlw_kg16 rrlw_vsn; do compile lw "$S" "$m" -O2; done
    local kpid=""
    if [ ! -f "$OUT/lw/rrtmg_lw_k_g.o" ] || [ "$(cat "$OUT/lw_kdata.txt" 2>/dev/null)" != "$tag" ]; then
      rm -f "$OUT/lw_kdata.txt"
      (cd "$OUT/lw" && "$FC" -fPIC -O0 -c "$KG" -o "$OUT/lw/rrtmg_lw_k_g.o" -module-dir "$OUT/lw" -I"$OUT/lw") &
      kpid=$!
    fi
    for m in $mods $code; do compile lw "$S" "$m" -O2; done
    if [ -n "$kpid" ]; then wait $kpid; fi
    echo "$tag" > "$OUT/lw_kdata.txt"
  else
    rm -f "$OUT/lw/rrtmg_lw_k_g.o"
    for m in $mods $code; do compile lw "$S" "$m" -O2; done
    # our own file: empty lw_kgb01..16 (data blob missing from the reference checkout)
    (cd "$OUT/lw" && "$FC" -fPIC -O0 -c "$HERE/lw_kg_stub.f90" -o "$OUT/lw/lw_kg_stub.o" -module-dir "$OUT/lw")
    echo "stub" > "$OUT/lw_kdata.txt"
  fi
  # our own stage driver (inatm -> setcoef -> taumol of the reference) for band-by-band checks
  (cd "$OUT/lw" && "$FC" -fPIC -O2 -c "$HERE/lw_stage_shim.f90" -o "$OUT/lw/lw_stage_shim.o" -module-dir "$OUT/lw" -I"$OUT/lw")
  "$FC" -shared -fPIC -o "$OUT/librrtmg_lw_ref.so" "$OUT"/lw/*.o
  echo "built $OUT/librrtmg_lw_ref.so"
}

case "$WHAT" in
  sw) build_sw ;;
  lw) build_lw ;;
  all) build_sw; build_lw ;;
esac
