! TEST INFRASTRUCTURE ONLY (oracle).  Not reference code.
!
! The reference's LW k-distribution data file climt/_lib/rrtmg_lw/rrtmg_lw_k_g.f90 is a
! missing large blob in the reference checkout (.MISSING_LARGE_BLOBS:3); it is the only
! definition of the sixteen loaders lw_kgb01..lw_kgb16 that rrtmg_lw_ini calls
! (rrtmg_lw_init.f90:80-95).  These empty loaders let the rest of the reference LW Fortran
! link so that its *algorithm* can be run on SYNTHETIC k-tables: the test harness writes the
! raw 16-g arrays of modules rrlw_kg01..16 (kao, kbo, selfrefo, forrefo, fracrefao, ...)
! through ctypes before calling rrtmg_lw_ini_wrapper, which then performs the reference's own
! 256->140 g-point reduction on them.  LW physical parity therefore stays UNPINNED.
subroutine lw_kgb01
end subroutine
subroutine lw_kgb02
end subroutine
subroutine lw_kgb03
end subroutine
subroutine lw_kgb04
end subroutine
subroutine lw_kgb05
end subroutine
subroutine lw_kgb06
end subroutine
subroutine lw_kgb07
end subroutine
subroutine lw_kgb08
end subroutine
subroutine lw_kgb09
end subroutine
subroutine lw_kgb10
end subroutine
subroutine lw_kgb11
end subroutine
subroutine lw_kgb12
end subroutine
subroutine lw_kgb13
end subroutine
subroutine lw_kgb14
end subroutine
subroutine lw_kgb15
end subroutine
subroutine lw_kgb16
end subroutine
