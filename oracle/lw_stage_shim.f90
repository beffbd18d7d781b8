! TEST INFRASTRUCTURE ONLY (oracle).  Our own bind(c) driver around the REFERENCE's longwave stage
! routines inatm -> setcoef -> taumol (rrtmg_lw_rad.nomcica.f90:458-512), linked next to the reference
! objects by oracle/build_ref.sh, so that per-(layer, g-point) optical depths and Planck fractions can
! be checked band by band.  One column per call; arrays are (nlay) / (nlay+1), layer 1 = surface.
subroutine lw_stage_taumol(nlay, play, plev, tlay, tlev, tsfc, h2o, o3, co2, ch4, n2o, o2, &
                           cfc11, cfc12, cfc22, ccl4, emis, taug_out, fracs_out, laytrop_out, pwvcm_out) bind(c)
  use iso_c_binding
  use parkind, only : im => kind_im, rb => kind_rb
  use parrrtm, only : nbndlw, ngptlw, mxmol, maxxsec
  use rrtmg_lw_rad_nomcica, only : inatm
  use rrtmg_lw_setcoef, only : setcoef
  use rrtmg_lw_taumol, only : taumol
  implicit none
  integer(kind=im), intent(in) :: nlay
  real(kind=rb), intent(in) :: play(1,nlay), plev(1,nlay+1), tlay(1,nlay), tlev(1,nlay+1), tsfc(1)
  real(kind=rb), intent(in) :: h2o(1,nlay), o3(1,nlay), co2(1,nlay), ch4(1,nlay), n2o(1,nlay), o2(1,nlay)
  real(kind=rb), intent(in) :: cfc11(1,nlay), cfc12(1,nlay), cfc22(1,nlay), ccl4(1,nlay), emis(1,nbndlw)
  real(kind=rb), intent(out) :: taug_out(nlay,ngptlw), fracs_out(nlay,ngptlw), pwvcm_out
  integer(kind=im), intent(out) :: laytrop_out
  integer(kind=im) :: nlayers, icld, iaer, inflag, iceflag, liqflag, laytrop, idrv, istart
  real(kind=rb) :: cldfr(1,nlay), taucld(nbndlw,1,nlay), cicewp(1,nlay), cliqwp(1,nlay), reice(1,nlay), reliq(1,nlay)
  real(kind=rb) :: tauaer(1,nlay,nbndlw)
  real(kind=rb) :: pavel(nlay+1), tavel(nlay+1), pz(0:nlay+1), tz(0:nlay+1), tbound, coldry(nlay+1), wbrodl(nlay+1)
  real(kind=rb) :: wkl(mxmol,nlay+1), wx(maxxsec,nlay+1), pwvcm, semiss(nbndlw), taua(nlay+1,nbndlw)
  real(kind=rb) :: cldfrac(nlay+1), tauc(nbndlw,nlay+1), ciwp(nlay+1), clwp(nlay+1), rei(nlay+1), rel(nlay+1)
  integer(kind=im) :: jp(nlay+1), jt(nlay+1), jt1(nlay+1), indself(nlay+1), indfor(nlay+1), indminor(nlay+1)
  real(kind=rb) :: planklay(nlay+1,nbndlw), planklev(0:nlay+1,nbndlw), plankbnd(nbndlw), dplankbnd_dt(nbndlw)
  real(kind=rb), dimension(nlay+1) :: colh2o, colco2, colo3, coln2o, colco, colch4, colo2, colbrd, fac00, fac01, fac10, fac11, &
       rat_h2oco2, rat_h2oco2_1, rat_h2oo3, rat_h2oo3_1, rat_h2on2o, rat_h2on2o_1, rat_h2och4, rat_h2och4_1, &
       rat_n2oco2, rat_n2oco2_1, rat_o3co2, rat_o3co2_1, selffac, selffrac, forfac, forfrac, minorfrac, scaleminor, scaleminorn2
  real(kind=rb) :: fracs(nlay+1,ngptlw), taug(nlay+1,ngptlw)
  icld = 0; iaer = 0; idrv = 0; istart = 1
  cldfr = 0._rb; taucld = 0._rb; cicewp = 0._rb; cliqwp = 0._rb; reice = 0._rb; reliq = 0._rb; tauaer = 0._rb
  call inatm (1, nlay, icld, iaer, play, plev, tlay, tlev, tsfc, h2o, o3, co2, ch4, n2o, o2, cfc11, cfc12, &
              cfc22, ccl4, emis, 0, 0, 0, cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer, &
              nlayers, pavel, pz, tavel, tz, tbound, semiss, coldry, wkl, wbrodl, wx, pwvcm, inflag, iceflag, liqflag, &
              cldfrac, tauc, ciwp, clwp, rei, rel, taua)
  call setcoef(nlayers, istart, pavel, tavel, tz, tbound, semiss, coldry, wkl, wbrodl, &
               laytrop, jp, jt, jt1, planklay, planklev, plankbnd, idrv, dplankbnd_dt, &
               colh2o, colco2, colo3, coln2o, colco, colch4, colo2, colbrd, fac00, fac01, fac10, fac11, &
               rat_h2oco2, rat_h2oco2_1, rat_h2oo3, rat_h2oo3_1, rat_h2on2o, rat_h2on2o_1, rat_h2och4, rat_h2och4_1, &
               rat_n2oco2, rat_n2oco2_1, rat_o3co2, rat_o3co2_1, selffac, selffrac, indself, forfac, forfrac, indfor, &
               minorfrac, scaleminor, scaleminorn2, indminor)
  call taumol(nlayers, pavel, wx, coldry, laytrop, jp, jt, jt1, planklay, planklev, plankbnd, &
              colh2o, colco2, colo3, coln2o, colco, colch4, colo2, colbrd, fac00, fac01, fac10, fac11, &
              rat_h2oco2, rat_h2oco2_1, rat_h2oo3, rat_h2oo3_1, rat_h2on2o, rat_h2on2o_1, rat_h2och4, rat_h2och4_1, &
              rat_n2oco2, rat_n2oco2_1, rat_o3co2, rat_o3co2_1, selffac, selffrac, indself, forfac, forfrac, indfor, &
              minorfrac, scaleminor, scaleminorn2, indminor, fracs, taug)
  taug_out(1:nlay,:) = taug(1:nlay,:)
  fracs_out(1:nlay,:) = fracs(1:nlay,:)
  laytrop_out = laytrop
  pwvcm_out = pwvcm
end subroutine lw_stage_taumol
