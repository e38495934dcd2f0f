! The four Nek5000 / nekStab externals core/krylov_decomposition.f calls besides the krylov_* seam: the wall clock, the abort,
! rzero, and arnoldi_checkpoint (core/eigensolvers.f:802; only reached when ifres is set, which the seam driver never does).
function dnekclock() result(t)
  implicit none
  real(8) :: t
  integer(8) :: c, r
  call system_clock(c, r)
  t = real(c, 8) / real(r, 8)
end function

subroutine nek_end
  stop 1
end subroutine

subroutine rzero(a, n)
  implicit none
  integer :: n
  real(8) :: a(n)
  a = 0.0d0
end subroutine

subroutine arnoldi_checkpoint(vx, vy, vz, pr, t, H, k)
  implicit none
  real(8) :: vx(*), vy(*), vz(*), pr(*), t(*), H(*)
  integer :: k
  write(*, *) 'arnoldi_checkpoint: not part of the seam test (ifres = .false.)'
end subroutine
