! The reference's Krylov-vector seam on top of libnekstab_hip.so: `module krylov_subspace` with `type(krylov_vector)` and the
! external subroutines krylov_inner_product / norm / normalize / cmult / add2 / sub2 / zero / copy / matmul and matvec(f, q),
! with the reference's names and argument lists (core/krylov_subspace.f:1-258, core/matvec.f:64-68), so that code written
! against the reference's module -- core/krylov_decomposition.f: arnoldi_factorization, update_hessenberg_matrix -- compiles
! UNCHANGED against this file (host/Makefile target `ref_seam` does exactly that, reading the reference's source from
! /root/reference at build time) and then runs with every vector resident in HBM.
!
! What differs from the reference's module, and why:
!   * krylov_vector holds a device handle (nsk_vec) instead of the arrays vx, vy, vz, pr, theta.  Zero-size components of
!     those names remain so that statements that only NAME them compile (core/krylov_decomposition.f:88 passes f%vx ... to
!     arnoldi_checkpoint); data crosses to the host through krylov_download / krylov_upload below.
!   * a vector is created on first use (krylov_zero / krylov_copy into an unallocated one) and released by a finalizer, so
!     that the local `type(krylov_vector) :: f, wrk` of the reference's routines need no change.
!   * lv, lp, n, n2 are run-time values (the reference fixes them in SIZE).
!   * matvec dispatches on ks_mode (NSK_DIRECT ... NSK_FORCE_SENSITIVITY), set by the caller, instead of uparam(1).
! Compile with explicit kinds here and -fdefault-real-8 for the reference's sources (Nek5000 is built with -r8).
module krylov_subspace
  use iso_c_binding
  use nekstab_hip
  implicit none
  private

  integer, save, public :: lv = 0, lp = 0
  integer, save, public :: n = 0, n2 = 0
  type, public :: krylov_vector
    type(c_ptr) :: h = c_null_ptr
    real(c_double), dimension(0) :: vx, vy, vz, pr
    real(c_double), dimension(0, 0) :: theta
    real(c_double) :: time = 0.0d0
  contains
    final :: krylov_vector_release
  end type krylov_vector

  type(krylov_vector), save, public :: ic_nwt, fc_nwt          ! core/krylov_subspace.f:17
  type(c_ptr), save, public :: ks_ctx = c_null_ptr              ! the library context every vector lives in
  integer(c_int), save, public :: ks_mode = NSK_DIRECT          ! what matvec(f, q) applies
  logical, save, public :: ks_time_component = .false.          ! uparam(1) = 2.1: the inner product adds p%time * q%time (:53-55)
  integer, save, public :: ks_live = 0                          ! vectors currently allocated (leak check of the tests)

  public :: krylov_subspace_attach, krylov_vector_need, krylov_vector_release, krylov_upload, krylov_download

contains

  subroutine krylov_subspace_attach(ctx, mode)
    type(c_ptr), intent(in) :: ctx
    integer(c_int), intent(in) :: mode
    real(c_double) :: dt
    integer(c_int) :: nsteps
    integer(c_long_long) :: nstate, nvel, npres
    ks_ctx = ctx; ks_mode = mode
    call nsk_check(nsk_get_info(ctx, dt, nsteps, nstate, nvel, npres), 'nsk_get_info')
    lv = int(nvel); lp = int(npres); n = lv; n2 = lp
  end subroutine

  subroutine krylov_vector_need(p)                              ! allocate on first use
    type(krylov_vector), intent(inout) :: p
    type(c_ptr) :: v(1)
    if (c_associated(p%h)) return
    call nsk_check(nsk_vec_alloc(ks_ctx, 1_c_int, v), 'nsk_vec_alloc')
    p%h = v(1); ks_live = ks_live + 1
  end subroutine

  impure elemental subroutine krylov_vector_release(p)
    type(krylov_vector), intent(inout) :: p
    type(c_ptr) :: v(1)
    integer(c_int) :: ierr
    if (.not. c_associated(p%h)) return
    v(1) = p%h
    if (c_associated(ks_ctx)) ierr = nsk_vec_free(ks_ctx, 1_c_int, v)
    p%h = c_null_ptr; ks_live = ks_live - 1
  end subroutine

  subroutine krylov_upload(p, vx, vy, pr)                       ! host arrays -> vector (quadrilateral cases)
    type(krylov_vector), intent(inout) :: p
    real(c_double), intent(in) :: vx(*), vy(*), pr(*)
    call krylov_vector_need(p)
    call nsk_check(nsk_vec_upload(ks_ctx, p%h, vx, vy, pr), 'nsk_vec_upload')
  end subroutine

  subroutine krylov_download(p, vx, vy, pr)
    type(krylov_vector), intent(in) :: p
    real(c_double), intent(out) :: vx(*), vy(*), pr(*)
    call nsk_check(nsk_vec_download(ks_ctx, p%h, vx, vy, pr), 'nsk_vec_download')
  end subroutine
end module krylov_subspace


! ---- the reference's external subroutines (core/krylov_subspace.f:24-258), same names and argument lists ------------------

subroutine krylov_inner_product(alpha, p, q)                    ! :24-62: bm1s-weighted velocity (+ scalars) product, pressure excluded
  use krylov_subspace
  use nekstab_hip
  use iso_c_binding
  implicit none
  type(krylov_vector), intent(in) :: p, q
  real(c_double), intent(out) :: alpha
  integer(c_int) :: ierr
  ierr = nsk_dot(ks_ctx, p%h, q%h, alpha)
  if (ierr == NSK_ENAN) then                                    ! :58 `if (isnan(alpha)) call nek_end`
    write(*, *) 'krylov_inner_product: NaN'; call nek_end
  endif
  call nsk_check(ierr, 'nsk_dot')
  if (ks_time_component) alpha = alpha + p%time * q%time
end subroutine krylov_inner_product

subroutine krylov_norm(alpha, p)                                ! :64-75
  use krylov_subspace
  use iso_c_binding
  implicit none
  type(krylov_vector), intent(in) :: p
  real(c_double), intent(out) :: alpha
  call krylov_inner_product(alpha, p, p)
  alpha = dsqrt(alpha)
end subroutine krylov_norm

subroutine krylov_normalize(p, alpha)                           ! :77-94
  use krylov_subspace
  use iso_c_binding
  implicit none
  type(krylov_vector), intent(inout) :: p
  real(c_double), intent(out) :: alpha
  real(c_double) :: inv_alpha
  call krylov_norm(alpha, p)
  inv_alpha = 1.0d0 / alpha
  call krylov_cmult(p, inv_alpha)
end subroutine krylov_normalize

subroutine krylov_cmult(p, alpha)                               ! :96-120 (velocity, pressure, scalars and the time component)
  use krylov_subspace
  use nekstab_hip
  use iso_c_binding
  implicit none
  type(krylov_vector) :: p
  real(c_double) :: alpha
  call nsk_check(nsk_scal(ks_ctx, p%h, alpha), 'nsk_scal')
  p%time = p%time * alpha
end subroutine krylov_cmult

subroutine krylov_add2(p, q)                                    ! :122-144
  use krylov_subspace
  use nekstab_hip
  use iso_c_binding
  implicit none
  type(krylov_vector) :: p, q
  call nsk_check(nsk_axpy(ks_ctx, p%h, 1.0d0, q%h), 'nsk_axpy')
  p%time = p%time + q%time
end subroutine krylov_add2

subroutine krylov_sub2(p, q)                                    ! :147-168
  use krylov_subspace
  use nekstab_hip
  use iso_c_binding
  implicit none
  type(krylov_vector) :: p, q
  call nsk_check(nsk_axpy(ks_ctx, p%h, -1.0d0, q%h), 'nsk_axpy')
  p%time = p%time - q%time
end subroutine krylov_sub2

subroutine krylov_zero(p)                                       ! :170-191
  use krylov_subspace
  use nekstab_hip
  use iso_c_binding
  implicit none
  type(krylov_vector) :: p
  call krylov_vector_need(p)
  call nsk_check(nsk_zero(ks_ctx, p%h), 'nsk_zero')
  p%time = 0.0d0
end subroutine krylov_zero

subroutine krylov_copy(p, q)                                    ! :193-214
  use krylov_subspace
  use nekstab_hip
  use iso_c_binding
  implicit none
  type(krylov_vector) :: p, q
  call krylov_vector_need(p)
  call nsk_check(nsk_copy(ks_ctx, p%h, q%h), 'nsk_copy')
  p%time = q%time
end subroutine krylov_copy

subroutine krylov_matmul(dq, Q, yvec, k)                        ! :216-258: dq = Q(1:k) . yvec, one pass over the basis on the device
  use krylov_subspace
  use nekstab_hip
  use iso_c_binding
  implicit none
  integer :: k, i
  type(krylov_vector) :: dq
  type(krylov_vector), dimension(k) :: Q
  real(c_double), dimension(k) :: yvec
  type(c_ptr) :: hq(k)
  real(c_double) :: yim(k)
  call krylov_vector_need(dq)
  dq%time = 0.0d0
  do i = 1, k
    hq(i) = Q(i)%h
    dq%time = dq%time + Q(i)%time * yvec(i)
  enddo
  yim = 0.0d0
  call nsk_check(nsk_basis_gemv(ks_ctx, hq, int(k, c_int), yvec, yim, dq%h, c_null_ptr), 'nsk_basis_gemv')
end subroutine krylov_matmul

subroutine matvec(f, q)                                         ! core/matvec.f:64-110: f = exp(L T) q (direct / adjoint / ...)
  use krylov_subspace
  use nekstab_hip
  use iso_c_binding
  implicit none
  type(krylov_vector) :: f, q
  call krylov_vector_need(f)
  call nsk_check(nsk_matvec(ks_ctx, ks_mode, f%h, q%h), 'nsk_matvec')
  f%time = q%time
end subroutine matvec
