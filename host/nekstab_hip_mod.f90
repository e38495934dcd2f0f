! iso_c_binding view of include/nekstab_hip.h -- the thin boundary north_star asks for:
! the Arnoldi loop / Hessenberg update / dense eigen-solve stay in Fortran on the host and
! reach the HIP matvec and Krylov BLAS-1 through these entry points.
module nekstab_hip
  use iso_c_binding
  implicit none
  integer(c_int), parameter :: NSK_DIRECT = 0, NSK_ADJOINT = 1, NSK_DIRECT_ADJOINT = 2, NSK_NEWTON = 3

  type, bind(c) :: nsk_case
    integer(c_int) :: ndim, nel, lx1, lxd
    integer(c_long_long) :: nglob
    type(c_ptr) :: x, y, gid, mask, ub, vb, spng, vert
    integer(c_long_long) :: nvert
    real(c_double) :: re, endtime, cfl
    integer(c_int) :: has_outflow
    real(c_double) :: tol_helm, tol_pres
    integer(c_int) :: tol_relative, schwarz_layers, max_helm_iter, max_pres_iter, nproj
    type(c_ptr) :: z = c_null_ptr, wb = c_null_ptr      ! hexahedral cases (ndim = 3) only
  end type

  interface
    integer(c_int) function nsk_init(c, ctx) bind(c, name='nsk_init')
      import
      type(nsk_case), intent(in) :: c
      type(c_ptr), intent(out) :: ctx
    end function
    integer(c_int) function nsk_finalize(ctx) bind(c, name='nsk_finalize')
      import
      type(c_ptr), value :: ctx
    end function
    type(c_ptr) function nsk_last_error() bind(c, name='nsk_last_error')
      import
    end function
    integer(c_int) function nsk_get_info(ctx, dt, nsteps, nstate, nvel, npres) bind(c, name='nsk_get_info')
      import
      type(c_ptr), value :: ctx
      real(c_double) :: dt
      integer(c_int) :: nsteps
      integer(c_long_long) :: nstate, nvel, npres
    end function
    integer(c_int) function nsk_vec_alloc(ctx, n, v) bind(c, name='nsk_vec_alloc')
      import
      type(c_ptr), value :: ctx
      integer(c_int), value :: n
      type(c_ptr) :: v(*)
    end function
    integer(c_int) function nsk_vec_upload(ctx, v, vx, vy, pr) bind(c, name='nsk_vec_upload')
      import
      type(c_ptr), value :: ctx, v
      real(c_double) :: vx(*), vy(*), pr(*)
    end function
    integer(c_int) function nsk_vec_download(ctx, v, vx, vy, pr) bind(c, name='nsk_vec_download')
      import
      type(c_ptr), value :: ctx, v
      real(c_double) :: vx(*), vy(*), pr(*)
    end function
    integer(c_int) function nsk_matvec(ctx, mode, f, q) bind(c, name='nsk_matvec')
      import
      type(c_ptr), value :: ctx, f, q
      integer(c_int), value :: mode
    end function
    integer(c_int) function nsk_norm(ctx, p, alpha) bind(c, name='nsk_norm')
      import
      type(c_ptr), value :: ctx, p
      real(c_double) :: alpha
    end function
    integer(c_int) function nsk_dot(ctx, p, q, alpha) bind(c, name='nsk_dot')
      import
      type(c_ptr), value :: ctx, p, q
      real(c_double) :: alpha
    end function
    integer(c_int) function nsk_scal(ctx, p, alpha) bind(c, name='nsk_scal')
      import
      type(c_ptr), value :: ctx, p
      real(c_double), value :: alpha
    end function
    integer(c_int) function nsk_copy(ctx, dst, src) bind(c, name='nsk_copy')
      import
      type(c_ptr), value :: ctx, dst, src
    end function
    integer(c_int) function nsk_orth(ctx, f, Q, j, h, beta) bind(c, name='nsk_orth')
      import
      type(c_ptr), value :: ctx, f
      type(c_ptr) :: Q(*)
      integer(c_int), value :: j
      real(c_double) :: h(*), beta
    end function
    integer(c_int) function nsk_basis_gemv(ctx, Q, k, yre, yim, re, im) bind(c, name='nsk_basis_gemv')
      import
      type(c_ptr), value :: ctx, re, im
      type(c_ptr) :: Q(*)
      integer(c_int), value :: k
      real(c_double) :: yre(*), yim(*)
    end function
    ! LAPACK from the image's OpenBLAS (SciPy's bundled copy exports scipy_-prefixed symbols);
    ! the reference links MKL/OpenBLAS for the same routine (core/lapack_wrapper.f:173)
    subroutine dgeev(jobvl, jobvr, n, a, lda, wr, wi, vl, ldvl, vr, ldvr, work, lwork, info) bind(c, name='scipy_dgeev_')
      import
      character(kind=c_char) :: jobvl, jobvr
      integer(c_int) :: n, lda, ldvl, ldvr, lwork, info
      real(c_double) :: a(lda,*), wr(*), wi(*), vl(ldvl,*), vr(ldvr,*), work(*)
    end subroutine
  end interface
contains
  subroutine nsk_check(ierr, what)
    integer(c_int), intent(in) :: ierr
    character(*), intent(in) :: what
    character(kind=c_char), pointer :: msg(:)
    integer :: i
    if (ierr /= 0) then
      call c_f_pointer(nsk_last_error(), msg, [512])
      write(*,'(a,a,a,i0,a)', advance='no') ' nsk error in ', what, ' (', ierr, '): '
      do i = 1, 512
        if (msg(i) == c_null_char) exit
        write(*,'(a)', advance='no') msg(i)
      enddo
      write(*,*)
      stop 2
    endif
  end subroutine
end module nekstab_hip
