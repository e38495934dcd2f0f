! iso_c_binding view of include/nekstab_hip.h -- the thin boundary north_star asks for:
! the Arnoldi loop / Hessenberg update / Schur restart / dense eigen-solves stay in Fortran on the host
! (host/krylov_host.f90) and reach the HIP matvec and Krylov BLAS-1 through these entry points.
! EVERY function the header declares is bound here (tests/test_host_cpu.py compares the two lists).
module nekstab_hip
  use iso_c_binding
  implicit none
  integer(c_int), parameter :: NSK_DIRECT = 0, NSK_ADJOINT = 1, NSK_DIRECT_ADJOINT = 2, NSK_NEWTON = 3, NSK_FORCE_SENSITIVITY = 4
  integer(c_int), parameter :: NSK_EINVAL = -1, NSK_EHIP = -2, NSK_ENAN = -3, NSK_ENOCONV = -4, NSK_ENOMEM = -5, NSK_ECOMM = -6

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

  type, bind(c) :: nsk_stats
    integer(c_long_long) :: steps, helm_iters, pres_iters, unconverged
    real(c_double) :: last_helm_res, last_pres_res
    integer(c_long_long) :: max_helm_iter, max_pres_iter, budget_helm, budget_pres, recaptures, retries, capped_solves
    real(c_double) :: worst_cap_ratio
    integer(c_long_long) :: total_capped_solves
    real(c_double) :: total_worst_cap_ratio
    integer(c_long_long) :: total_helm_iters, total_pres_iters, total_steps
    real(c_double) :: recapture_seconds
    integer(c_long_long) :: total_pres_jsum
    real(c_double) :: coarse_bytes_per_solve
    integer(c_long_long) :: step_budget_maps
    real(c_double) :: step_budget_helm_mean, step_budget_pres_mean
    integer(c_long_long) :: tail_maps
    integer(c_long_long) :: zero_arrays
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
    integer(c_int) function nsk_get_info(ctx, dt, nsteps, nstate, nvel, npres) bind(c, name='nsk_get_info')
      import
      type(c_ptr), value :: ctx
      real(c_double) :: dt
      integer(c_int) :: nsteps
      integer(c_long_long) :: nstate
      integer(c_long_long) :: nvel
      integer(c_long_long) :: npres
    end function
    integer(c_int) function nsk_set_nsteps(ctx, nsteps) bind(c, name='nsk_set_nsteps')
      import
      type(c_ptr), value :: ctx
      integer(c_int), value :: nsteps
    end function
    integer(c_int) function nsk_set_tolerances(ctx, tol_helm, tol_pres, relative) bind(c, name='nsk_set_tolerances')
      import
      type(c_ptr), value :: ctx
      real(c_double), value :: tol_helm
      real(c_double), value :: tol_pres
      integer(c_int), value :: relative
    end function
    integer(c_int) function nsk_set_option(ctx, name, val) bind(c, name='nsk_set_option')
      import
      type(c_ptr), value :: ctx
      character(kind=c_char), dimension(*) :: name
      real(c_double), value :: val
    end function
    integer(c_int) function nsk_vec_alloc(ctx, n, v) bind(c, name='nsk_vec_alloc')
      import
      type(c_ptr), value :: ctx
      integer(c_int), value :: n
      type(c_ptr), dimension(*) :: v
    end function
    integer(c_int) function nsk_vec_free(ctx, n, v) bind(c, name='nsk_vec_free')
      import
      type(c_ptr), value :: ctx
      integer(c_int), value :: n
      type(c_ptr), dimension(*) :: v
    end function
    integer(c_int) function nsk_vec_upload(ctx, v, vx, vy, pr) bind(c, name='nsk_vec_upload')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: v
      real(c_double), dimension(*) :: vx
      real(c_double), dimension(*) :: vy
      real(c_double), dimension(*) :: pr
    end function
    integer(c_int) function nsk_vec_download(ctx, v, vx, vy, pr) bind(c, name='nsk_vec_download')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: v
      real(c_double), dimension(*) :: vx
      real(c_double), dimension(*) :: vy
      real(c_double), dimension(*) :: pr
    end function
    integer(c_int) function nsk_vec_upload_scalar(ctx, v, m, theta) bind(c, name='nsk_vec_upload_scalar')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: v
      integer(c_int), value :: m
      real(c_double), dimension(*) :: theta
    end function
    integer(c_int) function nsk_vec_download_scalar(ctx, v, m, theta) bind(c, name='nsk_vec_download_scalar')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: v
      integer(c_int), value :: m
      real(c_double), dimension(*) :: theta
    end function
    integer(c_int) function nsk_vec_upload3(ctx, v, vx, vy, vz, pr) bind(c, name='nsk_vec_upload3')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: v
      real(c_double), dimension(*) :: vx
      real(c_double), dimension(*) :: vy
      real(c_double), dimension(*) :: vz
      real(c_double), dimension(*) :: pr
    end function
    integer(c_int) function nsk_vec_download3(ctx, v, vx, vy, vz, pr) bind(c, name='nsk_vec_download3')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: v
      real(c_double), dimension(*) :: vx
      real(c_double), dimension(*) :: vy
      real(c_double), dimension(*) :: vz
      real(c_double), dimension(*) :: pr
    end function
    integer(c_int) function nsk_matvec(ctx, mode, f, q) bind(c, name='nsk_matvec')
      import
      type(c_ptr), value :: ctx
      integer(c_int), value :: mode
      type(c_ptr), value :: f
      type(c_ptr), value :: q
    end function
    integer(c_int) function nsk_nonlinear_map(ctx, f, q, subtract_q) bind(c, name='nsk_nonlinear_map')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: f
      type(c_ptr), value :: q
      integer(c_int), value :: subtract_q
    end function
    integer(c_int) function nsk_set_baseflow(ctx, q) bind(c, name='nsk_set_baseflow')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: q
    end function
    integer(c_int) function nsk_set_orbit(ctx, q0, spng_str, endv) bind(c, name='nsk_set_orbit')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: q0
      real(c_double), value :: spng_str
      type(c_ptr), value :: endv
    end function
    integer(c_int) function nsk_dot(ctx, p, q, alpha) bind(c, name='nsk_dot')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: p
      type(c_ptr), value :: q
      real(c_double) :: alpha
    end function
    integer(c_int) function nsk_norm(ctx, p, alpha) bind(c, name='nsk_norm')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: p
      real(c_double) :: alpha
    end function
    integer(c_int) function nsk_scal(ctx, p, alpha) bind(c, name='nsk_scal')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: p
      real(c_double), value :: alpha
    end function
    integer(c_int) function nsk_axpy(ctx, p, alpha, q) bind(c, name='nsk_axpy')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: p
      real(c_double), value :: alpha
      type(c_ptr), value :: q
    end function
    integer(c_int) function nsk_copy(ctx, dst, src) bind(c, name='nsk_copy')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: dst
      type(c_ptr), value :: src
    end function
    integer(c_int) function nsk_zero(ctx, p) bind(c, name='nsk_zero')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: p
    end function
    integer(c_int) function nsk_orth(ctx, f, Q, j, h, beta) bind(c, name='nsk_orth')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: f
      type(c_ptr), dimension(*) :: Q
      integer(c_int), value :: j
      real(c_double), dimension(*) :: h
      real(c_double) :: beta
    end function
    integer(c_int) function nsk_basis_gemm(ctx, Q, k, Z, ldz) bind(c, name='nsk_basis_gemm')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), dimension(*) :: Q
      integer(c_int), value :: k
      real(c_double), dimension(*) :: Z
      integer(c_int), value :: ldz
    end function
    integer(c_int) function nsk_basis_gemv(ctx, Q, k, yre, yim, re, im) bind(c, name='nsk_basis_gemv')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), dimension(*) :: Q
      integer(c_int), value :: k
      real(c_double), dimension(*) :: yre
      real(c_double), dimension(*) :: yim
      type(c_ptr), value :: re
      type(c_ptr), value :: im
    end function
    integer(c_int) function nsk_seed_noise(ctx, v) bind(c, name='nsk_seed_noise')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: v
    end function
    integer(c_int) function nsk_clone(ctx, lane) bind(c, name='nsk_clone')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), intent(out) :: lane
    end function
    integer(c_int) function nsk_matvec_batch(lanes, b, mode, f, q) bind(c, name='nsk_matvec_batch')
      import
      type(c_ptr), dimension(*) :: lanes
      integer(c_int), value :: b
      integer(c_int), value :: mode
      type(c_ptr), dimension(*) :: f
      type(c_ptr), dimension(*) :: q
    end function
    integer(c_int) function nsk_get_stats(ctx, s) bind(c, name='nsk_get_stats')
      import
      type(c_ptr), value :: ctx
      type(nsk_stats) :: s
    end function
    integer(c_int) function nsk_shard_create(parent, part, rank, nranks, shard) bind(c, name='nsk_shard_create')
      import
      type(c_ptr), value :: parent
      integer(c_int), dimension(*) :: part
      integer(c_int), value :: rank
      integer(c_int), value :: nranks
      type(c_ptr), intent(out) :: shard
    end function
    integer(c_int) function nsk_init_local(sub, own, ctx) bind(c, name='nsk_init_local')
      import
      type(nsk_case), intent(in) :: sub
      integer(c_int), dimension(*) :: own
      type(c_ptr), intent(out) :: ctx
    end function
    integer(c_int) function nsk_local_info(ctx, vol_own, ctarg, fd_lmax, npr_own, nrows) bind(c, name='nsk_local_info')
      import
      type(c_ptr), value :: ctx
      real(c_double), intent(out) :: vol_own
      real(c_double), intent(out) :: ctarg
      real(c_double), intent(out) :: fd_lmax
      integer(c_long_long), intent(out) :: npr_own
      integer(c_long_long), intent(out) :: nrows
    end function
    integer(c_int) function nsk_local_rows(ctx, u, v, a) bind(c, name='nsk_local_rows')
      import
      type(c_ptr), value :: ctx
      integer(c_int), dimension(*) :: u
      integer(c_int), dimension(*) :: v
      real(c_double), dimension(*) :: a
    end function
    integer(c_int) function nsk_local_finish(ctx, vol, ctarg, fd_lmax, npr_glob, nrows, u, v, a) bind(c, name='nsk_local_finish')
      import
      type(c_ptr), value :: ctx
      real(c_double), value :: vol
      real(c_double), value :: ctarg
      real(c_double), value :: fd_lmax
      integer(c_long_long), value :: npr_glob
      integer(c_long_long), value :: nrows
      integer(c_int), dimension(*) :: u
      integer(c_int), dimension(*) :: v
      real(c_double), dimension(*) :: a
    end function
    integer(c_int) function nsk_shard_create_local(parent, part_sub, elem_glob, rank, nranks, shard) bind(c, name='nsk_shard_create_local')
      import
      type(c_ptr), value :: parent
      integer(c_int), dimension(*) :: part_sub
      integer(c_long_long), dimension(*) :: elem_glob
      integer(c_int), value :: rank
      integer(c_int), value :: nranks
      type(c_ptr), intent(out) :: shard
    end function
    integer(c_int) function nsk_shard_halo_counts(shard, vel, pres_send, pres_recv) bind(c, name='nsk_shard_halo_counts')
      import
      type(c_ptr), value :: shard
      integer(c_int), dimension(*) :: vel, pres_send, pres_recv
    end function
    integer(c_int) function nsk_shard_elems(shard, elems) bind(c, name='nsk_shard_elems')
      import
      type(c_ptr), value :: shard
      integer(c_long_long), dimension(*) :: elems
    end function
    integer(c_int) function nsk_shard_share_stream(shard, leader) bind(c, name='nsk_shard_share_stream')
      import
      type(c_ptr), value :: shard
      type(c_ptr), value :: leader
    end function
    integer(c_int) function nsk_group_matvec(shards, n, mode, f, q) bind(c, name='nsk_group_matvec')
      import
      type(c_ptr), dimension(*) :: shards
      integer(c_int), value :: n
      integer(c_int), value :: mode
      type(c_ptr), dimension(*) :: f
      type(c_ptr), dimension(*) :: q
    end function
    integer(c_int) function nsk_group_nonlinear_map(shards, n, f, q, subtract_q) bind(c, name='nsk_group_nonlinear_map')
      import
      type(c_ptr), dimension(*) :: shards
      integer(c_int), value :: n
      type(c_ptr), dimension(*) :: f
      type(c_ptr), dimension(*) :: q
      integer(c_int), value :: subtract_q
    end function
    integer(c_int) function nsk_group_set_baseflow(shards, n, q) bind(c, name='nsk_group_set_baseflow')
      import
      type(c_ptr), dimension(*) :: shards
      integer(c_int), value :: n
      type(c_ptr), dimension(*) :: q
    end function
    integer(c_int) function nsk_group_set_orbit(shards, n, q0, spng_str, endv) bind(c, name='nsk_group_set_orbit')
      import
      type(c_ptr), dimension(*) :: shards
      integer(c_int), value :: n
      type(c_ptr), dimension(*) :: q0
      real(c_double), value :: spng_str
      type(c_ptr), dimension(*) :: endv
    end function
    integer(c_int) function nsk_shard_release_parent(parent) bind(c, name='nsk_shard_release_parent')
      import
      type(c_ptr), value :: parent
    end function
    integer(c_int) function nsk_comm_init_host(shard, exchange, allreduce, user) bind(c, name='nsk_comm_init_host')
      import
      type(c_ptr), value :: shard
      type(c_funptr), value :: exchange
      type(c_funptr), value :: allreduce
      type(c_ptr), value :: user
    end function
    integer(c_int) function nsk_comm_unique_id(out128) bind(c, name='nsk_comm_unique_id')
      import
      integer(c_signed_char), dimension(*) :: out128
    end function
    integer(c_int) function nsk_comm_init_rccl(shard, id128) bind(c, name='nsk_comm_init_rccl')
      import
      type(c_ptr), value :: shard
      integer(c_signed_char), dimension(*) :: id128
    end function
    integer(c_int) function nsk_allreduce_host(shard, buf, n) bind(c, name='nsk_allreduce_host')
      import
      type(c_ptr), value :: shard
      real(c_double), dimension(*) :: buf
      integer(c_int), value :: n
    end function
    integer(c_int) function nsk_group_test(shards, n, which, vin, vout) bind(c, name='nsk_group_test')
      import
      type(c_ptr), dimension(*) :: shards
      integer(c_int), value :: n
      integer(c_int), value :: which
      type(c_ptr), dimension(*) :: vin
      type(c_ptr), dimension(*) :: vout
    end function
    integer(c_int) function nsk_local_dots(ctx, f, Q, nq, dots) bind(c, name='nsk_local_dots')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: f
      type(c_ptr), dimension(*) :: Q
      integer(c_int), value :: nq
      real(c_double), dimension(*) :: dots
    end function
    integer(c_int) function nsk_project_out(ctx, f, Q, nq, h) bind(c, name='nsk_project_out')
      import
      type(c_ptr), value :: ctx
      type(c_ptr), value :: f
      type(c_ptr), dimension(*) :: Q
      integer(c_int), value :: nq
      real(c_double), dimension(*) :: h
    end function
    integer(c_int) function nsk_get_step_iters(ctx, n, helm, pres, nsteps) bind(c, name='nsk_get_step_iters')
      import
      type(c_ptr), value :: ctx
      integer(c_int), value :: n
      integer(c_int), dimension(*) :: helm, pres
      integer(c_int) :: nsteps
    end function
    integer(c_int) function nsk_bench_kernel(ctx, name, reps, avg_us) bind(c, name='nsk_bench_kernel')
      import
      type(c_ptr), value :: ctx
      character(kind=c_char), dimension(*) :: name
      integer(c_int), value :: reps
      real(c_double) :: avg_us
    end function
    integer(c_int) function nsk_test_axhelm(ctx, u, h1, h2, w) bind(c, name='nsk_test_axhelm')
      import
      type(c_ptr), value :: ctx
      real(c_double), dimension(*) :: u
      real(c_double), value :: h1
      real(c_double), value :: h2
      real(c_double), dimension(*) :: w
    end function
    integer(c_int) function nsk_test_dssum(ctx, u, w) bind(c, name='nsk_test_dssum')
      import
      type(c_ptr), value :: ctx
      real(c_double), dimension(*) :: u
      real(c_double), dimension(*) :: w
    end function
    integer(c_int) function nsk_test_opdiv(ctx, u, v, w) bind(c, name='nsk_test_opdiv')
      import
      type(c_ptr), value :: ctx
      real(c_double), dimension(*) :: u
      real(c_double), dimension(*) :: v
      real(c_double), dimension(*) :: w
    end function
    integer(c_int) function nsk_test_opgradt(ctx, p, ox, oy) bind(c, name='nsk_test_opgradt')
      import
      type(c_ptr), value :: ctx
      real(c_double), dimension(*) :: p
      real(c_double), dimension(*) :: ox
      real(c_double), dimension(*) :: oy
    end function
    integer(c_int) function nsk_test_convect(ctx, adjoint, u, v, ox, oy) bind(c, name='nsk_test_convect')
      import
      type(c_ptr), value :: ctx
      integer(c_int), value :: adjoint
      real(c_double), dimension(*) :: u
      real(c_double), dimension(*) :: v
      real(c_double), dimension(*) :: ox
      real(c_double), dimension(*) :: oy
    end function
    integer(c_int) function nsk_test_eapply(ctx, p, w) bind(c, name='nsk_test_eapply')
      import
      type(c_ptr), value :: ctx
      real(c_double), dimension(*) :: p
      real(c_double), dimension(*) :: w
    end function
    integer(c_int) function nsk_test_helm_solve(ctx, rx, ry, order, ox, oy, iters) bind(c, name='nsk_test_helm_solve')
      import
      type(c_ptr), value :: ctx
      real(c_double), dimension(*) :: rx
      real(c_double), dimension(*) :: ry
      integer(c_int), value :: order
      real(c_double), dimension(*) :: ox
      real(c_double), dimension(*) :: oy
      integer(c_int) :: iters
    end function
    integer(c_int) function nsk_test_pres_solve(ctx, g, w, iters) bind(c, name='nsk_test_pres_solve')
      import
      type(c_ptr), value :: ctx
      real(c_double), dimension(*) :: g
      real(c_double), dimension(*) :: w
      integer(c_int) :: iters
    end function
    integer(c_int) function nsk_test_op3(ctx, which, vin, vout, a, iters) bind(c, name='nsk_test_op3')
      import
      type(c_ptr), value :: ctx
      integer(c_int), value :: which
      real(c_double), dimension(*) :: vin
      real(c_double), dimension(*) :: vout
      integer(c_int), value :: a
      integer(c_int) :: iters
    end function
    type(c_ptr) function nsk_last_error() bind(c, name='nsk_last_error')
      import
    end function
    ! LAPACK from the image's OpenBLAS (SciPy's bundled copy exports scipy_-prefixed symbols); the reference links
    ! MKL / OpenBLAS for the same routines: dgeev core/lapack_wrapper.f:173, dgees :53, dtrsen :119
    subroutine dgeev(jobvl, jobvr, n, a, lda, wr, wi, vl, ldvl, vr, ldvr, work, lwork, info) bind(c, name='scipy_dgeev_')
      import
      character(kind=c_char) :: jobvl, jobvr
      integer(c_int) :: n, lda, ldvl, ldvr, lwork, info
      real(c_double) :: a(lda,*), wr(*), wi(*), vl(ldvl,*), vr(ldvr,*), work(*)
    end subroutine
    subroutine dgees(jobvs, sort, sel, n, a, lda, sdim, wr, wi, vs, ldvs, work, lwork, bwork, info) bind(c, name='scipy_dgees_')
      import
      character(kind=c_char) :: jobvs, sort
      type(c_funptr), value :: sel
      integer(c_int) :: n, lda, sdim, ldvs, lwork, info
      real(c_double) :: a(lda,*), wr(*), wi(*), vs(ldvs,*), work(*)
      integer(c_int) :: bwork(*)
    end subroutine
    subroutine dtrsen(job, compq, sel, n, t, ldt, q, ldq, wr, wi, m, s, sep, work, lwork, iwork, liwork, info) bind(c, name='scipy_dtrsen_')
      import
      character(kind=c_char) :: job, compq
      integer(c_int) :: sel(*), n, ldt, ldq, m, lwork, iwork(*), liwork, info
      real(c_double) :: t(ldt,*), q(ldq,*), wr(*), wi(*), s, sep, work(*)
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
  ! nsk_set_option with a Fortran string
  subroutine nsk_option(ctx, name, val)
    type(c_ptr), intent(in) :: ctx
    character(*), intent(in) :: name
    real(c_double), intent(in) :: val
    character(kind=c_char) :: cname(len_trim(name) + 1)
    integer :: i
    do i = 1, len_trim(name)
      cname(i) = name(i:i)
    enddo
    cname(len_trim(name) + 1) = c_null_char
    call nsk_check(nsk_set_option(ctx, cname, val), 'nsk_set_option '//trim(name))
  end subroutine
end module nekstab_hip
