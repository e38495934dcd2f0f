! Fortran host of the eigensolver on top of libnekstab_hip.so.  Restates, call for call,
!   arnoldi_factorization      core/krylov_decomposition.f:7-104
!   update_hessenberg_matrix   core/krylov_decomposition.f:116-202   (one nsk_orth call)
!   krylov_schur, schur_tgt<=0 core/eigensolvers.f:141-388 (the committed cylinder example: 1cyl.usr:22)
!   eig + sort                 core/lapack_wrapper.f:129-251
!   Spectre_H / Spectre_NS     core/eigensolvers.f:572-604
! Usage:  arnoldi_host <case.bin> <k_dim> <outdir>
! case.bin is written by nekstab_amd/casefile.py (mesh arrays, base flow, seed vector).
program arnoldi_host
  use iso_c_binding
  use nekstab_hip
  implicit none
  character(len=512) :: casefile, outdir, arg
  integer :: k_dim, u, i, j, mstep
  integer(c_int) :: hdr(8), ierr, nsteps
  integer(c_long_long) :: nglob, nstate, nvel, npres
  real(c_double) :: rpar(3), dt, alpha, beta, t0, t1
  real(c_double), allocatable, target :: x(:), y(:), mask(:), ub(:), vb(:), spng(:), sx(:), sy(:), sp(:)
  integer(c_long_long), allocatable, target :: gid(:), vert(:)
  type(nsk_case) :: cs
  type(c_ptr) :: ctx
  type(c_ptr), allocatable :: Q(:)
  real(c_double), allocatable :: H(:,:), hcol(:)
  complex(c_double_complex), allocatable :: vals(:), vecs(:,:)
  real(c_double), allocatable :: residual(:)
  integer :: nloc, np2

  call get_command_argument(1, casefile)
  call get_command_argument(2, arg); read(arg, *) k_dim
  call get_command_argument(3, outdir)
  if (k_dim == 0) then                                  ! core/krylov_decomposition.f:64-67
    write(*,*) 'Krylov base dimension == 0! Increase it.. STOP'; stop 1
  endif

  open(newunit=u, file=trim(casefile), access='stream', form='unformatted', status='old')
  read(u) hdr; read(u) nglob; read(u) rpar
  nloc = hdr(2) * hdr(3) * hdr(3); np2 = hdr(2) * (hdr(3) - 2) * (hdr(3) - 2)
  allocate(x(nloc), y(nloc), gid(nloc), mask(nloc), ub(nloc), vb(nloc), spng(nloc), vert(4 * hdr(2)))
  allocate(sx(nloc), sy(nloc), sp(np2))
  read(u) x; read(u) y; read(u) gid; read(u) mask; read(u) ub; read(u) vb; read(u) spng; read(u) vert
  read(u) sx; read(u) sy; read(u) sp
  close(u)

  cs%ndim = hdr(1); cs%nel = hdr(2); cs%lx1 = hdr(3); cs%lxd = hdr(4); cs%nglob = nglob
  cs%x = c_loc(x); cs%y = c_loc(y); cs%gid = c_loc(gid); cs%mask = c_loc(mask)
  cs%ub = c_loc(ub); cs%vb = c_loc(vb); cs%spng = c_loc(spng); cs%vert = c_loc(vert); cs%nvert = hdr(5)
  cs%re = rpar(1); cs%endtime = rpar(2); cs%cfl = rpar(3); cs%has_outflow = hdr(6)
  cs%tol_helm = 1.0d-11; cs%tol_pres = 1.0d-1; cs%tol_relative = 1
  cs%schwarz_layers = 2; cs%max_helm_iter = 100; cs%max_pres_iter = 48; cs%nproj = 8
  call nsk_check(nsk_init(cs, ctx), 'nsk_init')
  call nsk_check(nsk_get_info(ctx, dt, nsteps, nstate, nvel, npres), 'nsk_get_info')
  write(*,'(a,es14.6,a,i0,a,i0)') ' dt = ', dt, '  nsteps = ', nsteps, '  state = ', nstate

  ! ----- allocate(Q(k_dim+1)), H(k_dim+1,k_dim)          core/eigensolvers.f:170-171
  allocate(Q(k_dim + 1), H(k_dim + 1, k_dim), hcol(k_dim))
  allocate(vals(k_dim), vecs(k_dim, k_dim), residual(k_dim))
  H = 0.0d0
  call nsk_check(nsk_vec_alloc(ctx, int(k_dim + 1, c_int), Q), 'nsk_vec_alloc')
  ! ----- seed: normalised to unit norm                   core/eigensolvers.f:263-282
  call nsk_check(nsk_vec_upload(ctx, Q(1), sx, sy, sp), 'nsk_vec_upload')
  call nsk_check(nsk_norm(ctx, Q(1), alpha), 'nsk_norm')
  call nsk_check(nsk_scal(ctx, Q(1), 1.0d0 / alpha), 'nsk_scal')

  ! ----- arnoldi_factorization(Q, H, 1, k_dim, k_dim)    core/krylov_decomposition.f:73-102
  call cpu_time(t0)
  do mstep = 1, k_dim
    call nsk_check(nsk_matvec(ctx, NSK_DIRECT, Q(mstep + 1), Q(mstep)), 'nsk_matvec')          ! :80
    call nsk_check(nsk_orth(ctx, Q(mstep + 1), Q, int(mstep, c_int), hcol, beta), 'nsk_orth')   ! :83
    H(1:mstep, mstep) = hcol(1:mstep)
    H(mstep + 1, mstep) = beta
  enddo
  call cpu_time(t1)
  write(*,'(a,i0,a,f8.3,a)') ' Arnoldi: ', k_dim, ' matvecs in ', t1 - t0, ' s (cpu time of the host thread)'

  ! ----- eig(H(1:k,1:k)); residual = |H(k+1,k) * vecs(k,:)|      core/eigensolvers.f:346-349
  call eig(H(1:k_dim, 1:k_dim), vecs, vals, k_dim)
  residual = abs(H(k_dim + 1, k_dim) * vecs(k_dim, :))

  ! ----- outpost_ks: spectra tables                      core/eigensolvers.f:572-604
  open(newunit=u, file=trim(outdir)//'/Spectre_Hd.dat', status='replace')
  do i = 1, k_dim
    write(u, '(3E15.7)') real(vals(i)), aimag(vals(i)), residual(i)
  enddo
  close(u)
  open(newunit=u, file=trim(outdir)//'/Spectre_NSd.dat', status='replace')
  do i = 1, k_dim
    write(u, '(3E15.7)') real(log(vals(i))) / rpar(2), aimag(log(vals(i))) / rpar(2), residual(i)   ! log_transform :908-915
  enddo
  close(u)
  open(newunit=u, file=trim(outdir)//'/ritz_full.txt', status='replace')
  do i = 1, k_dim
    write(u, '(3ES26.17)') real(vals(i)), aimag(vals(i)), residual(i)
  enddo
  close(u)
  open(newunit=u, file=trim(outdir)//'/HES.txt', status='replace')      ! arnoldi_checkpoint :889
  write(u, *) ((H(i, j), j = 1, k_dim), i = 1, k_dim + 1)
  close(u)
  call nsk_check(nsk_finalize(ctx), 'nsk_finalize')

contains

  subroutine eig(A, vecs, vals, n)
    ! dgeev('N','V') + complex pair assembly + sort by decreasing modulus  (core/lapack_wrapper.f:129-251)
    integer, intent(in) :: n
    real(c_double), intent(in) :: A(n, n)
    complex(c_double_complex), intent(out) :: vecs(n, n), vals(n)
    real(c_double) :: a2(n, n), wr(n), wi(n), vl(1, n), vr(n, n), work(8 * n)
    integer(c_int) :: info, nn, lw, one
    integer :: i, jj, imax
    complex(c_double_complex) :: tv, tcol(n)
    a2 = A; nn = n; lw = 8 * n; one = 1
    call dgeev('N', 'V', nn, a2, nn, wr, wi, vl, one, vr, nn, work, lw, info)
    vals = cmplx(wr, wi, kind=c_double_complex)
    i = 1
    do while (i <= n)
      if (wi(i) == 0.0d0) then
        vecs(:, i) = cmplx(vr(:, i), 0.0d0, kind=c_double_complex); i = i + 1
      else
        vecs(:, i) = cmplx(vr(:, i), vr(:, i + 1), kind=c_double_complex)
        vecs(:, i + 1) = cmplx(vr(:, i), -vr(:, i + 1), kind=c_double_complex); i = i + 2
      endif
    enddo
    do i = 1, n - 1                                   ! sort_eigendecomp: decreasing |lambda|
      imax = i
      do jj = i + 1, n
        if (abs(vals(jj)) > abs(vals(imax))) imax = jj
      enddo
      if (imax /= i) then
        tv = vals(i); vals(i) = vals(imax); vals(imax) = tv
        tcol = vecs(:, i); vecs(:, i) = vecs(:, imax); vecs(:, imax) = tcol
      endif
    enddo
  end subroutine

end program arnoldi_host
