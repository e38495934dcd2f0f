! Host side of the eigensolver in the reference's language, over the C-ABI (north_star: "the outer Arnoldi loop, Hessenberg
! update and Schur restart stay in Fortran on the host").  Restates, routine for routine,
!   arnoldi_factorization      core/krylov_decomposition.f:7-104
!   update_hessenberg_matrix   core/krylov_decomposition.f:116-202   (one nsk_orth call: both passes + normalisation on the device)
!   krylov_schur               core/eigensolvers.f:141-388
!   schur_condensation         core/eigensolvers.f:395-499           (Q <- Q Z through nsk_basis_gemm)
!   select_eigenvalues         core/eigensolvers.f:729-795
!   outpost_ks                 core/eigensolvers.f:508-721           (tables; eigenmodes Q y_i through nsk_basis_gemv)
!   eig / schur / ordschur     core/lapack_wrapper.f:7-251           (dgeev / dgees / dtrsen of the image's OpenBLAS)
! The Krylov basis lives on the device (opaque handles); the host holds H(k+1,k) and the k x k factorisations, which is
! what the reference replicates on every MPI rank.
module krylov_host
  use iso_c_binding
  use nekstab_hip
  use nek_fld
  implicit none
  private
  public :: arnoldi_factorization, krylov_schur, schur_condensation, schur_restart_dense, select_eigenvalues, eig, schur, ordschur, outpost_ks

contains

  ! ---- core/krylov_decomposition.f:73-102.  With `geom` + `outdir` present and ifres set, every step writes the reference's
  ! checkpoint (core/krylov_decomposition.f:88 -> arnoldi_checkpoint, core/eigensolvers.f:802-905: nek_fld.f90)
  subroutine arnoldi_factorization(ctx, Q, H, mstart, mend, ksize, mode, geom, outdir, evop, sampling_period, nsteps, ifres)
    type(c_ptr), intent(in) :: ctx
    integer, intent(in) :: mstart, mend, ksize
    type(c_ptr), intent(inout) :: Q(ksize + 1)
    real(c_double), intent(inout) :: H(ksize + 1, ksize)
    integer(c_int), intent(in) :: mode
    type(host_geom), intent(in), optional :: geom
    character(*), intent(in), optional :: outdir, evop
    real(c_double), intent(in), optional :: sampling_period
    integer, intent(in), optional :: nsteps
    logical, intent(in), optional :: ifres
    real(c_double) :: hcol(ksize), beta, telapsed, tmiss
    complex(c_double_complex), allocatable :: cvals(:), cvecs(:, :)
    integer :: mstep
    integer(8) :: tick0, tick1, tickrate
    if (ksize == 0) then                                   ! :64-67
      write(*,*) 'Krylov base dimension == 0! Increase it.. STOP'; stop 1
    endif
    do mstep = mstart, mend
      write(*,*) 'iteration current and total:', mstep, '/', mend                                  ! :75
      call system_clock(tick0, tickrate)                                                           ! :77 eetime0 = dnekclock()
      call nsk_check(nsk_matvec(ctx, mode, Q(mstep + 1), Q(mstep)), 'nsk_matvec')                 ! :80
      call nsk_check(nsk_orth(ctx, Q(mstep + 1), Q, int(mstep, c_int), hcol, beta), 'nsk_orth')    ! :83 update_hessenberg_matrix
      H(1:mstep, mstep) = hcol(1:mstep)
      H(mstep + 1, mstep) = beta
      if (present(ifres) .and. present(geom) .and. present(outdir)) then
        if (ifres) then
          allocate(cvals(mstep), cvecs(mstep, mstep))
          call eig(H(1:mstep, 1:mstep), cvecs, cvals, mstep)
          call arnoldi_checkpoint(ctx, geom, Q, H, mstep, ksize, outdir, evop, sampling_period, nsteps, cvals, cvecs)
          deallocate(cvals, cvecs)
        endif
      endif
      ! timing statistics, :92-98 (the reference prints hours and minutes; a step of this build takes a fraction of a second,
      ! so the wall seconds of the step follow on the same line: what bench.py's Fortran-host leg parses)
      call system_clock(tick1)
      telapsed = real(tick1 - tick0, c_double) / real(tickrate, c_double) / 3600.0d0
      tmiss = telapsed * (ksize - mstep)
      write(*,"(' Time per iteration/remaining:',I3,'h ',I2,'min /',I3,'h ',I2,'min   step_wall_s=',ES13.6)") &
        int(telapsed), ceiling((telapsed - int(telapsed)) * 60.), int(tmiss), ceiling((tmiss - int(tmiss)) * 60.), telapsed * 3600.0d0
    enddo
  end subroutine

  ! ---- eig: dgeev('N','V') + complex pair assembly + sort by decreasing modulus   core/lapack_wrapper.f:129-251
  subroutine eig(A, vecs, vals, n)
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

  ! selector of dgees: |lambda| > 0.9                  core/lapack_wrapper.f:258-270
  function select_eigvals(wr, wi) result(sel) bind(c)
    real(c_double), intent(in) :: wr, wi
    integer(c_int) :: sel
    sel = merge(1_c_int, 0_c_int, sqrt(wr * wr + wi * wi) > 0.9d0)
  end function

  ! ---- schur: dgees('V','S',select)                  core/lapack_wrapper.f:7-59
  subroutine schur(A, vecs, vals, n)
    integer, intent(in) :: n
    real(c_double), intent(inout) :: A(n, n)
    real(c_double), intent(out) :: vecs(n, n)
    complex(c_double_complex), intent(out) :: vals(n)
    real(c_double) :: wr(n), wi(n), work(8 * n)
    integer(c_int) :: bwork(n), nn, sdim, lw, info
    nn = n; lw = 8 * n
    call dgees('V', 'S', c_funloc(select_eigvals), nn, A, nn, sdim, wr, wi, vecs, nn, work, lw, bwork, info)
    vals = cmplx(wr, wi, kind=c_double_complex)
  end subroutine

  ! ---- ordschur: dtrsen('N','V')                     core/lapack_wrapper.f:70-122
  subroutine ordschur(T, Q, selected, n)
    integer, intent(in) :: n
    real(c_double), intent(inout) :: T(n, n), Q(n, n)
    logical, intent(in) :: selected(n)
    real(c_double) :: wr(n), wi(n), s, sep, work(max(1, n * n))
    integer(c_int) :: sel(n), iwork(max(1, n * n)), nn, m, lw, liw, info
    sel = merge(1_c_int, 0_c_int, selected)
    nn = n; lw = max(1, n * n); liw = max(1, n * n)
    call dtrsen('N', 'V', sel, nn, T, nn, Q, nn, wr, wi, m, s, sep, work, lw, iwork, liw, info)
  end subroutine

  ! ---- select_eigenvalues                            core/eigensolvers.f:729-795
  ! everything outside the circle of radius 1 - delta, at least the nev + 4 largest, and the conjugate partner of the
  ! smallest selected one
  subroutine select_eigenvalues(selected, mstart, vals, delta, nev, n)
    integer, intent(in) :: n, nev
    logical, intent(out) :: selected(n)
    integer, intent(out) :: mstart
    complex(c_double_complex), intent(in) :: vals(n)
    real(c_double), intent(in) :: delta
    integer :: idx(n), i, j, t, lo
    real(c_double) :: key(n), tk
    do i = 1, n
      idx(i) = i; key(i) = abs(vals(i))
    enddo
    do i = 2, n                                        ! ascending stable insertion sort (quicksort2, core/utils.f:31-147)
      tk = key(i); t = idx(i); j = i - 1
      do while (j >= 1)
        if (key(j) <= tk) exit
        key(j + 1) = key(j); idx(j + 1) = idx(j); j = j - 1
      enddo
      key(j + 1) = tk; idx(j + 1) = t
    enddo
    selected = abs(vals) >= (1.0d0 - delta)
    lo = max(1, n - (nev + 4) + 1)
    do i = lo, n
      selected(idx(i)) = .true.
    enddo
    if (lo - 1 >= 1) then
      if (aimag(vals(idx(lo))) == -aimag(vals(idx(lo - 1)))) selected(idx(lo - 1)) = .true.
    endif
    mstart = count(selected)
  end subroutine

  ! ---- schur_condensation                            core/eigensolvers.f:395-499
  ! host part: Schur form, selection, re-ordering, truncation of H; returns Z (the basis rotation) and ms (vectors kept)
  subroutine schur_restart_dense(H, ksize, schur_del, schur_tgt, Z, ms)
    integer, intent(in) :: ksize, schur_tgt
    real(c_double), intent(inout) :: H(ksize + 1, ksize)
    real(c_double), intent(in) :: schur_del
    real(c_double), intent(out) :: Z(ksize, ksize)
    integer, intent(out) :: ms
    real(c_double) :: b(ksize), T(ksize, ksize), bz(ksize)
    complex(c_double_complex) :: vals(ksize)
    logical :: selected(ksize)
    b = 0.0d0; b(ksize) = H(ksize + 1, ksize)                       ! :431-432
    T = H(1:ksize, 1:ksize)
    call schur(T, Z, vals, ksize)                                   ! :441
    call select_eigenvalues(selected, ms, vals, schur_del, schur_tgt, ksize)   ! :444
    call ordschur(T, Z, selected, ksize)                            ! :448
    H = 0.0d0
    H(1:ms, 1:ms) = T(1:ms, 1:ms)                                   ! :451-452
    bz = matmul(b, Z)
    H(ms + 1, 1:ms) = bz(1:ms)                                      ! :478-479
  end subroutine

  subroutine schur_condensation(ctx, mstart, H, Q, ksize, schur_del, schur_tgt)
    type(c_ptr), intent(in) :: ctx
    integer, intent(in) :: ksize, schur_tgt
    integer, intent(out) :: mstart
    real(c_double), intent(inout) :: H(ksize + 1, ksize)
    type(c_ptr), intent(inout) :: Q(ksize + 1)
    real(c_double), intent(in) :: schur_del
    real(c_double) :: Z(ksize, ksize)
    integer :: ms
    call schur_restart_dense(H, ksize, schur_del, schur_tgt, Z, ms)
    call nsk_check(nsk_basis_gemm(ctx, Q, int(ksize, c_int), Z, int(ksize, c_int)), 'nsk_basis_gemm')     ! :455-474  Q(:,1:k) <- Q(:,1:k) Z
    call nsk_check(nsk_copy(ctx, Q(ms + 1), Q(ksize + 1)), 'nsk_copy')          ! :482-485
    mstart = ms + 1
  end subroutine

  ! ---- krylov_schur                                  core/eigensolvers.f:141-388
  ! Q(1) holds the seed on entry (un-normalised); on return vals / vecs / residual describe H(1:k,1:k)
  ! geom / outdir / ifres: checkpoint after every Arnoldi step; restart_from > 0: continue from the checkpoint of that step
  ! (uparam(2) of the reference, core/eigensolvers.f:284-325) instead of starting from Q(1)
  subroutine krylov_schur(ctx, Q, H, vals, vecs, residual, k_dim, mode, schur_tgt, eigen_tol, schur_del, schur_cnt, matvecs, &
                          geom, outdir, evop, sampling_period, nsteps, ifres, restart_from)
    type(c_ptr), intent(in) :: ctx
    integer, intent(in) :: k_dim, schur_tgt
    type(host_geom), intent(in), optional :: geom
    character(*), intent(in), optional :: outdir, evop
    real(c_double), intent(in), optional :: sampling_period
    integer, intent(in), optional :: nsteps, restart_from
    logical, intent(in), optional :: ifres
    logical :: okr
    type(c_ptr), intent(inout) :: Q(k_dim + 1)
    real(c_double), intent(inout) :: H(k_dim + 1, k_dim)
    complex(c_double_complex), intent(out) :: vals(k_dim), vecs(k_dim, k_dim)
    real(c_double), intent(out) :: residual(k_dim)
    integer(c_int), intent(in) :: mode
    real(c_double), intent(in) :: eigen_tol, schur_del
    integer, intent(out) :: schur_cnt, matvecs
    real(c_double) :: alpha
    integer :: mstart, cnt
    logical :: converged
    H = 0.0d0
    call nsk_check(nsk_norm(ctx, Q(1), alpha), 'nsk_norm')                       ! krylov_normalize, :271-278
    call nsk_check(nsk_scal(ctx, Q(1), 1.0d0 / alpha), 'nsk_scal')
    mstart = 1; schur_cnt = 0; matvecs = 0; converged = .false.
    if (present(restart_from) .and. present(geom) .and. present(outdir)) then
      if (restart_from > 0) then                                                 ! :284-325
        call load_checkpoint(ctx, geom, Q, H, restart_from, k_dim, outdir, okr)
        if (.not. okr) then
          write(*,*) 'krylov_schur: no usable checkpoint of step ', restart_from, ' in ', trim(outdir); stop 1
        endif
        mstart = min(restart_from, k_dim) + 1
        if (restart_from > k_dim) then       ! the reference's k_dim < mstart branch (core/eigensolvers.f:295-301): sub-sampled Hessenberg matrix
          write(*,'(a,i0,a,i0,a)') ' restarted from the checkpoint of Arnoldi step ', restart_from, ', sub-sampled to k_dim = ', k_dim, ' (leading block of H, first k_dim+1 vectors)'
        else
          write(*,'(a,i0)') ' restarted from the checkpoint of Arnoldi step ', restart_from
        endif
      endif
    endif
    do while (.not. converged)                                                   ! :335-373
      if (present(ifres) .and. present(geom) .and. present(outdir)) then
        call arnoldi_factorization(ctx, Q, H, mstart, k_dim, k_dim, mode, geom, outdir, evop, sampling_period, nsteps, ifres)
      else
        call arnoldi_factorization(ctx, Q, H, mstart, k_dim, k_dim, mode)        ! :337
      endif
      matvecs = matvecs + k_dim - mstart + 1
      call eig(H(1:k_dim, 1:k_dim), vecs, vals, k_dim)                           ! :346
      residual = abs(H(k_dim + 1, k_dim) * vecs(k_dim, :))                       ! :349
      cnt = count(residual < eigen_tol)
      if (schur_tgt > 0 .and. cnt < schur_tgt .and. schur_cnt >= 50) then         ! the reference has no such cap: say so, never report it as converged
        write(*,'(a,i0,a,i0,a)') ' krylov_schur: 50 restarts without ', schur_tgt, ' converged eigenvalues (', cnt, ' so far): NOT CONVERGED, stopping'
        converged = .false.
        exit
      endif
      if (schur_tgt <= 0 .or. cnt >= schur_tgt) then        ! :354-371
        converged = .true.
      else
        schur_cnt = schur_cnt + 1
        call schur_condensation(ctx, mstart, H, Q, k_dim, schur_del, schur_tgt)
        write(*,'(a,i0,a,i0,a,i0)') ' Schur restart ', schur_cnt, ': ', cnt, ' converged, restarting from mstart = ', mstart
      endif
    enddo
  end subroutine

  ! ---- outpost_ks                                    core/eigensolvers.f:508-721
  ! spectra tables in the reference's formats; converged eigenmodes Q y_i (at most maxmodes), normalised so that
  ! |Re|^2 + |Im|^2 = 1 in the bm1s norm (:619-627), written as raw fp64 [vx | vy | pr] records next to the tables
  subroutine outpost_ks(ctx, vals, vecs, Q, residual, k_dim, outdir, evop, sampling_period, eigen_tol, maxmodes, nvel, npres, geom, nsteps)
    type(c_ptr), intent(in) :: ctx
    type(host_geom), intent(in), optional :: geom          ! present: the modes also as Nek field files <op>Re<session>0.f0000i / <op>Im... (:625-642)
    integer, intent(in), optional :: nsteps
    character(len=256) :: fn
    integer, intent(in) :: k_dim, maxmodes
    complex(c_double_complex), intent(in) :: vals(k_dim), vecs(k_dim, k_dim)
    type(c_ptr), intent(in) :: Q(k_dim + 1)
    real(c_double), intent(in) :: residual(k_dim), sampling_period, eigen_tol
    character(*), intent(in) :: outdir, evop
    integer(c_long_long), intent(in) :: nvel, npres
    type(c_ptr) :: w(2)
    real(c_double) :: yre(k_dim), yim(k_dim), ar, ai, beta
    real(c_double), allocatable :: vx(:), vy(:), pr(:)
    complex(c_double_complex) :: lam
    integer :: u1, u2, u3, um, i, outposted
    character(len=16) :: tag
    call nsk_check(nsk_vec_alloc(ctx, 2_c_int, w), 'nsk_vec_alloc')
    allocate(vx(nvel), vy(nvel), pr(npres))
    open(newunit=u1, file=trim(outdir)//'/Spectre_H'//trim(evop)//'.dat', status='replace')
    open(newunit=u2, file=trim(outdir)//'/Spectre_NS'//trim(evop)//'.dat', status='replace')
    open(newunit=u3, file=trim(outdir)//'/Spectre_NS'//trim(evop)//'_conv.dat', status='replace')
    outposted = 0
    do i = 1, k_dim
      lam = log(vals(i)) / sampling_period                                       ! log_transform, :908-915
      write(u1, '(3E15.7)') real(vals(i)), aimag(vals(i)), residual(i)           ! :590
      write(u2, '(3E15.7)') real(lam), aimag(lam), residual(i)
      if (residual(i) < eigen_tol .and. outposted < maxmodes) then               ! :599
        outposted = outposted + 1
        write(u3, '(2E15.7)') real(lam), aimag(lam)
        yre = real(vecs(:, i)); yim = aimag(vecs(:, i))
        call nsk_check(nsk_basis_gemv(ctx, Q, int(k_dim, c_int), yre, yim, w(1), w(2)), 'nsk_basis_gemv')    ! :607-615
        call nsk_check(nsk_dot(ctx, w(1), w(1), ar), 'nsk_dot')
        call nsk_check(nsk_dot(ctx, w(2), w(2), ai), 'nsk_dot')
        beta = 1.0d0 / sqrt(ar + ai)                                             ! :619-622
        call nsk_check(nsk_scal(ctx, w(1), beta), 'nsk_scal')
        call nsk_check(nsk_scal(ctx, w(2), beta), 'nsk_scal')
        write(tag, '(i5.5)') outposted
        call nsk_check(nsk_vec_download(ctx, w(1), vx, vy, pr), 'nsk_vec_download')
        open(newunit=um, file=trim(outdir)//'/'//trim(evop)//'Re'//trim(tag)//'.bin', access='stream', form='unformatted', status='replace')
        write(um) vx, vy, pr; close(um)
        call nsk_check(nsk_vec_download(ctx, w(2), vx, vy, pr), 'nsk_vec_download')
        open(newunit=um, file=trim(outdir)//'/'//trim(evop)//'Im'//trim(tag)//'.bin', access='stream', form='unformatted', status='replace')
        write(um) vx, vy, pr; close(um)
        if (present(geom)) then                                                  ! outpost(vx, vy, vz, pr, t, nRe / nIm), :625-642
          write(fn, '(a,a,a,a,a,a,i5.5)') trim(outdir), '/', trim(evop), 'Re', trim(geom%session), '0.f', outposted
          call state_write(ctx, geom, w(1), trim(fn), dble(outposted), nsteps + 1)
          write(fn, '(a,a,a,a,a,a,i5.5)') trim(outdir), '/', trim(evop), 'Im', trim(geom%session), '0.f', outposted
          call state_write(ctx, geom, w(2), trim(fn), dble(outposted), nsteps + 1)
        endif
      endif
    enddo
    close(u1); close(u2); close(u3)
    call nsk_check(nsk_vec_free(ctx, 2_c_int, w), 'nsk_vec_free')
  end subroutine

end module krylov_host
