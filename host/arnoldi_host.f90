! Fortran host of the eigensolver on top of libnekstab_hip.so: the driver the reference's `krylov_schur` call site
! (core/usr_extra.f:193-213) becomes.  The algorithms live in host/krylov_host.f90, the bindings in host/nekstab_hip_mod.f90.
! Usage:  arnoldi_host <case.bin> <k_dim> <outdir> [schur_tgt] [mode: d | a] [ifres: 0 | 1] [restart_from]
!   ifres = 1: the reference's checkpoint after every Arnoldi step (KRY / HES / Spectre files, core/eigensolvers.f:802-905);
!   restart_from = m > 0: continue from the checkpoint of step m in <outdir> (uparam(2), core/eigensolvers.f:284-325)
! case.bin is written by nekstab_amd/casefile.py: mesh arrays, base flow, seed vector and the inner-solver settings
! (nekstab_amd/settings.py: the production settings, not a copy of them in this file).
program arnoldi_host
  use iso_c_binding
  use nekstab_hip
  use krylov_host
  use nek_fld
  implicit none
  type(host_geom) :: geom
  integer :: ifres_i, restart_from
  character(len=512) :: casefile, outdir, arg
  character(len=1) :: evop
  integer :: k_dim, u, i, j, schur_tgt, schur_cnt, matvecs, ios
  integer(c_int) :: hdr(8), nsteps, mode
  integer(c_long_long) :: nglob, nstate, nvel, npres
  real(c_double) :: rpar(3), sett(8), dt, t0, t1
  real(c_double), allocatable, target :: x(:), y(:), mask(:), ub(:), vb(:), spng(:), sx(:), sy(:), sp(:)
  integer(c_long_long), allocatable, target :: gid(:), vert(:)
  type(nsk_case) :: cs
  type(c_ptr) :: ctx
  type(c_ptr), allocatable :: Q(:)
  real(c_double), allocatable :: H(:,:)
  complex(c_double_complex), allocatable :: vals(:), vecs(:,:)
  real(c_double), allocatable :: residual(:)
  integer :: nloc, np2

  call get_command_argument(1, casefile)
  call get_command_argument(2, arg); read(arg, *) k_dim
  call get_command_argument(3, outdir)
  schur_tgt = 0; evop = 'd'
  if (command_argument_count() >= 4) then
    call get_command_argument(4, arg); read(arg, *) schur_tgt
  endif
  if (command_argument_count() >= 5) call get_command_argument(5, evop)
  ifres_i = 0; restart_from = 0
  if (command_argument_count() >= 6) then
    call get_command_argument(6, arg); read(arg, *) ifres_i
  endif
  if (command_argument_count() >= 7) then
    call get_command_argument(7, arg); read(arg, *) restart_from
  endif
  mode = merge(NSK_ADJOINT, NSK_DIRECT, evop == 'a')

  open(newunit=u, file=trim(casefile), access='stream', form='unformatted', status='old')
  read(u) hdr; read(u) nglob; read(u) rpar
  nloc = hdr(2) * hdr(3) * hdr(3); np2 = hdr(2) * (hdr(3) - 2) * (hdr(3) - 2)
  allocate(x(nloc), y(nloc), gid(nloc), mask(nloc), ub(nloc), vb(nloc), spng(nloc), vert(4 * hdr(2)))
  allocate(sx(nloc), sy(nloc), sp(np2))
  read(u) x; read(u) y; read(u) gid; read(u) mask; read(u) ub; read(u) vb; read(u) spng; read(u) vert
  read(u) sx; read(u) sy; read(u) sp
  ! inner-solver and eigensolver settings: tol_helm, tol_pres, min_pres_iter, nproj, eigen_tol, schur_del, maxmodes, max_helm_iter
  read(u, iostat=ios) sett
  if (ios /= 0) then
    write(*,*) 'case file without a settings record: rewrite it with nekstab_amd/casefile.py'; stop 1
  endif
  close(u)

  cs%ndim = hdr(1); cs%nel = hdr(2); cs%lx1 = hdr(3); cs%lxd = hdr(4); cs%nglob = nglob
  cs%x = c_loc(x); cs%y = c_loc(y); cs%gid = c_loc(gid); cs%mask = c_loc(mask)
  cs%ub = c_loc(ub); cs%vb = c_loc(vb); cs%spng = c_loc(spng); cs%vert = c_loc(vert); cs%nvert = hdr(5)
  cs%re = rpar(1); cs%endtime = rpar(2); cs%cfl = rpar(3); cs%has_outflow = hdr(6)
  cs%tol_helm = sett(1); cs%tol_pres = sett(2); cs%tol_relative = 1
  cs%schwarz_layers = 2; cs%max_helm_iter = int(sett(8)); cs%max_pres_iter = 144; cs%nproj = int(sett(4))      ! (restarted 48-vector GMRES cycles)
  call nsk_check(nsk_init(cs, ctx), 'nsk_init')
  call nsk_option(ctx, 'min_pres_iter', sett(3))
  call nsk_check(nsk_get_info(ctx, dt, nsteps, nstate, nvel, npres), 'nsk_get_info')
  write(*,'(a,es14.6,a,i0,a,i0)') ' dt = ', dt, '  nsteps = ', nsteps, '  state = ', nstate
  write(*,'(a,es9.2,a,es9.2,a,i0,a,i0)') ' settings: tol_helm = ', sett(1), '  tol_pres = ', sett(2), '  min_pres_iter = ', int(sett(3)), '  nproj = ', int(sett(4))

  ! ----- allocate(Q(k_dim+1)), H(k_dim+1,k_dim)          core/eigensolvers.f:170-171
  allocate(Q(k_dim + 1), H(k_dim + 1, k_dim))
  allocate(vals(k_dim), vecs(k_dim, k_dim), residual(k_dim))
  call nsk_check(nsk_vec_alloc(ctx, int(k_dim + 1, c_int), Q), 'nsk_vec_alloc')
  call nsk_check(nsk_vec_upload(ctx, Q(1), sx, sy, sp), 'nsk_vec_upload')       ! seed, :263-282

  call cpu_time(t0)
  if (hdr(1) == 2) call geom_init(geom, int(hdr(2)), int(hdr(3)), x, y, '1cyl')
  if (hdr(1) == 2 .and. (ifres_i /= 0 .or. restart_from > 0)) then
    call krylov_schur(ctx, Q, H, vals, vecs, residual, k_dim, mode, schur_tgt, sett(5), sett(6), schur_cnt, matvecs, &
                      geom, trim(outdir), evop, rpar(2), int(nsteps), ifres_i /= 0, restart_from)
  else
    call krylov_schur(ctx, Q, H, vals, vecs, residual, k_dim, mode, schur_tgt, sett(5), sett(6), schur_cnt, matvecs)
  endif
  call cpu_time(t1)
  write(*,'(a,i0,a,i0,a,f8.3,a)') ' Krylov-Schur: ', matvecs, ' matvecs, ', schur_cnt, ' restarts in ', t1 - t0, ' s (cpu time of the host thread)'

  if (hdr(1) == 2) then
    call outpost_ks(ctx, vals, vecs, Q, residual, k_dim, trim(outdir), evop, rpar(2), sett(5), int(sett(7)), nvel, npres, geom, int(nsteps))
  else
    call outpost_ks(ctx, vals, vecs, Q, residual, k_dim, trim(outdir), evop, rpar(2), sett(5), int(sett(7)), nvel, npres)
  endif
  open(newunit=u, file=trim(outdir)//'/ritz_full.txt', status='replace')
  do i = 1, k_dim
    write(u, '(3ES26.17)') real(vals(i)), aimag(vals(i)), residual(i)
  enddo
  close(u)
  open(newunit=u, file=trim(outdir)//'/HES.txt', status='replace')      ! arnoldi_checkpoint :889
  write(u, *) ((H(i, j), j = 1, k_dim), i = 1, k_dim + 1)
  close(u)
  call nsk_check(nsk_finalize(ctx), 'nsk_finalize')
end program arnoldi_host
