! Boundary test: the REFERENCE'S OWN arnoldi_factorization + update_hessenberg_matrix (core/krylov_decomposition.f, compiled
! unchanged from /root/reference by host/Makefile's `ref_seam` target) drive libnekstab_hip.so through host/ref_seam/
! krylov_subspace_hip.f90.  Usage:  ref_seam_driver <case.bin> <k> <outdir> [mode: d | a]
! Writes <outdir>/HES_seam.txt = H(1:k+1, 1:k) row-major, and the number of live vectors after the run.
program ref_seam_driver
  use iso_c_binding
  use nekstab_hip
  use krylov_subspace
  implicit none
  integer :: nid, mstep
  logical :: ifres
  common /seam_size/ nid, mstep, ifres
  character(len=512) :: casefile, outdir, arg
  character(len=1) :: evop
  integer :: k, u, i, j, ios, nloc, np2
  integer(c_int) :: hdr(8), nsteps
  integer(c_long_long) :: nglob, nstate, nvel, npres
  real(c_double) :: rpar(3), sett(8), dt, beta, yv(3)
  real(c_double), allocatable, target :: x(:), y(:), mask(:), ub(:), vb(:), spng(:), sx(:), sy(:), sp(:)
  integer(c_long_long), allocatable, target :: gid(:), vert(:)
  type(nsk_case) :: cs
  type(c_ptr) :: ctx
  type(krylov_vector), allocatable :: Q(:)
  type(krylov_vector) :: comb
  real(c_double), allocatable :: H(:,:)

  call get_command_argument(1, casefile)
  call get_command_argument(2, arg); read(arg, *) k
  call get_command_argument(3, outdir)
  evop = 'd'
  if (command_argument_count() >= 4) call get_command_argument(4, evop)
  open(newunit=u, file=trim(casefile), access='stream', form='unformatted', status='old')
  read(u) hdr; read(u) nglob; read(u) rpar
  nloc = hdr(2) * hdr(3) * hdr(3); np2 = hdr(2) * (hdr(3) - 2) * (hdr(3) - 2)
  allocate(x(nloc), y(nloc), gid(nloc), mask(nloc), ub(nloc), vb(nloc), spng(nloc), vert(4 * hdr(2)), sx(nloc), sy(nloc), sp(np2))
  read(u) x; read(u) y; read(u) gid; read(u) mask; read(u) ub; read(u) vb; read(u) spng; read(u) vert
  read(u) sx; read(u) sy; read(u) sp
  read(u, iostat=ios) sett
  if (ios /= 0) stop 'case file without a settings record'
  close(u)
  cs%ndim = hdr(1); cs%nel = hdr(2); cs%lx1 = hdr(3); cs%lxd = hdr(4); cs%nglob = nglob
  cs%x = c_loc(x); cs%y = c_loc(y); cs%gid = c_loc(gid); cs%mask = c_loc(mask)
  cs%ub = c_loc(ub); cs%vb = c_loc(vb); cs%spng = c_loc(spng); cs%vert = c_loc(vert); cs%nvert = hdr(5)
  cs%re = rpar(1); cs%endtime = rpar(2); cs%cfl = rpar(3); cs%has_outflow = hdr(6)
  cs%tol_helm = sett(1); cs%tol_pres = sett(2); cs%tol_relative = 1
  cs%schwarz_layers = 2; cs%max_helm_iter = int(sett(8)); cs%max_pres_iter = 192; cs%nproj = int(sett(4))      ! (192: restarted GMRES cycles, the test runs tight tolerances)
  call nsk_check(nsk_init(cs, ctx), 'nsk_init')
  call nsk_option(ctx, 'min_pres_iter', sett(3))
  call nsk_check(nsk_get_info(ctx, dt, nsteps, nstate, nvel, npres), 'nsk_get_info')
  write(*,'(a,i0,a,i0)') ' nsteps = ', nsteps, '  state = ', nstate

  call krylov_subspace_attach(ctx, merge(NSK_ADJOINT, NSK_DIRECT, evop == 'a'))
  nid = 0; mstep = 0; ifres = .false.
  allocate(Q(k + 1), H(k + 1, k))
  H = 0.0d0
  call krylov_upload(Q(1), sx, sy, sp)                      ! the seed, normalised as core/eigensolvers.f:263-282 does
  call krylov_normalize(Q(1), beta)
  ! ---- the reference's routine, unchanged text
  call arnoldi_factorization(Q, H, 1, k, k)
  open(newunit=u, file=trim(outdir)//'/HES_seam.txt', status='replace')
  write(u, *) ((H(i, j), j = 1, k), i = 1, k + 1)
  close(u)
  ! krylov_matmul (core/krylov_subspace.f:216): a combination of the first three basis vectors, checked through its norm
  yv = (/ 0.6d0, -0.3d0, 0.2d0 /)
  call krylov_matmul(comb, Q(1:3), yv, 3)
  call krylov_norm(beta, comb)
  write(*,'(a,es24.16)') ' |Q(1:3) y| = ', beta
  write(*,'(a,i0)') ' live vectors before release: ', ks_live
  deallocate(Q)                                             ! finalizer: every handle goes back to the library
  call krylov_vector_release(comb)
  write(*,'(a,i0)') ' live vectors after release: ', ks_live
  call nsk_check(nsk_finalize(ctx), 'nsk_finalize')
end program ref_seam_driver
