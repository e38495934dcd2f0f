! Nek5000 field files (.f0000N) and the Arnoldi checkpoint of the reference, from the Fortran host (quadrilateral cases):
!   fld_write / fld_read      the single-file `outpost` format (SURVEY App. B: 132-byte header, endian tag, element map, X / U / P
!                             blocks per element; pressure on mesh 1)                               [core/IO.f:15-60 reads it back]
!   arnoldi_checkpoint        KRY<session>0.f<k+1>, Spectre_H<op><k>.dat, Spectre_NS<op><k>.dat, HES<session><k>
!                                                                                                  core/eigensolvers.f:802-905
!   load_checkpoint           H from HES<session><mstart>, Q(1 .. mstart+1) from the KRY files      core/eigensolvers.f:284-325
! The same formats as nekstab_amd/nekio.py + checkpoint.py write: a run checkpointed by one host restarts from the other
! (tests/test_fortran_host_gpu.py).  KRY files are written in double precision so that a restarted factorisation continues the
! same Krylov sequence to round-off (the reference writes them in single precision).
module nek_fld
  use iso_c_binding
  use nekstab_hip
  implicit none
  private
  public :: host_geom, geom_init, fld_write, fld_read, state_write, state_read, arnoldi_checkpoint, load_checkpoint

  type host_geom                       ! what the host needs to turn a device state into a field file
    integer :: nel = 0, lx1 = 0, lx2 = 0
    real(c_double), allocatable :: x(:), y(:)               ! GLL coordinates, element-major
    real(c_double), allocatable :: J21(:, :), J12(:, :)     ! Gauss -> GLL (output, map21) and GLL -> Gauss (input, map12)
    character(len=64) :: session = '1cyl'
  end type

contains

  ! Legendre polynomial and its derivative at x
  subroutine legendre(n, x, p, dp)
    integer, intent(in) :: n
    real(c_double), intent(in) :: x
    real(c_double), intent(out) :: p, dp
    real(c_double) :: p0, p1, p2
    integer :: k
    p0 = 1.0d0; p1 = x
    if (n == 0) then
      p = 1.0d0; dp = 0.0d0; return
    endif
    do k = 2, n
      p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
      p0 = p1; p1 = p2
    enddo
    p = p1
    dp = n * (x * p1 - p0) / (x * x - 1.0d0)
  end subroutine

  subroutine gl_nodes(n, z)            ! zeros of P_n
    integer, intent(in) :: n
    real(c_double), intent(out) :: z(n)
    real(c_double) :: x, p, dp, pi
    integer :: i, it
    pi = 4.0d0 * atan(1.0d0)
    do i = 1, n
      x = -cos(pi * (i - 0.25d0) / (n + 0.5d0))
      do it = 1, 100
        call legendre(n, x, p, dp)
        x = x - p / dp
        if (abs(p / dp) < 1d-15) exit
      enddo
      z(i) = x
    enddo
  end subroutine

  subroutine gll_nodes(n, z)           ! -1, zeros of P'_{n-1}, 1
    integer, intent(in) :: n
    real(c_double), intent(out) :: z(n)
    real(c_double) :: x, p, dp, d2, pi
    integer :: i, it, m
    m = n - 1
    pi = 4.0d0 * atan(1.0d0)
    z(1) = -1.0d0; z(n) = 1.0d0
    do i = 2, n - 1
      x = -cos(pi * (i - 1) / m)
      do it = 1, 100
        call legendre(m, x, p, dp)
        d2 = (2.0d0 * x * dp - m * (m + 1) * p) / (1.0d0 - x * x)      ! P''_m from Legendre's equation
        x = x - dp / d2
        if (abs(dp / d2) < 1d-15) exit
      enddo
      z(i) = x
    enddo
  end subroutine

  subroutine interp_matrix(nf, zf, nt, zt, J)       ! J(t, f): Lagrange interpolation from nodes zf to points zt
    integer, intent(in) :: nf, nt
    real(c_double), intent(in) :: zf(nf), zt(nt)
    real(c_double), intent(out) :: J(nt, nf)
    integer :: a, i, k
    real(c_double) :: l
    do a = 1, nt
      do i = 1, nf
        l = 1.0d0
        do k = 1, nf
          if (k /= i) l = l * (zt(a) - zf(k)) / (zf(i) - zf(k))
        enddo
        J(a, i) = l
      enddo
    enddo
  end subroutine

  subroutine geom_init(g, nel, lx1, x, y, session)
    type(host_geom), intent(out) :: g
    integer, intent(in) :: nel, lx1
    real(c_double), intent(in) :: x(:), y(:)
    character(*), intent(in) :: session
    real(c_double) :: zl(lx1), zg(lx1 - 2)
    g%nel = nel; g%lx1 = lx1; g%lx2 = lx1 - 2; g%session = session
    allocate(g%x(size(x)), g%y(size(y)), g%J21(lx1, lx1 - 2), g%J12(lx1 - 2, lx1))
    g%x = x; g%y = y
    call gll_nodes(lx1, zl); call gl_nodes(lx1 - 2, zg)
    call interp_matrix(lx1 - 2, zg, lx1, zl, g%J21)
    call interp_matrix(lx1, zl, lx1 - 2, zg, g%J12)
  end subroutine

  ! ---- the field file: X, U, P blocks of a 2-D case, wdsize = 8
  subroutine fld_write(path, g, vx, vy, p1, time, istep)
    character(*), intent(in) :: path
    type(host_geom), intent(in) :: g
    real(c_double), intent(in) :: vx(:), vy(:), p1(:), time
    integer, intent(in) :: istep
    character(len=132) :: hdr
    integer :: u, e, nn
    integer(c_int), allocatable :: emap(:)
    nn = g%lx1 * g%lx1
    write(hdr, '(a4,1x,i1,1x,i2,1x,i2,1x,i2,1x,i10,1x,i10,1x,es20.13,1x,i9,1x,i6,1x,i6,1x,a10,1x,es14.7,1x,a1)') &
      '#std', 8, g%lx1, g%lx1, 1, g%nel, g%nel, time, istep, 0, 1, 'XUP       ', 1.0d0, 'F'
    allocate(emap(g%nel))
    do e = 1, g%nel
      emap(e) = e
    enddo
    open(newunit=u, file=trim(path), access='stream', form='unformatted', status='replace')
    write(u) hdr
    write(u) 6.54321_c_float
    write(u) emap
    do e = 1, g%nel                                          ! X: per element [x(n), y(n)]
      write(u) g%x((e - 1) * nn + 1:e * nn), g%y((e - 1) * nn + 1:e * nn)
    enddo
    do e = 1, g%nel
      write(u) vx((e - 1) * nn + 1:e * nn), vy((e - 1) * nn + 1:e * nn)
    enddo
    write(u) p1
    close(u)
  end subroutine

  subroutine fld_read(path, g, vx, vy, p1, ok)
    character(*), intent(in) :: path
    type(host_geom), intent(in) :: g
    real(c_double), intent(out) :: vx(:), vy(:), p1(:)
    logical, intent(out) :: ok
    character(len=132) :: hdr
    character(len=4) :: tag
    real(c_float) :: endian
    integer :: u, e, nn, wd, nx, ny, nz, nel, nelg, ios
    integer(c_int), allocatable :: emap(:)
    real(c_double), allocatable :: xy(:)
    real(c_float), allocatable :: xy4(:)
    ok = .false.
    nn = g%lx1 * g%lx1
    open(newunit=u, file=trim(path), access='stream', form='unformatted', status='old', iostat=ios)
    if (ios /= 0) return
    read(u) hdr
    read(hdr, *, iostat=ios) tag, wd, nx, ny, nz, nel, nelg
    ! wdsize 8: this host's own checkpoints; wdsize 4: what the reference's outpost writes (KRY files of a run checkpointed by
    ! nekStab itself, param(63) = 0): read as real(4) and promoted
    if (ios /= 0 .or. (wd /= 8 .and. wd /= 4) .or. nx /= g%lx1 .or. nel /= g%nel .or. index(hdr, 'XUP') == 0) then
      close(u); return
    endif
    read(u) endian
    allocate(emap(nel), xy(2 * nn), xy4(2 * nn))
    read(u) emap
    do e = 1, nel
      call rd(2 * nn)
    enddo
    do e = 1, nel
      call rd(2 * nn)
      vx((emap(e) - 1) * nn + 1:emap(e) * nn) = xy(1:nn); vy((emap(e) - 1) * nn + 1:emap(e) * nn) = xy(nn + 1:2 * nn)
    enddo
    do e = 1, nel
      call rd(nn)
      p1((emap(e) - 1) * nn + 1:emap(e) * nn) = xy(1:nn)
    enddo
    close(u)
    ok = .true.
  contains
    subroutine rd(n)                                   ! n values of the file's word size into xy(1:n)
      integer, intent(in) :: n
      if (wd == 8) then
        read(u) xy(1:n)
      else
        read(u) xy4(1:n)
        xy(1:n) = real(xy4(1:n), c_double)
      endif
    end subroutine
  end subroutine

  ! device state <-> field file (pressure through map21 / map12, as outpost / load_fld do)
  subroutine state_write(ctx, g, v, path, time, istep)
    type(c_ptr), intent(in) :: ctx, v
    type(host_geom), intent(in) :: g
    character(*), intent(in) :: path
    real(c_double), intent(in) :: time
    integer, intent(in) :: istep
    real(c_double), allocatable :: vx(:), vy(:), pr(:), p1(:)
    real(c_double) :: pe(g%lx2, g%lx2), t(g%lx1, g%lx2), o(g%lx1, g%lx1)
    integer :: e, n1, n2
    n1 = g%lx1 * g%lx1; n2 = g%lx2 * g%lx2
    allocate(vx(g%nel * n1), vy(g%nel * n1), pr(g%nel * n2), p1(g%nel * n1))
    call nsk_check(nsk_vec_download(ctx, v, vx, vy, pr), 'nsk_vec_download')
    do e = 1, g%nel
      pe = reshape(pr((e - 1) * n2 + 1:e * n2), (/ g%lx2, g%lx2 /))       ! pe(a, b): a fastest (r), b (s)
      t = matmul(g%J21, pe)                                               ! r
      o = matmul(t, transpose(g%J21))                                     ! s
      p1((e - 1) * n1 + 1:e * n1) = reshape(o, (/ n1 /))
    enddo
    call fld_write(path, g, vx, vy, p1, time, istep)
  end subroutine

  subroutine state_read(ctx, g, v, path, ok)
    type(c_ptr), intent(in) :: ctx, v
    type(host_geom), intent(in) :: g
    character(*), intent(in) :: path
    logical, intent(out) :: ok
    real(c_double), allocatable :: vx(:), vy(:), pr(:), p1(:)
    real(c_double) :: pe(g%lx1, g%lx1), t(g%lx2, g%lx1), o(g%lx2, g%lx2)
    integer :: e, n1, n2
    n1 = g%lx1 * g%lx1; n2 = g%lx2 * g%lx2
    allocate(vx(g%nel * n1), vy(g%nel * n1), pr(g%nel * n2), p1(g%nel * n1))
    call fld_read(path, g, vx, vy, p1, ok)
    if (.not. ok) return
    do e = 1, g%nel
      pe = reshape(p1((e - 1) * n1 + 1:e * n1), (/ g%lx1, g%lx1 /))
      t = matmul(g%J12, pe)
      o = matmul(t, transpose(g%J12))
      pr((e - 1) * n2 + 1:e * n2) = reshape(o, (/ n2 /))
    enddo
    call nsk_check(nsk_vec_upload(ctx, v, vx, vy, pr), 'nsk_vec_upload')
  end subroutine

  function kry_name(g, i) result(s)
    type(host_geom), intent(in) :: g
    integer, intent(in) :: i
    character(len=96) :: s
    write(s, '(a,a,a,i5.5)') 'KRY', trim(g%session), '0.f', i
  end function

  ! ---- core/eigensolvers.f:802-905: after Arnoldi step k: vector k+1, the spectra of H(1:k,1:k), the Hessenberg matrix
  subroutine arnoldi_checkpoint(ctx, g, Q, H, k, ksize, outdir, evop, sampling_period, nsteps, vals, vecs)
    type(c_ptr), intent(in) :: ctx
    type(host_geom), intent(in) :: g
    integer, intent(in) :: k, ksize, nsteps
    type(c_ptr), intent(in) :: Q(ksize + 1)
    real(c_double), intent(in) :: H(ksize + 1, ksize), sampling_period
    character(*), intent(in) :: outdir, evop
    complex(c_double_complex), intent(in) :: vals(k), vecs(k, k)         ! eig(H(1:k,1:k)) from the caller (krylov_host: eig)
    real(c_double) :: residual(k)
    complex(c_double_complex) :: lam
    character(len=256) :: fn
    integer :: u, i, j
    if (k == 1) call state_write(ctx, g, Q(1), trim(outdir)//'/'//trim(kry_name(g, 1)), 0.0d0, nsteps + 1)      ! the initial condition, :280-282
    call state_write(ctx, g, Q(k + 1), trim(outdir)//'/'//trim(kry_name(g, k + 1)), dble(k), nsteps + 1)       ! whereyouwant("KRY", k+1), :843-849
    residual = abs(H(k + 1, k) * vecs(k, :))                                                                  ! :855
    write(fn, '(a,a,a,a,i4.4,a)') trim(outdir), '/Spectre_H', trim(evop), '', k, '.dat'
    open(newunit=u, file=trim(fn), status='replace')
    write(u, '(3E15.7)') (real(vals(i)), aimag(vals(i)), residual(i), i = 1, k)                               ! :866
    close(u)
    write(fn, '(a,a,a,a,i4.4,a)') trim(outdir), '/Spectre_NS', trim(evop), '', k, '.dat'
    open(newunit=u, file=trim(fn), status='replace')
    do i = 1, k
      lam = log(vals(i)) / sampling_period
      write(u, '(3E15.7)') real(lam), aimag(lam), residual(i)                                                 ! :874-877
    enddo
    close(u)
    write(fn, '(a,a,a,i4.4)') trim(outdir), '/HES', trim(g%session), k                                        ! :881
    open(newunit=u, file=trim(fn), status='replace')
    write(u, *) ((H(i, j), j = 1, k), i = 1, k + 1)                                                           ! :885
    close(u)
  end subroutine

  ! ---- restart (uparam(2) = mstart, core/eigensolvers.f:284-325): H(1:mstart+1, 1:mstart) and Q(1 .. mstart+1)
  subroutine load_checkpoint(ctx, g, Q, H, mstart, ksize, indir, ok)
    type(c_ptr), intent(in) :: ctx
    type(host_geom), intent(in) :: g
    integer, intent(in) :: mstart, ksize
    type(c_ptr), intent(in) :: Q(ksize + 1)
    real(c_double), intent(inout) :: H(ksize + 1, ksize)
    character(*), intent(in) :: indir
    logical, intent(out) :: ok
    character(len=256) :: fn
    real(c_double) :: hm((mstart + 1) * mstart)
    integer :: u, i, j, ios
    ok = .false.
    write(fn, '(a,a,a,i4.4)') trim(indir), '/HES', trim(g%session), mstart
    open(newunit=u, file=trim(fn), status='old', iostat=ios)
    if (ios /= 0) return
    read(u, *, iostat=ios) hm
    close(u)
    if (ios /= 0) return
    ! mstart > ksize ("subsampling", core/eigensolvers.f:295-301): the leading (ksize+1) x ksize block of the checkpointed
    ! factorisation and its first ksize+1 vectors -- a leading part of an Arnoldi factorisation is an Arnoldi factorisation
    H = 0.0d0
    do i = 1, min(mstart, ksize) + 1
      do j = 1, min(mstart, ksize)
        H(i, j) = hm((i - 1) * mstart + j)
      enddo
    enddo
    do i = 1, min(mstart, ksize) + 1
      call state_read(ctx, g, Q(i), trim(indir)//'/'//trim(kry_name(g, i)), ok)
      if (.not. ok) return
    enddo
  end subroutine
end module nek_fld
