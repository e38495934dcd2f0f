! CPU-only check of the host's dense linear algebra (tests/test_host_cpu.py): reads H(k+1,k) from <in>, applies eig and the
! host part of schur_condensation, writes the sorted eigen-decomposition, ms, the truncated H and the rotation Z to <out>.
program dense_check
  use iso_c_binding
  use krylov_host
  implicit none
  character(len=512) :: fin, fout, arg
  integer :: k, u, i, j, ms, schur_tgt
  real(c_double) :: schur_del
  real(c_double), allocatable :: H(:,:), Z(:,:)
  complex(c_double_complex), allocatable :: vals(:), vecs(:,:)
  call get_command_argument(1, fin); call get_command_argument(2, fout)
  call get_command_argument(3, arg); read(arg, *) k
  call get_command_argument(4, arg); read(arg, *) schur_tgt
  call get_command_argument(5, arg); read(arg, *) schur_del
  allocate(H(k + 1, k), Z(k, k), vals(k), vecs(k, k))
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) H; close(u)                       ! column-major
  call eig(H(1:k, 1:k), vecs, vals, k)
  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) (real(vals(i)), aimag(vals(i)), i = 1, k)
  write(u) ((real(vecs(i, j)), aimag(vecs(i, j)), i = 1, k), j = 1, k)
  call schur_restart_dense(H, k, schur_del, schur_tgt, Z, ms)
  write(u) real(ms, c_double)
  write(u) H
  write(u) Z
  close(u)
end program dense_check
