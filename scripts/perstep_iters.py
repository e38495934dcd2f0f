"""Per-time-step iteration counts of one map at the production settings (NSK_DEBUG=1, eager launches), for a late Krylov vector."""
import os, sys, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from nekstab_amd import krylov, mesh, seed
    from nekstab_amd.settings import production_context
    case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), 8)
    h = production_context(case)
    qx, qy = seed.add_noise(case)
    Q = h.alloc(12); H = np.zeros((12, 11))
    h.upload(Q[0], qx, qy, np.zeros(h.npres)); h.scal(Q[0], 1.0 / h.norm(Q[0]))
    krylov.arnoldi_factorization(h, Q, H, 1, 10, 0)
    print("MARK", flush=True); sys.stderr.write("MARK\n"); sys.stderr.flush()
    h.set_option("use_graph", 0)
    os.environ["NSK_DEBUG"] = "1"
    h2 = production_context(case)      # NSK_DEBUG is read at init: a second context, warmed by copying the vector
    v, f = h2.alloc(2)
    a = h.download(Q[10]); h2.upload(v, *a)
    h2.set_option("use_graph", 0)
    h2.matvec(f, v, 0); h2.matvec(f, v, 0)
    sys.exit(0)
out = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True)
lines = [l for l in out.stderr.splitlines() if l.startswith("step")]
lines = lines[len(lines) // 2:]                    # the second map (projection space warm)
hi = [int(re.search(r"helm_it=(\d+)", l).group(1)) for l in lines]
pi = [int(re.search(r"pres_it=(\d+)", l).group(1)) for l in lines]
print("helm per step:", hi)
print("pres per step:", pi)
import numpy as np
for name, v in (("helm", np.array(hi)), ("pres", np.array(pi))):
    print(name, "mean %.2f" % v.mean(), "steps 17+: mean %.2f max %d p90 %d | 17-40 max %d, 41-100 max %d, 101+ max %d" % (v[16:].mean(), v[16:].max(), np.percentile(v[16:], 90), v[16:40].max(), v[40:100].max(), v[100:].max()))
