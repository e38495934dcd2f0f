#!/usr/bin/env python3
"""Install an already compiled library (same flags as __graft_entry__.build()) and stamp it with the source hash, so that
build() does not compile the same sources again:  python3 scripts/install_built.py /tmp/_t.so"""
import hashlib, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
deps = [g.SRC] + [os.path.join(ROOT, "nekstab_amd", "csrc", f) for f in ("nsk_kernels.hpp", "nsk_persist.hpp", "nsk3_kernels.hpp", "nsk3_mfma.hpp", "nsk3_mfma_ops.hpp", "nsk3_setup.inc", "nsk_dev.hpp", "nsk_basis.hpp", "nsk_shard.inc", "nsk_crtrig.hpp")]
deps.append(os.path.join(ROOT, "include", "nekstab_hip.h"))
h = hashlib.sha256()
for p in deps:
    h.update(open(p, "rb").read())
shutil.copy(sys.argv[1], g.LIB)
open(g.LIB + ".srchash", "w").write(h.hexdigest())
print("installed", g.LIB, h.hexdigest()[:12])
