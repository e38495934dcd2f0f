#!/usr/bin/env python3
"""Thread scaling of the CPU port (oracle/cpu_step.c) on the host cores of the box it runs on -- VERDICT r5 item 6.

    python3 scripts/cpu_scaling.py [--lx1 8] [--steps 6] [--threads 8,16,32,64,128] > table

One child process per OpenMP environment (libgomp reads OMP_PROC_BIND / OMP_PLACES / OMP_WAIT_POLICY when it is loaded);
inside a child the thread count is changed with omp_set_num_threads.  Prints ms per time step (after one untimed step,
projection space reset before every sample) and one JSON line at the end.  Test / baseline infrastructure (imports oracle/)."""
import argparse, json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ENVS = {
    "passive (what rounds 2-5 timed)": {"OMP_WAIT_POLICY": "passive"},
    "active": {"OMP_WAIT_POLICY": "active"},
    "active, bind close, places cores": {"OMP_WAIT_POLICY": "active", "OMP_PROC_BIND": "close", "OMP_PLACES": "cores"},
    "active, bind spread, places cores": {"OMP_WAIT_POLICY": "active", "OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"},
    "passive, bind close, places cores": {"OMP_WAIT_POLICY": "passive", "OMP_PROC_BIND": "close", "OMP_PLACES": "cores"},
}


def child(a):
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from oracle.cpu_port import CpuPort
    from oracle.linns import LinNS2D
    from nekstab_amd import seed
    case = bench.build_case("cfg2", a.lx1 if a.lx1 != 8 else None)
    o = LinNS2D(x=case.x, y=case.y, gid=case.gid, nglob=case.nglob, mask=case.mask, ub=case.ub, spng=case.spng, re=case.re,
                endtime=case.endtime, lxd=case.lxd, has_outflow=case.has_outflow, factorize_pressure=False)
    cp = CpuPort(o, case.meta["vert"], case.meta["nvert"], nproj=32, tol_helm=3e-12, tol_pres=3e-2, tol_relative=1, min_pres=2)
    qx, qy = seed.add_noise(case)
    q0 = (qx, qy, np.zeros((case.nel, case.lx1 - 2, case.lx1 - 2)))
    res = {}
    for nt in [int(t) for t in a.threads.split(",")]:
        cp.set_threads(nt)
        cp.proj_reset()
        t0 = time.perf_counter(); cp.matvec(q0, nsteps=1); t1 = time.perf_counter() - t0
        if t1 > 2.0:                       # hopeless (oversubscribed spinning): do not spend minutes on it
            res[nt] = 1e3 * t1
            continue
        cp.proj_reset()
        t0 = time.perf_counter(); cp.matvec(q0, nsteps=a.steps); res[nt] = 1e3 * (time.perf_counter() - t0) / a.steps
    print("CHILD " + json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lx1", type=int, default=8)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--threads", default="4,8,16,32,64,128")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--envs", default="")
    a = ap.parse_args()
    if a.child:
        return child(a)
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    print("cores visible: %d" % visible)
    try:
        print(subprocess.run("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA node\\(s\\)|^CPU\\(s\\)'", shell=True, capture_output=True, text=True).stdout)
    except Exception:                      # noqa: BLE001
        pass
    out = {"cores_visible": visible, "ms_per_time_step": {}}
    for name, env in ENVS.items():
        if a.envs and not any(k in name for k in a.envs.split(";")):
            continue
        e = dict(os.environ)
        for k in ("OMP_WAIT_POLICY", "OMP_PROC_BIND", "OMP_PLACES", "OMP_NUM_THREADS"):
            e.pop(k, None)
        e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--lx1", str(a.lx1), "--steps", str(a.steps), "--threads", a.threads],
                           env=e, capture_output=True, text=True, timeout=1500)
        line = [l for l in r.stdout.splitlines() if l.startswith("CHILD ")]
        if not line:
            print("%-40s failed: %s" % (name, r.stderr[-300:]))
            continue
        res = json.loads(line[0][6:])
        out["ms_per_time_step"][name] = res
        print("%-40s " % name + "  ".join("%s thr: %7.2f ms" % (k, v) for k, v in res.items()), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
