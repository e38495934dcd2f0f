"""The driver's bench command, exactly: `python3 bench.py --gpus 1 --steps 20 --warmup 5` in a fresh child process.
BENCH_r04 was lost (rc 1, no JSON) because an optional diagnostic raised before the record was printed and no test ran this
command; this one does, and checks what the judge reads from the record."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, timeout=900, env=None):
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                         env=dict(os.environ, **(env or {})))
    wall = time.perf_counter() - t0
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    return out, wall, lines


def test_driver_command_prints_one_complete_record():
    out, wall, lines = _run(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1, out.stdout[-2000:]                      # ONE JSON line
    r = json.loads(lines[0])
    assert r["metric"].startswith("Arnoldi matvecs/sec") and r["unit"] == "matvecs/s" and r["n_gpus"] == 1
    assert r["steps"] == 20 and r["warmup"] == 5 and r["higher_is_better"] is True and r["dtype"] == "f64" and r["vs_baseline"] is None
    assert "BASELINE configs[1]" in r["config"]["workload"] and "E=1996" in r["config"]["workload"] and "lx1=8" in r["config"]["workload"] and "nsteps=183" in r["config"]["workload"]
    assert r["value"] > 5.0 and abs(r["value"] - 1e3 / r["ms_per_step"]) < 1e-6 * r["value"]
    assert r["ms_per_step"] * r["steps"] * 1e-3 < wall              # the timed region fits inside the process's wall time
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and 0.05 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["kernel"].startswith("k_helm<8>") and rf["avg_launch_us"] > 1.0 and "traffic" in rf
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "matvecs/s" and cb["value"] > 0.0 and cb["cores"] >= 1 and "sample" in cb
    assert cb["iterations"]["pres_iters_per_step"] < cb["without_projection_space"]["iterations"]["pres_iters_per_step"]      # the port runs the projection space too
    assert r["wall_time_kdim_s"] > 0 and r["leading_ritz"]["k"] == 128 and abs(r["leading_ritz"]["re"] - 0.7386874) < 2e-6 and abs(r["leading_ritz"]["im"] - 0.6972307) < 2e-6
    assert r["map_retries"] == 0 and r["capped_solves"] == 0
    fh = r.get("fortran_host")
    if os.path.exists(os.path.join(ROOT, "host", "arnoldi_host")):
        assert fh and "error" not in fh and 0.8 < fh["vs_python_host"] < 1.25, fh       # the outer loop in Fortran: the same rate
    assert "lanes" not in r and "same_build_other_settings" not in r and "step_time_budget" not in r        # not in the driver's record


def test_record_survives_failing_optional_sections(tmp_path):
    """Every section after the timed steps is optional: with the CPU port and the Fortran host made to fail (test hooks) the
    record still has the headline and the roofline, the failed fields say why, and the exit code is 0; --extras writes the
    diagnostics to their own file AFTER the record."""
    ex = os.path.join(str(tmp_path), "extras.json")
    out, wall, lines = _run(["--gpus", "1", "--steps", "3", "--warmup", "1", "--no-kdim", "--extras", "--extras-out", ex],
                            env={"NSK_BENCH_TEST_FAIL": "cpu_baseline,fortran_host"})
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["value"] > 1.0 and r["roofline"]["frac"] > 0.05
    assert "error" in r["cpu_baseline"] and "error" in r["fortran_host"]
    assert r["wall_time_kdim_s"] is None and "wall_time_kdim_note" in r
    e = json.load(open(ex))
    assert e["step_time_budget"]["busy_fraction"] > 0.2 and "production_without_projection_space" in e["same_build_other_settings"]
