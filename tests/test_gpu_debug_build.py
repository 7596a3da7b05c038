"""The attention parity cases once more on the -DBEVR_DEBUG build of the library (make DEBUG=1): in-kernel traps on
(a) a region fill or flush that would leave the padded table and (b) a conditional barrier whose predicate is not
workgroup-uniform (csrc/attn_tile.h).  A trap aborts the child process, so the cases run in a subprocess.
The small goldens have tables NARROWER than an LDS region (S = 8: 38 padded columns against 88 region columns), the
shape behind round 1's GPU abort; test_mid_step_region_move_and_all_padding_half drives the mid-step barrier."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEBUG_LIB = os.path.join(ROOT, "bevrender_amd", "lib_debug", "libbevrender_hip.so")


def test_attention_cases_on_the_trap_instrumented_build():
    if not os.path.exists(DEBUG_LIB):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "bevrender_amd", "csrc"), "-j", "4", "DEBUG=1",
                            "OUTDIR=../lib_debug"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, BEVRENDER_LIB=DEBUG_LIB)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x",
                        os.path.join(ROOT, "tests", "test_gpu_modules.py"),
                        os.path.join(ROOT, "tests", "test_gpu_ops.py"),
                        "-k", "tsa_module or sca_module or attention_core or mid_step"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert " passed" in r.stdout
