"""The round-6 forms of the x6 GEMM with pre-split weights (csrc/igemm_planes.hip: activation fragments straight from global
memory in 128 x 64 / 128 x 128 / 256 x 64 tiles, and the resident-B streaming kernel for K = 64 / 128 over the big maps) are
not dispatched by the product library -- in the step they measured neutral to +0.2 ms -- but they stay correct: the
diagnostic build forces each variant onto every eligible launch of the parity cases of
tests/test_hip_ops.py::test_x6_conv_with_presplit_weights (fp64 ATen, 2e-5 of the output scale, forward with the full epilogue
and data gradient with the layer scale folded into the pack; K = 64 ... 2 048, split contractions, 32 768 ... 131 072 rows)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TUNING = os.path.join(ROOT, "asy-vrnet_amd", "csrc", "libvrnet_hip_tuning.so")


@pytest.mark.gpu
@pytest.mark.parametrize("env", [
    {"VRNET_PLANES_REG21": "413", "VRNET_PLANES_REG22": "422"},      # 128 x 64 (3 per CU) / 128 x 128 tiles, four waves
    {"VRNET_PLANES_REG21": "414", "VRNET_PLANES_REG22": "423"},
    {"VRNET_PLANES_REG21": "812", "VRNET_PLANES_REG22": "821"},      # 256-row tiles, eight waves on one B stage
    {"VRNET_PLANES_REG21": "-1"},                                    # the per-shape rule that was measured in the step
    {"VRNET_PLANES_STREAM_MIN_ROWS": "32768", "VRNET_PLANES_STREAM_MAX_COLS": "4096"},      # resident B, barrier-free A streams
], ids=["reg413_422", "reg414_423", "reg812_821", "reg_rule", "stream"])
def test_forced_variants_pass_the_presplit_parity_cases(env):
    if not os.path.exists(TUNING):
        pytest.skip("diagnostic build absent (make -C asy-vrnet_amd/csrc tuning)")
    e = dict(os.environ, VRNET_HIP_LIB=TUNING, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_ops.py"), "-x", "-q", "-k", "presplit",
                        "-p", "no:cacheprovider"], env=e, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0 and " passed" in r.stdout, tail
