"""SURVEY.md section 5 (race detection / sanitizers): the CPU builds of the tree kernels + host engine (wavefront
emulator) and of the oracle compile with -fsanitize=address,undefined and run a selection of their own tests clean.
tools/run_sanitized.sh without arguments runs the whole emulator + oracle suites that way."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ with libasan / libubsan")
def test_emulator_and_oracle_run_clean_under_asan_ubsan():
    # a selection sized for the CPU suite (about a minute); tools/run_sanitized.sh alone runs all 35 tests (15 min)
    sel = ["tests/test_engine_emul.py::test_static", "tests/test_engine_emul.py::test_batch_encode",
           "tests/test_engine_emul.py::test_search_random_vs_oracle", "tests/test_oracle_golden.py::test_survey_kats"]
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD",)}
    p = subprocess.run(["bash", os.path.join(REPO, "tools", "run_sanitized.sh")] + sel, cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert " passed" in p.stdout
