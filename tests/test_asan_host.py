"""CPU sanitizer pass over the library's host side.

The host code of csrc/ (weight-norm folding, row / Winograd / MFMA-fragment packing, style tables, the workspace closed forms,
arena carving, the launch planning of every contraction) is compiled with -fsanitize=address,undefined against a stub of the
HIP runtime (tests/asan/hip_stub.cpp: device memory = host heap, launches validated and counted, nothing executed) and driven
over the shapes of BASELINE's configs (cfg2, cfg3's phoneme batch, cfg4's 256 mixed utterances, cfg5's 64 x 10 s per GPU,
token-count extremes), in fp32 and fp16 operand modes.  GPU AddressSanitizer is not available on this pool; the kernels
themselves are covered by the -m gpu parity tests."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_host_side_under_asan_ubsan(tmp_path):
    if not (os.path.exists("/opt/rocm/bin/hipcc") and os.path.exists("/opt/rocm/lib/llvm/bin/clang++") and shutil.which("nm")):
        pytest.skip("needs the ROCm clang toolchain")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan", "build_and_run.py"), str(tmp_path / "build")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=850)
    tail = r.stdout[-3000:]
    assert r.returncode == 0, tail
    assert "asan driver: all cases ran" in tail and "runtime error" not in r.stdout and "AddressSanitizer" not in r.stdout, tail
    assert tail.count("cfg5 per GPU") == 2  # both precisions reached the largest case
