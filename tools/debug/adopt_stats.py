import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import subprocess
for rad in ("0.01", "0.1", "0.5", "1.0", "3.0"):
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_hip_benchmarked_path.py", "-q", "-s", "-k", "distinct"], env=dict(os.environ, STTS_ADOPTION_RAD=rad), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    print("RAD", rad, "rc", r.returncode)
    for ln in out.splitlines():
        if "distinct utterances" in ln or "AssertionError" in ln or "assert " in ln: print("   ", ln[:400])
