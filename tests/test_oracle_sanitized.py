"""Sanitizer run of the CPU restatement (the reference has no race / memory checking; the C oracle is where an
out-of-bounds deck index or a signed overflow would hide): the golden-vector tests of tests/test_oracle_golden.py once more, in a
child interpreter, against oracle/libbalatro_oracle_asan.so (gcc -fsanitize=address,undefined).  CPU only -- GPU AddressSanitizer is not
available on the pool."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_golden_vectors_under_asan_ubsan():
    try:
        libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
        libubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("gcc not available")
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    lib = os.path.join(ROOT, "oracle", "libbalatro_oracle_asan.so")
    env = dict(os.environ, BALATRO_ORACLE_LIB=lib, LD_PRELOAD=f"{libasan} {libubsan}" if os.path.exists(libubsan) else libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the operator fixtures, the known answers, and the traces that reach the most code: jokers, boss blinds, shop, consumables, card states
    sel = ("test_mt_known_answers or test_deck_shuffle or test_classify_golden or test_score_hand_golden or test_reference_known_answers "
           "or test_sim_score_golden or c5_uniform_rich or consumables_scorer or boss_forced_scorer or cards_levels")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-k", sel,
                          "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, tail
    assert " passed" in out.stdout, tail
