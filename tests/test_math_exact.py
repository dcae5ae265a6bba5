"""CPU: ppf_math.h / ppf_core.h against the host libm.

pm_acosf and pm_atanf are compared with glibc on every one of the 2^32 floats; pm_atan2f,
the exact quantisation, the alpha-bin threshold table and the quantised-angle scheme on
hundreds of millions of structured and random inputs; the tabulated acos bins (pc_acos_bin) against the quantised
libm acosf on every float, and the key rebuilt from a pair's bins against the hashed key on 3*10^8 pairs; the threshold table is re-derived
from every ratio in [2^-63, 2^63].  These are the functions the GPU kernels run, so
bit-exact PPF keys and reference-identical alpha bins rest on this file."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "objective-slam_amd", "csrc")
BUILD = os.path.join(ROOT, "build")


@pytest.fixture(scope="module")
def exe():
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, "math_exhaustive")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-I", CSRC,
                    os.path.join(ROOT, "tests", "native", "math_exhaustive.c"), "-o", out, "-lm"], check=True)
    return out


@pytest.mark.parametrize("mode,arg", [("acosf", "1"), ("atanf", "1"), ("atan2f", "400000000"),
                                      ("quant", "200000000"), ("alphabin", "400000000"),
                                      ("hybrid", "400000000"), ("acosbin", "1"), ("pairbins", "300000000")])
def test_against_libm(exe, mode, arg):
    r = subprocess.run([exe, mode, arg], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches=0" in r.stdout


def test_alpha_threshold_table_is_exhaustively_derived():
    out = os.path.join(BUILD, "gen_alpha_table_check")
    os.makedirs(BUILD, exist_ok=True)
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-DCHECK_HEADER", "-I", CSRC,
                    os.path.join(ROOT, "tools", "gen_alpha_table.c"), "-o", out, "-lm"], check=True)
    r = subprocess.run([out, "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
