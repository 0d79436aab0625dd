"""Results must not depend on how the work is cut into launches (workgroup shapes are chosen by launch size): the stress scripts of
tests/stress/ as tests.  b3_consistency found the round-4 bug of the bf16x3 layer-0 kernels (stale split planes of a site group below its
workgroup's level); lds_poison_check runs when its helper library has been built (tests/helpers_native/lds_poison.hip, by build())."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args, timeout=600):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stress", script), *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    return p.stdout


def test_forward_results_do_not_depend_on_launch_shapes():
    out = _run("b3_consistency.py")
    assert out.count("identical") == 5 and "DIFFER" not in out, out


def test_encode_results_do_not_depend_on_the_launch_partition():
    out = _run("encode_consistency.py")
    assert out.count("identical") == 4 and "DIFFER" not in out, out


def test_no_kernel_reads_lds_or_workspace_it_never_wrote():
    if not os.path.exists(os.path.join(ROOT, "tests", "helpers_native", "liblds_poison.so")):
        pytest.skip("tests/helpers_native/liblds_poison.so not built")
    out = _run("lds_poison_check.py")
    assert out.count("identical") == 2, out


def test_text_pipeline_on_adversarial_contigs():
    """tests/stress/pipeline_stress.py: three rounds (adversarial columns; positions that step back / repeat, cut alleles; the reader's corner
    cases - runs of tabs, CRLF, no quality column, atoll-style positions): every chunk size and the host-parsed run give the one-chunk
    device-tokenised VCF byte for byte; its sites and calls equal the oracle chain's"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stress", "pipeline_stress.py"), "3", "8000"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0 and p.stdout.count(": ok") == 3 and "DIFFER" not in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])
