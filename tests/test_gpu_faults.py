"""The tests that pin the benchmarked update path (tests/test_gpu_parity.py::test_benchmarked_update_path_matches_reference_optimizer,
tests/test_gpu_configs.py::test_replayed_default_bench_mode_against_the_oracle_at_full_size) are only worth having if a broken
update fails them.  The fault-injection build (`make -C csrc fault` -> librecnet_hip_fault.so, loaded with RN_LIB_VARIANT=fault;
never part of the product library) leaves ONE store of the Adam epilogue of the pending d W_hh product out — RN_FAULT = 1: the
updated parameters, 2: the bf16 operand image of W_hh, 3: its transposed image (the backward chain's operand), 4: the Adam moments —
and the pinned test has to fail for every one of them, and pass in the same build without a fault."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TARGET = "tests/test_gpu_parity.py::test_benchmarked_update_path_matches_reference_optimizer"


def _child(fault, case):
    env = dict(os.environ)
    env["RN_LIB_VARIANT"] = "fault"
    env["RN_FAULT"] = str(fault)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "%s[%s-graph_split_update-bf16]" % (TARGET, case)],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return r.returncode, r.stdout[-1500:]


@pytest.mark.parametrize("case", ["lr_global_chain", "lr_local_chain"])
def test_the_fault_build_without_a_fault_passes(case):
    rc, out = _child(0, case)
    assert rc == 0, out


@pytest.mark.parametrize("fault", [1, 2, 3, 4])
@pytest.mark.parametrize("case", ["lr_global_chain", "lr_local_chain"])
def test_a_skipped_store_of_the_adam_epilogue_fails_the_pinned_test(case, fault):
    rc, out = _child(fault, case)
    assert rc == 1, "fault %d went unnoticed:\n%s" % (fault, out)          # pytest: 1 = tests failed (not a crash, not a usage error)
    assert "passed" not in out.splitlines()[-1] or "failed" in out.splitlines()[-1], out
