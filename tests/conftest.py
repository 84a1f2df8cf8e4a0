import os
import sys
import pytest
# PyTorch wheels bundle a complete ROCm stack (libamdhip64, libhsa-runtime64, librccl ...) under torch/lib with the same SONAMEs as
# /opt/rocm's.  Whichever copy a process loads first serves everyone, EXCEPT that RCCL dlopens its HSA runtime by path: with
# libdvslam_hip.so (linked against /opt/rocm) loaded before torch, a second HSA runtime comes up and ncclCommInitRank fails with
# "no ROCm-capable device".  Tests that use both therefore import torch first, as bench.py does; a C++ host without torch only ever
# sees /opt/rocm's copy.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_bindings
    oracle_bindings.build()
    return oracle_bindings


@pytest.fixture(scope="session")
def hiplib():
    """The HIP library must exist and load; GPU tests additionally need a device (no fallback)."""
    from dvslam_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        _lib.build_library()
    return _lib.lib()


@pytest.fixture(scope="session")
def hooks(hiplib):
    """lib/libdvslam_hip_test.so (-DDVS_TEST_HOOKS): the dvs_test_* entry points of include/dvslam_hip_test.h, which the product library
    does not export"""
    from dvslam_amd import _lib
    return _lib.test_lib()


@pytest.fixture(scope="session")
def gpu(hiplib):
    from dvslam_amd import device_count
    n = device_count()
    assert n >= 1, "no HIP device visible: -m gpu tests must run on the MI355X box (there is no CPU fallback)"
    return n
