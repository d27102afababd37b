import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "a-low-texture-robust-hybrid-feature-based-visual-odometry_amd")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """import the (hyphenated) product package under the alias hvo_amd"""
    if "hvo_amd" in sys.modules:
        return sys.modules["hvo_amd"]
    spec = importlib.util.spec_from_file_location(
        "hvo_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["hvo_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def load_oracle():
    """the CPU oracle -- test infrastructure only"""
    if "hvo_oracle" in sys.modules:
        return sys.modules["hvo_oracle"]
    spec = importlib.util.spec_from_file_location("hvo_oracle", os.path.join(ROOT, "oracle", "oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["hvo_oracle"] = mod
    spec.loader.exec_module(mod)
    return mod


def load_synth():
    load_pkg()
    return importlib.import_module("hvo_amd.synth")


@pytest.fixture(scope="session")
def hvo():
    return load_pkg()


@pytest.fixture(scope="session")
def orc():
    m = load_oracle()
    m.lib()
    return m


@pytest.fixture(scope="session")
def synth():
    return load_synth()


@pytest.fixture(scope="session")
def gpu_ctx(hvo):
    """one shared context for GPU tests (single process, single stream)"""
    ctx = hvo.Context(max_batch=4)
    yield ctx
    ctx.close()
