import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """The package directory is named `ik_llama.cpp_amd` (not an importable identifier): import it as ik_llama_cpp_amd."""
    name = "ik_llama_cpp_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg_dir = os.path.join(ROOT, "ik_llama.cpp_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def oracle():
    from oracle.bindings import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The real reference library; tests that need it are skipped when this host cannot run it."""
    from oracle.bindings import Ref, ref_path
    if ref_path() is None:
        pytest.skip("oracle/_ref/libggml_ref_*.so not available / not runnable on this CPU")
    return Ref()


@pytest.fixture(scope="session")
def backend(pkg):
    """The HIP backend on cuda:0.  No fallback: a missing library or GPU is an error for -m gpu tests."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test started without a visible GPU"
    be = pkg.Cdna4Backend(0)
    yield be
    be.close()
