"""Shared fixtures.  `-m gpu` tests need a HIP device and call through the C-ABI;
everything else runs on the CPU (oracle, host stage, symbol export, gloo)."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("objective-slam_amd")


@pytest.fixture(scope="session")
def ppf(pkg):
    return pkg.ppf


@pytest.fixture(scope="session")
def synth(pkg):
    return pkg.synth


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def built_lib(ppf):
    if not os.path.exists(ppf.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return ppf.lib()


def make_case(synth, M, S, seed, tau_d=0.05, model_id=0, noise=0.0, instance_points=None):
    mp, mn = synth.make_model(model_id, M)
    d = synth.d_dist_for(mp, tau_d)
    sp, sn, poses = synth.make_scene([model_id], S, seed, instance_points=instance_points or min(M, S // 2),
                                     noise_sigma=noise * d)
    return dict(mp=mp, mn=mn, sp=sp, sn=sn, d=d, truth=poses[0][1])


@pytest.fixture(scope="session")
def case_small(synth):
    return make_case(synth, 200, 500, 2001)


@pytest.fixture(scope="session")
def case_two_slices(synth):
    # M > 2046: two table slices / two LDS accumulators per reference point (both 16-bit halves of the first in use)
    return make_case(synth, 2300, 1500, 2002)


def cells_equal(a, b):
    return (len(a) == len(b) and np.array_equal(a["code"], b["code"]) and np.array_equal(a["count"], b["count"]))
