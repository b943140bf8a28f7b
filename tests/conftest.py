"""Shared test fixtures.

Markers
  gpu : needs a real MI355X (run with `-m gpu` on the GPU box).  Everything else runs on CPU.

The oracle (oracle/) is imported ONLY here and in test files: it is the checker, never the
thing under test on the GPU path.  CPU-only tests exercise the host-side Python layers by
swapping the oracle in at the one seam the reference itself has
(`modules.functional.backend._backend`).
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 1588147245   # the reference's seed (configs/__init__.py:3)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X GPU (deselected on the CPU box)')
    # a hung test must FAIL, not stall the GPU box for its whole time limit (pytest-timeout, when installed)
    if config.pluginmanager.hasplugin('timeout') and not getattr(config.option, 'timeout', None):
        config.option.timeout = 600


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle backend (builds oracle/libpvcnn_oracle.so with gcc on first use)."""
    from oracle import oracle_backend
    oracle_backend.build()
    return oracle_backend.OracleBackend()


@pytest.fixture()
def oracle_seam(oracle, monkeypatch):
    """Route pvcnn_amd's functional layer to the CPU oracle for the duration of one test."""
    from pvcnn_amd.modules.functional import backend as be
    monkeypatch.setattr(be, '_backend', oracle)
    return oracle


@pytest.fixture()
def hip():
    """The product backend (libpvcnn_hip.so).  Fails -- never skips -- if the library is missing."""
    from pvcnn_amd.modules.functional.backend import HipBackend
    b = HipBackend()
    b.lib   # force the dlopen
    return b


@pytest.fixture()
def gen():
    g = torch.Generator()
    g.manual_seed(SEED)
    return g


def synth_cloud(gen, b, n, kind='s3dis', dup=0.05):
    """Synthetic coordinates (B,3,N) that mirror the reference loaders."""
    if kind == 's3dis':        # block-local metres: U[0,1.5] x U[0,1.5] x U[0,3]
        c = torch.rand(b, 3, n, generator=gen) * torch.tensor([1.5, 1.5, 3.0]).view(1, 3, 1)
    elif kind == 'surface':    # points on three planes: ~10 % voxel occupancy, atomic hot spots
        c = torch.rand(b, 3, n, generator=gen)
        plane = torch.randint(0, 3, (b, n), generator=gen)
        for ax in range(3):
            c[:, ax, :] = torch.where(plane == ax, torch.full_like(c[:, ax, :], 0.5), c[:, ax, :])
    else:                      # unit cube
        c = torch.rand(b, 3, n, generator=gen)
    ndup = int(n * dup)
    if ndup > 0:               # exact duplicates (the S3DIS loader samples with replacement)
        src = torch.randint(0, n, (b, ndup), generator=gen)
        dst = torch.randint(0, n, (b, ndup), generator=gen)
        for bi in range(b):
            c[bi, :, dst[bi]] = c[bi, :, src[bi]]
    return c.contiguous()


def grid_coords(gen, b, n, r):
    """Float grid coordinates in [0, r-1] with a mix of fractional, integral and boundary values."""
    c = torch.rand(b, 3, n, generator=gen) * (r - 1)
    k = max(1, n // 16)
    c[:, :, :k] = torch.round(c[:, :, :k])          # integral: weight 1 on corner 000
    c[:, :, k:2 * k] = 0.0                          # lower boundary
    c[:, :, 2 * k:3 * k] = float(r - 1)             # upper boundary (never reads out of bounds)
    c[:, 0, 3 * k:4 * k] = torch.round(c[:, 0, 3 * k:4 * k])   # integral in x only
    return c.contiguous()
