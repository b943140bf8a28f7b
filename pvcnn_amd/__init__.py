"""pvcnn_amd -- MI355X (gfx950) native PVConv hot path behind the reference's operator API.

Layout (only what the hot path needs):
  csrc/      hand-written HIP kernels + the C ABI (include/pvcnn_hip.h) -> libpvcnn_hip.so
  _lib.py    ctypes binding of that C ABI (fails loudly when the library is missing)
  modules/   host-side mirror of the reference's `modules` package (same names, ctor/forward
             signatures, state_dict keys); `modules.functional.backend._backend` is the seam
  dp.py      one-process-per-GPU data-parallel gradient all-reduce (RCCL) harness
  workload.py  synthetic (B,C,N,R) inputs + the PVCNN/PVCNN++ network builders used by bench.py

`install_dropin()` registers `pvcnn_amd.modules` under the top-level name `modules`, so the
reference's unchanged `models/` (`from modules import PVConv, ...`) run on this backend.
"""
import importlib
import sys

__version__ = '0.1.0'

_DROPIN_SUBMODULES = (
    'ball_query', 'frustum', 'loss', 'pointnet', 'pvconv', 'se', 'shared_mlp', 'voxelization',
    'functional', 'functional.backend', 'functional.ball_query', 'functional.devoxelization',
    'functional.grouping', 'functional.interpolatation', 'functional.loss', 'functional.sampling',
    'functional.voxelization',
)


def install_dropin(name='modules'):
    """Alias pvcnn_amd.modules (and every sub-module the reference imports by path) as `name`."""
    impl = importlib.import_module('pvcnn_amd.modules')
    sys.modules[name] = impl
    for sub in _DROPIN_SUBMODULES:
        sys.modules[f'{name}.{sub}'] = importlib.import_module(f'pvcnn_amd.modules.{sub}')
    return impl
