"""Loading / saving checkpoints in the reference's format (train.py:186-199, 249-255).

The reference saves `{'epoch', 'model', 'optimizer', 'meters', 'configs'}` with `model` = the state_dict of an
`nn.DataParallel` wrapper, i.e. every key carries a `module.` prefix, and `configs` = a pickled object of its own
`utils.config.Config` class.  The modules of this package keep the reference's parameter names, so the released
`.pth.tar` files load once (a) the prefix is reconciled with how the target model is wrapped and (b) the pickled
`configs` object is tolerated without the reference's `utils` package being importable.
"""
import pickle

import torch

__all__ = ['load_reference_checkpoint', 'save_reference_checkpoint', 'strip_module_prefix']

_PREFIX = 'module.'


def strip_module_prefix(state_dict):
    """`module.x.y` -> `x.y` for every key (only if ALL keys carry the prefix: a DataParallel state_dict)."""
    keys = list(state_dict.keys())
    if keys and all(k.startswith(_PREFIX) for k in keys):
        return type(state_dict)((k[len(_PREFIX):], v) for k, v in state_dict.items())
    return state_dict


class _Opaque(dict):
    """Stand-in for classes of the reference's own packages (utils.config.Config -- a dict subclass --, ...) found in a
    checkpoint: accepts whatever the pickle stream does to it (construction, item / attribute state, appends)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.args, self.kwargs = args, kwargs

    def __setstate__(self, state):
        self.state = state

    def append(self, item):
        self.setdefault('_items', []).append(item)

    def extend(self, items):
        self.setdefault('_items', []).extend(items)

    def __call__(self, *args, **kwargs):
        return _Opaque(*args, **kwargs)


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return _Opaque


class _TolerantPickle:
    """pickle_module for torch.load: unknown classes become inert placeholders instead of ImportErrors."""
    __name__ = 'pvcnn_amd.checkpoint.tolerant_pickle'
    Unpickler = _TolerantUnpickler
    load = staticmethod(lambda f, **kw: _TolerantUnpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump, dumps, HIGHEST_PROTOCOL, PicklingError, UnpicklingError = (pickle.dump, pickle.dumps, pickle.HIGHEST_PROTOCOL,
                                                                      pickle.PicklingError, pickle.UnpicklingError)


def load_reference_checkpoint(path, model, optimizer=None, map_location='cpu', strict=True):
    """Load a reference `.pth.tar` (or a bare state_dict file) into `model`.
    -> {'epoch': int, 'meters': dict}.  `model` may itself be wrapped (DataParallel / DistributedDataParallel):
    the `module.` prefix is added or removed as needed."""
    blob = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_TolerantPickle)
    state = blob['model'] if isinstance(blob, dict) and 'model' in blob else blob
    state = strip_module_prefix(state)
    wrapped = all(k.startswith(_PREFIX) for k in model.state_dict().keys()) and len(model.state_dict()) > 0
    if wrapped:
        state = type(state)((_PREFIX + k, v) for k, v in state.items())
    model.load_state_dict(state, strict=strict)
    meta = {'epoch': -1, 'meters': {}}
    if isinstance(blob, dict):
        if optimizer is not None and blob.get('optimizer') is not None:
            optimizer.load_state_dict(blob['optimizer'])
        meta['epoch'] = blob.get('epoch', -1)
        meta['meters'] = blob.get('meters', {})
    return meta


def save_reference_checkpoint(path, model, optimizer=None, epoch=0, meters=None):
    """Write a checkpoint the reference's train.py / evaluate scripts load: DataParallel-style `module.` keys."""
    state = strip_module_prefix(model.state_dict())
    torch.save({'epoch': epoch, 'model': type(state)((_PREFIX + k, v) for k, v in state.items()),
                'optimizer': optimizer.state_dict() if optimizer is not None else None, 'meters': meters or {}}, path)
