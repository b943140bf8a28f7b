"""checkpoint-compatible loader for the reference's `.pth.tar` files (train.py:186-199: DataParallel
`module.` keys, pickled config object) and the prefetching input pipeline (CPU pass-through here, GPU in test_gpu_*)."""
import os
import sys
import types

import pytest
import torch

from conftest import ROOT

REF = '/root/reference'


def test_reference_style_checkpoint_round_trip(tmp_path, oracle_seam):
    from pvcnn_amd import workload
    from pvcnn_amd.checkpoint import load_reference_checkpoint, save_reference_checkpoint, strip_module_prefix
    torch.manual_seed(0)
    src = workload.PVCNN(13, 6, width_multiplier=0.125)
    # what the reference's train.py writes: a DataParallel state_dict + an object of its own Config class
    fake_mod = types.ModuleType('utils_for_test_config')
    exec('class Config(dict):\n    pass\n', fake_mod.__dict__)
    Config = fake_mod.Config
    Config.__module__, Config.__qualname__ = 'utils_for_test_config', 'Config'
    sys.modules['utils_for_test_config'] = fake_mod
    path = tmp_path / 'latest.pth.tar'
    try:
        torch.save({'epoch': 7, 'model': {'module.' + k: v for k, v in src.state_dict().items()}, 'optimizer': None,
                    'meters': {'acc/iou_test_best': 0.5}, 'configs': Config(model='pvcnn')}, path)
    finally:
        del sys.modules['utils_for_test_config']        # the loader must cope without the reference's packages
    dst = workload.PVCNN(13, 6, width_multiplier=0.125)
    meta = load_reference_checkpoint(str(path), dst)
    assert meta['epoch'] == 7 and meta['meters']['acc/iou_test_best'] == 0.5
    for (ka, va), (kb, vb) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    # wrapped target (DataParallel keeps the prefix) and the save direction
    wrapped = torch.nn.DataParallel(workload.PVCNN(13, 6, width_multiplier=0.125))
    load_reference_checkpoint(str(path), wrapped)
    assert torch.equal(wrapped.module.classifier[-1].weight, src.classifier[-1].weight)
    out = tmp_path / 'mine.pth.tar'
    save_reference_checkpoint(str(out), dst, epoch=3)
    blob = torch.load(out, weights_only=False)
    assert all(k.startswith('module.') for k in blob['model']) and blob['epoch'] == 3
    assert list(strip_module_prefix(blob['model']).keys()) == list(src.state_dict().keys())
    x, _ = workload.make_s3dis_batch(1, 256)
    src.eval(); dst.eval()
    with torch.no_grad():
        assert torch.equal(src(x), dst(x))


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'models')), reason='reference tree not mounted')
def test_checkpoint_written_from_the_reference_model_class_loads(tmp_path, oracle, oracle_seam):
    """state_dict of the reference's OWN model class (wrapped like train.py does) -> this package's network."""
    import importlib
    from pvcnn_amd import workload
    from pvcnn_amd.checkpoint import load_reference_checkpoint
    saved = {k: v for k, v in sys.modules.items() if k.split('.')[0] in ('modules', 'models')}
    for k in saved:
        del sys.modules[k]
    fake = types.ModuleType('modules.functional.backend')
    fake._backend = oracle
    sys.modules['modules.functional.backend'] = fake
    sys.path.insert(0, REF)
    try:
        ref_models = importlib.import_module('models.s3dis')
        torch.manual_seed(1)
        theirs = torch.nn.DataParallel(ref_models.PVCNN2(13, 6, width_multiplier=0.125))
        path = tmp_path / 'ref.pth.tar'
        torch.save({'epoch': 1, 'model': theirs.state_dict(), 'optimizer': None, 'meters': {}}, path)
        mine = workload.PVCNN2(13, 6, width_multiplier=0.125)
        load_reference_checkpoint(str(path), mine)
        x, _ = workload.make_s3dis_batch(1, 512)
        theirs.eval(); mine.eval()
        with torch.no_grad():
            assert torch.equal(theirs.module(x), mine(x))
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k.split('.')[0] in ('modules', 'models')]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_prefetcher_is_a_pass_through_on_cpu():
    from pvcnn_amd.pipeline import DevicePrefetcher, synthetic_stream
    from pvcnn_amd import workload
    batches = list(synthetic_stream(lambda i: workload.make_s3dis_batch(2, 64, seed=i), 3))
    got = list(DevicePrefetcher(iter(batches), 'cpu'))
    assert len(got) == 3
    for (x, y), (gx, gy) in zip(batches, got):
        assert torch.equal(x, gx) and torch.equal(y, gy)
    dicts = list(DevicePrefetcher(iter([{'features': torch.ones(2, 4, 8), 'one_hot_vectors': torch.zeros(2, 3)}]), 'cpu'))
    assert set(dicts[0]) == {'features', 'one_hot_vectors'}


@pytest.mark.gpu
def test_prefetcher_overlaps_and_delivers_on_gpu():
    from pvcnn_amd.pipeline import DevicePrefetcher, synthetic_stream
    from pvcnn_amd import workload
    batches = list(synthetic_stream(lambda i: workload.make_s3dis_batch(4, 1024, seed=i), 5))
    seen = 0
    for (x, y), (gx, gy) in zip(batches, DevicePrefetcher(iter(batches), 'cuda:0')):
        assert gx.is_cuda and torch.equal(gx.cpu(), x) and torch.equal(gy.cpu(), y)
        (gx * 2).sum().item()
        seen += 1
    assert seen == 5
