"""Cross-checks against the reference's OWN Python layers, executed here.

/root/reference's `modules/*.py` and `models/**` import fine once the native extension is
replaced at its seam (`modules.functional.backend._backend`).  With the
CPU oracle plugged into BOTH stacks, the reference's Python glue and pvcnn_amd's must agree
exactly: same state_dict keys/shapes, same forward outputs, same gradients.  These tests only
run where the reference tree is mounted (this build container); the GPU box relies on the
golden vectors in tests/golden/ that gen_golden.py produced from the same reference run.
"""
import importlib
import os
import sys
import types

import pytest
import torch

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'modules')), reason='reference tree not mounted')


@pytest.fixture()
def ref(oracle, oracle_seam):
    """Import the reference's `modules` / `models` with the oracle as their native backend."""
    saved = {k: v for k, v in sys.modules.items() if k == 'modules' or k.startswith('modules.') or k == 'models' or k.startswith('models.')}
    for k in saved:
        del sys.modules[k]
    fake = types.ModuleType('modules.functional.backend')
    fake._backend = oracle
    sys.modules['modules.functional.backend'] = fake
    sys.path.insert(0, REF)
    try:
        mods = importlib.import_module('modules')
        models = importlib.import_module('models.s3dis')
        yield types.SimpleNamespace(modules=mods, models=models, F=importlib.import_module('modules.functional'))
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == 'modules' or k.startswith('modules.') or k == 'models' or k.startswith('models.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def _same_state(a, b):
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    for k in sa:
        assert sa[k].shape == sb[k].shape, k


def test_pvconv_matches_reference(ref):
    from pvcnn_amd.modules import PVConv
    for kw in [dict(with_se=False, normalize=True), dict(with_se=True, normalize=False)]:
        torch.manual_seed(0)
        theirs = ref.modules.PVConv(9, 16, 3, 8, **kw)
        mine = PVConv(9, 16, 3, 8, **kw)
        _same_state(theirs, mine)
        mine.load_state_dict(theirs.state_dict())
        x = torch.rand(2, 9, 300)
        xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
        ya, ca = theirs((xa, xa[:, :3, :] if kw['normalize'] else xa[:, :3, :] - 0.5))
        yb, cb = mine((xb, xb[:, :3, :] if kw['normalize'] else xb[:, :3, :] - 0.5))
        assert torch.equal(ya, yb)
        ya.square().sum().backward(); yb.square().sum().backward()
        assert torch.equal(xa.grad, xb.grad)
        for (n1, p1), (n2, p2) in zip(theirs.named_parameters(), mine.named_parameters()):
            assert n1 == n2 and torch.equal(p1.grad, p2.grad), n1


def test_pointnet_modules_match_reference(ref):
    from pvcnn_amd.modules import PointNetSAModule, PointNetFPModule, PointNetAModule, BallQuery
    torch.manual_seed(1)
    feats, coords = torch.rand(2, 6, 256), torch.rand(2, 3, 256)
    a = ref.modules.PointNetSAModule(32, [0.2, 0.4], [8, 16], 6, [[8, 16], [8, 32]])
    b = PointNetSAModule(32, [0.2, 0.4], [8, 16], 6, [[8, 16], [8, 32]])
    _same_state(a, b); b.load_state_dict(a.state_dict())
    (fa, ca), (fb, cb) = a((feats, coords)), b((feats, coords))
    assert torch.equal(fa, fb) and torch.equal(ca, cb) and a.out_channels == b.out_channels
    a = ref.modules.PointNetFPModule(48 + 6, (32, 16)); b = PointNetFPModule(48 + 6, (32, 16))
    _same_state(a, b); b.load_state_dict(a.state_dict())
    assert torch.equal(a((coords, ca, fa, feats))[0], b((coords, cb, fb, feats))[0])
    a = ref.modules.PointNetAModule(6, [16, 32]); b = PointNetAModule(6, [16, 32])
    _same_state(a, b); b.load_state_dict(a.state_dict())
    assert torch.equal(a((feats, coords))[0], b((feats, coords))[0])
    assert torch.equal(ref.modules.BallQuery(0.3, 8)(coords, ca, feats), BallQuery(0.3, 8)(coords, cb, feats))


@pytest.mark.parametrize('name,n', [('PVCNN', 512), ('PVCNN2', 1024)])
def test_s3dis_networks_match_reference(ref, name, n):
    from pvcnn_amd import workload
    torch.manual_seed(2)
    theirs = getattr(ref.models, name)(13, 6, width_multiplier=0.125)
    mine = getattr(workload, name)(13, 6, width_multiplier=0.125)
    _same_state(theirs, mine)
    mine.load_state_dict(theirs.state_dict())
    theirs.eval(); mine.eval()
    x, _ = workload.make_s3dis_batch(2, n)
    with torch.no_grad():
        assert torch.equal(theirs(x), mine(x))


def test_shapenet_and_frustum_networks_match_reference(ref):
    """BASELINE configs[3] / configs[4] builders vs the reference's own classes (models/shapenet/pvcnn.py:9-42,
    models/kitti/frustum/frustum_net.py:14-113): same state_dict keys / shapes, same outputs."""
    import numpy as np
    from pvcnn_amd import workload
    sys.path.insert(0, REF)
    try:
        shapenet = importlib.import_module('models.shapenet')
        frustum = importlib.import_module('models.kitti.frustum')
    finally:
        sys.path.remove(REF)
    torch.manual_seed(4)
    theirs, mine = shapenet.PVCNN(50, 16, 3, 0.125), workload.PVCNNShapeNet(50, 16, 3, 0.125)
    _same_state(theirs, mine)
    mine.load_state_dict(theirs.state_dict())
    theirs.eval(); mine.eval()
    x, _ = workload.make_shapenet_batch(2, 512)
    assert x.shape == (2, 22, 512)
    with torch.no_grad():
        assert torch.equal(theirs(x), mine(x))
    templates = workload.frustum_size_templates()
    theirs = frustum.FrustumPVCNNE(3, 12, 8, 128, templates, 1, 0.25)
    mine = workload.FrustumPVCNNE(3, 12, 8, 128, templates, 1, 0.25)
    _same_state(theirs, mine)
    mine.load_state_dict(theirs.state_dict())
    theirs.eval(); mine.eval()
    inputs, _ = workload.make_frustum_batch(2, 256)
    with torch.no_grad():
        np.random.seed(5); want = theirs(inputs)
        np.random.seed(5); got = mine(inputs)
    assert list(want.keys()) == list(got.keys())
    for k in want:
        assert torch.equal(want[k], got[k]), k
    assert [m.resolution for m in mine.inst_seg_net.point_features[:4]] == [16, 16, 12, 12]


def test_full_width_parameter_counts(ref):
    # PVCNN 1xC 2,572,493 parameters; PVCNN++ 13,709,837
    from pvcnn_amd import workload
    assert sum(p.numel() for p in workload.PVCNN(13, 6, 1).parameters()) == 2572493
    assert sum(p.numel() for p in workload.PVCNN2(13, 6, 1).parameters()) == 13709837
    sys.path.insert(0, REF)
    try:
        shapenet = importlib.import_module('models.shapenet')
    finally:
        sys.path.remove(REF)
    assert (sum(p.numel() for p in workload.PVCNNShapeNet(50, 16, 3, 1).parameters())
            == sum(p.numel() for p in shapenet.PVCNN(50, 16, 3, 1).parameters()) == 4200434)


def test_reference_models_run_unchanged_on_the_dropin(oracle_seam):
    """`install_dropin()` makes the reference's unmodified models/ import pvcnn_amd.modules."""
    import pvcnn_amd
    saved = {k: v for k, v in sys.modules.items() if k.split('.')[0] in ('modules', 'models')}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, REF)
    try:
        impl = pvcnn_amd.install_dropin()
        models = importlib.import_module('models.s3dis')
        import modules
        assert modules is impl and modules.PVConv.__module__.startswith('pvcnn_amd.')
        net = models.PVCNN(13, 6, width_multiplier=0.125).eval()
        assert isinstance(net.point_features[0], impl.PVConv)
        with torch.no_grad():
            assert net(torch.rand(1, 9, 256)).shape == (1, 13, 256)
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k.split('.')[0] in ('modules', 'models')]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_baseline_config0_cpu_plumbing(oracle_seam):
    """BASELINE.json configs[0]: PVCNN (0.125xC) S3DIS, B=1, N=4096, R=32 on the CPU reference path
    (eval mode: BatchNorm1d on a (1,C) tensor raises in train mode, models/utils.py:11-12)."""
    from pvcnn_amd import workload
    torch.manual_seed(workload.SEED)
    net = workload.PVCNN(13, 6, width_multiplier=0.125).eval()
    x, _ = workload.make_s3dis_batch(1, 4096)
    with torch.no_grad():
        logits = net(x)
    assert logits.shape == (1, 13, 4096) and torch.isfinite(logits).all()
    assert [m.resolution for m in net.point_features[:4]] == [32, 16, 16, 16]


@pytest.mark.parametrize('name,ref_name,n', [('ReferencePVCNN', 'PVCNN', 512), ('ReferencePVCNN2', 'PVCNN2', 1024)])
def test_reference_composition_is_the_reference_forward(ref, name, ref_name, n):
    """tests/reference_composition.py claims to compose this package's modules exactly as the reference's forward() methods do
    (models/s3dis/pvcnn.py:34-46, pvcnnpp.py:44-59).  Pinned here against the reference's own classes on the CPU oracle stack, TRAIN
    mode (dropout p = 0: the two models would otherwise draw different masks from one generator), forward and every gradient: bit-equal."""
    import reference_composition as rc
    from pvcnn_amd import workload
    torch.manual_seed(3)
    theirs = getattr(ref.models, ref_name)(13, 6, width_multiplier=0.125)
    mine = getattr(rc, name)(13, 6, width_multiplier=0.125)
    _same_state(theirs, mine)
    mine.load_state_dict(theirs.state_dict())
    for net in (theirs, mine):
        net.train()
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    x, y = workload.make_s3dis_batch(2, n)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    la = torch.nn.functional.cross_entropy(theirs(xa), y)
    lb = torch.nn.functional.cross_entropy(mine(xb), y)
    assert torch.equal(la, lb)
    la.backward(); lb.backward()
    assert torch.equal(xa.grad, xb.grad)
    for (n1, p1), (n2, p2) in zip(theirs.named_parameters(), mine.named_parameters()):
        assert n1 == n2 and torch.equal(p1.grad, p2.grad), n1
    for (n1, b1), (n2, b2) in zip(theirs.named_buffers(), mine.named_buffers()):
        assert n1 == n2 and torch.equal(b1, b2), n1


def test_reference_composition_shapenet_is_the_reference_forward(ref):
    import reference_composition as rc
    from pvcnn_amd import workload
    sys.path.insert(0, REF)
    try:
        shapenet = importlib.import_module('models.shapenet')
    finally:
        sys.path.remove(REF)
    torch.manual_seed(4)
    theirs, mine = shapenet.PVCNN(50, 16, 3, 0.125), rc.ReferencePVCNNShapeNet(50, 16, 3, 0.125)
    _same_state(theirs, mine)
    mine.load_state_dict(theirs.state_dict())
    theirs.eval(); mine.eval()
    x, _ = workload.make_shapenet_batch(2, 512)
    with torch.no_grad():
        assert torch.equal(theirs(x), mine(x))


def test_adopt_keeps_a_reference_model_and_its_state_dict(ref):
    """pvcnn_amd.adopt(model) on the reference's own class instance: same parameters / state_dict keys, same outputs and gradients (on
    CPU tensors the adopted forwards fall back to the modules; the GPU side is tests/test_gpu_reference_composition.py)."""
    import pvcnn_amd
    from pvcnn_amd import workload
    torch.manual_seed(5)
    for name, n in (('PVCNN', 512), ('PVCNN2', 1024)):
        plain = getattr(ref.models, name)(13, 6, width_multiplier=0.125)
        adopted = getattr(ref.models, name)(13, 6, width_multiplier=0.125)
        adopted.load_state_dict(plain.state_dict())
        keys = list(adopted.state_dict().keys())
        assert pvcnn_amd.adopt(adopted) is adopted and list(adopted.state_dict().keys()) == keys
        assert type(adopted).__name__ == name and isinstance(adopted, getattr(ref.models, name))
        assert type(adopted).forward is getattr(workload, name).forward
        for net in (plain, adopted):
            net.train()
            for m in net.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
        x, y = workload.make_s3dis_batch(2, n)
        la = torch.nn.functional.cross_entropy(plain(x), y)
        lb = torch.nn.functional.cross_entropy(adopted(x), y)
        assert torch.equal(la, lb)
        la.backward(); lb.backward()
        for (n1, p1), (n2, p2) in zip(plain.named_parameters(), adopted.named_parameters()):
            assert n1 == n2 and torch.equal(p1.grad, p2.grad), n1


def test_adopt_of_the_reference_frustum_net(ref):
    """(round 6) pvcnn_amd.adopt(model) on the reference's own models.kitti.frustum.FrustumPVCNNE instance (frustum_net.py:105-113):
    the three sub-nets take workload's forwards, parameters / state_dict keys / class names stay, every returned head and every
    gradient is the plain instance's (CPU: the adopted forwards fall back to the modules), and the adopted model pickles."""
    import pickle
    import numpy as np
    import pvcnn_amd
    from pvcnn_amd import workload
    frustum = importlib.import_module('models.kitti.frustum')
    templates = workload.frustum_size_templates()
    torch.manual_seed(6)
    plain = frustum.FrustumPVCNNE(3, 12, 8, 64, templates, 1, 0.125)
    adopted = frustum.FrustumPVCNNE(3, 12, 8, 64, templates, 1, 0.125)
    adopted.load_state_dict(plain.state_dict())
    keys = list(adopted.state_dict().keys())
    assert pvcnn_amd.adopt(adopted) is adopted and list(adopted.state_dict().keys()) == keys
    assert type(adopted.inst_seg_net).forward is workload._FrustumSegmentation.forward
    assert type(adopted.inst_seg_net).__name__ == 'InstanceSegmentationPVCNN'
    assert type(adopted.center_reg_net).forward is workload._CloudRegressor.forward and not adopted.center_reg_net._coords_tuple
    assert type(adopted.box_est_net).forward is workload._CloudRegressor.forward and adopted.box_est_net._coords_tuple
    assert pvcnn_amd.adopt(adopted) is adopted                          # idempotent
    inputs, _ = workload.make_frustum_batch(3, 256)
    outs = []
    for net in (plain, adopted):
        net.train()
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        np.random.seed(3)                                               # logits_mask's host draws (functional/sampling.py:69-82)
        out = net(inputs)
        sum(v.float().square().mean() for v in out.values() if v.dtype.is_floating_point).backward()
        outs.append(out)
    assert outs[0].keys() == outs[1].keys()
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    for (n1, p1), (n2, p2) in zip(plain.named_parameters(), adopted.named_parameters()):
        assert n1 == n2 and (p1.grad is None) == (p2.grad is None) and (p1.grad is None or torch.equal(p1.grad, p2.grad)), n1
    clone = pickle.loads(pickle.dumps(adopted))
    assert type(clone) is type(adopted) and type(clone.inst_seg_net) is type(adopted.inst_seg_net)
    assert list(clone.state_dict().keys()) == keys
