"""Fused BatchNorm + ReLU/LeakyReLU (csrc/bnact.hip) against torch's own modules (fp64 reference on
the GPU).  The reference delegates BatchNorm to cuDNN (unpinned summation order): tolerance 1e-5
relative to each tensor's scale, running statistics and num_batches_tracked included."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)


@pytest.mark.parametrize('shape,bn_cls,eps,slope', [((4, 64, 16, 16, 16), nn.BatchNorm3d, 1e-4, 0.1),
                                                    ((2, 9, 32, 32, 32), nn.BatchNorm3d, 1e-4, 0.1),
                                                    ((3, 128, 4096), nn.BatchNorm1d, 1e-5, 0.0),
                                                    ((2, 35, 257, 9), nn.BatchNorm2d, 1e-5, 0.0),
                                                    ((2, 7, 1001), nn.BatchNorm1d, 1e-5, 0.0)])
@pytest.mark.parametrize('training', [True, False])
def test_bn_act_matches_torch(hip, shape, bn_cls, eps, slope, training):
    from pvcnn_amd.modules.functional.bnact import run_layers
    torch.manual_seed(0)
    act = nn.LeakyReLU(slope, True) if slope else nn.ReLU(True)
    mine = nn.Sequential(bn_cls(shape[1], eps=eps), act).to(DEV)
    ref = nn.Sequential(bn_cls(shape[1], eps=eps), nn.LeakyReLU(slope) if slope else nn.ReLU()).to(DEV).double()
    with torch.no_grad():
        mine[0].weight.uniform_(0.5, 1.5); mine[0].bias.uniform_(-0.5, 0.5)
        mine[0].running_mean.uniform_(-0.2, 0.2); mine[0].running_var.uniform_(0.5, 1.5)
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in mine.state_dict().items()})
    mine.train(training); ref.train(training)
    x = (torch.randn(*shape, device=DEV) * 2 + 0.7)
    xa, xb = x.clone().requires_grad_(), x.double().requires_grad_()
    g = torch.randn(*shape, device=DEV)
    for _ in range(2):      # two steps: running statistics accumulate
        ya = run_layers(mine, xa); yb = ref(xb)
    ya.backward(g); yb.backward(g.double())
    assert _rel(ya, yb.detach()) < 1e-5
    assert _rel(xa.grad, xb.grad) < 2e-5
    assert _rel(mine[0].weight.grad, ref[0].weight.grad) < 2e-5 and _rel(mine[0].bias.grad, ref[0].bias.grad) < 2e-5
    assert _rel(mine[0].running_mean, ref[0].running_mean) < 1e-5 and _rel(mine[0].running_var, ref[0].running_var) < 1e-5
    assert int(mine[0].num_batches_tracked) == int(ref[0].num_batches_tracked)


def test_bn_act_fused_into_devoxelize_is_bit_identical(hip):
    """PVConv with its last BatchNorm3d + LeakyReLU fused into the devoxelize gather vs the two separate ops:
    same bits forward and backward (the gather applies bnact_apply_kernel's expressions while staging)."""
    import copy
    from pvcnn_amd.modules import PVConv
    from pvcnn_amd.modules.functional._autograd import native
    torch.manual_seed(11)
    dev = 'cuda:0'
    for r, cin, cout in [(16, 9, 32), (32, 16, 16)]:
        fused = PVConv(cin, cout, 3, r).to(dev).train()
        plain = copy.deepcopy(fused)
        feats = torch.randn(2, cin, 1024, device=dev)
        coords = torch.rand(2, 3, 1024, device=dev) * 2 - 1
        fa, fb = feats.clone().requires_grad_(), feats.clone().requires_grad_()
        ya, _ = fused((fa, coords))
        ya.square().sum().backward()
        be = native()
        assert be.has_devox_bnact
        try:
            type(be).has_devox_bnact = False
            yb, _ = plain((fb, coords))
            yb.square().sum().backward()
        finally:
            type(be).has_devox_bnact = True
        assert torch.equal(ya, yb)
        assert torch.equal(fa.grad, fb.grad)
        for (na, pa), (nb, pb) in zip(fused.named_parameters(), plain.named_parameters()):
            assert torch.equal(pa.grad, pb.grad), na
        for (na, ba), (nb, bb) in zip(fused.named_buffers(), plain.named_buffers()):
            assert torch.equal(ba, bb), na
        # eval mode: running statistics
        fused.eval(); plain.eval()
        with torch.no_grad():
            ya, _ = fused((feats, coords))
            try:
                type(be).has_devox_bnact = False
                yb, _ = plain((feats, coords))
            finally:
                type(be).has_devox_bnact = True
        assert torch.equal(ya, yb)


def test_backward_kernels_take_batch_strided_gradients(hip):
    """Gradients that are channel slices of a wider tensor (torch.cat's backward) are consumed in place:
    same bits as with a contiguous copy, for the BN+act backward and the devoxelize backward."""
    dev = 'cuda:0'
    torch.manual_seed(5)
    b, c, n, r = 3, 24, 1000, 8
    wide = torch.randn(b, c + 40, n, device=dev)
    g_view = wide[:, 8:8 + c, :]
    assert not g_view.is_contiguous()
    g_copy = g_view.contiguous()
    x = torch.randn(b, c, n, device=dev)
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    _, mean, rstd = hip.bnact_forward(x, gamma, beta, None, None, True, 0.1, 1e-5, 0.1)
    for a, bb in zip(hip.bnact_backward(x, g_view, gamma, beta, mean, rstd, 0.1, True),
                     hip.bnact_backward(x, g_copy, gamma, beta, mean, rstd, 0.1, True)):
        assert torch.equal(a, bb)
    coords = torch.rand(b, 3, n, device=dev) * (r - 1)
    grid = torch.randn(b, c, r ** 3, device=dev)
    _, inds, wgts = hip.trilinear_devoxelize_forward(r, True, coords, grid)
    assert torch.equal(hip.trilinear_devoxelize_backward(g_view, inds, wgts, r),
                       hip.trilinear_devoxelize_backward(g_copy, inds, wgts, r))
    # dense-target path of the scatter (many points per voxel)
    coords2 = torch.rand(b, 3, n, device=dev) * 3
    _, inds2, wgts2 = hip.trilinear_devoxelize_forward(4, True, coords2, torch.randn(b, c, 64, device=dev))
    assert torch.equal(hip.trilinear_devoxelize_backward(g_view, inds2, wgts2, 4),
                       hip.trilinear_devoxelize_backward(g_copy, inds2, wgts2, 4))


@pytest.mark.parametrize('path', ['stats-pass', 'conv3d-epilogue', 'pwconv-epilogue'])
def test_batch_statistics_when_the_mean_dwarfs_the_spread(hip, path):
    """mean = 1e3, std = 1e-2: a variance from fp32 sums as E[x^2] - E[x]^2 has no correct digit left (it clamps to 0 and
    rstd becomes 1/sqrt(eps)); torch's BatchNorm gets it right.  The kernels accumulate SHIFTED sums -- around the
    channel's first element in the statistics pass, around the convolution's bias in the epilogue statistics -- and must
    agree with an fp64 BatchNorm of the same fp32 tensor: running_var to 1e-2 relative, output to 2e-2 of its scale (the
    normalisation y * scale + shift cancels 5 leading digits in fp32 whatever the statistics)."""
    from pvcnn_amd.modules.functional.bnact import run_layers
    torch.manual_seed(3)
    if path == 'stats-pass':
        net = nn.Sequential(nn.BatchNorm1d(16), nn.ReLU(True)).to(DEV).train()
        x = (torch.randn(4, 16, 2048, device=DEV, dtype=torch.float64) * 1e-2 + 1e3).float()
    elif path == 'pwconv-epilogue':
        net = nn.Sequential(nn.Conv1d(8, 16, 1), nn.BatchNorm1d(16), nn.ReLU(True)).to(DEV).train()
        with torch.no_grad():
            net[0].weight.mul_(1e-2); net[0].bias.fill_(1e3)
        x = torch.randn(4, 8, 2048, device=DEV)
    else:
        from pvcnn_amd.modules.pvconv import _VoxelConv3d
        net = nn.Sequential(_VoxelConv3d(4, 64, 3, stride=1, padding=1), nn.BatchNorm3d(64, eps=1e-4), nn.LeakyReLU(0.1, True)).to(DEV).train()
        with torch.no_grad():
            net[0].weight.mul_(1e-2); net[0].bias.fill_(1e3)
        x = torch.randn(2, 4, 8, 8, 8, device=DEV)
    import copy
    bn = [m for m in net if isinstance(m, nn.modules.batchnorm._BatchNorm)][0]
    tail = nn.Sequential(*[copy.deepcopy(m) for m in net if not isinstance(m, nn.modules.conv._ConvNd)]).double()
    bn_ref = tail[0]
    with torch.no_grad():
        y = net[0](x) if path != 'stats-pass' else x          # the fp32 tensor whose statistics are taken (it carries the
        #                                                       spread with only ~2-3 digits: the yardstick is fp64 BN of IT)
    got = run_layers(net, x)
    want = tail(y.double())
    assert _rel(bn.running_mean, bn_ref.running_mean) < 1e-6
    assert _rel(bn.running_var, bn_ref.running_var) < 1e-2, (bn.running_var[:4], bn_ref.running_var[:4])
    assert (got.double() - want).abs().max().item() < 2e-2 * want.abs().max().item()


@pytest.mark.parametrize('r,cin,cout,n,normalize', [(16, 9, 32, 1024, True), (32, 16, 24, 2048, True), (8, 12, 64, 600, False), (12, 6, 16, 1000, True)])
def test_se_tail_fused_into_the_gather_matches_the_separate_modules(hip, r, cin, cout, n, normalize):
    """PVConv WITH squeeze-and-excitation (cfg3 / cfg4): BatchNorm3d + LeakyReLU + SE3d + devoxelize + point-branch sum as one node
    (functional/bnact.py: BatchNormActSEDevoxelize) vs the same modules run one by one on the GPU.  Not bit-identical by construction:
    the squeeze is ONE sum over the grid instead of the reference's mean of means, and the BatchNorm-backward sums are assembled from
    per-(cloud, channel) partial sums -- 1e-5 of the tensor's largest entry, forward and every gradient, train and eval mode."""
    import copy
    from pvcnn_amd.modules import PVConv
    from pvcnn_amd.modules.functional._autograd import native
    torch.manual_seed(13)
    dev = 'cuda:0'
    fused = PVConv(cin, cout, 3, r, with_se=True, normalize=normalize).to(dev).train()
    plain = copy.deepcopy(fused)
    feats = torch.randn(3, cin, n, device=dev)
    coords = torch.rand(3, 3, n, device=dev) * (1.0 if normalize else 0.9) * 2 - 1
    if not normalize:
        coords = coords * 0.5
    wgt = torch.randn(3, cout, n, device=dev)
    fa, fb = feats.clone().requires_grad_(), feats.clone().requires_grad_()
    be = native()

    def rel(a, b):
        return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    ya, _ = fused((fa, coords))
    (ya * wgt).sum().backward()
    try:
        type(be).has_devox_bnact = False
        yb, _ = plain((fb, coords))
        (yb * wgt).sum().backward()
    finally:
        type(be).has_devox_bnact = True
    assert rel(ya, yb) < 1e-5, rel(ya, yb)
    assert rel(fa.grad, fb.grad) < 2e-5, rel(fa.grad, fb.grad)
    for (na, pa), (nb, pb) in zip(fused.named_parameters(), plain.named_parameters()):
        if pa.grad is None:
            assert pb.grad is None
            continue
        # a bias in front of a train-mode BatchNorm has a zero true gradient: judged against its layer's weight gradient
        scale = max(pb.grad.abs().max().item(), dict(plain.named_parameters())[na.replace('.bias', '.weight')].grad.abs().max().item()
                    if na.endswith('.bias') else 0.0, 1e-30)
        assert (pa.grad - pb.grad).abs().max().item() <= 3e-5 * scale, (na, (pa.grad - pb.grad).abs().max().item(), scale)
    for (na, ba), (nb, bb) in zip(fused.named_buffers(), plain.named_buffers()):
        assert torch.allclose(ba.float(), bb.float(), rtol=1e-6, atol=1e-7), na
    fused.eval(); plain.eval()
    with torch.no_grad():
        ya, _ = fused((feats, coords))
        try:
            type(be).has_devox_bnact = False
            yb, _ = plain((feats, coords))
        finally:
            type(be).has_devox_bnact = True
    assert rel(ya, yb) < 1e-5


@pytest.mark.parametrize('b,c,h,slices,affine', [(8, 64, 8, 8, True), (64, 1024, 128, 1, True), (3, 100, 12, 3, False), (1, 2048, 256, 2, True), (5, 16, 2, 1, True)])
def test_se_excitation_kernels_match_the_torch_chain(hip, b, c, h, slices, affine):
    """pvcnn_se_excite_fwd / _bwd (csrc/se.hip) vs the same algebra written with torch ops in fp64 (SE3d, reference modules/se.py:6-17,
    between the sums of BatchNormActSEDevoxelize), from the reduction pass's partial sums (C, B, slices, 2): 1e-5 of each tensor's
    largest entry; and run twice -> the same bits (sums in slice and cloud order)."""
    torch.manual_seed(5)
    dev = 'cuda:0'
    s3 = 4096
    part_a = torch.randn(c, b, slices, 2, device=dev) * (s3 / slices) ** 0.5
    part_p = torch.randn(c, b, slices, 2, device=dev) * (s3 / slices) ** 0.5
    gam = torch.randn(c, device=dev) if affine else None
    bet = torch.randn(c, device=dev) if affine else None
    w1 = torch.randn(h, c, device=dev) / c ** 0.5 * 8
    w2 = torch.randn(c, h, device=dev) / h ** 0.5
    a_sum, ax_sum, sq, hd, ex = hip.se_excite_forward(part_a, gam, bet, w1, w2, s3)
    out = hip.se_excite_backward(part_p, a_sum, ax_sum, gam, bet, sq, hd, ex, w1, w2, s3)
    again = hip.se_excite_backward(part_p, a_sum, ax_sum, gam, bet, sq, hd, ex, w1, w2, s3)
    for x, y in zip(out, again):
        assert torch.equal(x, y)
    d = torch.float64
    g = gam.to(d) if affine else torch.ones(c, device=dev, dtype=d)
    bt = bet.to(d) if affine else torch.zeros(c, device=dev, dtype=d)
    sa, sp = part_a.to(d).sum(2), part_p.to(d).sum(2)                      # (C, B, 2)
    A, AX, P, Q = sa[..., 0].t(), sa[..., 1].t(), sp[..., 0].t(), sp[..., 1].t()
    W1, W2 = w1.to(d), w2.to(d)
    sq_r = (g * AX + bt * A) / s3
    pre1 = sq_r @ W1.t()
    hd_r = torch.relu(pre1)
    ex_r = torch.sigmoid(hd_r @ W2.t())
    g_ex = g * Q + bt * P
    g_pre2 = g_ex * ex_r * (1 - ex_r)
    g_w2 = g_pre2.t() @ hd_r
    g_pre1 = (g_pre2 @ W2) * (hd.to(d) > 0)               # the kernel's own relu mask: a hidden unit at 0 +- 1 ulp may sit on either side
    g_w1 = g_pre1.t() @ sq_r
    g_mean = (g_pre1 @ W1) / s3
    sb = (ex_r * P + g_mean * A).sum(0)
    sg = (ex_r * Q + g_mean * AX).sum(0)

    def rel(x, y):
        return ((x.to(d) - y).abs().max() / y.abs().max().clamp_min(1e-300)).item()
    assert rel(a_sum, A) < 1e-6 and rel(ax_sum, AX) < 1e-6
    assert rel(sq, sq_r) < 1e-6 and rel(hd, hd_r) < 1e-5 and rel(ex, ex_r) < 1e-5
    for name, x, y in zip(('g_w1', 'g_w2', 'g_mean', 'sum_beta', 'sum_gamma'), out, (g_w1, g_w2, g_mean, sb, sg)):
        assert rel(x, y) < 1e-5, (name, rel(x, y))


@pytest.mark.parametrize('shape,p', [((16, 128, 4096), 0.3), ((3, 40, 1000), 0.5), ((2, 16, 260), 0.1), ((2, 8, 64, 32), 0.3)])
def test_dropout_fused_into_the_bn_relu_passes(hip, shape, p):
    """SharedMLP + nn.Dropout(p) of a classifier head (models/utils.py:15-36) as run_layers(..., tail_dropout=p): the Dropout rides on the
    BatchNorm + ReLU passes (csrc/bnact.hip).  With the kernel's own keep mask (pvcnn_dropout_keep_mask under the same seed) the
    forward output is bit-identical to dropout applied to the un-fused pair's output, every gradient agrees with autograd through
    `pair(x) * keep / (1 - p)` in fp64 to 1e-5, the mask keeps 1 - p of the elements (4 sigma), differs from call to call and repeats
    under the same torch seed."""
    from pvcnn_amd.modules.functional import bnact as B
    torch.manual_seed(3)
    dim2 = len(shape) == 4
    conv, norm = (nn.Conv2d, nn.BatchNorm2d) if dim2 else (nn.Conv1d, nn.BatchNorm1d)
    cin = 24
    net = nn.Sequential(conv(cin, shape[1], 1), norm(shape[1]), nn.ReLU(True)).to(DEV).train()
    x = torch.randn(shape[0], cin, *shape[2:], device=DEV)
    assert B.fused_dropout_ok(x)
    seeds = []
    real_randint = torch.randint

    def spy(*a, **k):
        t = real_randint(*a, **k)
        seeds.append(t)
        return t
    xa = x.clone().requires_grad_()
    torch.randint = spy
    try:
        torch.manual_seed(11)
        ya = B.run_layers(net, xa, tail_dropout=p)
        torch.manual_seed(11)
        yb = B.run_layers(net, x, tail_dropout=p)
        yc = B.run_layers(net, x, tail_dropout=p)
    finally:
        torch.randint = real_randint
    assert len(seeds) == 3 and torch.equal(seeds[0], seeds[1]) and not torch.equal(seeds[1], seeds[2])
    assert torch.equal(ya, yb) and not torch.equal(yb, yc)
    keep = hip.dropout_keep_mask(seeds[0], p, ya.numel()).view(ya.shape)
    n = keep.numel()
    assert abs(keep.float().mean().item() - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-5
    # per channel and per sample too (no stripes)
    assert (keep.float().mean(dim=tuple(i for i in range(keep.dim()) if i != 1)) - (1 - p)).abs().max().item() < 6 * (p * (1 - p) / (n / shape[1])) ** 0.5 + 1e-5
    with torch.no_grad():
        plain = B.run_layers(net, x)                                  # the un-fused pair (statistics of the same batch)
    want = torch.where(keep, plain * (1.0 / (1.0 - p)), torch.zeros_like(plain))
    assert torch.equal(ya.detach(), want)
    # gradients: autograd through the same function in fp64
    ref = nn.Sequential(conv(cin, shape[1], 1), norm(shape[1]), nn.ReLU()).to(DEV).double().train()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in net.state_dict().items()})
    xr = x.double().requires_grad_()
    g = torch.randn_like(ya)
    for q in net.parameters():
        q.grad = None
    ya.backward(g)
    (ref(xr) * keep.double() / (1.0 - p)).backward(g.double())
    assert _rel(xa.grad, xr.grad) < 2e-5
    for (na, pa), (nb, pb) in zip(net.named_parameters(), ref.named_parameters()):
        scale = max(pb.grad.abs().max().item(), dict(ref.named_parameters())[na.replace('.bias', '.weight')].grad.abs().max().item()
                    if na.endswith('.bias') else 0.0, 1e-30)
        assert (pa.grad.double() - pb.grad).abs().max().item() <= 3e-5 * scale, na


def test_classifier_head_with_fused_dropout_trains_like_the_module_stack(hip):
    """workload._classify (SharedMLP + Dropout pairs fused, last Conv1d on the package's GEMM) vs nn.Sequential.forward of the same
    head: identical in eval mode and with p = 0; in training mode the two draw different masks, so the check is statistical -- the
    mean and the second moment of the logits over 8 draws agree within their spread."""
    from pvcnn_amd import workload
    torch.manual_seed(5)
    layers, _ = workload._head(96, [64, 0.3, 32, 0.3, 13], 1, pointwise=True, classify=True)
    head = nn.Sequential(*layers).to(DEV)
    x = torch.randn(4, 96, 2048, device=DEV)
    head.eval()
    with torch.no_grad():
        assert (workload._classify(head, x) - head(x)).abs().max().item() < 1e-5 * head(x).abs().max().item()
    head.train()
    with torch.no_grad():
        fused = torch.stack([workload._classify(head, x) for _ in range(8)])
        plain = torch.stack([head(x) for _ in range(8)])
    for stat in (lambda t: t.mean(), lambda t: t.pow(2).mean()):
        a, b = stat(fused).item(), stat(plain).item()
        spread = max(abs(stat(plain[i]).item() - b) for i in range(8)) + abs(b) * 0.02
        assert abs(a - b) < 3 * spread, (a, b, spread)
    xa = x.clone().requires_grad_()
    workload._classify(head, xa).square().mean().backward()
    assert xa.grad is not None and torch.isfinite(xa.grad).all() and all(q.grad is not None and torch.isfinite(q.grad).all() for q in head.parameters())


@pytest.mark.parametrize('b,c,n', [(16, 1024, 4096), (2, 40, 256), (3, 64, 2048), (1, 8, 512)])
def test_apply_pass_emits_the_row_maxima_torch_max_returns(hip, b, c, n):
    """csrc/bnact.hip bnact_apply_pb_kernel<.., ROWMAX> + row_keys_decode (the global max-pool over the points behind the last
    SharedMLP, models/s3dis/pvcnn.py:37-43, riding on the BatchNorm + ReLU pass that writes the tensor): y and its amax buffer
    bit-identical to the plain pass, winners / values == y.max(dim=-1) on ReLU outputs -- rows that are all zero (a mix of -0 and
    +0: the first index), repeated maxima, and a NaN wins."""
    gen = torch.Generator().manual_seed(b * 1000 + c)
    x = torch.randn(b, c, n, generator=gen).mul(2).round().div(2).to(DEV)           # coarse values: plenty of exact ties
    x[0, 0] = -x[0, 0].abs() - 1.0                                                    # a row ReLU turns into -0 everywhere
    x[-1, -1, 7] = float('nan')
    gamma, beta = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.1
    gamma[0], beta[0] = 1.0, 0.0
    mean, rstd = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    seg = 256
    plain, _, _, amax_plain = hip.bnact_forward(x, gamma, beta, None, None, False, 0.1, 1e-5, 0.0, stats=(mean, rstd), amax_seg=seg)
    whole, amax, keys = hip.amax_and_row_keys(b, c, n, seg, x.device)
    whole.zero_()
    y, winners, values = hip.bnact_apply_rowmax(x, gamma, beta, mean, rstd, 0.0, seg, amax, keys)
    assert torch.equal(y.view(torch.int32), plain.view(torch.int32))
    assert torch.equal(amax, amax_plain)
    ref = y.max(dim=-1)
    nan_row = torch.isnan(ref.values)
    assert nan_row[-1, -1] and winners[-1, -1].item() == 7 and torch.isnan(values[-1, -1])
    assert torch.equal(winners[~nan_row], ref.indices[~nan_row])
    assert torch.equal(values[~nan_row].view(torch.int32), ref.values[~nan_row].view(torch.int32))
    assert winners[0, 0].item() == 0 and values[0, 0].item() == 0.0


def test_last_point_stage_hands_its_row_maxima_to_the_max_pool(hip):
    """workload.PVCNN's last SharedMLP under emit_row_max: tap_and_pool takes (winners, values) from the tensor instead of reading it
    (counted), and the pooled features and every gradient equal the path that reduces the tensor itself."""
    from pvcnn_amd import workload
    from pvcnn_amd.modules import SharedMLP
    from pvcnn_amd.modules.functional.bnact import emit_row_max
    torch.manual_seed(3)
    mlp = SharedMLP(32, 96).to(DEV).train()
    x = torch.randn(4, 32, 1024, device=DEV)
    outs = []
    for fused in (True, False):
        xa = x.clone().requires_grad_()
        mlp.zero_grad()
        with (emit_row_max(workload._last_norm(mlp)) if fused else __import__('contextlib').nullcontext()):
            y = mlp(xa)
        assert (getattr(y, '_pvcnn_row_max', None) is not None) == fused
        tap, pooled = workload.tap_and_pool(y)
        (tap.square().mean() + pooled.square().sum()).backward()
        outs.append((pooled.detach().clone(), xa.grad.clone(), [p.grad.clone() for p in mlp.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert all(torch.equal(a, b) for a, b in zip(outs[0][2], outs[1][2]))


def test_row_maxima_handed_over_by_the_batchnorm_pass_die_with_an_in_place_update(hip):
    """the (winners, values) a BatchNorm + ReLU pass attaches to its output are keyed by the tensor's in-place version.
    Two lines of defence: autograd itself refuses to use the output of the fused node after an in-place update (it is a view made
    inside a custom Function: RuntimeError, measured in round 5), and where a version moved without that error -- emulated here by a
    tag from an older version -- tap_and_pool reduces the tensor as it is instead of trusting the tag."""
    from pvcnn_amd import workload
    from pvcnn_amd.modules import SharedMLP
    from pvcnn_amd.modules.functional.bnact import emit_row_max
    torch.manual_seed(4)
    mlp = SharedMLP(16, 64).to(DEV).train()
    x = torch.randn(2, 16, 512, device=DEV).requires_grad_()
    with emit_row_max(workload._last_norm(mlp)):
        y = mlp(x)
    winners, values, version = y._pvcnn_row_max
    assert version == y._version
    _, pooled = workload.tap_and_pool(y)
    assert torch.equal(pooled, y.max(dim=-1).values)
    # a stale tag (older version, wrong contents): ignored
    y._pvcnn_row_max = (torch.zeros_like(winners), torch.full_like(values, -1.0), version - 1)
    _, pooled = workload.tap_and_pool(y)
    assert torch.equal(pooled, y.max(dim=-1).values)
    # ... and the in-place update itself is refused by autograd for this tensor
    y2 = mlp(x)
    y2.detach().add_(1.0)
    with pytest.raises(RuntimeError, match='modified inplace'):
        workload.tap_and_pool(y2)
