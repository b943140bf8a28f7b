"""ABI v12: amax buffers WITHOUT their word [0] (PVCNN_TABLE_ONLY: the producer launches nothing for the global maximum) and the two
backward-weight entries that take the global maximum from the table instead (x_amax_seg / gy_amax_seg > 0).  Same bits as the route
through word [0]; a poisoned word [0] is never read."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def hip():
    from pvcnn_amd.modules.functional.backend import HipBackend
    return HipBackend()


def _poisoned(buf, bits):
    out = buf.clone()
    out[0] = bits
    return out


@pytest.mark.parametrize('shape,seg', [((3, 5, 1000), 256), ((16, 64, 4096), 256), ((2, 7, 12 ** 3), 12), ((4, 9, 32 ** 3), 32)])
def test_the_table_of_a_table_only_buffer_is_the_table(hip, shape, seg):
    x = torch.randn(*shape, device=DEV) * torch.pow(10.0, torch.randint(-4, 4, (shape[0], 1, shape[2]), device=DEV).float())
    whole, table_only = hip.absmax_tiles(x, seg), hip.absmax_tiles(x, seg, want_global=False)
    assert torch.equal(whole[1:], table_only[1:]) and whole[0] == whole[1:].max()


@pytest.mark.parametrize('b,ci,co,r', [(2, 16, 24, 8), (3, 33, 40, 12), (4, 64, 64, 16), (2, 9, 64, 16), (2, 64, 64, 32), (1, 9, 32, 32)])
def test_conv3d_backward_weight_takes_the_maxima_from_the_tables(hip, b, ci, co, r):
    g = torch.Generator(device=DEV).manual_seed(b * 100 + r)
    x = torch.randn(b, ci, r, r, r, device=DEV, generator=g) * 37.0
    gy = torch.randn(b, co, r, r, r, device=DEV, generator=g) * 1e-3
    xa, ga = hip.conv_amax(x), hip.conv_amax(gy)
    by_word = hip.conv3d_backward_weight_f16(x, gy, xa[:1].clone(), ga[:1].clone(), with_bias=True)
    for poison in (0, 0x7f7fffff, 0x7fc00000):              # zero, the largest float, a NaN: word [0] must not matter
        by_table = hip.conv3d_backward_weight_f16(x, gy, _poisoned(xa, poison), _poisoned(ga, poison), with_bias=True)
        assert torch.equal(by_word[0], by_table[0]) and torch.equal(by_word[1], by_table[1]), (poison,)
    mixed = hip.conv3d_backward_weight_f16(x, gy, _poisoned(xa, 0), ga[:1].clone(), with_bias=True)      # a table and a 1-word buffer
    assert torch.equal(by_word[0], mixed[0])
    w = torch.zeros(co, ci, 3, 3, 3, dtype=torch.float64, device=DEV, requires_grad=True)
    torch.nn.functional.conv3d(x.double(), w, padding=1).backward(gy.double())
    assert ((by_word[0].double() - w.grad).abs().max() / w.grad.abs().max()).item() < 1e-5


@pytest.mark.parametrize('b,k,m,n', [(2, 64, 64, 1024), (16, 128, 1024, 4096), (3, 1472, 512, 1000), (16, 512, 256, 4096), (1, 40, 72, 260)])
def test_pwconv_backward_weight_takes_the_maxima_from_the_tables(hip, b, k, m, n):
    g = torch.Generator(device=DEV).manual_seed(n + k)
    x = torch.randn(b, k, n, device=DEV, generator=g) * 5.0
    gy = torch.randn(b, m, n, device=DEV, generator=g) * 1e-2
    xa, ga = hip.pw_amax(x), hip.pw_amax(gy)
    by_word = hip.pwconv_backward_weight_f16(x, gy, xa[:1].clone(), ga[:1].clone(), with_bias=True)
    for poison in (0, 0x7f7fffff):
        by_table = hip.pwconv_backward_weight_f16(x, gy, _poisoned(xa, poison), _poisoned(ga, poison), with_bias=True)
        assert torch.equal(by_word[0], by_table[0]) and torch.equal(by_word[1], by_table[1]), (poison,)
    ref = torch.einsum('bmn,bkn->mk', gy.double(), x.double())
    assert ((by_word[0].double() - ref).abs().max() / ref.abs().max()).item() < 1e-5


def test_a_training_step_never_reads_word_zero(hip):
    """The f16x2 autograd functions make their own amax buffers table-only: gradients equal to the ones with complete buffers."""
    from pvcnn_amd.modules.functional.conv3d import voxel_conv3d
    from pvcnn_amd.modules.functional.pwconv import pointwise_conv
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(2, 32, 16, 16, 16, device=DEV, generator=g, requires_grad=True)
    w = (torch.randn(48, 32, 3, 3, 3, device=DEV, generator=g) * 0.05).requires_grad_()
    y = voxel_conv3d(x, w, None, False, 2)
    gy = torch.randn(y.shape, device=DEV, generator=g)
    y.backward(gy)
    gw = hip.conv3d_backward_weight_f16(x.detach(), gy, hip.conv_amax(x.detach())[:1].clone(), hip.conv_amax(gy)[:1].clone())
    assert torch.equal(w.grad, gw)
    p = torch.randn(16, 512, 4096, device=DEV, generator=g, requires_grad=True)        # (large enough for the f16x2 backward-weight route)
    v = (torch.randn(256, 512, device=DEV, generator=g) * 0.05).requires_grad_()
    q = pointwise_conv(p, v.view(256, 512, 1), None, False, 2)
    gq = torch.randn(q.shape, device=DEV, generator=g)
    q.backward(gq)
    gv = hip.pwconv_backward_weight_f16(p.detach(), gq, hip.pw_amax(p.detach())[:1].clone(), hip.pw_amax(gq)[:1].clone())
    assert torch.equal(v.grad.view(256, 512), gv)
