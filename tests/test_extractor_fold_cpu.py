"""Host logic of the extractor: the eval-mode BatchNorm fold and its cache (reart_amd/networks/feature_extractor.py)."""
import numpy as np
import torch
import torch.nn as nn


def _reference_fold(conv, bn):
    w = conv.weight.detach().reshape(conv.weight.shape[0], -1).double()
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    return (w * scale[:, None]).t().numpy(), ((conv.bias.detach().double() - bn.running_mean.double()) * scale + bn.bias.detach().double()).numpy()


def test_fold_equals_conv_followed_by_eval_batchnorm_and_follows_parameter_changes():
    from reart_amd.networks.feature_extractor import _fold

    torch.manual_seed(0)
    conv, bn = nn.Conv2d(7, 5, 1), nn.BatchNorm2d(5)
    with torch.no_grad():
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.normal_(); bn.bias.normal_()
    bn.eval()
    x = torch.randn(3, 7, 4, 2)
    Wt, b = _fold(conv, bn)
    y = bn(conv(x)).detach()
    np.testing.assert_allclose(torch.einsum("bchw,cd->bdhw", x, Wt).numpy() + b.numpy()[None, :, None, None], y.numpy(), rtol=1e-5, atol=1e-5)
    # cached: the same tensors come back while nothing changed
    Wt2, b2 = _fold(conv, bn)
    assert Wt2 is Wt and b2 is b
    # an in-place update of any source tensor refolds
    with torch.no_grad():
        bn.running_var.mul_(3.0)
    Wt3, b3 = _fold(conv, bn)
    assert Wt3 is not Wt
    rw, rb = _reference_fold(conv, bn)
    np.testing.assert_allclose(Wt3.numpy(), rw, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b3.numpy(), rb, rtol=1e-6, atol=1e-6)
    # so does load_state_dict
    sd = {k: v.clone() for k, v in conv.state_dict().items()}
    sd["weight"] = sd["weight"] * 2.0
    conv.load_state_dict(sd)
    Wt4, _ = _fold(conv, bn)
    np.testing.assert_allclose(Wt4.numpy(), 2.0 * Wt3.numpy(), rtol=1e-6, atol=1e-7)
    # and a re-assigned parameter
    conv.weight = nn.Parameter(conv.weight.detach() * 0.5)
    Wt5, _ = _fold(conv, bn)
    np.testing.assert_allclose(Wt5.numpy(), Wt3.numpy(), rtol=1e-6, atol=1e-7)


def test_fold_cache_is_dropped_by_the_module_level_hooks():
    """Writes through .data bypass the version counter: invalidate_fold() -- which .train(), .to() and
    load_state_dict() call themselves -- drops the cached folds."""
    from reart_amd.networks.feature_extractor import PointNet2Msg2, _fold

    torch.manual_seed(1)
    net = PointNet2Msg2(8).eval()
    Wt, _ = _fold(net.conv1, net.bn1)
    assert _fold(net.conv1, net.bn1)[0] is Wt
    net.conv1.weight.data.mul_(2.0)                      # invisible to the key ...
    assert _fold(net.conv1, net.bn1)[0] is Wt
    net.invalidate_fold()                                # ... until the cache is dropped
    Wt2, _ = _fold(net.conv1, net.bn1)
    np.testing.assert_allclose(Wt2.numpy(), 2.0 * Wt.numpy(), rtol=1e-6, atol=1e-7)
    for action in (lambda: net.train() and net.eval(), lambda: net.double().float(), lambda: net.load_state_dict(net.state_dict())):
        _fold(net.conv1, net.bn1)
        assert hasattr(net.conv1, "_reart_folded")
        action()
        assert not hasattr(net.conv1, "_reart_folded")
    # the entry keeps its source tensors alive: a freed parameter's address cannot be handed to a new one under the key
    assert _fold(net.conv1, net.bn1) and net.conv1._reart_folded[3][0] is net.conv1.weight


def test_racer_counts_fit_the_chip():
    """reart_amd.utils.lap._racers: (cold, all) workgroups per matrix of a raced assignment solve -- never more than 256
    workgroups below 4096 columns, five cold racers at 4096, three warm ones when a previous solution is available."""
    from reart_amd.utils.lap import _racers, RACERS

    assert _racers(19, 4096, False) == (RACERS, RACERS) and _racers(19, 4096, True) == (RACERS, RACERS + 3)
    for B in (1, 9, 19, 40, 100):
        for n in (64, 1024, 2048):
            cold, total = _racers(B, n, False)
            assert cold == total and RACERS <= total <= 12 and (total == RACERS or B * total <= 256)
            cold, total = _racers(B, n, True)
            assert total - cold == 3 and cold >= 2 and total <= 15 and (total == RACERS or B * total <= 256)
