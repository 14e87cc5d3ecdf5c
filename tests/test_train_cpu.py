"""Host-side pieces of the training step that need no GPU."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'ml-pointconvformer_amd'))


def test_clip_grad_norm_matches_torch():
    """pcf_train.clip_grad_norm_ (over the optimizer's parameter lists) == torch.nn.utils.clip_grad_norm_ over
    model.parameters() (train_ScanNet_DDP_WarmUP.py:421): same total norm, same scaled gradients, both when the
    norm exceeds the bound and when it does not; parameters without a gradient are skipped."""
    import pcf_train
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    for max_norm, scale in ((10.0, 100.0), (10.0, 1e-3)):
        for p in net.parameters():
            p.grad = torch.randn_like(p) * scale
        net[3].bias.grad = None
        ref = copy.deepcopy(net)
        for p, q in zip(net.parameters(), ref.parameters()):
            q.grad = None if p.grad is None else p.grad.clone()
        opt = torch.optim.AdamW(net.parameters(), lr=0.1)
        got = pcf_train.clip_grad_norm_(opt, max_norm)
        want = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm)
        torch.testing.assert_close(got, want)
        for p, q in zip(net.parameters(), ref.parameters()):
            if q.grad is None:
                assert p.grad is None
            else:
                torch.testing.assert_close(p.grad, q.grad)


def test_make_optimizer_follows_the_yaml_keys():
    import pcf_train

    class Cfg:
        learning_rate, adamw_decay = 0.02, 0.05
    net = torch.nn.Linear(4, 4)
    opt = pcf_train.make_optimizer(Cfg, net)
    g = opt.param_groups[0]
    assert isinstance(opt, torch.optim.AdamW) and g['lr'] == 0.02 and g['weight_decay'] == 0.05
    assert not g['fused']                 # CPU parameters: the for-each form


def test_training_state_snapshot_restores_a_step():
    """pcf_train._training_state lists every tensor an optimisation step changes in place (parameters, BatchNorm running
    statistics and counters, optimizer moments and step counters): cloning them, stepping and copying the clones back
    returns model and optimizer to the exact state before the step -- what GraphedTrainingStep / DataParallelStep rely on
    to keep a capture's warm-up iterations out of training."""
    import pcf_train
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.BatchNorm1d(8), torch.nn.ReLU(), torch.nn.Linear(8, 3)).train()
    opt = torch.optim.AdamW(net.parameters(), lr=0.05, weight_decay=0.05)

    def step(seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(16, 6, generator=g)
        net(x).square().mean().backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    step(1)                                           # optimizer state exists from here on
    state = pcf_train._training_state(net, opt)
    n_expected = len(list(net.parameters())) * 4 + len(list(net.buffers()))          # p, exp_avg, exp_avg_sq, step + buffers
    assert len(state) == n_expected
    saved = [t.clone() for t in state]
    ref = copy.deepcopy(net)
    ref_opt = copy.deepcopy(opt.state_dict())
    step(2)
    assert any(not torch.equal(a, b) for a, b in zip(state, saved))
    with torch.no_grad():
        for t, v in zip(state, saved):
            t.copy_(v)
    for a, b in zip(net.state_dict().values(), ref.state_dict().values()):
        assert torch.equal(a, b)
    now = opt.state_dict()['state']
    for k, st in ref_opt['state'].items():
        for name, v in st.items():
            assert torch.equal(now[k][name], v), (k, name)
    # and the same step from the restored state reproduces the same result as from a deep copy
    step(3)
    opt2 = torch.optim.AdamW(ref.parameters(), lr=0.05, weight_decay=0.05)
    opt2.load_state_dict(ref_opt)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(16, 6, generator=g)
    ref(x).square().mean().backward()
    opt2.step()
    for a, b in zip(net.state_dict().values(), ref.state_dict().values()):
        torch.testing.assert_close(a, b, rtol=0, atol=0)
