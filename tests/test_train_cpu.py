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
