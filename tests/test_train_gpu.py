"""GPU: whole training iterations (post-kNN + inverse CSR + forward + loss + backward + clip + AdamW,
train_ScanNet_DDP_WarmUP.py:376-424) of the BASELINE model configs at their scene sizes, held to size-independent
properties: finite decreasing-or-stable loss, the fused edge graph against the layer-at-a-time kernels (loss equal to
1e-4, gradient direction equal: the 29-layer backward amplifies fp32 rounding to 1-10 % per tensor on ANY path, see
tests/golden/make_golden_models.py), and the kNN / CSR tables of the batch bit-exact against the C oracle."""
import numpy as np
import pytest
import torch

from oracle import knn_c

pytestmark = pytest.mark.gpu


def _batch(cfg, device, scenes, points, seed):
    import pcf_train
    sc = [pcf_train.synthetic_scene(points, cfg.grid_size, seed=seed + i, device=device) for i in range(scenes)]
    return pcf_train.pack_batch(sc, cfg.grid_size)


def _grads(net, cfg, batch, edges, crit):
    features, pointclouds, target, norms, _ = batch
    es, ef, ep, inv = edges
    net.zero_grad(set_to_none=True)
    pred = net(features, pointclouds, es, ef, ep, norms, *inv)
    loss = crit(pred.reshape(-1, cfg.num_classes), target)
    loss.backward()
    return float(loss.detach()), torch.cat([p.grad.reshape(-1) for p in net.parameters()]).double()


@pytest.mark.parametrize('name,scenes,points', [('configPCF_10cm_lite', 4, 40000), ('configPCF_5cm', 1, 150000),
                                                ('configPCF_10cm', 2, 40000), ('configPCF_2cm_PTF2', 1, 60000)])
def test_training_iteration(device, name, scenes, points):
    import pcf_model
    import pcf_train
    cfg = pcf_train.baseline_config(name)
    torch.manual_seed(1)
    net = pcf_model.PointConvFormer_Segmentation(cfg).to(device).train()
    opt = pcf_train.make_optimizer(cfg, net)
    crit = torch.nn.CrossEntropyLoss(ignore_index=cfg.ignore_label, label_smoothing=cfg.label_smoothing).to(device)
    batch = _batch(cfg, device, scenes, points, seed=4000)
    features, pointclouds, target, norms, stored = batch
    n0 = sum(stored[0])
    assert abs(n0 - scenes * points) < 0.15 * scenes * points and len(pointclouds) == 5
    edges = pcf_train.build_edges(cfg, pointclouds, stored)

    # ---- kNN and CSR tables of this batch against the C oracle: coarse levels whole, level 0 on a sub-sample ----
    es, ef, ep, inv = edges
    offs = [np.concatenate([[0], np.cumsum(c)]).astype(np.int32) for c in stored]
    for l in (2, 3, 4):
        pts = pointclouds[l][0].cpu().numpy()
        assert np.array_equal(es[l][0].cpu().numpy(), knn_c.knn_packed(pts, pts, offs[l], offs[l], 16)), f'self kNN level {l}'
        fine = pointclouds[l - 1][0].cpu().numpy()
        assert np.array_equal(ef[l - 1][0].cpu().numpy(), knn_c.knn_packed(fine, pts, offs[l - 1], offs[l], 16)), f'forward kNN {l}'
        assert np.array_equal(ep[l - 1][0].cpu().numpy(), knn_c.knn_packed(pts, fine, offs[l], offs[l - 1], 16)), f'propagate kNN {l}'
        want = knn_c.knn_inverse(es[l][0].cpu().numpy(), pts.shape[0])
        for got, w in zip((inv[0][0][l], inv[0][1][l], inv[0][2][l]), want):
            assert np.array_equal(got.cpu().numpy().reshape(-1)[:w.size], w), f'CSR level {l}'
    p0 = pointclouds[0][0].cpu().numpy()
    a, b = int(offs[0][0]), int(offs[0][1])                     # first scene; 2000 of its points as queries
    rows = np.random.default_rng(0).choice(b - a, 2000, replace=False) + a
    sub = knn_c.knn_packed(p0[a:b], p0[rows], np.array([0, b - a], np.int32), np.array([0, 2000], np.int32), 16) + a
    assert np.array_equal(es[0][0].cpu().numpy()[rows], sub), 'self kNN level 0 (sub-sample)'

    # ---- two optimisation steps: finite losses, parameters move, BatchNorm counters advance ----
    before = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    l0 = float(pcf_train.training_iteration(net, opt, crit, cfg, batch, edges))
    l1 = float(pcf_train.training_iteration(net, opt, crit, cfg, batch, edges))
    assert np.isfinite(l0) and np.isfinite(l1) and l1 < l0 + 0.5, (l0, l1)
    after = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.isfinite(after).all() and float((after - before).abs().max()) > 0
    nbt = [int(b) for n, b in net.named_buffers() if n.endswith('num_batches_tracked')]
    assert min(nbt) >= 2

    # ---- fused edge graph versus layer-at-a-time kernels on the same weights and batch ----
    if cfg.drop_path_rate > 0:
        for m in net.modules():           # freeze the stochastic depth draw so that both paths see the same graph
            if hasattr(m, 'drop_path') and hasattr(m.drop_path, 'drop_prob'):
                m.drop_path.draw = (lambda x: x.new_full((1, 1, 1), 1.0 / (1.0 - cfg.drop_path_rate)))
    loss_f, g_f = _grads(net, cfg, batch, edges, crit)
    assert _grads(net, cfg, batch, edges, crit)[0] == loss_f, 'the forward pass (no float atomics) must repeat bit for bit'
    cfg.NO_EDGE_CHAIN = True
    for m in net.modules():
        if hasattr(m, 'no_chain') or m.__class__.__name__ == 'WeightNet':
            m.no_chain = True
    loss_l, g_l = _grads(net, cfg, batch, edges, crit)
    cfg.NO_EDGE_CHAIN = False
    assert abs(loss_f - loss_l) < 1e-4 * max(1.0, abs(loss_l)), (loss_f, loss_l)
    cos = float((g_f @ g_l) / (g_f.norm() * g_l.norm()))
    assert cos > 0.99, cos


def test_fused_adamw_with_clipping_equals_torch(device):
    """pcf_optim.FusedAdamW.step(max_grad_norm) against clip_grad_norm_ + torch.optim.AdamW (train_ScanNet_DDP_WarmUP.py:237-241,
    :421) over 150 tensors (three argument lists), one of them larger than a chunk, for six steps with a learning-rate change:
    parameters, both moments, clipped gradients and the reported norm; then without clipping; state_dicts interchange."""
    import pcf_optim
    g = torch.Generator().manual_seed(3)
    shapes = [(70000,), (300, 40)] + [(int(torch.randint(1, 40, (1,), generator=g)), int(torch.randint(1, 30, (1,), generator=g)))
                                      for _ in range(148)]
    base = [torch.randn(*s, generator=g) for s in shapes]
    mine = [torch.nn.Parameter(t.clone().to(device)) for t in base]
    ref = [torch.nn.Parameter(t.clone().to(device)) for t in base]
    opt = pcf_optim.FusedAdamW(mine, lr=0.02, weight_decay=0.05)
    want = torch.optim.AdamW(ref, lr=0.02, weight_decay=0.05)
    for it in range(6):
        scale = 10.0 if it % 2 == 0 else 1e-3                # clipped and unclipped steps
        for a, b in zip(mine, ref):
            gr = torch.randn(a.shape, generator=g).to(device) * scale
            a.grad, b.grad = gr.clone(), gr.clone()
        if it == 3:
            for o in (opt, want):
                o.param_groups[0]['lr'] = 0.005
        clip = 10 if it < 5 else None
        norm = torch.nn.utils.clip_grad_norm_(ref, clip) if clip else None
        want.step()
        opt.step(max_grad_norm=clip)
        if clip:
            torch.testing.assert_close(opt.last_grad_norm, norm, rtol=1e-5, atol=0)
        for a, b in zip(mine, ref):
            torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-7)
            torch.testing.assert_close(a.grad, b.grad, rtol=2e-6, atol=1e-9)
            torch.testing.assert_close(opt.state[a]['exp_avg'], want.state[b]['exp_avg'], rtol=2e-6, atol=1e-8)
            torch.testing.assert_close(opt.state[a]['exp_avg_sq'], want.state[b]['exp_avg_sq'], rtol=2e-6, atol=1e-12)
        assert float(opt.state[mine[0]]['step']) == it + 1
    sd = opt.state_dict()
    again = pcf_optim.FusedAdamW(mine, lr=0.02, weight_decay=0.05)
    again.load_state_dict(sd)
    torch.optim.AdamW(ref, lr=0.02, weight_decay=0.05).load_state_dict(sd)          # torch accepts the same layout
    for a in mine:
        a.grad = torch.ones_like(a)
    again.step()
    assert float(again.state[mine[0]]['step']) == 7


@pytest.mark.parametrize('R,C,smoothing,ignored', [(144157, 20, 0.1, 0.2), (1000, 13, 0.0, 0.0), (257, 64, 0.3, 0.9), (5, 3, 0.0, 0.4)])
def test_fused_cross_entropy_equals_torch(device, R, C, smoothing, ignored):
    """pcf_fused.cross_entropy against nn.CrossEntropyLoss(ignore_index, label_smoothing) (train_ScanNet_DDP_WarmUP.py:243,
    :404): loss and logit gradient (scaled by an upstream factor), rows with the ignore label contribute nothing."""
    import pcf_fused
    g = torch.Generator().manual_seed(R + C)
    logits = (torch.randn(R, C, generator=g) * 3).to(device)
    target = torch.randint(0, C, (R,), generator=g)
    target[torch.rand(R, generator=g) < ignored] = -100
    target[0] = 1                                                # at least one valid row
    target = target.to(device)
    crit = torch.nn.CrossEntropyLoss(ignore_index=-100, label_smoothing=smoothing).to(device)
    a = logits.clone().requires_grad_(True)
    b = logits.clone().requires_grad_(True)
    assert pcf_fused.cross_entropy_supported(crit, a)
    la = pcf_fused.cross_entropy(a, target, -100, smoothing)
    lb = crit(b, target)
    (2.5 * la).backward()
    (2.5 * lb).backward()
    torch.testing.assert_close(la, lb, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(a.grad, b.grad, rtol=1e-5, atol=1e-8)
    assert float(a.grad[target == -100].abs().sum()) == 0.0
    again = pcf_fused.cross_entropy(logits, target, -100, smoothing)
    assert torch.equal(again, la.detach())                       # fixed summation order
    none_valid = pcf_fused.cross_entropy(logits[:4], torch.full((4,), -100, dtype=torch.int64, device=device), -100, smoothing)
    assert torch.isnan(none_valid)                               # torch: mean over zero rows
