"""GPU: a captured training iteration replays like the eager one (train_ScanNet_DDP_WarmUP.py:376-424 as
pcf_train.training_iteration / GraphedTrainingStep run it), for every BASELINE model YAML.

Round 2's captured configPCF_2cm_PTF2 iteration died on replay with a GPU memory fault.  Cause (DESIGN.md "graph replay
fault"): the kNN cell histogram and the CSR counters were cleared with hipMemsetAsync, which a capture turns into memset
NODES; on a relaunch the dependent counting sort ran on an uncleared histogram and scattered rows through garbage offsets.
The library now clears with its own kernels, so a captured iteration holds kernel nodes only -- asserted here -- and the
counting sorts bound their positions.

The replays run in a child process (tests/graph_replay_child.py): should a fault come back, it ends the child, this test
fails with the child's output, and the rest of the suite has already run (the file sorts last on purpose)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run_child(cases, timeout):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    cmd = [sys.executable, os.path.join(ROOT, 'tests', 'graph_replay_child.py'), '--cases', cases]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    recs = []
    for line in res.stdout.splitlines():
        if line.startswith('{'):
            recs.append(json.loads(line))
    tail = '\n'.join(res.stdout.splitlines()[-6:]) + '\n--- stderr ---\n' + res.stderr[-1500:]
    assert res.returncode == 0, f'child exited with {res.returncode} (negative / 134 = aborted, e.g. by a GPU fault):\n{tail}'
    assert recs and recs[-1].get('done'), tail
    if cases == 'edge':
        return [r for r in recs if 'edge_case' in r]
    tables = [r for r in recs if 'tables_case' in r]
    assert len(tables) == 2, tail
    for t in tables:          # neighbour tables and CSR of every replay bit-identical to eagerly built ones
        assert t['n_wrong'] == 0 and t['compared'] >= 6 * 22, (t['tables_case'], t['wrong'])
    return [r for r in recs if 'case' in r]


def _check(rec):
    tag = f"{rec['case']} {rec['scenes']}x{rec['points']}"
    calls = len(rec['eager_losses'])
    assert rec['finite'], tag
    # a captured iteration is made of kernel nodes only: no memset / memcpy nodes (see the module docstring)
    for kinds in rec['node_kinds']:
        assert set(kinds) == {'kernel'} and kinds['kernel'] > 500, (tag, kinds)
    # every call is exactly one optimisation step: BatchNorm counters and the optimizer's device-side counter
    assert rec['num_batches_tracked'][0] == rec['num_batches_tracked'][1], (tag, rec['num_batches_tracked'])
    assert rec['num_batches_tracked'][1][0] >= calls, (tag, rec['num_batches_tracked'])
    assert rec['optimizer_steps'] == [float(calls), float(calls)], (tag, rec['optimizer_steps'])
    if not rec['frozen_draws']:
        assert all(l == l and abs(l) < 1e3 for l in rec['graph_losses']), (tag, rec['graph_losses'])
        return
    # the forward pass has no float atomics: from equal parameters the losses agree to rounding (first call: both eager);
    # later calls start from parameters that differ by the backward's atomics noise through AdamW
    e, g = rec['eager_losses'], rec['graph_losses']
    assert abs(e[0] - g[0]) <= 1e-5 * max(1.0, abs(e[0])), (tag, e, g)
    for a, b in zip(e, g):
        assert abs(a - b) <= 5e-3 * max(1.0, abs(a)), (tag, e, g)
    # AdamW normalises every element's update to ~lr, including elements whose gradient is rounding noise (a bias in front
    # of a batch-statistics BatchNorm): those move at random on any path, the rest together
    assert rec['update_cosine'] > 0.9, (tag, rec['update_cosine'])


@pytest.mark.timeout(1500)
def test_replayed_iteration_equals_eager_iteration_all_configs():
    """Six calls on a rotating pool of two packed batches per model YAML at small scene sizes (where every tensor of the
    iteration shares allocator segments with its neighbours, and where the round-2 fault reproduced for every config),
    then configPCF_2cm_PTF2 with real stochastic-depth draws and at its own size, 2 x 120k points."""
    recs = _run_child('all', 1400)
    assert len(recs) == 6, [r['case'] for r in recs]
    for rec in recs:
        _check(rec)


def test_headline_workload_against_oracle_at_its_own_size(device):
    """BASELINE.json's headline workload itself -- PCFLayer(64 -> 64, 8 heads, C_mid 16, VI + BatchNorm, train mode), one
    cloud of N = 80 000 points, K = 16 (layers.py:306-416) -- through the fused HIP path against
    oracle/pcf_oracle.py:pcf_layer on the same tensors: output, feature gradient and every parameter gradient, random
    upstream gradient.  Tolerance 1e-3 of each tensor's largest entry (north_star's bar); analytically zero parameter
    gradients (a bias in front of a batch-statistics BatchNorm; the guidance shift q - key cancels) hold rounding noise on
    both sides and get an absolute bound of 1e-4 of the largest parameter gradient; the feature gradient is held row by
    row with an allowance of 0.2 % of the rows for ReLU / LeakyReLU masks that the 3e-7 difference between two correct
    forwards flips (11 of 80 000 rows between this build's own two forward engines)."""
    import pcf_cuda
    import pcf_layers
    from oracle import pcf_oracle as O
    N, K = 80000, 16
    g = torch.Generator().manual_seed(1)
    xyz = torch.rand(1, N, 3, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(1, N, 3, generator=g), dim=-1)
    feats = torch.randn(1, N, 64, generator=g)
    up = torch.randn(1, N, 64, generator=g)
    off = torch.tensor([0, N], dtype=torch.int32, device=device)
    idx = pcf_cuda.knn_packed(xyz[0].to(device), xyz[0].to(device), off, off, K)[None].contiguous()

    class Cfg(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    cfg = Cfg(attention_type='subtraction', BATCH_NORM=True, drop_path_rate=0., dropout_rate=0., USE_VI=True, USE_PE=True,
              PCONV_OPT=True, USE_CUDA_KERNEL=True, layer_norm_guidance=False)
    torch.manual_seed(7)
    layer = pcf_layers.PCFLayer(64, 64, cfg, weightnet=[12, 16], num_heads=8, guidance_feat_len=32).train()
    with torch.no_grad():
        for m in layer.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in layer.state_dict().items()}
    # ---- oracle (CPU) ----
    xo = feats.clone().requires_grad_(True)
    want, _ = O.pcf_layer(O.Params(sd, '', True), xyz, xo, idx.cpu(), nrm, num_heads=8)
    want.backward(up)
    # ---- HIP ----
    layer = layer.to(device)
    x = feats.to(device).requires_grad_(True)
    out, _ = layer(xyz.to(device), x, idx, nrm.to(device))
    out.backward(up.to(device))
    torch.cuda.synchronize()

    scale = float(want.abs().max())
    assert float((out.detach().cpu() - want.detach()).abs().max()) <= 1e-3 * scale, 'output'
    top = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    n_checked = 0
    for name, p in layer.named_parameters():
        ref = sd[name].grad
        assert ref is not None and p.grad is not None, name
        bound = 1e-3 * float(ref.abs().max()) + 1e-4 * top
        err = float((p.grad.cpu() - ref).abs().max())
        assert err <= bound, f'{name}: |diff| {err:.3e} > {bound:.3e} (largest entry {float(ref.abs().max()):.3e}, top {top:.3e})'
        n_checked += 1
    assert n_checked >= 24
    gx, rx = x.grad.cpu()[0], xo.grad[0]
    row_err = (gx - rx).abs().amax(-1)
    bad = int((row_err > 1e-3 * float(rx.abs().max())).sum())
    assert bad <= N // 500, f'feature gradient: {bad} of {N} rows beyond 1e-3 of the largest entry'
    assert float(row_err.median()) <= 1e-5 * float(rx.abs().max())


def test_fused_adamw_two_groups_one_global_clip(device):
    """pcf_optim.FusedAdamW with two parameter groups (a no-weight-decay group with its own learning rate, as fine-tuning
    scripts build them) and step(max_grad_norm): ONE 2-norm over the gradients of both groups, as
    clip_grad_norm_(model.parameters(), 10) followed by torch.optim.AdamW.step() (train_ScanNet_DDP_WarmUP.py:237-241, :421)
    -- parameters, clipped gradients, moments and the reported norm over four steps, clipped and unclipped."""
    import pcf_optim
    g = torch.Generator().manual_seed(9)
    shapes_a = [(5000,), (64, 33), (7,)] + [(11, 5)] * 20
    shapes_b = [(129,), (3, 3)] + [(17,)] * 10
    base_a = [torch.randn(*s, generator=g) for s in shapes_a]
    base_b = [torch.randn(*s, generator=g) for s in shapes_b]
    mk = lambda ts: [torch.nn.Parameter(t.clone().to(device)) for t in ts]
    ma, mb, ra, rb = mk(base_a), mk(base_b), mk(base_a), mk(base_b)
    groups = lambda a, b: [dict(params=a, lr=0.02, weight_decay=0.05), dict(params=b, lr=0.004, weight_decay=0.0)]
    opt = pcf_optim.FusedAdamW(groups(ma, mb), lr=0.02, weight_decay=0.05)
    want = torch.optim.AdamW(groups(ra, rb), lr=0.02, weight_decay=0.05)
    for it in range(4):
        scale = 10.0 if it % 2 == 0 else 1e-3                # clipped and unclipped steps
        for x, y in zip(ma + mb, ra + rb):
            gr = torch.randn(x.shape, generator=g).to(device) * scale
            x.grad, y.grad = gr.clone(), gr.clone()
        norm = torch.nn.utils.clip_grad_norm_(ra + rb, 10)
        want.step()
        opt.step(max_grad_norm=10)
        torch.testing.assert_close(opt.last_grad_norm, norm, rtol=1e-5, atol=0)
        for x, y in zip(ma + mb, ra + rb):
            torch.testing.assert_close(x, y, rtol=2e-6, atol=2e-7)
            torch.testing.assert_close(x.grad, y.grad, rtol=2e-6, atol=1e-9)
            torch.testing.assert_close(opt.state[x]['exp_avg'], want.state[y]['exp_avg'], rtol=2e-6, atol=1e-8)
        assert float(opt.state[ma[0]]['step']) == it + 1 and float(opt.state[mb[0]]['step']) == it + 1


def test_voxelize_float64_coordinates(device):
    """pcf_hip_voxelize_f64 (round 3): float64 coordinates near voxel faces are hashed from their double values -- bit-exact
    against the oracle and the reference's voxel sequence (tests/golden/make_voxelize_f64_golden.py); the float32 entry point
    on the rounded coordinates gives a different, equally self-consistent, answer."""
    import numpy as np
    import knn_post_dataloader_utils as U
    from conftest import GOLDEN
    from oracle import voxelize_oracle as V
    z = np.load(os.path.join(GOLDEN, 'vox_f64_faces.npz'), allow_pickle=False)
    coord, vs, key, ref_idx = z['coord'], float(z['voxel']), z['key'], z['idx']
    assert coord.dtype == np.float64
    idx = U.voxelize(coord, vs, mode='deterministic').cpu().numpy()
    want, _ = V.voxelize(coord, vs)
    assert np.array_equal(idx, want)
    assert np.array_equal(key[idx], key[ref_idx])
    idx32 = U.voxelize(coord.astype(np.float32), vs, mode='deterministic').cpu().numpy()
    assert np.array_equal(idx32, V.voxelize(coord.astype(np.float32), vs)[0]) and not np.array_equal(idx32, idx)


@pytest.mark.timeout(900)
def test_thread_per_edge_backward_engine_against_default_kernels():
    """pcf_hip_set_aggregate_engine(3): the thread-per-edge backward of small unguided layers (level-0 PointConv of the
    10cm / 5cm models) -- opt-in and unmeasured, its arithmetic checked against the oracle on the CPU
    (tests/test_abi_cpu.py) -- against the default kernels on the GPU, in a child process (first hardware run): the kernel
    is the one that ran, grad_w / grad_add and the whole CSR path are bit-identical (same fmaf chains), the atomics path
    agrees to rounding."""
    recs = _run_child('edge', 800)
    assert len(recs) == 4
    for r in recs:
        assert r['edge_kernel_ran'] == 2 and r['edge_kernel_in_default'] == 0, r
        assert max(r['atomic_err']) < 1e-5 and max(r['csr_err']) < 1e-5, r
        if r['default_is_generic_lds_kernel']:
            assert r['atomic_grad_w_equal'] and r['atomic_grad_add_equal'] and all(r['csr_equal']), r
    assert sum(r['default_is_generic_lds_kernel'] for r in recs) >= 3
