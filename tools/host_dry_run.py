"""Host-side dry run of the GPU-only Python path on a machine WITHOUT a GPU.

Every C-ABI launch (`pcf_cuda._call`) becomes a no-op, the "must be a CUDA tensor" checks are relaxed, streams / device
guards are stubbed, and one training iteration (kNN, CSR, forward, loss, backward, optimizer) of a BASELINE model YAML
runs on CPU tensors.  Kernel outputs are uninitialised memory, so nothing numerical is checked -- what this gives is

  * a smoke test of the Python control flow of code that otherwise only runs on the GPU box (autograd Functions, shape
    bookkeeping, workspace queries, DataParallelStep's eager halves), and
  * the host cost of an iteration without any device work: calls per iteration, seconds per iteration, and a cProfile
    of where the Python time goes (the eager iteration on the GPU is host-bound: DESIGN.md section 10 item 4).

    python tools/host_dry_run.py --model configPCF_10cm --points 2000 --scenes 2 --iters 5 [--profile]

Not part of the product path; never imported by it."""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'ml-pointconvformer_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CALLS = {'n': 0, 'by_name': {}, 'trace': None}


def install_stubs():
    import pcf_cuda
    import pcf_fused
    import knn_post_dataloader_utils as knn_utils

    def relaxed_check(t, name, dtype=None):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f'{name} must be a torch.Tensor')
        if not t.is_contiguous():
            raise RuntimeError(f'{name} must be contiguous')
        if dtype is not None and t.dtype != dtype:
            raise RuntimeError(f'{name} must be {dtype}, got {t.dtype}')

    def no_launch(fn, *args):
        CALLS['n'] += 1
        CALLS['by_name'][fn.__name__] = CALLS['by_name'].get(fn.__name__, 0) + 1
        if CALLS['trace'] is not None:          # entry point + its integer arguments (sizes, flags); pointers and floats left out
            ints = [a for a in args if isinstance(a, int) and not isinstance(a, bool) and abs(a) < (1 << 31)]
            CALLS['trace'].append(fn.__name__.replace('pcf_hip_', '') + ' ' + ' '.join(str(a) for a in ints))

    class NoGuard:
        def __init__(self, dev):
            pass

        def __enter__(self):
            return None

        def __exit__(self, *exc):
            return False

    for mod in (pcf_cuda, pcf_fused):
        if hasattr(mod, '_check_input'):
            mod._check_input = relaxed_check
        if hasattr(mod, '_stream'):
            mod._stream = lambda dev: 0
        if hasattr(mod, '_guard'):
            mod._guard = NoGuard
    pcf_cuda._call = no_launch
    import pcf_optim
    pcf_optim._stream = lambda dev: 0
    pcf_optim._guard = NoGuard
    knn_utils._device = lambda: torch.device('cpu')


def oracle_batch(cfg, scene_points, scenes, seed):
    """A packed batch in the collate layout, levels by the oracle's grid subsampling (no GPU)."""
    from oracle import grid_subsample_oracle as G
    rng = np.random.default_rng(seed)
    per_level = None
    for s in range(scenes):
        side = cfg.grid_size[0] * scene_points ** 0.5 * 1.05
        xy = rng.random((int(scene_points * 1.6), 2), dtype=np.float32) * np.float32(side)
        z = (0.35 * np.sin(1.1 * xy[:, 0]) + 0.25 * np.cos(0.7 * xy[:, 1])).astype(np.float32)
        pts = np.stack([xy[:, 0], xy[:, 1], z], 1).astype(np.float32)
        pts = G.grid_subsampling(pts, None, None, float(cfg.grid_size[0]))[0][:scene_points]
        nrm = np.tile(np.array([[0., 0., 1.]], np.float32), (pts.shape[0], 1))
        levels = [(pts, nrm)]
        for gs in cfg.grid_size[1:]:
            q, f, _ = G.grid_subsampling(levels[-1][0], levels[-1][1], None, float(gs))
            levels.append((q.astype(np.float32), f.astype(np.float32)) if q.shape[0] > 16 else levels[-1])
        per_level = [[lv] for lv in levels] if per_level is None else [a + [lv] for a, lv in zip(per_level, levels)]
    pointclouds = [torch.from_numpy(np.concatenate([p for p, _ in lv]))[None] for lv in per_level]
    norms = [torch.from_numpy(np.concatenate([n for _, n in lv]))[None] for lv in per_level]
    stored = [[int(p.shape[0]) for p, _ in lv] for lv in per_level]
    n0 = sum(stored[0])
    features = torch.randn(1, n0, 3)
    target = torch.randint(0, cfg.num_classes, (n0,))
    return features, pointclouds, target, norms, stored


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='configPCF_10cm')
    ap.add_argument('--points', type=int, default=2000)
    ap.add_argument('--scenes', type=int, default=2)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--profile', action='store_true')
    ap.add_argument('--top', type=int, default=35)
    ap.add_argument('--no-opt', action='store_true', help='forward + backward only (the CPU optimizer is not what the GPU path runs)')
    ap.add_argument('--sort', default='tottime')
    ap.add_argument('--trace', default=None, help='write the C-ABI calls of ONE iteration (entry point + integer arguments) to this file')
    ap.add_argument('--dp', action='store_true', help="the iteration as pcf_train.DataParallelStep's eager halves (one process)")
    args = ap.parse_args()
    torch.set_num_threads(1)
    install_stubs()
    import pcf_model
    import pcf_train
    cfg = pcf_train.baseline_config(args.model)
    torch.manual_seed(1)
    net = pcf_model.PointConvFormer_Segmentation(cfg).train()
    opt = pcf_train.make_optimizer(cfg, net)
    crit = torch.nn.CrossEntropyLoss(ignore_index=cfg.ignore_label, label_smoothing=cfg.label_smoothing)
    batch = oracle_batch(cfg, args.points, args.scenes, seed=5)
    print(f'{args.model}: levels {batch[4]}', flush=True)

    if args.dp:
        import pcf_dist
        dp = pcf_train.DataParallelStep(net, opt, crit, cfg, pcf_dist.GradBucket(list(net.parameters()), list(net.buffers())))

    def step():
        if args.dp:
            return dp(batch)
        if args.no_opt:
            loss = pcf_train.forward_backward(net, crit, cfg, batch)
            opt.zero_grad(set_to_none=True)
            return loss
        return pcf_train.training_iteration(net, opt, crit, cfg, batch)

    step()                                           # first call: lazy module state
    if args.trace:
        CALLS['trace'] = []
        step()
        with open(args.trace, 'w') as f:
            f.write('\n'.join(CALLS['trace']) + '\n')
        print(f'{len(CALLS["trace"])} calls written to {args.trace}', flush=True)
        CALLS['trace'] = None
    CALLS['n'], CALLS['by_name'] = 0, {}
    prof = cProfile.Profile() if args.profile else None
    t0 = time.perf_counter()
    if prof:
        prof.enable()
    for _ in range(args.iters):
        step()
    if prof:
        prof.disable()
    dt = (time.perf_counter() - t0) / args.iters
    print(f'host time per iteration (no device work, 1 thread): {dt * 1e3:.2f} ms; C-ABI calls per iteration: '
          f'{CALLS["n"] / args.iters:.0f}', flush=True)
    top = sorted(CALLS['by_name'].items(), key=lambda kv: -kv[1])[:12]
    print('most frequent entry points per iteration:', [(k.replace('pcf_hip_', ''), v // args.iters) for k, v in top])
    if prof:
        st = pstats.Stats(prof)
        st.sort_stats(args.sort).print_stats(args.top)


if __name__ == '__main__':
    main()
