"""Child process of tests/test_zz_graph_replay_gpu.py: captured-and-replayed training iterations against eager ones.

Runs in its own process so that a GPU memory fault (the failure this test exists for: round 2's captured
configPCF_2cm_PTF2 iteration faulted on replay) ends this process only and is reported by the parent as a failed
assertion with the child's output.  One JSON line per case on stdout.

Per case: two models with identical parameters and two FusedAdamW optimizers; the same rotating pool of two packed
batches; `calls` optimisation steps through pcf_train.training_iteration (eager) and through
pcf_train.GraphedTrainingStep (capture on first sight of a batch, replay afterwards -- the fifth call is the second
replay of the first graph, where the round-2 fault occurred)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'ml-pointconvformer_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def run_case(name, scenes, points, calls, freeze_draws, dev):
    import pcf_layers
    import pcf_model
    import pcf_train
    cfg = pcf_train.baseline_config(name)
    crit = torch.nn.CrossEntropyLoss(ignore_index=cfg.ignore_label, label_smoothing=cfg.label_smoothing).to(dev)
    pool = []
    for b in range(2):
        sc = [pcf_train.synthetic_scene(points, cfg.grid_size, seed=7000 + 10 * b + i, device=dev) for i in range(scenes)]
        pool.append(pcf_train.pack_batch(sc, cfg.grid_size))
    nets, opts = [], []
    for _ in range(2):
        torch.manual_seed(11)
        net = pcf_model.PointConvFormer_Segmentation(cfg).to(dev).train()
        if freeze_draws and cfg.drop_path_rate > 0:
            # both paths must see the same stochastic-depth factors: every block keeps its branch, scaled by 1 / keep
            # (the DropPath code path -- no fused residual tail -- is still the one that runs)
            for m in net.modules():
                if isinstance(m, pcf_layers.DropPath) and m.drop_prob > 0:
                    m.draw = (lambda x, k=1.0 - m.drop_prob: x.new_full((x.shape[0],) + (1,) * (x.dim() - 1), 1.0 / k))
        nets.append(net)
        opts.append(pcf_train.make_optimizer(cfg, net, capturable=True))
    start = torch.cat([p.detach().reshape(-1) for p in nets[0].parameters()]).clone()
    eager = [float(pcf_train.training_iteration(nets[0], opts[0], crit, cfg, pool[i % 2])) for i in range(calls)]
    torch.cuda.synchronize()
    gstep = pcf_train.GraphedTrainingStep(nets[1], opts[1], crit, cfg)
    gstep.keep_graph = True
    graphed = []
    for i in range(calls):
        graphed.append(float(gstep(pool[i % 2])))          # float(): synchronises, so a fault is pinned to its call
        print(json.dumps({'progress': name, 'call': i, 'loss': graphed[-1]}), flush=True)
    kinds = [pcf_train.graph_node_counts(g[0]) for g in gstep.graphs.values()]
    pe = torch.cat([p.detach().reshape(-1) for p in nets[0].parameters()])
    pg = torch.cat([p.detach().reshape(-1) for p in nets[1].parameters()])
    de, dg = (pe - start).double(), (pg - start).double()
    nbt = [[int(b) for n, b in net.named_buffers() if n.endswith('num_batches_tracked')] for net in nets]
    steps = [float(o.state[o.param_groups[0]['params'][0]]['step']) for o in opts]          # FusedAdamW: a view of its device record
    return {'case': name, 'scenes': scenes, 'points': points, 'levels': pool[0][4], 'eager_losses': eager, 'graph_losses': graphed,
            'node_kinds': kinds, 'update_cosine': float((de @ dg) / (de.norm() * dg.norm() + 1e-30)),
            'finite': bool(torch.isfinite(pg).all()), 'num_batches_tracked': [[min(v), max(v)] for v in nbt],
            'optimizer_steps': steps, 'frozen_draws': bool(freeze_draws)}


def run_tables_case(name, scenes, points, rounds, dev):
    """The neighbour tables and the CSR of a captured iteration, kept alive and compared after every replay with tables
    built eagerly from the same batch: the direct check of the round-2 failure (an uncleared cell histogram / CSR counter on
    relaunch gives wrong tables long before it gives a fault).  Two graphs, replayed alternately."""
    import pcf_model
    import pcf_train
    cfg = pcf_train.baseline_config(name)
    crit = torch.nn.CrossEntropyLoss(ignore_index=cfg.ignore_label, label_smoothing=cfg.label_smoothing).to(dev)
    pool = []
    for b in range(2):
        sc = [pcf_train.synthetic_scene(points, cfg.grid_size, seed=9000 + 10 * b + i, device=dev) for i in range(scenes)]
        pool.append(pcf_train.pack_batch(sc, cfg.grid_size))
    torch.manual_seed(13)
    net = pcf_model.PointConvFormer_Segmentation(cfg).to(dev).train()
    opt = pcf_train.make_optimizer(cfg, net, capturable=True)
    for batch in pool:
        pcf_train.training_iteration(net, opt, crit, cfg, batch)
    torch.cuda.synchronize()
    graphs = []
    for batch in pool:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            pcf_train.training_iteration(net, opt, crit, cfg, batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            edges = pcf_train.build_edges(cfg, batch[1], batch[4])
            pcf_train.training_iteration(net, opt, crit, cfg, batch, edges)
        graphs.append((g, edges, batch))
    wrong, compared = [], 0
    for r in range(rounds):
        for gi, (g, edges, batch) in enumerate(graphs):
            g.replay()
            torch.cuda.synchronize()
            es, ef, ep, inv = edges
            res, rf, rp, rinv = pcf_train.build_edges(cfg, batch[1], batch[4])
            for tag, got, want in (('self', es, res), ('forward', ef, rf), ('propagate', ep, rp)):
                for l, (a, b) in enumerate(zip(got, want)):
                    compared += 1
                    if not torch.equal(a, b):
                        wrong.append(f'round {r} graph {gi} {tag}[{l}]: {int((a != b).any(-1).sum())} rows')
            for rel in range(3):
                for part in range(3):
                    for l, (a, b) in enumerate(zip(inv[rel][part], rinv[rel][part])):
                        compared += 1
                        if not torch.equal(a, b):
                            wrong.append(f'round {r} graph {gi} csr{rel}.{part}[{l}]')
            print(json.dumps({'progress': name + ' tables', 'round': r, 'graph': gi, 'wrong_so_far': len(wrong)}), flush=True)
    return {'tables_case': name, 'scenes': scenes, 'points': points, 'levels': pool[0][4], 'compared': compared, 'wrong': wrong[:20],
            'n_wrong': len(wrong)}


def run_edge_engine_cases(dev):
    """The opt-in thread-per-edge backward (pcf_hip_set_aggregate_engine(3), csrc/aggregate.hip:agg_bwd_edge_kernel; its
    body is checked against the oracle on the CPU) against the default kernels on the GPU: pconv_backward (float atomics)
    and pconv_linear_opt_backward (contribution rows + CSR reduce: deterministic, so bit-identical), at the level-0
    PointConv shape of the 10cm / 5cm models and three more.  First hardware run of this kernel: in the child process."""
    import pcf_cuda
    out = []
    for (N, Nout, K, Ci, Ca, Cm, Co) in ((20000, 20000, 16, 6, 12, 16, 64), (5000, 1300, 16, 16, 16, 4, 32), (3000, 3000, 8, 3, 0, 16, 32),
                                         (777, 500, 16, 7, 5, 16, 24)):
        g = torch.Generator().manual_seed(N + Ci)
        x = torch.randn(1, N, Ci, generator=g).to(dev)
        idx = torch.randint(0, N, (1, Nout, K), generator=g)
        idx[0, ::7, 3] = -1                                                   # out-of-range entries: skipped on every path
        idx = idx.to(dev)
        w = torch.randn(1, Nout, K, Cm, generator=g).to(dev)
        add = torch.randn(1, Nout, K, Ca, generator=g).to(dev)
        gout = torch.randn(1, Nout, (Ci + Ca) * Cm, generator=g).to(dev)
        lin_w = (torch.randn(Co, (Ci + Ca) * Cm, generator=g) / 16).to(dev)
        lin_b = torch.zeros(Co, device=dev)
        gout_lin = torch.randn(1, Nout, Co, generator=g).to(dev)
        inv = pcf_cuda.compute_knn_inverse(idx, N)
        res = {}
        for engine in ('default', 'edge'):
            pcf_cuda.set_aggregate_engine(engine)
            pcf_cuda.launch_log(True)
            a = pcf_cuda.pconv_backward(gout, x, idx, w, add)
            _, pconv_out = pcf_cuda.pconv_linear_forward(x, idx, w, add, lin_w, lin_b)
            b = pcf_cuda.pconv_linear_opt_backward(gout_lin, x, inv[0], inv[1], inv[2], idx, w, add, lin_w, pconv_out)
            torch.cuda.synchronize()
            res[engine] = ([t.cpu() for t in a], [t.cpu() for t in b], pcf_cuda.read_launch_log())
            pcf_cuda.launch_log(False)
        pcf_cuda.set_aggregate_engine('default')
        (a0, b0, log0), (a1, b1, log1) = res['default'], res['edge']
        scale = lambda t: float(t.abs().max()) + 1e-30
        err = lambda p, q: (float((p - q).abs().max()) / scale(p)) if p.numel() else 0.0
        rec = {'edge_case': [N, Nout, K, Ci, Ca, Cm], 'edge_kernel_ran': sum('agg_bwd_edge_kernel' in l for l in log1),
               'edge_kernel_in_default': sum('agg_bwd_edge_kernel' in l for l in log0),
               # the generic LDS kernel runs the same fmaf chains (bit-identical grad_w / grad_add / contribution rows);
               # the matrix-core kernels of the default dispatch sum in another order
               'default_is_generic_lds_kernel': all('mfma' not in l for l in log0 if 'bwd' in l and 'agg' in l or 'pconv_bwd' in l),
               'atomic_err': [err(p, q) for p, q in zip(a0, a1)], 'csr_err': [err(p, q) for p, q in zip(b0, b1)],
               'atomic_grad_w_equal': bool(torch.equal(a0[1], a1[1])), 'atomic_grad_add_equal': bool(torch.equal(a0[2], a1[2])),
               'csr_equal': [bool(torch.equal(p, q)) for p, q in zip(b0, b1)]}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', default='small')
    args = ap.parse_args()
    dev = torch.device(os.environ.get('PCF_TEST_DEVICE', 'cuda:0'))          # tools/dry_env.py runs this on the CPU
    if args.cases == 'edge':
        run_edge_engine_cases(dev)
        print(json.dumps({'done': True}), flush=True)
        return
    small = [('configPCF_10cm_lite', 2, 3000), ('configPCF_10cm', 2, 3000), ('configPCF_5cm', 1, 6000),
             ('configPCF_2cm_PTF2', 2, 6000)]
    cases = [(n, s, p, 6, True) for n, s, p in small]
    if args.cases == 'all':
        cases.append(('configPCF_2cm_PTF2', 2, 6000, 6, False))          # real stochastic-depth draws under capture (philox)
        cases.append(('configPCF_2cm_PTF2', 2, 120000, 5, True))          # the configuration's own size (MAX_POINTS_NUM x BATCH_SIZE)
    # first the sharpest and cheapest check: tables of replayed iterations against eager ones (no fault even if they differ)
    for name, sc, pts in (('configPCF_2cm_PTF2', 2, 6000), ('configPCF_10cm', 2, 6000)):
        print(json.dumps(run_tables_case(name, sc, pts, 3, dev)), flush=True)
        torch.cuda.empty_cache()
    for c in cases:
        print(json.dumps(run_case(*c, dev)), flush=True)
        torch.cuda.empty_cache()
    print(json.dumps({'done': True}), flush=True)


if __name__ == '__main__':
    main()
