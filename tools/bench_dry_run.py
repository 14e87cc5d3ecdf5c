"""bench.py's layer workload (the headline line) executed top to bottom on a machine WITHOUT a GPU: launches are no-ops
(tools/host_dry_run.py's stubs), torch.cuda's stream / graph / event / synchronise calls are dummies, the process group is
a single CPU rank.  The numbers mean nothing; the point is that every line of the glue code runs -- timing loops, HIP-event
bookkeeping, roofline blocks, profile evidence, the parity block against the oracle, JSON assembly -- before the driver
runs it for real.  Prints the JSON line bench.py would print.

    python tools/bench_dry_run.py --points 2000 --steps 3 --warmup 1 --no-train"""
import contextlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'ml-pointconvformer_amd'), os.path.join(ROOT, 'tools')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import host_dry_run  # noqa: E402


class FakeEvent:
    def __init__(self, *a, **k):
        pass

    def record(self, *a):
        pass

    def elapsed_time(self, other):
        return 0.05


class FakeStream:
    cuda_stream = 0

    def wait_stream(self, other):
        pass


class FakeGraph:
    def __init__(self, *a, **k):
        pass

    def replay(self):
        pass

    def pool(self):
        return None


def main():
    host_dry_run.install_stubs()
    import pcf_cuda
    import pcf_dist

    def no_launch(fn, *args):          # as host_dry_run's, plus the HIP-event timeline bench.py reads
        tl, only = pcf_cuda._timeline, pcf_cuda._timeline_only
        if tl is not None and (only is None or fn.__name__ in only):
            tl.append((fn.__name__, FakeEvent(), FakeEvent()))
    pcf_cuda._call = no_launch
    # the oracle (CPU baseline, parity) indexes with the neighbour table: give it valid indices instead of uninitialised memory
    pcf_cuda.knn_packed = lambda ref, query, ro, qo, K, method='auto': torch.randint(0, ref.shape[0], (query.shape[0], K))
    cpu = torch.device('cpu')
    torch.cuda.is_available = lambda: True
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.Stream = FakeStream
    torch.cuda.current_stream = lambda *a, **k: FakeStream()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.CUDAGraph = FakeGraph
    torch.cuda.graph = lambda g, **k: contextlib.nullcontext()
    torch.cuda.Event = FakeEvent
    torch.cuda.is_current_stream_capturing = lambda: False
    if 'WORLD_SIZE' in os.environ:          # under torch.distributed.run: real ranks and real collectives, over gloo
        real_setup = pcf_dist.setup
        torch.cuda.set_device = lambda *a, **k: None

        def setup(backend=None):
            rank, world, local_rank = pcf_dist.env_rank()
            if not torch.distributed.is_initialized():
                torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
            return rank, world, local_rank, cpu
        pcf_dist.setup = setup
    else:
        pcf_dist.setup = lambda backend=None: (0, 1, 0, cpu)
        pcf_dist.fence = lambda dev=None: None
    if 'train' in sys.argv:
        # scenes and their levels come from GPU kernels (voxelisation, grid subsampling) whose outputs drive host code:
        # use the oracle-made batch of host_dry_run instead
        import pcf_train
        made = {}

        def synthetic_scene(n_points, grid_sizes, seed, device, n_features=3, n_classes=20):
            return {'n': n_points, 'seed': seed}

        def pack_batch(scenes, grid_sizes):
            cfg = pcf_train.baseline_config(made.get('model', 'configPCF_10cm'))
            return host_dry_run.oracle_batch(cfg, scenes[0]['n'], len(scenes), seed=scenes[0]['seed'])
        for i, a in enumerate(sys.argv):
            if a == '--model':
                made['model'] = sys.argv[i + 1]
        pcf_train.synthetic_scene, pcf_train.pack_batch = synthetic_scene, pack_batch
    import bench
    bench.main()


if __name__ == '__main__':
    main()
