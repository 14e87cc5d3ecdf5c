"""Run any script of this repository on a machine WITHOUT a GPU with every device interaction stubbed:

    python tools/dry_env.py <script.py> [args ...]

C-ABI launches are no-ops (tools/host_dry_run.py), torch.cuda's stream / graph / event / synchronise calls are dummies,
synthetic scenes and their levels come from the oracle's grid subsampling instead of the GPU kernels, graph node counts are
faked.  Numbers mean nothing; the Python control flow is what gets exercised."""
import contextlib
import os
import runpy
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


def install():
    host_dry_run.install_stubs()
    import pcf_cuda
    import pcf_train

    def no_launch(fn, *args):          # as host_dry_run's, plus the HIP-event timeline bench.py reads
        tl, only = pcf_cuda._timeline, pcf_cuda._timeline_only
        if tl is not None and (only is None or fn.__name__ in only):
            tl.append((fn.__name__, FakeEvent(), FakeEvent()))
    pcf_cuda._call = no_launch
    # host code indexes with neighbour tables (the oracle in bench.py's CPU baseline): valid indices, not uninitialised memory
    pcf_cuda.knn_packed = lambda ref, query, ro, qo, K, method='auto': torch.randint(0, max(ref.shape[0], 1), (query.shape[0], K))
    torch.cuda.is_available = lambda: True
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda: None
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.Stream = FakeStream
    torch.cuda.current_stream = lambda *a, **k: FakeStream()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.CUDAGraph = FakeGraph
    torch.cuda.graph = lambda g, **k: contextlib.nullcontext()
    torch.cuda.Event = FakeEvent
    torch.cuda.is_current_stream_capturing = lambda: False
    pcf_train.graph_node_counts = lambda g: {'kernel': 1000}
    cfg_of = {}

    def synthetic_scene(n_points, grid_sizes, seed, device, n_features=3, n_classes=20):
        return {'n': n_points, 'seed': seed, 'grid': tuple(grid_sizes)}

    def pack_batch(scenes, grid_sizes):
        key = tuple(grid_sizes)
        if key not in cfg_of:
            cfg_of[key] = next(pcf_train.baseline_config(n) for n in pcf_train.BASELINE_CONFIGS
                               if tuple(pcf_train.BASELINE_CONFIGS[n]['grid_size']) == key)
        return host_dry_run.oracle_batch(cfg_of[key], scenes[0]['n'], len(scenes), seed=scenes[0]['seed'])
    pcf_train.synthetic_scene, pcf_train.pack_batch = synthetic_scene, pack_batch


if __name__ == '__main__':
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    install()
    os.environ['PCF_TEST_DEVICE'] = 'cpu'          # scripts that build their own device read this
    sys.argv = sys.argv[1:]
    runpy.run_path(sys.argv[0], run_name='__main__')
