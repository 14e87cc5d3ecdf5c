"""CPU: the GPU-only Python path of a training iteration (kNN, CSR, every layer's autograd Functions, loss, backward,
optimizer; as plain iteration and as DataParallelStep's eager halves) executes without a GPU when every C-ABI launch is a
no-op (tools/host_dry_run.py, in a child process because it patches the package's launch hooks).  Nothing numerical is
checked -- kernel outputs are uninitialised -- only that the control flow of all four BASELINE model YAMLs holds together
and issues the expected number of entry-point calls."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.timeout(600)
@pytest.mark.parametrize('model,extra', [('configPCF_10cm_lite', []), ('configPCF_10cm', []), ('configPCF_5cm', []),
                                         ('configPCF_2cm_PTF2', []), ('configPCF_10cm_lite', ['--dp'])])
def test_training_iteration_control_flow_without_a_gpu(model, extra):
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'host_dry_run.py'), '--model', model, '--points', '1000', '--scenes', '2',
           '--iters', '1'] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=500)
    assert res.returncode == 0, res.stderr[-2000:]
    m = re.search(r'C-ABI calls per iteration: (\d+)', res.stdout)
    assert m and int(m.group(1)) > 400, res.stdout


@pytest.mark.timeout(600)
@pytest.mark.parametrize('model', ['configPCF_10cm_lite', 'configPCF_10cm', 'configPCF_5cm', 'configPCF_2cm_PTF2'])
def test_launch_sequence_of_an_iteration_is_pinned(model, tmp_path):
    """The ordered list of C-ABI calls of one training iteration (entry point + integer arguments: sizes, widths, flags,
    workspace bytes) for a fixed oracle-made batch, against tests/golden/launch_trace_<model>.txt.  A host-side refactor
    that changes which kernels run, in which order or with which sizes shows up here without a GPU; regenerate with
    `python tools/host_dry_run.py --model <model> --points 1000 --scenes 2 --iters 1 --trace tests/golden/launch_trace_<model>.txt`
    when the change is intended."""
    out = tmp_path / 'trace.txt'
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'host_dry_run.py'), '--model', model, '--points', '1000', '--scenes', '2',
           '--iters', '1', '--trace', str(out)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=500)
    assert res.returncode == 0, res.stderr[-2000:]
    got = out.read_text().splitlines()
    want = open(os.path.join(ROOT, 'tests', 'golden', f'launch_trace_{model}.txt')).read().splitlines()
    assert len(got) == len(want), (len(got), len(want))
    for i, (a, b) in enumerate(zip(got, want)):
        assert a == b, f'call {i}: {a!r} != {b!r}'
