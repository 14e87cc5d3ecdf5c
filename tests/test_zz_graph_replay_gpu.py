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
