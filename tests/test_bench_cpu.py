"""CPU: the host-side pieces of bench.py that the GPU-side measurement relies on (no GPU work here)."""
import argparse
import os

import torch

import bench


def test_profile_counters_name_their_source():
    """roofline.mfma_busy_frac / traffic are read from the committed rocprofv3 --pmc summaries, with the file named."""
    shape = {'N': 80000, 'K': 16, 'Ci': 16, 'Cm': 16, 'H': 8}
    got = bench.profile_counters('pcf_hip_pcf_chain_backward', shape)
    assert 0.2 < got['mfma_busy_frac'] < 0.6 and got['mfma_busy_source'].startswith('profiles/r')
    assert got['traffic'] > 5e8 and got['traffic_source'].startswith('profiles/r')
    agg = bench.profile_counters('pcf_hip_pcf_backward', shape)
    assert 1.0 < agg['traffic'] / (6272 * 80000) < 1.2          # 1.08x the algorithmic bytes of the aggregate backward
    other = bench.profile_counters('pcf_hip_pcf_backward', dict(shape, N=4096))
    assert other['traffic'] is None and other['mfma_busy_frac'] is not None          # traffic only at the profiled shape
    assert bench.profile_counters('no_such_entry_point', shape) == {'mfma_busy_frac': None, 'mfma_busy_source': None,
                                                                   'traffic': None, 'traffic_source': None}


def test_parity_record_units():
    """Errors are in units of each oracle tensor's largest entry; analytically zero parameter gradients are measured
    against 1e-4 of the largest parameter gradient; the feature gradient is also counted row by row."""
    g = torch.Generator().manual_seed(0)
    ref = {'output': torch.randn(1, 50, 8, generator=g), 'feature_grad': torch.randn(1, 50, 8, generator=g),
           'grad:w': torch.randn(8, 8, generator=g) * 100, 'grad:b': torch.randn(8, generator=g) * 1e-9}
    got = {k: v.clone() for k, v in ref.items()}
    got['output'] = got['output'] * (1 + 1e-4)
    got['grad:b'] = torch.zeros(8)                         # exact zero on the fused path, rounding noise in the oracle
    got['feature_grad'][0, 7] += 1.0                       # one row off (a flipped ReLU mask)
    rec = bench.parity_record(got, ref)
    assert rec['worst_tensor'] == 'output' and 5e-5 < rec['parity_max_rel_err'] < 2e-4
    assert rec['param_grads_max'] < 1e-4 and rec['param_grad_tensors'] == 2
    assert rec['feature_grad_rows_over_tolerance'] == 1 and rec['feature_grad_rows'] == 50


def test_self_launch_is_a_no_op_for_one_gpu_and_inside_torchrun(monkeypatch):
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('RANK', raising=False)
    assert bench.self_launch(argparse.Namespace(gpus=1)) is None
    monkeypatch.setenv('WORLD_SIZE', '8')
    assert bench.self_launch(argparse.Namespace(gpus=8)) is None          # the launcher's children must not launch again


def test_train_child_failure_is_reported_not_raised():
    """Without a GPU the child exits with an error: the parent gets a record, the headline measurement would go on."""
    rec = bench.train_in_child(argparse.Namespace(no_cpu_baseline=True))
    assert 'error' in rec and 'no GPU visible' in rec.get('stderr_tail', '')


def test_cpu_baseline_train_runs_the_oracle_model():
    import pcf_model
    import pcf_train
    cfg = pcf_train.baseline_config('configPCF_10cm_lite')
    torch.manual_seed(1)
    net = pcf_model.PointConvFormer_Segmentation(cfg).train()
    rec = bench.cpu_baseline_train(cfg, net, 1200)
    assert rec['kind'] == 'port' and rec['value'] > 0 and rec['unit'] == 'level-0 points/s' and 'oracle/pcf_oracle.py' in rec['sample']


def test_layer_workload_glue_code_runs_end_to_end_without_a_gpu():
    """tools/bench_dry_run.py: bench.py's main() for the headline workload with no-op launches and dummy stream / graph /
    event objects -- every line of the timing loops, HIP-event bookkeeping, roofline blocks, profile evidence, parity block
    and JSON assembly executes and the output parses as strict JSON with the contract's keys."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bench_dry_run.py'), '--points', '1500', '--steps', '2',
                          '--warmup', '1', '--no-train'], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1], parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'parity', 'parity_max_rel_err'):
        assert key in line, key
    assert line['config']['workload'].startswith('PCFLayer(64->64') and line['dtype'] == 'f32' and line['n_gpus'] == 1
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(line['roofline'])
    assert {'value', 'unit', 'cores', 'kind', 'sample'} <= set(line['cpu_baseline'])


def test_train_workload_glue_code_runs_end_to_end_without_a_gpu():
    """bench.py --workload train through tools/bench_dry_run.py: model build, eager iterations, GraphedTrainingStep (state
    snapshot, warm-up, capture, replay -- with dummy graph objects), timing, the oracle's model iteration as CPU baseline,
    JSON assembly; strict JSON (a NaN loss becomes null)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bench_dry_run.py'), '--workload', 'train', '--model',
                          'configPCF_2cm_PTF2', '--points', '1000', '--scenes', '2', '--steps', '2', '--warmup', '1'],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1], parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert line['hip_graph'] is True and line['step_path'].startswith('HIP-graph replay') and line['eager_ms_per_step'] > 0
    assert line['cpu_baseline']['kind'] == 'port' and line['config']['sync_bn'] is False


def test_graph_replay_child_control_flow_without_a_gpu():
    """tests/graph_replay_child.py (the body of the GPU replay test) through tools/dry_env.py: all four model YAMLs, eager
    iterations and GraphedTrainingStep calls with stubbed launches and dummy graphs.  In the dry environment a "capture"
    executes its body and a "replay" does nothing, so after six calls the graph path has taken exactly three real steps
    (the eager first step + two capture bodies): the warm-up iterations of both captures were rolled back."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'dry_env.py'), os.path.join(ROOT, 'tests', 'graph_replay_child.py'),
                          '--cases', 'small'], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    recs = [json.loads(l) for l in res.stdout.splitlines() if l.startswith('{')]
    cases = [r for r in recs if 'case' in r]
    assert recs[-1] == {'done': True} and len(cases) == 4
    for r in cases:
        assert r['optimizer_steps'] == [6.0, 3.0] and r['num_batches_tracked'] == [[6, 6], [3, 3]], r
