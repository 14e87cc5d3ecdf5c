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
